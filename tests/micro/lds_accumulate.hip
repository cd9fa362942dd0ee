// lds_accumulate.hip -- what does it cost to ADD a wavefront's 64 partial sums into per-splat accumulators in LDS?
//
// The quadrant-queue backward (round 3, profiles/r3_backward_q.md) merged the four quadrants' partial sums of a splat with
// ds_add_f32 and measured ~360 cycles per wave-wide instruction.  Is that the instruction (float atomics serialised per
// lane), the address pattern (a row of 9 floats per splat: bank = (9 slot + col) mod 32), or same-address collisions between
// the quadrants?  Every wave issues `iters` x 12 operations of one form and stamps s_memtime around them:
//   form 0  ds_add_f32, lane-linear addresses (conflict-free)
//   form 1  ds_add_f32, acc[slot(q, sg, r) * 9 + cj] with 16 distinct random slots per quadrant, quadrants disjoint
//   form 2  ds_add_f32, the same with all four quadrants on the SAME 16 slots (4 lanes per address)
//   form 3  ds_write_b32, pattern of form 1           (the cost of the addressing alone)
//   form 4  ds_read_b32 + v_add + ds_write_b32, pattern of form 1 (non-atomic read-modify-write)
//   form 5  ds_add_f32, rows of 12 floats instead of 9 (bank = (12 slot + col) mod 32)
//   form 6  ds_add_rtn_f32 (returning), pattern of form 1
// Build + run: hipcc --offload-arch=gfx950 -O2 lds_accumulate.hip -o lds_accumulate.bin && ./lds_accumulate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

template <int FORM>
__global__ __launch_bounds__(256) void kern(unsigned long long* __restrict__ out, int iters, float* sink) {
  extern __shared__ float lds[];
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  float* acc = lds + wv * 2048;                                   // 8 KB per wavefront
  for (int i = l; i < 2048; i += 64) acc[i] = 0.f;
  const int q = l >> 4, sg = (l >> 2) & 3, cj = l & 3;
  constexpr int ROW = FORM == 5 ? 12 : 9;
  int addr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // 16 pseudo-random slots of 128 per quadrant (the ring slots of the splats a quadrant pops in one step)
    int slot = ((sg * 4 + r) * 37 + 11 * q + 5) & 127;
    if (FORM == 2) slot = ((sg * 4 + r) * 37 + 5) & 127;          // all quadrants on the same slots
    else slot = (slot & ~3) | ((slot + q) & 3) | 0;               // (keeps quadrants mostly apart)
    if (FORM != 2) slot = (slot & 31) + 32 * q;                   // disjoint slot ranges per quadrant
    addr[r] = FORM == 0 ? (r * 64 + l) : slot * ROW + cj;
  }
  float v = 1.0f + l * 1e-3f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* p = acc + addr[r] + (FORM == 0 ? c * 256 : 4 * c);
        if (FORM == 3) { asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(size_t)(p - lds) * 4u), "v"(v) : "memory"); }
        else if (FORM == 4) { const float o = *(volatile float*)p; *(volatile float*)p = o + v; }
        else if (FORM == 6) { v += __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * 1e-9f; }
        else { asm volatile("ds_add_f32 %0, %1" :: "v"((unsigned)(size_t)(p - lds) * 4u), "v"(v) : "memory"); }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc[l] == 123.456f) sink[0] = acc[l] + v;
  if (l == 0) out[blockIdx.x * 4 + wv] = t1 - t0;
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, iters = 32;
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, sizeof(unsigned long long) * cus * 8 * 4);
  hipMalloc(&d_sink, 4);
  const char* names[7] = {"ds_add_f32 linear", "ds_add_f32 rows of 9", "ds_add_f32 4 lanes/address", "ds_write_b32 rows of 9",
                          "read+add+write rows of 9", "ds_add_f32 rows of 12", "ds_add_rtn_f32 rows of 9"};
  printf("cycles per wave-wide LDS operation: per-wave cycles / (12 x iters), then the SIMD's share (/ waves per SIMD)\n");
  for (int form = 0; form < 7; ++form) {
    printf("  %-28s", names[form]);
    for (int W : {1, 2, 3, 4}) {
      const size_t lds = 32768;                                   // 4 wavefronts x 8 KB
      const int blocks = cus * W;
      for (int rep = 0; rep < 2; ++rep) {
#define LAUNCH(F) hipLaunchKernelGGL(kern<F>, dim3(blocks), dim3(256), lds, 0, d_out, iters, d_sink)
        switch (form) { case 0: LAUNCH(0); break; case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break;
                        case 4: LAUNCH(4); break; case 5: LAUNCH(5); break; default: LAUNCH(6); }
        hipDeviceSynchronize();
      }
      std::vector<unsigned long long> h(blocks * 4);
      hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      const double med = (double)h[h.size() / 2] / (12.0 * iters);
      printf("  W=%d: %6.1f (%5.1f)", W, med, med / W);
    }
    printf("\n");
  }
  return 0;
}
