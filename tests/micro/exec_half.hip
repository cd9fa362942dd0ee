// exec_half.hip -- does gfx950 skip a 32-lane half of a wave64 vector instruction whose EXEC bits are all zero?
//
// If it does, a splat that reaches only the upper or only the lower four rows of an 8x8 tile (about half of the
// instances at sigma ~ 1 px) could run its sweep on half a wavefront for half the issue time.
// Every wave runs 64 x 16 v_fma_f32 / v_exp_f32 with EXEC = all lanes | lanes 0-31 | lanes 32-63 | lanes 0-15 | even lanes
// and stamps s_memtime around the loop; W blocks per CU of 256 threads = W waves per SIMD.
// Build + run: hipcc --offload-arch=gfx950 -O2 exec_half.hip -o exec_half.bin && ./exec_half.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define FMA1(acc) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))
#define EXP1(acc) asm volatile("v_exp_f32 %0, %0" : "+v"(acc))

template <int OP>
__global__ __launch_bounds__(256) void kern(unsigned long long* __restrict__ out, int iters, float x, float y, float* sink,
                                            unsigned long long mask) {
  extern __shared__ float pad[];
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = x * (float)(i + 1) + (float)threadIdx.x * 1e-6f;
  const int lane = threadIdx.x & 63;
  const bool on = (mask >> lane) & 1ull;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (on) {                                   // the whole loop runs under the partial EXEC mask
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { if (OP == 0) FMA1(a[i]); else EXP1(a[i]); }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 123.456f) sink[0] = s + pad[0];
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, iters = 64;
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, sizeof(unsigned long long) * cus * 8 * 4);
  hipMalloc(&d_sink, 4);
  const unsigned long long masks[5] = {~0ull, 0xFFFFFFFFull, 0xFFFFFFFF00000000ull, 0xFFFFull, 0x5555555555555555ull};
  const char* names[5] = {"all 64 lanes", "lanes 0-31", "lanes 32-63", "lanes 0-15", "even lanes"};
  for (int op = 0; op < 2; ++op) {
    printf("%s: cycles per 16 instructions on the SIMD (per-wave cycles / waves per SIMD)\n", op ? "v_exp_f32" : "v_fma_f32");
    for (int m = 0; m < 5; ++m) {
      printf("  %-14s", names[m]);
      for (int W : {1, 2, 4, 8}) {
        const size_t lds = (size_t)(160 * 1024 / W) - 1024;       // W blocks per CU
        const int blocks = cus * W;
        for (int rep = 0; rep < 2; ++rep) {
          if (op == 0) hipLaunchKernelGGL(kern<0>, dim3(blocks), dim3(256), W == 1 ? 64 * 1024 : lds > 65536 ? 65536 : lds, 0, d_out, iters, 1.0001f, 0.5f, d_sink, masks[m]);
          else hipLaunchKernelGGL(kern<1>, dim3(blocks), dim3(256), W == 1 ? 64 * 1024 : lds > 65536 ? 65536 : lds, 0, d_out, iters, 1.0001f, 0.5f, d_sink, masks[m]);
          hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("  W%d %7.1f", W, (double)h[h.size() / 2] / iters / W);
      }
      printf("\n");
    }
  }
  return 0;
}
