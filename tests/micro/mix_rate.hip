// mix_rate.hip -- what does it cost to MIX f32 MFMAs (v_mfma_f32_4x4x1_16b_f32) and plain VALU work in one wavefront's
// instruction stream, as a function of how they are grouped and of the number of wavefronts per SIMD?
// Body = A MFMAs back to back (4 accumulators, or ONE accumulator = a dependent chain with DEP) followed by B independent
// v_fma_f32, repeated; sched_barrier keeps the grouping.  "sum of parts" uses the saturated single-kind rates measured by
// issue_rate.hip (MFMA 6.0, v_fma 1.95 cycles).  W wavefronts per SIMD are enforced through the dynamic-LDS request.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define FMA1(acc) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))

template <int A, int B, bool DEP>
__global__ __launch_bounds__(256, 8) void kern(unsigned long long* out, int iters, float x, float y, float* sink) {
  extern __shared__ float dyn[];
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = x * (float)(i + 1) + (float)threadIdx.x * 1e-6f;
  f32x4 m4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) m4[i] = f32x4{x, y, x, y};
  if (threadIdx.x == 0) dyn[0] = x;
  constexpr int REP = (A + B >= 64) ? 1 : 64 / (A + B);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < REP; ++rep) {
#pragma unroll
      for (int i = 0; i < A; ++i) m4[DEP ? 0 : (i & 3)] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, m4[DEP ? 0 : (i & 3)], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < B; ++i) FMA1(a[i & 15]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += m4[i][0] + m4[i][1] + m4[i][2] + m4[i][3];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = (t1 - t0) / REP;
}

template <int A, int B, bool DEP>
static void run(unsigned long long* d_out, float* d_sink) {
  printf("%2d MFMA%s + %2d v_fma (sum of parts %5.1f):", A, DEP ? " (one chain)" : "", B, A * 6.0 + B * 1.95);
  for (int W : {1, 2, 3, 4, 5, 6, 8}) {
    const int blocks = 256 * W, iters = 3000;
    const size_t lds = (size_t)(160 * 1024 / W) - 1024;
    (void)hipFuncSetAttribute((const void*)kern<A, B, DEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((kern<A, B, DEP>), dim3(blocks), dim3(256), lds, 0, d_out, iters, 1.0001f, 0.9999f, d_sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(3); }
    std::vector<unsigned long long> h((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("  W%d %6.1f", W, (double)h[h.size() / 2] / iters / W);
  }
  printf("   (cycles per body on the SIMD)\n");
}

int main() {
  unsigned long long* d_out; float* d_sink;
  if (hipMalloc(&d_out, 256 * 8 * 4 * 8) != hipSuccess || hipMalloc(&d_sink, 64) != hipSuccess) return 2;
  run<0, 16, false>(d_out, d_sink);
  run<16, 0, false>(d_out, d_sink);
  run<16, 0, true>(d_out, d_sink);
  run<1, 3, false>(d_out, d_sink);
  run<4, 12, false>(d_out, d_sink);
  run<8, 24, false>(d_out, d_sink);
  run<24, 72, false>(d_out, d_sink);
  run<16, 48, true>(d_out, d_sink);
  return 0;
}
