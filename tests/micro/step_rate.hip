// step_rate.hip -- how fast can one SIMD run the inner step of the composites when nothing else is in the way?
//
// The step functions of vtgs_composite_q.hip (quadrant form: 24 exponent MFMAs + 16 exp/threshold/T updates + 16 colour
// MFMAs per 16 splats) and of vtgs_composite.hip (lane = pixel form: payload from LDS, colour on the VALU) are run in a
// loop on register-resident operands, with exactly W wavefronts per SIMD (W blocks of 256 threads per CU, enforced by the
// dynamic-LDS request).  Prints s_memtime cycles per step for the median wave and the SIMD-level cycles per step
// (= per-wave cycles / W): the floor the real kernels can approach once memory, LDS queues and control flow are hidden.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../vtgaussian-slam_amd/csrc -I../../include step_rate.hip -o step_rate.bin
#include "../../vtgaussian-slam_amd/csrc/vtgs_composite_q.hip"
#include "../../vtgaussian-slam_amd/csrc/vtgs_composite.hip"
#include <stdio.h>
#include <algorithm>
#include <vector>

using namespace vtgs;

template <int MODE>   // 0: quadrant step (clamp-free, optimistic), 1: clamped, 2: exact-first
__global__ __launch_bounds__(256, 5) void bench_q(unsigned long long* out, int iters, float seed, float* sink) {
  extern __shared__ float dyn[];
  __shared__ float4 ka[4][4];
  __shared__ float2 kb[4][4];
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float Phi[6], PT[4], PT2[4] = {0.f, 0.f, 0.f, 0.f};
  const float X = (float)(l & 3) - 1.5f, Y = (float)((l >> 2) & 3) - 1.5f;
  Phi[0] = 1.f; Phi[1] = X; Phi[2] = Y; Phi[3] = X * X; Phi[4] = X * Y; Phi[5] = Y * Y;
  if (l < 4) { ka[wv][l] = make_float4(-3.f - 0.01f * l + seed, 0.1f, -0.2f, -0.4f); kb[wv][l] = make_float2(0.05f, -0.45f); }
  for (int j = 0; j < 4; ++j) PT[j] = 0.1f * j + 0.01f * l;
  f32x4 C = {0.f, 0.f, 0.f, 0.f}, C2 = {0.f, 0.f, 0.f, 0.f};
  float T = 1.f;
  bool done = false, exact = false;
  if (threadIdx.x == 0) dyn[0] = seed;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const int slot = (it + l) & 3;
    if (MODE == 0) q_forward_step<false, false, false>(T, done, exact, C, C2, ka[wv], kb[wv], slot, Phi, PT, PT2);
    if (MODE == 1) q_forward_step<false, true, false>(T, done, exact, C, C2, ka[wv], kb[wv], slot, Phi, PT, PT2);
    if (MODE == 2) q_forward_step<false, true, true>(T, done, exact, C, C2, ka[wv], kb[wv], slot, Phi, PT, PT2);
    T = T * 0.5f + 0.5f;                          // keep the pixel alive (one extra op per step)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  if (C[0] + C[1] + C[2] + C[3] + T == 12345.f) sink[0] = T;
  if (l == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// lane = pixel form of vtgs_composite.hip: 24 exponent MFMAs, payload by LDS broadcast reads, colour on the VALU
__global__ __launch_bounds__(256, 3) void bench_px(unsigned long long* out, int iters, float seed, float* sink) {
  extern __shared__ float dyn[];
  __shared__ float4 pay[4][64];
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float K[6], Phi[6];
  const float X = (float)(l & 7) - 3.5f, Y = (float)(l >> 3) - 3.5f;
  Phi[0] = 1.f; Phi[1] = X; Phi[2] = Y; Phi[3] = X * X; Phi[4] = X * Y; Phi[5] = Y * Y;
  K[0] = -3.f - 0.01f * l + seed; K[1] = 0.1f; K[2] = -0.2f; K[3] = -0.4f; K[4] = 0.05f; K[5] = -0.45f;
  pay[wv][l] = make_float4(0.1f * l, 0.2f, 0.3f, 0.4f);
  float C[4] = {0.f, 0.f, 0.f, 0.f};
  float T = 1.f;
  bool done = false, exact = false;
  if (threadIdx.x == 0) dyn[0] = seed;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    px_forward_batch<0, false, false, false>(T, done, exact, C, K, Phi, pay[wv], nullptr);
    T = T * 0.5f + 0.5f;
    K[0] += 1e-6f;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  if (C[0] + C[1] + C[2] + C[3] + T == 12345.f) sink[0] = T;
  if (l == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static void run_px(int W, int iters, unsigned long long* d_out, float* d_sink) {
  const int blocks = 256 * W;
  const size_t lds = (size_t)(160 * 1024 / W) - 8192;
  (void)hipFuncSetAttribute((const void*)bench_px, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(bench_px, dim3(blocks), dim3(256), lds, 0, d_out, iters, 0.f, d_sink);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(3); }
  std::vector<unsigned long long> h((size_t)blocks * 4);
  (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2] / iters;
  printf("lane = pixel batch  waves/SIMD %d : median wave %7.1f cyc/batch  ->  %6.1f cyc/batch on the SIMD\n", W, med, med / W);
}

template <int MODE>
static void run(int W, int iters, unsigned long long* d_out, float* d_sink) {
  const int blocks = 256 * W;
  const size_t lds = (size_t)(160 * 1024 / W) - 2048;          // exactly W blocks fit a CU (W > 5: as many as the registers allow)
  (void)hipFuncSetAttribute((const void*)bench_q<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(bench_q<MODE>, dim3(blocks), dim3(256), lds, 0, d_out, iters, 0.f, d_sink);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(3); }
  std::vector<unsigned long long> h((size_t)blocks * 4);
  (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2] / iters, mx = (double)h.back() / iters;
  printf("mode %d  waves/SIMD %d : median wave %7.1f cyc/step (max %7.1f)  ->  %6.1f cyc/step on the SIMD\n", MODE, W, med, mx, med / W);
}

int main() {
  unsigned long long* d_out; float* d_sink;
  if (hipMalloc(&d_out, 256 * 8 * 4 * 8) != hipSuccess || hipMalloc(&d_sink, 64) != hipSuccess) return 2;
  printf("quadrant step: 24 exponent MFMA + 16 x (exp, [min], cmp, cndmask, mul, sub) + 16 colour MFMA per 16 splats x 64 pixels\n");
  for (int W : {1, 2, 3, 4, 5, 6, 8}) run<0>(W, 4000, d_out, d_sink);
  for (int W : {3, 5, 6, 8}) run<1>(W, 4000, d_out, d_sink);
  for (int W : {3, 5, 6, 8}) run<2>(W, 4000, d_out, d_sink);
  for (int W = 1; W <= 3; ++W) run_px(W, 4000, d_out, d_sink);
  return 0;
}
