// issue_rate.hip -- measured issue / co-execution rates on gfx950 that the composite kernels' design rests on.
//
// Questions ("verify with a 1/2/3/4-wave v_fma microbenchmark"):
//   * how many waves per SIMD does it take to fill the fp32 lanes with independent / dependent v_fma_f32 ?
//   * what do v_exp_f32, v_cmp+v_cndmask, v_pk_fma_f32 and a broadcast ds_read_b128 cost per wave-instruction ?
//   * do f32 MFMAs (4x4x1 16-block, 16x16x4) run BESIDE vector work of the same wave / of another wave on the SIMD,
//     or do they take the same lanes ?  And the bf16 32x32x16 MFMA ?
// Every wave stamps s_memtime around its loop; the host prints the median wave's cycles per loop body and per
// instruction for each (mode, waves/SIMD).  Blocks of 256 threads put one wave on every SIMD; W blocks per CU give W
// waves per SIMD.  Role-split modes use 512-thread blocks: waves 0-3 take role A, waves 4-7 role B (one of each per SIMD).
// Build + run: hipcc --offload-arch=gfx950 -O2 issue_rate.hip -o issue_rate.bin && ./issue_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define FMA1(acc) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))
#define EXP1(acc) asm volatile("v_exp_f32 %0, %0" : "+v"(acc))
#define PKFMA1(acc) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(px), "v"(py))
#define CMPSEL(acc) asm volatile("v_cmp_ge_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %2, %0, vcc" : "+v"(acc) : "v"(x), "v"(y) : "vcc")

enum Mode {
  M_FMA_INDEP = 0,     // 16 independent v_fma_f32
  M_FMA_DEP,           // 16 v_fma_f32, one dependent chain
  M_FMA_DEP4,          // 16 v_fma_f32 in four interleaved dependent chains
  M_EXP,               // 16 independent v_exp_f32
  M_CMPSEL,            // 8 x (v_cmp_ge_f32 + v_cndmask_b32) = 16 instructions
  M_PKFMA,             // 16 independent v_pk_fma_f32
  M_MFMA4,             // 16 v_mfma_f32_4x4x1_16b_f32 on 4 accumulators
  M_MFMA16,            // 8 v_mfma_f32_16x16x4_f32 on 4 accumulators
  M_MFMABF,            // 4 v_mfma_f32_32x32x16_bf16 on 2 accumulators
  M_MIX_F32,           // one wave: 4 x (1 mfma 4x4x1 + 3 v_fma) = 4 MFMA + 12 v_fma
  M_MIX_BF,            // one wave: 2 x (1 mfma bf16 32x32x16 + 8 v_fma) = 2 MFMA + 16 v_fma
  M_SPLIT_F32,         // waves 0-3: 16 v_fma;  waves 4-7: 16 mfma 4x4x1   (512-thread blocks)
  M_SPLIT_F32_16,      // waves 0-3: 16 v_fma;  waves 4-7: 8 mfma 16x16x4
  M_SPLIT_BF,          // waves 0-3: 16 v_fma;  waves 4-7: 4 mfma bf16 32x32x16
  M_LDS_B128,          // 16 broadcast ds_read_b128 + one wait
  M_COUNT
};
static const char* kNames[M_COUNT] = {
    "16 v_fma independent", "16 v_fma one chain", "16 v_fma four chains", "16 v_exp_f32", "8 x (v_cmp + v_cndmask)",
    "16 v_pk_fma_f32", "16 mfma_f32_4x4x1 (4 acc)", "8 mfma_f32_16x16x4 (4 acc)", "4 mfma_bf16_32x32x16 (2 acc)",
    "same wave: 4 mfma4x4x1 + 12 v_fma", "same wave: 2 mfma_bf16 + 16 v_fma", "split: A=16 v_fma | B=16 mfma4x4x1",
    "split: A=16 v_fma | B=8 mfma16x16x4", "split: A=16 v_fma | B=4 mfma_bf16", "16 ds_read_b128 broadcast"};

template <int MODE>
__global__ __launch_bounds__(512) void kern(unsigned long long* __restrict__ out, int iters, float x, float y, float* sink) {
  __shared__ float4 lds[64];
  if (threadIdx.x < 64) lds[threadIdx.x] = make_float4(x, y, x, y);
  __syncthreads();
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = x * (float)(i + 1) + (float)threadIdx.x * 1e-6f;
  f32x4 m4[4];
  f32x16 m16[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) m4[i] = f32x4{x, y, x, y};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) m16[i][j] = x + (float)j;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 p[16], px = {x, y}, py = {y, x};
#pragma unroll
  for (int i = 0; i < 16; ++i) p[i] = f32x2{x * (float)i, y};
  bf16x8 ba, bb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ba[i] = (short)(0x3f80 + i); bb[i] = (short)(0x3f00 + threadIdx.x % 7); }
  const int wave = (int)(threadIdx.x >> 6);
  const bool roleB = wave >= 4;
  float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);

  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {                      // 4 bodies per loop trip: loop overhead amortised
      if constexpr (MODE == M_FMA_INDEP) {
#pragma unroll
        for (int i = 0; i < 16; ++i) FMA1(a[i]);
      } else if constexpr (MODE == M_FMA_DEP) {
#pragma unroll
        for (int i = 0; i < 16; ++i) FMA1(a[0]);
      } else if constexpr (MODE == M_FMA_DEP4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) FMA1(a[i & 3]);
      } else if constexpr (MODE == M_EXP) {
#pragma unroll
        for (int i = 0; i < 16; ++i) EXP1(a[i]);
      } else if constexpr (MODE == M_CMPSEL) {
#pragma unroll
        for (int i = 0; i < 8; ++i) CMPSEL(a[i]);
      } else if constexpr (MODE == M_PKFMA) {
#pragma unroll
        for (int i = 0; i < 16; ++i) PKFMA1(p[i]);
      } else if constexpr (MODE == M_MFMA4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) m4[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, m4[i & 3], 0, 0, 0);
      } else if constexpr (MODE == M_MFMA16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) m4[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, m4[i & 3], 0, 0, 0);
      } else if constexpr (MODE == M_MFMABF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) m16[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, m16[i & 1], 0, 0, 0);
      } else if constexpr (MODE == M_MIX_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          m4[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, m4[i], 0, 0, 0);
          FMA1(a[3 * i]); FMA1(a[3 * i + 1]); FMA1(a[3 * i + 2]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if constexpr (MODE == M_MIX_BF) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          m16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, m16[i], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 8; ++j) FMA1(a[8 * i + j]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if constexpr (MODE == M_SPLIT_F32) {
        if (!roleB) {
#pragma unroll
          for (int i = 0; i < 16; ++i) FMA1(a[i]);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) m4[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, m4[i & 3], 0, 0, 0);
        }
      } else if constexpr (MODE == M_SPLIT_F32_16) {
        if (!roleB) {
#pragma unroll
          for (int i = 0; i < 16; ++i) FMA1(a[i]);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) m4[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, m4[i & 3], 0, 0, 0);
        }
      } else if constexpr (MODE == M_SPLIT_BF) {
        if (!roleB) {
#pragma unroll
          for (int i = 0; i < 16; ++i) FMA1(a[i]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) m16[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, m16[i & 1], 0, 0, 0);
        }
      } else if constexpr (MODE == M_LDS_B128) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float4 v = lds[(it + i) & 63];                // wave-uniform address: broadcast
          acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w;
        }
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  // keep everything alive
  float s = acc4.x + acc4.y + acc4.z + acc4.w;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += m4[i][0] + m4[i][1] + m4[i][2] + m4[i][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += m16[i][j];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
static void run(int waves_per_simd, int iters, unsigned long long* d_out, float* d_sink) {
  const bool split = MODE == M_SPLIT_F32 || MODE == M_SPLIT_F32_16 || MODE == M_SPLIT_BF;
  const int threads = split ? 512 : 256;
  const int blocks_per_cu = split ? waves_per_simd / 2 : waves_per_simd;
  if (blocks_per_cu < 1) return;
  const int blocks = 256 * blocks_per_cu;
  (void)hipMemset(d_out, 0, (size_t)blocks * 8 * sizeof(unsigned long long));
  for (int warm = 0; warm < 2; ++warm) {
    hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1.0001f, 0.9999f, d_sink);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed mode %d\n", MODE); exit(3); }
  std::vector<unsigned long long> h((size_t)blocks * 8);
  (void)hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> ta, tb;
  const int wpb = threads / 64;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < wpb; ++w) (w >= 4 ? tb : ta).push_back((double)h[(size_t)b * 8 + w]);
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
  const double bodies = 4.0 * iters;
  const double ca = med(ta) / bodies;
  if (split) {
    const double cb = med(tb) / bodies;
    printf("%-40s waves/SIMD %d : role A %7.1f cyc/body   role B %7.1f cyc/body\n", kNames[MODE], waves_per_simd, ca, cb);
  } else {
    // SIMD view: waves_per_simd waves each finished `bodies` bodies in ca*bodies cycles
    printf("%-40s waves/SIMD %d : %7.1f cyc/body/wave  = %6.2f cyc per body on the SIMD\n", kNames[MODE], waves_per_simd, ca,
           ca / waves_per_simd);
  }
}

template <int MODE>
static void sweep(unsigned long long* d_out, float* d_sink, int iters) {
  const bool split = MODE == M_SPLIT_F32 || MODE == M_SPLIT_F32_16 || MODE == M_SPLIT_BF;
  if (split) { run<MODE>(2, iters, d_out, d_sink); run<MODE>(4, iters, d_out, d_sink); }
  else for (int w = 1; w <= 4; ++w) run<MODE>(w, iters, d_out, d_sink);
}

int main() {
  unsigned long long* d_out; float* d_sink;
  if (hipMalloc(&d_out, 256 * 8 * 8 * sizeof(unsigned long long)) != hipSuccess) return 2;
  if (hipMalloc(&d_sink, 64) != hipSuccess) return 2;
  const int iters = 2000;
  printf("cycles = s_memtime ticks of the median wave; body = the instruction group named on the left\n");
  sweep<M_FMA_INDEP>(d_out, d_sink, iters);
  sweep<M_FMA_DEP>(d_out, d_sink, iters);
  sweep<M_FMA_DEP4>(d_out, d_sink, iters);
  sweep<M_EXP>(d_out, d_sink, iters);
  sweep<M_CMPSEL>(d_out, d_sink, iters);
  sweep<M_PKFMA>(d_out, d_sink, iters);
  sweep<M_MFMA4>(d_out, d_sink, iters);
  sweep<M_MFMA16>(d_out, d_sink, iters);
  sweep<M_MFMABF>(d_out, d_sink, iters);
  sweep<M_MIX_F32>(d_out, d_sink, iters);
  sweep<M_MIX_BF>(d_out, d_sink, iters);
  sweep<M_SPLIT_F32>(d_out, d_sink, iters);
  sweep<M_SPLIT_F32_16>(d_out, d_sink, iters);
  sweep<M_SPLIT_BF>(d_out, d_sink, iters);
  sweep<M_LDS_B128>(d_out, d_sink, iters);
  return 0;
}
