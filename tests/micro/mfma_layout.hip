// Self-checking micro-test of the two MFMA facts composite_*_mx relies on (gfx950):
//   v_mfma_f32_16x16x1_4b_f32 with cbsz=2/abid=b:  D[4*blk + r] of lane (j = L&15, q = L>>4)
//        = A(lane 16*b + 4*q + r) * B(lane 16*blk + j)          (4 blocks in the 4 register groups, A broadcast from block b)
//   the same with cbsz=0:  D[4*blk + r] of lane (j, q) = A(lane 16*blk + 4*q + r) * B(lane 16*blk + j)   (no sharing)
//   v_mfma_f32_16x16x4_f32:                          D[r] of lane (j, q) = sum_k A(lane (4q + r) + 16 k) * B(lane j + 16 k)
//   v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 blocks, block = L>>2):
//        D[v] of lane L = A(lane 4*(L>>2) + v) * B(lane L)                 (row v from the block's lanes, column = L&3)
//   the same with cbsz=4/abid=g:  D[v] of lane L = A(lane 4*g + v) * B(lane L)      (all blocks share block g's A rows)
//   the same with cbsz=2/abid=g:  D[v] of lane L = A(lane 16*(L>>4) + 4*g + v) * B(lane L)
//        (the four blocks of each 16-lane group share the A rows of the group's block g: one 4x4 pixel quadrant per group,
//         each quadrant with its OWN four splats -- what the quadrant-queue composites are built on)
// Exit code 0 = layout as assumed.  Build: hipcc --offload-arch=gfx950 mfma_layout.hip -o mfma_layout.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ABID>
__global__ void k16x1(float* out) {
  const int l = threadIdx.x;
  f32x16 d = {0};
  d = __builtin_amdgcn_mfma_f32_16x16x1f32(100.f + l, 1.f, d, 2, ABID, 0);
  f32x16 e = {0};
  e = __builtin_amdgcn_mfma_f32_16x16x1f32(1.f, 1000.f + l, e, 2, ABID, 0);
  for (int r = 0; r < 16; ++r) { out[l * 32 + r] = d[r]; out[l * 32 + 16 + r] = e[r]; }
}
// no broadcast (cbsz = 0): every block multiplies ITS OWN 16 A lanes with its own 16 B lanes
__global__ void k16x1_own(float* out) {
  const int l = threadIdx.x;
  f32x16 d = {0};
  d = __builtin_amdgcn_mfma_f32_16x16x1f32(100.f + l, 1.f, d, 0, 0, 0);
  f32x16 e = {0};
  e = __builtin_amdgcn_mfma_f32_16x16x1f32(1.f, 1000.f + l, e, 0, 0, 0);
  for (int r = 0; r < 16; ++r) { out[l * 32 + r] = d[r]; out[l * 32 + 16 + r] = e[r]; }
}
__global__ void k16x4(float* out) {
  const int l = threadIdx.x;
  f32x4 d = {0, 0, 0, 0};
  // A(lane) = 1 + lane, B(lane) = 2^-(lane&15): the sum over the 4 k-slots identifies both operands
  d = __builtin_amdgcn_mfma_f32_16x16x4f32((float)(1 + l), (float)(l & 15) + 1.f, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
__global__ void k4x4(float* out) {
  const int l = threadIdx.x;
  f32x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + l), (float)(3 + l), d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, 1.f, d, 0, 0, 0);            // accumulates: + 1 everywhere
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
// A broadcast for the 16-block form: cbsz = 4 makes all 16 blocks read block ABID's four A lanes
template <int ABID>
__global__ void k4x4_bcast(float* out) {
  const int l = threadIdx.x;
  f32x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + l), (float)(3 + l), d, 4, ABID, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
template <int ABID>
static int check4x4_bcast(float* d) {
  float h[64 * 4];
  k4x4_bcast<ABID><<<1, 64>>>(d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1000;
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v)
      if (h[l * 4 + v] != (float)(1 + 4 * ABID + v) * (float)(3 + l)) ++bad;
  if (bad) printf("4x4x1_16b cbsz=4 abid=%d: %d mismatches (lane 9: %g %g %g %g)\n", ABID, bad, h[36], h[37], h[38], h[39]);
  return bad;
}
template <int ABID>
__global__ void k4x4_group(float* out) {
  const int l = threadIdx.x;
  f32x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + l), (float)(3 + l), d, 2, ABID, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
template <int ABID>
static int check4x4_group(float* d) {
  float h[64 * 4];
  k4x4_group<ABID><<<1, 64>>>(d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1000;
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v)
      if (h[l * 4 + v] != (float)(1 + 16 * (l >> 4) + 4 * ABID + v) * (float)(3 + l)) ++bad;
  if (bad) printf("4x4x1_16b cbsz=2 abid=%d: %d mismatches (lane 21: %g %g %g %g)\n", ABID, bad, h[84], h[85], h[86], h[87]);
  return bad;
}
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error line %d\n", __LINE__); return 2; } } while (0)

template <int ABID>
static int check16x1(float* d) {
  float h[64 * 32];
  k16x1<ABID><<<1, 64>>>(d);
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int blk = 0; blk < 4; ++blk)
      for (int r = 0; r < 4; ++r) {
        const int j = l & 15, q = l >> 4;
        if (h[l * 32 + 4 * blk + r] != 100.f + 16 * ABID + 4 * q + r) ++bad;
        if (h[l * 32 + 16 + 4 * blk + r] != 1000.f + 16 * blk + j) ++bad;
      }
  if (bad) printf("16x16x1_4b cbsz=2 abid=%d: %d mismatches\n", ABID, bad);
  return bad;
}
int main() {
  float* d;
  CK(hipMalloc(&d, 64 * 32 * sizeof(float)));
  int bad = check16x1<0>(d) + check16x1<1>(d) + check16x1<2>(d) + check16x1<3>(d);
  {
    float g[64 * 32];
    k16x1_own<<<1, 64>>>(d);
    CK(hipMemcpy(g, d, sizeof(g), hipMemcpyDeviceToHost));
    int bad0 = 0;
    for (int l = 0; l < 64; ++l)
      for (int blk = 0; blk < 4; ++blk)
        for (int r = 0; r < 4; ++r) {
          const int j = l & 15, q = l >> 4;
          if (g[l * 32 + 4 * blk + r] != 100.f + 16 * blk + 4 * q + r) ++bad0;      // A row 4q+r of block blk
          if (g[l * 32 + 16 + 4 * blk + r] != 1000.f + 16 * blk + j) ++bad0;         // B column j of block blk
        }
    if (bad0) printf("16x16x1_4b cbsz=0: %d mismatches\n", bad0);
    bad += bad0;
  }
  float h[64 * 4];
  k16x4<<<1, 64>>>(d);
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int j = l & 15, q = l >> 4, row = 4 * q + r;
      float want = 0.f;
      for (int k = 0; k < 4; ++k) want += (float)(1 + row + 16 * k) * ((float)j + 1.f);
      if (h[l * 4 + r] != want) ++bad;
    }
  k4x4<<<1, 64>>>(d);
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  int bad4 = 0;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v)
      if (h[l * 4 + v] != (float)(1 + 4 * (l >> 2) + v) * (float)(3 + l) + 1.f) ++bad4;
  if (bad4) printf("4x4x1_16b: %d mismatches (lane 5: %g %g %g %g)\n", bad4, h[20], h[21], h[22], h[23]);
  bad += bad4;
  bad += check4x4_bcast<0>(d) + check4x4_bcast<5>(d) + check4x4_bcast<15>(d);
  bad += check4x4_group<0>(d) + check4x4_group<1>(d) + check4x4_group<2>(d) + check4x4_group<3>(d);
  printf(bad ? "MFMA layout DIFFERS from what the kernels assume (%d)\n" : "mfma layouts ok\n", bad);
  return bad ? 1 : 0;
}
