// Micro-test: decode the A/B/D lane layout of v_mfma_f32_16x16x1_4b_f32 and its cbsz/abid A-broadcast.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void k16(float* out) {
  const int l = threadIdx.x;
  f32x16 d = {0};
  // A = 100 + lane, B = 1 (so D tells which A lane fed each output), then A = 1, B = 1000 + lane
  d = __builtin_amdgcn_mfma_f32_16x16x1f32(100.f + l, 1.f, d, CBSZ, ABID, 0);
  for (int r = 0; r < 16; ++r) out[l * 32 + r] = d[r];
  f32x16 e = {0};
  e = __builtin_amdgcn_mfma_f32_16x16x1f32(1.f, 1000.f + l, e, CBSZ, ABID, 0);
  for (int r = 0; r < 16; ++r) out[l * 32 + 16 + r] = e[r];
}
template <int CBSZ, int ABID>
__global__ void k4(float* out) {
  const int l = threadIdx.x;
  f32x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(100.f + l, 1.f, d, CBSZ, ABID, 0);
  for (int r = 0; r < 4; ++r) out[l * 8 + r] = d[r];
  f32x4 e = {0, 0, 0, 0};
  e = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, 1000.f + l, e, CBSZ, ABID, 0);
  for (int r = 0; r < 4; ++r) out[l * 8 + 4 + r] = e[r];
}
static void dump16(const char* name, float* h) {
  printf("== %s: lane: D regs (A-source lanes) | D regs (B-source lanes)\n", name);
  for (int l = 0; l < 64; l += 1) {
    if (!(l < 20 || l % 16 == 0 || l == 63)) continue;
    printf("lane %2d: A:", l);
    for (int r = 0; r < 16; ++r) printf(" %3.0f", h[l * 32 + r] - 100.f);
    printf(" | B:");
    for (int r = 0; r < 16; ++r) printf(" %3.0f", h[l * 32 + 16 + r] - 1000.f);
    printf("\n");
  }
}
int main() {
  float *d, h[64 * 32];
  hipMalloc(&d, sizeof(h));
  k16<0, 0><<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); dump16("16x16x1 cbsz=0", h);
  k16<2, 1><<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); dump16("16x16x1 cbsz=2 abid=1", h);
  k16<2, 3><<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); dump16("16x16x1 cbsz=2 abid=3", h);
  float h4[64 * 8];
  k4<0, 0><<<1, 64>>>(d); hipMemcpy(h4, d, sizeof(h4), hipMemcpyDeviceToHost);
  printf("== 4x4x1 cbsz=0\n");
  for (int l = 0; l < 12; ++l) { printf("lane %2d: A:", l); for (int r = 0; r < 4; ++r) printf(" %3.0f", h4[l*8+r]-100.f); printf(" | B:"); for (int r = 0; r < 4; ++r) printf(" %3.0f", h4[l*8+4+r]-1000.f); printf("\n"); }
  k4<4, 5><<<1, 64>>>(d); hipMemcpy(h4, d, sizeof(h4), hipMemcpyDeviceToHost);
  printf("== 4x4x1 cbsz=4 abid=5\n");
  for (int l = 0; l < 12; ++l) { printf("lane %2d: A:", l); for (int r = 0; r < 4; ++r) printf(" %3.0f", h4[l*8+r]-100.f); printf(" | B:"); for (int r = 0; r < 4; ++r) printf(" %3.0f", h4[l*8+4+r]-1000.f); printf("\n"); }
  return 0;
}
