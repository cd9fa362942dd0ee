// Self-checking micro-test of the two gfx950 row-swap instructions the backward's quarter reduction relies on:
//   v_permlane32_swap(a, b): r0 = (a.lanes 0-31 , b.lanes 0-31),  r1 = (a.lanes 32-63, b.lanes 32-63)
//   v_permlane16_swap(a, b): r0 = (a.row0, b.row0, a.row2, b.row2), r1 = (a.row1, b.row1, a.row3, b.row3)   (rows = 16 lanes)
// so swap32_add / swap16_add / quarter_sum of vtgs_composite.hip leave row rho holding the 4-row total of register rho.
// Exit code 0 = as assumed.  Build: hipcc --offload-arch=gfx950 permlane_swap.hip -o permlane_swap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ float swap32_add(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ float swap16_add(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__global__ void k(float* out) {
  const int l = threadIdx.x;
  unsigned a = l, b = 100 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1]; out[128 + l] = s[0]; out[192 + l] = s[1];
  // register v of lane l holds 1000 v + l: row rho of the quarter sum must be sum over rows of register rho
  float p[4];
  for (int v = 0; v < 4; ++v) p[v] = 1000.f * v + l;
  out[256 + l] = swap16_add(swap32_add(p[0], p[2]), swap32_add(p[1], p[3]));
}
int main() {
  float* d;
  if (hipMalloc(&d, 320 * 4) != hipSuccess) return 2;
  k<<<1, 64>>>(d);
  float h[320];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int row = l >> 4, c = l & 15;
    bad += h[l] != (l < 32 ? l : 100 + (l - 32));
    bad += h[64 + l] != (l < 32 ? 32 + l : 100 + l);
    bad += h[128 + l] != ((row & 1) ? 100 + 16 * (row - 1) + c : 16 * row + c);
    bad += h[192 + l] != ((row & 1) ? 100 + 16 * row + c : 16 * (row + 1) + c);
    float want = 0.f;
    for (int r = 0; r < 4; ++r) want += 1000.f * row + (16 * r + c);      // register rho = row, summed over the 4 rows
    bad += h[256 + l] != want;
  }
  printf(bad ? "permlane swap semantics DIFFER from what the kernels assume (%d)\n" : "permlane swaps ok\n", bad);
  return bad ? 1 : 0;
}
