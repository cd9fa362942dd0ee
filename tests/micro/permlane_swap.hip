#include <hip/hip_runtime.h>
__global__ void k(float* out) {
  const int l = threadIdx.x;
  unsigned a = l, b = 100 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1]; out[128 + l] = s[0]; out[192 + l] = s[1];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int v = 0; v < 4; ++v) { for (int l = 0; l < 64; l += 1) printf("%g ", h[v * 64 + l]); printf("\n"); }
  return 0;
}
