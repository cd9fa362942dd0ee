// vtgs_sh.hip -- spherical-harmonics colours of the operator's `shs` argument (gfx950).
//
// The operator takes either `colors_precomp` or `shs` (it "must raise if both / neither"); the reference only ever
// passes colours (utils/slam_helpers.py:152-159, sh_degree = 0 at utils/recon_helpers.py:22), so this half of the surface is
// off the hot path: one thread per Gaussian in front of the rasterizer (colours from the SH coefficients and the viewing
// direction) and one behind it (dL/dcolour -> dL/dshs and the direction's share of dL/dmeans3D).  No kernel of the rasterizer
// changes: the composites see colours.
//
// Semantics [UPSTREAM-PUBLIC] (the published evaluation, restated; oracle/gs_oracle.py::sh_colors is the float64 form with
// autograd behind it): dir = (mean - campos) / |mean - campos|; colour = sum_k Y_k(dir) sh[k] + 0.5, real SH basis up to
// degree 3 with the usual constants; negative channels are clamped to 0 and pass no gradient.
#include "../../include/vtgs.h"
#include "vtgs_internal.h"

namespace vtgs {

constexpr float kSH0 = 0.28209479177387814f;
constexpr float kSH1 = 0.4886025119029199f;
constexpr float kSH2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
constexpr float kSH3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f,
                           1.445305721320277f, -0.5900435899266435f};

// the basis values Y[0 .. (deg+1)^2) at the unit direction (x, y, z)
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float (&Y)[16]) {
  Y[0] = kSH0;
  if (deg > 0) {
    Y[1] = -kSH1 * y; Y[2] = kSH1 * z; Y[3] = -kSH1 * x;
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      Y[4] = kSH2[0] * xy; Y[5] = kSH2[1] * yz; Y[6] = kSH2[2] * (2.f * zz - xx - yy); Y[7] = kSH2[3] * xz; Y[8] = kSH2[4] * (xx - yy);
      if (deg > 2) {
        Y[9] = kSH3[0] * y * (3.f * xx - yy); Y[10] = kSH3[1] * xy * z; Y[11] = kSH3[2] * y * (4.f * zz - xx - yy);
        Y[12] = kSH3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); Y[13] = kSH3[4] * x * (4.f * zz - xx - yy);
        Y[14] = kSH3[5] * z * (xx - yy); Y[15] = kSH3[6] * x * (xx - 3.f * yy);
      }
    }
  }
}

// shs [n, coeffs, 3]; colors [n, 3]; clamped [n] bit c = channel c was negative (clamped to 0: no gradient through it)
__global__ __launch_bounds__(256) void sh_forward_kernel(int n, int deg, int coeffs, const float* __restrict__ means3D,
                                                         const float* __restrict__ campos, const float* __restrict__ shs,
                                                         float* __restrict__ colors, uint8_t* __restrict__ clamped) {
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= n) return;
  float dx = means3D[3 * i] - campos[0], dy = means3D[3 * i + 1] - campos[1], dz = means3D[3 * i + 2] - campos[2];
  const float inv = rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-30f));
  dx *= inv; dy *= inv; dz *= inv;
  float Y[16];
  sh_basis(deg, dx, dy, dz, Y);
  const int m = (deg + 1) * (deg + 1);
  const float* sh = shs + (size_t)i * coeffs * 3;
  float c[3] = {0.5f, 0.5f, 0.5f};
  for (int k = 0; k < m; ++k) { c[0] = fmaf(Y[k], sh[3 * k], c[0]); c[1] = fmaf(Y[k], sh[3 * k + 1], c[1]); c[2] = fmaf(Y[k], sh[3 * k + 2], c[2]); }
  uint8_t bits = 0;
  for (int ch = 0; ch < 3; ++ch) {
    if (c[ch] < 0.f) { bits |= (uint8_t)(1u << ch); c[ch] = 0.f; }
    colors[3 * i + ch] = c[ch];
  }
  clamped[i] = bits;
}

// g_shs [n, coeffs, 3] (coefficients beyond the active degree get 0); g_means3D [n, 3] = the direction's share only
__global__ __launch_bounds__(256) void sh_backward_kernel(int n, int deg, int coeffs, const float* __restrict__ means3D,
                                                          const float* __restrict__ campos, const float* __restrict__ shs,
                                                          const uint8_t* __restrict__ clamped, const float* __restrict__ g_colors,
                                                          float* __restrict__ g_shs, float* __restrict__ g_means3D) {
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= n) return;
  const float px = means3D[3 * i] - campos[0], py = means3D[3 * i + 1] - campos[1], pz = means3D[3 * i + 2] - campos[2];
  const float inv = rsqrtf(fmaxf(px * px + py * py + pz * pz, 1e-30f));
  const float x = px * inv, y = py * inv, z = pz * inv;
  const uint8_t bits = clamped[i];
  float g[3];
  for (int ch = 0; ch < 3; ++ch) g[ch] = (bits >> ch) & 1u ? 0.f : g_colors[3 * i + ch];
  float Y[16];
  sh_basis(deg, x, y, z, Y);
  const int m = (deg + 1) * (deg + 1);
  const float* sh = shs + (size_t)i * coeffs * 3;
  float* gs = g_shs ? g_shs + (size_t)i * coeffs * 3 : nullptr;
  if (gs)
    for (int k = 0; k < coeffs; ++k)
      for (int ch = 0; ch < 3; ++ch) gs[3 * k + ch] = k < m ? Y[k] * g[ch] : 0.f;
  if (!g_means3D) return;
  // dL/d(dir) = sum_k dY_k/d(dir) (g . sh[k]); then through dir = p / |p|
  float w[16];
  for (int k = 0; k < 16; ++k) w[k] = k < m ? g[0] * sh[3 * k] + g[1] * sh[3 * k + 1] + g[2] * sh[3 * k + 2] : 0.f;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (deg > 0) {
    gy += -kSH1 * w[1]; gz += kSH1 * w[2]; gx += -kSH1 * w[3];
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z;
      gx += kSH2[0] * y * w[4] + kSH2[2] * (-2.f * x) * w[6] + kSH2[3] * z * w[7] + kSH2[4] * 2.f * x * w[8];
      gy += kSH2[0] * x * w[4] + kSH2[1] * z * w[5] + kSH2[2] * (-2.f * y) * w[6] + kSH2[4] * (-2.f * y) * w[8];
      gz += kSH2[1] * y * w[5] + kSH2[2] * 4.f * z * w[6] + kSH2[3] * x * w[7];
      if (deg > 2) {
        gx += kSH3[0] * 6.f * x * y * w[9] + kSH3[1] * y * z * w[10] + kSH3[2] * (-2.f * x * y) * w[11] + kSH3[3] * (-6.f * x * z) * w[12]
              + kSH3[4] * (4.f * zz - 3.f * xx - yy) * w[13] + kSH3[5] * 2.f * x * z * w[14] + kSH3[6] * 3.f * (xx - yy) * w[15];
        gy += kSH3[0] * 3.f * (xx - yy) * w[9] + kSH3[1] * x * z * w[10] + kSH3[2] * (4.f * zz - xx - 3.f * yy) * w[11]
              + kSH3[3] * (-6.f * y * z) * w[12] + kSH3[4] * (-2.f * x * y) * w[13] + kSH3[5] * (-2.f * y * z) * w[14]
              + kSH3[6] * (-6.f * x * y) * w[15];
        gz += kSH3[1] * x * y * w[10] + kSH3[2] * 8.f * y * z * w[11] + kSH3[3] * (6.f * zz - 3.f * xx - 3.f * yy) * w[12]
              + kSH3[4] * 8.f * x * z * w[13] + kSH3[5] * (xx - yy) * w[14];
      }
    }
  }
  // d(p / |p|)/dp = (I - dir dir^T) / |p|
  const float dot = gx * x + gy * y + gz * z;
  g_means3D[3 * i] = (gx - x * dot) * inv;
  g_means3D[3 * i + 1] = (gy - y * dot) * inv;
  g_means3D[3 * i + 2] = (gz - z * dot) * inv;
}

}  // namespace vtgs

using namespace vtgs;

extern "C" {

int vtgs_sh_forward(int32_t n, int32_t degree, int32_t coeffs, const float* means3D, const float* campos, const float* shs,
                    float* out_colors, uint8_t* out_clamped, void* stream) {
  if (n < 0 || degree < 0 || degree > 3 || coeffs < (degree + 1) * (degree + 1) || coeffs > 16 || !campos) return VTGS_ERR_INVALID_ARGUMENT;
  if (n == 0) return VTGS_OK;
  if (!means3D || !shs || !out_colors || !out_clamped) return VTGS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sh_forward_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, degree, coeffs, means3D, campos, shs,
                     out_colors, out_clamped);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

int vtgs_sh_backward(int32_t n, int32_t degree, int32_t coeffs, const float* means3D, const float* campos, const float* shs,
                     const uint8_t* clamped, const float* g_colors, float* g_shs, float* g_means3D, void* stream) {
  if (n < 0 || degree < 0 || degree > 3 || coeffs < (degree + 1) * (degree + 1) || coeffs > 16 || !campos) return VTGS_ERR_INVALID_ARGUMENT;
  if (n == 0) return VTGS_OK;
  if (!means3D || !shs || !clamped || !g_colors || (!g_shs && !g_means3D)) return VTGS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sh_backward_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, degree, coeffs, means3D, campos,
                     shs, clamped, g_colors, g_shs, g_means3D);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

}  // extern "C"
