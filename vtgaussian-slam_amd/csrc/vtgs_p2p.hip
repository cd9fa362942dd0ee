// vtgs_p2p.hip -- point-to-plane consistency of two depth frames on the device.
//
// Replaces the host path of `compute_point2plane_dist` (src/vtgaussian_slam.py:1070-1155): kornia normals -> numpy ->
// Open3D KD-tree on the host, called per tracking iteration at base-frame boundaries (:1929, :1956, :2158, :2185).
//   p2p_target   one thread per pixel of the TARGET (latest) frame: world point ((x - cx + 0.5)/fx convention of
//                get_pointcloud, :76-101), normal (3x3 Sobel / 8 of the K^-1 [u,v,1] d point image with replicate
//                padding, cross, normalise: kornia.geometry.depth_to_normals; rotated to the world, :1158-1178) and
//                the "seen by the source camera" flag (get_frustum_mask, :1046-1065).
//   p2p_source   one thread per pixel of the SOURCE (current) frame: world point, "seen by the target camera" flag,
//                then the NEAREST target point within `threshold` -- found exactly, not approximately: the target
//                points are the target frame's pixels, so every candidate within `threshold` of the source point
//                projects into a window of known radius around the source point's own projection into the target
//                frame (derivation at window_radius below).  Writes n . (p_source - p_target) per matched pixel.
// The reduction (sum of squares / max / mean of the 100 largest) is a torch reduction on the device in the Python layer.
#include "../../include/vtgs.h"
#include "vtgs_internal.h"

namespace vtgs {

struct P2PFrame { float k[9]; float w2c[16]; float c2w[16]; };   // row-major; c2w = rigid inverse

__device__ __forceinline__ P2PFrame load_frame(const float* __restrict__ k, const float* __restrict__ w2c) {
  P2PFrame f;
#pragma unroll
  for (int i = 0; i < 9; ++i) f.k[i] = k[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) f.w2c[i] = w2c[i];
  // rigid inverse: R^T, -R^T t
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) f.c2w[4 * r + c] = f.w2c[4 * c + r];
    f.c2w[4 * r + 3] = -(f.w2c[0 + r] * f.w2c[3] + f.w2c[4 + r] * f.w2c[7] + f.w2c[8 + r] * f.w2c[11]);
  }
  f.c2w[12] = f.c2w[13] = f.c2w[14] = 0.f; f.c2w[15] = 1.f;
  return f;
}
__device__ __forceinline__ float3 xform(const float (&m)[16], float3 p) {
  return make_float3(fmaf(m[0], p.x, fmaf(m[1], p.y, fmaf(m[2], p.z, m[3]))),
                     fmaf(m[4], p.x, fmaf(m[5], p.y, fmaf(m[6], p.z, m[7]))),
                     fmaf(m[8], p.x, fmaf(m[9], p.y, fmaf(m[10], p.z, m[11]))));
}
// get_frustum_mask: uv = K (w2c p); z = uv.z + 1e-8; 0 < u < W, 0 < v < H, z > 0   (K general 3x3)
__device__ __forceinline__ bool in_frustum(const P2PFrame& f, float3 pw, int W, int H) {
  const float3 c = xform(f.w2c, pw);
  const float ux = f.k[0] * c.x + f.k[1] * c.y + f.k[2] * c.z, uy = f.k[3] * c.x + f.k[4] * c.y + f.k[5] * c.z;
  const float z = f.k[6] * c.x + f.k[7] * c.y + f.k[8] * c.z + 1e-8f;
  const float u = ux / z, v = uy / z;
  return u < (float)W && u > 0.f && v < (float)H && v > 0.f && z > 0.f;
}
__device__ __forceinline__ float3 backproject(const P2PFrame& f, int x, int y, float d) {      // get_pointcloud, factor 1
  const float3 pc = make_float3(((float)x - f.k[2] + 0.5f) / f.k[0] * d, ((float)y - f.k[5] + 0.5f) / f.k[4] * d, d);
  return xform(f.c2w, pc);
}

__global__ __launch_bounds__(256) void p2p_target(int W, int H, const float* __restrict__ depth, const uint8_t* __restrict__ mask,
                                                  const float* __restrict__ k, const float* __restrict__ w2c_t,
                                                  const float* __restrict__ w2c_s, int frustum,
                                                  float4* __restrict__ tgt_p, float4* __restrict__ tgt_n) {
  const int pix = (int)(blockIdx.x * 256u + threadIdx.x);
  if (pix >= W * H) return;
  const P2PFrame ft = load_frame(k, w2c_t), fs = load_frame(k, w2c_s);
  const int x = pix % W, y = pix / W;
  const float d = depth[pix];
  bool ok = d > 0.f && (!mask || mask[pix]);
  const float3 pw = backproject(ft, x, y, d);
  if (ok && frustum) ok = in_frustum(fs, pw, W, H);
  // kornia depth_to_normals: xyz = K^-1 [u, v, 1] d (pinhole K: ((u - cx)/fx, (v - cy)/fy, 1) d), Sobel / 8, replicate pad
  float3 gx = make_float3(0.f, 0.f, 0.f), gy = gx;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = min(max(x + dx, 0), W - 1), yy = min(max(y + dy, 0), H - 1);
      const float dd = depth[yy * W + xx];
      const float3 q = make_float3(((float)xx - ft.k[2]) / ft.k[0] * dd, ((float)yy - ft.k[5]) / ft.k[4] * dd, dd);
      const float wx = (float)dx * (dy == 0 ? 2.f : 1.f) * 0.125f, wy = (float)dy * (dx == 0 ? 2.f : 1.f) * 0.125f;
      gx.x = fmaf(wx, q.x, gx.x); gx.y = fmaf(wx, q.y, gx.y); gx.z = fmaf(wx, q.z, gx.z);
      gy.x = fmaf(wy, q.x, gy.x); gy.y = fmaf(wy, q.y, gy.y); gy.z = fmaf(wy, q.z, gy.z);
    }
  float3 n = make_float3(gx.y * gy.z - gx.z * gy.y, gx.z * gy.x - gx.x * gy.z, gx.x * gy.y - gx.y * gy.x);
  const float inv = 1.f / fmaxf(sqrtf(n.x * n.x + n.y * n.y + n.z * n.z), 1e-12f);
  n.x *= inv; n.y *= inv; n.z *= inv;
  const float3 nw = make_float3(ft.c2w[0] * n.x + ft.c2w[1] * n.y + ft.c2w[2] * n.z, ft.c2w[4] * n.x + ft.c2w[5] * n.y + ft.c2w[6] * n.z,
                                ft.c2w[8] * n.x + ft.c2w[9] * n.y + ft.c2w[10] * n.z);
  tgt_p[pix] = make_float4(pw.x, pw.y, pw.z, ok ? 1.f : 0.f);
  tgt_n[pix] = make_float4(nw.x, nw.y, nw.z, 0.f);
}

// Window that is guaranteed to hold every target pixel whose point lies within t of the source point.  In the target
// camera the source point is (X, Y, Z) and a candidate (X', Y', Z') with |X' - X|, |Y' - Y|, |Z' - Z| <= t; candidates are
// pixel centres u' = fx X'/Z' + cx - 0.5.  |X'/Z' - X/Z| <= |X' - X| / Z' + |X| |1/Z' - 1/Z| <= t (1 + |X|/Z) / (Z - t), so
// |u' - u| <= fx t (1 + |X|/Z) / (Z - t); +1 covers the rounding to integer pixels.  Z <= 1.5 t: the whole image.
__global__ __launch_bounds__(256) void p2p_source(int W, int H, const float* __restrict__ depth, const uint8_t* __restrict__ mask,
                                                  const float* __restrict__ k, const float* __restrict__ w2c_t,
                                                  const float* __restrict__ w2c_s, int frustum, float threshold,
                                                  const float4* __restrict__ tgt_p, const float4* __restrict__ tgt_n,
                                                  float* __restrict__ out_dist, uint8_t* __restrict__ out_matched) {
  const int pix = (int)(blockIdx.x * 256u + threadIdx.x);
  if (pix >= W * H) return;
  const P2PFrame ft = load_frame(k, w2c_t), fs = load_frame(k, w2c_s);
  const int x = pix % W, y = pix / W;
  const float d = depth[pix];
  bool ok = d > 0.f && (!mask || mask[pix]);
  const float3 pw = backproject(fs, x, y, d);
  if (ok && frustum) ok = in_frustum(ft, pw, W, H);
  float best = threshold * threshold, dist = 0.f;
  bool found = false;
  if (ok) {
    const float3 c = xform(ft.w2c, pw);
    int x0 = 0, x1 = W - 1, y0 = 0, y1 = H - 1;
    if (c.z > 1.5f * threshold) {
      const float u = ft.k[0] * c.x / c.z + ft.k[2] - 0.5f, v = ft.k[4] * c.y / c.z + ft.k[5] - 0.5f;
      const float ru = ft.k[0] * threshold * (1.f + fabsf(c.x) / c.z) / (c.z - threshold) + 1.f;
      const float rv = ft.k[4] * threshold * (1.f + fabsf(c.y) / c.z) / (c.z - threshold) + 1.f;
      x0 = max(0, (int)floorf(u - ru)); x1 = min(W - 1, (int)ceilf(u + ru));
      y0 = max(0, (int)floorf(v - rv)); y1 = min(H - 1, (int)ceilf(v + rv));
    }
    for (int yy = y0; yy <= y1; ++yy)
      for (int xx = x0; xx <= x1; ++xx) {
        const float4 q = tgt_p[yy * W + xx];
        const float ex = pw.x - q.x, ey = pw.y - q.y, ez = pw.z - q.z;
        const float d2 = ex * ex + ey * ey + ez * ez;
        if (q.w != 0.f && d2 < best) {
          best = d2; found = true;
          const float4 nn = tgt_n[yy * W + xx];
          dist = nn.x * ex + nn.y * ey + nn.z * ez;
        }
      }
  }
  out_dist[pix] = found ? dist : 0.f;
  out_matched[pix] = found ? 1 : 0;
}

}  // namespace vtgs

using namespace vtgs;

extern "C" {

size_t vtgs_point2plane_scratch_bytes(int32_t width, int32_t height) {
  if (width <= 0 || height <= 0) return 0;
  return (size_t)width * height * 2 * sizeof(float4);
}

int vtgs_point2plane(int32_t width, int32_t height, const float* depth_target, const float* depth_source,
                     const uint8_t* mask_target, const uint8_t* mask_source, const float* intrinsics, const float* w2c_target,
                     const float* w2c_source, float threshold, int32_t frustum, void* scratch, size_t scratch_bytes,
                     float* out_dist, uint8_t* out_matched, void* stream) {
  if (width <= 0 || height <= 0 || !depth_target || !depth_source || !intrinsics || !w2c_target || !w2c_source || !scratch ||
      !out_dist || !out_matched || !(threshold > 0.f))
    return VTGS_ERR_INVALID_ARGUMENT;
  if (scratch_bytes < vtgs_point2plane_scratch_bytes(width, height)) return VTGS_ERR_WORKSPACE_TOO_SMALL;
  const int P = width * height;
  float4* tp = (float4*)scratch;
  float4* tn = tp + P;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(p2p_target, dim3((P + 255) / 256), dim3(256), 0, st, width, height, depth_target, mask_target, intrinsics,
                     w2c_target, w2c_source, frustum, tp, tn);
  hipLaunchKernelGGL(p2p_source, dim3((P + 255) / 256), dim3(256), 0, st, width, height, depth_source, mask_source, intrinsics,
                     w2c_target, w2c_source, frustum, threshold, (const float4*)tp, (const float4*)tn, out_dist, out_matched);
  return hipGetLastError() == hipSuccess ? VTGS_OK : VTGS_ERR_HIP;
}

}  // extern "C"
