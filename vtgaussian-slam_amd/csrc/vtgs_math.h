// vtgs_math.h -- per-Gaussian projection maths shared by the HIP kernels (device) and by the
// test-only host build in tests/hostsim (g++), so the hand-derived backward can be checked against
// autograd on the CPU.  No memory access, no intrinsics: plain float arithmetic.
//
// Semantics: SURVEY.md Appendix A1/A5 (published 3DGS EWA projection; the reference's call sites fix
// the conventions: viewmatrix/projmatrix arrive transposed, utils/recon_helpers.py:8,12-13;
// quaternions are (w,x,y,z), utils/slam_external.py:29-32).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VTGS_HD __host__ __device__ __forceinline__
#else
#define VTGS_HD inline
#endif

namespace vtgs {

constexpr float kNearCull   = 0.2f;
constexpr float kDilation   = 0.3f;
constexpr float kFovClamp   = 1.3f;
constexpr float kAlphaMin   = 1.0f / 255.0f;
constexpr float kAlphaMax   = 0.99f;
constexpr float kTStop      = 1e-4f;
constexpr int   kBinTile    = 16;   // granularity that decides which pixels a splat may reach
constexpr int   kSubTile    = 8;    // granularity of the composite (one wavefront = 8x8 pixels)
// View-tied splats sit exactly on pixel centres in their own base frame (src/vtgaussian_slam.py:87-88
// back-projects (x - cx + 0.5)/fx), so (u +- r)/16 is an exact integer for 1 splat in 16 and float32
// rounding noise (~1e-7 tile) would decide the tile rectangle at random.  The floor is therefore taken
// 1e-4 tile (1.6e-3 px) above the value: the lattice case resolves to the exact-arithmetic answer.
constexpr float kRectEps    = 1e-4f;

struct CamParams {          // scalar camera state broadcast to every thread
  float V[16];              // viewmatrix memory order: t_j = sum_i p_i V[4*i+j]
  float PV[16];             // projmatrix memory order
  float fx, fy;             // focal lengths in pixels = W/(2 tanfovx), H/(2 tanfovy)
  float limx, limy;         // 1.3 * tanfov
  float mod;                // scale_modifier
  int   W, H;
  int   gx16, gy16;         // 16x16 tile grid
  int   gx8, gy8;           // 8x8 tile grid
  int   row8_begin, row8_end; // band of 8-pixel tile rows rendered by this call
  int   radius_rule;
};

struct Splat {              // forward result for one Gaussian
  float u, v;               // pixel centre, rounded to float32 ...
  float ulo, vlo;           // ... and what the rounding took: (u + ulo, v + vlo) is the centre to ~2^-45 relative (round 5)
  float A, B, C;            // conic (inverse 2-D covariance)
  float depth;              // view-space z (sort key and depth channel)
  int   radius;             // 0 => culled
  int   x0, y0, x1, y1;     // half-open rectangle of 16x16 tiles
};

// intermediate values the backward needs again (recomputed, never stored in HBM)
struct SplatAux {
  float tx, ty, tz;         // view-space mean, x/y possibly clamped
  float xmul, ymul;         // 0 when the clamp was active, else 1
  float a, b, c;            // dilated 2-D covariance
  float cov3[6];            // xx xy xz yy yz zz
  float R[9];               // rotation from the (un-normalised) quaternion, row-major
  float hw_inv;             // 1/(h.w + 1e-7)
  float hx, hy;
};

VTGS_HD void quat_to_R(const float q[4], float R[9]) {
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
  R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
  R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma3 = R diag(s^2) R^T, upper triangle
VTGS_HD void cov3_from_scale_rot(const float s[3], float mod, const float R[9], float cov[6]) {
  const float s0 = mod * s[0], s1 = mod * s[1], s2 = mod * s[2];
  const float q0 = s0 * s0, q1 = s1 * s1, q2 = s2 * s2;
  cov[0] = R[0] * R[0] * q0 + R[1] * R[1] * q1 + R[2] * R[2] * q2;
  cov[1] = R[0] * R[3] * q0 + R[1] * R[4] * q1 + R[2] * R[5] * q2;
  cov[2] = R[0] * R[6] * q0 + R[1] * R[7] * q1 + R[2] * R[8] * q2;
  cov[3] = R[3] * R[3] * q0 + R[4] * R[4] * q1 + R[5] * R[5] * q2;
  cov[4] = R[3] * R[6] * q0 + R[4] * R[7] * q1 + R[5] * R[8] * q2;
  cov[5] = R[6] * R[6] * q0 + R[7] * R[7] * q1 + R[8] * R[8] * q2;
}

VTGS_HD int splat_radius(float lam_max, float opacity, int rule) {
  const float r3 = ceilf(3.f * sqrtf(lam_max));
  if (rule == 1) {
    const float ext = 2.f * logf(fmaxf(255.f * opacity, 1.f));
    return (int)fminf(r3, ceilf(sqrtf(ext * lam_max)));
  }
  return (int)r3;
}

VTGS_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// M = J * Rw2c (2x3), with Rw2c[i][j] = V[4*j+i]
VTGS_HD void ewa_M(const CamParams& cam, float tx, float ty, float tz, float M[6]) {
  const float iz = 1.f / tz, iz2 = iz * iz;
  const float j00 = cam.fx * iz, j02 = -cam.fx * tx * iz2;
  const float j11 = cam.fy * iz, j12 = -cam.fy * ty * iz2;
  // row 0 of J = (j00, 0, j02); row 1 = (0, j11, j12);  (J Rw)[r][k] = sum_i J[r][i] * V[4*k+i]
  for (int k = 0; k < 3; ++k) {
    M[k]     = j00 * cam.V[4 * k + 0] + j02 * cam.V[4 * k + 2];
    M[3 + k] = j11 * cam.V[4 * k + 1] + j12 * cam.V[4 * k + 2];
  }
}

// The half-open rectangle of 16x16 tiles under a splat's centre and radius (with the lattice epsilon
// of kRectEps).  project_splat calls it; so does the end phase of project_and_bin, which rebuilds the rectangle of a deferred
// splat from its stored centre and radius -- the same floats through the same operations, hence the same rectangle.
VTGS_HD void tile_rect(const CamParams& cam, float u, float v, int radius, int& x0, int& y0, int& x1, int& y1) {
  const float rf = (float)radius;
  const float it = 1.f / (float)kBinTile;
  x0 = clampi((int)floorf((u - rf) * it + kRectEps), 0, cam.gx16);
  x1 = clampi((int)floorf((u + rf + (float)(kBinTile - 1)) * it + kRectEps), 0, cam.gx16);
  y0 = clampi((int)floorf((v - rf) * it + kRectEps), 0, cam.gy16);
  y1 = clampi((int)floorf((v + rf + (float)(kBinTile - 1)) * it + kRectEps), 0, cam.gy16);
}

// Pixel centre.  A float32 centre at |u| ~ 1000 px carries ~6e-5 px of rounding -- against exponent slopes of a few per
// pixel that is the 3e-4 relative noise on every alpha that put the 99.9th percentile of the gradient error at 1.4e-3 of
// the float64 oracle at 1200x680 (CPU study: the float32 oracle itself drops from 2e-3 to 1.5e-5 when
// only the centre OFFSETS are exact).  The homogeneous coordinates and the divide are therefore carried in double (a
// dozen half-rate instructions per Gaussian), and the centre leaves as a float32 pair: the consumers form
// (u - tile centre) + ulo, whose first term is exact.  (A function of its own since round 6: gather_splat_grads needs the
// centre, and nothing else of the projection, while it sums the records.)
VTGS_HD void pixel_centre(const CamParams& cam, float x, float y, float z, float& u, float& v, float& ulo, float& vlo) {
  const float* P = cam.PV;
  const double xd = (double)x, yd = (double)y, zd = (double)z;
  const double hxd = (double)P[0] * xd + ((double)P[4] * yd + ((double)P[8] * zd + (double)P[12]));
  const double hyd = (double)P[1] * xd + ((double)P[5] * yd + ((double)P[9] * zd + (double)P[13]));
  const double hwd = (double)P[3] * xd + ((double)P[7] * yd + ((double)P[11] * zd + (double)P[15])) + 1e-7;
  // quotient: the float32 one, then one Newton step in double (relative error ~2^-45)
  const float inv = 1.f / (float)hwd;
  double qx = (double)((float)hxd * inv), qy = (double)((float)hyd * inv);
  qx += (hxd - qx * hwd) * (double)inv;
  qy += (hyd - qy * hwd) * (double)inv;
  const double hwid = 0.5 * (double)cam.W, hhd = 0.5 * (double)cam.H;            // ((q + 1) W - 1) / 2 = q W/2 + (W - 1)/2
  const double ud = qx * hwid + (hwid - 0.5), vd = qy * hhd + (hhd - 0.5);
  u = (float)ud; v = (float)vd;
  ulo = (float)(ud - (double)u); vlo = (float)(vd - (double)v);
}

// Forward projection of one Gaussian.  Returns false when culled (radius = 0).
// cov_precomp != NULL: the caller gives the 3-D covariance itself (xx xy xz yy yz zz -- the operator's `cov3D_precomp`, used as
// it is: no scale modifier) instead of scale + rotation; every existing caller passes the default and pays nothing.
// centre4 != NULL: (u, v, ulo, vlo) as pixel_centre returned them for this mean -- the caller formed the centre ahead of the
// rest (gather_splat_grads) and the double-precision chain is not run a second time.
VTGS_HD bool project_splat(const CamParams& cam, const float mean[3], const float scale[3],
                           const float quat[4], float opacity, Splat& out, SplatAux& aux, const float* cov_precomp = nullptr,
                           const float* centre4 = nullptr) {
  out.radius = 0; out.x0 = out.y0 = out.x1 = out.y1 = 0;
  const float x = mean[0], y = mean[1], z = mean[2];
  const float* V = cam.V;
  // view-space mean; the z chain is the sort key and its order is part of the contract with the oracle
  const float tz = fmaf(V[2], x, fmaf(V[6], y, fmaf(V[10], z, V[14])));
  out.depth = tz;
  if (!(tz > kNearCull)) return false;
  float tx = fmaf(V[0], x, fmaf(V[4], y, fmaf(V[8], z, V[12])));
  float ty = fmaf(V[1], x, fmaf(V[5], y, fmaf(V[9], z, V[13])));
  const float* P = cam.PV;
  const float hx = fmaf(P[0], x, fmaf(P[4], y, fmaf(P[8], z, P[12])));
  const float hy = fmaf(P[1], x, fmaf(P[5], y, fmaf(P[9], z, P[13])));
  const float hw = fmaf(P[3], x, fmaf(P[7], y, fmaf(P[11], z, P[15])));
  const float hw_inv = 1.f / (hw + 1e-7f);
  aux.hw_inv = hw_inv; aux.hx = hx; aux.hy = hy;
  if (centre4) { out.u = centre4[0]; out.v = centre4[1]; out.ulo = centre4[2]; out.vlo = centre4[3]; }
  else pixel_centre(cam, x, y, z, out.u, out.v, out.ulo, out.vlo);

  if (cov_precomp) {
    for (int i = 0; i < 6; ++i) aux.cov3[i] = cov_precomp[i];
    for (int i = 0; i < 9; ++i) aux.R[i] = 0.f;
  } else {
    quat_to_R(quat, aux.R);
    cov3_from_scale_rot(scale, cam.mod, aux.R, aux.cov3);
  }

  const float rx = tx / tz, ry = ty / tz;
  aux.xmul = (rx < -cam.limx || rx > cam.limx) ? 0.f : 1.f;
  aux.ymul = (ry < -cam.limy || ry > cam.limy) ? 0.f : 1.f;
  tx = fminf(cam.limx, fmaxf(-cam.limx, rx)) * tz;
  ty = fminf(cam.limy, fmaxf(-cam.limy, ry)) * tz;
  aux.tx = tx; aux.ty = ty; aux.tz = tz;

  float M[6];
  ewa_M(cam, tx, ty, tz, M);
  const float* S = aux.cov3;
  // Sigma2 = M Sigma3 M^T
  const float m0[3] = {S[0] * M[0] + S[1] * M[1] + S[2] * M[2], S[1] * M[0] + S[3] * M[1] + S[4] * M[2],
                       S[2] * M[0] + S[4] * M[1] + S[5] * M[2]};          // Sigma3 * M_row0
  const float m1[3] = {S[0] * M[3] + S[1] * M[4] + S[2] * M[5], S[1] * M[3] + S[3] * M[4] + S[4] * M[5],
                       S[2] * M[3] + S[4] * M[4] + S[5] * M[5]};          // Sigma3 * M_row1
  const float a = M[0] * m0[0] + M[1] * m0[1] + M[2] * m0[2] + kDilation;
  const float b = M[0] * m1[0] + M[1] * m1[1] + M[2] * m1[2];
  const float c = M[3] * m1[0] + M[4] * m1[1] + M[5] * m1[2] + kDilation;
  aux.a = a; aux.b = b; aux.c = c;
  const float det = a * c - b * b;
  if (det == 0.f || !(det == det)) return false;
  const float det_inv = 1.f / det;
  out.A = c * det_inv; out.B = -b * det_inv; out.C = a * det_inv;
  const float mid = 0.5f * (a + c);
  const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  const int radius = splat_radius(lam, opacity, cam.radius_rule);
  const float rf = (float)radius;
  if (!(fabsf(out.u) < 1e9f) || !(fabsf(out.v) < 1e9f) || !(rf < 1e9f)) return false;
  tile_rect(cam, out.u, out.v, radius, out.x0, out.y0, out.x1, out.y1);
  if ((out.x1 - out.x0) * (out.y1 - out.y0) <= 0) return false;
  out.radius = radius;
  return true;
}

// Tile-row partition: can this Gaussian be skipped by a call that renders 16-pixel tile rows [row_b, row_e)
// only?  True only when project_splat would give it a tile rectangle that misses those rows (or cull it) -- decided from the
// mean and the scales alone, before the rotation, the opacity and the covariance algebra are touched:
//   lambda_max(Sigma2) <= smax(J)^2 smax(Rw)^2 lambda_max(Sigma3) + 0.3,   lambda_max(Sigma3) = (mod max scale)^2,
//   smax(J)^2 = the larger eigenvalue of J J^T (2x2, J as in ewa_M with the clamped tx/tz, ty/tz),
//   smax(Rw)^2 <= the largest absolute row sum of Rw^T Rw (exactly 1 for a rigid view matrix),
// and the radius of either rule is at most ceil(3 sqrt(lam)), lam <= lambda_max + sqrt(0.1) (the floor inside the root).
// Two pixels of slack cover ceil() and the float32 rounding of v on both sides; a NaN makes every comparison false: not skipped.
// `margin_px` widens the radius and `growth` multiplies the scales: the owned sets of the tile-row partition
// (vtgs_band_owner_mask) ask "could this Gaussian meet the rows after the pose has moved a little / the scales have grown";
// (0, 1) is the test project_and_bin makes, to the bit (r + 0 and smax * 1 are exact).
VTGS_HD bool outside_tile_rows_ext(const CamParams& cam, const float mean[3], const float scale[3], int row_b, int row_e,
                                   float margin_px, float growth) {
  const float x = mean[0], y = mean[1], z = mean[2];
  const float* V = cam.V;
  const float tz = fmaf(V[2], x, fmaf(V[6], y, fmaf(V[10], z, V[14])));
  if (!(tz > kNearCull)) return tz <= kNearCull;               // culled by the near plane on every rank (NaN: not decided here)
  const float* P = cam.PV;
  const float hy = fmaf(P[1], x, fmaf(P[5], y, fmaf(P[9], z, P[13])));
  const float hw = fmaf(P[3], x, fmaf(P[7], y, fmaf(P[11], z, P[15])));
  const float v = ((hy * (1.f / (hw + 1e-7f)) + 1.f) * (float)cam.H - 1.f) * 0.5f;
  const float smax = cam.mod * fmaxf(fabsf(scale[0]), fmaxf(fabsf(scale[1]), fabsf(scale[2]))) * growth;
  float gram = 0.f;                                            // smax(Rw)^2 <= max_i sum_j |(Rw^T Rw)_ij|
  for (int i = 0; i < 3; ++i) {
    float row = 0.f;
    for (int j = 0; j < 3; ++j) row += fabsf(V[4 * i] * V[4 * j] + V[4 * i + 1] * V[4 * j + 1] + V[4 * i + 2] * V[4 * j + 2]);
    gram = fmaxf(gram, row);
  }
  const float tx = fmaf(V[0], x, fmaf(V[4], y, fmaf(V[8], z, V[12])));
  const float ty = fmaf(V[1], x, fmaf(V[5], y, fmaf(V[9], z, V[13])));
  const float iz = 1.f / tz;
  const float rx = fminf(cam.limx, fmaxf(-cam.limx, tx * iz)), ry = fminf(cam.limy, fmaxf(-cam.limy, ty * iz));
  const float ja = cam.fx * cam.fx * (1.f + rx * rx), jc = cam.fy * cam.fy * (1.f + ry * ry), jb = cam.fx * cam.fy * rx * ry;
  const float hd = 0.5f * (ja - jc);
  const float j2 = (0.5f * (ja + jc) + sqrtf(hd * hd + jb * jb)) * iz * iz;
  const float lam = j2 * gram * smax * smax * 1.001f + kDilation + 0.32f;
  const float r = 3.f * sqrtf(lam) + 2.f + margin_px;          // ceil() and one pixel of slack
  const float it = 1.f / (float)kBinTile;
  const bool above = (v + r + (float)(kBinTile - 1)) * it + kRectEps < (float)row_b + 1.f;   // => y1 = floor(.) <= row_b
  const bool below = (v - r) * it >= (float)row_e;                                          // => y0 = floor(. + eps) >= row_e
  return above || below;
}

// The 16-pixel tile row of a Gaussian's projected centre, clamped to the image (a centre above / below the image counts for the
// first / last row); -1 behind the near plane.  Owner bands of the tile-row partition (partition.OwnerExchange): every rank
// computes it from the same bytes, so every rank names the same owner.
VTGS_HD int centre_tile_row(const CamParams& cam, const float mean[3]) {
  const float x = mean[0], y = mean[1], z = mean[2];
  const float* V = cam.V;
  const float tz = fmaf(V[2], x, fmaf(V[6], y, fmaf(V[10], z, V[14])));
  if (!(tz > kNearCull)) return -1;
  const float* P = cam.PV;
  const float hy = fmaf(P[1], x, fmaf(P[5], y, fmaf(P[9], z, P[13])));
  const float hw = fmaf(P[3], x, fmaf(P[7], y, fmaf(P[11], z, P[15])));
  const float v = ((hy * (1.f / (hw + 1e-7f)) + 1.f) * (float)cam.H - 1.f) * 0.5f;
  const float r = floorf(v * (1.f / (float)kBinTile));
  const float top = (float)(cam.gy16 - 1);
  return (int)(r > 0.f ? (r < top ? r : top) : 0.f);              // (NaN: row 0)
}

VTGS_HD bool outside_tile_rows(const CamParams& cam, const float mean[3], const float scale[3], int row_b, int row_e) {
  return outside_tile_rows_ext(cam, mean, scale, row_b, row_e, 0.f, 1.f);
}

// Smallest value of q(d) = 1/2 (A dx^2 + C dy^2) + B dx dy over the pixel-centre rectangle
// [px0,px1] x [py0,py1] for a splat centred at (u,v).  A splat reaches a pixel only where
// q <= ln(255 o), so a tile whose minimum exceeds that bound receives nothing from it.
VTGS_HD float min_quadratic_over_rect(float A, float B, float C, float u, float v,
                                      float px0, float py0, float px1, float py1) {
  // d = centre - pixel, pixel in the rectangle => dx in [u-px1, u-px0]
  const float dx0 = u - px1, dx1 = u - px0, dy0 = v - py1, dy1 = v - py0;
  if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return 0.f;
  float best = 3.0e38f;
  // vertical edges (dx fixed): minimise over dy  ->  dy* = -B dx / C
  for (int e = 0; e < 2; ++e) {
    const float dx = e ? dx1 : dx0;
    float dy = (C > 0.f) ? (-B * dx / C) : dy0;
    dy = fminf(dy1, fmaxf(dy0, dy));
    best = fminf(best, 0.5f * (A * dx * dx + C * dy * dy) + B * dx * dy);
  }
  for (int e = 0; e < 2; ++e) {
    const float dy = e ? dy1 : dy0;
    float dx = (A > 0.f) ? (-B * dy / A) : dx0;
    dx = fminf(dx1, fmaxf(dx0, dx));
    best = fminf(best, 0.5f * (A * dx * dx + C * dy * dy) + B * dx * dy);
  }
  return best;
}

// Per-Gaussian sums produced by the backward composite (over all pixels the splat reached):
//   with u_p = G_p * dL/dalpha_p (G = exp(power), un-clamped), d = centre - pixel
//   m[0]=sum u   m[1]=sum u dx  m[2]=sum u dy  m[3]=sum u dx^2  m[4]=sum u dx dy  m[5]=sum u dy^2
//   m[6..8] = sum_p w_p * dL/dcolor_p[ch]      (w = alpha * T)
struct SplatMoments { float m[9]; };

struct SplatGrads {
  float mean3D[3], mean2D[3], color[3], opacity, scale[3], rot[4];
};

// Backward of project_splat + the conic/opacity part of the composite.  `sp`/`aux` are recomputed
// by the caller with project_splat on the same inputs.
// g_cov6 != NULL (cov3D_precomp): dL/d(xx xy xz yy yz zz) is written there -- an off-diagonal entry fills two places of the
// symmetric matrix, so it gets twice the matrix gradient -- and the scale / rotation gradients are zero.
VTGS_HD void splat_backward(const CamParams& cam, const float scale[3], const float quat[4], float opacity,
                            const Splat& sp, const SplatAux& aux, const SplatMoments& mo, SplatGrads& g, float* g_cov6 = nullptr) {
  const float* m = mo.m;
  g.color[0] = m[6]; g.color[1] = m[7]; g.color[2] = m[8];
  g.opacity = m[0];
  // dL/d(pixel centre) and dL/d(conic)
  const float o = opacity;
  const float dLdu = o * (-sp.A * m[1] - sp.B * m[2]);
  const float dLdv = o * (-sp.C * m[2] - sp.B * m[1]);
  const float dA = -0.5f * o * m[3], dB = -o * m[4], dC = -0.5f * o * m[5];
  g.mean2D[0] = dLdu * 0.5f * (float)cam.W;
  g.mean2D[1] = dLdv * 0.5f * (float)cam.H;
  g.mean2D[2] = 0.f;

  // conic -> dilated covariance (a,b,c);  1/(det^2 + 1e-7) as in the published backward
  const float a = aux.a, b = aux.b, c = aux.c;
  const float det = a * c - b * b;
  const float d2i = 1.f / (det * det + 1e-7f);
  const float da = d2i * (-c * c * dA + b * c * dB - b * b * dC);
  const float dc = d2i * (-b * b * dA + a * b * dB - a * a * dC);
  const float db = d2i * (2.f * b * c * dA - (a * c + b * b) * dB + 2.f * a * b * dC);
  // symmetric gradient matrix G2 = [[da, db/2],[db/2, dc]]
  const float h = 0.5f * db;

  float M[6];
  ewa_M(cam, aux.tx, aux.ty, aux.tz, M);
  // dL/dSigma3 = M^T G2 M  (symmetric 3x3); row r of (G2 M):
  float GM0[3], GM1[3];
  for (int k = 0; k < 3; ++k) { GM0[k] = da * M[k] + h * M[3 + k]; GM1[k] = h * M[k] + dc * M[3 + k]; }
  float G3[9];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) G3[3 * i + k] = M[i] * GM0[k] + M[3 + i] * GM1[k];

  // dL/dM = 2 G2 M Sigma3
  const float* S = aux.cov3;
  const float Sf[9] = {S[0], S[1], S[2], S[1], S[3], S[4], S[2], S[4], S[5]};
  float dM[6];
  for (int k = 0; k < 3; ++k) {
    dM[k]     = 2.f * (GM0[0] * Sf[k] + GM0[1] * Sf[3 + k] + GM0[2] * Sf[6 + k]);
    dM[3 + k] = 2.f * (GM1[0] * Sf[k] + GM1[1] * Sf[3 + k] + GM1[2] * Sf[6 + k]);
  }
  // M = J Rw  =>  dL/dJ[r][i] = sum_k dM[r][k] * Rw[i][k],  Rw[i][k] = V[4*k+i]
  const float* V = cam.V;
  const float dJ00 = dM[0] * V[0] + dM[1] * V[4] + dM[2] * V[8];
  const float dJ02 = dM[0] * V[2] + dM[1] * V[6] + dM[2] * V[10];
  const float dJ11 = dM[3] * V[1] + dM[4] * V[5] + dM[5] * V[9];
  const float dJ12 = dM[3] * V[2] + dM[4] * V[6] + dM[5] * V[10];
  const float tz = aux.tz, iz = 1.f / tz, iz2 = iz * iz, iz3 = iz2 * iz;
  const float dtx = aux.xmul * (-cam.fx * iz2) * dJ02;
  const float dty = aux.ymul * (-cam.fy * iz2) * dJ12;
  const float dtz = -cam.fx * iz2 * dJ00 - cam.fy * iz2 * dJ11
                    + 2.f * cam.fx * aux.tx * iz3 * dJ02 + 2.f * cam.fy * aux.ty * iz3 * dJ12;
  // t = Rw mu + tau  =>  dL/dmu_k = sum_i dt_i * Rw[i][k] = sum_i dt_i V[4*k+i]
  float gm[3];
  for (int k = 0; k < 3; ++k) gm[k] = dtx * V[4 * k + 0] + dty * V[4 * k + 1] + dtz * V[4 * k + 2];

  // pixel centre -> mean through the perspective divide of the full projection
  const float P_du = dLdu * 0.5f * (float)cam.W, P_dv = dLdv * 0.5f * (float)cam.H;  // dL/dndc
  const float wi = aux.hw_inv;
  const float dhx = P_du * wi, dhy = P_dv * wi;
  const float dhw = -(P_du * aux.hx + P_dv * aux.hy) * wi * wi;
  const float* P = cam.PV;
  for (int k = 0; k < 3; ++k) gm[k] += dhx * P[4 * k + 0] + dhy * P[4 * k + 1] + dhw * P[4 * k + 3];
  g.mean3D[0] = gm[0]; g.mean3D[1] = gm[1]; g.mean3D[2] = gm[2];

  if (g_cov6) {
    g_cov6[0] = G3[0]; g_cov6[1] = 2.f * G3[1]; g_cov6[2] = 2.f * G3[2];
    g_cov6[3] = G3[4]; g_cov6[4] = 2.f * G3[5]; g_cov6[5] = G3[8];
    g.scale[0] = g.scale[1] = g.scale[2] = 0.f;
    g.rot[0] = g.rot[1] = g.rot[2] = g.rot[3] = 0.f;
    return;
  }
  // Sigma3 = R diag(s'^2) R^T, s' = mod*s:  dL/ds_i = 2 s'_i mod (R^T G3 R)_ii ; dL/dR = 2 G3 R diag(s'^2)
  const float* R = aux.R;
  float sp2[3], dR[9];
  for (int i = 0; i < 3; ++i) {
    const float si = cam.mod * scale[i];
    sp2[i] = si * si;
    // column i of R: R[0*3+i], R[1*3+i], R[2*3+i]
    float acc = 0.f;
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) acc += R[3 * r + i] * G3[3 * r + k] * R[3 * k + i];
    g.scale[i] = 2.f * si * cam.mod * acc;
  }
  for (int r = 0; r < 3; ++r)
    for (int i = 0; i < 3; ++i)
      dR[3 * r + i] = 2.f * (G3[3 * r + 0] * R[0 + i] + G3[3 * r + 1] * R[3 + i] + G3[3 * r + 2] * R[6 + i]) * sp2[i];
  // R(q) entries are quadratic in the raw quaternion (no normalisation inside the operator)
  const float qr = quat[0], qx = quat[1], qy = quat[2], qz = quat[3];
  g.rot[0] = 2.f * (-qz * dR[1] + qy * dR[2] + qz * dR[3] - qx * dR[5] - qy * dR[6] + qx * dR[7]);
  g.rot[1] = 2.f * (qy * dR[1] + qz * dR[2] + qy * dR[3] - 2.f * qx * dR[4] - qr * dR[5] + qz * dR[6] + qr * dR[7] - 2.f * qx * dR[8]);
  g.rot[2] = 2.f * (-2.f * qy * dR[0] + qx * dR[1] + qr * dR[2] + qx * dR[3] + qz * dR[5] - qr * dR[6] + qz * dR[7] - 2.f * qy * dR[8]);
  g.rot[3] = 2.f * (-2.f * qz * dR[0] - qr * dR[1] + qx * dR[2] + qr * dR[3] - 2.f * qz * dR[4] + qy * dR[5] + qx * dR[6] + qy * dR[7]);
}

}  // namespace vtgs
