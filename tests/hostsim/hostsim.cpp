// TEST INFRASTRUCTURE ONLY -- host (g++) build of vtgaussian-slam_amd/csrc/vtgs_math.h so the per-Gaussian
// projection forward and the hand-derived backward can be checked against the autograd oracle on a CPU-only
// box.  Never loaded by the product package; the product path runs these same inline functions on the GPU.
#include "../../vtgaussian-slam_amd/csrc/vtgs_math.h"
#include <cstdint>

using namespace vtgs;

static CamParams make_cam(int W, int H, float tanfovx, float tanfovy, float mod, int rule, const float* V, const float* PV) {
  CamParams c;
  for (int i = 0; i < 16; ++i) { c.V[i] = V[i]; c.PV[i] = PV[i]; }
  c.W = W; c.H = H;
  c.fx = (float)W / (2.f * tanfovx); c.fy = (float)H / (2.f * tanfovy);
  c.limx = kFovClamp * tanfovx; c.limy = kFovClamp * tanfovy;
  c.mod = mod;
  c.gx16 = (W + 15) / 16; c.gy16 = (H + 15) / 16; c.gx8 = (W + 7) / 8; c.gy8 = (H + 7) / 8;
  c.row8_begin = 0; c.row8_end = c.gy8; c.radius_rule = rule;
  return c;
}

extern "C" {

// out[n][12] = u v A B C depth radius x0 y0 x1 y1 visible
void hostsim_project(int W, int H, float tanfovx, float tanfovy, float mod, int rule, const float* V, const float* PV,
                     int n, const float* means, const float* scales, const float* rots, const float* opac, float* out) {
  const CamParams cam = make_cam(W, H, tanfovx, tanfovy, mod, rule, V, PV);
  for (int i = 0; i < n; ++i) {
    Splat sp{}; SplatAux aux{};
    const bool vis = project_splat(cam, means + 3 * i, scales + 3 * i, rots + 4 * i, opac[i], sp, aux);
    float* o = out + 12 * i;
    o[0] = sp.u; o[1] = sp.v; o[2] = sp.A; o[3] = sp.B; o[4] = sp.C; o[5] = sp.depth; o[6] = (float)sp.radius;
    o[7] = (float)sp.x0; o[8] = (float)sp.y0; o[9] = (float)sp.x1; o[10] = (float)sp.y1; o[11] = vis ? 1.f : 0.f;
  }
}

// moments[n][9] -> grads[n][17] = mean3D(3) mean2D(3) color(3) opacity(1) scale(3) rot(4)
void hostsim_backward(int W, int H, float tanfovx, float tanfovy, float mod, int rule, const float* V, const float* PV,
                      int n, const float* means, const float* scales, const float* rots, const float* opac,
                      const float* moments, float* grads) {
  const CamParams cam = make_cam(W, H, tanfovx, tanfovy, mod, rule, V, PV);
  for (int i = 0; i < n; ++i) {
    Splat sp{}; SplatAux aux{};
    float* g = grads + 17 * i;
    for (int k = 0; k < 17; ++k) g[k] = 0.f;
    if (!project_splat(cam, means + 3 * i, scales + 3 * i, rots + 4 * i, opac[i], sp, aux)) continue;
    SplatMoments mo;
    for (int k = 0; k < 9; ++k) mo.m[k] = moments[9 * i + k];
    SplatGrads sg;
    splat_backward(cam, scales + 3 * i, rots + 4 * i, opac[i], sp, aux, mo, sg);
    for (int k = 0; k < 3; ++k) { g[k] = sg.mean3D[k]; g[3 + k] = sg.mean2D[k]; g[6 + k] = sg.color[k]; g[10 + k] = sg.scale[k]; }
    g[9] = sg.opacity;
    for (int k = 0; k < 4; ++k) g[13 + k] = sg.rot[k];
  }
}

// out[n] = 1 where a call over tile rows [row_b, row_e) may skip the Gaussian before projecting it (conservative y-cull)
void hostsim_outside_rows(int W, int H, float tanfovx, float tanfovy, float mod, int rule, const float* V, const float* PV,
                          int n, const float* means, const float* scales, int row_b, int row_e, uint8_t* out) {
  const CamParams cam = make_cam(W, H, tanfovx, tanfovy, mod, rule, V, PV);
  for (int i = 0; i < n; ++i) out[i] = outside_tile_rows(cam, means + 3 * i, scales + 3 * i, row_b, row_e) ? 1 : 0;
}

// min of the quadratic form over a rectangle (tile culling predicate)
float hostsim_min_quadratic(float A, float B, float C, float u, float v, float px0, float py0, float px1, float py1) {
  return min_quadratic_over_rect(A, B, C, u, v, px0, py0, px1, py1);
}
}
