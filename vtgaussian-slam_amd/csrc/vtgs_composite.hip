// vtgs_composite.hip -- front-to-back alpha composite (forward) and its backward (gfx950, wave64).
//
// Work decomposition: ONE WAVEFRONT = ONE 8x8 PIXEL TILE, lane <-> pixel.  A workgroup is four
// independent wavefronts (the 2x2 tiles of one 16x16 block, so they gather mostly the same splats from
// the same L1/L2); there is no LDS and no barrier in the forward.  The tile's depth-sorted list is walked
// in chunks of 64: each lane gathers one splat record (coalesced index read + 32-byte record gather),
// then the chunk is replayed splat by splat with the record broadcast from its lane into scalar
// registers (v_readlane), so the per-pixel arithmetic reads splat parameters as SGPR operands.
//
// Semantics per pixel: skip power>0; alpha=min(.99,o*exp(power)); skip alpha<1/255;
// stop before adding when T(1-alpha)<1e-4; C+=c*alpha*T; D+=z*alpha*T; out=C+T*bg.  exp is evaluated as
// exp2 of a pre-scaled quadratic form (v_exp_f32).
//
// Backward (Appendix A4, restated front-to-back): with g = dL/dcolor at the pixel, Cg = g.(out - T_final*bg)
// and the running prefix P_k = sum_{j<=k} (g.c_j) alpha_j T_j,
//     dL/dalpha_k = T_k (g.c_k) - (Cg - P_k + T_final (g.bg)) / (1 - alpha_k)
// so the list is replayed in the SAME order as the forward (identical skip/stop decisions by construction,
// no per-pixel contributor count to store) and the per-pixel state is two scalars (T, P).  Per splat the
// wavefront reduces nine sums over its 64 pixels (SplatMoments, vtgs_math.h) and stores them as one
// 48-byte record per (splat, tile) instance -- plain stores, no float atomics, bitwise reproducible.
// gather_splat_grads then sums each splat's contiguous run of records and runs splat_backward.
#include "vtgs_internal.h"

namespace vtgs {

constexpr float kLog2e = 1.4426950408889634f;

struct ChunkRec {     // one splat of the current chunk, held by one lane
  float u, v, qa, qb, qc, op, depth, c0, c1, c2;
};

__device__ __forceinline__ ChunkRec gather_chunk(const uint32_t* __restrict__ sorted_gid, const GeomRec* __restrict__ geom,
                                                 const float* __restrict__ colors, uint32_t pos, bool in) {
  ChunkRec r{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (in) {
    const uint32_t gid = sorted_gid[pos];
    const float4* gp = reinterpret_cast<const float4*>(geom + gid);
    const float4 g0 = gp[0], g1 = gp[1];
    r.u = g0.x; r.v = g0.y;
    r.qa = -0.5f * kLog2e * g0.z;     // exp(power) = exp2(dx*(qa*dx + qb*dy) + qc*dy*dy)
    r.qb = -kLog2e * g0.w;
    r.qc = -0.5f * kLog2e * g1.x;
    r.op = g1.y; r.depth = g1.z;
    r.c0 = colors[3 * gid]; r.c1 = colors[3 * gid + 1]; r.c2 = colors[3 * gid + 2];
  }
  return r;
}

// tile -> wavefront mapping shared by forward and backward
struct TileCoord { int tile, px, py; bool tile_ok, inside; };

__device__ __forceinline__ TileCoord tile_coord(const CamScalars& cs, uint32_t nblk16, int gx16, int gx8, int gy8) {
  const uint32_t b = xcd_swizzle(blockIdx.x, nblk16);
  const int row16_0 = cs.row8_begin >> 1;
  const int t16x = (int)(b % (uint32_t)gx16), t16y = row16_0 + (int)(b / (uint32_t)gx16);
  // the wavefront index is wave-uniform; saying so keeps list bounds and readlane selects in SGPRs
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), l = lane_id();
  const int t8x = 2 * t16x + (w & 1), t8y = 2 * t16y + (w >> 1);
  TileCoord tc;
  tc.tile_ok = t8x < gx8 && t8y < gy8 && t8y >= cs.row8_begin && t8y < cs.row8_end;
  tc.tile = t8y * gx8 + t8x;
  tc.px = t8x * kSubTile + (l & 7);
  tc.py = t8y * kSubTile + (l >> 3);
  tc.inside = tc.tile_ok && tc.px < cs.W && tc.py < cs.H;
  return tc;
}

__global__ __launch_bounds__(256) void composite_forward(
    CamScalars cs, const float* __restrict__ bg, uint32_t nblk16,
    const uint32_t* __restrict__ tile_off, const uint32_t* __restrict__ sorted_gid,
    const GeomRec* __restrict__ geom, const float* __restrict__ colors,
    float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ final_T,
    const Counters* __restrict__ ctr) {
  if (ctr->overflow) return;
  const int gx16 = (cs.W + kBinTile - 1) / kBinTile;
  const int gx8 = (cs.W + kSubTile - 1) / kSubTile, gy8 = (cs.H + kSubTile - 1) / kSubTile;
  const TileCoord tc = tile_coord(cs, nblk16, gx16, gx8, gy8);
  if (!tc.tile_ok) return;                       // wave-uniform
  const int l = lane_id();
  const float pxf = (float)tc.px, pyf = (float)tc.py;
  const uint32_t s = tile_off[tc.tile], e = tile_off[tc.tile + 1];

  float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
  bool done = !tc.inside;
  for (uint32_t base = s; base < e; base += 64u) {
    if (__ballot(!done) == 0ull) break;
    const int n = (int)min(64u, e - base);
    const ChunkRec r = gather_chunk(sorted_gid, geom, colors, base + (uint32_t)l, l < n);
    for (int j = 0; j < n; ++j) {
      const float su = bcast_f(r.u, j), sv = bcast_f(r.v, j);
      const float sa = bcast_f(r.qa, j), sb = bcast_f(r.qb, j), sc = bcast_f(r.qc, j);
      const float so = bcast_f(r.op, j);
      const float dx = su - pxf, dy = sv - pyf;
      const float p2 = dx * (sa * dx + sb * dy) + sc * dy * dy;
      const float alpha = fminf(kAlphaMax, so * __builtin_amdgcn_exp2f(p2));
      const float Tn = T * (1.f - alpha);
      bool hit = !done && p2 <= 0.f && alpha >= kAlphaMin;
      if (hit && Tn < kTStop) { done = true; hit = false; }
      if (__ballot(hit) != 0ull) {               // wave-uniform: nobody adds this splat -> skip colour reads
        const float wgt = hit ? alpha * T : 0.f;
        C0 = fmaf(bcast_f(r.c0, j), wgt, C0);
        C1 = fmaf(bcast_f(r.c1, j), wgt, C1);
        C2 = fmaf(bcast_f(r.c2, j), wgt, C2);
        D = fmaf(bcast_f(r.depth, j), wgt, D);
        T = hit ? Tn : T;
      }
    }
  }
  if (tc.inside) {
    const size_t P = (size_t)cs.W * cs.H, pix = (size_t)tc.py * cs.W + tc.px;
    out_color[pix] = C0 + T * bg[0];
    out_color[P + pix] = C1 + T * bg[1];
    out_color[2 * P + pix] = C2 + T * bg[2];
    out_depth[pix] = D;
    final_T[pix] = T;
  }
}

__global__ __launch_bounds__(256) void composite_backward(
    CamScalars cs, const float* __restrict__ bg, uint32_t nblk16,
    const uint32_t* __restrict__ tile_off, const uint32_t* __restrict__ sorted_gid,
    const uint32_t* __restrict__ sorted_inst, const GeomRec* __restrict__ geom, const float* __restrict__ colors,
    const float* __restrict__ out_color, const float* __restrict__ grad_color, const float* __restrict__ final_T,
    float* __restrict__ grad_inst) {
  const int gx16 = (cs.W + kBinTile - 1) / kBinTile;
  const int gx8 = (cs.W + kSubTile - 1) / kSubTile, gy8 = (cs.H + kSubTile - 1) / kSubTile;
  const TileCoord tc = tile_coord(cs, nblk16, gx16, gx8, gy8);
  if (!tc.tile_ok) return;
  const int l = lane_id();
  const float pxf = (float)tc.px, pyf = (float)tc.py;
  const uint32_t s = tile_off[tc.tile], e = tile_off[tc.tile + 1];

  float g0 = 0.f, g1 = 0.f, g2 = 0.f, Cg = 0.f, Bg = 0.f;
  if (tc.inside) {
    const size_t P = (size_t)cs.W * cs.H, pix = (size_t)tc.py * cs.W + tc.px;
    g0 = grad_color[pix]; g1 = grad_color[P + pix]; g2 = grad_color[2 * P + pix];
    const float Tf = final_T[pix];
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    Cg = g0 * (out_color[pix] - Tf * b0) + g1 * (out_color[P + pix] - Tf * b1) + g2 * (out_color[2 * P + pix] - Tf * b2);
    Bg = Tf * (g0 * b0 + g1 * b1 + g2 * b2);
  }
  float T = 1.f, Pfx = 0.f;
  bool done = !tc.inside;
  uint32_t base = s;
  for (; base < e; base += 64u) {
    if (__ballot(!done) == 0ull) break;
    const int n = (int)min(64u, e - base);
    const ChunkRec r = gather_chunk(sorted_gid, geom, colors, base + (uint32_t)l, l < n);
    const uint32_t my_inst = (l < n) ? sorted_inst[base + (uint32_t)l] : 0u;
    for (int j = 0; j < n; ++j) {
      const float su = bcast_f(r.u, j), sv = bcast_f(r.v, j);
      const float sa = bcast_f(r.qa, j), sb = bcast_f(r.qb, j), sc = bcast_f(r.qc, j);
      const float so = bcast_f(r.op, j);
      const float dx = su - pxf, dy = sv - pyf;
      const float p2 = dx * (sa * dx + sb * dy) + sc * dy * dy;
      const float G = __builtin_amdgcn_exp2f(p2);
      const float alpha = fminf(kAlphaMax, so * G);
      const float Tn = T * (1.f - alpha);
      bool hit = !done && p2 <= 0.f && alpha >= kAlphaMin;
      if (hit && Tn < kTStop) { done = true; hit = false; }
      float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f, m6 = 0.f, m7 = 0.f, m8 = 0.f;
      if (__ballot(hit) != 0ull) {
        const float c0 = bcast_f(r.c0, j), c1 = bcast_f(r.c1, j), c2 = bcast_f(r.c2, j);
        if (hit) {
          const float gc = g0 * c0 + g1 * c1 + g2 * c2;
          const float wgt = alpha * T;
          Pfx = fmaf(gc, wgt, Pfx);
          const float dLda = T * gc - (Cg - Pfx + Bg) / (1.f - alpha);
          const float uu = G * dLda;                 // clamp at 0.99 passes the gradient through
          m0 = uu; m1 = uu * dx; m2 = uu * dy; m3 = m1 * dx; m4 = m1 * dy; m5 = m2 * dy;
          m6 = wgt * g0; m7 = wgt * g1; m8 = wgt * g2;
          T = Tn;
        }
        m0 = wave_sum(m0); m1 = wave_sum(m1); m2 = wave_sum(m2); m3 = wave_sum(m3); m4 = wave_sum(m4);
        m5 = wave_sum(m5); m6 = wave_sum(m6); m7 = wave_sum(m7); m8 = wave_sum(m8);
      }
      // lane k < 12 stores float k of this instance's record
      const uint32_t inst = (uint32_t)bcast_i((int)my_inst, j);
      float val = 0.f;
      val = (l == 0) ? m0 : val; val = (l == 1) ? m1 : val; val = (l == 2) ? m2 : val;
      val = (l == 3) ? m3 : val; val = (l == 4) ? m4 : val; val = (l == 5) ? m5 : val;
      val = (l == 6) ? m6 : val; val = (l == 7) ? m7 : val; val = (l == 8) ? m8 : val;
      if (l < kGradRec) grad_inst[(size_t)inst * kGradRec + l] = val;
    }
  }
  // every pixel finished before the end of the list: the remaining instances contribute nothing
  for (; base < e; base += 64u) {
    const int n = (int)min(64u, e - base);
    if (l < n) {
      const uint32_t inst = sorted_inst[base + (uint32_t)l];
      float4* p = reinterpret_cast<float4*>(grad_inst + (size_t)inst * kGradRec);
      p[0] = p[1] = p[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// one thread per Gaussian: sum its instance records (fixed order), then the projection backward
__global__ __launch_bounds__(256) void gather_splat_grads(
    CamScalars cs, const float* __restrict__ Vp, const float* __restrict__ PVp, int n,
    const float* __restrict__ means3D, const float* __restrict__ opacities,
    const float* __restrict__ scales, const float* __restrict__ rotations,
    const GaussAux* __restrict__ gaux, const float* __restrict__ grad_inst,
    float* __restrict__ g_means3D, float* __restrict__ g_means2D, float* __restrict__ g_colors,
    float* __restrict__ g_opacities, float* __restrict__ g_scales, float* __restrict__ g_rotations) {
  const CamParams cam = load_cam(cs, Vp, PVp);
  const int gid = (int)(blockIdx.x * 256u + threadIdx.x);
  if (gid >= n) return;
  SplatGrads g;
  for (int i = 0; i < 3; ++i) { g.mean3D[i] = g.mean2D[i] = g.color[i] = g.scale[i] = 0.f; }
  g.opacity = 0.f; g.rot[0] = g.rot[1] = g.rot[2] = g.rot[3] = 0.f;
  const GaussAux ga = gaux[gid];
  if (ga.inst_cnt) {
    SplatMoments mo;
    for (int k = 0; k < 9; ++k) mo.m[k] = 0.f;
    const float4* rec = reinterpret_cast<const float4*>(grad_inst + (size_t)ga.inst_base * kGradRec);
    for (uint32_t i = 0; i < ga.inst_cnt; ++i) {
      const float4 a = rec[3 * i], b = rec[3 * i + 1], c = rec[3 * i + 2];
      mo.m[0] += a.x; mo.m[1] += a.y; mo.m[2] += a.z; mo.m[3] += a.w;
      mo.m[4] += b.x; mo.m[5] += b.y; mo.m[6] += b.z; mo.m[7] += b.w;
      mo.m[8] += c.x;
    }
    const float mean[3] = {means3D[3 * gid], means3D[3 * gid + 1], means3D[3 * gid + 2]};
    const float sc[3] = {scales[3 * gid], scales[3 * gid + 1], scales[3 * gid + 2]};
    const float4 q4 = reinterpret_cast<const float4*>(rotations)[gid];
    const float q[4] = {q4.x, q4.y, q4.z, q4.w};
    const float op = opacities[gid];
    Splat sp; SplatAux aux;
    if (project_splat(cam, mean, sc, q, op, sp, aux)) splat_backward(cam, sc, q, op, sp, aux, mo, g);
  }
  for (int i = 0; i < 3; ++i) {
    g_means3D[3 * gid + i] = g.mean3D[i];
    g_means2D[3 * gid + i] = g.mean2D[i];
    g_colors[3 * gid + i] = g.color[i];
    g_scales[3 * gid + i] = g.scale[i];
  }
  g_opacities[gid] = g.opacity;
  reinterpret_cast<float4*>(g_rotations)[gid] = make_float4(g.rot[0], g.rot[1], g.rot[2], g.rot[3]);
}

__global__ __launch_bounds__(256) void mark_visible_kernel(const float* __restrict__ Vp, int n,
                                                           const float* __restrict__ means3D, uint8_t* __restrict__ out) {
  const int gid = (int)(blockIdx.x * 256u + threadIdx.x);
  if (gid >= n) return;
  const float x = means3D[3 * gid], y = means3D[3 * gid + 1], z = means3D[3 * gid + 2];
  const float tz = fmaf(Vp[2], x, fmaf(Vp[6], y, fmaf(Vp[10], z, Vp[14])));
  out[gid] = tz > kNearCull ? 1 : 0;
}

}  // namespace vtgs
