// vtgs_composite_common.h -- pieces shared by the composite kernels of vtgs_composite.hip (scalar, quad, lane = pixel)
// and vtgs_composite_q.hip (quadrant queues): tile <-> wavefront mapping, the per-chunk gather, the rank-6 exponent.
#pragma once
#include "vtgs_internal.h"

namespace vtgs {

constexpr float kLog2e = 1.4426950408889634f;

struct ChunkRec {     // one splat of the current chunk, held by one lane
  float u, v, qa, qb, qc, op, depth, c0, c1, c2, ulo, vlo;
};

__device__ __forceinline__ ChunkRec gather_chunk(const uint32_t* __restrict__ sorted_gid, const GeomRec* __restrict__ geom,
                                                 const float* __restrict__ colors, uint32_t pos, bool in) {
  ChunkRec r{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (in) {
    const uint32_t gid = sorted_gid[pos];
    const float4* gp = reinterpret_cast<const float4*>(geom + gid);
    const float4 g0 = gp[0], g1 = gp[1];
    r.u = g0.x; r.v = g0.y;
    r.ulo = centre_lo_x(__float_as_uint(g1.w)); r.vlo = centre_lo_y(__float_as_uint(g1.w));
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

template <int WAVES>
__device__ __forceinline__ TileCoord tile_coord(const CamScalars& cs, uint32_t nblk, int gx16, int gx8, int gy8) {
  const uint32_t b = xcd_swizzle(blockIdx.x, nblk);
  const int l = lane_id();
  int t8x, t8y;
  if (WAVES == 4) {          // workgroup = the 2x2 tiles of one 16x16 block
    const int row16_0 = cs.row8_begin >> 1;
    const int t16x = (int)(b % (uint32_t)gx16), t16y = row16_0 + (int)(b / (uint32_t)gx16);
    // the wavefront index is wave-uniform; saying so keeps list bounds and readlane selects in SGPRs
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    t8x = 2 * t16x + (w & 1); t8y = 2 * t16y + (w >> 1);
  } else {                   // workgroup = one wavefront = one 8x8 tile; blocks walk 2x2 groups so neighbours stay close
    const uint32_t g = b >> 2, q = b & 3u;
    const int row16_0 = cs.row8_begin >> 1;
    const int t16x = (int)(g % (uint32_t)gx16), t16y = row16_0 + (int)(g / (uint32_t)gx16);
    t8x = 2 * t16x + (int)(q & 1u); t8y = 2 * t16y + (int)(q >> 1);
  }
  TileCoord tc;
  tc.tile_ok = t8x < gx8 && t8y < gy8 && t8y >= cs.row8_begin && t8y < cs.row8_end;
  tc.tile = t8y * gx8 + t8x;
  tc.px = t8x * kSubTile + (l & 7);
  tc.py = t8y * kSubTile + (l >> 3);
  tc.inside = tc.tile_ok && tc.px < cs.W && tc.py < cs.H;
  return tc;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MxSplat {           // what lane L holds for splat L of the current 64-chunk
  float K[6];
  float4 pay;              // c0 c1 c2 depth          (dual render: c0 c1 c2 c3)
  float2 pay2;             //                          (dual render: c4 c5)
  bool hot;                // opacity above kClampGuard: alpha of this splat can reach the 0.99 clamp
};
// log2(alpha) = log2(o) - (a non-negative quadratic form, up to ~1e-5 of rounding): with o <= 0.98 the min(0.99, .) never
// binds.  A chunk without hot splats takes the clamp-free sweeps below (one instruction less per pair in the forward, four
// in the backward, where alpha_unclamped T == alpha T == w).
constexpr float kClampGuard = 0.98f;

// Coefficients of log2(alpha_unclamped) = sum_m K_m Phi_m(pixel - tile centre) for one splat, from its 32-byte geometry
// record.  Written with explicit fmaf so that every kernel that calls it gets the SAME bits (the quadrant-queue composite is
// tested bit for bit against the lane = pixel one); contraction of the remaining products is switched off.
__device__ __forceinline__ void tile_coefficients(const float4& g0, const float4& g1, float cx, float cy, float (&K)[6]) {
#pragma clang fp contract(off)
  // centre relative to the tile centre: the float32 difference of two nearby numbers is exact (or rounds at the 1e-6 px
  // level for a far splat), then the part of the centre that float32 could not hold (GeomRec::centre_lo)
  const uint32_t lo = __float_as_uint(g1.w);
  const float sx = (g0.x - cx) + centre_lo_x(lo), sy = (g0.y - cy) + centre_lo_y(lo);
  const float qa = -0.5f * kLog2e * g0.z, qb = -kLog2e * g0.w, qc = -0.5f * kLog2e * g1.x;
  K[0] = fmaf(qa * sx, sx, fmaf(qb * sx, sy, fmaf(qc * sy, sy, __log2f(g1.y))));
  K[1] = fmaf(-2.f * qa, sx, -(qb * sy));
  K[2] = fmaf(-2.f * qc, sy, -(qb * sx));
  K[3] = qa; K[4] = qb; K[5] = qc;
}

// DUAL: two renders over the same geometry in one pass -- the second render's colours ride along
// as channels 3..5; the depth image (which the fused caller discards) is not produced.
template <bool DUAL = false>
__device__ __forceinline__ MxSplat mx_gather_gid(uint32_t gid, const GeomRec* __restrict__ geom,
                                                 const float* __restrict__ colors, bool in, float cx, float cy,
                                                 const float* __restrict__ colors_b = nullptr) {
  MxSplat m;
  m.K[0] = -1e30f; m.K[1] = m.K[2] = m.K[3] = m.K[4] = m.K[5] = 0.f;
  m.pay = make_float4(0.f, 0.f, 0.f, 0.f);
  m.pay2 = make_float2(0.f, 0.f);
  m.hot = false;
  if (in) {
    const float4* gp = reinterpret_cast<const float4*>(geom + gid);
    const float4 g0 = gp[0], g1 = gp[1];
    tile_coefficients(g0, g1, cx, cy, m.K);
    m.hot = g1.y > kClampGuard;
    m.pay = make_float4(colors[3 * gid], colors[3 * gid + 1], colors[3 * gid + 2], DUAL ? colors_b[3 * gid] : g1.z);
    if (DUAL) m.pay2 = make_float2(colors_b[3 * gid + 1], colors_b[3 * gid + 2]);
  }
  return m;
}
template <bool DUAL = false>
__device__ __forceinline__ MxSplat mx_gather(const uint32_t* __restrict__ sorted_gid, const GeomRec* __restrict__ geom,
                                             const float* __restrict__ colors, uint32_t pos, bool in, float cx, float cy,
                                             const float* __restrict__ colors_b = nullptr) {
  return mx_gather_gid<DUAL>(in ? sorted_gid[pos] : 0u, geom, colors, in, cx, cy, colors_b);
}

constexpr int kExactFirstEndings = 3;   // pixels (of 64) ending in a batch that make the next batch skip the optimistic sweep

template <int G>
__device__ __forceinline__ f32x4 px_exponents(const float (&K)[6], const float (&Phi)[6]) {
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < 6; ++m) d = __builtin_amdgcn_mfma_f32_4x4x1f32(K[m], Phi[m], d, 4, G, 0);
  return d;
}


}  // namespace vtgs
