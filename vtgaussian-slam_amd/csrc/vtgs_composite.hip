// vtgs_composite.hip -- front-to-back alpha composite (forward) and its backward (gfx950, wave64).
//
// Work decomposition: ONE WAVEFRONT = ONE 8x8 PIXEL TILE; a workgroup is four independent wavefronts (the 2x2 tiles
// of one 16x16 block, XCD-swizzled so neighbours gather the same splats from the same L2); no workgroup barriers.
//
// Two implementations of each direction live here:
//   * matrix-core kernels (composite_forward_mx / composite_backward_mx, default): the exponent of 64 pixels x 16
//     splats is a rank-6 bilinear form evaluated by six 4-block f32 MFMAs; see the block comments below.
//   * scalar kernels (composite_forward / composite_backward, VTGS_FWD_IMPL=0 / VTGS_BWD_IMPL=0): lane = pixel, one
//     splat at a time broadcast with v_readlane.  They are the first correct path of round 1 and stay as an
//     independent implementation the GPU tests cross-check against.
//
// Semantics per pixel: skip power>0; alpha=min(.99,o*exp(power)); skip alpha<1/255;
// stop before adding when T(1-alpha)<1e-4; C+=c*alpha*T; D+=z*alpha*T; out=C+T*bg.  exp is v_exp_f32 (exp2) of a
// pre-scaled quadratic form.
//
// Backward (Appendix A4, restated front-to-back): with g = dL/dcolor at the pixel, Cg = g.(out - T_final*bg)
// and the running prefix P_k = sum_{j<=k} (g.c_j) alpha_j T_j,
//     dL/dalpha_k = T_k (g.c_k) - (Cg - P_k + T_final (g.bg)) / (1 - alpha_k)
// so the list is replayed in the SAME order as the forward (identical skip/stop decisions by construction,
// no per-pixel contributor count to store) and the per-pixel state is two scalars (T, P).  Per splat the nine
// sums over the tile's 64 pixels are formed by an f32 MFMA contraction and stored as one 40-byte record per
// (splat, tile) instance (56 bytes in the dual render) -- plain stores, no float atomics, bitwise reproducible.  gather_splat_grads then
// re-centres and sums each splat's contiguous run of records and runs splat_backward (vtgs_math.h).
#include "vtgs_internal.h"
#include "vtgs_composite_common.h"

// Scheduling groups (profiles/r2_issue_rates.md 2: an MFMA and a vector instruction that alternate one by one cost 1.5x the
// sum of their parts).  Measured on the headline scene: backward sweeps + contraction grouped 190.6 us against 197.8 us
// ungrouped; the same grouping in the lane = pixel FORWARD raises its register pressure and slows it down, so it stays off.
#ifndef VTGS_PX_GROUP
#define VTGS_PX_GROUP 1
#endif
#ifndef VTGS_PX_GROUP2
#define VTGS_PX_GROUP2 1
#endif
#ifndef VTGS_PX_GROUP_FWD
#define VTGS_PX_GROUP_FWD 0
#endif
#ifndef VTGS_BWD_PREFETCH
#define VTGS_BWD_PREFETCH 1
#endif

// Which kernels this translation unit emits.  libvtgs.so (the product) holds the DEFAULT composites only -- the quadrant-queue
// forward (vtgs_composite_q.hip), the lane = pixel backward, the gradient gather; the other implementations (scalar, quad form,
// lane = pixel forward, quadrant-queue backward) are independent cross-checks for the tests and live in the test-only
// libvtgs_xcheck.so, built from the same sources with -DVTGS_XCHECK_BUILD=1 (csrc/vtgs_xcheck.hip).
#ifndef VTGS_XCHECK_BUILD
#define VTGS_XCHECK_BUILD 0
#endif

namespace vtgs {

#if VTGS_XCHECK_BUILD
// Scalar ("readlane") forward composite: lane = pixel, splats broadcast one at a time.  Kept as an independent
// second implementation (VTGS_FWD_IMPL=0) that the GPU tests cross-check against the matrix-core kernel.
__global__ __launch_bounds__(256) void composite_forward(
    CamScalars cs, const float* __restrict__ bg, uint32_t nblk,
    const uint32_t* __restrict__ tile_cnt, uint32_t tile_cap, const uint32_t* __restrict__ sorted_gid,
    const GeomRec* __restrict__ geom, const float* __restrict__ colors,
    float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ final_T,
    const Counters* __restrict__ ctr) {
  if (ctr->overflow) return;                     // bins hold unwritten slots after an overflow
  const int gx16 = (cs.W + kBinTile - 1) / kBinTile;
  const int gx8 = (cs.W + kSubTile - 1) / kSubTile, gy8 = (cs.H + kSubTile - 1) / kSubTile;
  const TileCoord tc = tile_coord<4>(cs, nblk, gx16, gx8, gy8);
  if (!tc.tile_ok) return;                       // wave-uniform
  const int l = lane_id();
  const float pxf = (float)tc.px, pyf = (float)tc.py;
  const BinRange br = bin_range(cs, (uint32_t)tc.tile, tile_cap);
  const uint32_t s = br.s, e = s + min(tile_cnt[tc.tile], br.cap);

  float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
  bool done = !tc.inside;
  for (uint32_t base = s; base < e; base += 64u) {
    if (__ballot(!done) == 0ull) break;
    const int n = (int)min(64u, e - base);
    const ChunkRec r = gather_chunk(sorted_gid, geom, colors, base + (uint32_t)l, l < n);
    for (int j = 0; j < n; ++j) {
      const float dx = (bcast_f(r.u, j) - pxf) + bcast_f(r.ulo, j), dy = (bcast_f(r.v, j) - pyf) + bcast_f(r.vlo, j);
      const float p2 = dx * (bcast_f(r.qa, j) * dx + bcast_f(r.qb, j) * dy) + bcast_f(r.qc, j) * dy * dy;
      const float alpha = fminf(kAlphaMax, bcast_f(r.op, j) * __builtin_amdgcn_exp2f(p2));
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

#endif  // VTGS_XCHECK_BUILD

// ---------------------------------------------------------------------------------------------------
// Forward composite, matrix-core form ("mx").
//
// exponent[pixel p, splat k] = log2(alpha_unclamped) = sum_m K_m(k) * Phi_m(p) is bilinear of rank 6:
//     Phi = (1, X, Y, X^2, XY, Y^2)  with X,Y = pixel - tile centre
//     K   = (qa sx^2 + qb sx sy + qc sy^2 + log2 o,  -2 qa sx - qb sy,  -2 qc sy - qb sx,  qa, qb, qc),  s = splat centre - tile centre
// so 64 pixels x 16 splats come out of SIX v_mfma_f32_16x16x1_4b_f32 (4 blocks of 16x16, exact f32 fma chains).
// With cbsz=2/abid=b the A operand of all four blocks is taken from lanes 16b..16b+15: those lanes' OWN coefficient
// registers (lane L of the wavefront gathered splat L of the 64-chunk) -- no broadcast instruction at all.
// B is each lane's own pixel monomials (lane L <-> pixel L of the 8x8 tile), constant for the tile.
// Result layout (measured, tests/micro/mfma_layout.hip): lane (j = L&15, q = L>>4), register 4*blk + r holds
//     pixel 16*blk + j,  splat 16*b + 4*q + r.
// Each lane therefore owns 4 pixels x 4 splats per batch; the per-splat payload (colour, depth) is read per quad
// from LDS, the transmittance is chained across the four quad-lanes of a pixel with one LDS exchange per batch
// (T_in = T_batch * prod of the earlier quads' local products), colour sums stay per lane and are reduced across
// the quad-lanes once per tile.  The stop rule T(1-alpha) < 1e-4 is evaluated exactly on a wave-uniform slow path
// that is taken only for batches in which some pixel's transmittance actually crosses 1e-4.
// The `power > 0` skip of the scalar form cannot trigger for a positive-definite conic and is not evaluated here
// (the exponent carries ~1e-5 absolute rounding from the expansion; alpha is still clamped to 0.99).
// ---------------------------------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ f32x16 mx_exponents(const float (&K)[6], const float (&Phi)[6]) {
  f32x16 d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < 6; ++m) d = __builtin_amdgcn_mfma_f32_16x16x1f32(K[m], Phi[m], d, 2, B, 0);
  return d;
}

#if VTGS_XCHECK_BUILD
template <bool DUAL>
struct MxFwdState {
  float Tb[4];             // transmittance of pixel (blk, j) at the start of the batch; 0 = finished (replicated over q)
  float Tfin[4];           // set by the lane that stopped the pixel: T before the stopping splat
  float C[4][DUAL ? 6 : 4];   // this lane's share of colour (3) + depth sums of pixel (blk, j); dual: 6 colour sums
};

template <int B, bool DUAL>
__device__ __forceinline__ void mx_forward_batch(MxFwdState<DUAL>& st, const float (&K)[6], const float (&Phi)[6],
                                                 const float4* __restrict__ lds_pay, float4* __restrict__ lds_xch,
                                                 int l, const float2* __restrict__ lds_pay2 = nullptr) {
  const int j = l & 15, q = l >> 4;
  const f32x16 d = mx_exponents<B>(K, Phi);
  float4 pay[4];
  float2 pay2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    pay[r] = lds_pay[16 * B + 4 * q + r];
    if (DUAL) pay2[r] = lds_pay2[16 * B + 4 * q + r];
  }
  const unsigned long long lower_q = 0x0001000100010001ull & ((q == 0) ? 0ull : ((1ull << (16 * q)) - 1ull));
  float2* xch2 = reinterpret_cast<float2*>(lds_xch);
  // two half-passes over the pixel groups {0,1} and {2,3} (keeps the live register set small)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float a[2][4], pl[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int blk = 2 * h + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float al = fminf(kAlphaMax, __builtin_amdgcn_exp2f(d[4 * blk + r]));
        a[i][r] = (al >= kAlphaMin) ? al : 0.f;
      }
      pl[i][0] = 1.f - a[i][0];
      pl[i][1] = pl[i][0] * (1.f - a[i][1]);
      pl[i][2] = pl[i][1] * (1.f - a[i][2]);
      pl[i][3] = pl[i][2] * (1.f - a[i][3]);
    }
    float2* xch = xch2 + 64 * h;
    xch[l] = make_float2(pl[0][3], pl[1][3]);
    const float2 x0 = xch[j], x1 = xch[16 + j], x2 = xch[32 + j], x3 = xch[48 + j];
    float Tin[2], Tend[2];
    bool cross = false;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int blk = 2 * h + i;
      const float P0 = i ? x0.y : x0.x, P1 = i ? x1.y : x1.x, P2 = i ? x2.y : x2.x, P3 = i ? x3.y : x3.x;
      const float e1 = P0, e2 = e1 * P1, e3 = e2 * P2;
      const float pre = (q == 0) ? 1.f : (q == 1) ? e1 : (q == 2) ? e2 : e3;
      Tin[i] = st.Tb[blk] * pre;
      Tend[i] = st.Tb[blk] * (e3 * P3);
      cross = cross || (st.Tb[blk] > 0.f && Tend[i] < kTStop);
    }
    const bool slow = __ballot(cross) != 0ull;                 // wave-uniform; rarely true
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int blk = 2 * h + i;
      const float t0 = Tin[i], t1 = t0 * pl[i][0], t2 = t0 * pl[i][1], t3 = t0 * pl[i][2];
      const float t[4] = {t0, t1, t2, t3};
      float w[4] = {a[i][0] * t0, a[i][1] * t1, a[i][2] * t2, a[i][3] * t3};
      const bool was_alive = st.Tb[blk] > 0.f;
      bool pixel_stopped = false;
      if (slow) {
        // exact stop rule: the first (quad, splat) in list order with T*(1-alpha) < 1e-4 ends the pixel, before adding
        bool livep = true, any = false;
        float tstop = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool stop = a[i][r] > 0.f && Tin[i] * pl[i][r] < kTStop;
          if (livep && stop) { tstop = t[r]; any = true; }
          livep = livep && !stop;
          w[r] = livep ? w[r] : 0.f;
        }
        const unsigned long long bal = __ballot(any && was_alive) >> j;
        const bool mine = was_alive && (bal & lower_q) == 0ull;    // no quad in front of mine ended the pixel
        pixel_stopped = (bal & 0x0001000100010001ull) != 0ull;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = mine ? w[r] : 0.f;
        if (mine && any) st.Tfin[blk] = tstop;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st.C[blk][0] = fmaf(w[r], pay[r].x, st.C[blk][0]);
        st.C[blk][1] = fmaf(w[r], pay[r].y, st.C[blk][1]);
        st.C[blk][2] = fmaf(w[r], pay[r].z, st.C[blk][2]);
        st.C[blk][3] = fmaf(w[r], pay[r].w, st.C[blk][3]);
        if constexpr (DUAL) {
          st.C[blk][4] = fmaf(w[r], pay2[r].x, st.C[blk][4]);
          st.C[blk][5] = fmaf(w[r], pay2[r].y, st.C[blk][5]);
        }
      }
      st.Tb[blk] = (was_alive && !pixel_stopped) ? Tend[i] : 0.f;
    }
  }
}

template <int WAVES, bool DUAL>
__global__ __launch_bounds__(64 * WAVES, DUAL ? 3 : 4) void composite_forward_mx(
    CamScalars cs, const float* __restrict__ bg, uint32_t nblk,
    const uint32_t* __restrict__ tile_cnt, uint32_t tile_cap, const uint32_t* __restrict__ sorted_gid,
    const GeomRec* __restrict__ geom, const float* __restrict__ colors,
    float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ final_T,
    const Counters* __restrict__ ctr, const float* __restrict__ colors_b, float* __restrict__ out_color_b) {
  constexpr int NC = DUAL ? 6 : 4;                          // accumulated channels; + 1 row for T_final in the reduction
  __shared__ float4 lds_pay_all[WAVES][64];
  __shared__ float2 lds_pay2_all[DUAL ? WAVES : 1][DUAL ? 64 : 1];
  __shared__ float4 lds_xch_all[WAVES][64];
  __shared__ float lds_red_all[WAVES][(NC + 1) * 64 * 4];   // [value 0..NC][q][pixel 0..63]
  if (ctr->overflow) return;                                // bins hold unwritten slots after an overflow
  const int gx16 = (cs.W + kBinTile - 1) / kBinTile;
  const int gx8 = (cs.W + kSubTile - 1) / kSubTile, gy8 = (cs.H + kSubTile - 1) / kSubTile;
  const TileCoord tc = tile_coord<WAVES>(cs, nblk, gx16, gx8, gy8);   // lane L <-> pixel L of the tile (x = L&7, y = L>>3)
  if (!tc.tile_ok) return;
  const int l = lane_id();
  const int wv = (WAVES == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float4* lds_pay = lds_pay_all[wv];
  float2* lds_pay2 = lds_pay2_all[DUAL ? wv : 0];
  float4* lds_xch = lds_xch_all[wv];
  float* lds_red = lds_red_all[wv];
  const int j = l & 15, q = l >> 4;
  const int tx0 = tc.px - (l & 7), ty0 = tc.py - (l >> 3);
  const float cx = (float)tx0 + 3.5f, cy = (float)ty0 + 3.5f;
  const float X = (float)(l & 7) - 3.5f, Y = (float)(l >> 3) - 3.5f;
  const float Phi[6] = {1.f, X, Y, X * X, X * Y, Y * Y};
  const BinRange br = bin_range(cs, (uint32_t)tc.tile, tile_cap);
  const uint32_t s = br.s, e = s + min(tile_cnt[tc.tile], br.cap);

  MxFwdState<DUAL> st;
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    const int p = 16 * blk + j;                               // pixel (blk, j) of the tile
    const bool in_img = (tx0 + (p & 7)) < cs.W && (ty0 + (p >> 3)) < cs.H;
    st.Tb[blk] = in_img ? 1.f : 0.f;
    st.Tfin[blk] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) st.C[blk][c] = 0.f;
  }
  for (uint32_t base = s; base < e; base += 64u) {
    const bool alive = st.Tb[0] > 0.f || st.Tb[1] > 0.f || st.Tb[2] > 0.f || st.Tb[3] > 0.f;
    if (__ballot(alive) == 0ull) break;
    const int n = (int)min(64u, e - base);
    const MxSplat m = mx_gather<DUAL>(sorted_gid, geom, colors, base + (uint32_t)l, l < n, cx, cy, colors_b);
    lds_pay[l] = m.pay;
    if (DUAL) lds_pay2[l] = m.pay2;
    mx_forward_batch<0, DUAL>(st, m.K, Phi, lds_pay, lds_xch, l, lds_pay2);
    if (n > 16) mx_forward_batch<1, DUAL>(st, m.K, Phi, lds_pay, lds_xch, l, lds_pay2);
    if (n > 32) mx_forward_batch<2, DUAL>(st, m.K, Phi, lds_pay, lds_xch, l, lds_pay2);
    if (n > 48) mx_forward_batch<3, DUAL>(st, m.K, Phi, lds_pay, lds_xch, l, lds_pay2);
  }
  // reduce the four quad-lanes of every pixel: lane L outputs pixel L = (blk = L>>4, j = L&15)
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    const int p = 16 * blk + j;
#pragma unroll
    for (int c = 0; c < NC; ++c) lds_red[(c * 4 + q) * 64 + p] = st.C[blk][c];
    lds_red[(NC * 4 + q) * 64 + p] = st.Tfin[blk];
  }
  float out[NC + 1];
#pragma unroll
  for (int c = 0; c < NC + 1; ++c)
    out[c] = lds_red[(c * 4 + 0) * 64 + l] + lds_red[(c * 4 + 1) * 64 + l] + lds_red[(c * 4 + 2) * 64 + l] + lds_red[(c * 4 + 3) * 64 + l];
  const float Tb_mine = (q == 0) ? st.Tb[0] : (q == 1) ? st.Tb[1] : (q == 2) ? st.Tb[2] : st.Tb[3];   // pixel L has blk == q
  const float T = (Tb_mine > 0.f) ? Tb_mine : out[NC];
  if (tc.inside) {
    const size_t P = (size_t)cs.W * cs.H, pix = (size_t)tc.py * cs.W + tc.px;
    out_color[pix] = out[0] + T * bg[0];
    out_color[P + pix] = out[1] + T * bg[1];
    out_color[2 * P + pix] = out[2] + T * bg[2];
    if constexpr (DUAL) {
      out_color_b[pix] = out[3] + T * bg[0];
      out_color_b[P + pix] = out[4] + T * bg[1];
      out_color_b[2 * P + pix] = out[5] + T * bg[2];
    } else {
      out_depth[pix] = out[3];
    }
    final_T[pix] = T;
  }
}
template __global__ void composite_forward_mx<4, false>(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*, const GeomRec*, const float*, float*, float*, float*, const Counters*, const float*, float*);
template __global__ void composite_forward_mx<4, true>(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*, const GeomRec*, const float*, float*, float*, float*, const Counters*, const float*, float*);
#endif  // VTGS_XCHECK_BUILD


#if VTGS_XCHECK_BUILD
// ---------------------------------------------------------------------------------------------------
// Forward composite, lane = pixel ("px").  Same bilinear exponent, but from v_mfma_f32_4x4x1_16b_f32 with the A operand
// broadcast (cbsz = 4, abid = g): all 16 blocks multiply the four splats held by lanes 4g..4g+3 with their own four
// pixels, so lane L receives the exponents of splats 4g..4g+3 at ITS pixel L -- four MFMAs per rank-1 term give a lane
// one pixel x 16 splats (same MAC count as the 4-block form above, layout checked by tests/micro/mfma_layout.hip).
// The transmittance chain then runs inside the lane: no quad exchange, no per-lane selects, no reduction at the end;
// the per-splat payload is an LDS broadcast read.  A batch of 16 is first composited without the stop test; only if
// some pixel of the wavefront ends inside it (T(1 - alpha) < 1e-4) is the batch redone with the exact per-splat rule
// (wave-uniform branch, rare).  Per pixel this is the scalar kernel's recurrence: w = alpha T, C += w c, T' = T - w.
// ---------------------------------------------------------------------------------------------------
// State of a pixel: T = transmittance in front of the next splat, FROZEN once the pixel has ended (then it is the value
// the reference reports as final T: the transmittance in front of the ending splat); `done` marks ended / off-image pixels.
// `exact` (wave-uniform): after a batch in which several pixels ended, the next batch goes straight to the exact sweep, and
// keeps doing so while pixels keep ending -- in saturating scenes they end batch after batch, and optimistic-then-redo
// would cost more; where endings are sparse the optimistic sweep stays the first choice.
template <int B, bool DUAL, bool CLAMP, bool EXACT_FIRST>
__device__ __forceinline__ void px_forward_batch(float& T, bool& done, bool& exact, float (&C)[DUAL ? 6 : 4],
                                                 const float (&K)[6], const float (&Phi)[6],
                                                 const float4* __restrict__ lds_pay, const float2* __restrict__ lds_pay2) {
  constexpr int NC = DUAL ? 6 : 4;
  const f32x4 d[4] = {px_exponents<4 * B>(K, Phi), px_exponents<4 * B + 1>(K, Phi), px_exponents<4 * B + 2>(K, Phi),
                      px_exponents<4 * B + 3>(K, Phi)};
  auto alpha = [&](int k) {
    const float g = __builtin_amdgcn_exp2f(d[k >> 2][k & 3]);
    const float al = CLAMP ? fminf(kAlphaMax, g) : g;           // CLAMP == false: no splat of the chunk can reach 0.99
    return (al >= kAlphaMin) ? al : 0.f;
  };
  if constexpr (!EXACT_FIRST) {
    // optimistic sweep: no stop test
    float Tn = done ? 0.f : T, Cn[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) Cn[c] = C[c];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float ak = alpha(k);
      const float4 py = lds_pay[16 * B + k];                     // same address in every lane: LDS broadcast
      const float w = ak * Tn;
      Cn[0] = fmaf(w, py.x, Cn[0]); Cn[1] = fmaf(w, py.y, Cn[1]); Cn[2] = fmaf(w, py.z, Cn[2]); Cn[3] = fmaf(w, py.w, Cn[3]);
      if constexpr (DUAL) {
        const float2 p2 = lds_pay2[16 * B + k];
        Cn[4] = fmaf(w, p2.x, Cn[4]); Cn[5] = fmaf(w, p2.y, Cn[5]);
      }
      Tn = Tn - w;                                               // T (1 - alpha), with the product already at hand
    }
#if VTGS_PX_GROUP_FWD
    // keep the 24 exponent MFMAs together, ahead of the vector sweep (tests/micro/mix_rate.hip: MFMA and vector
    // instructions alternating one by one cost 1.5x the sum of their separate issue times)
    __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
    __builtin_amdgcn_sched_group_barrier(0x102, 400, 0);
#endif
    if (__ballot(!done && Tn < kTStop) == 0ull) {                // nobody ends inside this batch
      T = done ? T : Tn;
#pragma unroll
      for (int c = 0; c < NC; ++c) C[c] = Cn[c];
      return;
    }
  }
  // exact sweep: the first splat with T (1 - alpha) < 1e-4 ends the pixel BEFORE it is added.  A live pixel has
  // T >= 1e-4, so an invalid pair (alpha = 0) can never trigger the test.
  const bool was_done = done;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float ak = alpha(k);
    const float wk = ak * T;
    const float tn = T - wk;                                     // the same expression as in the optimistic sweep
    const bool stop = tn < kTStop;
    const bool live = !done && !stop;
    const float4 py = lds_pay[16 * B + k];
    const float w = live ? wk : 0.f;
    C[0] = fmaf(w, py.x, C[0]); C[1] = fmaf(w, py.y, C[1]); C[2] = fmaf(w, py.z, C[2]); C[3] = fmaf(w, py.w, C[3]);
    if constexpr (DUAL) {
      const float2 p2 = lds_pay2[16 * B + k];
      C[4] = fmaf(w, p2.x, C[4]); C[5] = fmaf(w, p2.y, C[5]);
    }
    T = live ? tn : T;
    done = done || stop;
  }
  // several pixels ended in this batch: a saturation front is passing, the next batch will very likely end more
  // (exact-first costs 14 instead of 10 instructions per pair, optimistic + redo 24: worth it above ~30 % odds)
  exact = __builtin_popcountll(__ballot(done && !was_done)) >= kExactFirstEndings;
}

template <int WAVES, bool DUAL>
__global__ __launch_bounds__(64 * WAVES, 3) void composite_forward_px(
    CamScalars cs, const float* __restrict__ bg, uint32_t nblk,
    const uint32_t* __restrict__ tile_cnt, uint32_t tile_cap, const uint32_t* __restrict__ sorted_gid,
    const GeomRec* __restrict__ geom, const float* __restrict__ colors,
    float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ final_T,
    const Counters* __restrict__ ctr, const float* __restrict__ colors_b, float* __restrict__ out_color_b) {
  constexpr int NC = DUAL ? 6 : 4;
  __shared__ float4 lds_pay_all[WAVES][64];
  __shared__ float2 lds_pay2_all[DUAL ? WAVES : 1][DUAL ? 64 : 1];
  if (ctr->overflow) return;                                // bins hold unwritten slots after an overflow
  const int gx16 = (cs.W + kBinTile - 1) / kBinTile;
  const int gx8 = (cs.W + kSubTile - 1) / kSubTile, gy8 = (cs.H + kSubTile - 1) / kSubTile;
  const TileCoord tc = tile_coord<WAVES>(cs, nblk, gx16, gx8, gy8);   // lane L <-> pixel L of the tile (x = L&7, y = L>>3)
  if (!tc.tile_ok) return;
  const int l = lane_id();
  const int wv = (WAVES == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float4* lds_pay = lds_pay_all[wv];
  float2* lds_pay2 = lds_pay2_all[DUAL ? wv : 0];
  const int tx0 = tc.px - (l & 7), ty0 = tc.py - (l >> 3);
  const float cx = (float)tx0 + 3.5f, cy = (float)ty0 + 3.5f;
  const float X = (float)(l & 7) - 3.5f, Y = (float)(l >> 3) - 3.5f;
  const float Phi[6] = {1.f, X, Y, X * X, X * Y, Y * Y};
  const BinRange br = bin_range(cs, (uint32_t)tc.tile, tile_cap);
  const uint32_t s = br.s, e = s + min(tile_cnt[tc.tile], br.cap);

  float T = 1.f, C[NC];
  bool done = !tc.inside, exact = false;
#pragma unroll
  for (int c = 0; c < NC; ++c) C[c] = 0.f;
  for (uint32_t base = s; base < e; base += 64u) {
    if (__ballot(!done) == 0ull) break;
    const int n = (int)min(64u, e - base);
    const MxSplat m = mx_gather<DUAL>(sorted_gid, geom, colors, base + (uint32_t)l, l < n, cx, cy, colors_b);
    lds_pay[l] = m.pay;
    if (DUAL) lds_pay2[l] = m.pay2;
    // wave-uniform choices per chunk: clamp-free sweeps unless some splat may reach the 0.99 clamp; exact sweep first
    // while pixels keep ending (set by the batches themselves)
    const bool hot = __ballot(m.hot) != 0ull;
#define VTGS_PX_FWD(CL, EX)                                                                           \
    {                                                                                                 \
      px_forward_batch<0, DUAL, CL, EX>(T, done, exact, C, m.K, Phi, lds_pay, lds_pay2);              \
      if (n > 16) px_forward_batch<1, DUAL, CL, EX>(T, done, exact, C, m.K, Phi, lds_pay, lds_pay2);  \
      if (n > 32) px_forward_batch<2, DUAL, CL, EX>(T, done, exact, C, m.K, Phi, lds_pay, lds_pay2);  \
      if (n > 48) px_forward_batch<3, DUAL, CL, EX>(T, done, exact, C, m.K, Phi, lds_pay, lds_pay2);  \
    }
    if (exact) VTGS_PX_FWD(true, true)                        // (one exact-first body: the clamped form is always valid)
    else       { if (hot) VTGS_PX_FWD(true, false) else VTGS_PX_FWD(false, false) }
#undef VTGS_PX_FWD
  }
  const float Tout = T;                                     // running value, or frozen in front of the ending splat
  if (tc.inside) {
    const size_t P = (size_t)cs.W * cs.H, pix = (size_t)tc.py * cs.W + tc.px;
    out_color[pix] = C[0] + Tout * bg[0];
    out_color[P + pix] = C[1] + Tout * bg[1];
    out_color[2 * P + pix] = C[2] + Tout * bg[2];
    if constexpr (DUAL) {
      out_color_b[pix] = C[3] + Tout * bg[0];
      out_color_b[P + pix] = C[4] + Tout * bg[1];
      out_color_b[2 * P + pix] = C[5] + Tout * bg[2];
    } else {
      out_depth[pix] = C[3];
    }
    final_T[pix] = Tout;
  }
}
template __global__ void composite_forward_px<4, false>(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*, const GeomRec*, const float*, float*, float*, float*, const Counters*, const float*, float*);
template __global__ void composite_forward_px<4, true>(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*, const GeomRec*, const float*, float*, float*, float*, const Counters*, const float*, float*);
#endif  // VTGS_XCHECK_BUILD


#if VTGS_XCHECK_BUILD
// ---------------------------------------------------------------------------------------------------
// Backward composite.  Pixel-major replay produces, per (splat k, pixel p), two scalars:
//     u_kp = G_kp * dL/dalpha_kp      and      w_kp = alpha_kp * T_kp
// and the nine per-splat sums are a contraction over the 64 pixels of the tile:
//     S[k][0..5] = sum_p u_kp * phi_j(p),  phi = (1, X, Y, X^2, XY, Y^2),  X,Y = pixel - tile centre
//     S[k][6..8] = sum_p w_kp * g_ch(p)                                   (g = dL/dcolor)
// i.e. [16 splats x 64 px] x [64 px x 9] per batch of 16 splats -- run on the matrix cores with the exact-f32
// MFMA (v_mfma_f32_16x16x4_f32), which is otherwise idle and issues beside the VALU work of the co-resident
// wavefronts.  The lane<->pixel layout of the replay is transposed into the MFMA A-operand layout
// (lane = splat row i + 16 * k-slot) through a per-wavefront LDS image [16][65] (stride 65: conflict-free for
// both the row-wise ds_write_b32 and the column-wise ds_read_b32).  B (phi | g) lives in 16 VGPRs per lane for the
// whole tile.  The tile-local moments are re-centred on the splat in gather_splat_grads (the record carries the
// tile id), so nothing here depends on the splat position and there is no cross-lane reduction at all.
// ---------------------------------------------------------------------------------------------------
constexpr int kRowStride = 65;

__global__ __launch_bounds__(256) void composite_backward(
    CamScalars cs, const float* __restrict__ bg, uint32_t nblk16,
    const uint32_t* __restrict__ tile_cnt, uint32_t tile_cap, const uint32_t* __restrict__ sorted_gid,
    const uint32_t* __restrict__ sorted_inst, const GeomRec* __restrict__ geom, const float* __restrict__ colors,
    const float* __restrict__ out_color, const float* __restrict__ grad_color, const float* __restrict__ final_T,
    float* __restrict__ grad_inst, const Counters* __restrict__ ctr) {
  __shared__ float lds[4][2][16 * kRowStride];
  if (ctr->overflow) return;
  const int gx16 = (cs.W + kBinTile - 1) / kBinTile;
  const int gx8 = (cs.W + kSubTile - 1) / kSubTile, gy8 = (cs.H + kSubTile - 1) / kSubTile;
  const TileCoord tc = tile_coord<4>(cs, nblk16, gx16, gx8, gy8);
  if (!tc.tile_ok) return;
  const int l = lane_id();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* __restrict__ Us = lds[wv][0];
  float* __restrict__ Ws = lds[wv][1];
  const float pxf = (float)tc.px, pyf = (float)tc.py;
  const BinRange br = bin_range(cs, (uint32_t)tc.tile, tile_cap);
  const uint32_t s = br.s, e = s + min(tile_cnt[tc.tile], br.cap);
  if (s == e) return;
  const size_t P = (size_t)cs.W * cs.H;

  float g0 = 0.f, g1 = 0.f, g2 = 0.f, Cg = 0.f, Bg = 0.f;
  if (tc.inside) {
    const size_t pix = (size_t)tc.py * cs.W + tc.px;
    g0 = grad_color[pix]; g1 = grad_color[P + pix]; g2 = grad_color[2 * P + pix];
    const float Tf = final_T[pix];
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    Cg = g0 * (out_color[pix] - Tf * b0) + g1 * (out_color[P + pix] - Tf * b1) + g2 * (out_color[2 * P + pix] - Tf * b2);
    Bg = Tf * (g0 * b0 + g1 * b1 + g2 * b2);
  }
  // B operand: lane (j = l&15, kk = l>>4), step t  <->  pixel p = 16*kk + t of the tile (p = lane index of the replay)
  const int bj = l & 15, bk = l >> 4;
  const int tx0 = tc.px - (l & 7), ty0 = tc.py - (l >> 3);          // tile origin
  float Bv[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int p = 16 * bk + t;
    const float X = (float)(p & 7) - 3.5f, Y = (float)(p >> 3) - 3.5f;
    float v = 0.f;
    v = (bj == 0) ? 1.f : v; v = (bj == 1) ? X : v; v = (bj == 2) ? Y : v;
    v = (bj == 3) ? X * X : v; v = (bj == 4) ? X * Y : v; v = (bj == 5) ? Y * Y : v;
    if (bj >= 6 && bj <= 8) {
      const int qx = tx0 + (p & 7), qy = ty0 + (p >> 3);
      if (qx < cs.W && qy < cs.H) v = grad_color[(size_t)(bj - 6) * P + (size_t)qy * cs.W + qx];
    }
    Bv[t] = v;
  }
  const int a_off = bj * kRowStride + 16 * bk;                         // A operand: row = splat bj, k-slot bk
  const uint32_t tile_bits = (uint32_t)tc.tile;

  float T = 1.f, Pfx = 0.f;
  bool done = !tc.inside;
  uint32_t base = s;
  for (; base < e; base += 64u) {
    if (__ballot(!done) == 0ull) break;
    const int n = (int)min(64u, e - base);
    const ChunkRec r = gather_chunk(sorted_gid, geom, colors, base + (uint32_t)l, l < n);
    const uint32_t my_inst = (l < n) ? sorted_inst[base + (uint32_t)l] : 0u;
    for (int jb = 0; jb < n; jb += 16) {
      const int nb = min(16, n - jb);
      for (int jj = 0; jj < nb; ++jj) {
        const int j = jb + jj;
        const float su = bcast_f(r.u, j), sv = bcast_f(r.v, j);
        const float sa = bcast_f(r.qa, j), sb = bcast_f(r.qb, j), sc = bcast_f(r.qc, j);
        const float so = bcast_f(r.op, j);
        const float dx = (su - pxf) + bcast_f(r.ulo, j), dy = (sv - pyf) + bcast_f(r.vlo, j);
        const float p2 = dx * (sa * dx + sb * dy) + sc * dy * dy;
        const float G = __builtin_amdgcn_exp2f(p2);
        const float alpha = fminf(kAlphaMax, so * G);
        const float Tn = T * (1.f - alpha);
        bool hit = !done && p2 <= 0.f && alpha >= kAlphaMin;
        if (hit && Tn < kTStop) { done = true; hit = false; }
        float uu = 0.f, ww = 0.f;
        if (__ballot(hit) != 0ull) {
          const float c0 = bcast_f(r.c0, j), c1 = bcast_f(r.c1, j), c2 = bcast_f(r.c2, j);
          const float gc = g0 * c0 + g1 * c1 + g2 * c2;
          const float wgt = alpha * T;
          const float Pn = fmaf(gc, wgt, Pfx);
          const float dLda = T * gc - (Cg - Pn + Bg) * __builtin_amdgcn_rcpf(1.f - alpha);
          uu = hit ? G * dLda : 0.f;            // clamp at 0.99 passes the gradient through
          ww = hit ? wgt : 0.f;
          Pfx = hit ? Pn : Pfx;
          T = hit ? Tn : T;
        }
        Us[jj * kRowStride + l] = uu;
        Ws[jj * kRowStride + l] = ww;
      }
      // contraction over the 64 pixels on the matrix cores
      f32x4 D1 = {0.f, 0.f, 0.f, 0.f}, D2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        D1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Us[a_off + t], Bv[t], D1, 0, 0, 0);
        D2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ws[a_off + t], Bv[t], D2, 0, 0, 0);
      }
      // D: col = l&15, row = 4*(l>>4) + reg.  Column j<6 from D1 (u x phi), 6..8 from D2 (w x g), 9 = tile id.
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = 4 * bk + rr;
        const uint32_t inst = (uint32_t)__shfl((int)my_inst, jb + row, 64);
        float val = (bj < 6) ? D1[rr] : D2[rr];
        val = (bj == 9) ? __uint_as_float(tile_bits) : val;
        val = (bj > 9) ? 0.f : val;
        if (row < nb && bj < kGradRec) grad_inst[(size_t)inst * kGradRec + bj] = val;
      }
    }
  }
  // every pixel finished before the end of the list: the remaining instances contribute nothing
  for (; base < e; base += 64u) {
    const int n = (int)min(64u, e - base);
    if (l < n) {
      const uint32_t inst = sorted_inst[base + (uint32_t)l];
      float2* p = reinterpret_cast<float2*>(grad_inst + (size_t)inst * kGradRec);
      p[0] = p[1] = p[2] = p[3] = make_float2(0.f, 0.f);
      p[4] = make_float2(0.f, __uint_as_float(tile_bits));
    }
  }
}

#endif  // VTGS_XCHECK_BUILD

// ---------------------------------------------------------------------------------------------------
// Backward composite, matrix-core form.  Same lane layout as composite_forward_mx (lane = pixel column j x splat
// quad q, exponents from six 4-block MFMAs).  Besides the transmittance, the gradient prefix
//     P_k = sum_{i<=k} (g.c_i) alpha_i T_i      is chained across the quad-lanes:  P_in(q) = P_batch + sum_{q'<q} T_in(q') S(q')
// with S the T-free local sum.  Per (pixel, splat) the lane produces u' = alpha_unclamped * dL/dalpha (= o * u of the
// scalar form; the gather kernel divides the six geometric sums by o) and w = alpha*T and writes them into two
// per-wavefront LDS images laid out [pixel quarter][splat][16 pixels + 4 pad].
//
// The contraction over the 64 pixels runs on v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 blocks = 4 splat groups x
// 4 pixel quarters, each accumulating [4 splats] x [4 columns] over its 16 pixels -- one chain per column group
// (Phi0..3 | Phi4,5 | g0..2 [| g3..5]).  f32 MFMAs share the fp32 FMA lanes with the VALU on CDNA (same peak, no
// co-execution: measured by ablation), so what counts is MACs issued: the 16x16x4 form spends
// 32 instructions x 1024 MACs on 9 useful columns of 16, this form 48 x 256 with 9 (12) of 12 (16) useful.  The four
// pixel-quarter partials are summed across the 16-lane rows with v_permlane32_swap / v_permlane16_swap butterflies
// (two values per swap), which leaves lane (column c, splat group sg, row rho) holding the total of splat 4 sg + rho.
// ---------------------------------------------------------------------------------------------------
constexpr int kImgRow = 20;                 // floats per (splat, pixel quarter) row: 16 pixels + 4 pad (bank spread)
constexpr int kImgQuarter = 16 * kImgRow;   // 320 floats = 5 x 64 banks: the quarter does not move the bank
constexpr int kImgFloats = 4 * kImgQuarter; // one image (u' or w) per wavefront
constexpr int kPhiQuarter = 8 * kImgRow;    // Phi table [pixel quarter][8 columns][16 + 4 pad], shared by the workgroup

__device__ __forceinline__ float swap32_add(float a, float b) {    // rows (a0+a2, a1+a3, b0+b2, b1+b3)
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float a, float b) {    // rows (a0+a1, b0+b1, a2+a3, b2+b3)
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over the four 16-lane rows of the four registers of one chain: row rho of the result = total of register rho
__device__ __forceinline__ float quarter_sum(const f32x4& p) {
  return swap16_add(swap32_add(p[0], p[2]), swap32_add(p[1], p[3]));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Per-pixel state; the two components of every f32x2 are the pixel blocks 2h and 2h+1 of half-pass h, so the
// element-wise arithmetic below compiles to packed v_pk_{mul,add,fma}_f32 (two pixels per instruction).
template <int NG>
struct MxBwdState {
  f32x2 Tb[2], Pb[2];      // transmittance / gradient prefix of pixel (blk, j) at the start of the batch (replicated over q)
  f32x2 CB[2];             // g.(out - T_final bg) + T_final (g.bg)
  float gown[NG];          // dL/dcolor of the lane's OWN pixel (lane L <-> pixel L): B operand of the g.c products
};

// g.c for 64 pixels x 16 splats: a rank-3 (dual: rank-6) bilinear form, so it comes out of the matrix cores in the
// same layout as the exponents (A = the lanes' own splat colours, B = the lanes' own pixel gradients).
template <int B, bool DUAL>
__device__ __forceinline__ f32x16 mx_gdotc(const MxSplat& m, const float (&g)[DUAL ? 6 : 3]) {
  f32x16 d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_16x16x1f32(m.pay.x, g[0], d, 2, B, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x1f32(m.pay.y, g[1], d, 2, B, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x1f32(m.pay.z, g[2], d, 2, B, 0);
  if constexpr (DUAL) {
    d = __builtin_amdgcn_mfma_f32_16x16x1f32(m.pay.w, g[3], d, 2, B, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x1f32(m.pay2.x, g[4], d, 2, B, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x1f32(m.pay2.y, g[5], d, 2, B, 0);
  }
  return d;
}

template <int B, bool DUAL>
__device__ __forceinline__ void mx_backward_batch(MxBwdState<DUAL ? 6 : 3>& st, const MxSplat& m, const float (&Phi)[6],
                                                  float4* __restrict__ lds_xch, float* __restrict__ Us,
                                                  float* __restrict__ Ws, int l) {
  const int j = l & 15, q = l >> 4;
  const f32x16 d = mx_exponents<B>(m.K, Phi);
  const f32x16 gcv = mx_gdotc<B, DUAL>(m, st.gown);
  const f32x2 one = {1.f, 1.f};
  // two half-passes over the pixel groups {0,1} and {2,3}: halves the number of live per-pair values.
  // Everything that does not depend on the incoming transmittance / prefix is folded before the quad-lane exchange:
  //   u' = Gm (t gc - (CB - P) / (1 - a)),  t = Tin pl[r-1]   ==>   u' = Tin gt - (CB - P) gr,   w = Tin wl
  //   with gt = Gm pl[r-1] gc,  gr = Gm / (1 - a),  wl = a pl[r-1]   (pl[-1] = 1)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x2 gt[4], gr[4], wl[4], pl[4], S[4];
    const bool alive0 = st.Tb[h].x > 0.f, alive1 = st.Tb[h].y > 0.f;     // pixel not finished at the start of the batch
    {
      f32x2 Gm[4], a[4], gc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float Gp0 = __builtin_amdgcn_exp2f(d[8 * h + r]), Gp1 = __builtin_amdgcn_exp2f(d[8 * h + 4 + r]);
        const float al0 = fminf(kAlphaMax, Gp0), al1 = fminf(kAlphaMax, Gp1);
        const bool v0 = al0 >= kAlphaMin, v1 = al1 >= kAlphaMin;
        a[r].x = v0 ? al0 : 0.f; a[r].y = v1 ? al1 : 0.f;
        Gm[r].x = (v0 && alive0) ? Gp0 : 0.f;                  // alpha_unclamped where this pair can contribute, else 0
        Gm[r].y = (v1 && alive1) ? Gp1 : 0.f;                  // (the 0.99 clamp passes the gradient through)
        gc[r].x = gcv[8 * h + r]; gc[r].y = gcv[8 * h + 4 + r];
      }
      const f32x2 om1 = one - a[1], om2 = one - a[2], om3 = one - a[3];
      pl[0] = one - a[0];
      pl[1] = pl[0] * om1;
      pl[2] = pl[1] * om2;
      pl[3] = pl[2] * om3;
      // 1 / (1 - a_r) for the four splats from ONE reciprocal of their product (v_rcp_f32 is quarter rate):
      //   1/pl3 -> 1/om3 = pl2/pl3,  1/pl2 = om3/pl3 -> 1/om2 = pl1/pl2, ...      (om >= 0.01, so pl3 >= 1e-8)
      const f32x2 i3 = {__builtin_amdgcn_rcpf(pl[3].x), __builtin_amdgcn_rcpf(pl[3].y)};
      const f32x2 i2 = i3 * om3, i1 = i2 * om2;
      gr[0] = Gm[0] * (i1 * om1); gr[1] = Gm[1] * (pl[0] * i1); gr[2] = Gm[2] * (pl[1] * i2); gr[3] = Gm[3] * (pl[2] * i3);
      const f32x2 tg1 = pl[0] * gc[1], tg2 = pl[1] * gc[2], tg3 = pl[2] * gc[3];
      S[0] = gc[0] * a[0];
      S[1] = __builtin_elementwise_fma(tg1, a[1], S[0]);
      S[2] = __builtin_elementwise_fma(tg2, a[2], S[1]);
      S[3] = __builtin_elementwise_fma(tg3, a[3], S[2]);
      wl[0] = a[0]; wl[1] = a[1] * pl[0]; wl[2] = a[2] * pl[1]; wl[3] = a[3] * pl[2];
      gt[0] = Gm[0] * gc[0]; gt[1] = Gm[1] * tg1; gt[2] = Gm[2] * tg2; gt[3] = Gm[3] * tg3;
    }
    // exchange (local product, local T-free prefix sum) of both pixel groups among the four quad-lanes of a pixel
    float4* xch = lds_xch + 64 * h;
    xch[l] = make_float4(pl[3].x, pl[3].y, S[3].x, S[3].y);
    const float4 x0 = xch[j], x1 = xch[16 + j], x2 = xch[32 + j], x3 = xch[48 + j];
    const f32x2 P0 = {x0.x, x0.y}, P1 = {x1.x, x1.y}, P2 = {x2.x, x2.y}, P3 = {x3.x, x3.y};
    const f32x2 S0 = {x0.z, x0.w}, S1 = {x1.z, x1.w}, S2 = {x2.z, x2.w}, S3 = {x3.z, x3.w};
    const f32x2 T0 = st.Tb[h];
    const f32x2 T1 = T0 * P0, T2 = T1 * P1, T3 = T2 * P2;
    const f32x2 Q0 = st.Pb[h];
    const f32x2 Q1 = __builtin_elementwise_fma(T0, S0, Q0), Q2 = __builtin_elementwise_fma(T1, S1, Q1),
                Q3 = __builtin_elementwise_fma(T2, S2, Q2);
    const f32x2 Tin = (q == 0) ? T0 : (q == 1) ? T1 : (q == 2) ? T2 : T3;
    const f32x2 Pin = (q == 0) ? Q0 : (q == 1) ? Q1 : (q == 2) ? Q2 : Q3;
    const f32x2 Tend = T3 * P3;
    const f32x2 Pend = __builtin_elementwise_fma(T3, S3, Q3);
    const bool cross = (alive0 && Tend.x < kTStop) || (alive1 && Tend.y < kTStop);
    bool stopped0 = false, stopped1 = false;
    if (__ballot(cross) != 0ull) {                              // wave-uniform; rarely true
      // exact stop rule: the first (quad, splat) in list order with T*(1-alpha) < 1e-4 ends the pixel before adding;
      // everything from there on (this quad and the quads behind it) contributes nothing
      const unsigned long long lower_q = 0x0001000100010001ull & ((q == 0) ? 0ull : ((1ull << (16 * q)) - 1ull));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float tin = i ? Tin.y : Tin.x;
        const bool was_alive = i ? alive1 : alive0;
        bool livep = true, any = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float wr = i ? wl[r].y : wl[r].x, plr = i ? pl[r].y : pl[r].x;     // wl > 0  <=>  alpha >= 1/255
          const bool stop = wr > 0.f && tin * plr < kTStop;
          any = any || (livep && stop);
          livep = livep && !stop;
          if (!livep) { if (i) { gt[r].y = 0.f; gr[r].y = 0.f; wl[r].y = 0.f; } else { gt[r].x = 0.f; gr[r].x = 0.f; wl[r].x = 0.f; } }
        }
        const unsigned long long bal = __ballot(any && was_alive) >> j;
        const bool pixel_stopped = (bal & 0x0001000100010001ull) != 0ull;
        if (i) stopped1 = pixel_stopped; else stopped0 = pixel_stopped;
        if ((bal & lower_q) != 0ull) {                          // a quad in front of mine already ended the pixel
#pragma unroll
          for (int r = 0; r < 4; ++r) { if (i) { gt[r].y = 0.f; gr[r].y = 0.f; wl[r].y = 0.f; } else { gt[r].x = 0.f; gr[r].x = 0.f; wl[r].x = 0.f; } }
        }
      }
    }
    // no predicates from here: gt = gr = 0 where the pair cannot contribute, wl = 0 for invalid pairs, Tin = 0 for dead pixels
    const f32x2 CB = st.CB[h];
    float* __restrict__ us = Us + (2 * h) * kImgQuarter + (4 * q) * kImgRow + j;   // pixel block = pixel quarter
    float* __restrict__ ws = Ws + (2 * h) * kImgQuarter + (4 * q) * kImgRow + j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x2 Pr = __builtin_elementwise_fma(Tin, S[r], Pin);      // prefix including this splat
      const f32x2 u = __builtin_elementwise_fma(Pr - CB, gr[r], Tin * gt[r]);
      const f32x2 w = Tin * wl[r];
      us[r * kImgRow] = u.x; us[r * kImgRow + kImgQuarter] = u.y;
      ws[r * kImgRow] = w.x; ws[r * kImgRow + kImgQuarter] = w.y;
    }
    st.Tb[h].x = (alive0 && !stopped0) ? Tend.x : 0.f;
    st.Tb[h].y = (alive1 && !stopped1) ? Tend.y : 0.f;
    st.Pb[h] = Pend;
  }
}

// ---- lane = pixel replay (PX): the backward twin of composite_forward_px -------------------------------------------------
// Exponents and g.c for the lane's own pixel x 16 splats from the A-broadcast 16-block MFMA; T and the gradient prefix P
// run inside the lane with the forward's recurrence (w = alpha T, T' = T - w: the stop decisions are the forward's by
// construction); u' and w go to the same LDS images, so the contraction below is shared with the quad form.  A pixel that
// has ended carries T = 0 and CB = P, which makes every later u' and w exactly zero without a mask.
// NG = image-gradient channels: 3 (one render), 6 (dual render) or 4 (dual render whose second image passes a gradient
// through its FIRST channel only -- the [z, 1, z^2] render under get_loss, where the silhouette only feeds comparisons and
// z^2 a detached uncertainty, src/vtgaussian_slam.py:466-521)
template <int G, int NG>
__device__ __forceinline__ f32x4 px_gdotc(const MxSplat& m, const float (&g)[NG]) {
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(m.pay.x, g[0], d, 4, G, 0);
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(m.pay.y, g[1], d, 4, G, 0);
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(m.pay.z, g[2], d, 4, G, 0);
  if constexpr (NG >= 4) d = __builtin_amdgcn_mfma_f32_4x4x1f32(m.pay.w, g[3], d, 4, G, 0);
  if constexpr (NG == 6) {
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(m.pay2.x, g[4], d, 4, G, 0);
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(m.pay2.y, g[5], d, 4, G, 0);
  }
  return d;
}

struct PxBwdState { float T, P, CB; bool done; };   // T frozen once the pixel has ended (as in the forward); done: ended / off-image

// One batch of 16 splats for the lane's pixel, in two sweeps so that no per-splat division is needed:
//   front to back:  alpha_k, w_k = alpha_k T_k (-> LDS), T_{k+1} = T_k - w_k, P_k = P_{k-1} + (g.c_k) w_k
//   back to front:  A = colour still behind splat k, seen from behind it = (CB - P_k) / T_{k+1}; anchored ONCE per batch at
//                   its end (for a pixel that has ended: in front of its ending splat, where T is frozen), then
//                   A_{k-1} = A_k + alpha_k (g.c_k - A_k), and
//                   u'_k = alpha_unclamped_k dL/dalpha_k = G_k T_k (g.c_k - A_k)          (the reference's recurrence)
// dL/dalpha_k = T_k g.c_k - (CB - P_k)/(1 - alpha_k) is the same thing since (CB - P_k)/(1 - alpha_k) = T_k A_k.
// `exact` is the forward's switch: straight to the exact sweep while pixels keep ending.
template <int B, int NG, bool CLAMP, bool EXACT_FIRST>
__device__ __forceinline__ void px_backward_batch(PxBwdState& st, bool& exact, const MxSplat& m, const float (&Phi)[6],
                                                  const float (&gown)[NG], float* __restrict__ Us,
                                                  float* __restrict__ Ws, int l) {
  const f32x4 d[4] = {px_exponents<4 * B>(m.K, Phi), px_exponents<4 * B + 1>(m.K, Phi), px_exponents<4 * B + 2>(m.K, Phi),
                      px_exponents<4 * B + 3>(m.K, Phi)};
  const f32x4 gcv[4] = {px_gdotc<4 * B, NG>(m, gown), px_gdotc<4 * B + 1, NG>(m, gown), px_gdotc<4 * B + 2, NG>(m, gown),
                        px_gdotc<4 * B + 3, NG>(m, gown)};
  float* __restrict__ us = Us + (l >> 4) * kImgQuarter + (l & 15);   // image [pixel quarter][splat][16 px]
  float* __restrict__ ws = Ws + (l >> 4) * kImgQuarter + (l & 15);
  float a[16], gT[16];                                         // alpha_k and G_k T_k (0 where the pair contributes nothing)
  float Pn = st.P;
  bool swept = false;
  const bool was_done = st.done;
  if constexpr (!EXACT_FIRST) {
    // front to back, optimistic: no stop test
    float Tn = st.done ? 0.f : st.T;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float Gp = __builtin_amdgcn_exp2f(d[k >> 2][k & 3]);
      float w;
      if constexpr (CLAMP) {
        const float al = fminf(kAlphaMax, Gp);
        const bool valid = al >= kAlphaMin;
        a[k] = valid ? al : 0.f;
        gT[k] = valid ? Gp * Tn : 0.f;                         // the 0.99 clamp passes the gradient through
        w = a[k] * Tn;
      } else {                                                 // no clamp possible in this chunk: alpha == alpha_unclamped
        a[k] = (Gp >= kAlphaMin) ? Gp : 0.f;
        w = a[k] * Tn;
        gT[k] = w;
      }
      ws[k * kImgRow] = w;
      Pn = fmaf(gcv[k >> 2][k & 3], w, Pn);
      Tn = Tn - w;
    }
#if VTGS_PX_GROUP
    __builtin_amdgcn_sched_group_barrier(0x008, 24 + 4 * NG, 0);
    __builtin_amdgcn_sched_group_barrier(0x302, 400, 0);
#endif
    if (__ballot(!st.done && Tn < kTStop) == 0ull) { st.T = st.done ? st.T : Tn; swept = true; }
    else Pn = st.P;
  }
  if (!swept) {
    // exact sweep: from the ending splat on alpha and G T are zero, so the back sweep passes A through and writes zeros
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float Gp = __builtin_amdgcn_exp2f(d[k >> 2][k & 3]);
      const float al = CLAMP ? fminf(kAlphaMax, Gp) : Gp;
      const float av = (al >= kAlphaMin) ? al : 0.f;
      const float wk = av * st.T;
      const float tn = st.T - wk;
      const bool stop = tn < kTStop;                           // a live pixel has T >= 1e-4: alpha = 0 cannot trigger it
      const bool live = !st.done && !stop;
      a[k] = live ? av : 0.f;
      gT[k] = live ? ((al >= kAlphaMin) ? Gp * st.T : 0.f) : 0.f;
      const float w = live ? wk : 0.f;
      ws[k * kImgRow] = w;
      Pn = fmaf(gcv[k >> 2][k & 3], w, Pn);
      st.T = live ? tn : st.T;
      st.done = st.done || stop;
    }
    exact = __builtin_popcountll(__ballot(st.done && !was_done)) >= kExactFirstEndings;   // as in the forward
  }
  // anchor: colour behind the batch (behind the ending splat for an ended pixel) per unit of transmittance there; T > 0
  float A = (st.CB - Pn) * __builtin_amdgcn_rcpf(st.T);
  st.P = Pn;
  // back to front
#pragma unroll
  for (int k = 15; k >= 0; --k) {
    const float t = gcv[k >> 2][k & 3] - A;
    us[k * kImgRow] = gT[k] * t;
    A = fmaf(a[k], t, A);
  }
}

// B1 (lane = pixel form of the dual render only): the second image's gradient is zero outside its first channel (CamScalars /
// FrameEpilogue flag 8, promised by the caller -- get_loss, whose loss touches the [z, 1, z^2] render through z alone).  Then
// g.c has four terms, not six; the second set's one colour sum rides in the idle fourth column of chain w (no fourth chain);
// a splat brings one colour of the second set, which leaves the registers for the single render's record prefetch; and a
// record is 12 floats (three aligned float4) instead of 14: 88 matrix instructions per batch instead of 112, 48-byte records.
template <int WAVES, bool DUAL, bool PXL, bool B1 = false>
__global__ __launch_bounds__(64 * WAVES, 3) void composite_backward_mx(
    CamScalars cs, const float* __restrict__ bg, uint32_t nblk,
    const uint32_t* __restrict__ tile_cnt, uint32_t tile_cap, const uint32_t* __restrict__ sorted_gid,
    const uint32_t* __restrict__ sorted_inst, const GeomRec* __restrict__ geom, const float* __restrict__ colors,
    const float* __restrict__ out_color, const float* __restrict__ grad_color, const float* __restrict__ final_T,
    float* __restrict__ grad_inst, const Counters* __restrict__ ctr, const float* __restrict__ colors_b,
    const float* __restrict__ out_color_b, const float* __restrict__ grad_color_b, uint32_t* __restrict__ dbg) {
  static_assert(!B1 || (DUAL && PXL), "B1 is a form of the lane = pixel dual backward");
  constexpr int NG = DUAL ? (B1 ? 4 : 6) : 3;                 // image-gradient channels
  constexpr bool DUAL6 = DUAL && !B1;                         // both colour sets in full
#ifdef VTGS_Q_STAMPS
  const unsigned long long st0 = __builtin_amdgcn_s_memtime();
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_gather = 0ull, st_batch = 0ull, st_loop0 = 0ull;
  uint32_t nbatches = 0u;
#endif
  constexpr int REC = DUAL ? (B1 ? kGradRecDual1 : kGradRecDual) : kGradRec;   // floats per (splat, tile) record
  __shared__ float4 lds_xch_all[WAVES][PXL ? 1 : 128];
#ifdef VTGS_AB_BWD_PAD                                          // occupancy experiment: extra LDS so that fewer workgroups fit a CU
  __shared__ float ab_pad[VTGS_AB_BWD_PAD];
  if (cs.W < 0) { ab_pad[threadIdx.x] = 1.f; __syncthreads(); if (ab_pad[(threadIdx.x + 1) & 255] == 2.f) return; }
#endif
  __shared__ __attribute__((aligned(16))) float lds_uw[WAVES][2][kImgFloats];
  __shared__ __attribute__((aligned(16))) float lds_phi[4 * kPhiQuarter];
  // the contraction's dL/dcolor operands live in LDS ([wave][second set?][quarter][column][16 + 4]), not in 16 / 32 VGPRs:
  // round 3 -- the registers hold the NEXT chunk's geometry records instead (the gathers were 12 % of the wavefront's life)
  __shared__ __attribute__((aligned(16))) float lds_g_all[WAVES][(DUAL6 ? 2 : 1) * 16 * kImgRow];
  // Everything the wavefront needs before its first batch is requested up front -- flag, list length, image values, first
  // list entries -- and the flag is looked at afterwards: the prologue was four dependent round trips (15 % of the
  // wavefront's life, profiles/r3_stamps.md)
  const uint32_t overflow_flag = ctr->overflow;
  // Phi table for the contraction's B operands: [pixel quarter][column 0..7][pixel 0..15 (+4 pad)], columns 6,7 = 0
  for (int i = (int)threadIdx.x; i < 4 * kPhiQuarter; i += 64 * WAVES) {
    const int pqt = i / kPhiQuarter, c = (i - pqt * kPhiQuarter) / kImgRow, kk = i - pqt * kPhiQuarter - c * kImgRow;
    const int p = 16 * pqt + kk;
    const float PX = (float)(p & 7) - 3.5f, PY = (float)(p >> 3) - 3.5f;
    float v = 0.f;
    v = (c == 0) ? 1.f : v; v = (c == 1) ? PX : v; v = (c == 2) ? PY : v;
    v = (c == 3) ? PX * PX : v; v = (c == 4) ? PX * PY : v; v = (c == 5) ? PY * PY : v;
    lds_phi[i] = (kk < 16) ? v : 0.f;
  }
  const int gx16 = (cs.W + kBinTile - 1) / kBinTile;
  const int gx8 = (cs.W + kSubTile - 1) / kSubTile, gy8 = (cs.H + kSubTile - 1) / kSubTile;
  TileCoord tc = tile_coord<WAVES>(cs, nblk, gx16, gx8, gy8);
  const bool tile_ok = tc.tile_ok;                              // wave-uniform
  if (!tile_ok) { tc.tile = 0; tc.inside = false; }              // (its speculative loads below read tile 0's bin)
  const int l = lane_id();
  const int wv = (WAVES == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float4* lds_xch = lds_xch_all[wv];
  float* __restrict__ Us = lds_uw[wv][0];
  float* __restrict__ Ws = lds_uw[wv][1];
  const BinRange br = bin_range(cs, (uint32_t)tc.tile, tile_cap);
  const uint32_t s = br.s;
  const uint32_t list_len = min(tile_cnt[tc.tile], br.cap);
  // first list entries (a bin holds at least 64 slots: reading past a short list stays inside the workspace; masked later)
  uint32_t gid_first = sorted_gid[s + (uint32_t)lane_id()], inst_first = sorted_inst[s + (uint32_t)lane_id()];
  const size_t P = (size_t)cs.W * cs.H;
  const int j = l & 15, q = l >> 4;
  const int tx0 = tc.px - (l & 7), ty0 = tc.py - (l >> 3);
  const float cx = (float)tx0 + 3.5f, cy = (float)ty0 + 3.5f;
  const float X = (float)(l & 7) - 3.5f, Y = (float)(l >> 3) - 3.5f;
  const float Phi[6] = {1.f, X, Y, X * X, X * Y, Y * Y};
  const float b0 = bg[0], b1 = bg[1], b2 = bg[2];

  MxBwdState<NG> st;
  PxBwdState ps{1.f, 0.f, 0.f, !tc.inside};
  bool px_exact = false;
  if (PXL && tc.inside) {                                        // lane L <-> pixel L
    const size_t pix = (size_t)tc.py * cs.W + tc.px;
    const float g0 = grad_color[pix], g1 = grad_color[P + pix], g2 = grad_color[2 * P + pix];
    const float Tf = final_T[pix];
    ps.CB = g0 * (out_color[pix] - Tf * b0) + g1 * (out_color[P + pix] - Tf * b1) + g2 * (out_color[2 * P + pix] - Tf * b2)
            + Tf * (g0 * b0 + g1 * b1 + g2 * b2);
    if constexpr (DUAL6) {
      const float g3 = grad_color_b[pix], g4 = grad_color_b[P + pix], g5 = grad_color_b[2 * P + pix];
      ps.CB += g3 * (out_color_b[pix] - Tf * b0) + g4 * (out_color_b[P + pix] - Tf * b1) + g5 * (out_color_b[2 * P + pix] - Tf * b2)
               + Tf * (g3 * b0 + g4 * b1 + g5 * b2);
    } else if constexpr (B1) {
      const float g3 = grad_color_b[pix];
      ps.CB += g3 * (out_color_b[pix] - Tf * b0) + Tf * (g3 * b0);
    }
  }
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    if (PXL) break;
    if constexpr (!B1) {
    const int p = 16 * blk + j;
    const int qx = tx0 + (p & 7), qy = ty0 + (p >> 3);
    const bool in_img = qx < cs.W && qy < cs.H;
    float cb = 0.f;
    if (in_img) {
      const size_t pix = (size_t)qy * cs.W + qx;
      const float g0 = grad_color[pix], g1 = grad_color[P + pix], g2 = grad_color[2 * P + pix];
      const float Tf = final_T[pix];
      cb = g0 * (out_color[pix] - Tf * b0) + g1 * (out_color[P + pix] - Tf * b1) + g2 * (out_color[2 * P + pix] - Tf * b2)
           + Tf * (g0 * b0 + g1 * b1 + g2 * b2);
      if constexpr (DUAL) {
        const float g3 = grad_color_b[pix], g4 = grad_color_b[P + pix], g5 = grad_color_b[2 * P + pix];
        cb += g3 * (out_color_b[pix] - Tf * b0) + g4 * (out_color_b[P + pix] - Tf * b1) + g5 * (out_color_b[2 * P + pix] - Tf * b2)
              + Tf * (g3 * b0 + g4 * b1 + g5 * b2);
      }
    }
    if (blk & 1) { st.Tb[blk >> 1].y = in_img ? 1.f : 0.f; st.CB[blk >> 1].y = cb; st.Pb[blk >> 1].y = 0.f; }
    else         { st.Tb[blk >> 1].x = in_img ? 1.f : 0.f; st.CB[blk >> 1].x = cb; st.Pb[blk >> 1].x = 0.f; }
    }
  }
#pragma unroll
  for (int c = 0; c < NG; ++c) st.gown[c] = 0.f;
  if (tc.inside) {                                              // lane L <-> pixel L (tc.px, tc.py)
    const size_t pix = (size_t)tc.py * cs.W + tc.px;
    st.gown[0] = grad_color[pix]; st.gown[1] = grad_color[P + pix]; st.gown[2] = grad_color[2 * P + pix];
    if constexpr (DUAL) st.gown[3] = grad_color_b[pix];
    if constexpr (DUAL6) { st.gown[4] = grad_color_b[P + pix]; st.gown[5] = grad_color_b[2 * P + pix]; }
  }
  // Contraction roles: l = cj + 4 sg + 16 pq -- column cj of every group, splat group sg, pixel quarter pq.
  // A operand: row (l & 3) of block (sg, pq) = splat 4 sg + (l & 3) = image row (l & 15); B operand: column cj.
  const int cj = l & 3, pq = q;
  // dL/dcolor as the contraction's B operand: row (pq, channel) of set a at [(4 pq + channel) * kImgRow], 16 pixels + pad,
  // channel 3 = 0; dual: set b 320 floats on.  Written from the lanes' OWN pixel gradients (lane L = pixel L = pixel
  // (L & 15) of quarter L >> 4): no second trip to the image.
  float* lds_g = lds_g_all[wv];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    lds_g[(4 * pq + c) * kImgRow + j] = (c < 3) ? st.gown[c < 3 ? c : 0] : (B1 ? st.gown[B1 ? 3 : 0] : 0.f);   // B1: the second set's
    if constexpr (DUAL6) lds_g[16 * kImgRow + (4 * pq + c) * kImgRow + j] = (c < 3) ? st.gown[3 + (c < 3 ? c : 0)] : 0.f;   // one live channel
  }
  __syncthreads();                                            // the Phi table (and, per wavefront, the g image); before any exit
  if (overflow_flag) return;                                  // uniform over the grid: the forward did not complete
  if (!tile_ok || list_len == 0u) return;
  const uint32_t e = s + list_len;
  const float4* __restrict__ Ga4 = reinterpret_cast<const float4*>(lds_g + (4 * pq + cj) * kImgRow);
  const float4* __restrict__ Gb4 = reinterpret_cast<const float4*>(lds_g + 16 * kImgRow + (4 * pq + cj) * kImgRow);
  const float4* Ua4 = reinterpret_cast<const float4*>(Us + pq * kImgQuarter + (l & 15) * kImgRow);   // alias Us / Ws: no restrict
  const float4* Wa4 = reinterpret_cast<const float4*>(Ws + pq * kImgQuarter + (l & 15) * kImgRow);
  const float4* __restrict__ PhiA4 = reinterpret_cast<const float4*>(lds_phi + pq * kPhiQuarter + cj * kImgRow);
  const float4* __restrict__ PhiB4 = reinterpret_cast<const float4*>(lds_phi + pq * kPhiQuarter + (4 + cj) * kImgRow);
  const uint32_t tile_bits = (uint32_t)tc.tile;
  const bool skip_a = PXL && !B1 && (cs.bwd_flags & 1u) != 0u;   // (B1: chain w also carries the second set's column)
  // record columns: chain a -> 0..3, chain b -> 4, 5 (its lanes cj = 2, 3 hold padding and store nothing), chain w -> 6..8
  // and the tile id in 9 (dual: w -> 6..8 + tile id in 12, w2 -> 9..11, its lane cj = 3 stores nothing)
  const int col_b = 4 + cj;
  const bool b_live = cj < 2;
  // B1: chain w -> 6..9 (9 = the second set's first colour), the tile id in 10 from chain b's idle lane cj = 2
  const int col_w = (DUAL6 && cj == 3) ? 12 : 6 + cj;
  const int col_w2 = 9 + cj;

  uint32_t base = s;
#ifdef VTGS_Q_STAMPS
  st_loop0 = __builtin_amdgcn_s_memtime();
#endif
  // Software pipeline of the gather, two deep as in composite_forward_q: while chunk c is composited, the list entries of
  // chunk c + 2 and the geometry records + colours of chunk c + 1 are in flight (a lane past the end reads entry 0 of its bin).
  auto entry = [&](const uint32_t* __restrict__ list, uint32_t b) { const uint32_t p = b + (uint32_t)l; return list[p < e ? p : s]; };
  // the speculative first read went past the end of a short list, into slots nobody wrote: those lanes take entry 0 instead
  // (every id that is gathered through must be a written one -- an unwritten slot holds whatever the memory held before)
  const bool first_in = (uint32_t)l < list_len;
  const uint32_t gid0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)gid_first);    // lane 0 (all lanes active here): entry 0
  const uint32_t inst0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)inst_first);
  uint32_t gid_cur = first_in ? gid_first : gid0;
  uint32_t inst_cur = first_in ? inst_first : inst0;
  uint32_t gid_nxt = entry(sorted_gid, s + 64u), inst_nxt = entry(sorted_inst, s + 64u);
  float4 g0n, g1n;
  float cn[NG];
  auto fetch = [&](uint32_t gid) {
    const float4* gp = reinterpret_cast<const float4*>(geom + gid);
    g0n = gp[0]; g1n = gp[1];
    cn[0] = colors[3 * gid]; cn[1] = colors[3 * gid + 1]; cn[2] = colors[3 * gid + 2];
    if constexpr (DUAL) cn[3] = colors_b[3 * gid];
    if constexpr (DUAL6) { cn[4] = colors_b[3 * gid + 1]; cn[5] = colors_b[3 * gid + 2]; }
  };
  // (dual render: 14 more registers in flight across the batches push the kernel into scratch -- there the records are
  // requested at the top of their own chunk, as in round 2; the entries still come one chunk ahead)
  constexpr bool kRecordsAhead = !DUAL6;
  if constexpr (kRecordsAhead) fetch(gid_cur);
  for (; base < e; base += 64u) {
    const bool alive = PXL ? !ps.done : (st.Tb[0].x > 0.f || st.Tb[0].y > 0.f || st.Tb[1].x > 0.f || st.Tb[1].y > 0.f);
    if (__ballot(alive) == 0ull) break;
    const int n = (int)min(64u, e - base);
#ifdef VTGS_Q_STAMPS
    const unsigned long long sg0 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (!kRecordsAhead) fetch(gid_cur);
    MxSplat m;                                                  // this chunk's splats, from the records requested one chunk ago
    m.K[0] = -1e30f; m.K[1] = m.K[2] = m.K[3] = m.K[4] = m.K[5] = 0.f;
    m.pay = make_float4(0.f, 0.f, 0.f, 0.f); m.pay2 = make_float2(0.f, 0.f); m.hot = false;
    if (l < n) {
      tile_coefficients(g0n, g1n, cx, cy, m.K);
      m.hot = g1n.y > kClampGuard;
      m.pay = make_float4(cn[0], cn[1], cn[2], DUAL ? cn[DUAL ? 3 : 0] : g1n.z);
      if constexpr (DUAL6) m.pay2 = make_float2(cn[DUAL6 ? 4 : 0], cn[DUAL6 ? 5 : 0]);
    }
    const uint32_t my_inst = (l < n) ? inst_cur : 0u;
    gid_cur = gid_nxt; inst_cur = inst_nxt;
    gid_nxt = entry(sorted_gid, base + 128u); inst_nxt = entry(sorted_inst, base + 128u);
    if constexpr (kRecordsAhead) fetch(gid_cur);                 // next chunk's records: in flight during this chunk's batches
    const bool hot = __ballot(m.hot) != 0ull;                 // wave-uniform: some splat of the chunk may hit the 0.99 clamp
#ifdef VTGS_Q_STAMPS
    asm volatile("" :: "v"(m.K[0]), "v"(m.pay.x));
    __builtin_amdgcn_s_waitcnt(0x0070);                        // vmcnt(0): the chunk's gathers have arrived
    const unsigned long long sg1 = __builtin_amdgcn_s_memtime();
    st_gather += sg1 - sg0;
#endif
    const bool chunk_exact = px_exact;                        // exact sweep first for this chunk (pixels kept ending before it)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (16 * b >= n) break;                                 // wave-uniform
      if constexpr (PXL) {
#define VTGS_PX_BWD(CL, EX)                                                                               \
        {                                                                                                 \
          if (b == 0) px_backward_batch<0, NG, CL, EX>(ps, px_exact, m, Phi, st.gown, Us, Ws, l);         \
          if (b == 1) px_backward_batch<1, NG, CL, EX>(ps, px_exact, m, Phi, st.gown, Us, Ws, l);         \
          if (b == 2) px_backward_batch<2, NG, CL, EX>(ps, px_exact, m, Phi, st.gown, Us, Ws, l);         \
          if (b == 3) px_backward_batch<3, NG, CL, EX>(ps, px_exact, m, Phi, st.gown, Us, Ws, l);         \
        }
        if (chunk_exact) VTGS_PX_BWD(true, true)                // (one exact-first body: the clamped form is always valid)
        else             { if (hot) VTGS_PX_BWD(true, false) else VTGS_PX_BWD(false, false) }
#undef VTGS_PX_BWD
      } else {
        if constexpr (!B1) {
          if (b == 0) mx_backward_batch<0, DUAL>(st, m, Phi, lds_xch, Us, Ws, l);
          if (b == 1) mx_backward_batch<1, DUAL>(st, m, Phi, lds_xch, Us, Ws, l);
          if (b == 2) mx_backward_batch<2, DUAL>(st, m, Phi, lds_xch, Us, Ws, l);
          if (b == 3) mx_backward_batch<3, DUAL>(st, m, Phi, lds_xch, Us, Ws, l);
        }
      }
      const int nb = min(16, n - 16 * b);
      f32x4 Pa = {0.f, 0.f, 0.f, 0.f}, Pb = {0.f, 0.f, 0.f, 0.f}, Pw = {0.f, 0.f, 0.f, 0.f}, Pw2 = {0.f, 0.f, 0.f, 0.f};
      // SKIPA (wave-uniform, CamScalars::bwd_flags bit 0): dL/d(first colour set) is wanted by nobody -- a tracking iteration --
      // so chain w (16 of the 48 / 64 contraction MFMAs and its operand reads) is left out; its record columns stay unwritten
      // and the gather kernel does not store what it sums from them
#define VTGS_CONTRACT(SKIPA)                                                                                        \
      {                                                                                                             \
        _Pragma("unroll") for (int t4 = 0; t4 < 4; ++t4) {                                                          \
          const float4 ua = Ua4[t4], wa = Wa4[t4], ba = PhiA4[t4], bb = PhiB4[t4];                                  \
          const float uav[4] = {ua.x, ua.y, ua.z, ua.w}, wav[4] = {wa.x, wa.y, wa.z, wa.w};                         \
          const float bav[4] = {ba.x, ba.y, ba.z, ba.w}, bbv[4] = {bb.x, bb.y, bb.z, bb.w};                         \
          float gav[4] = {0.f, 0.f, 0.f, 0.f}, gbv[4] = {0.f, 0.f, 0.f, 0.f};                                       \
          if constexpr (!(SKIPA)) {                                                                                 \
            const float4 ga = Ga4[t4];                                                                              \
            gav[0] = ga.x; gav[1] = ga.y; gav[2] = ga.z; gav[3] = ga.w;                                             \
          }                                                                                                         \
          if constexpr (DUAL6) {                                                                                    \
            const float4 gb = Gb4[t4];                                                                              \
            gbv[0] = gb.x; gbv[1] = gb.y; gbv[2] = gb.z; gbv[3] = gb.w;                                             \
          }                                                                                                         \
          _Pragma("unroll") for (int e4 = 0; e4 < 4; ++e4) {                                                        \
            Pa = __builtin_amdgcn_mfma_f32_4x4x1f32(uav[e4], bav[e4], Pa, 0, 0, 0);                                 \
            Pb = __builtin_amdgcn_mfma_f32_4x4x1f32(uav[e4], bbv[e4], Pb, 0, 0, 0);                                 \
            if constexpr (!(SKIPA)) Pw = __builtin_amdgcn_mfma_f32_4x4x1f32(wav[e4], gav[e4], Pw, 0, 0, 0);         \
            if constexpr (DUAL6) Pw2 = __builtin_amdgcn_mfma_f32_4x4x1f32(wav[e4], gbv[e4], Pw2, 0, 0, 0);          \
          }                                                                                                         \
        }                                                                                                           \
        if (VTGS_PX_GROUP2) {                                                                                       \
          __builtin_amdgcn_sched_group_barrier(0x100, (DUAL6 ? 24 : 20) - ((SKIPA) ? 4 : 0), 1);  /* the image / Phi / g reads first ... */ \
          __builtin_amdgcn_sched_group_barrier(0x008, (DUAL6 ? 64 : 48) - ((SKIPA) ? 16 : 0), 1); /* ... then the contraction MFMAs back to back */ \
        }                                                                                                           \
      }
      if (skip_a) VTGS_CONTRACT(true) else VTGS_CONTRACT(false)
#undef VTGS_CONTRACT
      // lane (cj, sg, rho = l >> 4) ends up with the totals of splat 4 sg + rho
      const float Fa = quarter_sum(Pa), Fb = quarter_sum(Pb), Fw = skip_a ? 0.f : quarter_sum(Pw);   // (skip_a is wave-uniform)
      const float Fw2 = DUAL6 ? quarter_sum(Pw2) : 0.f;          // cross-lane: must run with all lanes active
      const int srow = (l & 12) + (l >> 4);                     // 4 sg + rho
      const uint32_t inst = (uint32_t)__shfl((int)my_inst, 16 * b + srow, 64);
      if (srow < nb && inst < cs.scratch_records) {
        float* __restrict__ rec = grad_inst + (size_t)inst * REC;
        rec[cj] = Fa;
        if (b_live) rec[col_b] = Fb;
        if constexpr (B1) {
          if (cj == 2) rec[10] = __uint_as_float(tile_bits);
          rec[col_w] = Fw;
        } else {
          rec[col_w] = (cj == 3) ? __uint_as_float(tile_bits) : Fw;
        }
        if constexpr (DUAL6) { if (cj < 3) rec[col_w2] = Fw2; }
      }
#ifdef VTGS_Q_STAMPS
      ++nbatches;
#endif
    }
#ifdef VTGS_Q_STAMPS
    st_batch += __builtin_amdgcn_s_memtime() - sg1;
#endif
  }
  for (; base < e; base += 64u) {
    const int n = (int)min(64u, e - base);
    const uint32_t inst = (l < n) ? sorted_inst[base + (uint32_t)l] : 0xFFFFFFFFu;
    if (l < n && inst < cs.scratch_records) {
      if constexpr (B1) {
        float4* p = reinterpret_cast<float4*>(grad_inst + (size_t)inst * REC);
        p[0] = p[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        p[2] = make_float4(0.f, 0.f, __uint_as_float(tile_bits), 0.f);
      } else if constexpr (DUAL) {
        float2* p = reinterpret_cast<float2*>(grad_inst + (size_t)inst * REC);
        p[0] = p[1] = p[2] = p[3] = p[4] = p[5] = make_float2(0.f, 0.f);
        p[6] = make_float2(__uint_as_float(tile_bits), 0.f);
      } else {
        float2* p = reinterpret_cast<float2*>(grad_inst + (size_t)inst * REC);
        p[0] = p[1] = p[2] = p[3] = make_float2(0.f, 0.f);
        p[4] = make_float2(0.f, __uint_as_float(tile_bits));
      }
    }
  }
#ifdef VTGS_Q_STAMPS
  if (dbg && l == 0) {
    uint32_t* o = dbg + 64 + kStampWords * tc.tile;
    const unsigned long long se = __builtin_amdgcn_s_memtime();
    o[0] = (uint32_t)(st_loop0 - st0); o[1] = 0u; o[2] = (uint32_t)st_gather; o[3] = (uint32_t)st_batch;
    o[4] = (uint32_t)(se - st0); o[5] = nbatches; o[6] = e - s; o[7] = 0u;
    o[8] = (uint32_t)rt0; o[9] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    o[10] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); o[11] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // HW_ID, XCC_ID
  }
#endif
}
#if VTGS_XCHECK_BUILD
template __global__ void composite_backward_mx<4, false, false>(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*, const uint32_t*, const GeomRec*, const float*, const float*, const float*, const float*, float*, const Counters*, const float*, const float*, const float*, uint32_t*);
template __global__ void composite_backward_mx<4, true, false>(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*, const uint32_t*, const GeomRec*, const float*, const float*, const float*, const float*, float*, const Counters*, const float*, const float*, const float*, uint32_t*);
#else
template __global__ void composite_backward_mx<4, false, true>(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*, const uint32_t*, const GeomRec*, const float*, const float*, const float*, const float*, float*, const Counters*, const float*, const float*, const float*, uint32_t*);
template __global__ void composite_backward_mx<4, true, true>(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*, const uint32_t*, const GeomRec*, const float*, const float*, const float*, const float*, float*, const Counters*, const float*, const float*, const float*, uint32_t*);
template __global__ void composite_backward_mx<4, true, true, true>(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*, const uint32_t*, const GeomRec*, const float*, const float*, const float*, const float*, float*, const Counters*, const float*, const float*, const float*, uint32_t*);
#endif


#if !VTGS_XCHECK_BUILD
// one thread per Gaussian: re-centre and sum its instance records (fixed order), then the projection backward.
// FRAME (dual render of the fused caller chain): the adjoint of vtgs_prepare_frame runs here, on the gradients while they
// are still in registers -- same formulas and the same block reduction as prepare_frame_backward_kernel (vtgs_frame.hip);
// the results agree to float32 rounding -- instead of writing six dense [N, .] arrays for that kernel to read back
// (~160 bytes per Gaussian less traffic, one launch less; on a rank of the tile-row partition no dense zero arrays at all).
// COV3D (the operator's cov3D_precomp): `scales` holds the six covariance entries per Gaussian and g_scales receives their
// six gradients; `rotations` / g_rotations are not touched.
// B1: the 12-float records of composite_backward_mx<.., B1> (second image differentiated through its first channel only).
template <bool DUAL, bool FRAME = false, bool COV3D = false, bool B1 = false>
__global__ __launch_bounds__(256) void gather_splat_grads(
    CamScalars cs, const float* __restrict__ Vp, const float* __restrict__ PVp, int n,
    const float* __restrict__ means3D, const float* __restrict__ opacities,
    const float* __restrict__ scales, const float* __restrict__ rotations,
    const GaussAux* __restrict__ gaux, const float* __restrict__ grad_inst, int moments_scaled_by_opacity,
    float* __restrict__ g_means3D, float* __restrict__ g_means2D, float* __restrict__ g_colors,
    float* __restrict__ g_opacities, float* __restrict__ g_scales, float* __restrict__ g_rotations,
    const Counters* __restrict__ ctr, float* __restrict__ g_colors_b, FrameEpilogue fe = FrameEpilogue{}) {
  // Workgroups walk the map from its END: a SLAM map is appended to (densification, new submaps), and the appended Gaussians
  // are the ones the optimiser has grown the most -- the workgroups with the most records.  Started last they were the kernel's
  // tail; started first, the light workgroups fill in behind them.  (The per-workgroup pose partials keep their row: the
  // block index below is only the position in the map.)
  const uint32_t blk = gridDim.x - 1u - blockIdx.x;
  const int gid = (int)(blk * 256u + threadIdx.x);
  if (ctr->overflow) {
    // The forward did not complete: nothing valid to differentiate.  The Python layer never gets here (its checked forward
    // answers an overflow before it returns); a bare C-ABI caller that ignored the overflow of an asynchronous forward
    // gets DEFINED gradients -- zeros -- instead of whatever the output buffers held.
    if (gid < n) {
      if constexpr (FRAME) {
        const int row = fe.idx ? fe.idx[gid] : gid;
        if (fe.flags & 1u) {
          fe.g_means3D[3 * row] = fe.g_means3D[3 * row + 1] = fe.g_means3D[3 * row + 2] = 0.f;
          reinterpret_cast<float4*>(fe.g_unnorm_rot)[row] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (fe.flags & 4u) {
          fe.g_logit[row] = 0.f; fe.g_log_scales[row] = 0.f;
          fe.g_rgb[3 * row] = fe.g_rgb[3 * row + 1] = fe.g_rgb[3 * row + 2] = 0.f;
        }
      } else {
        for (int i = 0; i < 3; ++i) {
          if (g_means3D) g_means3D[3 * gid + i] = 0.f;
          if (g_means2D) g_means2D[3 * gid + i] = 0.f;
          if (g_colors) g_colors[3 * gid + i] = 0.f;
          if (g_scales && !COV3D) g_scales[3 * gid + i] = 0.f;
          if constexpr (DUAL) { if (g_colors_b) g_colors_b[3 * gid + i] = 0.f; }
        }
        if constexpr (COV3D) { if (g_scales) for (int i = 0; i < 6; ++i) g_scales[6 * gid + i] = 0.f; }
        if (g_opacities) g_opacities[gid] = 0.f;
        if (g_rotations && !COV3D) reinterpret_cast<float4*>(g_rotations)[gid] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if constexpr (FRAME) {
      if ((fe.flags & 2u) && threadIdx.x < 12) fe.pose_partials[(size_t)blk * 12 + threadIdx.x] = 0.f;
    }
    return;
  }
  const CamParams cam = load_cam(cs, Vp, PVp);
  const bool live = gid < n;                     // nobody leaves: the wavefront sums the records of its big splats together
  // (an early exit for wavefronts without any instance -- 7/8 of them on a rank of an 8-way partition -- was measured and
  //  dropped: it puts the camera's scalar loads behind the gaux load for the wavefronts that stay, +2 us whole frame, +1 in a band)
  GaussAux ga = live ? gaux[gid] : GaussAux{0u, 0u};
  if ((unsigned long long)ga.inst_base + ga.inst_cnt > (unsigned long long)cs.scratch_records) ga.inst_cnt = 0u;   // (CamScalars::scratch_records)
  SplatGrads g;
  for (int i = 0; i < 3; ++i) { g.mean3D[i] = g.mean2D[i] = g.color[i] = g.scale[i] = 0.f; }
  g.opacity = 0.f; g.rot[0] = g.rot[1] = g.rot[2] = g.rot[3] = 0.f;
  float cb0 = 0.f, cb1 = 0.f, cb2 = 0.f;         // dual: dL/d(second render's colours)
  // ---- the gradient records of the wavefront's 64 splats -----------------------------------------------------------------
  // Round 6.  Until round 5 every lane walked ITS OWN records in a loop (the first four requested ahead, the rest one
  // dependent load after the other): a wavefront took as long as its largest splat, and an optimised SLAM map has a heavy tail
  // -- after 20 frames of mapping 8 % of the Gaussians of the synthetic Replica sequence have 7 .. 64 instances, so nearly every
  // wavefront holds one and ran 16 .. 64 serial trips to memory (152 us against 54 us on the fresh map,
  // gpurun_out/r6/slamlate_b_dens.txt).  Now: the first kGatherAhead records of every lane as before (in flight with the
  // projection's inputs; 93 % of the splats of a fresh map have no more) and the EXCESS records of the wavefront's splats as one
  // list, 64 records per step, every lane one record: position j of the list belongs to the last lane whose first position V is
  // <= j (six-step search over the lanes' scan with ds_bpermute); the lane re-centres its record with the owner's centre and
  // parks the ten sums in a 3 KB LDS window; then every owner adds ITS rows of the window, in index order -- the order the
  // per-lane loop had -- from LDS, not from memory.  No float atomics, no cross-lane float reduction.  (The list for ALL records
  // cost the fresh map 9 us: three windows where four loads in flight did, gpurun_out/r6/timing_g_head.log.)
  // A Gaussian with more than kBigInst instances (a splat grown over a hole of the map has thousands) is summed by the whole
  // wavefront instead, 64 records at a time with a butterfly at the end (its own fixed order).
  constexpr uint32_t kBigInst = 64;
  const bool big = ga.inst_cnt > kBigInst;
  // dual: 14 floats = seven float2 (56-byte stride); single render: 10 floats = five float2 (40-byte stride)
  auto load_record = [&](uint32_t inst, float4& a, float4& b, float4& c, float4& d) {
    if constexpr (B1) {
      const float4* rec = reinterpret_cast<const float4*>(grad_inst) + (size_t)inst * (kGradRecDual1 / 4);
      a = rec[0]; b = rec[1]; c = rec[2]; d = c;
    } else if constexpr (DUAL) {
      const float2* rec = reinterpret_cast<const float2*>(grad_inst) + (size_t)inst * (kGradRecDual / 2);
      const float2 f0 = rec[0], f1 = rec[1], f2 = rec[2], f3 = rec[3], f4 = rec[4], f5 = rec[5], f6 = rec[6];
      a = make_float4(f0.x, f0.y, f1.x, f1.y); b = make_float4(f2.x, f2.y, f3.x, f3.y);
      c = make_float4(f4.x, f4.y, f5.x, f5.y); d = make_float4(f6.x, 0.f, 0.f, 0.f);
    } else {
      const float2* rec = reinterpret_cast<const float2*>(grad_inst) + (size_t)inst * (kGradRec / 2);
      const float2 f0 = rec[0], f1 = rec[1], f2 = rec[2], f3 = rec[3], f4 = rec[4];
      a = make_float4(f0.x, f0.y, f1.x, f1.y); b = make_float4(f2.x, f2.y, f3.x, f3.y);
      c = make_float4(f4.x, f4.y, 0.f, 0.f); d = c;
    }
  };
  // record = tile-local moments (U0, UX, UY, UXX, UXY, UYY), colour sums (3 or 6), tile id; (u, v) = the splat's centre
  auto add_record = [&](SplatMoments& M, float& c0, float& c1, float& c2, float4 a, float4 b, float4 c, float4 d, float u, float v,
                        float ulo, float vlo) {
    uint32_t tile;
    if constexpr (B1) {
      tile = __float_as_uint(c.z);
      c0 += c.y;
    } else if constexpr (DUAL) {
      tile = __float_as_uint(d.x);
      c0 += c.y; c1 += c.z; c2 += c.w;
    } else {
      tile = __float_as_uint(c.y);
    }
    const int ty = (int)(tile / (uint32_t)cam.gx8), tx = (int)(tile - (uint32_t)ty * (uint32_t)cam.gx8);
    const float sx = (u - ((float)(tx * kSubTile) + 3.5f)) + ulo, sy = (v - ((float)(ty * kSubTile) + 3.5f)) + vlo;   // as tile_coefficients
    const float U0 = a.x, UX = a.y, UY = a.z, UXX = a.w, UXY = b.x, UYY = b.y;
    M.m[0] += U0;
    M.m[1] += sx * U0 - UX;                                     // d = centre - pixel = s - X
    M.m[2] += sy * U0 - UY;
    M.m[3] += sx * sx * U0 - 2.f * sx * UX + UXX;
    M.m[4] += sx * sy * U0 - sx * UY - sy * UX + UXY;
    M.m[5] += sy * sy * U0 - 2.f * sy * UY + UYY;
    M.m[6] += b.z; M.m[7] += b.w; M.m[8] += c.x;
  };
  float sc[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, op = 0.f;
  float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, g6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  Splat sp{}; SplatAux aux{};
  SplatMoments mo;
  for (int k = 0; k < 9; ++k) mo.m[k] = 0.f;
  bool ok = false;
  constexpr uint32_t kGatherAhead = 4;
  // the wavefront's list of EXCESS records: lane l's sit at positions [V, V + my_cnt) (wave-level scan; a big splat is not in it)
  constexpr int kRow = 12;                                          // floats per parked record: 9 sums + up to 3 second-set colours
  __shared__ __attribute__((aligned(16))) float lds_rows[4][64 * kRow];
  float* __restrict__ rows = lds_rows[threadIdx.x >> 6];
  const int ln = lane_id();
  const uint32_t my_cnt = (live && !big && ga.inst_cnt > kGatherAhead) ? ga.inst_cnt - kGatherAhead : 0u;
  const uint32_t incl_cnt = wave_incl_scan(my_cnt);
  const uint32_t V = incl_cnt - my_cnt;
  const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl_cnt, 63);
  auto owner_of = [&](uint32_t j) {                                 // last lane whose V <= j (j < T: that lane holds position j)
    int lo = 0, hi = 63;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int mid = (lo + hi + 1) >> 1;
      const uint32_t vm = (uint32_t)__builtin_amdgcn_ds_bpermute(mid << 2, (int)V);
      const bool le = vm <= j;
      lo = le ? mid : lo; hi = le ? hi : mid - 1;
    }
    return lo;
  };
  // the lane's first records are requested up front: their addresses only need gaux, so they are in flight together with the
  // inputs of the projection instead of behind its arithmetic
  float4 pa[kGatherAhead], pb[kGatherAhead], pc[kGatherAhead], pd[kGatherAhead];
  if (ga.inst_cnt && !big) {
#pragma unroll
    for (uint32_t i = 0; i < kGatherAhead; ++i)            // past the end: the last record again (a cache hit), unused
      load_record(ga.inst_base + min(i, ga.inst_cnt - 1u), pa[i], pb[i], pc[i], pd[i]);
  }
  // inputs of the projection; of the projection itself only the pixel centre is formed before the records are summed -- the
  // rest (and with it ~25 live registers: the backward's intermediates) follows behind the record phase
  float mean[3] = {0.f, 0.f, 0.f}, cu = 0.f, cv = 0.f, culo = 0.f, cvlo = 0.f;
  if (ga.inst_cnt) {
    mean[0] = means3D[3 * gid]; mean[1] = means3D[3 * gid + 1]; mean[2] = means3D[3 * gid + 2];
    if constexpr (COV3D) {
      for (int i = 0; i < 6; ++i) c6[i] = scales[6 * gid + i];
      op = opacities[gid];
    } else if (FRAME && cs.raw_act) {
      // the forward's VTGS_FORWARD_RAW_ACTIVATIONS (kernel-uniform): parameters in, activations here; the Gaussian's rotation
      // is read only when somebody wants its gradient (frame flag 1) -- the covariance s^2 I does not depend on it
      sc[0] = sc[1] = sc[2] = __expf(scales[gid]);
      op = 1.f / (1.f + __expf(-opacities[gid]));
      if constexpr (FRAME) {
        if (fe.flags & 1u) {
          const float4 u = reinterpret_cast<const float4*>(fe.unnorm_rot)[gid];
          const float un = rsqrtf(fmaxf(u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w, 1e-24f));
          q[0] = u.x * un; q[1] = u.y * un; q[2] = u.z * un; q[3] = u.w * un;
        }
      }
    } else {
      sc[0] = scales[3 * gid]; sc[1] = scales[3 * gid + 1]; sc[2] = scales[3 * gid + 2];
      const float4 q4 = reinterpret_cast<const float4*>(rotations)[gid];
      q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
      op = opacities[gid];
    }
    pixel_centre(cam, mean[0], mean[1], mean[2], cu, cv, culo, cvlo);
    if (!big) {
#pragma unroll
      for (uint32_t i = 0; i < kGatherAhead; ++i)
        if (i < ga.inst_cnt) add_record(mo, cb0, cb1, cb2, pa[i], pb[i], pc[i], pd[i], cu, cv, culo, cvlo);
    }
  }
  // (the list's first window goes out here, behind the lanes' own records: requested with them it kept sixteen more registers
  //  alive through the projection -- 144-158 VGPRs, three wavefronts per SIMD instead of four)
  float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra, rd = ra;
  int own = 0;
  bool have = (uint32_t)ln < T;
  if (T) {                                                          // wave-uniform
    own = owner_of(have ? (uint32_t)ln : 0u);                       // (every lane takes part in the permutes)
    const uint32_t ob = (uint32_t)__builtin_amdgcn_ds_bpermute(own << 2, (int)ga.inst_base);
    const uint32_t ov = (uint32_t)__builtin_amdgcn_ds_bpermute(own << 2, (int)V);
    if (have) load_record(ob + kGatherAhead + ((uint32_t)ln - ov), ra, rb, rc, rd);
  }
  for (uint32_t j0 = 0; j0 < T; j0 += 64u) {                        // wave-uniform
    {
      // my record of this window, re-centred with ITS owner's centre, parked in row `ln`
      const float ou = __int_as_float(__builtin_amdgcn_ds_bpermute(own << 2, __float_as_int(cu)));
      const float ovv = __int_as_float(__builtin_amdgcn_ds_bpermute(own << 2, __float_as_int(cv)));
      const float oul = __int_as_float(__builtin_amdgcn_ds_bpermute(own << 2, __float_as_int(culo)));
      const float ovl = __int_as_float(__builtin_amdgcn_ds_bpermute(own << 2, __float_as_int(cvlo)));
      SplatMoments one;
      for (int k = 0; k < 9; ++k) one.m[k] = 0.f;
      float o0 = 0.f, o1 = 0.f, o2 = 0.f;
      if (have) add_record(one, o0, o1, o2, ra, rb, rc, rd, ou, ovv, oul, ovl);
      float4* row = reinterpret_cast<float4*>(rows + ln * kRow);
      row[0] = make_float4(one.m[0], one.m[1], one.m[2], one.m[3]);
      row[1] = make_float4(one.m[4], one.m[5], one.m[6], one.m[7]);
      row[2] = make_float4(one.m[8], o0, o1, o2);
    }
    // the next window's records go out before this window's rows are added up
    const uint32_t jn = j0 + 64u + (uint32_t)ln;
    const bool have_n = jn < T;
    if (j0 + 64u < T) {                                             // wave-uniform
      own = owner_of(have_n ? jn : 0u);
      const uint32_t ob = (uint32_t)__builtin_amdgcn_ds_bpermute(own << 2, (int)ga.inst_base);
      const uint32_t ov = (uint32_t)__builtin_amdgcn_ds_bpermute(own << 2, (int)V);
      if (have_n) load_record(ob + kGatherAhead + (jn - ov), ra, rb, rc, rd);
    }
    have = have_n;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // the rows are written (one wavefront: LDS runs in order) ...
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ... and every owner adds its rows of the window, in index order
    const uint32_t r0 = V > j0 ? V - j0 : 0u, r1 = min(V + my_cnt, j0 + 64u);
    for (uint32_t r = r0; r + j0 < r1 && my_cnt; ++r) {
      const float4* row = reinterpret_cast<const float4*>(rows + r * kRow);
      const float4 x0 = row[0], x1 = row[1], x2 = row[2];
      mo.m[0] += x0.x; mo.m[1] += x0.y; mo.m[2] += x0.z; mo.m[3] += x0.w;
      mo.m[4] += x1.x; mo.m[5] += x1.y; mo.m[6] += x1.z; mo.m[7] += x1.w;
      mo.m[8] += x2.x; cb0 += x2.y;
      if constexpr (DUAL && !B1) { cb1 += x2.z; cb2 += x2.w; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // (the next window overwrites the rows)
    __builtin_amdgcn_wave_barrier();
  }
  for (unsigned long long rest = __ballot(big); rest; rest &= rest - 1ull) {     // wave-uniform
    const int src = __builtin_ctzll(rest);
    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)ga.inst_base, src);
    const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)ga.inst_cnt, src);
    const float u = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cu), src));
    const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), src));
    const float ulo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(culo), src));
    const float vlo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cvlo), src));
    SplatMoments part;
    for (int k = 0; k < 9; ++k) part.m[k] = 0.f;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    for (uint32_t i = (uint32_t)lane_id(); i < cnt; i += 64u) {   // lane-strided partial sums, then a butterfly: fixed order
      float4 a, b, c, d;
      load_record(base + i, a, b, c, d);
      add_record(part, p0, p1, p2, a, b, c, d, u, v, ulo, vlo);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) part.m[k] = wave_sum(part.m[k]);
    if constexpr (DUAL) { p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2); }
    if (lane_id() == src) { mo = part; cb0 = p0; cb1 = p1; cb2 = p2; }
  }
  if (ga.inst_cnt) {
    const float centre4[4] = {cu, cv, culo, cvlo};                                            // (formed above: not computed twice)
    ok = project_splat(cam, mean, sc, q, op, sp, aux, COV3D ? c6 : nullptr, centre4);
  }
  if (ok) {
    if (moments_scaled_by_opacity) {        // the matrix-core backward accumulates u' = o*u
      const float io = 1.f / op;
      for (int k = 0; k < 6; ++k) mo.m[k] *= io;
    }
    splat_backward(cam, sc, q, op, sp, aux, mo, g, COV3D ? g6 : nullptr);
  }
  if constexpr (FRAME) {
    __shared__ float red[4][12];
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    if (live) {
      const FramePose P = load_pose(fe.cam_q, fe.cam_t, fe.depth_w2c);
      const int row = fe.idx ? fe.idx[gid] : gid;                              // owned sets: the Gaussian's row in the map
      const float x = fe.means3D_world[3 * row], y = fe.means3D_world[3 * row + 1], z = fe.means3D_world[3 * row + 2];
      float cx, cy, cz, zz;
      pose_apply(P, x, y, z, cx, cy, cz, zz);
      const float dz = cb0 + 2.f * zz * cb2;                                   // colours of the second render: [z, 1, z^2]
      const float g0 = g.mean3D[0] + dz * P.zr[0], g1 = g.mean3D[1] + dz * P.zr[1], g2 = g.mean3D[2] + dz * P.zr[2];
      if (fe.flags & 1u) {
        fe.g_means3D[3 * row] = P.R[0] * g0 + P.R[3] * g1 + P.R[6] * g2;
        fe.g_means3D[3 * row + 1] = P.R[1] * g0 + P.R[4] * g1 + P.R[7] * g2;
        fe.g_means3D[3 * row + 2] = P.R[2] * g0 + P.R[5] * g1 + P.R[8] * g2;
        const float4 u = reinterpret_cast<const float4*>(fe.unnorm_rot)[row];
        const float un = rsqrtf(fmaxf(u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w, 1e-24f));
        const float r[4] = {u.x * un, u.y * un, u.z * un, u.w * un};
        const float dot = r[0] * g.rot[0] + r[1] * g.rot[1] + r[2] * g.rot[2] + r[3] * g.rot[3];
        reinterpret_cast<float4*>(fe.g_unnorm_rot)[row] = make_float4((g.rot[0] - r[0] * dot) * un, (g.rot[1] - r[1] * dot) * un,
                                                                       (g.rot[2] - r[2] * dot) * un, (g.rot[3] - r[3] * dot) * un);
      }
      if (fe.flags & 4u) {
        // (op, sc[0] = sigmoid(logit), exp(log-scale): the forward's values, loaded or formed above; a Gaussian without
        //  instances never loaded them and has zero gradients)
        fe.g_logit[row] = g.opacity * op * (1.f - op);
        fe.g_log_scales[row] = sc[0] * (g.scale[0] + g.scale[1] + g.scale[2]);
        fe.g_rgb[3 * row] = g.color[0]; fe.g_rgb[3 * row + 1] = g.color[1]; fe.g_rgb[3 * row + 2] = g.color[2];
      }
      if (fe.flags & 2u) {
        acc[0] = g0; acc[1] = g1; acc[2] = g2;
        acc[3] = g0 * x; acc[4] = g0 * y; acc[5] = g0 * z;
        acc[6] = g1 * x; acc[7] = g1 * y; acc[8] = g1 * z;
        acc[9] = g2 * x; acc[10] = g2 * y; acc[11] = g2 * z;
      }
    }
    if (fe.flags & 2u) {                               // fixed-order block reduction, as in prepare_frame_backward_kernel
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[k] = wave_sum(acc[k]);
      const int wv = (int)(threadIdx.x >> 6), l = lane_id();
      if (l == 0)
        for (int k = 0; k < 12; ++k) red[wv][k] = acc[k];
      __syncthreads();
      if (threadIdx.x < 12) fe.pose_partials[(size_t)blk * 12 + threadIdx.x] =
          red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
    return;
  }
  if (!live) return;
  // an output nobody asked for is NULL (wave-uniform tests): 68 bytes per Gaussian when all six are wanted, 24 in a
  // tracking iteration of the unfused loops (means3D + the screen-space term)
  if (g_means3D) { for (int i = 0; i < 3; ++i) g_means3D[3 * gid + i] = g.mean3D[i]; }
  if (g_means2D) { for (int i = 0; i < 3; ++i) g_means2D[3 * gid + i] = g.mean2D[i]; }
  if (g_colors) { for (int i = 0; i < 3; ++i) g_colors[3 * gid + i] = g.color[i]; }
  if constexpr (COV3D) {
    if (g_scales) { for (int i = 0; i < 6; ++i) g_scales[6 * gid + i] = g6[i]; }
  } else {
    if (g_scales) { for (int i = 0; i < 3; ++i) g_scales[3 * gid + i] = g.scale[i]; }
    if (g_rotations) reinterpret_cast<float4*>(g_rotations)[gid] = make_float4(g.rot[0], g.rot[1], g.rot[2], g.rot[3]);
  }
  if (g_opacities) g_opacities[gid] = g.opacity;
  if constexpr (DUAL) { if (g_colors_b) { g_colors_b[3 * gid] = cb0; g_colors_b[3 * gid + 1] = cb1; g_colors_b[3 * gid + 2] = cb2; } }
}
template __global__ void gather_splat_grads<false>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, const GaussAux*, const float*, int, float*, float*, float*, float*, float*, float*, const Counters*, float*, FrameEpilogue);
template __global__ void gather_splat_grads<true>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, const GaussAux*, const float*, int, float*, float*, float*, float*, float*, float*, const Counters*, float*, FrameEpilogue);
template __global__ void gather_splat_grads<true, true>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, const GaussAux*, const float*, int, float*, float*, float*, float*, float*, float*, const Counters*, float*, FrameEpilogue);
template __global__ void gather_splat_grads<true, true, false, true>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, const GaussAux*, const float*, int, float*, float*, float*, float*, float*, float*, const Counters*, float*, FrameEpilogue);
template __global__ void gather_splat_grads<false, false, true>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, const GaussAux*, const float*, int, float*, float*, float*, float*, float*, float*, const Counters*, float*, FrameEpilogue);

__global__ __launch_bounds__(256) void mark_visible_kernel(const float* __restrict__ Vp, int n,
                                                           const float* __restrict__ means3D, uint8_t* __restrict__ out) {
  const int gid = (int)(blockIdx.x * 256u + threadIdx.x);
  if (gid >= n) return;
  const float x = means3D[3 * gid], y = means3D[3 * gid + 1], z = means3D[3 * gid + 2];
  const float tz = fmaf(Vp[2], x, fmaf(Vp[6], y, fmaf(Vp[10], z, Vp[14])));
  out[gid] = tz > kNearCull ? 1 : 0;
}

#endif  // !VTGS_XCHECK_BUILD

}  // namespace vtgs
