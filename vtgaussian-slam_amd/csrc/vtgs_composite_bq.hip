// vtgs_composite_bq.hip -- backward composite with per-quadrant splat queues (gfx950, wave64).
//
// composite_backward_mx (vtgs_composite.hip) replays every splat of an 8x8 tile's list at all 64 pixels: 0.18 G
// (pixel, splat) pairs at the headline shape, ~85 % of them below alpha = 1/255.  Here, as in composite_forward_q, the 16
// lanes of a 4x4 QUADRANT walk their own queue of the tile's list -- the entries whose alpha >= 1/255 box reaches the
// quadrant (the 4-bit masks composite_forward_q left in the workspace; recomputed here when another forward ran) -- so a
// step is 64 pixels x the next 16 splats OF EACH QUADRANT: 8.5 steps per tile instead of 14.4 batches.
//
//   * lane L <-> pixel (4 (q & 1) + (i & 3), 4 (q >> 1) + (i >> 2)), q = L >> 4, i = L & 15.  Lane (q, j) also OWNS the j-th
//     entry its quadrant pops in a step: it gathers that splat's geometry record and colour itself (requested one step
//     ahead), so the exponent and g.c matrix instructions (v_mfma_f32_4x4x1, cbsz = 2: A from lanes 4g..4g+3 of the
//     lane's own 16-lane group) read their operands from registers -- no coefficient table in LDS;
//   * the sweeps are those of px_backward_batch (front to back: alpha, w = alpha T, T, P; one anchor A = (CB - P) / T per
//     step; back to front: u' = G T (g.c - A), A += alpha (g.c - A));
//   * the pixel contraction per quadrant: [16 splats x 16 px] x [16 px x 9] on v_mfma_f32_4x4x1 -- the quadrant's four
//     blocks are its four splat groups, so there is no cross-lane sum afterwards -- through ONE 4 KB LDS image per
//     wavefront (XOR-swizzled rows, no padding) that w and u' use one after the other: the front sweep writes w, the colour
//     products read it, the back sweep overwrites it with u' (LDS operations of a wavefront execute in order);
//   * a splat's sums arrive from up to four quadrants in different steps: they are added into per-wavefront LDS accumulators
//     (three row-contiguous arrays, 9 floats per ring slot) and leave as ONE 40-byte record per (splat, tile) instance -- the
//     format gather_splat_grads reads -- when the chunk retires: when all four queues have popped its last entry, the same
//     invariant that frees the chunk's ring slots.  Round 4: the add is a plain read-modify-write, one quadrant after the other
//     (the contraction runs with the column operand first, so a lane holds all nine sums of the splat it popped itself);
//     round 3 used ds_add_f32, which gfx950 applies one lane at a time -- 768 cycles per wave-wide instruction
//     (profiles/r4_lds_accumulate.md): 445 us then, 194 us now.  Fixed order (quadrants 0..3 within a step, steps in
//     sequence, the step sequence of a tile a function of its list alone): bitwise reproducible run to run.
//
// Semantics: SURVEY.md Appendix A4 as restated in vtgs_composite.hip (the recurrences of px_backward_batch).
#include "vtgs_internal.h"
#include "vtgs_composite_common.h"

#ifndef VTGS_XCHECK_BUILD
#define VTGS_XCHECK_BUILD 0
#endif
#if VTGS_XCHECK_BUILD                          // a cross-check implementation: test-only library (csrc/vtgs_xcheck.hip)

#ifndef VTGS_BQ_CHUNKS
#define VTGS_BQ_CHUNKS 2
#endif

namespace vtgs {

constexpr int kBqChunks = VTGS_BQ_CHUNKS;      // 64-entry chunks of the list in flight
constexpr int kBqRing = 64 * kBqChunks;        // ring slots (accumulators, ids, queue capacity)
constexpr int kBqDummy = kBqRing;              // slot of a lane that popped past the end of its queue

__device__ __forceinline__ uint32_t quadrant_mask_bq(const float4& g0, const float4& g1, float sx, float sy) {
  // the same box test as composite_forward_q (vtgs_composite_q.hip): only used when that kernel did not leave its masks
  float tau = (__log2f(g1.y) + 7.99435344f) * 0.69314718f;               // ln(255 o)
  tau += 1e-4f * tau + 1e-4f;
  // (hardware reciprocal and square root, 1 ulp each: the IEEE forms expand to ~35 instructions per list entry and append;
  //  the box only has to be conservative, and it carries 1e-6 relative + 1e-5 px of slack)
  const float idet = __builtin_amdgcn_rcpf(fmaxf(g0.z * g1.x - g0.w * g0.w, 1e-30f));
  const float k2 = 2.f * fmaxf(tau, 0.f) * idet;
  const float hx = __builtin_amdgcn_sqrtf(k2 * g1.x) * 1.000002f + 1e-5f, hy = __builtin_amdgcn_sqrtf(k2 * g0.z) * 1.000002f + 1e-5f;
  const bool left = sx - hx <= -0.5f && sx + hx >= -3.5f, right = sx + hx >= 0.5f && sx - hx <= 3.5f;
  const bool top = sy - hy <= -0.5f && sy + hy >= -3.5f, bottom = sy + hy >= 0.5f && sy - hy <= 3.5f;
  return (left && top ? 1u : 0u) | (right && top ? 2u : 0u) | (left && bottom ? 4u : 0u) | (right && bottom ? 8u : 0u);
}

// exponents / g.c of the lane's pixel x the four splats held by lanes 4G..4G+3 of the lane's own 16-lane group
template <int G>
__device__ __forceinline__ f32x4 bq_exponents(const float (&K)[6], const float (&Phi)[6]) {
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < 6; ++m) d = __builtin_amdgcn_mfma_f32_4x4x1f32(K[m], Phi[m], d, 2, G, 0);
  return d;
}
template <int G>
__device__ __forceinline__ f32x4 bq_gdotc(const float (&col)[3], const float (&g)[3]) {
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c) d = __builtin_amdgcn_mfma_f32_4x4x1f32(col[c], g[c], d, 2, G, 0);
  return d;
}

struct BqPixel { float T, P, CB; bool done; };   // T frozen once the pixel has ended (as in the forward)

// image [quadrant][splat row 0..15][16 px], rows XOR-swizzled in 4-float granules: pixel granule g of row k sits at
// granule g ^ (k >> 2), which makes the column-wise ds_write_b32 and the row-wise ds_read_b128 conflict-free without padding
__device__ __forceinline__ int img_read_off(int q, int row, int t4) { return q * 256 + row * 16 + 4 * (t4 ^ (row >> 2)); }

// The sweeps of one step for the lane's pixel x the 16 splats its quadrant popped: the recurrences of px_backward_batch
// (vtgs_composite.hip).  Front to back: alpha_k, w_k = alpha_k T_k -> image row k, T, P.  `wbase[g]` = the lane's image
// offset for splat group g (rows 4g..4g+3 are 16 floats apart).  Returns with a[], gT[], the anchor in A.
template <bool CLAMP, bool EXACT_FIRST>
__device__ __forceinline__ void bq_front(BqPixel& px, bool& exact, const f32x4 (&d)[4], const f32x4 (&gcv)[4], float* __restrict__ img,
                                         const int (&wbase)[4], float (&a)[16], float (&gT)[16], float& A) {
  float Pn = px.P;
  bool swept = false;
  const bool was_done = px.done;
  if constexpr (!EXACT_FIRST) {
    float Tn = px.done ? 0.f : px.T;                             // optimistic: no stop test
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float Gp = __builtin_amdgcn_exp2f(d[k >> 2][k & 3]);
      float w;
      if constexpr (CLAMP) {
        const float al = fminf(kAlphaMax, Gp);
        const bool valid = al >= kAlphaMin;
        a[k] = valid ? al : 0.f;
        gT[k] = valid ? Gp * Tn : 0.f;                           // the 0.99 clamp passes the gradient through
        w = a[k] * Tn;
      } else {                                                   // no splat of this step can reach the clamp
        a[k] = (Gp >= kAlphaMin) ? Gp : 0.f;
        w = a[k] * Tn;
        gT[k] = w;
      }
      img[wbase[k >> 2] + 16 * (k & 3)] = w;
      Pn = fmaf(gcv[k >> 2][k & 3], w, Pn);
      Tn = Tn - w;
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 36, 0);          // exponent + g.c MFMAs back to back, then the sweep
    __builtin_amdgcn_sched_group_barrier(0x302, 400, 0);
    if (__ballot(!px.done && Tn < kTStop) == 0ull) { px.T = px.done ? px.T : Tn; swept = true; }
    else Pn = px.P;
  }
  if (!swept) {
    // exact: the first splat with T (1 - alpha) < 1e-4 ends the pixel BEFORE it is added; from there on alpha = G T = w = 0
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float Gp = __builtin_amdgcn_exp2f(d[k >> 2][k & 3]);
      const float al = CLAMP ? fminf(kAlphaMax, Gp) : Gp;
      const float av = (al >= kAlphaMin) ? al : 0.f;
      const float wk = av * px.T;
      const float tn = px.T - wk;
      const bool stop = tn < kTStop;                             // a live pixel has T >= 1e-4: alpha = 0 cannot trigger it
      const bool live = !px.done && !stop;
      a[k] = live ? av : 0.f;
      gT[k] = live ? ((al >= kAlphaMin) ? Gp * px.T : 0.f) : 0.f;
      const float w = live ? wk : 0.f;
      img[wbase[k >> 2] + 16 * (k & 3)] = w;
      Pn = fmaf(gcv[k >> 2][k & 3], w, Pn);
      px.T = live ? tn : px.T;
      px.done = px.done || stop;
    }
    exact = __builtin_popcountll(__ballot(px.done && !was_done)) >= kExactFirstEndings;   // as in the forward
  }
  // anchor: colour behind the step (behind the ending splat for an ended pixel) per unit of transmittance there; T > 0
  A = (px.CB - Pn) * __builtin_amdgcn_rcpf(px.T);
  px.P = Pn;
}

constexpr int kPhiRow = 20;                    // Phi table row: 16 pixels + 4 pad (bank spread of the b128 row reads)

__device__ __forceinline__ int bq_wrap(int x) { return x >= kBqRing ? x - kBqRing : x; }   // x < 2 kBqRing

template <int R>
__device__ __forceinline__ int quad_bcast(int v) {     // value of lane (l & ~3) + R, in every lane of the 4-lane quad
  return __builtin_amdgcn_mov_dpp(v, R | (R << 2) | (R << 4) | (R << 6), 0xf, 0xf, true);
}

__global__ __launch_bounds__(256, 3) void composite_backward_q(
    CamScalars cs, const float* __restrict__ bg, uint32_t nblk,
    const uint32_t* __restrict__ tile_cnt, uint32_t tile_cap, const uint32_t* __restrict__ sorted_gid,
    const uint32_t* __restrict__ sorted_inst, const uint8_t* __restrict__ qmask, const GeomRec* __restrict__ geom,
    const float* __restrict__ colors, const float* __restrict__ out_color, const float* __restrict__ grad_color,
    const float* __restrict__ final_T, float* __restrict__ grad_inst, const Counters* __restrict__ ctr,
    uint32_t* __restrict__ step_counters) {
  __shared__ __attribute__((aligned(16))) float lds_img[4][4 * 256];          // one image per wavefront: u', then w
  // per ring slot: U0 UX UY UXX | UXY UYY | c0 c1 c2 (pad) -- three arrays so that one lane moves one splat's sums with
  // ds_read/write_b128 + _b64 + _b128.  Round 4: plain read-modify-write, one quadrant after the other (see the merge below);
  // ds_add_f32 costs ~768 cycles per wave-wide instruction on gfx950 -- LDS float atomics are applied one lane at a time,
  // 12 cycles each, whatever the addresses (tests/micro/lds_accumulate.hip, profiles/r4_lds_accumulate.md)
  __shared__ __attribute__((aligned(16))) float4 lds_accA[4][kBqRing + 1];
  __shared__ __attribute__((aligned(16))) float2 lds_accB[4][kBqRing + 1];
  __shared__ __attribute__((aligned(16))) float4 lds_accW[4][kBqRing + 1];
  __shared__ uint32_t lds_gid[4][kBqRing + 1], lds_inst[4][kBqRing + 1];
  __shared__ uint8_t lds_q[4][4][kBqRing];
  __shared__ __attribute__((aligned(16))) float lds_phi[4 * 8 * kPhiRow];     // [quadrant][column 0..7][16 px + 4 pad]
  __shared__ __attribute__((aligned(16))) float lds_g[4][4 * 4 * kPhiRow];   // per wavefront: dL/dcolor [quadrant][channel][16 px + 4 pad]
  // Everything this wavefront needs from memory before its first step is requested HERE, in one go -- the flags, its list
  // length, its pixels' image values, the first list entries -- and only then are the flags looked at: the prologue was four
  // dependent round trips to L2 / HBM (flag, length, images, list) and 15 % of the wavefront's life
  // (profiles/r3_backward_stamps.md).
  const int gx16 = (cs.W + kBinTile - 1) / kBinTile;
  const int gx8 = (cs.W + kSubTile - 1) / kSubTile, gy8 = (cs.H + kSubTile - 1) / kSubTile;
  const uint32_t blk = xcd_swizzle(blockIdx.x, nblk);
  const int l = lane_id();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // workgroup = the 2x2 tiles of a 16x16 block
  const int row16_0 = cs.row8_begin >> 1;
  const int t8x = 2 * (int)(blk % (uint32_t)gx16) + (wv & 1), t8y = 2 * (row16_0 + (int)(blk / (uint32_t)gx16)) + (wv >> 1);
  const bool tile_ok = t8x < gx8 && t8y < gy8 && t8y >= cs.row8_begin && t8y < cs.row8_end;   // wave-uniform
  const int tile = tile_ok ? t8y * gx8 + t8x : 0;
  const int q = l >> 4, i = l & 15;
  const int tx0 = t8x * kSubTile, ty0 = t8y * kSubTile;
  const int lx = 4 * (q & 1) + (i & 3), ly = 4 * (q >> 1) + (i >> 2);
  const int pxx = tx0 + lx, pyy = ty0 + ly;
  const bool inside = tile_ok && pxx < cs.W && pyy < cs.H;
  const size_t P = (size_t)cs.W * cs.H;
  const uint32_t overflow = ctr->overflow, masks_valid = ctr->qmask_valid;
  const BinRange br = bin_range(cs, (uint32_t)tile, tile_cap);
  const uint32_t s = br.s;
  const uint32_t list_len = min(tile_cnt[tile], br.cap);
  float gown[3] = {0.f, 0.f, 0.f}, oc[3] = {0.f, 0.f, 0.f}, Tf = 0.f;
  if (inside) {
    const size_t pix = (size_t)pyy * cs.W + pxx;
    gown[0] = grad_color[pix]; gown[1] = grad_color[P + pix]; gown[2] = grad_color[2 * P + pix];
    oc[0] = out_color[pix]; oc[1] = out_color[P + pix]; oc[2] = out_color[2 * P + pix];
    Tf = final_T[pix];
  }
  const int cj = l & 3, row = l & 15;
  // first list entries (a bin holds at least 64 slots: reading past a short list stays inside the bin; masked later)
  const bool have_mask_array = qmask != nullptr;
  uint32_t gid_n = sorted_gid[s + (uint32_t)l], inst_n = sorted_inst[s + (uint32_t)l];
  uint32_t mask_n = have_mask_array ? (uint32_t)qmask[s + (uint32_t)l] : 0u;

  // Phi table for the contraction's B operands: columns (1, X, Y, X^2, XY, Y^2, 0, 0) at the quadrant's 16 pixels, X, Y
  // relative to the TILE centre (the records carry tile-centred moments, as composite_backward_mx writes them)
  for (int idx = (int)threadIdx.x; idx < 4 * 8 * kPhiRow; idx += 256) {
    const int qq = idx / (8 * kPhiRow), c = (idx - qq * 8 * kPhiRow) / kPhiRow, t = idx - qq * 8 * kPhiRow - c * kPhiRow;
    const float PX = (float)(4 * (qq & 1) + (t & 3)) - 3.5f, PY = (float)(4 * (qq >> 1) + ((t >> 2) & 3)) - 3.5f;
    float v = 0.f;
    v = (c == 0) ? 1.f : v; v = (c == 1) ? PX : v; v = (c == 2) ? PY : v;
    v = (c == 3) ? PX * PX : v; v = (c == 4) ? PX * PY : v; v = (c == 5) ? PY * PY : v;
    lds_phi[idx] = (t < 16) ? v : 0.f;
  }
  float* __restrict__ img = lds_img[wv];
  float4* __restrict__ accA = lds_accA[wv];
  float2* __restrict__ accB = lds_accB[wv];
  float4* __restrict__ accW = lds_accW[wv];
  uint32_t* __restrict__ tgid = lds_gid[wv];
  uint32_t* __restrict__ tinst = lds_inst[wv];
  // dL/dcolor as the B operand of the colour contraction: [quadrant][channel 0..2, 3 = 0][16 px + 4 pad], from the lanes' own
  // pixel gradients (no second trip to the image)
  float* __restrict__ gimg = lds_g[wv];
#pragma unroll
  for (int c = 0; c < 4; ++c) gimg[(4 * q + c) * kPhiRow + i] = (c < 3) ? gown[c < 3 ? c : 0] : 0.f;
  // ring state: accumulators start at zero (and are zeroed again when a chunk retires); dummy slot included
  for (int k = l; k < kBqRing + 1; k += 64) {
    accA[k] = make_float4(0.f, 0.f, 0.f, 0.f); accB[k] = make_float2(0.f, 0.f); accW[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (l == 0) { tgid[kBqDummy] = 0u; tinst[kBqDummy] = 0u; }
  __syncthreads();                                              // before any per-wavefront exit
  if (overflow) return;                                         // uniform over the grid: the forward did not complete
  if (!tile_ok || list_len == 0u) return;
  const uint32_t e = s + list_len;
  // the speculative first read went past the end of a short list, into slots nobody wrote: those lanes take entry 0 instead
  // (every id that is gathered through must be a written one)
  {
    const uint32_t gid0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)gid_n);      // lane 0 (all lanes active here): entry 0
    const uint32_t inst0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)inst_n);
    const bool first_in = (uint32_t)l < list_len;
    gid_n = first_in ? gid_n : gid0;
    inst_n = first_in ? inst_n : inst0;
    mask_n = first_in ? mask_n : 0u;
  }
  const float cx = (float)tx0 + 3.5f, cy = (float)ty0 + 3.5f;
  const float X = (float)lx - 3.5f, Y = (float)ly - 3.5f;
  const float Phi[6] = {1.f, X, Y, X * X, X * Y, Y * Y};
  BqPixel px{1.f, 0.f, 0.f, !inside};
  {
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    px.CB = gown[0] * (oc[0] - Tf * b0) + gown[1] * (oc[1] - Tf * b1) + gown[2] * (oc[2] - Tf * b2)
            + Tf * (gown[0] * b0 + gown[1] * b1 + gown[2] * b2);
  }
  // Contraction roles: l = cj + 4 sg + 16 q -- column cj of every chain, splat group sg of quadrant q.  A operand: row
  // (l & 15) = 4 sg + cj of the quadrant's image; B operand: column cj (chain a: Phi 0..3, chain b: Phi 4, 5, chain w: dL/dcolor).
  const float4* __restrict__ PhiA4 = reinterpret_cast<const float4*>(lds_phi + (8 * q + cj) * kPhiRow);
  const float4* __restrict__ PhiB4 = reinterpret_cast<const float4*>(lds_phi + (8 * q + 4 + cj) * kPhiRow);
  const float4* __restrict__ G4 = reinterpret_cast<const float4*>(gimg + (4 * q + cj) * kPhiRow);
  // the lane's image offsets for the four splat groups: row k = 4 g + r at wbase[g] + 16 r (swizzle: granule ^ (k >> 2) = ^ g)
  int wbase[4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) wbase[gq] = q * 256 + 64 * gq + 4 * ((i >> 2) ^ gq) + (i & 3);
  bool px_exact = false;
  const uint32_t tile_bits = (uint32_t)tile;
  const bool have_masks = have_mask_array && masks_valid != 0u;               // uniform

  int head_v = 0, c0 = 0, c1 = 0, c2 = 0;                        // the lane's OWN queue: head, queued entries by chunk in flight
  int inflight = 0, n0 = 0, n1 = 0, n2 = 0;                      // chunks in flight and their lengths (wave-uniform)
  uint32_t base = s, wslot = 0u, rslot = 0u, nsteps = 0u;       // next chunk to append goes to slots wslot.., oldest sits at rslot..
  // list entries one chunk ahead (a lane past the end reads entry 0 of its own bin)
  auto lpos = [&](uint32_t b) { const uint32_t p = b + (uint32_t)l; return p < e ? p : s; };

  // the step in flight: cur = popped and gathered, to be computed; nxt = popped, gathers in flight
#ifdef VTGS_Q_STAMPS
  const unsigned long long st0 = __builtin_amdgcn_s_memtime();
  unsigned long long st_app = 0ull, st_pop = 0ull, st_sweep = 0ull, st_contr = 0ull, st_acc = 0ull, st_prom = 0ull, st_ret = 0ull;
#define BQ_STAMP(var) { asm volatile("" ::: "memory"); const unsigned long long now__ = __builtin_amdgcn_s_memtime(); var += now__ - tlast; tlast = now__; }
  unsigned long long tlast = st0;
#else
#define BQ_STAMP(var)
#endif
  bool have_cur = false;
  float curK[6] = {-1e30f, 0.f, 0.f, 0.f, 0.f, 0.f}, curC[3] = {0.f, 0.f, 0.f};
  int cur_slot = kBqDummy, cur_d0 = 0, cur_d1 = 0, cur_d2 = 0;   // entries of the current step by chunk in flight (oldest first)
  bool cur_hot = false;

  for (;;) {
    const bool all_done = __ballot(!px.done) == 0ull;
    if (all_done) break;                                        // every pixel of the tile has ended: nothing more can contribute
    if (inflight < kBqChunks && base < e) {
      // ---- append the next 64-entry chunk: ids into the ring, wavefront-ballot compaction into the four queues ----------
      const uint32_t pos = base + (uint32_t)l;
      const bool in = pos < e;
      const uint32_t gid = gid_n, inst = inst_n;
      uint32_t mask = mask_n;
      gid_n = sorted_gid[lpos(base + 64u)]; inst_n = sorted_inst[lpos(base + 64u)];
      if (have_masks) {
        mask_n = (uint32_t)qmask[lpos(base + 64u)];
      } else {                                                  // another forward ran: the box test of composite_forward_q, here
        const float4* gp = reinterpret_cast<const float4*>(geom + gid);
        const float4 g0 = gp[0], g1 = gp[1];
        mask = quadrant_mask_bq(g0, g1, g0.x - cx, g0.y - cy);
      }
      mask = in ? mask : 0u;
      const int slot = (int)wslot + l;
      tgid[slot] = gid; tinst[slot] = inst;
      const int tail_v = head_v + c0 + c1 + c2;
      int add_v = 0;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const bool in_q = (mask >> qq) & 1u;
        const unsigned long long bal = __ballot(in_q);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        const int tail = __builtin_amdgcn_readlane(tail_v, 16 * qq);
        if (in_q) lds_q[wv][qq][bq_wrap(bq_wrap(tail + rank))] = (uint8_t)slot;
        const int add = (int)__builtin_popcountll(bal);
        add_v = (q == qq) ? add : add_v;
      }
      const int len = (int)min(64u, e - base);
      if (inflight == 0) { c0 += add_v; n0 = len; }
      else if (inflight == 1) { c1 += add_v; n1 = len; }
      else { c2 += add_v; n2 = len; }
      base += 64u; wslot = (wslot == (uint32_t)(kBqRing - 64)) ? 0u : wslot + 64u; ++inflight;
      BQ_STAMP(st_app)
      continue;
    }
    // ---- pop the NEXT step (min(16, count) entries from every queue) and request its splats: lane (q, j) owns entry j ----
    const int cnt = c0 + c1 + c2;
    const int avail = min(16, cnt);
    const bool nxt_valid = i < avail;
    int nxt_slot = kBqDummy;
    if (nxt_valid) nxt_slot = (int)lds_q[wv][q][bq_wrap(head_v + i)];
    const uint32_t ngid = tgid[nxt_slot];
    const float4* ngp = reinterpret_cast<const float4*>(geom + ngid);
    const float4 ng0 = ngp[0], ng1 = ngp[1];
    const float nc0 = colors[3 * ngid], nc1 = colors[3 * ngid + 1], nc2 = colors[3 * ngid + 2];
    head_v = bq_wrap(head_v + avail);
    int nxt_d0, nxt_d1, nxt_d2;
    {
      int t = avail;
      nxt_d0 = min(t, c0); c0 -= nxt_d0; t -= nxt_d0;
      nxt_d1 = min(t, c1); c1 -= nxt_d1; t -= nxt_d1;
      nxt_d2 = t; c2 -= t;
    }
    const bool have_nxt = __ballot(nxt_valid) != 0ull;
    BQ_STAMP(st_pop)

    if (have_cur) {
      // ---- compute the current step: the lane's pixel x the 16 splats of its quadrant ---------------------------------------
      ++nsteps;
      const bool hot = __ballot(cur_hot) != 0ull;               // wave-uniform: some splat of this step may hit the 0.99 clamp
      const f32x4 d[4] = {bq_exponents<0>(curK, Phi), bq_exponents<1>(curK, Phi), bq_exponents<2>(curK, Phi), bq_exponents<3>(curK, Phi)};
      const f32x4 gcv[4] = {bq_gdotc<0>(curC, gown), bq_gdotc<1>(curC, gown), bq_gdotc<2>(curC, gown), bq_gdotc<3>(curC, gown)};
      float a[16], gT[16], A;
      if (px_exact) bq_front<true, true>(px, px_exact, d, gcv, img, wbase, a, gT, A);     // (one exact-first body: the clamped form is always valid)
      else if (hot) bq_front<true, false>(px, px_exact, d, gcv, img, wbase, a, gT, A);
      else bq_front<false, false>(px, px_exact, d, gcv, img, wbase, a, gT, A);
      BQ_STAMP(st_sweep)
      // ---- colour sums: w x dL/dcolor over the quadrant's 16 pixels (the image holds w now) ---------------------------------
      f32x4 Pw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {                            // two pixel granules at a time: 16 operand registers, not 32
        float4 wa[2], ga[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          wa[t] = *reinterpret_cast<const float4*>(img + img_read_off(q, row, 2 * h2 + t));
          ga[t] = G4[2 * h2 + t];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float wav[4] = {wa[t].x, wa[t].y, wa[t].z, wa[t].w}, gav[4] = {ga[t].x, ga[t].y, ga[t].z, ga[t].w};
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) Pw = __builtin_amdgcn_mfma_f32_4x4x1f32(gav[e4], wav[e4], Pw, 0, 0, 0);
        }
      }
      // ---- back to front: u'_k = G_k T_k (g.c_k - A), A <- A + alpha_k (g.c_k - A); u' takes the image over from w -----------
      // (LDS operations of one wavefront execute in order: these stores cannot overtake the row reads above)
#pragma unroll
      for (int k = 15; k >= 0; --k) {
        const float t = gcv[k >> 2][k & 3] - A;
        img[wbase[k >> 2] + 16 * (k & 3)] = gT[k] * t;
        A = fmaf(a[k], t, A);
      }
      f32x4 Pa = {0.f, 0.f, 0.f, 0.f}, Pb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        float4 ua[2], pa4[2], pb4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          ua[t] = *reinterpret_cast<const float4*>(img + img_read_off(q, row, 2 * h2 + t));
          pa4[t] = PhiA4[2 * h2 + t]; pb4[t] = PhiB4[2 * h2 + t];
        }
#pragma unroll
        for (int t4 = 0; t4 < 2; ++t4) {
          const float uav[4] = {ua[t4].x, ua[t4].y, ua[t4].z, ua[t4].w};
          const float bav[4] = {pa4[t4].x, pa4[t4].y, pa4[t4].z, pa4[t4].w}, bbv[4] = {pb4[t4].x, pb4[t4].y, pb4[t4].z, pb4[t4].w};
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            Pa = __builtin_amdgcn_mfma_f32_4x4x1f32(bav[e4], uav[e4], Pa, 0, 0, 0);
            Pb = __builtin_amdgcn_mfma_f32_4x4x1f32(bbv[e4], uav[e4], Pb, 0, 0, 0);
          }
        }
      }
#ifdef VTGS_Q_STAMPS
      asm volatile("" :: "v"(Pa[0]), "v"(Pb[0]), "v"(Pw[0]));
#endif
      BQ_STAMP(st_contr)
      // ---- add the quadrant's partial sums to the splats' ring accumulators.  With the column operand first, lane (q, j) holds
      // all nine sums of ITS OWN splat (row j = the entry it popped): Pa = U0 UX UY UXX, Pb = UXY UYY, Pw = c0 c1 c2.  Two
      // quadrants can hold the same splat in the same step, so the quadrants take turns -- four exec-masked read-modify-write
      // rounds; the LDS operations of one wavefront execute in order, a round reads what the round before it wrote.  Fixed
      // order (0, 1, 2, 3 within a step, steps in sequence): bitwise reproducible.
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        if (q == qq) {
          float4 a = accA[cur_slot];
          float2 b = accB[cur_slot];
          float4 w = accW[cur_slot];
          a.x += Pa[0]; a.y += Pa[1]; a.z += Pa[2]; a.w += Pa[3];
          b.x += Pb[0]; b.y += Pb[1];
          w.x += Pw[0]; w.y += Pw[1]; w.z += Pw[2];
          accA[cur_slot] = a; accB[cur_slot] = b; accW[cur_slot] = w;
        }
        // the next round's loads must stay BEHIND this round's stores: to the compiler the rounds are mutually exclusive
        // branches of one thread (it would hoist or merge the loads: lost updates -- the first build of this merge did
        // exactly that); a wavefront-scope fence costs no instruction, the hardware keeps one wavefront's LDS operations in order
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
#ifdef VTGS_Q_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      BQ_STAMP(st_acc)
    }
    // ---- the popped step becomes the current one ----------------------------------------------------------------------------
    have_cur = have_nxt;
    cur_slot = nxt_slot; cur_d0 = nxt_d0; cur_d1 = nxt_d1; cur_d2 = nxt_d2;
    cur_hot = nxt_valid && ng1.y > kClampGuard;
    {
      float K[6];
      tile_coefficients(ng0, ng1, cx, cy, K);
      curK[0] = nxt_valid ? K[0] : -1e30f;
#pragma unroll
      for (int m = 1; m < 6; ++m) curK[m] = nxt_valid ? K[m] : 0.f;
      curC[0] = nxt_valid ? nc0 : 0.f; curC[1] = nxt_valid ? nc1 : 0.f; curC[2] = nxt_valid ? nc2 : 0.f;
    }
#ifdef VTGS_Q_STAMPS
    asm volatile("" :: "v"(curK[0]), "v"(curC[0]));
#endif
    BQ_STAMP(st_prom)
    // ---- retire: the oldest chunk has left all four queues AND the step still to be computed holds none of its entries ---
    while (inflight > 0 && __ballot(c0 > 0) == 0ull && __ballot(have_cur && cur_d0 > 0) == 0ull) {
      const int slot = (int)rslot + l;
      const float4 a = accA[slot];
      const float2 b = accB[slot];
      const float4 w = accW[slot];
      accA[slot] = make_float4(0.f, 0.f, 0.f, 0.f); accB[slot] = make_float2(0.f, 0.f); accW[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (l < n0) {
        float2* p = reinterpret_cast<float2*>(grad_inst + (size_t)tinst[slot] * kGradRec);
        p[0] = make_float2(a.x, a.y); p[1] = make_float2(a.z, a.w); p[2] = b;
        p[3] = make_float2(w.x, w.y); p[4] = make_float2(w.z, __uint_as_float(tile_bits));
      }
      c0 = c1; c1 = c2; c2 = 0; n0 = n1; n1 = n2; n2 = 0;
      cur_d0 = cur_d1; cur_d1 = cur_d2; cur_d2 = 0;              // what the step in flight holds of the NEW oldest chunk
      rslot = (rslot == (uint32_t)(kBqRing - 64)) ? 0u : rslot + 64u; --inflight;
    }
    BQ_STAMP(st_ret)
    if (inflight == 0 && base >= e && !have_cur) break;         // the list is exhausted, every chunk has retired, nothing in flight
  }
  // every pixel ended before the end of the list (or the loop ran out): chunks still in flight leave with what they have
  // gathered so far, entries never appended contributed nothing
  while (inflight > 0) {
    const int slot = (int)rslot + l;
    if (l < n0) {
      const float4 a = accA[slot];
      const float2 b = accB[slot];
      const float4 w = accW[slot];
      float2* p = reinterpret_cast<float2*>(grad_inst + (size_t)tinst[slot] * kGradRec);
      p[0] = make_float2(a.x, a.y); p[1] = make_float2(a.z, a.w); p[2] = b;
      p[3] = make_float2(w.x, w.y); p[4] = make_float2(w.z, __uint_as_float(tile_bits));
    }
    n0 = n1; n1 = n2; n2 = 0;
    rslot = (rslot == (uint32_t)(kBqRing - 64)) ? 0u : rslot + 64u; --inflight;
  }
  for (; base < e; base += 64u) {
    const uint32_t pos = base + (uint32_t)l;
    if (pos < e) {
      float2* p = reinterpret_cast<float2*>(grad_inst + (size_t)sorted_inst[pos] * kGradRec);
      p[0] = p[1] = p[2] = p[3] = make_float2(0.f, 0.f);
      p[4] = make_float2(0.f, __uint_as_float(tile_bits));
    }
  }
#ifdef VTGS_Q_STAMPS
  if (step_counters && l == 0) {
    uint32_t* o = step_counters + 64 + kStampWords * tile;
    const unsigned long long se = __builtin_amdgcn_s_memtime();
    o[0] = (uint32_t)st_app; o[1] = (uint32_t)st_pop; o[2] = (uint32_t)st_sweep; o[3] = (uint32_t)st_contr;
    o[4] = (uint32_t)(se - st0); o[5] = nsteps; o[6] = (uint32_t)st_acc; o[7] = (uint32_t)st_prom | ((uint32_t)(st_ret >> 4) << 20);
  }
#endif
  if (step_counters && l == 0) atomicAdd(&step_counters[blockIdx.x & 63u], nsteps);   // measurement only (VTGS_COUNT_STEPS)
}

}  // namespace vtgs

#endif  // VTGS_XCHECK_BUILD
