// vtgs_composite_q.hip -- composites with per-quadrant splat queues (gfx950, wave64).
//
// The lane = pixel kernels of vtgs_composite.hip evaluate every splat of an 8x8 tile's list at all 64 pixels; at view-tied
// densities (sigma ~ 1 px) a splat's alpha >= 1/255 footprint covers ~2 of the tile's four 4x4 quadrants, so about half of
// those (pixel, splat) rows are dead by construction.  Here one wavefront is still one 8x8 tile and one lane one pixel, but
// the 16 lanes of a 4x4 QUADRANT walk their own list:
//
//   * lane L <-> pixel (x, y) = (4 (q & 1) + (i & 3), 4 (q >> 1) + (i >> 2)),  q = L >> 4 the quadrant, i = L & 15;
//   * per 64-entry chunk of the tile's depth-sorted list, lane L gathers splat L, derives from the bounding box of its
//     alpha >= 1/255 ellipse which quadrants it can reach (4 bits), and the wavefront compacts the chunk into four FIFO
//     queues in LDS with __ballot + mbcnt ranks ("wavefront-ballot compaction" of the LDS-staged list).  The order inside
//     a queue is the list order, so every pixel still sees its contributors front to back;
//   * a STEP pops up to 16 entries from every queue: v_mfma_f32_4x4x1_16b_f32 with cbsz = 2 / abid = g multiplies, in each
//     16-lane group, the four splats held by the group's lanes 4g..4g+3 with the group's own 16 pixels -- four MFMAs per
//     rank-1 term give every lane its pixel x 16 splats OF ITS OWN QUADRANT (tests/micro/mfma_layout.hip checks the
//     layout).  Up to three chunks are in flight: the loop appends chunks while the table ring has room and otherwise
//     steps until every queue has popped the last entry of the OLDEST chunk, whose 64 table slots are then free again
//     (an explicit invariant -- counted per quadrant and chunk -- not an argument about queue lengths: entries that
//     reach no quadrant at all exist).  With three chunks queued, every quadrant pops full groups of 16 until the list
//     runs out: 8.0 steps per tile on the headline scene, the minimum its queue lengths allow, against 9.7 when a step
//     was taken as soon as ONE queue held 16 (profiles/r3_queue_policy.md);
//   * colour accumulation is a rank-1 update per splat as well, C[ch][pixel] += c[ch][k] w[k][pixel], and runs on the
//     same MFMA (A = the splat's four channels transposed onto the lanes of block g, B = the lane's w_k): no payload
//     reads in the inner loop.  An exact-f32 MFMA is a k-ordered fmaf chain, and a skipped pair has w = 0 exactly, so
//     images are BIT-IDENTICAL to composite_forward_px (tests/test_gpu_parity.py).
//
// Where the time goes (profiles/r2_forward_study.md).  A step costs ~1,100 cycles of one wavefront's time and 520-660
// cycles of the SIMD's at 3-8 resident wavefronts; more than half of that is the vector sweep (exp, threshold, weight,
// transmittance: 290 cycles per 16 splats, independent of occupancy), the 24 exponent MFMAs take ~120 and the 16 colour
// MFMAs ~125.  MFMAs and vector instructions are kept in separate groups (sched_group_barrier): alternating one by one
// they cost 1.5x the sum of their parts.  The gather is software-pipelined two chunks deep.  With the prefetch registers
// the kernel needs ~120 VGPRs (4 wavefronts per SIMD); the 64-register form without prefetch runs no faster at 6.
//
// Semantics per pixel: SURVEY.md Appendix A3 (see vtgs_composite.hip).
#include "vtgs_internal.h"
#include "vtgs_composite_common.h"
#include "vtgs_sort_common.h"

namespace vtgs {

#ifndef VTGS_Q_CHUNKS
#define VTGS_Q_CHUNKS 3
#endif
constexpr int kQChunks = VTGS_Q_CHUNKS;                 // 64-entry chunks of the tile's list in flight at once
constexpr int kQRing = 64 * kQChunks;       // table / queue ring: 192 slots
constexpr int kQDummy = kQRing;             // table slot 192: a splat that reaches nothing (popped past the end of a queue)
constexpr int kSortedIdsInLds = 416;        // sorted ids the wavefront keeps in LDS for its own chunks (4 workgroups per CU: 40 KB each)

struct QuadCoord { int tile, px, py, q, i; bool tile_ok, inside; };

__device__ __forceinline__ QuadCoord quad_coord(const CamScalars& cs, uint32_t nblk, int gx16, int gx8, int gy8) {
  const uint32_t b = xcd_swizzle(blockIdx.x, nblk);
  const int l = lane_id();
  const int row16_0 = cs.row8_begin >> 1;
  const int t16x = (int)(b % (uint32_t)gx16), t16y = row16_0 + (int)(b / (uint32_t)gx16);
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // workgroup = the 2x2 tiles of a 16x16 block
  const int t8x = 2 * t16x + (w & 1), t8y = 2 * t16y + (w >> 1);
  QuadCoord qc;
  qc.q = l >> 4; qc.i = l & 15;
  qc.tile_ok = t8x < gx8 && t8y < gy8 && t8y >= cs.row8_begin && t8y < cs.row8_end;
  qc.tile = t8y * gx8 + t8x;
  qc.px = t8x * kSubTile + 4 * (qc.q & 1) + (qc.i & 3);
  qc.py = t8y * kSubTile + 4 * (qc.q >> 1) + (qc.i >> 2);
  qc.inside = qc.tile_ok && qc.px < cs.W && qc.py < cs.H;
  return qc;
}

// Quadrants (bit q) the splat's alpha >= 1/255 region can reach: bounding box of the ellipse {q <= tau}, tau = ln(255 o) with
// the binning's conservative slack -- a quadrant outside the box holds no pixel with alpha >= 1/255.  (sx, sy) = splat
// centre relative to the tile centre; the quadrants' pixel centres are X, Y in {-3.5..-0.5} and {0.5..3.5}.
__device__ __forceinline__ uint32_t quadrant_mask(const float4& g0, const float4& g1, float sx, float sy) {
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

// exponents of the lane's pixel x the four splats held by lanes 4g..4g+3 of the lane's OWN 16-lane group
template <int G>
__device__ __forceinline__ f32x4 q_exponents(const float (&K)[6], const float (&Phi)[6]) {
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < 6; ++m) d = __builtin_amdgcn_mfma_f32_4x4x1f32(K[m], Phi[m], d, 2, G, 0);
  return d;
}

template <int R>
__device__ __forceinline__ f32x4 q_colour(float pt, float w, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(pt, w, c, 2, R, 0);        // C[ch] += c[ch][splat 4j + R] * w
}

template <bool CLAMP>
__device__ __forceinline__ float q_alpha(float d) {
  const float g = __builtin_amdgcn_exp2f(d);
  const float al = CLAMP ? fminf(kAlphaMax, g) : g;                     // CLAMP == false: no splat in flight can reach 0.99
  return (al >= kAlphaMin) ? al : 0.f;
}

// One step for the lane's pixel: 16 splats of its quadrant's queue.  T: transmittance in front of the next splat, frozen once
// the pixel has ended; `done`: ended / off-image; `exact` (wave-uniform): go straight to the exact sweep while pixels keep
// ending (the policy of composite_forward_px).  PT[G]: lane (q, 4r + ch) holds channel ch of splat 4G + r.
//
// THREE PHASES, kept apart with sched_barrier: (A) the 24 exponent MFMAs back to back, (B) the vector sweep -- alpha, weight
// and transmittance of the 16 splats -- with no matrix instruction in it, (C) the 16 colour MFMAs back to back.
// tests/micro/mix_rate.hip: an f32 MFMA and a vector instruction that alternate in one wavefront's stream cost 1.5x the sum
// of their separate issue times at every occupancy (1 MFMA : 3 v_fma), while groups of 24 : 72 cost the plain sum; left
// to itself the scheduler interleaves them one by one to hide the MFMA latency, which is exactly the wrong thing here.
#ifndef VTGS_Q_STEP_ABL
#define VTGS_Q_STEP_ABL 0      // tests/micro/step_rate.hip only: 1 = no exponent MFMAs, 2 = no colour MFMAs, 4 = no vector sweep
#endif
template <bool DUAL, bool CLAMP, bool EXACT_FIRST>
__device__ __forceinline__ void q_forward_step(float& T, bool& done, bool& exact, f32x4& C, f32x4& C2,
                                               const float4* ka, const float2* kb, int slot, const float (&Phi)[6],
                                               const float (&PT)[4], const float (&PT2)[4]) {
  float v[16];                                                  // exponents, then (in place) the weights w = alpha T
  auto exponents = [&]() {
    const float4 a4 = ka[slot];
    const float2 b2 = kb[slot];
    const float K[6] = {a4.x, a4.y, a4.z, a4.w, b2.x, b2.y};
    if constexpr ((VTGS_Q_STEP_ABL & 1) != 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { v[r] = K[r % 6] + Phi[r % 5]; asm volatile("" : "+v"(v[r])); }
      return;
    }
    // term by term across the four splat groups: consecutive MFMAs belong to different accumulation chains (a dependent
    // v_mfma_f32_4x4x1 issues every ~6 cycles, independent ones every ~3: tests/micro/mix_rate.hip)
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0, d3 = d0;
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(K[m], Phi[m], d0, 2, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(K[m], Phi[m], d1, 2, 1, 0);
      d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(K[m], Phi[m], d2, 2, 2, 0);
      d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(K[m], Phi[m], d3, 2, 3, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = d0[r]; v[4 + r] = d1[r]; v[8 + r] = d2[r]; v[12 + r] = d3[r]; }
  };
  exponents();
  bool swept = false;
  if constexpr ((VTGS_Q_STEP_ABL & 4) != 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(v[k]));
    swept = true;
  } else
  if constexpr (!EXACT_FIRST) {
    float Tn = done ? 0.f : T;                                  // optimistic sweep: no stop test
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      v[k] = q_alpha<CLAMP>(v[k]) * Tn;
      Tn = Tn - v[k];
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);         // the 24 exponent MFMAs as one group ...
    __builtin_amdgcn_sched_group_barrier(0x002, 96, 0);         // ... then the vector sweep
    if (__ballot(!done && Tn < kTStop) == 0ull) { T = done ? T : Tn; swept = true; }
  }
  if (!swept) {
    // exact sweep: the first splat with T (1 - alpha) < 1e-4 ends the pixel BEFORE it is added.  After an optimistic sweep
    // the coefficients are read and the exponents produced AGAIN (24 MFMAs on the rare path) instead of being kept alive
    // next to the weights that replaced them: ~20 registers less on the common path.
    if constexpr (!EXACT_FIRST) {
      asm volatile("" : "+v"(slot));                            // not the same loads as far as CSE is concerned
      exponents();
    }
    const bool was_done = done;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float wk = q_alpha<CLAMP>(v[k]) * T;
      const float tn = T - wk;
      const bool stop = tn < kTStop;
      const bool live = !done && !stop;
      v[k] = live ? wk : 0.f;
      T = live ? tn : T;
      done = done || stop;
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 24, 1);
    __builtin_amdgcn_sched_group_barrier(0x002, 160, 1);
    exact = __builtin_popcountll(__ballot(done && !was_done)) >= kExactFirstEndings;
  }
  if constexpr ((VTGS_Q_STEP_ABL & 2) != 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" :: "v"(v[k]));
    return;
  }
  // colour: C[ch] += c[ch][k] w[k], a rank-1 update per splat on the matrix cores (k-ordered fmaf chain: bit-identical to
  // the lane = pixel kernel; four independent accumulators would issue ~20 % faster at 8 waves per SIMD, tests/micro/
  // step_rate.hip, but cost 12 registers, which at this kernel's occupancy is the scarcer resource)
#define VTGS_Q_COLOUR(G)                                                                       \
  C = q_colour<0>(PT[G], v[4 * G + 0], C); C = q_colour<1>(PT[G], v[4 * G + 1], C);            \
  C = q_colour<2>(PT[G], v[4 * G + 2], C); C = q_colour<3>(PT[G], v[4 * G + 3], C);            \
  if constexpr (DUAL) {                                                                        \
    C2 = q_colour<0>(PT2[G], v[4 * G + 0], C2); C2 = q_colour<1>(PT2[G], v[4 * G + 1], C2);    \
    C2 = q_colour<2>(PT2[G], v[4 * G + 2], C2); C2 = q_colour<3>(PT2[G], v[4 * G + 3], C2);    \
  }
  VTGS_Q_COLOUR(0) VTGS_Q_COLOUR(1) VTGS_Q_COLOUR(2) VTGS_Q_COLOUR(3)
#undef VTGS_Q_COLOUR
  __builtin_amdgcn_sched_group_barrier(0x008, DUAL ? 32 : 16, 2);
}

#ifndef VTGS_Q_WAVES
#define VTGS_Q_WAVES 4
#endif
// MODE 0: one colour set + the depth image; 1: dual render (two colour sets); 2: dual render whose second set is the fused
// caller chain's [z, 1, z^2] and whose second image is consumed as get_loss consumes it (VTGS_FORWARD_SECOND_IS_DEPTH,
// include/vtgs.h) -- the kernel of mode 0 with z in the depth column: plane 0 = sum w z, plane 1 = 1 - T_final (= sum w up to
// rounding), plane 2 = plane 0 squared.  Four payload columns instead of six: one matrix-instruction group per step instead
// of two, the registers and LDS of the single render (4 wavefronts per SIMD instead of 3).
template <int MODE>
__global__ __launch_bounds__(256, MODE == 1 ? 3 : VTGS_Q_WAVES) void composite_forward_q(
    CamScalars cs, const float* __restrict__ bg, uint32_t nblk,
    const uint32_t* __restrict__ tile_cnt, uint32_t tile_cap, uint32_t* sorted_gid,
    const GeomRec* __restrict__ geom, const float* __restrict__ colors,
    float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ final_T,
    const Counters* __restrict__ ctr, const float* __restrict__ colors_b, float* __restrict__ out_color_b,
    int sort_mode, const unsigned long long* __restrict__ bin_keys, const uint32_t* __restrict__ bin_vals,
    uint32_t* sorted_inst, FinalizeArgs fin, uint8_t* __restrict__ qmask, uint32_t* __restrict__ step_counters) {
  constexpr bool DUAL = MODE == 1, ZL = MODE == 2;           // ZL: second image from the depth column
  // per wavefront: the table of the ring's entries (+ one dummy slot) and the four queues of table slots.  Queue bytes are
  // stored twice, 128 apart, so a pop reads [head & 127, head & 127 + 16) without wrapping.
  // The three tables of a wavefront are carved from one block: the counting sort of the tile's list (vtgs_sort_common.h)
  // uses the same bytes as staging before the first chunk is appended.
  constexpr int kTabFloats = (kQRing + 1) * 10 + 2;             // ka 4 + pa 4 + kb 2 floats per slot, 16-byte multiples
  static_assert(kTabFloats * 4 >= kCountSortMax * 8 + 256 * 4, "the sort's staging must fit into the table block");
  __shared__ __attribute__((aligned(16))) float lds_tab[4][kTabFloats];
  __shared__ float4 lds_pb[DUAL ? 4 : 1][DUAL ? kQRing + 1 : 1];//                   (dual: c4 c5 0 0)
  __shared__ uint8_t lds_q[4][4][kQRing];
  __shared__ uint32_t lds_sgid[4][kSortedIdsInLds];             // the wavefront's own copy of the first sorted Gaussian ids
#ifdef VTGS_AB_FWD_PAD                                          // occupancy experiment: extra LDS so that fewer workgroups fit a CU
  __shared__ float ab_pad[VTGS_AB_FWD_PAD];
  if (cs.W < 0) { ab_pad[threadIdx.x] = 1.f; __syncthreads(); if (ab_pad[(threadIdx.x + 1) & 255] == 2.f) return; }
#endif
#ifdef VTGS_Q_STAMPS
  const unsigned long long st0 = __builtin_amdgcn_s_memtime();
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st1 = st0, st2 = st0, st_app = 0ull, st_step = 0ull;
  unsigned long long st_sort[2] = {st0, st0};
#define VTGS_SORT_STAMP st_sort
#else
#define VTGS_SORT_STAMP nullptr
#endif
  // bail (uniform over the grid): the forward cannot complete -- every tile is composited as EMPTY, so the caller's image is
  // the background colour instead of whatever its memory held (a run-ahead caller looks at the result record later).
  bool bail = false;
  if (sort_mode) {
    // Nothing ran between the binning and this kernel: the first workgroup does what finalize_forward does (longest list,
    // statistics, overflow flags, the host's record -- it is dispatched first, so the record still leaves early), and
    // every wavefront decides for itself whether its bin is safe to read: slots below min(count, capacity) are all
    // written unless the INSTANCE capacity overflowed (then an entry may have been dropped after its slot was taken).
    if (blockIdx.x == 0u) finalize_block<256>(fin.tile_cnt, fin.tiles, fin.ctr, fin.capacity, fin.tile_cap, fin.block_stats,
                                              fin.nblocks, fin.host_record, fin.plan, fin.plan_next);
    bail = (unsigned long long)ctr->inst_total > fin.capacity;
  } else {
    bail = ctr->overflow != 0u;                                 // bins hold unwritten slots after an overflow
  }
  if (qmask && blockIdx.x == 0u && threadIdx.x == 0u) fin.ctr->qmask_valid = 1u;
  const int gx16 = (cs.W + kBinTile - 1) / kBinTile;
  const int gx8 = (cs.W + kSubTile - 1) / kSubTile, gy8 = (cs.H + kSubTile - 1) / kSubTile;
  const QuadCoord qc = quad_coord(cs, nblk, gx16, gx8, gy8);
  if (!qc.tile_ok) return;
  const int l = lane_id();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float4* ka = reinterpret_cast<float4*>(lds_tab[wv]);                                   // K0..K3
  float4* pa = reinterpret_cast<float4*>(lds_tab[wv] + 4 * (kQRing + 1));                // c0 c1 c2 depth   (dual: c0 c1 c2 c3)
  float2* kb = reinterpret_cast<float2*>(lds_tab[wv] + 8 * (kQRing + 1));                // K4, K5
  float4* pb = lds_pb[DUAL ? wv : 0];
  uint32_t* sgid = lds_sgid[wv];
  const int q = qc.q, i = qc.i;
  const uint8_t* myq = lds_q[wv][q];
  const int lx = 4 * (q & 1) + (i & 3), ly = 4 * (q >> 1) + (i >> 2);
  const float cx = (float)(qc.px - lx) + 3.5f, cy = (float)(qc.py - ly) + 3.5f;
  const float X = (float)lx - 3.5f, Y = (float)ly - 3.5f;
  const float Phi[6] = {1.f, X, Y, X * X, X * Y, Y * Y};
  const BinRange br = bin_range(cs, (uint32_t)qc.tile, tile_cap);
  // A bin whose list outgrew it (the flag is raised by the first workgroup, possibly after this wavefront has started) is
  // composited as empty too: nobody sorted it -- sort_long_lists and sort_tiles skip such a bin -- so beyond 512 slots its
  // sorted list would be whatever the workspace held before (unchecked ids from unwritten memory).
  const uint32_t cnt_raw = tile_cnt[qc.tile];
  const uint32_t s = br.s, e = s + ((bail || cnt_raw > br.cap) ? 0u : cnt_raw);
  // sort_mode != 0 (the host picks it when no bin can hold more than 1024 entries): the wavefront sorts its own tile's list
  // here -- 1 = payload packed into the key, 2 = key + value -- instead of a sort kernel before this one: one launch less,
  // and the list's trip through memory overlaps with the other wavefronts' compositing.  The sorted list still goes to
  // global memory (the backward and a second render read it); the wavefront itself reads its first kSortedIdsInLds ids from
  // the LDS copy the counting sort leaves behind (round 3), the rest -- or everything after the bitonic network -- from memory.
  bool ids_in_lds = false;                                      // wave-uniform: the first kSortedIdsInLds sorted ids are in sgid
  if (sort_mode) {
    const uint32_t L = e - s;
    const size_t sz = (size_t)s;
    bool counted = false;
    const bool presorted = (sort_mode & 4) != 0 && L > (uint32_t)kCountSortMax;   // sort_long_lists has been here (lists > 512)
    if (presorted) {
    } else if (L > 1u && L <= (uint32_t)kCountSortMax) {                // the common case: bucket pass + in-bucket ranks, ids kept in LDS
      unsigned long long* stage = reinterpret_cast<unsigned long long*>(lds_tab[wv]);
      uint32_t* cnt = reinterpret_cast<uint32_t*>(lds_tab[wv]) + 2 * kCountSortMax;
      if (L <= 64u) counted = wave_count_sort<1>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, stage, cnt, sgid, kSortedIdsInLds);
      else if (L <= 128u) counted = wave_count_sort<2>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, stage, cnt, sgid, kSortedIdsInLds);
      else if (L <= 256u) counted = wave_count_sort<4>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, stage, cnt, sgid, kSortedIdsInLds);
      else counted = wave_count_sort<8>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, stage, cnt, sgid, kSortedIdsInLds);
      ids_in_lds = counted;
    }
    if (counted || presorted) {
    } else if (L == 1u) {
      if (l == 0) { sorted_gid[s] = (uint32_t)bin_keys[s]; sorted_inst[s] = bin_vals[s]; }
    } else if ((sort_mode & 3) == 1) {
      if (L <= 64u) { if (L) wave_sort_tile<1, true>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, VTGS_SORT_STAMP); }
      else if (L <= 128u) wave_sort_tile<2, true>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, VTGS_SORT_STAMP);
      else if (L <= 256u) wave_sort_tile<4, true>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, VTGS_SORT_STAMP);
      else if (L <= 512u) wave_sort_tile<8, true>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, VTGS_SORT_STAMP);
      else if (L <= 1024u) wave_sort_tile<16, true>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, VTGS_SORT_STAMP);
      // (longer: planned bins -- sort_tiles' long-list pass has sorted it ahead of this kernel)
    } else {
      if (L <= 64u) { if (L) wave_sort_tile<1, false>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, VTGS_SORT_STAMP); }
      else if (L <= 128u) wave_sort_tile<2, false>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, VTGS_SORT_STAMP);
      else if (L <= 256u) wave_sort_tile<4, false>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, VTGS_SORT_STAMP);
      else if (L <= 512u) wave_sort_tile<8, false>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, VTGS_SORT_STAMP);
      else if (L <= 1024u) wave_sort_tile<16, false>(bin_keys, bin_vals, sorted_gid, sorted_inst, sz, L, l, VTGS_SORT_STAMP);
    }
    if (!(counted && L <= (uint32_t)kSortedIdsInLds)) {
      // some of the list will be re-read from memory: the wavefront's own stores before its own loads -- program order within
      // one wavefront, no cache maintenance (a device-scope fence here writes L2 back for every tile: 590 us instead of 90)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_s_waitcnt(0);
    }
  }
#ifdef VTGS_Q_STAMPS
  st1 = __builtin_amdgcn_s_memtime();
#endif
  // dummy slot + queue bytes start defined (a pop past the end of a queue reads bytes that were never written)
  ka[kQDummy] = make_float4(-1e30f, 0.f, 0.f, 0.f);
  kb[kQDummy] = make_float2(0.f, 0.f);
  pa[kQDummy] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (DUAL) pb[kQDummy] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < (4 * kQRing) / 256; ++k)                   // 4 queues x 192 bytes = 192 dwords per wavefront
    reinterpret_cast<uint32_t*>(lds_q[wv][0])[64 * k + l] = 0u;

  float T = 1.f;
  f32x4 C = {0.f, 0.f, 0.f, 0.f}, C2 = {0.f, 0.f, 0.f, 0.f};
  bool done = !qc.inside, exact = false;
  // Ring state of the lane's OWN queue (uniform per 16-lane group): head position 0..kQRing-1 and the number of queued
  // entries by chunk in flight, oldest first.  A pop takes from the oldest chunk first (the queue is a FIFO in list order), so
  // "c0 == 0 in every quadrant" says that nothing references the oldest chunk's 64 table slots any more.
  int head_v = 0, c0 = 0, c1 = 0, c2 = 0;
  bool hot0 = false, hot1 = false, hot2 = false;                // per chunk in flight: some splat may reach the 0.99 clamp
  int inflight = 0;                                             // chunks in flight (wave-uniform)
  uint32_t base = s, wslot = 0u, nsteps = 0u;                   // wslot: first table slot of the next chunk (0, 64, 128)
  // Two-deep software pipeline of the gather: while chunk c is composited the list entry of chunk c+2 and the geometry
  // record + colours of chunk c+1 are in flight (two dependent trips to L2 / Infinity Cache per chunk otherwise sit on the
  // wavefront's critical path: gather + compaction alone is 35 us of this kernel, profiles/r2_forward_ablation.md).
  // Loads are unconditional -- a lane past the end of the list reads entry 0 of its own bin -- and masked afterwards.
  auto entry = [&](uint32_t b) {
    const uint32_t p = b + (uint32_t)l, pp = (p < e ? p : s) - s;
    if (ids_in_lds && pp < (uint32_t)kSortedIdsInLds) return sgid[pp];          // (no trip to L2 for the wavefront's own list)
    return sorted_gid[s + pp];
  };
  uint32_t gid_cur = 0u, gid_nxt = 0u;
  float4 g0n = make_float4(0.f, 0.f, 0.f, 0.f), g1n = g0n;
  float c0n = 0.f, c1n = 0.f, c2n = 0.f, d0n = 0.f, d1n = 0.f, d2n = 0.f;
  auto fetch = [&](uint32_t gid) {
    const float4* gp = reinterpret_cast<const float4*>(geom + gid);
    g0n = gp[0]; g1n = gp[1];
    c0n = colors[3 * gid]; c1n = colors[3 * gid + 1]; c2n = colors[3 * gid + 2];
    if (DUAL) { d0n = colors_b[3 * gid]; d1n = colors_b[3 * gid + 1]; d2n = colors_b[3 * gid + 2]; }
    if (ZL) d0n = colors_b[3 * gid];
  };
  // ---- append one 64-entry chunk whose records have arrived: table, wavefront-ballot compaction into the four queues -------
  // (a macro, not a lambda: called from four places, the closure kept the ring counters in scratch memory)
#define VTGS_Q_APPEND(G0, G1, FC0, FC1, FC2, FD0, FD1, FD2)                                                              \
  {                                                                                                                      \
    const float4 g0 = (G0), g1 = (G1);                                                                                   \
    const uint32_t pos = base + (uint32_t)l;                                                                             \
    const bool in = pos < e;                                                                                             \
    const int slot = (int)wslot + l;                                                                                     \
    {                                                                                                                    \
      float K[6];                                                                                                        \
      tile_coefficients(g0, g1, cx, cy, K);                                                                              \
      ka[slot] = make_float4(K[0], K[1], K[2], K[3]);                                                                    \
      kb[slot] = make_float2(K[4], K[5]);                                                                                \
      pa[slot] = make_float4((FC0), (FC1), (FC2), (DUAL || ZL) ? (FD0) : g1.z);                                          \
      if (DUAL) pb[slot] = make_float4((FD1), (FD2), 0.f, 0.f);                                                          \
    }                                                                                                                    \
    const bool hot = __ballot(in && g1.y > kClampGuard) != 0ull;                                                         \
    const uint32_t mask = in ? quadrant_mask(g0, g1, g0.x - cx, g0.y - cy) : 0u;                                         \
    if (qmask && in) qmask[pos] = (uint8_t)mask;          /* kept for the backward (composite_backward_q) */             \
    const int tail_v = head_v + c0 + c1 + c2;                                                                            \
    int add_v = 0;                                                                                                       \
    _Pragma("unroll") for (int qq = 0; qq < 4; ++qq) {                                                                   \
      const bool in_q = (mask >> qq) & 1u;                                                                               \
      const unsigned long long bal = __ballot(in_q);                                                                     \
      const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u)); \
      const int tail = __builtin_amdgcn_readlane(tail_v, 16 * qq);                                                       \
      if (in_q) {                                                                                                        \
        int p = tail + rank;                              /* tail < 2 kQRing, rank < 64 */                               \
        p -= (p >= kQRing) ? kQRing : 0;                                                                                 \
        p -= (p >= kQRing) ? kQRing : 0;                                                                                 \
        lds_q[wv][qq][p] = (uint8_t)slot;                                                                                \
      }                                                                                                                  \
      const int add = (int)__builtin_popcountll(bal);                                                                    \
      add_v = (q == qq) ? add : add_v;                                                                                   \
    }                                                                                                                    \
    if (inflight == 0) { c0 += add_v; hot0 = hot; }                                                                      \
    else if (inflight == 1) { c1 += add_v; hot1 = hot; }                                                                 \
    else { c2 += add_v; hot2 = hot; }                                                                                    \
    base += 64u; wslot = (wslot == (uint32_t)(kQRing - 64)) ? 0u : wslot + 64u; ++inflight;                              \
  }
  static_assert(kQChunks == 3, "the prologue below requests exactly the ring's three chunks");
  if (s < e && __ballot(!done) != 0ull) {
    // The ring starts empty and takes up to three chunks at once: their records are requested TOGETHER (one round trip to L2 /
    // Infinity Cache instead of three back to back -- ~8 % of the wavefront's life, profiles/r3_stamps.md), then appended.
    const uint32_t gidB = entry(s + 64u), gidC = entry(s + 128u);
    gid_cur = entry(s + 192u);
    gid_nxt = entry(s + 256u);
#define VTGS_Q_FETCH(tag, gid_expr)                                                                        \
    const uint32_t gid_##tag = (gid_expr);                                                                  \
    const float4* gp_##tag = reinterpret_cast<const float4*>(geom + gid_##tag);                             \
    const float4 g0_##tag = gp_##tag[0], g1_##tag = gp_##tag[1];                                            \
    const float c0_##tag = colors[3 * gid_##tag], c1_##tag = colors[3 * gid_##tag + 1], c2_##tag = colors[3 * gid_##tag + 2]; \
    const float d0_##tag = (DUAL || ZL) ? colors_b[3 * gid_##tag] : 0.f, d1_##tag = DUAL ? colors_b[3 * gid_##tag + 1] : 0.f, \
                d2_##tag = DUAL ? colors_b[3 * gid_##tag + 2] : 0.f;
    VTGS_Q_FETCH(a, entry(s))
    VTGS_Q_FETCH(b, gidB)
    VTGS_Q_FETCH(c, gidC)
#undef VTGS_Q_FETCH
#ifdef VTGS_Q_STAMPS
    const unsigned long long sa = __builtin_amdgcn_s_memtime();
#endif
    VTGS_Q_APPEND(g0_a, g1_a, c0_a, c1_a, c2_a, d0_a, d1_a, d2_a)
#ifdef VTGS_Q_STAMPS
    __builtin_amdgcn_s_waitcnt(0xc07f);
    { const unsigned long long sb = __builtin_amdgcn_s_memtime(); st_app += sb - sa; st2 = sb; }
#endif
    if (base < e) VTGS_Q_APPEND(g0_b, g1_b, c0_b, c1_b, c2_b, d0_b, d1_b, d2_b)
    if (base < e) VTGS_Q_APPEND(g0_c, g1_c, c0_c, c1_c, c2_c, d0_c, d1_c, d2_c)
    fetch(gid_cur);                                                // the fourth chunk's records: in flight during the first steps
  }
#ifdef VTGS_Q_STAMPS
  const unsigned long long st3 = __builtin_amdgcn_s_memtime();   // loop entry: everything up to here is sort + prologue
#endif

  for (;;) {
    if (inflight < kQChunks && base < e) {
      // ---- append the next 64-entry chunk (its records were requested one append ago) ------------------------------------------
      if (__ballot(!done) == 0ull) break;
#ifdef VTGS_Q_STAMPS
      const unsigned long long sa = __builtin_amdgcn_s_memtime();
#endif
      VTGS_Q_APPEND(g0n, g1n, c0n, c1n, c2n, d0n, d1n, d2n)
      gid_cur = gid_nxt;
      gid_nxt = entry(base + 64u);
      fetch(gid_cur);                                               // next chunk's data: in flight during this chunk's steps
#ifdef VTGS_Q_STAMPS
      __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): the table / queue writes have landed
      { const unsigned long long sb = __builtin_amdgcn_s_memtime(); st_app += sb - sa; }
#endif
#if defined(VTGS_Q_ABL) && VTGS_Q_ABL == 1                                 // ablation: gather + compaction only, no steps
      asm volatile("" :: "v"(c0));
      head_v += c0 + c1 + c2; head_v -= (head_v >= kQRing) ? kQRing : 0; c0 = c1 = c2 = 0; inflight = 0;
#endif
      continue;
    }
    if (inflight == 0) break;                                     // list exhausted, every chunk retired
    if (__ballot(c0 > 0) == 0ull) {                               // the oldest chunk is drained in all four queues: retire it
      c0 = c1; c1 = c2; c2 = 0;
      hot0 = hot1; hot1 = hot2; hot2 = false;
      --inflight;
      continue;
    }
    if (__ballot(!done) == 0ull) break;                           // every pixel of the tile has ended
    // ---- one step: pop min(16, count) entries from every queue ---------------------------------------------------------
#ifdef VTGS_Q_STAMPS
    const unsigned long long ss = __builtin_amdgcn_s_memtime();
#endif
    const int avail = min(16, c0 + c1 + c2);
    const int hm = head_v;
    int sl[5];
    auto qwrap = [](int x) { return x >= kQRing ? x - kQRing : x; };
    sl[4] = (int)myq[qwrap(hm + i)];
#pragma unroll
    for (int j = 0; j < 4; ++j) sl[j] = (int)myq[qwrap(hm + 4 * j + (i >> 2))];
    sl[4] = (i < avail) ? sl[4] : kQDummy;
#pragma unroll
    for (int j = 0; j < 4; ++j) sl[j] = (4 * j + (i >> 2) < avail) ? sl[j] : kQDummy;
    float PT[4], PT2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      PT[j] = reinterpret_cast<const float*>(pa)[4 * sl[j] + (i & 3)];
      PT2[j] = DUAL ? reinterpret_cast<const float*>(pb)[4 * sl[j] + (i & 3)] : 0.f;
    }
    head_v += avail; head_v -= (head_v >= kQRing) ? kQRing : 0;
    {
      int t = avail;
      const int d0 = min(t, c0); c0 -= d0; t -= d0;
      const int d1 = min(t, c1); c1 -= d1; t -= d1;
      c2 -= t;
    }
    ++nsteps;
    if (exact) q_forward_step<DUAL, true, true>(T, done, exact, C, C2, ka, kb, sl[4], Phi, PT, PT2);
    else if (hot0 || hot1 || hot2) q_forward_step<DUAL, true, false>(T, done, exact, C, C2, ka, kb, sl[4], Phi, PT, PT2);
    else q_forward_step<DUAL, false, false>(T, done, exact, C, C2, ka, kb, sl[4], Phi, PT, PT2);
#ifdef VTGS_Q_STAMPS
    asm volatile("" :: "v"(T), "v"(C[0]));
    st_step += __builtin_amdgcn_s_memtime() - ss;
#endif
  }
#ifdef VTGS_Q_STAMPS
  const unsigned long long st_loop_end = __builtin_amdgcn_s_memtime();
  if (step_counters && l == 0) {
    uint32_t* o = step_counters + 64 + kStampWords * qc.tile;
    o[8] = (uint32_t)rt0; o[9] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    o[10] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); o[11] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // HW_ID, XCC_ID
    const unsigned long long se = __builtin_amdgcn_s_memtime();
    o[0] = (uint32_t)(st1 - st0); o[1] = (uint32_t)(st2 - st1); o[2] = (uint32_t)st_app; o[3] = (uint32_t)st_step;
    o[4] = (uint32_t)(se - st0); o[5] = nsteps; o[6] = (uint32_t)(st_loop_end - st0);   // (list length: tile_cnt)
    o[7] = (uint32_t)(st3 - st0);                                // entry -> loop entry (sort + LDS init + the prologue's three appends)
  }
#endif
  if (step_counters && l == 0) atomicAdd(&step_counters[blockIdx.x & 63u], nsteps);   // measurement only (VTGS_COUNT_STEPS)
  if (qc.inside) {
    const size_t P = (size_t)cs.W * cs.H, pix = (size_t)qc.py * cs.W + qc.px;
    out_color[pix] = C[0] + T * bg[0];
    out_color[P + pix] = C[1] + T * bg[1];
    out_color[2 * P + pix] = C[2] + T * bg[2];
    if constexpr (DUAL) {
      out_color_b[pix] = C[3] + T * bg[0];
      out_color_b[P + pix] = C2[0] + T * bg[1];
      out_color_b[2 * P + pix] = C2[1] + T * bg[2];
    } else if constexpr (ZL) {
      const float D = C[3] + T * bg[0];
      out_color_b[pix] = D;
      out_color_b[P + pix] = (1.f - T) + T * bg[1];
      out_color_b[2 * P + pix] = D * D;
    } else {
      out_depth[pix] = C[3];
    }
    final_T[pix] = T;
  }
}
template __global__ void composite_forward_q<0>(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, uint32_t*, const GeomRec*, const float*, float*, float*, float*, const Counters*, const float*, float*, int, const unsigned long long*, const uint32_t*, uint32_t*, FinalizeArgs, uint8_t*, uint32_t*);
template __global__ void composite_forward_q<1>(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, uint32_t*, const GeomRec*, const float*, float*, float*, float*, const Counters*, const float*, float*, int, const unsigned long long*, const uint32_t*, uint32_t*, FinalizeArgs, uint8_t*, uint32_t*);
template __global__ void composite_forward_q<2>(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, uint32_t*, const GeomRec*, const float*, float*, float*, float*, const Counters*, const float*, float*, int, const unsigned long long*, const uint32_t*, uint32_t*, FinalizeArgs, uint8_t*, uint32_t*);

}  // namespace vtgs
