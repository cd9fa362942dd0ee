// vtgs_internal.h -- workspace layout, device-side records and wavefront helpers (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "vtgs_math.h"
#include "../../include/vtgs.h"

namespace vtgs {

// ------------------------------------------------------------------------------------------------
// HBM layout.  One caller-owned byte buffer, carved deterministically from (N, W, H, capacity) so
// that backward finds what forward wrote.  All regions 256-byte aligned.
//
//   counters     256 B     Counters (zeroed per forward)
//   geom         N x 32 B  GeomRec: what the composite needs per splat; read by gather, L2/MALL resident
//   gaux         N x 8 B   first instance id + instance count (instances of a splat are contiguous)
//   block_stats  ceil(N/1024) x 16 B  per-workgroup {visible, 16x16 tiles touched} partials (no same-address atomics)
//   tile_cnt     T8 x 4    per 8x8-tile list length (zeroed per forward, filled by slot reservation)
//   keys         T8 x cap_t x 8   tile bins: (depth bits << 32 | gaussian), written in place by the projection kernel
//   vals         T8 x cap_t x 4   instance id travelling with the key
//   sorted_gid   T8 x cap_t x 4   per-tile front-to-back Gaussian ids (tile t: [t*cap_t, t*cap_t + tile_cnt[t]))
//   sorted_inst  T8 x cap_t x 4   per-tile instance ids (address of the per-instance gradient record)
//   final_T      P x 4     per-pixel transmittance after the last contributor
//   qmask        T8 x cap_t x 1   which 4x4 quadrants of its tile a sorted list entry can reach (quadrant-queue composites)
//   defer_list   N x 48    the splats project_and_bin left to bin_deferred_splats (round 6): the first Counters::defer_total entries
//                          (up to 64 candidate tiles each) and, from the END, Counters::defer_large entries (larger splats)
// Every tile owns a fixed-capacity bin (cap_t, a caller hint like the instance capacity), so binning is one
// pass: no prefix scan over tiles and no scatter pass.  finalize_forward raises the overflow flags right after the
// binning (a dropped instance leaves an unwritten bin slot, so sort and composite bail on the flag) and the caller
// re-runs with the sizes reported in the result record.
// ------------------------------------------------------------------------------------------------
struct Counters {
  // the first two words are ONE 64-bit counter for project_and_bin: a workgroup takes its instance range and its stretch of the
  // deferred-splat list with a single atomic (the deferred count in the LOW word, so that an instance count running past 2^32 --
  // it keeps counting past the capacity -- carries out of the top and not into its neighbour)
  uint32_t defer_total;     // splats project_and_bin left to bin_deferred_splats (entries of the deferred list)
  uint32_t inst_total;      // instances requested (keeps counting past capacity)
  uint32_t overflow;        // bit 0: inst_total > instance capacity, bit 1: a tile list > tile capacity (finalize_forward)
  uint32_t qmask_valid;     // composite_forward_q wrote the quadrant masks of this forward's lists (composite_backward_q reads them)
  uint32_t defer_large;     // deferred splats beyond kGroupArea candidate tiles: listed from the END of the deferred list
  uint32_t pad[11];
  // byte 64: image of the public VtgsForwardInfo, written by finalize_forward, copied to the host by vtgs_forward
  unsigned long long info_instances, info_needed, info_r16;
  uint32_t info_visible, info_max_list, info_overflow, info_complete;
  unsigned long long info_slots;
};
static_assert(sizeof(Counters) <= 256, "counters block");
static_assert(offsetof(Counters, info_instances) == 64 && offsetof(Counters, defer_total) == 0 && offsetof(Counters, inst_total) == 4, "Counters layout");

struct alignas(16) GeomRec {   // 32 bytes
  float u, v;                  // pixel centre (float32)
  float A, B, C;               // conic
  float opacity;
  float depth;
  uint32_t centre_lo;          // what float32 rounding took from (u, v): two truncated-float32 halves (pack_centre_lo)
};
// (ulo, vlo) with |.| <= ulp(u)/2: sign + exponent + 7 mantissa bits each -- 2^-8 of 6e-5 px is more than enough
__host__ __device__ inline uint32_t pack_centre_lo(float ulo, float vlo) {
  union { float f; uint32_t u; } a, b;
  a.f = ulo; b.f = vlo;
  return (a.u & 0xFFFF0000u) | (b.u >> 16);
}
__host__ __device__ inline float centre_lo_x(uint32_t p) { union { float f; uint32_t u; } a; a.u = p & 0xFFFF0000u; return a.f; }
__host__ __device__ inline float centre_lo_y(uint32_t p) { union { float f; uint32_t u; } a; a.u = p << 16; return a.f; }

struct alignas(8) GaussAux { uint32_t inst_base, inst_cnt; };

struct alignas(16) BlockStats { uint32_t visible, pad; unsigned long long r16; };

// What bin_deferred_splats needs of a splat that project_and_bin left aside (more than kDeferArea candidate tiles): written by the
// splat's own lane into the forward's deferred list, read back 16 lanes (a wavefront, a workgroup) at a time -- no dependent look-ups.
struct alignas(16) DeferRec {   // 48 bytes
  float u, v, A, B;             // centre (float32 part: the reach test is conservative by 1e-4 anyway), conic
  float C, tau, depth;          // tau = ln(255 o) with its slack; depth bits = the sort key's high word
  uint32_t gid;                 // the Gaussian
  uint32_t cxy;                 // candidate walk (8x8 tiles), already shrunk to the ellipse's bounding box: cx0 | cy0 << 16,
  int32_t cw, ch;               // ... width, height
  uint32_t inst_base;           // first instance id of the splat: project_and_bin reserved cw * ch ids for it (an upper bound --
                                // the ids behind the tiles it turns out to reach stay unused -- so that bin_deferred_splats needs
                                // no counter of its own: same-address atomics retire one per ~14 ns)
};
static_assert(sizeof(DeferRec) == 48, "DeferRec layout");

constexpr int kGradRec = 10;       // floats per instance gradient record: 6 moments, 3 colour sums, tile id (40-byte stride, float2 access)
constexpr int kGradRecDual = 14;   // dual render: 6 moments + 6 colour sums + tile id + one pad word (56-byte stride, float2 access)
constexpr int kGradRecDual1 = 12;  // dual render, second image differentiated through its first channel only: 6 moments + 3 + 1
                                   // colour sums + tile id + one pad word (48-byte stride, three aligned float4)

constexpr int kStampWords = 12;    // -DVTGS_Q_STAMPS: words per tile in the debug region (8 phase stamps, start / end in 10 ns units of
                                   // the chip-wide constant clock, HW_ID, XCC_ID)
struct WsLayout {
  size_t counters, geom, gaux, block_stats, tile_cnt, keys, vals, sorted_gid, sorted_inst, final_T, qmask, dbg, plan, defer_list, total;
  uint32_t tiles8, tile_cap;      // tile_cap: slots per bin (uniform bins) or the average over the bins (planned bins)
  bool planned;                   // VTGS_TILE_CAPACITY_PLANNED: bin t = [plan[t], plan[t+1]) instead of [t * tile_cap, (t+1) * tile_cap)
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline WsLayout make_layout(int32_t n, int32_t w, int32_t h, uint64_t cap, uint32_t tile_cap) {
  (void)cap;   // the instance capacity bounds instance ids (backward scratch); the bins are sized by tile_cap
  WsLayout L;
  const uint32_t gx8 = (uint32_t)(w + kSubTile - 1) / kSubTile, gy8 = (uint32_t)(h + kSubTile - 1) / kSubTile;
  L.tiles8 = gx8 * gy8;
  L.planned = (tile_cap & VTGS_TILE_CAPACITY_PLANNED) != 0u;
  tile_cap &= ~VTGS_TILE_CAPACITY_PLANNED;
  L.tile_cap = tile_cap;
  const size_t slots = (size_t)L.tiles8 * tile_cap;
  size_t o = 0;
  L.counters = o;    o += 256;                                   // counters and tile_cnt are adjacent: ONE memset clears both
  L.tile_cnt = o;    o += align256(((size_t)L.tiles8 + 1) * 4);
  L.geom = o;        o += align256((size_t)n * sizeof(GeomRec));
  L.gaux = o;        o += align256((size_t)n * sizeof(GaussAux));
  L.block_stats = o; o += align256(((size_t)(n + 1023) / 1024 + 1) * sizeof(BlockStats));
  L.keys = o;        o += align256(slots * 8);
  L.vals = o;        o += align256(slots * 4);
  L.sorted_gid = o;  o += align256(slots * 4);
  L.sorted_inst = o; o += align256(slots * 4);
  L.final_T = o;     o += align256((size_t)w * h * 4);
  L.qmask = o;       o += align256(slots);               // quadrant mask (4 bits) of every sorted list entry, written by composite_forward_q
#ifdef VTGS_Q_STAMPS
  L.dbg = o;         o += align256(256 + (size_t)L.tiles8 * kStampWords * 4)    // diagnostic build: + 12 words of stamps per tile
                          + align256(((size_t)(n + 1023) / 1024 + 1) * 32);         // ... + 8 per workgroup of project_and_bin
#else
  L.dbg = o;         o += 256;                           // 64 step counters (measurement only, VTGS_COUNT_STEPS)
#endif
  L.plan = o;        o += align256(((size_t)L.tiles8 + 1) * 4);   // planned bins: this forward's copy of the caller's plan
  L.defer_list = o;  o += align256(((size_t)n + 1) * sizeof(DeferRec));   // the deferred splats of this forward, in no particular order
  L.total = o;
  return L;
}

// camera scalars passed by value to kernels; the two matrices stay behind device pointers and are
// fetched with scalar loads (wave-uniform addresses)
struct CamScalars {
  int W, H;
  float tanfovx, tanfovy, mod;
  int radius_rule;
  int row8_begin, row8_end;
  const uint32_t* bin_plan;       // planned bins: offsets [tiles8 + 1] (the workspace's copy); NULL = uniform bins of tile_cap slots
  uint32_t bin_limit;             // ... and the slots the workspace holds: no bin reaches past it, whatever the plan says
  uint32_t bwd_flags;             // composite_backward_mx: bit 0 = nobody wants dL/d(first colour set) -- the tracking loop detaches
                                  // the Gaussians, and the pose gradient does not need it: that contraction chain is skipped
  uint32_t scratch_records;       // backward: gradient records the caller's scratch holds.  An instance id at or beyond it is
                                  // neither written (composite) nor read (gather: that Gaussian's gradient is zero): a scratch
                                  // sized from a wrong or stale count gives wrong numbers, not an out-of-bounds access
  uint32_t raw_act;               // round 6 (VTGS_FORWARD_RAW_ACTIVATIONS / frame flag 16): `opacities` are logits, `scales` log-scales [N]
                                  // (isotropic), `rotations` is not read -- project_and_bin and gather_splat_grads apply the activations
  uint32_t no_defer;              // project_and_bin: VTGS_FORWARD_EXPECT_NO_DEFERRED honoured (whole frame, uniform bins): nothing is listed for
                                  // bin_deferred_splats, which is then not launched
#ifdef VTGS_Q_STAMPS
  uint32_t* dbg_proj;             // diagnostic build: 8 words of stamps per workgroup of project_and_bin
#endif
};

// where tile t's bin lives: a wave-uniform branch on a kernel argument (uniform bins pay nothing for the planned form)
struct BinRange { uint32_t s, cap; };
#if defined(__HIPCC__)
__device__ __forceinline__ BinRange bin_range(const uint32_t* __restrict__ plan, uint32_t limit, uint32_t tile, uint32_t tile_cap) {
  BinRange r;
  if (plan) {
    const uint32_t a = min(plan[tile], limit), b = min(plan[tile + 1], limit);
    r.s = a; r.cap = b > a ? b - a : 0u;                      // a plan that outgrew the workspace is flagged by finalize_forward
  } else { r.s = tile * tile_cap; r.cap = tile_cap; }
  return r;
}
__device__ __forceinline__ BinRange bin_range(const CamScalars& cs, uint32_t tile, uint32_t tile_cap) {
  return bin_range(cs.bin_plan, cs.bin_limit, tile, tile_cap);
}
#endif

#if defined(__HIPCC__)
__device__ __forceinline__ CamParams load_cam(const CamScalars& cs, const float* __restrict__ V,
                                              const float* __restrict__ PV) {
  CamParams c;
#pragma unroll
  for (int i = 0; i < 16; ++i) { c.V[i] = V[i]; c.PV[i] = PV[i]; }
  c.W = cs.W; c.H = cs.H;
  c.fx = (float)cs.W / (2.f * cs.tanfovx);
  c.fy = (float)cs.H / (2.f * cs.tanfovy);
  c.limx = kFovClamp * cs.tanfovx; c.limy = kFovClamp * cs.tanfovy;
  c.mod = cs.mod;
  c.gx16 = (cs.W + kBinTile - 1) / kBinTile; c.gy16 = (cs.H + kBinTile - 1) / kBinTile;
  c.gx8 = (cs.W + kSubTile - 1) / kSubTile;  c.gy8 = (cs.H + kSubTile - 1) / kSubTile;
  c.row8_begin = cs.row8_begin; c.row8_end = cs.row8_end;
  c.radius_rule = cs.radius_rule;
  return c;
}

// ---------------------------------------------------------------- wavefront (64-lane) helpers ----
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ float bcast_f(float v, int src_lane) {   // src_lane must be wave-uniform
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}
__device__ __forceinline__ int bcast_i(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }

__device__ __forceinline__ float wave_sum(float v) {   // butterfly; every lane ends with the total
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}
// Integer reductions / scans over the wavefront through DPP row operations + four readlanes instead of six ds_bpermute round
// trips each (a crossbar trip is ~100+ cycles of latency on a wavefront's critical path -- the counting sort of the forward
// composite makes four such reductions per tile, the projection three per workgroup).  VTGS_WAVE_DPP=0: the shuffle forms.
#ifndef VTGS_WAVE_DPP
#define VTGS_WAVE_DPP 1
#endif
template <int CTRL>
__device__ __forceinline__ int dpp_all(int v) {                 // a DPP control under which every lane has a valid source
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {   // row shifts: a lane whose source is outside its row reads 0
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// op over the 16 lanes of each row (every lane ends with its row's result): lane ^ 1, lane ^ 2, mirror within 8, mirror within 16
#define VTGS_ROW_ALLREDUCE(v, OP)                 \
  v = OP(v, dpp_all<0xB1>(v));                    \
  v = OP(v, dpp_all<0x4E>(v));                    \
  v = OP(v, dpp_all<0x141>(v));                   \
  v = OP(v, dpp_all<0x140>(v))
__device__ __forceinline__ int wave_max_i(int v) {
#if VTGS_WAVE_DPP
  VTGS_ROW_ALLREDUCE(v, max);
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
#else
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m, 64));
  return v;
#endif
}
__device__ __forceinline__ int wave_min_i(int v) {
#if VTGS_WAVE_DPP
  VTGS_ROW_ALLREDUCE(v, min);
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
#else
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = min(v, __shfl_xor(v, m, 64));
  return v;
#endif
}
// inclusive prefix sum over lanes
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#if VTGS_WAVE_DPP
  v += dpp_or_zero<0x111>(v);                                   // row_shr:1, 2, 4, 8: inclusive scan inside each row of 16
  v += dpp_or_zero<0x112>(v);
  v += dpp_or_zero<0x114>(v);
  v += dpp_or_zero<0x118>(v);
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31),
                 r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
  const int l = lane_id();
  return v + (l >= 16 ? r0 : 0u) + (l >= 32 ? r1 : 0u) + (l >= 48 ? r2 : 0u);
#else
  const int l = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(v, d, 64);
    if (l >= d) v += t;
  }
  return v;
#endif
}

// XCD-aware block remap (8 XCDs, blocks dealt round-robin): gives each XCD a contiguous chunk of
// the index space so neighbouring tiles share an L2.  Bijective for any nblk.
__device__ __forceinline__ uint32_t xcd_swizzle(uint32_t bid, uint32_t nblk) {
  const uint32_t q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, idx = bid >> 3;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}
#endif  // __HIPCC__

// ---- what follows the binning: longest list, statistics, overflow flags, the host-visible record -------------------------
// Run by ONE workgroup of THREADS threads (all of them call it: it contains a barrier).
struct FinalizeArgs {
  const uint32_t* tile_cnt; uint32_t tiles; Counters* ctr; unsigned long long capacity; uint32_t tile_cap;
  const BlockStats* block_stats; uint32_t nblocks; VtgsForwardInfo* host_record;
  const uint32_t* plan;      // planned bins: the plan this forward binned into (workspace copy), else NULL
  uint32_t* plan_next;       // ... and the caller's persistent plan, rewritten from this forward's list lengths
};

// capacity the next plan gives a bin whose list is `cnt` long now: half again as much + 16, in multiples of 16
__host__ __device__ inline uint32_t planned_bin_capacity(uint32_t cnt) { return (cnt + (cnt >> 1) + 31u) & ~15u; }
template <int THREADS>
__device__ __forceinline__ void finalize_block(const uint32_t* __restrict__ tile_cnt, uint32_t tiles, Counters* __restrict__ ctr,
                                               unsigned long long capacity, uint32_t tile_cap,
                                               const BlockStats* __restrict__ block_stats, uint32_t nblocks,
                                               VtgsForwardInfo* host_record, const uint32_t* __restrict__ plan = nullptr,
                                               uint32_t* __restrict__ plan_next = nullptr) {
  constexpr int kW = THREADS / 64;
  __shared__ uint32_t wmax[kW], svis[kW], sover[kW], swsum[kW], sslots[kW];
  __shared__ unsigned long long sr16[kW], slisted[kW];
  const uint32_t t = threadIdx.x;
  uint32_t mx = 0, over = 0, slots = 0;
  unsigned long long listed = 0;                                 // entries of all lists = the instances actually binned
  for (uint32_t i = t; i < tiles; i += (uint32_t)THREADS) {
    const uint32_t c = tile_cnt[i];
    mx = max(mx, c);
    listed += c;
    slots += planned_bin_capacity(c);                            // what bins sized to these lists take in total
    if (plan) {                                                  // planned bins: every bin against its own capacity
      const uint32_t limit = tiles * tile_cap, a = min(plan[i], limit), b = min(plan[i + 1], limit);
      over |= (c > (b > a ? b - a : 0u)) ? 1u : 0u;
    }
  }
  mx = (uint32_t)wave_max_i((int)mx);
  over = (uint32_t)wave_max_i((int)over);
  uint32_t vis = 0; unsigned long long r16 = 0;
  for (uint32_t i = t; i < nblocks; i += (uint32_t)THREADS) { vis += block_stats[i].visible; r16 += block_stats[i].r16; }
  for (int m = 1; m < 64; m <<= 1) {
    vis += (uint32_t)__shfl_xor((int)vis, m, 64);
    r16 += (unsigned long long)__shfl_xor((long long)r16, m, 64);
    slots += (uint32_t)__shfl_xor((int)slots, m, 64);
    listed += (unsigned long long)__shfl_xor((long long)listed, m, 64);
  }
  // planned bins: the NEXT plan from this forward's list lengths (the caller's persistent buffer; this forward and its
  // backward read the workspace copy).  Wavefront w owns a contiguous run of tiles, walked 64 at a time (coalesced) with a
  // wavefront scan and a running carry; the runs' totals are scanned over the wavefronts in between.
  const uint32_t wv = t >> 6, ln = t & 63u;
  const uint32_t per = ((tiles + (uint32_t)kW - 1u) / (uint32_t)kW + 63u) & ~63u;        // tiles per wavefront, whole rows of 64
  const uint32_t lo = min(wv * per, tiles), hi = min(lo + per, tiles);
  uint32_t run_total = 0;
  if (plan_next) {
    for (uint32_t i = lo + ln; i < hi; i += 64u) run_total += planned_bin_capacity(tile_cnt[i]);
    for (int m = 1; m < 64; m <<= 1) run_total += (uint32_t)__shfl_xor((int)run_total, m, 64);
  }
  if (ln == 0) { wmax[wv] = mx; svis[wv] = vis; sr16[wv] = r16; sover[wv] = over; swsum[wv] = run_total; sslots[wv] = slots; slisted[wv] = listed; }
  __syncthreads();
  unsigned long long slots_next = 0;
  for (int i = 0; i < kW; ++i) slots_next += sslots[i];
  if (plan_next) {
    uint32_t carry = 0;
    for (uint32_t i = 0; i < wv; ++i) carry += swsum[i];
    for (uint32_t i0 = lo; i0 < hi; i0 += 64u) {                 // (wave-uniform bounds)
      const uint32_t i = i0 + ln;
      const uint32_t cap = i < hi ? planned_bin_capacity(tile_cnt[i]) : 0u;
      const uint32_t incl = wave_incl_scan(cap);
      if (i < hi) plan_next[i] = carry + incl - cap;
      carry += (uint32_t)__shfl((int)incl, 63, 64);
    }
    if (t == 0) plan_next[tiles] = (uint32_t)slots_next;
  }
  if (t == 0) {
    uint32_t m = 0, v = 0, ov = 0; unsigned long long r = 0, binned = 0;
    for (int i = 0; i < kW; ++i) { m = max(m, wmax[i]); v += svis[i]; r += sr16[i]; ov |= sover[i]; binned += slisted[i]; }
    // instance IDS handed out: what the capacity and the backward's scratch must hold.  >= the instances binned: a splat binned
    // by bin_deferred_splats holds one id per CANDIDATE tile (DeferRec::inst_base)
    const uint32_t total = ctr->inst_total;
    const bool bin_over = plan ? (ov != 0u) : (m > tile_cap);
    const uint32_t ovf = (((unsigned long long)total > capacity) ? 1u : 0u) | (bin_over ? 2u : 0u);
    const unsigned long long slots = slots_next;     // what bins sized to this forward's lists take in total (planned bins)
    ctr->overflow = ovf;
    ctr->info_instances = ovf ? 0ull : binned;
    ctr->info_needed = total; ctr->info_r16 = r;
    ctr->info_visible = v; ctr->info_max_list = m; ctr->info_overflow = ovf; ctr->info_slots = slots; ctr->info_complete = 1u;
    if (host_record) {                         // the caller's pinned record, device-addressable
      host_record->instances = ovf ? 0ull : binned;
      host_record->instances_needed = total; host_record->tiles16_touched = r;
      host_record->visible = v; host_record->max_tile_list = m; host_record->overflow = ovf;
      host_record->bin_slots_needed = slots;
      __threadfence_system();
      host_record->complete = 1u;              // last: the host treats the record as landed once this is set
    }
  }
}

// ---- pose transform of the fused caller chain (vtgs_frame.hip; also the epilogue of gather_splat_grads<.., FRAME>) --------
struct FramePose { float R[9]; float t[3]; float zr[4]; };   // rotation from the normalised quaternion, translation, depth row

__device__ __forceinline__ FramePose load_pose(const float* __restrict__ q, const float* __restrict__ t,
                                               const float* __restrict__ w2c) {
  // (multiply-adds spelled out, like pose_apply below: every kernel that loads the pose gets the same nine floats -- left to the
  //  compiler, a * b + c is contracted differently from kernel to kernel and two routes to one image differ in the last bit)
  FramePose p;
  const float n = rsqrtf(fmaf(q[3], q[3], fmaf(q[2], q[2], fmaf(q[1], q[1], q[0] * q[0]))));
  const float r = q[0] * n, x = q[1] * n, y = q[2] * n, z = q[3] * n;
  p.R[0] = fmaf(-2.f, fmaf(y, y, z * z), 1.f); p.R[1] = 2.f * fmaf(x, y, -(r * z));       p.R[2] = 2.f * fmaf(x, z, r * y);
  p.R[3] = 2.f * fmaf(x, y, r * z);            p.R[4] = fmaf(-2.f, fmaf(x, x, z * z), 1.f); p.R[5] = 2.f * fmaf(y, z, -(r * x));
  p.R[6] = 2.f * fmaf(x, z, -(r * y));         p.R[7] = 2.f * fmaf(y, z, r * x);           p.R[8] = fmaf(-2.f, fmaf(x, x, y * y), 1.f);
  p.t[0] = t[0]; p.t[1] = t[1]; p.t[2] = t[2];
  p.zr[0] = w2c[8]; p.zr[1] = w2c[9]; p.zr[2] = w2c[10]; p.zr[3] = w2c[11];    // row 2 of the row-major 4x4
  return p;
}

// The transform itself, with the multiply-adds spelled out: three kernels evaluate it (prepare_frame_kernel,
// prepare_frame_pose_kernel, the epilogue of gather_splat_grads) and the compiler contracts a * b + c differently from one
// kernel to the next -- the camera-frame means of two routes then differ in the last bit, and tests that compare routes
// bit for bit (tests/test_gpu_fused_frame.py) see it.
__device__ __forceinline__ void pose_apply(const FramePose& P, float x, float y, float z, float& cx, float& cy, float& cz, float& zz) {
  cx = fmaf(P.R[2], z, fmaf(P.R[1], y, fmaf(P.R[0], x, P.t[0])));
  cy = fmaf(P.R[5], z, fmaf(P.R[4], y, fmaf(P.R[3], x, P.t[1])));
  cz = fmaf(P.R[8], z, fmaf(P.R[7], y, fmaf(P.R[6], x, P.t[2])));
  zz = fmaf(P.zr[2], cz, fmaf(P.zr[1], cy, fmaf(P.zr[0], cx, P.zr[3])));
}

// What vtgs_prepare_frame_backward needs besides the operator gradients, for the backward that runs it in the gather
// kernel (vtgs_backward_dual_frame): flags as there (bit 0 geometry, bit 1 pose, bit 2 appearance).
// `idx` (owned sets of the tile-row partition, vtgs_prepare_frame_owned): the rasterizer saw the compact arrays of the listed
// Gaussians; row gid of them is Gaussian idx[gid] of the map, which is where the parameters are read and the gradients land.
struct FrameEpilogue {
  uint32_t flags;
  const float* means3D_world; const float* unnorm_rot; const float* cam_q; const float* cam_t; const float* depth_w2c;
  float* g_rgb; float* g_means3D; float* g_logit; float* g_log_scales; float* g_unnorm_rot; float* pose_partials;
  const int32_t* idx;
};

}  // namespace vtgs
