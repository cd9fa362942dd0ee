// vtgs_sort_common.h -- the one-wavefront-per-list register sort, shared by sort_tiles (vtgs_binning.hip) and by the
// quadrant-queue forward composite, which sorts its own tile's list when the bins cannot hold lists above 1024 entries.
#pragma once
#include "vtgs_internal.h"
#include <type_traits>

namespace vtgs {

// Per-tile depth sort.  Network: the all-ascending form of the bitonic sorter -- each merge level starts with a
// mirror step (e <-> e ^ (k2-1)) followed by half-cleaners (e <-> e ^ j).  Every comparator puts the smaller key at
// the lower index, so a list of any length behaves as if padded with +inf up to the next power of two.
//
// Fast path (lists <= 1024, i.e. practically all of them): ONE WAVEFRONT PER TILE, keys in registers.  Blocked
// layout, element e = lane*E + r with E = n2/64 registers per lane: strides below E are in-lane compare-exchanges,
// strides >= E are lane-xor exchanges (DPP / ds_bpermute), no LDS traffic and no barriers.
// Slow path (longer lists): the whole workgroup cooperates, in LDS up to kSortLds entries, else in place in global
// memory (L2-resident) with the same network.
constexpr int kSortLds = 2048;
constexpr int kWaveSortMax = 1024;        // key + value form: 16 keys per lane
constexpr int kWaveSortMaxPacked = 2048;  // packed form (payload in the key): 32 keys per lane still fit the register budget

// HASV = false sorts the keys alone (the payload travels in their low bits, see wave_sort_tile): a third less to move.
//
// Code size matters more than the exchange instruction here: fully unrolled, the five list-length variants of this network
// came to ~35 K instructions (~250 KB), far beyond the instruction cache that the CUs share, and the kernel took the same
// 27-45 us whether it sorted 1,600 or 12,750 lists and whether lanes exchanged through ds_bpermute or DPP -- it was
// fetching instructions.  So only what must be unrolled is (register indices: the in-lane comparators); the merge levels
// and the cross-lane half-cleaners are real loops over a run-time lane mask.
template <int E, bool HASV>
__device__ __forceinline__ void in_lane_cleaners(unsigned long long (&k)[E], uint32_t (&v)[E]) {   // partner = e ^ j, j = E/2 .. 1
#pragma unroll
  for (int j = E >> 1; j > 0; j >>= 1) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
      if (!(r & j)) {
        const int p = r | j;
        const bool sw = k[r] > k[p];
        const unsigned long long a = k[r], b = k[p];
        const uint32_t va = v[r], vb = v[p];
        k[r] = sw ? b : a; k[p] = sw ? a : b;
        if (HASV) { v[r] = sw ? vb : va; v[p] = sw ? va : vb; }
      }
    }
  }
}

#ifndef VTGS_SORT_DPP
#define VTGS_SORT_DPP 0
#endif

__device__ __forceinline__ unsigned long long lane_fetch64(unsigned long long x, int byte_addr) {   // x of lane byte_addr / 4
  const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(uint32_t)x);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(uint32_t)(x >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// x of lane (lane ^ MASK) for the masks the network uses, without the LDS crossbar: DPP row operations inside a 16-lane row,
// v_permlane16_swap / v_permlane32_swap across rows (semantics: tests/micro/permlane_swap.hip).  OFF by default
// (VTGS_SORT_DPP = 0): measured inside composite_forward_q on one box it is SLOWER than ds_bpermute (133.4 / 131.5 us against
// 127.3 / 126.2, gpurun_out/r3/timing_e.txt).  The network takes 19 k of the wavefront's 62 k cycles (profiles/r3_stamps.md),
// but not because a crossbar round trip is slow: the sorting wavefront competes for vector issue with three wavefronts that
// are compositing, and a ds_bpermute is work for the otherwise idle LDS unit while DPP moves (plus the wait states they need
// behind the compare-exchange that produced their source) are more vector instructions.  Kept for the record and for A/B.
template <int MASK>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t x, int lane) {
  constexpr int kQuad1 = 0xB1, kQuad2 = 0x4E, kQuad3 = 0x1B, kRowMirror = 0x140, kHalfMirror = 0x141, kRor8 = 0x128;
  auto dpp = [](uint32_t v, auto ctrl) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, decltype(ctrl)::value, 0xf, 0xf, false); };
  auto x16 = [&](uint32_t v) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (lane & 16) ? r[0] : r[1];
  };
  auto x32 = [&](uint32_t v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane & 32) ? r[0] : r[1];
  };
  if constexpr (MASK == 1) return dpp(x, std::integral_constant<int, kQuad1>{});
  else if constexpr (MASK == 2) return dpp(x, std::integral_constant<int, kQuad2>{});
  else if constexpr (MASK == 3) return dpp(x, std::integral_constant<int, kQuad3>{});
  else if constexpr (MASK == 4) return dpp(dpp(x, std::integral_constant<int, kHalfMirror>{}), std::integral_constant<int, kQuad3>{});
  else if constexpr (MASK == 7) return dpp(x, std::integral_constant<int, kHalfMirror>{});
  else if constexpr (MASK == 8) return dpp(x, std::integral_constant<int, kRor8>{});
  else if constexpr (MASK == 15) return dpp(x, std::integral_constant<int, kRowMirror>{});
  else if constexpr (MASK == 16) return x16(x);
  else if constexpr (MASK == 31) return x16(dpp(x, std::integral_constant<int, kRowMirror>{}));
  else if constexpr (MASK == 32) return x32(x);
  else { static_assert(MASK == 63, "lane mask not used by the network"); return x32(x16(dpp(x, std::integral_constant<int, kRowMirror>{}))); }
}
// DPP: lists up to VTGS_SORT_DPP_MAX_E keys per lane (the common ones: 4 keys per lane = 256 entries); the long-list forms
// keep the LDS crossbar -- with 16 or 32 keys per lane the extra temporaries of the row swaps push them into scratch
#ifndef VTGS_SORT_DPP_MAX_E
#define VTGS_SORT_DPP_MAX_E 8
#endif
template <int MASK, bool DPP>
__device__ __forceinline__ unsigned long long lane_xor64(unsigned long long x, int lane) {
  if constexpr (DPP)
    return ((unsigned long long)lane_xor32<MASK>((uint32_t)(x >> 32), lane) << 32) | lane_xor32<MASK>((uint32_t)x, lane);
  else
    return lane_fetch64(x, (lane ^ MASK) << 2);
}
template <int MASK, bool DPP>
__device__ __forceinline__ uint32_t lane_xor_v(uint32_t x, int lane) {
  if constexpr (DPP) return lane_xor32<MASK>(x, lane);
  else return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ MASK) << 2, (int)x);
}

// mirror step of a merge level: lane ^ M, in-lane index mirrored.  Registers r and E-1-r trade places with their mirror images
// in the partner lane: done pair by pair, so that only two fetched keys are live at a time (a [E] array of them doubles the
// register footprint of the E = 32 form)
template <int E, bool HASV, int M>
__device__ __forceinline__ void mirror_step(unsigned long long (&k)[E], uint32_t (&v)[E], int lane) {
  constexpr bool kDpp = VTGS_SORT_DPP != 0 && E <= VTGS_SORT_DPP_MAX_E;
  const bool lower = (lane & ((M + 1) >> 1)) == 0;
#pragma unroll
  for (int r = 0; r < E / 2 + (E == 1 ? 1 : 0); ++r) {
    const int m2 = E - 1 - r;
    const unsigned long long pa = lane_xor64<M, kDpp>(k[m2], lane);          // partner's mirror of r
    const uint32_t va = HASV ? lane_xor_v<M, kDpp>(v[m2], lane) : 0u;
    unsigned long long pb = 0ull; uint32_t vb = 0u;
    if (m2 != r) {
      pb = lane_xor64<M, kDpp>(k[r], lane);                                   // partner's mirror of E-1-r
      vb = HASV ? lane_xor_v<M, kDpp>(v[r], lane) : 0u;
    }
    const bool ta = lower ? (pa < k[r]) : (pa > k[r]);
    k[r] = ta ? pa : k[r];
    if (HASV) v[r] = ta ? va : v[r];
    if (m2 != r) {
      const bool tb = lower ? (pb < k[m2]) : (pb > k[m2]);
      k[m2] = tb ? pb : k[m2];
      if (HASV) v[m2] = tb ? vb : v[m2];
    }
  }
}
// half-cleaner across lanes: lane ^ m, same register
template <int E, bool HASV, int m>
__device__ __forceinline__ void cleaner_step(unsigned long long (&k)[E], uint32_t (&v)[E], int lane) {
  constexpr bool kDpp = VTGS_SORT_DPP != 0 && E <= VTGS_SORT_DPP_MAX_E;
  const bool lower = (lane & m) == 0;
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const unsigned long long pk = lane_xor64<m, kDpp>(k[r], lane);
    const uint32_t pv = HASV ? lane_xor_v<m, kDpp>(v[r], lane) : 0u;
    const bool take = lower ? (pk < k[r]) : (pk > k[r]);
    k[r] = take ? pk : k[r];
    if (HASV) v[r] = take ? pv : v[r];
  }
}

template <int E, bool HASV>
__device__ __forceinline__ void wave_sort_regs(unsigned long long (&k)[E], uint32_t (&v)[E], int lane) {
  // levels inside a lane (k2 <= E): compile-time register pairs
#pragma unroll
  for (int k2 = 2; k2 <= E; k2 <<= 1) {
#pragma unroll
    for (int r = 0; r < E; ++r) {                                // mirror step: partner = r ^ (k2 - 1)
      const int p = r ^ (k2 - 1);
      if (r < p) {
        const bool sw = k[r] > k[p];
        const unsigned long long a = k[r], b = k[p];
        const uint32_t va = v[r], vb = v[p];
        k[r] = sw ? b : a; k[p] = sw ? a : b;
        if (HASV) { v[r] = sw ? vb : va; v[p] = sw ? va : vb; }
      }
    }
#pragma unroll
    for (int j = k2 >> 2; j > 0; j >>= 1) {
#pragma unroll
      for (int r = 0; r < E; ++r) {
        if (!(r & j)) {
          const int p = r | j;
          const bool sw = k[r] > k[p];
          const unsigned long long a = k[r], b = k[p];
          const uint32_t va = v[r], vb = v[p];
          k[r] = sw ? b : a; k[p] = sw ? a : b;
          if (HASV) { v[r] = sw ? vb : va; v[p] = sw ? va : vb; }
        }
      }
    }
  }
  // levels across lanes: lane mask M = 1, 3, 7, .. 63 (k2 = 2E .. 64E)
  if constexpr (!(VTGS_SORT_DPP != 0 && E <= VTGS_SORT_DPP_MAX_E)) {
    // long lists: the LDS crossbar with run-time lane masks -- merge levels and cleaners are real loops (code size, above)
#pragma unroll 1
    for (int M = 1; M < 64; M = 2 * M + 1) {
      {
        const int addr = (lane ^ M) << 2;
        const bool lower = (lane & ((M + 1) >> 1)) == 0;
#pragma unroll
        for (int r = 0; r < E / 2 + (E == 1 ? 1 : 0); ++r) {
          const int m2 = E - 1 - r;
          const unsigned long long pa = lane_fetch64(k[m2], addr);          // partner's mirror of r
          const uint32_t va = HASV ? (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v[m2]) : 0u;
          unsigned long long pb = 0ull; uint32_t vb = 0u;
          if (m2 != r) {
            pb = lane_fetch64(k[r], addr);                                   // partner's mirror of E-1-r
            vb = HASV ? (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v[r]) : 0u;
          }
          const bool ta = lower ? (pa < k[r]) : (pa > k[r]);
          k[r] = ta ? pa : k[r];
          if (HASV) v[r] = ta ? va : v[r];
          if (m2 != r) {
            const bool tb = lower ? (pb < k[m2]) : (pb > k[m2]);
            k[m2] = tb ? pb : k[m2];
            if (HASV) v[m2] = tb ? vb : v[m2];
          }
        }
      }
#pragma unroll 1
      for (int m = (M + 1) >> 2; m > 0; m >>= 1) {                 // half-cleaners across lanes: lane ^ m
        const int addr = (lane ^ m) << 2;
        const bool lower = (lane & m) == 0;
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const unsigned long long pk = lane_fetch64(k[r], addr);
          const uint32_t pv = HASV ? (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v[r]) : 0u;
          const bool take = lower ? (pk < k[r]) : (pk > k[r]);
          k[r] = take ? pk : k[r];
          if (HASV) v[r] = take ? pv : v[r];
        }
      }
      in_lane_cleaners<E, HASV>(k, v);
    }
    return;
  }
  // short lists: DPP / row-swap exchanges.  The level and the cleaner distance stay loop variables; the exchange needs its
  // lane mask at compile time, hence the switches.
#pragma unroll 1
  for (int lvl = 0; lvl < 6; ++lvl) {
    switch (lvl) {
      case 0: mirror_step<E, HASV, 1>(k, v, lane); break;
      case 1: mirror_step<E, HASV, 3>(k, v, lane); break;
      case 2: mirror_step<E, HASV, 7>(k, v, lane); break;
      case 3: mirror_step<E, HASV, 15>(k, v, lane); break;
      case 4: mirror_step<E, HASV, 31>(k, v, lane); break;
      default: mirror_step<E, HASV, 63>(k, v, lane); break;
    }
#pragma unroll 1
    for (int c = lvl - 1; c >= 0; --c) {                          // half-cleaners: lane ^ 2^c
      switch (c) {
        case 0: cleaner_step<E, HASV, 1>(k, v, lane); break;
        case 1: cleaner_step<E, HASV, 2>(k, v, lane); break;
        case 2: cleaner_step<E, HASV, 4>(k, v, lane); break;
        case 3: cleaner_step<E, HASV, 8>(k, v, lane); break;
        default: cleaner_step<E, HASV, 16>(k, v, lane); break;
      }
    }
    in_lane_cleaners<E, HASV>(k, v);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Counting sort of one tile's bin by one wavefront (round 3).  Inside composite_forward_q the bitonic network above was the
// largest single phase of a wavefront's life (31 %, profiles/r3_stamps.md): ~650 vector instructions + 168 crossbar
// exchanges for a 225-entry list, issued in competition with three compositing wavefronts.  Depths of one 8x8 tile are a
// narrow, smooth range, so a bucket pass does nearly all of the ordering:
//   1. keys in registers (striped: element r * 64 + lane), min / max of their depth bits over the wavefront;
//   2. bucket = floor((depth bits - min) * 256 / (range + 1)) -- monotone in the depth -- counted with one returning LDS
//      atomic per key, which also hands the key its arrival index in the bucket;
//   3. exclusive scan of the 256 counts (four per lane + a wavefront scan), longest bucket;
//   4. keys to a staging array at bucket base + arrival index, then every key counts the keys of ITS bucket that are smaller
//      (full 64-bit compare: depth bits, then Gaussian id -- the published stable order) and so learns its final position.
// ~150 vector instructions and ~40 LDS operations instead of the network, no payload packing (any N), and the wavefront's
// own copy of the sorted ids stays in LDS for its first chunks.  A bucket longer than kCountSortBucketMax (many exactly equal
// depths) sends the tile to the network instead: same lists, bit for bit (tests/test_gpu_parity.py: sort tests).
constexpr int kCountSortMax = 512;           // entries: 8 keys per lane
constexpr int kCountSortBucketMax = 12;

// unsigned order through the signed reductions of vtgs_internal.h: flipping the top bit maps one order onto the other
__device__ __forceinline__ uint32_t wave_min_u(uint32_t v) { return (uint32_t)wave_min_i((int)(v ^ 0x80000000u)) ^ 0x80000000u; }
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v) { return (uint32_t)wave_max_i((int)(v ^ 0x80000000u)) ^ 0x80000000u; }

// stage: 64 E x 8 bytes, cnt: 256 x 4 bytes (LDS of this wavefront alone); lgid: the wavefront's LDS copy of the sorted ids,
// first lgid_cap entries.  Returns false (nothing written) when a bucket is too long for step 4.
template <int E>
__device__ __forceinline__ bool wave_count_sort(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                uint32_t* __restrict__ sorted_gid, uint32_t* __restrict__ sorted_inst, size_t s,
                                                uint32_t L, int lane, unsigned long long* __restrict__ stage,
                                                uint32_t* __restrict__ cnt, uint32_t* __restrict__ lgid, uint32_t lgid_cap) {
  unsigned long long k[E]; uint32_t v[E];
  uint32_t dmin = 0xFFFFFFFFu, dmax = 0u;
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const uint32_t e = (uint32_t)r * 64u + (uint32_t)lane;
    const bool in = e < L;
    k[r] = in ? keys[s + e] : ~0ull;
    v[r] = in ? vals[s + e] : 0u;
    const uint32_t d = (uint32_t)(k[r] >> 32);
    dmin = in ? min(dmin, d) : dmin;
    dmax = in ? max(dmax, d) : dmax;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) cnt[64 * j + lane] = 0u;
  dmin = wave_min_u(dmin); dmax = wave_max_u(dmax);
  const float scale = 256.0f / ((float)(dmax - dmin) + 1.0f);
  uint32_t b[E], pib[E];
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const bool in = (uint32_t)r * 64u + (uint32_t)lane < L;
    b[r] = min(255u, (uint32_t)((float)((uint32_t)(k[r] >> 32) - dmin) * scale));
    pib[r] = 0u;
    if (in) pib[r] = atomicAdd(&cnt[b[r]], 1u);
  }
  // exclusive scan of the 256 counts: lane l owns buckets 4 l .. 4 l + 3
  const uint4 c = reinterpret_cast<const uint4*>(cnt)[lane];
  const uint32_t t1 = c.x + c.y, t2 = t1 + c.z, t3 = t2 + c.w;
  const uint32_t excl = wave_incl_scan(t3) - t3;
  const uint32_t longest = wave_max_u(max(max(c.x, c.y), max(c.z, c.w)));
  if (longest > (uint32_t)kCountSortBucketMax) return false;                 // wave-uniform
  reinterpret_cast<uint4*>(cnt)[lane] = make_uint4((excl << 16) | c.x, ((excl + c.x) << 16) | c.y, ((excl + t1) << 16) | c.z,
                                                   ((excl + t2) << 16) | c.w);
  uint32_t base[E], nb[E];
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const bool in = (uint32_t)r * 64u + (uint32_t)lane < L;
    const uint32_t pc = cnt[b[r]];
    base[r] = pc >> 16; nb[r] = pc & 0xFFFFu;
    if (in) stage[base[r] + pib[r]] = k[r];
  }
  uint32_t rank[E];
#pragma unroll
  for (int r = 0; r < E; ++r) rank[r] = 0u;
  for (uint32_t j = 0; j < longest; ++j) {                                    // wave-uniform trip count, <= kCountSortBucketMax
#pragma unroll
    for (int r = 0; r < E; ++r) {
      const bool in = (uint32_t)r * 64u + (uint32_t)lane < L && j < nb[r];
      const unsigned long long o = stage[in ? base[r] + j : 0u];
      rank[r] += (in && o < k[r]) ? 1u : 0u;
    }
  }
#pragma unroll
  for (int r = 0; r < E; ++r) {
    if ((uint32_t)r * 64u + (uint32_t)lane < L) {
      const uint32_t pos = base[r] + rank[r];
      sorted_gid[s + pos] = (uint32_t)k[r];
      sorted_inst[s + pos] = v[r];
      if (pos < lgid_cap) lgid[pos] = (uint32_t)k[r];
    }
  }
  return true;
}

// The same counting sort by a whole 256-thread workgroup for ONE list of 513 .. 2048 entries (dense maps: 2 M Gaussians at
// 640x480 average ~1,300 per tile): 1,024 buckets, eight keys per thread, stage / cnt in the workgroup's LDS (16 KB + 4 KB).
// Replaces the 32-keys-per-lane register network for such lists (round 3: sort_tiles 113 -> see profiles/r3_long_list_sort.txt).
// All 256 threads call it (barriers inside); returns false -- nothing written -- when a bucket is too long for the rank pass.
constexpr int kBlockSortMax = 2048;
constexpr int kBlockSortBuckets = 1024;
__device__ __forceinline__ bool block_count_sort(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                 uint32_t* __restrict__ sorted_gid, uint32_t* __restrict__ sorted_inst, size_t s,
                                                 uint32_t L, uint32_t t, unsigned long long* __restrict__ stage,
                                                 uint32_t* __restrict__ cnt) {
  constexpr int E = kBlockSortMax / 256;                         // 8 keys per thread: element r * 256 + t
  __shared__ uint32_t red[8];
  unsigned long long k[E];
  uint32_t dmin = 0xFFFFFFFFu, dmax = 0u;
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const uint32_t e = (uint32_t)r * 256u + t;
    const bool in = e < L;
    k[r] = in ? keys[s + e] : ~0ull;
    const uint32_t d = (uint32_t)(k[r] >> 32);
    dmin = in ? min(dmin, d) : dmin;
    dmax = in ? max(dmax, d) : dmax;
  }
#pragma unroll
  for (int j = 0; j < kBlockSortBuckets / 256; ++j) cnt[256 * j + t] = 0u;
  dmin = wave_min_u(dmin); dmax = wave_max_u(dmax);
  if ((t & 63u) == 0u) { red[t >> 6] = dmin; red[4 + (t >> 6)] = dmax; }
  __syncthreads();
  dmin = min(min(red[0], red[1]), min(red[2], red[3]));
  dmax = max(max(red[4], red[5]), max(red[6], red[7]));
  const float scale = (float)kBlockSortBuckets / ((float)(dmax - dmin) + 1.0f);
  uint32_t bp[E];                                               // bucket (10 bits) | arrival index in the bucket << 16
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const bool in = (uint32_t)r * 256u + t < L;
    const uint32_t b = min((uint32_t)kBlockSortBuckets - 1u, (uint32_t)((float)((uint32_t)(k[r] >> 32) - dmin) * scale));
    bp[r] = b;
    if (in) bp[r] |= atomicAdd(&cnt[b], 1u) << 16;
  }
  __syncthreads();
  // exclusive scan of the 1,024 counts: thread t owns buckets 4 t .. 4 t + 3
  const uint4 c = reinterpret_cast<const uint4*>(cnt)[t];
  const uint32_t t1 = c.x + c.y, t2 = t1 + c.z, t3 = t2 + c.w;
  const uint32_t incl = wave_incl_scan(t3);
  uint32_t longest = wave_max_u(max(max(c.x, c.y), max(c.z, c.w)));
  __syncthreads();                                              // (red is read above by everybody before it is rewritten)
  if ((t & 63u) == 63u) red[t >> 6] = incl;
  if ((t & 63u) == 0u) red[4 + (t >> 6)] = longest;
  __syncthreads();
  uint32_t excl = incl - t3;
  for (uint32_t w = 0; w < (t >> 6); ++w) excl += red[w];
  longest = max(max(red[4], red[5]), max(red[6], red[7]));
  if (longest > (uint32_t)kCountSortBucketMax) return false;    // workgroup-uniform
  reinterpret_cast<uint4*>(cnt)[t] = make_uint4((excl << 16) | c.x, ((excl + c.x) << 16) | c.y, ((excl + t1) << 16) | c.z,
                                                ((excl + t2) << 16) | c.w);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < E; ++r) {
    if ((uint32_t)r * 256u + t < L) stage[(cnt[bp[r] & 0xFFFFu] >> 16) + (bp[r] >> 16)] = k[r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const uint32_t e = (uint32_t)r * 256u + t;
    if (e < L) {
      const uint32_t pc = cnt[bp[r] & 0xFFFFu], base = pc >> 16, nb = pc & 0xFFFFu;
      uint32_t rank = 0u;
      for (uint32_t j = 0; j < nb; ++j) rank += (stage[base + j] < k[r]) ? 1u : 0u;
      const uint32_t pos = base + rank;
      sorted_gid[s + pos] = (uint32_t)k[r];
      sorted_inst[s + pos] = vals[s + e];
    }
  }
  __syncthreads();                                              // stage / cnt are the caller's to reuse
  return true;
}

// PACKED (Gaussian ids below 2^21, list positions below 2^11 -- the host decides): the low key word becomes
// (gid << 11 | bin slot); the order (depth, gid) is unchanged, the instance id is fetched from its slot afterwards.
constexpr uint32_t kSlotBits = 11u;            // (the host sends ids below 2^21 only: 21 + 11 bits)

template <int E, bool PACKED>
__device__ __forceinline__ void wave_sort_tile(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                               uint32_t* __restrict__ sorted_gid, uint32_t* __restrict__ sorted_inst,
                                               size_t s, uint32_t L, int lane, unsigned long long* stamp = nullptr) {
  unsigned long long k[E]; uint32_t v[E];
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const uint32_t e = (uint32_t)lane * E + r;
    k[r] = (e < L) ? keys[s + e] : ~0ull;
    if (PACKED) {
      if (e < L) k[r] = (k[r] & 0xFFFFFFFF00000000ull) | (unsigned long long)((((uint32_t)k[r]) << kSlotBits) | e);
      v[r] = 0u;
    } else {
      v[r] = (e < L) ? vals[s + e] : 0u;
    }
  }
#ifdef VTGS_Q_STAMPS
  if (stamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp[0] = __builtin_amdgcn_s_memtime(); }
#endif
  wave_sort_regs<E, !PACKED>(k, v, lane);
#ifdef VTGS_Q_STAMPS
  if (stamp) { asm volatile("" :: "v"(k[0])); stamp[1] = __builtin_amdgcn_s_memtime(); }
#endif
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const uint32_t e = (uint32_t)lane * E + r;
    if (e < L) {
      if (PACKED) {
        const uint32_t lo = (uint32_t)k[r];
        sorted_gid[s + e] = lo >> kSlotBits;
        sorted_inst[s + e] = vals[s + (lo & ((1u << kSlotBits) - 1u))];
      } else {
        sorted_gid[s + e] = (uint32_t)k[r]; sorted_inst[s + e] = v[r];
      }
    }
  }
}


}  // namespace vtgs
