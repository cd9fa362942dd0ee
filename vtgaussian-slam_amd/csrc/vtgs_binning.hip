// vtgs_binning.hip -- projection, per-tile binning and per-tile depth sort (gfx950, wave64).
//
// Pipeline (all on the caller's stream, no host round trip):
//   project_and_bin   one thread per Gaussian: EWA projection (vtgs_math.h), then walks the 8x8 tiles under its
//                     16x16-tile rectangle, keeps those the splat's alpha>=1/255 ellipse can reach, reserves a slot
//                     in each tile's fixed-capacity bin with run-aggregated atomics (Gaussians of a view-tied map are
//                     stored in raster order, so neighbouring lanes hit the same tile and one atomic serves a whole
//                     run of lanes) and writes the 64-bit key (depth bits | Gaussian id) + instance id in place.
//   sort_tiles        per-tile bitonic sort of (key, value): one wavefront per tile, keys in registers; key order ==
//                     the stable (tile, depth) order of the published algorithm because ties fall back to Gaussian id.
//   finalize_forward  one workgroup: longest list, statistics, overflow flags (the later kernels bail on them), the
//                     host-visible result record.
//
// The replaced implementation [UPSTREAM-PUBLIC] sorts all (tile|depth) keys with a global radix sort (several passes
// over 12 B x R) and reads the total back to the host to size it.  Here the tile is resolved by binning (one pass,
// no prefix scan) and only the short per-tile lists are sorted, on chip.
#include "../../include/vtgs.h"
#include "vtgs_internal.h"
#include "vtgs_sort_common.h"

namespace vtgs {

// Reserve one slot per active lane in counters[tile]; lanes of a run (consecutive active lanes with the same tile)
// share one atomic.  Split in two so the caller can overlap the atomic's round trip with other work:
// reserve_issue() launches the atomic and returns what is needed later, reserve_resolve() turns it into the slot.
struct Reservation { uint32_t base; int head_lane; int rank; bool act; int tile; };

__device__ __forceinline__ Reservation reserve_issue(uint32_t* __restrict__ counters, int tile, bool act) {
  const int l = lane_id();
  const int prev_tile = __shfl_up(tile, 1, 64);
  const int prev_act = __shfl_up((int)act, 1, 64);
  const bool head = act && (l == 0 || !prev_act || prev_tile != tile);
  const unsigned long long H = __ballot(head);
  const unsigned long long A = __ballot(act);
  const unsigned long long stops = H | ~A;                       // lanes where a run cannot continue
  const unsigned long long le_mask = (l == 63) ? ~0ull : ((2ull << l) - 1ull);   // bits <= l
  const unsigned long long heads_le = H & le_mask;
  Reservation r;
  r.head_lane = heads_le ? (63 - __builtin_clzll(heads_le)) : 0;   // head lane of my run
  r.rank = l - r.head_lane;
  r.act = act; r.tile = tile;
  r.base = 0;
  if (head) {
    const unsigned long long above = stops & ~le_mask;            // bits > l
    const int e = above ? __builtin_ctzll(above) : 64;
    r.base = atomicAdd(&counters[tile], (uint32_t)(e - l));
  }
  return r;
}
__device__ __forceinline__ uint32_t reserve_resolve(const Reservation& r) {   // all 64 lanes must call it
  return (uint32_t)__shfl((int)r.base, r.head_lane, 64) + (uint32_t)r.rank;
}

struct TileWalk {       // candidate 8x8 tiles of one splat: those under its 16x16-tile rectangle, in the band
  int cx0, cy0, cw, ch;
};

__device__ __forceinline__ TileWalk make_walk(const CamParams& cam, const Splat& sp, bool reach) {
  TileWalk w{0, 0, 0, 0};
  if (!reach) return w;
  const int x0 = 2 * sp.x0, x1 = min(2 * sp.x1, cam.gx8);
  const int y0 = max(2 * sp.y0, cam.row8_begin), y1 = min(min(2 * sp.y1, cam.gy8), cam.row8_end);
  if (x1 > x0 && y1 > y0) { w.cx0 = x0; w.cy0 = y0; w.cw = x1 - x0; w.ch = y1 - y0; }
  return w;
}

// does the splat's alpha >= 1/255 region reach the pixel centres of 8x8 tile (tx,ty)?
// Same minimum as vtgs_math.h's min_quadratic_over_rect (the four edge minima of q = 1/2 (A dx^2 + C dy^2) + B dx dy, or 0
// when the centre lies inside), with the two divisions hoisted out of the per-tile loop (they were half of this kernel's
// vector instructions) and v_med3_f32 for the clamps.  The differences are rounding-level; tau carries 1e-4 of slack.
struct ReachForm { float hA, B, hC, kx, ky; bool regular; };

__device__ __forceinline__ ReachForm make_reach_form(const Splat& sp) {
  ReachForm f;
  f.hA = 0.5f * sp.A; f.B = sp.B; f.hC = 0.5f * sp.C;
  f.regular = sp.A > 0.f && sp.C > 0.f;                        // always true for a projected (positive-definite) conic
  // (hardware reciprocals, 1 ulp: q is evaluated AT the clamped minimiser, so an error eps in it raises q by O(eps^2))
  f.kx = f.regular ? -sp.B * __builtin_amdgcn_rcpf(sp.C) : 0.f;   // dy* = kx dx on a vertical edge
  f.ky = f.regular ? -sp.B * __builtin_amdgcn_rcpf(sp.A) : 0.f;   // dx* = ky dy on a horizontal edge
  return f;
}

__device__ __forceinline__ bool tile_reached(const CamParams& cam, const Splat& sp, const ReachForm& f, float tau, int tx,
                                             int ty) {
  const float px0 = (float)(tx * kSubTile), py0 = (float)(ty * kSubTile);
  const float px1 = fminf(px0 + (float)(kSubTile - 1), (float)(cam.W - 1));
  const float py1 = fminf(py0 + (float)(kSubTile - 1), (float)(cam.H - 1));
  if (!f.regular) return min_quadratic_over_rect(sp.A, sp.B, sp.C, sp.u, sp.v, px0, py0, px1, py1) <= tau;
  // d = centre - pixel, pixel in the rectangle => dx in [u - px1, u - px0]
  const float dx0 = sp.u - px1, dx1 = sp.u - px0, dy0 = sp.v - py1, dy1 = sp.v - py0;
  const bool inside = dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f;
  auto q = [&](float dx, float dy) { return fmaf(f.hC * dy, dy, dx * fmaf(f.hA, dx, f.B * dy)); };
  const float q0 = q(dx0, __builtin_amdgcn_fmed3f(f.kx * dx0, dy0, dy1));
  const float q1 = q(dx1, __builtin_amdgcn_fmed3f(f.kx * dx1, dy0, dy1));
  const float q2 = q(__builtin_amdgcn_fmed3f(f.ky * dy0, dx0, dx1), dy0);
  const float q3 = q(__builtin_amdgcn_fmed3f(f.ky * dy1, dx0, dx1), dy1);
  return (inside ? 0.f : fminf(fminf(q0, q1), fminf(q2, q3))) <= tau;
}

__device__ __forceinline__ void store_geom(GeomRec* __restrict__ rec, const Splat& sp, float op) {
  float4* gp = reinterpret_cast<float4*>(rec);
  gp[0] = make_float4(sp.u, sp.v, sp.A, sp.B);
  gp[1] = make_float4(sp.C, op, sp.depth, __uint_as_float(pack_centre_lo(sp.ulo, sp.vlo)));
}

constexpr int kProjBlock = 1024;     // threads per workgroup: one allocation atomic per 1024 Gaussians
constexpr uint32_t kWinEntries = 16384;  // windowed LDS tile table (LDSBINS == 2): 64 KB, two workgroups per CU

// A Gaussian whose candidate walk exceeds kDeferArea tiles is DEFERRED: its own lane does not walk it.  (Rounds 2-5: only walks
// of more than 64 tiles were taken away from their lane, by the lane's wavefront at the end of the kernel, each with an atomic of
// its own.  But every lane of a wavefront waits for the longest walk among the 64, and an optimised SLAM map has a heavy tail
// of splat sizes: after 20 frames of mapping 8 % of the Gaussians of the synthetic Replica sequence cover 7 .. 64 tiles, so
// nearly every wavefront held one and ran 16 .. 64 trips of both passes where a fresh view-tied map runs 4 -- project_and_bin
// 160 us against 46 us, gpurun_out/r6/slamlate_b_dens.txt.)  project_and_bin only LISTS them -- 48-byte records, one list for the
// forward from the front of an array (up to kGroupArea candidates) and one from its end (larger); a workgroup takes its stretches
// with the atomic that takes its instance range -- and a second kernel, bin_deferred_splats, bins them balanced over its lanes:
//   * up to kGroupArea candidate tiles: one 16-lane group per splat, 16 tiles per step;
//   * up to kWaveArea: one wavefront per splat, 64 tiles per step;
//   * more: the whole workgroup, 256 tiles per step (a splat grown over a hole of the map covers thousands).
// A deferred splat's instance ids are reserved by project_and_bin, one per CANDIDATE tile (an upper bound: the ids behind the
// tiles it does not reach stay unused), so the second kernel needs no counter of its own.  Instance ids of a splat follow the
// raster order of its walk, as everywhere.
// A kernel of its own because inlined at the end of project_and_bin the same code kept twenty more scalar registers alive
// through the common path (88 against 68), which that kernel answers with s_load re-materialisation inside its loops: 62 us
// against 52 us at the headline shape (gpurun_out/r6/timing_g_head*.log); as a called function it dragged its registers and a
// scratch frame into the caller (120 VGPRs).  With an empty list every workgroup of the second kernel leaves after one load.
#ifndef VTGS_DEFER_AREA
#define VTGS_DEFER_AREA 9
#endif
constexpr int kDeferArea = VTGS_DEFER_AREA;      // candidate tiles a lane walks itself (3 x 3: what a few-pixel splat can straddle)
constexpr int kGroupArea = 64;                   // deferred splats up to this many candidates: a 16-lane group each
constexpr int kDeferBlock = 256;                 // threads per workgroup of bin_deferred_splats
constexpr int kWaveArea = 512;                   // ... larger ones up to this many: one WAVEFRONT each, 64 tiles per step; beyond: the workgroup
static_assert(kDeferArea >= 1 && kDeferArea <= 64, "the reach mask of the common path holds 64 candidates");

// What the walk of one splat needs besides the splat: the reach threshold, the hoisted reach form, the candidate rectangle
// (shrunk to the bounding box of the alpha >= 1/255 ellipse).  One function for the common path and for the end phase, which
// rebuilds it from the stored geometry record: the same floats through the same operations.
struct WalkSetup { TileWalk w; ReachForm rf; float tau; int area; };

__device__ __forceinline__ WalkSetup setup_walk(const CamParams& cam, const Splat& sp, float op, bool vis) {
  WalkSetup ws;
  // alpha = o*G >= 1/255 somewhere  <=>  q <= ln(255 o); slack keeps the test conservative
  ws.tau = -1.f;
  if (vis && op * 255.f >= 1.f) { ws.tau = __log2f(255.f * op) * 0.69314718f; ws.tau += 1e-4f * ws.tau + 1e-4f; }   // (v_log_f32: 1 ulp, inside the slack)
  const bool reach = vis && ws.tau >= 0.f;
  ws.rf = make_reach_form(sp);
  ws.w = make_walk(cam, sp, reach);
  if (reach && ws.rf.regular) {
    // shrink the walk to the tiles under the bounding box of the alpha >= 1/255 ellipse (half-widths sqrt(2 tau C / det),
    // sqrt(2 tau A / det); tau carries the slack): a few-pixel splat then tests ~4 candidates instead of the 16 under its
    // 16x16-tile rectangle.  The exact test still decides; the box only removes tiles it cannot pass.
    // (hardware reciprocal / square root as in composite_forward_q's quadrant box, with the same 2e-6 relative + 1e-5 px of
    //  slack: the IEEE forms are ~35 instructions per Gaussian and the box only has to be conservative)
    const float idet = __builtin_amdgcn_rcpf(fmaxf(sp.A * sp.C - sp.B * sp.B, 1e-30f));
    const float k2 = 2.f * ws.tau * idet;
    const float hx = __builtin_amdgcn_sqrtf(k2 * sp.C) * 1.000002f + 1e-5f, hy = __builtin_amdgcn_sqrtf(k2 * sp.A) * 1.000002f + 1e-5f;
    const float inv8 = 1.f / (float)kSubTile;
    const int bx0 = (int)ceilf((sp.u - hx - (float)(kSubTile - 1)) * inv8), bx1 = (int)floorf((sp.u + hx) * inv8);
    const int by0 = (int)ceilf((sp.v - hy - (float)(kSubTile - 1)) * inv8), by1 = (int)floorf((sp.v + hy) * inv8);
    const int x0 = max(ws.w.cx0, bx0), x1 = min(ws.w.cx0 + ws.w.cw, bx1 + 1);
    const int y0 = max(ws.w.cy0, by0), y1 = min(ws.w.cy0 + ws.w.ch, by1 + 1);
    if (x1 > x0 && y1 > y0) { ws.w.cx0 = x0; ws.w.cy0 = y0; ws.w.cw = x1 - x0; ws.w.ch = y1 - y0; }
    else { ws.w.cw = 0; ws.w.ch = 0; }
  }
  ws.area = ws.w.cw * ws.w.ch;
  return ws;
}

constexpr int kBigArea = kDeferArea;          // (the name the body of project_and_bin uses)

// LDSBINS (tile count fits the LDS table): the workgroup first histograms its instances per tile in LDS, then takes ONE
// global slot range per touched tile -- all those device-scope atomics are in flight together, so their round trip
// (microseconds: they execute at the memory side) is paid once per workgroup instead of once per step of the lock-step
// walk -- and hands out the slots with LDS atomics.  Otherwise: run-aggregated, software-pipelined global atomics.
// MODE (kernel-uniform facts, compiled apart so that the common case -- whole frame, uniform bins -- keeps its scalar-register
// budget: either feature alone cost that case 13 us of 54 through s_load re-materialisation, batch P of round 3):
//   bit 0  a band of the tile-row partition (row cull before the projection, early exit of workgroups with nothing in the band)
//   bit 1  planned bins (bin t = [plan[t], plan[t+1]), include/vtgs.h)
// CamScalars::no_defer (VTGS_FORWARD_EXPECT_NO_DEFERRED): nothing is deferred -- a lane walks up to 64 candidates itself and a larger splat is
// walked by its wavefront at the end of the kernel, every lane a tile, with an instance-range atomic of its own (the kernel of
// rounds 2-5).  Correct for every map, fast for the maps the hint is given for (none of these splats), and no second launch.
struct BigWalk {            // wave-uniform copy of one lane's walk and reach test
  TileWalk w; Splat sp; ReachForm rf; float tau; int area;
};
__device__ __forceinline__ BigWalk broadcast_walk(const TileWalk& w, const Splat& sp, const ReachForm& rf, float tau, int area,
                                                  int src) {
  BigWalk b;
  auto bi = [&](int v) { return __builtin_amdgcn_readlane(v, src); };
  auto bf = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
  b.w.cx0 = bi(w.cx0); b.w.cy0 = bi(w.cy0); b.w.cw = bi(w.cw); b.w.ch = bi(w.ch);
  b.sp = Splat{};
  b.sp.u = bf(sp.u); b.sp.v = bf(sp.v); b.sp.A = bf(sp.A); b.sp.B = bf(sp.B); b.sp.C = bf(sp.C);
  b.rf.hA = bf(rf.hA); b.rf.B = bf(rf.B); b.rf.hC = bf(rf.hC); b.rf.kx = bf(rf.kx); b.rf.ky = bf(rf.ky);
  b.rf.regular = bi(rf.regular ? 1 : 0) != 0;
  b.tau = bf(tau); b.area = bi(area);
  return b;
}

// The per-Gaussian inputs of the projection besides the mean.  raw (CamScalars::raw_act, kernel-uniform): the fused caller chain's
// isotropic map as its PARAMETERS -- logit, log-scale -- with the activations of utils/slam_helpers.py:127-160 applied here
// (sigmoid, exp on all three axes) and the rotation left at the identity: the covariance s^2 I does not depend on it.
__device__ __forceinline__ void load_activations(bool raw, int gid, const float* __restrict__ opacities, const float* __restrict__ scales,
                                                 const float* __restrict__ rotations, float& op, float (&sc)[3], float (&q)[4]) {
  if (raw) {
    const float lo = opacities[gid], ls = scales[gid];
    op = 1.f / (1.f + __expf(-lo));
    sc[0] = sc[1] = sc[2] = __expf(ls);
    q[0] = 1.f; q[1] = q[2] = q[3] = 0.f;
  } else {
    sc[0] = scales[3 * gid]; sc[1] = scales[3 * gid + 1]; sc[2] = scales[3 * gid + 2];
    const float4 q4 = reinterpret_cast<const float4*>(rotations)[gid];
    q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
    op = opacities[gid];
  }
}

template <int LDSBINS, int MODE>
__device__ __forceinline__ void project_and_bin_body(
    CamScalars cs, const float* __restrict__ Vp, const float* __restrict__ PVp, int n,
    const float* __restrict__ means3D, const float* __restrict__ opacities,
    const float* __restrict__ scales, const float* __restrict__ rotations,
    int32_t* __restrict__ radii, GeomRec* __restrict__ geom, GaussAux* __restrict__ gaux,
    uint32_t* __restrict__ tile_cnt, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals,
    Counters* __restrict__ ctr, BlockStats* __restrict__ block_stats, unsigned long long capacity, uint32_t tile_cap,
    DeferRec* __restrict__ defer_list) {
  constexpr int kWaves = kProjBlock / 64;
  extern __shared__ uint32_t lds_tile[];                         // LDSBINS: one entry per 8x8 tile of this call's band
#ifdef VTGS_Q_STAMPS
  unsigned long long pst[6];
  pst[0] = __builtin_amdgcn_s_memtime();
  const unsigned long long prt0 = __builtin_amdgcn_s_memrealtime();
#define VTGS_P_STAMP(i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pst[i] = __builtin_amdgcn_s_memtime(); }
#else
#define VTGS_P_STAMP(i)
#endif
  const CamParams cam = load_cam(cs, Vp, PVp);
  const int tile0 = cam.row8_begin * cam.gx8;                    // first tile of the band (tile-row multi-GPU partition)
  const int tiles8 = (cam.row8_end - cam.row8_begin) * cam.gx8;
  const int gid = (int)(blockIdx.x * (uint32_t)kProjBlock + threadIdx.x);
  const int l = lane_id();
  const bool valid = gid < n;

  Splat sp{}; SplatAux aux;
  float op = 0.f;
  bool vis = false;
  constexpr bool banded = (MODE & 1) != 0;                      // a rank of the tile-row partition
  constexpr bool planned = (MODE & 2) != 0;
  constexpr bool precomp = (MODE & 4) != 0;                     // `scales` holds cov3D_precomp [N,6], `rotations` is not read
  // (a RUN-TIME fact, not a template parameter: two instantiations contract the projection's multiply-adds differently, and the
  //  first forward of a shape -- no hint yet -- and the later ones would differ in the last bit of their images)
  const bool defer = cs.no_defer == 0u;                         // kernel-uniform; no_defer: no deferred list (see BigWalk)
  if constexpr (precomp) {
    if (valid) {                                                // (no row cull ahead of the projection: it is built on the scales)
      const float mean[3] = {means3D[3 * gid], means3D[3 * gid + 1], means3D[3 * gid + 2]};
      float c6[6];
      for (int i = 0; i < 6; ++i) c6[i] = scales[6 * gid + i];
      const float unused3[3] = {0.f, 0.f, 0.f}, unused4[4] = {1.f, 0.f, 0.f, 0.f};
      op = opacities[gid];
      vis = project_splat(cam, mean, unused3, unused4, op, sp, aux, c6);
      radii[gid] = vis ? sp.radius : 0;
      if (vis) store_geom(geom + gid, sp, op);
    }
  }
  if (valid && !banded && !precomp) {
    // whole frame: the four input streams are requested together (one trip to memory on the kernel's latency chain)
    const float mean[3] = {means3D[3 * gid], means3D[3 * gid + 1], means3D[3 * gid + 2]};
    float sc[3], q[4];
    load_activations(cs.raw_act != 0u, gid, opacities, scales, rotations, op, sc, q);
    vis = project_splat(cam, mean, sc, q, op, sp, aux);
    radii[gid] = vis ? sp.radius : 0;
    // the geometry record is stored NOW, whole (two dwordx4): carrying the centre's float32 remainder to a store behind the
    // instance count cost two vector registers (and with them the second workgroup per CU), and storing the remainder alone
    // here -- a 4-byte write into a 32-byte sector the record's other words fill later -- cost 27 MB of write traffic
    if (vis) store_geom(geom + gid, sp, op);
  }
  if (valid && banded && !precomp) {
    // a band: a Gaussian that cannot meet the band's rows is dropped on its mean and scales alone (outside_tile_rows),
    // before the rotation / opacity loads and the covariance algebra -- 7/8 of them at 8 ranks.  It reports radius 0 on this
    // rank: radii are complete as the MAXIMUM over the ranks.  The survivors pay a second trip to memory.
    const float mean[3] = {means3D[3 * gid], means3D[3 * gid + 1], means3D[3 * gid + 2]};
    float sc[3];
    if (cs.raw_act) { sc[0] = sc[1] = sc[2] = __expf(scales[gid]); }
    else { sc[0] = scales[3 * gid]; sc[1] = scales[3 * gid + 1]; sc[2] = scales[3 * gid + 2]; }
    if (!outside_tile_rows(cam, mean, sc, cam.row8_begin / 2, (cam.row8_end + 1) / 2)) {
      float q[4] = {1.f, 0.f, 0.f, 0.f};
      if (cs.raw_act) { op = 1.f / (1.f + __expf(-opacities[gid])); }
      else {
        const float4 q4 = reinterpret_cast<const float4*>(rotations)[gid];
        q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
        op = opacities[gid];
      }
      vis = project_splat(cam, mean, sc, q, op, sp, aux);
      if (vis) store_geom(geom + gid, sp, op);
    }
    radii[gid] = vis ? sp.radius : 0;
  }
  if (banded) {
    // view-tied maps are stored in raster order, so most WORKGROUPS hold nothing that meets the band -- those leave here,
    // after one trip to memory, instead of walking the reservation chain (three barriers, the instance-range atomic, the
    // per-tile reservations) with nothing to reserve
    if (__syncthreads_or(vis ? 1 : 0) == 0) {
      if (valid) gaux[gid] = GaussAux{0u, 0u};
      if (threadIdx.x == 0) { BlockStats bs; bs.visible = 0; bs.pad = 0; bs.r16 = 0; block_stats[blockIdx.x] = bs; }
      return;
    }
  }
  // alpha = o*G >= 1/255 somewhere  <=>  q <= ln(255 o); slack keeps the test conservative
  float tau = -1.f;
  if (vis && op * 255.f >= 1.f) { tau = __log2f(255.f * op) * 0.69314718f; tau += 1e-4f * tau + 1e-4f; }   // (v_log_f32: 1 ulp, inside the slack)
  const bool reach = vis && tau >= 0.f;
  const ReachForm rf = make_reach_form(sp);
  TileWalk w = make_walk(cam, sp, reach);
  if (reach && rf.regular) {
    // shrink the walk to the tiles under the bounding box of the alpha >= 1/255 ellipse (half-widths sqrt(2 tau C / det),
    // sqrt(2 tau A / det); tau carries the slack): a few-pixel splat then tests ~4 candidates instead of the 16 under its
    // 16x16-tile rectangle.  The exact test below still decides; the box only removes tiles it cannot pass.
    // (hardware reciprocal / square root as in composite_forward_q's quadrant box, with the same 2e-6 relative + 1e-5 px of
    //  slack: the IEEE forms are ~35 instructions per Gaussian and the box only has to be conservative)
    const float idet = __builtin_amdgcn_rcpf(fmaxf(sp.A * sp.C - sp.B * sp.B, 1e-30f));
    const float k2 = 2.f * tau * idet;
    const float hx = __builtin_amdgcn_sqrtf(k2 * sp.C) * 1.000002f + 1e-5f, hy = __builtin_amdgcn_sqrtf(k2 * sp.A) * 1.000002f + 1e-5f;
    const float inv8 = 1.f / (float)kSubTile;
    const int bx0 = (int)ceilf((sp.u - hx - (float)(kSubTile - 1)) * inv8), bx1 = (int)floorf((sp.u + hx) * inv8);
    const int by0 = (int)ceilf((sp.v - hy - (float)(kSubTile - 1)) * inv8), by1 = (int)floorf((sp.v + hy) * inv8);
    const int x0 = max(w.cx0, bx0), x1 = min(w.cx0 + w.cw, bx1 + 1);
    const int y0 = max(w.cy0, by0), y1 = min(w.cy0 + w.ch, by1 + 1);
    if (x1 > x0 && y1 > y0) { w.cx0 = x0; w.cy0 = y0; w.cw = x1 - x0; w.ch = y1 - y0; }
    else { w.cw = 0; w.ch = 0; }
  }
  VTGS_P_STAMP(1)                                                // inputs arrived, projection + walk set up
  const int area_all = w.cw * w.ch;
  const bool big = area_all > (defer ? kBigArea : 64);         // deferred: listed for bin_deferred_splats, not walked here (no-defer form: by the wavefront, below)
  const int area = big ? 0 : area_all;

  // the per-tile table is cleared only now: the input loads above are in flight while it happens, and the workgroups of a
  // band that left above never touch it
  if constexpr (LDSBINS == 1) {
    for (int i = (int)threadIdx.x; i < tiles8; i += kProjBlock) lds_tile[i] = 0u;
    __syncthreads();
  }
  // pass 1: which candidate tiles does this splat really reach (remembered as a bitmask for the first 64)
  uint32_t cnt = 0;
  unsigned long long reach_mask = 0ull;
  for (int i = 0, tx = 0, ty = 0; i < area; ++i) {
    const bool hit = tile_reached(cam, sp, rf, tau, w.cx0 + tx, w.cy0 + ty);
    cnt += hit ? 1u : 0u;
    if (i < 64 && hit) reach_mask |= 1ull << i;
    if constexpr (LDSBINS == 1) { if (hit) atomicAdd(&lds_tile[(w.cy0 + ty) * cam.gx8 + w.cx0 + tx - tile0], 1u); }
    if (++tx == w.cw) { tx = 0; ++ty; }
  }
  VTGS_P_STAMP(2)                                                // table cleared, pass 1 (reach tests + LDS histogram)
  // Instances of one splat are contiguous: reserve [base, base+cnt).  ONE atomic per workgroup on the
  // global counter: same-address atomics serialise at the memory side (~14 ns each measured), so per-wavefront
  // atomics on one cache line cost more than the whole projection.
  __shared__ uint32_t s_wave_cnt[kWaves], s_wave_vis[kWaves], s_wave_r16[kWaves], s_block_base;
  __shared__ uint32_t s_wave_def[kWaves], s_def_base;              // deferred splats of the workgroup, per wavefront; its stretch of the list
  __shared__ uint32_t s_wave_lrg[kWaves], s_lrg_base;              // ... those beyond kGroupArea candidates: the list's other end
  const int wv = (int)(threadIdx.x >> 6);
  // a deferred splat takes its instance ids here too: one per candidate tile, an upper bound (DeferRec::inst_base)
  const uint32_t cnt_ids = (defer && big) ? (uint32_t)area_all : cnt;   // (no-defer form: a big splat takes its ids itself, cnt = 0 here)
  const uint32_t incl = wave_incl_scan(cnt_ids);
  const uint32_t r16 = vis ? (uint32_t)((sp.x1 - sp.x0) * (sp.y1 - sp.y0)) : 0u;
  const uint32_t r16_incl = wave_incl_scan(r16);
  const unsigned long long vb = __ballot(vis);
  const bool large = defer && area_all > kGroupArea;               // (implies big)
  // (no-defer form: def_b counts the splats the deferring form WOULD list -- see the phantom id below)
  const unsigned long long def_b = __ballot(defer ? (big && !large) : (area_all > kBigArea)), lrg_b = __ballot(large);
  if (l == 63) {
    s_wave_cnt[wv] = incl; s_wave_vis[wv] = (uint32_t)__popcll(vb); s_wave_r16[wv] = r16_incl;
    s_wave_def[wv] = (uint32_t)__popcll(def_b); s_wave_lrg[wv] = (uint32_t)__popcll(lrg_b);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0, v = 0, nd = 0, nl = 0; unsigned long long r = 0;
    // (unrolled sixteen times the compiler requests all 80 partials at once and the KERNEL's register count follows: 94 VGPRs,
    //  the second workgroup per CU gone -- with three arrays, until round 5, the same loop just fitted under 64)
#pragma unroll 4
    for (int k = 0; k < kWaves; ++k) { tot += s_wave_cnt[k]; v += s_wave_vis[k]; r += s_wave_r16[k]; nd += s_wave_def[k]; nl += s_wave_lrg[k]; }
    if (!defer) {
      // The caller expected no splat beyond kDeferArea candidates and this workgroup holds one: ONE id that nobody uses, so that
      // the record's instances_needed exceeds its instances -- the only sign the host needs to drop the hint (the ids handed
      // out are an upper bound in the deferring form too)
      tot += nd ? 1u : 0u; nd = 0u;
    }
    // the instance range and the stretch of the deferred list with ONE atomic (Counters: the 64-bit pair)
    const unsigned long long old = (tot | nd) ? atomicAdd(reinterpret_cast<unsigned long long*>(ctr), ((unsigned long long)tot << 32) | (unsigned long long)nd) : 0ull;
    // (the large ones are rare -- none in a fresh view-tied map: their counter is touched only by the workgroups that hold one;
    //  both atomics are in flight together)
    const uint32_t oldl = nl ? atomicAdd(&ctr->defer_large, nl) : 0u;
    s_block_base = (uint32_t)(old >> 32); s_def_base = (uint32_t)old; s_lrg_base = oldl;
    BlockStats bs;
    bs.visible = v; bs.pad = 0; bs.r16 = r;
    block_stats[blockIdx.x] = bs;
  }
  if constexpr (LDSBINS == 1) {
    // one global reservation per tile this workgroup touches; the table entry becomes the running slot index
    for (int i = (int)threadIdx.x; i < tiles8; i += kProjBlock) {
      const uint32_t c = lds_tile[i];
      if (c) lds_tile[i] = atomicAdd(&tile_cnt[tile0 + i], c);
    }
  }
  __syncthreads();
  VTGS_P_STAMP(3)                                                // scans, the instance-range atomic, per-tile global reservations
  uint32_t wave_base = s_block_base;
  for (int k = 0; k < wv; ++k) wave_base += s_wave_cnt[k];
  const uint32_t inst_base = wave_base + incl - cnt_ids;

  if (defer && big) {
    // listed for bin_deferred_splats, with everything it needs (DeferRec).  HERE, between the passes: the splat is still in registers.
    // (up to kGroupArea candidates: from the front of the list; larger: from its end, position n - 1 - index)
    uint32_t pos = (large ? s_lrg_base : s_def_base) + (uint32_t)__popcll((large ? lrg_b : def_b) & ((1ull << l) - 1ull));
    for (int k = 0; k < wv; ++k) pos += large ? s_wave_lrg[k] : s_wave_def[k];
    if (large) pos = (uint32_t)n - 1u - pos;
    float4* __restrict__ rec = reinterpret_cast<float4*>(defer_list + pos);
    rec[0] = make_float4(sp.u, sp.v, sp.A, sp.B);
    rec[1] = make_float4(sp.C, tau, sp.depth, __uint_as_float((uint32_t)gid));
    rec[2] = make_float4(__uint_as_float((uint32_t)w.cx0 | ((uint32_t)w.cy0 << 16)), __int_as_float(w.cw), __int_as_float(w.ch),
                         __uint_as_float(inst_base));
  }

  if (valid && !big) {
    // the geometry record is only ever reached through a tile list: a splat without instances (culled, or outside this
    // call's band of tile rows -- 7/8 of them on each rank of an 8-way partition) does not need one
    // (its geometry record was stored right after the projection)
    gaux[gid] = GaussAux{inst_base, cnt};
  }

  // pass 2: reserve a slot in every reached tile, lanes in lock-step so runs can share atomics.  The returning
  // atomic of step i is consumed in step i+1, so its round trip overlaps the next step's work.
  const unsigned long long key = ((unsigned long long)__float_as_uint(sp.depth) << 32) | (unsigned long long)(uint32_t)gid;
  uint32_t ord = 0;
  if constexpr (LDSBINS == 2) {
    // ---- WINDOWED table (round 5): the frame has more 8x8 tiles than a table that leaves room for two workgroups per CU
    // (> 20 K: 1752x1168 has 32 K), but the 1,024 Gaussians of a workgroup of a raster-ordered map reach a few tile ROWS.  The
    // table covers kWinRows(gx8) rows at a time, starting at the first row any lane of the workgroup reaches: one pass for a
    // view-tied map, ceil(rows / window) passes for a map in any order (each lane takes part with the tiles of that pass only;
    // instance ids follow the walk order of the splat whatever the pass).  Before: run-aggregated global atomics, 404 us at
    // 5 M Gaussians (profiles/r4_shapes.md).
    __shared__ int s_wy0[kWaves], s_wy1[kWaves];
    {
      const int y0 = area > 0 ? w.cy0 : 0x7fffffff, y1 = area > 0 ? w.cy0 + w.ch : 0;
      const int m0 = wave_min_i(y0), m1 = wave_max_i(y1);
      if (l == 0) { s_wy0[wv] = m0; s_wy1[wv] = m1; }
    }
    __syncthreads();
    int wy0 = 0x7fffffff, wy1 = 0;
    for (int k = 0; k < kWaves; ++k) { wy0 = min(wy0, s_wy0[k]); wy1 = max(wy1, s_wy1[k]); }
    const int win_rows = max(1, (int)(kWinEntries / (uint32_t)cam.gx8));
    const int cwd = max(w.cw, 1);
    // row of candidate i without an integer division (two of them per reached tile and loop were ~50 instructions each):
    // i < 64 and cwd <= 64, so (i * ceil(2^16 / cwd)) >> 16 is exact (the error i * (ceil - 2^16 / cwd) < 64 <= 2^16 / cwd); the
    // hardware reciprocal is exact for the powers of two, the only widths whose quotient is an integer
    const uint32_t row_magic = (uint32_t)ceilf(65536.f * __builtin_amdgcn_rcpf((float)cwd));
    auto row_of = [&](int i) { return (int)(((uint32_t)i * row_magic) >> 16); };
    for (int py = wy0; py < wy1; py += win_rows) {                 // (workgroup-uniform bounds)
      const int rows = min(win_rows, wy1 - py), entries = rows * cam.gx8, t0 = py * cam.gx8;
      for (int i = (int)threadIdx.x; i < entries; i += kProjBlock) lds_tile[i] = 0u;
      __syncthreads();
      for (unsigned long long m = reach_mask; m; m &= m - 1ull) {  // histogram of this pass's tiles
        const int i = __builtin_ctzll(m), ry = row_of(i), tty = w.cy0 + ry, ttx = w.cx0 + (i - ry * cwd);
        if (tty >= py && tty < py + rows) atomicAdd(&lds_tile[tty * cam.gx8 + ttx - t0], 1u);
      }
      __syncthreads();
      for (int i = (int)threadIdx.x; i < entries; i += kProjBlock) {   // one global reservation per touched tile
        const uint32_t c = lds_tile[i];
        if (c) lds_tile[i] = atomicAdd(&tile_cnt[t0 + i], c);
      }
      __syncthreads();
      for (unsigned long long m = reach_mask; m; m &= m - 1ull) {
        const int i = __builtin_ctzll(m), ry = row_of(i), tty = w.cy0 + ry, ttx = w.cx0 + (i - ry * cwd);
        if (tty >= py && tty < py + rows) {
          const int tile = tty * cam.gx8 + ttx;
          const uint32_t slot = atomicAdd(&lds_tile[tile - t0], 1u);
          const unsigned long long id = (unsigned long long)inst_base + (uint32_t)__builtin_popcountll(reach_mask & ((1ull << i) - 1ull));
          const BinRange br = planned ? bin_range(cs, (uint32_t)tile, tile_cap) : BinRange{(uint32_t)tile * tile_cap, tile_cap};
          if (id < capacity && slot < br.cap) {
            const size_t pos = (size_t)br.s + slot;
            keys[pos] = key;
            vals[pos] = (uint32_t)id;
          }
        }
      }
      if (py + win_rows < wy1) __syncthreads();                    // the next pass clears the table
    }
  } else if constexpr (LDSBINS == 1) {
    // pass 2, LDS form: every lane walks its own reached tiles; the slot comes from the LDS table
    for (int i = 0, tx = 0, ty = 0; i < area; ++i) {
      const int ttx = w.cx0 + tx, tty = w.cy0 + ty;
      const bool hit = (i < 64) ? ((reach_mask >> i) & 1ull) != 0ull : tile_reached(cam, sp, rf, tau, ttx, tty);
      if (hit) {
        const int tile = tty * cam.gx8 + ttx;
        const uint32_t slot = atomicAdd(&lds_tile[tile - tile0], 1u);
        const unsigned long long id = (unsigned long long)inst_base + ord;
        const BinRange br = planned ? bin_range(cs, (uint32_t)tile, tile_cap) : BinRange{(uint32_t)tile * tile_cap, tile_cap};
        if (id < capacity && slot < br.cap) {                    // an overflowing bin / id is dropped and flagged later
          const size_t pos = (size_t)br.s + slot;
          keys[pos] = key;
          vals[pos] = (uint32_t)id;
        }
        ++ord;
      }
      if (++tx == w.cw) { tx = 0; ++ty; }
    }
  } else {
  Reservation pend;
  pend.base = 0; pend.head_lane = 0; pend.rank = 0; pend.act = false; pend.tile = -1;
  auto consume = [&](const Reservation& r) {                  // all 64 lanes call it
    const uint32_t slot = reserve_resolve(r);
    if (r.act) {
      const unsigned long long id = (unsigned long long)inst_base + ord;
      const BinRange br = planned ? bin_range(cs, (uint32_t)r.tile, tile_cap) : BinRange{(uint32_t)r.tile * tile_cap, tile_cap};
      if (id < capacity && slot < br.cap) {                    // an overflowing bin / id is dropped and flagged later
        const size_t pos = (size_t)br.s + slot;
        keys[pos] = key;
        vals[pos] = (uint32_t)id;
      }
      ++ord;
    }
  };
  const int max_area = wave_max_i(area);
  if (max_area <= 64) {
    // common case: step k handles every lane's k-th REACHED tile (k-th set bit of its mask).  Raster-ordered
    // neighbours have near-identical masks, so runs still form, and there are ~3 steps instead of ~16.
    const int steps = wave_max_i((int)cnt);
    unsigned long long m = reach_mask;
    for (int k = 0; k <= steps; ++k) {
      Reservation cur;
      cur.base = 0; cur.head_lane = 0; cur.rank = 0; cur.act = false; cur.tile = -1;
      if (k < steps) {                                       // wave-uniform
        const bool act = m != 0ull;
        const int i = act ? __builtin_ctzll(m) : 0;
        m &= m - 1ull;
        const int cwd = max(w.cw, 1);
        const int tty = w.cy0 + i / cwd, ttx = w.cx0 + (i - (i / cwd) * cwd);
        cur = reserve_issue(tile_cnt, act ? (tty * cam.gx8 + ttx) : -1, act);
      }
      consume(pend);
      pend = cur;
    }
  } else {
    // a splat of this wavefront covers more than 64 candidate tiles: walk all candidates in lock-step
    for (int i = 0, tx = 0, ty = 0; i <= max_area; ++i) {
      Reservation cur;
      cur.base = 0; cur.head_lane = 0; cur.rank = 0; cur.act = false; cur.tile = -1;
      if (i < max_area) {                                    // wave-uniform
        const bool in = i < area;
        const int ttx = w.cx0 + tx, tty = w.cy0 + ty;
        const bool act = in && ((i < 64) ? ((reach_mask >> i) & 1ull) != 0ull : tile_reached(cam, sp, rf, tau, ttx, tty));
        cur = reserve_issue(tile_cnt, act ? (tty * cam.gx8 + ttx) : -1, act);
        if (in && ++tx == w.cw) { tx = 0; ++ty; }
      }
      consume(pend);
      pend = cur;
    }
  }
  }  // (global-atomic form)

  VTGS_P_STAMP(4)                                                // records + pass 2 (slots, bin entries stored)
#ifdef VTGS_Q_STAMPS
  if (threadIdx.x == 0 && cs.dbg_proj) {
    uint32_t* o = cs.dbg_proj + 8 * blockIdx.x;
    o[0] = (uint32_t)(pst[1] - pst[0]); o[1] = (uint32_t)(pst[2] - pst[1]); o[2] = (uint32_t)(pst[3] - pst[2]);
    o[3] = (uint32_t)(pst[4] - pst[3]); o[4] = (uint32_t)(pst[4] - pst[0]);
    o[5] = (uint32_t)prt0; o[6] = (uint32_t)__builtin_amdgcn_s_memrealtime(); o[7] = (uint32_t)area_all;
  }
#endif
  if (!defer) {
    // ---- the big splats of this wavefront, one after the other, every lane a tile -------------------------------------------
    for (unsigned long long rest = __ballot(big); rest; rest &= rest - 1ull) {      // wave-uniform
      const int src = __builtin_ctzll(rest);
      const BigWalk b = broadcast_walk(w, sp, rf, tau, area_all, src);
      uint32_t total = 0;
      for (int i0 = 0; i0 < b.area; i0 += 64) {
        const int i = i0 + l;
        const int ty = i / b.w.cw, tx = i - ty * b.w.cw;
        const bool hit = i < b.area && tile_reached(cam, b.sp, b.rf, b.tau, b.w.cx0 + tx, b.w.cy0 + ty);
        total += (uint32_t)__builtin_popcountll(__ballot(hit));
      }
      unsigned long long got = 0ull;                               // (the 64-bit pair of Counters: the instance total is its high word)
      if (l == src && total) got = atomicAdd(reinterpret_cast<unsigned long long*>(ctr), (unsigned long long)total << 32);
      const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(got >> 32), src);
      const unsigned long long key_src = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(key >> 32), src) << 32) |
                                         (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, src);
      uint32_t done = 0;
      for (int i0 = 0; i0 < b.area; i0 += 64) {
        const int i = i0 + l;
        const int ty = i / b.w.cw, tx = i - ty * b.w.cw;
        const bool hit = i < b.area && tile_reached(cam, b.sp, b.rf, b.tau, b.w.cx0 + tx, b.w.cy0 + ty);
        const unsigned long long hb = __ballot(hit);
        if (hit) {
          const int tile = (b.w.cy0 + ty) * cam.gx8 + b.w.cx0 + tx;
          const uint32_t slot = atomicAdd(&tile_cnt[tile], 1u);
          const uint32_t rank = (uint32_t)__builtin_popcountll(hb & ((1ull << l) - 1ull));
          const unsigned long long id = (unsigned long long)base + done + rank;            // raster order of the walk
          const BinRange br = planned ? bin_range(cs, (uint32_t)tile, tile_cap) : BinRange{(uint32_t)tile * tile_cap, tile_cap};
          if (id < capacity && slot < br.cap) {
            const size_t pos = (size_t)br.s + slot;
            keys[pos] = key_src;
            vals[pos] = (uint32_t)id;
          }
        }
        done += (uint32_t)__builtin_popcountll(hb);
      }
      if (l == src) gaux[gid] = GaussAux{base, total};
    }
  }
}


// The kernels proper.  MODE 0 (whole frame, uniform bins) compiles to 79 scalar registers by itself; the other modes need 82-90,
// which costs the second workgroup per CU (13 us of 54 at the headline shape, profiles/r3_project_ab.txt): they are capped at
// 80, a few scalars then live in vector-register lanes.  (The cap on MODE 0 would push it to 65 vector registers -- the same cliff
// from the other side -- hence two definitions.)
template <int LDSBINS, int MODE>
__global__ __launch_bounds__(kProjBlock) void project_and_bin(
    CamScalars cs, const float* __restrict__ Vp, const float* __restrict__ PVp, int n,
    const float* __restrict__ means3D, const float* __restrict__ opacities,
    const float* __restrict__ scales, const float* __restrict__ rotations,
    int32_t* __restrict__ radii, GeomRec* __restrict__ geom, GaussAux* __restrict__ gaux,
    uint32_t* __restrict__ tile_cnt, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals,
    Counters* __restrict__ ctr, BlockStats* __restrict__ block_stats, unsigned long long capacity, uint32_t tile_cap,
    DeferRec* __restrict__ defer_list) {
  project_and_bin_body<LDSBINS, MODE>(cs, Vp, PVp, n, means3D, opacities, scales, rotations, radii, geom, gaux, tile_cnt, keys, vals, ctr, block_stats, capacity, tile_cap, defer_list);
}
template <int LDSBINS, int MODE>
__global__ __launch_bounds__(kProjBlock) __attribute__((amdgpu_num_sgpr(80))) void project_and_bin_capped(
    CamScalars cs, const float* __restrict__ Vp, const float* __restrict__ PVp, int n,
    const float* __restrict__ means3D, const float* __restrict__ opacities,
    const float* __restrict__ scales, const float* __restrict__ rotations,
    int32_t* __restrict__ radii, GeomRec* __restrict__ geom, GaussAux* __restrict__ gaux,
    uint32_t* __restrict__ tile_cnt, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals,
    Counters* __restrict__ ctr, BlockStats* __restrict__ block_stats, unsigned long long capacity, uint32_t tile_cap,
    DeferRec* __restrict__ defer_list) {
  project_and_bin_body<LDSBINS, MODE>(cs, Vp, PVp, n, means3D, opacities, scales, rotations, radii, geom, gaux, tile_cnt, keys, vals, ctr, block_stats, capacity, tile_cap, defer_list);
}
template __global__ void project_and_bin<0, 0>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin<1, 0>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin<2, 0>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<2, 1>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<2, 2>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<2, 3>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<0, 1>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<0, 2>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<0, 3>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<1, 1>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<1, 2>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<1, 3>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<0, 4>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<0, 5>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<1, 4>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template __global__ void project_and_bin_capped<1, 5>(CamScalars, const float*, const float*, int, const float*, const float*, const float*, const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);

// The deferred splats of a forward (see kDeferArea).  project_and_bin leaves TWO lists in one array: splats of up to kGroupArea
// candidate tiles from the front (Counters::defer_total), larger ones from the end (Counters::defer_large) -- one list for the
// whole forward each, so the work spreads evenly however the large splats cluster in the map (with one list per projection
// workgroup the densified end of a SLAM map kept a few workgroups busy for 98 us, gpurun_out/r6/slamlate_j_dens.txt).  Every
// entry brings its instance range along (DeferRec::inst_base), so there is no counter, no scan and no barrier between entries.
// The first `large_grid` workgroups take the large list (they are dispatched first: theirs are the long chains), the others
// the small one -- until run r of round 6 one workgroup did both for its 64 entries, one tier after the other behind a
// barrier: 55 us for the 150 K entries of the heavy-tailed test map, of which the waiting for atomics is 19.
//   * small (<= kGroupArea candidates): a 16-lane group forms the reach mask, writes (first id, count) into gaux and bins the
//     mask's bits -- the slot atomics of ALL the group's four entries of the round go out before anything is stored: the round
//     trip of those atomics (~1.5 us) is what bounds this part, and with one waited for after the other the same work took 105 us
//     (gpurun_out/r6/timing_k_tail.log);
//   * large, up to kWaveArea: a wavefront walks the candidates 64 at a time, keeps the hit ballots, then bins with four steps'
//     atomics in flight;
//   * beyond (a splat grown over a hole of the map: thousands of tiles): the whole workgroup, 256 candidates per step; the
//     per-step, per-wavefront counts go through LDS once, then the steps are binned four at a time without a barrier.
// Everything a splat needs arrives in its 48-byte DeferRec: no dependent loads.
constexpr int kDeferChunk = 64;                  // small-list entries per workgroup and round: four per 16-lane group
constexpr int kWaveSteps = kWaveArea / 64;
constexpr int kMaxHugeSteps = 512;               // 131,072 candidate tiles (a 2896 x 2896-pixel splat on an 8-pixel grid) per huge splat and pass
// diagnostic builds only (-DVTGS_EXP_DEFER=1: no slot atomics, 2: no bin stores): where this kernel's time goes
#ifndef VTGS_EXP_DEFER
#define VTGS_EXP_DEFER 0
#endif
#define DEFER_SLOT(p) ((VTGS_EXP_DEFER & 1) ? 0u : atomicAdd((p), 1u))
template <bool PLANNED>
__global__ __launch_bounds__(kDeferBlock) void bin_deferred_splats(
    CamScalars cs, GaussAux* __restrict__ gaux, uint32_t* __restrict__ tile_cnt,
    unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals, Counters* __restrict__ ctr,
    const DeferRec* __restrict__ list, uint32_t n, unsigned long long capacity, uint32_t tile_cap, uint32_t large_grid) {
  constexpr int kWaves = kDeferBlock / 64, kGroups = kDeferBlock / 16, kPerGroup = kDeferChunk / kGroups, kSteps = kGroupArea / 16;
  static_assert(kPerGroup * kGroups == kDeferChunk, "whole groups");
  constexpr bool planned = PLANNED;
  const bool large_role = blockIdx.x < large_grid;                // workgroup-uniform
  const uint32_t total = large_role ? ctr->defer_large : ctr->defer_total;   // (complete: project_and_bin has retired)
  const uint32_t blk = large_role ? blockIdx.x : blockIdx.x - large_grid, nblk = large_role ? large_grid : gridDim.x - large_grid;
  if (blk * (uint32_t)(large_role ? kWaves : kDeferChunk) >= total) return;  // workgroup-uniform: nothing (left) for this workgroup
  // camera scalars the walk needs (no matrices: the projection is done)
  CamParams cam{};
  cam.W = cs.W; cam.H = cs.H;
  cam.gx8 = (cs.W + kSubTile - 1) / kSubTile; cam.gy8 = (cs.H + kSubTile - 1) / kSubTile;
  const int l = lane_id(), wv = (int)(threadIdx.x >> 6);
  struct Rec { float4 a, b, c; };
  auto splat_of = [&](const Rec& r, Splat& sp, ReachForm& rf) {
    sp = Splat{};
    sp.u = r.a.x; sp.v = r.a.y; sp.A = r.a.z; sp.B = r.a.w; sp.C = r.b.x;
    rf = make_reach_form(sp);
  };
  auto put = [&](int tile, uint32_t slot, unsigned long long key, unsigned long long id) {
    const BinRange br = planned ? bin_range(cs, (uint32_t)tile, tile_cap) : BinRange{(uint32_t)tile * tile_cap, tile_cap};
    if ((VTGS_EXP_DEFER & 2) == 0 && id < capacity && slot < br.cap) {   // an overflowing bin / id is dropped and flagged later
      const size_t pos = (size_t)br.s + slot;
      keys[pos] = key;
      vals[pos] = (uint32_t)id;
    }
  };
  if (!large_role) {
    // ---- the small list: no LDS, no barrier
    const int grp = (int)(threadIdx.x >> 4), sub = l & 15, gsh = l & 48;  // 16-lane group of the workgroup, lane in it, its bit offset
    for (uint32_t c0 = blk * (uint32_t)kDeferChunk; c0 < total; c0 += nblk * (uint32_t)kDeferChunk) {   // workgroup-uniform
      const uint32_t ne = min((uint32_t)kDeferChunk, total - c0);
      Rec rs[kPerGroup];
      unsigned long long ms[kPerGroup];
#pragma unroll
      for (int j = 0; j < kPerGroup; ++j) {                        // all records requested first
        const uint32_t e = (uint32_t)(grp + kGroups * j);
        const float4* p = reinterpret_cast<const float4*>(list + c0 + (e < ne ? e : 0u));
        rs[j] = Rec{p[0], p[1], p[2]};
      }
#pragma unroll
      for (int j = 0; j < kPerGroup; ++j) {
        const uint32_t e = (uint32_t)(grp + kGroups * j);
        ms[j] = 0ull;
        if (e >= ne) continue;                                     // (group-uniform)
        const Rec& r = rs[j];
        const uint32_t cxy = __float_as_uint(r.c.x);
        const int cx0 = (int)(cxy & 0xFFFFu), cy0 = (int)(cxy >> 16), cw = __float_as_int(r.c.y), ch = __float_as_int(r.c.z);
        const int area = min(cw * ch, kGroupArea);                 // (<= kGroupArea by construction of the list)
        Splat sp; ReachForm rf;
        splat_of(r, sp, rf);
        unsigned long long m = 0ull;
        for (int i0 = 0; i0 < area; i0 += 16) {                    // at most four steps
          const int i = i0 + sub;
          const int ty = i / cw, tx = i - ty * cw;
          const bool hit = i < area && tile_reached(cam, sp, rf, r.b.y, cx0 + tx, cy0 + ty);
          m |= ((__ballot(hit) >> gsh) & 0xFFFFull) << i0;
        }
        ms[j] = m;
        if (sub == 0) gaux[__float_as_uint(r.b.w)] = GaussAux{__float_as_uint(r.c.w), (uint32_t)__popcll(m)};
      }
      uint32_t slot[kPerGroup][kSteps];
#pragma unroll
      for (int j = 0; j < kPerGroup; ++j) {                        // all slot atomics of the round
        const uint32_t cxy = __float_as_uint(rs[j].c.x);
        const int cx0 = (int)(cxy & 0xFFFFu), cy0 = (int)(cxy >> 16), cw = __float_as_int(rs[j].c.y);
#pragma unroll
        for (int k = 0; k < kSteps; ++k) {
          const int i = 16 * k + sub;
          slot[j][k] = 0u;
          if ((ms[j] >> i) & 1ull) {
            const int ty = i / cw, tx = i - ty * cw;
            slot[j][k] = DEFER_SLOT(&tile_cnt[(cy0 + ty) * cam.gx8 + cx0 + tx]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < kPerGroup; ++j) {                        // the bin entries
        const unsigned long long m = ms[j];
        if (m == 0ull) continue;
        const uint32_t cxy = __float_as_uint(rs[j].c.x);
        const int cx0 = (int)(cxy & 0xFFFFu), cy0 = (int)(cxy >> 16), cw = __float_as_int(rs[j].c.y);
        const uint32_t base = __float_as_uint(rs[j].c.w);
        const unsigned long long key = ((unsigned long long)__float_as_uint(rs[j].b.z) << 32) | (unsigned long long)__float_as_uint(rs[j].b.w);
#pragma unroll
        for (int k = 0; k < kSteps; ++k) {
          const int i = 16 * k + sub;
          if ((m >> i) & 1ull) {
            const int ty = i / cw, tx = i - ty * cw;
            put((cy0 + ty) * cam.gx8 + cx0 + tx, slot[j][k], key, (unsigned long long)base + (uint32_t)__popcll(m & ((1ull << i) - 1ull)));   // raster order of the walk
          }
        }
      }
    }
    return;
  }
  // ---- the large list (entry e sits at list[n - 1 - e]): one entry per wavefront and round; what is beyond kWaveArea waits in
  // s_huge for the whole workgroup
  __shared__ uint32_t s_huge[kWaves], s_nhuge;
  __shared__ uint32_t s_cnt[kMaxHugeSteps][kWaves];                // huge splats: hits per step and wavefront
  auto load_large = [&](uint32_t e) { const float4* p = reinterpret_cast<const float4*>(list + (n - 1u - e)); return Rec{p[0], p[1], p[2]}; };
  for (uint32_t c0 = blk * (uint32_t)kWaves; c0 < total; c0 += nblk * (uint32_t)kWaves) {               // workgroup-uniform
    if (threadIdx.x == 0) s_nhuge = 0u;
    __syncthreads();
    const uint32_t e = c0 + (uint32_t)wv;
    if (e < total) {                                                                                  // wave-uniform
      const Rec r = load_large(e);
      const uint32_t cxy = __float_as_uint(r.c.x);
      const int cx0 = (int)(cxy & 0xFFFFu), cy0 = (int)(cxy >> 16), cw = __float_as_int(r.c.y), ch = __float_as_int(r.c.z);
      const int area = cw * ch;
      if (area > kWaveArea) {
        if (l == 0) s_huge[atomicAdd(&s_nhuge, 1u)] = e;
      } else {
        Splat sp; ReachForm rf;
        splat_of(r, sp, rf);
        const uint32_t base = __float_as_uint(r.c.w);
        const unsigned long long key = ((unsigned long long)__float_as_uint(r.b.z) << 32) | (unsigned long long)__float_as_uint(r.b.w);
        unsigned long long hb[kWaveSteps];                        // the hit ballots of all steps (wave-uniform: scalar registers)
        uint32_t cnt = 0;
#pragma unroll
        for (int st = 0; st < kWaveSteps; ++st) {
          hb[st] = 0ull;
          if (64 * st < area) {                                    // wave-uniform
            const int i = 64 * st + l;
            const int ty = i / cw, tx = i - ty * cw;
            hb[st] = __ballot(i < area && tile_reached(cam, sp, rf, r.b.y, cx0 + tx, cy0 + ty));
            cnt += (uint32_t)__popcll(hb[st]);
          }
        }
        if (l == 0) gaux[__float_as_uint(r.b.w)] = GaussAux{base, cnt};
        uint32_t done = 0;
#pragma unroll
        for (int s4 = 0; s4 < kWaveSteps; s4 += 4) {
          if (64 * s4 >= area) break;                              // wave-uniform
          uint32_t slot4[4]; int tile4[4]; uint32_t rank4[4];
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const unsigned long long b = hb[s4 + h];
            const int i = 64 * (s4 + h) + l;
            const int ty = i / cw, tx = i - ty * cw;
            tile4[h] = (cy0 + ty) * cam.gx8 + cx0 + tx;
            rank4[h] = done + (uint32_t)__popcll(b & ((1ull << l) - 1ull));
            done += (uint32_t)__popcll(b);
            slot4[h] = ((b >> l) & 1ull) ? DEFER_SLOT(&tile_cnt[tile4[h]]) : 0u;
          }
#pragma unroll
          for (int h = 0; h < 4; ++h)
            if ((hb[s4 + h] >> l) & 1ull) put(tile4[h], slot4[h], key, (unsigned long long)base + rank4[h]);
        }
      }
    }
    __syncthreads();
    const uint32_t nh = s_nhuge;
    for (uint32_t k = 0; k < nh; ++k) {                                                                // workgroup-uniform
      const Rec r = load_large(s_huge[k]);
      const uint32_t cxy = __float_as_uint(r.c.x);
      const int cx0 = (int)(cxy & 0xFFFFu), cy0 = (int)(cxy >> 16), cw = __float_as_int(r.c.y), ch = __float_as_int(r.c.z);
      const int area = cw * ch;
      Splat sp; ReachForm rf;
      splat_of(r, sp, rf);
      const uint32_t base = __float_as_uint(r.c.w);
      const unsigned long long key = ((unsigned long long)__float_as_uint(r.b.z) << 32) | (unsigned long long)__float_as_uint(r.b.w);
      uint32_t done = 0;                                           // ids handed out so far
      for (int p0 = 0; p0 < area; p0 += kMaxHugeSteps * kDeferBlock) {                                 // (one pass unless the splat covers > 131,072 tiles)
        const int pend = min(area, p0 + kMaxHugeSteps * kDeferBlock);
        __syncthreads();                                           // (the previous user of s_cnt is through)
        for (int i0 = p0, st = 0; i0 < pend; i0 += kDeferBlock, ++st) {
          const int i = i0 + (int)threadIdx.x;
          const int ty = i / cw, tx = i - ty * cw;
          const bool hit = i < pend && tile_reached(cam, sp, rf, r.b.y, cx0 + tx, cy0 + ty);
          const uint32_t c = (uint32_t)__popcll(__ballot(hit));
          if (l == 0) s_cnt[st][wv] = c;
        }
        __syncthreads();
        for (int i0 = p0, st = 0; i0 < pend; i0 += 4 * kDeferBlock, st += 4) {                         // four steps' atomics in flight
          uint32_t slot4[4]; int tile4[4]; uint32_t rank4[4]; bool hit4[4];
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const int i = i0 + h * kDeferBlock + (int)threadIdx.x;
            const int ty = i / cw, tx = i - ty * cw;
            hit4[h] = i < pend && tile_reached(cam, sp, rf, r.b.y, cx0 + tx, cy0 + ty);
            const unsigned long long b = __ballot(hit4[h]);
            uint32_t before = 0, step_total = 0;
            if (i0 + h * kDeferBlock < pend) {                     // workgroup-uniform (s_cnt holds this pass's steps only)
              for (int kk = 0; kk < kWaves; ++kk) { const uint32_t c = s_cnt[st + h][kk]; before += kk < wv ? c : 0u; step_total += c; }
            }
            tile4[h] = (cy0 + ty) * cam.gx8 + cx0 + tx;
            rank4[h] = done + before + (uint32_t)__popcll(b & ((1ull << l) - 1ull));
            done += step_total;
            slot4[h] = hit4[h] ? DEFER_SLOT(&tile_cnt[tile4[h]]) : 0u;
          }
#pragma unroll
          for (int h = 0; h < 4; ++h)
            if (hit4[h]) put(tile4[h], slot4[h], key, (unsigned long long)base + rank4[h]);
        }
      }
      if (threadIdx.x == 0) gaux[__float_as_uint(r.b.w)] = GaussAux{base, done};
    }
    // (the next round's first barrier separates this round's readers of s_huge / s_cnt from its writers)
  }
}
template __global__ void bin_deferred_splats<false>(CamScalars, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, const DeferRec*, uint32_t, unsigned long long, uint32_t, uint32_t);
template __global__ void bin_deferred_splats<true>(CamScalars, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, const DeferRec*, uint32_t, unsigned long long, uint32_t, uint32_t);

// One workgroup right after the binning: longest tile list, statistics, overflow flags and the image of the
// host-visible VtgsForwardInfo (finalize_block, vtgs_internal.h; the quadrant-queue forward runs it in its first workgroup
// instead when it also sorts the lists -- then nothing is launched between the binning and the composite).
__global__ __launch_bounds__(1024) void finalize_forward(const uint32_t* __restrict__ tile_cnt, uint32_t tiles,
                                                         Counters* __restrict__ ctr, unsigned long long capacity,
                                                         uint32_t tile_cap, const BlockStats* __restrict__ block_stats,
                                                         uint32_t nblocks, VtgsForwardInfo* host_record,
                                                         const uint32_t* __restrict__ plan, uint32_t* __restrict__ plan_next) {
  finalize_block<1024>(tile_cnt, tiles, ctr, capacity, tile_cap, block_stats, nblocks, host_record, plan, plan_next);
}

__device__ __forceinline__ void order_pair(unsigned long long* k, uint32_t* v, uint32_t i, uint32_t p) {
  const unsigned long long a = k[i], b = k[p];
  if (a > b) {
    k[i] = b; k[p] = a;
    const uint32_t va = v[i], vb = v[p];
    v[i] = vb; v[p] = va;
  }
}

__device__ __forceinline__ void sort_network(unsigned long long* k, uint32_t* v, uint32_t L, uint32_t n2, uint32_t t) {
  for (uint32_t k2 = 2; k2 <= n2; k2 <<= 1) {
    const uint32_t half = k2 >> 1;
    for (uint32_t q = t; q < (n2 >> 1); q += 256u) {                 // mirror step
      const uint32_t blk = q / half, off = q - blk * half;
      const uint32_t i = blk * k2 + off, p = blk * k2 + (k2 - 1u - off);
      if (p < L) order_pair(k, v, i, p);
    }
    __syncthreads();
    for (uint32_t j = half >> 1; j > 0; j >>= 1) {                    // half-cleaners
      for (uint32_t q = t; q < (n2 >> 1); q += 256u) {
        const uint32_t i = 2u * q - (q & (j - 1u)), p = i + j;
        if (p < L) order_pair(k, v, i, p);
      }
      __syncthreads();
    }
  }
}

// grid = ceil(tiles/4) workgroups of 4 wavefronts; wavefront w of workgroup b owns tile tile_first + 4*b' + w (b' XCD-swizzled)
// (tile_first, tiles) = the call's band of tiles: the grid covers only those, so that a band (multi-GPU partition) still
// spreads over all eight XCDs instead of landing in the one XCD whose share of a full-frame grid it would be.
// WIDE adds the 32-keys-per-lane register form for packed lists of 1025..2048 entries (dense maps: 2 M Gaussians at 640x480
// average ~1,300 per tile; one wavefront per list instead of the whole workgroup going through them one at a time: 192 ->
// ~40 us there).  It costs 125 instead of 80 VGPRs, so the host launches it only when the bin capacity allows such lists.
template <bool WIDE>
__global__ __launch_bounds__(256) void sort_tiles(const uint32_t* __restrict__ tile_cnt, unsigned long long* __restrict__ keys,
                                                  uint32_t* __restrict__ vals, uint32_t* __restrict__ sorted_gid,
                                                  uint32_t* __restrict__ sorted_inst, uint32_t tile_first, uint32_t tiles,
                                                  uint32_t tile_cap, const Counters* __restrict__ ctr, int packed,
                                                  const uint32_t* __restrict__ plan, uint32_t bin_limit,
                                                  unsigned long long long_only_capacity, int mid_done) {
  // long_only_capacity != 0 (planned bins with the sort fused into the forward composite): only the lists that one wavefront
  // of the composite cannot sort (> 1,024 entries) are sorted here, ahead of it.  finalize has not run yet in that mode, so
  // the kernel makes the composite's own safety test: nothing if the INSTANCE capacity overflowed (an entry may have been
  // dropped after its slot was taken), and no list whose bin overflowed (the composite bails on those).
  __shared__ unsigned long long sk[kSortLds];
  __shared__ uint32_t sv[kSortLds];
  const bool long_only = long_only_capacity != 0ull;
  if (long_only ? ((unsigned long long)ctr->inst_total > long_only_capacity) : (ctr->overflow != 0u)) return;
  const uint32_t nblk = (tiles + 3u) >> 2;
  const uint32_t b = xcd_swizzle(blockIdx.x, nblk);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  {
    const uint32_t tile = tile_first + 4u * b + (uint32_t)wv;
    if (4u * b + (uint32_t)wv < tiles) {
      const BinRange br = bin_range(plan, bin_limit, tile, tile_cap);
      const size_t s = (size_t)br.s;
      const uint32_t cnt = tile_cnt[tile];
      uint32_t L = (long_only && (cnt > br.cap || cnt <= (uint32_t)kWaveSortMax)) ? 0u : min(cnt, br.cap);
      if (mid_done && L > (uint32_t)kCountSortMax && L <= (uint32_t)kBlockSortMax) L = 0u;   // sorted by sort_long_lists
      if (L == 1u) {
        if (lane == 0) { sorted_gid[s] = (uint32_t)keys[s]; sorted_inst[s] = vals[s]; }
      } else if (packed) {
        if (L <= 64u) { if (L) wave_sort_tile<1, true>(keys, vals, sorted_gid, sorted_inst, s, L, lane); }
        else if (L <= 128u) wave_sort_tile<2, true>(keys, vals, sorted_gid, sorted_inst, s, L, lane);
        else if (L <= 256u) wave_sort_tile<4, true>(keys, vals, sorted_gid, sorted_inst, s, L, lane);
        else if (L <= 512u) wave_sort_tile<8, true>(keys, vals, sorted_gid, sorted_inst, s, L, lane);
        else if (L <= (uint32_t)kWaveSortMax) wave_sort_tile<16, true>(keys, vals, sorted_gid, sorted_inst, s, L, lane);
        else if (WIDE && L <= (uint32_t)kWaveSortMaxPacked) { if constexpr (WIDE) wave_sort_tile<32, true>(keys, vals, sorted_gid, sorted_inst, s, L, lane); }
      } else {
        if (L <= 64u) { if (L) wave_sort_tile<1, false>(keys, vals, sorted_gid, sorted_inst, s, L, lane); }
        else if (L <= 128u) wave_sort_tile<2, false>(keys, vals, sorted_gid, sorted_inst, s, L, lane);
        else if (L <= 256u) wave_sort_tile<4, false>(keys, vals, sorted_gid, sorted_inst, s, L, lane);
        else if (L <= 512u) wave_sort_tile<8, false>(keys, vals, sorted_gid, sorted_inst, s, L, lane);
        else if (L <= (uint32_t)kWaveSortMax) wave_sort_tile<16, false>(keys, vals, sorted_gid, sorted_inst, s, L, lane);
      }
    }
  }
  // long lists: the whole workgroup takes them one at a time (workgroup-uniform control flow)
  const uint32_t t = threadIdx.x;
  for (uint32_t q = 0; q < 4u; ++q) {
    const uint32_t tile = tile_first + 4u * b + q;
    if (4u * b + q >= tiles) break;
    const BinRange br = bin_range(plan, bin_limit, tile, tile_cap);
    const size_t s = (size_t)br.s;
    const uint32_t cnt = tile_cnt[tile];
    const uint32_t L = (long_only && cnt > br.cap) ? 0u : min(cnt, br.cap);
    if (L <= (uint32_t)((WIDE && packed) ? kWaveSortMaxPacked : kWaveSortMax) || (mid_done && L <= (uint32_t)kBlockSortMax)) continue;
    uint32_t n2 = 1;
    while (n2 < L) n2 <<= 1;
    __syncthreads();
    if (L <= (uint32_t)kSortLds) {
      for (uint32_t i = t; i < L; i += 256u) { sk[i] = keys[s + i]; sv[i] = vals[s + i]; }
      __syncthreads();
      sort_network(sk, sv, L, n2, t);
      for (uint32_t i = t; i < L; i += 256u) { sorted_gid[s + i] = (uint32_t)sk[i]; sorted_inst[s + i] = sv[i]; }
    } else {
      sort_network(keys + s, vals + s, L, n2, t);   // __syncthreads() orders the workgroup's global accesses
      for (uint32_t i = t; i < L; i += 256u) { sorted_gid[s + i] = (uint32_t)keys[s + i]; sorted_inst[s + i] = vals[s + i]; }
    }
  }
}

// Lists of 513 .. 2,048 entries (1,025 .. 2,048 ahead of a forward that sorts the rest itself), ONE list per workgroup: the
// workgroup counting sort of vtgs_sort_common.h, or -- when it declines a list (a bucket of more than 12 near-equal depths) --
// the LDS network.  sort_tiles (mid_done = 1) leaves those lists alone.  Dense maps (2 M Gaussians at 640x480, ~1,300 per
// tile): the 32-keys-per-lane register network took 113 us for them, this takes profiles/r3_long_list_sort.txt.
__global__ __launch_bounds__(256) void sort_long_lists(const uint32_t* __restrict__ tile_cnt, unsigned long long* __restrict__ keys,
                                                       uint32_t* __restrict__ vals, uint32_t* __restrict__ sorted_gid,
                                                       uint32_t* __restrict__ sorted_inst, uint32_t tile_first, uint32_t tiles,
                                                       uint32_t tile_cap, const Counters* __restrict__ ctr,
                                                       const uint32_t* __restrict__ plan, uint32_t bin_limit,
                                                       unsigned long long long_only_capacity, int counting, uint32_t lo) {
  __shared__ unsigned long long sk[kSortLds];
  __shared__ uint32_t sv[kSortLds];
  static_assert(kSortLds >= kBlockSortMax && kSortLds >= kBlockSortBuckets, "LDS staging of the workgroup counting sort");
  const bool long_only = long_only_capacity != 0ull;
  if (long_only ? ((unsigned long long)ctr->inst_total > long_only_capacity) : (ctr->overflow != 0u)) return;
  const uint32_t tile = tile_first + xcd_swizzle(blockIdx.x, tiles);
  const BinRange br = bin_range(plan, bin_limit, tile, tile_cap);
  const uint32_t cnt = tile_cnt[tile];
  if (cnt > br.cap) return;                                     // an overflowing bin: every consumer bails on the flag
  const uint32_t L = cnt;                                       // lo = 512: the lists the wavefront-level counting sort does not take
  if (L <= lo || L > (uint32_t)kBlockSortMax) return;           // workgroup-uniform
  const size_t s = (size_t)br.s;
  const uint32_t t = threadIdx.x;
  if (counting && block_count_sort(keys, vals, sorted_gid, sorted_inst, s, L, t, sk, sv)) return;   // (counting = 0: cross-check switch)
  uint32_t n2 = 1;
  while (n2 < L) n2 <<= 1;
  for (uint32_t i = t; i < L; i += 256u) { sk[i] = keys[s + i]; sv[i] = vals[s + i]; }
  __syncthreads();
  sort_network(sk, sv, L, n2, t);
  for (uint32_t i = t; i < L; i += 256u) { sorted_gid[s + i] = (uint32_t)sk[i]; sorted_inst[s + i] = sv[i]; }
}

template __global__ void sort_tiles<false>(const uint32_t*, unsigned long long*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t, uint32_t, const Counters*, int, const uint32_t*, uint32_t, unsigned long long, int);
template __global__ void sort_tiles<true>(const uint32_t*, unsigned long long*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t, uint32_t, const Counters*, int, const uint32_t*, uint32_t, unsigned long long, int);

}  // namespace vtgs
