// vtgs_api.hip -- the extern "C" surface declared in include/vtgs.h: argument checks, workspace carving,
// kernel launches on the caller's stream.  No device allocation, no retained pointers.
#include "../../include/vtgs.h"
#include "vtgs_internal.h"

#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <string>

namespace vtgs {
// kernels (vtgs_binning.hip / vtgs_composite.hip)
template <int LDSBINS, int MODE>
__global__ void project_and_bin(CamScalars, const float*, const float*, int, const float*, const float*, const float*,
                                const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*,
                                Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template <int LDSBINS, int MODE>
__global__ void project_and_bin_capped(CamScalars, const float*, const float*, int, const float*, const float*, const float*,
                                const float*, int32_t*, GeomRec*, GaussAux*, uint32_t*, unsigned long long*, uint32_t*,
                                Counters*, BlockStats*, unsigned long long, uint32_t, DeferRec*);
template <bool PLANNED>
__global__ void bin_deferred_splats(CamScalars, GaussAux*, uint32_t*, unsigned long long*, uint32_t*, Counters*, const DeferRec*,
                                    uint32_t, unsigned long long, uint32_t, uint32_t);
__global__ void finalize_forward(const uint32_t*, uint32_t, Counters*, unsigned long long, uint32_t, const BlockStats*,
                                 uint32_t, VtgsForwardInfo*, const uint32_t*, uint32_t*);
template <bool WIDE>
__global__ void sort_tiles(const uint32_t*, unsigned long long*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t, uint32_t,
                           const Counters*, int, const uint32_t*, uint32_t, unsigned long long, int);
__global__ void sort_long_lists(const uint32_t*, unsigned long long*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t, uint32_t,
                                const Counters*, const uint32_t*, uint32_t, unsigned long long, int, uint32_t);
template <int MODE>
__global__ void composite_forward_q(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, uint32_t*,
                                    const GeomRec*, const float*, float*, float*, float*, const Counters*, const float*,
                                    float*, int, const unsigned long long*, const uint32_t*, uint32_t*, FinalizeArgs, uint8_t*, uint32_t*);
template <int WAVES, bool DUAL, bool PX, bool B1>
__global__ void composite_backward_mx(CamScalars, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*,
                                      const uint32_t*, const GeomRec*, const float*, const float*, const float*,
                                      const float*, float*, const Counters*, const float*, const float*, const float*, uint32_t*);
template <bool DUAL, bool FRAME, bool COV3D, bool B1>
__global__ void gather_splat_grads(CamScalars, const float*, const float*, int, const float*, const float*, const float*,
                                   const float*, const GaussAux*, const float*, int, float*, float*, float*, float*, float*,
                                   float*, const Counters*, float*, FrameEpilogue);
__global__ void mark_visible_kernel(const float*, int, const float*, uint8_t*);
__global__ void band_owner_kernel(CamScalars, const float*, const float*, int, const float*, const float*, int, const float*,
                                  const float*, float, float, const uint8_t*, uint8_t*, uint32_t*, int32_t*);
__global__ void uniform_plan_kernel(uint32_t* plan, uint32_t entries, uint32_t slots_per_bin) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < entries) plan[i] = i * slots_per_bin;
}
}  // namespace vtgs

using namespace vtgs;

static thread_local char g_hip_err[256] = "";

static int hip_fail(hipError_t e, const char* what) {
  snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, hipGetErrorString(e));
  return VTGS_ERR_HIP;
}
#define VTGS_HIP(call)                                        \
  do {                                                        \
    hipError_t e__ = (call);                                  \
    if (e__ != hipSuccess) return hip_fail(e__, #call);       \
  } while (0)

// ---- the cross-check composites (scalar, quad form, lane = pixel forward, quadrant-queue backward) are NOT in this library:
// they live in the test-only libvtgs_xcheck.so (csrc/vtgs_xcheck.hip), opened next to this library the first time an
// implementation switch asks for one of them.
typedef int (*XcheckForwardFn)(int, int, const CamScalars*, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*,
                               const GeomRec*, const float*, float*, float*, float*, const Counters*, const float*, float*, void*);
typedef int (*XcheckBackwardFn)(int, int, const CamScalars*, const float*, uint32_t, const uint32_t*, uint32_t, const uint32_t*,
                                const uint32_t*, const uint8_t*, const GeomRec*, const float*, const float*, const float*,
                                const float*, float*, const Counters*, const float*, const float*, const float*, uint32_t*, void*);
static XcheckForwardFn g_xcheck_forward = nullptr;
static XcheckBackwardFn g_xcheck_backward = nullptr;
static std::once_flag g_xcheck_once;
static bool xcheck_available() {
  std::call_once(g_xcheck_once, [] {
    Dl_info self;
    if (!dladdr((const void*)&vtgs_abi_version, &self) || !self.dli_fname) return;
    std::string path(self.dli_fname);                                 // .../libvtgs.so -> .../libvtgs_xcheck.so (an experiment
    const size_t dot = path.rfind(".so");                             // build libfoo.so looks for ITS libfoo_xcheck.so)
    if (dot == std::string::npos) return;
    path = path.substr(0, dot) + "_xcheck.so";
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    typedef uint32_t (*AbiFn)(void);
    AbiFn abi = (AbiFn)dlsym(h, "vtgs_xcheck_abi_version");
    if (!abi || abi() != VTGS_ABI_VERSION) return;                       // built from other headers: not usable
    g_xcheck_forward = (XcheckForwardFn)dlsym(h, "vtgs_xcheck_forward");
    g_xcheck_backward = (XcheckBackwardFn)dlsym(h, "vtgs_xcheck_backward");
  });
  return g_xcheck_forward && g_xcheck_backward;
}
static int xcheck_missing() {
  snprintf(g_hip_err, sizeof(g_hip_err), "a cross-check implementation was selected (VTGS_FWD_IMPL / VTGS_BWD_IMPL) but the test-only "
                                         "libvtgs_xcheck.so is not next to libvtgs.so (or was built from another ABI)");
  return VTGS_ERR_INVALID_ARGUMENT;
}

// ---- optional per-kernel event timing (vtgs_profile_*) -------------------------------------------------
struct ProfSlot { const char* name; hipEvent_t a, b; };
static bool g_prof_on = false;
static ProfSlot g_prof[8192];
static int g_prof_n = 0, g_prof_created = 0;

// VTGS_DEBUG_SYNC=1 (environment, read once): every bracketed kernel is followed by a stream synchronise and a line on stderr --
// a GPU fault then names the kernel that was in flight instead of surfacing at some later call (round 6: a fault in a new kernel
// showed up as "Memory access fault" with nothing to say which of seven launches it belonged to)
static int g_debug_sync = -1;
static bool debug_sync() {
  if (g_debug_sync < 0) { const char* v = getenv("VTGS_DEBUG_SYNC"); g_debug_sync = (v && v[0] == '1') ? 1 : 0; }
  return g_debug_sync == 1;
}

struct ProfScope {
  hipStream_t st; int idx; const char* nm;
  ProfScope(const char* name, hipStream_t s) : st(s), idx(-1), nm(name) {
    if (!g_prof_on || g_prof_n >= 8192) return;
    idx = g_prof_n++;
    if (idx >= g_prof_created) {
      if (hipEventCreate(&g_prof[idx].a) != hipSuccess || hipEventCreate(&g_prof[idx].b) != hipSuccess) { idx = -1; --g_prof_n; return; }
      g_prof_created = idx + 1;
    }
    g_prof[idx].name = name;
    (void)hipEventRecord(g_prof[idx].a, st);
  }
  ~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(g_prof[idx].b, st);
    if (debug_sync()) {
      const hipError_t e = hipStreamSynchronize(st);
      fprintf(stderr, "[vtgs] %s: %s\n", nm, e == hipSuccess ? "done" : hipGetErrorString(e));
      fflush(stderr);
    }
  }
};

static bool band_of(const VtgsCamera* cam, int* row8_begin, int* row8_end, int* rows16, int* row16_0) {
  const int gy16 = (cam->image_height + kBinTile - 1) / kBinTile;
  const int gy8 = (cam->image_height + kSubTile - 1) / kSubTile;
  int b = cam->tile_row_begin, e = cam->tile_row_end;
  if (b == 0 && e == 0) e = gy16;
  if (b < 0 || e > gy16 || b >= e) return false;
  *row8_begin = 2 * b;
  *row8_end = (2 * e < gy8) ? 2 * e : gy8;
  *rows16 = e - b;
  *row16_0 = b;
  return true;
}

static bool cam_ok(const VtgsCamera* cam) {
  return cam && cam->image_width > 0 && cam->image_height > 0 && cam->tanfovx > 0.f && cam->tanfovy > 0.f &&
         cam->bg && cam->viewmatrix && cam->projmatrix &&
         (cam->radius_rule == VTGS_RADIUS_3SIGMA || cam->radius_rule == VTGS_RADIUS_OPACITY);
}

static CamScalars scalars_of(const VtgsCamera* cam, int row8_begin, int row8_end) {
  CamScalars cs;
  cs.W = cam->image_width; cs.H = cam->image_height;
  cs.tanfovx = cam->tanfovx; cs.tanfovy = cam->tanfovy; cs.mod = cam->scale_modifier;
  cs.radius_rule = cam->radius_rule;
  cs.row8_begin = row8_begin; cs.row8_end = row8_end;
  cs.bin_plan = nullptr; cs.bin_limit = 0u;                  // planned bins: set by the caller once the workspace layout is known
  cs.bwd_flags = 0u;
  cs.scratch_records = 0xFFFFFFFFu;
  cs.raw_act = 0u;
  cs.no_defer = 0u;
#ifdef VTGS_Q_STAMPS
  cs.dbg_proj = nullptr;
#endif
  return cs;
}

extern "C" {

uint32_t vtgs_abi_version(void) { return VTGS_ABI_VERSION; }

const char* vtgs_strerror(int status) {
  switch (status) {
    case VTGS_OK: return "ok";
    case VTGS_ERR_INVALID_ARGUMENT: return "invalid argument";
    case VTGS_ERR_WORKSPACE_TOO_SMALL: return "workspace too small";
    case VTGS_ERR_INSTANCE_OVERFLOW: return "more (Gaussian,tile) instances than instance_capacity or a tile list longer than tile_capacity";
    case VTGS_ERR_HIP: return "HIP runtime error";
    case VTGS_ERR_STALE_WORKSPACE: return "workspace holds no completed forward for these sizes";
    default: return "unknown status";
  }
}

const char* vtgs_last_hip_error(void) { return g_hip_err; }

size_t vtgs_workspace_bytes(int32_t n, int32_t width, int32_t height, uint64_t instance_capacity, uint32_t tile_capacity) {
  if (n < 0 || width <= 0 || height <= 0 || (tile_capacity & ~VTGS_TILE_CAPACITY_PLANNED) == 0) return 0;
  return make_layout(n, width, height, instance_capacity, tile_capacity).total;
}

size_t vtgs_workspace_clear_bytes(int32_t image_width, int32_t image_height) {
  if (image_width <= 0 || image_height <= 0) return 0;
  const size_t tiles8 = (size_t)((image_width + kSubTile - 1) / kSubTile) * (size_t)((image_height + kSubTile - 1) / kSubTile);
  return 256 + align256((tiles8 + 1) * 4);       // WsLayout: counters at byte 0, tile_cnt behind them
}

size_t vtgs_backward_scratch_bytes(int32_t n, uint64_t instances) {
  (void)n;
  return align256((size_t)(instances ? instances : 1) * kGradRec * sizeof(float));
}

size_t vtgs_backward_dual_scratch_bytes(int32_t n, uint64_t instances) {
  (void)n;
  return align256((size_t)(instances ? instances : 1) * kGradRecDual * sizeof(float));
}

#include <stdlib.h>
#include <time.h>
// Implementation switches: defaults from the environment, read once; vtgs_set_option overrides them at run time.
struct Option { const char* name; int dflt; int value; };
static Option g_options[] = {{"VTGS_FWD_IMPL", 3, -1}, {"VTGS_BWD_IMPL", 2, -1}, {"VTGS_BIN_IMPL", 1, -1}, {"VTGS_SORT_PACKED", 1, -1},
                             {"VTGS_SORT_FUSED", 1, -1}, {"VTGS_COUNT_STEPS", 0, -1}, {"VTGS_SORT_LONG_COUNTING", 1, -1},
                             {"VTGS_DUAL_B1", 1, -1},    // 0: ignore frame flag 8 (the full dual backward: the cross-check of the one-channel form)
                             {"VTGS_DEPTH_LITE", 1, -1}}; // 0: ignore VTGS_FORWARD_SECOND_IS_DEPTH (the full dual forward)
enum { OPT_FWD_IMPL = 0, OPT_BWD_IMPL, OPT_BIN_IMPL, OPT_SORT_PACKED, OPT_SORT_FUSED, OPT_COUNT_STEPS, OPT_SORT_LONG_COUNTING, OPT_DUAL_B1,
       OPT_DEPTH_LITE, OPT_COUNT };
static std::once_flag g_options_once;
static void options_init() {                                   // thread-safe: the first caller reads the environment
  std::call_once(g_options_once, [] {
    for (int i = 0; i < OPT_COUNT; ++i) {
      const char* v = getenv(g_options[i].name);
      if (v) g_options[i].dflt = atoi(v);
      if (g_options[i].value < 0) g_options[i].value = g_options[i].dflt;
    }
  });
}
static inline int option(int which) { options_init(); return g_options[which].value; }
int vtgs_set_option(const char* name, int value) {
  options_init();
  if (!name) return VTGS_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, g_options[i].name) == 0) { g_options[i].value = value < 0 ? g_options[i].dflt : value; return VTGS_OK; }
  return VTGS_ERR_INVALID_ARGUMENT;
}
int vtgs_get_option(const char* name) {
  options_init();
  if (!name) return -1;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, g_options[i].name) == 0) return g_options[i].value;
  return -1;
}

// colors_b / out_color_b != NULL: dual render (two colour sets over the same geometry, no depth image)
static int launch_composite_forward(const VtgsCamera* cam, const CamScalars& cs, int rows16, const WsLayout& L,
                                    char* ws, const float* colors, float* out_color, float* out_depth,
                                    float* image_state, hipStream_t st, const float* colors_b = nullptr,
                                    float* out_color_b = nullptr, int sort_mode = 0, FinalizeArgs fin = FinalizeArgs{},
                                    bool write_qmask = false, bool second_is_depth = false) {
  // dual render <=> a second output image.  (Round 4: this was inferred from colors_b, and an EMPTY dual render -- n = 0, where
  // the header lets every per-Gaussian pointer be NULL -- took the single-render kernel with its NULL depth plane: a write
  // through address 0 + the band's pixel offset, found by tests/test_gpu_owned_sets.py::test_an_empty_list_renders_the_background.)
  const bool dual = out_color_b != nullptr;
  const int gx16 = (cam->image_width + kBinTile - 1) / kBinTile;
  uint32_t* steps = nullptr;                                 // measurement only: steps of the quadrant-queue forward, 64 partial sums
  if (option(OPT_COUNT_STEPS) == 1 && write_qmask) {
    steps = (uint32_t*)(ws + L.dbg);
    VTGS_HIP(hipMemsetAsync(steps, 0, 256, st));
  }
  const uint32_t nblk = (uint32_t)(gx16 * rows16);
  const int impl = option(OPT_FWD_IMPL);                     // 2 = lane-per-pixel matrix-core kernel (default), 1 = pixel x splat-quad
                                                           // matrix-core kernel, 0 = scalar kernel (read per call)
  {
    ProfScope ps__(dual ? "composite_forward_dual" : "composite_forward", st);
    if (impl == 3 && dual && second_is_depth)               // the single render's kernel with z in its depth column (see the kernel)
      hipLaunchKernelGGL((composite_forward_q<2>), dim3(nblk), dim3(256), 0, st, cs, cam->bg, nblk,
                         (const uint32_t*)(ws + L.tile_cnt), L.tile_cap, (uint32_t*)(ws + L.sorted_gid),
                         (const GeomRec*)(ws + L.geom), colors, out_color, (float*)nullptr, image_state,
                         (const Counters*)(ws + L.counters), colors_b, out_color_b, sort_mode,
                         (const unsigned long long*)(ws + L.keys), (const uint32_t*)(ws + L.vals), (uint32_t*)(ws + L.sorted_inst), fin,
                         write_qmask ? (uint8_t*)(ws + L.qmask) : (uint8_t*)nullptr, steps);
    else if (impl == 3 && dual)                             // quadrant queues (vtgs_composite_q.hip)
      hipLaunchKernelGGL((composite_forward_q<1>), dim3(nblk), dim3(256), 0, st, cs, cam->bg, nblk,
                         (const uint32_t*)(ws + L.tile_cnt), L.tile_cap, (uint32_t*)(ws + L.sorted_gid),
                         (const GeomRec*)(ws + L.geom), colors, out_color, (float*)nullptr, image_state,
                         (const Counters*)(ws + L.counters), colors_b, out_color_b, sort_mode,
                         (const unsigned long long*)(ws + L.keys), (const uint32_t*)(ws + L.vals), (uint32_t*)(ws + L.sorted_inst), fin,
                         write_qmask ? (uint8_t*)(ws + L.qmask) : (uint8_t*)nullptr, steps);
    else if (impl == 3)
      hipLaunchKernelGGL((composite_forward_q<0>), dim3(nblk), dim3(256), 0, st, cs, cam->bg, nblk,
                         (const uint32_t*)(ws + L.tile_cnt), L.tile_cap, (uint32_t*)(ws + L.sorted_gid),
                         (const GeomRec*)(ws + L.geom), colors, out_color, out_depth, image_state,
                         (const Counters*)(ws + L.counters), (const float*)nullptr, (float*)nullptr, sort_mode,
                         (const unsigned long long*)(ws + L.keys), (const uint32_t*)(ws + L.vals), (uint32_t*)(ws + L.sorted_inst), fin,
                         write_qmask ? (uint8_t*)(ws + L.qmask) : (uint8_t*)nullptr, steps);
    else {                                                  // a cross-check implementation: the test-only library
      if (!xcheck_available()) return xcheck_missing();
      const int rc = g_xcheck_forward(impl, dual ? 1 : 0, &cs, cam->bg, nblk, (const uint32_t*)(ws + L.tile_cnt), L.tile_cap,
                                      (const uint32_t*)(ws + L.sorted_gid), (const GeomRec*)(ws + L.geom), colors, out_color,
                                      dual ? (float*)nullptr : out_depth, image_state, (const Counters*)(ws + L.counters),
                                      colors_b, out_color_b, (void*)st);
      if (rc != 0) return hip_fail(hipGetLastError(), "vtgs_xcheck_forward");
    }
  }
  VTGS_HIP(hipGetLastError());
  return VTGS_OK;
}

static int forward_impl(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors, const float* colors_b,
                        const float* opacities, const float* scales, const float* rotations, float* out_color,
                        float* out_depth, float* out_color_b, int32_t* out_radii, void* workspace, size_t workspace_bytes,
                        uint64_t instance_capacity, uint32_t tile_capacity, VtgsForwardInfo* info, uint32_t flags,
                        void* stream, bool dual, uint32_t* bin_plan = nullptr, bool cov3d = false) {
  // cov3d: `scales` is cov3D_precomp [N,6] and `rotations` is not read (vtgs_forward_cov3d: single render, uniform bins)
  if (!cam_ok(cam) || n < 0 || !out_color || (dual ? !out_color_b : !out_depth) || !workspace || instance_capacity == 0 ||
      instance_capacity > 0xFFFFFFFFull || (tile_capacity & ~VTGS_TILE_CAPACITY_PLANNED) == 0 ||
      ((tile_capacity & VTGS_TILE_CAPACITY_PLANNED) != 0) != (bin_plan != nullptr) || (cov3d && (dual || bin_plan)))
    return VTGS_ERR_INVALID_ARGUMENT;
  const bool raw_act = (flags & VTGS_FORWARD_RAW_ACTIVATIONS) != 0u;      // logits / log-scales in, rotations not read (dual render only)
  if (raw_act && (!dual || cov3d)) return VTGS_ERR_INVALID_ARGUMENT;
  if (n > 0 && (!means3D || !colors || (dual && !colors_b) || !opacities || !scales || (!rotations && !cov3d && !raw_act) || !out_radii))
    return VTGS_ERR_INVALID_ARGUMENT;
  int r8b, r8e, rows16, row16_0;
  if (!band_of(cam, &r8b, &r8e, &rows16, &row16_0)) return VTGS_ERR_INVALID_ARGUMENT;
  const WsLayout L = make_layout(n, cam->image_width, cam->image_height, instance_capacity, tile_capacity);
  if (workspace_bytes < L.total) return VTGS_ERR_WORKSPACE_TOO_SMALL;
  if ((uint64_t)L.tiles8 * L.tile_cap >= (1ull << 32)) return VTGS_ERR_INVALID_ARGUMENT;   // bin offsets are 32-bit in the composites
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  CamScalars cs = scalars_of(cam, r8b, r8e);
  cs.raw_act = raw_act ? 1u : 0u;
  Counters* ctr = (Counters*)(ws + L.counters);
#ifdef VTGS_Q_STAMPS
  cs.dbg_proj = (uint32_t*)(ws + L.dbg + align256(256 + (size_t)L.tiles8 * kStampWords * 4));
#endif
  if (L.planned) {
    // this forward (and its backward) bins into a COPY of the caller's plan: the persistent one is rewritten for the next
    // forward by finalize_forward.  A plan that does not fit the workspace (its total is on the device) makes every bin
    // look empty-capacity to nobody: finalize_forward flags it (plan[tiles] > slots) like any other overflow.
    VTGS_HIP(hipMemcpyAsync(ws + L.plan, bin_plan, ((size_t)L.tiles8 + 1) * 4, hipMemcpyDeviceToDevice, st));
    cs.bin_plan = (const uint32_t*)(ws + L.plan); cs.bin_limit = L.tiles8 * L.tile_cap;
  }

  // (per-kernel profiling only) an EMPTY bracket: what two event records cost with nothing between them -- every kernel's bracket
  // is inflated by about that much (the brackets of round 5 summed to more than the step); vtgs_profile_collect reports it as
  // "_empty_bracket" and bench.py subtracts it
  { ProfScope ps__("_empty_bracket", st); }
  // counters + per-tile list lengths in one fill (adjacent in the layout, padded to 256 B).  (Round 5, measured and dropped:
  // clearing inside project_and_bin under a token costs +27 us -- the check sits on every workgroup's latency chain -- and a
  // clear kernel of this library takes the same 4.6-5 us as the runtime's fill; profiles/r5_negative_results.md)
  // VTGS_FORWARD_WORKSPACE_CLEARED (round 6): the caller did it (vtgs_prepare_frame_slot, in the launch it makes anyway)
  if (!(flags & VTGS_FORWARD_WORKSPACE_CLEARED))
    VTGS_HIP(hipMemsetAsync(ws + L.counters, 0, 256 + align256(((size_t)L.tiles8 + 1) * 4), st));
  if (rows16 * kBinTile < cam->image_height || cam->tile_row_begin != 0) {
    // band mode: pixels outside the band are written as zero (include/vtgs.h); the caller keeps only
    // its own rows when it assembles the bands
    const size_t P = (size_t)cam->image_width * cam->image_height;
    VTGS_HIP(hipMemsetAsync(out_color, 0, 3 * P * sizeof(float), st));
    if (dual) VTGS_HIP(hipMemsetAsync(out_color_b, 0, 3 * P * sizeof(float), st));
    else VTGS_HIP(hipMemsetAsync(out_depth, 0, P * sizeof(float), st));
  }
  bool no_defer = false;                                        // VTGS_FORWARD_EXPECT_NO_DEFERRED honoured: no second binning launch
  if (n > 0) {
    // LDS-binned form while the per-tile table (4 B per 8x8 tile of this call's band) leaves room for two workgroups
    // per CU: 79 KB = 20 K tiles (1200x680: 12.7 K; 1296x968: 19.6 K; a band of 1752x1168 on >= 2 GPUs).  Beyond that
    // the run-aggregated global-atomic walk is faster again (5 M splats at 1752x1168: 488 us vs 525 us with a 128 KB
    // table and one workgroup per CU).
    const size_t table_bytes = (size_t)(r8e - r8b) * (size_t)((cam->image_width + kSubTile - 1) / kSubTile) * 4;
    bool lds_bins = table_bytes <= (79u << 10) && option(OPT_BIN_IMPL) == 1;   // (VTGS_BIN_IMPL = 2: the windowed table at every size)
    // a larger frame (1752x1168: 32 K tiles): the windowed table (VTGS_BIN_IMPL = 1, default) -- not for the cov3D form,
    // which has no windowed instantiation; VTGS_BIN_IMPL = 0: global atomics everywhere (cross-check)
    const bool win_bins = !lds_bins && option(OPT_BIN_IMPL) >= 1 && !cov3d &&
                          (uint32_t)((cam->image_width + kSubTile - 1) / kSubTile) <= 16384u;
    if (lds_bins && table_bytes > (64u << 10)) {
      static bool raised[64] = {false};                       // the attribute sticks to the function, per DEVICE
      int dev_id = 0;
      if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) { (void)hipGetLastError(); dev_id = -1; }
      if (dev_id < 0 || !raised[dev_id]) {
        if (hipFuncSetAttribute((const void*)project_and_bin<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10) == hipSuccess &&
            hipFuncSetAttribute((const void*)project_and_bin_capped<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10) == hipSuccess &&
            hipFuncSetAttribute((const void*)project_and_bin_capped<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10) == hipSuccess &&
            hipFuncSetAttribute((const void*)project_and_bin_capped<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10) == hipSuccess &&
            hipFuncSetAttribute((const void*)project_and_bin_capped<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10) == hipSuccess &&
            hipFuncSetAttribute((const void*)project_and_bin_capped<1, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10) == hipSuccess) {
          if (dev_id >= 0) raised[dev_id] = true;
        } else { (void)hipGetLastError(); lds_bins = false; }
      }
    }
    {
      ProfScope ps__("project_and_bin", st);
      const int mode = ((r8b > 0 || r8e < (cam->image_height + kSubTile - 1) / kSubTile) ? 1 : 0) | (L.planned ? 2 : 0) | (cov3d ? 4 : 0);
      if (mode == 0 && (flags & VTGS_FORWARD_EXPECT_NO_DEFERRED)) { cs.no_defer = 1u; no_defer = true; }   // (the hint: whole frame, uniform bins)
#define VTGS_LAUNCH_PROJECT(LDS, MODE, SHMEM)                                                                          \
      hipLaunchKernelGGL((VTGS_PROJECT_KERNEL(LDS, MODE)), dim3((n + 1023) / 1024), dim3(1024), SHMEM, st, cs,             \
                         cam->viewmatrix, cam->projmatrix, n, means3D, opacities, scales, rotations, out_radii,         \
                         (GeomRec*)(ws + L.geom), (GaussAux*)(ws + L.gaux), (uint32_t*)(ws + L.tile_cnt),              \
                         (unsigned long long*)(ws + L.keys), (uint32_t*)(ws + L.vals), ctr,                            \
                         (BlockStats*)(ws + L.block_stats), (unsigned long long)instance_capacity, L.tile_cap,            \
                         (DeferRec*)(ws + L.defer_list))
      // (LDS: 0 = run-aggregated global atomics, 1 = the band's whole tile table in LDS, 2 = a window of kWinEntries tiles)
      if (lds_bins) {
#define VTGS_PROJECT_KERNEL(LDS, MODE) project_and_bin<LDS, MODE>
        if (mode == 0) VTGS_LAUNCH_PROJECT(1, 0, table_bytes);
#undef VTGS_PROJECT_KERNEL
#define VTGS_PROJECT_KERNEL(LDS, MODE) project_and_bin_capped<LDS, MODE>
        else if (mode == 1) VTGS_LAUNCH_PROJECT(1, 1, table_bytes);
        else if (mode == 2) VTGS_LAUNCH_PROJECT(1, 2, table_bytes);
        else if (mode == 4) VTGS_LAUNCH_PROJECT(1, 4, table_bytes);
        else if (mode == 5) VTGS_LAUNCH_PROJECT(1, 5, table_bytes);
        else VTGS_LAUNCH_PROJECT(1, 3, table_bytes);
      } else if (win_bins) {
#undef VTGS_PROJECT_KERNEL
#define VTGS_PROJECT_KERNEL(LDS, MODE) project_and_bin<LDS, MODE>
        if (mode == 0) VTGS_LAUNCH_PROJECT(2, 0, (size_t)(64u << 10));
#undef VTGS_PROJECT_KERNEL
#define VTGS_PROJECT_KERNEL(LDS, MODE) project_and_bin_capped<LDS, MODE>
        else if (mode == 1) VTGS_LAUNCH_PROJECT(2, 1, (size_t)(64u << 10));
        else if (mode == 2) VTGS_LAUNCH_PROJECT(2, 2, (size_t)(64u << 10));
        else VTGS_LAUNCH_PROJECT(2, 3, (size_t)(64u << 10));
      } else {
#undef VTGS_PROJECT_KERNEL
#define VTGS_PROJECT_KERNEL(LDS, MODE) project_and_bin<LDS, MODE>
        if (mode == 0) VTGS_LAUNCH_PROJECT(0, 0, 0);
#undef VTGS_PROJECT_KERNEL
#define VTGS_PROJECT_KERNEL(LDS, MODE) project_and_bin_capped<LDS, MODE>
        else if (mode == 1) VTGS_LAUNCH_PROJECT(0, 1, 0);
        else if (mode == 2) VTGS_LAUNCH_PROJECT(0, 2, 0);
        else if (mode == 4) VTGS_LAUNCH_PROJECT(0, 4, 0);
        else if (mode == 5) VTGS_LAUNCH_PROJECT(0, 5, 0);
        else VTGS_LAUNCH_PROJECT(0, 3, 0);
      }
#undef VTGS_PROJECT_KERNEL
#undef VTGS_LAUNCH_PROJECT
    }
    VTGS_HIP(hipGetLastError());
    if (!no_defer) {
      // the splats project_and_bin left aside (more than kDeferArea candidate tiles): binned here.  The lists' lengths are on the
      // device, so the grid is what lists of N / 8 small and N / 256 large entries need (longer lists: more rounds per
      // workgroup); with empty lists every workgroup leaves after one load.  The large list's workgroups come first.
      ProfScope ps__("bin_deferred_splats", st);
      const uint32_t lgrid = (uint32_t)min((long long)1024, max((long long)16, (long long)n / 1024));
      const uint32_t dgrid = lgrid + (uint32_t)min((long long)4096, max((long long)64, ((long long)n / 8 + 63) / 64));
      if (L.planned)
        hipLaunchKernelGGL((bin_deferred_splats<true>), dim3(dgrid), dim3(256), 0, st, cs, (GaussAux*)(ws + L.gaux), (uint32_t*)(ws + L.tile_cnt),
                           (unsigned long long*)(ws + L.keys), (uint32_t*)(ws + L.vals), ctr, (const DeferRec*)(ws + L.defer_list),
                           (uint32_t)n, (unsigned long long)instance_capacity, L.tile_cap, lgrid);
      else
        hipLaunchKernelGGL((bin_deferred_splats<false>), dim3(dgrid), dim3(256), 0, st, cs, (GaussAux*)(ws + L.gaux), (uint32_t*)(ws + L.tile_cnt),
                           (unsigned long long*)(ws + L.keys), (uint32_t*)(ws + L.vals), ctr, (const DeferRec*)(ws + L.defer_list),
                           (uint32_t)n, (unsigned long long)instance_capacity, L.tile_cap, lgrid);
    }
    VTGS_HIP(hipGetLastError());
  }
  // overflow flags and statistics first: on overflow some bin slots were never written, so the consumers must bail
  // asynchronous mode: the record goes straight into the caller's pinned buffer when the device can address it
  // (one store from the kernel instead of a copy command behind the composite)
  VtgsForwardInfo* host_record = nullptr;
  if ((flags & (VTGS_FORWARD_ASYNC | VTGS_FORWARD_CHECKED)) && info) {
    if (flags & VTGS_FORWARD_CHECKED) ((volatile VtgsForwardInfo*)info)->complete = 0u;
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, info, 0) == hipSuccess) host_record = (VtgsForwardInfo*)dp;
    else (void)hipGetLastError();                               // not mapped: fall back to the copy below
  }
  const uint32_t gx8 = (uint32_t)((cam->image_width + kSubTile - 1) / kSubTile);
  const uint32_t band_first = (uint32_t)r8b * gx8, band_tiles = (uint32_t)(r8e - r8b) * gx8;
  const int packed = (n <= (1 << 21) && option(OPT_SORT_PACKED) == 1) ? 1 : 0;
  // Who sorts what.  With the quadrant-queue forward and VTGS_SORT_FUSED (the defaults) the forward sorts the lists of up to
  // 512 entries itself (counting sort at the top of each wavefront; up to 1,024 with its network when nobody pre-sorted them)
  // and its first workgroup does finalize_forward's job.  Ahead of it, only where such lists can exist:
  //   pre512   sort_long_lists: lists of 513 .. 2,048 entries, one workgroup each (bins of >= 768 slots: the package sizes
  //            bins at 1.5 x the longest list, so smaller bins hold no list beyond 512 entries; planned bins: always);
  //   pre2048  a pass of sort_tiles over the lists beyond 2,048 entries (bins that large, or planned bins).
  // Both make the forward's own safety test (finalize has not run yet): nothing if the instance capacity overflowed, no
  // list whose bin overflowed.  Otherwise (other forwards, VTGS_SORT_FUSED = 0): finalize_forward, sort_long_lists, sort_tiles.
  const bool fused_sort = option(OPT_FWD_IMPL) == 3 && option(OPT_SORT_FUSED) == 1;
  // (VTGS_FORWARD_EXPECT_SHORT_LISTS: bins of 768 .. 1,024 slots without the pre-sort pass -- the composite's own network
  // takes a list of 513 .. 1,024 entries that turns up against the caller's expectation)
  const bool pre512 = L.planned || L.tile_cap > 1024u || (L.tile_cap >= 768u && !(flags & VTGS_FORWARD_EXPECT_SHORT_LISTS));
  const bool pre2048 = L.planned || L.tile_cap > 2048u;
  FinalizeArgs fin;
  fin.tile_cnt = (const uint32_t*)(ws + L.tile_cnt); fin.tiles = L.tiles8; fin.ctr = ctr;
  fin.capacity = (unsigned long long)instance_capacity; fin.tile_cap = L.tile_cap;
  fin.block_stats = (const BlockStats*)(ws + L.block_stats); fin.nblocks = (uint32_t)((n + 1023) / 1024);
  fin.host_record = host_record;
  fin.plan = cs.bin_plan; fin.plan_next = L.planned ? bin_plan : nullptr;
  if (!fused_sort) {
    ProfScope ps__("finalize_forward", st);
    hipLaunchKernelGGL(finalize_forward, dim3(1), dim3(1024), 0, st, fin.tile_cnt, fin.tiles, ctr, fin.capacity, fin.tile_cap,
                       fin.block_stats, fin.nblocks, host_record, fin.plan, fin.plan_next);
  }
  VTGS_HIP(hipGetLastError());
  const unsigned long long own_checks = fused_sort ? (unsigned long long)instance_capacity : 0ull;   // != 0: no overflow flag yet
  if (pre512 || !fused_sort) {
    ProfScope ps__("sort_tiles", st);
    if (pre512)
      hipLaunchKernelGGL(sort_long_lists, dim3(band_tiles), dim3(256), 0, st, (const uint32_t*)(ws + L.tile_cnt),
                         (unsigned long long*)(ws + L.keys), (uint32_t*)(ws + L.vals), (uint32_t*)(ws + L.sorted_gid),
                         (uint32_t*)(ws + L.sorted_inst), band_first, band_tiles, L.tile_cap, (const Counters*)ctr, cs.bin_plan,
                         cs.bin_limit, own_checks, option(OPT_SORT_LONG_COUNTING), 512u);
    if (!fused_sort || pre2048)
      hipLaunchKernelGGL(sort_tiles<false>, dim3((band_tiles + 3) / 4), dim3(256), 0, st, (const uint32_t*)(ws + L.tile_cnt),
                         (unsigned long long*)(ws + L.keys), (uint32_t*)(ws + L.vals), (uint32_t*)(ws + L.sorted_gid),
                         (uint32_t*)(ws + L.sorted_inst), band_first, band_tiles, L.tile_cap, (const Counters*)ctr, packed,
                         cs.bin_plan, cs.bin_limit, own_checks, pre512 ? 1 : 0);
  }
  VTGS_HIP(hipGetLastError());
  int rc = launch_composite_forward(cam, cs, rows16, L, ws, colors, out_color, out_depth, (float*)(ws + L.final_T), st,
                                    dual ? colors_b : nullptr, dual ? out_color_b : nullptr,
                                    fused_sort ? ((packed ? 1 : 2) | (pre512 ? 4 : 0)) : 0, fin, true,
                                    dual && (flags & VTGS_FORWARD_SECOND_IS_DEPTH) && option(OPT_DEPTH_LITE) != 0);
  if (rc != VTGS_OK) return rc;
  // result record: assembled on the device by finalize_forward at byte 64 of the counters block
  static_assert(sizeof(VtgsForwardInfo) == 48, "VtgsForwardInfo layout is mirrored in Counters");
  const char* image = (const char*)ctr + offsetof(Counters, info_instances);
  if (flags & VTGS_FORWARD_ASYNC) {
    if (!info) return VTGS_ERR_INVALID_ARGUMENT;
    if (!host_record) VTGS_HIP(hipMemcpyAsync(info, image, sizeof(VtgsForwardInfo), hipMemcpyDeviceToHost, st));
    return VTGS_OK;                                             // (else finalize_forward already stored it there)
  }
  if ((flags & VTGS_FORWARD_CHECKED) && host_record) {
    // The record lands in the caller's pinned memory when finalize_forward retires -- right after the binning, while
    // the sort and the composite are still queued behind it.  Wait for it on the host (bounded: a stream error would
    // never set the flag), so that a capacity overflow is answered before the caller sees an image.
    volatile VtgsForwardInfo* rec = (volatile VtgsForwardInfo*)info;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    bool landed = false;
    for (uint64_t spin = 0;; ++spin) {
      if (rec->complete) { landed = true; break; }
      if ((spin & 0x3FFu) == 0x3FFu) {
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) > 20) break;
        if (hipStreamQuery(st) == hipSuccess) { landed = rec->complete != 0u; break; }   // stream drained: now or never
      }
      __builtin_ia32_pause();
    }
    if (!landed) {
      VTGS_HIP(hipStreamSynchronize(st));
      if (!rec->complete) { snprintf(g_hip_err, sizeof(g_hip_err), "vtgs_forward: result record never arrived"); return VTGS_ERR_HIP; }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return info->overflow ? VTGS_ERR_INSTANCE_OVERFLOW : VTGS_OK;
  }
  VtgsForwardInfo host;
  VTGS_HIP(hipMemcpyAsync(&host, image, sizeof(VtgsForwardInfo), hipMemcpyDeviceToHost, st));
  VTGS_HIP(hipStreamSynchronize(st));
  if (info) *info = host;
  return host.overflow ? VTGS_ERR_INSTANCE_OVERFLOW : VTGS_OK;
}

int vtgs_forward(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors, const float* opacities,
                 const float* scales, const float* rotations, float* out_color, float* out_depth, int32_t* out_radii,
                 void* workspace, size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity,
                 VtgsForwardInfo* info, uint32_t flags, void* stream) {
  return forward_impl(cam, n, means3D, colors, nullptr, opacities, scales, rotations, out_color, out_depth, nullptr,
                      out_radii, workspace, workspace_bytes, instance_capacity, tile_capacity, info, flags, stream, false);
}

int vtgs_forward_dual(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors_a, const float* colors_b,
                      const float* opacities, const float* scales, const float* rotations, float* out_color_a,
                      float* out_color_b, int32_t* out_radii, void* workspace, size_t workspace_bytes,
                      uint64_t instance_capacity, uint32_t tile_capacity, VtgsForwardInfo* info, uint32_t flags,
                      void* stream) {
  return forward_impl(cam, n, means3D, colors_a, colors_b, opacities, scales, rotations, out_color_a, nullptr, out_color_b,
                      out_radii, workspace, workspace_bytes, instance_capacity, tile_capacity, info, flags, stream, true);
}

int vtgs_forward_planned(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors, const float* opacities,
                         const float* scales, const float* rotations, float* out_color, float* out_depth, int32_t* out_radii,
                         void* workspace, size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity,
                         uint32_t* bin_plan, VtgsForwardInfo* info, uint32_t flags, void* stream) {
  return forward_impl(cam, n, means3D, colors, nullptr, opacities, scales, rotations, out_color, out_depth, nullptr,
                      out_radii, workspace, workspace_bytes, instance_capacity, tile_capacity, info, flags, stream, false,
                      bin_plan);
}

int vtgs_forward_dual_planned(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors_a,
                              const float* colors_b, const float* opacities, const float* scales, const float* rotations,
                              float* out_color_a, float* out_color_b, int32_t* out_radii, void* workspace,
                              size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity, uint32_t* bin_plan,
                              VtgsForwardInfo* info, uint32_t flags, void* stream) {
  return forward_impl(cam, n, means3D, colors_a, colors_b, opacities, scales, rotations, out_color_a, nullptr, out_color_b,
                      out_radii, workspace, workspace_bytes, instance_capacity, tile_capacity, info, flags, stream, true,
                      bin_plan);
}

uint32_t vtgs_bin_plan_entries(int32_t width, int32_t height) {
  if (width <= 0 || height <= 0) return 0;
  return (uint32_t)((width + kSubTile - 1) / kSubTile) * (uint32_t)((height + kSubTile - 1) / kSubTile) + 1u;
}

int vtgs_bin_plan_uniform(int32_t width, int32_t height, uint32_t slots_per_bin, uint32_t* bin_plan, void* stream) {
  const uint32_t entries = vtgs_bin_plan_entries(width, height);
  if (!entries || !bin_plan || slots_per_bin == 0 || (uint64_t)(entries - 1) * slots_per_bin >= (1ull << 31))
    return VTGS_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(uniform_plan_kernel, dim3((entries + 255) / 256), dim3(256), 0, (hipStream_t)stream, bin_plan, entries,
                     slots_per_bin);
  VTGS_HIP(hipGetLastError());
  return VTGS_OK;
}

int vtgs_forward_shared(const VtgsCamera* cam, int32_t n, const float* colors, float* out_color, float* out_depth,
                        const void* workspace, size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity,
                        float* image_state, void* stream) {
  if (!cam_ok(cam) || n < 0 || (n > 0 && !colors) || !out_color || !out_depth || !workspace || !image_state ||
      instance_capacity == 0 || instance_capacity > 0xFFFFFFFFull || (tile_capacity & ~VTGS_TILE_CAPACITY_PLANNED) == 0)
    return VTGS_ERR_INVALID_ARGUMENT;                          // (n = 0: per-Gaussian arrays may be NULL, include/vtgs.h)
  int r8b, r8e, rows16, row16_0;
  if (!band_of(cam, &r8b, &r8e, &rows16, &row16_0)) return VTGS_ERR_INVALID_ARGUMENT;
  const WsLayout L = make_layout(n, cam->image_width, cam->image_height, instance_capacity, tile_capacity);
  if (workspace_bytes < L.total) return VTGS_ERR_WORKSPACE_TOO_SMALL;
  hipStream_t st = (hipStream_t)stream;
  CamScalars cs = scalars_of(cam, r8b, r8e);
  if (L.planned) { cs.bin_plan = (const uint32_t*)((const char*)workspace + L.plan); cs.bin_limit = L.tiles8 * L.tile_cap; }
  if (rows16 * kBinTile < cam->image_height || cam->tile_row_begin != 0) {
    const size_t P = (size_t)cam->image_width * cam->image_height;
    VTGS_HIP(hipMemsetAsync(out_color, 0, 3 * P * sizeof(float), st));
    VTGS_HIP(hipMemsetAsync(out_depth, 0, P * sizeof(float), st));
  }
  return launch_composite_forward(cam, cs, rows16, L, (char*)workspace, colors, out_color, out_depth, image_state, st);
}

static int backward_impl(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors, const float* colors_b,
                         const float* opacities, const float* scales, const float* rotations, const float* out_color,
                         const float* out_color_b, const float* grad_color, const float* grad_color_b,
                         const void* workspace, size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity,
                         const float* image_state, void* scratch, size_t scratch_bytes, float* g_means3D, float* g_means2D,
                         float* g_colors, float* g_colors_b, float* g_opacities, float* g_scales, float* g_rotations,
                         void* stream, bool dual, const FrameEpilogue* frame = nullptr, bool cov3d = false) {
  if (!cam_ok(cam) || n < 0 || !out_color || !grad_color || !workspace || !scratch || instance_capacity == 0 ||
      instance_capacity > 0xFFFFFFFFull || (tile_capacity & ~VTGS_TILE_CAPACITY_PLANNED) == 0 || (dual && (!out_color_b || !grad_color_b)) || (frame && !dual))
    return VTGS_ERR_INVALID_ARGUMENT;
  const bool raw_act = frame && (frame->flags & 16u) != 0u;                // the forward's VTGS_FORWARD_RAW_ACTIVATIONS
  if (raw_act && frame->idx) return VTGS_ERR_INVALID_ARGUMENT;             // (an owned list renders from fully prepared compact arrays)
  if (n > 0 && (!means3D || !colors || !opacities || !scales || (!rotations && !cov3d && !raw_act) || (dual && !colors_b)))
    return VTGS_ERR_INVALID_ARGUMENT;
  if (cov3d && (dual || frame)) return VTGS_ERR_INVALID_ARGUMENT;
  // any output may be NULL (a gradient nobody asked for -- the tracking loop detaches the Gaussians,
  // src/vtgaussian_slam.py:428-449 -- is then neither stored nor its array allocated), but not all of them
  if (n > 0 && !frame && !g_means3D && !g_means2D && !g_colors && !g_opacities && !g_scales && !g_rotations && !(dual && g_colors_b))
    return VTGS_ERR_INVALID_ARGUMENT;
  if (n > 0 && frame) {                                      // the gradients leave through the frame epilogue instead
    const FrameEpilogue& f = *frame;
    if (!f.means3D_world || !f.cam_q || !f.cam_t || !f.depth_w2c) return VTGS_ERR_INVALID_ARGUMENT;
    if ((f.flags & 1u) && (!f.unnorm_rot || !f.g_means3D || !f.g_unnorm_rot)) return VTGS_ERR_INVALID_ARGUMENT;
    if ((f.flags & 2u) && !f.pose_partials) return VTGS_ERR_INVALID_ARGUMENT;
    if ((f.flags & 4u) && (!f.g_rgb || !f.g_logit || !f.g_log_scales)) return VTGS_ERR_INVALID_ARGUMENT;
  }
  int r8b, r8e, rows16, row16_0;
  if (!band_of(cam, &r8b, &r8e, &rows16, &row16_0)) return VTGS_ERR_INVALID_ARGUMENT;
  if (n == 0) return VTGS_OK;
  const WsLayout L = make_layout(n, cam->image_width, cam->image_height, instance_capacity, tile_capacity);
  if (workspace_bytes < L.total) return VTGS_ERR_WORKSPACE_TOO_SMALL;
  if (scratch_bytes < (dual ? kGradRecDual : kGradRec) * sizeof(float)) return VTGS_ERR_INVALID_ARGUMENT;
  hipStream_t st = (hipStream_t)stream;
  const char* ws = (const char*)workspace;
  CamScalars cs = scalars_of(cam, r8b, r8e);
  if (L.planned) { cs.bin_plan = (const uint32_t*)(ws + L.plan); cs.bin_limit = L.tiles8 * L.tile_cap; }
  const float* state = image_state ? image_state : (const float*)(ws + L.final_T);
  const int gx16 = (cam->image_width + kBinTile - 1) / kBinTile;
  const uint32_t nblk16 = (uint32_t)(gx16 * rows16);
  uint32_t* dbg = option(OPT_COUNT_STEPS) == 1 ? (uint32_t*)(const_cast<char*>(ws) + L.dbg) : nullptr;   // measurement only
  int bwd_impl = option(OPT_BWD_IMPL);                       // 2 = lane-per-pixel matrix-core replay (default), 1 = pixel x splat-quad replay,
  // dL/d(first colour set) not asked for (tracking: the Gaussians are detached, src/vtgaussian_slam.py:428-449, and the pose
  // gradient flows through means3D and the SECOND set's [z, 1, z^2] only): the lane = pixel backward skips that contraction chain
  if (frame ? !(frame->flags & 4u) : !g_colors) cs.bwd_flags |= 1u;
  cs.raw_act = raw_act ? 1u : 0u;
  if (dual && bwd_impl == 0) bwd_impl = 1;                   // 0 = scalar kernel (single render only); read per call
  if (dual && bwd_impl == 3) bwd_impl = 2;                   // 3 = quadrant queues (single render only so far)
  // FrameEpilogue flag 8 (the caller's promise: grad_color_b is zero outside its first channel -- get_loss, whose loss reaches
  // the [z, 1, z^2] render through z alone): four gradient channels instead of six, three contraction chains, 48-byte records
  const bool b1 = dual && frame && (frame->flags & 8u) && bwd_impl == 2 && option(OPT_DUAL_B1) != 0;
  {
    const size_t rec_bytes = (size_t)(b1 ? kGradRecDual1 : (dual ? kGradRecDual : kGradRec)) * sizeof(float);
    const size_t recs = scratch_bytes / rec_bytes;
    cs.scratch_records = recs > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)recs;
  }
  {
    ProfScope ps__(dual ? "composite_backward_dual" : "composite_backward", st);
    if (b1)
      hipLaunchKernelGGL((composite_backward_mx<4, true, true, true>), dim3(nblk16), dim3(256), 0, st, cs, cam->bg, nblk16,
                         (const uint32_t*)(ws + L.tile_cnt), L.tile_cap, (const uint32_t*)(ws + L.sorted_gid),
                         (const uint32_t*)(ws + L.sorted_inst), (const GeomRec*)(ws + L.geom), colors, out_color,
                         grad_color, state, (float*)scratch, (const Counters*)(ws + L.counters), colors_b, out_color_b,
                         grad_color_b, dbg);
    else if (dual && bwd_impl == 2)
      hipLaunchKernelGGL((composite_backward_mx<4, true, true, false>), dim3(nblk16), dim3(256), 0, st, cs, cam->bg, nblk16,
                         (const uint32_t*)(ws + L.tile_cnt), L.tile_cap, (const uint32_t*)(ws + L.sorted_gid),
                         (const uint32_t*)(ws + L.sorted_inst), (const GeomRec*)(ws + L.geom), colors, out_color,
                         grad_color, state, (float*)scratch, (const Counters*)(ws + L.counters), colors_b, out_color_b,
                         grad_color_b, dbg);
    else if (!dual && bwd_impl == 2)                        // lane = pixel replay (the default)
      hipLaunchKernelGGL((composite_backward_mx<4, false, true, false>), dim3(nblk16), dim3(256), 0, st, cs, cam->bg, nblk16,
                         (const uint32_t*)(ws + L.tile_cnt), L.tile_cap, (const uint32_t*)(ws + L.sorted_gid),
                         (const uint32_t*)(ws + L.sorted_inst), (const GeomRec*)(ws + L.geom), colors, out_color,
                         grad_color, state, (float*)scratch, (const Counters*)(ws + L.counters), (const float*)nullptr,
                         (const float*)nullptr, (const float*)nullptr, dbg);
    else {                                                  // quad form, scalar, quadrant queues: the test-only library
      if (!xcheck_available()) return xcheck_missing();
      const int rc = g_xcheck_backward(bwd_impl, dual ? 1 : 0, &cs, cam->bg, nblk16, (const uint32_t*)(ws + L.tile_cnt), L.tile_cap,
                                       (const uint32_t*)(ws + L.sorted_gid), (const uint32_t*)(ws + L.sorted_inst),
                                       (const uint8_t*)(ws + L.qmask), (const GeomRec*)(ws + L.geom), colors, out_color, grad_color,
                                       state, (float*)scratch, (const Counters*)(ws + L.counters), colors_b, out_color_b,
                                       grad_color_b, dbg, (void*)st);
      if (rc != 0) return hip_fail(hipGetLastError(), "vtgs_xcheck_backward");
    }
  }
  VTGS_HIP(hipGetLastError());
  {
    ProfScope ps__(dual ? "gather_splat_grads_dual" : "gather_splat_grads", st);
    if (b1)
      hipLaunchKernelGGL((gather_splat_grads<true, true, false, true>), dim3((n + 255) / 256), dim3(256), 0, st, cs, cam->viewmatrix, cam->projmatrix, n,
                         means3D, opacities, scales, rotations, (const GaussAux*)(ws + L.gaux), (const float*)scratch, 1,
                         (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                         (const Counters*)(ws + L.counters), (float*)nullptr, *frame);
    else if (dual && frame)
      hipLaunchKernelGGL((gather_splat_grads<true, true, false, false>), dim3((n + 255) / 256), dim3(256), 0, st, cs, cam->viewmatrix, cam->projmatrix, n,
                         means3D, opacities, scales, rotations, (const GaussAux*)(ws + L.gaux), (const float*)scratch, 1,
                         (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                         (const Counters*)(ws + L.counters), (float*)nullptr, *frame);
    else if (dual)
      hipLaunchKernelGGL((gather_splat_grads<true, false, false, false>), dim3((n + 255) / 256), dim3(256), 0, st, cs, cam->viewmatrix, cam->projmatrix, n,
                         means3D, opacities, scales, rotations, (const GaussAux*)(ws + L.gaux), (const float*)scratch, 1,
                         g_means3D, g_means2D, g_colors, g_opacities, g_scales, g_rotations,
                         (const Counters*)(ws + L.counters), g_colors_b, FrameEpilogue{});
    else if (cov3d)
      hipLaunchKernelGGL((gather_splat_grads<false, false, true, false>), dim3((n + 255) / 256), dim3(256), 0, st, cs, cam->viewmatrix, cam->projmatrix, n,
                         means3D, opacities, scales, rotations, (const GaussAux*)(ws + L.gaux), (const float*)scratch, bwd_impl != 0 ? 1 : 0,
                         g_means3D, g_means2D, g_colors, g_opacities, g_scales, (float*)nullptr,
                         (const Counters*)(ws + L.counters), (float*)nullptr, FrameEpilogue{});
    else
      hipLaunchKernelGGL((gather_splat_grads<false, false, false, false>), dim3((n + 255) / 256), dim3(256), 0, st, cs, cam->viewmatrix, cam->projmatrix, n,
                         means3D, opacities, scales, rotations, (const GaussAux*)(ws + L.gaux), (const float*)scratch, bwd_impl != 0 ? 1 : 0,
                         g_means3D, g_means2D, g_colors, g_opacities, g_scales, g_rotations,
                         (const Counters*)(ws + L.counters), (float*)nullptr, FrameEpilogue{});
  }
  VTGS_HIP(hipGetLastError());
  return VTGS_OK;
}

int vtgs_backward(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors, const float* opacities,
                  const float* scales, const float* rotations, const float* out_color, const float* grad_color,
                  const void* workspace, size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity,
                  const float* image_state, void* scratch, size_t scratch_bytes, float* g_means3D, float* g_means2D,
                  float* g_colors, float* g_opacities, float* g_scales, float* g_rotations, void* stream) {
  return backward_impl(cam, n, means3D, colors, nullptr, opacities, scales, rotations, out_color, nullptr, grad_color, nullptr,
                       workspace, workspace_bytes, instance_capacity, tile_capacity, image_state, scratch, scratch_bytes,
                       g_means3D, g_means2D, g_colors, nullptr, g_opacities, g_scales, g_rotations, stream, false);
}

int vtgs_forward_cov3d(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors, const float* opacities,
                       const float* cov3D, float* out_color, float* out_depth, int32_t* out_radii, void* workspace,
                       size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity, VtgsForwardInfo* info,
                       uint32_t flags, void* stream) {
  return forward_impl(cam, n, means3D, colors, nullptr, opacities, cov3D, nullptr, out_color, out_depth, nullptr, out_radii,
                      workspace, workspace_bytes, instance_capacity, tile_capacity, info, flags, stream, false, nullptr, true);
}

int vtgs_backward_cov3d(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors, const float* opacities,
                        const float* cov3D, const float* out_color, const float* grad_color, const void* workspace,
                        size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity, void* scratch,
                        size_t scratch_bytes, float* g_means3D, float* g_means2D, float* g_colors, float* g_opacities,
                        float* g_cov3D, void* stream) {
  return backward_impl(cam, n, means3D, colors, nullptr, opacities, cov3D, nullptr, out_color, nullptr, grad_color, nullptr,
                       workspace, workspace_bytes, instance_capacity, tile_capacity, nullptr, scratch, scratch_bytes,
                       g_means3D, g_means2D, g_colors, nullptr, g_opacities, g_cov3D, nullptr, stream, false, nullptr, true);
}

int vtgs_backward_dual(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors_a, const float* colors_b,
                       const float* opacities, const float* scales, const float* rotations, const float* out_color_a,
                       const float* out_color_b, const float* grad_color_a, const float* grad_color_b,
                       const void* workspace, size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity,
                       void* scratch, size_t scratch_bytes, float* g_means3D, float* g_means2D, float* g_colors_a,
                       float* g_colors_b, float* g_opacities, float* g_scales, float* g_rotations, void* stream) {
  return backward_impl(cam, n, means3D, colors_a, colors_b, opacities, scales, rotations, out_color_a, out_color_b,
                       grad_color_a, grad_color_b, workspace, workspace_bytes, instance_capacity, tile_capacity, nullptr,
                       scratch, scratch_bytes, g_means3D, g_means2D, g_colors_a, g_colors_b, g_opacities, g_scales,
                       g_rotations, stream, true);
}

int vtgs_backward_dual_frame(const VtgsCamera* cam, int32_t n, const float* means_cam, const float* colors_a, const float* colors_b,
                             const float* opacities, const float* scales, const float* rotations, const float* out_color_a,
                             const float* out_color_b, const float* grad_color_a, const float* grad_color_b,
                             const void* workspace, size_t workspace_bytes, uint64_t instance_capacity,
                             uint32_t tile_capacity, void* scratch, size_t scratch_bytes, uint32_t flags,
                             const float* means3D, const float* unnorm_rotations, const float* cam_q, const float* cam_t,
                             const float* depth_w2c, float* g_rgb_colors, float* g_means3D, float* g_logit_opacities,
                             float* g_log_scales, float* g_unnorm_rotations, float* pose_partials, void* stream) {
  return vtgs_backward_dual_frame_owned(cam, n, nullptr, means_cam, colors_a, colors_b, opacities, scales, rotations, out_color_a,
                                        out_color_b, grad_color_a, grad_color_b, workspace, workspace_bytes, instance_capacity,
                                        tile_capacity, scratch, scratch_bytes, flags, means3D, unnorm_rotations, cam_q, cam_t,
                                        depth_w2c, g_rgb_colors, g_means3D, g_logit_opacities, g_log_scales, g_unnorm_rotations,
                                        pose_partials, stream);
}

int vtgs_backward_dual_frame_owned(const VtgsCamera* cam, int32_t n, const int32_t* owned_idx, const float* means_cam,
                                   const float* colors_a, const float* colors_b, const float* opacities, const float* scales,
                                   const float* rotations, const float* out_color_a, const float* out_color_b,
                                   const float* grad_color_a, const float* grad_color_b, const void* workspace,
                                   size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity, void* scratch,
                                   size_t scratch_bytes, uint32_t flags, const float* means3D, const float* unnorm_rotations,
                                   const float* cam_q, const float* cam_t, const float* depth_w2c, float* g_rgb_colors,
                                   float* g_means3D, float* g_logit_opacities, float* g_log_scales, float* g_unnorm_rotations,
                                   float* pose_partials, void* stream) {
  FrameEpilogue f;
  f.flags = flags;
  f.means3D_world = means3D; f.unnorm_rot = unnorm_rotations; f.cam_q = cam_q; f.cam_t = cam_t; f.depth_w2c = depth_w2c;
  f.g_rgb = g_rgb_colors; f.g_means3D = g_means3D; f.g_logit = g_logit_opacities; f.g_log_scales = g_log_scales;
  f.g_unnorm_rot = g_unnorm_rotations; f.pose_partials = pose_partials;
  f.idx = owned_idx;
  return backward_impl(cam, n, means_cam, colors_a, colors_b, opacities, scales, rotations, out_color_a, out_color_b,
                       grad_color_a, grad_color_b, workspace, workspace_bytes, instance_capacity, tile_capacity, nullptr,
                       scratch, scratch_bytes, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream, true, &f);
}

int vtgs_band_owner_mask(const VtgsCamera* cam, int32_t n, const float* means3D, const float* scales, int32_t scales_are_log,
                         const float* cam_q, const float* cam_t, float margin_px, float growth, const uint8_t* owned,
                         uint8_t* mask_out, uint32_t* escapes, int32_t* centre_rows, void* stream) {
  if (!cam_ok(cam) || n < 0 || (cam_q == nullptr) != (cam_t == nullptr) || !(margin_px >= 0.f) || !(growth >= 1.f) ||
      (!mask_out && !escapes && !centre_rows) || (escapes && !owned))
    return VTGS_ERR_INVALID_ARGUMENT;
  int r8b, r8e, rows16, row16_0;
  if (!band_of(cam, &r8b, &r8e, &rows16, &row16_0)) return VTGS_ERR_INVALID_ARGUMENT;
  if (n == 0) return VTGS_OK;
  if (!means3D || !scales) return VTGS_ERR_INVALID_ARGUMENT;
  const CamScalars cs = scalars_of(cam, r8b, r8e);
  ProfScope ps__("band_owner_mask", (hipStream_t)stream);
  hipLaunchKernelGGL(band_owner_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, cs, cam->viewmatrix,
                     cam->projmatrix, n, means3D, scales, scales_are_log ? 1 : 0, cam_q, cam_t, margin_px, growth, owned, mask_out,
                     escapes, centre_rows);
  VTGS_HIP(hipGetLastError());
  return VTGS_OK;
}

int vtgs_profile_enable(int on) {
  g_prof_on = on != 0;
  g_prof_n = 0;
  return VTGS_OK;
}

int vtgs_profile_collect(VtgsProfileEntry* out, int32_t max_entries, int32_t* n_entries) {
  if (!out || !n_entries || max_entries <= 0) return VTGS_ERR_INVALID_ARGUMENT;
  VTGS_HIP(hipDeviceSynchronize());
  int n = 0;
  for (int i = 0; i < g_prof_n; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_prof[i].a, g_prof[i].b) != hipSuccess) continue;
    int k = 0;
    for (; k < n; ++k) if (strncmp(out[k].name, g_prof[i].name, sizeof(out[k].name)) == 0) break;
    if (k == n) {
      if (n == max_entries) continue;
      memset(&out[n], 0, sizeof(out[n]));
      strncpy(out[n].name, g_prof[i].name, sizeof(out[n].name) - 1);
      ++n;
    }
    out[k].total_ms += ms;
    out[k].launches += 1;
  }
  *n_entries = n;
  g_prof_n = 0;
  return VTGS_OK;
}

int vtgs_debug_layout(int32_t n, int32_t width, int32_t height, uint64_t instance_capacity, uint32_t tile_capacity,
                      uint64_t out[12]) {
  if (n < 0 || width <= 0 || height <= 0 || !out || (tile_capacity & ~VTGS_TILE_CAPACITY_PLANNED) == 0) return VTGS_ERR_INVALID_ARGUMENT;
  const WsLayout L = make_layout(n, width, height, instance_capacity, tile_capacity);
  out[0] = L.counters; out[1] = L.geom; out[2] = L.gaux; out[3] = L.tile_cnt; out[4] = L.sorted_gid;
  out[5] = L.sorted_inst; out[6] = L.final_T; out[7] = L.tiles8; out[8] = L.qmask; out[9] = L.dbg;
  out[10] = L.plan; out[11] = (uint64_t)L.tiles8 * L.tile_cap;
  return VTGS_OK;
}

int vtgs_mark_visible(const VtgsCamera* cam, int32_t n, const float* means3D, uint8_t* out_visible, void* stream) {
  if (!cam || !cam->viewmatrix || n < 0 || (n > 0 && (!means3D || !out_visible))) return VTGS_ERR_INVALID_ARGUMENT;
  if (n == 0) return VTGS_OK;
  hipLaunchKernelGGL(mark_visible_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, cam->viewmatrix, n,
                     means3D, out_visible);
  VTGS_HIP(hipGetLastError());
  return VTGS_OK;
}

}  // extern "C"
