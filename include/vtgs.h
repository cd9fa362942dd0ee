/* vtgs.h -- C ABI of the MI355X-native differentiable Gaussian-splat rasterizer (libvtgs.so).
 *
 * Drop-in boundary.  The reference reaches its rasterizer through the Python module
 * `diff_gaussian_rasterization` (src/vtgaussian_slam.py:38, utils/recon_helpers.py:2), whose
 * autograd.Function binds three native entry points of an un-vendored CUDA extension
 * (requirements.txt:19): `_C.rasterize_gaussians`, `_C.rasterize_gaussians_backward`, `_C.mark_visible`.
 * Those pybind/torch-typed entry points are what this header replaces: plain pointers, sizes and a
 * HIP stream, no torch types, no C++ exceptions.  Call sites served:
 *   src/vtgaussian_slam.py:461,466,747      GaussianRasterizer(raster_settings=cam)(**rendervar)
 *   utils/eval_helpers.py:240,247,431,443,728,733   forward-only renders
 *   utils/recon_helpers.py:14-26            the 11-field settings record (-> VtgsCamera)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; all arrays are contiguous row-major
 *     float32 (radii int32); N = number of Gaussians; images are CHW.
 *   - the caller owns all memory: inputs, outputs, the forward workspace (kept alive until backward,
 *     like the geom/binning/image buffers the replaced extension hands back to Python) and the
 *     backward scratch.  The library never allocates device memory, never frees, keeps no pointer
 *     after a call returns, and enqueues everything on `stream`.
 *   - outputs are fully overwritten (gradient arrays are written for every row, zeros included).
 *   - host waits: vtgs_forward with VTGS_FORWARD_SYNC waits for the whole forward; with VTGS_FORWARD_CHECKED it waits
 *     only for the result record, which the device writes right after the binning (the sort and the composite are
 *     still running when the call returns); VTGS_FORWARD_ASYNC and vtgs_backward never wait.
 *   - all functions return a VtgsStatus; vtgs_strerror() gives a static message.
 */
#ifndef VTGS_H
#define VTGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTGS_ABI_VERSION 16

typedef enum VtgsStatus {
  VTGS_OK = 0,
  VTGS_ERR_INVALID_ARGUMENT = 1,   /* null pointer, negative size, bad band, bad rule            */
  VTGS_ERR_WORKSPACE_TOO_SMALL = 2,/* workspace_bytes < vtgs_workspace_bytes(...)                */
  VTGS_ERR_INSTANCE_OVERFLOW = 3,  /* more (Gaussian,tile) instances than instance_capacity, or a    */
                                   /* tile list longer than tile_capacity; VtgsForwardInfo says how  */
                                   /* many (instances_needed, max_tile_list); re-allocate and call   */
                                   /* again (the image of the failed call is the background colour,  */
                                   /* its radii are valid, its lists are not)                         */
  VTGS_ERR_HIP = 4,                /* a HIP runtime call failed (see vtgs_last_hip_error)          */
  VTGS_ERR_STALE_WORKSPACE = 5     /* reserved                                                     */
} VtgsStatus;

/* radius_rule: how the screen-space extent of a splat is derived from its 2-D covariance.      */
#define VTGS_RADIUS_3SIGMA  0      /* ceil(3 sqrt(lambda_max)) -- the published rule (default)  */
#define VTGS_RADIUS_OPACITY 1      /* min(3 sigma, ceil(sqrt(2 ln(255 o) lambda_max)))          */

/* Mirrors GaussianRasterizationSettings (utils/recon_helpers.py:14-26).  `sh_degree`, `campos`
 * and `prefiltered` of that record do not influence a colors_precomp render and are not passed.   */
typedef struct VtgsCamera {
  int32_t image_width;
  int32_t image_height;
  float   tanfovx;
  float   tanfovy;
  float   scale_modifier;
  int32_t radius_rule;       /* VTGS_RADIUS_*                                                       */
  int32_t tile_row_begin;    /* band of 16-pixel tile rows this call renders (tile-row multi-GPU     */
  int32_t tile_row_end;      /* partition); begin == end == 0 means the whole image                  */
  const float* bg;           /* [3]  background colour                                               */
  const float* viewmatrix;   /* [16] w2c transposed, as stored at utils/recon_helpers.py:8           */
  const float* projmatrix;   /* [16] full projection, as stored at utils/recon_helpers.py:13         */
} VtgsCamera;

/* Result record of a forward.  Assembled on the device and copied to `info` (host). */
typedef struct VtgsForwardInfo {
  uint64_t instances;        /* (Gaussian, 8x8 tile) instances binned by this call = entries of all   */
                             /* tile lists (0 on overflow)                                            */
  uint64_t instances_needed; /* instance IDS handed out: what instance_capacity and the backward's    */
                             /* scratch must hold.  >= instances: a splat of more than nine candidate */
                             /* tiles holds one id per CANDIDATE tile, and the ids of the candidates  */
                             /* it does not reach stay unused (ABI 16).  On overflow: what did not fit */
  uint64_t tiles16_touched;  /* R of SURVEY 8(d): sum over Gaussians of 16x16 tiles in their rect    */
  uint32_t visible;          /* Gaussians with radii > 0                                             */
  uint32_t max_tile_list;    /* longest per-tile list                                                */
  uint32_t overflow;         /* bit 0: instance_capacity too small; bit 1: tile_capacity too small.      */
                             /* Non-zero: the outputs are invalid                                        */
  uint32_t complete;         /* 1 once the record has been written (async mode: poll / wait on it)   */
  uint64_t bin_slots_needed; /* bin slots in total if every bin were sized to its own list (planned bins,   */
                             /* below): compare with tiles x tile_capacity to see what uniform bins waste   */
} VtgsForwardInfo;

/* ---- Planned bins ------------------------------------------------------------------------------------------------
 * By default every 8x8 tile owns a bin of `tile_capacity` slots (21 bytes each): one pass, no scan -- but ONE dense
 * tile (a pile of Gaussians along one ray) sizes every bin.  With planned bins the bins follow a PLAN: a persistent,
 * caller-owned device array of vtgs_bin_plan_entries(width, height) offsets (bin t = [plan[t], plan[t+1])), which every
 * forward rewrites from its own list lengths (half again as much + 16 per bin) for the NEXT forward of that view -- the
 * hints a SLAM loop needs live on the device, per tile.  A forward whose lists outgrow the plan it was given reports
 * overflow bit 1 as usual; the plan has then already been rewritten from the exact lengths, so the caller repeats the
 * call (with a workspace of at least info.bin_slots_needed slots) and it fits.
 *   vtgs_bin_plan_uniform   fills a plan with uniform bins (the first forward of a view).
 *   vtgs_forward_planned / vtgs_forward_dual_planned   = vtgs_forward / vtgs_forward_dual with a plan.  Pass
 *     tile_capacity = VTGS_TILE_CAPACITY_PLANNED | c  HERE AND TO EVERY CALL THAT TAKES THE WORKSPACE (backward, shared
 *     render, workspace_bytes, debug_layout): the workspace then holds tiles x c slots in total, plan[tiles] must not
 *     exceed that, and the forward keeps its own copy of the plan in the workspace for its backward.
 * Planned bins cost one more dependent load per tile in every kernel that walks a list, and a pass of the sort kernel for
 * the lists beyond 1,024 entries ahead of the forward composite (which sorts the others itself, as with uniform bins):
 * they are for skewed scenes, the default stays uniform.                                                              */
#define VTGS_TILE_CAPACITY_PLANNED 0x80000000u
uint32_t vtgs_bin_plan_entries(int32_t width, int32_t height);
int vtgs_bin_plan_uniform(int32_t width, int32_t height, uint32_t slots_per_bin, uint32_t* bin_plan, void* stream);

/* vtgs_forward flags */
#define VTGS_FORWARD_SYNC  0u   /* wait for the forward, fill `info`, return VTGS_ERR_INSTANCE_OVERFLOW on overflow  */
#define VTGS_FORWARD_ASYNC 1u   /* do not synchronise: `info` must be PINNED host memory that stays valid until the  */
                                /* stream reaches the copy; the caller synchronises (event / stream) before reading */
                                /* it and must treat info->overflow == 1 as VTGS_ERR_INSTANCE_OVERFLOW               */
#define VTGS_FORWARD_CHECKED 2u /* enqueue everything, then wait on the HOST for the result record only: `info` must */
                                /* be pinned, device-mapped host memory (hipHostMalloc / a pinned torch tensor).  The */
                                /* record is written by the one-workgroup kernel that follows the binning, i.e. about */
                                /* a quarter into the forward; on overflow the remaining kernels bail and the call    */
                                /* returns VTGS_ERR_INSTANCE_OVERFLOW like the synchronous mode -- the caller always  */
                                /* holds a valid image when the call returns VTGS_OK.  Falls back to a full wait when */
                                /* `info` is not device-mapped.                                                       */
#define VTGS_FORWARD_EXPECT_SHORT_LISTS 4u /* hint, OR-ed to one of the above: the caller expects no tile list beyond 512      */
                                /* entries (its last forward of this view had none).  With bins of 768..1024 slots the  */
                                /* pre-sort pass for long lists is then not launched; a list of 513..1024 entries that    */
                                /* turns up anyway is sorted by the composite itself (slower, same result).  Ignored    */
                                /* for larger and for planned bins.                                                      */
#define VTGS_FORWARD_SECOND_IS_DEPTH 8u /* vtgs_forward_dual*, OR-ed to one of the modes: a PROMISE about colors_b and about how  */
                                /* out_color_b is consumed.  colors_b is the fused caller chain's [z, 1, z^2]                 */
                                /* (utils/slam_helpers.py transformed_params2depthplussilhouette) and the caller uses the     */
                                /* second image as get_loss does (src/vtgaussian_slam.py:466-521): plane 0 (depth) in full,   */
                                /* plane 1 (silhouette) in comparisons only, plane 2 only through isnan(plane 2 - plane 0^2). */
                                /* Then plane 0 = sum w z as always, plane 1 = 1 - T_final (what sum w is, up to float32     */
                                /* rounding) and plane 2 = plane 0 squared (NaN exactly where the true difference is) come    */
                                /* from the single render's kernel with z in its depth column.  VTGS_DEPTH_LITE=0 /           */
                                /* vtgs_set_option ignores the bit.                                                           */
#define VTGS_FORWARD_RAW_ACTIVATIONS 16u /* vtgs_forward_dual*, OR-ed to one of the modes (round 6): the map is ISOTROPIC and its    */
                                /* activations are applied by the projection itself -- `opacities` holds the LOGITS [N],           */
                                /* `scales` the LOG-scales [N] (one per Gaussian: exp(.) on all three axes), `rotations` is     */
                                /* not read (may be NULL: the covariance of an isotropic Gaussian does not depend on it).  What  */
                                /* vtgs_prepare_frame_slot then still has to write is means_cam and the depth colours          */
                                /* (its other three outputs NULL): 36 instead of 92 bytes per Gaussian and iteration.          */
                                /* The matching backward is vtgs_backward_dual_frame with flags bit 4 (16).                     */
#define VTGS_FORWARD_WORKSPACE_CLEARED 32u /* OR-ed to one of the modes: the caller has ZEROED the first vtgs_workspace_clear_bytes()   */
                                /* bytes of `workspace` (the counters block and the per-tile list lengths) on this stream --    */
                                /* vtgs_prepare_frame_slot does it in its own launch -- so the forward issues no fill command   */
                                /* (~5 us of a SLAM iteration: every command costs the queue that much).                        */
#define VTGS_FORWARD_EXPECT_NO_DEFERRED 64u /* hint, OR-ed to one of the modes: the caller expects no splat beyond nine candidate      */
                                /* 8x8 tiles (its last forward of this view handed out exactly as many ids as it binned           */
                                /* instances: instances_needed == instances).  The projection then bins every splat itself --      */
                                /* a lane up to 64 candidates, its wavefront beyond -- and the second kernel (an empty launch   */
                                /* on such a map: ~5 us of queue time) is not launched.  Any map renders correctly with the       */
                                /* hint, a heavy-tailed one slowly; a workgroup that meets such a splat takes ONE     */
                                /* unused id, so instances_needed > instances tells the caller to drop the hint.  Whole frame,     */
                                /* uniform bins only (ignored for a band, planned bins, cov3D).                                    */
#define VTGS_FORWARD_MODE_MASK 3u

uint32_t    vtgs_abi_version(void);
/* Implementation switches (tests and ablations): "VTGS_FWD_IMPL" (3 = per-quadrant splat queues, default) and
 * "VTGS_FWD_IMPL" / "VTGS_BWD_IMPL" (2 = lane-per-pixel matrix-core composites, the backward's default; 1 = pixel x
 * splat-quad form; 0 = scalar kernels), "VTGS_BIN_IMPL" (1 = LDS-binned slot
 * reservation where the tile table fits, 0 = global atomics), "VTGS_SORT_PACKED" (1 = payload in the key's low bits for
 * N <= 2^21), "VTGS_SORT_FUSED" (1 = the quadrant-queue forward sorts its own tile's list -- lists beyond 512 entries are pre-sorted
 * by a kernel ahead of it where bins of that size exist -- and does the finalize step; 0 = separate finalize and sort launches), "VTGS_SORT_LONG_COUNTING" (1 = lists of 513 .. 2048 entries by the workgroup counting sort, 0 = by the
 * LDS network: cross-check), "VTGS_COUNT_STEPS" (1 = the quadrant-queue forward counts its steps, see vtgs_debug_layout;
 * measurement only).  Defaults come from the environment variables of the same names, read ONCE at first use.
 * vtgs_set_option returns VTGS_ERR_INVALID_ARGUMENT for an unknown name; value < 0 restores the default.           */
int         vtgs_set_option(const char* name, int value);
int         vtgs_get_option(const char* name);   /* current value, or -1 for an unknown name */
const char* vtgs_strerror(int status);
const char* vtgs_last_hip_error(void);   /* message of the last failed HIP call on this host thread */

/* Bytes of forward workspace for N Gaussians, a width x height image, room for `instance_capacity`
 * (Gaussian,tile) instances in total and for `tile_capacity` instances per 8x8 tile (every tile owns a
 * fixed-capacity bin, so binning needs no prefix scan).  Safe first guesses: 8*N and 512; vtgs_forward
 * reports the exact needs (instances_needed, max_tile_list) on overflow.                               */
size_t vtgs_workspace_bytes(int32_t n, int32_t width, int32_t height, uint64_t instance_capacity,
                            uint32_t tile_capacity);
/* The head of the workspace a forward needs zeroed before it runs (counters + per-tile list lengths): the forward clears it
 * itself unless the caller says VTGS_FORWARD_WORKSPACE_CLEARED. */
size_t vtgs_workspace_clear_bytes(int32_t image_width, int32_t image_height);

/* Bytes of backward scratch for a forward that binned `instances` instances.  When the count is not known
 * yet (asynchronous forward), pass its instance_capacity: the scratch is indexed by instance id < capacity.
 * If the forward overflowed, the backward kernels see the device-side flag and write nothing.            */
size_t vtgs_backward_scratch_bytes(int32_t n, uint64_t instances);

/* EMPTY INPUT (n = 0) -- one rule for every entry point below.  A rank of the tile-row partition may own nothing, a map may
 * be empty before its first frame: with n = 0 every PER-GAUSSIAN pointer (inputs, radii, per-Gaussian gradient outputs, index
 * lists) may be NULL.  Everything that is not per-Gaussian keeps its requirements: a forward still needs its output images, its
 * workspace and -- in the asynchronous / checked modes -- its record; it composites every tile as empty (background colour,
 * depth 0, final transmittance 1) and reports zero instances.  A backward with n = 0 touches nothing and returns VTGS_OK; the
 * pose-gradient helpers reduce zero rows to zeros.  No entry point infers a render VARIANT (single / dual, frame epilogue,
 * owned set) from a per-Gaussian pointer: variants follow from which entry point was called and from the image / flag
 * arguments (tests/test_gpu_empty_inputs.py calls all of them with n = 0 and NULL arrays on poisoned memory).             */

/* Forward.  Replaces `_C.rasterize_gaussians` for the colors_precomp + scales/rotations signature
 * the reference uses (cov3D_precomp: vtgs_forward_cov3d; shs: vtgs_sh_forward in front of this call).
 *   means3D[N,3] opacities[N,1] colors[N,3] scales[N,3] rotations[N,4](w,x,y,z)
 *   out_color[3,H,W] out_depth[1,H,W] out_radii[N] (0 = culled; in band mode also 0 for a Gaussian that cannot
 *   meet the band's rows -- it is skipped before it is projected -- so the radii of a partitioned frame are the
 *   element-wise MAXIMUM over the ranks, and info.visible / tiles16_touched count what this band looked at)
 * `info`: host record (may be NULL with VTGS_FORWARD_SYNC); `flags`: VTGS_FORWARD_*.                    */
int vtgs_forward(const VtgsCamera* cam, int32_t n,
                 const float* means3D, const float* colors, const float* opacities,
                 const float* scales, const float* rotations,
                 float* out_color, float* out_depth, int32_t* out_radii,
                 void* workspace, size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity,
                 VtgsForwardInfo* info, uint32_t flags, void* stream);

/* Second render over the SAME geometry (identical cam/means3D/opacities/scales/rotations as the
 * vtgs_forward that filled `workspace`) with other per-Gaussian colours -- the depth/silhouette pass
 * at src/vtgaussian_slam.py:466 right after the RGB pass at :461.  Skips projection, binning and
 * sorting; writes its per-pixel state to `image_state` ([H*W] floats, caller-owned) so both renders
 * can be differentiated.                                                                             */
int vtgs_forward_shared(const VtgsCamera* cam, int32_t n, const float* colors,
                        float* out_color, float* out_depth,
                        const void* workspace, size_t workspace_bytes, uint64_t instance_capacity,
                        uint32_t tile_capacity, float* image_state, void* stream);

/* Backward.  Replaces `_C.rasterize_gaussians_backward`.
 *   grad_color[3,H,W] = dL/d out_color;  out_color = what the forward wrote (the gradient of the
 *   depth image is not propagated -- the reference discards that output, src/vtgaussian_slam.py:461).
 *   workspace / instance_capacity / tile_capacity: exactly what the forward was given (they fix the layout).
 *   image_state: NULL to use the state stored in the workspace by vtgs_forward, or the buffer a
 *   vtgs_forward_shared call filled.
 *   scratch: vtgs_backward_scratch_bytes(n, info.instances_needed) bytes (records are addressed by instance id), contents
 *            undefined on entry and exit.
 *   In band mode (tile_row_begin/end) the gradients are this band's partial sums; pixels outside the
 *   band are written as zero by the forward and ignored by the backward.
 *   g_means2D[N,3] holds the NDC-scaled screen-space gradient in [:, :2] (what means2D.grad shows at
 *   utils/slam_external.py:100-103), zeros in [:, 2].
 *   Any of the six outputs may be NULL (not all): that gradient is then not stored -- the tracking loop
 *   detaches the Gaussians (src/vtgaussian_slam.py:428-449) and needs 24 of the 68 bytes per Gaussian.
 *   The same holds for the seven outputs of vtgs_backward_dual.                                       */
int vtgs_backward(const VtgsCamera* cam, int32_t n,
                  const float* means3D, const float* colors, const float* opacities,
                  const float* scales, const float* rotations,
                  const float* out_color, const float* grad_color,
                  const void* workspace, size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity,
                  const float* image_state, void* scratch, size_t scratch_bytes,
                  float* g_means3D, float* g_means2D, float* g_colors, float* g_opacities,
                  float* g_scales, float* g_rotations, void* stream);

/* cov3D_precomp form of the operator (the other half of its both-or-neither check): the 3-D covariance of every
 * Gaussian is given as six floats (xx xy xz yy yz zz, used as they are: no scale modifier) instead of scales + rotations, and
 * the backward returns dL/dcov3D[N,6] (an off-diagonal entry fills two places of the symmetric matrix).  The reference never
 * passes it (utils/slam_helpers.py:152-159); single render, uniform bins (tile_capacity without VTGS_TILE_CAPACITY_PLANNED),
 * whole frame or a band.  Everything else as vtgs_forward / vtgs_backward (any gradient output may be NULL, not all).          */
int vtgs_forward_cov3d(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors, const float* opacities,
                       const float* cov3D, float* out_color, float* out_depth, int32_t* out_radii, void* workspace,
                       size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity, VtgsForwardInfo* info,
                       uint32_t flags, void* stream);
int vtgs_backward_cov3d(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors, const float* opacities,
                        const float* cov3D, const float* out_color, const float* grad_color, const void* workspace,
                        size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity, void* scratch,
                        size_t scratch_bytes, float* g_means3D, float* g_means2D, float* g_colors, float* g_opacities,
                        float* g_cov3D, void* stream);

/* Spherical-harmonics colours: the operator's `shs` argument (the other half of its colours-or-SHs check; the reference only
 * passes colours, sh_degree = 0 at utils/recon_helpers.py:22).  Per-Gaussian pre- and post-op around the rasterizer, no kernel
 * of which changes: colours[n,3] = clamp0(sum_k Y_k(dir) shs[n,k,:] + 0.5), dir = (mean - campos)/|mean - campos|, real SH basis
 * of `degree` 0..3 ((degree+1)^2 <= coeffs <= 16 coefficients per channel, shs laid out [n, coeffs, 3]); out_clamped[n] holds one
 * bit per channel that was negative.  The backward turns dL/dcolours into dL/dshs[n,coeffs,3] (0 beyond the active degree) and
 * the viewing direction's share of dL/dmeans3D[n,3] (either output may be NULL, not both).                                    */
int vtgs_sh_forward(int32_t n, int32_t degree, int32_t coeffs, const float* means3D, const float* campos, const float* shs,
                    float* out_colors, uint8_t* out_clamped, void* stream);
int vtgs_sh_backward(int32_t n, int32_t degree, int32_t coeffs, const float* means3D, const float* campos, const float* shs,
                     const uint8_t* clamped, const float* g_colors, float* g_shs, float* g_means3D, void* stream);

/* ---- Dual render ("6 channels") -------------------------------------------------------------------
 * The two back-to-back renders of get_loss (src/vtgaussian_slam.py:461 RGB, :466 [z,1,z^2]) as ONE composite pass:
 * same cam / means3D / opacities / scales / rotations, two colour sets.  Every exponent, alpha and transmittance is
 * computed once; out_color_a / out_color_b are bit-identical to vtgs_forward + vtgs_forward_shared.  No depth image
 * (the reference discards it at both call sites).  vtgs_backward_dual differentiates both images at once:
 * grad_color_a/b in, ONE set of geometry gradients (= the sum over the two renders, as autograd would accumulate
 * it) plus g_colors_a / g_colors_b out.  Scratch: vtgs_backward_dual_scratch_bytes (56-byte records).            */
size_t vtgs_backward_dual_scratch_bytes(int32_t n, uint64_t instances);
int vtgs_forward_dual(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors_a, const float* colors_b,
                      const float* opacities, const float* scales, const float* rotations, float* out_color_a,
                      float* out_color_b, int32_t* out_radii, void* workspace, size_t workspace_bytes,
                      uint64_t instance_capacity, uint32_t tile_capacity, VtgsForwardInfo* info, uint32_t flags,
                      void* stream);
int vtgs_forward_planned(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors, const float* opacities,
                         const float* scales, const float* rotations, float* out_color, float* out_depth, int32_t* out_radii,
                         void* workspace, size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity,
                         uint32_t* bin_plan, VtgsForwardInfo* info, uint32_t flags, void* stream);
int vtgs_forward_dual_planned(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors_a,
                              const float* colors_b, const float* opacities, const float* scales, const float* rotations,
                              float* out_color_a, float* out_color_b, int32_t* out_radii, void* workspace,
                              size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity, uint32_t* bin_plan,
                              VtgsForwardInfo* info, uint32_t flags, void* stream);
int vtgs_backward_dual(const VtgsCamera* cam, int32_t n, const float* means3D, const float* colors_a, const float* colors_b,
                       const float* opacities, const float* scales, const float* rotations, const float* out_color_a,
                       const float* out_color_b, const float* grad_color_a, const float* grad_color_b,
                       const void* workspace, size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity,
                       void* scratch, size_t scratch_bytes, float* g_means3D, float* g_means2D, float* g_colors_a,
                       float* g_colors_b, float* g_opacities, float* g_scales, float* g_rotations, void* stream);

/* Replaces `_C.mark_visible` (GaussianRasterizer.markVisible; unused by the reference driver).
 * out_visible[N] bytes: 1 when the point passes the near-plane test of the forward.                 */
int vtgs_mark_visible(const VtgsCamera* cam, int32_t n, const float* means3D,
                      uint8_t* out_visible, void* stream);

/* ---- Fused caller chain, isotropic maps (log_scales [N,1], every reference config) -------------
 * vtgs_prepare_frame replaces the element-wise PyTorch launches of utils/slam_helpers.py:323-385 (transform_to_frame),
 * :127-160 (transformed_params2rendervar) and :217-287 (get_depth_and_silhouette / ...depthplussilhouette):
 *   out_means_cam = R(q/|q|) means3D + t        out_opacities = sigmoid(logit_opacities)
 *   out_scales    = exp(log_scales) tiled x3     out_rotations = normalize(unnorm_rotations)
 *   out_depth_colors = [z, 1, z^2], z = row 2 of depth_w2c (row-major 4x4, the first-frame camera) applied to means_cam
 * cam_q [4] (w,x,y,z, un-normalised), cam_t [3], depth_w2c [16] are DEVICE pointers (they are slices of parameters).
 * vtgs_prepare_frame_backward is its adjoint for two renders over the same geometry (a = RGB pass, b = depth/silhouette
 * pass): it takes the two sets of operator gradients plus dL/d(depth colours) and writes (flags bit 0) the gradients of
 * means3D / unnorm_rotations, (flags bit 2) those of logit_opacities / log_scales -- the reference's gaussians_grad=False
 * detaches only the former pair -- and (flags bit 1) per-workgroup partial sums of
 * [dL/dt (3) | dL/dR (9, row-major)] into pose_partials[vtgs_pose_partial_rows(n)][12] (16-byte aligned: the reduction reads
 * a row as three 16-byte words; otherwise VTGS_ERR_INVALID_ARGUMENT); the caller sums the rows and
 * takes the 12 -> 7 step through the quaternion.  No atomics: results are bitwise reproducible.  The four *_b inputs
 * may all be NULL (after vtgs_backward_dual the *_a set already holds the sum over both renders).                       */
uint32_t vtgs_pose_partial_rows(int32_t n);
/* Round 6 (ABI 16): the two slot helpers below folded into their neighbours, for callers that hold the reference's camera
 * tensors [1,4,T] / [1,3,T] (contiguous):
 *   vtgs_prepare_frame_slot   = vtgs_pose_slot_gather + vtgs_prepare_frame in ONE launch: the pose is read in place (column t)
 *                               and out_pose7 (device, 7 floats: q, t) receives it contiguous for the backward's entry points;
 *                               out_opacities, out_scales and out_rotations may be NULL TOGETHER (the caller renders with
 *                               VTGS_FORWARD_RAW_ACTIVATIONS): only means_cam and the depth colours are written;
 *                               clear / clear_bytes (may be NULL / 0; 16-byte aligned, a multiple of 16): that many bytes are
 *                               zeroed on the stream as well -- the head of the forward's workspace, see
 *                               VTGS_FORWARD_WORKSPACE_CLEARED and vtgs_workspace_clear_bytes;
 *   vtgs_pose_gradient_slot   = vtgs_pose_gradient + vtgs_pose_slot_scatter in ONE launch: writes the FULL-SIZE gradients
 *                               (4 T and 3 T floats, zero except column t); cam_q = the contiguous q of out_pose7.             */
int vtgs_prepare_frame_slot(int32_t n, const float* means3D, const float* logit_opacities, const float* log_scales,
                            const float* unnorm_rotations, const float* cam_unnorm_rots, const float* cam_trans, int32_t frames,
                            int32_t t, const float* depth_w2c, float* out_means_cam, float* out_opacities, float* out_scales,
                            float* out_rotations, float* out_depth_colors, float* out_pose7, void* clear, size_t clear_bytes, void* stream);
int vtgs_pose_gradient_slot(const float* pose_partials, uint32_t rows, const float* cam_q, int32_t frames, int32_t t,
                            float* g_cam_unnorm_rots, float* g_cam_trans, void* stream);
/* The pose of frame t out of the reference's camera tensors, cam_unnorm_rots [1,4,frames] and cam_trans [1,3,frames]
 * (src/vtgaussian_slam.py:160-167), as seven contiguous floats (q[4], t[3]); and its adjoint: full-size gradients that are zero
 * except column t.  One launch each -- tensor indexing costs ten (two strided copies, four zero-fills, four slice copies). */
int vtgs_pose_slot_gather(const float* cam_unnorm_rots, const float* cam_trans, int32_t frames, int32_t t, float* pose7, void* stream);
int vtgs_pose_slot_scatter(const float* g_q, const float* g_t, int32_t frames, int32_t t, float* g_cam_unnorm_rots, float* g_cam_trans,
                           void* stream);   /* g_q[4] / g_t[3]: either may be NULL (= zeros) */

/* Sums the partial rows and takes the 12 -> 7 step through the normalised quaternion: g_cam_q[4], g_cam_t[3] (device). */
int vtgs_pose_gradient(const float* pose_partials, uint32_t rows, const float* cam_q, float* g_cam_q, float* g_cam_t,
                       void* stream);
/* The 7-float pose reduction each rank of the tile-row partition all-reduces (bench.py --gpus N): for
 * points already in the camera frame (pose = identity), out7 = {sum g (3), sum p x g (3), sum g_z}; partials =
 * vtgs_pose_partial_rows(n) x 7 floats of scratch.  Two launches, fixed summation order.                            */
int vtgs_pose7_reduce(int32_t n, const float* points, const float* g_points, float* partials, float* out7, void* stream);
int vtgs_prepare_frame(int32_t n, const float* means3D, const float* logit_opacities, const float* log_scales,
                       const float* unnorm_rotations, const float* cam_q, const float* cam_t, const float* depth_w2c,
                       float* out_means_cam, float* out_opacities, float* out_scales, float* out_rotations,
                       float* out_depth_colors, void* stream);
int vtgs_prepare_frame_backward(int32_t n, uint32_t flags, const float* means3D, const float* logit_opacities,
                                const float* log_scales, const float* unnorm_rotations, const float* cam_q,
                                const float* cam_t, const float* depth_w2c, const float* g_means_a, const float* g_means_b,
                                const float* g_depth_colors, const float* g_opac_a, const float* g_opac_b,
                                const float* g_scales_a, const float* g_scales_b, const float* g_rot_a, const float* g_rot_b,
                                float* g_means3D, float* g_logit_opacities, float* g_log_scales, float* g_unnorm_rotations,
                                float* pose_partials, void* stream);

/* vtgs_backward_dual followed by vtgs_prepare_frame_backward, in ONE pass over the Gaussians: the adjoint of
 * vtgs_prepare_frame is applied to each Gaussian's operator gradients while they are still in registers (same formulas
 * and reduction order as vtgs_prepare_frame_backward; results agree to float32 rounding), so the six dense [N, .] gradient arrays
 * of vtgs_backward_dual are never written or read back.  Inputs as vtgs_backward_dual (means_cam, opacities, scales,
 * rotations, colors_b = the outputs of vtgs_prepare_frame) plus means3D / unnorm_rotations / cam_q / cam_t / depth_w2c
 * as given to vtgs_prepare_frame; flags as vtgs_prepare_frame_backward.  Outputs (each may be NULL when its flag is
 * clear): bit 0 g_means3D [N,3], g_unnorm_rotations [N,4]; bit 1 pose_partials [vtgs_pose_partial_rows(n)][12];
 * bit 2 g_rgb_colors [N,3], g_logit_opacities [N], g_log_scales [N].  No screen-space (means2D) gradient.
 * flags bit 3 (8) is a PROMISE about an input: grad_color_b is zero outside its first channel (get_loss differentiates the
 * [z, 1, z^2] render through z alone: the silhouette only feeds comparisons, z^2 a detached uncertainty,
 * src/vtgaussian_slam.py:466-521).  Planes 1 and 2 of grad_color_b are then not read: four image-gradient channels instead of
 * six, one colour of the second set per splat, 48-byte instead of 56-byte gradient records.  Results equal those without the
 * bit whenever the promise holds (tests/test_gpu_fused_frame.py).  VTGS_DUAL_B1=0 / vtgs_set_option ignores the bit.
 * flags bit 4 (16), round 6: the forward ran with VTGS_FORWARD_RAW_ACTIVATIONS -- `opacities` = logit_opacities [N], `scales` =
 * log_scales [N], `rotations` not read (may be NULL); the rotation of a Gaussian (unnorm_rotations, normalised) is read only when
 * bit 0 asks for its gradient.  Not with an owned list (owned_idx must be NULL: the compact arrays are prepared in full).     */
int vtgs_backward_dual_frame(const VtgsCamera* cam, int32_t n, const float* means_cam, const float* colors_a, const float* colors_b,
                             const float* opacities, const float* scales, const float* rotations, const float* out_color_a,
                             const float* out_color_b, const float* grad_color_a, const float* grad_color_b,
                             const void* workspace, size_t workspace_bytes, uint64_t instance_capacity,
                             uint32_t tile_capacity, void* scratch, size_t scratch_bytes, uint32_t flags,
                             const float* means3D, const float* unnorm_rotations, const float* cam_q, const float* cam_t,
                             const float* depth_w2c, float* g_rgb_colors, float* g_means3D, float* g_logit_opacities,
                             float* g_log_scales, float* g_unnorm_rotations, float* pose_partials, void* stream);

/* ---- Owned sets of the tile-row partition ---------------------------------------------------------
 * A rank of the partition renders a band of tile rows (VtgsCamera.tile_row_begin / _end); 7 of 8 Gaussians of the map
 * cannot meet it.  Instead of running the per-Gaussian kernels (vtgs_prepare_frame, projection and binning, the gradient
 * gather) over the whole map on every rank, the rank keeps a LIST of the Gaussians that could meet its rows for any pose /
 * scale near the current ones and hands the rasterizer the compact arrays of those (the reference has no counterpart: its
 * rasterizer is single-GPU, src/vtgaussian_slam.py:431-468 renders the whole map).
 *
 * vtgs_band_owner_mask: the band test of the projection kernel (mean + scale only) for all n Gaussians of the map -- with the
 *   radius widened by margin_px and the scale multiplied by growth (>= 1).  means3D are world points moved by the pose
 *   (cam_q [4] un-normalised, cam_t [3]; device pointers, as vtgs_prepare_frame takes them), or camera-frame points as the plain
 *   operator gets them (cam_q = cam_t = NULL).  scales: log_scales [n] of an isotropic map (scales_are_log = 1) or the
 *   operator's scales [n,3] (0).
 *   mask_out (may be NULL): 1 byte per Gaussian, 1 = could meet the band: the caller compacts the indices (ascending).
 *   escapes (may be NULL; needs owned = a mask written earlier): *escapes += the number of Gaussians with owned[i] == 0
 *   that could meet the band now.  Called with (margin 1 px, growth 1) before a render of the list it proves, while the
 *   counter stays 0, that the render equals the render of the whole map on the band: a Gaussian outside the list would have
 *   been dropped by the projection kernel's own band test.  The counter is the caller's (zeroed when the list is built).
 *   centre_rows (may be NULL): centre_rows[i] = the 16-pixel tile row of Gaussian i's projected centre, clamped to the image
 *   (-1 behind the near plane) -- the OWNER band of the Gaussian, the same on every rank (partition.OwnerExchange).
 * vtgs_prepare_frame_owned: vtgs_prepare_frame for the rows owned_idx[0..n_owned) of the map, written to compact rows
 *   0..n_owned; the colours are gathered too (out_rgb_colors [n_owned,3]) -- everything the rasterizer reads has n_owned rows.
 * vtgs_backward_dual_frame_owned: vtgs_backward_dual_frame over the compact arrays (n = n_owned); means3D,
 *   unnorm_rotations and the g_* outputs are the MAP's arrays, read / written at row owned_idx[i]; rows outside the list are
 *   not touched (the caller zero-fills them).  owned_idx == NULL is vtgs_backward_dual_frame.  pose_partials has
 *   vtgs_pose_partial_rows(n_owned) rows.                                                                              */
int vtgs_band_owner_mask(const VtgsCamera* cam, int32_t n, const float* means3D, const float* scales, int32_t scales_are_log,
                         const float* cam_q, const float* cam_t, float margin_px, float growth, const uint8_t* owned,
                         uint8_t* mask_out, uint32_t* escapes, int32_t* centre_rows, void* stream);
int vtgs_prepare_frame_owned(int32_t n_owned, const int32_t* owned_idx, const float* means3D, const float* logit_opacities,
                             const float* log_scales, const float* unnorm_rotations, const float* rgb_colors,
                             const float* cam_q, const float* cam_t, const float* depth_w2c, float* out_means_cam,
                             float* out_opacities, float* out_scales, float* out_rotations, float* out_depth_colors,
                             float* out_rgb_colors, void* stream);
int vtgs_backward_dual_frame_owned(const VtgsCamera* cam, int32_t n, const int32_t* owned_idx, const float* means_cam,
                                   const float* colors_a, const float* colors_b, const float* opacities, const float* scales,
                                   const float* rotations, const float* out_color_a, const float* out_color_b,
                                   const float* grad_color_a, const float* grad_color_b, const void* workspace,
                                   size_t workspace_bytes, uint64_t instance_capacity, uint32_t tile_capacity, void* scratch,
                                   size_t scratch_bytes, uint32_t flags, const float* means3D, const float* unnorm_rotations,
                                   const float* cam_q, const float* cam_t, const float* depth_w2c, float* g_rgb_colors,
                                   float* g_means3D, float* g_logit_opacities, float* g_log_scales, float* g_unnorm_rotations,
                                   float* pose_partials, void* stream);

/* ---- SSIM of the mapping loss ------------------------------------------------------------------
 * Replaces utils/slam_external.py:66-97 (calc_ssim): mean SSIM of two [C,H,W] images with the 11x11 Gaussian window
 * (sigma 1.5, zero padding, per-channel).  vtgs_ssim_forward writes one partial sum of the SSIM map per workgroup into
 * partial_sums[vtgs_ssim_partial_rows(C,H,W)] (mean = sum / (C*H*W)) and, if grad_maps is not NULL, three [C,H,W]
 * derivative maps for the backward.  vtgs_ssim_backward gives dL/dimg1 for dL/d(mean SSIM) = *upstream (device scalar).
 * img2 is treated as a constant (it is the ground-truth image in the reference).                                      */
uint32_t vtgs_ssim_partial_rows(int32_t channels, int32_t height, int32_t width);
int vtgs_ssim_forward(const float* img1, const float* img2, int32_t channels, int32_t height, int32_t width,
                      float* partial_sums, float* grad_maps, void* stream);
int vtgs_ssim_backward(const float* img1, const float* img2, const float* grad_maps, const float* upstream,
                       int32_t channels, int32_t height, int32_t width, float* grad_img1, void* stream);

/* Masked L1 terms of get_loss (src/vtgaussian_slam.py:519-608) with their gradient images, one pass over the pixels.
 * im, gt_im [3,P]; depth_sil [3,P] = the [z,1,z^2] render; gt_depth [P].  mode 0 = tracking (mask: gt_depth > 0, finite
 * depth and uncertainty, silhouette > sil_thres; colour and depth sums over the mask), mode 1 = mapping (depth over the
 * mask without the silhouette test, colour over all pixels), mode 2 = tracking with the colour sum over ALL pixels
 * (src/vtgaussian_slam.py:601-602: neither use_sil_for_loss nor ignore_outlier_depth_loss).  partial_sums[vtgs_masked_l1_partial_rows(P)][3] =
 * {sum |gt_im - im|, sum |gt_depth - depth|, mask count} per workgroup; g_im [3,P] and g_depth_sil [3,P] receive the
 * derivatives of the two sums (channels 1 and 2 of depth_sil only feed detached masks: zero).                          */
uint32_t vtgs_masked_l1_partial_rows(int32_t pixels);
int vtgs_masked_l1(const float* im, const float* depth_sil, const float* gt_im, const float* gt_depth, int32_t pixels,
                   float sil_thres, int32_t mode, float* partial_sums, float* g_im, float* g_depth_sil, void* stream);

/* Silhouette-threshold sweep of tracking iteration 0 (src/vtgaussian_slam.py:472-510): for each of n_thresholds <= 8
 * candidates c_k, over the pixels with silhouette > c_k and gt_depth > 0:  partial_sums[row][2k] = sum over the three
 * channels of (gt_im - im)^2, partial_sums[row][2k+1] = pixel count; rows = vtgs_masked_l1_partial_rows(P), row stride
 * 2*n_thresholds.  The masked MSE of candidate k is sum_k / (3 * count_k); the caller takes the arg-min like the
 * reference.  thresholds is a HOST array (copied into the launch).                                                    */
int vtgs_silhouette_sweep(const float* im, const float* silhouette, const float* gt_im, const float* gt_depth,
                          int32_t pixels, const float* thresholds, int32_t n_thresholds, float* partial_sums, void* stream);

/* The Replica branch of get_loss as a whole (src/vtgaussian_slam.py:519-608, 678-679).  mode 0 = tracking:
 * w_im * masked sum |gt_im - im| + w_depth * masked sum |gt_depth - depth| (mask: gt_depth > 0, finite depth and
 * uncertainty, silhouette > sil_thres).  mode 1 = mapping: w_im * (0.8 * mean |gt_im - im| + 0.2 * (1 - SSIM)) +
 * w_depth * masked mean |gt_depth - depth| (same mask without the silhouette test).  mode 2 = mode 0 with the colour sum
 * taken over all pixels (:601-602).  w_depth = 0 gives the loss of use_l1 = False (no depth term, :591-596).  Images are
 * [3,H,W] / [1,H,W].
 * forward: 2-3 launches; out8 (device, 8 floats) = {loss, mask count, sum |d im|, sum |d depth|, mean SSIM, the weighted
 * colour term, the weighted depth term, 0}; scratch =
 * vtgs_loss_scratch_floats(H, W) floats; ssim_grad_maps = 9*H*W floats (mode 1 with a backward to follow, else NULL).
 * backward: 1-2 launches writing g_im [3,H,W] and g_depth_sil [3,H,W] = upstream[0] * dloss/d(.) with `upstream` a
 * DEVICE scalar (the gradient arriving at the loss): no host wait, no element-wise multiplies afterwards.
 * The TUM / ScanNet / ScanNet++ branches (src/vtgaussian_slam.py:511-611) add detached per-pixel masks to the same sums:
 * extra_mask [H*W] floats (NULL = none; 0 = pixel excluded) is ANDed into the depth / tracking-colour mask -- the caller
 * forms it from the visibility mask (:536-584, :376-404), the far-depth filter (:586-588) and the 50 x median outlier
 * mask (:525-528); color_weight [3*H*W] (mapping only, NULL = none) = 10 * additional_mask + 0.8 replaces the constant
 * 0.8 of the colour L1 term (:609-611, l1_loss_v1_mask).                                                             */
size_t vtgs_loss_scratch_floats(int32_t height, int32_t width);
int vtgs_slam_loss_forward(int32_t mode, const float* im, const float* depth_sil, const float* gt_im, const float* gt_depth,
                           int32_t height, int32_t width, float sil_thres, float w_im, float w_depth, float* scratch,
                           float* ssim_grad_maps, float* out8, const float* extra_mask, const float* color_weight,
                           void* stream);
/* The same, with get_loss's bookkeeping of the render it belongs to (src/vtgaussian_slam.py:681-689: seen = radius > 0, the
 * running maximum of the screen-space radius) in the SAME launch as the loss's last step -- vtgs_seen_and_max_radius folded in:
 * one launch less per iteration.  n = 0: exactly vtgs_slam_loss_forward.  radii / max_2d_radius 16-byte aligned.  (ABI 16) */
int vtgs_slam_loss_forward_seen(int32_t mode, const float* im, const float* depth_sil, const float* gt_im, const float* gt_depth,
                                int32_t height, int32_t width, float sil_thres, float w_im, float w_depth, float* scratch,
                                float* ssim_grad_maps, float* out8, const float* extra_mask, const float* color_weight,
                                int32_t n, const int32_t* radii, float* max_2d_radius, uint8_t* seen, void* stream);
int vtgs_slam_loss_backward(int32_t mode, const float* im, const float* depth_sil, const float* gt_im, const float* gt_depth,
                            int32_t height, int32_t width, float sil_thres, float w_im, float w_depth,
                            const float* ssim_grad_maps, const float* fwd_out8, const float* upstream, float* g_im,
                            float* g_depth_sil, const float* extra_mask, const float* color_weight, void* stream);

/* ---- The same loss for ONE BAND of the tile-row partition ------------------------------------------------
 * A rank of the partition renders pixel rows [row_begin, row_end) and owns the loss terms of those rows.  Every term of
 * get_loss is a sum over pixels except two: the mapping depth term is a masked MEAN (src/vtgaussian_slam.py:592-597:
 * the band's sum over the count of ALL bands) and the SSIM window is 11x11 (utils/slam_external.py:78-87: rows
 * [row_begin - 5, row_end + 5) of `im` must hold what the neighbouring ranks rendered; the caller exchanges them).
 *   vtgs_slam_loss_band_sums   sums8 = {sum |gt_im - im| (weighted), sum |gt_depth - depth|, mask count, sum of the SSIM
 *                              map over the band's rows, 0 x 4}: additive over the bands -- the caller all-reduces them.
 *                              scratch / ssim_grad_maps as vtgs_slam_loss_forward (full-frame sizes; only band rows used).
 *   vtgs_slam_loss_band_share  out8 as vtgs_slam_loss_forward for this band's SHARE of the loss: the shares of all bands
 *                              sum to the full-frame loss (first_band carries the constant 0.2 w_im of 0.2 (1 - SSIM));
 *                              out8[1] is the count over all bands, out8[4] the full-frame mean SSIM.
 *   vtgs_slam_loss_band_backward  upstream[0] * d share / d im, d depth_sil.  Writes g_depth_sil rows [row_begin, row_end)
 *                              and g_im rows [row_begin, row_end) (mapping: [row_begin - 5, row_end + 5) clipped to the
 *                              image -- the part outside the band is the neighbours' renders' gradient, to be sent back);
 *                              every other row is left untouched (hand in zero-filled images).
 * The summation order inside a band differs from the full-frame call: values agree to float32 rounding.
 * vtgs_silhouette_sweep_band: vtgs_silhouette_sweep over pixels [pixel_begin, pixel_end) (rows x width of the band).   */
int vtgs_slam_loss_band_sums(int32_t mode, const float* im, const float* depth_sil, const float* gt_im, const float* gt_depth,
                             int32_t height, int32_t width, int32_t row_begin, int32_t row_end, float sil_thres,
                             float* scratch, float* ssim_grad_maps, float* sums8, const float* extra_mask,
                             const float* color_weight, void* stream);
int vtgs_slam_loss_band_share(int32_t mode, const float* own_sums8, const float* all_sums8, int32_t height, int32_t width,
                              float w_im, float w_depth, int32_t has_color_weight, int32_t first_band, float* out8,
                              void* stream);
int vtgs_slam_loss_band_backward(int32_t mode, const float* im, const float* depth_sil, const float* gt_im,
                                 const float* gt_depth, int32_t height, int32_t width, int32_t row_begin, int32_t row_end,
                                 float sil_thres, float w_im, float w_depth, const float* ssim_grad_maps,
                                 const float* share_out8, const float* upstream, float* g_im, float* g_depth_sil,
                                 const float* extra_mask, const float* color_weight, void* stream);
int vtgs_silhouette_sweep_band(const float* im, const float* silhouette, const float* gt_im, const float* gt_depth,
                               int32_t pixels, int32_t pixel_begin, int32_t pixel_end, const float* thresholds,
                               int32_t n_thresholds, float* partial_sums, void* stream);

/* ---- Adam over the parameter groups ------------------------------------------------------------------
 * Replaces torch.optim.Adam as the reference configures it (src/vtgaussian_slam.py:180-187: one group per tensor with its
 * own lr, betas (0.9, 0.999), eps 1e-8 tracking / 1e-15 mapping, no weight decay, no amsgrad) by ONE launch over up to
 * VTGS_ADAM_MAX_GROUPS tensors:  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 * p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps).   groups is a HOST array of device pointers.    */
#define VTGS_ADAM_MAX_GROUPS 8
typedef struct VtgsAdamGroup {
  float* param;         /* [count] updated in place */
  const float* grad;    /* [count] */
  float* exp_avg;       /* [count] first moment, updated in place */
  float* exp_avg_sq;    /* [count] second moment, updated in place */
  uint64_t count;
  float lr;
  float eps;
} VtgsAdamGroup;
int vtgs_adam_step(const VtgsAdamGroup* groups, int32_t n_groups, int32_t step, float beta1, float beta2, void* stream);
/* The same update for the listed ROWS only of per-Gaussian tensors (each group is [n_total_rows, count / n_total_rows], row-major):
 * rows[n_rows] are row indices; every other row -- parameter and both moments -- is left as it is.  A rank of the tile-row
 * partition that runs Adam for the Gaussians of its owner band only (partition.OwnerExchange; SURVEY.md 8e "each GPU runs Adam
 * on its owned slice").  Element for element the arithmetic of vtgs_adam_step.                                          */
int vtgs_adam_step_rows(const VtgsAdamGroup* groups, int32_t n_groups, int32_t step, float beta1, float beta2,
                        const int32_t* rows, int32_t n_rows, int32_t n_total_rows, void* stream);

/* The bookkeeping at the end of get_loss (src/vtgaussian_slam.py:681-689) in one launch instead of three element-wise ones:
 *   seen[i] = radii[i] > 0;   max_2D_radius[i] = max(max_2D_radius[i], (float)radii[i])   (a culled Gaussian has radius 0 and
 *   the running maximum is never negative, so the unmasked maximum equals the reference's masked update).
 * radii[N] int32 (the operator's second output), max_2d_radius[N] float32 in place, seen[N] one byte per Gaussian (0 / 1). */
int vtgs_seen_and_max_radius(int32_t n, const int32_t* radii, float* max_2d_radius, uint8_t* seen, void* stream);

/* ---- Point-to-plane consistency of two depth frames -----------------------------------------------
 * Replaces the host path of compute_point2plane_dist (src/vtgaussian_slam.py:1070-1155: kornia normals, numpy, Open3D
 * KD-tree) that the reference runs per tracking iteration at base-frame boundaries (:1929, :1956, :2158, :2185).
 * depth_target = the latest (reference) frame, depth_source = the current frame, both [H*W] floats on the device;
 * mask_* optional [H*W] bytes (the `varmask`s); intrinsics [9], w2c_* [16] row-major DEVICE pointers (the poses are slices
 * of parameters); threshold = max correspondence distance (0.02 in the reference); frustum = keep only points the other
 * camera sees.  out_dist[H*W] = n_target . (p_source - p_nearest_target) per source pixel, out_matched[H*W] = 1 where a
 * target point lies within threshold (nearest neighbour found exactly by a bounded window search, csrc/vtgs_p2p.hip).
 * The caller reduces: sum of squares ('sum'), max |.| ('max'), mean of the 100 largest |.| ('max100').             */
size_t vtgs_point2plane_scratch_bytes(int32_t width, int32_t height);
int vtgs_point2plane(int32_t width, int32_t height, const float* depth_target, const float* depth_source,
                     const uint8_t* mask_target, const uint8_t* mask_source, const float* intrinsics, const float* w2c_target,
                     const float* w2c_source, float threshold, int32_t frustum, void* scratch, size_t scratch_bytes,
                     float* out_dist, uint8_t* out_matched, void* stream);

/* Per-kernel timing with HIP events recorded on the stream each kernel is launched on (used by bench.py for
 * the roofline of the dominant kernel).  While enabled, every kernel launch of the library is bracketed by two
 * events; vtgs_profile_collect synchronises the device, sums elapsed time per kernel name since enabling and
 * clears the log.  Not thread-safe; meant for a measurement phase, not for production calls.               */
typedef struct VtgsProfileEntry {
  char     name[40];
  double   total_ms;
  uint32_t launches;
  uint32_t pad;
} VtgsProfileEntry;
int vtgs_profile_enable(int on);
int vtgs_profile_collect(VtgsProfileEntry* out, int32_t max_entries, int32_t* n_entries);

/* Introspection for tests: byte offsets of the workspace regions for (n, width, height, capacities).
 * out[0..7] = counters, geom (N x 8 f32: u v A B C opacity depth pad), gaux (N x {first instance, count}),
 * tile_counts (tiles8 x u32), sorted_gid (tiles8 x tile_capacity x u32), sorted_inst (same), final_T (P x f32),
 * tiles8 (count, not an offset).  8x8 tiles are numbered row-major over ceil(W/8) x ceil(H/8); tile t owns
 * entries [t * tile_capacity, t * tile_capacity + count[t]).  out[8] = quadrant masks (tiles8 x tile_capacity x u8, bit q =
 * the entry's alpha >= 1/255 box reaches 4x4 quadrant q of its tile; written by the default forward), out[9] = 64 x u32
 * partial sums of the forward's queue steps (only with option VTGS_COUNT_STEPS = 1).  out[10] = the forward's copy of the
 * bin plan ((tiles8 + 1) x u32; planned bins only: tile t then owns [plan[t], plan[t] + count[t]) instead), out[11] =
 * bin slots in total (a count).                                                                              */
int vtgs_debug_layout(int32_t n, int32_t width, int32_t height, uint64_t instance_capacity, uint32_t tile_capacity,
                      uint64_t out[12]);

#ifdef __cplusplus
}
#endif
#endif /* VTGS_H */
