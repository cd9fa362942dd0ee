// vtgs_torch.cpp -- the drop-in operator's autograd node in C++ (host plumbing only: PyTorch provides device memory and the
// autograd graph; every kernel is behind the C ABI of include/vtgs.h, libvtgs.so).
//
// Why: through the Python autograd.Function one forward + backward costs ~200 us of host time (module call, ctypes
// marshalling, tensor allocation in Python, the engine re-entering Python for the backward: profiles/r2_shapes.md,
// tools/host_overhead.py) -- more than the kernels of three of the five BASELINE shapes.  Here the forward is one call
// from Python and the backward never touches the interpreter.
//
// Division of labour with diff_gaussian_rasterization/__init__.py: Python keeps the POLICY (capacity hints and hysteresis,
// checked vs run-ahead mode, the pinned result records and their bookkeeping, the retry after an overflow) and hands the
// decisions in; this file does the per-call work (checks, allocations, vtgs_forward / vtgs_backward).
//
// Built in-tree by __graft_entry__.build() / vtgaussian-slam_amd/build.py (torch.utils.cpp_extension, g++; no device code).
#include <torch/extension.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

#include "../../include/vtgs.h"

namespace {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

at::Tensor need(const at::Tensor& t, const char* name, int64_t tail, int64_t n, const at::Device& dev) {
  TORCH_CHECK(t.device() == dev, name, " is on ", t.device(), ", expected ", dev);
  TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32");
  TORCH_CHECK(t.numel() == n * tail, name, " must have ", n, "x", tail, " elements");
  return t.contiguous();
}

// tests only (diff_gaussian_rasterization.poison_workspaces): every workspace / scratch block starts as 0xFF bytes, the images
// as NaN, so that a kernel that reads a slot nobody wrote -- which passes silently when the caching allocator hands back
// clean memory -- shows up as a wrong result (the cause of the GPU fault of round 3).
bool g_poison = false;

// Rows of a carved block start at multiples of four floats: rotations / their gradients are read and written as float4, and a
// segment that follows 3 n or 13 n floats is 16-byte aligned only when n % 4 == 0 (the owned lists of the tile-row
// partition have any length).
inline int64_t pad4(int64_t n) { return (n + 3) & ~int64_t(3); }

// Bytes to ALLOCATE for a block of `nbytes` (workspace, record scratch): from 64 MB up, the next multiple of 1/8 of the largest
// power of two below it (<= 12.5 % more).  A SLAM map grows by a few thousand Gaussians per frame and the capacities creep
// with it: every frame asked the caching allocator for a block slightly larger than any it had cached -- 21 GB reserved after
// 164 frames around 2 GB in use (round 5, bench_slam.py --densify).  On the grid the sizes repeat and the blocks are reused.
inline int64_t alloc_bytes(size_t nbytes) {
  if (nbytes < (size_t(64) << 20)) return (int64_t)nbytes;
  size_t p = 1;
  while ((p << 1) <= nbytes) p <<= 1;
  const size_t g = p >> 3;
  return (int64_t)((nbytes + g - 1) / g * g);
}

// Instances a CHECKED (or SYNC) forward counted, read from its result record right after the call returned -- the backward's
// scratch holds one record per instance, and the capacity policy keeps 3.6 x the last need as room for run-ahead forwards:
// sizing the scratch by the capacity committed 3.6 x the memory (device memory is not demand-paged: ADVICE r4).  0 = not
// known at this point (run-ahead forward, overflow): the backward then sizes by the capacity, which bounds every instance id.
inline int64_t counted_instances(int status, int64_t fwd_flags, int64_t slot_ptr) {
  if (status != VTGS_OK || (fwd_flags & VTGS_FORWARD_MODE_MASK) == VTGS_FORWARD_ASYNC || slot_ptr == 0) return 0;
  const VtgsForwardInfo* info = reinterpret_cast<const VtgsForwardInfo*>(slot_ptr);
  return (info->complete && !info->overflow) ? (int64_t)info->instances_needed : 0;   // (instance IDS handed out: >= the instances binned)
}

// The verdict of a RUN-AHEAD forward, read at the end of the node's backward (an overflow has to leave
// `loss.backward()`, before any optimizer step).  Round 5 first did this with a Python post-hook on the graph node (~14 us of
// host time per iteration, the interpreter back in the backward); the pinned record is plain host memory, so the node reads it
// itself: wait for `complete` (the device writes it right after the binning -- long before the backward was enqueued), and on
// overflow mark the record as reported (complete = 2: the Python bookkeeping that follows at the next forward raises the
// capacities and stays silent) and throw -- or, with deferral on (N-rank loops, partition.phase_overflows), count it.
std::atomic<bool> g_check_in_backward{true}, g_defer_overflow{false};
// RenderFrame without an owned list: the render applies the activations itself (VTGS_FORWARD_RAW_ACTIVATIONS, round 6) and
// vtgs_prepare_frame_slot writes the camera-frame means and the depth colours only.  VTGS_FRAME_RAW=0: the full prepare.
std::atomic<bool> g_frame_raw{true};
std::atomic<int64_t> g_deferred{0};

// The records are handed out round-robin (64 per device and stream) and a record is settled and handed on 64 forwards later: a
// backward that runs that late (a retained graph, many evaluation renders in between) would read ANOTHER forward's record --
// spin on its complete = 0, report its overflow as this one's, or mark it reported so that its own backward stays silent
// (ADVICE r5).  The host keeps a generation number in the spare bytes of the 64-byte slot (bytes 48..55, which the device
// never writes; the Python pool bumps it whenever the slot changes hands); the node remembers the generation it was given and
// skips the check when the slot has moved on -- the pool settled this forward's verdict before it handed the slot on.
inline int64_t slot_generation(int64_t slot_ptr) {
  return slot_ptr ? (int64_t)*reinterpret_cast<volatile uint64_t*>(slot_ptr + (int64_t)sizeof(VtgsForwardInfo)) : 0;
}

void check_run_ahead(int64_t slot_ptr, int64_t fwd_flags, int64_t generation) {
  if (!g_check_in_backward.load() || slot_ptr == 0 || (fwd_flags & VTGS_FORWARD_MODE_MASK) != VTGS_FORWARD_ASYNC) return;
  if (slot_generation(slot_ptr) != generation) return;          // the slot belongs to a later forward now
  volatile VtgsForwardInfo* info = reinterpret_cast<volatile VtgsForwardInfo*>(slot_ptr);
  if (!info->complete) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!info->complete) {
      std::this_thread::yield();
      TORCH_CHECK(std::chrono::steady_clock::now() - t0 < std::chrono::seconds(20),
                  "vtgs_forward: timed out waiting for the result record of a run-ahead forward");
    }
  }
  if (info->overflow && info->complete != 2u) {
    info->complete = 2u;
    if (g_defer_overflow.load()) { g_deferred.fetch_add(1); return; }
    TORCH_CHECK(false, "vtgs_forward (run-ahead mode): the workspace of this iteration's forward overflowed -- after three "
                "forwards of this shape that needed the same, this one binned more than three times as much -- so the image it "
                "returned (the background colour) is INVALID, and so are these gradients. The capacities are raised at the next "
                "forward; redo the iteration, or set VTGS_FORWARD_MODE=checked to have every forward verified before it returns.");
  }
}

struct CamRecord {            // VtgsCamera with the three device tensors it points at kept alive
  VtgsCamera c;
  at::Tensor bg, view, proj;
};

CamRecord camera_from(const at::Tensor& cam_bytes, const at::Tensor& bg, const at::Tensor& view, const at::Tensor& proj) {
  TORCH_CHECK(cam_bytes.device().is_cpu() && cam_bytes.scalar_type() == at::kByte &&
              cam_bytes.numel() == (int64_t)sizeof(VtgsCamera), "camera record: ", sizeof(VtgsCamera), " bytes on the CPU");
  CamRecord r;
  std::memcpy(&r.c, cam_bytes.data_ptr(), sizeof(VtgsCamera));
  r.bg = bg; r.view = view; r.proj = proj;
  r.c.bg = bg.data_ptr<float>(); r.c.viewmatrix = view.data_ptr<float>(); r.c.projmatrix = proj.data_ptr<float>();
  return r;
}

// forward returns {color, radii, depth, workspace, status}; status (CPU int64 scalar) is VTGS_OK or
// VTGS_ERR_INSTANCE_OVERFLOW -- after an overflow the outputs are undefined and Python calls again with larger capacities.
struct Rasterize : public torch::autograd::Function<Rasterize> {
  static variable_list forward(AutogradContext* ctx, at::Tensor means3D, at::Tensor means2D, at::Tensor colors, at::Tensor opac,
                               at::Tensor scales, at::Tensor rot, at::Tensor cam_bytes, at::Tensor bg, at::Tensor view,
                               at::Tensor proj, int64_t capacity, int64_t tile_cap, int64_t bin_plan, int64_t slot_ptr,
                               int64_t flags, int64_t stream) {
    const at::Device dev = means3D.device();
    TORCH_CHECK(dev.is_cuda(), "GaussianRasterizer needs tensors on a HIP device (torch 'cuda'); no CPU path exists");
    c10::DeviceGuard guard(dev);
    const int64_t n = means3D.size(0);
    means3D = need(means3D, "means3D", 3, n, dev);
    colors = need(colors, "colors_precomp", 3, n, dev);
    opac = need(opac, "opacities", 1, n, dev);
    scales = need(scales, "scales", 3, n, dev);
    rot = need(rot, "rotations", 4, n, dev);
    CamRecord cam = camera_from(cam_bytes, bg, view, proj);
    const int64_t H = cam.c.image_height, W = cam.c.image_width;
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    at::Tensor images = at::empty({4, H, W}, f32);               // colour + depth in one allocation
    at::Tensor color = images.narrow(0, 0, 3), depth = images.narrow(0, 3, 1);
    at::Tensor radii = at::empty({n}, f32.dtype(at::kInt));
    const size_t nbytes = vtgs_workspace_bytes((int32_t)n, (int32_t)W, (int32_t)H, (uint64_t)capacity, (uint32_t)tile_cap);
    at::Tensor workspace = at::empty({alloc_bytes(nbytes)}, f32.dtype(at::kByte));
    if (g_poison) { workspace.fill_(0xFF); images.fill_(std::nanf("")); }
    // bin_plan != 0: planned bins (tile_cap carries VTGS_TILE_CAPACITY_PLANNED; the backward needs nothing else)
    const int st = bin_plan
        ? vtgs_forward_planned(&cam.c, (int32_t)n, means3D.data_ptr<float>(), colors.data_ptr<float>(), opac.data_ptr<float>(),
                               scales.data_ptr<float>(), rot.data_ptr<float>(), color.data_ptr<float>(), depth.data_ptr<float>(),
                               radii.data_ptr<int32_t>(), workspace.data_ptr(), nbytes, (uint64_t)capacity, (uint32_t)tile_cap,
                               reinterpret_cast<uint32_t*>(bin_plan), reinterpret_cast<VtgsForwardInfo*>(slot_ptr),
                               (uint32_t)flags, reinterpret_cast<void*>(stream))
        : vtgs_forward(&cam.c, (int32_t)n, means3D.data_ptr<float>(), colors.data_ptr<float>(), opac.data_ptr<float>(),
                       scales.data_ptr<float>(), rot.data_ptr<float>(), color.data_ptr<float>(), depth.data_ptr<float>(),
                       radii.data_ptr<int32_t>(), workspace.data_ptr(), nbytes, (uint64_t)capacity, (uint32_t)tile_cap,
                       reinterpret_cast<VtgsForwardInfo*>(slot_ptr), (uint32_t)flags, reinterpret_cast<void*>(stream));
    TORCH_CHECK(st == VTGS_OK || st == VTGS_ERR_INSTANCE_OVERFLOW, "vtgs_forward failed: ", vtgs_strerror(st), " (",
                vtgs_last_hip_error(), ")");
    ctx->save_for_backward({means3D, colors, opac, scales, rot, color, workspace, cam_bytes, bg, view, proj});
    ctx->saved_data["instances"] = counted_instances(st, flags, slot_ptr);
    ctx->saved_data["slot_ptr"] = slot_ptr;
    ctx->saved_data["slot_gen"] = slot_generation(slot_ptr);
    ctx->saved_data["fwd_flags"] = flags;
    ctx->saved_data["capacity"] = capacity;
    ctx->saved_data["tile_cap"] = tile_cap;
    ctx->saved_data["stream"] = stream;
    ctx->saved_data["n"] = n;
    ctx->set_materialize_grads(false);
    ctx->mark_non_differentiable({radii, depth, workspace});
    at::Tensor status = at::empty({}, at::TensorOptions().dtype(at::kLong));
    status.fill_((int64_t)st);
    ctx->mark_non_differentiable({status});
    return {color, radii, depth, workspace, status};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const at::Tensor &means3D = saved[0], &colors = saved[1], &opac = saved[2], &scales = saved[3], &rot = saved[4], &color = saved[5],
                     &workspace = saved[6];
    const at::Device dev = means3D.device();
    c10::DeviceGuard guard(dev);
    CamRecord cam = camera_from(saved[7], saved[8], saved[9], saved[10]);
    const int64_t n = ctx->saved_data["n"].toInt(), capacity = ctx->saved_data["capacity"].toInt(),
                  tile_cap = ctx->saved_data["tile_cap"].toInt(), stream = ctx->saved_data["stream"].toInt();
    at::Tensor grad_color = grads[0].defined() ? grads[0].to(at::kFloat).contiguous() : at::zeros_like(color);
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    // one allocation for the gradients somebody asked for (68 bytes per Gaussian when all six are; a tracking iteration of
    // the unfused loops, src/vtgaussian_slam.py:428-449, detaches the Gaussians and needs 24): the others are neither
    // allocated nor stored by the kernel
    static const int64_t kWidth[6] = {3, 3, 3, 1, 3, 4};       // means3D, means2D, colours, opacities, scales, rotations
    int64_t total = 0;
    bool want[6];
    for (int i = 0; i < 6; ++i) { want[i] = ctx->needs_input_grad(i); total += want[i] ? kWidth[i] : 0; }
    const int64_t np = pad4(n);                                  // segment stride: every array starts 16-byte aligned
    at::Tensor flat = at::empty({std::max<int64_t>(total, 1) * std::max<int64_t>(np, 1)}, f32);
    at::Tensor g[6];
    float* gp[6];
    int64_t off = 0;
    for (int i = 0; i < 6; ++i) {
      gp[i] = nullptr;
      if (!want[i]) continue;
      g[i] = flat.narrow(0, off * np, kWidth[i] * n).view({n, kWidth[i]});
      gp[i] = g[i].data_ptr<float>();
      off += kWidth[i];
    }
    if (n > 0 && total > 0) {
      // The scratch is indexed by instance id: the count the forward read from its OWN record before it returned (checked
      // forwards), else the instance capacity, which bounds the ids whatever was counted.  (Round 3 re-read the pinned slot
      // here; it is shared round-robin with later forwards: ADVICE r3.)
      const int64_t counted = ctx->saved_data["instances"].toInt();
      const size_t sbytes = vtgs_backward_scratch_bytes((int32_t)n, (uint64_t)(counted > 0 ? counted : capacity));
      at::Tensor scratch = at::empty({alloc_bytes(sbytes)}, f32.dtype(at::kByte));
      if (g_poison) scratch.fill_(0xFF);
      const int st = vtgs_backward(&cam.c, (int32_t)n, means3D.data_ptr<float>(), colors.data_ptr<float>(), opac.data_ptr<float>(),
                                   scales.data_ptr<float>(), rot.data_ptr<float>(), color.data_ptr<float>(),
                                   grad_color.data_ptr<float>(), workspace.data_ptr(), (size_t)workspace.numel(),
                                   (uint64_t)capacity, (uint32_t)tile_cap, nullptr, scratch.data_ptr(), sbytes,
                                   gp[0], gp[1], gp[2], gp[3], gp[4], gp[5], reinterpret_cast<void*>(stream));
      TORCH_CHECK(st == VTGS_OK, "vtgs_backward failed: ", vtgs_strerror(st), " (", vtgs_last_hip_error(), ")");
    }
    check_run_ahead(ctx->saved_data["slot_ptr"].toInt(), ctx->saved_data["fwd_flags"].toInt(), ctx->saved_data["slot_gen"].toInt());   // (after the launches)
    at::Tensor none;
    return {g[0], g[1], g[2], g[3], g[4], g[5], none, none, none, none, none, none, none, none, none, none};
  }
};

std::vector<at::Tensor> rasterize(at::Tensor means3D, at::Tensor means2D, at::Tensor colors, at::Tensor opac, at::Tensor scales,
                                  at::Tensor rot, at::Tensor cam_bytes, at::Tensor bg, at::Tensor view, at::Tensor proj,
                                  int64_t capacity, int64_t tile_cap, int64_t bin_plan, int64_t slot_ptr, int64_t flags,
                                  int64_t stream) {
  return Rasterize::apply(means3D, means2D, colors, opac, scales, rot, cam_bytes, bg, view, proj, capacity, tile_cap, bin_plan,
                          slot_ptr, flags, stream);
}

// ---- the fused frame render (diff_gaussian_rasterization/fused.py: pose transform + render variables + both renders) ----------
// Same division of labour: Python decides (capacities, mode, result record, which gradients are wanted, the owned set of a rank
// of the tile-row partition); this node does the per-call work -- vtgs_prepare_frame[_owned], vtgs_forward_dual[_planned] and, in
// the backward, vtgs_backward_dual_frame_owned + vtgs_pose_gradient -- without the interpreter.  Why now: with the kernels of
// one BAND a rank's iteration is bound by this host path.
// forward returns {im, depth_sil, radii, workspace, status}.
struct RenderFrame : public torch::autograd::Function<RenderFrame> {
  static variable_list forward(AutogradContext* ctx, at::Tensor means3D, at::Tensor rgb, at::Tensor unnorm_rot, at::Tensor logit_op,
                               at::Tensor log_scales, at::Tensor cam_rots, at::Tensor cam_trans, int64_t t_idx, at::Tensor depth_w2c,
                               at::Tensor cam_bytes,
                               at::Tensor bg, at::Tensor view, at::Tensor proj, int64_t capacity, int64_t tile_cap, int64_t bin_plan,
                               int64_t slot_ptr, int64_t fwd_flags, int64_t frame_flags, int64_t stream,
                               c10::optional<at::Tensor> owned_idx, c10::optional<at::Tensor> owned_idx64,
                               c10::optional<at::Tensor> owned_mask, c10::optional<at::Tensor> owned_escapes) {
    const at::Device dev = means3D.device();
    TORCH_CHECK(dev.is_cuda(), "render_frame needs tensors on a HIP device (torch 'cuda'); no CPU path exists");
    c10::DeviceGuard guard(dev);
    const int64_t n_map = means3D.size(0);
    const bool owned = owned_idx.has_value() && owned_idx->defined();
    const int64_t n = owned ? owned_idx->numel() : n_map;
    means3D = need(means3D, "means3D", 3, n_map, dev);
    rgb = need(rgb, "rgb_colors", 3, n_map, dev);
    unnorm_rot = need(unnorm_rot, "unnorm_rotations", 4, n_map, dev);
    logit_op = need(logit_op, "logit_opacities", 1, n_map, dev);
    log_scales = need(log_scales, "log_scales", 1, n_map, dev);
    // The pose: column t_idx of the reference's camera tensors [1,4,T] / [1,3,T] (src/vtgaussian_slam.py:160-167), read IN PLACE
    // by vtgs_prepare_frame_slot, which also leaves it as seven contiguous floats for the backward; the gradient goes back
    // full-size from vtgs_pose_gradient_slot (round 6: the two slot launches of round 5 folded into their neighbours)
    TORCH_CHECK(cam_rots.dim() == 3 && cam_trans.dim() == 3 && cam_rots.size(0) == 1 && cam_trans.size(0) == 1 && cam_rots.size(1) == 4 &&
                cam_trans.size(1) == 3 && cam_rots.size(2) == cam_trans.size(2) && t_idx >= 0 && t_idx < cam_rots.size(2),
                "camera tensors [1,4,T] / [1,3,T] and a frame index inside them");
    const int64_t frames = cam_rots.size(2);
    cam_rots = need(cam_rots.reshape({4 * frames}), "cam_unnorm_rots", 4 * frames, 1, dev);
    cam_trans = need(cam_trans.reshape({3 * frames}), "cam_trans", 3 * frames, 1, dev);
    depth_w2c = need(depth_w2c, "first-frame w2c", 16, 1, dev);
    CamRecord cam = camera_from(cam_bytes, bg, view, proj);
    const int64_t H = cam.c.image_height, W = cam.c.image_width;
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    void* st_ = reinterpret_cast<void*>(stream);
    // one block for the render variables: means_cam 3, opacities 1, scales 3, rotations 4, depth colours 3 (+ compact colours 3)
    // raw: means_cam 3 + depth colours 3 -- the kernels read logits / log-scales themselves (isotropic map: no rotation)
    const bool raw = !owned && g_frame_raw.load();
    if (raw) { fwd_flags |= (int64_t)VTGS_FORWARD_RAW_ACTIVATIONS; frame_flags |= 16; }
    const int64_t np = pad4(n);                                   // segment stride (rot is read as float4)
    at::Tensor vars = at::empty({std::max<int64_t>(np, 4) * (owned ? 17 : raw ? 6 : 14)}, f32);
    float* v = vars.data_ptr<float>();
    float *means_cam = v, *opac = v + 3 * np, *scales = v + 4 * np, *rot = v + 7 * np, *dcol = v + 11 * np, *rgb_c = v + 14 * np;
    if (raw) { dcol = v + 3 * np; opac = logit_op.data_ptr<float>(); scales = log_scales.data_ptr<float>(); rot = nullptr; }
    at::Tensor pose7 = at::empty({7}, f32);
    at::Tensor cam_q = pose7.narrow(0, 0, 4), cam_t = pose7.narrow(0, 4, 3);
    // the forward's workspace, ahead of the prepare step: its head (counters, list lengths) is zeroed by that launch
    const size_t nbytes = vtgs_workspace_bytes((int32_t)n, (int32_t)W, (int32_t)H, (uint64_t)capacity, (uint32_t)tile_cap);
    at::Tensor workspace = at::empty({alloc_bytes(nbytes)}, f32.dtype(at::kByte));
    at::Tensor images = at::empty({6, H, W}, f32);
    if (g_poison) { workspace.fill_(0xFF); images.fill_(std::nanf("")); }
    const size_t clear_bytes = owned ? 0 : vtgs_workspace_clear_bytes((int32_t)W, (int32_t)H);
    if (clear_bytes) fwd_flags |= (int64_t)VTGS_FORWARD_WORKSPACE_CLEARED;
    int rc;
    if (owned) {
      rc = vtgs_pose_slot_gather(cam_rots.data_ptr<float>(), cam_trans.data_ptr<float>(), (int32_t)frames, (int32_t)t_idx,
                                 pose7.data_ptr<float>(), st_);     // (the band test below wants the pose ahead of the transform)
      TORCH_CHECK(rc == VTGS_OK, "vtgs_pose_slot_gather failed: ", vtgs_strerror(rc));
      TORCH_CHECK(owned_idx->scalar_type() == at::kInt && owned_idx->is_contiguous() && owned_idx->device() == dev &&
                  owned_idx64.has_value() && owned_mask.has_value() && owned_escapes.has_value(), "owned set: int32 index list on the device");
      rc = vtgs_band_owner_mask(&cam.c, (int32_t)n_map, means3D.data_ptr<float>(), log_scales.data_ptr<float>(), 1,
                                cam_q.data_ptr<float>(), cam_t.data_ptr<float>(), 1.f, 1.f, owned_mask->data_ptr<uint8_t>(), nullptr,
                                reinterpret_cast<uint32_t*>(owned_escapes->data_ptr<int32_t>()), nullptr, st_);
      TORCH_CHECK(rc == VTGS_OK, "vtgs_band_owner_mask failed: ", vtgs_strerror(rc));
      rc = vtgs_prepare_frame_owned((int32_t)n, owned_idx->data_ptr<int32_t>(), means3D.data_ptr<float>(), logit_op.data_ptr<float>(),
                                    log_scales.data_ptr<float>(), unnorm_rot.data_ptr<float>(), rgb.data_ptr<float>(),
                                    cam_q.data_ptr<float>(), cam_t.data_ptr<float>(), depth_w2c.data_ptr<float>(), means_cam, opac,
                                    scales, rot, dcol, rgb_c, st_);
    } else {
      rc = vtgs_prepare_frame_slot((int32_t)n, means3D.data_ptr<float>(), logit_op.data_ptr<float>(), log_scales.data_ptr<float>(),
                                   unnorm_rot.data_ptr<float>(), cam_rots.data_ptr<float>(), cam_trans.data_ptr<float>(),
                                   (int32_t)frames, (int32_t)t_idx, depth_w2c.data_ptr<float>(), means_cam, raw ? nullptr : opac,
                                   raw ? nullptr : scales, raw ? nullptr : rot, dcol, pose7.data_ptr<float>(),
                                   clear_bytes ? workspace.data_ptr() : nullptr, clear_bytes, st_);
    }
    TORCH_CHECK(rc == VTGS_OK, "vtgs_prepare_frame failed: ", vtgs_strerror(rc), " (", vtgs_last_hip_error(), ")");
    const float* colors_a = owned ? rgb_c : rgb.data_ptr<float>();
    at::Tensor im = images.narrow(0, 0, 3), depth_sil = images.narrow(0, 3, 3);
    at::Tensor radii = at::empty({n}, f32.dtype(at::kInt));
    const int st = bin_plan
        ? vtgs_forward_dual_planned(&cam.c, (int32_t)n, means_cam, colors_a, dcol, opac, scales, rot, im.data_ptr<float>(),
                                    depth_sil.data_ptr<float>(), radii.data_ptr<int32_t>(), workspace.data_ptr(), nbytes,
                                    (uint64_t)capacity, (uint32_t)tile_cap, reinterpret_cast<uint32_t*>(bin_plan),
                                    reinterpret_cast<VtgsForwardInfo*>(slot_ptr), (uint32_t)fwd_flags, st_)
        : vtgs_forward_dual(&cam.c, (int32_t)n, means_cam, colors_a, dcol, opac, scales, rot, im.data_ptr<float>(),
                            depth_sil.data_ptr<float>(), radii.data_ptr<int32_t>(), workspace.data_ptr(), nbytes, (uint64_t)capacity,
                            (uint32_t)tile_cap, reinterpret_cast<VtgsForwardInfo*>(slot_ptr), (uint32_t)fwd_flags, st_);
    TORCH_CHECK(st == VTGS_OK || st == VTGS_ERR_INSTANCE_OVERFLOW, "vtgs_forward_dual failed: ", vtgs_strerror(st), " (",
                vtgs_last_hip_error(), ")");
    if (owned)                                                    // the radii of the map: 0 outside the list
      radii = at::zeros({n_map}, f32.dtype(at::kInt)).index_copy_(0, *owned_idx64, radii);
    at::Tensor idx_saved = owned ? *owned_idx : at::Tensor();
    ctx->save_for_backward({means3D, rgb, unnorm_rot, cam_q, cam_t, depth_w2c, vars, images, workspace, cam_bytes, bg, view, proj,
                            idx_saved, logit_op, log_scales});
    ctx->saved_data["instances"] = counted_instances(st, fwd_flags, slot_ptr);
    ctx->saved_data["slot_ptr"] = slot_ptr;
    ctx->saved_data["slot_gen"] = slot_generation(slot_ptr);
    ctx->saved_data["fwd_flags"] = fwd_flags;
    ctx->saved_data["capacity"] = capacity;
    ctx->saved_data["tile_cap"] = tile_cap;
    ctx->saved_data["stream"] = stream;
    ctx->saved_data["n"] = n;
    ctx->saved_data["n_map"] = n_map;
    ctx->saved_data["frame_flags"] = frame_flags;
    ctx->saved_data["frames"] = frames;
    ctx->saved_data["t_idx"] = t_idx;
    ctx->set_materialize_grads(false);
    at::Tensor status = at::empty({}, at::TensorOptions().dtype(at::kLong));
    status.fill_((int64_t)st);
    ctx->mark_non_differentiable({radii, workspace, status});
    return {im, depth_sil, radii, workspace, status};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const at::Tensor &means3D = saved[0], &rgb = saved[1], &unnorm_rot = saved[2], &cam_q = saved[3], &cam_t = saved[4],
                     &depth_w2c = saved[5], &vars = saved[6], &images = saved[7], &workspace = saved[8], &idx = saved[13];
    const at::Device dev = means3D.device();
    c10::DeviceGuard guard(dev);
    CamRecord cam = camera_from(saved[9], saved[10], saved[11], saved[12]);
    const int64_t n = ctx->saved_data["n"].toInt(), n_map = ctx->saved_data["n_map"].toInt(),
                  capacity = ctx->saved_data["capacity"].toInt(), tile_cap = ctx->saved_data["tile_cap"].toInt(),
                  stream = ctx->saved_data["stream"].toInt(), flags = ctx->saved_data["frame_flags"].toInt(),
                  frames = ctx->saved_data["frames"].toInt(), t_idx = ctx->saved_data["t_idx"].toInt();
    const bool owned = idx.defined();
    void* st_ = reinterpret_cast<void*>(stream);
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    at::Tensor im = images.narrow(0, 0, 3), depth_sil = images.narrow(0, 3, 3);
    at::Tensor g_im = grads[0].defined() ? grads[0].to(at::kFloat).contiguous() : at::zeros_like(im);
    at::Tensor g_ds = grads[1].defined() ? grads[1].to(at::kFloat).contiguous() : at::zeros_like(depth_sil);
    const bool want_g = flags & 1, want_p = flags & 2, want_a = flags & 4;
    // the map-sized gradients somebody asked for, one block (with a list the kernel writes the listed rows only: zeros first)
    const int64_t width = (want_g ? 7 : 0) + (want_a ? 5 : 0);
    const int64_t npm = std::max<int64_t>(pad4(n_map), 4);      // segment stride (g_unnorm_rotations is written as float4)
    at::Tensor flat = owned ? at::zeros({std::max<int64_t>(width, 1) * npm}, f32) : at::empty({std::max<int64_t>(width, 1) * npm}, f32);
    float* f = flat.data_ptr<float>();
    at::Tensor g_means3D, g_ur, g_rgb, g_logit, g_ls;
    int64_t off = 0;
    auto take = [&](int64_t w) { at::Tensor t = flat.narrow(0, off * npm, w * n_map).view({n_map, w}); off += w; return t; };
    if (want_g) { g_means3D = take(3); g_ur = take(4); }
    if (want_a) { g_rgb = take(3); g_logit = take(1); g_ls = take(1); }
    (void)f;
    auto ptr = [](const at::Tensor& t) -> float* { return t.defined() ? t.data_ptr<float>() : nullptr; };
    const uint32_t rows = vtgs_pose_partial_rows((int32_t)n);
    at::Tensor partials = want_p ? at::empty({std::max<int64_t>(rows, 1), 12}, f32) : at::Tensor();
    at::Tensor g_q, g_t;
    if (n > 0 && flags != 0) {
      const float* v = vars.data_ptr<float>();
      const int64_t np = pad4(n);
      const float *means_cam = v, *opac = v + 3 * np, *scales = v + 4 * np, *rot = v + 7 * np, *dcol = v + 11 * np, *rgb_c = v + 14 * np;
      if (flags & 16) { dcol = v + 3 * np; opac = saved[14].data_ptr<float>(); scales = saved[15].data_ptr<float>(); rot = nullptr; }   // (raw activations)
      const int64_t counted = ctx->saved_data["instances"].toInt();   // (see Rasterize::backward)
      const size_t sbytes = vtgs_backward_dual_scratch_bytes((int32_t)n, (uint64_t)(counted > 0 ? counted : capacity));
      at::Tensor scratch = at::empty({alloc_bytes(sbytes)}, f32.dtype(at::kByte));
      if (g_poison) scratch.fill_(0xFF);
      const int st = vtgs_backward_dual_frame_owned(
          &cam.c, (int32_t)n, owned ? idx.data_ptr<int32_t>() : nullptr, means_cam, owned ? rgb_c : rgb.data_ptr<float>(), dcol, opac,
          scales, rot, im.data_ptr<float>(), depth_sil.data_ptr<float>(), g_im.data_ptr<float>(), g_ds.data_ptr<float>(),
          workspace.data_ptr(), (size_t)workspace.numel(), (uint64_t)capacity, (uint32_t)tile_cap, scratch.data_ptr(), sbytes,
          (uint32_t)flags, means3D.data_ptr<float>(), unnorm_rot.data_ptr<float>(), cam_q.data_ptr<float>(), cam_t.data_ptr<float>(),
          depth_w2c.data_ptr<float>(), ptr(g_rgb), ptr(g_means3D), ptr(g_logit), ptr(g_ls), ptr(g_ur), ptr(partials), st_);
      TORCH_CHECK(st == VTGS_OK, "vtgs_backward_dual_frame failed: ", vtgs_strerror(st), " (", vtgs_last_hip_error(), ")");
      if (want_p) {
        at::Tensor qt = at::empty({7 * frames}, f32);
        g_q = qt.narrow(0, 0, 4 * frames).view({1, 4, frames}); g_t = qt.narrow(0, 4 * frames, 3 * frames).view({1, 3, frames});
        const int sp = vtgs_pose_gradient_slot(partials.data_ptr<float>(), rows, cam_q.data_ptr<float>(), (int32_t)frames, (int32_t)t_idx,
                                               g_q.data_ptr<float>(), g_t.data_ptr<float>(), st_);
        TORCH_CHECK(sp == VTGS_OK, "vtgs_pose_gradient_slot failed: ", vtgs_strerror(sp));
      }
    } else if (want_p) {                                           // nothing rendered: zero pose gradient, no launch
      g_q = at::zeros({1, 4, frames}, f32); g_t = at::zeros({1, 3, frames}, f32);
    }
    if (n == 0 && !owned && width > 0) flat.zero_();
    check_run_ahead(ctx->saved_data["slot_ptr"].toInt(), ctx->saved_data["fwd_flags"].toInt(), ctx->saved_data["slot_gen"].toInt());   // (after the launches)
    at::Tensor none;
    return {g_means3D, g_rgb, g_ur, g_logit.defined() ? g_logit : none, g_ls.defined() ? g_ls : none, g_q, g_t, none,
            none, none, none, none, none, none, none, none, none, none, none, none, none, none, none, none};
  }
};

std::vector<at::Tensor> render_frame(at::Tensor means3D, at::Tensor rgb, at::Tensor unnorm_rot, at::Tensor logit_op, at::Tensor log_scales,
                                     at::Tensor cam_rots, at::Tensor cam_trans, int64_t t_idx, at::Tensor depth_w2c, at::Tensor cam_bytes, at::Tensor bg,
                                     at::Tensor view, at::Tensor proj, int64_t capacity, int64_t tile_cap, int64_t bin_plan,
                                     int64_t slot_ptr, int64_t fwd_flags, int64_t frame_flags, int64_t stream,
                                     c10::optional<at::Tensor> owned_idx, c10::optional<at::Tensor> owned_idx64,
                                     c10::optional<at::Tensor> owned_mask, c10::optional<at::Tensor> owned_escapes) {
  return RenderFrame::apply(means3D, rgb, unnorm_rot, logit_op, log_scales, cam_rots, cam_trans, t_idx, depth_w2c, cam_bytes, bg, view, proj, capacity,
                            tile_cap, bin_plan, slot_ptr, fwd_flags, frame_flags, stream, owned_idx, owned_idx64, owned_mask,
                            owned_escapes);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("render_frame", &render_frame, "fused.render_frame forward with a C++ autograd node behind it");
  m.def("rasterize", &rasterize, "GaussianRasterizer forward with a C++ autograd node behind it");
  m.def("abi_version", []() { return (int64_t)vtgs_abi_version(); });
  m.def("set_poison", [](bool on) { g_poison = on; });
  m.def("set_check_in_backward", [](bool on) { g_check_in_backward = on; });
  m.def("set_defer_overflow", [](bool on) { g_defer_overflow = on; });
  m.def("set_frame_raw", [](bool on) { g_frame_raw = on; });
  m.def("take_deferred_overflows", []() { return g_deferred.exchange(0); });
}
