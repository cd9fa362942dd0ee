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
#include <cmath>
#include <cstring>

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
    at::Tensor workspace = at::empty({(int64_t)nbytes}, f32.dtype(at::kByte));
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
    at::Tensor flat = at::empty({std::max<int64_t>(total, 1) * n}, f32);
    at::Tensor g[6];
    float* gp[6];
    int64_t off = 0;
    for (int i = 0; i < 6; ++i) {
      gp[i] = nullptr;
      if (!want[i]) continue;
      g[i] = flat.narrow(0, off * n, kWidth[i] * n).view({n, kWidth[i]});
      gp[i] = g[i].data_ptr<float>();
      off += kWidth[i];
    }
    if (n > 0 && total > 0) {
      // The scratch is indexed by instance id, and the instance CAPACITY bounds the ids whatever the forward counted.  (Round 3
      // sized it from the pinned result record behind slot_ptr; that slot is shared round-robin with later forwards, so a
      // backward that runs 64 forwards after its own forward could read another forward's -- smaller -- count: ADVICE r3.
      // Untouched tail pages of the block cost nothing.)
      const size_t sbytes = vtgs_backward_scratch_bytes((int32_t)n, (uint64_t)capacity);
      at::Tensor scratch = at::empty({(int64_t)sbytes}, f32.dtype(at::kByte));
      if (g_poison) scratch.fill_(0xFF);
      const int st = vtgs_backward(&cam.c, (int32_t)n, means3D.data_ptr<float>(), colors.data_ptr<float>(), opac.data_ptr<float>(),
                                   scales.data_ptr<float>(), rot.data_ptr<float>(), color.data_ptr<float>(),
                                   grad_color.data_ptr<float>(), workspace.data_ptr(), (size_t)workspace.numel(),
                                   (uint64_t)capacity, (uint32_t)tile_cap, nullptr, scratch.data_ptr(), sbytes,
                                   gp[0], gp[1], gp[2], gp[3], gp[4], gp[5], reinterpret_cast<void*>(stream));
      TORCH_CHECK(st == VTGS_OK, "vtgs_backward failed: ", vtgs_strerror(st), " (", vtgs_last_hip_error(), ")");
    }
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

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("rasterize", &rasterize, "GaussianRasterizer forward with a C++ autograd node behind it");
  m.def("abi_version", []() { return (int64_t)vtgs_abi_version(); });
  m.def("set_poison", [](bool on) { g_poison = on; });
}
