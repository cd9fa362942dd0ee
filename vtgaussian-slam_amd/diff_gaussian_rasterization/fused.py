"""Fused frame render (SURVEY.md 8f-1 + 8f-2): pose transform, render-variable builders and BOTH renders of one
`get_loss` call (src/vtgaussian_slam.py:431-468) as one autograd node.

    im, depth_sil, radii = render_frame(params, time_idx, raster_settings, first_frame_w2c,
                                        gaussians_grad=..., camera_grad=...)

is equivalent to the reference's

    tg  = transform_to_frame(params, time_idx, gaussians_grad, camera_grad)          utils/slam_helpers.py:323-385
    im, radii, _      = Renderer(cam)(**transformed_params2rendervar(params, tg))                       :127-160
    depth_sil, _, _   = Renderer(cam)(**transformed_params2depthplussilhouette(params, w2c, tg))        :255-287

but -- for isotropic maps (log_scales [N,1], every reference config; anisotropic ones take the chain above on the HIP
operator, `render_frame_unfused`) -- runs the element-wise chain as one HIP kernel
each way (vtgs_prepare_frame / vtgs_prepare_frame_backward), projects / bins / sorts once and composites BOTH
renders in one six-channel pass each way (vtgs_forward_dual / vtgs_backward_dual; VTGS_DUAL=0 selects the earlier
vtgs_forward + vtgs_forward_shared + 2 x vtgs_backward route, kept as the cross-check) and reduces dL/dmeans to the 7
pose scalars on the device (vtgs_prepare_frame_backward + vtgs_pose_gradient).
Opt-in: the unmodified driver keeps working through the plain GaussianRasterizer.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

import os

from . import (_Camera, _ForwardState, _camera_for, _RADIUS_RULES, _check, _device_guard, _lib, _run_backward, _run_backward_dual,
               _run_forward, _scratch, _scratch_instances, _settle, _stream_ptr, _I32, _P, VTGS_FORWARD_SECOND_IS_DEPTH)

_lib.vtgs_pose_partial_rows.restype, _lib.vtgs_pose_partial_rows.argtypes = ctypes.c_uint32, [_I32]
_lib.vtgs_prepare_frame.restype, _lib.vtgs_prepare_frame.argtypes = ctypes.c_int, [_I32] + [_P] * 13
_lib.vtgs_prepare_frame_backward.restype = ctypes.c_int
_lib.vtgs_prepare_frame_backward.argtypes = [_I32, ctypes.c_uint32] + [_P] * 22
_lib.vtgs_pose_gradient.restype, _lib.vtgs_pose_gradient.argtypes = ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, _P, _P]
_lib.vtgs_prepare_frame_owned.restype, _lib.vtgs_prepare_frame_owned.argtypes = ctypes.c_int, [_I32] + [_P] * 16
_lib.vtgs_pose_slot_gather.restype, _lib.vtgs_pose_slot_gather.argtypes = ctypes.c_int, [_P, _P, _I32, _I32, _P, _P]
_lib.vtgs_pose_slot_scatter.restype, _lib.vtgs_pose_slot_scatter.argtypes = ctypes.c_int, [_P, _P, _I32, _I32, _P, _P, _P]


class _PoseSlot(torch.autograd.Function):
    """(q[4], t[3]) of frame `t_idx` out of the reference's camera tensors [1,4,T] / [1,3,T], and the adjoint, ONE launch each way.
    `params['cam_unnorm_rots'][0, :, t]` is two `select`s: forward two strided copies (the kernels want contiguous floats),
    backward two zero-fills and two slice copies per tensor -- ten launches of ~5 us around seven floats, 6 % of a tracking
    iteration at 1 M Gaussians (kernel trace of round 5)."""

    @staticmethod
    def forward(ctx, rots, trans, t_idx: int):
        T = int(rots.shape[2])
        pose7 = torch.empty(7, dtype=torch.float32, device=rots.device)
        with _device_guard(rots.device):
            _check(_lib.vtgs_pose_slot_gather(rots.data_ptr(), trans.data_ptr(), T, int(t_idx), pose7.data_ptr(), _stream_ptr(rots.device)),
                   "vtgs_pose_slot_gather")
        ctx.T, ctx.t_idx, ctx.shapes = T, int(t_idx), (rots.shape, trans.shape)
        return pose7[:4], pose7[4:]

    @staticmethod
    def backward(ctx, g_q, g_t):
        ref = g_q if g_q is not None else g_t
        dev = ref.device
        g_rots = torch.empty(ctx.shapes[0], dtype=torch.float32, device=dev)
        g_trans = torch.empty(ctx.shapes[1], dtype=torch.float32, device=dev)
        c = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_q, g_t = c(g_q), c(g_t)
        ptr = lambda g: None if g is None else g.data_ptr()
        with _device_guard(dev):
            _check(_lib.vtgs_pose_slot_scatter(ptr(g_q), ptr(g_t), ctx.T, ctx.t_idx, g_rots.data_ptr(), g_trans.data_ptr(), _stream_ptr(dev)),
                   "vtgs_pose_slot_scatter")
        return g_rots, g_trans, None


def _pose_of_frame(params, time_idx: int, camera_grad: bool):
    rots, trans = params["cam_unnorm_rots"], params["cam_trans"]
    if (rots.dim() == 3 and trans.dim() == 3 and rots.shape[0] == 1 and trans.shape[0] == 1 and rots.shape[1] == 4 and trans.shape[1] == 3
            and rots.shape[2] == trans.shape[2] and rots.dtype is torch.float32 and trans.dtype is torch.float32
            and rots.is_contiguous() and trans.is_contiguous() and rots.is_cuda and 0 <= int(time_idx) < rots.shape[2]):
        if not camera_grad:
            rots, trans = rots.detach(), trans.detach()
        return _PoseSlot.apply(rots, trans, int(time_idx))
    q, t = rots[0, :, time_idx], trans[0, :, time_idx]                 # any other layout: plain indexing
    return (q, t) if camera_grad else (q.detach(), t.detach())


class _RenderFrame(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, rgb, unnorm_rot, logit_op, log_scales, cam_q, cam_t, depth_w2c, cam: _Camera, flags: int,
                owned=None):
        dev = means3D.device
        n_map = means3D.shape[0]
        n = n_map if owned is None else int(owned.idx.numel())       # rows the rasterizer sees
        f32 = lambda t: t.detach() if (t.dtype is torch.float32 and t.is_contiguous()) else t.detach().to(torch.float32).contiguous()
        means3D, rgb, unnorm_rot, logit_op, log_scales = map(f32, (means3D, rgb, unnorm_rot, logit_op, log_scales))
        cam_q, cam_t, depth_w2c = f32(cam_q).reshape(-1), f32(cam_t).reshape(-1), f32(depth_w2c).reshape(-1)
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        means_cam, opac, scales, rot, dcol = new(n, 3), new(n, 1), new(n, 3), new(n, 4), new(n, 3)
        stream = _stream_ptr(dev)
        dual = os.environ.get("VTGS_DUAL", "1") != "0"             # read per call, like the other implementation switches
        if owned is None:
            _check(_lib.vtgs_prepare_frame(n, means3D.data_ptr(), logit_op.data_ptr(), log_scales.data_ptr(),
                                           unnorm_rot.data_ptr(), cam_q.data_ptr(), cam_t.data_ptr(), depth_w2c.data_ptr(),
                                           means_cam.data_ptr(), opac.data_ptr(), scales.data_ptr(), rot.data_ptr(),
                                           dcol.data_ptr(), stream), "vtgs_prepare_frame")
        else:
            # a rank of the tile-row partition with a list of the Gaussians that can meet its rows (partition.OwnedSet): the
            # rasterizer gets compact arrays of those -- colours included -- and the map's other rows are never read
            if not dual or os.environ.get("VTGS_FRAME_EPILOGUE", "1") == "0":
                raise RuntimeError("owned sets run on the dual render with the frame epilogue (VTGS_DUAL / VTGS_FRAME_EPILOGUE "
                                   "select cross-check routes that have no list form)")
            owned.check(cam, means3D, log_scales, cam_q, cam_t, stream)   # the exact band test of the whole map, counted
            rgb_map, rgb = rgb, new(n, 3)
            _check(_lib.vtgs_prepare_frame_owned(n, owned.idx.data_ptr(), means3D.data_ptr(), logit_op.data_ptr(),
                                                 log_scales.data_ptr(), unnorm_rot.data_ptr(), rgb_map.data_ptr(),
                                                 cam_q.data_ptr(), cam_t.data_ptr(), depth_w2c.data_ptr(), means_cam.data_ptr(),
                                                 opac.data_ptr(), scales.data_ptr(), rot.data_ptr(), dcol.data_ptr(),
                                                 rgb.data_ptr(), stream), "vtgs_prepare_frame_owned")
        state = None
        if dual:
            im, radii, depth_sil, fs = _run_forward(cam, means_cam, rgb, opac, scales, rot, colors_b=dcol,
                                                    want_async=(flags & 7) != 0,
                                                    extra_flags=VTGS_FORWARD_SECOND_IS_DEPTH if flags & 16 else 0)
        else:
            im, radii, _, fs = _run_forward(cam, means_cam, rgb, opac, scales, rot, want_async=(flags & 7) != 0)
            H, W = cam.H, cam.W
            depth_sil, depth2, state = new(3, H, W), new(1, H, W), new(H * W)
            _check(_lib.vtgs_forward_shared(ctypes.byref(cam.c), n, dcol.data_ptr(), depth_sil.data_ptr(), depth2.data_ptr(),
                                            fs.workspace.data_ptr(), fs.workspace.numel(), fs.capacity, fs.tile_cap,
                                            state.data_ptr(), stream), "vtgs_forward_shared")
        ctx.fs, ctx.state, ctx.flags, ctx.n, ctx.dual, ctx.owned, ctx.n_map = fs, state, flags & 15, n, dual, owned, n_map
        if owned is not None:                       # the radii of the map: 0 outside the list (as on any band)
            radii = torch.zeros(n_map, dtype=torch.int32, device=dev).index_copy_(0, owned.idx64, radii)
        ctx.save_for_backward(means3D, rgb, unnorm_rot, logit_op, log_scales, cam_q, cam_t, depth_w2c,
                              means_cam, opac, scales, rot, dcol, im, depth_sil)
        ctx.set_materialize_grads(False)            # no zero-filled gradient for the radii output
        ctx.mark_non_differentiable(radii)
        return im, depth_sil, radii

    @staticmethod
    def backward(ctx, g_im, g_ds, _g_radii):
        (means3D, rgb, unnorm_rot, logit_op, log_scales, cam_q, cam_t, depth_w2c,
         means_cam, opac, scales, rot, dcol, im, depth_sil) = ctx.saved_tensors
        dev, n, flags, fs = means3D.device, ctx.n, ctx.flags, ctx.fs
        g_im = torch.zeros_like(im) if g_im is None else g_im.to(torch.float32).contiguous()
        g_ds = torch.zeros_like(depth_sil) if g_ds is None else g_ds.to(torch.float32).contiguous()
        if ctx.dual and os.environ.get("VTGS_FRAME_EPILOGUE", "1") != "0":
            # one pass, and the adjoint of vtgs_prepare_frame applied in the gather kernel: no dense operator gradients
            return _RenderFrame._backward_fused(ctx, g_im, g_ds)
        if ctx.dual:                                                # one pass; the geometry gradients arrive summed
            ga = _run_backward_dual(fs, means_cam, rgb, dcol, opac, scales, rot, im, depth_sil, g_im, g_ds)
            g_dcol, gb = ga[6], (None,) * 6
        else:
            ga = _run_backward(fs, means_cam, rgb, opac, scales, rot, im, g_im)
            fsb = _ForwardState()
            (fsb.cam, fsb.n, fsb.workspace, fsb.capacity, fsb.tile_cap, fsb.instances, fsb.image_state,
             fsb.key, fsb.pending) = fs.cam, fs.n, fs.workspace, fs.capacity, fs.tile_cap, fs.instances, ctx.state, fs.key, None
            gb = _run_backward(fsb, means_cam, dcol, opac, scales, rot, depth_sil, g_ds)
            g_dcol = gb[2]
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        ptr = lambda t: None if t is None else t.data_ptr()
        want_g, want_p, want_a = bool(flags & 1), bool(flags & 2), bool(flags & 4)
        g_means3D = new(n, 3) if want_g else None
        g_ur = new(n, 4) if want_g else None
        g_logit = new(n, 1) if want_a else None
        g_ls = new(n, 1) if want_a else None
        rows = int(_lib.vtgs_pose_partial_rows(n))
        partials = new(max(rows, 1), 12) if want_p else None
        if n > 0:
            _check(_lib.vtgs_prepare_frame_backward(
                n, flags & 7, means3D.data_ptr(), logit_op.data_ptr(), log_scales.data_ptr(), unnorm_rot.data_ptr(),
                cam_q.data_ptr(), cam_t.data_ptr(), depth_w2c.data_ptr(), ga[0].data_ptr(), ptr(gb[0]), g_dcol.data_ptr(),
                ga[3].data_ptr(), ptr(gb[3]), ga[4].data_ptr(), ptr(gb[4]), ga[5].data_ptr(), ptr(gb[5]),
                ptr(g_means3D), ptr(g_logit), ptr(g_ls), ptr(g_ur), ptr(partials), _stream_ptr(dev)),
                "vtgs_prepare_frame_backward")
        g_q = g_t = None
        if want_p and n == 0:                                     # nothing rendered: zero pose gradient, no launch
            g_q, g_t = torch.zeros(4, device=dev), torch.zeros(3, device=dev)
        elif want_p:                                              # 12 partial sums per workgroup -> dL/dq, dL/dt
            g_q, g_t = new(4), new(3)
            _check(_lib.vtgs_pose_gradient(partials.data_ptr(), rows, cam_q.data_ptr(), g_q.data_ptr(), g_t.data_ptr(),
                                           _stream_ptr(dev)), "vtgs_pose_gradient")
        return (g_means3D, ga[2] if want_a else None, g_ur, g_logit, g_ls, g_q, g_t, None, None, None, None)


def _backward_fused(ctx, g_im, g_ds):
    (means3D, rgb, unnorm_rot, logit_op, log_scales, cam_q, cam_t, depth_w2c,
     means_cam, opac, scales, rot, dcol, im, depth_sil) = ctx.saved_tensors
    dev, n, flags, fs, owned, n_map = means3D.device, ctx.n, ctx.flags, ctx.fs, ctx.owned, ctx.n_map
    # with a list the kernel writes the rows of the list only: the map-sized gradients start as zeros
    new = (lambda *s: torch.empty(s, dtype=torch.float32, device=dev)) if owned is None else \
          (lambda *s: torch.zeros(s, dtype=torch.float32, device=dev))
    ptr = lambda t: None if t is None else t.data_ptr()
    want_g, want_p, want_a = bool(flags & 1), bool(flags & 2), bool(flags & 4)
    g_means3D = new(n_map, 3) if want_g else None
    g_ur = new(n_map, 4) if want_g else None
    g_rgb = new(n_map, 3) if want_a else None
    g_logit = new(n_map, 1) if want_a else None
    g_ls = new(n_map, 1) if want_a else None
    rows = int(_lib.vtgs_pose_partial_rows(n))
    partials = torch.empty((max(rows, 1), 12), dtype=torch.float32, device=dev) if want_p else None
    g_q = g_t = None
    if n > 0:
        sbytes = _lib.vtgs_backward_dual_scratch_bytes(n, _scratch_instances(fs))
        scratch = _scratch(sbytes, dev)
        with _device_guard(dev):
          _check(_lib.vtgs_backward_dual_frame_owned(
            ctypes.byref(fs.cam.c), n, None if owned is None else owned.idx.data_ptr(), means_cam.data_ptr(), rgb.data_ptr(),
            dcol.data_ptr(), opac.data_ptr(),
            scales.data_ptr(), rot.data_ptr(), im.data_ptr(), depth_sil.data_ptr(), g_im.data_ptr(), g_ds.data_ptr(),
            fs.workspace.data_ptr(), fs.workspace.numel(), fs.capacity, fs.tile_cap, scratch.data_ptr(), sbytes, flags,
            means3D.data_ptr(), unnorm_rot.data_ptr(), cam_q.data_ptr(), cam_t.data_ptr(), depth_w2c.data_ptr(),
            ptr(g_rgb), ptr(g_means3D), ptr(g_logit), ptr(g_ls), ptr(g_ur), ptr(partials), _stream_ptr(dev)),
            "vtgs_backward_dual_frame_owned")
        _settle(fs)
        if want_p:                                                # 12 partial sums per workgroup -> dL/dq, dL/dt
            g_q, g_t = torch.empty(4, device=dev), torch.empty(3, device=dev)
            _check(_lib.vtgs_pose_gradient(partials.data_ptr(), rows, cam_q.data_ptr(), g_q.data_ptr(), g_t.data_ptr(),
                                           _stream_ptr(dev)), "vtgs_pose_gradient")
    elif want_p:                                                  # nothing rendered: zero pose gradient, no launch
        g_q, g_t = torch.zeros(4, device=dev), torch.zeros(3, device=dev)
    return (g_means3D, g_rgb, g_ur, g_logit, g_ls, g_q, g_t, None, None, None, None)


_RenderFrame._backward_fused = staticmethod(_backward_fused)


def _pose_tensors(params, time_idx: int, camera_grad: bool):
    """(cam_unnorm_rots [1,4,T], cam_trans [1,3,T], frame index) for the C++ node, which reads column `time_idx` in place and
    returns full-size gradients (round 6: no slot-gather / slot-scatter launches).  Any other layout: the pose as [1,4,1] /
    [1,3,1] tensors taken with plain indexing, frame index 0."""
    rots, trans = params["cam_unnorm_rots"], params["cam_trans"]
    if (rots.dim() == 3 and trans.dim() == 3 and rots.shape[0] == 1 and trans.shape[0] == 1 and rots.shape[1] == 4 and trans.shape[1] == 3
            and rots.shape[2] == trans.shape[2] and rots.dtype is torch.float32 and trans.dtype is torch.float32
            and rots.is_contiguous() and trans.is_contiguous() and rots.is_cuda and 0 <= int(time_idx) < rots.shape[2]):
        return (rots, trans, int(time_idx)) if camera_grad else (rots.detach(), trans.detach(), int(time_idx))
    q, t = rots[0, :, time_idx].reshape(1, 4, 1), trans[0, :, time_idx].reshape(1, 3, 1)
    q, t = q.to(torch.float32).contiguous(), t.to(torch.float32).contiguous()
    return (q, t, 0) if camera_grad else (q.detach(), t.detach(), 0)


def _render_frame_ext(means3D, rgb, unnorm_rot, logit_op, log_scales, pose, depth_w2c, cam: _Camera, flags: int, owned):
    """The same render through the C++ autograd node (csrc/vtgs_torch.cpp `RenderFrame`): the policy of `_run_forward` -- capacities,
    checked or run-ahead mode, the pinned result record, the retry after an overflow -- stays here, the per-call work and the
    whole backward run without the interpreter.  With the kernels of one band of the tile-row partition a rank's iteration is
    bound by this host path."""
    from . import (PLANNED, VTGS_ERR_INSTANCE_OVERFLOW, VTGS_FORWARD_ASYNC, VTGS_FORWARD_CHECKED, VTGS_FORWARD_SECOND_IS_DEPTH,
                   _FORWARD_MODE, _async_ok,
                   _caps_in_use, _choose_capacities, _drain, _ext, _forward_hints, _grow_after_overflow, _plan_for, _record_info,
                   _settle_after_backward, _slot_lock, _slot_pool)
    device = means3D.device
    n = int(means3D.shape[0]) if owned is None else int(owned.idx.numel())
    stream = _stream_ptr(device)
    key = (device.index, n, cam.W, cam.H, cam.band)
    fs = _ForwardState()
    fs.cam, fs.n, fs.image_state, fs.key, fs.pending = cam, n, None, key, None
    want_async = (flags & 7) != 0
    o = (None, None, None, None) if owned is None else (owned.idx, owned.idx64, owned.mask, owned.escapes)
    with _slot_lock:
        pool = _slot_pool(device, stream)
        _drain(pool)
        slot = pool.take(fs)
        capacity, tile_cap = _choose_capacities(key, n)
        run_ahead = want_async and _FORWARD_MODE == "auto" and _async_ok.get(key) == (capacity, tile_cap)
        info = pool.info[slot]
        for _attempt in range(6):
            info.complete = 0
            plan = _plan_for(key, device, tile_cap).data_ptr() if tile_cap & PLANNED else 0
            im, depth_sil, radii, workspace, status = _ext.render_frame(
                means3D, rgb, unnorm_rot, logit_op, log_scales, pose[0], pose[1], pose[2], depth_w2c, cam.bytes, cam.bg, cam.view, cam.proj, capacity,
                tile_cap, plan, pool.ptr[slot], (VTGS_FORWARD_ASYNC if run_ahead else VTGS_FORWARD_CHECKED) | _forward_hints(key, tile_cap)
                | (VTGS_FORWARD_SECOND_IS_DEPTH if flags & 16 else 0), flags & 15, stream, *o)
            if run_ahead:
                break
            if int(status) == VTGS_ERR_INSTANCE_OVERFLOW:     # the record says what is needed: grow whichever was short
                capacity, tile_cap = _grow_after_overflow(key, n, device, info, capacity, tile_cap, workspace)
                continue
            break
        else:
            raise RuntimeError("vtgs_forward_dual: instance capacity kept overflowing")
        fs.workspace, fs.capacity, fs.tile_cap = workspace, capacity, tile_cap
        if run_ahead:
            fs._instances = None
            fs.pending = (pool, slot, device, stream)
            pool.pending.append(fs)
            _settle_after_backward(im, fs)             # (the record is read inside loss.backward(), before any optimizer step)
        else:
            pool.owner[slot] = None
            _caps_in_use.setdefault(key, (capacity, tile_cap))
            _record_info(key, n, cam.W, cam.H, capacity, info)
            fs._instances = int(info.instances_needed)
    return im, depth_sil, radii


def render_frame(params: Dict[str, torch.Tensor], time_idx: int, raster_settings, first_frame_w2c: torch.Tensor,
                 gaussians_grad: bool, camera_grad: bool, radius_rule: Optional[str] = None, tile_rows=None, owned=None,
                 get_loss_contract: bool = False):
    """RGB render + [z,1,z^2] render of frame `time_idx` (see module docstring).  Returns (im [3,H,W],
    depth_sil [3,H,W], radii [N] int32).  `tile_rows=(begin, end)`: this rank's band of 16-pixel tile rows (multi-GPU
    partition, `partition.band_for_rank`): pixels outside the band come back as zero and the gradients are the band's
    share -- the pose gradient of a rank is then 7 floats to all-reduce, with no dense per-Gaussian array behind it.
    `owned` (a `partition.OwnedSet` built for the same band): only the Gaussians of the list are transformed, projected and
    binned, and only their gradients are gathered; the set checks on the device, before every render, that no Gaussian
    outside the list could meet the band (`owned.escaped()` reads the count).
    `get_loss_contract=True` is the caller's PROMISE to treat `depth_sil` as get_loss does (src/vtgaussian_slam.py:466-521):
    the silhouette (plane 1) only feeds comparisons, z^2 (plane 2) only the NaN test of a detached uncertainty
    isnan(plane 2 - plane 0^2), and the gradient sent back into `depth_sil` is zero outside plane 0.  Then (include/vtgs.h:
    VTGS_FORWARD_SECOND_IS_DEPTH, frame flag 8) the forward is the single render's kernel with z in its depth column --
    plane 1 = 1 - T_final, which is what sum w equals up to float32 rounding, plane 2 = plane 0 squared, NaN exactly where the
    true difference is -- and the backward carries four image-gradient channels instead of six.  A caller that reads
    planes 1 / 2 in any other way, or differentiates through them, must leave this off."""
    dev = params["means3D"].device
    if dev.type != "cuda":
        raise RuntimeError("render_frame needs tensors on a HIP device (torch 'cuda'); no CPU path exists")
    if owned is not None and params["log_scales"].shape[1] != 1:
        raise RuntimeError("owned sets are built for isotropic maps (log_scales [N,1], every reference config)")
    if params["log_scales"].shape[1] != 1:
        # anisotropic maps (log_scales [N,3]: no reference config, but transform_to_frame has the branch,
        # utils/slam_helpers.py:376-383 -- the rotations are composed with the camera's): the pose transform and the
        # render-variable builders run as the reference's own element-wise chain, both renders on the HIP operator with
        # the second one over the first one's bins
        im, depth_sil, radii, _ = render_frame_unfused(params, time_idx, raster_settings, first_frame_w2c, gaussians_grad,
                                                       camera_grad, radius_rule, tile_rows)
        return im, depth_sil, radii
    import os
    rule = _RADIUS_RULES[radius_rule or os.environ.get("VTGS_RADIUS_RULE", "3sigma")]
    cam = _camera_for(raster_settings, dev, rule, None if tile_rows is None else (int(tile_rows[0]), int(tile_rows[1])))
    # like the reference, gaussians_grad=False detaches only the geometry (means3D, unnorm_rotations); colours,
    # opacities and scales keep their gradient whenever they require one (utils/slam_helpers.py:362-367, 152-159)
    g = lambda x: x if gaussians_grad else x.detach()
    grad_on = torch.is_grad_enabled()
    appearance = any(params[k].requires_grad for k in ("rgb_colors", "logit_opacities", "log_scales"))
    flags = ((1 if gaussians_grad and grad_on else 0) | (2 if camera_grad and grad_on else 0)
             | (4 if appearance and grad_on else 0))
    if get_loss_contract:                           # 8: frame flag of the backward; 16: this module's mark for the forward flag
        flags |= 16 | (8 if flags else 0)
    if owned is not None:
        if not owned.scales_are_log:
            raise ValueError("this owned set was built for the plain operator (OwnedSet.for_operator)")
        owned.admit(params["means3D"].shape[0], tile_rows)
    from . import _ext
    if (_ext is not None and hasattr(_ext, "render_frame") and os.environ.get("VTGS_DUAL", "1") != "0"
            and os.environ.get("VTGS_FRAME_EPILOGUE", "1") != "0" and os.environ.get("VTGS_FUSED_EXT", "1") != "0"
            and not torch.cuda.is_current_stream_capturing()):
        return _render_frame_ext(g(params["means3D"]), params["rgb_colors"], g(params["unnorm_rotations"]), params["logit_opacities"],
                                 params["log_scales"], _pose_tensors(params, time_idx, camera_grad), first_frame_w2c.to(dev), cam, flags,
                                 owned)
    q, t = _pose_of_frame(params, time_idx, camera_grad)
    return _RenderFrame.apply(g(params["means3D"]), params["rgb_colors"], g(params["unnorm_rotations"]),
                              params["logit_opacities"], params["log_scales"], q, t,
                              first_frame_w2c.to(dev), cam, flags, owned)


def render_frame_unfused(params, time_idx: int, raster_settings, first_frame_w2c, gaussians_grad: bool, camera_grad: bool,
                         radius_rule: Optional[str] = None, tile_rows=None):
    """The reference's own chain (src/vtgaussian_slam.py:431-468) on the plain operator: `transform_to_frame`, the two
    render-variable builders, the colour render and the [z,1,z^2] render over the same bins.  Returns (im, depth_sil, radii,
    means2D): the colour render keeps `means2D` in the graph (its .grad is the screen-space gradient of the COLOUR render
    only, :460-462).  Any map, isotropic or not."""
    import slam_callers as sc                      # device-agnostic restatement of utils/slam_helpers.py (pinned by golden vectors)
    from . import GaussianRasterizer
    tg = sc.transform_to_frame(params, time_idx, gaussians_grad=gaussians_grad, camera_grad=camera_grad)
    rv = sc.transformed_params2rendervar(params, tg)
    dv = sc.transformed_params2depthplussilhouette(params, first_frame_w2c, tg)
    if rv["means2D"].requires_grad:
        rv["means2D"].retain_grad()
    rast = GaussianRasterizer(raster_settings=raster_settings, radius_rule=radius_rule, tile_rows=tile_rows)
    im, radius, _ = rast(**rv)
    depth_sil, _ = rast.render_shared(dv["colors_precomp"], like=(rv["means3D"], dv["means2D"], rv["opacities"], rv["scales"],
                                                                  rv["rotations"]))
    return im, depth_sil, radius, rv["means2D"]
