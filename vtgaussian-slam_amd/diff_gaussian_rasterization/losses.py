"""Loss kernels: `tracking_loss` / `mapping_loss` == the Replica branches of get_loss
(src/vtgaussian_slam.py:519-608, 678-679) with the masked L1 sums and their gradients in one pass (vtgs_masked_l1);
`fused_ssim(img1, img2)` == utils/slam_external.py:66-97 `calc_ssim(img1, img2)` (mean
SSIM, 11x11 Gaussian window, sigma 1.5, zero padding, per channel) as one HIP kernel each way instead of ten grouped
convolutions.  img2 is treated as a constant (the ground-truth image at the reference's call site,
src/vtgaussian_slam.py:608)."""
from __future__ import annotations

import ctypes

import torch

from . import _I32, _P, _check, _lib, _stream_ptr

_lib.vtgs_ssim_partial_rows.restype, _lib.vtgs_ssim_partial_rows.argtypes = ctypes.c_uint32, [_I32, _I32, _I32]
_lib.vtgs_ssim_forward.restype, _lib.vtgs_ssim_forward.argtypes = ctypes.c_int, [_P, _P, _I32, _I32, _I32, _P, _P, _P]
_lib.vtgs_ssim_backward.restype, _lib.vtgs_ssim_backward.argtypes = ctypes.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P]


class _FusedSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        if not img1.is_cuda:
            raise RuntimeError("fused_ssim needs tensors on a HIP device (torch 'cuda'); no CPU path exists")
        a = img1.detach().to(torch.float32).contiguous()
        b = img2.detach().to(torch.float32).contiguous()
        C, H, W = a.shape[-3], a.shape[-2], a.shape[-1]
        rows = int(_lib.vtgs_ssim_partial_rows(C, H, W))
        partial = torch.empty(rows, dtype=torch.float32, device=a.device)
        need = ctx.needs_input_grad[0]
        gmaps = torch.empty((3, C, H, W), dtype=torch.float32, device=a.device) if need else None
        _check(_lib.vtgs_ssim_forward(a.data_ptr(), b.data_ptr(), C, H, W, partial.data_ptr(),
                                      gmaps.data_ptr() if need else None, _stream_ptr(a.device)), "vtgs_ssim_forward")
        ctx.save_for_backward(a, b, gmaps if need else a)
        ctx.need = need
        return partial.sum() / float(C * H * W)

    @staticmethod
    def backward(ctx, g):
        if not ctx.need:
            return None, None
        a, b, gmaps = ctx.saved_tensors
        C, H, W = a.shape[-3], a.shape[-2], a.shape[-1]
        up = g.detach().to(torch.float32).reshape(1).contiguous()
        out = torch.empty_like(a)
        _check(_lib.vtgs_ssim_backward(a.data_ptr(), b.data_ptr(), gmaps.data_ptr(), up.data_ptr(), C, H, W, out.data_ptr(),
                                       _stream_ptr(a.device)), "vtgs_ssim_backward")
        return out, None


def fused_ssim(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """Mean SSIM of two [C,H,W] images; differentiable with respect to img1."""
    return _FusedSSIM.apply(img1, img2)


_lib.vtgs_seen_and_max_radius.restype, _lib.vtgs_seen_and_max_radius.argtypes = ctypes.c_int, [_I32, _P, _P, _P, _P]


def seen_and_max_radius(radius: torch.Tensor, max_2d_radius: torch.Tensor, seen: torch.Tensor) -> None:
    """`seen = radius > 0` and `max_2D_radius = max(max_2D_radius, radius)` (src/vtgaussian_slam.py:681-689) in one launch:
    radius int32 [N], max_2d_radius float32 [N] updated in place, seen bool [N] written."""
    if not radius.is_cuda:
        raise RuntimeError("seen_and_max_radius needs tensors on a HIP device (torch 'cuda'); no CPU path exists")
    _check(_lib.vtgs_seen_and_max_radius(radius.numel(), radius.data_ptr(), max_2d_radius.data_ptr(), seen.data_ptr(),
                                         _stream_ptr(radius.device)), "vtgs_seen_and_max_radius")


_lib.vtgs_masked_l1_partial_rows.restype, _lib.vtgs_masked_l1_partial_rows.argtypes = ctypes.c_uint32, [_I32]
_lib.vtgs_masked_l1.restype = ctypes.c_int
_lib.vtgs_masked_l1.argtypes = [_P, _P, _P, _P, _I32, ctypes.c_float, _I32, _P, _P, _P, _P]


class _MaskedL1(torch.autograd.Function):
    """(sum |gt_im - im|, sum |gt_depth - depth|, mask count) with the masks of get_loss; differentiable in im, depth_sil."""

    @staticmethod
    def forward(ctx, im, depth_sil, gt_im, gt_depth, sil_thres: float, mode: int):
        if not im.is_cuda:
            raise RuntimeError("the fused losses need tensors on a HIP device (torch 'cuda'); no CPU path exists")
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        a, d, ga, gd = f32(im), f32(depth_sil), f32(gt_im), f32(gt_depth)
        P = a.shape[-1] * a.shape[-2]
        rows = int(_lib.vtgs_masked_l1_partial_rows(P))
        partial = torch.empty((rows, 3), dtype=torch.float32, device=a.device)
        g_im, g_ds = torch.empty_like(a), torch.empty_like(d)
        _check(_lib.vtgs_masked_l1(a.data_ptr(), d.data_ptr(), ga.data_ptr(), gd.data_ptr(), P, float(sil_thres), int(mode),
                                   partial.data_ptr(), g_im.data_ptr(), g_ds.data_ptr(), _stream_ptr(a.device)),
               "vtgs_masked_l1")
        ctx.save_for_backward(g_im, g_ds)
        s = partial.sum(0)
        ctx.mark_non_differentiable(s[2:3])
        return s[0], s[1], s[2]

    @staticmethod
    def backward(ctx, g_sum_im, g_sum_depth, _g_count):
        g_im, g_ds = ctx.saved_tensors
        return (None if g_sum_im is None else g_im * g_sum_im, None if g_sum_depth is None else g_ds * g_sum_depth,
                None, None, None, None)


_lib.vtgs_loss_scratch_floats.restype, _lib.vtgs_loss_scratch_floats.argtypes = ctypes.c_size_t, [_I32, _I32]
_lib.vtgs_slam_loss_forward.restype = ctypes.c_int
_lib.vtgs_slam_loss_forward.argtypes = [_I32, _P, _P, _P, _P, _I32, _I32, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P, _P,
                                        _P, _P, _P]
_lib.vtgs_slam_loss_backward.restype = ctypes.c_int
_lib.vtgs_slam_loss_backward.argtypes = [_I32, _P, _P, _P, _P, _I32, _I32, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P, _P,
                                         _P, _P, _P, _P, _P]
_lib.vtgs_slam_loss_forward_seen.restype = ctypes.c_int
_lib.vtgs_slam_loss_forward_seen.argtypes = [_I32, _P, _P, _P, _P, _I32, _I32, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P, _P,
                                             _P, _P, _I32, _P, _P, _P, _P]


class _SlamLoss(torch.autograd.Function):
    """The whole Replica branch of get_loss as one node: 2-3 launches for the value, 1-2 for both gradient images (the
    upstream gradient is read on the device).  mode 0 = tracking, 1 = mapping, 2 = tracking with the unmasked colour sum."""

    @staticmethod
    def forward(ctx, im, depth_sil, gt_im, gt_depth, mode: int, sil_thres: float, w_im: float, w_depth: float,
                extra_mask=None, color_weight=None, bookkeeping=None):
        # bookkeeping = (radius int32 [N], max_2D_radius float32 [N], seen bool [N]) of the render this loss belongs to: get_loss's
        # `seen` / running-maximum update (src/vtgaussian_slam.py:681-689) rides in the loss's last launch (round 6)
        if not im.is_cuda:
            raise RuntimeError("the fused losses need tensors on a HIP device (torch 'cuda'); no CPU path exists")
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        a, d, ga, gd = f32(im), f32(depth_sil), f32(gt_im), f32(gt_depth)
        if a.shape[-3] != 3 or d.shape[-3] != 3:
            raise ValueError("im and depth_sil must be [3,H,W]")
        H, W = a.shape[-2], a.shape[-1]
        em = None if extra_mask is None else extra_mask.detach().to(device=a.device, dtype=torch.float32).reshape(-1).contiguous()
        cw = None if color_weight is None else color_weight.detach().to(device=a.device, dtype=torch.float32).expand(3, H, W).contiguous()
        if em is not None and em.numel() != H * W:
            raise ValueError("extra_mask must have H*W elements")
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        scratch = torch.empty(int(_lib.vtgs_loss_scratch_floats(H, W)), dtype=torch.float32, device=a.device)
        gmaps = torch.empty((3, 3, H, W), dtype=torch.float32, device=a.device) if (mode == 1 and need) else None
        out = torch.empty(8, dtype=torch.float32, device=a.device)
        n_bk, bk = (0, (None, None, None)) if bookkeeping is None else (int(bookkeeping[0].numel()), [t.data_ptr() for t in bookkeeping])
        _check(_lib.vtgs_slam_loss_forward_seen(mode, a.data_ptr(), d.data_ptr(), ga.data_ptr(), gd.data_ptr(), H, W,
                                                float(sil_thres), float(w_im), float(w_depth), scratch.data_ptr(),
                                                None if gmaps is None else gmaps.data_ptr(), out.data_ptr(),
                                                None if em is None else em.data_ptr(), None if cw is None else cw.data_ptr(),
                                                n_bk, bk[0], bk[1], bk[2], _stream_ptr(a.device)), "vtgs_slam_loss_forward_seen")
        ctx.em, ctx.cw = em, cw
        ctx.save_for_backward(a, d, ga, gd, out, gmaps if gmaps is not None else out)
        ctx.cfg = (mode, float(sil_thres), float(w_im), float(w_depth), need, gmaps is not None)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(out)
        return out[0], out              # out = {loss, mask count, sum |gt_im - im| (weighted), sum |gt_depth - depth|, mean SSIM,
                                        #        weighted colour term, weighted depth term, 0}

    @staticmethod
    def backward(ctx, g, _g_terms=None):
        mode, sil_thres, w_im, w_depth, need, has_maps = ctx.cfg
        if g is None or not need:
            return (None,) * 11
        a, d, ga, gd, out, gmaps = ctx.saved_tensors
        H, W = a.shape[-2], a.shape[-1]
        up = g.detach().to(torch.float32).reshape(1).contiguous()
        g_im, g_ds = torch.empty_like(a), torch.empty_like(d)
        _check(_lib.vtgs_slam_loss_backward(mode, a.data_ptr(), d.data_ptr(), ga.data_ptr(), gd.data_ptr(), H, W, sil_thres,
                                            w_im, w_depth, gmaps.data_ptr() if has_maps else None, out.data_ptr(),
                                            up.data_ptr(), g_im.data_ptr(), g_ds.data_ptr(),
                                            None if ctx.em is None else ctx.em.data_ptr(),
                                            None if ctx.cw is None else ctx.cw.data_ptr(), _stream_ptr(a.device)),
               "vtgs_slam_loss_backward")
        return g_im, g_ds, None, None, None, None, None, None, None, None, None


def tracking_loss(im, depth_sil, gt_im, gt_depth, sil_thres: float, w_im: float = 0.5, w_depth: float = 0.025,
                  extra_mask=None, return_terms: bool = False, colour_over_all_pixels: bool = False, bookkeeping=None):
    """Tracking loss of get_loss (src/vtgaussian_slam.py:519-605): w_im * masked L1 SUM of colour + w_depth * masked L1
    SUM of depth over gt_depth > 0 & finite & silhouette > sil_thres [& extra_mask].  `extra_mask` [H,W] / [1,H,W]
    (bool or float, detached) carries the masks of the TUM / ScanNet / ScanNet++ branches -- build it with
    `visibility_mask`, `far_depth_mask`, `outlier_depth_mask` below and AND them together.
    return_terms: also the detached device vector {loss, mask count, colour sum, depth sum, -, w_im * colour term,
    w_depth * depth term, 0}.  colour_over_all_pixels: the branch with neither use_sil_for_loss nor
    ignore_outlier_depth_loss (:601-602) -- the colour sum ignores the mask; w_depth = 0: use_l1 = False (:591-596)."""
    loss, terms = _SlamLoss.apply(im, depth_sil, gt_im, gt_depth, 2 if colour_over_all_pixels else 0, sil_thres, w_im, w_depth,
                                  extra_mask, None, bookkeeping)
    return (loss, terms) if return_terms else loss


def mapping_loss(im, depth_sil, gt_im, gt_depth, w_im: float = 1.0, w_depth: float = 1.0, extra_mask=None,
                 additional_mask=None, return_terms: bool = False, bookkeeping=None):
    """Mapping loss (src/vtgaussian_slam.py:592-611): w_depth * masked L1 MEAN of depth + w_im * (0.8 * L1 mean + 0.2 *
    (1 - SSIM)) of colour; with `additional_mask` the colour L1 becomes mean(|im - gt| * (10 * additional_mask + 0.8))
    (l1_loss_v1_mask, utils/slam_helpers.py:8-9).  `extra_mask`: the outlier-depth mask when ignore_outlier_depth_loss."""
    cw = None if additional_mask is None else 10.0 * additional_mask.to(torch.float32) + 0.8
    loss, terms = _SlamLoss.apply(im, depth_sil, gt_im, gt_depth, 1, 0.0, w_im, w_depth, extra_mask, cw, bookkeeping)
    return (loss, terms) if return_terms else loss


# ---- one band of the tile-row partition: the same loss, summed over the ranks ---------------------------
_lib.vtgs_slam_loss_band_sums.restype = ctypes.c_int
_lib.vtgs_slam_loss_band_sums.argtypes = [_I32, _P, _P, _P, _P, _I32, _I32, _I32, _I32, ctypes.c_float, _P, _P, _P, _P, _P, _P]
_lib.vtgs_slam_loss_band_share.restype = ctypes.c_int
_lib.vtgs_slam_loss_band_share.argtypes = [_I32, _P, _P, _I32, _I32, ctypes.c_float, ctypes.c_float, _I32, _I32, _P, _P]
_lib.vtgs_slam_loss_band_backward.restype = ctypes.c_int
_lib.vtgs_slam_loss_band_backward.argtypes = [_I32, _P, _P, _P, _P, _I32, _I32, _I32, _I32, ctypes.c_float, ctypes.c_float,
                                              ctypes.c_float, _P, _P, _P, _P, _P, _P, _P, _P]


class _BandLoss(torch.autograd.Function):
    """A band's share of the get_loss value (see include/vtgs.h, "ONE BAND"): value 3-4 launches + the caller's reduction
    of eight floats, gradient images 1-2 launches."""

    @staticmethod
    def forward(ctx, im, depth_sil, gt_im, gt_depth, rows, mode: int, sil_thres: float, w_im: float, w_depth: float,
                extra_mask, color_weight, reduce, first_band: bool):
        if not im.is_cuda:
            raise RuntimeError("the fused losses need tensors on a HIP device (torch 'cuda'); no CPU path exists")
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        a, d, ga, gd = f32(im), f32(depth_sil), f32(gt_im), f32(gt_depth)
        if a.shape[-3] != 3 or d.shape[-3] != 3:
            raise ValueError("im and depth_sil must be [3,H,W]")
        H, W = a.shape[-2], a.shape[-1]
        r0, r1 = int(rows[0]), int(rows[1])
        if not 0 <= r0 < r1 <= H:
            raise ValueError(f"rows {rows} outside an image of {H} rows")
        em = None if extra_mask is None else extra_mask.detach().to(device=a.device, dtype=torch.float32).reshape(-1).contiguous()
        cw = None if color_weight is None else color_weight.detach().to(device=a.device, dtype=torch.float32).expand(3, H, W).contiguous()
        if em is not None and em.numel() != H * W:
            raise ValueError("extra_mask must have H*W elements (the full frame)")
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        st = _stream_ptr(a.device)
        scratch = torch.empty(int(_lib.vtgs_loss_scratch_floats(H, W)), dtype=torch.float32, device=a.device)
        gmaps = torch.empty((3, 3, H, W), dtype=torch.float32, device=a.device) if (mode == 1 and need) else None
        own = torch.empty(8, dtype=torch.float32, device=a.device)
        _check(_lib.vtgs_slam_loss_band_sums(mode, a.data_ptr(), d.data_ptr(), ga.data_ptr(), gd.data_ptr(), H, W, r0, r1,
                                             float(sil_thres), scratch.data_ptr(), None if gmaps is None else gmaps.data_ptr(),
                                             own.data_ptr(), None if em is None else em.data_ptr(),
                                             None if cw is None else cw.data_ptr(), st), "vtgs_slam_loss_band_sums")
        if reduce is None:
            total = own
        else:
            total = own.clone()
            got = reduce(total)                 # in place (dist.all_reduce) or returning the reduced tensor
            total = total if got is None else got
        out = torch.empty(8, dtype=torch.float32, device=a.device)
        _check(_lib.vtgs_slam_loss_band_share(mode, own.data_ptr(), total.data_ptr(), H, W, float(w_im), float(w_depth),
                                              0 if cw is None else 1, 1 if first_band else 0, out.data_ptr(), st),
               "vtgs_slam_loss_band_share")
        ctx.em, ctx.cw = em, cw
        ctx.save_for_backward(a, d, ga, gd, out, gmaps if gmaps is not None else out)
        ctx.cfg = (mode, float(sil_thres), float(w_im), float(w_depth), need, gmaps is not None, r0, r1)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(out, total)
        return out[0], out, total

    @staticmethod
    def backward(ctx, g, _g_terms=None, _g_total=None):
        mode, sil_thres, w_im, w_depth, need, has_maps, r0, r1 = ctx.cfg
        if g is None or not need:
            return (None,) * 13
        a, d, ga, gd, out, gmaps = ctx.saved_tensors
        H, W = a.shape[-2], a.shape[-1]
        up = g.detach().to(torch.float32).reshape(1).contiguous()
        g_im, g_ds = torch.zeros_like(a), torch.zeros_like(d)      # rows outside the band (+ SSIM halo) stay zero
        _check(_lib.vtgs_slam_loss_band_backward(mode, a.data_ptr(), d.data_ptr(), ga.data_ptr(), gd.data_ptr(), H, W, r0, r1,
                                                 sil_thres, w_im, w_depth, gmaps.data_ptr() if has_maps else None,
                                                 out.data_ptr(), up.data_ptr(), g_im.data_ptr(), g_ds.data_ptr(),
                                                 None if ctx.em is None else ctx.em.data_ptr(),
                                                 None if ctx.cw is None else ctx.cw.data_ptr(), _stream_ptr(a.device)),
               "vtgs_slam_loss_band_backward")
        return (g_im, g_ds) + (None,) * 11


def band_loss(im, depth_sil, gt_im, gt_depth, rows, mode: str = "mapping", sil_thres: float = 0.0, w_im: float = 1.0,
              w_depth: float = 1.0, extra_mask=None, additional_mask=None, reduce=None, first_band: bool = True,
              colour_over_all_pixels: bool = False, return_terms: bool = False):
    """This band's share of `tracking_loss` / `mapping_loss` (mode "tracking" / "mapping") over pixel rows
    rows = (begin, end) of full-frame [3,H,W] images: the shares of all bands sum to the full-frame loss, the gradients of
    the shares are the full-frame gradients restricted to what this band's terms touch.  `reduce(t)`: sums the 8-float
    device tensor `t` over the ranks, in place or returning the result (`torch.distributed.all_reduce`); None = a single
    band.  Mapping needs rows [begin-5, end+5) of `im` filled with the neighbours' renders (partition.halo_exchange) and
    `first_band=True` on exactly one rank.  return_terms: (share, the 8-float record of the share, the reduced sums
    {colour sum, depth sum, mask count, SSIM sum})."""
    if mode == "tracking":
        m, cw = (2 if colour_over_all_pixels else 0), None
    elif mode == "mapping":
        m, cw = 1, (None if additional_mask is None else 10.0 * additional_mask.to(torch.float32) + 0.8)
    else:
        raise ValueError("mode is 'tracking' or 'mapping'")
    loss, terms, total = _BandLoss.apply(im, depth_sil, gt_im, gt_depth, rows, m, sil_thres, w_im, w_depth, extra_mask, cw,
                                         reduce, first_band)
    return (loss, terms, total) if return_terms else loss


# ---- detached masks of the TUM / ScanNet / ScanNet++ branches (device-side torch ops: plumbing, no gradients) ----------
def outlier_depth_mask(gt_depth, depth):
    """src/vtgaussian_slam.py:525-528 (ignore_outlier_depth_loss): |gt - depth| (0 where gt <= 0) below 50 x its median,
    and gt > 0.  gt_depth, depth: [1,H,W]."""
    err = torch.abs(gt_depth - depth.detach()) * (gt_depth > 0)
    return (err < 50 * err.median()) & (gt_depth > 0)


def far_depth_mask(gt_depth, far_depth_filter_thres: float):
    """src/vtgaussian_slam.py:586-588: gt_depth < far_depth_filter_thres."""
    return gt_depth < far_depth_filter_thres


def visibility_mask(gt_depth, intrinsics, curr_w2c, overlaps, vis_mask_thres: float = 0.05):
    """src/vtgaussian_slam.py:536-584 with get_vis_mask (:376-404): every pixel (gt_depth >= 0) is back-projected with
    the current pose estimate, projected into each overlapping base frame (w2c, gt depth [1,H,W]) and kept if the
    bilinearly sampled depth there agrees with its own depth in that frame to vis_mask_thres (relative to the smaller of
    the two); the masks of the overlapping frames are ORed (one frame for TUM, first / mid / last for ScanNet(++)).
    Returns [H,W] bool on gt_depth's device."""
    import torch.nn.functional as F
    H, W = gt_depth.shape[-2], gt_depth.shape[-1]
    dev = gt_depth.device
    k = intrinsics.to(dev, torch.float32)
    ys, xs = torch.where(gt_depth[0] >= 0)
    z = gt_depth[0, ys, xs]
    pts_cam = torch.stack(((xs - k[0, 2]) / k[0, 0] * z, (ys - k[1, 2]) / k[1, 1] * z, z), dim=-1)
    pts4 = torch.cat([pts_cam, torch.ones_like(pts_cam[:, :1])], dim=1)
    pts = (torch.inverse(curr_w2c.to(dev, torch.float32)) @ pts4.T).T[:, :3]
    out = None
    for w2c, gd in overlaps:
        p4 = torch.cat([pts, torch.ones_like(pts[:, :1])], dim=1)
        tp = (w2c.to(dev, torch.float32) @ p4.T).T[:, :3]
        p2 = (k @ tp.T).T
        pz = p2[:, 2:] + 1e-5
        uv = (p2 / pz)[:, :2]
        grid = torch.stack((uv[:, 0] / (W - 1) * 2.0 - 1.0, uv[:, 1] / (H - 1) * 2.0 - 1.0), dim=-1).reshape(1, 1, -1, 2)
        samp = F.grid_sample(gd.to(dev, torch.float32).reshape(1, 1, H, W), grid, padding_mode="zeros", align_corners=True).reshape(-1)
        vis = (torch.abs(samp - pz[:, 0]) < vis_mask_thres * torch.minimum(samp, pz[:, 0])).reshape(H, W)
        out = vis if out is None else (out | vis)
    return out


_lib.vtgs_silhouette_sweep.restype = ctypes.c_int
_lib.vtgs_silhouette_sweep.argtypes = [_P, _P, _P, _P, _I32, ctypes.POINTER(ctypes.c_float), _I32, _P, _P]
_lib.vtgs_silhouette_sweep_band.restype = ctypes.c_int
_lib.vtgs_silhouette_sweep_band.argtypes = [_P, _P, _P, _P, _I32, _I32, _I32, ctypes.POINTER(ctypes.c_float), _I32, _P, _P]


def silhouette_sweep(im, silhouette, gt_im, gt_depth, candidates, rows=None):
    """Per candidate c: (sum over channels of (gt_im - im)^2, pixel count) over silhouette > c & gt_depth > 0, as a
    [K,2] float64 tensor on the device (one kernel; src/vtgaussian_slam.py:476-496).  rows = (begin, end): only those
    pixel rows (a band of the tile-row partition; the sums of all bands add up to the full-frame ones)."""
    if not im.is_cuda:
        raise RuntimeError("the fused losses need tensors on a HIP device (torch 'cuda'); no CPU path exists")
    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    a, s, ga, gd = f32(im), f32(silhouette), f32(gt_im), f32(gt_depth)
    K = len(candidates)
    if not 1 <= K <= 8:
        raise ValueError("1..8 candidate thresholds")
    W = a.shape[-1]
    P = W * a.shape[-2]
    p0, p1 = (0, P) if rows is None else (int(rows[0]) * W, int(rows[1]) * W)
    if not 0 <= p0 < p1 <= P:
        raise ValueError(f"rows {rows} outside the image")
    partial = torch.empty((int(_lib.vtgs_masked_l1_partial_rows(p1 - p0)), K, 2), dtype=torch.float32, device=a.device)
    th = (ctypes.c_float * K)(*[float(c) for c in candidates])
    _check(_lib.vtgs_silhouette_sweep_band(a.data_ptr(), s.data_ptr(), ga.data_ptr(), gd.data_ptr(), P, p0, p1, th, K,
                                           partial.data_ptr(), _stream_ptr(a.device)), "vtgs_silhouette_sweep_band")
    return partial.sum(0, dtype=torch.float64)


def best_silhouette_threshold(im, silhouette, gt_im, gt_depth, candidates=(0.990, 0.993, 0.995, 0.997, 0.999),
                              return_mse: bool = False):
    """Tracking iteration 0 of the Replica branch: the candidate with the smallest masked colour MSE, first one on ties
    (src/vtgaussian_slam.py:472-510) -- one kernel and one host read instead of five masked gathers and five .item().
    return_mse: (threshold, its MSE), the pair the reference appends to sil_thres_ls / presence_sil_mask_mse_ls."""
    sums = silhouette_sweep(im, silhouette, gt_im, gt_depth, candidates).cpu()
    best, best_mse = candidates[0], float("inf")
    for k, c in enumerate(candidates):
        cnt = float(sums[k, 1])
        mse = float(sums[k, 0]) / (3.0 * cnt) if cnt > 0 else float("inf")      # an empty mask gives NaN in the reference
        if mse < best_mse:
            best, best_mse = c, mse
    return (best, best_mse) if return_mse else best
