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


def tracking_loss(im, depth_sil, gt_im, gt_depth, sil_thres: float, w_im: float = 0.5, w_depth: float = 0.025):
    """Replica tracking loss: w_im * masked L1 SUM of colour + w_depth * masked L1 SUM of depth."""
    s_im, s_d, _ = _MaskedL1.apply(im, depth_sil, gt_im, gt_depth, sil_thres, 0)
    return w_im * s_im + w_depth * s_d


def mapping_loss(im, depth_sil, gt_im, gt_depth, w_im: float = 1.0, w_depth: float = 1.0):
    """Mapping loss: w_depth * masked L1 MEAN of depth + w_im * (0.8 * L1 mean + 0.2 * (1 - SSIM)) of colour."""
    s_im, s_d, cnt = _MaskedL1.apply(im, depth_sil, gt_im, gt_depth, 0.0, 1)
    l_im = 0.8 * s_im / float(im.numel()) + 0.2 * (1.0 - fused_ssim(im, gt_im))
    return w_im * l_im + w_depth * s_d / cnt


_lib.vtgs_silhouette_sweep.restype = ctypes.c_int
_lib.vtgs_silhouette_sweep.argtypes = [_P, _P, _P, _P, _I32, ctypes.POINTER(ctypes.c_float), _I32, _P, _P]


def silhouette_sweep(im, silhouette, gt_im, gt_depth, candidates):
    """Per candidate c: (sum over channels of (gt_im - im)^2, pixel count) over silhouette > c & gt_depth > 0, as a
    [K,2] float64 tensor on the device (one kernel; src/vtgaussian_slam.py:476-496)."""
    if not im.is_cuda:
        raise RuntimeError("the fused losses need tensors on a HIP device (torch 'cuda'); no CPU path exists")
    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    a, s, ga, gd = f32(im), f32(silhouette), f32(gt_im), f32(gt_depth)
    K = len(candidates)
    if not 1 <= K <= 8:
        raise ValueError("1..8 candidate thresholds")
    P = a.shape[-1] * a.shape[-2]
    rows = int(_lib.vtgs_masked_l1_partial_rows(P))
    partial = torch.empty((rows, K, 2), dtype=torch.float32, device=a.device)
    th = (ctypes.c_float * K)(*[float(c) for c in candidates])
    _check(_lib.vtgs_silhouette_sweep(a.data_ptr(), s.data_ptr(), ga.data_ptr(), gd.data_ptr(), P, th, K, partial.data_ptr(),
                                      _stream_ptr(a.device)), "vtgs_silhouette_sweep")
    return partial.sum(0, dtype=torch.float64)


def best_silhouette_threshold(im, silhouette, gt_im, gt_depth, candidates=(0.990, 0.993, 0.995, 0.997, 0.999)) -> float:
    """Tracking iteration 0 of the Replica branch: the candidate with the smallest masked colour MSE, first one on ties
    (src/vtgaussian_slam.py:472-510) -- one kernel and one host read instead of five masked gathers and five .item()."""
    sums = silhouette_sweep(im, silhouette, gt_im, gt_depth, candidates).cpu()
    best, best_mse = candidates[0], float("inf")
    for k, c in enumerate(candidates):
        cnt = float(sums[k, 1])
        mse = float(sums[k, 0]) / (3.0 * cnt) if cnt > 0 else float("inf")      # an empty mask gives NaN in the reference
        if mse < best_mse:
            best, best_mse = c, mse
    return best
