"""Loss kernels: `fused_ssim(img1, img2)` == utils/slam_external.py:66-97 `calc_ssim(img1, img2)` (mean
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
