"""Point-to-plane consistency of two depth frames on the device: the GPU form of
`compute_point2plane_dist` (src/vtgaussian_slam.py:1070-1155), which the reference evaluates on the host -- kornia
normals -> numpy -> Open3D KD-tree -- at every tracking iteration of a base-frame boundary.

    d = compute_point2plane_dist(depth_latest, depth_curr, intrinsics, latest_w2c, curr_w2c, frustum=True, method='sum')

Same arguments as the reference minus the dataset look-up: the caller passes the two depth maps ([1,H,W] or [H,W], the
`depth.permute(2, 0, 1)` of dataset[frame_id]) instead of (dataset, frame ids).  Returns a 0-dim device tensor (no host
wait).  No CPU path: CPU tensors raise.
"""
from __future__ import annotations

import ctypes

import torch

from . import _I32, _P, _SZ, _check, _lib, _stream_ptr

_lib.vtgs_point2plane_scratch_bytes.restype, _lib.vtgs_point2plane_scratch_bytes.argtypes = _SZ, [_I32, _I32]
_lib.vtgs_point2plane.restype = ctypes.c_int
_lib.vtgs_point2plane.argtypes = [_I32, _I32, _P, _P, _P, _P, _P, _P, _P, ctypes.c_float, _I32, _P, _SZ, _P, _P, _P]


def point2plane_pairs(depth_latest, depth_curr, intrinsics, latest_w2c, curr_w2c, frustum=True, latest_varmask=None,
                      curr_varmask=None, threshold: float = 0.02):
    """(dist [H,W] float32, matched [H,W] bool): n_latest . (p_curr - nearest p_latest) per pixel of the current frame."""
    if not depth_latest.is_cuda:
        raise RuntimeError("point2plane needs tensors on a HIP device (torch 'cuda'); no CPU path exists")
    dev = depth_latest.device
    f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
    d0, d1 = f32(depth_latest), f32(depth_curr)
    H, W = d0.shape[-2], d0.shape[-1]
    if d1.shape[-2:] != d0.shape[-2:] or d0.numel() != H * W or d1.numel() != H * W:
        raise ValueError("the two depth maps must be [H,W] / [1,H,W] of equal size")
    k = f32(intrinsics)[:3, :3].contiguous()
    w0, w1 = f32(latest_w2c).reshape(16), f32(curr_w2c).reshape(16)
    u8 = lambda m: None if m is None else m.detach().to(device=dev).reshape(-1).to(torch.uint8).contiguous()
    m0, m1 = u8(latest_varmask), u8(curr_varmask)
    sbytes = int(_lib.vtgs_point2plane_scratch_bytes(W, H))
    scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
    dist = torch.empty((H, W), dtype=torch.float32, device=dev)
    matched = torch.empty((H, W), dtype=torch.uint8, device=dev)
    _check(_lib.vtgs_point2plane(W, H, d0.data_ptr(), d1.data_ptr(), None if m0 is None else m0.data_ptr(),
                                 None if m1 is None else m1.data_ptr(), k.data_ptr(), w0.data_ptr(), w1.data_ptr(),
                                 float(threshold), 1 if frustum else 0, scratch.data_ptr(), sbytes, dist.data_ptr(),
                                 matched.data_ptr(), _stream_ptr(dev)), "vtgs_point2plane")
    return dist, matched.bool()


def compute_point2plane_dist(depth_latest, depth_curr, intrinsics, latest_w2c, curr_w2c, frustum=True, latest_varmask=None,
                             curr_varmask=None, method: str = "sum", threshold: float = 0.02):
    dist, matched = point2plane_pairs(depth_latest, depth_curr, intrinsics, latest_w2c, curr_w2c, frustum, latest_varmask,
                                      curr_varmask, threshold)
    if method == "sum":                       # unmatched pixels hold 0
        return (dist ** 2).sum()
    vals = dist.abs()[matched]
    if method == "max":
        return vals.max()
    if method == "max100":
        return vals.topk(min(100, vals.numel()))[0].mean()
    raise ValueError(f"unknown method {method!r}")
