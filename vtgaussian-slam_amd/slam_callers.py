"""Host-side mirror of the reference's caller chain around the rasterizer.

These are the functions that sit on either side of the operator in the reference's tracking / mapping loops,
restated device-agnostically (the reference hard-codes 'cuda': utils/slam_external.py:28,
utils/slam_helpers.py:122,229,348,371) with the same names, argument meaning and results:

    build_rotation                          utils/slam_external.py:25-42
    quat_mult                               utils/slam_helpers.py:24-31
    transform_to_frame                      utils/slam_helpers.py:323-385
    get_depth_and_silhouette                utils/slam_helpers.py:217-234
    transformed_params2rendervar            utils/slam_helpers.py:127-160
    transformed_params2depthplussilhouette  utils/slam_helpers.py:255-287
    l1_loss_v1, calc_ssim                   utils/slam_helpers.py:5-6, utils/slam_external.py:66-97
    tracking_loss / mapping_loss            src/vtgaussian_slam.py:407-689 (Replica branch of get_loss)

The unmodified reference driver does not need this module (its own helpers run as they are on PyTorch-ROCm);
it exists for `bench_slam.py` and the tests, which cannot import reference code on the GPU box.  Parity is pinned
by tests/golden/helpers_*.npz, captured from the reference's own modules (tests/golden/make_helper_fixtures.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


def build_rotation(q: torch.Tensor) -> torch.Tensor:
    """[B,4] (w,x,y,z), normalised inside -> [B,3,3]."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(dim=1)
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def quat_mult(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    """Hamilton product of (w,x,y,z) quaternions, [B,4] x [B,4] (q1 may have B = 1)."""
    w1, x1, y1, z1 = q1.unbind(dim=-1)
    w2, x2, y2, z2 = q2.unbind(dim=-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)


def transform_to_frame(params: Dict[str, torch.Tensor], time_idx: int, gaussians_grad: bool, camera_grad: bool,
                       latest_w2c: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """World -> frame-`time_idx` camera: the only place the pose enters the autograd graph.
    Isotropic maps (log_scales [N,1], every reference config) pass the rotations through untouched;
    anisotropic maps compose them with the camera rotation."""
    rot_param = params["cam_unnorm_rots"][..., time_idx]
    trans_param = params["cam_trans"][..., time_idx]
    if not camera_grad:
        rot_param, trans_param = rot_param.detach(), trans_param.detach()
    cam_rot = F.normalize(rot_param)
    dev, dt = cam_rot.device, torch.float32
    rel_w2c = torch.eye(4, device=dev, dtype=dt)
    rel_w2c = torch.cat([torch.cat([build_rotation(cam_rot)[0], trans_param.reshape(3, 1)], dim=1),
                         rel_w2c[3:4]], dim=0)
    if latest_w2c is not None:
        rel_w2c = latest_w2c @ rel_w2c
    pts, unnorm_rots = params["means3D"], params["unnorm_rotations"]
    if not gaussians_grad:
        pts, unnorm_rots = pts.detach(), unnorm_rots.detach()
    out = {"means3D": pts @ rel_w2c[:3, :3].t() + rel_w2c[:3, 3]}
    if params["log_scales"].shape[1] == 1:
        out["unnorm_rotations"] = unnorm_rots
    else:
        out["unnorm_rotations"] = quat_mult(cam_rot, F.normalize(unnorm_rots))
    return out


def get_depth_and_silhouette(pts_3D: torch.Tensor, w2c: torch.Tensor) -> torch.Tensor:
    """Per-splat colours of the second render: [z, 1, z^2] with z in the camera `w2c`."""
    r = w2c[2]
    z = pts_3D[:, 0] * r[0] + pts_3D[:, 1] * r[1] + pts_3D[:, 2] * r[2] + r[3]      # (a [N,3] x [3] gemv is 10x slower)
    return torch.stack([z, torch.ones_like(z), z * z], dim=1)


def _scales(params):
    ls = params["log_scales"]
    return torch.exp(ls.expand(-1, 3) if ls.shape[1] == 1 else ls)


def _common_rendervar(params, transformed):
    return {
        "means3D": transformed["means3D"],
        "rotations": F.normalize(transformed["unnorm_rotations"]),
        "opacities": torch.sigmoid(params["logit_opacities"]),
        "scales": _scales(params),
        # zeros that receive the screen-space gradient (the reference calls retain_grad() on it)
        "means2D": torch.zeros_like(params["means3D"], requires_grad=True) + 0,
    }


def transformed_params2rendervar(params, transformed) -> Dict[str, torch.Tensor]:
    rv = _common_rendervar(params, transformed)
    rv["colors_precomp"] = params["rgb_colors"]
    return rv


def transformed_params2depthplussilhouette(params, w2c, transformed) -> Dict[str, torch.Tensor]:
    rv = _common_rendervar(params, transformed)
    rv["colors_precomp"] = get_depth_and_silhouette(transformed["means3D"], w2c)
    return rv


def l1_loss_v1(x, y):
    return (x - y).abs().mean()


def _gauss_window(size: int, sigma: float, channels: int, like: torch.Tensor) -> torch.Tensor:
    g = torch.tensor([math.exp(-(i - size // 2) ** 2 / (2 * sigma ** 2)) for i in range(size)])
    g = (g / g.sum()).to(like)
    return (g[:, None] @ g[None, :]).expand(channels, 1, size, size).contiguous()


def calc_ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """Mean SSIM with an 11x11 Gaussian window (sigma 1.5), zero padding, per-channel (grouped) filtering."""
    ch = img1.size(-3)
    a, b = img1.reshape(1, ch, *img1.shape[-2:]), img2.reshape(1, ch, *img2.shape[-2:])
    w = _gauss_window(window_size, 1.5, ch, a)
    blur = lambda t: F.conv2d(t, w, padding=window_size // 2, groups=ch)
    mu1, mu2 = blur(a), blur(b)
    s11, s22, s12 = blur(a * a) - mu1 * mu1, blur(b * b) - mu2 * mu2, blur(a * b) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))).mean()


def split_depth_silhouette(depth_sil: torch.Tensor):
    """(depth [1,H,W], silhouette [H,W], detached uncertainty [1,H,W]) from the [z,1,z^2] render."""
    depth = depth_sil[0:1]
    return depth, depth_sil[1], (depth_sil[2:3] - depth ** 2).detach()


def tracking_loss(im, depth_sil, gt_im, gt_depth, sil_thres: float, w_im: float = 0.5, w_depth: float = 0.025):
    """Replica tracking branch of get_loss: masked L1 SUMS over pixels with valid depth, finite renders and
    silhouette > sil_thres (src/vtgaussian_slam.py:513-605, 678-679)."""
    depth, sil, unc = split_depth_silhouette(depth_sil)
    mask = (gt_depth > 0) & ~torch.isnan(depth) & ~torch.isnan(unc) & (sil > sil_thres)[None]
    mask = mask.detach()
    l_depth = (gt_depth - depth).abs()[mask].sum()              # boolean indexing, like the reference (nonzero + gather)
    l_im = (gt_im - im).abs()[mask.expand(3, -1, -1)].sum()
    return w_im * l_im + w_depth * l_depth


def mapping_loss(im, depth_sil, gt_im, gt_depth, w_im: float = 1.0, w_depth: float = 1.0):
    """Mapping branch: masked L1 MEAN on depth, 0.8 L1 + 0.2 (1 - SSIM) on colour (src/vtgaussian_slam.py:592-608)."""
    depth, _, unc = split_depth_silhouette(depth_sil)
    mask = ((gt_depth > 0) & ~torch.isnan(depth) & ~torch.isnan(unc)).detach()
    l_depth = (gt_depth - depth).abs()[mask].mean()
    l_im = 0.8 * l1_loss_v1(im, gt_im) + 0.2 * (1.0 - calc_ssim(im, gt_im))
    return w_im * l_im + w_depth * l_depth


def best_silhouette_threshold(im, silhouette, gt_im, gt_depth, candidates=(0.990, 0.993, 0.995, 0.997, 0.999)) -> float:
    """Replica, tracking iteration 0: the candidate whose masked colour MSE is smallest (src/vtgaussian_slam.py:472-510)."""
    best, best_mse = candidates[0], float("inf")
    diff2 = (gt_im - im).detach() ** 2
    for c in candidates:
        m = ((silhouette > c) & (gt_depth[0] > 0))[None].expand(3, -1, -1)
        mse = diff2[m].mean().item() if bool(m.any()) else float("inf")
        if mse < best_mse:
            best, best_mse = c, mse
    return best
