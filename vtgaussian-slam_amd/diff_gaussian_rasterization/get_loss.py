"""`get_loss` of the reference with its own signature (src/vtgaussian_slam.py:407-689), on the fused operators.

The reference's tracking and mapping loops call ONE function per iteration:

    loss, variables, losses = get_loss(params, curr_data, variables, iter_time_idx, loss_weights, use_sil_for_loss,
                                       sil_thres, use_l1, ignore_outlier_depth_loss, tracking=True, ...)

A maintainer who replaces that function by this one (`from diff_gaussian_rasterization.get_loss import get_loss`) gets the
fused caller chain without touching the loops: pose transform + render-variable builders + both renders as the dual
composite (`fused.render_frame`), the masks of every dataset branch (`losses.visibility_mask`, `far_depth_mask`,
`outlier_depth_mask`), the threshold sweep of Replica's tracking iteration 0 and the whole loss as one autograd node.
Same arguments, same return value (`loss, variables, weighted_losses[, presence_sil_mask_mse_ls, sil_thres_ls]`), same
bookkeeping of `variables['seen']` / `['max_2D_radius']` and of the two threshold lists.

Every branch of the reference is covered: `use_l1=False` (no depth term, :591-596), tracking with neither
`use_sil_for_loss` nor `ignore_outlier_depth_loss` (the colour sum over all pixels, :601-602), all dataset masks.
Differences:
  * `variables['means2D']` (:460-462: the screen-space positions of the COLOUR render, whose `.grad` feeds
    `accumulate_mean2d_gradient`, utils/slam_external.py:100-103, when `use_gaussian_splatting_densification` is on --
    every shipped config has it off).  The dual backward sums the geometry gradients of both renders, so by default
    `variables['means2D']` is a placeholder whose `.grad` raises with this explanation instead of handing a stale tensor or
    a KeyError to the densification code.  `get_loss.SCREEN_SPACE_GRADIENT = True` (module attribute, or the environment
    variable VTGS_SCREEN_SPACE_GRADIENT=1) selects the two-render route -- `transform_to_frame` + the plain operator for the
    colour render with `means2D.retain_grad()`, the depth / silhouette render over the same bins -- and sets the real tensor;
  * `visualize_tracking_loss` is ignored (plots);
  * the entries of `weighted_losses` other than 'loss' are detached (the loops only log them).
Unsupported argument combinations are rejected BEFORE `params` / `variables` are touched.  HIP only: no CPU path.
"""
from __future__ import annotations

import os

import torch

from . import losses as _l
from .fused import render_frame, render_frame_unfused

__all__ = ["get_loss", "SCREEN_SPACE_GRADIENT"]

SCREEN_SPACE_GRADIENT = os.environ.get("VTGS_SCREEN_SPACE_GRADIENT", "0") == "1"


class _NoScreenSpaceGradient:
    """Placeholder for variables['means2D'] on the fused route: reading `.grad` explains what to switch on."""

    @property
    def grad(self):
        raise RuntimeError(
            "variables['means2D'].grad: the fused get_loss composites both renders in one pass and has no colour-render-only "
            "screen-space gradient (src/vtgaussian_slam.py:460-462). Densification from that gradient "
            "(use_gaussian_splatting_densification) needs `diff_gaussian_rasterization.get_loss.SCREEN_SPACE_GRADIENT = True` "
            "(or VTGS_SCREEN_SPACE_GRADIENT=1), which renders the two images separately.")


def _render_separately(params, iter_time_idx, cam, w2c, gaussians_grad, camera_grad):
    """The reference's own chain (:431-468) on the plain operator: the colour render keeps `means2D` in the graph."""
    return render_frame_unfused(params, iter_time_idx, cam, w2c, gaussians_grad, camera_grad)


def _cuda_f32(v):
    if isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.float32 and v.is_contiguous():
        return v                                   # (what .cuda().float().contiguous() returns for such a tensor: itself)
    t = v if isinstance(v, torch.Tensor) else torch.tensor(v)
    return t.cuda().float().contiguous()


def get_loss(params, curr_data, variables, iter_time_idx, loss_weights, use_sil_for_loss,
             sil_thres, use_l1, ignore_outlier_depth_loss, tracking=False,
             mapping=False, do_ba=False, plot_dir=None, visualize_tracking_loss=False,
             tracking_iteration=None, additional_mask=None, dataset_name=None,
             presence_sil_mask_mse_ls=None, sil_thres_ls=None, far_depth_filter_thres=None, vis_mask_thres=0.05,
             curr_w2c=None, overlap_w2c=None, overlap_gtdepth=None, overlap_last_w2c=None, overlap_last_gtdepth=None,
             overlap_mid_w2c=None, overlap_mid_gtdepth=None):
    # argument combinations this function cannot serve are rejected before anything is converted in place
    if tracking and use_sil_for_loss and dataset_name not in ("replica", "tum", "scannet", "scannetpp"):
        raise ValueError(f"get_loss: no presence mask is defined for dataset_name={dataset_name!r} "
                         "(src/vtgaussian_slam.py:470-514 knows replica, tum, scannet, scannetpp)")
    if tracking and overlap_w2c is not None and dataset_name not in ("replica", "tum", "scannet", "scannetpp"):
        raise ValueError(f"get_loss: visibility mask undefined for dataset_name={dataset_name!r}")
    # :416-426 -- everything on the device, float32, contiguous (a no-op for tensors that already are).  A tensor that has to
    # be converted is REPLACED in the caller's dict, as in the reference (whose Parameters already live on the device).
    for k, v in params.items():
        params[k] = _cuda_f32(v)
    for k, v in variables.items():
        if k not in ("means2D", "seen"):           # both are REPLACED below before anybody reads them: converting the bool
            variables[k] = _cuda_f32(v)            # `seen` of the previous call to float32 was an 8 us launch per iteration
    # :428-449 -- who gets a gradient
    if tracking:
        gaussians_grad, camera_grad = False, True
    elif mapping and do_ba:
        gaussians_grad, camera_grad = True, True
    else:
        gaussians_grad, camera_grad = True, False
    # :451-468 -- both renders
    if SCREEN_SPACE_GRADIENT:
        im, depth_sil, radius, means2D = _render_separately(params, iter_time_idx, curr_data["cam"], curr_data["w2c"],
                                                            gaussians_grad, camera_grad)
        variables["means2D"] = means2D                 # gradient only accumulated from the colour render (:462)
    else:
        # (get_loss_contract: of the [z, 1, z^2] render this function differentiates z alone, compares the silhouette with
        #  thresholds and asks of z^2 only whether the detached uncertainty is NaN, :466-521; the loss nodes write zeros into
        #  the other two gradient planes)
        im, depth_sil, radius = render_frame(params, iter_time_idx, curr_data["cam"], curr_data["w2c"], gaussians_grad, camera_grad,
                                             get_loss_contract=True)
        variables["means2D"] = _NoScreenSpaceGradient()
    gt_im, gt_depth = curr_data["im"], curr_data["depth"]
    depth = depth_sil[0:1].detach()

    # :470-514 -- the silhouette threshold of the presence mask
    thr = None
    if dataset_name == "replica":
        if tracking and use_sil_for_loss:
            if tracking_iteration == 0:
                thr, mse = _l.best_silhouette_threshold(im.detach(), depth_sil.detach()[1], gt_im, gt_depth, return_mse=True)
                presence_sil_mask_mse_ls.append(mse)
                sil_thres_ls.append(thr)
            else:
                thr = sil_thres_ls[-1]
    elif dataset_name in ("tum", "scannet", "scannetpp"):
        thr = sil_thres

    # :523-588 -- detached masks beyond gt_depth > 0 & finite (those two live in the loss node)
    masks = []
    if ignore_outlier_depth_loss:
        masks.append(_l.outlier_depth_mask(gt_depth, depth))
    if tracking and overlap_w2c is not None and dataset_name != "replica":
        if dataset_name == "tum":
            overlaps = [(overlap_w2c, overlap_gtdepth)]
        elif dataset_name in ("scannet", "scannetpp"):
            overlaps = [(overlap_w2c, overlap_gtdepth), (overlap_mid_w2c, overlap_mid_gtdepth),
                        (overlap_last_w2c, overlap_last_gtdepth)]
        else:
            overlaps = []
        masks.append(_l.visibility_mask(gt_depth, curr_data["intrinsics"], curr_w2c, overlaps, vis_mask_thres)[None])
    if tracking and far_depth_filter_thres is not None and dataset_name not in ("replica", "scannetpp"):
        masks.append(_l.far_depth_mask(gt_depth, far_depth_filter_thres))
    extra = None
    for m in masks:
        extra = m if extra is None else (extra & m)

    # :681-689 -- bookkeeping, decided here because it rides in the loss's last launch when it can:
    # max_2D_radius[seen] = max(radius[seen], max_2D_radius[seen]) without the boolean-mask gathers (each of them waits for
    # the device to count its elements -- 0.3 - 1.2 ms per iteration at 1 M Gaussians): a culled Gaussian has radius 0 and
    # the running maximum is never negative, so the element-wise maximum over ALL Gaussians is the same in-place update
    mx = variables["max_2D_radius"]
    fast_bk = (mx.dtype == torch.float32 and mx.is_contiguous() and radius.dtype == torch.int32 and radius.is_contiguous()
               and mx.data_ptr() % 16 == 0 and radius.data_ptr() % 16 == 0 and mx.numel() == radius.numel() and radius.numel() > 0)
    seen = torch.empty(radius.shape, dtype=torch.bool, device=radius.device) if fast_bk else None
    bk = (radius, mx, seen) if fast_bk else None

    # :590-611 -- the loss
    w_im = float(loss_weights["im"])
    w_depth = float(loss_weights["depth"]) if use_l1 else 0.0   # use_l1 = False: no depth term at all (:591-596)
    if tracking:
        loss, terms = _l.tracking_loss(im, depth_sil, gt_im, gt_depth, thr if use_sil_for_loss else float("-inf"),
                                       w_im=w_im, w_depth=w_depth, extra_mask=extra, return_terms=True,
                                       colour_over_all_pixels=not (use_sil_for_loss or ignore_outlier_depth_loss), bookkeeping=bk)
    else:
        loss, terms = _l.mapping_loss(im, depth_sil, gt_im, gt_depth, w_im=w_im, w_depth=w_depth, extra_mask=extra,
                                      additional_mask=additional_mask, return_terms=True, bookkeeping=bk)
    weighted_losses = {"im": terms[5]}                          # formed by the loss kernel: no element-wise launches here
    if use_l1:
        weighted_losses["depth"] = terms[6]

    # :681-689 -- bookkeeping: done inside the loss's last launch (above) or, for other tensor layouts, with torch
    if not fast_bk:
        seen = radius > 0
        torch.maximum(mx, radius.to(mx.dtype), out=mx)
    variables["seen"] = seen
    weighted_losses["loss"] = loss
    if presence_sil_mask_mse_ls is not None:
        return loss, variables, weighted_losses, presence_sil_mask_mse_ls, sil_thres_ls
    return loss, variables, weighted_losses
