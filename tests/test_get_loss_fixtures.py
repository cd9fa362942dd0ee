"""a10: the loss glue of get_loss for all four dataset branches, against fixtures captured from the reference's OWN
`get_loss` / `get_vis_mask` (tests/golden/make_get_loss_fixtures.py ran src/vtgaussian_slam.py:376-404, 407-689 in the
build container with the CPU oracle behind the rasterizer operator).  CPU part: the detached masks (device-agnostic torch
ops).  GPU part: the fused loss node (csrc/vtgs_loss.hip) -- value, per-term values and both gradient images."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    z = np.load(os.path.join(HERE, "golden", "get_loss.npz"))
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}


def _overlaps(fx, idx):
    return [(fx[f"overlap{i}_w2c"], fx[f"overlap{i}_gtdepth"]) for i in idx]


def test_visibility_masks_equal_the_reference(fx):
    """visibility_mask == get_vis_mask of the reference for each overlapping frame (a pixel whose sampled depth sits within
    float rounding of the threshold may differ: none does on this fixture) and ORs them like the ScanNet branch."""
    from diff_gaussian_rasterization.losses import visibility_mask
    masks = []
    for i in range(3):
        m = visibility_mask(fx["gt_depth"], fx["intrinsics"], fx["curr_w2c"], _overlaps(fx, [i]), 0.05)
        assert m.dtype == torch.bool and m.shape == fx[f"vis_mask{i}"].shape
        assert (m != fx[f"vis_mask{i}"]).sum().item() == 0
        assert 0.05 < m.double().mean().item() < 0.999            # a mask that actually cuts something
        masks.append(m)
    both = visibility_mask(fx["gt_depth"], fx["intrinsics"], fx["curr_w2c"], _overlaps(fx, [0, 1, 2]), 0.05)
    assert torch.equal(both, masks[0] | masks[1] | masks[2])


def _masks_for(fx, name, depth):
    from diff_gaussian_rasterization.losses import far_depth_mask, outlier_depth_mask, visibility_mask
    gt_depth = fx["gt_depth"].to(depth.device)
    if name == "tum_tracking_vis_far":
        vis = visibility_mask(gt_depth, fx["intrinsics"], fx["curr_w2c"], _overlaps(fx, [0]), 0.05)
        return vis[None] & far_depth_mask(gt_depth, 5.0)
    if name == "scannet_tracking_vis3_outlier":
        vis = visibility_mask(gt_depth, fx["intrinsics"], fx["curr_w2c"], _overlaps(fx, [0, 1, 2]), 0.05)
        return vis[None] & far_depth_mask(gt_depth, 6.0) & outlier_depth_mask(gt_depth, depth)
    return None


CASES = {   # name: (mode, sil_thres or None (= the sweep), uses additional_mask)
    "replica_tracking_iter0": ("tracking", None, False),
    "replica_mapping": ("mapping", 0.5, False),
    "tum_tracking_vis_far": ("tracking", 0.9, False),
    "scannet_tracking_vis3_outlier": ("tracking", 0.9, False),
    "scannetpp_mapping_additional_mask": ("mapping", 0.5, True),
    "tum_tracking_no_l1": ("tracking_no_l1", 0.9, False),                 # use_l1 = False: no depth term (:591-596)
    "tum_tracking_unmasked_colour": ("tracking_unmasked", None, False),   # colour summed over ALL pixels (:601-602)
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_fused_loss_equals_the_reference_get_loss(gpu_device, fx, name):
    from diff_gaussian_rasterization import losses
    dev = gpu_device
    mode, sil_thres, use_add = CASES[name]
    im = fx[name + "_im"].to(dev).requires_grad_(True)
    ds = fx[name + "_depth_sil"].to(dev).requires_grad_(True)
    gt_im, gt_depth = fx["gt_im"].to(dev), fx["gt_depth"].to(dev)
    if sil_thres is None and mode == "tracking":              # Replica, tracking iteration 0: five-candidate sweep
        sil_thres = losses.best_silhouette_threshold(im.detach(), ds.detach()[1], gt_im, gt_depth)
        assert abs(sil_thres - fx[name + "_sil_thres_chosen"]) < 1e-9
    extra = _masks_for(fx, name, ds.detach()[0:1])
    if mode == "tracking":
        loss = losses.tracking_loss(im, ds, gt_im, gt_depth, sil_thres, w_im=0.5, w_depth=1.0, extra_mask=extra)
    elif mode == "tracking_no_l1":                            # what get_loss.get_loss passes for use_l1 = False
        loss = losses.tracking_loss(im, ds, gt_im, gt_depth, sil_thres, w_im=0.5, w_depth=0.0, extra_mask=extra)
    elif mode == "tracking_unmasked":                         # ... and for use_sil_for_loss = ignore_outlier_depth_loss = False
        loss = losses.tracking_loss(im, ds, gt_im, gt_depth, float("-inf"), w_im=0.5, w_depth=1.0, extra_mask=extra,
                                    colour_over_all_pixels=True)
    else:
        loss = losses.mapping_loss(im, ds, gt_im, gt_depth, w_im=0.5, w_depth=1.0,
                                   additional_mask=fx["additional_mask"].to(dev) if use_add else None)
    loss.backward()
    ref = fx[name + "_loss"]
    assert abs(loss.item() - ref) <= 2e-5 * abs(ref), (loss.item(), ref)
    for got, want, what in ((im.grad, fx[name + "_g_im"], "d loss / d im"), (ds.grad, fx[name + "_g_depth_sil"], "d loss / d depth_sil")):
        want = want.to(dev)
        scale = want.abs().max().item()
        if mode == "tracking_no_l1" and what.endswith("depth_sil"):
            assert scale == 0 and (got is None or got.abs().max().item() == 0)   # no depth term: no gradient to the second render
            continue
        assert scale > 0
        err = (got - want).abs().max().item() / scale
        assert err <= 2e-4, f"{name}: {what} differs by {err:.2e} of its maximum"
