"""`diff_gaussian_rasterization.get_loss.get_loss`: the reference's get_loss (src/vtgaussian_slam.py:407-689) with its own
signature on the fused operators.  CPU part: the signature (names, order, defaults) is the reference's.  GPU part: values,
gradients and bookkeeping against the UNFUSED composition of pieces that are each pinned to the reference elsewhere -- the
restated helper chain (tests/test_slam_callers_golden.py), the drop-in operator (tests/test_gpu_parity.py) and the loss
node + masks (tests/test_get_loss_fixtures.py)."""
import ast
import inspect
import os

import pytest
import torch

REF = "/root/reference/src/vtgaussian_slam.py"

# (name, default) in the reference's order; `...` = no default
SIGNATURE = [("params", ...), ("curr_data", ...), ("variables", ...), ("iter_time_idx", ...), ("loss_weights", ...),
             ("use_sil_for_loss", ...), ("sil_thres", ...), ("use_l1", ...), ("ignore_outlier_depth_loss", ...),
             ("tracking", False), ("mapping", False), ("do_ba", False), ("plot_dir", None),
             ("visualize_tracking_loss", False), ("tracking_iteration", None), ("additional_mask", None),
             ("dataset_name", None), ("presence_sil_mask_mse_ls", None), ("sil_thres_ls", None),
             ("far_depth_filter_thres", None), ("vis_mask_thres", 0.05), ("curr_w2c", None), ("overlap_w2c", None),
             ("overlap_gtdepth", None), ("overlap_last_w2c", None), ("overlap_last_gtdepth", None),
             ("overlap_mid_w2c", None), ("overlap_mid_gtdepth", None)]


def test_signature_is_the_reference_signature():
    src = open(os.path.join(os.path.dirname(__file__), "..", "vtgaussian-slam_amd", "diff_gaussian_rasterization", "get_loss.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "get_loss")
    names = [a.arg for a in fn.args.args]
    defaults = [...] * (len(names) - len(fn.args.defaults)) + [ast.literal_eval(d) for d in fn.args.defaults]
    assert list(zip(names, defaults)) == SIGNATURE
    if os.path.exists(REF):                                   # build container only: the table above IS the reference's
        ref = next(n for n in ast.parse(open(REF).read()).body if isinstance(n, ast.FunctionDef) and n.name == "get_loss")
        rn = [a.arg for a in ref.args.args]
        rd = [...] * (len(rn) - len(ref.args.defaults)) + [ast.literal_eval(d) for d in ref.args.defaults]
        assert list(zip(rn, rd)) == SIGNATURE


def _scene(dev, n, W, H, seed):
    import sys
    from oracle import gs_oracle as go
    from parity_util import to_settings
    scene, cam = go.view_tied_scene(n, W, H, seed=seed)
    T = 4
    g = torch.Generator().manual_seed(seed)
    params = {
        "means3D": scene["means3D"], "rgb_colors": scene["colors_precomp"], "unnorm_rotations": scene["rotations"] * 1.7,
        "logit_opacities": torch.logit(scene["opacities"].clamp(1e-4, 1 - 1e-4)), "log_scales": torch.log(scene["scales"][:, :1]),
        "cam_unnorm_rots": torch.tensor([1.0, 0.01, -0.02, 0.005]).reshape(1, 4, 1).repeat(1, 1, T),
        "cam_trans": (0.01 * torch.randn(1, 3, T, generator=g)),
    }
    params = {k: torch.nn.Parameter(v.to(dev).float().contiguous()) for k, v in params.items()}
    st = to_settings(cam, dev)
    k = torch.tensor([[W / 2.0, 0, W / 2.0 - 0.5], [0, W / 2.0, H / 2.0 - 0.5], [0, 0, 1]])
    return params, st, k, go


CASES = [   # dataset, tracking, iteration, ignore_outlier, additional mask, far-depth thr, overlaps
    ("replica", True, 0, False, False, None, 0),
    ("replica", True, 3, False, False, None, 0),
    ("replica", False, None, False, False, None, 0),
    ("tum", True, 1, False, False, 5.0, 1),
    ("scannet", True, 1, True, False, 6.0, 3),
    ("scannetpp", False, None, True, True, None, 0),
]


@pytest.mark.gpu
@pytest.mark.parametrize("dataset,tracking,it,outlier,use_add,far,n_over", CASES)
def test_get_loss_mirror_equals_the_unfused_composition(gpu_device, dataset, tracking, it, outlier, use_add, far, n_over):
    import diff_gaussian_rasterization as dgr
    import slam_callers as sc
    from diff_gaussian_rasterization import losses
    from diff_gaussian_rasterization.get_loss import get_loss
    dev = gpu_device
    W, H, N = 160, 120, 24000
    params, st, K, go = _scene(dev, N, W, H, seed=17)
    g = torch.Generator().manual_seed(2)
    w2c0 = torch.eye(4, device=dev)
    with torch.no_grad():                                    # observations: the scene itself from a nearby pose, plus noise
        gt_p = {k: v.detach().clone() for k, v in params.items()}
        gt_p["cam_trans"] = gt_p["cam_trans"] + 0.004
        tg = sc.transform_to_frame(gt_p, 1, False, False)
        gim, _, _ = dgr.GaussianRasterizer(raster_settings=st)(**sc.transformed_params2rendervar(gt_p, tg))
        gds, _, _ = dgr.GaussianRasterizer(raster_settings=st)(**sc.transformed_params2depthplussilhouette(gt_p, w2c0, tg))
        gt_im = (gim + 0.03 * torch.randn(3, H, W, generator=g).to(dev)).clamp(0, 1)
        gt_depth = torch.where(gds[1:2] > 0.5, gds[0:1] / gds[1:2].clamp(min=1e-6), torch.zeros_like(gds[0:1]))
        gt_depth[:, 30:40, 50:70] = 0.0
        gt_depth[:, 80:84, 20:40] *= 60.0                    # outliers for the 50 x median mask
    curr = {"cam": st, "im": gt_im, "depth": gt_depth, "w2c": w2c0, "intrinsics": K.to(dev), "id": 1}
    weights = {"im": 0.5, "depth": 1.0}
    add_mask = (torch.rand(3, H, W, generator=g) > 0.7).to(dev) if use_add else None
    curr_w2c = torch.eye(4, device=dev)
    overlaps = []
    for i in range(3):
        o = torch.eye(4, device=dev)
        o[:3, 3] = torch.tensor([0.03 * (i + 1), -0.02, 0.01 * i], device=dev)
        overlaps.append((o, (gt_depth * (1.0 + 0.02 * i)).clone()))
    okw = {}
    if n_over >= 1:
        okw.update(curr_w2c=curr_w2c, overlap_w2c=overlaps[0][0], overlap_gtdepth=overlaps[0][1])
    if n_over == 3:
        okw.update(overlap_mid_w2c=overlaps[1][0], overlap_mid_gtdepth=overlaps[1][1],
                   overlap_last_w2c=overlaps[2][0], overlap_last_gtdepth=overlaps[2][1])
    mse_ls, thr_ls = ([], [0.995]) if dataset == "replica" and tracking else (None, None)
    if it == 0:
        thr_ls = []
    variables = {"max_2D_radius": torch.zeros(N, device=dev), "means2D_gradient_accum": torch.zeros(N, device=dev),
                 "denom": torch.zeros(N, device=dev)}

    # ---- the mirror
    for v in params.values():
        v.grad = None
    out = get_loss(params, curr, variables, 1, weights, True, 0.9, True, outlier, tracking=tracking, mapping=not tracking,
                   tracking_iteration=it, additional_mask=add_mask, dataset_name=dataset, presence_sil_mask_mse_ls=mse_ls,
                   sil_thres_ls=thr_ls, far_depth_filter_thres=far, **okw)
    loss, variables_out, wl = out[0], out[1], out[2]
    loss.backward()
    got = {k: (None if v.grad is None else v.grad.clone()) for k, v in params.items()}
    assert (len(out) == 5) == (mse_ls is not None)
    if it == 0:
        assert len(thr_ls) == 1 and len(mse_ls) == 1 and thr_ls[0] in (0.990, 0.993, 0.995, 0.997, 0.999)
    thr = thr_ls[-1] if dataset == "replica" and tracking else 0.9

    # ---- the unfused composition
    for v in params.values():
        v.grad = None
    tg = sc.transform_to_frame(params, 1, gaussians_grad=not tracking, camera_grad=tracking)
    im, radius, _ = dgr.GaussianRasterizer(raster_settings=st)(**sc.transformed_params2rendervar(params, tg))
    ds, _, _ = dgr.GaussianRasterizer(raster_settings=st)(**sc.transformed_params2depthplussilhouette(params, w2c0, tg))
    if it == 0:
        assert losses.best_silhouette_threshold(im.detach(), ds.detach()[1], gt_im, gt_depth) == thr
    extra = None
    def AND(a, b): return b if a is None else (a & b)
    if outlier:
        extra = AND(extra, losses.outlier_depth_mask(gt_depth, ds.detach()[0:1]))
    if tracking and n_over:
        extra = AND(extra, losses.visibility_mask(gt_depth, K.to(dev), curr_w2c, overlaps[:n_over], 0.05)[None])
    if tracking and far is not None and dataset not in ("replica", "scannetpp"):
        extra = AND(extra, losses.far_depth_mask(gt_depth, far))
    if tracking:
        ref = losses.tracking_loss(im, ds, gt_im, gt_depth, thr, w_im=0.5, w_depth=1.0, extra_mask=extra)
    else:
        ref = losses.mapping_loss(im, ds, gt_im, gt_depth, w_im=0.5, w_depth=1.0, extra_mask=extra, additional_mask=add_mask)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 2e-5 * abs(ref.item()), (loss.item(), ref.item())
    assert abs((wl["im"] + wl["depth"]).item() - loss.item()) <= 2e-5 * abs(loss.item())
    assert wl["loss"] is loss
    for k, v in params.items():
        if v.grad is None:
            assert got[k] is None or float(got[k].abs().max()) == 0, k
            continue
        assert got[k] is not None, k
        scale = v.grad.abs().max().item()
        if k == "unnorm_rotations":
            continue                                        # isotropic: float noise around zero in both routes
        assert (got[k] - v.grad).abs().max().item() <= 2e-3 * scale + 1e-7, (k, (got[k] - v.grad).abs().max().item(), scale)
    seen = radius > 0
    assert torch.equal(variables_out["seen"], seen)
    assert torch.equal(variables_out["max_2D_radius"], torch.where(seen, radius.float(), torch.zeros_like(radius, dtype=torch.float32)))


@pytest.mark.gpu
def test_get_loss_mirror_rare_branches_and_screen_space_gradient(gpu_device):
    """The branches no shipped configuration takes: use_l1 = False (no depth term, no 'depth' entry), the unmasked tracking
    colour sum, an unknown dataset (refused BEFORE params / variables are touched), and variables['means2D']: a placeholder
    that explains itself on the fused route, the colour render's own screen-space tensor (with its gradient after
    backward, equal to what the plain operator returns) on the two-render route."""
    import diff_gaussian_rasterization as dgr
    import slam_callers as sc
    from diff_gaussian_rasterization import get_loss as gl
    from diff_gaussian_rasterization import losses
    get_loss = gl.get_loss
    dev = gpu_device
    params, st, K, go = _scene(dev, 2000, 64, 48, seed=3)
    g = torch.Generator().manual_seed(4)
    curr = {"cam": st, "im": torch.rand(3, 48, 64, generator=g).to(dev), "depth": (torch.rand(1, 48, 64, generator=g) + 1.0).to(dev),
            "w2c": torch.eye(4, device=dev), "intrinsics": K.to(dev), "id": 0}
    variables = {"max_2D_radius": torch.zeros(2000, device=dev)}
    w = {"im": 0.5, "depth": 1.0}

    def unfused(**kw):
        tg = sc.transform_to_frame(params, 0, gaussians_grad=False, camera_grad=True)
        im, _, _ = dgr.GaussianRasterizer(raster_settings=st)(**sc.transformed_params2rendervar(params, tg))
        ds, _, _ = dgr.GaussianRasterizer(raster_settings=st)(**sc.transformed_params2depthplussilhouette(params, curr["w2c"], tg))
        return losses.tracking_loss(im, ds, curr["im"], curr["depth"], **kw)

    loss, variables, wl = get_loss(params, curr, variables, 0, w, True, 0.9, False, False, tracking=True, dataset_name="tum")
    ref = unfused(sil_thres=0.9, w_im=0.5, w_depth=0.0)
    assert set(wl) == {"im", "loss"} and abs(loss.item() - ref.item()) <= 2e-5 * abs(ref.item())
    loss, variables, wl = get_loss(params, curr, variables, 0, w, False, 0.9, True, False, tracking=True, dataset_name="tum")
    ref = unfused(sil_thres=float("-inf"), w_im=0.5, w_depth=1.0, colour_over_all_pixels=True)
    masked = unfused(sil_thres=float("-inf"), w_im=0.5, w_depth=1.0)
    assert set(wl) == {"im", "depth", "loss"} and abs(loss.item() - ref.item()) <= 2e-5 * abs(ref.item())
    assert ref.item() >= masked.item()
    cpu_params = {k: v.detach().cpu() for k, v in params.items()}
    snapshot = dict(cpu_params)
    with pytest.raises(ValueError):                          # a dataset the reference has no presence mask for
        get_loss(cpu_params, curr, variables, 0, w, True, 0.9, True, False, tracking=True, dataset_name="kitti")
    assert all(cpu_params[k] is snapshot[k] for k in snapshot)              # refused before anything was converted
    with pytest.raises(RuntimeError, match="SCREEN_SPACE_GRADIENT"):
        variables["means2D"].grad
    try:
        gl.SCREEN_SPACE_GRADIENT = True
        for v in params.values():
            v.grad = None
        loss, variables, wl = get_loss(params, curr, variables, 0, w, True, 0.9, True, False, mapping=True, dataset_name="tum")
        loss.backward()
        m2d = variables["means2D"]
        assert isinstance(m2d, torch.Tensor) and m2d.shape == (2000, 3) and m2d.grad is not None
        # the same gradient from the plain operator, colour render alone
        tg = sc.transform_to_frame(params, 0, gaussians_grad=True, camera_grad=False)
        rv = sc.transformed_params2rendervar(params, tg)
        rv["means2D"].retain_grad()
        im, _, _ = dgr.GaussianRasterizer(raster_settings=st)(**rv)
        ds, _, _ = dgr.GaussianRasterizer(raster_settings=st)(**sc.transformed_params2depthplussilhouette(params, curr["w2c"], tg))
        losses.mapping_loss(im, ds, curr["im"], curr["depth"], w_im=0.5, w_depth=1.0).backward()
        scale = rv["means2D"].grad.abs().max().item()
        assert scale > 0 and (m2d.grad - rv["means2D"].grad).abs().max().item() <= 1e-4 * scale
    finally:
        gl.SCREEN_SPACE_GRADIENT = False
    loss, variables, wl = get_loss(params, curr, variables, 0, w, True, 0.9, True, False, mapping=True, dataset_name="tum")
    assert torch.isfinite(loss) and set(wl) == {"im", "depth", "loss"} and "seen" in variables
