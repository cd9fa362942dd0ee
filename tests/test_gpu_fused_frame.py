"""Fused caller chain (vtgs_prepare_frame* + render_frame): the HIP kernel against golden vectors captured from the
reference's own helpers, and the fused operator against the unfused chain (slam_callers + two GaussianRasterizer calls)."""
import ctypes
import os

import numpy as np
import pytest
import torch

import slam_callers as sc
from oracle import gs_oracle as go
from parity_util import to_settings

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_prepare_frame_kernel_matches_reference_fixture(gpu_device):
    """Inputs/outputs of transform_to_frame + both render-variable builders, captured from the reference's modules."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import fused  # noqa: F401  (sets the ctypes signatures)
    t = np.load(os.path.join(G, "helpers_transform.npz"))
    dev = gpu_device
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ti = int(t["time_idx"])
    means, logit, ls, ur = T(t["in_means3D"]), T(t["in_logit_opacities"]), T(t["in_log_scales"]), T(t["in_unnorm_rotations"])
    q, tr = T(t["in_cam_unnorm_rots"][0, :, ti]), T(t["in_cam_trans"][0, :, ti])
    w2c = T(t["first_frame_w2c"]).reshape(-1)
    n = means.shape[0]
    out = [torch.empty(n, k, device=dev) for k in (3, 1, 3, 4, 3)]
    st = dgr._lib.vtgs_prepare_frame(n, means.data_ptr(), logit.data_ptr(), ls.data_ptr(), ur.data_ptr(), q.data_ptr(),
                                     tr.data_ptr(), w2c.data_ptr(), *[o.data_ptr() for o in out],
                                     torch.cuda.current_stream().cuda_stream)
    assert st == 0
    for o, name in zip(out, ("rgb_means3D", "rgb_opacities", "rgb_scales", "rgb_rotations", "dep_colors_precomp")):
        np.testing.assert_allclose(o.cpu().numpy(), t[name], rtol=2e-5, atol=2e-6, err_msg=name)


def _params(dev, n, W, H, seed, T=3):
    scene, cam = go.view_tied_scene(n, W, H, seed=seed)
    g = torch.Generator().manual_seed(seed)
    p = {"means3D": scene["means3D"], "rgb_colors": scene["colors_precomp"],
         "unnorm_rotations": scene["rotations"] * (1 + 0.3 * torch.rand(n, 1, generator=g)),
         "logit_opacities": torch.randn(n, 1, generator=g), "log_scales": torch.log(scene["scales"][:, :1]),
         "cam_unnorm_rots": torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1).repeat(1, 1, T) + 0.01 * torch.randn(1, 4, T, generator=g),
         "cam_trans": 0.01 * torch.randn(1, 3, T, generator=g)}
    return {k: torch.nn.Parameter(v.to(dev)) for k, v in p.items()}, cam


def _images_agree(a, b, tol=2e-5, flips=3):
    """Two float32 routes to the same image: the camera-frame means of the torch chain and of vtgs_prepare_frame differ in the
    last bit, so a pair whose alpha sits within one ulp of 1/255 -- or a pixel whose transmittance sits within one ulp of the
    1e-4 stop -- is kept by one route and dropped by the other (seeds 5 / 7 of this scene: ONE pixel each, off by 1.3e-4 /
    7e-4 with the silhouette moving by the same amount).  Everywhere else the images agree to `tol`; a flipped pixel is bounded
    by what one such pair can carry."""
    err = (a - b).abs().max(0).values
    scale = a.abs().max().item()
    assert int((err > tol * scale).sum()) <= flips, (int((err > tol * scale).sum()), err.max().item())
    assert err.max().item() <= 4e-3 * max(scale, 1.0), err.max().item()


@pytest.mark.parametrize("gaussians_grad,camera_grad", [(False, True), (True, False), (True, True)])
def test_fused_render_frame_equals_unfused_chain(gpu_device, gaussians_grad, camera_grad):
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization.fused import render_frame
    dev = gpu_device
    params, cam = _params(dev, 12000, 160, 120, seed=5)
    st = to_settings(cam, dev)
    w2c = torch.eye(4, device=dev)
    w2c[:3, 3] = torch.tensor([0.02, -0.01, 0.03], device=dev)
    g = torch.Generator().manual_seed(3)
    g1 = (torch.rand(3, 120, 160, generator=g) * 2 - 1).to(dev)
    g2 = (torch.rand(3, 120, 160, generator=g) * 2 - 1).to(dev)
    t_idx = 2
    # unfused reference chain
    tg = sc.transform_to_frame(params, t_idx, gaussians_grad, camera_grad)
    im0, r0, _ = dgr.GaussianRasterizer(raster_settings=st)(**sc.transformed_params2rendervar(params, tg))
    ds0, _, _ = dgr.GaussianRasterizer(raster_settings=st)(**sc.transformed_params2depthplussilhouette(params, w2c, tg))
    ((im0 * g1).sum() + (ds0 * g2).sum()).backward()
    ref = {k: (None if v.grad is None else v.grad.clone()) for k, v in params.items()}
    for v in params.values():
        v.grad = None
    # fused
    im1, ds1, r1 = render_frame(params, t_idx, st, w2c, gaussians_grad, camera_grad)
    ((im1 * g1).sum() + (ds1 * g2).sum()).backward()
    assert torch.equal(r0, r1)
    _images_agree(im0, im1)
    _images_agree(ds0, ds1)
    for k, v in params.items():
        if ref[k] is None:
            assert v.grad is None or float(v.grad.abs().max()) == 0, k
            continue
        assert v.grad is not None, k
        scale = ref[k].abs().max().item()
        err = (ref[k] - v.grad).abs().max().item()
        if k == "unnorm_rotations":                     # isotropic: exactly zero in exact arithmetic
            assert err <= 1e-4 * ref["log_scales"].abs().max().item() if ref["log_scales"] is not None else True
            continue
        assert err <= 2e-3 * scale + 1e-7, (k, err, scale)


@pytest.mark.parametrize("gaussians_grad,camera_grad", [(False, True), (True, True)])
def test_render_frame_takes_anisotropic_maps(gpu_device, gaussians_grad, camera_grad):
    """log_scales [N,3] (utils/slam_helpers.py:376-383: the rotations are composed with the camera's): render_frame -- and
    with it the get_loss mirror -- serves them through the reference's own element-wise chain on the HIP operator; against
    two independent renders of the same chain."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization.fused import render_frame
    dev = gpu_device
    params, cam = _params(dev, 9000, 160, 120, seed=7)
    g = torch.Generator().manual_seed(17)
    n = params["means3D"].shape[0]
    params["log_scales"] = torch.nn.Parameter((params["log_scales"].detach().cpu() + 0.4 * torch.randn(n, 3, generator=g)).to(dev))
    params["unnorm_rotations"] = torch.nn.Parameter(torch.randn(n, 4, generator=g).to(dev))
    st = to_settings(cam, dev)
    w2c = torch.eye(4, device=dev)
    g1 = (torch.rand(3, 120, 160, generator=g) * 2 - 1).to(dev)
    g2 = (torch.rand(3, 120, 160, generator=g) * 2 - 1).to(dev)
    tg = sc.transform_to_frame(params, 1, gaussians_grad, camera_grad)
    assert not torch.equal(tg["unnorm_rotations"], params["unnorm_rotations"])          # the quat_mult branch
    im0, r0, _ = dgr.GaussianRasterizer(raster_settings=st)(**sc.transformed_params2rendervar(params, tg))
    ds0, _, _ = dgr.GaussianRasterizer(raster_settings=st)(**sc.transformed_params2depthplussilhouette(params, w2c, tg))
    ((im0 * g1).sum() + (ds0 * g2).sum()).backward()
    ref = {k: (None if v.grad is None else v.grad.clone()) for k, v in params.items()}
    for v in params.values():
        v.grad = None
    im1, ds1, r1 = render_frame(params, 1, st, w2c, gaussians_grad, camera_grad)
    ((im1 * g1).sum() + (ds1 * g2).sum()).backward()
    assert torch.equal(r0, r1) and torch.equal(im0, im1) and torch.equal(ds0, ds1)     # same kernels, second render over the first's bins
    for k, v in params.items():
        if ref[k] is None:
            assert v.grad is None or float(v.grad.abs().max()) == 0, k
            continue
        scale = ref[k].abs().max().item()
        assert (ref[k] - v.grad).abs().max().item() <= 1e-4 * scale + 1e-7, k
    assert ref["cam_unnorm_rots"].abs().max().item() > 0


@pytest.mark.parametrize("shape,n,opacity_boost", [((160, 120), 12000, 0.0), ((333, 201), 60000, 0.0), ((96, 64), 40000, 6.0)])
def test_dual_composite_equals_two_renders(gpu_device, monkeypatch, shape, n, opacity_boost):
    """vtgs_forward_dual / vtgs_backward_dual (one six-channel pass) against vtgs_forward + vtgs_forward_shared + two
    vtgs_backward (VTGS_DUAL=0): images bit-identical, gradients equal up to float32 summation order.  The third case
    saturates most pixels (stop rule, wave-uniform slow path); a non-zero background exercises the bg terms."""
    from diff_gaussian_rasterization.fused import render_frame
    dev = gpu_device
    W, H = shape
    params, cam = _params(dev, n, W, H, seed=9)
    if opacity_boost:
        with torch.no_grad():
            params["logit_opacities"] += opacity_boost
    st = to_settings(cam, dev, bg=torch.tensor([0.3, 0.1, 0.7]))
    w2c = torch.eye(4, device=dev)
    w2c[:3, 3] = torch.tensor([0.02, -0.01, 0.03], device=dev)
    g = torch.Generator().manual_seed(4)
    g1 = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)
    g2 = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VTGS_DUAL", mode)
        for v in params.values():
            v.grad = None
        im, ds, radii = render_frame(params, 1, st, w2c, True, True)
        ((im * g1).sum() + (ds * g2).sum()).backward()
        res[mode] = (im.detach().clone(), ds.detach().clone(), radii.clone(), {k: v.grad.clone() for k, v in params.items()})
    assert torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][1], res["1"][1]) and torch.equal(res["0"][2], res["1"][2])
    if opacity_boost:                                     # the saturating case must really stop pixels early (T < 1e-4)
        assert float((res["1"][1][1] > 0.9999).float().mean()) > 0.1
    for k in params:
        a, b = res["0"][3][k], res["1"][3][k]
        scale = a.abs().max().item()
        if k == "unnorm_rotations":                       # isotropic: float noise around zero in both routes
            continue
        assert (a - b).abs().max().item() <= 2e-4 * scale + 1e-7, (k, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("gaussians_grad,camera_grad", [(False, True), (True, False), (True, True)])
def test_frame_epilogue_in_the_gather_kernel_matches_the_two_kernel_route(gpu_device, monkeypatch, gaussians_grad, camera_grad):
    """vtgs_backward_dual_frame (adjoint of vtgs_prepare_frame applied inside gather_splat_grads, default) against
    vtgs_backward_dual + vtgs_prepare_frame_backward (VTGS_FRAME_EPILOGUE=0): same formulas, same reduction order; what is
    left is float32 rounding (the compiler contracts multiply-adds differently in the two instantiations of the gather
    kernel): 4e-6 of the largest gradient of each tensor."""
    from diff_gaussian_rasterization.fused import render_frame
    dev = gpu_device
    params, cam = _params(dev, 30000, 200, 136, seed=11)
    st = to_settings(cam, dev)
    w2c = torch.eye(4, device=dev)
    w2c[:3, 3] = torch.tensor([0.02, -0.01, 0.03], device=dev)
    g = torch.Generator().manual_seed(6)
    g1 = (torch.rand(3, 136, 200, generator=g) * 2 - 1).to(dev)
    g2 = (torch.rand(3, 136, 200, generator=g) * 2 - 1).to(dev)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VTGS_FRAME_EPILOGUE", mode)
        for v in params.values():
            v.grad = None
        im, ds, _ = render_frame(params, 1, st, w2c, gaussians_grad, camera_grad)
        ((im * g1).sum() + (ds * g2).sum()).backward()
        res[mode] = {k: (None if v.grad is None else v.grad.clone()) for k, v in params.items()}
    for k in params:
        a, b = res["0"][k], res["1"][k]
        assert (a is None) == (b is None), k
        if a is not None:
            scale = a.abs().max().item()
            if k == "unnorm_rotations":                   # isotropic map: float noise around an exact zero in both routes
                scale = res["0"]["log_scales"].abs().max().item()
            assert (a - b).abs().max().item() <= 4e-6 * scale + 1e-12, (k, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("gaussians_grad,camera_grad,contract,band", [(False, True, True, None), (True, False, True, None),
                                                                      (True, True, False, None), (False, True, True, (2, 6))])
def test_raw_activations_in_the_render_kernels_match_the_full_prepare(gpu_device, gaussians_grad, camera_grad, contract, band):
    """Round 6, VTGS_FORWARD_RAW_ACTIVATIONS (default of the C++ node without an owned list): project_and_bin and
    gather_splat_grads read logits / log-scales and apply sigmoid / exp themselves, the rotation of the isotropic map is not read
    (identity in the forward, the normalised parameter in the backward only when its gradient is wanted), and
    vtgs_prepare_frame_slot writes the camera-frame means and the depth colours only -- against set_frame_raw(False), the
    full prepare step.  Same formulas on the same float32 inputs: images within 2e-6 of the image maximum (the covariance
    s^2 R R^T of a unit quaternion is s^2 I up to rounding), radii equal, gradients within 1e-5 of each tensor's largest."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization.fused import render_frame
    if dgr._ext is None or not hasattr(dgr._ext, "set_frame_raw"):
        pytest.skip("the C++ node is not built")
    dev = gpu_device
    W, H = 200, 136
    params, cam = _params(dev, 30000, W, H, seed=13)
    st = to_settings(cam, dev, bg=torch.tensor([0.2, 0.4, 0.1]))
    w2c = torch.eye(4, device=dev)
    w2c[:3, 3] = torch.tensor([0.02, -0.01, 0.03], device=dev)
    g = torch.Generator().manual_seed(8)
    g1 = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)
    g2 = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)
    if contract:
        g2[1:] = 0                                        # the promise: the second image is differentiated through z alone
    res = {}
    try:
        for raw in (False, True):
            dgr.set_frame_raw(raw)
            for v in params.values():
                v.grad = None
            im, ds, radii = render_frame(params, 1, st, w2c, gaussians_grad, camera_grad, tile_rows=band, get_loss_contract=contract)
            ((im * g1).sum() + (ds[:1] * g2[:1]).sum() if contract else (im * g1).sum() + (ds * g2).sum()).backward()
            torch.cuda.synchronize()
            res[raw] = (im.detach().clone(), ds.detach().clone(), radii.clone(),
                        {k: (None if v.grad is None else v.grad.clone()) for k, v in params.items()})
    finally:
        dgr.set_frame_raw(True)
    # (identity against a normalised quaternion: the covariance differs in the last bit, so a radius that sits on an integer
    #  and a pair on the 1/255 threshold may fall on the other side -- _images_agree bounds both)
    dr = (res[False][2] - res[True][2]).abs()
    assert int((dr > 0).sum()) <= 3 and int(dr.max()) <= 1, (int((dr > 0).sum()), int(dr.max()))
    for i in (0, 1):
        a, b = res[False][i], res[True][i]
        if contract and i == 1:
            a, b = a[:1], b[:1]                           # (planes 1, 2: compared with thresholds / isnan only)
        _images_agree(a, b, tol=2e-6)
    seen = 0
    for k in params:
        a, b = res[False][3][k], res[True][3][k]
        assert (a is None) == (b is None), k
        if a is None:
            continue
        seen += 1
        scale = a.abs().max().item()
        if k == "unnorm_rotations":                       # isotropic map: float noise around an exact zero in both routes
            scale = max(v.abs().max().item() for kk, v in res[False][3].items() if v is not None and kk != "unnorm_rotations")
        assert (a - b).abs().max().item() <= 1e-5 * scale + 1e-12, (k, (a - b).abs().max().item(), scale)
    assert seen >= 2


def test_render_frame_bands_add_up_to_the_full_frame(gpu_device):
    """Tile-row partition of the fused operator: the band images tile the frame bit-exactly and the band
    gradients -- pose gradient and appearance gradients -- sum to the full-frame ones (float32 summation order)."""
    from diff_gaussian_rasterization.fused import render_frame
    from diff_gaussian_rasterization.partition import all_bands, pixel_rows
    dev = gpu_device
    W, H = 200, 136
    params, cam = _params(dev, 30000, W, H, seed=13)
    st = to_settings(cam, dev)
    w2c = torch.eye(4, device=dev)
    g = torch.Generator().manual_seed(8)
    g1 = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)
    g2 = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)
    keys = ("rgb_colors", "logit_opacities", "log_scales", "cam_unnorm_rots", "cam_trans")

    def run(band):
        for v in params.values():
            v.grad = None
        im, ds, radii = render_frame(params, 1, st, w2c, False, True, tile_rows=band)
        ((im * g1).sum() + (ds * g2).sum()).backward()
        return im.detach().clone(), ds.detach().clone(), radii.clone(), {k: params[k].grad.clone() for k in keys}
    im_f, ds_f, r_f, g_f = run(None)
    im_s, ds_s = torch.zeros_like(im_f), torch.zeros_like(ds_f)
    g_s = {k: torch.zeros_like(v) for k, v in g_f.items()}
    for band in all_bands(H, 3):
        im, ds, r, gb = run(band)
        y0, y1 = pixel_rows(band, H)
        assert float(im[:, :y0].abs().max() if y0 else 0) == 0 and float(im[:, y1:].abs().max() if y1 < H else 0) == 0
        assert torch.equal(im[:, y0:y1], im_f[:, y0:y1]) and torch.equal(ds[:, y0:y1], ds_f[:, y0:y1])
        assert bool(((r == 0) | (r == r_f)).all())                   # band-local radii: complete as the maximum over the ranks
        r_max = r.clone() if band == all_bands(H, 3)[0] else torch.maximum(r_max, r)
        im_s += im; ds_s += ds
        for k in keys:
            g_s[k] += gb[k]
    assert torch.equal(im_s, im_f) and torch.equal(ds_s, ds_f) and torch.equal(r_max, r_f)
    for k in keys:
        scale = g_f[k].abs().max().item()
        assert (g_s[k] - g_f[k]).abs().max().item() <= 2e-5 * scale + 1e-9, (k, (g_s[k] - g_f[k]).abs().max().item(), scale)


def test_pose7_reduce_matches_torch(gpu_device):
    """vtgs_pose7_reduce (what each rank of the tile-row partition all-reduces) against the plain tensor expressions."""
    from diff_gaussian_rasterization.partition import pose7_reduce
    g = torch.Generator().manual_seed(5)
    for n in (1, 255, 100003):
        p = torch.randn(n, 3, generator=g).to(gpu_device)
        gr = torch.randn(n, 3, generator=g).to(gpu_device)
        ref = torch.cat([gr.double().sum(0), torch.cross(p.double(), gr.double(), dim=1).sum(0), gr[:, 2:3].double().sum(0)])
        got = pose7_reduce(p, gr).double()
        assert (got - ref).abs().max().item() <= 1e-4 * (ref.abs().max().item() + n ** 0.5)


def test_cxx_node_equals_python_node_and_survives_capacity_growth(gpu_device, monkeypatch):
    """fused.render_frame through the C++ autograd node (csrc/vtgs_torch.cpp RenderFrame, the default) against the Python
    autograd.Function (VTGS_FUSED_EXT=0): images, radii and every gradient bit for bit -- both make the same C-ABI calls.  The
    second scene has the same (N, W, H) as the first and several times its instances: the capacities the first render settled on
    overflow, the checked forward grows them and runs again inside the policy wrapper of either route."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization.fused import render_frame
    dev = gpu_device
    if dgr._ext is None or not hasattr(dgr._ext, "render_frame"):
        pytest.skip("lib/vtgs_torch.so not built")
    W, H, n = 192, 128, 20000
    small, cam = _params(dev, n, W, H, seed=21)
    st = to_settings(cam, dev)
    w2c = torch.eye(4, device=dev)
    big = {k: torch.nn.Parameter(v.detach().clone()) for k, v in small.items()}
    with torch.no_grad():
        big["log_scales"] += 1.8                                   # 6 x the radius
    g = torch.Generator().manual_seed(8)
    g1 = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)
    g2 = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)

    def run(params):
        for v in params.values():
            v.grad = None
        im, ds, radii = render_frame(params, 1, st, w2c, True, True)
        ((im * g1).sum() + (ds * g2).sum()).backward()
        dgr.settle_pending()
        return im.detach().clone(), ds.detach().clone(), radii.clone(), {k: v.grad.clone() for k, v in params.items()}, \
            dgr.last_forward_info()["instances"]

    out = {}
    for route in ("cxx", "python"):
        monkeypatch.setenv("VTGS_FUSED_EXT", "1" if route == "cxx" else "0")
        for state in (dgr._capacity_hint, dgr._tile_cap_hint, dgr._caps_in_use, dgr._async_ok, dgr._need_hist):
            state.clear()                                              # both routes start from the first-forward capacities
        a = run(small)
        b = run(big)
        assert b[4] > 3.7 * a[4] + 4096, (a[4], b[4])                # beyond the 3.6 x room the first render left: it overflowed
        out[route] = (a, b)
    for which in (0, 1):
        x, y = out["cxx"][which], out["python"][which]
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2]), which
        for k in x[3]:
            assert torch.equal(x[3][k], y[3][k]), (which, k)


@pytest.mark.parametrize("route", ["cxx", "python"])
def test_render_frame_of_an_empty_map(gpu_device, monkeypatch, route):
    """N = 0 (a submap before its first densification): both images are the background, the pose gradient is zero and the
    per-Gaussian gradients are empty -- through either autograd node (an empty dual render once took the single-render kernel
    with its NULL depth plane: DESIGN.md 7.6)."""
    from diff_gaussian_rasterization.fused import render_frame
    monkeypatch.setenv("VTGS_FUSED_EXT", "1" if route == "cxx" else "0")
    dev = gpu_device
    full, cam = _params(dev, 100, 96, 64, seed=2)
    params = {k: torch.nn.Parameter(v.detach()[:0].clone() if k not in ("cam_unnorm_rots", "cam_trans") else v.detach().clone())
              for k, v in full.items()}
    st, w2c = to_settings(cam, dev), torch.eye(4, device=dev)
    im, ds, radii = render_frame(params, 1, st, w2c, True, True)
    (im.sum() + ds.sum()).backward()
    assert im.shape == (3, 64, 96) and radii.shape == (0,) and float(im.detach().abs().max()) == 0 and float(ds.detach().abs().max()) == 0
    assert float(params["cam_trans"].grad.abs().max()) == 0 and float(params["cam_unnorm_rots"].grad.abs().max()) == 0
    for k in ("means3D", "rgb_colors", "logit_opacities", "log_scales", "unnorm_rotations"):
        assert params[k].grad is None or params[k].grad.numel() == 0, k


@pytest.mark.parametrize("route", ["cxx", "python"])
def test_run_ahead_overflow_surfaces_inside_backward_before_the_optimizer_step(gpu_device, monkeypatch, route):
    """a run-ahead forward whose workspace overflows returned the background colour; the error has to
    leave `loss.backward()` -- through the C++ node too, whose backward never enters the interpreter (a post-hook on its graph
    node reads the record) -- so that an `optimizer.step()` written after it never runs on those gradients."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization.fused import render_frame
    from diff_gaussian_rasterization.optim import FusedAdam
    dev = gpu_device
    if route == "cxx" and (dgr._ext is None or not hasattr(dgr._ext, "render_frame")):
        pytest.skip("lib/vtgs_torch.so not built")
    monkeypatch.setenv("VTGS_FUSED_EXT", "1" if route == "cxx" else "0")
    W, H, n = 160, 96, 8001                                         # (n % 4 != 0: the C++ node's carved blocks stay aligned)
    base, cam = _params(dev, n, W, H, seed=5)
    st, w2c = to_settings(cam, dev), torch.eye(4, device=dev)
    for state in (dgr._capacity_hint, dgr._tile_cap_hint, dgr._caps_in_use, dgr._async_ok, dgr._need_hist):
        state.clear()
    params = {k: torch.nn.Parameter(v.detach().clone()) for k, v in base.items()}
    opt = FusedAdam([{"params": [v], "lr": 1e-3} for v in params.values()])
    ahead = 0
    for it in range(6):                                             # a steady loop: the later forwards run ahead
        opt.zero_grad(set_to_none=True)
        im, ds, _ = render_frame(params, 1, st, w2c, True, True)
        (im.sum() + ds.sum()).backward()
        opt.step()
    dgr.settle_pending()
    with torch.no_grad():
        params["log_scales"] += 2.6                                 # ~13 x the radius: far beyond three times the last need
    before = {k: v.detach().clone() for k, v in params.items()}
    opt.zero_grad(set_to_none=True)
    im, ds, _ = render_frame(params, 1, st, w2c, True, True)
    stepped = False
    with pytest.raises(RuntimeError, match="run-ahead mode"):
        (im.sum() + ds.sum()).backward()
        stepped = True                                              # (not reached: the error leaves backward())
        opt.step()
    assert not stepped
    for k, v in params.items():
        assert torch.equal(v.detach(), before[k]), k
    opt.zero_grad(set_to_none=True)                                 # the capacities were raised by the failed settle: valid now
    im, ds, _ = render_frame(params, 1, st, w2c, True, True)
    (im.sum() + ds.sum()).backward()
    dgr.settle_pending()
    assert float(im.detach().abs().max()) > 0


@pytest.mark.parametrize("route", ["cxx", "python"])
@pytest.mark.parametrize("scene", ["plain", "saturating"])
@pytest.mark.parametrize("gaussians_grad,camera_grad", [(False, True), (True, False), (True, True)])
def test_depth_only_backward_equals_the_full_dual_backward(gpu_device, monkeypatch, route, scene, gaussians_grad, camera_grad):
    """render_frame(get_loss_contract=True) -- frame flag 8: four image-gradient channels, 48-byte records, the second set's colour
    sum in chain w's fourth column -- against the full dual backward (VTGS_DUAL_B1=0 makes the library ignore the flag) on a
    gradient whose depth_sil planes 1 and 2 are zero, as get_loss sends it (a plain scene and a saturating one, whose pixels
    end inside their lists): the per-pair arithmetic is the same (the two
    dropped terms were exact zeros), so what is left is the contraction order of the compiler: 2e-6 of each tensor's largest
    gradient.  Tracking (Gaussians detached), mapping and bundle-adjustment flags, both autograd nodes."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization.fused import render_frame
    dev = gpu_device
    monkeypatch.setenv("VTGS_FUSED_EXT", "1" if route == "cxx" else "0")
    params, cam = _params(dev, 40000, 232, 136, seed=21)
    if scene == "saturating":                           # opaque, 2.7 x larger splats: pixels end inside their lists (the records of
        with torch.no_grad():                           # the entries behind are the zero-filled ones), lists of several hundred entries
            params["logit_opacities"] += 6.0
            params["log_scales"] += 1.0
    st = to_settings(cam, dev)
    w2c = torch.eye(4, device=dev)
    w2c[:3, 3] = torch.tensor([0.01, 0.02, -0.02], device=dev)
    g = torch.Generator().manual_seed(9)
    g1 = (torch.rand(3, 136, 232, generator=g) * 2 - 1).to(dev)
    g2 = (torch.rand(3, 136, 232, generator=g) * 2 - 1).to(dev)
    g2[1:] = 0.0
    res = {}
    try:
        for b1 in (0, 1):
            assert dgr._lib.vtgs_set_option(b"VTGS_DUAL_B1", b1) == 0
            for v in params.values():
                v.grad = None
            im, ds, _ = render_frame(params, 1, st, w2c, gaussians_grad, camera_grad, get_loss_contract=True)
            ((im * g1).sum() + (ds * g2).sum()).backward()
            res[b1] = {k: (None if v.grad is None else v.grad.clone()) for k, v in params.items()}
    finally:
        dgr._lib.vtgs_set_option(b"VTGS_DUAL_B1", -1)
    seen_any = False
    for k in params:
        a, b = res[0][k], res[1][k]
        assert (a is None) == (b is None), k
        if a is not None:
            seen_any = True
            scale = a.abs().max().item()
            if k == "unnorm_rotations":                   # isotropic map: float noise around an exact zero in both routes
                scale = res[0]["log_scales"].abs().max().item() if res[0]["log_scales"] is not None else 1.0
            # (saturating: the back sweep's anchor (CB - P) / T divides by a transmittance near the 1e-4 stop, which carries the
            #  last-bit difference of the two CB expressions -- the full form adds its two zero terms -- to a few 1e-6)
            tol = 2e-6 if scene == "plain" else 2e-5
            assert (a - b).abs().max().item() <= tol * scale + 1e-12, (k, (a - b).abs().max().item(), scale)
    assert seen_any


def test_get_loss_sends_no_gradient_into_silhouette_and_depth_squared(gpu_device, monkeypatch):
    """The promise behind frame flag 8: whatever get_loss is asked for (tracking with the silhouette mask, mapping), the
    gradient that reaches the [z, 1, z^2] render is zero in planes 1 and 2 (src/vtgaussian_slam.py:466-521).  Checked on the
    Python node, whose output tensor takes a hook."""
    from diff_gaussian_rasterization import fused, get_loss as gl
    dev = gpu_device
    monkeypatch.setenv("VTGS_FUSED_EXT", "0")
    params, cam = _params(dev, 30000, 200, 136, seed=5)
    st = to_settings(cam, dev)
    w2c = torch.eye(4, device=dev)
    with torch.no_grad():
        im, ds, _ = fused.render_frame(params, 1, st, w2c, False, False)
    gt = {"cam": st, "w2c": w2c, "im": (im + 0.05).clamp(0, 1), "depth": ds[0:1] * 1.02, "id": 1,
          "intrinsics": torch.eye(3, device=dev), "iter_gt_w2c_list": [w2c, w2c]}
    caught = []
    real = fused.render_frame

    def spy(*a, **k):
        out = real(*a, **k)
        if out[1].requires_grad:
            out[1].register_hook(lambda g: caught.append(g.detach().clone()))
        return out
    monkeypatch.setattr(gl, "render_frame", spy)
    variables = {"max_2D_radius": torch.zeros(params["means3D"].shape[0], device=dev),
                 "means2D_gradient_accum": torch.zeros(params["means3D"].shape[0], device=dev),
                 "denom": torch.zeros(params["means3D"].shape[0], device=dev)}
    for tracking in (True, False):
        for v in params.values():
            v.grad = None
        out = gl.get_loss(params, gt, variables, 1, {"im": 0.5, "depth": 1.0}, True, 0.5, True, True, tracking=tracking,
                          mapping=not tracking, tracking_iteration=1, dataset_name="replica",
                          presence_sil_mask_mse_ls=[0.0] if tracking else None, sil_thres_ls=[0.99] if tracking else None)
        out[0].backward()
    assert len(caught) == 2
    for g in caught:
        assert g[0].abs().max().item() > 0.0
        assert g[1:].abs().max().item() == 0.0


@pytest.mark.parametrize("route", ["cxx", "python"])
def test_get_loss_contract_forward_against_the_full_dual_forward(gpu_device, monkeypatch, route):
    """VTGS_FORWARD_SECOND_IS_DEPTH (render_frame(get_loss_contract=True)): the single render's kernel with z in its depth
    column against the full dual forward (VTGS_DEPTH_LITE=0 makes the library ignore the flag).  The colour image and plane 0
    (sum w z) are the same sums of the same terms: bit-identical.  Plane 1 is 1 - T_final instead of sum w: equal up to the
    rounding of ~35 float32 operations per pixel (2e-6).  Plane 2 is plane 0 squared: its difference to plane 0^2 -- all
    get_loss asks of it -- is exactly zero where the full render's is finite."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization.fused import render_frame
    dev = gpu_device
    monkeypatch.setenv("VTGS_FUSED_EXT", "1" if route == "cxx" else "0")
    params, cam = _params(dev, 60000, 264, 200, seed=31)
    st = to_settings(cam, dev, bg=torch.tensor([0.1, 0.2, 0.3]))     # a background that shows in all six planes
    w2c = torch.eye(4, device=dev)
    out = {}
    try:
        for lite in (0, 1):
            assert dgr._lib.vtgs_set_option(b"VTGS_DEPTH_LITE", lite) == 0
            with torch.no_grad():
                out[lite] = render_frame(params, 1, st, w2c, False, False, get_loss_contract=True)
    finally:
        dgr._lib.vtgs_set_option(b"VTGS_DEPTH_LITE", -1)
    (im0, ds0, r0), (im1, ds1, r1) = out[0], out[1]
    assert torch.equal(im0, im1) and torch.equal(r0, r1)
    assert torch.equal(ds0[0], ds1[0])
    assert (ds0[1] - ds1[1]).abs().max().item() <= 2e-6
    assert torch.equal(ds1[2], ds1[0] * ds1[0])
    assert torch.isfinite(ds0[2] - ds0[0] ** 2).all() and ds0[1].max().item() > 0.9


def test_deferred_run_ahead_overflow_is_recorded_not_raised(gpu_device, monkeypatch):
    """N-rank mode (`defer_run_ahead_overflow(True)`): the overflow of a run-ahead forward does not leave this rank's
    `backward()` -- it is counted for `partition.phase_overflows`, the collective that raises on every rank at the end of the
    phase -- and the capacities are raised as always, so the next forward of the shape is valid."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization.fused import render_frame
    from diff_gaussian_rasterization.partition import phase_overflows
    dev = gpu_device
    W, H, n = 160, 96, 8001
    base, cam = _params(dev, n, W, H, seed=5)
    st, w2c = to_settings(cam, dev), torch.eye(4, device=dev)
    for state in (dgr._capacity_hint, dgr._tile_cap_hint, dgr._caps_in_use, dgr._async_ok, dgr._need_hist):
        state.clear()
    params = {k: torch.nn.Parameter(v.detach().clone()) for k, v in base.items()}
    dgr.deferred_overflows()
    dgr.defer_run_ahead_overflow(True)
    try:
        for it in range(6):                                         # a steady loop: the later forwards run ahead
            for v in params.values():
                v.grad = None
            im, ds, _ = render_frame(params, 1, st, w2c, True, True)
            (im.sum() + ds.sum()).backward()
        dgr.settle_pending()
        assert dgr.deferred_overflows() == 0
        with torch.no_grad():
            params["log_scales"] += 2.6
        im, ds, _ = render_frame(params, 1, st, w2c, True, True)
        (im.sum() + ds.sum()).backward()                            # no exception on this rank
        with pytest.raises(RuntimeError, match="redo the phase"):   # (single process: the sum over ranks is this rank's count)
            phase_overflows(device=dev)
        assert dgr.deferred_overflows() == 0
        im, ds, _ = render_frame(params, 1, st, w2c, True, True)    # capacities were raised: valid now
        (im.sum() + ds.sum()).backward()
        assert phase_overflows(device=dev) == 0
        assert float(im.detach().abs().max()) > 0
    finally:
        dgr.defer_run_ahead_overflow(False)
        dgr.deferred_overflows()
