"""Planned bins (include/vtgs.h "Planned bins"; VERDICT r2 item 8: two-level bins): bins sized per tile by a plan the device
rewrites after every forward, so that one pile of Gaussians along a ray no longer sizes every bin of the frame.  Same lists,
same images, same gradients as uniform bins -- bit for bit -- in a fraction of the workspace."""
import pytest
import torch

from oracle import gs_oracle as go
from parity_util import GRAD_KEYS, to_settings

pytestmark = pytest.mark.gpu


def _run(scene, cam, dev, grad_color, mode, monkeypatch, tile_rows=None):
    import diff_gaussian_rasterization as dgr
    monkeypatch.setattr(dgr, "_BINS_MODE", mode)
    leaves = {k: v.detach().to(dev).clone().requires_grad_(True) for k, v in scene.items()}
    rast = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev), tile_rows=tile_rows)
    color, radii, depth = rast(**leaves)
    (color * grad_color.to(dev)).sum().backward()
    dgr.settle_pending()
    fs = rast._last_state
    lists = dgr.debug_tile_lists(rast)
    return {"color": color.detach().cpu(), "depth": depth.detach().cpu(), "radii": radii.cpu(),
            "grads": {k: leaves[k].grad.cpu() for k in GRAD_KEYS}, "offs": lists[0], "gid": lists[1],
            "planned": bool(fs.tile_cap & dgr.PLANNED), "ws_bytes": fs.workspace.numel(), "info": dgr.last_forward_info()}


def _same(a, b):
    assert torch.equal(a["offs"], b["offs"]) and torch.equal(a["gid"], b["gid"])          # the same lists in the same order
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["depth"], b["depth"]) and torch.equal(a["radii"], b["radii"])
    for k in GRAD_KEYS:
        assert torch.equal(a["grads"][k], b["grads"][k]), k


@pytest.mark.parametrize("scene_name", ["view_tied", "random_aniso"])
def test_planned_bins_reproduce_uniform_bins(gpu_device, monkeypatch, scene_name):
    if scene_name == "view_tied":
        scene, cam = go.view_tied_scene(40000, 333, 201, seed=3)
    else:
        scene, cam = go.random_scene(6000, 200, 136, seed=5, anisotropic=True)
    g = torch.Generator().manual_seed(1)
    grad_color = torch.rand(3, cam.image_height, cam.image_width, generator=g) * 2 - 1
    uni = _run(scene, cam, gpu_device, grad_color, "uniform", monkeypatch)
    assert not uni["planned"]
    for attempt in range(3):        # the first planned forward of a view starts from a uniform plan; the plan then follows the view
        pl = _run(scene, cam, gpu_device, grad_color, "planned", monkeypatch)
        assert pl["planned"]
        _same(uni, pl)
    # a band of the frame (tile-row partition) through planned bins
    band = (2, 5)
    _same(_run(scene, cam, gpu_device, grad_color, "uniform", monkeypatch, band),
          _run(scene, cam, gpu_device, grad_color, "planned", monkeypatch, band))
    back = _run(scene, cam, gpu_device, grad_color, "uniform", monkeypatch)
    assert not back["planned"]
    _same(uni, back)


def test_the_plan_follows_a_changing_view(gpu_device, monkeypatch):
    """Planned bins through a sequence of views whose lists grow: every forward rewrites the plan for the next one; a jump
    that outgrows a bin is caught by the checked forward and repeated with the plan the failed attempt left behind."""
    import diff_gaussian_rasterization as dgr
    scene, cam = go.view_tied_scene(30000, 320, 240, seed=9)
    g = torch.Generator().manual_seed(4)
    grad_color = torch.rand(3, cam.image_height, cam.image_width, generator=g) * 2 - 1
    for step, growth in enumerate((1.0, 1.1, 1.2, 1.9, 1.0)):
        sc = dict(scene, scales=scene["scales"] * growth)
        uni = _run(sc, cam, gpu_device, grad_color, "uniform", monkeypatch)
        pl = _run(sc, cam, gpu_device, grad_color, "planned", monkeypatch)
        _same(uni, pl)
        assert pl["info"]["instances"] == uni["info"]["instances"]


def test_one_dense_tile_no_longer_sizes_every_bin(gpu_device, monkeypatch):
    """A pile of 20 000 Gaussians on one pixel of an otherwise ordinary frame (500 k view-tied Gaussians, 1200x680): with
    uniform bins the workspace is the longest list times every tile; `auto` switches that view to planned bins after the
    first overflow report and needs the lists' total instead.  Identical results."""
    import diff_gaussian_rasterization as dgr
    W, H = 1200, 680
    scene, cam = go.view_tied_scene(500_000, W, H, seed=2)
    pile = 20_000
    g = torch.Generator().manual_seed(11)
    z = 1.0 + 3.0 * torch.rand(pile, generator=g)
    fx = W / 2.0
    px, py = 700.3, 333.6
    xyz = torch.stack([(px - (W / 2 - 0.5)) / fx * z, (py - (H / 2 - 0.5)) / fx * z, z], dim=1)
    extra = {"means3D": xyz, "means2D": torch.zeros(pile, 3), "colors_precomp": torch.rand(pile, 3, generator=g),
             "opacities": torch.full((pile, 1), 0.004), "scales": (z / fx)[:, None].repeat(1, 3),
             "rotations": torch.tensor([[1.0, 0, 0, 0]]).repeat(pile, 1)}
    sc = {k: torch.cat([scene[k], extra[k]], 0) for k in scene}
    grad_color = torch.rand(3, H, W, generator=g) * 2 - 1
    auto = _run(sc, cam, gpu_device, grad_color, "auto", monkeypatch)
    assert auto["planned"] and auto["info"]["max_tile_list"] >= pile
    assert auto["ws_bytes"] < 400 << 20, auto["ws_bytes"]
    again = _run(sc, cam, gpu_device, grad_color, "auto", monkeypatch)                 # steady state: the plan fits, no retry
    assert again["planned"] and again["ws_bytes"] == auto["ws_bytes"]
    uni = _run(sc, cam, gpu_device, grad_color, "uniform", monkeypatch)
    assert not uni["planned"] and uni["ws_bytes"] > 20 * auto["ws_bytes"]
    _same(uni, auto)
    _same(uni, again)


@pytest.mark.parametrize("contract", [False, True])
def test_fused_frame_and_shared_render_through_planned_bins(gpu_device, monkeypatch, contract):
    """The other consumers of the bins -- the dual render of `render_frame` with its frame-epilogue backward, and the second
    render over the first one's bins (`render_shared`) -- read the forward's copy of the plan: identical to uniform bins.
    contract: `render_frame(get_loss_contract=True)` -- the single render's forward kernel with z in its depth column and the
    four-channel backward -- through `vtgs_forward_dual_planned` (the gradient sent into planes 1, 2 is zero, as promised)."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization.fused import render_frame
    dev = gpu_device
    scene, cam = go.view_tied_scene(15000, 200, 136, seed=21)
    n = scene["means3D"].shape[0]
    g = torch.Generator().manual_seed(7)
    base = {"means3D": scene["means3D"], "rgb_colors": scene["colors_precomp"], "unnorm_rotations": scene["rotations"],
            "logit_opacities": torch.randn(n, 1, generator=g), "log_scales": torch.log(scene["scales"][:, :1]),
            "cam_unnorm_rots": torch.tensor([1.0, 0.003, -0.002, 0.001]).reshape(1, 4, 1),
            "cam_trans": torch.tensor([0.002, -0.001, 0.003]).reshape(1, 3, 1)}
    st = to_settings(cam, dev)
    w2c = torch.eye(4, device=dev)
    g1 = (torch.rand(3, cam.image_height, cam.image_width, generator=g) * 2 - 1).to(dev)
    g2 = (torch.rand(3, cam.image_height, cam.image_width, generator=g) * 2 - 1).to(dev)

    def run(mode):
        monkeypatch.setattr(dgr, "_BINS_MODE", mode)
        p = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in base.items()}
        im, ds, radii = render_frame(p, 0, st, w2c, True, True, get_loss_contract=contract)
        ((im * g1).sum() + ((ds * g2)[:1] if contract else ds * g2).sum()).backward()
        out = {"im": im.detach().clone(), "ds": ds.detach().clone(), "radii": radii.clone(),
               **{"g_" + k: v.grad.clone() for k, v in p.items() if v.grad is not None}}
        leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in scene.items()}
        rast = dgr.GaussianRasterizer(raster_settings=st)
        c, _, _ = rast(**leaves)
        z = leaves["means3D"][:, 2:3]
        c2, _ = rast.render_shared(torch.cat([z, torch.ones_like(z), z * z], 1).detach(),
                                   like=tuple(leaves[k] for k in ("means3D", "means2D", "opacities", "scales", "rotations")))
        ((c * g1).sum() + (c2 * g2).sum()).backward()
        dgr.settle_pending()
        out.update(c=c.detach().clone(), c2=c2.detach().clone(), planned=bool(rast._last_state.tile_cap & dgr.PLANNED),
                   **{"s_" + k: v.grad.clone() for k, v in leaves.items() if v.grad is not None})
        return out
    uni = run("uniform")
    for _ in range(2):
        pl = run("planned")
        assert pl["planned"] and not uni["planned"]
        for k, v in uni.items():
            if k != "planned":
                assert torch.equal(v, pl[k]), k


@pytest.mark.parametrize("order", ["raster", "shuffled"])
def test_windowed_tile_table_equals_global_atomic_bins(gpu_device, order):
    """Round 5: a frame with more 8x8 tiles than the LDS tile table holds (1752x1168: 32 K) bins through a WINDOW of the table
    that follows the tile rows a workgroup's Gaussians reach -- one pass for a raster-ordered (view-tied) map, several passes
    for a map in any order.  Same lists (after the sort), same image, same radii and bit-identical gradients as the
    run-aggregated global-atomic binning (VTGS_BIN_IMPL = 0)."""
    import diff_gaussian_rasterization as dgr
    from oracle import gs_oracle as go
    from parity_util import GRAD_KEYS, to_settings
    dev = gpu_device
    n, W, H = 300_000, 1752, 1168
    scene, cam = go.view_tied_scene(n, W, H, seed=11)
    if order == "shuffled":
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(3))
        scene = {k: v[perm].contiguous() for k, v in scene.items()}
    g = torch.Generator().manual_seed(4)
    grad_color = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)
    out = {}
    try:
        for impl in (0, 1):
            dgr.set_option("VTGS_BIN_IMPL", impl)
            leaves = {k: v.to(dev).requires_grad_(True) for k, v in scene.items()}
            rast = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev))
            c, r, d = rast(**leaves)
            c.backward(grad_color)
            dgr.settle_pending()
            offs, gid, _ = dgr.debug_tile_lists(rast)
            out[impl] = (c.detach().clone(), r.clone(), d.detach().clone(), offs, gid, {k: leaves[k].grad.clone() for k in GRAD_KEYS})
    finally:
        dgr.set_option("VTGS_BIN_IMPL", -1)
    a, b = out[0], out[1]
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])          # list lengths and the sorted lists
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for k in GRAD_KEYS:
        assert torch.equal(a[5][k], b[5][k]), k
    assert int(a[3][-1]) > n                                             # (something was binned)
