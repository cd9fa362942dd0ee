"""Owned sets of the tile-row partition (include/vtgs.h "Owned sets", partition.OwnedSet): a rank that renders a band of
tile rows from the LIST of Gaussians that can meet it must get what the band render of the whole map gives -- image, radii
and per-Gaussian gradients bit for bit, the pose gradient to float32 rounding -- and must notice on the device when a Gaussian
outside the list could have met the band."""
import pytest
import torch

from oracle import gs_oracle as go
from parity_util import to_settings

pytestmark = pytest.mark.gpu
KEYS = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")


def _params(dev, n, W, H, seed, T=3):
    scene, cam = go.view_tied_scene(n, W, H, seed=seed)
    g = torch.Generator().manual_seed(seed)
    p = {"means3D": scene["means3D"], "rgb_colors": scene["colors_precomp"],
         "unnorm_rotations": scene["rotations"] * (1 + 0.3 * torch.rand(n, 1, generator=g)),
         "logit_opacities": torch.randn(n, 1, generator=g), "log_scales": torch.log(scene["scales"][:, :1]),
         "cam_unnorm_rots": torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1).repeat(1, 1, T) + 0.01 * torch.randn(1, 4, T, generator=g),
         "cam_trans": 0.01 * torch.randn(1, 3, T, generator=g)}
    return {k: torch.nn.Parameter(v.to(dev)) for k, v in p.items()}, cam


def _render(params, t_idx, st, w2c, band, owned, g1, g2, gaussians_grad, camera_grad, contract=False):
    from diff_gaussian_rasterization.fused import render_frame
    for v in params.values():
        v.grad = None
    im, ds, radii = render_frame(params, t_idx, st, w2c, gaussians_grad, camera_grad, tile_rows=band, owned=owned,
                                 get_loss_contract=contract)
    ((im * g1).sum() + ((ds * g2)[:1] if contract else ds * g2).sum()).backward()     # (the contract: no gradient into planes 1, 2)
    return im.detach(), ds.detach(), radii, {k: (None if v.grad is None else v.grad.clone()) for k, v in params.items()}


@pytest.mark.parametrize("contract", [False, True])
@pytest.mark.parametrize("gaussians_grad,camera_grad", [(False, True), (True, True)])
@pytest.mark.parametrize("band", [(0, 2), (3, 5), (6, 8)])
def test_render_of_the_list_equals_the_band_render_of_the_map(gpu_device, band, gaussians_grad, camera_grad, contract):
    from diff_gaussian_rasterization.partition import OwnedSet
    dev = gpu_device
    W, H, n = 208, 128, 30000                              # 8 tile rows
    params, cam = _params(dev, n, W, H, seed=9)
    st, w2c = to_settings(cam, dev), torch.eye(4, device=dev)
    g = torch.Generator().manual_seed(1)
    g1 = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)
    g2 = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)
    own = OwnedSet(params, 1, st, w2c, band, margin_px=8.0, growth=1.1)
    assert 0 < len(own) < 0.75 * n, len(own)               # a band of 2 rows of 8: most of the map is not on the list
    assert bool((own.idx[1:] > own.idx[:-1]).all())
    im0, ds0, r0, ref = _render(params, 1, st, w2c, band, None, g1, g2, gaussians_grad, camera_grad, contract)
    im1, ds1, r1, got = _render(params, 1, st, w2c, band, own, g1, g2, gaussians_grad, camera_grad, contract)
    assert own.escaped() == 0
    assert torch.equal(im0, im1) and torch.equal(ds0, ds1) and torch.equal(r0, r1)
    assert int((r0 > 0).sum()) > 100
    for k in KEYS:
        if ref[k] is None:
            assert got[k] is None or float(got[k].abs().max()) == 0, k
        else:
            assert torch.equal(ref[k], got[k]), (k, float((ref[k] - got[k]).abs().max()))
            off_list = torch.ones(n, dtype=torch.bool, device=dev)
            off_list[own.idx64] = False
            assert float(got[k][off_list].abs().max()) == 0, k          # nothing lands outside the list
    for k in ("cam_unnorm_rots", "cam_trans"):
        a, b = ref[k][0, :, 1], got[k][0, :, 1]
        assert float((a - b).norm()) <= 2e-5 * float(a.norm()), (k, a, b)
        assert float(got[k][0, :, 0].abs().max()) == 0 and float(got[k][0, :, 2].abs().max()) == 0


def test_a_phase_of_pose_steps_stays_inside_the_margin(gpu_device):
    """A tracking phase: the pose moves by Adam-sized steps (2 mm / 0.02 deg per iteration, 40 iterations); the list built at
    the start stays exact for every iteration and the device counter stays 0."""
    from diff_gaussian_rasterization.partition import OwnedSet
    dev = gpu_device
    W, H, n, band = 208, 128, 30000, (2, 4)
    params, cam = _params(dev, n, W, H, seed=4)
    st, w2c = to_settings(cam, dev), torch.eye(4, device=dev)
    g1 = torch.ones(3, H, W, device=dev)
    g2 = torch.ones(3, H, W, device=dev)
    own = OwnedSet(params, 1, st, w2c, band)               # default margin: 32 px, scales x 1.25
    for it in range(0, 40, 13):
        with torch.no_grad():
            params["cam_trans"][0, :, 1] += 13 * torch.tensor([0.002, -0.002, 0.002], device=dev)
            params["cam_unnorm_rots"][0, 1:, 1] += 13 * 0.00017
        a = _render(params, 1, st, w2c, band, None, g1, g2, False, True)
        b = _render(params, 1, st, w2c, band, own, g1, g2, False, True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), it
    assert own.escaped() == 0


def test_a_gaussian_that_escapes_the_list_is_counted(gpu_device):
    """The smallest margin (1 px): a pose jump of a few degrees brings Gaussians into the band that are not on the list.  The render of the list
    is then NOT the band render -- and the counter says so; after a rebuild both agree again."""
    from diff_gaussian_rasterization.partition import OwnedSet
    dev = gpu_device
    W, H, n, band = 208, 128, 30000, (2, 4)
    params, cam = _params(dev, n, W, H, seed=4)
    st, w2c = to_settings(cam, dev), torch.eye(4, device=dev)
    g1 = torch.ones(3, H, W, device=dev)
    g2 = torch.ones(3, H, W, device=dev)
    own = OwnedSet(params, 1, st, w2c, band, margin_px=1.0, growth=1.0)
    _render(params, 1, st, w2c, band, own, g1, g2, False, True)
    assert own.escaped() == 0
    with torch.no_grad():
        params["cam_unnorm_rots"][0, 1, 1] += 0.05         # ~5.7 degrees about x: the image moves by ~20 rows
    a = _render(params, 1, st, w2c, band, None, g1, g2, False, True)
    b = _render(params, 1, st, w2c, band, own, g1, g2, False, True)
    assert own.escaped() > 0
    assert not torch.equal(a[0], b[0])
    own = OwnedSet(params, 1, st, w2c, band, margin_px=1.0, growth=1.0)
    b = _render(params, 1, st, w2c, band, own, g1, g2, False, True)
    assert own.escaped() == 0 and torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    # scales that outgrow `growth` escape too
    with torch.no_grad():
        params["log_scales"] += 1.0
    _render(params, 1, st, w2c, band, own, g1, g2, False, True)
    assert own.escaped() > 0


def test_list_of_another_band_or_map_is_refused(gpu_device):
    from diff_gaussian_rasterization.fused import render_frame
    from diff_gaussian_rasterization.partition import OwnedSet
    dev = gpu_device
    params, cam = _params(dev, 5000, 208, 128, seed=2)
    st, w2c = to_settings(cam, dev), torch.eye(4, device=dev)
    own = OwnedSet(params, 1, st, w2c, (2, 4))
    with pytest.raises(ValueError, match="margin_px"):
        OwnedSet(params, 1, st, w2c, (2, 4), margin_px=0.0)
    with pytest.raises(ValueError, match="tile rows"):
        render_frame(params, 1, st, w2c, False, True, tile_rows=(4, 6), owned=own)
    with pytest.raises(ValueError, match="tile rows"):
        render_frame(params, 1, st, w2c, False, True, owned=own)
    smaller = {k: (torch.nn.Parameter(v[:4000].detach().clone()) if k in KEYS else v) for k, v in params.items()}
    with pytest.raises(ValueError, match="rebuild"):
        render_frame(smaller, 1, st, w2c, False, True, tile_rows=(2, 4), owned=own)


def test_plain_operator_over_the_list_equals_its_band_render(gpu_device):
    """`GaussianRasterizer(..., tile_rows=band, owned=OwnedSet.for_operator(...))`: the operator over the listed rows of its
    inputs -- anisotropic scales included -- gives the band render of all rows, and the gradients of all six inputs, bit for bit."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization.partition import OwnedSet
    dev = gpu_device
    W, H, n, band = 208, 128, 30000, (3, 5)
    scene, cam = go.view_tied_scene(n, W, H, seed=6)
    g = torch.Generator().manual_seed(2)
    scene["scales"] = scene["scales"] * (0.7 + 0.6 * torch.rand(n, 3, generator=g))
    st = to_settings(cam, dev)
    g1 = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)

    def run(owned):
        leaves = {k: v.to(dev).requires_grad_(True) for k, v in scene.items()}
        im, radii, depth = dgr.GaussianRasterizer(raster_settings=st, tile_rows=band, owned=owned)(**leaves)
        (im * g1).sum().backward()
        return im.detach(), radii, depth.detach(), {k: v.grad for k, v in leaves.items()}

    own = OwnedSet.for_operator(scene["means3D"].to(dev), scene["scales"].to(dev), st, band, margin_px=4.0, growth=1.05)
    assert 0 < len(own) < 0.75 * n
    a, b = run(None), run(own)
    assert own.escaped() == 0
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k
    with pytest.raises(ValueError, match="for_operator"):
        dgr.GaussianRasterizer(raster_settings=st, tile_rows=band, owned=OwnedSet(_params(dev, n, W, H, 6)[0], 1, st, torch.eye(4, device=dev), band))(
            **{k: v.to(dev) for k, v in scene.items()})


@pytest.mark.parametrize("route", ["cxx", "python"])
def test_an_empty_list_renders_the_background(gpu_device, monkeypatch, route):
    """A band no Gaussian can meet (the whole map projects into the upper third of the frame): the list is empty, the render is
    the band render of the map -- background -- and every gradient is zero, through either autograd node."""
    from diff_gaussian_rasterization.fused import render_frame
    from diff_gaussian_rasterization.partition import OwnedSet
    monkeypatch.setenv("VTGS_FUSED_EXT", "1" if route == "cxx" else "0")
    dev = gpu_device
    W, H, n, band = 208, 128, 4000, (6, 8)
    params, cam = _params(dev, n, W, H, seed=3)
    with torch.no_grad():
        m = params["means3D"]
        m[:, 1] = -0.45 * m[:, 2] - 0.1 * m[:, 1].abs()           # y / z < -0.45: rows near the top edge and above it
    st, w2c = to_settings(cam, dev), torch.eye(4, device=dev)
    own = OwnedSet(params, 1, st, w2c, band, margin_px=4.0, growth=1.05)
    assert len(own) == 0
    g1 = torch.ones(3, H, W, device=dev)
    a = _render(params, 1, st, w2c, band, None, g1, g1, True, True)
    b = _render(params, 1, st, w2c, band, own, g1, g1, True, True)
    assert own.escaped() == 0
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert float(b[0].abs().max()) == 0 and int(b[2].max()) == 0
    for k, v in b[3].items():
        assert v is None or float(v.abs().max()) == 0, k
