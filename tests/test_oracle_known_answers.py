"""Analytic known answers that pin the oracle (the reference holds no fixtures for this path: SURVEY 8c)."""
import math

import pytest
import torch

from oracle import gs_oracle as go

W, H, F = 64, 48, 40.0


def _cam(bg=None):
    cam = go.setup_camera(W, H, [[F, 0, W / 2 - 0.5], [0, F, H / 2 - 0.5], [0, 0, 1]], torch.eye(4))
    return cam if bg is None else cam._replace(bg=torch.tensor(bg))


def _render(means, scale, opac, cols, cam=None, **kw):
    n = means.shape[0]
    rot = torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1)
    return go.rasterize(means.double(), torch.zeros(n, 3).double(), opac.double(), cols.double(),
                        scale.double(), rot.double(), cam or _cam(), return_aux=True, **kw)


def test_empty_and_culled():
    c, r, d, aux = _render(torch.zeros(0, 3), torch.zeros(0, 3), torch.zeros(0, 1), torch.zeros(0, 3))
    assert c.shape == (3, H, W) and float(c.abs().max()) == 0 and r.numel() == 0
    # z <= 0.2 is culled (radii 0); exactly 0.2 too
    c, r, d, aux = _render(torch.tensor([[0.0, 0, 0.2], [0.0, 0, -3.0], [0.0, 0, 0.21]]), torch.full((3, 3), 0.01),
                           torch.full((3, 1), 0.9), torch.ones(3, 3))
    assert r.tolist()[:2] == [0, 0] and r[2] > 0
    # far off-screen: no tile touched => radii 0
    c, r, d, aux = _render(torch.tensor([[50.0, 0, 2.0]]), torch.full((1, 3), 0.05), torch.full((1, 1), 0.9), torch.ones(1, 3))
    assert r.tolist() == [0] and float(c.abs().max()) == 0


def test_single_centred_splat_closed_form():
    f32 = lambda x: float(torch.tensor(x, dtype=torch.float32))      # the inputs are float32 tensors
    z, s, o = 2.0, f32(0.1), f32(0.8)
    col = torch.tensor([[0.2, 0.5, 0.9]])
    c, r, d, aux = _render(torch.tensor([[0.0, 0, z]]), torch.full((1, 3), 0.1), torch.tensor([[0.8]]), col)
    sig2 = (s * F / z) ** 2 + 0.3                      # isotropic, on the optical axis: J J^T = (f/z)^2 I
    assert int(r[0]) == math.ceil(3 * math.sqrt(sig2))
    yy, xx = torch.meshgrid(torch.arange(H).double(), torch.arange(W).double(), indexing="ij")
    uc, vc = W / 2 - 1.0, H / 2 - 1.0                  # u = f X/Z + cx - 0.5
    al = o * torch.exp(-0.5 * ((xx - uc) ** 2 + (yy - vc) ** 2) / sig2)
    al = torch.where(al < go.ALPHA_MIN, torch.zeros_like(al), al)
    for ch in range(3):
        assert (c[ch] - col[0, ch].double() * al).abs().max().item() < 1e-7   # h.w + 1e-7 moves the centre by ~1e-7 px
    assert (d[0] - z * al).abs().max().item() < 1e-7
    assert (aux["T_final"] - (1 - al)).abs().max().item() < 1e-7


def test_saturation_order_and_stop_rule():
    cy, cx = H // 2 - 1, W // 2 - 1
    big = torch.full((1, 3), 0.2)
    # alpha saturates at 0.99
    c, *_ = _render(torch.tensor([[0.0, 0, 2.0]]), big, torch.ones(1, 1), torch.ones(1, 3))
    assert abs(float(c[0, cy, cx]) - go.ALPHA_MAX) < 1e-12
    # nearer splat first regardless of index order
    m = torch.tensor([[0.0, 0, 3.0], [0.0, 0, 2.0]])
    c, *_ = _render(m, big.repeat(2, 1), torch.tensor([[0.5], [1.0]]), torch.tensor([[1.0, 0, 0], [0.0, 1.0, 0]]))
    A = go.ALPHA_MAX
    assert abs(float(c[1, cy, cx]) - A) < 1e-12 and abs(float(c[0, cy, cx]) - 0.5 * (1 - A)) < 1e-12
    # equal depth: the lower Gaussian index is in front (stable sort)
    m = torch.tensor([[0.0, 0, 2.0], [0.0, 0, 2.0]])
    c, *_ = _render(m, big.repeat(2, 1), torch.tensor([[1.0], [0.5]]), torch.tensor([[1.0, 0, 0], [0.0, 1.0, 0]]))
    assert abs(float(c[0, cy, cx]) - A) < 1e-12 and abs(float(c[1, cy, cx]) - 0.5 * (1 - A)) < 1e-12
    # stop rule: T(1-alpha) < 1e-4 => the entry is NOT added
    m = torch.tensor([[0.0, 0, 2.0], [0.0, 0, 2.5], [0.0, 0, 3.0]])
    c, _, _, aux = _render(m, big.repeat(3, 1), torch.tensor([[1.0], [0.9], [1.0]]), torch.eye(3))
    o9 = float(torch.tensor(0.9))          # the float32 opacity that was passed in
    assert float(c[2, cy, cx]) == 0.0 and abs(float(c[1, cy, cx]) - o9 * (1 - A)) < 1e-12
    assert abs(float(aux["T_final"][cy, cx]) - (1 - A) * (1 - o9)) < 1e-12


def test_silhouette_identity_and_background():
    scene, cam = go.view_tied_scene(1500, 96, 64, seed=4)
    ones = dict(scene, colors_precomp=torch.ones(1500, 3))
    c, _, _, aux = go.rasterize(cam=cam, return_aux=True, dtype=torch.float64, **ones)
    assert (c[1] + aux["T_final"] - 1).abs().max().item() < 1e-12      # sum of weights + T_final == 1
    cam_bg = cam._replace(bg=torch.tensor([0.25, 0.5, 0.75]))
    cb, *_ = go.rasterize(cam=cam_bg, dtype=torch.float64, **ones)
    for ch, b in enumerate((0.25, 0.5, 0.75)):
        assert (cb[ch] - (c[ch] + aux["T_final"] * b)).abs().max().item() < 1e-12


def test_radius_rules():
    lam = torch.tensor([1.3, 4.0, 100.0])
    assert go.splat_radius(lam, torch.tensor([0.5, 0.5, 0.5]), "3sigma").tolist() == [4.0, 6.0, 30.0]
    r = go.splat_radius(lam, torch.tensor([0.5, 0.01, 1.0]), "opacity")
    assert r[0] <= 4 and r[1] <= 6 and r[2] <= 30 and r[1] < 6
    with pytest.raises(ValueError):
        go.splat_radius(lam, lam, "nope")


def test_band_render_equals_rows_of_full_render():
    scene, cam = go.view_tied_scene(2500, 96, 80, seed=8)
    full, *_ = go.rasterize(cam=cam, **scene)
    band, *_ = go.rasterize(cam=cam, tile_rows=(1, 3), **scene)
    assert torch.equal(band[:, 16:48], full[:, 16:48])
    assert float(band[:, :16].abs().max()) == 0 and float(band[:, 48:].abs().max()) == 0
