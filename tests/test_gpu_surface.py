"""The halves of the operator surface the reference never uses: `shs` instead of
`colors_precomp`, `cov3D_precomp` instead of scales + rotations -- each against the float64 oracle (its autograd is the
backward oracle), and the both-or-neither checks."""
import pytest
import torch

from oracle import gs_oracle as go
from parity_util import HIP_CENTRE_ERR_PX, audit_outliers, grad_error, tainted_gaussians, to_settings

pytestmark = pytest.mark.gpu


def _settings(cam, dev, degree):
    st = to_settings(cam, dev)
    return st._replace(sh_degree=degree)


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_sh_colours_against_the_oracle(gpu_device, degree):
    """vtgs_sh_forward / vtgs_sh_backward: colours, dL/dshs and the direction's share of dL/dmeans3D against autograd of the
    float64 restatement, with coefficients large enough that a good share of the channels is clamped at 0."""
    from diff_gaussian_rasterization import surface
    dev = gpu_device
    g = torch.Generator().manual_seed(10 + degree)
    n, K = 5000, 16
    means = torch.randn(n, 3, generator=g) * 2 + torch.tensor([0.0, 0.0, 4.0])
    shs = torch.randn(n, K, 3, generator=g) * 2.0
    campos = torch.tensor([0.3, -0.2, 0.1])
    w = torch.randn(n, 3, generator=g)
    m64, s64 = means.double().requires_grad_(True), shs.double().requires_grad_(True)
    ref = go.sh_colors(m64, s64, campos.double(), degree)
    (ref * w.double()).sum().backward()
    md, sd = means.to(dev).requires_grad_(True), shs.to(dev).requires_grad_(True)
    got = surface.sh_colors(md, sd, campos.to(dev), degree)
    (got * w.to(dev)).sum().backward()
    clamped = (ref == 0).double().mean().item()
    assert 0.05 < clamped < 0.6, clamped
    # a channel within float32 rounding of 0 may be clamped on one side only: compare away from the kink
    safe = (ref.detach().abs() > 1e-5) | (ref.detach() == 0)
    assert ((ref.detach() - got.detach().cpu().double()).abs()[safe]).max().item() <= 2e-6
    rows = safe.all(dim=1)
    assert grad_error(s64.grad[rows], sd.grad.cpu()[rows])[0] <= 1e-5
    if degree == 0:                                 # no dependence on the direction: autograd has no gradient at all
        assert m64.grad is None or float(m64.grad.abs().max()) == 0.0
        assert float(md.grad.abs().max()) == 0.0
    else:
        assert grad_error(m64.grad[rows], md.grad.cpu()[rows])[0] <= 1e-4
    if degree < 3:                                  # coefficients beyond the active degree get exactly zero
        assert float(sd.grad[:, (degree + 1) ** 2:].abs().max()) == 0.0


def _cov6(scales, rotations):
    R = go.quat_to_rotmat(rotations.double())
    S = R @ torch.diag_embed(scales.double() ** 2) @ R.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)


@pytest.mark.parametrize("band", [None, (2, 6)])
def test_cov3d_precomp_against_the_oracle(gpu_device, band):
    """GaussianRasterizer(..., cov3D_precomp=...): image, radii and the five gradients against the float64 oracle given the
    same covariances (audited outliers as everywhere), whole frame and a band; and against the scales + rotations call of the
    same Gaussians (the covariance it builds is this one to float32 rounding)."""
    import diff_gaussian_rasterization as dgr
    dev = gpu_device
    scene, cam = go.random_scene(3000, 200, 136, seed=12, anisotropic=True)
    cov64 = _cov6(scene["scales"], torch.nn.functional.normalize(scene["rotations"]))
    g = torch.Generator().manual_seed(5)
    grad_color = torch.rand(3, cam.image_height, cam.image_width, generator=g) * 2 - 1
    # oracle
    L = {k: scene[k].double().requires_grad_(True) for k in ("means3D", "means2D", "opacities", "colors_precomp")}
    c64 = cov64.clone().requires_grad_(True)
    ref_c, ref_r, ref_d, aux = go.rasterize(L["means3D"], L["means2D"], L["opacities"], L["colors_precomp"], None, None, cam,
                                           cov3D_precomp=c64, tile_rows=band, return_aux=True)
    (ref_c * grad_color.double()).sum().backward()
    # operator
    D = {k: scene[k].to(dev).requires_grad_(True) for k in ("means3D", "means2D", "opacities", "colors_precomp")}
    cd = cov64.float().to(dev).requires_grad_(True)
    rast = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev), tile_rows=band)
    color, radii, depth = rast(means3D=D["means3D"], means2D=D["means2D"], opacities=D["opacities"], colors_precomp=D["colors_precomp"],
                               cov3D_precomp=cd)
    (color * grad_color.to(dev)).sum().backward()
    diff = ref_r != radii.cpu()
    assert diff.double().mean().item() <= 2e-3
    taint = torch.zeros(3000, dtype=torch.bool)
    for r_img, g_img in ((ref_c, color), (ref_d, depth)):
        a = audit_outliers(r_img, g_img.detach().cpu(), aux, scene["opacities"], cam, 1e-4, centre_err_px=HIP_CENTRE_ERR_PX)
        assert not a["unexplained"] and a["max_rel"] <= 8e-3, a
        taint |= tainted_gaussians(aux, a["tiles"], 3000)
    for name, r, h in (("means3D", L["means3D"].grad, D["means3D"].grad), ("means2D", L["means2D"].grad, D["means2D"].grad),
                       ("opacities", L["opacities"].grad, D["opacities"].grad), ("colors", L["colors_precomp"].grad, D["colors_precomp"].grad),
                       ("cov3D", c64.grad, cd.grad)):
        mx, p999 = grad_error(r[~taint & ~diff], h.cpu()[~taint & ~diff])
        assert mx <= 2e-3 and p999 <= 2e-3, (name, mx, p999)
    # the scales + rotations signature on the same Gaussians
    S = {k: scene[k].to(dev) for k in scene}
    S["rotations"] = torch.nn.functional.normalize(S["rotations"])
    c2, r2, d2 = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev), tile_rows=band)(**S)
    # (on a band the scales + rotations call drops a Gaussian that cannot meet the rows before projecting it and reports radius 0
    #  for it, include/vtgs.h; the covariance call has no such pre-cull)
    met = r2 > 0
    assert (r2 != radii)[met].double().mean().item() <= 2e-3 and bool((radii[~met] >= 0).all())
    assert float((c2 - color.detach()).abs().max()) <= 2e-2 and float((c2 - color.detach()).abs().mean()) <= 1e-5


def test_render_with_shs_through_the_operator(gpu_device):
    """GaussianRasterizer(..., shs=...) with sh_degree = 2: the image and dL/dshs, dL/dmeans3D against the oracle render of the
    oracle's SH colours; and the both / neither checks of the published surface."""
    import diff_gaussian_rasterization as dgr
    dev = gpu_device
    scene, cam = go.view_tied_scene(6000, 160, 120, seed=3)
    g = torch.Generator().manual_seed(6)
    shs = torch.randn(6000, 9, 3, generator=g) * 0.4
    shs[:, 0] += 1.0
    grad_color = torch.rand(3, 120, 160, generator=g) * 2 - 1
    L = {k: scene[k].double().requires_grad_(True) for k in ("means3D", "means2D", "opacities", "scales", "rotations")}
    s64 = shs.double().requires_grad_(True)
    cols = go.sh_colors(L["means3D"], s64, cam.campos.double(), 2)
    ref_c, ref_r, ref_d, aux = go.rasterize(L["means3D"], L["means2D"], L["opacities"], cols, L["scales"], L["rotations"], cam,
                                           return_aux=True)
    (ref_c * grad_color.double()).sum().backward()
    D = {k: scene[k].to(dev).requires_grad_(True) for k in ("means3D", "means2D", "opacities", "scales", "rotations")}
    sd = shs.to(dev).requires_grad_(True)
    rast = dgr.GaussianRasterizer(raster_settings=_settings(cam, dev, 2))
    color, radii, depth = rast(means3D=D["means3D"], means2D=D["means2D"], opacities=D["opacities"], shs=sd, scales=D["scales"],
                               rotations=D["rotations"])
    (color * grad_color.to(dev)).sum().backward()
    a = audit_outliers(ref_c, color.detach().cpu(), aux, scene["opacities"], cam, 1e-4, centre_err_px=HIP_CENTRE_ERR_PX)
    assert not a["unexplained"] and a["max_rel"] <= 8e-3
    taint = tainted_gaussians(aux, a["tiles"], 6000)
    for name, r, h in (("shs", s64.grad, sd.grad), ("means3D", L["means3D"].grad, D["means3D"].grad),
                       ("opacities", L["opacities"].grad, D["opacities"].grad)):
        mx, p999 = grad_error(r[~taint], h.cpu()[~taint])
        assert mx <= 2e-3 and p999 <= 1e-3, (name, mx, p999)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=D["means3D"], means2D=D["means2D"], opacities=D["opacities"], scales=D["scales"], rotations=D["rotations"])
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=D["means3D"], means2D=D["means2D"], opacities=D["opacities"], shs=sd, colors_precomp=scene["colors_precomp"].to(dev),
             scales=D["scales"], rotations=D["rotations"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=D["means3D"], means2D=D["means2D"], opacities=D["opacities"], shs=sd, scales=D["scales"], rotations=D["rotations"],
             cov3D_precomp=torch.zeros(6000, 6, device=dev))
