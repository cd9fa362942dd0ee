"""Every BASELINE.json configuration at its real size on the GPU (four of five never ran in a -m gpu test).

A full frame of 0.5-5 M Gaussians cannot be composited by the CPU oracle in test time, so each configuration is checked
three ways:
  (i)   a few 16-pixel tile rows against the float64 oracle -- colour, depth and the gradients of a loss that lives on
        those rows -- with the oracle restricted to the Gaussians whose tile rectangle meets the rows
        (parity_util.oracle_rows); every pixel above the 1e-4 tolerance must be explained by the outlier audit;
  (ii)  the radii of ALL Gaussians against the oracle's vectorised `preprocess`;
  (iii) size-independent properties of the full frame: linearity in the colours (render(c) + render(1-c) == render(1),
        the silhouette), silhouette in [0, 1], depth output == channel 0 of the [z, 1, z^2] render, every 8x8 tile list
        strictly (depth bits, index)-ordered, and two tile-row bands reassembling the full frame bit for bit.
The N > 2^21 key + value sort and the global-atomic binning (> 20 K tiles) run here at the sizes that select them.
"""
import pytest
import torch

from oracle import gs_oracle as go
from parity_util import (HIP_CENTRE_ERR_PX, GRAD_KEYS, audit_outliers, grad_error, oracle_instances_8x8, oracle_rows, rows_mask, run_hip,
                         tainted_gaussians, to_settings)

pytestmark = pytest.mark.gpu

# name: (N, W, H, tile rows checked against the oracle)
CONFIGS = {
    "replica_room0_500k_1200x680": (500_000, 1200, 680, [0, 21, 42]),          # BASELINE.json configs[1]
    "headline_1M_1200x680": (1_000_000, 1200, 680, [7, 30]),                    # the shape the metric is quoted on
    "tum_fr1_desk_300k_640x480": (300_000, 640, 480, [0, 14, 29]),              # configs[2]
    "scannet_2M_640x480": (2_000_000, 640, 480, [11]),                          # configs[3] (4-GPU config, here whole)
    "scannetpp_5M_1752x1168": (5_000_000, 1752, 1168, [36]),                    # configs[4]: N > 2^21, 32 K tiles
}
IMG_TOL, GRAD_TOL = 1e-4, 1e-3


def _grad_check(ref, got, sel, what, max_tol, p999_tol):
    """<= 1e-3 relative on the gradients, three ways: the largest difference against the largest gradient, the relative L2
    error, and the 99.9th percentile of the element-wise relative error (floor: 1e-3 of the largest gradient, as everywhere).
    The loss of these tests lives on a few tile rows, so most selected Gaussians only graze it and their gradients span five
    orders of magnitude; the small ones are sums of terms that cancel a hundredfold.  Until round 4 every term carried the
    ~3e-4 relative noise of float32 pixel centres (2^-24 x 1200 px against exponent slopes of a few per pixel) and the
    percentile was held to 1.25 x what the float32 CPU oracle achieves (1.55e-3 at 1200x680).  Round 5: the kernels carry the
    centre as a float32 pair (vtgs_math.h::project_splat, GeomRec::centre_lo) and the PLAIN 1e-3 holds at every BASELINE
    shape -- the headline frame measures 2-4e-4 (profiles/r5_grad_error_breakdown.txt)."""
    for k in GRAD_KEYS:
        if k == "rotations":
            continue                      # isotropic scene: exactly zero in exact arithmetic
        r, g = ref[k][sel].double(), got[k][sel].double()
        scale = ref[k].abs().max().item()
        if scale == 0 or r.numel() == 0:
            continue
        d = (r - g).abs()
        mx = (d.max() / scale).item()
        l2 = (d.norm() / (r.norm() + 1e-300)).item()
        p999 = torch.quantile((d / (r.abs() + 1e-3 * scale)).reshape(-1)[:4_000_000], 0.999).item()
        assert mx <= max_tol and l2 <= max_tol and (p999_tol is None or p999 <= p999_tol), \
            f"{what} grad {k}: max {mx:.2e} of max|ref|, rel L2 {l2:.2e}, p99.9 rel {p999:.2e} (bound {p999_tol})"


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_rows_radii_and_properties(gpu_device, name):
    import diff_gaussian_rasterization as dgr
    n, W, H, rows = CONFIGS[name]
    dev = gpu_device
    scene, cam = go.view_tied_scene(n, W, H, seed=5)
    g = torch.Generator().manual_seed(17)
    mask = rows_mask(cam, rows)
    grad_color = torch.rand(3, H, W, generator=g) * 2 - 1
    grad_color[:, ~mask] = 0                                            # the loss lives on the checked rows
    # ---- (i) rows against the oracle --------------------------------------------------------------------------
    ref_c, ref_r, ref_d, ref_g, keep, aux, idx = oracle_rows(scene, cam, rows, grad_color)
    got_c, got_r, got_d, got_g = run_hip(scene, cam, dev, grad_color)
    info = dgr.last_forward_info()
    sub_op = scene["opacities"][idx]
    hc = torch.where(mask[None, :, None], got_c.double(), ref_c)        # outside the rows the oracle image is just bg
    hd = torch.where(mask[None, :, None], got_d.double(), ref_d)
    a_c = audit_outliers(ref_c, hc, aux, sub_op, cam, IMG_TOL, centre_err_px=HIP_CENTRE_ERR_PX)
    a_d = audit_outliers(ref_d, hd, aux, sub_op, cam, IMG_TOL, centre_err_px=HIP_CENTRE_ERR_PX)
    n_px = int(mask.sum()) * W
    for label, a in (("colour", a_c), ("depth", a_d)):
        assert not a["unexplained"], f"{name} {label}: pixels above {IMG_TOL} not on a discrete decision: {a['unexplained'][:5]}"
        assert a["outliers"] <= 1e-3 * n_px and a["max_rel"] <= 8e-3, (name, label, a["outliers"], n_px, a["max_rel"])
    # ---- (ii) radii of all N -----------------------------------------------------------------------------------
    diff = ref_r != got_r
    assert diff.double().mean().item() <= 1e-3 and (ref_r - got_r).abs().max().item() <= 1, \
        f"{name}: radii differ for {int(diff.sum())} of {n} Gaussians"
    assert ((ref_r > 0) != (got_r > 0)).sum().item() <= 2
    # gradients: Gaussians that do not meet the rows get exactly nothing; the others match the oracle.  Gaussians that
    # share a 16x16 tile with an audited outlier pixel carry that pixel's flipped decision: looser bound for them.
    away = ~keep & ~diff                                                # (a radius on the other side of ceil() moves a rectangle)
    for k in GRAD_KEYS:
        assert float(got_g[k][away].abs().max()) == 0.0, f"{name}: gradient {k} outside the rows"
    taint = tainted_gaussians(aux, a_c["tiles"] | a_d["tiles"], idx.numel())
    taint_full = torch.zeros(n, dtype=torch.bool)
    taint_full[idx[taint]] = True
    _grad_check(ref_g, got_g, keep & ~taint_full, f"{name} (clean)", GRAD_TOL, GRAD_TOL)
    if taint_full.any():
        _grad_check(ref_g, got_g, taint_full, f"{name} (beside an outlier pixel)", 2e-2, None)
    # ---- (iii) full-frame properties ---------------------------------------------------------------------------
    st = to_settings(cam, dev)
    leaves = {k: v.to(dev) for k, v in scene.items()}
    z = leaves["means3D"][:, 2:3]
    with torch.no_grad():
        rast = dgr.GaussianRasterizer(raster_settings=st)
        c_a, r_a, d_a = rast(**leaves)
        offs, gid, geom = dgr.debug_tile_lists(rast)
        c_b, _, _ = dgr.GaussianRasterizer(raster_settings=st)(**dict(leaves, colors_precomp=1.0 - leaves["colors_precomp"]))
        c_1, _, _ = dgr.GaussianRasterizer(raster_settings=st)(**dict(leaves, colors_precomp=torch.ones_like(leaves["colors_precomp"])))
        c_z, _, d_z = dgr.GaussianRasterizer(raster_settings=st)(**dict(leaves, colors_precomp=torch.cat([z, torch.ones_like(z), z * z], 1)))
    assert torch.equal(c_a.cpu(), got_c) and torch.equal(r_a.cpu(), got_r)          # no-grad forward == grad-mode forward
    sil = c_1[0]
    assert float(sil.min()) >= 0.0 and float(sil.max()) <= 1.0 + 1e-6
    assert float((c_a + c_b - c_1).abs().max()) <= 2e-6, "render(c) + render(1 - c) != render(1)"
    assert float((c_z[1] - sil).abs().max()) <= 1e-6 and float((c_z[0] - d_z[0]).abs().max()) <= 1e-5 * float(d_z.abs().max())
    assert float((d_a - d_z).abs().max()) == 0.0                                     # depth output does not depend on the colours
    # every list strictly (depth bits, index)-ordered, all instances accounted for
    lens = offs[1:] - offs[:-1]
    assert int(offs[-1]) == info["instances"] and int(lens.max()) == info["max_tile_list"]
    zbits = geom[:, 6].view(torch.int32).long()
    key = (zbits[gid] << 32) | gid
    tile_of = torch.repeat_interleave(torch.arange(lens.numel()), lens)
    assert bool(((key[1:] > key[:-1]) | (tile_of[1:] != tile_of[:-1])).all()), f"{name}: a tile list is not strictly ordered"
    # two bands reassemble the frame bit for bit
    gy16 = (H + 15) // 16
    half = gy16 // 2
    with torch.no_grad():
        top, _, dt = dgr.GaussianRasterizer(raster_settings=st, tile_rows=(0, half))(**leaves)
        bot, _, db = dgr.GaussianRasterizer(raster_settings=st, tile_rows=(half, gy16))(**leaves)
    assert torch.equal(top[:, :half * 16], c_a[:, :half * 16]) and torch.equal(bot[:, half * 16:], c_a[:, half * 16:])
    assert torch.equal(dt[:, :half * 16], d_a[:, :half * 16]) and torch.equal(db[:, half * 16:], d_a[:, half * 16:])


def _hip_instance_keys(rast):
    import diff_gaussian_rasterization as dgr
    offs, gid, geom = dgr.debug_tile_lists(rast)
    lens = offs[1:] - offs[:-1]
    tile_of = torch.repeat_interleave(torch.arange(lens.numel()), lens)
    return (tile_of << 32) | gid, offs, gid, geom


@pytest.mark.parametrize("n,w,h,kind,rule", [(3000, 160, 120, "tied", "3sigma"), (60000, 152, 104, "tied", "3sigma"),
                                             (5000, 200, 136, "aniso", "3sigma"), (4000, 333, 201, "iso", "3sigma"),
                                             (1_000_000, 1200, 680, "tied", "3sigma"),
                                             # the tile rectangles of the other radius rule
                                             (60000, 152, 104, "tied", "opacity"), (5000, 200, 136, "aniso", "opacity")])
def test_tile_lists_equal_the_oracle_lists(gpu_device, n, w, h, kind, rule):
    """Index parity (bit-exact work): the kernel's 8x8 lists against the oracle's 16x16 parent lists filtered by the exact
    reach predicate.  Membership: strict oracle set <= kernel set <= oracle set with twice the kernel's documented slack
    (a member of the band contributes alpha < 1/255 everywhere in the tile: output-invariant).  Order: the kernel's depth
    bits equal the oracle's float32 sort key bit for bit and each list is strictly (depth bits, index)-ordered -- together
    with the set relation that is the oracle's order."""
    import diff_gaussian_rasterization as dgr
    if kind == "tied":
        scene, cam = go.view_tied_scene(n, w, h, seed=n % 89)
    else:
        scene, cam = go.random_scene(n, w, h, seed=n % 89, anisotropic=(kind == "aniso"))
    dev = gpu_device
    rast = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev), radius_rule=rule)
    with torch.no_grad():
        _, radii, _ = rast(**{k: v.to(dev) for k, v in scene.items()})
    hip_keys, offs, gid, geom = _hip_instance_keys(rast)
    with torch.no_grad():
        f = {k: v.double() for k, v in scene.items()}
        sp = go.preprocess(f["means3D"], f["means2D"], f["opacities"], f["scales"], f["rotations"], cam, rule)
    same_rect = sp.radii == radii.cpu()               # a float32 ceil() on the other side moves the rectangle: exclude those
    assert (~same_rect).double().mean().item() <= 1e-3
    strict = oracle_instances_8x8(sp, scene["opacities"], cam)
    loose = oracle_instances_8x8(sp, scene["opacities"], cam, 2e-4, 2e-4)
    keep = lambda keys: keys[same_rect[keys & 0xFFFFFFFF]]
    strict, loose, hip = keep(strict), keep(loose), keep(hip_keys)
    missing = ~torch.isin(strict, hip)
    extra = ~torch.isin(hip, loose)
    assert not bool(missing.any()), f"{int(missing.sum())} (Gaussian, tile) instances the oracle needs are not in the kernel's lists"
    assert not bool(extra.any()), f"{int(extra.sum())} instances in the kernel's lists lie outside the oracle's reach band"
    assert strict.numel() <= hip.numel() <= loose.numel()
    # depth key bits of every binned Gaussian == the oracle's float32 key
    binned = torch.unique(gid)
    assert torch.equal(geom[binned, 6].view(torch.int32), sp.zkey[binned].view(torch.int32))
    lens = offs[1:] - offs[:-1]
    zbits = geom[:, 6].view(torch.int32).long()
    key = (zbits[gid] << 32) | gid
    tile_of = torch.repeat_interleave(torch.arange(lens.numel()), lens)
    assert bool(((key[1:] > key[:-1]) | (tile_of[1:] != tile_of[:-1])).all())


def test_dual_render_against_two_oracle_renders(gpu_device):
    """f2: the six-channel dual composite (vtgs_forward_dual / vtgs_backward_dual) against TWO float64
    oracle renders -- the RGB pass and the [z, 1, z^2] pass of src/vtgaussian_slam.py:461,466 -- images and the summed
    gradients of a loss on both images."""
    import diff_gaussian_rasterization as dgr
    from parity_util import run_oracle
    scene, cam = go.view_tied_scene(9000, 168, 120, seed=29)
    H, W = cam.image_height, cam.image_width
    z = scene["means3D"][:, 2:3]
    dcol = torch.cat([z, torch.ones_like(z), z * z], dim=1)
    g = torch.Generator().manual_seed(31)
    g_a = torch.rand(3, H, W, generator=g) * 2 - 1
    g_b = (torch.rand(3, H, W, generator=g) * 2 - 1) * torch.tensor([0.3, 1.0, 0.05])[:, None, None]
    ref_a = run_oracle(scene, cam, g_a)
    ref_b = run_oracle(dict(scene, colors_precomp=dcol), cam, g_b)
    dev = gpu_device
    t = {k: v.to(dev).contiguous() for k, v in scene.items()}
    cam_rec = dgr._Camera(to_settings(cam, dev), dev, 0, None)
    col_a, col_b = t["colors_precomp"], dcol.to(dev).contiguous()
    im_a, radii, im_b, fs = dgr._run_forward(cam_rec, t["means3D"], col_a, t["opacities"], t["scales"], t["rotations"],
                                             want_async=False, colors_b=col_b)
    grads = dgr._run_backward_dual(fs, t["means3D"], col_a, col_b, t["opacities"], t["scales"], t["rotations"], im_a, im_b,
                                   g_a.to(dev), g_b.to(dev))
    g_means3D, g_means2D, g_ca, g_op, g_sc, g_rot, g_cb = [x.cpu() for x in grads]
    a1 = audit_outliers(ref_a[0], im_a.cpu(), ref_a[4], scene["opacities"], cam, IMG_TOL, centre_err_px=HIP_CENTRE_ERR_PX)
    a2 = audit_outliers(ref_b[0], im_b.cpu(), ref_b[4], scene["opacities"], cam, IMG_TOL, centre_err_px=HIP_CENTRE_ERR_PX)
    for a in (a1, a2):
        assert not a["unexplained"] and a["max_rel"] <= 8e-3 and a["frac"] <= 1e-3, a
    taint = tainted_gaussians(ref_a[4], a1["tiles"] | a2["tiles"], z.shape[0])
    ref = {"means3D": ref_a[3]["means3D"] + ref_b[3]["means3D"], "means2D": ref_a[3]["means2D"] + ref_b[3]["means2D"],
           "opacities": ref_a[3]["opacities"] + ref_b[3]["opacities"], "scales": ref_a[3]["scales"] + ref_b[3]["scales"],
           "colors_precomp": ref_a[3]["colors_precomp"], "rotations": ref_a[3]["rotations"]}
    got = {"means3D": g_means3D, "means2D": g_means2D, "opacities": g_op, "scales": g_sc, "colors_precomp": g_ca,
           "rotations": g_rot}
    _grad_check(ref, got, ~taint, "dual (clean)", GRAD_TOL, GRAD_TOL)
    mx, p999 = grad_error(ref_b[3]["colors_precomp"][~taint], g_cb[~taint])
    assert mx <= 2 * GRAD_TOL and p999 <= GRAD_TOL, ("second colour set", mx, p999)
