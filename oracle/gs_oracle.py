"""CPU oracle for the differentiable Gaussian-splat rasterizer  --  TEST INFRASTRUCTURE ONLY.

This module is the parity checker and the reported CPU baseline.  It is never imported by the
product package (`vtgaussian-slam_amd/diff_gaussian_rasterization`); only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may use it.

PARITY UNPINNED.  The arithmetic restated here lives in a third-party dependency of the reference
that is absent from /root/reference and carries no version pin:
    requirements.txt:19  git+https://github.com/pengchongH/diff-gaussian-rasterization-w-depth-smallerGSradii.git
The reference holds no tests, golden vectors or fixtures for this path, so the
oracle is a restatement of the *published* 3D-Gaussian-splatting EWA rasterizer (with the w-depth fork's
extra depth accumulator) anchored on the reference's own call sites:
    utils/recon_helpers.py:4-27      camera record: viewmatrix = w2c^T, projmatrix = w2c^T . P^T, bg, tanfov
    utils/slam_helpers.py:127-160    the six tensors handed to the operator (rgb pass)
    utils/slam_helpers.py:217-287    the [z, 1, z^2] colour channels (depth / silhouette pass)
    src/vtgaussian_slam.py:460-468   (color, radii, depth) 3-tuple; means2D.retain_grad(); silhouette = ch1
    src/vtgaussian_slam.py:681-683   radii > 0 is the `seen` mask
and on analytic known answers (tests/test_oracle_known_answers.py).  The one rule that cannot be read
anywhere (the fork's "smaller radii") is isolated in `splat_radius()` behind `radius_rule`.

Semantics restated:
  preprocess  : view transform, cull z <= 0.2, EWA covariance J W S W^T J^T with the 1.3*tanfov clamp,
                +0.3 dilation, conic, eigen-radius, pixel centre ((ndc+1)*S-1)/2, 16x16 tile rectangle.
  binning     : key = (tile, float32 depth bits), stable => ties keep Gaussian index order.  The tile
                rectangle floor() is taken RECT_EPS above its argument (lattice-aligned splats, see RECT_EPS).
  composite   : front-to-back; skip power>0; alpha=min(.99, o*exp(power)); skip alpha<1/255;
                stop before adding when T*(1-alpha) < 1e-4; C += c*alpha*T; D += z*alpha*T; out = C + T*bg.
  backward    : torch autograd through the forward above, with the two places where the published
                hand-written backward deviates from the true derivative mimicked on purpose:
                (1) the 0.99 clamp passes gradient straight through, (2) a clamped t.x/t.y carries no
                gradient to t.z.  The depth image is a non-differentiable output.
`means2D` (all zeros at the call site, utils/slam_helpers.py:158) is added in NDC units to the pixel
centre so that autograd yields the NDC-scaled screen-space gradient the operator reports in means2D.grad.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional, Tuple

import torch

TILE = 16                 # binning granularity that defines which pixels a splat may reach
def _f32(x: float) -> float:
    """Thresholds are the float32 constants of the kernels (1.f/255.f, 0.99f, 1e-4f, 0.2f), so a float32 input
    sitting exactly on one of them resolves the same way whatever dtype the oracle runs in."""
    return float(torch.tensor(x, dtype=torch.float32))


ALPHA_MIN = _f32(1.0 / 255.0)
ALPHA_MAX = _f32(0.99)
T_STOP = _f32(1e-4)
NEAR_CULL = _f32(0.2)
DILATION = 0.3
FOV_CLAMP = 1.3
RECT_EPS = 1e-4           # tile units; see kRectEps in vtgaussian-slam_amd/csrc/vtgs_math.h


class OracleCamera(NamedTuple):
    """Same 11 fields, same order, as the record built at utils/recon_helpers.py:14-26."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool


def setup_camera(w, h, k, w2c, near=0.01, far=100.0, dtype=torch.float32) -> OracleCamera:
    """Restates the 9 lines of maths at utils/recon_helpers.py:5-13 (that file cannot be imported:
    it needs the absent rasterizer package).  Returns CPU tensors."""
    fx, fy, cx, cy = float(k[0][0]), float(k[1][1]), float(k[0][2]), float(k[1][2])
    w2c = torch.as_tensor(w2c, dtype=torch.float32)
    cam_center = torch.inverse(w2c)[:3, 3]
    view = w2c.unsqueeze(0).transpose(1, 2)
    proj = torch.tensor([[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0],
                         [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
                         [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
                         [0.0, 0.0, 1.0, 0.0]], dtype=torch.float32).unsqueeze(0).transpose(1, 2)
    full = view.bmm(proj)
    return OracleCamera(h, w, w / (2 * fx), h / (2 * fy), torch.zeros(3, dtype=torch.float32), 1.0,
                        view.to(dtype), full.to(dtype), 0, cam_center.to(dtype), False)


def splat_radius(lam_max: torch.Tensor, opacity: torch.Tensor, rule: str) -> torch.Tensor:
    """Screen-space radius in pixels (float, already ceil'ed).

    "3sigma"  : ceil(3 sqrt(lambda_max))  -- the published rule, and the default.
    "opacity" : ceil(sqrt(2 ln(255 o) lambda_max)), never above the 3-sigma value -- one plausible
                reading of the unreadable "smallerGSradii" fork (hard parts).
    """
    r3 = torch.ceil(3.0 * torch.sqrt(lam_max))
    if rule == "3sigma":
        return r3
    if rule == "opacity":
        ext = 2.0 * torch.log(torch.clamp(255.0 * opacity, min=1.0))
        return torch.minimum(r3, torch.ceil(torch.sqrt(ext * lam_max)))
    raise ValueError(f"unknown radius rule {rule!r}")


def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    """(w,x,y,z) -> R, *not* re-normalised (the caller normalises: utils/slam_helpers.py:155).
    Same convention as utils/slam_external.py:29-41."""
    r, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)


def depth_sort_key(means3D: torch.Tensor, V: torch.Tensor) -> torch.Tensor:
    """float32 view-space z used as the sort key.  Evaluated as the fused chain
    fma(V02,x, fma(V12,y, fma(V22,z, V32))) -- the order the HIP preprocess uses.  Each fma is emulated as
    an exact float64 product-sum (a product of two float32 is exact in float64) rounded to float32."""
    m = means3D.detach().to(torch.float32).to(torch.float64)
    v = V.detach().to(torch.float32).to(torch.float64)
    f32 = lambda t: t.to(torch.float32).to(torch.float64)
    acc = f32(m[:, 2] * v[2, 2] + v[3, 2])
    acc = f32(m[:, 1] * v[1, 2] + acc)
    acc = f32(m[:, 0] * v[0, 2] + acc)
    return acc.to(torch.float32)


class Splats(NamedTuple):
    visible: torch.Tensor     # [N] bool
    xy: torch.Tensor          # [N,2] pixel centre (includes the means2D hook)
    conic: torch.Tensor       # [N,3] (A,B,C)
    depth: torch.Tensor       # [N] view z (differentiable copy)
    zkey: torch.Tensor        # [N] float32 sort key
    radii: torch.Tensor       # [N] int32
    rect: torch.Tensor        # [N,4] int64 tile rectangle (x0,y0,x1,y1), half-open


def preprocess(means3D, means2D, opacities, scales, rotations, cam, radius_rule="3sigma",
               cov3D_precomp=None) -> Splats:
    dt = means3D.dtype
    H, W = int(cam.image_height), int(cam.image_width)
    V = cam.viewmatrix.reshape(4, 4).to(dt)
    PV = cam.projmatrix.reshape(4, 4).to(dt)
    N = means3D.shape[0]
    ones = torch.ones(N, 1, dtype=dt)
    p4 = torch.cat([means3D, ones], dim=1)
    t = (p4 @ V)[:, :3]                                  # row-vector convention: memory holds w2c^T
    hom = p4 @ PV
    ndc = hom[:, :3] / (hom[:, 3:4] + 1e-7)

    if cov3D_precomp is None:
        R = quat_to_rotmat(rotations)
        S2 = (scales * cam.scale_modifier) ** 2
        cov3 = R @ torch.diag_embed(S2) @ R.transpose(1, 2)
    else:                                                # upper-triangular 6-vector
        c = cov3D_precomp
        cov3 = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4],
                            c[:, 2], c[:, 4], c[:, 5]], dim=-1).reshape(-1, 3, 3)

    fx = W / (2.0 * cam.tanfovx)
    fy = H / (2.0 * cam.tanfovy)
    tz = t[:, 2]
    safe_tz = torch.where(tz.abs() < 1e-12, torch.full_like(tz, 1e-12), tz)
    limx, limy = FOV_CLAMP * cam.tanfovx, FOV_CLAMP * cam.tanfovy
    rx, ry = t[:, 0] / safe_tz, t[:, 1] / safe_tz
    # a clamped component is a constant as far as the published backward is concerned
    tx = torch.where((rx < -limx) | (rx > limx), (rx.clamp(-limx, limx) * safe_tz).detach(), t[:, 0])
    ty = torch.where((ry < -limy) | (ry > limy), (ry.clamp(-limy, limy) * safe_tz).detach(), t[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / safe_tz, zero, -fx * tx / (safe_tz * safe_tz),
                     zero, fy / safe_tz, -fy * ty / (safe_tz * safe_tz)], dim=-1).reshape(-1, 2, 3)
    Rw2c = V[:3, :3].t()
    M = J @ Rw2c
    cov2 = M @ cov3 @ M.transpose(1, 2)
    a = cov2[:, 0, 0] + DILATION
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + DILATION
    det = a * c - b * b
    det_ok = det != 0
    sdet = torch.where(det_ok, det, torch.ones_like(det))
    conic = torch.stack([c / sdet, -b / sdet, a / sdet], dim=-1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))   # the larger eigenvalue
    radius = splat_radius(lam.detach(), opacities.detach().reshape(-1), radius_rule)

    u = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5 + means2D[:, 0] * (0.5 * W)
    v = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5 + means2D[:, 1] * (0.5 * H)
    xy = torch.stack([u, v], dim=-1)

    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    ud, vd, rd = u.detach(), v.detach(), radius
    finite = torch.isfinite(ud) & torch.isfinite(vd) & torch.isfinite(rd)
    ud, vd, rd = [torch.where(finite, q, torch.zeros_like(q)) for q in (ud, vd, rd)]
    x0 = torch.clamp(torch.floor((ud - rd) / TILE + RECT_EPS), 0, gx).long()
    x1 = torch.clamp(torch.floor((ud + rd + TILE - 1) / TILE + RECT_EPS), 0, gx).long()
    y0 = torch.clamp(torch.floor((vd - rd) / TILE + RECT_EPS), 0, gy).long()
    y1 = torch.clamp(torch.floor((vd + rd + TILE - 1) / TILE + RECT_EPS), 0, gy).long()
    area = (x1 - x0) * (y1 - y0)
    visible = (tz.detach() > NEAR_CULL) & det_ok & (area > 0) & finite
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    rect = torch.stack([x0, y0, x1, y1], dim=-1)
    return Splats(visible, xy, conic, tz, depth_sort_key(means3D, V), radii, rect)


SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


def sh_colors(means3D: torch.Tensor, shs: torch.Tensor, campos: torch.Tensor, degree: int) -> torch.Tensor:
    """Colours from spherical harmonics, the operator's `shs` argument [UPSTREAM-PUBLIC: the published real-SH evaluation up to
    degree 3, view direction = normalised (mean - campos), + 0.5, negative channels clamped to 0].  shs [N, K, 3] with
    K >= (degree + 1)^2.  Differentiable (autograd is the backward oracle).  The reference never takes this path
    (sh_degree = 0, colours only: utils/recon_helpers.py:22, utils/slam_helpers.py:152-159)."""
    d = means3D - campos.reshape(1, 3).to(means3D.dtype)
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if degree > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
    if degree > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2.0 * zz - xx - yy) * shs[:, 6]
               + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
    if degree > 2:
        res = (res + SH_C3[0] * y * (3.0 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
               + SH_C3[2] * y * (4.0 * zz - xx - yy) * shs[:, 11] + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * shs[:, 12]
               + SH_C3[4] * x * (4.0 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
               + SH_C3[6] * x * (xx - 3.0 * yy) * shs[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def build_tile_lists(sp: Splats, gx: int, gy: int):
    """Stable (tile, depth-bits) ordering; returns sorted Gaussian ids and [tiles+1] offsets."""
    vis = torch.nonzero(sp.visible).reshape(-1)
    if vis.numel() == 0:
        return torch.zeros(0, dtype=torch.long), torch.zeros(gx * gy + 1, dtype=torch.long)
    r = sp.rect[vis]
    w = r[:, 2] - r[:, 0]
    h = r[:, 3] - r[:, 1]
    cnt = w * h
    owner = torch.repeat_interleave(torch.arange(vis.numel()), cnt)
    first = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(int(cnt.sum())) - first[owner]
    ty = r[owner, 1] + local // w[owner]
    tx = r[owner, 0] + local % w[owner]
    tile = ty * gx + tx
    gid = vis[owner]
    zbits = sp.zkey[gid].view(torch.int32).long()          # positive floats: bit order == value order
    key = (tile << 32) | zbits
    order = torch.sort(key, stable=True).indices           # instances were emitted in Gaussian order
    counts = torch.bincount(tile, minlength=gx * gy)
    offs = torch.zeros(gx * gy + 1, dtype=torch.long)
    offs[1:] = torch.cumsum(counts, 0)
    return gid[order], offs


def composite_tile(px, py, xy, conic, op, col, dep, bg):
    """One tile.  px,py [P]; per-list tensors [L,...].  Returns color [P,C], depth [P], T_final [P]."""
    dx = xy[None, :, 0] - px[:, None]
    dy = xy[None, :, 1] - py[:, None]
    power = -0.5 * (conic[None, :, 0] * dx * dx + conic[None, :, 2] * dy * dy) - conic[None, :, 1] * dx * dy
    a_raw = op[None, :] * torch.exp(torch.clamp(power, max=0.0))
    alpha = a_raw + (torch.clamp(a_raw, max=ALPHA_MAX) - a_raw).detach()     # clamp, gradient straight through
    skip = (power > 0) | (alpha < ALPHA_MIN)
    a_eff = torch.where(skip, torch.zeros_like(alpha), alpha)
    om = 1.0 - a_eff
    Tcum = torch.cumprod(om, dim=1)
    stopped = torch.cumsum((Tcum < T_STOP).to(torch.int32), dim=1) > 0       # this entry and all later ones
    live = ~stopped
    T_before = torch.cat([torch.ones_like(Tcum[:, :1]), Tcum[:, :-1]], dim=1)
    wgt = torch.where(live, a_eff * T_before, torch.zeros_like(a_eff))
    T_final = torch.prod(torch.where(live, om, torch.ones_like(om)), dim=1)
    color = wgt @ col + T_final[:, None] * bg[None, :]
    depth = wgt @ dep.detach()
    return color, depth, T_final


def rasterize(means3D, means2D, opacities, colors_precomp, scales, rotations, cam,
              cov3D_precomp=None, radius_rule: str = "3sigma", dtype=None,
              tile_rows: Optional[Tuple[int, int]] = None, return_aux: bool = False, tile_row_list=None):
    """Differentiable restatement of `GaussianRasterizer(raster_settings)(...)`.

    Returns (color [C,H,W], radii [N] int32, depth [1,H,W]).  `tile_rows=(r0,r1)` composites only
    that band of 16-pixel tile rows (rest of the image stays at bg / 0): used for the bounded CPU
    baseline and the tile-row partition tests; `tile_row_list=[r, ...]` does the same for scattered rows
    (the full-size configuration tests check a few rows of a frame that is too large to composite on the CPU).
    """
    dt = dtype or means3D.dtype
    cast = lambda x: None if x is None else x.to(dt)
    means3D, means2D, opacities, colors_precomp = map(cast, (means3D, means2D, opacities, colors_precomp))
    scales, rotations, cov3D_precomp = map(cast, (scales, rotations, cov3D_precomp))
    H, W = int(cam.image_height), int(cam.image_width)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    sp = preprocess(means3D, means2D, opacities, scales, rotations, cam, radius_rule, cov3D_precomp)
    sorted_gid, offs = build_tile_lists(sp, gx, gy)
    C = colors_precomp.shape[1]
    bg = cam.bg.to(dt).reshape(-1)[:C]
    op = opacities.reshape(-1)

    rows = []
    r0, r1 = (0, gy) if tile_rows is None else tile_rows
    col_img = bg[:, None, None].expand(C, H, W).clone()
    dep_img = torch.zeros(H, W, dtype=dt)
    T_img = torch.ones(H, W, dtype=dt)
    ar = torch.arange(TILE)
    n_eval = 0
    for ty in (range(r0, r1) if tile_row_list is None else tile_row_list):
        y0, y1 = ty * TILE, min((ty + 1) * TILE, H)
        for tx in range(gx):
            x0, x1 = tx * TILE, min((tx + 1) * TILE, W)
            s, e = int(offs[ty * gx + tx]), int(offs[ty * gx + tx + 1])
            if e == s:
                continue
            ids = sorted_gid[s:e]
            yy, xx = torch.meshgrid(ar[: y1 - y0] + y0, ar[: x1 - x0] + x0, indexing="ij")
            px, py = xx.reshape(-1).to(dt), yy.reshape(-1).to(dt)
            n_eval += px.numel() * (e - s)
            c, d, T = composite_tile(px, py, sp.xy[ids], sp.conic[ids], op[ids],
                                     colors_precomp[ids], sp.depth[ids], bg)
            col_img[:, y0:y1, x0:x1] = c.t().reshape(C, y1 - y0, x1 - x0)
            dep_img[y0:y1, x0:x1] = d.reshape(y1 - y0, x1 - x0)
            T_img[y0:y1, x0:x1] = T.detach().reshape(y1 - y0, x1 - x0)
    out = (col_img, sp.radii, dep_img.unsqueeze(0).detach())
    if return_aux:
        return out + ({"T_final": T_img, "tile_offsets": offs, "sorted_gid": sorted_gid,
                       "splats": sp, "pair_evals": n_eval},)
    return out


# ---------------------------------------------------------------------------------------------
# Synthetic view-tied scene generator shared by tests, smoke() and bench.py.
# ---------------------------------------------------------------------------------------------
def view_tied_scene(n: int, width: int, height: int, seed: int = 0, z_range=(1.0, 6.0)):
    """One isotropic Gaussian per pixel centre back-projected from a smoothed random depth map, plus
    half-scale Gaussians at random sub-pixel positions until `n` is reached (models the 2x edge
    densification, src/vtgaussian_slam.py:300-340).  Scale = depth / f (src/vtgaussian_slam.py:105-110).
    Returns a dict of float32 CPU tensors shaped like the operator inputs, and the camera record."""
    g = torch.Generator().manual_seed(seed)
    fx = fy = width / 2.0
    cx, cy = width / 2.0 - 0.5, height / 2.0 - 0.5
    k = [[fx, 0, cx], [0, fy, cy], [0, 0, 1]]
    cam = setup_camera(width, height, k, torch.eye(4))
    z = torch.rand(1, 1, height, width, generator=g) * (z_range[1] - z_range[0]) + z_range[0]
    z = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(z, (4, 4, 4, 4), mode="replicate"), 9, 1)[0, 0]
    P = width * height
    n_pix = min(n, P)
    if n_pix < P:
        pick = torch.randperm(P, generator=g)[:n_pix].sort().values
    else:
        pick = torch.arange(P)
    ys, xs = (pick // width).float(), (pick % width).float()
    zz = z.reshape(-1)[pick]
    scale = zz / fx
    n_extra = n - n_pix
    if n_extra > 0:
        ex = torch.rand(n_extra, generator=g) * (width - 1)
        ey = torch.rand(n_extra, generator=g) * (height - 1)
        ez = z[ey.round().long(), ex.round().long()] * (1.0 + 0.01 * (torch.rand(n_extra, generator=g) - 0.5))
        xs, ys, zz = torch.cat([xs, ex]), torch.cat([ys, ey]), torch.cat([zz, ez])
        scale = torch.cat([scale, 0.5 * ez / fx])
    means3D = torch.stack([(xs - cx + 0.5) / fx * zz, (ys - cy + 0.5) / fy * zz, zz], dim=-1)   # u == xs exactly
    scene = {
        "means3D": means3D.contiguous(),
        "means2D": torch.zeros(n, 3),
        "opacities": torch.sigmoid(torch.rand(n, 1, generator=g) * 4.0 - 2.0),
        "colors_precomp": torch.rand(n, 3, generator=g),
        "scales": scale[:, None].repeat(1, 3).contiguous(),
        "rotations": torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(n, 1),
    }
    return scene, cam


def random_scene(n: int, width: int, height: int, seed: int = 0, anisotropic: bool = True,
                 w2c: Optional[torch.Tensor] = None, fov_scale: float = 1.0):
    """General (anisotropic, arbitrary orientation, off-screen and behind-camera members) test scene."""
    g = torch.Generator().manual_seed(seed)
    fx = fy = fov_scale * width / 2.0
    cx, cy = width / 2.0 - 0.5 + 1.7, height / 2.0 - 0.5 - 2.3
    k = [[fx, 0, cx], [0, fy, cy], [0, 0, 1]]
    w2c = torch.eye(4) if w2c is None else w2c
    cam = setup_camera(width, height, k, w2c)
    z = torch.rand(n, generator=g) * 6.0 - 0.5                    # some behind / inside the near cull
    x = (torch.rand(n, generator=g) * 2.6 - 1.3) * z.abs() * cam.tanfovx
    y = (torch.rand(n, generator=g) * 2.6 - 1.3) * z.abs() * cam.tanfovy
    pc = torch.stack([x, y, z, torch.ones(n)], dim=-1)
    means3D = (torch.inverse(w2c) @ pc.t()).t()[:, :3].contiguous()
    base = z.abs().clamp(min=0.3) / fx * torch.exp(torch.rand(n, generator=g) * 3.0 - 0.5)
    if anisotropic:
        scales = base[:, None] * torch.exp(torch.rand(n, 3, generator=g) * 1.6 - 0.8)
        rot = torch.nn.functional.normalize(torch.randn(n, 4, generator=g))
    else:
        scales = base[:, None].repeat(1, 3)
        rot = torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(n, 1)
    scene = {
        "means3D": means3D,
        "means2D": torch.zeros(n, 3),
        "opacities": torch.sigmoid(torch.rand(n, 1, generator=g) * 8.0 - 4.0),
        "colors_precomp": torch.rand(n, 3, generator=g) * 2.0 - 0.5,
        "scales": scales.contiguous(),
        "rotations": rot.contiguous(),
    }
    return scene, cam
