"""Shared comparison helpers for the GPU parity tests (HIP path vs oracle/gs_oracle.py)."""
import numpy as np
import torch

from oracle import gs_oracle as go

GRAD_KEYS = ["means3D", "means2D", "opacities", "colors_precomp", "scales", "rotations"]


def to_settings(cam, device, bg=None):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=(cam.bg if bg is None else bg).to(device), scale_modifier=cam.scale_modifier,
        viewmatrix=cam.viewmatrix.to(device), projmatrix=cam.projmatrix.to(device), sh_degree=0,
        campos=cam.campos.to(device), prefiltered=False)


def run_oracle(scene, cam, grad_color=None, dtype=torch.float64, radius_rule="3sigma", tile_rows=None):
    leaves = {k: v.detach().to(dtype).clone().requires_grad_(grad_color is not None) for k, v in scene.items()}
    color, radii, depth, aux = go.rasterize(cam=cam, radius_rule=radius_rule, tile_rows=tile_rows, return_aux=True, **leaves)
    grads = None
    if grad_color is not None:
        (color * grad_color.to(dtype)).sum().backward()
        grads = {k: leaves[k].grad for k in GRAD_KEYS}
    return color.detach(), radii, depth.detach(), grads, aux


def run_hip(scene, cam, device, grad_color=None, radius_rule=None, tile_rows=None, bg=None):
    from diff_gaussian_rasterization import GaussianRasterizer
    leaves = {k: v.detach().to(device=device, dtype=torch.float32).clone().requires_grad_(grad_color is not None)
              for k, v in scene.items()}
    rast = GaussianRasterizer(raster_settings=to_settings(cam, device, bg), radius_rule=radius_rule, tile_rows=tile_rows)
    color, radii, depth = rast(**leaves)
    grads = None
    if grad_color is not None:
        (color * grad_color.to(device)).sum().backward()
        grads = {k: leaves[k].grad.cpu() for k in GRAD_KEYS}
    return color.detach().cpu(), radii.cpu(), depth.detach().cpu(), grads


def image_error(ref, got):
    """max |diff| relative to the image's max magnitude, and the fraction of pixels above 1e-4 of it."""
    ref, got = ref.double(), got.double()
    scale = ref.abs().max().item() + 1e-12
    d = (ref - got).abs() / scale
    return d.max().item(), (d > 1e-4).double().mean().item()


def grad_error(ref, got):
    """(max |diff| / max |ref|,  99.9th percentile of element-wise relative error with a 1e-3*max floor)."""
    ref, got = ref.double(), got.double()
    scale = ref.abs().max().item()
    if scale == 0:
        return got.abs().max().item(), 0.0
    d = (ref - got).abs()
    rel = d / (ref.abs() + 1e-3 * scale)
    return (d.max() / scale).item(), torch.quantile(rel.reshape(-1)[:4_000_000], 0.999).item()


# ---------------------------------------------------------------------------------------------------------------
# Full-size configurations: the oracle composites a few 16-pixel tile rows of a frame that is too large to render on
# the CPU, on the sub-scene of Gaussians whose tile rectangle meets those rows.
# ---------------------------------------------------------------------------------------------------------------
def oracle_rows(scene, cam, rows, grad_color=None, dtype=torch.float64, radius_rule="3sigma"):
    """Oracle render of the 16-pixel tile rows `rows` only.  A no-grad pass of the oracle's own `preprocess` over all N
    Gaussians gives the radii and the tile rectangles; the differentiable oracle then runs on the Gaussians whose
    rectangle meets one of the rows (their relative order -- the depth tie-break -- is unchanged by the selection).
    Returns (color [C,H,W] with only those rows filled, radii [N] of the full scene, depth, grads scattered back to
    N rows (zero elsewhere), keep [N] bool, aux of the sub-scene render, index of the sub-scene in the full scene)."""
    n = scene["means3D"].shape[0]
    with torch.no_grad():
        full = {k: v.detach().to(dtype) for k, v in scene.items()}
        sp = go.preprocess(full["means3D"], full["means2D"], full["opacities"], full["scales"], full["rotations"], cam,
                           radius_rule)
    keep = torch.zeros(n, dtype=torch.bool)
    for r in rows:
        keep |= (sp.rect[:, 1] <= r) & (r < sp.rect[:, 3])
    keep &= sp.visible
    idx = torch.nonzero(keep).reshape(-1)
    leaves = {k: v[idx].detach().to(dtype).clone().requires_grad_(grad_color is not None) for k, v in scene.items()}
    color, _, depth, aux = go.rasterize(cam=cam, radius_rule=radius_rule, tile_row_list=list(rows), return_aux=True, **leaves)
    grads = None
    if grad_color is not None:
        (color * grad_color.to(dtype)).sum().backward()
        grads = {}
        for k in GRAD_KEYS:
            g = torch.zeros((n,) + tuple(leaves[k].shape[1:]), dtype=dtype)
            g[idx] = leaves[k].grad
            grads[k] = g
    return color.detach(), sp.radii, depth.detach(), grads, keep, aux, idx


def rows_mask(cam, rows):
    """[H] bool: pixel rows covered by the 16-pixel tile rows `rows`."""
    m = torch.zeros(int(cam.image_height), dtype=torch.bool)
    for r in rows:
        m[r * 16: min((r + 1) * 16, int(cam.image_height))] = True
    return m


# ---------------------------------------------------------------------------------------------------------------
# Index parity: which (Gaussian, 8x8 tile) instances the composite may see.  The kernel bins a splat into an 8x8 tile
# iff the tile lies under the splat's 16x16-tile rectangle AND the minimum of q = 1/2 (A dx^2 + C dy^2) + B dx dy over the
# tile's pixel centres is <= tau = ln(255 o), where the kernel evaluates tau with a documented conservative slack
# (tau (1 + 1e-4) + 1e-4, csrc/vtgs_binning.hip).  The float64 restatement below yields the STRICT set (no slack: every
# member has a pixel with alpha >= 1/255, so dropping it would change the image) and a LOOSE set (twice the slack);
# the kernel's lists must sit between the two, in exactly the oracle's (depth bits, index) order.
# ---------------------------------------------------------------------------------------------------------------
def _min_quadratic_over_rect64(A, B, C, u, v, px0, py0, px1, py1):
    dx0, dx1, dy0, dy1 = u - px1, u - px0, v - py1, v - py0
    inside = (dx0 <= 0) & (dx1 >= 0) & (dy0 <= 0) & (dy1 >= 0)
    q = lambda dx, dy: 0.5 * (A * dx * dx + C * dy * dy) + B * dx * dy
    best = torch.full_like(u, float("inf"))
    for dx in (dx0, dx1):                                    # vertical edges: minimise over dy
        dy = torch.minimum(dy1, torch.maximum(dy0, -B * dx / C))
        best = torch.minimum(best, q(dx, dy))
    for dy in (dy0, dy1):
        dx = torch.minimum(dx1, torch.maximum(dx0, -B * dy / A))
        best = torch.minimum(best, q(dx, dy))
    return torch.where(inside, torch.zeros_like(best), best)


def oracle_instances_8x8(sp, opacities, cam, rel_slack=0.0, abs_slack=0.0, rows8=None):
    """Sorted int64 keys (tile8 << 32 | gaussian) of the (Gaussian, 8x8 tile) instances under the predicate
    q_min <= tau (1 + rel_slack) + abs_slack.  `sp` = oracle Splats (float64), `rows8` = optional (begin, end) band of
    8-pixel tile rows."""
    H, W = int(cam.image_height), int(cam.image_width)
    gx8, gy8 = (W + 7) // 8, (H + 7) // 8
    op = opacities.reshape(-1).to(torch.float64)
    ok = sp.visible & (op * 255.0 >= 1.0)
    vis = torch.nonzero(ok).reshape(-1)
    if vis.numel() == 0:
        return torch.zeros(0, dtype=torch.long)
    r = sp.rect[vis]
    x0, x1 = 2 * r[:, 0], torch.clamp(2 * r[:, 2], max=gx8)
    y0, y1 = 2 * r[:, 1], torch.clamp(2 * r[:, 3], max=gy8)
    if rows8 is not None:
        y0, y1 = torch.clamp(y0, min=rows8[0]), torch.clamp(y1, max=rows8[1])
    w, h = torch.clamp(x1 - x0, min=0), torch.clamp(y1 - y0, min=0)
    cnt = w * h
    owner = torch.repeat_interleave(torch.arange(vis.numel()), cnt)
    first = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(int(cnt.sum())) - first[owner]
    ty = y0[owner] + local // w[owner]
    tx = x0[owner] + local % w[owner]
    gid = vis[owner]
    px0, py0 = (tx * 8).to(torch.float64), (ty * 8).to(torch.float64)
    px1 = torch.clamp(px0 + 7, max=W - 1)
    py1 = torch.clamp(py0 + 7, max=H - 1)
    con, xy = sp.conic.detach().to(torch.float64)[gid], sp.xy.detach().to(torch.float64)[gid]
    qmin = _min_quadratic_over_rect64(con[:, 0], con[:, 1], con[:, 2], xy[:, 0], xy[:, 1], px0, py0, px1, py1)
    tau = torch.log(255.0 * op[gid])
    hit = qmin <= tau * (1.0 + rel_slack) + abs_slack
    keys = ((ty * gx8 + tx) << 32) | gid
    return torch.sort(keys[hit]).values


# ---------------------------------------------------------------------------------------------------------------
# Outlier audit: a pixel whose HIP colour differs from the float64 oracle by more than the 1e-4 tolerance must sit on a
# discrete decision of the composite -- a pair whose alpha is within float32 rounding of 1/255 (counted or skipped), or a
# transmittance within rounding of the 1e-4 stop -- as seen in the ORACLE's own per-pair values.  Anything else fails.
# ---------------------------------------------------------------------------------------------------------------
# |ln alpha - ln(1/255)| a float32 implementation cannot resolve: 5e-5 for the exponent arithmetic itself (DESIGN.md 2,
# deviation 3) plus the float32 representation of the splat centre -- u, v are O(W) pixels, so they carry an absolute
# error of a few half-ulps of max(W, H) (6e-5 px at 1024, the published CUDA operator has the same), which moves the
# exponent by |grad_centre q| times that: up to ~3 per pixel at the rim of a sigma ~ 1 px splat.
LN_ALPHA_MARGIN = 5e-5
CENTRE_HALF_ULPS = 4.0
T_STOP_MARGIN = 1e-3        # |T (1 - alpha) / 1e-4 - 1|: a product of up to ~100 float32 factors, each with the above


def audit_outliers(ref, got, aux, opacities, cam, tol=1e-4, max_report=5000):
    """ref/got: [C,H,W] images (any number of channels).  aux: oracle aux (float64 render) whose `splats`, `sorted_gid`,
    `tile_offsets` describe the scene `opacities` belongs to.  Returns dict(outliers, explained, unexplained [(y,x)...],
    max_rel, tiles) -- `tiles` = set of 16x16 tile ids that hold an outlier pixel."""
    ref, got = ref.double(), got.double()
    scale = ref.abs().max().item() + 1e-12
    err = ((ref - got).abs() / scale).amax(dim=0)
    ys, xs = torch.nonzero(err > tol, as_tuple=True)
    out = {"outliers": int(ys.numel()), "explained": 0, "unexplained": [], "max_rel": float(err.max()), "tiles": set(),
           "frac": float(ys.numel()) / err.numel()}
    if ys.numel() > max_report:
        out["unexplained"] = [("too many outliers", int(ys.numel()))]
        return out
    sp, offs, sg = aux["splats"], aux["tile_offsets"], aux["sorted_gid"]
    gx = (int(cam.image_width) + 15) // 16
    op_all = opacities.reshape(-1).double()
    ln_min = float(torch.log(torch.tensor(go.ALPHA_MIN, dtype=torch.float64)))
    delta = CENTRE_HALF_ULPS * 2.0 ** -24 * max(int(cam.image_width), int(cam.image_height))
    for y, x in zip(ys.tolist(), xs.tolist()):
        t = (y // 16) * gx + x // 16
        out["tiles"].add(t)
        ids = sg[int(offs[t]): int(offs[t + 1])]
        xy, con, op = sp.xy.detach().double()[ids], sp.conic.detach().double()[ids], op_all[ids]
        dx, dy = xy[:, 0] - x, xy[:, 1] - y
        power = -0.5 * (con[:, 0] * dx * dx + con[:, 2] * dy * dy) - con[:, 1] * dx * dy
        a_raw = op * torch.exp(torch.clamp(power, max=0.0))
        alpha = torch.clamp(a_raw, max=go.ALPHA_MAX)
        valid = (power <= 0) & (alpha >= go.ALPHA_MIN)
        a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
        Tcum = torch.cumprod(1.0 - a_eff, 0)
        stopped = torch.cumsum((Tcum < go.T_STOP).int(), 0) > 0
        reach = ~torch.cat([torch.zeros(1, dtype=torch.bool), stopped[:-1]])      # pairs the pixel still looks at
        ln_a = torch.log(torch.clamp(a_raw, min=1e-300))
        gq = torch.sqrt((con[:, 0] * dx + con[:, 1] * dy) ** 2 + (con[:, 2] * dy + con[:, 1] * dx) ** 2)   # |d power / d centre|
        m_alpha = ((ln_a - ln_min).abs() - gq * delta)[reach & (power <= 0)]
        m_T = (Tcum / go.T_STOP - 1.0).abs()[reach & valid]
        ok = (m_alpha.numel() and float(m_alpha.min()) <= LN_ALPHA_MARGIN) or (m_T.numel() and float(m_T.min()) <= T_STOP_MARGIN)
        if ok:
            out["explained"] += 1
        else:
            out["unexplained"].append((y, x, float(err[y, x]), float(m_alpha.min()) if m_alpha.numel() else None,
                                       float(m_T.min()) if m_T.numel() else None))
    return out


def tiles_of(aux, sel, cam):
    """16x16 tile ids under the oracle rectangles of the selected Gaussians."""
    gx = (int(cam.image_width) + 15) // 16
    out = set()
    for i in torch.nonzero(sel).reshape(-1).tolist():
        x0, y0, x1, y1 = aux["splats"].rect[i].tolist()
        out.update(ty * gx + tx for ty in range(y0, y1) for tx in range(x0, x1))
    return out


def tainted_gaussians(aux, tiles, n):
    """[n] bool: Gaussians in the 16x16 tile list of a tile that holds an audited outlier pixel (their gradients carry
    that pixel's flipped decision)."""
    m = torch.zeros(n, dtype=torch.bool)
    offs, sg = aux["tile_offsets"], aux["sorted_gid"]
    for t in tiles:
        m[sg[int(offs[t]): int(offs[t + 1])]] = True
    return m
