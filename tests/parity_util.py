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
# The margins are BOUNDS derived from float32 rounding, not tuned values ("audit margins"):
#
# LN_ALPHA_MARGIN -- |ln alpha(float32) - ln alpha(exact)| from the exponent arithmetic alone.  The kernels form
#   log2 alpha = sum_m K_m Phi_m  (csrc/vtgs_composite_common.h::tile_coefficients + six MFMA steps): 7 roundings in K_0
#   (three products, three fma, the log2 of the opacity), 3 each in K_1, K_2, 6 in the MFMA chain -- 19 roundings, each at most
#   a relative half-ulp (2^-24) of a partial result that is itself at most M = the sum of the absolute values of the terms
#   (|q_a| s_x^2 + |q_b s_x s_y| + |q_c| s_y^2 + |log2 o| + |K_1 X| + |K_2 Y| + |q_a| X^2 + |q_b X Y| + |q_c| Y^2, s = splat centre -
#   tile centre, (X, Y) = pixel - tile centre).  For a pair near the threshold of a sigma ~ 1 px splat M stays below 64
#   (|log2(1/255)| = 8, the quadratic terms of a centre up to 7.5 px from the tile centre: < 56), which gives
#   19 x 2^-24 x 64 x ln 2 = 5.0e-5; the audit evaluates M for every pair and scales the margin by max(1, M / 64), so the
#   bound also holds for the sharp or distant splats of the anisotropic scenes.  (v_exp_f32 / v_log_f32: one ulp of the RESULT
#   each, 1.2e-7 relative -- nothing.  A float32 implementation that evaluates the quadratic form directly from the pixel offset
#   -- the CPU oracle in float32, the published CUDA operator -- makes ~8 roundings of smaller partials: inside the same bound.)
# centre error -- what the implementation's pixel-centre arithmetic loses, in pixels; it moves ln alpha by |d ln alpha / d centre|
#   (`gq` below, up to ~3 per pixel at the rim of a sigma ~ 1 px splat) times that.
#   * an implementation that forms the centre in float32 (the CPU oracle in float32, the published operator):
#     u = ((ndc + 1) W - 1) / 2 rounds at the divide, at ndc + 1, at the product (magnitude W) and again in u - pixel:
#     CENTRE_HALF_ULPS = 4 half-ulps of max(W, H), 4 x 2^-24 x max(W, H) px (2.9e-4 px at 1200).
#   * the HIP kernels since round 5: the centre travels as a float32 PAIR (error 2^-45 relative) and the only rounding left is
#     (u - tile centre) + lo at magnitude < 16 px: HIP_CENTRE_ERR_PX = 2^-20 px.
# T stop -- |T (1 - alpha) / 1e-4 - 1|, T a product of (1 - alpha_k): d ln(1 - alpha_k) = alpha_k / (1 - alpha_k) x d ln alpha_k, so the
#   relative error of the product at list position j is bounded PER PIXEL by
#       sum_{k <= j, counted} alpha_k / (1 - alpha_k) x (LN_ALPHA_MARGIN + gq_k x centre error)  +  2 j x 2^-24
#   (a clamped alpha = 0.99 carries no exponent error; the second term: one rounding each of 1 - alpha and of the running
#   product).  T_STOP_MARGIN = 1e-3 is what that bound comes to for a stack that reaches T = 1e-4 through alphas <= 0.5
#   (sum alpha / (1 - alpha) <= 2 sum alpha <= 2 x 9.2) and stays as the documented typical value; the audit uses the
#   per-pixel bound.
# tests/test_audit_margins.py (CPU): the float32 CPU oracle against the float64 one needs exactly these margins -- every outlier
# it produces is explained by them, and with the margins at zero the same outliers are NOT explained.
LN_ALPHA_MARGIN = 5e-5
CENTRE_HALF_ULPS = 4.0
HIP_CENTRE_ERR_PX = 2.0 ** -20
T_STOP_MARGIN = 1e-3


def float32_centre_err_px(cam) -> float:
    return CENTRE_HALF_ULPS * 2.0 ** -24 * max(int(cam.image_width), int(cam.image_height))


def audit_outliers(ref, got, aux, opacities, cam, tol=1e-4, max_report=5000, centre_err_px=None, ln_alpha_margin=LN_ALPHA_MARGIN):
    """ref/got: [C,H,W] images (any number of channels).  aux: oracle aux (float64 render) whose `splats`, `sorted_gid`,
    `tile_offsets` describe the scene `opacities` belongs to.  `centre_err_px`: what the audited implementation's centre
    arithmetic loses (default: a float32 centre, float32_centre_err_px; the HIP kernels: HIP_CENTRE_ERR_PX).  Returns
    dict(outliers, explained, unexplained [(y,x)...], max_rel, tiles) -- `tiles` = set of 16x16 tile ids that hold an outlier
    pixel."""
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
    delta = float32_centre_err_px(cam) if centre_err_px is None else float(centre_err_px)
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
        # M of every pair (comment above), in log2 units, around the centre of the pixel's 8x8 tile
        l2e = 1.4426950408889634
        sx, sy = xy[:, 0] - (8 * (x // 8) + 3.5), xy[:, 1] - (8 * (y // 8) + 3.5)
        X, Y = x - (8 * (x // 8) + 3.5), y - (8 * (y // 8) + 3.5)
        qa, qb, qc = 0.5 * l2e * con[:, 0].abs(), l2e * con[:, 1].abs(), 0.5 * l2e * con[:, 2].abs()
        M = (qa * sx * sx + qb * (sx * sy).abs() + qc * sy * sy + torch.log2(torch.clamp(op, min=1e-300)).abs()
             + (2 * qa * sx.abs() + qb * sy.abs()) * abs(X) + (2 * qc * sy.abs() + qb * sx.abs()) * abs(Y)
             + qa * X * X + qb * abs(X * Y) + qc * Y * Y)
        ln_m = ln_alpha_margin * torch.clamp(M / 64.0, min=1.0)
        m_alpha = ((ln_a - ln_min).abs() - gq * delta - ln_m)[reach & (power <= 0)]
        # the per-pixel bound on the relative error of T (1 - alpha) at every list position (comment above)
        d_ln = torch.where(valid & (a_raw < go.ALPHA_MAX), a_eff / (1.0 - a_eff) * (ln_m + gq * delta), torch.zeros_like(a_eff))
        t_bound = torch.cumsum(d_ln, 0) + 2.0 * torch.cumsum(valid.double(), 0) * 2.0 ** -24
        m_T = ((Tcum / go.T_STOP - 1.0).abs() - t_bound)[reach & valid]
        ok = (m_alpha.numel() and float(m_alpha.min()) <= 0.0) or (m_T.numel() and float(m_T.min()) <= 0.0)
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
