"""Which gradient carries the p999 of bench.py's parity block?  Same frame, same audited rows, per input tensor:
max |d| / max |ref|, the 99.9th percentile of |d| / (|ref| + 1e-3 max |ref|), and where in |ref| the large ratios sit."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vtgaussian-slam_amd"), os.path.join(ROOT, "tests")]
from oracle import gs_oracle as go
from parity_util import HIP_CENTRE_ERR_PX, GRAD_KEYS, audit_outliers, oracle_rows, rows_mask, tainted_gaussians, to_settings
import diff_gaussian_rasterization as dgr
dev = torch.device("cuda:0")
N, W, H = 1_000_000, 1200, 680
rows = [4, 19, 20, 21, 22, 37]
scene, cam = go.view_tied_scene(N, W, H, seed=0)
g = torch.Generator().manual_seed(1)
grad_color = torch.rand(3, H, W, generator=g) * 2 - 1
mask = rows_mask(cam, rows)
gsel = grad_color.clone(); gsel[:, ~mask] = 0
torch.set_num_threads(16)
ref_c, ref_r, ref_d, ref_g, keep, aux, idx = oracle_rows(scene, cam, rows, gsel)
leaves = {k: v.to(dev).requires_grad_(True) for k, v in scene.items()}
rast = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev))
color, radii, depth = rast(**leaves)
color.backward(gsel.to(dev))
got_c, got_d = color.detach().cpu().double(), depth.detach().cpu().double()
hc = torch.where(mask[None, :, None], got_c, ref_c); hd = torch.where(mask[None, :, None], got_d, ref_d)
sub_op = scene["opacities"][idx]
a_c = audit_outliers(ref_c, hc, aux, sub_op, cam, 1e-4, centre_err_px=HIP_CENTRE_ERR_PX); a_d = audit_outliers(ref_d, hd, aux, sub_op, cam, 1e-4, centre_err_px=HIP_CENTRE_ERR_PX)
taint = tainted_gaussians(aux, a_c["tiles"] | a_d["tiles"], idx.numel())
tf = torch.zeros(N, dtype=torch.bool); tf[idx[taint]] = True; tf |= (ref_r != radii.cpu())
gy16 = (H + 15) // 16
rect = aux["splats"].rect
rowset = torch.zeros(gy16 + 1, dtype=torch.bool); rowset[rows] = True
cov = torch.cumsum(rowset.long(), 0)
inner_sub = (cov[(rect[:, 3] - 1).clamp(0, gy16)] - cov[rect[:, 1].clamp(0, gy16)] + rowset[rect[:, 1].clamp(0, gy16)].long()
             == (rect[:, 3] - rect[:, 1])) & (rect[:, 3] > rect[:, 1])
inner = torch.zeros(N, dtype=torch.bool); inner[idx[inner_sub]] = True
sel = keep & ~tf & inner
print("gaussians compared:", int(sel.sum()))
for k in GRAD_KEYS:
    r, h = ref_g[k][sel].double(), leaves[k].grad.cpu()[sel].double()
    scale = r.abs().max().item()
    if scale == 0:
        print(k, "all zero in the oracle; max |got| =", h.abs().max().item()); continue
    d = (r - h).abs()
    rel = d / (r.abs() + 1e-3 * scale)
    q = torch.quantile(rel.reshape(-1)[:4_000_000], torch.tensor([0.5, 0.99, 0.999, 0.9999], dtype=torch.float64))
    big = rel > 1e-3
    print(f"{k:15s} max_rel {d.max().item() / scale:.2e}  p50 {q[0]:.1e} p99 {q[1]:.1e} p999 {q[2]:.1e} p9999 {q[3]:.1e}  "
          f"elements > 1e-3: {int(big.sum())} of {rel.numel()}; their |ref|/max median {(r.abs()[big] / scale).median().item() if big.any() else 0:.1e}, "
          f"their |d|/max median {(d[big] / scale).median().item() if big.any() else 0:.1e}")
    if k in ("means3D", "scales", "means2D"):
        for c in range(r.shape[1]):
            rc, dc = r[:, c], d[:, c]
            relc = dc / (rc.abs() + 1e-3 * scale)
            print(f"    column {c}: max|ref| {rc.abs().max().item():.2e}  p999 {torch.quantile(relc, 0.999).item():.1e}")
