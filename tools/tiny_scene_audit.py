"""VERDICT r4 (weak 2): where does the 2.9e-3 (99.9th percentile = the maximum of 75 elements) of the 5x3-pixel scene come from?
Per input tensor: the worst element of the default kernels AND of the scalar cross-check kernels against the float64 oracle,
the Gaussian it belongs to and that Gaussian's projected shape."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vtgaussian-slam_amd"), os.path.join(ROOT, "tests")]
from oracle import gs_oracle as go
from parity_util import GRAD_KEYS, run_oracle, run_hip
import diff_gaussian_rasterization as dgr
dev = torch.device("cuda:0")
scene, cam = go.random_scene(25, 5, 3, seed=6, anisotropic=True)
g = torch.Generator().manual_seed(99)
grad_color = torch.rand(3, 3, 5, generator=g) * 2 - 1
ref_c, ref_r, ref_d, ref_g, aux = run_oracle(scene, cam, grad_color)
sp = aux["splats"]
for name, (fi, bi) in {"default (matrix-core) kernels": (-1, -1), "scalar kernels": (0, 0)}.items():
    dgr.set_option("VTGS_FWD_IMPL", fi); dgr.set_option("VTGS_BWD_IMPL", bi)
    got_c, got_r, got_d, got_g = run_hip(scene, cam, dev, grad_color)
    print(name, "image max |d|", float((ref_c.double() - got_c.double()).abs().max()))
    for k in GRAD_KEYS:
        a, b = ref_g[k].double(), got_g[k].double()
        sc = a.abs().max()
        rel = (a - b).abs() / (a.abs() + 1e-3 * sc)
        i = int(rel.reshape(-1).argmax()); gi = i // a.shape[1] if a.dim() > 1 else i
        con = sp.conic[gi].tolist(); det = con[0] * con[2] - con[1] ** 2
        print(f"  {k:15s} worst rel {float(rel.reshape(-1)[i]):.2e} |ref|/max {float(a.reshape(-1)[i].abs() / sc):.2e} |err|/max "
              f"{float((a - b).abs().reshape(-1)[i] / sc):.2e}  gaussian {gi} radius {int(ref_r[gi])} centre ({float(sp.xy[gi, 0]):.1f}, "
              f"{float(sp.xy[gi, 1]):.1f}) conic ({con[0]:.4f}, {con[1]:.4f}, {con[2]:.4f}) cond {max(con[0], con[2]) ** 2 / max(det, 1e-30):.1f}")
dgr.reset_options()
