"""Timing of the GPU point-to-plane check at the Replica frame size vs the CPU oracle (KD-tree) on the same frames."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vtgaussian-slam_amd"), os.path.join(ROOT, "tests")]
from diff_gaussian_rasterization import point2plane as p2p
from oracle import p2p_oracle as po
from test_point2plane import _frames
H, W = int(os.environ.get("P2P_H", 680)), int(os.environ.get("P2P_W", 1200))
d0, d1, k, w0, w1 = _frames(1, H, W, 600.0 * W / 1200)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
args = (t(d0)[None], t(d1)[None], t(k), t(w0), t(w1))
for _ in range(3): v = p2p.compute_point2plane_dist(*args, method="sum")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): v = p2p.compute_point2plane_dist(*args, method="sum")
torch.cuda.synchronize(); gpu_ms = (time.perf_counter() - t0) / 20 * 1e3
t0 = time.perf_counter(); ref = po.compute_point2plane_dist(d0, d1, k, w0, w1, method="sum"); cpu_s = time.perf_counter() - t0
print(f"point-to-plane {W}x{H}: GPU {gpu_ms:.3f} ms per call (sum = {float(v):.6f}); CPU oracle (scipy KD-tree) {cpu_s:.2f} s (sum = {ref:.6f})")
