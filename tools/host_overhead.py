"""Host-side floor of one forward+backward through the Python layer (tiny scene: the GPU work is negligible)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vtgaussian-slam_amd"), os.path.join(ROOT, "tests")]
from oracle import gs_oracle as go
from parity_util import to_settings
import diff_gaussian_rasterization as dgr
dev = torch.device("cuda:0")
scene, cam = go.view_tied_scene(2000, 64, 48, seed=0)
leaves = {k: v.to(dev).requires_grad_(True) for k, v in scene.items()}
st = to_settings(cam, dev)
g = torch.rand(3, 48, 64, device=dev)
def step():
    for t in leaves.values(): t.grad = None
    c, r, d = dgr.GaussianRasterizer(raster_settings=st)(**leaves)
    c.backward(g)
for _ in range(50): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
print(f"host floor per forward+backward: {dt * 1e6:.0f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
