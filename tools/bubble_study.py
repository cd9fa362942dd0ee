"""The fwd+bwd loop of bench.py (gradients reset every step), 70 steps, for kernel traces of the queue bubble at the step
boundary.  MODE = base | onethread (backward on the calling thread) | stream (a non-default torch stream) | both"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'vtgaussian-slam_amd'), os.path.join(ROOT, 'tests')]
from oracle import gs_oracle as go
from parity_util import to_settings
import diff_gaussian_rasterization as dgr
mode = os.environ.get('MODE', 'base')
dev = torch.device('cuda:0')
N, W, H = 1000000, 1200, 680
scene, cam = go.view_tied_scene(N, W, H, seed=0)
leaves = {k: v.to(dev).requires_grad_(True) for k, v in scene.items()}
rast = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev))
g = torch.rand(3, H, W, device=dev)
if mode in ('onethread', 'both'):
    torch.autograd.set_multithreading_enabled(False)
def step():
    for t in leaves.values():
        t.grad = None
    c, r, d = rast(**leaves)
    if mode != 'fwdonly':
        c.backward(g)
def loop():
    for it in range(10): step()
    torch.cuda.synchronize(); t = time.time()
    for it in range(60): step()
    torch.cuda.synchronize()
    print(mode, 'step %.4f ms' % ((time.time() - t) / 60 * 1e3), flush=True)
if mode in ('stream', 'both'):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        loop()
else:
    loop()
