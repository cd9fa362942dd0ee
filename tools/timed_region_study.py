"""Why is bench.py's host-clocked mean (ms_per_step) larger than the median of its event-timed steps?  One timed region exactly as
bench.py runs it (fence, K steps, fence) with a hipEvent after every step and host time stamps around every call."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'vtgaussian-slam_amd'), os.path.join(ROOT, 'tests')]
from oracle import gs_oracle as go
from parity_util import to_settings
import diff_gaussian_rasterization as dgr
dev = torch.device('cuda:0')
N, W, H = 1000000, 1200, 680
K = int(os.environ.get('K', '50'))
scene, cam = go.view_tied_scene(N, W, H, seed=0)
leaves = {k: v.to(dev).requires_grad_(True) for k, v in scene.items()}
rast = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev))
g = (torch.rand(3, H, W) * 2 - 1).to(dev)
def step():
    for t in leaves.values():
        t.grad = None
    c, r, d = rast(**leaves)
    c.backward(g)
    return rast._last_state.pending is not None
W = int(os.environ.get('W', '10'))
if 'RAMP_MS' in os.environ:
    t_r = time.perf_counter()
    while (time.perf_counter() - t_r) * 1e3 < float(os.environ['RAMP_MS']): step()
import gc
if os.environ.get('NOGC') == '1': gc.disable()
for rnd in range(4):
    for _ in range(W): step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record()
    ahead = 0
    for i in range(K):
        h0 = time.perf_counter()
        ahead += step()
        ev[i + 1].record()
        host.append((time.perf_counter() - h0) * 1e3)
    t_enq = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
    print(f'round {rnd}: host-clocked {(t1 - t0) / K * 1e3:.4f} ms/step; events total {sum(d) / K:.4f} ms/step, first step {d[0]:.3f}, '
          f'median {sorted(d)[K // 2]:.4f}, max {max(d):.3f}; host per call median {sorted(host)[K // 2]:.3f} max {max(host):.3f} ms; '
          f'enqueue done {(t_enq - t0) * 1e3:.2f} ms of {(t1 - t0) * 1e3:.2f}; run-ahead forwards {ahead}/{K}', flush=True)
    print('   all event steps', ' '.join(f'{x:.3f}' for x in d), flush=True)
    print('   all host calls ', ' '.join(f'{x:.3f}' for x in host), flush=True)
    print('   first five event steps', [round(x, 3) for x in d[:5]], 'first five host calls', [round(x, 3) for x in host[:5]], flush=True)
