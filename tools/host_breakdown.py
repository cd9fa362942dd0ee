"""Where the host time of one forward+backward goes (tiny scene: the GPU work is negligible), layer by layer."""
import ctypes, os, sys, time, torch
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
N = 1000


def timed(fn, n=N):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


rast = dgr.GaussianRasterizer(raster_settings=st)
camobj = dgr._camera_for(st, dev, 0, None)
det = {k: v.detach() for k, v in leaves.items()}
args = (camobj, det["means3D"], det["colors_precomp"], det["opacities"], det["scales"], det["rotations"])
print(f"_run_forward (allocations + vtgs_forward incl. the wait for the record): {timed(lambda: dgr._run_forward(*args)):.0f} us")
def fwd_nograd():
    with torch.no_grad():
        rast(**leaves)
print(f"module call, no grad:                                                   {timed(fwd_nograd):.0f} us")
print(f"module call, grad mode (autograd node built), no backward:              {timed(lambda: rast(**leaves)):.0f} us")
def new_module():
    dgr.GaussianRasterizer(raster_settings=st)
print(f"GaussianRasterizer(...) construction:                                    {timed(new_module):.1f} us")
c, r, d, fs = None, None, None, None
def fb():
    for t in leaves.values(): t.grad = None
    c, r, d = rast(**leaves)
    c.backward(g)
print(f"forward + backward:                                                      {timed(fb):.0f} us")
color, radii, depth, fs = dgr._run_forward(*args)
bargs = (fs, det["means3D"], det["colors_precomp"], det["opacities"], det["scales"], det["rotations"], color, g)
print(f"_run_backward (allocations + vtgs_backward, no wait):                    {timed(lambda: dgr._run_backward(*bargs)):.0f} us")
x = torch.zeros(8, device=dev, requires_grad=True)
class Id(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a): return a * 1
    @staticmethod
    def backward(ctx, ga): return ga
def trivial():
    x.grad = None
    Id.apply(x).sum().backward()
print(f"a trivial autograd.Function forward + backward (framework floor):        {timed(trivial):.0f} us")
