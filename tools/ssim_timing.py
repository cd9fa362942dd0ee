"""Time the SSIM kernels alone (1200x680, 3 channels): forward with gradient maps + backward, HIP events, 100 iterations."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vtgaussian-slam_amd")]
from diff_gaussian_rasterization import losses
dev = torch.device("cuda:0")
H, W = int(os.environ.get("ABL_H", 680)), int(os.environ.get("ABL_W", 1200))
a = torch.rand(3, H, W, device=dev, requires_grad=True)
b = torch.rand(3, H, W, device=dev)
for _ in range(10):
    a.grad = None
    losses.fused_ssim(a, b).backward()
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
tf = tb = 0.0
for _ in range(100):
    a.grad = None
    e0.record(); s = losses.fused_ssim(a, b); e1.record(); s.backward(); e2.record()
    torch.cuda.synchronize()
    tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
print(os.environ.get("ABL_TAG", ""), f"ssim forward (+ the mean's reduction) {tf * 10:.1f} us, backward {tb * 10:.1f} us")
