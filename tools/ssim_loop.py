"""A few forward + backward passes of the SSIM kernels at 1200x680 (the mapping loss's two launches), for rocprofv3 passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'vtgaussian-slam_amd')]
from diff_gaussian_rasterization import losses as L
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
H, W = 680, 1200
a = torch.rand(3, H, W, generator=g).to(dev).requires_grad_(True)
b = (a.detach() + 0.05 * torch.randn(3, H, W, generator=g).to(dev)).clamp(0, 1)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    a.grad = None
    v = L.fused_ssim(a, b)
    v.backward()
torch.cuda.synchronize()
print('ssim', float(v), 'grad abs sum', float(a.grad.abs().sum()))
