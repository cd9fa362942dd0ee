"""The statistics behind tests/test_gpu_parity.py::test_scalar_and_matrix_core_kernels_agree, printed (which key, how far from the
tolerance) -- to tell a real disagreement from a statistic that sits on its threshold.   VTGS_LIBRARY=... python tools/xcheck_stat.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'vtgaussian-slam_amd'), os.path.join(ROOT, 'tests')]
from oracle import gs_oracle as go
import test_gpu_parity as T
from parity_util import GRAD_KEYS, grad_error, to_settings
dev = torch.device('cuda:0')
scene, cam = go.random_scene(6000, 200, 136, seed=41, anisotropic=True, w2c=T._w2c(41))
g = torch.Generator().manual_seed(8)
grad_color = torch.rand(3, 136, 200, generator=g) * 2 - 1
T._opt("VTGS_FWD_IMPL", "1"); T._opt("VTGS_BWD_IMPL", "1")
c1, r1, d1, g1 = T.run_hip(scene, cam, dev, grad_color)
T._opt("VTGS_FWD_IMPL", "0"); T._opt("VTGS_BWD_IMPL", "0")
c0, r0, d0, g0 = T.run_hip(scene, cam, dev, grad_color)
for k in GRAD_KEYS:
    mx, p999 = grad_error(g0[k].double(), g1[k])
    print(k, 'max %.3e p99.9 %.3e' % (mx, p999))
import diff_gaussian_rasterization as dgr
print(dgr.last_forward_info())
# is it a decision flip?  pixels where the two forwards differ by more than 1e-4 of the image maximum, and the statistics without
# the Gaussians of their 8x8 tiles
sc = c0.abs().max().item()
bad = ((c0 - c1).abs().amax(0) > 1e-4 * sc) | ((d0 - d1).abs().squeeze(0) > 1e-4 * d0.abs().max().item())
ys, xs = torch.nonzero(bad, as_tuple=True)
print('pixels above 1e-4:', int(bad.sum()), 'max colour diff %.3e' % ((c0 - c1).abs().max().item() / sc), list(zip(ys.tolist(), xs.tolist()))[:8])
rast = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev))
with torch.no_grad():
    rast(**{k: v.to(dev) for k, v in scene.items()})
offs, gid, _ = dgr.debug_tile_lists(rast)
gx8 = (cam.image_width + 7) // 8
taint = torch.zeros(scene["means3D"].shape[0], dtype=torch.bool)
for y, x in zip(ys.tolist(), xs.tolist()):
    t = (y // 8) * gx8 + x // 8
    taint[gid[offs[t]:offs[t + 1]].long().cpu()] = True
print('tainted Gaussians:', int(taint.sum()))
for k in GRAD_KEYS:
    mx, p999 = grad_error(g0[k].double()[~taint], g1[k][~taint])
    print(k, 'clean: max %.3e p99.9 %.3e' % (mx, p999))
