"""Queue steps per tile of the quadrant-queue forward at the headline shape (device counter, VTGS_COUNT_STEPS).

    python tools/forward_steps.py        # env ABL_N, ABL_W, ABL_H as tools/kernel_timing.py
"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'vtgaussian-slam_amd'), os.path.join(ROOT, 'tests')]
from oracle import gs_oracle as go
from parity_util import to_settings
import diff_gaussian_rasterization as dgr
dev = torch.device('cuda:0')
N = int(os.environ.get('ABL_N', '1000000')); W = int(os.environ.get('ABL_W', '1200')); H = int(os.environ.get('ABL_H', '680'))
scene, cam = go.view_tied_scene(N, W, H, seed=0)
dgr.set_option("VTGS_COUNT_STEPS", 1)
rast = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev))
with torch.no_grad():
    rast(**{k: v.to(dev) for k, v in scene.items()})
steps = dgr.debug_forward_steps(rast)
info = dgr.last_forward_info()
tiles = ((W + 7) // 8) * ((H + 7) // 8)
print({"steps": steps, "tiles8": tiles, "steps_per_tile": round(steps / tiles, 3), "instances": info["instances"],
       "batches_of_16_per_tile": round(info["instances"] / 16 / tiles, 3), "pairs_forward_G": round(steps * 1024 / 1e9, 4),
       "pairs_if_batched_G": round(info["instances"] * 64 / 1e9, 4)})
