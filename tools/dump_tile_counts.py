"""Writes the per-8x8-tile list lengths of the headline scene to gpurun_out/tile_counts.npy (scheduling studies)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vtgaussian-slam_amd"), os.path.join(ROOT, "tests")]
from oracle import gs_oracle as go
from parity_util import to_settings
import diff_gaussian_rasterization as dgr
dev = torch.device("cuda:0")
scene, cam = go.view_tied_scene(1_000_000, 1200, 680, seed=0)
leaves = {k: v.to(dev) for k, v in scene.items()}
rast = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev))
with torch.no_grad():
    rast(**leaves)
offs, gid, geom = dgr.debug_tile_lists(rast)
cnt = (offs[1:] - offs[:-1]).numpy()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "tile_counts.npy"), cnt)
print("tiles", cnt.shape, "mean", cnt.mean(), "max", cnt.max(), "min", cnt.min())
