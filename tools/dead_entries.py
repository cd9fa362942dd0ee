"""How many entries of the 8x8 tile lists contribute to NO pixel of their tile (alpha < 1/255 at all 64 pixel centres)?  The
binning's reach test is exact over the continuous rectangle of pixel centres, not over the 64 centres themselves; such entries
cost the composites a full share of every batch they sit in.  Headline scene, every 7th tile."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'vtgaussian-slam_amd'), os.path.join(ROOT, 'tests')]
from oracle import gs_oracle as go
from parity_util import to_settings
import diff_gaussian_rasterization as dgr
dev = torch.device('cuda:0')
N, W, H = 1000000, 1200, 680
scene, cam = go.view_tied_scene(N, W, H, seed=0)
rast = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev))
with torch.no_grad():
    rast(**{k: v.to(dev) for k, v in scene.items()})
offs, gid, geom = dgr.debug_tile_lists(rast)
gx8 = (W + 7) // 8
tot = dead = 0
valid_pairs = 0
hist = torch.zeros(65, dtype=torch.long)
for t in range(0, offs.numel() - 1, 7):
    s, e = int(offs[t]), int(offs[t + 1])
    if e == s:
        continue
    g = geom[gid[s:e]].double()
    ty, tx = t // gx8, t % gx8
    ys, xs = torch.meshgrid(torch.arange(8) + 8 * ty, torch.arange(8) + 8 * tx, indexing='ij')
    px, py = xs.reshape(-1).double(), ys.reshape(-1).double()
    dx = g[:, 0:1] - px[None, :]; dy = g[:, 1:2] - py[None, :]
    power = -0.5 * (g[:, 2:3] * dx * dx + g[:, 4:5] * dy * dy) - g[:, 3:4] * dx * dy
    alpha = torch.clamp(g[:, 5:6] * torch.exp(power), max=0.99)
    ok = (alpha >= 1.0 / 255.0) & (py[None, :] < H) & (px[None, :] < W)
    n_ok = ok.sum(1)
    tot += e - s; dead += int((n_ok == 0).sum()); valid_pairs += int(n_ok.sum())
    hist += torch.bincount(n_ok.clamp(max=64), minlength=65)
print(f'sampled entries {tot}: {dead} ({100.0 * dead / tot:.2f} %) reach no pixel centre; valid pairs per entry {valid_pairs / tot:.2f} of 64')
c = torch.cumsum(hist, 0).double() / hist.sum()
print('entries with <= k valid pixels:', {k: round(float(c[k]), 3) for k in (0, 1, 2, 4, 8, 16, 32)})
