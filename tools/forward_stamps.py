"""Where one wavefront of composite_forward_q spends its time (diagnostic build with -DVTGS_Q_STAMPS: s_memtime stamps,
8 words per tile in the workspace's debug region).

    python vtgaussian-slam_amd/build.py --out vtgaussian-slam_amd/lib/libvtgs_stamps.so -DVTGS_Q_STAMPS
    VTGS_LIBRARY=.../libvtgs_stamps.so python tools/forward_stamps.py
"""
import ctypes, os, sys, torch
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
leaves = {k: v.to(dev) for k, v in scene.items()}
with torch.no_grad():
    for _ in range(4):
        rast(**leaves)
torch.cuda.synchronize()
fs = rast._last_state
out = (ctypes.c_uint64 * 12)()
dgr._lib.vtgs_debug_layout(fs.n, fs.cam.W, fs.cam.H, fs.capacity, fs.tile_cap, out)
tiles = int(out[7])
st_raw = fs.workspace[int(out[9]) + 256: int(out[9]) + 256 + 48 * tiles].view(torch.int32).reshape(tiles, 12).cpu()
st = st_raw[:, :8].double()
fwd_raw = st_raw.clone()
# ---- project_and_bin: 8 words per workgroup behind the per-tile stamps
nblk = (N + 1023) // 1024
poff = int(out[9]) + ((256 + 48 * tiles + 255) // 256) * 256
pj = fs.workspace[poff: poff + 32 * nblk].view(torch.int32).reshape(nblk, 8).cpu()
pjd = pj.double()
print("project_and_bin (ticks per workgroup of 1024 Gaussians):")
for i, nm in enumerate(["inputs + projection + walk set-up", "table clear + pass 1 (reach tests, LDS histogram)",
                        "scans + instance atomic + per-tile global reservations", "records + pass 2 (slots, bin entries)", "whole workgroup"]):
    c = pjd[:, i]
    print(f"  {nm:58s} mean {c.mean():9.1f}  median {c.median():9.1f}  p90 {c.quantile(0.9):9.1f}  max {c.max():9.1f}")
import numpy as np
pt0 = (pj[:, 5].long() & 0xFFFFFFFF).numpy().astype(np.int64); pt1 = (pj[:, 6].long() & 0xFFFFFFFF).numpy().astype(np.int64)
okp = pt1 >= pt0
b0 = pt0[okp].min()
life = (pt1[okp] - pt0[okp]) * 0.01
print(f"  {okp.sum()} workgroups, span {(pt1[okp].max() - b0) * 0.01:.1f} us, workgroup life mean {life.mean():.1f} us (p10 {np.quantile(life, .1):.1f}, p90 {np.quantile(life, .9):.1f}); "
      f"starts: {[round(float(x), 1) for x in np.sort((pt0[okp] - b0) * 0.01)[::100]]}")
names = ["sort (entry -> sorted list re-readable)", "first append", "all appends", "all steps", "whole wavefront", "steps", "entry -> loop exit (before the image stores)"]
for i, nm in enumerate(names):
    c = st[:, i]
    print(f"{nm:42s} mean {c.mean():10.1f}  median {c.median():10.1f}  p90 {c.quantile(0.9):10.1f}  max {c.max():10.1f}")
pro = st[:, 7] - st[:, 0]                       # sort end -> loop entry: LDS init, the first three chunks' gathers and appends
loop = st[:, 6] - st[:, 7]                      # loop entry -> loop exit
ctl = loop - st[:, 3] - (st[:, 2] - 0)          # minus steps, minus the stamped appends (the prologue's first append is in both: see below)
for nm, c in (("  sort end -> loop entry (prologue)", pro), ("  loop entry -> loop exit", loop),
              ("  loop minus steps minus appends (control, retire, waits)", ctl)):
    print(f"{nm:42s} mean {c.mean():10.1f}  median {c.median():10.1f}  p90 {c.quantile(0.9):10.1f}  max {c.max():10.1f}")
per_step = (st[:, 3] / st[:, 5].clamp(min=1)).median()
print(f"ticks per step (median tile) {per_step:.0f}")
# ---- backward (composite_backward_mx writes its own stamps into the same region)
lv = {k: v.clone().requires_grad_(True) for k, v in leaves.items()}
g = torch.rand(3, H, W, device=dev)
for _ in range(3):
    c, _, _ = rast(**lv)
    c.backward(g)
torch.cuda.synchronize()
fs = rast._last_state
st_raw = fs.workspace[int(out[9]) + 256: int(out[9]) + 256 + 48 * tiles].view(torch.int32).reshape(tiles, 12).cpu()
st = st_raw[:, :8].double()
print("backward:")
if dgr.get_option("VTGS_BWD_IMPL") == 3:
    w7 = st[:, 7].long() & 0xFFFFFFFF
    cols = (("appends", st[:, 0]), ("pops (queue, ids, gathers requested)", st[:, 1]), ("exponent + g.c MFMAs, front sweep", st[:, 2]),
            ("colour products, back sweep, u' products", st[:, 3]), ("accumulator adds (waited for)", st[:, 6]),
            ("promote: tile coefficients of the next step", (w7 & 0xFFFFF).double()), ("retire: records out", ((w7 >> 20) * 16).double()),
            ("whole wavefront", st[:, 4]), ("steps", st[:, 5]))
else:
    cols = (("prologue (entry -> list loop)", st[:, 0]), ("waiting for the chunk gathers", st[:, 2]),
            ("batches (sweeps + contraction + records)", st[:, 3]), ("whole wavefront", st[:, 4]), ("batches", st[:, 5]),
            ("list length", st[:, 6]))
for nm, c in cols:
    print(f"{nm:42s} mean {c.mean():10.1f}  median {c.median():10.1f}  p90 {c.quantile(0.9):10.1f}  max {c.max():10.1f}")
print(f"ticks per batch / step (median tile) {((st[:, 3] if dgr.get_option('VTGS_BWD_IMPL') != 3 else st[:, 1] + st[:, 2] + st[:, 3] + st[:, 6]) / st[:, 5].clamp(min=1)).median():.0f}")


def timeline(raw, name):
    """Occupancy over the kernel's life from the per-wavefront start / end stamps (10 ns units of the chip-wide constant clock):
    how many of the wavefronts are resident at each instant, and how much of the kernel runs below 50 / 80 % of the peak."""
    import numpy as np
    t0 = (raw[:, 8].long() & 0xFFFFFFFF).numpy().astype(np.int64)
    t1 = (raw[:, 9].long() & 0xFFFFFFFF).numpy().astype(np.int64)
    ok = (t1 >= t0) & ((raw[:, 9] != 0).numpy())
    t0, t1 = t0[ok], t1[ok]
    base = t0.min()
    t0, t1 = (t0 - base) * 0.01, (t1 - base) * 0.01            # us
    span = t1.max()
    grid = np.linspace(0, span, 400)
    occ = np.array([((t0 <= g) & (t1 > g)).sum() for g in grid])
    peak = occ.max()
    xcc = (raw[:, 11].long() & 0xF).numpy()[ok]
    print(f"{name}: {ok.sum()} wavefronts, span {span:.1f} us, mean wavefront life {np.mean(t1 - t0):.1f} us (p10 {np.quantile(t1 - t0, .1):.1f}, p90 {np.quantile(t1 - t0, .9):.1f}), "
          f"peak residency {peak}, mean residency {occ.mean():.0f} ({occ.mean() / peak:.2f} of peak)")
    print(f"   time below 80 % of peak: {(occ < 0.8 * peak).mean() * span:.1f} us, below 50 %: {(occ < 0.5 * peak).mean() * span:.1f} us; "
          f"wavefront-us = {np.sum(t1 - t0):.0f} => at peak residency the work would take {np.sum(t1 - t0) / peak:.1f} us")
    print("   residency at 10 % steps of the span:", [int(occ[int(i * 39.9)]) for i in range(0, 11)])
    print("   wavefronts per XCC:", [int((xcc == x).sum()) for x in range(8)])
    # life of a wavefront against WHEN it started (late starters run on an emptying chip): is a wavefront faster alone?
    order = np.argsort(t0)
    life = (t1 - t0)[order]
    per_unit = None
    if name == "backward":
        bat = raw[:, 3].double().numpy()[ok][order] / np.maximum(raw[:, 5].double().numpy()[ok][order], 1)   # ticks per batch
        per_unit = bat
    else:
        stp = raw[:, 3].double().numpy()[ok][order] / np.maximum(raw[:, 5].double().numpy()[ok][order], 1)   # ticks per step
        per_unit = stp
    n = len(life)
    print("   by start order (tenths): mean life us", [round(float(life[i * n // 10:(i + 1) * n // 10].mean()), 1) for i in range(10)])
    print("   by start order (tenths): ticks per batch/step", [int(per_unit[i * n // 10:(i + 1) * n // 10].mean()) for i in range(10)])
    last = order[-600:]
    print(f"   the last 600 starters: mean life {float((t1 - t0)[last].mean()):.1f} us, ticks per batch/step {float(per_unit[-600:].mean()):.0f}")
    starts = np.sort(t0)
    print("   start times (us) of every 1000th wavefront:", [round(float(starts[i]), 1) for i in range(0, len(starts), 1000)])


timeline(st_raw, "backward")
timeline(fwd_raw, "forward")
