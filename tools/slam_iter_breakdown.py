"""Where one tracking / mapping iteration of bench_slam.py spends its time (synchronised phases)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vtgaussian-slam_amd"), os.path.join(ROOT, "tests")]
import diff_gaussian_rasterization as dgr
import slam_callers as sc
from oracle import gs_oracle as go
from parity_util import to_settings
dev = torch.device("cuda:0")
N, W, H = 1_000_000, 1200, 680
scene, cam = go.view_tied_scene(N, W, H, seed=0)
st = to_settings(cam, dev)
params = {"means3D": scene["means3D"], "rgb_colors": scene["colors_precomp"], "unnorm_rotations": scene["rotations"],
          "logit_opacities": torch.full((N, 1), 2.0), "log_scales": torch.log(scene["scales"][:, :1]),
          "cam_unnorm_rots": torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1).repeat(1, 1, 2), "cam_trans": torch.zeros(1, 3, 2)}
params = {k: torch.nn.Parameter(v.to(dev)) for k, v in params.items()}
w2c = torch.eye(4, device=dev)
gt_im, gt_depth = torch.rand(3, H, W, device=dev), torch.rand(1, H, W, device=dev) + 1
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for mode in ("tracking", "mapping"):
    acc = {}
    for it in range(8):
        t0 = sync()
        tg = sc.transform_to_frame(params, 1, gaussians_grad=(mode == "mapping"), camera_grad=(mode == "tracking"))
        rv = sc.transformed_params2rendervar(params, tg); dv = sc.transformed_params2depthplussilhouette(params, w2c, tg)
        t1 = sync()
        im, radius, _ = dgr.GaussianRasterizer(raster_settings=st)(**rv)
        t2 = sync()
        ds, _, _ = dgr.GaussianRasterizer(raster_settings=st)(**dv)
        t3 = sync()
        loss = sc.tracking_loss(im, ds, gt_im, gt_depth, 0.99) if mode == "tracking" else sc.mapping_loss(im, ds, gt_im, gt_depth)
        t4 = sync()
        loss.backward()
        t5 = sync()
        for p in params.values(): p.grad = None
        if it >= 3:
            for k, v in (("helpers", t1 - t0), ("render rgb", t2 - t1), ("render depth", t3 - t2), ("loss", t4 - t3), ("backward", t5 - t4)):
                acc[k] = acc.get(k, 0) + v * 1e3 / 5
    print(mode, {k: round(v, 2) for k, v in acc.items()}, "allocated GB", round(torch.cuda.memory_allocated() / 1e9, 2),
          "reserved GB", round(torch.cuda.memory_reserved() / 1e9, 2), flush=True)

# ---- fused path (render_frame), including the optimiser step
from diff_gaussian_rasterization.fused import render_frame
lrs_t = dict(means3D=0.0, rgb_colors=0.0, unnorm_rotations=0.0, logit_opacities=0.0, log_scales=0.0, cam_unnorm_rots=0.0004, cam_trans=0.002)
for mode in ("tracking", "mapping"):
    opt = torch.optim.Adam([{"params": [v], "name": k, "lr": lrs_t[k]} for k, v in params.items()])
    acc = {}
    for it in range(8):
        t0 = sync()
        im, ds, radius = render_frame(params, 1, st, w2c, gaussians_grad=(mode == "mapping"), camera_grad=(mode == "tracking"))
        t1 = sync()
        loss = sc.tracking_loss(im, ds, gt_im, gt_depth, 0.99) if mode == "tracking" else sc.mapping_loss(im, ds, gt_im, gt_depth)
        t2 = sync()
        loss.backward()
        t3 = sync()
        opt.step(); opt.zero_grad(set_to_none=True)
        t4 = sync()
        if it >= 3:
            for k, v in (("render_frame fwd", t1 - t0), ("loss", t2 - t1), ("backward", t3 - t2), ("adam", t4 - t3)):
                acc[k] = acc.get(k, 0) + v * 1e3 / 5
    print("fused", mode, {k: round(v, 2) for k, v in acc.items()}, flush=True)
# backward split: loss part vs rasterizer part
im, ds, radius = render_frame(params, 1, st, w2c, gaussians_grad=False, camera_grad=True)
g1, g2 = torch.rand_like(im), torch.rand_like(ds)
for it in range(5):
    im, ds, radius = render_frame(params, 1, st, w2c, gaussians_grad=False, camera_grad=True)
    t0 = sync(); torch.autograd.backward([im, ds], [g1, g2]); t1 = sync()
print("fused tracking: operator backward alone %.2f ms" % ((t1 - t0) * 1e3), flush=True)
dgr.profile_enable(True)
for it in range(5):
    im, ds, radius = render_frame(params, 1, st, w2c, gaussians_grad=False, camera_grad=True)
    torch.autograd.backward([im, ds], [g1, g2])
p = dgr.profile_collect(); dgr.profile_enable(False)
print({k: round(v[0] / 5 * 1e3, 1) for k, v in p.items()}, "(us per iteration, summed over both renders)")
