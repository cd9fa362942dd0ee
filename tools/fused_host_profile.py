"""Host time of one fused SLAM iteration by piece (tiny scene: GPU work negligible).  Wall-clock wrappers, so the pieces
that run on the autograd engine's thread are seen too."""
import functools, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vtgaussian-slam_amd"), os.path.join(ROOT, "tests")]
from oracle import gs_oracle as go
from parity_util import to_settings
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import fused, losses, optim

acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    raw = f.__func__ if isinstance(f, staticmethod) else f
    @functools.wraps(raw)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return raw(*a, **k)
        finally:
            d = acc.setdefault(label, [0.0, 0]); d[0] += time.perf_counter() - t; d[1] += 1
    setattr(obj, name, staticmethod(g) if isinstance(obj, type) and name in ("forward", "backward") else g)

wrap(fused._RenderFrame, "forward", "render_frame: autograd forward (prepare_frame + dual forward + wait)")
wrap(fused._RenderFrame, "backward", "render_frame: autograd backward (dual backward + frame epilogue)")
wrap(losses._SlamLoss, "forward", "loss node: forward")
wrap(losses._SlamLoss, "backward", "loss node: backward")
wrap(optim.FusedAdam, "step", "FusedAdam.step")

dev = torch.device("cuda:0")
N, W, H, T = 2000, 64, 48, 3
scene, cam = go.view_tied_scene(N, W, H, seed=0)
st = to_settings(cam, dev)
params = {"means3D": scene["means3D"], "rgb_colors": scene["colors_precomp"], "unnorm_rotations": scene["rotations"],
          "logit_opacities": torch.logit(scene["opacities"].clamp(1e-4, 1 - 1e-4)), "log_scales": torch.log(scene["scales"][:, :1]),
          "cam_unnorm_rots": torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1).repeat(1, 1, T), "cam_trans": torch.zeros(1, 3, T)}
params = {k: torch.nn.Parameter(v.to(dev).float().contiguous()) for k, v in params.items()}
w2c = torch.eye(4, device=dev)
gt_im, gt_depth = torch.rand(3, H, W, device=dev), torch.rand(1, H, W, device=dev) + 1.0
lrs = dict(means3D=0.0, rgb_colors=0.0025, unnorm_rotations=0.0, logit_opacities=0.05, log_scales=0.005, cam_unnorm_rots=4e-4, cam_trans=2e-3)

def loop(tracking, iters):
    opt = optim.FusedAdam([{"params": [v], "name": k, "lr": lrs[k]} for k, v in params.items()], skip_frozen=True)
    for _ in range(iters):
        im, ds, _ = fused.render_frame(params, 1, st, w2c, gaussians_grad=not tracking, camera_grad=tracking)
        loss = (losses.tracking_loss(im, ds, gt_im, gt_depth, 0.5) if tracking else losses.mapping_loss(im, ds, gt_im, gt_depth))
        loss.backward()
        opt.step(); opt.zero_grad(set_to_none=True)

for tracking in (True, False):
    loop(tracking, 50); torch.cuda.synchronize(); acc.clear()
    t0 = time.perf_counter(); loop(tracking, 300); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300 * 1e6
    print(f"{'tracking' if tracking else 'mapping'} iteration: {dt:.0f} us of wall clock")
    for k, (s, n) in acc.items():
        print(f"    {k:75s} {s / 300 * 1e6:7.1f} us")
