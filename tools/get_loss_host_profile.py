"""Host time of one SLAM iteration through the get_loss mirror (tiny scene: the GPU work is negligible, so the wall clock per
iteration IS the host's): what bench_slam.py's early frames are bound by (round 6: the trace of a fresh map shows the GPU idle
for a quarter of a tracking iteration, gpurun_out/r6/slamlate_j_dens.txt).  cProfile of the calling thread on top."""
import cProfile, io, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vtgaussian-slam_amd"), os.path.join(ROOT, "tests")]
from oracle import gs_oracle as go
from parity_util import to_settings
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization.get_loss import get_loss
from diff_gaussian_rasterization.optim import FusedAdam

dev = torch.device("cuda:0")
N, W, H, T = 2000, 64, 48, 3
scene, cam = go.view_tied_scene(N, W, H, seed=0)
st = to_settings(cam, dev)
params = {"means3D": scene["means3D"], "rgb_colors": scene["colors_precomp"], "unnorm_rotations": scene["rotations"],
          "logit_opacities": torch.logit(scene["opacities"].clamp(1e-4, 1 - 1e-4)), "log_scales": torch.log(scene["scales"][:, :1]),
          "cam_unnorm_rots": torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1).repeat(1, 1, T), "cam_trans": torch.zeros(1, 3, T)}
params = {k: torch.nn.Parameter(v.to(dev).float().contiguous()) for k, v in params.items()}
w2c = torch.eye(4, device=dev)
curr = {"cam": st, "im": torch.rand(3, H, W, device=dev), "depth": torch.rand(1, H, W, device=dev) + 1.0, "id": 1, "w2c": w2c}
variables = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom")}
lrs_t = dict(means3D=0.0, rgb_colors=0.0, unnorm_rotations=0.0, logit_opacities=0.0, log_scales=0.0, cam_unnorm_rots=4e-4, cam_trans=2e-3)
lrs_m = dict(means3D=0.0, rgb_colors=0.0025, unnorm_rotations=0.0, logit_opacities=0.05, log_scales=0.005, cam_unnorm_rots=1e-8, cam_trans=1e-7)


def loop(tracking, iters):
    global variables
    opt = FusedAdam([{"params": [v], "name": k, "lr": (lrs_t if tracking else lrs_m)[k]} for k, v in params.items()], skip_frozen=True)
    mse, thr = [], []
    for it in range(iters):
        if tracking:
            loss, variables, _l, mse, thr = get_loss(params, curr, variables, 1, {"im": 0.5, "depth": 0.025}, True, 0.99, True, False,
                                                     tracking=True, tracking_iteration=it, dataset_name="replica",
                                                     presence_sil_mask_mse_ls=mse, sil_thres_ls=thr)
        else:
            loss, variables, _l = get_loss(params, curr, variables, 1, {"im": 1.0, "depth": 1.0}, False, 0.99, True, False, mapping=True,
                                           dataset_name="replica")
        loss.backward()
        opt.step(); opt.zero_grad(set_to_none=True)


for tracking in (True, False):
    loop(tracking, 60); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(tracking, 400); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 400 * 1e6
    print(f"{'tracking' if tracking else 'mapping'} iteration through get_loss: {dt:.0f} us of wall clock (host-bound scene)", flush=True)
    pr = cProfile.Profile(); pr.enable(); loop(tracking, 200); torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:26]), flush=True)
