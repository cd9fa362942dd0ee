"""Generates tests/golden/loop.npz: a K-iteration pose / parameter TRAJECTORY of the reference's own optimisation loops
(VERDICT r3 item 6 -- until now the loops of bench_slam.py were only compared with themselves).

    python tests/golden/make_loop_fixtures.py          # needs /root/reference; build container only

What runs, all of it the reference's code (src/vtgaussian_slam.py), imported exactly as make_get_loss_fixtures.py imports it
(stand-in modules for the absent third-party imports, the 'cuda' -> CPU device shim, the float32 CPU oracle behind the
`diff_gaussian_rasterization` operator slot):

    tracking   optimizer = initialize_optimizer(params, config['tracking']['lrs'], tracking=True)          :180-185
               5 x { get_loss(..., tracking=True, tracking_iteration=it, dataset_name='replica', ...)       :407-689
                     loss.backward(); optimizer.step(); optimizer.zero_grad(set_to_none=True) }             :1889-1891
    mapping    optimizer = initialize_optimizer(params, config['mapping']['lrs'], tracking=False)           :187
               5 x { get_loss(..., mapping=True, dataset_name='replica'); backward; step; zero_grad }       :2545-2702

with the Replica learning rates and loss weights (configs/replica/room0.py:75-108).  Stored: the start parameters, the
observation, and after EVERY iteration the loss, the camera pose of the frame (tracking) and the trainable Gaussian
parameters (mapping).  tests/test_loop_fixture.py replays the same iterations on the GPU through the get_loss mirror and
FusedAdam.  Only numbers are stored; no reference source text; the reference never travels to the GPU box.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_get_loss_fixtures as base          # noqa: E402  (the device shim and the stand-in modules)

REF = base.REF
TRACK_LRS = dict(means3D=0.0, rgb_colors=0.0, unnorm_rotations=0.0, logit_opacities=0.0, log_scales=0.0,
                 cam_unnorm_rots=0.0004, cam_trans=0.002)                      # configs/replica/room0.py:78-86
MAP_LRS = dict(means3D=0, rgb_colors=0.0025, unnorm_rotations=0, logit_opacities=0.05, log_scales=0.005,
               cam_unnorm_rots=1e-8, cam_trans=1e-7)                           # :99-107
ITERS = 5


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted; fixtures can only be generated in the build container")
    sys.dont_write_bytecode = True
    base.install_cpu_shim()
    go = base.install_stand_ins()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "src"))
    spec = importlib.util.spec_from_file_location("ref_vtgaussian_slam", os.path.join(REF, "src", "vtgaussian_slam.py"))
    ref = importlib.util.module_from_spec(spec)
    saved_argv, sys.argv = sys.argv, ["vtgaussian_slam.py"]
    try:
        spec.loader.exec_module(ref)
    finally:
        sys.argv = saved_argv
    from utils.slam_external import build_rotation

    W, H, F = 72, 56, 60.0
    g = torch.Generator().manual_seed(20251104)
    k = torch.tensor([[F, 0, W / 2 - 0.5], [0, F, H / 2 - 0.5], [0, 0, 1.0]])
    first_w2c = torch.eye(4)
    cam = go.setup_camera(W, H, k.numpy(), first_w2c.numpy())
    scene, _ = go.view_tied_scene(2600, W, H, seed=11, z_range=(1.5, 4.0))
    n = scene["means3D"].shape[0]
    T, t_idx = 3, 1

    # the observation: the ground-truth map seen from a pose 0.25 deg / 1 cm away from where tracking starts
    q_gt = torch.nn.functional.normalize(torch.tensor([[1.0, 0.0012, -0.0018, 0.0009]]))
    t_gt = torch.tensor([0.006, -0.004, 0.007])
    w2c_gt = torch.eye(4)
    w2c_gt[:3, :3] = build_rotation(q_gt)[0]
    w2c_gt[:3, 3] = t_gt
    with torch.no_grad():
        pts = scene["means3D"]
        pts_cam = (w2c_gt[:3, :3] @ pts.T).T + w2c_gt[:3, 3]
        sc_gt = dict(scene, means3D=pts_cam.contiguous(), opacities=torch.full((n, 1), 0.95))
        z = pts_cam[:, 2:3]
        gt_im, _, _ = go.rasterize(cam=cam, **sc_gt)
        gt_ds, _, _ = go.rasterize(cam=cam, **dict(sc_gt, colors_precomp=torch.cat([z, torch.ones_like(z), z * z], 1)))
    gt_im = (gt_im + 0.01 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    gt_depth = gt_ds[0:1] / gt_ds[1:2].clamp(min=1e-6) + 0.005 * torch.randn(1, H, W, generator=g)
    gt_depth[:, :4, :7] = 0.0                                   # an invalid-depth hole

    gg = torch.Generator().manual_seed(77)
    start = {
        "means3D": scene["means3D"].clone(),
        "rgb_colors": (scene["colors_precomp"] + 0.05 * torch.randn(n, 3, generator=gg)).clamp(0, 1),
        "unnorm_rotations": torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1),
        "logit_opacities": torch.full((n, 1), 2.0) + 0.3 * torch.randn(n, 1, generator=gg),
        "log_scales": torch.log(scene["scales"][:, :1] * 1.1),
        "cam_unnorm_rots": torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1).repeat(1, 1, T).contiguous(),
        "cam_trans": torch.zeros(1, 3, T),
    }
    params = {kk: torch.nn.Parameter(v.clone()) for kk, v in start.items()}
    variables = {"max_2D_radius": torch.zeros(n), "means2D_gradient_accum": torch.zeros(n), "denom": torch.zeros(n),
                 "timestep": torch.zeros(n)}

    def curr_data():
        return {"cam": cam, "im": gt_im.clone(), "depth": gt_depth.clone(), "id": t_idx, "intrinsics": k.clone(),
                "w2c": first_w2c.clone(), "iter_gt_w2c_list": None}

    store = {"W": np.int64(W), "H": np.int64(H), "focal": np.float64(F), "t_idx": np.int64(t_idx), "iters": np.int64(ITERS),
             "gt_im": gt_im.numpy(), "gt_depth": gt_depth.numpy(), "q_gt": q_gt[0].numpy(), "t_gt": t_gt.numpy(),
             **{"start_" + kk: v.numpy().copy() for kk, v in start.items()},
             **{"track_lr_" + kk: np.float64(v) for kk, v in TRACK_LRS.items()},
             **{"map_lr_" + kk: np.float64(v) for kk, v in MAP_LRS.items()}}

    # ---- tracking: the reference's loop body (:1794-1891) ------------------------------------------------------------------
    opt = ref.initialize_optimizer(params, TRACK_LRS, tracking=True)
    mse_ls, thr_ls = [], []
    for it in range(ITERS):
        loss, variables, losses, mse_ls, thr_ls = ref.get_loss(
            params, curr_data(), variables, t_idx, {"im": 0.5, "depth": 0.025}, True, 0.99, True, False, tracking=True,
            plot_dir=None, visualize_tracking_loss=False, tracking_iteration=it, dataset_name="replica",
            presence_sil_mask_mse_ls=mse_ls, sil_thres_ls=thr_ls)
        loss.backward()
        store[f"track{it}_loss"] = np.float64(loss.item())
        store[f"track{it}_grad_q"] = params["cam_unnorm_rots"].grad[0, :, t_idx].numpy().copy()
        store[f"track{it}_grad_t"] = params["cam_trans"].grad[0, :, t_idx].numpy().copy()
        with torch.no_grad():
            opt.step()
            opt.zero_grad(set_to_none=True)
        store[f"track{it}_q"] = params["cam_unnorm_rots"][0, :, t_idx].detach().numpy().copy()
        store[f"track{it}_t"] = params["cam_trans"][0, :, t_idx].detach().numpy().copy()
        print(f"tracking {it}: loss {loss.item():.4f}  t {store[f'track{it}_t']}  thr {thr_ls[-1] if thr_ls else None}")
    store["track_sil_thres"] = np.float64(thr_ls[-1])

    # ---- mapping: the reference's loop body on the current frame (:2545-2702, first submap: one get_loss per iteration) ----
    opt = ref.initialize_optimizer(params, MAP_LRS, tracking=False)
    for it in range(ITERS):
        loss, variables, losses = ref.get_loss(params, curr_data(), variables, t_idx, {"im": 1.0, "depth": 1.0}, False, 0.5, True,
                                               False, mapping=True, dataset_name="replica")
        loss.backward()
        store[f"map{it}_loss"] = np.float64(loss.item())
        with torch.no_grad():
            opt.step()
            opt.zero_grad(set_to_none=True)
        for kk in ("rgb_colors", "logit_opacities", "log_scales"):
            store[f"map{it}_{kk}"] = params[kk].detach().numpy().copy()
        store[f"map{it}_q"] = params["cam_unnorm_rots"][0, :, t_idx].detach().numpy().copy()
        store[f"map{it}_t"] = params["cam_trans"][0, :, t_idx].detach().numpy().copy()
        print(f"mapping {it}: loss {loss.item():.5f}")
    np.savez_compressed(os.path.join(HERE, "loop.npz"), **store)
    print("wrote loop.npz", sum(v.nbytes for v in store.values() if hasattr(v, "nbytes")) // 1024, "KiB")


if __name__ == "__main__":
    main()
