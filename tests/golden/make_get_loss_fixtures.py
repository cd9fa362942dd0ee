"""Generates tests/golden/get_loss.npz and tests/golden/driver_helpers.npz by running the reference's OWN `get_loss` and `get_vis_mask`
(src/vtgaussian_slam.py:376-404, 407-689) in the build container -- the loss glue of the four dataset branches
(Replica silhouette sweep, TUM / ScanNet / ScanNet++ visibility mask, far-depth filter, 50 x median outlier mask,
`additional_mask`) is captured from the reference instead of being restated.

    python tests/golden/make_get_loss_fixtures.py          # needs /root/reference

src/vtgaussian_slam.py imports, at module level, third-party packages that are absent here and that `get_loss` never
touches (cv2, wandb, Open3D odometry, the dataset loaders, the evaluation helpers).  This script -- and only this
script -- registers empty stand-in modules for those names so that the file can be imported, maps the reference's
hard-coded 'cuda' device to the CPU (as make_helper_fixtures.py does), and puts the float32 CPU ORACLE
(oracle/gs_oracle.py, test infrastructure) behind `diff_gaussian_rasterization.GaussianRasterizer`, the operator this
repository replaces.  What is stored: the operator outputs get_loss saw (im, depth_sil), its other inputs, and what it
returned -- the loss, the per-term weighted losses, d loss / d im and d loss / d depth_sil (captured with tensor hooks) and
the masks of get_vis_mask.  driver_helpers.npz holds, captured the same way, the submap bookkeeping functions
(quantize_selected_time_idx, concat_keyframes_params_base_frame, concat_global, update_params_ls, update_variables_ls,
:884-1020) and the reference's own pieces of compute_point2plane_dist (get_pointcloud, the frustum mask, trans_normal_c2w,
:1070-1155).  No reference source text is stored; the reference never travels to the GPU box.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def install_cpu_shim():
    torch.Tensor.cuda = lambda self, *a, **k: self
    for name in ("zeros", "ones", "eye", "zeros_like", "ones_like", "tensor", "empty"):
        orig = getattr(torch, name)

        def wrap(*a, __orig=orig, **k):
            if "device" in k and str(k["device"]).startswith("cuda"):
                k["device"] = "cpu"
            return __orig(*a, **k)
        setattr(torch, name, wrap)


captured = {}


def install_stand_ins():
    """Empty modules for imports get_loss never uses + the oracle behind the rasterizer operator."""
    from oracle import gs_oracle as go

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mod("cv2")
    mod("wandb")
    names = ["load_dataset_config", "ICLDataset", "ReplicaDataset", "ReplicaV2Dataset", "AzureKinectDataset", "ScannetDataset",
             "Ai2thorDataset", "Record3DDataset", "RealsenseDataset", "TUMDataset", "ScannetPPDataset", "NeRFCaptureDataset"]
    pkg = mod("datasets")
    pkg.__path__ = []
    mod("datasets.gradslam_datasets", **{n: None for n in names})
    mod("utils.eval_helpers", report_loss=None, report_progress=None, eval=None)
    mod("utils.recon_helpers", setup_camera=go.setup_camera)
    mod("visual_odometer", VisualOdometer=None)

    class Renderer:
        """The operator boundary: (color, radii, depth) from the float32 oracle; hooks record dL/d(color)."""
        calls = 0

        def __init__(self, raster_settings):
            self.cam = raster_settings

        def __call__(self, means3D, means2D, opacities, colors_precomp, scales, rotations):
            color, radii, depth = go.rasterize(means3D, means2D, opacities, colors_precomp, scales, rotations, self.cam)
            tag = "im" if Renderer.calls % 2 == 0 else "depth_sil"
            Renderer.calls += 1
            captured[tag] = color.detach().clone()
            captured["g_" + tag] = torch.zeros_like(color)
            color.register_hook(lambda g, t=tag: captured.__setitem__("g_" + t, g.detach().clone()))
            return color, radii, depth
    mod("diff_gaussian_rasterization", GaussianRasterizer=Renderer, GaussianRasterizationSettings=go.OracleCamera)
    return go


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted; fixtures can only be generated in the build container")
    sys.dont_write_bytecode = True
    install_cpu_shim()
    go = install_stand_ins()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "src"))
    spec = importlib.util.spec_from_file_location("ref_vtgaussian_slam", os.path.join(REF, "src", "vtgaussian_slam.py"))
    ref = importlib.util.module_from_spec(spec)
    saved_argv, sys.argv = sys.argv, ["vtgaussian_slam.py"]
    try:
        spec.loader.exec_module(ref)          # defines get_loss / get_vis_mask; `__main__` block does not run
    finally:
        sys.argv = saved_argv

    W, H, F = 72, 56, 60.0
    g = torch.Generator().manual_seed(20251004)
    k = torch.tensor([[F, 0, W / 2 - 0.5], [0, F, H / 2 - 0.5], [0, 0, 1.0]])
    first_w2c = torch.eye(4)
    cam = go.setup_camera(W, H, k.numpy(), first_w2c.numpy())
    scene, _ = go.view_tied_scene(2600, W, H, seed=7, z_range=(1.5, 4.0))
    n = scene["means3D"].shape[0]
    T = 3

    def fresh_params():
        gg = torch.Generator().manual_seed(99)
        return {
            "means3D": torch.nn.Parameter(scene["means3D"].clone()),
            "rgb_colors": torch.nn.Parameter(scene["colors_precomp"].clone()),
            "unnorm_rotations": torch.nn.Parameter(torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1)),
            "logit_opacities": torch.nn.Parameter(torch.logit(scene["opacities"].clamp(0.02, 0.98) * 0.0 + 0.93) + 0.8 * torch.randn(n, 1, generator=gg)),
            "log_scales": torch.nn.Parameter(torch.log(scene["scales"][:, :1] * 1.6)),
            "cam_unnorm_rots": torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1).repeat(1, 1, T)
                                                  + 0.004 * torch.randn(1, 4, T, generator=gg)),
            "cam_trans": torch.nn.Parameter(0.01 * torch.randn(1, 3, T, generator=gg)),
        }

    # ground truth: the scene rendered by the oracle from the first pose + noise, holes in the depth, NaN-free
    with torch.no_grad():
        z = scene["means3D"][:, 2:3]
        gt_im, _, _ = go.rasterize(cam=cam, **scene)
        gt_ds, _, _ = go.rasterize(cam=cam, **dict(scene, colors_precomp=torch.cat([z, torch.ones_like(z), z * z], 1)))
    gt_im = (gt_im + 0.05 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    gt_depth = gt_ds[0:1] / gt_ds[1:2].clamp(min=1e-6) + 0.02 * torch.randn(1, H, W, generator=g)
    gt_depth[:, :6, :9] = 0.0                                   # invalid-depth hole
    gt_depth[:, 30:34, 40:52] = 7.5                             # far outliers (far-depth filter / 50 x median mask)
    intr = k.clone()
    t_idx = 1

    def curr_data():
        return {"cam": cam, "im": gt_im.clone(), "depth": gt_depth.clone(), "id": t_idx, "intrinsics": intr.clone(),
                "w2c": first_w2c.clone(), "iter_gt_w2c_list": None}

    def pose(rot_noise, trans):
        q = torch.nn.functional.normalize(torch.tensor([[1.0, 0, 0, 0]]) + rot_noise * torch.randn(1, 4, generator=g))
        from utils.slam_external import build_rotation
        m = torch.eye(4)
        m[:3, :3] = build_rotation(q)[0]
        m[:3, 3] = torch.tensor(trans)
        return m

    overlaps = [(pose(0.02, [0.05, -0.02, 0.03]), (gt_depth * (1 + 0.03 * torch.randn(1, H, W, generator=g))).clamp(min=0)),
                (pose(0.03, [-0.04, 0.03, 0.02]), (gt_depth * (1 + 0.03 * torch.randn(1, H, W, generator=g))).clamp(min=0)),
                (pose(0.015, [0.02, 0.05, -0.04]), (gt_depth * (1 + 0.03 * torch.randn(1, H, W, generator=g))).clamp(min=0))]
    curr_w2c = pose(0.01, [0.01, 0.0, -0.01])
    add_mask = (torch.rand(3, H, W, generator=g) > 0.7)

    cases = {
        # name: kwargs of get_loss (besides params / curr_data / variables / iter_time_idx)
        "replica_tracking_iter0": dict(loss_weights={"im": 0.5, "depth": 1.0}, use_sil_for_loss=True, sil_thres=0.99, use_l1=True,
                                       ignore_outlier_depth_loss=False, tracking=True, tracking_iteration=0,
                                       dataset_name="replica", presence_sil_mask_mse_ls=[], sil_thres_ls=[]),
        "replica_mapping": dict(loss_weights={"im": 0.5, "depth": 1.0}, use_sil_for_loss=False, sil_thres=0.5, use_l1=True,
                                ignore_outlier_depth_loss=False, mapping=True, dataset_name="replica"),
        "tum_tracking_vis_far": dict(loss_weights={"im": 0.5, "depth": 1.0}, use_sil_for_loss=True, sil_thres=0.9, use_l1=True,
                                     ignore_outlier_depth_loss=False, tracking=True, tracking_iteration=3, dataset_name="tum",
                                     far_depth_filter_thres=5.0, vis_mask_thres=0.05, curr_w2c=curr_w2c,
                                     overlap_w2c=overlaps[0][0], overlap_gtdepth=overlaps[0][1]),
        "scannet_tracking_vis3_outlier": dict(loss_weights={"im": 0.5, "depth": 1.0}, use_sil_for_loss=True, sil_thres=0.9,
                                              use_l1=True, ignore_outlier_depth_loss=True, tracking=True, tracking_iteration=5,
                                              dataset_name="scannet", far_depth_filter_thres=6.0, vis_mask_thres=0.05,
                                              curr_w2c=curr_w2c, overlap_w2c=overlaps[0][0], overlap_gtdepth=overlaps[0][1],
                                              overlap_mid_w2c=overlaps[1][0], overlap_mid_gtdepth=overlaps[1][1],
                                              overlap_last_w2c=overlaps[2][0], overlap_last_gtdepth=overlaps[2][1]),
        "scannetpp_mapping_additional_mask": dict(loss_weights={"im": 0.5, "depth": 1.0}, use_sil_for_loss=False, sil_thres=0.5,
                                                  use_l1=True, ignore_outlier_depth_loss=False, mapping=True,
                                                  dataset_name="scannetpp", additional_mask=add_mask.clone()),
        # the two branches no shipped configuration takes (round 3): no depth term at all (:591-596), and the tracking colour
        # sum over ALL pixels when neither use_sil_for_loss nor ignore_outlier_depth_loss is set (:601-602)
        "tum_tracking_no_l1": dict(loss_weights={"im": 0.5, "depth": 1.0}, use_sil_for_loss=True, sil_thres=0.9, use_l1=False,
                                   ignore_outlier_depth_loss=False, tracking=True, tracking_iteration=2, dataset_name="tum"),
        "tum_tracking_unmasked_colour": dict(loss_weights={"im": 0.5, "depth": 1.0}, use_sil_for_loss=False, sil_thres=0.9,
                                             use_l1=True, ignore_outlier_depth_loss=False, tracking=True, tracking_iteration=2,
                                             dataset_name="tum"),
    }
    store = {"W": np.int64(W), "H": np.int64(H), "gt_im": gt_im.numpy(), "gt_depth": gt_depth.numpy(), "intrinsics": intr.numpy(),
             "curr_w2c": curr_w2c.numpy(), "additional_mask": add_mask.numpy(),
             **{f"overlap{i}_w2c": o[0].numpy() for i, o in enumerate(overlaps)},
             **{f"overlap{i}_gtdepth": o[1].numpy() for i, o in enumerate(overlaps)}}
    for name, kw in cases.items():
        captured.clear()
        params = fresh_params()
        variables = {"max_2D_radius": torch.zeros(n), "means2D_gradient_accum": torch.zeros(n), "denom": torch.zeros(n),
                     "timestep": torch.zeros(n)}
        out = ref.get_loss(params, curr_data(), variables, t_idx, **kw)
        loss, weighted = out[0], out[2]
        loss.backward()
        store[name + "_im"] = captured["im"].numpy()
        store[name + "_depth_sil"] = captured["depth_sil"].numpy()
        store[name + "_g_im"] = captured["g_im"].numpy()
        store[name + "_g_depth_sil"] = captured["g_depth_sil"].numpy()
        store[name + "_loss"] = np.float64(loss.item())
        store[name + "_loss_im"] = np.float64(weighted["im"].item())
        store[name + "_loss_depth"] = np.float64(weighted["depth"].item()) if "depth" in weighted else np.float64("nan")
        if len(out) == 5:
            store[name + "_sil_thres_chosen"] = np.float64(out[4][-1])
        print(name, "loss", loss.item(), {k: float(v) for k, v in weighted.items()})
    # the visibility masks of get_vis_mask on their own (all valid-depth pixels back-projected with curr_w2c)
    ys, xs = torch.where(gt_depth[0] >= 0)
    zz = gt_depth[0, ys, xs]
    pts_cam = torch.stack(((xs - intr[0, 2]) / intr[0, 0] * zz, (ys - intr[1, 2]) / intr[1, 1] * zz, zz), -1)
    pts = (torch.inverse(curr_w2c) @ torch.cat([pts_cam, torch.ones_like(pts_cam[:, :1])], 1).T).T[:, :3]
    for i, (w2c, gd) in enumerate(overlaps):
        store[f"vis_mask{i}"] = ref.get_vis_mask(w2c, pts, intr, gd, 0.05, H, W).numpy()
    np.savez_compressed(os.path.join(OUT, "get_loss.npz"), **store)
    print("wrote get_loss.npz", {k: v.shape for k, v in store.items() if hasattr(v, "shape") and v.ndim > 0})

    # ---- f4: submap bookkeeping (src/vtgaussian_slam.py:884-1020) and the torch pieces of compute_point2plane_dist
    #      (get_pointcloud :76-128 with factor=1, get_frustum_mask :1046-1065, trans_normal_c2w :1158-1178) ---------------
    d = {}
    sizes = [37, 52, 41, 29]
    keys5 = ["means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales"]
    dims = {"means3D": 3, "rgb_colors": 3, "unnorm_rotations": 4, "logit_opacities": 1, "log_scales": 1}
    vkeys = ["max_2D_radius", "means2D_gradient_accum", "denom", "timestep"]
    params_ls, variables_ls = [], []
    for i, m in enumerate(sizes):
        pr = {kk: torch.randn(m, dims[kk], generator=g) for kk in keys5}
        pr["cam_unnorm_rots"] = torch.randn(1, 4, 12, generator=g)
        pr["cam_trans"] = torch.randn(1, 3, 12, generator=g)
        params_ls.append(pr)
        vr = {kk: torch.rand(m, generator=g) for kk in vkeys}
        vr["scene_radius"] = torch.tensor(2.0 + i)
        variables_ls.append(vr)
        for kk, vv in {**pr, **vr}.items():
            d[f"in{i}_{kk}"] = vv.numpy().copy()
    selected = [3, 11, 4, 9]                 # frames -> base frames {0, 1, 2} with 4 frames per base frame
    nfe = 4
    d["selected_time_idx"] = np.array(selected)
    d["num_frames_each_base_frame"] = np.int64(nfe)
    d["quantized"] = np.array(ref.quantize_selected_time_idx(selected, nfe))
    cat_p, cat_v, num_gs = ref.concat_keyframes_params_base_frame(params_ls, variables_ls, selected, nfe)
    d["cat_num_gs"] = np.array(num_gs)
    for kk, vv in cat_p.items():
        d["cat_p_" + kk] = vv.detach().numpy().copy()
    for kk, vv in cat_v.items():
        d["cat_v_" + kk] = vv.detach().numpy().copy()
    glob_p = {kk: torch.randn(23, dims[kk], generator=g) for kk in keys5}
    glob_v = {kk: torch.rand(23, generator=g) for kk in vkeys}
    for kk, vv in {**glob_p, **glob_v}.items():
        d["glob_" + kk] = vv.numpy().copy()
    gp, gv, gnum = ref.concat_global(cat_p, cat_v, list(num_gs), glob_p, glob_v)
    d["global_num_gs"] = np.array(gnum)
    for kk, vv in gp.items():
        d["global_p_" + kk] = vv.detach().numpy().copy()
    for kk, vv in gv.items():
        d["global_v_" + kk] = vv.detach().numpy().copy()
    # after an "optimisation" (perturbed copies), write back into the per-base-frame lists
    new_p = {kk: (vv.detach() + 0.5) for kk, vv in cat_p.items()}
    new_v = {kk: (vv.detach() * 2.0) for kk, vv in cat_v.items()}
    upd_p = ref.update_params_ls([dict(x) for x in params_ls], selected, new_p, list(num_gs), nfe)
    upd_v = ref.update_variables_ls([dict(x) for x in variables_ls], selected, new_v, list(num_gs), nfe)
    for i in range(len(sizes)):
        for kk in keys5:
            d[f"upd{i}_{kk}"] = upd_p[i][kk].numpy().copy()
        for kk in vkeys:
            d[f"upd{i}_{kk}"] = upd_v[i][kk].numpy().copy()
    # point-to-plane pieces
    Hp, Wp = 40, 52
    kp = torch.tensor([[45.0, 0, Wp / 2 - 0.3], [0, 44.0, Hp / 2 + 0.2], [0, 0, 1.0]])
    depth0 = 2.0 + 0.4 * torch.rand(1, Hp, Wp, generator=g)
    depth0[:, :3, :5] = 0.0
    col0 = torch.rand(3, Hp, Wp, generator=g)
    w2c_a, w2c_b = pose(0.02, [0.03, -0.01, 0.02]), pose(0.03, [-0.02, 0.04, 0.01])
    mask0 = (depth0 > 0).reshape(-1)
    pc = ref.get_pointcloud(col0, depth0, kp, w2c_a, mask=mask0, factor=1)
    nrm_cam = torch.nn.functional.normalize(torch.randn(int(mask0.sum()), 3, generator=g), dim=1)
    d.update(p2p_depth0=depth0.numpy(), p2p_color0=col0.numpy(), p2p_k=kp.numpy(), p2p_w2c_a=w2c_a.numpy(), p2p_w2c_b=w2c_b.numpy(),
             p2p_pointcloud=pc.numpy(), p2p_normals_cam=nrm_cam.numpy(),
             p2p_normals_world=ref.trans_normal_c2w(nrm_cam.numpy(), w2c_a),
             p2p_frustum=ref.get_frustum_mask(w2c_b, kp, pc[:, :3], Hp, Wp).numpy())
    np.savez_compressed(os.path.join(OUT, "driver_helpers.npz"), **d)
    print("wrote driver_helpers.npz", len(d), "arrays")


if __name__ == "__main__":
    main()
