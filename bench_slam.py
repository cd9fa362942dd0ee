#!/usr/bin/env python3
"""SLAM-loop benchmark (metric 2, on the synthetic sequence): frames/s of the reference's per-frame
tracking + mapping optimisation, driven through the drop-in operator exactly the way `get_loss` drives it
(src/vtgaussian_slam.py:407-689): per iteration one RGB render and one [z,1,z^2] render, the Replica tracking /
mapping losses, `loss.backward()`, Adam.  Iteration counts, loss weights and learning rates are those of
configs/replica/room0.py (tracking 60 it, lrs 4e-4 / 2e-3; mapping 100 it, lrs 2.5e-3 / 5e-2 / 5e-3).

The Replica data set is not available offline, so the sequence is synthetic: one view-tied submap
(N Gaussians, one per pixel + edge splats) observed from a slowly moving camera; ground-truth colour and
depth of every frame are rendered from the ground-truth pose.  Tracking starts each frame from the previous estimate
and must recover the motion -- the reported pose error is an end-to-end check of the pose gradient.

    python bench_slam.py --frames 3                        # one JSON line
    python bench_slam.py --frames 3 --shared-geometry      # depth/silhouette pass reuses the RGB pass's binning (8f-2)
    python bench_slam.py --frames 3 --fused                # + fused caller chain (8f-1) and loss kernels (8f-3)
"""
import argparse
import json
import math
import os

import numpy as np
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "vtgaussian-slam_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def pose_quat_trans(angle_deg: float, axis, trans):
    a = math.radians(angle_deg) / 2
    ax = torch.tensor(axis, dtype=torch.float32)
    ax = ax / ax.norm()
    return torch.cat([torch.tensor([math.cos(a)]), math.sin(a) * ax]), torch.tensor(trans, dtype=torch.float32)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=680)
    ap.add_argument("--tracking-iters", type=int, default=60)
    ap.add_argument("--mapping-iters", type=int, default=100)
    ap.add_argument("--shared-geometry", action="store_true", help="depth/silhouette pass reuses the RGB pass's binning (8f-2)")
    ap.add_argument("--fused", action="store_true", help="fused pose transform + render variables + both renders (8f-1), "
                                                         "loss kernels, threshold sweep and one-launch Adam (8f-3)")
    ap.add_argument("--get-loss", action="store_true", help="drive both loops through diff_gaussian_rasterization.get_loss."
                                                            "get_loss, the function with the reference's own signature "
                                                            "(implies --fused): what a one-import swap in the driver gives")
    ap.add_argument("--graph", action="store_true",
                    help="capture one iteration of each phase (get_loss + backward: ~25 launches) into a hipGraph "
                         "(torch.cuda.graph) and replay it; the optimiser step stays eager (its bias corrections are host "
                         "scalars).  Opt-in: the reference's loops would need these few lines around their get_loss call "
                         "(needs --get-loss)")
    ap.add_argument("--global-submaps", type=int, default=0,
                    help="K > 0: every mapping iteration makes the reference's SECOND get_loss call as well, over the global "
                         "set = K fixed submaps (+) the current one (src/vtgaussian_slam.py:2545-2556, 944-977): (K+1) N "
                         "Gaussians rendered, gradients to the current submap only (needs --get-loss)")
    ap.add_argument("--base-frame-every", type=int, default=0,
                    help="E > 0: frame t (1-based) is a BASE frame when t %% E == 0, the others are ordinary frames of the "
                         "current submap -- the two mapping regimes of the reference (configs/replica/room0.py:35: E = 40).  "
                         "Ordinary frame: every mapping iteration draws ONE keyframe at random from the frames of the submap so "
                         "far (src/vtgaussian_slam.py:2563-2585) and renders THAT view; with --global-submaps the second "
                         "get_loss call is made when the draw is the submap's base frame (:2600-2604).  Base frame: the "
                         "current view every iteration and, with --global-submaps, BOTH calls every iteration (:2545-2557).  "
                         "0 (default): every frame maps on its own view, second call (if any) every iteration -- round 3's loop")
    ap.add_argument("--warmup-frames", type=int, default=0,
                    help="frames processed BEFORE the clock starts (tracked and mapped like the others, not counted): the first "
                         "frame of a process settles capacities, allocator pools and clocks and runs 1.5-3 x slower per iteration")
    ap.add_argument("--emulate-window", type=int, default=0,
                    help="W > 0 (with --base-frame-every): an ordinary frame draws its keyframe as if the submap already held W "
                         "frames -- with probability 1/W the submap's base frame (then the second get_loss call is made), else "
                         "one of the frames really present.  A short run has windows of 2-3 frames, where the base frame is "
                         "drawn every second or third iteration; over the 39 ordinary frames of a 40-frame submap the mean of "
                         "1/(i+1) is 0.084, i.e. W = 12")
    ap.add_argument("--densify", action="store_true",
                    help="the densification step of every ORDINARY frame's mapping phase (src/vtgaussian_slam.py:2349-2380 -> "
                         "add_new_gaussians_base_frame, :732-813): one forward-only silhouette render under the tracked pose, "
                         "depth_error.median(), a Gaussian for every pixel whose silhouette is below 0.5 or whose rendered depth "
                         "lies behind the observation by more than 50 x the median error -- plus, for one such pixel in twelve "
                         "(the edge share of Appendix B), four half-scale Gaussians on the 2 x grid -- appended to the current "
                         "submap, so N grows over a submap's 40 frames as in the reference.  One GPU only")
    ap.add_argument("--owned-sets", default="auto", choices=["auto", "on", "off"],
                    help="N-rank loop: run the per-Gaussian kernels over the list of Gaussians that can meet the rank's band "
                         "(partition.OwnedSet, rebuilt at every phase) instead of over the whole map.  auto = from 2 M Gaussians up: "
                         "below that the whole-map check before every render costs what the shorter kernels save (DESIGN.md 5)")
    ap.add_argument("--mapping-exchange", default="allreduce", choices=["allreduce", "owner"],
                    help="N-rank mapping: allreduce = one flat all-reduce of the trainable per-Gaussian gradients (20 B each) and Adam "
                         "on every row on every rank; owner = partition.OwnerExchange (needs owned sets): halo gradients to the owner "
                         "band, Adam on the owned rows, the updated rows back to the ranks that list them, one all-reduce per phase")
    ap.add_argument("--autograd-threads", default="auto", choices=["auto", "engine", "caller"],
                    help="caller = torch.autograd.set_multithreading_enabled(False): backward() runs on the calling thread instead "
                         "of being handed to the engine's device thread (~40 us per iteration).  auto = caller on N > 1 ranks, "
                         "where a rank's iteration is bound by the host (DESIGN.md 5), the engine's default on one GPU")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N ranks (started by torch.distributed.run): 'nccl' = RCCL, one GPU per rank; 'gloo' = rehearsal of the "
                         "N-rank code path with every rank on GPU 0 and the collectives staged through the host")
    args = ap.parse_args(argv)
    if args.get_loss:
        args.fused = True
    if args.global_submaps and not args.get_loss:
        ap.error("--global-submaps needs --get-loss")
    if args.graph and not args.get_loss:
        ap.error("--graph needs --get-loss")
    if args.graph and args.base_frame_every:
        ap.error("--graph replays ONE captured view: it cannot follow the per-iteration keyframe draw of --base-frame-every")
    return args


def run(args) -> dict:
    """One run of the loop; returns the JSON record (bench.py embeds a short run of it as its `slam` block)."""
    assert torch.cuda.is_available(), "bench_slam.py needs an MI355X"
    # N ranks = the tile-row partition: every rank holds all Gaussians, renders one band of 16-pixel tile rows,
    # owns the loss terms of its rows; per iteration ONE small all-reduce in tracking (7 pose-gradient floats), and in
    # mapping the SSIM halo rows, 8 loss sums and the trainable per-Gaussian gradients (20 B each).  Adam runs replicated.
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    backend = getattr(args, "backend", "nccl")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if (world > 1 and backend == "nccl") else 0)
    torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if world > 1 and (args.get_loss or args.graph or not args.fused):
        raise SystemExit("bench_slam.py: the N-rank loop runs on the fused route (--fused, without --get-loss / --graph)")
    if args.graph:
        # Everything -- the eager iterations too -- runs on ONE side stream: a graph cannot be captured on the default stream,
        # and the AccumulateGrad nodes of the parameters remember the stream they were created under (a node created by an
        # eager iteration on the default stream and still referenced would drag a cross-stream sync into the capture).
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(side)

    at_mode = getattr(args, "autograd_threads", "auto")
    caller_thread = at_mode == "caller" or (at_mode == "auto" and world > 1)
    if caller_thread:
        torch.autograd.set_multithreading_enabled(False)

    import diff_gaussian_rasterization as dgr
    import slam_callers as sc
    from oracle import gs_oracle as go            # scene generator only
    from parity_util import to_settings

    WU = max(0, getattr(args, "warmup_frames", 0))
    N, W, H, T = args.n, args.width, args.height, args.frames + WU + 1
    from diff_gaussian_rasterization import partition as pt
    band = pt.band_for_rank(H, world, rank) if world > 1 else None
    scene, cam = go.view_tied_scene(N, W, H, seed=0)
    settings = to_settings(cam, dev)
    first_w2c = torch.eye(4, device=dev)
    fx = W / 2.0
    gt_params = {
        "means3D": scene["means3D"].to(dev), "rgb_colors": scene["colors_precomp"].to(dev),
        "unnorm_rotations": scene["rotations"].to(dev), "logit_opacities": torch.full((N, 1), 3.0, device=dev),
        "log_scales": torch.log(scene["scales"][:, :1]).to(dev),
    }
    if getattr(args, "densify", False):
        # The world beyond the first frame's frustum: a ring of view-tied Gaussians around the frame (96 px wide at frame 0,
        # smooth depth), present in the OBSERVATIONS only.  The map starts as the first frame's back-projection, so the camera
        # path uncovers ~1.6 px of unmapped border per frame -- what the reference's densification step finds in a real
        # sequence (newly visible surface with valid depth; N grows by 10-30 % over a submap).
        m = 96
        yy, xx = torch.meshgrid(torch.arange(-m, H + m, dtype=torch.float32), torch.arange(-m, W + m, dtype=torch.float32), indexing="ij")
        ring = ((xx < 0) | (xx >= W) | (yy < 0) | (yy >= H)).reshape(-1)
        rx, ry = xx.reshape(-1)[ring], yy.reshape(-1)[ring]
        rz = 3.5 + 1.2 * torch.sin(rx / 97.0) * torch.cos(ry / 71.0)
        gr = torch.Generator().manual_seed(77)
        ring_params = {
            "means3D": torch.stack([(rx - (W / 2.0 - 0.5) + 0.5) / fx * rz, (ry - (H / 2.0 - 0.5) + 0.5) / fx * rz, rz], 1).to(dev),
            "rgb_colors": torch.rand(rx.numel(), 3, generator=gr).to(dev),
            "unnorm_rotations": torch.tensor([1.0, 0, 0, 0]).repeat(rx.numel(), 1).to(dev),
            "logit_opacities": torch.full((rx.numel(), 1), 3.0, device=dev), "log_scales": torch.log(rz / fx)[:, None].to(dev)}
        obs_params = {k: torch.cat([gt_params[k], ring_params[k]], 0) for k in gt_params}
    else:
        obs_params = gt_params
    # ground-truth camera path: 0.15 deg and 4 mm per frame
    gt_rots = torch.zeros(1, 4, T, device=dev)
    gt_trans = torch.zeros(1, 3, T, device=dev)
    for t in range(T):
        q, tr = pose_quat_trans(0.15 * t, (0.3, 1.0, 0.1), (0.004 * t, -0.002 * t, 0.003 * t))
        gt_rots[0, :, t], gt_trans[0, :, t] = q.to(dev), tr.to(dev)

    from diff_gaussian_rasterization.fused import render_frame
    from diff_gaussian_rasterization import losses as fl
    if args.fused:                                    # loss kernels, threshold sweep and one-launch Adam (8f-3)
        from diff_gaussian_rasterization.optim import FusedAdam
        track_loss = lambda im, ds, gi, gd, thr: fl.tracking_loss(im, ds, gi, gd, thr)
        map_loss = lambda im, ds, gi, gd: fl.mapping_loss(im, ds, gi, gd)
        import functools
        pick_threshold, make_adam = fl.best_silhouette_threshold, functools.partial(FusedAdam, skip_frozen=True)
        if world > 1:                                 # the same terms, summed over the bands
            track_loss = lambda im, ds, gi, gd, thr: pt.band_tracking_loss(im, ds, gi, gd, band, thr)
            map_loss = lambda im, ds, gi, gd: pt.band_mapping_loss(im, ds, gi, gd, band, rank, world)
            pick_threshold = lambda im, sil, gi, gd: pt.band_silhouette_threshold(im, sil, gi, gd, band, world)
    else:                                             # exactly the reference's PyTorch formulation
        track_loss = lambda im, ds, gi, gd, thr: sc.tracking_loss(im, ds, gi, gd, thr)
        map_loss = lambda im, ds, gi, gd: sc.mapping_loss(im, ds, gi, gd)
        pick_threshold, make_adam = sc.best_silhouette_threshold, torch.optim.Adam

    # N ranks: the list of Gaussians that can meet this rank's band under the pose of view t_idx, built when a phase first
    # renders that view and dropped at the end of the phase (owned_done: the device counters are read there -- a Gaussian that
    # escaped a list would make the phase's renders differ from the band render of the whole map, and stops the run)
    mode_owned = getattr(args, "owned_sets", "auto")
    use_owned = world > 1 and args.fused and (mode_owned == "on" or (mode_owned == "auto" and args.n >= 2_000_000))
    owned_sets, owned_stats = {}, {"built": 0, "listed": 0, "escapes": 0}

    def owned_for(params, t_idx):
        if not use_owned:
            return None
        own = owned_sets.get(t_idx)
        if own is None:
            own = owned_sets[t_idx] = pt.OwnedSet(params, t_idx, settings, first_w2c, band)
            owned_stats["built"] += 1
            owned_stats["listed"] += len(own)
        return own

    # --mapping-exchange owner: one exchange object per mapping phase, built on a union list whose margin covers every view of
    # the submap's window (bound on the image shift between the phase's reference view and the window's views, from the poses)
    use_owner = use_owned and getattr(args, "mapping_exchange", "allreduce") == "owner"
    owner_stats = {"phases": 0, "bytes_sent": 0, "iterations": 0, "halo_rows": 0, "own_rows": 0}
    if getattr(args, "mapping_exchange", "allreduce") == "owner" and not use_owned and world > 1:
        raise SystemExit("bench_slam.py: --mapping-exchange owner needs owned sets (--owned-sets on, or auto with N >= 2 M)")

    def make_exchange(params, t_ref, views):
        with torch.no_grad():
            q = torch.nn.functional.normalize(params["cam_unnorm_rots"][0].detach(), dim=0).cpu()      # [4, T]
            tr = params["cam_trans"][0].detach().cpu()
            ang = max(2.0 * math.acos(min(1.0, abs(float((q[:, t_ref] * q[:, v]).sum())))) for v in views)
            dt = max(float((tr[:, t_ref] - tr[:, v]).norm()) for v in views)
        lim = 1.3 * max(settings.tanfovx, settings.tanfovy)
        shift = fx * (ang * (1.0 + lim * lim) + dt / 1.0)                    # (scene depths >= 1 m)
        union = pt.OwnedSet(params, t_ref, settings, first_w2c, band, margin_px=32.0 + 1.5 * shift, with_centre_rows=True)
        ex = pt.OwnerExchange(union, H, rank, world)
        owner_stats["phases"] += 1; owner_stats["halo_rows"] += ex.halo_rows; owner_stats["own_rows"] += int(ex.own_rows.numel())
        return ex

    if world > 1:
        dgr.defer_run_ahead_overflow(True)          # nobody raises alone: pt.phase_overflows at the end of every phase

    def owned_done():
        if world > 1:
            pt.phase_overflows(device=dev)
        if not use_owned:
            return
        esc_t = torch.zeros(1, dtype=torch.float32, device=dev)
        for o in owned_sets.values():
            esc_t += o.escapes
        esc = int(pt.all_reduce_sum(esc_t).item())          # every rank learns of an escape on any rank (and stops with it)
        owned_sets.clear()
        owned_stats["escapes"] += esc
        if esc:
            raise SystemExit(f"bench_slam.py: {esc} Gaussians escaped an owned set within one phase (margin too small)")

    def render_pair(params, t_idx, gaussians_grad, camera_grad, tile_rows=None, contract=False):
        # contract: the caller feeds both images to the fused tracking / mapping loss (whole frame or band), which treats the
        # [z, 1, z^2] render as get_loss does -- fused.render_frame(get_loss_contract=True), DESIGN.md 7 row f2'
        if args.fused:
            return render_frame(params, t_idx, settings, first_w2c, gaussians_grad, camera_grad, tile_rows=tile_rows,
                                owned=owned_for(params, t_idx) if tile_rows is not None else None, get_loss_contract=contract)
        tg = sc.transform_to_frame(params, t_idx, gaussians_grad=gaussians_grad, camera_grad=camera_grad)
        rv = sc.transformed_params2rendervar(params, tg)
        dv = sc.transformed_params2depthplussilhouette(params, first_w2c, tg)
        rast = dgr.GaussianRasterizer(raster_settings=settings)
        im, radius, _ = rast(**rv)
        if args.shared_geometry:
            depth_sil, _ = rast.render_shared(dv["colors_precomp"], like=(rv["means3D"], rv["means2D"], rv["opacities"],
                                                                          rv["scales"], rv["rotations"]))
        else:
            depth_sil, _, _ = dgr.GaussianRasterizer(raster_settings=settings)(**dv)
        return im, depth_sil, radius

    # ground-truth observations
    gts = []
    with torch.no_grad():
        for t in range(T):
            p = dict(obs_params, cam_unnorm_rots=gt_rots, cam_trans=gt_trans)
            im, ds, _ = render_pair(p, t, False, False)
            sil = ds[1]
            depth = torch.where(sil > 0.5, ds[0] / sil.clamp(min=1e-6), torch.zeros_like(sil))[None]
            gts.append((im.clone(), depth.clone()))

    # the map being optimised: same geometry, perturbed appearance; poses unknown except frame 0
    g = torch.Generator().manual_seed(1)
    params = {
        "means3D": torch.nn.Parameter(gt_params["means3D"].clone()),
        "rgb_colors": torch.nn.Parameter((gt_params["rgb_colors"] + 0.05 * torch.randn(N, 3, generator=g).to(dev)).clamp(0, 1)),
        "unnorm_rotations": torch.nn.Parameter(gt_params["unnorm_rotations"].clone()),
        "logit_opacities": torch.nn.Parameter(torch.full((N, 1), 2.0, device=dev)),
        "log_scales": torch.nn.Parameter(gt_params["log_scales"].clone()),
        "cam_unnorm_rots": torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0], device=dev).reshape(1, 4, 1).repeat(1, 1, T)),
        "cam_trans": torch.nn.Parameter(torch.zeros(1, 3, T, device=dev)),
    }
    track_lrs = dict(means3D=0.0, rgb_colors=0.0, unnorm_rotations=0.0, logit_opacities=0.0, log_scales=0.0,
                     cam_unnorm_rots=0.0004, cam_trans=0.002)
    map_lrs = dict(means3D=0.0, rgb_colors=0.0025, unnorm_rotations=0.0, logit_opacities=0.05, log_scales=0.005,
                   cam_unnorm_rots=1e-8, cam_trans=1e-7)

    # the reference's global set: K earlier ("fixed") submaps that cover the same view, concatenated IN FRONT of the current
    # one every iteration (concat_global, src/vtgaussian_slam.py:944-977, called at :2510 and again at :2734 after every step)
    fixed = []
    for k in range(args.global_submaps):
        sk, _ = go.view_tied_scene(N, W, H, seed=100 + k)
        fixed.append({"means3D": sk["means3D"].to(dev), "rgb_colors": sk["colors_precomp"].to(dev),
                      "unnorm_rotations": sk["rotations"].to(dev), "logit_opacities": torch.full((N, 1), 2.0, device=dev),
                      "log_scales": torch.log(sk["scales"][:, :1]).to(dev)})
    variables_global = {"max_2D_radius": torch.zeros(N * (1 + len(fixed)), device=dev),
                        "means2D_gradient_accum": torch.zeros(N * (1 + len(fixed)), device=dev),
                        "denom": torch.zeros(N * (1 + len(fixed)), device=dev)}

    def concat_global():
        out = {k: torch.cat([f[k] for f in fixed] + [params[k]], dim=0) for k in fixed[0]}
        out["cam_unnorm_rots"], out["cam_trans"] = params["cam_unnorm_rots"], params["cam_trans"]
        return out

    mirror_get_loss = None
    variables = {"max_2D_radius": torch.zeros(N, device=dev), "means2D_gradient_accum": torch.zeros(N, device=dev),
                 "denom": torch.zeros(N, device=dev)}
    if args.get_loss:
        from diff_gaussian_rasterization.get_loss import get_loss as mirror_get_loss

    def curr_data(t):                             # what the driver hands to get_loss for frame t (tracking_curr_data / iter_data of the driver)
        return {"cam": settings, "im": gts[t][0], "depth": gts[t][1], "id": t, "w2c": first_w2c}

    def capture(fn):
        """fn() -> loss (forward + loss of one iteration).  Two eager warm-up iterations on the capture stream (they move no
        parameter), then the capture of forward + backward; returns (graph, static loss tensor)."""
        for _ in range(2):
            for v in params.values():
                v.grad = None
            fn().backward()
        for v in params.values():
            v.grad = None
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
            loss = fn()
            loss.backward()
        return g, loss

    def pose_error(t):
        with torch.no_grad():
            dt = (params["cam_trans"][0, :, t] - gt_trans[0, :, t]).norm().item() * 100          # cm
            q1 = torch.nn.functional.normalize(params["cam_unnorm_rots"][0, :, t], dim=0)
            dq = (q1 * gt_rots[0, :, t]).sum().abs().clamp(max=1.0)
            return dt, math.degrees(2 * math.acos(dq.item()))

    # untimed warm-up (kernel selection / compilation inside MIOpen, allocator pools, capacity hints): a few iterations
    # of each phase on a throw-away copy of the parameters
    warm = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    for _ in range(3):
        im, depth_sil, _ = render_pair(warm, 1, gaussians_grad=False, camera_grad=True, tile_rows=band, contract=True)
        track_loss(im, depth_sil, gts[1][0], gts[1][1], 0.99).backward()
        im, depth_sil, _ = render_pair(warm, 1, gaussians_grad=True, camera_grad=False, tile_rows=band, contract=True)
        map_loss(im, depth_sil, gts[1][0], gts[1][1]).backward()
    del warm
    owned_done()

    track_ms, map_ms, errs_before, errs_after = [], [], [], []
    dens_added, dens_ms = [], []
    redone = {"tracking": 0, "mapping": 0}           # N ranks: phases redone from their snapshot after a deferred overflow
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    frame_kind, frame_s, second_frac = [], [], []
    for t in range(1, T):
        if t == WU + 1 and WU:                     # the clock starts here: drop what the warm-up frames recorded
            torch.cuda.synchronize()
            track_ms, map_ms, errs_before, errs_after, frame_kind, frame_s, second_frac = [], [], [], [], [], [], []
            dens_added, dens_ms = [], []
            t_all = time.perf_counter()
        torch.cuda.synchronize(); t_frame = time.perf_counter()
        gt_im, gt_depth = gts[t]
        with torch.no_grad():                     # forward-propagate the previous pose (constant-position prior)
            params["cam_unnorm_rots"][..., t] = params["cam_unnorm_rots"][..., t - 1]
            params["cam_trans"][..., t] = params["cam_trans"][..., t - 1]
        errs_before.append(pose_error(t))
        # ---- tracking
        # N ranks: a run-ahead overflow is recorded, not raised (every rank must take the same decision), and the phase's
        # optimizer keeps consuming that iteration's invalid gradients until the end of the phase: the phase is redone ONCE from
        # a snapshot of what its optimizer moves (partition.PhaseSnapshot); the clock covers both attempts
        snap = pt.PhaseSnapshot(params, ("cam_unnorm_rots", "cam_trans")) if world > 1 else None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for attempt in (0, 1):
          opt = make_adam([{"params": [v], "name": k, "lr": track_lrs[k]} for k, v in params.items()])
          sil_thres, best = 0.99, (float("inf"), None, None)
          mse_ls, thr_ls = [], []
          graph = None
          for it in range(args.tracking_iters):
              if args.graph and it >= 1:
                  if graph is None:                  # iteration 0 (the threshold sweep reads the device) ran eagerly; capture 1..
                      def track_fn(it=it):
                          return mirror_get_loss(params, curr_data(t), variables, t, {"im": 0.5, "depth": 0.025}, True, 0.99, True,
                                                 False, tracking=True, plot_dir=None, visualize_tracking_loss=False,
                                                 tracking_iteration=it, dataset_name="replica",
                                                 presence_sil_mask_mse_ls=mse_ls, sil_thres_ls=thr_ls)[0]
                      loss = _losses = None          # (no reference to an earlier iteration's autograd graph survives)
                      graph, loss = capture(track_fn)
                  graph.replay()
                  if it % 10 == 9 and loss.item() < best[0]:
                      best = (loss.item(), params["cam_unnorm_rots"][..., t].clone(), params["cam_trans"][..., t].clone())
                  opt.step()                         # eager: one launch; the replay overwrites the gradients, no zero_grad
                  continue
              if args.get_loss:                      # the call of src/vtgaussian_slam.py:1803-1806, argument for argument
                  loss, variables, _losses, mse_ls, thr_ls = mirror_get_loss(
                      params, curr_data(t), variables, t, {"im": 0.5, "depth": 0.025}, True, 0.99, True, False, tracking=True,
                      plot_dir=None, visualize_tracking_loss=False, tracking_iteration=it, dataset_name="replica",
                      presence_sil_mask_mse_ls=mse_ls, sil_thres_ls=thr_ls)
              else:
                  im, depth_sil, _ = render_pair(params, t, gaussians_grad=False, camera_grad=True, tile_rows=band, contract=True)
                  if it == 0:
                      sil_thres = pick_threshold(im, depth_sil[1], gt_im, gt_depth)
                  loss = track_loss(im, depth_sil, gt_im, gt_depth, sil_thres)
              loss.backward()
              if world > 1:                          # the pose gradient of the frame = the sum over the bands (7 floats)
                  pose_grad = torch.cat([params["cam_unnorm_rots"].grad.reshape(-1), params["cam_trans"].grad.reshape(-1)])
                  pt.all_reduce_sum(pose_grad)
                  nq = params["cam_unnorm_rots"].grad.numel()
                  params["cam_unnorm_rots"].grad.copy_(pose_grad[:nq].view_as(params["cam_unnorm_rots"].grad))
                  params["cam_trans"].grad.copy_(pose_grad[nq:].view_as(params["cam_trans"].grad))
              with torch.no_grad():
                  lv = loss.detach()
                  if it % 10 == 9 or it == 0:       # the reference keeps the best pose; checking it costs a host sync
                      if world > 1:
                          lv = pt.all_reduce_sum(lv.clone().reshape(1))
                      if lv.item() < best[0]:
                          best = (lv.item(), params["cam_unnorm_rots"][..., t].clone(), params["cam_trans"][..., t].clone())
              opt.step(); opt.zero_grad(set_to_none=True)
          if graph is not None:
              torch.cuda.synchronize()
              dgr.check_captured()
              del graph
              dgr.forget_captured()
              opt.zero_grad(set_to_none=True)
          try:
              owned_done()
              break
          except pt.RunAheadOverflow:
              if attempt:
                  raise
              redone["tracking"] += 1
              snap.restore()
        torch.cuda.synchronize(); track_ms.append((time.perf_counter() - t0) * 1e3 / args.tracking_iters)
        errs_after.append(pose_error(t))
        # ---- densification (the reference's add_new_gaussians_base_frame, ordinary frames only: :2366-2375)
        if getattr(args, "densify", False) and world == 1 and not (args.base_frame_every > 0 and t % args.base_frame_every == 0):
            torch.cuda.synchronize(); td0 = time.perf_counter()
            with torch.no_grad():
                # forward only, under the tracked pose (the reference renders [z, 1, z^2] alone, :747; the fused route's dual pass
                # renders the colour image beside it)
                _im, ds, _r = render_pair(params, t, gaussians_grad=False, camera_grad=False)
                sil, rd, gd = ds[1], ds[0], gts[t][1][0]
                err = (gd - rd).abs() * (gd > 0)
                hole = (sil < 0.5) | ((rd > gd) & (err > 50 * err.median()))          # :748-756
                idx = (hole & (gd > 0)).reshape(-1).nonzero().reshape(-1)               # (a host read, like :758's torch.sum(...) > 0)
                if idx.numel():
                    fxy = W / 2.0                                                       # the synthetic camera of SURVEY 8d
                    px, py = (idx % W).float(), (idx // W).float()
                    dense = idx[idx % 12 == 0]                                          # the 2 x grid: one hole pixel in twelve, four sub-pixels each
                    if dense.numel():
                        off = torch.tensor([[-0.25, -0.25], [0.25, -0.25], [-0.25, 0.25], [0.25, 0.25]], device=dev)
                        dx = ((dense % W).float()[:, None] + off[None, :, 0]).reshape(-1)
                        dy = ((dense // W).float()[:, None] + off[None, :, 1]).reshape(-1)
                        px, py = torch.cat([px, dx]), torch.cat([py, dy])
                        src = torch.cat([idx, dense.repeat_interleave(4)])
                        half = torch.cat([torch.ones(idx.numel(), device=dev), torch.full((4 * dense.numel(),), 0.5, device=dev)])
                    else:
                        src, half = idx, torch.ones(idx.numel(), device=dev)
                    z = gd.reshape(-1)[src] * 1.005                                     # get_pointcloud, :76-121 (factor 1.005)
                    pts_cam = torch.stack([(px - (W / 2.0 - 0.5) + 0.5) / fxy * z, (py - (H / 2.0 - 0.5) + 0.5) / fxy * z, z,
                                           torch.ones_like(z)], 1)
                    w2c_t = torch.eye(4, device=dev)
                    w2c_t[:3, :3] = sc.build_rotation(torch.nn.functional.normalize(params["cam_unnorm_rots"][..., t].detach()))
                    w2c_t[:3, 3] = params["cam_trans"][0, :, t].detach()
                    new = {"means3D": (torch.inverse(w2c_t) @ pts_cam.T).T[:, :3].contiguous(),
                           "rgb_colors": gts[t][0].reshape(3, -1).T[src].contiguous(),
                           "unnorm_rotations": torch.tensor([1.0, 0, 0, 0], device=dev).repeat(src.numel(), 1),
                           "logit_opacities": torch.zeros(src.numel(), 1, device=dev),      # initialize_new_params, :692-720
                           "log_scales": torch.log(z / fxy * half)[:, None]}
                    for k, v in new.items():
                        params[k] = torch.nn.Parameter(torch.cat([params[k].detach(), v], 0))
                    n_now = params["means3D"].shape[0]
                    variables = {k: torch.zeros(n_now, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom")}
                    variables_global = {k: torch.zeros(n_now + N * len(fixed), device=dev) for k in variables}
                    dens_added.append(src.numel())
                else:
                    dens_added.append(0)
            torch.cuda.synchronize(); dens_ms.append((time.perf_counter() - td0) * 1e3)
        # ---- mapping
        # Which view an iteration renders, and whether it makes the second call over the global set (module docstring of
        # --base-frame-every).  The draw is the reference's np.random.randint over the submap's frames so far; the same seed
        # on every rank of the N-rank loop.
        E = args.base_frame_every
        is_base = E > 0 and t % E == 0
        w0 = (t // E) * E if E > 0 else t                 # first frame of the current submap's window (its base frame)
        window = list(range(w0, t + 1))
        rng = np.random.RandomState(1000 + t)
        frame_kind.append("base" if (is_base or E == 0) else "ordinary")
        second_calls = 0
        snap = pt.PhaseSnapshot(params, [k for k, lr in map_lrs.items() if lr != 0.0]) if world > 1 else None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for attempt in (0, 1):
          opt = make_adam([{"params": [v], "name": k, "lr": map_lrs[k]} for k, v in params.items()], lr=0.0, eps=1e-15)
          rng = np.random.RandomState(1000 + t)           # (a redone phase draws the same keyframes)
          second_calls = 0
          graph = None
          exchange = make_exchange(params, t, window) if use_owner else None
          owner_checked = set()
          for it in range(args.mapping_iters):
              if E > 0 and not is_base and args.emulate_window > 0:
                  draw = rng.randint(0, args.emulate_window)
                  kf = w0 if draw == 0 else (window[1 + (draw - 1) % (len(window) - 1)] if len(window) > 1 else w0)
                  second = bool(fixed) and kf == w0 and kf % E == 0
              elif E > 0 and not is_base:
                  kf = window[rng.randint(0, len(window))]
                  second = bool(fixed) and kf == w0 and kf % E == 0
              else:
                  kf, second = t, bool(fixed)
              second_calls += int(second)
              gt_im, gt_depth = gts[kf]
              if args.graph:
                  if graph is None:
                      def map_fn():
                          loss = mirror_get_loss(params, curr_data(t), variables, t, {"im": 1.0, "depth": 1.0}, False, 0.99, True,
                                                 False, mapping=True, dataset_name="replica")[0]
                          if fixed:
                              loss = loss + mirror_get_loss(concat_global(), curr_data(t), variables_global, t,
                                                            {"im": 1.0, "depth": 1.0}, False, 0.99, True, False, mapping=True,
                                                            dataset_name="replica")[0]
                          return loss
                      loss = _losses = None
                      graph, loss = capture(map_fn)
                  graph.replay()
                  opt.step()
                  continue
              if args.get_loss:                      # (the mapping loops call it the same way, without the threshold lists)
                  loss, variables, _losses = mirror_get_loss(params, curr_data(kf), variables, kf, {"im": 1.0, "depth": 1.0}, False,
                                                             0.99, True, False, mapping=True, dataset_name="replica")
                  if second:                         # the second call of the reference's mapping iteration (:2551-2556, :2600-2604)
                      loss_global, variables_global, _lg = mirror_get_loss(
                          concat_global(), curr_data(kf), variables_global, kf, {"im": 1.0, "depth": 1.0}, False, 0.99, True, False,
                          mapping=True, dataset_name="replica")
                      loss = loss + loss_global
              else:
                  im, depth_sil, _ = render_pair(params, kf, gaussians_grad=True, camera_grad=False, tile_rows=band, contract=True)
                  loss = map_loss(im, depth_sil, gt_im, gt_depth)
              loss.backward()
              if exchange is not None:               # halo gradients -> owner bands; Adam on the owned rows; updated rows -> listers
                  if kf not in owner_checked:
                      owner_checked.add(kf)
                      if not exchange.covers(owned_sets[kf]):
                          raise SystemExit(f"bench_slam.py: the list of view {kf} is not inside the phase's union list")
                  owner_stats["bytes_sent"] += exchange.reduce_grads(params)
                  opt.step(rows=exchange.update_rows)
                  owner_stats["bytes_sent"] += exchange.publish(params)
                  owner_stats["iterations"] += 1
                  opt.zero_grad(set_to_none=True)
                  continue
              if world > 1:                          # trainable per-Gaussian gradients: one flat all-reduce, 20 B per Gaussian
                  pt.allreduce_param_grads(params)
              opt.step(); opt.zero_grad(set_to_none=True)
          if graph is not None:
              torch.cuda.synchronize()
              dgr.check_captured()
              del graph
              dgr.forget_captured()
              opt.zero_grad(set_to_none=True)
          if exchange is not None:
              exchange.gather_all(params)            # every rank gets every owner's rows back before the next phase
          try:
              owned_done()
              break
          except pt.RunAheadOverflow:
              if attempt:
                  raise
              redone["mapping"] += 1
              snap.restore()
        torch.cuda.synchronize(); map_ms.append((time.perf_counter() - t0) * 1e3 / args.mapping_iters)
        frame_s.append(time.perf_counter() - t_frame)
        second_frac.append(second_calls / max(args.mapping_iters, 1))
        if os.environ.get("VTGS_SLAM_SPLAT_SIZES") and t % int(os.environ["VTGS_SLAM_SPLAT_SIZES"]) == 0:
            # diagnostic (round 6): how many 8x8-tile instances each Gaussian has under the current view -- through the plain
            # operator over the same bins (the fused route keeps no handle on its workspace)
            with torch.no_grad():
                tg = sc.transform_to_frame(params, t, gaussians_grad=False, camera_grad=False)
                rv = sc.transformed_params2rendervar(params, tg)
                rast = dgr.GaussianRasterizer(raster_settings=settings)
                _c, rad, _d = rast(**rv)
                offs, gid, _geom = dgr.debug_tile_lists(rast)
                per = torch.bincount(gid, minlength=rad.numel())
                edges = [0, 1, 2, 3, 5, 7, 9, 13, 17, 33, 65, 257, 1 << 30]
                hist = {f"{a}..{b - 1}": int(((per >= a) & (per < b)).sum()) for a, b in zip(edges[:-1], edges[1:])}
                r = rad.float().cpu()
                sc_now = torch.exp(params["log_scales"].detach()).reshape(-1).cpu()
                print(f"[bench_slam] frame {t}: N {rad.numel()}, instances {int(per.sum())}, per-Gaussian instance histogram {hist}; "
                      f"radius p50 {r.quantile(0.5):.0f} p99 {r.quantile(0.99):.0f} p99.9 {r.quantile(0.999):.0f} max {r.max():.0f}; "
                      f"scale ratio to the first frame's median: p50 {float(sc_now.median() / scene['scales'][:, 0].median()):.2f} "
                      f"p99 {float(sc_now.quantile(0.99) / scene['scales'][:, 0].median()):.2f} max {float(sc_now.max() / scene['scales'][:, 0].median()):.2f}",
                      file=sys.stderr, flush=True)
        if os.environ.get("VTGS_SLAM_VERBOSE"):
            print(f"[bench_slam] frame {t}: tracking {track_ms[-1]:.3f} ms/it, mapping {map_ms[-1]:.3f} ms/it, "
                  f"instances {dgr.last_forward_info().get('instances')}, longest tile list {dgr.last_forward_info().get('max_tile_list')}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    total = time.perf_counter() - t_all
    map_ms_frames = list(map_ms)
    if world > 1:                                  # the slowest rank's clock
        tmax = torch.tensor([total, sum(track_ms) / len(track_ms), sum(map_ms) / len(map_ms)], dtype=torch.float64)
        if backend == "nccl":
            tmax = tmax.to(dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        total = float(tmax[0])
        track_ms, map_ms = [float(tmax[1])], [float(tmax[2])]
    regimes = None
    if args.base_frame_every:
        if world > 1:                              # the slowest rank's clock, frame by frame
            fs = torch.tensor(frame_s, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(fs, op=dist.ReduceOp.MAX)
            frame_s = [float(x) for x in fs]
        mean = lambda xs: sum(xs) / len(xs) if xs else None
        ord_s = [x for x, k in zip(frame_s, frame_kind) if k == "ordinary"]
        base_s = [x for x, k in zip(frame_s, frame_kind) if k == "base"]
        regimes = {
            "ordinary_frame": {"frames": len(ord_s), "s_per_frame": mean(ord_s),
                               "mapping_ms_per_iter": mean([m for m, k in zip(map_ms_frames, frame_kind) if k == "ordinary"]),
                               "second_get_loss_call_frac": mean([f for f, k in zip(second_frac, frame_kind) if k == "ordinary"]),
                               "mapping_view": "one keyframe drawn per iteration from the submap's frames so far"},
            "base_frame": {"frames": len(base_s), "s_per_frame": mean(base_s),
                           "mapping_ms_per_iter": mean([m for m, k in zip(map_ms_frames, frame_kind) if k == "base"]),
                           "second_get_loss_call_frac": mean([f for f, k in zip(second_frac, frame_kind) if k == "base"]),
                           "mapping_view": "the current frame, both get_loss calls every iteration" if fixed else "the current frame"},
        }
        if ord_s and base_s:                       # configs/replica/room0.py:35: one base frame in 40
            regimes["frames_per_s_mix_39_to_1"] = round(40.0 / (39.0 * mean(ord_s) + mean(base_s)), 3)
    out = {
        "metric": "SLAM frames/s, tracking+mapping loop (synthetic Replica-room0-like sequence)",
        "value": round(args.frames / total, 3), "unit": "frames/s", "n_gpus": world, "higher_is_better": True,
        "data": "synthetic", "dtype": "f32",
        "config": {"workload": f"view-tied submap N={N}, {W}x{H}; {args.tracking_iters} tracking + {args.mapping_iters} "
                               f"mapping iterations per frame, 2 renders fwd+bwd per iteration (configs/replica/room0.py)",
                   "frames": args.frames, "warmup_frames_not_counted": WU, "shared_geometry": bool(args.shared_geometry or args.fused), "fused_callers": bool(args.fused),
                   "autograd_backward_on": "calling thread" if caller_thread else "engine device thread (PyTorch default)",
                   "through_get_loss_mirror": bool(args.get_loss), "iteration_replayed_from_a_hipgraph": bool(args.graph),
                   "mapping_get_loss_calls_per_iteration": (2 if fixed else 1) if not args.base_frame_every else "see regimes",
                   "base_frame_every": args.base_frame_every or None, "emulated_window_frames": args.emulate_window or None,
                   "gaussians_in_global_set": N * (1 + len(fixed)) if fixed else None,
                   "partition": "none" if world == 1 else f"tile-row bands x{world} ({backend}): all-reduce of the pose gradient "
                                                          "(tracking); SSIM halo rows + 8 loss sums + 20 B/Gaussian (mapping)"},
        "tracking_ms_per_iter": round(sum(track_ms) / len(track_ms), 3),
        "mapping_ms_per_iter": round(sum(map_ms) / len(map_ms), 3),
        "pose_error_before_tracking_cm_deg": [[round(a, 3), round(b, 4)] for a, b in errs_before],
        "pose_error_after_tracking_cm_deg": [[round(a, 3), round(b, 4)] for a, b in errs_after],
        "regimes": regimes,
        "densification": None if not getattr(args, "densify", False) else {
            "frames": len(dens_added), "gaussians_added_per_frame_mean": round(sum(dens_added) / max(len(dens_added), 1), 1),
            "ms_per_frame_mean": round(sum(dens_ms) / max(len(dens_ms), 1), 2), "gaussians_at_the_end": int(params["means3D"].shape[0]),
            "what": "forward-only render + depth_error.median() + back-projection + append, every ordinary frame "
                    "(src/vtgaussian_slam.py:732-813)"},
        "phases_redone_after_a_run_ahead_overflow": None if world == 1 else redone,
        "owned_sets": None if not use_owned else {
            "lists_built": owned_stats["built"], "mean_listed_fraction_of_map": round(owned_stats["listed"] / max(owned_stats["built"], 1) / N, 4),
            "escapes": owned_stats["escapes"], "margin_px": 32.0, "scale_growth": 1.25, "rank": rank},
        "mapping_exchange": None if world == 1 else (
            {"route": "all-reduce of 20 B per Gaussian, Adam on every row", "bytes_per_iteration": 20 * N} if not use_owner else
            {"route": "owner bands: halo gradients -> owner, Adam on owned rows, updated rows -> listers; one all-reduce per phase",
             "bytes_sent_per_iteration_this_rank": round(owner_stats["bytes_sent"] / max(owner_stats["iterations"], 1)),
             "all_reduce_bytes_per_iteration_for_comparison": 20 * N,
             "halo_rows_mean": round(owner_stats["halo_rows"] / max(owner_stats["phases"], 1)),
             "own_rows_mean": round(owner_stats["own_rows"] / max(owner_stats["phases"], 1))}),
    }
    ms = torch.cuda.memory_stats()
    out["allocator"] = {"device_allocs": ms.get("num_device_alloc", 0), "device_frees": ms.get("num_device_free", 0),
                        "alloc_retries": ms.get("num_alloc_retries", 0),
                        "reserved_gb": round(torch.cuda.memory_reserved() / 1e9, 2)}
    return out


def main():
    out = run(parse_args())
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
