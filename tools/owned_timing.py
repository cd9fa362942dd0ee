"""What an owned set buys one rank of the tile-row partition: the SLAM iteration's render (fused frame, forward + backward) on
band `--rank` of `--world`, with the per-Gaussian kernels over the whole map (round 3) and over the list (partition.OwnedSet).

    python tools/owned_timing.py [--n 1000000] [--world 8] [--rank 3] [--iters 40]

Prints one JSON line: per-iteration milliseconds of both routes for the tracking and the mapping form of the iteration, the
per-kernel averages of the library's own event timers, the list's size, what building and checking it cost."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vtgaussian-slam_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=680)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--caller-thread-autograd", action="store_true", help="torch.autograd.set_multithreading_enabled(False)")
    args = ap.parse_args()
    if args.caller_thread_autograd:
        torch.autograd.set_multithreading_enabled(False)
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import partition as pt
    from diff_gaussian_rasterization.fused import render_frame
    from oracle import gs_oracle as go            # scene generator only
    from parity_util import to_settings
    dev = torch.device("cuda", 0)
    N, W, H = args.n, args.width, args.height
    scene, cam = go.view_tied_scene(N, W, H, seed=0)
    st, w2c = to_settings(cam, dev), torch.eye(4, device=dev)
    params = {"means3D": scene["means3D"], "rgb_colors": scene["colors_precomp"], "unnorm_rotations": scene["rotations"],
              "logit_opacities": torch.full((N, 1), 2.0), "log_scales": torch.log(scene["scales"][:, :1]),
              "cam_unnorm_rots": torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1).repeat(1, 1, 2), "cam_trans": torch.zeros(1, 3, 2)}
    params = {k: torch.nn.Parameter(v.to(dev)) for k, v in params.items()}
    band = pt.band_for_rank(H, args.world, args.rank)
    g1, g2 = torch.ones(3, H, W, device=dev), torch.ones(3, H, W, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    own = pt.OwnedSet(params, 1, st, w2c, band)
    torch.cuda.synchronize(); build_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    own2 = pt.OwnedSet(params, 1, st, w2c, band)
    torch.cuda.synchronize(); build_ms = min(build_ms, (time.perf_counter() - t0) * 1e3)
    del own2

    def one(owned, gaussians_grad, camera_grad):
        for v in params.values():
            v.grad = None
        im, ds, _ = render_frame(params, 1, st, w2c, gaussians_grad, camera_grad, tile_rows=band, owned=owned)
        ((im * g1).sum() + (ds * g2).sum()).backward()

    out = {"workload": f"N={N}, {W}x{H}, band {band} = rank {args.rank} of {args.world}", "listed": len(own),
           "listed_fraction": round(len(own) / N, 4), "build_ms": round(build_ms, 3)}
    for phase, (gg, cg) in (("warm", (False, True)), ("tracking", (False, True)), ("mapping", (True, False)), ("tracking_again", (False, True))):
        for name, owned in (("whole_map", None), ("owned_list", own)):
            for _ in range(5):
                one(owned, gg, cg)
            torch.cuda.synchronize()
            dgr.settle_pending()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                one(owned, gg, cg)
            e1.record(); torch.cuda.synchronize()
            dgr.settle_pending()
            dgr.profile_enable(True)
            for _ in range(5):
                one(owned, gg, cg)
            prof = dgr.profile_collect()
            dgr.profile_enable(False)
            info = dgr.last_forward_info()
            key = (0, N if owned is None else len(own), W, H, band)
            ra = dgr._async_ok.get(key) is not None and dgr._async_ok.get(key) == dgr._choose_capacities(key, key[1])
            out[f"{phase}_{name}"] = {"run_ahead": ra, "caps": [int(x) for x in dgr._choose_capacities(key, key[1])], "instances": info.get("instances"), "max_tile_list": info.get("max_tile_list"), "ms_per_iter": round(e0.elapsed_time(e1) / args.iters, 4),
                                      "kernels_us": {k: round(v[0] / v[1] * 1e3, 1) for k, v in prof.items()}}
    camc = dgr._Camera(st, dev, dgr._RADIUS_RULES["3sigma"], band)
    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    a = (camc, f32(params["means3D"]), f32(params["log_scales"]), f32(params["cam_unnorm_rots"][0, :, 1]),
         f32(params["cam_trans"][0, :, 1]))
    dgr.profile_enable(True)
    for _ in range(20):
        own.check(*a)
    prof = dgr.profile_collect()
    dgr.profile_enable(False)
    out["check_us"] = {k: round(v[0] / v[1] * 1e3, 1) for k, v in prof.items()}
    out["escaped"] = own.escaped()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
