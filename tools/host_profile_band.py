"""cProfile of the host side of one band iteration (fused frame render, forward + backward) -- where the ~0.5 ms per iteration
of a rank of the tile-row partition go once the kernels have shrunk to its band.  python tools/host_profile_band.py [owned]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vtgaussian-slam_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import partition as pt
    from diff_gaussian_rasterization.fused import render_frame
    from oracle import gs_oracle as go            # scene generator only
    from parity_util import to_settings
    dev = torch.device("cuda", 0)
    N, W, H = 1_000_000, 1200, 680
    scene, cam = go.view_tied_scene(N, W, H, seed=0)
    st, w2c = to_settings(cam, dev), torch.eye(4, device=dev)
    params = {"means3D": scene["means3D"], "rgb_colors": scene["colors_precomp"], "unnorm_rotations": scene["rotations"],
              "logit_opacities": torch.full((N, 1), 2.0), "log_scales": torch.log(scene["scales"][:, :1]),
              "cam_unnorm_rots": torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1).repeat(1, 1, 2), "cam_trans": torch.zeros(1, 3, 2)}
    params = {k: torch.nn.Parameter(v.to(dev)) for k, v in params.items()}
    band = pt.band_for_rank(H, 8, 3)
    own = pt.OwnedSet(params, 1, st, w2c, band) if "owned" in sys.argv else None
    g1 = torch.ones(3, H, W, device=dev)

    def one():
        for v in params.values():
            v.grad = None
        im, ds, _ = render_frame(params, 1, st, w2c, False, True, tile_rows=band, owned=own)
        (im * g1).sum().backward()

    for _ in range(20):
        one()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        one()
    pr.disable()
    torch.cuda.synchronize()
    dgr.settle_pending()
    ps = pstats.Stats(pr)
    ps.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
