"""Generates tests/golden/helpers_*.npz by IMPORTING the reference's own helper modules in the build
container (they never travel to the GPU box; only the .npz files do).

    python tests/golden/make_helper_fixtures.py          # needs /root/reference

What is captured (inputs and outputs only -- no reference source text):
  * utils/slam_external.py:25-42   build_rotation
  * utils/slam_helpers.py:24-31    quat_mult
  * utils/slam_helpers.py:46-106   matrix_to_quaternion
  * utils/slam_helpers.py:323-385  transform_to_frame          (fwd + grads to cam_unnorm_rots / cam_trans)
  * utils/slam_helpers.py:127-160  transformed_params2rendervar
  * utils/slam_helpers.py:255-287  transformed_params2depthplussilhouette  (and 217-234 via it)
  * utils/slam_external.py:66-97   calc_ssim ; :49-51 calc_psnr ; utils/slam_helpers.py:5-21 L1 losses
The reference hard-codes device='cuda' (utils/slam_external.py:28, utils/slam_helpers.py:122,229,348,371);
this script -- and only this script -- maps those to CPU.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def install_cpu_shim():
    torch.Tensor.cuda = lambda self, *a, **k: self
    for name in ("zeros", "ones", "eye", "zeros_like", "ones_like", "tensor", "empty"):
        orig = getattr(torch, name)

        def wrap(*a, __orig=orig, **k):
            if "device" in k and str(k["device"]).startswith("cuda"):
                k["device"] = "cpu"
            return __orig(*a, **k)
        setattr(torch, name, wrap)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted; fixtures can only be generated in the build container")
    sys.dont_write_bytecode = True
    install_cpu_shim()
    sys.path.insert(0, REF)
    from utils import slam_external as se
    from utils import slam_helpers as sh

    g = torch.Generator().manual_seed(20250614)
    rnd = lambda *s: torch.randn(*s, generator=g)

    # ---- quaternion / rotation helpers -------------------------------------------------------
    q = rnd(64, 4)
    rot = se.build_rotation(q)
    q1, q2 = torch.nn.functional.normalize(rnd(32, 4)), torch.nn.functional.normalize(rnd(32, 4))
    qm = sh.quat_mult(q1, q2)
    Rm = se.build_rotation(torch.nn.functional.normalize(rnd(48, 4)))
    m2q = sh.matrix_to_quaternion(Rm)
    np.savez_compressed(os.path.join(OUT, "helpers_quat.npz"), q=q.numpy(), build_rotation=rot.numpy(),
                        q1=q1.numpy(), q2=q2.numpy(), quat_mult=qm.numpy(), R=Rm.numpy(),
                        matrix_to_quaternion=m2q.numpy())

    # ---- pose transform + render-variable builders (isotropic, as every config uses) -----------
    n, T = 257, 5
    params = {
        "means3D": torch.nn.Parameter(rnd(n, 3) * 2 + torch.tensor([0.0, 0.0, 4.0])),
        "rgb_colors": torch.nn.Parameter(torch.rand(n, 3, generator=g)),
        "unnorm_rotations": torch.nn.Parameter(torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1)),
        "logit_opacities": torch.nn.Parameter(rnd(n, 1)),
        "log_scales": torch.nn.Parameter(rnd(n, 1) * 0.3 - 4.0),
        "cam_unnorm_rots": torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1).repeat(1, 1, T)
                                              + 0.05 * rnd(1, 4, T)),
        "cam_trans": torch.nn.Parameter(0.1 * rnd(1, 3, T)),
    }
    t_idx = 3
    w2c0 = torch.eye(4)
    w2c0[:3, :3] = se.build_rotation(torch.nn.functional.normalize(rnd(1, 4)))[0]
    w2c0[:3, 3] = 0.2 * rnd(3)
    tg = sh.transform_to_frame(params, t_idx, gaussians_grad=True, camera_grad=True)
    rv = sh.transformed_params2rendervar(params, tg)
    dv = sh.transformed_params2depthplussilhouette(params, w2c0, tg)
    wm, wc = rnd(n, 3), rnd(n, 3)
    loss = (rv["means3D"] * wm).sum() + (dv["colors_precomp"] * wc).sum()
    loss.backward()
    np.savez_compressed(
        os.path.join(OUT, "helpers_transform.npz"), time_idx=np.int64(t_idx), first_frame_w2c=w2c0.numpy(),
        wm=wm.numpy(), wc=wc.numpy(),
        **{"in_" + k: v.detach().numpy() for k, v in params.items()},
        **{"rgb_" + k: v.detach().numpy() for k, v in rv.items()},
        **{"dep_" + k: v.detach().numpy() for k, v in dv.items()},
        grad_cam_unnorm_rots=params["cam_unnorm_rots"].grad.numpy(), grad_cam_trans=params["cam_trans"].grad.numpy(),
        grad_means3D=params["means3D"].grad.numpy())

    # ---- image losses ------------------------------------------------------------------------
    a, b = torch.rand(3, 40, 56, generator=g), torch.rand(3, 40, 56, generator=g)
    mask = torch.rand(3, 40, 56, generator=g) > 0.3
    np.savez_compressed(os.path.join(OUT, "helpers_losses.npz"), a=a.numpy(), b=b.numpy(), mask=mask.numpy(),
                        ssim=se.calc_ssim(a, b).numpy(), psnr=se.calc_psnr(a, b).numpy(),
                        l1=sh.l1_loss_v1(a, b).numpy(), l1_mask=sh.l1_loss_v1_mask(a, b, mask).numpy())

    # ---- camera records for the five BASELINE resolutions (utils/recon_helpers.py:5-13 maths; that
    #      module itself cannot be imported -- it needs the absent rasterizer package -- so these are
    #      evaluated from the intrinsics of configs/data/*.yaml with the same torch ops) -------------
    cams = {}
    for name, (w, h, fx, fy, cx, cy) in {
            "synthetic": (320, 240, 160.0, 160.0, 159.5, 119.5),
            "replica": (1200, 680, 600.0, 600.0, 599.5, 339.5),
            "tum_fr1": (640, 480, 517.3, 516.5, 318.6, 255.3),
            "scannet": (640, 480, 577.59, 578.73, 318.9, 242.68),
            "scannetpp": (1752, 1168, 1371.3, 1371.3, 876.0, 584.0)}.items():
        w2c = torch.eye(4)
        view = w2c.unsqueeze(0).transpose(1, 2)
        near, far = 0.01, 100
        proj = torch.tensor([[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0],
                             [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
                             [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
                             [0.0, 0.0, 1.0, 0.0]]).float().unsqueeze(0).transpose(1, 2)
        cams[name + "_whk"] = np.array([w, h, fx, fy, cx, cy], dtype=np.float64)
        cams[name + "_view"] = view.numpy()
        cams[name + "_proj"] = view.bmm(proj).numpy()
    np.savez_compressed(os.path.join(OUT, "cameras.npz"), **cams)
    print("wrote", [f for f in os.listdir(OUT) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
