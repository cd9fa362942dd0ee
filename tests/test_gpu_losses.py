"""fused_ssim (vtgs_ssim_*) against the SSIM value captured from the reference's calc_ssim and against the PyTorch
restatement (value and gradient) on odd image sizes."""
import os

import numpy as np
import pytest
import torch

import slam_callers as sc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_ssim_matches_reference_fixture(gpu_device):
    from diff_gaussian_rasterization.losses import fused_ssim
    d = np.load(os.path.join(G, "helpers_losses.npz"))
    a, b = torch.from_numpy(d["a"]).to(gpu_device), torch.from_numpy(d["b"]).to(gpu_device)
    np.testing.assert_allclose(fused_ssim(a, b).item(), float(d["ssim"]), rtol=2e-5)


@pytest.mark.parametrize("shape", [(3, 40, 56), (3, 97, 131), (1, 33, 31), (3, 680, 1200)])
def test_ssim_value_and_gradient_match_conv_restatement(gpu_device, shape):
    from diff_gaussian_rasterization.losses import fused_ssim
    g = torch.Generator().manual_seed(shape[1])
    a = torch.rand(*shape, generator=g).to(gpu_device).requires_grad_(True)
    b = (a.detach() + 0.1 * torch.randn(*shape, generator=g).to(gpu_device)).clamp(0, 1)
    ref = sc.calc_ssim(a, b)
    ref.backward()
    gref = a.grad.clone()
    a.grad = None
    out = fused_ssim(a, b)
    (out * 1.0).backward()
    assert abs(out.item() - ref.item()) <= 2e-5 * abs(ref.item())
    assert (a.grad - gref).abs().max().item() <= 1e-3 * gref.abs().max().item()


@pytest.mark.parametrize("mode", ["tracking", "mapping"])
def test_fused_slam_losses_match_the_restatement(gpu_device, mode):
    """tracking_loss / mapping_loss kernels vs slam_callers (itself pinned by reference fixtures): value and gradients,
    with invalid depth, NaNs in the render and low-silhouette pixels present."""
    from diff_gaussian_rasterization import losses
    dev = gpu_device
    g = torch.Generator().manual_seed(11)
    H, W = 97, 131
    im = torch.rand(3, H, W, generator=g).to(dev).requires_grad_(True)
    z = torch.rand(H, W, generator=g) + 1
    sil = torch.rand(H, W, generator=g) * 0.2 + 0.85
    ds = torch.stack([z, sil, z * z + 0.01 * torch.rand(H, W, generator=g)])
    ds[0, 5, 7] = float("nan"); ds[2, 9, 3] = float("nan")
    ds = ds.to(dev).requires_grad_(True)
    gt_im = torch.rand(3, H, W, generator=g).to(dev)
    gt_depth = (z + 0.2 * torch.randn(H, W, generator=g))[None].clone()
    gt_depth[0, :4, :] = 0
    gt_depth = gt_depth.to(dev)
    if mode == "tracking":
        ref = sc.tracking_loss(im, ds, gt_im, gt_depth, 0.95)
        fn = lambda: losses.tracking_loss(im, ds, gt_im, gt_depth, 0.95)
    else:
        ref = sc.mapping_loss(im, ds, gt_im, gt_depth)
        fn = lambda: losses.mapping_loss(im, ds, gt_im, gt_depth)
    ref.backward()
    r_im, r_ds = im.grad.clone(), torch.nan_to_num(ds.grad.clone())
    im.grad = None; ds.grad = None
    out = fn()
    out.backward()
    assert abs(out.item() - ref.item()) <= 2e-5 * abs(ref.item())
    assert (im.grad - r_im).abs().max().item() <= 1e-3 * r_im.abs().max().item() + 1e-9
    assert (torch.nan_to_num(ds.grad) - r_ds).abs().max().item() <= 1e-5 * r_ds.abs().max().item() + 1e-9
