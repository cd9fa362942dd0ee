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
