"""Shared comparison helpers for the GPU parity tests (HIP path vs oracle/gs_oracle.py)."""
import numpy as np
import torch

from oracle import gs_oracle as go

GRAD_KEYS = ["means3D", "means2D", "opacities", "colors_precomp", "scales", "rotations"]


def to_settings(cam, device, bg=None):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=(cam.bg if bg is None else bg).to(device), scale_modifier=cam.scale_modifier,
        viewmatrix=cam.viewmatrix.to(device), projmatrix=cam.projmatrix.to(device), sh_degree=0,
        campos=cam.campos.to(device), prefiltered=False)


def run_oracle(scene, cam, grad_color=None, dtype=torch.float64, radius_rule="3sigma", tile_rows=None):
    leaves = {k: v.detach().to(dtype).clone().requires_grad_(grad_color is not None) for k, v in scene.items()}
    color, radii, depth, aux = go.rasterize(cam=cam, radius_rule=radius_rule, tile_rows=tile_rows, return_aux=True, **leaves)
    grads = None
    if grad_color is not None:
        (color * grad_color.to(dtype)).sum().backward()
        grads = {k: leaves[k].grad for k in GRAD_KEYS}
    return color.detach(), radii, depth.detach(), grads, aux


def run_hip(scene, cam, device, grad_color=None, radius_rule=None, tile_rows=None, bg=None):
    from diff_gaussian_rasterization import GaussianRasterizer
    leaves = {k: v.detach().to(device=device, dtype=torch.float32).clone().requires_grad_(grad_color is not None)
              for k, v in scene.items()}
    rast = GaussianRasterizer(raster_settings=to_settings(cam, device, bg), radius_rule=radius_rule, tile_rows=tile_rows)
    color, radii, depth = rast(**leaves)
    grads = None
    if grad_color is not None:
        (color * grad_color.to(device)).sum().backward()
        grads = {k: leaves[k].grad.cpu() for k in GRAD_KEYS}
    return color.detach().cpu(), radii.cpu(), depth.detach().cpu(), grads


def image_error(ref, got):
    """max |diff| relative to the image's max magnitude, and the fraction of pixels above 1e-4 of it."""
    ref, got = ref.double(), got.double()
    scale = ref.abs().max().item() + 1e-12
    d = (ref - got).abs() / scale
    return d.max().item(), (d > 1e-4).double().mean().item()


def grad_error(ref, got):
    """(max |diff| / max |ref|,  99.9th percentile of element-wise relative error with a 1e-3*max floor)."""
    ref, got = ref.double(), got.double()
    scale = ref.abs().max().item()
    if scale == 0:
        return got.abs().max().item(), 0.0
    d = (ref - got).abs()
    rel = d / (ref.abs() + 1e-3 * scale)
    return (d.max() / scale).item(), torch.quantile(rel.reshape(-1)[:4_000_000], 0.999).item()
