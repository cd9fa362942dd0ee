"""The two halves of the operator surface the reference never uses ("must raise if both / neither of shs /
colors_precomp or of scales + rotations / cov3D_precomp are given"): spherical-harmonics colours and a precomputed 3-D covariance.

`shs` [N, K, 3] (K >= (sh_degree + 1)^2, sh_degree and campos from the settings record): a per-Gaussian pre-op in front of the
rasterizer (csrc/vtgs_sh.hip) turns them into colours, its backward turns dL/dcolours into dL/dshs and the viewing direction's
share of dL/dmeans3D; the rasterizer's kernels see colours.  `cov3D_precomp` [N, 6]: vtgs_forward_cov3d / vtgs_backward_cov3d
(the projection and gather kernels compiled with the covariance read from memory instead of built from scale + rotation).
Both are checked forwards through this Python node -- the run-ahead policy and the C++ node serve the reference's signature.
"""
import ctypes

import torch

from . import (_P, _SZ, _U64, _I32, VTGS_ERR_INSTANCE_OVERFLOW, VTGS_FORWARD_CHECKED, PLANNED, _Camera, _ForwardState, _VtgsCamera,
               _check, _device_guard, _lib, _require, _scratch, _slot_lock, _slot_pool, _stream_ptr, _tile_capacity_for, _workspace)

_lib.vtgs_sh_forward.restype, _lib.vtgs_sh_forward.argtypes = ctypes.c_int, [_I32, _I32, _I32, _P, _P, _P, _P, _P, _P]
_lib.vtgs_sh_backward.restype, _lib.vtgs_sh_backward.argtypes = ctypes.c_int, [_I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P]
_lib.vtgs_forward_cov3d.restype = ctypes.c_int
_lib.vtgs_forward_cov3d.argtypes = [ctypes.POINTER(_VtgsCamera), _I32, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _U64, ctypes.c_uint32, _P,
                                    ctypes.c_uint32, _P]
_lib.vtgs_backward_cov3d.restype = ctypes.c_int
_lib.vtgs_backward_cov3d.argtypes = [ctypes.POINTER(_VtgsCamera), _I32, _P, _P, _P, _P, _P, _P, _P, _SZ, _U64, ctypes.c_uint32, _P, _SZ,
                                     _P, _P, _P, _P, _P, _P]


class _SHColors(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, shs, campos, degree: int):
        dev, n = means3D.device, means3D.shape[0]
        if shs.dim() != 3 or shs.shape[0] != n or shs.shape[2] != 3:
            raise ValueError("shs must be [N, K, 3]")
        K = int(shs.shape[1])
        if not 0 <= degree <= 3 or K < (degree + 1) ** 2 or K > 16:
            raise ValueError(f"sh_degree {degree} needs (degree + 1)^2 <= K <= 16 coefficients, shs has {K}")
        m = _require(means3D, "means3D", 3, n, dev)
        s = shs.detach().to(torch.float32).contiguous()
        c = campos.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        colors = torch.empty((n, 3), dtype=torch.float32, device=dev)
        clamped = torch.empty((n,), dtype=torch.uint8, device=dev)
        with _device_guard(dev):
            _check(_lib.vtgs_sh_forward(n, degree, K, m.data_ptr(), c.data_ptr(), s.data_ptr(), colors.data_ptr(), clamped.data_ptr(),
                                        _stream_ptr(dev)), "vtgs_sh_forward")
        ctx.save_for_backward(m, s, c, clamped)
        ctx.degree, ctx.K = degree, K
        return colors

    @staticmethod
    def backward(ctx, g_colors):
        m, s, c, clamped = ctx.saved_tensors
        dev, n = m.device, m.shape[0]
        g = g_colors.to(torch.float32).contiguous()
        g_shs = torch.empty_like(s) if ctx.needs_input_grad[1] else None
        g_m = torch.empty_like(m) if ctx.needs_input_grad[0] else None
        if g_shs is not None or g_m is not None:
            ptr = lambda t: None if t is None else t.data_ptr()
            with _device_guard(dev):
                _check(_lib.vtgs_sh_backward(n, ctx.degree, ctx.K, m.data_ptr(), c.data_ptr(), s.data_ptr(), clamped.data_ptr(), g.data_ptr(),
                                             ptr(g_shs), ptr(g_m), _stream_ptr(dev)), "vtgs_sh_backward")
        return g_m, g_shs, None, None


def sh_colors(means3D: torch.Tensor, shs: torch.Tensor, campos: torch.Tensor, degree: int) -> torch.Tensor:
    """colours [N, 3] of the operator's `shs` argument (differentiable w.r.t. means3D and shs)."""
    if not means3D.is_cuda:
        raise RuntimeError("GaussianRasterizer needs tensors on a HIP device (torch 'cuda'); no CPU path exists")
    return _SHColors.apply(means3D, shs, campos, int(degree))


class _RasterizeCov3D(torch.autograd.Function):
    """(means3D, means2D, colors_precomp, opacities, cov3D_precomp) -> (color, radii, depth); five gradients back."""

    @staticmethod
    def forward(ctx, means3D, means2D, colors_precomp, opacities, cov3D, cam: _Camera):
        dev, n = means3D.device, means3D.shape[0]
        means3D = _require(means3D, "means3D", 3, n, dev)
        colors = _require(colors_precomp, "colors_precomp", 3, n, dev)
        opac = _require(opacities, "opacities", 1, n, dev)
        cov = _require(cov3D, "cov3D_precomp", 6, n, dev)
        H, W = cam.H, cam.W
        stream = _stream_ptr(dev)
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((n,), dtype=torch.int32, device=dev)
        capacity, tile_cap = 8 * n + 65536, 512
        fs = _ForwardState()
        with _slot_lock, _device_guard(dev):
            pool = _slot_pool(dev, stream)
            slot = pool.take(fs)
            info = pool.info[slot]
            for _attempt in range(8):
                info.complete = 0
                nbytes, ws = _workspace(n, W, H, capacity, tile_cap, dev)
                st = _lib.vtgs_forward_cov3d(ctypes.byref(cam.c), n, means3D.data_ptr(), colors.data_ptr(), opac.data_ptr(), cov.data_ptr(),
                                             color.data_ptr(), depth.data_ptr(), radii.data_ptr(), ws.data_ptr(), nbytes, capacity, tile_cap,
                                             pool.ptr[slot], VTGS_FORWARD_CHECKED, stream)
                if st == VTGS_ERR_INSTANCE_OVERFLOW:          # uniform bins only on this path: grow whichever was short
                    if info.overflow & 1:
                        capacity = int(info.instances_needed * 1.5) + 4096
                    if info.overflow & 2:
                        tile_cap = _tile_capacity_for(info.max_tile_list)
                    continue
                _check(st, "vtgs_forward_cov3d")
                break
            else:
                raise RuntimeError("vtgs_forward_cov3d: instance capacity kept overflowing")
            pool.owner[slot] = None
            fs.cam, fs.n, fs.workspace, fs.capacity, fs.tile_cap, fs.image_state, fs.key, fs.pending = cam, n, ws, capacity, tile_cap, None, None, None
            fs._instances = int(info.instances_needed)
        ctx.fs = fs
        ctx.save_for_backward(means3D, colors, opac, cov, color)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii, depth)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, *unused):
        means3D, colors, opac, cov, color = ctx.saved_tensors
        fs, dev, n = ctx.fs, means3D.device, means3D.shape[0]
        g = (torch.zeros_like(color) if grad_color is None else grad_color).to(torch.float32).contiguous()
        need = ctx.needs_input_grad
        new = lambda w: torch.empty((n, w), dtype=torch.float32, device=dev)
        g_means3D, g_means2D = (new(3) if need[0] else None), (new(3) if need[1] else None)
        g_colors, g_opac, g_cov = (new(3) if need[2] else None), (new(1) if need[3] else None), (new(6) if need[4] else None)
        if n > 0 and any(need[:5]):
            sbytes = _lib.vtgs_backward_scratch_bytes(n, max(fs._instances, 1))
            scratch = _scratch(sbytes, dev)
            ptr = lambda t: None if t is None else t.data_ptr()
            with _device_guard(dev):
                _check(_lib.vtgs_backward_cov3d(ctypes.byref(fs.cam.c), n, means3D.data_ptr(), colors.data_ptr(), opac.data_ptr(),
                                                cov.data_ptr(), color.data_ptr(), g.data_ptr(), fs.workspace.data_ptr(), fs.workspace.numel(),
                                                fs.capacity, fs.tile_cap, scratch.data_ptr(), sbytes, ptr(g_means3D), ptr(g_means2D),
                                                ptr(g_colors), ptr(g_opac), ptr(g_cov), _stream_ptr(dev)), "vtgs_backward_cov3d")
        return g_means3D, g_means2D, g_colors, g_opac, g_cov, None


def rasterize_cov3d(cam: _Camera, means3D, means2D, colors_precomp, opacities, cov3D_precomp):
    if means2D is None:
        means2D = torch.zeros_like(means3D)
    return _RasterizeCov3D.apply(means3D, means2D, colors_precomp, opacities, cov3D_precomp, cam)
