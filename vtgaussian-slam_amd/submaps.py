"""Resident submaps: the reference's per-base-frame bookkeeping without the per-frame
GPU -> CPU -> GPU shuttle.

The reference keeps the map as a list of per-base-frame parameter dicts (`params_ls`, `variables_ls`) and, at the end of
EVERY frame, moves every tensor of every submap to the host (src/vtgaussian_slam.py:2832-2843); the next frame uploads the
selected ones again with `.cuda()` inside `concat_keyframes_params_base_frame` (:913, :929) and `concat_global` (:950).  At
~1 M Gaussians x 56 B per submap that is ~60 MB each way per submap and frame over PCIe -- and an MI355X has 288 GB of HBM
for a map of a few GB.  This module mirrors the reference's four functions by name and behaviour, device-agnostically (the
tensors stay wherever they are), plus `keep_resident`, the replacement of the shuttle loop.

Pinned by tests/golden/driver_helpers.npz, captured by running the reference's own functions
(tests/golden/make_get_loss_fixtures.py).  Host-side logic only: no kernels, testable without a GPU.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch

GAUSSIAN_KEYS = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")
VARIABLE_KEYS = ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")


def quantize_selected_time_idx(selected_time_idx: Sequence[int], num_frames_each_base_frame: int) -> List[int]:
    """Frame indices -> distinct base-frame (submap) indices (src/vtgaussian_slam.py:884-894; same `set` order)."""
    return list(set(int(idx / num_frames_each_base_frame) for idx in selected_time_idx))


def _f32(v, device):
    t = v if isinstance(v, torch.Tensor) else torch.tensor(v)
    return t.to(device=device, dtype=torch.float32).contiguous()


def concat_keyframes_params_base_frame(params_ls, variables_ls, selected_time_idx, num_frames_each_base_frame, device=None):
    """src/vtgaussian_slam.py:900-941: concatenate the selected submaps' Gaussians (and per-Gaussian statistics) into one
    optimisable parameter dict; camera tensors come from the last selected submap.  `device=None`: wherever the first
    selected submap lives (resident submaps are already on the GPU: the concatenation is a device-side copy) -- except that
    host-resident submaps go to the GPU when there is one, as in the reference; pass device="cpu" to keep them there."""
    q = quantize_selected_time_idx(selected_time_idx, num_frames_each_base_frame)
    dev = device or params_ls[q[0]]["means3D"].device
    if device is None and torch.device(dev).type == "cpu" and torch.cuda.is_available():
        dev = torch.device("cuda")             # the reference forces .cuda() here (:913): host-resident submaps are uploaded
    num_gs = [params_ls[idx]["means3D"].shape[0] for idx in q]
    params = {k: torch.cat([_f32(params_ls[idx][k], dev) for idx in q], dim=0) for k in GAUSSIAN_KEYS}
    params["cam_unnorm_rots"] = params_ls[q[-1]]["cam_unnorm_rots"]
    params["cam_trans"] = params_ls[q[-1]]["cam_trans"]
    params = {k: torch.nn.Parameter(_f32(v.detach() if isinstance(v, torch.Tensor) else v, dev).requires_grad_(True))
              for k, v in params.items()}
    variables = {k: torch.cat([_f32(variables_ls[idx][k], dev) for idx in q], dim=0) for k in VARIABLE_KEYS}
    variables["scene_radius"] = variables_ls[q[-1]]["scene_radius"]
    return params, variables, num_gs


def concat_global(cat_params, cat_variables, cat_num_gs_per_frame=None, global_params=None, global_variables=None):
    """src/vtgaussian_slam.py:944-977: prepend the fixed global Gaussians to the concatenated local ones."""
    dev = cat_params["means3D"].device
    params = {k: torch.cat((_f32(global_params[k], dev), _f32(cat_params[k], dev)), dim=0)
              for k in GAUSSIAN_KEYS if k in global_params}
    params["cam_unnorm_rots"] = cat_params["cam_unnorm_rots"]
    params["cam_trans"] = cat_params["cam_trans"]
    variables = {k: torch.cat((_f32(global_variables[k], dev), _f32(cat_variables[k], dev)), dim=0)
                 for k in VARIABLE_KEYS if k in global_variables}
    variables["scene_radius"] = cat_variables["scene_radius"]
    if cat_num_gs_per_frame is not None:
        return params, variables, [global_params["means3D"].shape[0]] + list(cat_num_gs_per_frame)
    return params, variables


def update_params_ls(params_ls, selected_time_idx, cat_params, num_gs_per_frame, num_frames_each_base_frame):
    """src/vtgaussian_slam.py:980-1004: split the optimised concatenation back into the per-submap dicts (views of the
    concatenated tensors, as in the reference: no copy)."""
    split = {k: torch.split(cat_params[k], list(num_gs_per_frame), dim=0) for k in GAUSSIAN_KEYS if k in cat_params}
    for i, idx in enumerate(quantize_selected_time_idx(selected_time_idx, num_frames_each_base_frame)):
        for k in list(params_ls[idx].keys()):
            if k in split:
                params_ls[idx][k] = split[k][i]
    for param in params_ls:                    # (the reference compares against 'cam_unnorm_rot' -- sic -- so only cam_trans is shared)
        if "cam_trans" in param:
            param["cam_trans"] = cat_params["cam_trans"]
    return params_ls


def update_variables_ls(variables_ls, selected_time_idx, cat_variables, num_gs_per_frame, num_frames_each_base_frame):
    """src/vtgaussian_slam.py:1007-1020."""
    split = {k: torch.split(cat_variables[k], list(num_gs_per_frame), dim=0) for k in VARIABLE_KEYS if k in cat_variables}
    for i, idx in enumerate(quantize_selected_time_idx(selected_time_idx, num_frames_each_base_frame)):
        for k in list(variables_ls[idx].keys()):
            if k in split:
                variables_ls[idx][k] = split[k][i]
    return variables_ls


def keep_resident(params_ls: List[Dict], variables_ls: List[Dict]) -> Tuple[int, int]:
    """Replaces the end-of-frame shuttle (src/vtgaussian_slam.py:2832-2843): detach the tensors (the reference's
    `.detach().cpu()` also cuts the graph) and LEAVE THEM WHERE THEY ARE.  Views produced by update_params_ls are turned
    into owning tensors so that the big concatenation of the frame can be freed.  Returns (bytes resident on devices,
    bytes on the host) for memory accounting against the 288 GB of an MI355X."""
    on_dev = on_host = 0
    for group in (params_ls, variables_ls):
        for d in group:
            for k, v in d.items():
                if isinstance(v, torch.Tensor):
                    # a torch.split piece shares the storage of the whole concatenation (detach() of a view reports
                    # _base None, so compare sizes): copy it out, or every submap would pin -- and alias -- that buffer
                    owns = v.untyped_storage().nbytes() == v.numel() * v.element_size()
                    t = v.detach() if owns else v.detach().clone()
                    d[k] = t
                    nbytes = t.numel() * t.element_size()
                    if t.device.type == "cpu":
                        on_host += nbytes
                    else:
                        on_dev += nbytes
    return on_dev, on_host
