"""f4, host side: resident-submap bookkeeping (vtgaussian-slam_amd/submaps.py) against the reference's own
concat / split functions (src/vtgaussian_slam.py:884-1020), captured in tests/golden/driver_helpers.npz."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "vtgaussian-slam_amd"))
import submaps  # noqa: E402

KEYS5 = list(submaps.GAUSSIAN_KEYS)
VKEYS = list(submaps.VARIABLE_KEYS)


@pytest.fixture(scope="module")
def fx():
    z = np.load(os.path.join(HERE, "golden", "driver_helpers.npz"))
    return {k: z[k] for k in z.files}


def _lists(fx):
    params_ls, variables_ls = [], []
    for i in range(4):
        params_ls.append({k: torch.from_numpy(fx[f"in{i}_{k}"]) for k in KEYS5 + ["cam_unnorm_rots", "cam_trans"]})
        variables_ls.append({k: torch.from_numpy(fx[f"in{i}_{k}"]) for k in VKEYS + ["scene_radius"]})
    return params_ls, variables_ls


def test_concat_split_equal_the_reference(fx):
    params_ls, variables_ls = _lists(fx)
    sel, nfe = fx["selected_time_idx"].tolist(), int(fx["num_frames_each_base_frame"])
    assert submaps.quantize_selected_time_idx(sel, nfe) == fx["quantized"].tolist()
    cat_p, cat_v, num_gs = submaps.concat_keyframes_params_base_frame(params_ls, variables_ls, sel, nfe, device="cpu")
    assert num_gs == fx["cat_num_gs"].tolist()
    for k in KEYS5 + ["cam_unnorm_rots", "cam_trans"]:
        assert isinstance(cat_p[k], torch.nn.Parameter) and cat_p[k].requires_grad
        assert np.array_equal(cat_p[k].detach().numpy(), fx["cat_p_" + k]), k
    for k in VKEYS + ["scene_radius"]:
        assert np.array_equal(np.asarray(cat_v[k]), fx["cat_v_" + k]), k
    glob_p = {k: torch.from_numpy(fx["glob_" + k]) for k in KEYS5}
    glob_v = {k: torch.from_numpy(fx["glob_" + k]) for k in VKEYS}
    gp, gv, gnum = submaps.concat_global(cat_p, cat_v, list(num_gs), glob_p, glob_v)
    assert gnum == fx["global_num_gs"].tolist()
    for k in KEYS5 + ["cam_unnorm_rots", "cam_trans"]:
        assert np.array_equal(gp[k].detach().numpy(), fx["global_p_" + k]), k
    for k in VKEYS + ["scene_radius"]:
        assert np.array_equal(np.asarray(gv[k]), fx["global_v_" + k]), k
    new_p = {k: v.detach() + 0.5 for k, v in cat_p.items()}
    new_v = {k: (v.detach() * 2.0 if isinstance(v, torch.Tensor) else v) for k, v in cat_v.items()}
    upd_p = submaps.update_params_ls([dict(x) for x in params_ls], sel, new_p, list(num_gs), nfe)
    upd_v = submaps.update_variables_ls([dict(x) for x in variables_ls], sel, new_v, list(num_gs), nfe)
    for i in range(4):
        for k in KEYS5:
            assert np.array_equal(upd_p[i][k].numpy(), fx[f"upd{i}_{k}"]), (i, k)
        for k in VKEYS:
            assert np.array_equal(upd_v[i][k].numpy(), fx[f"upd{i}_{k}"]), (i, k)
    # the shuttle replacement: nothing moves, views become owners, the graph is cut
    on_dev, on_host = submaps.keep_resident(upd_p, upd_v)
    assert on_dev == 0 and on_host > 0
    before = {(i, k): v.clone() for i, d in enumerate(upd_p) for k, v in d.items() if isinstance(v, torch.Tensor)}
    for v in new_p.values():                       # the frame's concatenation changes afterwards (the next optimiser step) ...
        v.add_(1000.0)
    for i, d in enumerate(upd_p):                  # ... and no kept submap may notice: owners, not views of it
        for k, v in d.items():
            if isinstance(v, torch.Tensor) and k in KEYS5:
                assert torch.equal(v, before[(i, k)]), (i, k)
    for d in upd_p + upd_v:
        for k, v in d.items():
            if isinstance(v, torch.Tensor):
                assert not v.requires_grad and v.device.type == "cpu"
                if k in KEYS5 + VKEYS:
                    assert v.untyped_storage().nbytes() == v.numel() * v.element_size(), k


@pytest.mark.gpu
def test_resident_submaps_on_the_device(fx):
    """The same bookkeeping with the submaps resident on the GPU: values equal the reference's (which went through the host),
    nothing leaves the device, and keep_resident accounts the bytes to the device."""
    dev = torch.device("cuda:0")
    params_ls, variables_ls = _lists(fx)
    params_ls = [{k: v.to(dev) for k, v in d.items()} for d in params_ls]
    variables_ls = [{k: v.to(dev) for k, v in d.items()} for d in variables_ls]
    sel, nfe = fx["selected_time_idx"].tolist(), int(fx["num_frames_each_base_frame"])
    cat_p, cat_v, num_gs = submaps.concat_keyframes_params_base_frame(params_ls, variables_ls, sel, nfe)
    assert num_gs == fx["cat_num_gs"].tolist()
    for k in KEYS5 + ["cam_unnorm_rots", "cam_trans"]:
        assert cat_p[k].is_cuda and cat_p[k].requires_grad
        assert np.array_equal(cat_p[k].detach().cpu().numpy(), fx["cat_p_" + k]), k
    new_p = {k: v.detach() + 0.5 for k, v in cat_p.items()}
    new_v = {k: (v.detach() * 2.0 if isinstance(v, torch.Tensor) else v) for k, v in cat_v.items()}
    upd_p = submaps.update_params_ls([dict(x) for x in params_ls], sel, new_p, list(num_gs), nfe)
    upd_v = submaps.update_variables_ls([dict(x) for x in variables_ls], sel, new_v, list(num_gs), nfe)
    for i in range(4):
        for k in KEYS5:
            assert np.array_equal(upd_p[i][k].cpu().numpy(), fx[f"upd{i}_{k}"]), (i, k)
    on_dev, on_host = submaps.keep_resident(upd_p, upd_v)
    assert on_dev > 0 and on_host == 0
    for d in upd_p + upd_v:
        for k, v in d.items():
            if isinstance(v, torch.Tensor):
                assert v.is_cuda and not v.requires_grad
                if k in KEYS5 + VKEYS:                       # owners: the frame's concatenation can be freed
                    assert v.untyped_storage().nbytes() == v.numel() * v.element_size(), k
