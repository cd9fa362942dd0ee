"""The three modes of vtgs_forward through the bare C ABI (include/vtgs.h): SYNC waits for the stream, ASYNC returns at
once and leaves the result record in pinned memory, CHECKED (what the package uses) waits for the record only.  Same
images, same radii, same record -- with the sort and the finalize step inside the forward composite (default) and with
their own launches (VTGS_SORT_FUSED=0)."""
import ctypes

import pytest
import torch

from oracle import gs_oracle as go
from parity_util import to_settings

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [1, 0])
def test_forward_modes_agree(gpu_device, fused):
    import diff_gaussian_rasterization as dgr
    lib = dgr._lib
    dev = gpu_device
    dgr.set_option("VTGS_SORT_FUSED", fused)
    scene, cam = go.view_tied_scene(20000, 200, 136, seed=41)
    t = {k: v.to(dev).contiguous() for k, v in scene.items()}
    n, W, H = 20000, 200, 136
    camobj = dgr._camera_for(to_settings(cam, dev), dev, 0, None)
    cap, tcap = 8 * n + 65536, 512
    nbytes = lib.vtgs_workspace_bytes(n, W, H, cap, tcap)
    stream = torch.cuda.current_stream(dev).cuda_stream
    res = {}
    for mode in (dgr.VTGS_FORWARD_SYNC, dgr.VTGS_FORWARD_ASYNC, dgr.VTGS_FORWARD_CHECKED):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        color = torch.full((3, H, W), float("nan"), device=dev)
        depth = torch.full((1, H, W), float("nan"), device=dev)
        radii = torch.full((n,), -7, dtype=torch.int32, device=dev)
        slot = torch.zeros(64, dtype=torch.uint8).pin_memory()
        info = dgr._VtgsForwardInfo.from_address(slot.data_ptr())
        st = lib.vtgs_forward(ctypes.byref(camobj.c), n, t["means3D"].data_ptr(), t["colors_precomp"].data_ptr(),
                              t["opacities"].data_ptr(), t["scales"].data_ptr(), t["rotations"].data_ptr(), color.data_ptr(),
                              depth.data_ptr(), radii.data_ptr(), ws.data_ptr(), nbytes, cap, tcap, slot.data_ptr(), mode, stream)
        assert st == 0, lib.vtgs_strerror(st)
        if mode == dgr.VTGS_FORWARD_ASYNC:
            torch.cuda.synchronize()                              # the record is the caller's to wait for
        assert info.complete == 1 and info.overflow == 0
        torch.cuda.synchronize()
        res[mode] = (color.cpu(), depth.cpu(), radii.cpu(), (info.instances, info.instances_needed, info.visible, info.max_tile_list))
        assert torch.isfinite(res[mode][0]).all() and int(res[mode][2].min()) >= 0
    a = res[dgr.VTGS_FORWARD_SYNC]
    for mode in (dgr.VTGS_FORWARD_ASYNC, dgr.VTGS_FORWARD_CHECKED):
        b = res[mode]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3]
    assert a[3][0] == a[3][1] > 0 and a[3][2] > 0 and a[3][3] > 0


def test_overflow_is_reported_in_every_mode(gpu_device):
    """Capacities that are too small: every mode reports VTGS_ERR_INSTANCE_OVERFLOW (SYNC, CHECKED) or the overflow bits in
    the record (ASYNC) together with the needs, and nothing faults -- also with the finalize step inside the composite."""
    import diff_gaussian_rasterization as dgr
    lib = dgr._lib
    dev = gpu_device
    scene, cam = go.view_tied_scene(20000, 200, 136, seed=41)
    t = {k: v.to(dev).contiguous() for k, v in scene.items()}
    n, W, H = 20000, 200, 136
    camobj = dgr._camera_for(to_settings(cam, dev), dev, 0, None)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for cap, tcap, bit in ((1000, 512, 1), (8 * n, 64, 2)):
        nbytes = lib.vtgs_workspace_bytes(n, W, H, cap, tcap)
        for mode in (dgr.VTGS_FORWARD_SYNC, dgr.VTGS_FORWARD_ASYNC, dgr.VTGS_FORWARD_CHECKED):
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            color, depth = torch.empty(3, H, W, device=dev), torch.empty(1, H, W, device=dev)
            radii = torch.empty(n, dtype=torch.int32, device=dev)
            slot = torch.zeros(64, dtype=torch.uint8).pin_memory()
            info = dgr._VtgsForwardInfo.from_address(slot.data_ptr())
            st = lib.vtgs_forward(ctypes.byref(camobj.c), n, t["means3D"].data_ptr(), t["colors_precomp"].data_ptr(),
                                  t["opacities"].data_ptr(), t["scales"].data_ptr(), t["rotations"].data_ptr(), color.data_ptr(),
                                  depth.data_ptr(), radii.data_ptr(), ws.data_ptr(), nbytes, cap, tcap, slot.data_ptr(), mode, stream)
            torch.cuda.synchronize()
            if mode == dgr.VTGS_FORWARD_ASYNC:
                assert st == 0
            else:
                assert st == dgr.VTGS_ERR_INSTANCE_OVERFLOW
            assert info.complete == 1 and (info.overflow & bit)
            assert info.instances_needed > 0 and info.max_tile_list > 0


def test_planned_forward_through_the_bare_abi(gpu_device):
    """vtgs_forward_planned as a C caller drives it: a uniform plan that is too small for some bins reports overflow bit 2
    (and nothing faults -- every bin is clamped to the workspace, even under a garbage plan); the forward has rewritten the
    plan from its exact list lengths, so the repeat fits, reports the same needs and gives vtgs_forward's image."""
    import diff_gaussian_rasterization as dgr
    lib = dgr._lib
    dev = gpu_device
    scene, cam = go.view_tied_scene(20000, 200, 136, seed=41)
    t = {k: v.to(dev).contiguous() for k, v in scene.items()}
    n, W, H = 20000, 200, 136
    camobj = dgr._camera_for(to_settings(cam, dev), dev, 0, None)
    stream = torch.cuda.current_stream(dev).cuda_stream
    cap = 8 * n + 65536
    args = (ctypes.byref(camobj.c), n, t["means3D"].data_ptr(), t["colors_precomp"].data_ptr(), t["opacities"].data_ptr(),
            t["scales"].data_ptr(), t["rotations"].data_ptr())

    def forward(tcap, plan):
        nbytes = lib.vtgs_workspace_bytes(n, W, H, cap, tcap)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        color, depth = torch.full((3, H, W), float("nan"), device=dev), torch.full((1, H, W), float("nan"), device=dev)
        radii = torch.empty(n, dtype=torch.int32, device=dev)
        slot = torch.zeros(64, dtype=torch.uint8).pin_memory()
        info = dgr._VtgsForwardInfo.from_address(slot.data_ptr())
        if plan is None:
            st = lib.vtgs_forward(*args, color.data_ptr(), depth.data_ptr(), radii.data_ptr(), ws.data_ptr(), nbytes, cap, tcap,
                                  slot.data_ptr(), dgr.VTGS_FORWARD_SYNC, stream)
        else:
            st = lib.vtgs_forward_planned(*args, color.data_ptr(), depth.data_ptr(), radii.data_ptr(), ws.data_ptr(), nbytes, cap,
                                          tcap, plan.data_ptr(), slot.data_ptr(), dgr.VTGS_FORWARD_SYNC, stream)
        torch.cuda.synchronize()
        return st, color.cpu(), depth.cpu(), (int(info.instances), int(info.instances_needed), int(info.max_tile_list),
                                               int(info.overflow), int(info.bin_slots_needed))
    st, ref_c, ref_d, ref_info = forward(512, None)
    assert st == 0 and ref_info[3] == 0 and ref_info[4] > 0
    tiles = (W + 7) // 8 * ((H + 7) // 8)
    assert lib.vtgs_bin_plan_entries(W, H) == tiles + 1
    plan = torch.empty(tiles + 1, dtype=torch.int32, device=dev)
    per_bin = 32                                                     # too few for the longest lists of this scene
    assert ref_info[2] > per_bin
    assert lib.vtgs_bin_plan_uniform(W, H, per_bin, plan.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    assert plan.cpu().tolist() == [per_bin * i for i in range(tiles + 1)]
    avg = -(-ref_info[4] // tiles) + 1                               # workspace: what the lists need in total (+ rounding)
    st, _, _, info1 = forward(dgr.PLANNED | avg, plan)
    assert st == dgr.VTGS_ERR_INSTANCE_OVERFLOW and info1[3] == 2 and info1[1] == ref_info[1] and info1[4] == ref_info[4]
    new_plan = plan.cpu().long()
    assert int(new_plan[-1]) == ref_info[4] and bool((new_plan[1:] >= new_plan[:-1]).all())
    st, c, d, info2 = forward(dgr.PLANNED | avg, plan)               # the plan the failed attempt left behind fits
    assert st == 0 and info2[:3] == ref_info[:3] and info2[3] == 0
    assert torch.equal(c, ref_c) and torch.equal(d, ref_d)
    # a garbage plan (decreasing, far beyond the workspace): an overflow report, no fault, and a usable plan afterwards
    plan.copy_(torch.randint(0, 2 ** 31 - 1, (tiles + 1,), dtype=torch.int32))
    st, _, _, info3 = forward(dgr.PLANNED | avg, plan)
    assert st == dgr.VTGS_ERR_INSTANCE_OVERFLOW and info3[3] & 2
    st, c, d, info4 = forward(dgr.PLANNED | avg, plan)
    assert st == 0 and torch.equal(c, ref_c) and torch.equal(d, ref_d)
