"""Reads of memory nobody wrote, and capacity growth under an unchanged driver.

Round 3 met a GPU memory access fault (gpurun_out/r3/run_g.log): lanes past the end of a SHORT tile list used unwritten
workspace slots as Gaussian ids for the geometry gather (fixed in 0adb68e).  On memory the caching allocator hands back clean
such a read passes silently, so these tests POISON every workspace / scratch block with 0xFF bytes (an id of 0xFFFFFFFF is far
outside every array, a float of 0xFFFFFFFF is a NaN) and the images with NaN, and compare with a run on zero-filled memory
bit for bit.  Run once: a read through a poisoned id is a fault, not a flaky difference.
"""
import ctypes
import os

import pytest
import torch

from oracle import gs_oracle as go
from parity_util import GRAD_KEYS, to_settings

pytestmark = pytest.mark.gpu


def _short_list_scene(seed=5, W=200, H=136):
    """Empty tiles (the right half of the frame and a horizontal stripe hold nothing) and many lists of 1..63 entries: a
    view-tied scene thinned to ~0.15 Gaussians per pixel, plus a few dense tiles so that multi-chunk lists exist too."""
    scene, cam = go.view_tied_scene(W * H + 3000, W, H, seed=seed)
    g = torch.Generator().manual_seed(seed)
    m = scene["means3D"]
    u = m[:, 0] / m[:, 2] * (W / 2.0) + (W / 2.0 - 0.5) - 0.5
    v = m[:, 1] / m[:, 2] * (H / 2.0) + (H / 2.0 - 0.5) - 0.5
    keep = (torch.rand(m.shape[0], generator=g) < 0.15) & (u < 0.55 * W) & ~((v > 60) & (v < 84))
    keep |= (u > 16) & (u < 40) & (v > 16) & (v < 40)                       # a dense patch: lists of several chunks
    return {k: t[keep].contiguous() for k, t in scene.items()}, cam


def _clear_policy(dgr):
    for d in (dgr._capacity_hint, dgr._caps_in_use, dgr._tile_cap_hint, dgr._async_ok, dgr._need_hist, dgr._slots_hint,
              dgr._no_deferred, dgr._no_defer_cooldown):
        d.clear()


def _operator_step(dgr, scene, cam, dev, grad_color, tile_rows=None):
    leaves = {k: v.to(dev).requires_grad_(True) for k, v in scene.items()}
    rast = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev), tile_rows=tile_rows)
    c, r, d = rast(**leaves)
    dcol = torch.stack([leaves["means3D"][:, 2], torch.ones_like(leaves["means3D"][:, 2]), leaves["means3D"][:, 2] ** 2], 1)
    c2, d2 = rast.render_shared(dcol, like=(leaves["means3D"], leaves["means2D"], leaves["opacities"], leaves["scales"],
                                            leaves["rotations"]))
    ((c * grad_color).sum() + (c2 * grad_color.flip(0)).sum()).backward()
    dgr.settle_pending()
    lists = dgr.debug_tile_lists(rast)
    return [c.detach().cpu(), r.cpu(), d.detach().cpu(), c2.detach().cpu(), d2.detach().cpu()] + \
           [leaves[k].grad.cpu() for k in GRAD_KEYS], lists


@pytest.mark.parametrize("route", ["ext", "python"])
def test_poisoned_workspaces_change_nothing(gpu_device, route, monkeypatch):
    """Forward (fused sort, quadrant queues), shared second render, both backwards (lane = pixel) and the gather kernel on
    a scene with empty tiles and many lists shorter than one 64-entry chunk: 0xFF-filled workspace + scratch and NaN images
    give the bits of a run on zero-filled blocks -- through the C++ autograd node and through the Python one."""
    import diff_gaussian_rasterization as dgr
    dev = gpu_device
    if route == "python":
        monkeypatch.setattr(dgr, "_ext", None)
    elif dgr._ext is None:
        pytest.skip("the C++ autograd node is not built")
    scene, cam = _short_list_scene()
    g = torch.Generator().manual_seed(1)
    grad_color = (torch.rand(3, cam.image_height, cam.image_width, generator=g) * 2 - 1).to(dev)
    _clear_policy(dgr)
    try:
        dgr.poison_workspaces(False)
        torch.cuda.empty_cache()
        ref, lists = _operator_step(dgr, scene, cam, dev, grad_color)
        cnt = (lists[0][1:] - lists[0][:-1])
        assert int((cnt == 0).sum()) > 50 and int(((cnt > 0) & (cnt < 64)).sum()) > 100 and int(cnt.max()) > 128, \
            (int((cnt == 0).sum()), int(((cnt > 0) & (cnt < 64)).sum()), int(cnt.max()))
        dgr.poison_workspaces(True)
        for _ in range(2):                                     # (twice: the second run re-uses the first one's poisoned blocks)
            got, _ = _operator_step(dgr, scene, cam, dev, grad_color)
            for i, (a, b) in enumerate(zip(ref, got)):
                assert torch.equal(a, b), i
        # a band of tile rows (the row cull of the projection, the band grid of the composites)
        dgr.poison_workspaces(False)
        ref_b, _ = _operator_step(dgr, scene, cam, dev, grad_color, tile_rows=(2, 5))
        dgr.poison_workspaces(True)
        got_b, _ = _operator_step(dgr, scene, cam, dev, grad_color, tile_rows=(2, 5))
        for i, (a, b) in enumerate(zip(ref_b, got_b)):
            assert torch.equal(a, b), i
    finally:
        dgr.poison_workspaces(False)


@pytest.mark.parametrize("contract", [False, True])
def test_poisoned_workspaces_dual_render_and_frame_epilogue(gpu_device, contract):
    """The fused caller chain (dual forward, dual backward with the frame epilogue in the gather kernel) under poison.
    contract: the get_loss forms -- the single render's forward with z in its depth column, the four-channel backward with its
    12-float records (one pad word that nobody writes and nobody may read)."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import fused
    dev = gpu_device
    scene, cam = _short_list_scene(seed=8)
    n = scene["means3D"].shape[0]
    g = torch.Generator().manual_seed(2)
    H, W = cam.image_height, cam.image_width
    g_im = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)
    g_ds = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)

    def step():
        params = {
            "means3D": scene["means3D"].to(dev).requires_grad_(True),
            "rgb_colors": scene["colors_precomp"].to(dev).requires_grad_(True),
            "unnorm_rotations": scene["rotations"].to(dev).requires_grad_(True),
            "logit_opacities": torch.logit(scene["opacities"]).to(dev).requires_grad_(True),
            "log_scales": torch.log(scene["scales"][:, :1]).to(dev).requires_grad_(True),
            "cam_unnorm_rots": torch.tensor([1.0, 0.002, -0.001, 0.0015], device=dev).reshape(1, 4, 1).requires_grad_(True),
            "cam_trans": torch.tensor([0.004, -0.003, 0.002], device=dev).reshape(1, 3, 1).requires_grad_(True),
        }
        im, ds, radii = fused.render_frame(params, 0, to_settings(cam, dev), torch.eye(4, device=dev), gaussians_grad=True,
                                           camera_grad=True, get_loss_contract=contract)
        ((im * g_im).sum() + ((ds * g_ds)[:1] if contract else ds * g_ds).sum()).backward()
        dgr.settle_pending()
        return [im.detach().cpu(), ds.detach().cpu(), radii.cpu()] + [params[k].grad.cpu() for k in sorted(params)]

    _clear_policy(dgr)
    try:
        dgr.poison_workspaces(False)
        torch.cuda.empty_cache()
        ref = step()
        assert n > 0 and all(torch.isfinite(t.float()).all() for t in ref)
        dgr.poison_workspaces(True)
        got = step()
        for i, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), i
    finally:
        dgr.poison_workspaces(False)


def _abi_forward(dgr, t, cam, dev, n, W, H, cap, tcap, flags, plan=None, poison=True):
    lib = dgr._lib
    camobj = dgr._camera_for(to_settings(cam, dev), dev, 0, None)
    nbytes = lib.vtgs_workspace_bytes(n, W, H, cap, tcap)
    ws = torch.full((nbytes,), 0xFF if poison else 0, dtype=torch.uint8, device=dev)
    color = torch.full((3, H, W), float("nan"), device=dev)
    depth = torch.full((1, H, W), float("nan"), device=dev)
    radii = torch.full((n,), -7, dtype=torch.int32, device=dev)
    slot = torch.zeros(64, dtype=torch.uint8).pin_memory()
    info = dgr._VtgsForwardInfo.from_address(slot.data_ptr())
    stream = torch.cuda.current_stream(dev).cuda_stream
    args = (ctypes.byref(camobj.c), n, t["means3D"].data_ptr(), t["colors_precomp"].data_ptr(), t["opacities"].data_ptr(),
            t["scales"].data_ptr(), t["rotations"].data_ptr(), color.data_ptr(), depth.data_ptr(), radii.data_ptr(),
            ws.data_ptr(), nbytes, cap, tcap)
    if plan is None:
        st = lib.vtgs_forward(*args, slot.data_ptr(), flags, stream)
    else:
        st = lib.vtgs_forward_planned(*args, plan.data_ptr(), slot.data_ptr(), flags, stream)
    torch.cuda.synchronize()
    rec = (int(info.instances), int(info.instances_needed), int(info.max_tile_list), int(info.overflow), int(info.complete))
    return st, color.cpu(), depth.cpu(), radii.cpu(), rec, ws


def test_overflowing_bins_above_512_slots_are_not_read(gpu_device):
    """a bin of more than 512 slots whose list outgrew it is sorted by nobody (sort_long_lists and sort_tiles
    skip it), and the fused forward used to walk its `sorted_gid` slots all the same -- ids from unwritten memory, straight
    into the geometry gather, before the overflow flag could stop anything.  Uniform bins of 768 slots and planned bins of
    600 under lists of ~1,800 entries, workspace 0xFF: the run-ahead (asynchronous) forward reports the overflow, nothing
    faults, and its image is the background colour instead of leftovers."""
    import diff_gaussian_rasterization as dgr
    dev = gpu_device
    n, W, H = 60000, 96, 64
    scene, cam = go.view_tied_scene(n, W, H, seed=21)
    t = {k: v.to(dev).contiguous() for k, v in scene.items()}
    cap = 8 * n + 65536
    for mode in (dgr.VTGS_FORWARD_ASYNC, dgr.VTGS_FORWARD_CHECKED, dgr.VTGS_FORWARD_SYNC):
        st, color, depth, radii, rec, _ = _abi_forward(dgr, t, cam, dev, n, W, H, cap, 768, mode)
        assert rec[4] == 1 and rec[3] == 2 and rec[2] > 768, rec
        assert st == (0 if mode == dgr.VTGS_FORWARD_ASYNC else dgr.VTGS_ERR_INSTANCE_OVERFLOW)
        assert torch.equal(color, torch.zeros_like(color)) and torch.equal(depth, torch.zeros_like(depth))   # bg = 0
        assert int(radii.min()) >= 0
    tiles = (W + 7) // 8 * ((H + 7) // 8)
    plan = torch.empty(tiles + 1, dtype=torch.int32, device=dev)
    assert dgr._lib.vtgs_bin_plan_uniform(W, H, 600, plan.data_ptr(), torch.cuda.current_stream(dev).cuda_stream) == 0
    st, color, depth, radii, rec, _ = _abi_forward(dgr, t, cam, dev, n, W, H, cap, dgr.PLANNED | 600, dgr.VTGS_FORWARD_ASYNC, plan)
    assert st == 0 and rec[3] & 2 and rec[4] == 1
    assert torch.equal(color, torch.zeros_like(color))
    # an instance-capacity overflow: every tile is composited as empty
    st, color, depth, radii, rec, _ = _abi_forward(dgr, t, cam, dev, n, W, H, 5000, 4096, dgr.VTGS_FORWARD_ASYNC)
    assert st == 0 and rec[3] & 1 and torch.equal(color, torch.zeros_like(color))


def test_lists_beyond_512_without_the_presort_pass(gpu_device):
    """VTGS_FORWARD_EXPECT_SHORT_LISTS (bins of 768..1024 slots launched without the pre-sort pass): lists of 513..1024
    entries that turn up anyway are sorted by the composite's own network -- same lists, same image as with the pass."""
    import diff_gaussian_rasterization as dgr
    dev = gpu_device
    n, W, H = 24000, 96, 64
    scene, cam = go.view_tied_scene(n, W, H, seed=22)
    t = {k: v.to(dev).contiguous() for k, v in scene.items()}
    cap = 8 * n + 65536
    out = {}
    for name, flags in (("pass", dgr.VTGS_FORWARD_SYNC), ("hint", dgr.VTGS_FORWARD_SYNC | dgr.VTGS_FORWARD_EXPECT_SHORT_LISTS)):
        st, color, depth, radii, rec, ws = _abi_forward(dgr, t, cam, dev, n, W, H, cap, 1024, flags)
        assert st == 0 and rec[3] == 0 and 512 < rec[2] <= 1024, rec
        lay = (ctypes.c_uint64 * 12)()
        assert dgr._lib.vtgs_debug_layout(n, W, H, cap, 1024, lay) == 0
        tiles = int(lay[7])
        cnt = ws[int(lay[3]): int(lay[3]) + 4 * tiles].view(torch.int32).cpu()
        gid = ws[int(lay[4]): int(lay[4]) + 4 * tiles * 1024].view(torch.int32).reshape(tiles, 1024).cpu()
        out[name] = (color, depth, radii, cnt, [gid[i, :int(cnt[i])].clone() for i in range(tiles)])
    a, b = out["pass"], out["hint"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert all(torch.equal(x, y) for x, y in zip(a[4], b[4]))
    assert torch.isfinite(a[0]).all() and float(a[0].abs().max()) > 0


def test_alternating_views_never_raise(gpu_device):
    """VERDICT r3 item 3 -- the pattern of the reference's mapping loop over random keyframes (src/vtgaussian_slam.py:
    2563-2585) under one (N, W, H): three similar views, then one that bins more than TWICE the instances, sixty grad-mode
    iterations through the plain GaussianRasterizer.  No exception; every image and gradient equals a checked forward's bit
    for bit; and the loop does run ahead of its records where it is safe (three times the last need fits both capacities)."""
    import diff_gaussian_rasterization as dgr
    dev = gpu_device
    n, W, H = 9000, 200, 136
    scene, cam = go.view_tied_scene(n, W, H, seed=33)
    st = to_settings(cam, dev)
    g = torch.Generator().manual_seed(4)
    grad_color = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)

    def view(it):
        sc = dict(scene)
        if it % 4 == 3:                                            # the odd one out: wider splats, > 2 x the instances
            sc["scales"] = scene["scales"] * 3.0
        else:
            sc["means3D"] = scene["means3D"] + torch.tensor([0.002 * (it % 4), -0.001 * (it % 4), 0.0])
        return sc

    def run(mode):
        _clear_policy(dgr)
        dgr._FORWARD_MODE = mode
        outs, ahead, inst = [], 0, []
        for it in range(60):
            leaves = {k: v.to(dev).requires_grad_(True) for k, v in view(it).items()}
            rast = dgr.GaussianRasterizer(raster_settings=st)
            c, r, d = rast(**leaves)
            ahead += rast._last_state.pending is not None
            c.backward(grad_color)
            outs.append((c.detach().clone(), r.clone(), leaves["means3D"].grad.clone(), leaves["scales"].grad.clone()))
            if mode == "checked":
                inst.append(dgr.last_forward_info()["instances"])
        dgr.settle_pending()
        return outs, ahead, inst

    try:
        ref, ahead_ref, inst = run("checked")
        assert ahead_ref == 0
        assert min(inst[3::4]) > 2 * max(inst[0::4] + inst[1::4] + inst[2::4]), (inst[:8])
        got, ahead, _ = run("auto")                                 # must not raise
        assert ahead >= 10, ahead
        for it, (a, b) in enumerate(zip(ref, got)):
            for x, y in zip(a, b):
                assert torch.equal(x, y), it
    finally:
        dgr._FORWARD_MODE = os.environ.get("VTGS_FORWARD_MODE", "auto")



def test_a_scratch_smaller_than_the_instance_ids_gives_wrong_numbers_not_a_fault(gpu_device, monkeypatch):
    """the backward's scratch is sized from the instance count of the forward's record; nothing on the device used to tie
    the ids in the tile lists to that size, so a wrong or stale count was an out-of-bounds write (composite) and read (gather).
    Now both kernels know how many records the scratch holds (CamScalars::scratch_records): an id at or beyond it is neither
    written nor read, and the Gaussians whose records would lie there get a ZERO gradient.  Here the scratch is cut to a
    third of the ids on purpose -- on poisoned memory, so that a read past it would be a NaN or a fault."""
    import diff_gaussian_rasterization as dgr
    monkeypatch.setattr(dgr, "_ext", None)                        # (the Python node sizes the scratch through _scratch_instances)
    dev = gpu_device
    scene, cam = go.view_tied_scene(20000, 200, 136, seed=9)
    leaves = {k: v.to(dev).requires_grad_(True) for k, v in scene.items()}
    g = torch.Generator().manual_seed(2)
    grad_color = (torch.rand(3, 136, 200, generator=g) * 2 - 1).to(dev)
    _clear_policy(dgr)
    dgr.poison_workspaces(True)
    try:
        c, r, d = dgr.GaussianRasterizer(raster_settings=to_settings(cam, dev))(**leaves)
        full = {}
        (c * grad_color).sum().backward(retain_graph=True)
        torch.cuda.synchronize()
        full = {k: leaves[k].grad.clone() for k in GRAD_KEYS}
        ids = dgr.last_forward_info()["instances_needed"]
        for t in leaves.values():
            t.grad = None
        real = dgr._scratch_instances
        monkeypatch.setattr(dgr, "_scratch_instances", lambda fs: max(real(fs) // 3, 1))
        (c * grad_color).sum().backward()
        torch.cuda.synchronize()                                  # a fault would surface here
    finally:
        dgr.poison_workspaces(False)
    cut = {k: leaves[k].grad for k in GRAD_KEYS}
    same = zero = 0
    for k in ("means3D", "opacities", "colors_precomp"):
        assert bool(torch.isfinite(cut[k]).all()), k
        eq = (cut[k] == full[k]).reshape(cut[k].shape[0], -1).all(dim=1)
        zr = (cut[k] == 0).reshape(cut[k].shape[0], -1).all(dim=1)
        assert bool((eq | zr).all()), f"{k}: a Gaussian has a gradient that is neither the full one nor zero"
        same, zero = int(eq.sum()), int((zr & ~eq).sum())
    assert ids > 3 and zero > 0 and same > 0, (ids, same, zero)    # some Gaussians kept their records, some lost them
