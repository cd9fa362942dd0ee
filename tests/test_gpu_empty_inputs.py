"""Empty input through the bare C ABI (include/vtgs.h "EMPTY INPUT").

Round 4 met a GPU memory access fault because a launcher chose the single-render kernel from a per-Gaussian pointer that the
header lets be NULL at n = 0.  Here EVERY exported render / backward / frame / pose / bookkeeping entry point is
called with n = 0 and NULL per-Gaussian arrays, on poisoned images and workspaces: the call returns VTGS_OK, a forward writes
the background colour into every pixel (depth 0, final transmittance 1, record: zero instances), a backward and the helpers
touch nothing they should not, and nothing faults.  Run ONCE (a fault is a fault, not a flake).

Every call is followed by a synchronise OF ITS OWN (`_call`): the fault of an asynchronous launch surfaces at the next
synchronise, and round 5 met an abort at a synchronise that stood behind ten entry points -- raised on a runtime thread, without
the runtime's memory-fault line -- whose launcher could not be named afterwards.  With one synchronise per entry
point the traceback's line IS the entry point."""
import ctypes

import pytest
import torch

from oracle import gs_oracle as go
from parity_util import to_settings

pytestmark = pytest.mark.gpu
NULL = None


def _call(lib, name, *args, what=None):
    st = getattr(lib, name)(*args)
    assert st == 0, (name, what, lib.vtgs_strerror(st))
    torch.cuda.synchronize()               # this entry point's launches have retired (or the process has died HERE)


def _blocks(lib, dev, W, H, cap, tcap, dual):
    nbytes = lib.vtgs_workspace_bytes(0, W, H, cap, tcap)
    ws = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device=dev)
    a = torch.full((3, H, W), float("nan"), device=dev)
    b = torch.full((3 if dual else 1, H, W), float("nan"), device=dev)
    slot = torch.zeros(64, dtype=torch.uint8).pin_memory()
    return nbytes, ws, a, b, slot


@pytest.mark.parametrize("band", [None, (1, 4)])
def test_every_entry_point_with_an_empty_map(gpu_device, band):
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import fused, losses, optim, partition   # noqa: F401  (their ctypes signatures)
    lib, dev = dgr._lib, gpu_device
    W, H = 104, 72
    _, cam = go.view_tied_scene(16, W, H, seed=3)
    bg = torch.tensor([0.25, 0.5, 0.75])
    camobj = dgr._camera_for(to_settings(cam, dev, bg), dev, 0, band)
    c = ctypes.byref(camobj.c)
    stream = torch.cuda.current_stream(dev).cuda_stream
    cap, tcap = 4096, 64
    rows = slice(0, H) if band is None else slice(16 * band[0], min(16 * band[1], H))
    expect = torch.zeros(3, H, W)
    expect[:, rows] = bg[:, None, None]

    def check_forward(st, a, b, slot, dual, ws=None, state=None):
        assert st == 0, lib.vtgs_strerror(st)
        torch.cuda.synchronize()
        assert torch.equal(a.cpu(), expect)
        if dual:
            assert torch.equal(b.cpu(), expect)
        else:
            assert torch.equal(b.cpu(), torch.zeros(1, H, W))
        if slot is not None:
            info = dgr._VtgsForwardInfo.from_address(slot.data_ptr())
            assert info.complete == 1 and info.overflow == 0 and info.instances == 0 and info.visible == 0

    # ---- forwards: single, dual, planned, dual planned, in every mode; the shared second render ---------------------------
    plan = torch.empty(lib.vtgs_bin_plan_entries(W, H), dtype=torch.int32, device=dev)
    assert lib.vtgs_bin_plan_uniform(W, H, tcap, plan.data_ptr(), stream) == 0
    kept = None
    for mode in (dgr.VTGS_FORWARD_SYNC, dgr.VTGS_FORWARD_ASYNC, dgr.VTGS_FORWARD_CHECKED):
        nbytes, ws, a, b, slot = _blocks(lib, dev, W, H, cap, tcap, False)
        st = lib.vtgs_forward(c, 0, NULL, NULL, NULL, NULL, NULL, a.data_ptr(), b.data_ptr(), NULL, ws.data_ptr(), nbytes, cap, tcap,
                              slot.data_ptr(), mode, stream)
        check_forward(st, a, b, slot, False)
        kept = (nbytes, ws, a)
        nbytes, ws, a, b, slot = _blocks(lib, dev, W, H, cap, tcap, True)
        st = lib.vtgs_forward_dual(c, 0, NULL, NULL, NULL, NULL, NULL, NULL, a.data_ptr(), b.data_ptr(), NULL, ws.data_ptr(), nbytes,
                                   cap, tcap, slot.data_ptr(), mode, stream)
        check_forward(st, a, b, slot, True)
        kept_dual = (nbytes, ws, a, b)
        ptcap = tcap | dgr.PLANNED
        nbytes, ws, a, b, slot = _blocks(lib, dev, W, H, cap, ptcap, False)
        st = lib.vtgs_forward_planned(c, 0, NULL, NULL, NULL, NULL, NULL, a.data_ptr(), b.data_ptr(), NULL, ws.data_ptr(), nbytes, cap,
                                      ptcap, plan.data_ptr(), slot.data_ptr(), mode, stream)
        check_forward(st, a, b, slot, False)
        nbytes, ws, a, b, slot = _blocks(lib, dev, W, H, cap, ptcap, True)
        st = lib.vtgs_forward_dual_planned(c, 0, NULL, NULL, NULL, NULL, NULL, NULL, a.data_ptr(), b.data_ptr(), NULL, ws.data_ptr(),
                                           nbytes, cap, ptcap, plan.data_ptr(), slot.data_ptr(), mode, stream)
        check_forward(st, a, b, slot, True)
    nbytes, ws, _ = kept
    a2 = torch.full((3, H, W), float("nan"), device=dev)
    d2 = torch.full((1, H, W), float("nan"), device=dev)
    state = torch.full((H * W,), float("nan"), device=dev)
    st = lib.vtgs_forward_shared(c, 0, NULL, a2.data_ptr(), d2.data_ptr(), ws.data_ptr(), nbytes, cap, tcap, state.data_ptr(), stream)
    check_forward(st, a2, d2, None, False)
    assert torch.equal(state.cpu().reshape(H, W)[rows], torch.ones(H, W)[rows])

    # ---- backwards: nothing to differentiate, nothing touched --------------------------------------------------------------
    g = torch.rand(3, H, W, device=dev)
    scratch = torch.full((256,), 0xFF, dtype=torch.uint8, device=dev)
    _call(lib, "vtgs_backward", c, 0, NULL, NULL, NULL, NULL, NULL, kept[2].data_ptr(), g.data_ptr(), ws.data_ptr(), nbytes, cap, tcap,
          NULL, scratch.data_ptr(), 256, NULL, NULL, NULL, NULL, NULL, NULL, stream)
    nbd, wsd, ad, bd = kept_dual
    _call(lib, "vtgs_backward_dual", c, 0, NULL, NULL, NULL, NULL, NULL, NULL, ad.data_ptr(), bd.data_ptr(), g.data_ptr(), g.data_ptr(),
          wsd.data_ptr(), nbd, cap, tcap, scratch.data_ptr(), 256, NULL, NULL, NULL, NULL, NULL, NULL, NULL, stream)
    q = torch.tensor([1.0, 0, 0, 0], device=dev)
    t = torch.zeros(3, device=dev)
    w2c = torch.eye(4, device=dev).reshape(-1).contiguous()
    for flags in (1, 2, 4, 7, 2 | 8, 7 | 8):
        _call(lib, "vtgs_backward_dual_frame", c, 0, NULL, NULL, NULL, NULL, NULL, NULL, ad.data_ptr(), bd.data_ptr(), g.data_ptr(),
              g.data_ptr(), wsd.data_ptr(), nbd, cap, tcap, scratch.data_ptr(), 256, flags, NULL, NULL,
              q.data_ptr(), t.data_ptr(), w2c.data_ptr(), NULL, NULL, NULL, NULL, NULL, NULL, stream, what=flags)
        _call(lib, "vtgs_backward_dual_frame_owned", c, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, ad.data_ptr(), bd.data_ptr(),
              g.data_ptr(), g.data_ptr(), wsd.data_ptr(), nbd, cap, tcap, scratch.data_ptr(), 256,
              flags, NULL, NULL, q.data_ptr(), t.data_ptr(), w2c.data_ptr(), NULL, NULL, NULL, NULL,
              NULL, NULL, stream, what=flags)
    assert bool((scratch == 0xFF).all()), "an empty backward wrote into its scratch"

    # ---- the caller chain around the operator ---------------------------------------------------------------------------------
    _call(lib, "vtgs_prepare_frame", 0, NULL, NULL, NULL, NULL, q.data_ptr(), t.data_ptr(), w2c.data_ptr(), NULL, NULL, NULL, NULL, NULL,
          stream)
    _call(lib, "vtgs_prepare_frame_owned", 0, NULL, NULL, NULL, NULL, NULL, NULL, q.data_ptr(), t.data_ptr(), w2c.data_ptr(), NULL, NULL,
          NULL, NULL, NULL, NULL, stream)
    _call(lib, "vtgs_prepare_frame_backward", 0, 7, NULL, NULL, NULL, NULL, q.data_ptr(), t.data_ptr(), w2c.data_ptr(), *([NULL] * 14),
          stream)
    assert lib.vtgs_pose_partial_rows(0) == 0
    gq = torch.full((4,), float("nan"), device=dev)
    gt = torch.full((3,), float("nan"), device=dev)
    _call(lib, "vtgs_pose_gradient", NULL, 0, q.data_ptr(), gq.data_ptr(), gt.data_ptr(), stream)
    assert torch.equal(gq.cpu(), torch.zeros(4)) and torch.equal(gt.cpu(), torch.zeros(3))
    out7 = torch.full((7,), float("nan"), device=dev)
    _call(lib, "vtgs_pose7_reduce", 0, NULL, NULL, NULL, out7.data_ptr(), stream)
    assert torch.equal(out7.cpu(), torch.zeros(7))
    _call(lib, "vtgs_mark_visible", c, 0, NULL, NULL, stream)
    esc = torch.zeros(1, dtype=torch.int32, device=dev)
    _call(lib, "vtgs_band_owner_mask", c, 0, NULL, NULL, 1, NULL, NULL, 32.0, 1.25, NULL, NULL, NULL, esc.data_ptr(), stream)
    _call(lib, "vtgs_seen_and_max_radius", 0, NULL, NULL, NULL, stream)
    grp = (optim._Group * 1)()
    grp[0] = optim._Group(None, None, None, None, 0, 1e-3, 1e-15)
    _call(lib, "vtgs_adam_step", grp, 1, 1, 0.9, 0.999, stream)
    _call(lib, "vtgs_adam_step_rows", grp, 1, 1, 0.9, 0.999, NULL, 0, 1, stream)
    assert int(esc.item()) == 0


@pytest.mark.parametrize("ext", [True, False])
def test_empty_map_through_the_operator_and_the_fused_frame(gpu_device, ext, monkeypatch):
    """The same through both autograd nodes (C++ and Python) of the plain operator and of fused.render_frame: an empty map
    renders the background, its backward returns empty gradients and a zero pose gradient."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import fused
    dev = gpu_device
    if not ext:
        monkeypatch.setattr(dgr, "_ext", None)
    elif dgr._ext is None:
        pytest.skip("the C++ autograd node is not built")
    W, H = 104, 72
    _, cam = go.view_tied_scene(16, W, H, seed=3)
    bg = torch.tensor([0.25, 0.5, 0.75])
    st = to_settings(cam, dev, bg)
    dgr.poison_workspaces(True)
    try:
        leaves = {"means3D": torch.zeros(0, 3), "means2D": torch.zeros(0, 3), "opacities": torch.zeros(0, 1),
                  "colors_precomp": torch.zeros(0, 3), "scales": torch.zeros(0, 3), "rotations": torch.zeros(0, 4)}
        leaves = {k: v.to(dev).requires_grad_(True) for k, v in leaves.items()}
        color, radii, depth = dgr.GaussianRasterizer(raster_settings=st)(**leaves)
        assert torch.equal(color.cpu(), bg[:, None, None].expand(3, H, W)) and radii.numel() == 0
        assert torch.equal(depth.cpu(), torch.zeros(1, H, W))
        color.sum().backward()
        dgr.settle_pending()
        assert all(v.grad is None or v.grad.numel() == 0 for v in leaves.values())
    finally:
        dgr.poison_workspaces(False)
