"""SURVEY 8e on the device: the band forms of the loss kernels (vtgs_slam_loss_band_*, include/vtgs.h "ONE BAND") against
the full-frame loss node they partition, and the gradients the backward is asked for (src/vtgaussian_slam.py:428-449: the
tracking loop detaches the Gaussians).  One process plays every rank in turn: the reduction over the ranks is a sum of the
bands' eight floats, computed in a first pass."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _frames(dev, H=150, W=203, seed=0):
    """A rendered-looking pair of images with a ground truth: smooth fields + noise, holes in the depth, a dark silhouette rim."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    base = torch.stack([0.5 + 0.4 * torch.sin(7 * xx + 3 * yy), 0.5 + 0.4 * torch.cos(5 * yy - 2 * xx), 0.3 + 0.5 * xx * yy])
    gt_im = (base + 0.03 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    im = (base + 0.05 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    z = 2.0 + yy + 0.5 * torch.sin(9 * xx)
    sil = (0.9 + 0.12 * torch.rand(H, W, generator=g)).clamp(max=1.0)
    sil[:, :6] = 0.4
    ds = torch.stack([z * sil, sil, z * z * sil + 0.01])
    gt_depth = (z * (1 + 0.02 * torch.randn(H, W, generator=g)))[None]
    gt_depth[:, 40:55, 30:80] = 0.0
    gt_depth[:, 100:104, 150:170] *= 30.0
    return [t.to(dev) for t in (im, ds, gt_im, gt_depth)]


def _bands(H, world):
    from diff_gaussian_rasterization.partition import all_bands, pixel_rows
    return [pixel_rows(b, H) for b in all_bands(H, world)]


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("masks", ["plain", "extra+weights"])
def test_band_shares_of_the_mapping_loss_sum_to_the_full_frame(gpu_device, world, masks):
    from diff_gaussian_rasterization import losses
    dev = gpu_device
    im0, ds0, gt_im, gt_depth = _frames(dev)
    H, W = im0.shape[-2:]
    extra = add = None
    if masks != "plain":
        g = torch.Generator().manual_seed(5)
        extra = (torch.rand(H, W, generator=g) > 0.2).to(dev)
        add = (torch.rand(1, H, W, generator=g) > 0.7).float().to(dev)
    im, ds = im0.clone().requires_grad_(True), ds0.clone().requires_grad_(True)
    full, terms = losses.mapping_loss(im, ds, gt_im, gt_depth, w_im=0.7, w_depth=1.3, extra_mask=extra, additional_mask=add,
                                      return_terms=True)
    full.backward()
    ref_gi, ref_gd = im.grad.clone(), ds.grad.clone()
    bands = _bands(H, world)
    kw = dict(mode="mapping", w_im=0.7, w_depth=1.3, extra_mask=extra, additional_mask=add)
    with torch.no_grad():                                           # pass 1: every band's sums -> what the all-reduce returns
        total = sum(losses.band_loss(im0, ds0, gt_im, gt_depth, rows, return_terms=True, **kw)[2] for rows in bands)
    assert abs(total[2].item() - terms[1].item()) == 0              # the mask count is exact
    im.grad = ds.grad = None
    shares = []
    for r, rows in enumerate(bands):                                # pass 2: the shares and their gradients
        share, rec, tot = losses.band_loss(im, ds, gt_im, gt_depth, rows, reduce=lambda t: total.clone(), first_band=(r == 0),
                                           return_terms=True, **kw)
        assert rec[1].item() == terms[1].item() and abs(rec[4].item() - terms[4].item()) <= 2e-6
        share.backward()
        shares.append(share.detach())
    got = torch.stack(shares).sum()
    assert abs(got.item() - full.item()) <= 3e-6 * abs(full.item()), (got.item(), full.item())
    for got_g, ref_g, what in ((im.grad, ref_gi, "im"), (ds.grad, ref_gd, "depth_sil")):
        err = (got_g - ref_g).abs().max().item() / ref_g.abs().max().item()
        assert err <= 2e-5, f"d loss / d {what}: {err:.2e}"


@pytest.mark.parametrize("all_pixels", [False, True])
def test_band_shares_of_the_tracking_loss_sum_to_the_full_frame(gpu_device, all_pixels):
    from diff_gaussian_rasterization import losses
    dev = gpu_device
    im0, ds0, gt_im, gt_depth = _frames(dev, seed=3)
    H, W = im0.shape[-2:]
    g = torch.Generator().manual_seed(6)
    extra = (torch.rand(1, H, W, generator=g) > 0.1).to(dev)
    im, ds = im0.clone().requires_grad_(True), ds0.clone().requires_grad_(True)
    thr = float("-inf") if all_pixels else 0.95
    full = losses.tracking_loss(im, ds, gt_im, gt_depth, thr, w_im=0.5, w_depth=1.0, extra_mask=extra,
                                colour_over_all_pixels=all_pixels)
    full.backward()
    ref_gi, ref_gd = im.grad.clone(), ds.grad.clone()
    im.grad = ds.grad = None
    shares = []
    for rows in _bands(H, 4):
        share = losses.band_loss(im, ds, gt_im, gt_depth, rows, mode="tracking", sil_thres=thr, w_im=0.5, w_depth=1.0,
                                 extra_mask=extra, colour_over_all_pixels=all_pixels)
        share.backward()
        shares.append(share.detach())
    got = torch.stack(shares).sum()
    assert abs(got.item() - full.item()) <= 3e-6 * abs(full.item())
    assert torch.equal(im.grad, ref_gi) and torch.equal(ds.grad, ref_gd)       # signs times weights: no rounding to differ by


def test_band_sums_of_the_threshold_sweep(gpu_device):
    from diff_gaussian_rasterization import losses
    from diff_gaussian_rasterization.partition import all_bands, band_silhouette_threshold
    im, ds, gt_im, gt_depth = _frames(gpu_device, seed=8)
    H = im.shape[-2]
    cands = (0.90, 0.93, 0.95, 0.97, 0.99)
    full = losses.silhouette_sweep(im, ds[1], gt_im, gt_depth, cands)
    parts = sum(losses.silhouette_sweep(im, ds[1], gt_im, gt_depth, cands, rows=rows) for rows in _bands(H, 3))
    assert torch.equal(parts[:, 1], full[:, 1])                                # pixel counts
    assert ((parts[:, 0] - full[:, 0]).abs() <= 1e-6 * full[:, 0].abs()).all()
    one = band_silhouette_threshold(im, ds[1], gt_im, gt_depth, all_bands(H, 1)[0], 1, cands)
    assert one == losses.best_silhouette_threshold(im, ds[1], gt_im, gt_depth, cands)
    with pytest.raises(ValueError):
        losses.silhouette_sweep(im, ds[1], gt_im, gt_depth, cands, rows=(10, H + 1))


def test_band_loss_rejects_what_it_cannot_serve(gpu_device):
    from diff_gaussian_rasterization import losses
    im, ds, gt_im, gt_depth = _frames(gpu_device)
    H = im.shape[-2]
    for rows in ((0, 0), (-1, 5), (5, H + 1), (9, 3)):
        with pytest.raises(ValueError):
            losses.band_loss(im, ds, gt_im, gt_depth, rows)
    with pytest.raises(ValueError):
        losses.band_loss(im, ds, gt_im, gt_depth, (0, 16), mode="both")
    with pytest.raises(RuntimeError):
        losses.band_loss(im.cpu(), ds.cpu(), gt_im.cpu(), gt_depth.cpu(), (0, 16))


@pytest.mark.parametrize("node", ["c++", "python"])
def test_backward_stores_only_the_gradients_asked_for(gpu_device, node, monkeypatch):
    """Tracking through the plain operator: the Gaussians are detached, only means3D (and the screen-space term) need a
    gradient.  What is asked for is bit-identical to the all-six backward, the rest is None."""
    import diff_gaussian_rasterization as dgr
    from oracle import gs_oracle as go
    from parity_util import to_settings
    if node == "python":
        monkeypatch.setattr(dgr, "_ext", None)
    elif dgr._ext is None:
        pytest.skip("vtgs_torch.so not built")
    dev = gpu_device
    scene, cam = go.view_tied_scene(20000, 320, 240, seed=4)
    settings = to_settings(cam, dev)
    g = torch.Generator().manual_seed(2)
    grad_color = (torch.rand(3, cam.image_height, cam.image_width, generator=g) * 2 - 1).to(dev)

    def run(wanted):
        leaves = {k: v.to(dev).requires_grad_(k in wanted) for k, v in scene.items()}
        color, _, _ = dgr.GaussianRasterizer(raster_settings=settings)(**leaves)
        color.backward(grad_color)
        dgr.settle_pending()
        return {k: v.grad for k, v in leaves.items()}
    ref = run(set(scene))
    assert all(v is not None for v in ref.values())
    for wanted in ({"means3D", "means2D"}, {"colors_precomp", "opacities", "scales"}, {"rotations"}, {"opacities"}):
        got = run(wanted)
        for k in scene:
            if k in wanted:
                assert torch.equal(got[k], ref[k]), (wanted, k)
            else:
                assert got[k] is None, (wanted, k)


# ---- two ranks on one GPU (gloo, collectives staged through the host): the N-rank iteration against the 1-rank one ---------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(dev):
    from oracle import gs_oracle as go
    from parity_util import to_settings
    N, W, H = 6000, 160, 112                                       # 7 tile rows: bands of 4 and 3
    scene, cam = go.view_tied_scene(N, W, H, seed=31)
    settings = to_settings(cam, dev)
    g = torch.Generator().manual_seed(13)
    params = {
        "means3D": scene["means3D"].to(dev), "rgb_colors": scene["colors_precomp"].to(dev),
        "unnorm_rotations": scene["rotations"].to(dev),
        "logit_opacities": (2.0 + 0.5 * torch.randn(N, 1, generator=g)).to(dev),
        "log_scales": torch.log(scene["scales"][:, :1]).to(dev),
        "cam_unnorm_rots": torch.tensor([0.9999, 0.004, -0.003, 0.002]).reshape(1, 4, 1).to(dev),
        "cam_trans": torch.tensor([0.004, -0.002, 0.003]).reshape(1, 3, 1).to(dev),
    }
    gt_im = torch.rand(3, H, W, generator=g).to(dev)
    gt_depth = (1.5 + 4 * torch.rand(1, H, W, generator=g)).to(dev)
    gt_depth[:, 20:30, 40:90] = 0.0
    return params, settings, gt_im, gt_depth, H


def _iteration(params, settings, gt_im, gt_depth, H, rank, world, kind, owned_sets=False):
    """One tracking / mapping iteration the way bench_slam.py's N-rank loop does it; returns (loss, gradients).
    owned_sets: every rank renders from the list of Gaussians that can meet its band (partition.OwnedSet)."""
    import torch.distributed as dist
    from diff_gaussian_rasterization import losses, partition as pt
    from diff_gaussian_rasterization.fused import render_frame
    dev = gt_im.device
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    band = pt.band_for_rank(H, world, rank) if world > 1 else None
    first_w2c = torch.eye(4, device=dev)
    own = pt.OwnedSet(p, 0, settings, first_w2c, band) if (owned_sets and world > 1) else None
    if kind == "tracking":
        im, ds, _ = render_frame(p, 0, settings, first_w2c, False, True, tile_rows=band, owned=own)
        thr = (pt.band_silhouette_threshold(im, ds[1], gt_im, gt_depth, band, world) if world > 1
               else losses.best_silhouette_threshold(im, ds[1], gt_im, gt_depth))
        loss = (pt.band_tracking_loss(im, ds, gt_im, gt_depth, band, thr) if world > 1
                else losses.tracking_loss(im, ds, gt_im, gt_depth, thr))
        loss.backward()
        grads = {k: p[k].grad.clone() for k in ("cam_unnorm_rots", "cam_trans")}
        grads["thr"] = torch.tensor([thr])
    else:
        im, ds, _ = render_frame(p, 0, settings, first_w2c, True, False, tile_rows=band, owned=own)
        loss = (pt.band_mapping_loss(im, ds, gt_im, gt_depth, band, rank, world, ignore_outlier_depth_loss=True) if world > 1
                else losses.mapping_loss(im, ds, gt_im, gt_depth, extra_mask=losses.outlier_depth_mask(gt_depth, ds[0:1])))
        loss.backward()
        if world > 1:
            pt.allreduce_param_grads(p)
        grads = {k: p[k].grad.clone() for k in ("rgb_colors", "logit_opacities", "log_scales")}
    total = loss.detach().clone().reshape(1)
    if world > 1:
        for v in grads.values():
            if v.is_cuda and v.numel() <= 16:
                pt.all_reduce_sum(v)                                # the pose gradient: summed over the bands
        pt.all_reduce_sum(total)
    if own is not None:
        assert 0 < len(own) < params["means3D"].shape[0]
        assert pt.phase_escapes([own]) == 0                         # (collective: every rank asks)
    return total, grads


def _rank_main(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    res = {}
    for kind in ("tracking", "mapping"):
        loss, grads = _iteration(*_problem(dev), rank, world, kind, owned_sets=True)
        res[kind] = {"loss": loss.cpu(), **{k: v.cpu() for k, v in grads.items()}}
    # margin violation: lists built with the smallest margin for a pose ~6 degrees away.  Rank 1's band receives Gaussians
    # that are not on its list; BOTH ranks learn of it (phase_escapes is a collective) and rebuild; then nothing escapes.
    from diff_gaussian_rasterization import partition as pt
    from diff_gaussian_rasterization.fused import render_frame
    params, settings, _gi, _gd, H = _problem(dev)
    band, w2c = pt.band_for_rank(H, world, rank), torch.eye(4, device=dev)
    stale = dict(params, cam_unnorm_rots=params["cam_unnorm_rots"] + torch.tensor([0, 0.05, 0, 0.0], device=dev).reshape(1, 4, 1))
    own = pt.OwnedSet(stale, 0, settings, w2c, band, margin_px=1.0, growth=1.0)
    with torch.no_grad():
        render_frame(params, 0, settings, w2c, False, False, tile_rows=band, owned=own)
    res["escapes_mine_before"] = own.escaped()
    res["escapes_before_rebuild"] = pt.phase_escapes([own])
    own = pt.OwnedSet(params, 0, settings, w2c, band, margin_px=1.0, growth=1.0)
    with torch.no_grad():
        render_frame(params, 0, settings, w2c, False, False, tile_rows=band, owned=own)
    res["escapes_after_rebuild"] = pt.phase_escapes([own])
    # owner exchange: three mapping iterations with the gradients sent to the owner band, Adam on the owned
    # rows, the updated rows published -- against the all-reduce + Adam-on-every-row route on a copy, both rendering from lists
    from diff_gaussian_rasterization.optim import FusedAdam
    map_lrs = dict(means3D=0.0, rgb_colors=0.0025, unnorm_rotations=0.0, logit_opacities=0.05, log_scales=0.005,
                   cam_unnorm_rots=1e-8, cam_trans=1e-7)
    params, settings, gt_im, gt_depth, H = _problem(dev)
    pa = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    pb = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    union = pt.OwnedSet(pa, 0, settings, w2c, band, margin_px=48.0, with_centre_rows=True)
    ex = pt.OwnerExchange(union, H, rank, world)
    own = pt.OwnedSet(pa, 0, settings, w2c, band)
    res["covers"] = ex.covers(own)
    oa = FusedAdam([{"params": [v], "name": k, "lr": map_lrs[k]} for k, v in pa.items()], lr=0.0, eps=1e-15, skip_frozen=True)
    ob = FusedAdam([{"params": [v], "name": k, "lr": map_lrs[k]} for k, v in pb.items()], lr=0.0, eps=1e-15, skip_frozen=True)
    keys = ("rgb_colors", "logit_opacities", "log_scales")
    worst, sent = 0.0, 0
    for it in range(3):
        for p, route in ((pa, "owner"), (pb, "allreduce")):
            im, ds, _ = render_frame(p, 0, settings, w2c, True, False, tile_rows=band, owned=own)
            pt.band_mapping_loss(im, ds, gt_im, gt_depth, band, rank, world, ignore_outlier_depth_loss=True).backward()
        # (iteration 0: both copies hold the same parameters, so the two routes' summed gradients must agree on the owned rows)
        local = {k: pa[k].grad.clone() for k in keys}
        sent += ex.reduce_grads(pa)
        pt.allreduce_param_grads(pb)
        if it == 0:
            for k in keys:
                a, b = pa[k].grad[ex.own_rows], pb[k].grad[ex.own_rows]
                worst = max(worst, float((a - b).abs().max() / b.abs().max()))
        oa.step(rows=ex.update_rows); ob.step()
        sent += ex.publish(pa)
        oa.zero_grad(); ob.zero_grad()
    listed = own.mask.bool()
    res["listed_diff"] = max(float((pa[k].detach()[listed] - pb[k].detach()[listed]).abs().max()) for k in keys)
    ex.gather_all(pa)
    res["final_q99"] = {k: float((pa[k].detach() - pb[k].detach()).abs().reshape(-1).quantile(0.99)) for k in keys}
    res["moved_median"] = {k: float((pb[k].detach() - params[k]).abs().median()) for k in keys}
    res["reduced_grad_rel"], res["bytes_owner_route"] = worst, sent
    res["bytes_allreduce_route"] = 3 * 20 * params["means3D"].shape[0]
    res["halo_rows"], res["own_rows"] = ex.halo_rows, int(ex.own_rows.numel())
    if rank == 0:
        torch.save(res, out)
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_the_single_rank_iteration(gpu_device, tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "r0.pt")
    mp.get_context("spawn")
    mp.spawn(_rank_main, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["escapes_before_rebuild"] > 0 and got["escapes_before_rebuild"] >= got["escapes_mine_before"]
    assert got["escapes_after_rebuild"] == 0
    # owner exchange: the owner's summed gradient is the all-reduced one; three Adam iterations on the owned rows + publish leave
    # the parameters where all-reduce + Adam on every row leaves them (Adam normalises noise-level gradients: quantiles, as in
    # tests/test_loop_fixture.py), for a fraction of the bytes
    assert got["covers"] and got["reduced_grad_rel"] <= 2e-6, got["reduced_grad_rel"]
    for k, lr in (("rgb_colors", 0.0025), ("logit_opacities", 0.05), ("log_scales", 0.005)):
        assert got["moved_median"][k] > 0.3 * lr, (k, got["moved_median"])
        assert got["final_q99"][k] <= 0.02 * 3 * lr, (k, got["final_q99"])
    assert got["listed_diff"] <= 3 * 0.05 and 0 < got["halo_rows"] < got["own_rows"]
    # (a 160 x 112 frame in two bands with a 48-pixel margin is nearly all halo: the byte saving shows at real sizes, bench_slam.py)
    assert got["bytes_owner_route"] < got["bytes_allreduce_route"], (got["bytes_owner_route"], got["bytes_allreduce_route"])
    for kind in ("tracking", "mapping"):
        loss, grads = _iteration(*_problem(gpu_device), 0, 1, kind)
        ref = {"loss": loss.cpu(), **{k: v.cpu() for k, v in grads.items()}}
        assert abs(got[kind]["loss"].item() - ref["loss"].item()) <= 1e-5 * abs(ref["loss"].item()), kind
        for k, v in ref.items():
            if k in ("loss",):
                continue
            if k == "thr":
                assert torch.equal(got[kind][k], v)
                continue
            scale = v.abs().max().item()
            assert scale > 0, (kind, k)
            err = (got[kind][k] - v).abs().max().item() / scale
            assert err <= 2e-4, f"{kind}: d loss / d {k} differs by {err:.2e} of its maximum"


def _rccl_main(rank, world, port, out):
    import torch.distributed as dist
    from diff_gaussian_rasterization import partition as pt
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert not pt._host_staged()
    t = torch.arange(8, dtype=torch.float32, device=dev)
    pt.all_reduce_sum(t)
    r = torch.tensor([0, 3, 0, 7], dtype=torch.int32, device=dev)
    pt.allreduce_radii(r)
    p = {k: torch.nn.Parameter(torch.ones(5, w, device=dev)) for k, w in (("rgb_colors", 3), ("logit_opacities", 1), ("log_scales", 1))}
    for v in p.values():
        v.grad = torch.full_like(v, 2.0)
    nbytes = pt.allreduce_param_grads(p)
    img = torch.rand(3, 32, 16, device=dev, requires_grad=True)
    full = pt.halo_exchange(img, (0, 2), 32, rank, world)                    # one rank: no neighbour, a differentiable copy
    full.sum().backward()
    med = pt.global_median(torch.tensor([3.0, 1.0, 2.0, 9.0], device=dev))
    torch.cuda.synchronize()
    torch.save({"t": t.cpu(), "r": r.cpu(), "bytes": nbytes, "g": p["rgb_colors"].grad.cpu(), "halo_equal": bool(torch.equal(full, img)),
                "halo_grad": img.grad.cpu(), "median": med.cpu()}, out)
    dist.destroy_process_group()


def test_collectives_of_the_partition_run_on_rccl(gpu_device, tmp_path):
    """One rank, backend "nccl" (= RCCL): the collective helpers of the N-rank loop take device tensors as they are (no host
    staging) and RCCL initialises on this box -- the N-GPU run itself is the driver's."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_main, args=(1, _free_port(), out), nprocs=1, join=True)
    got = torch.load(out)
    assert torch.equal(got["t"], torch.arange(8, dtype=torch.float32)) and got["r"].tolist() == [0, 3, 0, 7]
    assert got["bytes"] == 20 * 5 and bool((got["g"] == 2).all()) and got["halo_equal"]
    assert bool((got["halo_grad"] == 1).all()) and got["median"].item() == 2.0
