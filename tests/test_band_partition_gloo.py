"""Tile-row partition across ranks: bands tile the image exactly, and band renders + summed band
gradients reproduce the full-frame result.  Runs world_size=2 over gloo on the CPU, with the oracle standing in
for the device operator (the partition arithmetic and the collective pattern are what is under test)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diff_gaussian_rasterization.partition import all_bands, band_for_rank, pixel_rows, tile_rows_total
from oracle import gs_oracle as go


def test_bands_tile_the_image():
    for h in (16, 17, 240, 480, 680, 1168):
        rows = tile_rows_total(h)
        for world in (1, 2, 3, 4, 8):
            if world > rows:
                with pytest.raises(ValueError):
                    band_for_rank(h, world, 0)
                continue
            bands = all_bands(h, world)
            assert bands[0][0] == 0 and bands[-1][1] == rows
            assert all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
            sizes = [e - b for b, e in bands]
            assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
            assert pixel_rows(bands[-1], h)[1] == h
    assert all_bands(680, 4) == [(0, 11), (11, 22), (22, 33), (33, 43)]          # SURVEY 8e example


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    scene, cam = go.view_tied_scene(1500, 96, 80, seed=12)
    H, W = cam.image_height, cam.image_width
    g = torch.Generator().manual_seed(3)
    grad_color = torch.rand(3, H, W, generator=g) * 2 - 1
    leaves = {k: v.clone().requires_grad_(True) for k, v in scene.items()}
    band = band_for_rank(H, world, rank)
    color, radii, depth = go.rasterize(cam=cam, tile_rows=band, **leaves)
    y0, y1 = pixel_rows(band, H)
    mask = torch.zeros(1, H, 1)
    mask[:, y0:y1] = 1
    (color * grad_color * mask).sum().backward()
    # tracking collective: 7 pose scalars; mapping collective: the per-Gaussian gradients
    gm = leaves["means3D"].grad
    pose = torch.cat([gm.sum(0), torch.cross(leaves["means3D"].detach(), gm, dim=1).sum(0), gm[:, 2:3].sum(0)])
    dist.all_reduce(pose)
    grads = {k: leaves[k].grad.clone() for k in ("means3D", "opacities", "colors_precomp", "scales")}
    for v in grads.values():
        dist.all_reduce(v)
    img = color.detach() * mask
    dist.all_reduce(img)
    if rank == 0:
        torch.save({"img": img, "pose": pose, **grads}, out)
    dist.destroy_process_group()


def test_two_rank_band_render_matches_full_frame(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    scene, cam = go.view_tied_scene(1500, 96, 80, seed=12)
    g = torch.Generator().manual_seed(3)
    grad_color = torch.rand(3, cam.image_height, cam.image_width, generator=g) * 2 - 1
    leaves = {k: v.clone().requires_grad_(True) for k, v in scene.items()}
    color, _, _ = go.rasterize(cam=cam, **leaves)
    (color * grad_color).sum().backward()
    assert torch.equal(got["img"], color.detach())                    # bands are disjoint: bit-identical image
    for k in ("means3D", "opacities", "colors_precomp", "scales"):
        ref = leaves[k].grad
        assert (got[k] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-7, k
    gm = leaves["means3D"].grad
    pose = torch.cat([gm.sum(0), torch.cross(leaves["means3D"].detach(), gm, dim=1).sum(0), gm[:, 2:3].sum(0)])
    assert (got["pose"] - pose).abs().max().item() <= 1e-3 * pose.abs().max().item()
