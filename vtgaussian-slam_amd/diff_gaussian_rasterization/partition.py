"""Tile-row image partition for multi-GPU rendering: contiguous bands of 16-pixel tile rows,
Gaussians replicated, one band per rank.  Pure Python, no device work."""
from typing import List, Tuple


def tile_rows_total(image_height: int) -> int:
    return (int(image_height) + 15) // 16


def band_for_rank(image_height: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(begin, end) tile rows of `rank`; the first `rows % world_size` ranks get one extra row.
    Raises if there are more ranks than tile rows (an empty band is not a valid render)."""
    rows = tile_rows_total(image_height)
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    if world_size > rows:
        raise ValueError(f"{world_size} ranks for {rows} tile rows: reduce the number of GPUs")
    per, rem = divmod(rows, world_size)
    b = rank * per + min(rank, rem)
    return b, b + per + (1 if rank < rem else 0)


def all_bands(image_height: int, world_size: int) -> List[Tuple[int, int]]:
    return [band_for_rank(image_height, world_size, r) for r in range(world_size)]


def pixel_rows(band: Tuple[int, int], image_height: int) -> Tuple[int, int]:
    return band[0] * 16, min(band[1] * 16, int(image_height))



def pose7_reduce(points, g_points):
    """{sum g, sum p x g, sum g_z} of camera-frame points and their gradients as ONE 7-float device tensor (two launches):
    what every rank of the tile-row partition contributes to the pose-gradient all-reduce."""
    import ctypes

    import torch

    from . import _I32, _P, _check, _lib, _stream_ptr
    _lib.vtgs_pose_partial_rows.restype, _lib.vtgs_pose_partial_rows.argtypes = ctypes.c_uint32, [_I32]
    _lib.vtgs_pose7_reduce.restype, _lib.vtgs_pose7_reduce.argtypes = ctypes.c_int, [_I32, _P, _P, _P, _P, _P]
    if not points.is_cuda:
        raise RuntimeError("pose7_reduce needs tensors on a HIP device (torch 'cuda'); no CPU path exists")
    p = points.detach().to(torch.float32).contiguous()
    g = g_points.detach().to(torch.float32).contiguous()
    n = p.shape[0]
    rows = int(_lib.vtgs_pose_partial_rows(n))
    partials = torch.empty((max(rows, 1), 7), dtype=torch.float32, device=p.device)
    out = torch.empty(7, dtype=torch.float32, device=p.device)
    _check(_lib.vtgs_pose7_reduce(n, p.data_ptr(), g.data_ptr(), partials.data_ptr(), out.data_ptr(), _stream_ptr(p.device)),
           "vtgs_pose7_reduce")
    return out
