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
