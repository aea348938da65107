"""Work partitioning over GPUs: which rows / frames a rank owns and how per-rank buffers
stitch back into one image. Pure index arithmetic (no GPU, no torch), shared by the HIP
path and by the CPU tests.

Image tiles are row bands: band b = rows [b*band_rows, (b+1)*band_rows) belongs to rank
b % world (kl_set_shard in include/Math/kl_render.h). A rank's buffers hold its rows in
increasing row order. Sample ranges split the frame indices: rank g renders frames
[first + g*n/world, first + (g+1)*n/world) as a sum."""
import numpy as np


def rows_of_rank(height, rank, world, band_rows):
    """global row indices owned by `rank`, in the order its local buffer stores them"""
    rows = np.arange(height)
    return rows[(rows // band_rows) % world == rank]


def max_local_rows(height, world, band_rows):
    return max(len(rows_of_rank(height, r, world, band_rows)) for r in range(world))


def stitch(parts, height, world, band_rows):
    """parts[r] = (rows_r, width, C) local buffer of rank r (may carry padding rows at the end)"""
    width, ch = parts[0].shape[1], parts[0].shape[2]
    out = np.zeros((height, width, ch), dtype=parts[0].dtype)
    for r in range(world):
        rows = rows_of_rank(height, r, world, band_rows)
        out[rows] = parts[r][:len(rows)]
    return out


def frame_range(first, n, rank, world):
    """frames [lo, hi) of rank `rank` when n frames starting at `first` are split over `world` ranks"""
    lo = first + (n * rank) // world
    hi = first + (n * (rank + 1)) // world
    return lo, hi


def untile_index(height, world, band_rows):
    """For the all-gathered buffer of `world` padded per-rank blocks (each max_local_rows rows): the row of that buffer
    which holds image row y, for every y. One index gather with it un-tiles the image (what a HIP un-tile kernel computes
    per pixel: rank = (y // band_rows) % world, local row = (y // band_rows // world) * band_rows + y % band_rows)."""
    pad = max_local_rows(height, world, band_rows)
    y = np.arange(height)
    band = y // band_rows
    return (band % world) * pad + (band // world) * band_rows + y % band_rows
