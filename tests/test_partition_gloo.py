"""CPU suite for the N>1 path: the partition arithmetic and both collectives, with two gloo
ranks. The per-rank pixels come from the CPU oracle here (no GPU in this suite); on the GPU
the same helpers are fed by the HIP renderer (tests/test_gpu_scene.py checks its shards)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, free_port
from klrender import partition


def test_rows_partition_is_exact_cover():
    for height, world, band in ((54, 2, 8), (33, 3, 5), (36, 8, 4), (10, 4, 16), (1080, 8, 8), (7, 8, 1)):
        seen = np.zeros(height, dtype=int)
        for r in range(world):
            rows = partition.rows_of_rank(height, r, world, band)
            assert np.all(np.diff(rows) > 0)
            seen[rows] += 1
        assert np.all(seen == 1)
        assert partition.max_local_rows(height, world, band) >= -(-height // world) - band


def test_frame_ranges_cover():
    for first, n, world in ((0, 64, 8), (5, 10, 3), (0, 3, 8)):
        got = []
        for r in range(world):
            lo, hi = partition.frame_range(first, n, r, world)
            got += list(range(lo, hi))
        assert got == list(range(first, first + n))


def test_untile_index_is_the_inverse_of_the_row_band_layout():
    """the one-gather un-tile (klrender/dist.py TileGather, and the k_untile kernel of kl_group_gather, which computes the same
    expression per pixel): row y of the image sits at row untile_index[y] of the all-gathered, padded per-rank blocks"""
    rng = np.random.default_rng(3)
    for height, world, band in ((54, 2, 8), (33, 3, 5), (50, 3, 8), (36, 8, 4), (10, 4, 16), (2160, 8, 8), (7, 8, 1)):
        full = rng.random((height, 3, 4), dtype=np.float32)
        pad = partition.max_local_rows(height, world, band)
        gathered = np.zeros((world * pad, 3, 4), dtype=np.float32)
        for r in range(world):
            rows = partition.rows_of_rank(height, r, world, band)
            gathered[r * pad:r * pad + len(rows)] = full[rows]
        idx = partition.untile_index(height, world, band)
        assert np.array_equal(gathered[idx], full)
        parts = [gathered[r * pad:(r + 1) * pad] for r in range(world)]
        assert np.array_equal(partition.stitch(parts, height, world, band), full)


def test_live_frames_counts_the_frames_before_the_freeze():
    from klrender import dist as kdist
    assert kdist.live_frames(0, 64, 0) == 64 and kdist.live_frames(0, 64, 30) == 30
    assert kdist.live_frames(20, 20, 30) == 10 and kdist.live_frames(40, 8, 30) == 0


def _worker(rank, world, port, height, width, band_rows, n_frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "path-tracer_amd")):
        sys.path.insert(0, p)
    import kloracle as ko
    from klrender import dist as kdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = ko.Scene(builtin=True)
        cam = ko.default_camera()
        p = ko.default_params(width=width, height=height, mode=ko.MODE_PT, max_frames=0, max_bounces=6)
        # image tiles: this rank's rows of the running mean
        full = ko.render_mean(sc, cam, p, n_frames, nthreads=2)
        rows = partition.rows_of_rank(height, rank, world, band_rows)
        local = torch.from_numpy(np.ascontiguousarray(full[rows]))
        img = kdist.gather_tiles(local, height, band_rows).numpy()
        np.save(os.path.join(out_dir, "tiles_%d.npy" % rank), img)
        # sample ranges: this rank's sum of raw frames
        lo, hi = partition.frame_range(0, n_frames, rank, world)
        acc = np.zeros((height, width, 4), dtype=np.float32)
        for f in range(lo, hi):
            acc += sc.render_raw(cam, p, f, nthreads=2)
        mean = kdist.reduce_samples(torch.from_numpy(acc), n_frames).numpy()
        np.save(os.path.join(out_dir, "samples_%d.npy" % rank), mean)
        if rank == 0:
            np.save(os.path.join(out_dir, "full.npy"), full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height,band_rows", [(27, 4), (20, 16)])
def test_two_rank_gather_and_reduce(tmp_path, height, band_rows):
    world, width, n_frames = 2, 48, 6
    port = free_port()
    mp.spawn(_worker, args=(world, port, height, width, band_rows, n_frames, str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    for r in range(world):
        tiles = np.load(tmp_path / ("tiles_%d.npy" % r))
        assert np.array_equal(tiles.view(np.uint32), full.view(np.uint32))      # tiles: bit-identical
        samples = np.load(tmp_path / ("samples_%d.npy" % r))
        assert np.allclose(samples[..., :3], full[..., :3], rtol=2e-5, atol=1e-6)  # sums: fp32 re-association only
        assert np.all(samples[..., 3] == 1.0)
