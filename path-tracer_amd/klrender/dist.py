"""Collectives of the multi-GPU partitions, over torch.distributed (backend "nccl" = RCCL on
ROCm, one process per GPU; "gloo" in the CPU tests). Two exchange steps exist on this path
and nothing else crosses GPUs:

  * image tiles    -> one all_gather of the per-rank accumulation rows, then a local un-tile;
  * sample ranges  -> one all_reduce(sum) of the per-rank accumulation sums, then a divide.

Both run once per report interval (after all frames), never per sample."""
import numpy as np
import torch
import torch.distributed as dist

from . import partition


class TileGather:
    """Buffers and index of the image-tile exchange for one (height, width, world, band_rows), allocated once: the padded
    send block, the all-gathered blocks and the row index that un-tiles them with ONE gather (no per-rank Python loop)."""

    def __init__(self, height, width, band_rows, device, group=None, channels=4, dtype=torch.float32):
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.height, self.width, self.band_rows = height, width, band_rows
        self.pad_rows = partition.max_local_rows(height, self.world, band_rows)
        self.rows = len(partition.rows_of_rank(height, self.rank, self.world, band_rows))
        self.gathered = torch.empty((self.world * self.pad_rows, width, channels), dtype=dtype, device=device)
        # this rank's block of the gathered buffer doubles as its send buffer (in-place all_gather: no staging copy)
        self.send = self.gathered[self.rank * self.pad_rows:(self.rank + 1) * self.pad_rows]
        if self.rows < self.pad_rows:
            self.send[self.rows:].zero_()
        self.index = torch.as_tensor(partition.untile_index(height, self.world, band_rows), device=device, dtype=torch.long)
        self.full = torch.empty((height, width, channels), dtype=dtype, device=device)

    def local_view(self):
        """(local_rows, W, C) view of the send block: where this rank's rows go before gather()"""
        return self.send[:self.rows]

    def gather(self):
        dist.all_gather_into_tensor(self.gathered, self.send, group=self.group)
        torch.index_select(self.gathered, 0, self.index, out=self.full)
        return self.full


def gather_tiles(local, height, band_rows, group=None):
    """local: (local_rows, W, 4) float32 tensor of this rank's row bands. Returns the full (H, W, 4) image on every
    rank. Ranks may own different row counts (ragged last band): blocks are padded to the largest count."""
    tg = TileGather(height, local.shape[1], band_rows, local.device, group, channels=local.shape[2], dtype=local.dtype)
    tg.local_view().copy_(local)
    return tg.gather()


def reduce_samples(local_sum, n_frames_total, group=None):
    """local_sum: (H, W, 4) float32 tensor, the SUM of this rank's raw frames (KL_ACCUM_SUM).
    Returns the mean over all ranks' frames on every rank (alpha forced to 1)."""
    dist.all_reduce(local_sum, op=dist.ReduceOp.SUM, group=group)
    local_sum.div_(float(n_frames_total))
    local_sum[..., 3] = 1.0
    return local_sum


def _device_of(renderer):
    return torch.device("cuda", renderer.device)


def _split_timing(renderer, timing, t0):
    """`timing` (a dict, measurement runs only): wait for the render, note its wall time, return the exchange's start"""
    import time
    if timing is None:
        return None
    renderer.synchronize()
    t1 = time.perf_counter()
    timing["render_s"] = timing.get("render_s", 0.0) + (t1 - t0)
    return t1


def _end_timing(timing, device, t1):
    import time
    if timing is None:
        return
    torch.cuda.synchronize(device)
    timing["exchange_s"] = timing.get("exchange_s", 0.0) + (time.perf_counter() - t1)


def render_tiles(renderer, n_frames, band_rows, first_frame=0, group=None, gatherer=None, timing=None):
    """Image-tile partition on the HIP path: shards `renderer` by row bands, renders, gathers. `gatherer`: a TileGather
    to reuse between calls (its buffers are then not reallocated). `timing`: a dict that receives the wall time of the
    render and of the exchange (all-gather + un-tile) separately, at the price of a wait between the two."""
    import time
    t0 = time.perf_counter()
    from . import ACCUM_MEAN, ACCUM_SUM
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if renderer.accum_mode == ACCUM_SUM:    # left over from a sample-range render: tiles hold the running mean
        renderer.set_accum_mode(ACCUM_MEAN)
    if renderer.shard != (rank, world, band_rows):
        renderer.set_shard(rank, world, band_rows)
    else:
        renderer.reset()
    renderer.render_frames(n_frames, first_frame=first_frame)
    tg = gatherer or TileGather(renderer.height, renderer.width, band_rows, _device_of(renderer), group)
    assert tg.rows == renderer.local_rows
    # torch's stream may still read the block (previous gather): wait, then let the renderer's stream copy into it
    torch.cuda.current_stream(tg.gathered.device).synchronize()
    t1 = _split_timing(renderer, timing, t0)
    if tg.rows:
        view = tg.local_view()
        renderer.copy_accum_to_device(view.data_ptr(), view.numel() * 4)
    out = tg.gather()
    _end_timing(timing, tg.gathered.device, t1)
    return out


def live_frames(first_frame, n_frames, max_frames):
    """frames of [first_frame, first_frame + n_frames) that are rendered, i.e. lie before the SAMPLES freeze
    (RawFrame.glsl:426: a frame at or past SAMPLES repeats the accumulation and adds nothing to a sum)"""
    if not max_frames:
        return n_frames
    return max(0, min(first_frame + n_frames, max_frames) - first_frame)


def render_sample_ranges(renderer, n_frames, first_frame=0, group=None, out=None, timing=None):
    """Sample-range partition on the HIP path: every rank sums its frame range over the full image; returns the
    mean over all ranks' frames. `out`: an (H, W, 4) float32 tensor on the renderer's device to reuse between calls.
    `timing`: as in render_tiles (render and all-reduce + divide timed separately)."""
    import time
    from . import ACCUM_SUM
    t0 = time.perf_counter()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = partition.frame_range(first_frame, n_frames, rank, world)
    if renderer.shard[1] != 1:              # left over from an image-tile render: sample ranges cover the whole image
        renderer.set_shard(0, 1, 1)
    renderer.set_accum_mode(ACCUM_SUM)
    if hi > lo:
        renderer.render_frames(hi - lo, first_frame=lo)
    shape = (renderer.height, renderer.width, 4)
    local = out if out is not None else torch.empty(shape, dtype=torch.float32, device=_device_of(renderer))
    assert tuple(local.shape) == shape and local.dtype == torch.float32 and local.is_contiguous()
    # `local` may still be read by the previous call's all_reduce / divide on torch's stream, and the copy below runs
    # on the renderer's stream: wait for torch's side first
    torch.cuda.current_stream(local.device).synchronize()
    t1 = _split_timing(renderer, timing, t0)
    if hi > lo:
        renderer.copy_accum_to_device(local.data_ptr(), local.numel() * 4)
    else:
        local.zero_()
    live = live_frames(first_frame, n_frames, renderer.get_params().max_frames)
    res = reduce_samples(local, max(live, 1), group)
    _end_timing(timing, local.device, t1)
    return res


def upload_scene_once(renderer, scene, src=0, group=None):
    """The tree is built ONCE per job: rank `src` uploads `scene` (spheres, quads, tris, materials; the other ranks may pass
    None), exports the packed scene -- records, tree, shading records, flat-quad table -- as one block of device memory,
    and one broadcast hands it to every other rank, which imports it instead of building the same tree again (N ranks of a
    node would otherwise run N multi-threaded host builds side by side). With the "nccl" backend the block travels from
    GPU to GPU over RCCL; with "gloo" (CPU tests, rehearsals) through host memory. Returns the block (a uint8 tensor on the
    renderer's device): another context on this device can import it again without a new build or broadcast."""
    rank = dist.get_rank(group)
    dev = _device_of(renderer)
    nbytes = torch.zeros(1, dtype=torch.int64)
    if rank == src:
        renderer.upload_scene(*scene)
        nbytes[0] = renderer.scene_blob_bytes()
    via_host = dist.get_backend(group) != "nccl"
    size = nbytes if via_host else nbytes.to(dev)
    dist.broadcast(size, src=src, group=group)
    n = int(size.item())
    block = torch.empty(n, dtype=torch.uint8, device=dev)
    if rank == src:
        renderer.scene_export(block.data_ptr(), n)
    if via_host:
        host = block.cpu() if rank == src else torch.empty(n, dtype=torch.uint8)
        dist.broadcast(host, src=src, group=group)
        if rank != src:
            block.copy_(host)
    else:
        dist.broadcast(block, src=src, group=group)
    if rank != src:
        torch.cuda.synchronize(dev)
        renderer.scene_import(block.data_ptr(), n)
    return block
