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


def gather_tiles(local, height, band_rows, group=None):
    """local: (local_rows, W, 4) float32 tensor of this rank's row bands. Returns the full
    (H, W, 4) image on every rank. Ranks may own different row counts (ragged last band):
    buffers are padded to the largest count for the all_gather."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    width, ch = local.shape[1], local.shape[2]
    pad_rows = partition.max_local_rows(height, world, band_rows)
    buf = torch.zeros((pad_rows, width, ch), dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = torch.empty((height, width, ch), dtype=local.dtype, device=local.device)
    for r in range(world):
        rows = torch.as_tensor(partition.rows_of_rank(height, r, world, band_rows), device=local.device)
        if len(rows):
            out[rows] = parts[r][:len(rows)]
    return out


def reduce_samples(local_sum, n_frames_total, group=None):
    """local_sum: (H, W, 4) float32 tensor, the SUM of this rank's raw frames (KL_ACCUM_SUM).
    Returns the mean over all ranks' frames on every rank (alpha forced to 1)."""
    dist.all_reduce(local_sum, op=dist.ReduceOp.SUM, group=group)
    local_sum.div_(float(n_frames_total))
    local_sum[..., 3] = 1.0
    return local_sum


def _device_of(renderer):
    return torch.device("cuda", renderer.device)


def render_tiles(renderer, n_frames, band_rows, first_frame=0, group=None):
    """Image-tile partition on the HIP path: shards `renderer` by row bands, renders, gathers."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    renderer.set_shard(rank, world, band_rows)
    renderer.render_frames(n_frames, first_frame=first_frame)
    rows = renderer.local_rows
    local = torch.empty((rows, renderer.width, 4), dtype=torch.float32, device=_device_of(renderer))
    if rows:
        # the copy runs on the renderer's own stream and returns when it has landed; nothing of torch's is pending on `local`
        renderer.copy_accum_to_device(local.data_ptr(), local.numel() * 4)
    return gather_tiles(local, renderer.height, band_rows, group)


def live_frames(first_frame, n_frames, max_frames):
    """frames of [first_frame, first_frame + n_frames) that are rendered, i.e. lie before the SAMPLES freeze
    (RawFrame.glsl:426: a frame at or past SAMPLES repeats the accumulation and adds nothing to a sum)"""
    if not max_frames:
        return n_frames
    return max(0, min(first_frame + n_frames, max_frames) - first_frame)


def render_sample_ranges(renderer, n_frames, first_frame=0, group=None, out=None):
    """Sample-range partition on the HIP path: every rank sums its frame range over the full image; returns the
    mean over all ranks' frames. `out`: an (H, W, 4) float32 tensor on the renderer's device to reuse between calls."""
    from . import ACCUM_SUM
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = partition.frame_range(first_frame, n_frames, rank, world)
    renderer.set_accum_mode(ACCUM_SUM)
    if hi > lo:
        renderer.render_frames(hi - lo, first_frame=lo)
    shape = (renderer.height, renderer.width, 4)
    local = out if out is not None else torch.empty(shape, dtype=torch.float32, device=_device_of(renderer))
    assert tuple(local.shape) == shape and local.dtype == torch.float32 and local.is_contiguous()
    # `local` may still be read by the previous call's all_reduce / divide on torch's stream, and the copy below runs
    # on the renderer's stream: wait for torch's side first
    torch.cuda.current_stream(local.device).synchronize()
    if hi > lo:
        renderer.copy_accum_to_device(local.data_ptr(), local.numel() * 4)
    else:
        local.zero_()
    live = live_frames(first_frame, n_frames, renderer.get_params().max_frames)
    return reduce_samples(local, max(live, 1), group)
