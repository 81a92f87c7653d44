"""Sharding the render across the GPUs of one node.  No counterpart in the reference, which is
single-GPU; the scene is replicated, the only exchange is the finished framebuffer.

Two partitions, both free of data-path collectives while rendering:

  tiles    8x8 screen tiles dealt round-robin (tile t -> rank t % world).  Every rank renders ALL frames for its
           own pixels with the same per-frame seeds, so each pixel is bit-identical to the 1-GPU image.  Exchange:
           one gather of packed RGBA32F tiles to rank 0.

  samples  frames dealt round-robin (frame f -> rank f % world) over the FULL image; rank r keeps the running mean
           of its own n_r frames.  Exchange: one sum-reduce of (n_r / N) * mean_r to rank 0.  The sum order differs
           from the sequential running mean, so pixels agree to ~1e-6 relative, not bit for bit.

The functions take torch tensors and a torch.distributed module, so the same code runs over RCCL on GPUs and over
gloo on CPU tensors (tests/test_multigpu_cpu.py).
"""
from __future__ import annotations

import numpy as np


def frames_of_rank(first_frame: int, n_frames: int, rank: int, world: int) -> np.ndarray:
    """Global frame numbers rank `rank` renders under the `samples` partition."""
    f = np.arange(first_frame, first_frame + n_frames, dtype=np.int64)
    return f[(f % world) == rank]


def render_sample_share(ctx, seeds_by_frame, first_frame: int, n_frames: int, rank: int, world: int, bounces: int = 5,
                        local_done: int = 0) -> int:
    """Render this rank's frames.  seeds_by_frame[i] is the host-PCG seed of global frame first_frame + i (the same
    table on every rank, so the per-pixel PCG streams equal the sequential run's).  The blend weight uses the LOCAL
    frame count so that the rank's framebuffer is the mean of its own samples.  Returns the new local count."""
    for f in frames_of_rank(first_frame, n_frames, rank, world):
        ctx.render_frame(int(seeds_by_frame[f - first_frame]), local_done, bounces)
        local_done += 1
    return local_done


def gather_tiles(dist, packed, rank: int, world: int, dst: int = 0):
    """packed: [n_max, 4] float32 tensor holding this rank's pixels in pt_partition_pixels order (rows past the
    rank's own count are padding).  Returns the list of all ranks' tensors on dst, None elsewhere."""
    bufs = [packed.new_empty(packed.shape) for _ in range(world)] if rank == dst else None
    dist.gather(packed, bufs, dst=dst)
    return bufs


def assemble_tiles(bufs, width: int, height: int, world: int, partition_pixels) -> np.ndarray:
    """Host-side twin of pt_unpack_from: scatter every rank's packed pixels into a full [H, W, 4] image."""
    img = np.zeros((height * width, 4), np.float32)
    for r in range(world):
        pix = partition_pixels(width, height, 0, r, world)
        img[pix] = bufs[r][: pix.size].cpu().numpy()
    return img.reshape(height, width, 4)


def reduce_samples(dist, mean_rgba, n_local: int, n_total: int, dst: int = 0):
    """mean_rgba: this rank's running-mean image as a tensor (modified in place).  After the call rank dst holds
    sum_r (n_r / N) * mean_r = the mean over all N samples; alpha is restored to 1."""
    mean_rgba.mul_(float(n_local) / float(max(n_total, 1)))
    dist.reduce(mean_rgba, dst=dst, op=dist.ReduceOp.SUM)
    mean_rgba[..., 3] = 1.0
    return mean_rgba
