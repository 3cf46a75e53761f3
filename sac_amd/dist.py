"""Multi-GPU sharding of the optimizer/predictor path.

Frames are independent (fresh predictor and coder state per frame, src/codec.cpp:156-159,240-242),
so the path shards by contiguous frame ranges, one process per GPU, with NO collective on the data
path. The only communication is the final gather of the per-frame results (8 doubles + 4 ints per
frame) to the rank that runs the range coder / writes the file, in frame order.
"""
import numpy as np


def frame_shard(nframes: int, world: int, rank: int):
    """Contiguous, balanced [lo, hi) range of frames for `rank` (the first nframes % world ranks
    get one extra frame). Every frame belongs to exactly one rank; order is preserved."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(nframes, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def gather_frame_rows(local_rows: np.ndarray, nframes: int, world: int, rank: int, group=None):
    """All ranks contribute their shard's rows ([hi-lo, ncol] float64, frame order); every rank gets the
    full [nframes, ncol] table in frame order. Uses one all_gather of equal-sized padded blocks."""
    import torch
    import torch.distributed as dist
    local_rows = np.ascontiguousarray(local_rows, dtype=np.float64)
    ncol = local_rows.shape[1] if local_rows.ndim == 2 else 0
    if world == 1:
        return local_rows.reshape(nframes, ncol)
    per = (nframes + world - 1) // world
    pad = np.zeros((per, ncol), np.float64)
    lo, hi = frame_shard(nframes, world, rank)
    pad[:hi - lo] = local_rows
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(pad).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out = np.zeros((nframes, ncol), np.float64)
    for r in range(world):
        l, h = frame_shard(nframes, world, r)
        out[l:h] = parts[r].cpu().numpy()[:h - l]
    return out
