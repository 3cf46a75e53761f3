"""CPU, world_size 2 over gloo: the N>1 path (frame sharding + ordered gather of per-frame results).
The per-frame work is replaced by a deterministic stand-in here (no GPU in this tier); what is under
test is that every frame is processed exactly once and results come back in frame order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sac_amd.dist import frame_shard, gather_frame_rows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frame_result(f):  # stand-in for (8 selected coefs, cost) of frame f
    return np.array([f, f * 0.5, 1000.0 + f, -f], np.float64)


def _worker(rank, world, port, nframes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = frame_shard(nframes, world, rank)
    local = np.stack([_frame_result(f) for f in range(lo, hi)]) if hi > lo else np.zeros((0, 4))
    full = gather_frame_rows(local, nframes, world, rank)
    q.put((rank, lo, hi, full))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nframes", [15, 4, 1])
def test_two_rank_shard_and_gather(nframes):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_frame_result(f) for f in range(nframes)])
    covered = []
    for rank, lo, hi, full in got:
        assert np.array_equal(full, want), rank
        covered += list(range(lo, hi))
    assert sorted(covered) == list(range(nframes))


def test_frame_shard_properties():
    for n in (0, 1, 7, 15, 4800):
        for w in (1, 2, 4, 8):
            spans = [frame_shard(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [h - l for l, h in spans]
            assert max(sizes) - min(sizes) <= 1
