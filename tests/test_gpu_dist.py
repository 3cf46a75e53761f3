"""GPU: one frame's candidate population split over two ranks, both ranks on this box's one GPU (own
process and sacopt context each, gloo for the collectives): sac_amd.dist.optimize_split through the C ABI must give
the rows of the library's single-GPU sacopt_optimize_frames, bit for bit."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, profile, lens, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sac_amd import sacopt
    from sac_amd.dist import SacoptEvaluator, optimize_split
    from sac_amd.synth import stereo_clip
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rsqrt_table.npz"))
    ctx = sacopt.Context(0, d["tab"].astype(np.float32), int(d["bits"]))
    frames = [stereo_clip(50 + k, n) for k, n in enumerate(lens)]
    stats = {}
    best, traces = optimize_split(SacoptEvaluator(ctx), profile, frames, sacopt.base_profile(profile), sacopt.NCOEF[profile],
                                  world, rank, stats=stats)
    q.put((rank, best, traces, stats))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("profile,lens", [(2, [40000, 6000]), (1, [36000])])
def test_candidate_split_two_ranks_one_gpu(rsqrt_table, profile, lens):
    import torch.multiprocessing as mp
    from sac_amd import sacopt
    from sac_amd.synth import stereo_clip
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mctx = mp.get_context("spawn")
    q = mctx.Queue()
    procs = [mctx.Process(target=_worker, args=(r, 2, port, profile, lens, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ctx = sacopt.Context(0, *rsqrt_table)
    frames = [stereo_clip(50 + k, n) for k, n in enumerate(lens)]
    best, traces, _ = ctx.optimize(profile, frames)
    ctx.close()
    for rank, b, t, stats in got:
        for k in range(len(lens)):
            assert t[k].tobytes() == np.ascontiguousarray(traces[k]).tobytes(), (rank, k)
            assert b[k].tobytes() == np.ascontiguousarray(best[k]).tobytes(), (rank, k)
