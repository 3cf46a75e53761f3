"""Does an initialised RCCL communicator (its streams / hardware queues) slow the library's calls down? One rank, backend nccl."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.distributed as dist
torch.cuda.init()
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
FRAME = 176400
use = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 15
if use:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    x = torch.ones(1024, device="cuda:0")
    dist.all_reduce(x); outl = [torch.empty_like(x)]; dist.all_gather(outl, x); dist.barrier()
    torch.cuda.synchronize()
L, R = stereo_clip(1000, nfr * FRAME)
dL = torch.from_numpy(L).cuda(); dR = torch.from_numpy(R).cuda()
o = [torch.empty_like(dL) for _ in range(6)]
frames = [(dL.data_ptr() + 4 * f * FRAME, dR.data_ptr() + 4 * f * FRAME, FRAME) for f in range(nfr)]
outs = [{k: o[i].data_ptr() + 4 * f * FRAME for i, k in enumerate(("e0", "e1", "u0", "u1", "m0", "m1"))} for f in range(nfr)]
with sacopt.Context(0) as ctx:
    ctx.encode(2, frames, device_out=outs); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        if use: dist.barrier()
        t = time.perf_counter(); ctx.encode(2, frames, device_out=outs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        if use:
            x = torch.ones(1024, device="cuda:0"); dist.all_gather([torch.empty_like(x)], x)
    print(f"rccl={use} frames={nfr}: " + " ".join(f"{1e3 * t:.1f}" for t in ts) + " ms")
if use: dist.destroy_process_group()
