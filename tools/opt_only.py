"""Optimize-only calls on the bench clip (for a rocprofv3 --kernel-trace timeline of the optimizer's rounds)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
FRAME, NF = 176400, int(sys.argv[1]) if len(sys.argv) > 1 else 15
L, R = stereo_clip(1000, NF * FRAME)
dL = torch.from_numpy(L).cuda(); dR = torch.from_numpy(R).cuda()
frames = [(dL.data_ptr() + 4 * f * FRAME, dR.data_ptr() + 4 * f * FRAME, FRAME) for f in range(NF)]
with sacopt.Context(0) as ctx:
    ctx.optimize(2, frames); torch.cuda.synchronize()
    t = time.perf_counter(); ctx.optimize(2, frames); torch.cuda.synchronize()
    print(f"optimize {(time.perf_counter() - t) * 1e3:.1f} ms")
