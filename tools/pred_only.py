"""Full-frame-pass-only calls on the bench clip (for a rocprofv3 --kernel-trace timeline: tools/prof_timeline.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
FRAME, NF = 176400, 15
L, R = stereo_clip(1000, NF * FRAME)
dL = torch.from_numpy(L).cuda(); dR = torch.from_numpy(R).cuda()
outb = [torch.empty_like(dL) for _ in range(4)]
frames = [(dL.data_ptr() + 4 * f * FRAME, dR.data_ptr() + 4 * f * FRAME, FRAME) for f in range(NF)]
outs = [dict(e0=outb[0].data_ptr() + 4 * f * FRAME, e1=outb[1].data_ptr() + 4 * f * FRAME,
             u0=outb[2].data_ptr() + 4 * f * FRAME, u1=outb[3].data_ptr() + 4 * f * FRAME) for f in range(NF)]
with sacopt.Context(0) as ctx:
    ctx.predict(2, frames, device_out=outs); torch.cuda.synchronize()
    time.sleep(0.05)
    t = time.perf_counter(); ctx.predict(2, frames, device_out=outs); torch.cuda.synchronize()
    print(f"predict {(time.perf_counter() - t) * 1e3:.1f} ms")
