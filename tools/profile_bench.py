"""Encode time (Optimize + full-frame pass) of the 60 s clip for every profile, one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
FRAME, NF = 176400, 15
L, R = stereo_clip(1000, NF * FRAME)
dL = torch.from_numpy(L).cuda(); dR = torch.from_numpy(R).cuda()
outb = [torch.empty_like(dL) for _ in range(6)]
frames = [(dL.data_ptr() + 4 * f * FRAME, dR.data_ptr() + 4 * f * FRAME, FRAME) for f in range(NF)]
outs = [dict(e0=outb[0].data_ptr() + 4 * f * FRAME, e1=outb[1].data_ptr() + 4 * f * FRAME, u0=outb[2].data_ptr() + 4 * f * FRAME,
             u1=outb[3].data_ptr() + 4 * f * FRAME, m0=outb[4].data_ptr() + 4 * f * FRAME, m1=outb[5].data_ptr() + 4 * f * FRAME) for f in range(NF)]
ctx = sacopt.Context(0)
for prof, name in ((0, "fast"), (1, "normal"), (2, "high")):
    ctx.encode(prof, frames, device_out=outs); torch.cuda.synchronize()
    t = time.perf_counter(); ctx.encode(prof, frames, device_out=outs); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{name:7s} encode (optimize + predict + remap) {dt * 1000:8.1f} ms  {NF * FRAME / dt / 1e6:7.2f} M samples/s", flush=True)

ctx.close()
