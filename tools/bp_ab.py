"""A/B of sacopt_coded_sizes across library builds in one process-per-variant run: 360 buffers x 22050 samples. usage: bp_ab.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
torch.cuda.init()
FRAME, nbuf, n = 176400, 360, 22050
with sacopt.Context(0) as ctx:
    L, R = stereo_clip(1000, FRAME)
    res = ctx.predict(2, [(L, R)])[0]
    host = [res["u%d" % (i % 2)][(37 * i) % (FRAME - n + 1):][:n] for i in range(nbuf)]
    dev = [torch.from_numpy(np.ascontiguousarray(h)).cuda() for h in host]
    mb = [max(int(h.max()), 1).bit_length() - 1 for h in host]
    bufs = [(d.data_ptr(), n) for d in dev]
    ctx.coded_sizes(bufs, mb)
    t = time.perf_counter(); got = ctx.coded_sizes(bufs, mb); dt = time.perf_counter() - t
    print(f"{sacopt.LIB_PATH}: {dt*1e3:8.1f} ms  checksum {int(got.sum())}", flush=True)
