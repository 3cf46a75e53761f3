"""Per-launch fixed cost of k_lms1c: time of sacopt_lms_pair for different lengths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
ctx = sacopt.Context(0)
L, R = stereo_clip(1000, 176400)
L = L[77175:77175 + 32768]; R = R[77175:77175 + 32768]
p0, p1 = ctx.lpc3_pair(0.997, L, R)
x0 = L - p0; x1 = R - p1
for n in (512, 1024, 2048, 4096, 8192, 16384, 32768):
    ctx.lms_pair(1280, 640, 1e-5, 5e-6, x0[:n], x1[:n])
    ts = []
    for _ in range(3):
        ctx.lms_pair(1280, 640, 1e-5, 5e-6, x0[:n], x1[:n]); ts.append(ctx.last_timing()["lms_ms"])
    print(f"n {n:6d}: kernel {min(ts):8.3f} ms  {1e3 * min(ts) / n:6.3f} us/sample", flush=True)

ctx.close()
