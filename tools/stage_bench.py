#!/usr/bin/env python3
"""Per-stage device timings (one chain each) -- development aid."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 22050
ctx = sacopt.Context(0)
L, R = stereo_clip(3, n)
rs = np.random.RandomState(0)
x0 = rs.randn(n) * 300; x1 = 0.5 * x0 + rs.randn(n) * 200
for rep in range(2):
    for (S, D, m1, m2) in [(1280, 640, 1e-5, 5e-6), (256, 128, 5e-4, 1e-4), (32, 16, 3e-3, 3e-3), (8, 4, 0.01, 0.01)]:
        ctx.lms_pair(S, D, m1, m2, x0, x1); t = ctx.last_timing()
        print(f"lms{S:5d}: {t['lms_ms'] if S==1280 else t['total_ms']:8.2f} ms  {1e3*t['total_ms']/n:7.3f} us/sample  {2.4e3*t['total_ms']/n*1e0:8.0f} cyc/sample")
    ctx.lpc3_pair(0.997, L, R); t = ctx.last_timing(); print(f"lpc3    : {t['total_ms']:8.2f} ms  {1e3*t['total_ms']/n:7.3f} us/sample")
    for prof in (0, 1, 2):
        ctx.predict(prof, [(L, R)]); t = ctx.last_timing(); print(f"predict{prof}: {t['total_ms']:8.2f} ms  {1e3*t['total_ms']/n:7.3f} us/sample")

ctx.close()
