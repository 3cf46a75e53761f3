"""Cycles per sample of the (1280,640) stage kernels on realistic stage inputs (LPC3 residual of the synthetic clip's
optimizer window), product library: k_lms1c (one chain: two workgroups) through sacopt_lms_pair."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
n = 22050
print("lib", sacopt.LIB_PATH)
with sacopt.Context(0) as ctx:
    for seed, off in ((1000, 77175), (1003, 20000)):
        L, R = stereo_clip(seed, 176400)
        L = L[off:off + n]; R = R[off:off + n]
        p0, p1 = ctx.lpc3_pair(0.997, L, R)
        x0 = L - p0; x1 = R - p1
        for rep in range(2):
            ctx.lms_pair(1280, 640, 1e-5, 5e-6, x0, x1); t = ctx.last_timing()
        print(f"seed {seed}: k_lms1c {2.4e6 * t['lms_ms'] / n:8.0f} cycles/sample ({t['lms_ms']:.2f} ms for {n} samples)", flush=True)
        for (S, D, m1, m2) in ((256, 128, 5e-4, 1e-4), (32, 16, 3e-3, 3e-3), (8, 4, 0.01, 0.01)):
            ctx.lms_pair(S, D, m1, m2, x0, x1); t = ctx.last_timing()
            print(f"   ({S},{D}) {2.4e6 * t['total_ms'] / n:8.0f} cycles/sample", flush=True)
