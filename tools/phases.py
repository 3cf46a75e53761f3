"""P2 passes per sample of k_lms1c (debug flag 128 of sacopt_lms_pair) on realistic stage inputs."""
import os as _os; _os.environ.setdefault("SACOPT_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_ablate", "libsacopt.so"))  # instrumented build: make -C tools
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
n = 22050
ctx = sacopt.Context(0)
for seed, off in ((1000, 77175), (1003, 20000)):
    L, R = stereo_clip(seed, 176400)
    L = L[off:off + n]; R = R[off:off + n]
    p0, p1 = ctx.lpc3_pair(0.997, L, R)
    x0 = L - p0; x1 = R - p1          # real stage-1 inputs
    for extra in [int(a) for a in sys.argv[1:]] or [0]:
        os.environ["SACOPT_DBG"] = str(128 + extra)
        q0, q1 = ctx.lms_pair(1280, 640, 1e-5, 5e-6, x0, x1)
        a0, a1 = int(q0[0]), int(q1[0])
        print(f"seed {seed} dbg {extra}: P2 passes per sample: ch0 {(a0 & 0xffffffff) / n:.3f}  ch1 {(a1 & 0xffffffff) / n:.3f}"
              f"   walk iterations per sample (dbg 1024): ch0 {(a0 >> 32) / n:.3f}  ch1 {(a1 >> 32) / n:.3f}")

ctx.close()
