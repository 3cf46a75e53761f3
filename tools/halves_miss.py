"""(256,128) stage on realistic stage inputs: time per sample and how often the exact first half misses wave 1's 32
candidates (debug flag 128 of sacopt_lms_pair)."""
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
    x0 = L - p0; x1 = R - p1
    os.environ.pop("SACOPT_DBG", None)
    q0, q1 = ctx.lms_pair(1280, 640, 1e-5, 5e-6, x0, x1)
    y0 = x0 - q0; y1 = x1 - q1        # real stage-2 inputs
    ctx.lms_pair(256, 128, 5e-4, 1e-4, y0, y1); t = ctx.last_timing()
    os.environ["SACOPT_DBG"] = "128"
    r0, r1 = ctx.lms_pair(256, 128, 5e-4, 1e-4, y0, y1)
    os.environ.pop("SACOPT_DBG", None)
    print(f"seed {seed}: {1e3 * t['total_ms'] / n:.3f} us/sample ({2.4e3 * t['total_ms'] / n:.0f} cycles), misses per sample {r0[0] / n:.4f}")

ctx.close()
