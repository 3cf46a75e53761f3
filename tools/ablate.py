"""Marginal real-time cost of the phases of k_lms<1280,640> (debug flags of sacopt_lms_pair) on
realistic stage inputs (LPC3 residual of the synthetic clip's optimizer window)."""
import os as _os; _os.environ.setdefault("SACOPT_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_ablate", "libsacopt.so"))  # instrumented build: make -C tools
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
n = 22050
ctx = sacopt.Context(0)
L, R = stereo_clip(1000, 176400)
L = L[77175:77175 + n]; R = R[77175:77175 + n]
p0, p1 = ctx.lpc3_pair(0.997, L, R)
x0 = L - p0; x1 = R - p1
modes = [int(a) for a in sys.argv[1:]] or [0, 8, 16, 16 + 32, 16 + 32 + 64, 2, 4, 2 + 4, 2 + 16 + 32 + 64, 2 + 4 + 16 + 32 + 64]
for mode in modes:
    os.environ["SACOPT_DBG"] = str(mode)
    ctx.lms_pair(1280, 640, 1e-5, 5e-6, x0, x1); t = ctx.last_timing()
    print(f"mode {mode:4d} (256=two-channel kernel,8=serial chain,16=single P2,2=no update): {2.4e6*t['total_ms']/n:8.0f} cyc/sample", flush=True)

ctx.close()
