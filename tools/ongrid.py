"""How far from the DPP-tree guess do the exact boundary values of k_lms1c lie? (ablation build, debug flag 2048 of
sacopt_lms_pair). Per sample: the largest |offset - 8| over the 14 boundaries, in ulps of the boundary's grid; bins
<=7 (= inside the 16 candidates everywhere), <=15, <=23, <=31, <=63, <=127, more, and 'some boundary value not on the
guess's binade grid'. Tells how many candidates a P1 pass would need for the walk to stay on exact table entries."""
import os as _os; _os.environ.setdefault("SACOPT_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_ablate", "libsacopt.so"))  # instrumented build: make -C tools
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
n = 22050
ctx = sacopt.Context(0)
for seed, off in ((1000, 77175), (1003, 20000), (1007, 120000)):
    L, R = stereo_clip(seed, 176400)
    L = L[off:off + n]; R = R[off:off + n]
    p0, p1 = ctx.lpc3_pair(0.997, L, R)
    os.environ["SACOPT_DBG"] = "2048"
    q0, q1 = ctx.lms_pair(1280, 640, 1e-5, 5e-6, L - p0, R - p1)
    os.environ.pop("SACOPT_DBG", None)
    for q in (q0, q1):
        h = q[:9].astype(np.int64)
        tot = h[:8].sum()
        print(f"seed {seed}: samples {tot}  <=7 {h[0]/tot:.3f}  <=15 {h[1]/tot:.3f}  <=23 {h[2]/tot:.3f}  <=31 {h[3]/tot:.3f}  <=63 {h[4]/tot:.3f}  <=127 {h[5]/tot:.3f}  more {h[6]/tot:.3f}  off-grid {h[7]/tot:.3f}   retried {h[8]/tot:.3f}", flush=True)
ctx.close()
