"""Per-wave phase clocks of k_lms2 (two-channel (1280,640) kernel) with REPL copies of one chain in the launch
(-DH_PROF -DSACOPT_ABLATE build: make -C tools ab ABFLAGS="-DH_PROF -DSACOPT_ABLATE", moved to tools/live_prof/).
Per wave, cycles per sample: products until the barrier | sum until its barrier | wait there | update."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SACOPT_LIB", os.path.join(ROOT, "tools", "live_prof", "libsacopt.so"))
sys.path[:0] = [ROOT]
os.environ["SACOPT_DBG"] = "256"   # sacopt_lms_pair: the two-channel kernel whatever the number of chains
import numpy as np
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
n = 4096
L, R = stereo_clip(1000, 176400)
L = L[77175:77175 + n]; R = R[77175:77175 + n]
for repl in [int(a) for a in sys.argv[1:]] or [1, 224, 448]:
    os.environ["SACOPT_REPL"] = str(repl)
    ctx = sacopt.Context(0)
    p0, p1 = ctx.lpc3_pair(0.997, L, R)
    for _ in range(2):
        q0, q1 = ctx.lms_pair(1280, 640, 1e-5, 5e-6, L - p0, R - p1)
        t = ctx.last_timing()
    tb = (n - 1) & ~127
    ph = (q0[tb:tb + 32].reshape(8, 4) / n).round().astype(int)
    print(f"REPL={repl}: kernel {t['lms_ms']:.2f} ms = {t['lms_ms'] * 2.4e6 / n:.0f} cycles/sample; per wave: products | sum | wait | update (clock64 units per sample)")
    for w in range(8):
        print("   ", w, [int(v) for v in ph[w]], int(ph[w].sum()))
    ctx.close()
