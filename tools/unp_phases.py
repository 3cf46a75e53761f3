"""Phase clocks of k_unpredict<high> (library built with -DUN_PROF, SACOPT_LIB=tools/_prof/libsacopt.so): cycles of wave 0
between barriers, per stereo sample. A = products (+ waiting for the slowest update wave), B = exact sums, C = mix/unmap/
reconstruct on wave 7, D = wave 0's own update, K = Cholesky (amortised)."""
import os as _os; _os.environ.setdefault("SACOPT_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_prof", "libsacopt.so"))  # instrumented build: make -C tools
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
ctx = sacopt.Context(0)
n = 8192 + 40
for prof in (1, 2):
    L, R = stereo_clip(5, n)
    enc = ctx.predict(prof, [(L, R)])[0]
    v = ctx.unpredict(prof, [dict(e0=enc["e0"], e1=enc["e1"])])[0]
    ph = v[0][8192:8197].astype(np.int64) * 1024 / n
    wb = v[1][8192:8208].astype(np.int64) * 1024 // n; wd = v[1][8208:8224].astype(np.int64) * 1024 // n
    print("  busy cycles/sample per wave in B:", list(wb), " in D:", list(wd))
    print(f"profile {prof}: cycles/sample A {ph[0]:.0f} B {ph[1]:.0f} C {ph[2]:.0f} D {ph[3]:.0f} K {ph[4]:.0f} total {ph.sum():.0f}; kernel {ctx.last_timing()['lms_ms'] * 1e3 / n:.2f} us/sample")

ctx.close()
