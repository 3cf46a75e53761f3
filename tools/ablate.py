import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sac_amd import sacopt
n = 22050
ctx = sacopt.Context(0)
rs = np.random.RandomState(0)
x0 = rs.randn(n) * 300; x1 = 0.5 * x0 + rs.randn(n) * 200
for mode in (0, 8, 16, 16+32, 16+32+64, 2, 2+16+32+64):
    os.environ["SACOPT_DBG"] = str(mode)
    for (S, D, m1, m2) in [(1280, 640, 1e-5, 5e-6)]:
        ctx.lms_pair(S, D, m1, m2, x0, x1); t = ctx.last_timing()
        print(f"mode {mode} (8=serial chain,16=single P2,32=no P1,64=no P0,2=no update) lms{S:5d}: {2.4e6*t['total_ms']/n:8.0f} cyc/sample")
