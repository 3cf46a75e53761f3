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
x0 = L - p0; x1 = R - p1          # real stage-1 inputs
os.environ["SACOPT_DBG"] = "128"
q0, q1 = ctx.lms_pair(1280, 640, 1e-5, 5e-6, x0, x1)
names = ["products+bar", "P0+bar", "P1+bar", "walk", "P2", "pred write+bar", "P2 rounds", "update"]
for nm, v in zip(names, q0[:8]):
    print(f"{nm:16s} {v / n:10.1f} per sample")
print("total", sum(q0[:6]) / n + q0[7] / n)
