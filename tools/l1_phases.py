"""Per-wave phase clocks of k_lms1c (library built with -DL1_PROF, SACOPT_LIB=tools/_prof/libsacopt.so). Per wave, cycles
per sample: [0] products until the first barrier is reached, [1] from there until the wave reaches the barrier after the
sum (waiting at the first barrier + P1 + internal barrier + walk/verification for wave 0), [2] waiting at that barrier,
[3] update."""
import os as _os; _os.environ.setdefault("SACOPT_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_prof", "libsacopt.so"))  # instrumented build: make -C tools
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
n = 8192
ctx = sacopt.Context(0)
L, R = stereo_clip(1000, 176400)
L = L[77175:77175 + n]; R = R[77175:77175 + n]
p0, p1 = ctx.lpc3_pair(0.997, L, R)
q0, q1 = ctx.lms_pair(1280, 640, 1e-5, 5e-6, L - p0, R - p1)
ph = (q0[:64].reshape(16, 4) / n).round().astype(int)
print("wave: products | to-barrier-after-sum | wait | update   (cycles per sample), total", ph[0].sum())
for w in range(16):
    print(w, list(ph[w]))


print('early prediction: samples updated with the prediction', q0[64] / n, ' redone', q0[65] / n, ' verification passes per sample', q0[67] / n)
ctx.close()
