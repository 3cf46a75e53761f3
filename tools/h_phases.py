"""Per-wave phase clocks of the (256,128) stage kernel (library built with -DH_PROF: make -C tools prof). Per wave, cycles per
sample: [0] products until the first barrier is reached, [1] from there until the barrier behind the sums (wait + guess + MFMA
chain), [2] wait there + look-up of the candidate (+ the second pass on a miss), [3] update."""
import os as _os; _os.environ.setdefault("SACOPT_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_prof", "libsacopt.so"))
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
x0 = L - p0; x1 = R - p1
q0, q1 = ctx.lms_pair(256, 128, 5e-4, 1e-4, x0, x1)
ph = (q0[:16].reshape(4, 4) / n).round().astype(int)
print("wave: products | sums | look-up | update   (cycles per sample), total", ph[0].sum())
for w in range(4):
    print(w, [int(v) for v in ph[w]])
ctx.close()
