"""Per-wave phase clocks of the two-channel (1280,640) kernel (library built with -DH_PROF and -DSACOPT_ABLATE so that the kernel can be
forced: tools/_sprof, see below). Per wave, cycles per sample: [0] update tail + products until the first barrier is reached,
[1] from there until the barrier behind the sum (wait + candidate pass; walk + verification on waves 0, 1), [2] wait there, [3] update.
  build:  /tmp/build_variant.sh or: make -C tools ab ABFLAGS="-DH_PROF -DSACOPT_ABLATE" ; SACOPT_LIB=tools/_ab/libsacopt.so"""
import os as _os; _os.environ.setdefault("SACOPT_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_ab", "libsacopt.so"))
_os.environ["SACOPT_DBG"] = "256"   # sacopt_lms_pair: the two-channel kernel
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
ph = (q0[:32].reshape(8, 4) / n).round().astype(int)
print("wave: products | sum | wait | update   (cycles per sample), total", ph[0].sum())
for w in range(8):
    print(w, [int(v) for v in ph[w]])
ctx.close()
