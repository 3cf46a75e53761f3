"""How often k_lms1c's early prediction is used, redone, and how many verification passes a sample takes on the hostile stage
inputs of tests/test_gpu_parity.py::test_lms_pair_hostile_inputs_vs_oracle (instrumented build: make -C tools prof)."""
import os as _os; _os.environ.setdefault("SACOPT_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_prof", "libsacopt.so"))
import os, sys
import numpy as np
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
from sac_amd import sacopt
rs = np.random.RandomState(77)
n = 3000
x0 = rs.randn(n) * 10.0 ** rs.uniform(-4, 3.2, n)
x1 = rs.randn(n) * 10.0 ** rs.uniform(-4, 3.2, n)
x0[500:700] = 0.0; x1[600:650] = 0.0
x1[1000:1400] = -x0[1000:1400]
x0[1500] = 3e5; x1[1501] = -3e5
x0[2000:2300] = np.round(x0[2000:2300])
with sacopt.Context(0) as ctx:
    q0, q1 = ctx.lms_pair(1280, 640, 1e-5, 5e-6, x0, x1)
    for c, q in enumerate((q0, q1)):
        print(f"channel {c}: predicted {q[64] / n:.3f} of the samples, redone {q[65] / n:.4f}, verification passes per sample {q[67] / n:.3f}")
