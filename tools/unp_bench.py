"""Decode timing: sacopt_unpredict_stereo_frames on full frames (encode on the GPU first, check the round trip)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 15
n = int(sys.argv[2]) if len(sys.argv) > 2 else 176400
ctx = sacopt.Context(0)
base = [stereo_clip(1 + i, n) for i in range(min(nfr, 4))]
clips = [base[i % len(base)] for i in range(nfr)]
for prof in (0, 1, 2):
    enc = ctx.predict(prof, clips)
    t = time.time()
    dec = ctx.unpredict(prof, [dict(e0=d["e0"], e1=d["e1"]) for d in enc])
    wall = time.time() - t
    ok = all(np.array_equal(v[0], c[0]) and np.array_equal(v[1], c[1]) for v, c in zip(dec, clips))
    lt = ctx.last_timing(); tot, kms = lt["total_ms"], lt["lms_ms"]
    print(f"profile {prof}: {nfr} frames x {n}: kernel {kms:.1f} ms, call {tot:.1f} ms, wall {wall*1e3:.1f} ms, "
          f"{nfr*n/(tot*1e-3)/1e6:.2f} M samples/s, {kms*1e3/n:.3f} us/sample/frame, round trip {'ok' if ok else 'MISMATCH'}", flush=True)

ctx.close()
