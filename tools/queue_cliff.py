"""The encode step with N extra CU-masked streams (hardware queues) alive in the process: where is the cliff?"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
hip = C.CDLL(None)  # torch's runtime, already mapped
for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
    try:
        hip = C.CDLL(name, mode=os.RTLD_NOLOAD | os.RTLD_GLOBAL); break
    except OSError:
        continue
FRAME = 176400
L, R = stereo_clip(1000, 15 * FRAME)
frames = [(L[f * FRAME:(f + 1) * FRAME], R[f * FRAME:(f + 1) * FRAME]) for f in range(15)]
extra = []
for n in [int(a) for a in sys.argv[1:]] or [0]:
    while len(extra) < n:
        s = C.c_void_p()
        mask = (C.c_uint32 * 8)(*([0xffffffff] * 8))
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(8), mask)
        assert rc == 0, rc
        extra.append(s)
    with sacopt.Context(0) as ctx:
        ctx.encode(2, frames)
        t = time.perf_counter(); ctx.encode(2, frames); dt = time.perf_counter() - t
    print(f"{n} extra CU-masked streams in the process: 15-frame encode call {dt * 1e3:.1f} ms (host buffers)", flush=True)
