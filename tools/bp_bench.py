"""sacopt_coded_sizes: time for the buffers of one optimizer round (12 candidates x 15 frames x 2 channels, 22050-sample
windows) and for the 30 channel-frames of the 60 s clip, device-resident residuals; the restated host coder on one core beside it."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_lib
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
torch.cuda.init()
FRAME = 176400
orc = oracle_lib.Oracle()
with sacopt.Context(0) as ctx:
    L, R = stereo_clip(1000, 2 * FRAME)
    res = ctx.predict(2, [(L[:FRAME], R[:FRAME]), (L[FRAME:], R[FRAME:])])
    for label, nbuf, n in (("one optimizer round", 360, 22050), ("two rounds' worth", 720, 22050), ("60 s clip, full frames", 30, FRAME), ("64 full frames", 64, FRAME)):
        src = [res[i % 2]["u%d" % ((i // 2) % 2)] for i in range(nbuf)]
        off = [(37 * i) % (FRAME - n + 1) for i in range(nbuf)]
        host = [s[o:o + n] for s, o in zip(src, off)]
        dev = [torch.from_numpy(np.ascontiguousarray(h)).cuda() for h in host]
        mb = [max(int(h.max()), 1).bit_length() - 1 for h in host]
        bufs = [(d.data_ptr(), n) for d in dev]
        ctx.coded_sizes(bufs, mb); torch.cuda.synchronize()
        t = time.perf_counter(); got = ctx.coded_sizes(bufs, mb); torch.cuda.synchronize(); dt = time.perf_counter() - t
        t = time.perf_counter()
        want = orc.lib.oracle_encode_mono_normal(host[0].ctypes.data_as(C.c_void_p), C.c_int(n), C.c_int(mb[0]), None, C.c_int(0))
        dc = time.perf_counter() - t
        assert got[0] == want
        print(f"{label:26s}: {nbuf:4d} buffers x {n:6d} samples: GPU {dt*1e3:8.1f} ms ({nbuf*n/dt/1e6:6.2f} M samples/s; kernel {ctx.last_timing()['lms_ms']:.1f} ms)   "
              f"host coder, one core: {dc*1e3:7.1f} ms per buffer ({n/dc/1e6:5.2f} M samples/s)", flush=True)
