"""Are two workgroups of k_lms2 resident on one CU at the same time? REPL copies of one chain in one launch
(build: make -C tools ab ABFLAGS="-DS2_OCC -DSACOPT_ABLATE", moved to tools/live_occ/); every workgroup records its CU
(HW_ID, XCC_ID) and the wall-clock interval it was resident; counts, per CU, the workgroups and the pairs that overlapped."""
import os, sys, ctypes as C, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SACOPT_LIB", os.path.join(ROOT, "tools", "live_occ", "libsacopt.so"))
sys.path[:0] = [ROOT]
os.environ["SACOPT_DBG"] = "256"   # sacopt_lms_pair: the two-channel kernel whatever the number of chains
import numpy as np
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
n = 2048
L, R = stereo_clip(1000, 176400)
L = L[77175:77175 + n]; R = R[77175:77175 + n]
for repl in [int(a) for a in sys.argv[1:]] or [256, 512]:
    os.environ["SACOPT_REPL"] = str(repl)
    ctx = sacopt.Context(0)
    p0, p1 = ctx.lpc3_pair(0.997, L, R)
    for _ in range(2):
        ctx.lms_pair(1280, 640, 1e-5, 5e-6, L - p0, R - p1)
        t = ctx.last_timing()
    rec = np.zeros(4 * repl)
    assert sacopt.lib().sacopt_s2occ_read(rec.ctypes.data_as(C.c_void_p), C.c_int(4 * repl)) == 0
    rec = rec.reshape(repl, 4)
    cu = collections.defaultdict(list)
    for t0, t1, hw, xcc in rec:
        hw = int(hw)
        cu[(int(xcc), (hw >> 13) & 7, (hw >> 8) & 15)].append((t0, t1))   # XCC, SE_ID, CU_ID
    pairs = sum(1 for v in cu.values() for i in range(len(v)) for j in range(i) if min(v[i][1], v[j][1]) - max(v[i][0], v[j][0]) > 0.5 * min(v[i][1] - v[i][0], v[j][1] - v[j][0]))
    dur = rec[:, 1] - rec[:, 0]
    print(f"REPL={repl}: kernel {t['lms_ms']:.2f} ms; {len(cu)} CUs used, workgroups per CU {collections.Counter(len(v) for v in cu.values())}, "
          f"pairs resident together (> half of the shorter one's time): {pairs}; residency per workgroup {dur.mean() / 100:.0f} us mean, {dur.min() / 100:.0f}..{dur.max() / 100:.0f} us (100 MHz clock)")
    ctx.close()
