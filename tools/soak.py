"""Soak: PredictStereoFrame (high and normal profile) of many random clips of odd lengths through the library, bit-compared with
the C restatement (oracle/) -- block edges, chunk edges, the pipelined full-frame path, both (1280,640) kernels.
usage: soak.py [seed] [nclips]   (test infrastructure: it uses the oracle)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nclips = int(sys.argv[2]) if len(sys.argv) > 2 else 48
rs = np.random.RandomState(seed)
oracle_lib.build_oracle()
orc = oracle_lib.Oracle()
_t = np.load(os.path.join(ROOT, 'tests', 'golden', 'rsqrt_table.npz'), allow_pickle=False)
tab, bits = _t['tab'].astype(np.float32), int(_t['bits'])   # the fixture's rsqrtss table on both sides
orc.set_rsqrt_table(tab, bits)
bad = 0
with sacopt.Context(0, tab, bits) as ctx:
    for prof in (2, 1):
        lens = [int(rs.choice([rs.randint(1, 600), rs.randint(600, 5000), rs.randint(5000, 17000), rs.randint(17000, 42000)], p=[.2, .4, .25, .15])) for _ in range(nclips)]
        clips = []
        for i, n in enumerate(lens):
            L, R = stereo_clip(int(rs.randint(1, 1 << 30)), n)
            clips.append((L, R))
        t = time.time()
        # one call of all clips (> 20 chains: two-channel kernel) and single-clip calls (k_lms1c)
        allres = ctx.predict(prof, clips)
        for i, (L, R) in enumerate(clips):
            ref = orc.predict(prof, L, R)
            one = ctx.predict(prof, [(L, R)])[0] if i % 3 == 0 else None
            for k in ("e0", "e1"):
                if not np.array_equal(allres[i][k], ref[k]): bad += 1; print(f"MISMATCH batch prof {prof} clip {i} n {len(L)} {k}", flush=True)
                if one is not None and not np.array_equal(one[k], ref[k]): bad += 1; print(f"MISMATCH single prof {prof} clip {i} n {len(L)} {k}", flush=True)
        print(f"profile {prof}: {nclips} clips, {sum(lens)} samples, {time.time() - t:.1f} s, mismatches so far {bad}", flush=True)
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
