"""Debug aid: first mismatch of sacopt_unpredict_stereo_frames against the golden fixture cases."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from sac_amd import sacopt
from test_oracle_golden import unpredict_cases
g = dict(np.load(os.path.join(ROOT, "tests/golden/golden_unpredict.npz")))
d = np.load(os.path.join(ROOT, "tests/golden/rsqrt_table.npz"))
ctx = sacopt.Context(0, d["tab"].astype(np.float32), int(d["bits"]))
for name, prof, e, kw, want in unpredict_cases(g):
    cf = kw.get("coefs")
    fr = dict(e0=e[0], e1=e[1], used0=kw.get("used0"), used1=kw.get("used1"), minmax=kw.get("minmax"))
    v = ctx.unpredict(prof, [fr], coefs=None if cf is None else np.asarray(cf)[None, :])[0]
    for c in (0, 1):
        bad = np.nonzero(v[c] != want[c])[0]
        if len(bad):
            i = bad[0]
            print(name, "ch", c, "first mismatch at", i, "of", len(want[c]), "n_bad", len(bad), "got", v[c][i:i+4], "want", want[c][i:i+4], "e", e[c][i:i+4])
        else:
            print(name, "ch", c, "ok")

ctx.close()
