"""Is the optimizer worth running? Coded size of frames predicted with the base profile against the same frames predicted
with the parameters FrameCoder::Optimize selects, both through the bit-plane + range coder as restated
in oracle/sac_coder.c (byte-identical to the reference's: tests/test_oracle_vs_ref.py; analysis only) -- and the CostGolomb
estimate the optimizer minimised beside it.
usage: coded_bytes.py [nframes] [profile]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle_lib
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
FRAME = 176400
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prof = int(sys.argv[2]) if len(sys.argv) > 2 else 2
orc = oracle_lib.Oracle()
L, R = stereo_clip(1000, nfr * FRAME)
frames = [(L[f * FRAME:(f + 1) * FRAME], R[f * FRAME:(f + 1) * FRAME]) for f in range(nfr)]
rows = []
with sacopt.Context(0) as ctx:
    best, traces, _ = ctx.optimize(prof, frames)
    base = ctx.predict(prof, frames, maps=True)
    opt = ctx.predict(prof, frames, coefs=best, maps=True)
    for f, (l, r) in enumerate(frames):
        size = lambda o: sum(orc.encode_channel(o[f"u{c}"], o["maxbpn"][c], o[f"m{c}"], o["maxbpn"][2 + c], orc.remap_tables(x), want_bytes=False)[0]
                             for c, x in enumerate((l, r)))
        b, o = size(base[f]), size(opt[f])
        tr = traces[f]
        rows.append(dict(frame=f, base_bytes=b, opt_bytes=o, saving_pct=100. * (b - o) / b, golomb_base=float(tr[0, 2]), golomb_best=float(tr[:, 2].min()),
                         best=[float(x) for x in best[f]]))
        print(json.dumps(rows[-1]), flush=True)
tb, to = sum(r["base_bytes"] for r in rows), sum(r["opt_bytes"] for r in rows)
print(json.dumps(dict(profile=prof, frames=nfr, base_bytes=tb, opt_bytes=to, saving_pct=100. * (tb - to) / tb, pcm_bytes=nfr * FRAME * 4)))
