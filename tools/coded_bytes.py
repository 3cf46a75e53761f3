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
import time
with sacopt.Context(0) as ctx:
    t0 = time.perf_counter(); best, traces, _ = ctx.optimize(prof, frames); t_g = time.perf_counter() - t0
    ctx.set_cost_function(sacopt.COST_BITPLANE)
    t0 = time.perf_counter(); best_b, traces_b, _ = ctx.optimize(prof, frames); t_b = time.perf_counter() - t0
    ctx.set_cost_function(sacopt.COST_GOLOMB)
    base = ctx.predict(prof, frames, maps=True)
    opt = ctx.predict(prof, frames, coefs=best, maps=True)
    opt_b = ctx.predict(prof, frames, coefs=best_b, maps=True)
    print(json.dumps(dict(optimize_s_cost_golomb=t_g, optimize_s_cost_bitplane=t_b)), flush=True)
    for f, (l, r) in enumerate(frames):
        size = lambda o: sum(orc.encode_channel(o[f"u{c}"], o["maxbpn"][c], o[f"m{c}"], o["maxbpn"][2 + c], orc.remap_tables(x), want_bytes=False)[0]
                             for c, x in enumerate((l, r)))
        b, o, ob = size(base[f]), size(opt[f]), size(opt_b[f])
        tr = traces[f]
        rows.append(dict(frame=f, base_bytes=b, opt_bytes=o, saving_pct=100. * (b - o) / b, opt_bytes_bitplane_objective=ob,
                         saving_pct_bitplane_objective=100. * (b - ob) / b, window_bits_base=float(traces_b[f][0, 2]), window_bits_best=float(traces_b[f][:, 2].min()), golomb_base=float(tr[0, 2]), golomb_best=float(tr[:, 2].min()),
                         best=[float(x) for x in best[f]]))
        print(json.dumps(rows[-1]), flush=True)
tb, to, tob = sum(r["base_bytes"] for r in rows), sum(r["opt_bytes"] for r in rows), sum(r["opt_bytes_bitplane_objective"] for r in rows)
print(json.dumps(dict(profile=prof, frames=nfr, base_bytes=tb, opt_bytes_cost_golomb=to, saving_pct_cost_golomb=100. * (tb - to) / tb,
                      opt_bytes_cost_bitplane=tob, saving_pct_cost_bitplane=100. * (tb - tob) / tb, pcm_bytes=nfr * FRAME * 4)))
