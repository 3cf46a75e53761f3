"""Randomised differential run of the library against the C restatement (oracle/), wider than tests/ and tools/soak.py: all four
profiles, random coefficient vectors inside the profile's ranges, ten signal families (tonal clip, silence, DC, full-scale
square, sparse alphabets that switch Remap on, white noise, very quiet noise, impulses, 17-24 bit values, bursts), lengths on and around every block
edge of the kernels (8, 32, 64, 256, 512, 2048, 4096, 7056, 16384) and random ones. Per round and profile:
  predict   one call of all clips AND single-clip calls (both (1280,640) kernels)   == oracle: e, u, m, maxbpn
  unpredict plain residuals, and mapped ones with the tables of Remap::Analyse      == oracle and == the input samples
  optimize  a few short clips, random number of optimised coefficients                == oracle: winners and cost trace
  coded sizes of the plain and mapped s2u buffers of every clip                       == oracle coder's byte counts
  payload bytes of both variants (sacopt_encode_payloads) of every third clip         == oracle coder's bytes
  get_costs on random windows with carried state, cost_golomb / _noalloc / cost_meanrms on raw buffers, the decoder on
  residuals no encoder wrote (clamp, Unmap tables), encode in both modes and Optimize with the other two cost functions
usage: fuzz_parity.py [seed] [rounds] [clips per round] [full-size frames]      (test infrastructure: it uses the oracle)"""
import ctypes as C
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib
from sac_amd import sacopt
from sac_amd.synth import stereo_clip

EDGES = [1, 2, 7, 8, 9, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1025, 2047, 2048, 2049, 4095, 4096, 4097,
         7055, 7056, 7057, 8192, 16383, 16384, 16385]


def signal(rs, kind, n):
    s = int(rs.randint(1, 1 << 30))
    if kind == 0:
        return stereo_clip(s, n)
    if kind == 1:
        return np.zeros(n, np.int32), np.zeros(n, np.int32)
    if kind == 2:
        return np.full(n, int(rs.randint(-32768, 32768)), np.int32), np.full(n, int(rs.randint(-32768, 32768)), np.int32)
    if kind == 3:   # full-scale square waves of different periods
        p0, p1 = int(rs.randint(2, 200)), int(rs.randint(2, 200))
        t = np.arange(n)
        return np.where((t // p0) & 1, 32767, -32768).astype(np.int32), np.where((t // p1) & 1, -32768, 32767).astype(np.int32)
    if kind == 4:   # sparse alphabet: Remap pays
        L, R = stereo_clip(s, n)
        q0, q1 = int(rs.randint(2, 40)), int(rs.randint(2, 40))
        return ((L // q0) * q0).astype(np.int32), ((R // q1) * q1).astype(np.int32)
    if kind == 5:
        r = np.random.RandomState(s)
        return r.randint(-32768, 32768, n).astype(np.int32), r.randint(-32768, 32768, n).astype(np.int32)
    if kind == 6:
        r = np.random.RandomState(s)
        return r.randint(-3, 4, n).astype(np.int32), r.randint(-1, 2, n).astype(np.int32)
    if kind == 8:   # beyond 16 bits: values outside Remap's +-32768 table ("val too large", map.cpp:116), predictions outside it
        L, R = stereo_clip(s, n)
        k0, k1 = int(rs.randint(2, 200)), int(rs.randint(2, 200))
        return (L * k0).astype(np.int32), (R * k1).astype(np.int32)
    if kind == 9:   # silence, then a burst, then silence again (adaptive stages wake up and decay)
        L, R = stereo_clip(s, n)
        a, b = sorted(int(x) for x in rs.randint(0, n + 1, 2))
        L = L.copy(); R = R.copy()
        L[:a] = 0; R[:a] = 0; L[b:] = 0; R[b:] = 0
        return L, R
    r = np.random.RandomState(s)   # impulses on silence
    L = np.zeros(n, np.int32); R = np.zeros(n, np.int32)
    k = max(1, n // 97)
    L[r.randint(0, n, k)] = r.randint(-32768, 32768, k); R[r.randint(0, n, k)] = r.randint(-32768, 32768, k)
    return L, R


def s2u_inv(u):   # utils.h:62-66
    return np.where(u & 1, ((u.astype(np.int64) + 1) >> 1), -(u.astype(np.int64) >> 1)).astype(np.int32)


KEYS = ("e0", "e1", "u0", "u1", "m0", "m1", "maxbpn")
def fuzz(ctx, orc, seed, rounds, nclips, nbig=0, log=print):
    """Returns the number of mismatches (each one logged). ctx and orc must use the same rsqrtss table."""
    rs = np.random.RandomState(seed)
    bad = [0]

    def check(ok, *what):
        if not ok:
            bad[0] += 1
            log("MISMATCH", *what)

    def coder_size(buf, maxbpn):
        b = np.ascontiguousarray(buf, dtype=np.int32)
        return orc.lib.oracle_encode_mono_normal(b.ctypes.data_as(C.c_void_p), C.c_int(len(b)), C.c_int(int(maxbpn)), None, C.c_int(0))

    for rnd in range(rounds):
        for prof in (2, 1, 0, 3):
            t = time.time()
            base = orc.base_profile(prof)
            lens = [int(rs.choice(EDGES)) if rs.rand() < 0.5 else int(rs.choice([rs.randint(1, 600), rs.randint(600, 5000), rs.randint(5000, 20000), rs.randint(20000, 45000)], p=[.3, .45, .2, .05]))
                    for _ in range(nclips)]
            kinds = [int(rs.randint(0, 10)) for _ in range(nclips)]
            clips = [signal(rs, k, n) for k, n in zip(kinds, lens)]
            # one coefficient vector per clip: the profile's defaults or a random point of its ranges
            coefs = np.array([[(v[2] if rs.rand() < 0.3 else v[0] + rs.rand() * (v[1] - v[0])) for v in base] for _ in range(nclips)])
            allres = ctx.predict(prof, clips, coefs=coefs, maps=True)
            refs = []
            for i, (L, R) in enumerate(clips):
                ref = orc.predict(prof, L, R, coefs=coefs[i], maps=True)
                refs.append(ref)
                for k in KEYS:
                    check(np.array_equal(allres[i][k], ref[k]), "predict batch", prof, i, kinds[i], len(L), k)
                if i % 4 == 0:
                    one = ctx.predict(prof, [(L, R)], coefs=coefs[i:i + 1], maps=True)[0]
                    for k in KEYS:
                        check(np.array_equal(one[k], ref[k]), "predict single", prof, i, kinds[i], len(L), k)
            # decoder: plain residuals of every clip in one call; mapped ones for the sparse clips
            dec = ctx.unpredict(prof, [dict(e0=r["e0"], e1=r["e1"]) for r in refs], coefs=coefs)
            for i, (L, R) in enumerate(clips):
                # (the high profiles' encoder does not clamp but their decoder does: identical while the prediction stays in range)
                o = orc.unpredict(prof, refs[i]["e0"], refs[i]["e1"], coefs=coefs[i])
                check(np.array_equal(dec[i][0], o[0]) and np.array_equal(dec[i][1], o[1]), "unpredict vs oracle", prof, i, kinds[i], len(L))
                check(np.array_equal(dec[i][0], L) and np.array_equal(dec[i][1], R), "unpredict round trip", prof, i, kinds[i], len(L))   # (e = v - p wraps, and so does v = e + p)
            if prof in (1, 2):
                sp = [i for i in range(nclips) if kinds[i] in (3, 4, 7)][:6]
                if sp:
                    fr = [dict(e0=s2u_inv(refs[i]["m0"]), e1=s2u_inv(refs[i]["m1"]), used0=orc.remap_tables(clips[i][0]), used1=orc.remap_tables(clips[i][1])) for i in sp]
                    dm = ctx.unpredict(prof, fr, coefs=coefs[sp])
                    for j, i in enumerate(sp):   # (a diverged filter's residuals are not invertible -- Map(pred, INT32_MIN) is 0 --: the oracle's decoder is the yardstick)
                        o = orc.unpredict(prof, fr[j]["e0"], fr[j]["e1"], coefs=coefs[i], used0=fr[j]["used0"], used1=fr[j]["used1"])
                        check(np.array_equal(dm[j][0], o[0]) and np.array_equal(dm[j][1], o[1]), "mapped unpredict vs oracle", prof, i, kinds[i], lens[i])
                        if np.abs(refs[i]["e0"].astype(np.int64)).max() < 2 ** 30 and np.abs(refs[i]["e1"].astype(np.int64)).max() < 2 ** 30:
                            check(np.array_equal(dm[j][0], clips[i][0]) and np.array_equal(dm[j][1], clips[i][1]), "mapped round trip", prof, i, kinds[i], lens[i])
            # coded sizes of both residual forms
            bufs, mb = [], []
            for r in refs:
                bufs += [r["u0"], r["u1"], r["m0"], r["m1"]]
                mb += [int(r["maxbpn"][0]), int(r["maxbpn"][1]), int(r["maxbpn"][2]), int(r["maxbpn"][3])]
            got = ctx.coded_sizes(bufs, mb)
            for j in range(len(bufs)):
                check(int(got[j]) == coder_size(bufs[j], mb[j]), "coded size", prof, j // 4, j % 4, len(bufs[j]), mb[j])
            # the payload BYTES of both variants (EncodeMonoFrame_Normal / _Mapped) for a third of the clips
            pj = [i for i in range(nclips) if i % 3 == 0 and lens[i] <= 20000]
            pb, pm_, pu = [], [], []
            for i in pj:
                r = refs[i]
                for chn, src in enumerate(clips[i]):
                    pb += [r[f"u{chn}"], r[f"m{chn}"]]; pm_ += [int(r["maxbpn"][chn]), int(r["maxbpn"][2 + chn])]; pu += [None, orc.remap_tables(src)]
            if pb:
                pay = ctx.encode_payloads(pb, pm_, pu)
                for j, (b_, m_, u_) in enumerate(zip(pb, pm_, pu)):
                    b_ = np.ascontiguousarray(b_, dtype=np.int32)
                    cap_ = 16 * len(b_) + 70000
                    o_ = np.zeros(cap_, np.uint8)
                    if u_ is None:
                        n_ = orc.lib.oracle_encode_mono_normal(b_.ctypes.data_as(C.c_void_p), C.c_int(len(b_)), C.c_int(m_), o_.ctypes.data_as(C.c_void_p), C.c_int(cap_))
                    else:
                        n_ = orc.lib.oracle_encode_mono_mapped(b_.ctypes.data_as(C.c_void_p), C.c_int(len(b_)), C.c_int(m_), u_.ctypes.data_as(C.c_void_p),
                                                               o_.ctypes.data_as(C.c_void_p), C.c_int(cap_))
                    check(pay[j] == o_[:n_].tobytes(), "payload bytes", prof, pj[j // 4], j % 4, len(b_), m_, len(pay[j]), n_)
            # optimizer: a few short clips in one call
            pick = [i for i in range(nclips) if 64 <= lens[i] <= 6000][:4]
            if pick:
                ncl = int(rs.randint(1, sacopt.NCOEF[prof] + 1))
                ptab = None
                if rs.rand() < 0.5:   # a profile table of the caller's: ranges shrunk around the defaults (the grid's length is decided by FP accumulation, codec.cpp:392)
                    ptab = base.copy()
                    ptab[:, 0] += rs.rand(len(base)) * (ptab[:, 2] - ptab[:, 0])
                    ptab[:, 1] -= rs.rand(len(base)) * (ptab[:, 1] - ptab[:, 2])
                best, traces, _ = ctx.optimize(prof, [clips[i] for i in pick], prof=ptab, ncoef_loop=ncl)
                for j, i in enumerate(pick):
                    ob, ot = orc.optimize(prof, *clips[i], prof=ptab, ncoef_loop=ncl)
                    check(best[j].tobytes() == ob[:best.shape[1]].tobytes() and traces[j].tobytes() == ot.tobytes(), "optimize", prof, i, kinds[i], lens[i], ncl, ptab is not None)
            # ---- the other entry points, on a few clips of this round
            small = [i for i in range(nclips) if 200 <= lens[i] <= 9000][:5]
            if small:
                fr = [clips[i] for i in small]
                # GetCost on windows, several candidates per frame, carried running mean (codec.cpp:342-352, 315-339)
                ncand = int(rs.randint(1, 5))
                cands = np.array([[[(v[2] if rs.rand() < 0.5 else v[0] + rs.rand() * (v[1] - v[0])) for v in base] for _ in range(ncand)] for _ in small])
                frm = [int(rs.randint(0, len(c[0]) // 2)) for c in fr]
                nn = [int(rs.randint(1, len(c[0]) - f + 1)) for c, f in zip(fr, frm)]
                st0 = rs.rand(len(small)) * 8
                costs, st = ctx.get_costs(prof, fr, cands, frm=frm, n=nn, state=st0)
                for j, (L, R) in enumerate(fr):
                    mean = float(st0[j])
                    for k in range(ncand):
                        r = orc.predict(prof, L, R, frm=frm[j], n=nn[j], coefs=cands[j, k])
                        c1, mean = orc.cost_golomb(r["e0"], mean)
                        c2, mean = orc.cost_golomb(r["e1"], mean)
                        okc = np.array([(c1 + c2) / 2.]).tobytes() == costs[j, k:k + 1].tobytes()
                        check(okc, "get_costs", prof, small[j], kinds[small[j]], frm[j], nn[j], k, "got", costs[j, k], "want", (c1 + c2) / 2., c1, c2)
                        if not okc:
                            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                            np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_getcosts_{seed}_{rnd}_{prof}_{small[j]}_{k}.npz"), L=L, R=R, cand=cands[j, k], frm=frm[j], n=nn[j],
                                     st0=st0[j], e0=r["e0"], e1=r["e1"])
                    check(np.array([mean]).tobytes() == st[j:j + 1].tobytes(), "get_costs state", prof, small[j])
                # the cost functions on raw buffers (residuals of a diverged filter included)
                for j in small[:3]:
                    for key in ("e0", "e1"):
                        b = refs[j][key]; s0_ = float(rs.rand() * 5)
                        g = ctx.cost_golomb(b, s0_); w = orc.cost_golomb(b, s0_)
                        check(np.array(g).tobytes() == np.array(w).tobytes(), "cost_golomb", prof, j, kinds[j], len(b))
                        g = ctx.cost_golomb_noalloc(b, s0_)
                        check(np.array(g).tobytes() == np.array(w).tobytes(), "cost_golomb_noalloc", prof, j, kinds[j], len(b))
                        check(np.float64(ctx.cost_meanrms(b)).tobytes() == np.float64(orc.cost_meanrms(b)).tobytes(), "cost_meanrms", prof, j, kinds[j], len(b))
                # the decoder on residuals that no encoder wrote, with the high profile's clamp and Unmap tables
                if prof in (1, 2):
                    gfr = []
                    for j in small[:3]:
                        n_ = min(len(clips[j][0]), 1500)
                        amp = int(rs.choice([3, 300, 40000, 2 ** 31 - 1]))
                        g = dict(e0=rs.randint(-amp, amp + 1, n_).astype(np.int32), e1=rs.randint(-amp // 2 - 1, amp // 2 + 2, n_).astype(np.int32))
                        if amp > 40000: g["e0"][rs.randint(0, n_, 3)] = -2 ** 31
                        if rs.rand() < 0.5: g["used0"] = orc.remap_tables(clips[j][0])
                        if rs.rand() < 0.5: g["used1"] = orc.remap_tables(clips[j][1])
                        if prof == 2 and rs.rand() < 0.7: g["minmax"] = (int(rs.randint(-40000, 0)), int(rs.randint(0, 40000)), int(rs.randint(-40000, 0)), int(rs.randint(0, 40000)))
                        gfr.append(g)
                    dg = ctx.unpredict(prof, gfr, coefs=coefs[small[:3]])
                    for j, g, v in zip(small[:3], gfr, dg):
                        o = orc.unpredict(prof, g["e0"], g["e1"], coefs=coefs[j], used0=g.get("used0"), used1=g.get("used1"), minmax=g.get("minmax"))
                        check(np.array_equal(v[0], o[0]) and np.array_equal(v[1], o[1]), "unpredict garbage", prof, j, len(g["e0"]), sorted(g.keys()))
            # FrameCoder::Predict with Optimize enabled, both modes, and the other two cost functions (short clips: the oracle is slow)
            tiny = [i for i in range(nclips) if 64 <= lens[i] <= 2500 and kinds[i] in (0, 4, 5, 6)][:2]
            if tiny:
                ncl = int(rs.randint(1, sacopt.NCOEF[prof] + 1))
                fr = [clips[i] for i in tiny]
                for sel in (False, True):
                    best, outs = ctx.encode(prof, fr, ncoef_loop=ncl, maps=True, selected=sel)
                    for j, i in enumerate(tiny):
                        ob, _ = orc.optimize(prof, *clips[i], ncoef_loop=ncl)
                        check(best[j].tobytes() == ob[:best.shape[1]].tobytes(), "encode winners", prof, i, sel, ncl)
                        r = orc.predict(prof, *clips[i], coefs=(ob[:best.shape[1]] if sel else None), maps=True)
                        for k in KEYS:
                            check(np.array_equal(outs[j][k], r[k]), "encode residuals", prof, i, sel, k)
                cf = int(rs.randint(1, 3))
                ctx.set_cost_function(cf)
                best, traces, _ = ctx.optimize(prof, fr, ncoef_loop=ncl)
                ctx.set_cost_function(0)
                for j, i in enumerate(tiny):
                    ob, ot = orc.optimize(prof, *clips[i], ncoef_loop=ncl, cost_fn=cf)
                    check(best[j].tobytes() == ob[:best.shape[1]].tobytes() and traces[j].tobytes() == ot.tobytes(), "optimize cost_fn", prof, i, cf, ncl)
            # a whole-batch Optimize (hundreds of chains per round: the shared stream sets, the lead stage's launch sequences)
            if prof == 2:
                many = [i for i in range(nclips) if 64 <= lens[i] <= 4000]
                if len(many) > 16:
                    best, traces, _ = ctx.optimize(prof, [clips[i] for i in many])
                    for j, i in enumerate(many):
                        ob, ot = orc.optimize(prof, *clips[i])
                        check(best[j].tobytes() == ob[:best.shape[1]].tobytes() and traces[j].tobytes() == ot.tobytes(), "optimize batch", prof, i, kinds[i], lens[i], len(many))
            log(f"round {rnd} profile {prof}: {nclips} clips, {sum(lens)} samples, {time.time() - t:.1f} s, mismatches so far {bad[0]}")
    # ---- full-size frames (argv[4] of them): the chunk-pipelined full-frame pass with carried stage state, chunked optimizer windows
    if nbig:
        t = time.time()
        prof = 2
        base = orc.base_profile(prof)
        lens = [int(rs.randint(100000, 200001)) for _ in range(nbig)]
        kinds = [int(rs.choice([0, 3, 4, 5, 7, 8, 9])) for _ in range(nbig)]
        clips = [signal(rs, k, n) for k, n in zip(kinds, lens)]
        coefs = np.array([[(v[2] if rs.rand() < 0.5 else v[0] + rs.rand() * (v[1] - v[0])) for v in base] for _ in range(nbig)])
        allres = ctx.predict(prof, clips, coefs=coefs, maps=True)
        refs = [orc.predict(prof, L, R, coefs=coefs[i], maps=True) for i, (L, R) in enumerate(clips)]
        for i in range(nbig):
            for k in KEYS:
                check(np.array_equal(allres[i][k], refs[i][k]), "big predict", i, kinds[i], lens[i], k)
        dec = ctx.unpredict(prof, [dict(e0=r["e0"], e1=r["e1"]) for r in refs], coefs=coefs)
        for i, (L, R) in enumerate(clips):
            check(np.array_equal(dec[i][0], L) and np.array_equal(dec[i][1], R), "big unpredict round trip", i, kinds[i], lens[i])
        best, outs = ctx.encode(prof, clips[:2], maps=True, selected=True)
        for i in range(min(2, nbig)):
            ob, _ = orc.optimize(prof, *clips[i])
            check(best[i].tobytes() == ob[:best.shape[1]].tobytes(), "big encode winners", i, kinds[i], lens[i])
            r = orc.predict(prof, *clips[i], coefs=ob[:best.shape[1]], maps=True)
            for k in KEYS:
                check(np.array_equal(outs[i][k], r[k]), "big encode residuals", i, kinds[i], lens[i], k)
        log(f"big: {nbig} frames, {sum(lens)} samples, kinds {kinds}, {time.time() - t:.1f} s, mismatches so far {bad[0]}")
    return bad[0]


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    nclips = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    nbig = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    oracle_lib.build_oracle()
    orc = oracle_lib.Oracle()
    _t = np.load(os.path.join(ROOT, "tests", "golden", "rsqrt_table.npz"), allow_pickle=False)
    tab, bits = _t["tab"].astype(np.float32), int(_t["bits"])   # the fixture's rsqrtss table on both sides
    orc.set_rsqrt_table(tab, bits)
    with sacopt.Context(0, tab, bits) as ctx:
        nbad = fuzz(ctx, orc, seed, rounds, nclips, nbig, log=lambda *a: print(*a, flush=True))
    print("FUZZ", "FAILED" if nbad else "ok")
    sys.exit(1 if nbad else 0)
