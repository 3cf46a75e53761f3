"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libsacref.so).

Run in the build container only (needs /root/reference to build oracle/_ref):
    make -C oracle ref && python tests/golden/make_golden.py && python tests/golden/make_golden.py unpredict
The fixtures are data only: seeded synthetic inputs (re-generated from sac_amd.synth at test
time, and also stored) and the reference's outputs. They pin the behaviour of the x86
`rsqrtss` of the CPU they were generated on (Intel Xeon here) -- the probed table travels with
them and every test that compares against a fixture runs with THAT table.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle_lib import Oracle, Ref  # noqa: E402
from sac_amd.synth import stereo_clip  # noqa: E402


def main():
    ref = Ref()
    orc = Oracle()
    tab, bits = orc.probe_rsqrt()
    assert bits >= 0
    # sanity: the table reproduces the native instruction the reference just used
    rs = np.random.RandomState(0)
    xs = np.abs(rs.randn(20000).astype(np.float32)) * np.float32(10.0) ** rs.randint(-3, 12, 20000).astype(np.float32) + np.float32(1e-3)
    orc.set_rsqrt_table(tab, bits)
    for x in xs[:5000]:
        assert np.float32(orc.rsqrt(float(x))).tobytes() == np.float32(ref.rsqrt(float(x))).tobytes()
    np.savez_compressed(os.path.join(HERE, "rsqrt_table.npz"), tab=tab, bits=np.int32(bits))

    out = {}
    # --- PredictStereoFrame, 3 profiles, seed 11, 4096 samples out of a 6000-sample frame
    L, R = stereo_clip(11, 6000)
    out["pred_L"], out["pred_R"] = L, R
    for prof in (0, 1, 2):
        d = ref.predict(prof, L, R, frm=0, n=4096, maps=True)
        for k, v in d.items():
            out[f"pred{prof}_{k}"] = v
        d = ref.predict(prof, L, R, frm=1500, n=2000, maps=True)  # window not at frame start
        for k, v in d.items():
            out[f"predw{prof}_{k}"] = v
    # non-default coefficients (high)
    cf = np.array([0.9899, 2e-5, 3e-6, 3e-4, 2e-4, 1e-3, 0.02, 0.9995])
    out["predc_coefs"] = cf
    d = ref.predict(2, L, R, frm=0, n=3000, coefs=cf)
    out["predc_e0"], out["predc_e1"] = d["e0"], d["e1"]
    # --- per-stage traces, first 96 samples
    for prof in (1, 2):
        p, v, e0, e1 = ref.cascade_trace(prof, L[:96], R[:96])
        out[f"trace{prof}_p"], out[f"trace{prof}_v"] = p, v
    # --- stage-level drivers
    x0 = (rs.randn(3000) * 300.0)
    x1 = 0.5 * x0 + rs.randn(3000) * 200.0
    out["lms_x0"], out["lms_x1"] = x0, x1
    for (S, D, m1, m2) in [(1280, 640, 1e-5, 5e-6), (256, 128, 5e-4, 1e-4), (32, 16, 3e-3, 3e-3), (8, 4, 0.01, 0.01)]:
        p0, p1 = ref.lms_pair(S, D, m1, m2, x0, x1)
        out[f"lms{S}_p0"], out[f"lms{S}_p1"] = p0, p1
    p0, p1 = ref.lpc3_pair(0.997, L[:3000], R[:3000])
    out["lpc3_p0"], out["lpc3_p1"] = p0, p1
    pv = rs.randn(2000, 4) * 100.0
    val = rs.randn(2000) * 100.0
    out["rls_pv"], out["rls_val"] = pv, val
    out["rls4_p"] = ref.rls_run(4, 0.998, pv, val)
    out["rls3_p"] = ref.rls_run(3, 0.998, np.ascontiguousarray(pv[:, :3]), val)
    # --- CostGolomb with carried state
    e = rs.randint(-3000, 3000, 5000).astype(np.int32)
    out["cost_e"] = e
    c1, s1 = ref.cost_golomb(e[:2500], 0.0)
    c2, s2 = ref.cost_golomb(e[2500:], s1)
    out["cost_vals"] = np.array([c1, s1, c2, s2, ref.cost_meanrms(e)])
    # --- Optimize on one 8192-sample frame (window 1024): trace of 84 GetCost calls + winners
    Lo, Ro = stereo_clip(12, 8192)
    out["opt_L"], out["opt_R"] = Lo, Ro
    best, trace = ref.optimize_trace(2, Lo, Ro)
    best2, n = ref.optimize(2, Lo, Ro)
    assert n == 8 and np.array_equal(best, best2), "trace replay deviates from FrameCoder::Optimize"
    out["opt_best"], out["opt_trace"] = best, trace
    # second frame, larger (window 2048)
    Lo2, Ro2 = stereo_clip(13, 16384)
    out["opt2_L"], out["opt2_R"] = Lo2, Ro2
    best, trace = ref.optimize_trace(2, Lo2, Ro2)
    out["opt2_best"], out["opt2_trace"] = best, trace
    # normal profile with the documented deviation (loop over its 6 coefs)
    best, trace = ref.optimize_trace(1, Lo, Ro, ncoef_loop=6)
    out["optn_best"], out["optn_trace"] = best[:6], trace
    # base profiles
    for prof in (0, 1, 2):
        out[f"base{prof}"] = ref.base_profile(prof)
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    print("wrote", len(out), "arrays")


def main_unpredict():
    """tests/golden/golden_unpredict.npz: FrameCoder::UnpredictStereoFrame (codec.cpp:216-235) of the reference."""
    ref = Ref()
    out = {}
    u2s = lambda u: np.where(u & 1, (u + 1) // 2, -(u // 2)).astype(np.int32)
    L, R = stereo_clip(41, 3000)
    Ls, Rs = (L // 8 * 8).astype(np.int32), (R // 8 * 8 + 3).astype(np.int32)  # sparse value grid: the map acts
    out["L"], out["R"], out["Ls"], out["Rs"] = L, R, Ls, Rs
    orc = Oracle()
    out["used0"], out["used1"] = orc.remap_tables(Ls), orc.remap_tables(Rs)
    rs = np.random.RandomState(42)
    g0 = rs.randint(-200, 200, 1500).astype(np.int32); g1 = rs.randint(-50, 50, 1500).astype(np.int32)
    out["g0"], out["g1"] = g0, g1
    out["mm"] = np.array([-300, 200, -100, 400], np.int32)
    out["coefs2"] = np.array([0.99, 2e-5, 1e-5, 3e-4, 2e-4, 2e-3, 0.02, 0.999])
    for prof in (0, 1, 2):
        d = ref.predict(prof, L, R)
        out[f"e{prof}_0"], out[f"e{prof}_1"] = d["e0"], d["e1"]
        v = ref.unpredict(prof, d["e0"], d["e1"])
        assert np.array_equal(v[0], L) and np.array_equal(v[1], R)
        out[f"v{prof}_0"], out[f"v{prof}_1"] = v
        v = ref.unpredict(prof, g0, g1)  # residuals no encoder wrote
        out[f"vg{prof}_0"], out[f"vg{prof}_1"] = v
        if prof:
            d = ref.predict(prof, Ls, Rs, maps=True)
            out[f"m{prof}_0"], out[f"m{prof}_1"] = u2s(d["m0"]), u2s(d["m1"])
            v = ref.unpredict(prof, out[f"m{prof}_0"], out[f"m{prof}_1"], used0=out["used0"], used1=out["used1"])
            assert np.array_equal(v[0], Ls) and np.array_equal(v[1], Rs)
            out[f"vm{prof}_0"], out[f"vm{prof}_1"] = v
    d = ref.predict(2, L, R)
    v = ref.unpredict(2, d["e0"], d["e1"], minmax=out["mm"])  # narrow clamp (profile.cpp:163,170)
    out["vc2_0"], out["vc2_1"] = v
    v = ref.unpredict(2, g0, g1, coefs=out["coefs2"])
    out["vk2_0"], out["vk2_1"] = v
    np.savez_compressed(os.path.join(HERE, "golden_unpredict.npz"), **out)
    print("wrote", len(out), "arrays")


def main_config5():
    """tests/golden/golden_config5.npz: BASELINE configs[4]'s shape -- 24-bit 96 kHz, framesize 384000 (optimizer window
    48000), stereo pairs of an 8-channel frame treated as frames.
      * r_*  : the REAL reference (24-tap OLS as shipped): FrameCoder::Optimize trace + winners of pair 0, SHA-256 of its
               full-frame residuals (the reference handles any int32 sample in PredictStereoFrame; only its CLI / WAV
               reader / Remap table are 16-bit);
      * lpc32: LPC3(0.997, 24, 8, 8, 500) stage outputs from the REAL reference class (lpc.h:181 takes S and D);
      * o32_*: profile 3 (StereoHigh with that 32-tap stage) from the ORACLE -- no reference profile has this shape, so
               these rows are "parity unpinned" (GPU vs restatement only).
    About 5 minutes of CPU."""
    import hashlib
    from concurrent.futures import ProcessPoolExecutor
    out = {}
    n = 384000
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a, dtype=np.int32).tobytes()).digest(), np.uint8)
    out["seeds"] = np.array([320, 321, 322, 323], np.int32)
    with ProcessPoolExecutor(3) as ex:
        f_ref = ex.submit(_c5_ref, n)
        f_o32 = ex.submit(_c5_o32, n)
        f_o32s = ex.submit(_c5_o32_small)
        out.update(f_ref.result()); out.update(f_o32.result()); out.update(f_o32s.result())
    ref = Ref()
    L, R = stereo_clip(11, 6000)
    out["lpc32_p0"], out["lpc32_p1"] = ref.lpc3_pair(0.997, L[:3000], R[:3000], S=24, D=8)
    np.savez_compressed(os.path.join(HERE, "golden_config5.npz"), **out)
    print("wrote", len(out), "arrays")


def _sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a, dtype=np.int32).tobytes()).digest(), np.uint8)


def _c5_ref(n):
    ref = Ref()
    L, R = stereo_clip(320, n, rate=96000, bits=24)
    best, trace = ref.optimize_trace(2, L, R)
    d = ref.predict(2, L, R, maps=True)
    return {"r_best": best, "r_trace": trace, "r_e0_sha": _sha(d["e0"]), "r_e1_sha": _sha(d["e1"]), "r_maxbpn": d["maxbpn"][:2].copy()}


def _c5_o32(n):
    orc = Oracle()   # native rsqrtss of this (Intel) host == the fixture table (checked by main())
    L, R = stereo_clip(321, n, rate=96000, bits=24)
    best, trace = orc.optimize(3, L, R)
    return {"o32_best": best, "o32_trace": trace}


def _c5_o32_small():
    orc = Oracle()
    L, R = stereo_clip(11, 6000)
    d = orc.predict(3, L, R, frm=0, n=4096)
    best, trace = orc.optimize(3, *stereo_clip(12, 8192))
    return {"o32_e0": d["e0"], "o32_e1": d["e1"], "o32s_best": best, "o32s_trace": trace}


if __name__ == "__main__":
    {"unpredict": main_unpredict, "config5": main_config5}.get((sys.argv[1:] or [""])[0], main)()
