"""CPU, build container only: the oracle against the REAL reference compiled into
oracle/_ref/libsacref.so, both using the native rsqrtss of this CPU. Skipped where the
reference build is absent."""
import numpy as np
import pytest

from oracle_lib import Oracle, Ref, have_ref
from sac_amd.synth import stereo_clip, degenerate_clips

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def pair():
    o = Oracle()
    o.set_rsqrt_table(None, 0)  # native instruction, like the reference
    return o, Ref()


def test_rsqrt_probe_matches_native(pair):
    o, r = pair
    tab, bits = o.probe_rsqrt()
    assert bits >= 0
    o.set_rsqrt_table(tab, bits)
    rs = np.random.RandomState(1)
    xs = (np.abs(rs.randn(4000)) * 10.0 ** rs.randint(-3, 30, 4000) + 1e-3).astype(np.float32)
    for x in xs:
        assert np.float32(o.rsqrt(float(x))).tobytes() == np.float32(r.rsqrt(float(x))).tobytes()
    assert o.rsqrt(float("inf")) == r.rsqrt(float("inf")) == 0.0
    o.set_rsqrt_table(None, 0)


@pytest.mark.parametrize("prof,seed,n", [(0, 21, 5000), (1, 22, 4000), (2, 23, 3000)])
def test_predict(pair, prof, seed, n):
    o, r = pair
    L, R = stereo_clip(seed, n + 500)
    a = o.predict(prof, L, R, frm=250, n=n, maps=True)
    b = r.predict(prof, L, R, frm=250, n=n, maps=True)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_optimize(pair):
    o, r = pair
    L, R = stereo_clip(24, 4096)
    bo, to = o.optimize(2, L, R)
    br, tr = r.optimize_trace(2, L, R)
    b2, n = r.optimize(2, L, R)
    assert n == 8
    assert to.tobytes() == tr.tobytes() and bo.tobytes() == br.tobytes() == b2.tobytes()


def test_loud_clip_high_amplitude(pair):
    # full-scale square-ish input: exercises large |err| (Remap::Map loops, big Golomb means)
    o, r = pair
    rs = np.random.RandomState(5)
    L = (np.sign(np.sin(np.arange(2000) * 0.05)) * 30000 + rs.randint(-2000, 2000, 2000)).astype(np.int32)
    R = (-L // 2 + rs.randint(-500, 500, 2000)).astype(np.int32)
    for prof in (0, 1, 2):
        a = o.predict(prof, L, R, maps=True); b = r.predict(prof, L, R, maps=True)
        for k in a:
            assert np.array_equal(a[k], b[k]), (prof, k)


def test_silence(pair):
    o, r = pair
    Z = np.zeros(1000, np.int32)
    for prof in (0, 1, 2):
        a = o.predict(prof, Z, Z, maps=True); b = r.predict(prof, Z, Z, maps=True)
        for k in a:
            assert np.array_equal(a[k], b[k]), (prof, k)


def test_samples_beyond_remap_table(pair):
    # samples outside Remap's +-32768 table ("val too large", map.cpp:116,123: not marked) and
    # predictions outside it (isUsed false, map.cpp:134-135)
    o, r = pair
    rs = np.random.RandomState(5)
    L = (np.sign(np.sin(np.arange(900) * 0.05)) * 90000 + rs.randint(-6000, 6000, 900)).astype(np.int32)
    R = (-L // 2 + rs.randint(-1500, 1500, 900)).astype(np.int32)
    for prof in (0, 1, 2):
        a = o.predict(prof, L, R, maps=True); b = r.predict(prof, L, R, maps=True)
        for k in a:
            assert np.array_equal(a[k], b[k]), (prof, k)


def test_frame_record_bytes_from_oracle_residuals(pair):
    # the reference's Encode()/WriteEncoded() fed with the oracle's residuals writes the reference's own frame bytes
    import oracle_lib
    o, r = pair
    L, R = stereo_clip(25, 6000)
    for prof in (0, 1, 2):
        res = o.predict(prof, L, R, maps=True)
        assert oracle_lib.ref_encode_residuals(r, prof, L, R, res) == oracle_lib.ref_encode_frame(r, prof, L, R), prof


def test_degenerate_inputs(pair):
    # constant / periodic / sparse / 24-bit inputs: rank-deficient covariance, Cholesky failures keep the
    # previous solution (lpc.h:216-218), large Golomb means
    o, r = pair
    for name, (L, R) in degenerate_clips(2500).items():
        for prof in (0, 1, 2):
            a = o.predict(prof, L, R, maps=True); b = r.predict(prof, L, R, maps=True)
            for k in a:
                assert np.array_equal(a[k], b[k]), (name, prof, k)


def _sparse_clip(seed, n):
    # samples on a coarse value grid: most values unused, so Remap::Map changes the residuals
    L, R = stereo_clip(seed, n)
    return (L // 8 * 8).astype(np.int32), (R // 8 * 8 + 3).astype(np.int32)


@pytest.mark.parametrize("prof,seed,n", [(0, 31, 4000), (1, 32, 3000), (2, 33, 2500)])
def test_unpredict_plain(pair, prof, seed, n):
    # FrameCoder::UnpredictStereoFrame (codec.cpp:216-235): oracle == reference, and decode(encode(x)) == x
    o, r = pair
    L, R = stereo_clip(seed, n)
    enc = o.predict(prof, L, R)
    a = o.unpredict(prof, enc["e0"], enc["e1"])
    b = r.unpredict(prof, enc["e0"], enc["e1"])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[0], L) and np.array_equal(a[1], R)


@pytest.mark.parametrize("prof,seed,n", [(1, 34, 2500), (2, 35, 2000)])
def test_unpredict_mapped(pair, prof, seed, n):
    # enc_mapped channels: residuals go through Remap::Unmap (map.cpp:154-168); one channel mapped, then both
    o, r = pair
    L, R = _sparse_clip(seed, n)
    enc = o.predict(prof, L, R, maps=True)
    u2s = lambda u: np.where(u & 1, (u + 1) // 2, -(u // 2)).astype(np.int32)  # inverse of S2U (utils.h:56-61)
    m0, m1 = u2s(enc["m0"]), u2s(enc["m1"])
    assert not np.array_equal(m0, enc["e0"])  # the map does something on this input
    t0, t1 = o.remap_tables(L), o.remap_tables(R)
    for kw, ee in ((dict(used0=t0), (m0, enc["e1"])), (dict(used0=t0, used1=t1), (m0, m1))):
        a = o.unpredict(prof, *ee, **kw)
        b = r.unpredict(prof, *ee, **kw)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert np.array_equal(a[0], L) and np.array_equal(a[1], R)


def test_unpredict_high_clamp(pair):
    # StereoHigh::Unpredict clamps the prediction to framestats[ch].minval/maxval (profile.cpp:163,170); the encoder
    # does not, so a narrow range decodes to something else than the input -- the oracle must follow the reference
    o, r = pair
    L, R = stereo_clip(36, 2000)
    enc = o.predict(2, L, R)
    mm = [-300, 200, -100, 400]
    a = o.unpredict(2, enc["e0"], enc["e1"], minmax=mm)
    b = r.unpredict(2, enc["e0"], enc["e1"], minmax=mm)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert not np.array_equal(a[0], L)
    wide = [int(L.min()), int(L.max()), int(R.min()), int(R.max())]
    a = o.unpredict(2, enc["e0"], enc["e1"], minmax=wide)
    b = r.unpredict(2, enc["e0"], enc["e1"], minmax=wide)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_unpredict_coefs_and_garbage_residuals(pair):
    # non-default coefficients; residuals that no encoder produced (decoder state diverges, still deterministic)
    o, r = pair
    rs = np.random.RandomState(9)
    e0 = rs.randint(-200, 200, 1500).astype(np.int32); e1 = rs.randint(-50, 50, 1500).astype(np.int32)
    for prof, cf in ((0, [0.993]), (1, [0.997, 3e-4, 2e-4, 2e-3, 0.02, 0.999]), (2, [0.99, 2e-5, 1e-5, 3e-4, 2e-4, 2e-3, 0.02, 0.999])):
        a = o.unpredict(prof, e0, e1, coefs=cf)
        b = r.unpredict(prof, e0, e1, coefs=cf)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), prof


def _spd(seed, n):
    rs = np.random.RandomState(seed)
    M = rs.randn(n, 3 * n) * 50.0
    return M @ M.T, rs.randn(n) * 1e4


def test_chol_solve(pair):
    # Cholesky::Factor + Solve (cholesky.cpp:16-77): all-orders form with regp (LPC3), full-order form (LPC2), the
    # Cholesky::Test matrix, and an indefinite matrix ("matrix indefinit": Factor returns 1)
    o, r = pair
    for n, regp, fast in ((24, 500., False), (32, 500., False), (24, 0., True), (4, 0., True)):
        A, b = _spd(n, n)
        if n == 4:
            A = np.array([[9, 3, -6, 12], [3, 26, -7, -11], [-6, -7, 9, 7], [12, -11, 7, 65]], float); b = np.array([72., 34, 22, 326])
        so, fo = o.chol_solve(A, b, regp, fast)
        sr, fr = r.chol_solve(A, b, regp, fast)
        assert not fo and not fr and so.tobytes() == sr.tobytes(), n
        if n == 4:
            assert np.abs(so - [2, 4, 3, 5]).sum() < 1e-14
    A, b = _spd(3, 24)
    A[5, 5] = -1e9
    assert o.chol_solve(A, b, 500., False)[1] and r.chol_solve(A, b, 500., False)[1]


def test_bitplane_coder_bytes(pair):
    # SURVEY 8f.4, first half: the bit-plane context-mixing coder + MapEncoder + range coder restated in oracle/sac_coder.c
    # produce the reference's payload BYTES (FrameCoder::Encode, codec.cpp:237-280), plain and mapped, all profiles'
    # residual statistics, a sparse-valued clip (the mapped variant wins) and silence
    import oracle_lib
    o, r = pair
    cases = [(0, stereo_clip(26, 5000)), (1, stereo_clip(27, 7000)), (2, stereo_clip(28, 6000)), (1, _sparse_clip(29, 6000)),
             (0, (np.zeros(1500, np.int32), np.zeros(1500, np.int32)))]
    mapped_seen = plain_seen = False
    for prof, (L, R) in cases:
        res = o.predict(prof, L, R, maps=True)
        want = oracle_lib.ref_encode_channels(r, prof, L, R, res)
        for ch, (s, u, m) in enumerate(((L, res["u0"], res["m0"]), (R, res["u1"], res["m1"]))):
            got = o.encode_channel(u, res["maxbpn"][ch], m, res["maxbpn"][2 + ch], o.remap_tables(s))
            assert got == want[ch], (prof, ch, len(got[0]), len(want[ch][0]), got[1:], want[ch][1:])
            mapped_seen |= got[1]; plain_seen |= not got[1]
    assert mapped_seen and plain_seen


def test_bitplane_coder_values_beyond_the_planes(pair):
    # the restated coder against the reference's Encode() where the buffer has bits above maxbpn (the mapped variant is fed
    # the same buffer, so the plain one -- strictly smaller: no table in front -- is what the reference keeps)
    import ctypes as C
    import oracle_lib
    o, r = pair
    bufs = oracle_lib.out_of_plane_buffers()
    for k in range(0, len(bufs), 2):
        (u0, mb), (u1, _) = bufs[k], bufs[k + 1]
        z = np.zeros(len(u0), np.int32)
        want = oracle_lib.ref_encode_channels(r, 1, z, z, dict(u0=u0, u1=u1, m0=u0, m1=u1, maxbpn=np.array([mb] * 4, np.int32)))
        for ch, u in enumerate((u0, u1)):
            cap = 16 * len(u) + 70000
            out = np.zeros(cap, np.uint8)
            n = o.lib.oracle_encode_mono_normal(u.ctypes.data_as(C.c_void_p), C.c_int(len(u)), C.c_int(mb), out.ctypes.data_as(C.c_void_p), C.c_int(cap))
            assert not want[ch][1] and out[:n].tobytes() == want[ch][0], (k, ch, n, len(want[ch][0]))


def test_remap_map_beyond_int32(pair):
    # Remap::Map where the reference's int32 arithmetic overflows (undefined; its build wraps): the restatement, compiled
    # with -fwrapv, returns what the reference's compiled code returns. The diverging clip: whole PredictStereoFrame + Map.
    import oracle_lib
    o, r = pair
    src, pred, err = oracle_lib.map_wrap_cases()
    assert np.array_equal(o.remap_map(src, pred, err), r.remap_map(src, pred, err))
    L, R, coefs = oracle_lib.diverging_impulse_clip()
    a = o.predict(2, L, R, coefs=coefs)
    b = r.predict(2, L, R, coefs=coefs)
    assert a["e0"].min() == -2 ** 31 and (a["u0"] < 0).any()   # the case is what it claims to be
    for k in ("e0", "e1", "u0", "u1"):
        assert np.array_equal(a[k], b[k]), k
    p0 = L - a["e0"]
    assert np.array_equal(o.remap_map(L, p0, a["e0"]), r.remap_map(L, p0, a["e0"]))
    # the cost functions on such residuals: s2u wraps negative, abs(INT32_MIN) stays negative
    for k in ("e0", "e1"):
        assert o.cost_golomb(a[k], 1.5) == r.cost_golomb(a[k], 1.5), k
        assert o.cost_meanrms(a[k]) == r.cost_meanrms(a[k]), k
    z = np.array([5, -2 ** 31, -7, 2 ** 30, -2 ** 30 - 3], np.int32)
    assert o.cost_meanrms(z) == r.cost_meanrms(z) == 3.0 and o.cost_golomb(z, 0.) == r.cost_golomb(z, 0.)


def test_optimize_with_cost_mean_rms(pair):
    # FrameCoder::Optimize's schedule driven with the reference's OTHER CostFunction, CostMeanRMS (codec.cpp:302-312)
    o, r = pair
    L, R = stereo_clip(45, 4096)
    bo, to = o.optimize(2, L, R, cost_fn=1)
    br, tr = r.optimize_trace(2, L, R, meanrms=True)
    assert to.tobytes() == tr.tobytes() and bo.tobytes() == br.tobytes()
