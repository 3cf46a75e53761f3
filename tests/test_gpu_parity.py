"""GPU: libsacopt (HIP, through the C ABI) against the oracle and the golden fixtures.
Bit-exact everywhere: int32 residuals, FP64 stage outputs / costs / selected parameters
compared as raw bits. All runs use the FIXTURE's rsqrtss table (Intel), so they do not depend
on the CPU of the GPU box."""
import numpy as np
import pytest

from sac_amd import sacopt
from sac_amd.synth import stereo_clip, degenerate_clips

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.fixture(scope="module")
def ctx(rsqrt_table):
    c = sacopt.Context(0, *rsqrt_table)
    yield c
    c.close()


# ---------------------------------------------------------------- stage kernels vs golden
@pytest.mark.parametrize("S,D,m1,m2", [(8, 4, 0.01, 0.01), (32, 16, 3e-3, 3e-3), (256, 128, 5e-4, 1e-4), (1280, 640, 1e-5, 5e-6)])
def test_lms_pair_golden(ctx, golden, S, D, m1, m2):
    p0, p1 = ctx.lms_pair(S, D, m1, m2, golden["lms_x0"], golden["lms_x1"])
    assert same_bits(p0, golden[f"lms{S}_p0"]) and same_bits(p1, golden[f"lms{S}_p1"])


@pytest.mark.parametrize("S,D,m1,m2", [(256, 128, 5e-4, 1e-4), (1280, 640, 1e-5, 5e-6)])
def test_lms_pair_hostile_inputs_vs_oracle(ctx, oracle, S, D, m1, m2):
    # stage inputs that defeat the guesses of the speculative sums: magnitudes spread over 24 binades, runs of zeros, spikes,
    # cancellation (x1 = -x0 in stretches). k_lms1c then predicts wrongly more often (update redone from the kept weights),
    # walks off its candidates, re-walks after failed verifications; the (256,128) stage misses its candidates. Bit-equal anyway.
    rs = np.random.RandomState(77)
    n = 3000
    x0 = rs.randn(n) * 10.0 ** rs.uniform(-4, 3.2, n)
    x1 = rs.randn(n) * 10.0 ** rs.uniform(-4, 3.2, n)
    x0[500:700] = 0.0; x1[600:650] = 0.0
    x1[1000:1400] = -x0[1000:1400]
    x0[1500] = 3e5; x1[1501] = -3e5
    x0[2000:2300] = np.round(x0[2000:2300])
    p0, p1 = ctx.lms_pair(S, D, m1, m2, x0, x1)
    o0, o1 = oracle.lms_pair(S, D, m1, m2, x0, x1)
    assert np.isfinite(o0).all() and np.isfinite(o1).all()
    assert same_bits(p0, o0) and same_bits(p1, o1)


@pytest.mark.parametrize("prof", [2, 1])
def test_random_lengths_vs_oracle(ctx, oracle, prof):
    # 28 clips of odd lengths (1 .. 30000 samples) in ONE call (more than 20 chains: the two-channel (1280,640) kernel) and every
    # third one alone (k_lms1c): block edges (256), chunk edges (512, window chunks, the pipelined full-frame path beyond 16384)
    rs = np.random.RandomState(5 + prof)
    lens = [1, 2, 255, 256, 257, 511, 513, 4095, 4097, 16385] + [int(rs.randint(3, 3000)) for _ in range(10)] + \
           [int(rs.randint(3000, 17000)) for _ in range(6)] + [int(rs.randint(17000, 30000)) for _ in range(2)]
    clips = [stereo_clip(int(rs.randint(1, 1 << 30)), n) for n in lens]
    res = ctx.predict(prof, clips)
    for i, (L, R) in enumerate(clips):
        ref = oracle.predict(prof, L, R)
        assert same_bits(res[i]["e0"], ref["e0"]) and same_bits(res[i]["e1"], ref["e1"]), (i, len(L))
        if i % 3 == 0:
            one = ctx.predict(prof, [(L, R)])[0]
            assert same_bits(one["e0"], ref["e0"]) and same_bits(one["e1"], ref["e1"]), (i, len(L))


def test_lpc3_pair_golden(ctx, golden):
    p0, p1 = ctx.lpc3_pair(0.997, golden["pred_L"][:3000], golden["pred_R"][:3000])
    assert same_bits(p0, golden["lpc3_p0"]) and same_bits(p1, golden["lpc3_p1"])


def test_rls_golden(ctx, golden):
    # sacopt_rls_run drives the cascades' own mixer kernel k_mix<NS> (lane per P[i][j], DPP row sums), 700 samples per
    # launch with P / w carried from launch to launch as the pipelines carry them
    assert same_bits(ctx.rls_run(4, 0.998, golden["rls_pv"], golden["rls_val"]), golden["rls4_p"])
    assert same_bits(ctx.rls_run(3, 0.998, np.ascontiguousarray(golden["rls_pv"][:, :3]), golden["rls_val"]), golden["rls3_p"])


@pytest.mark.parametrize("prof", [1, 2])
def test_stage_traces_golden(ctx, golden, prof):
    # every stage's prediction p[k] and input val[ch][k] of the first 96 samples, as the stage kernels hand them on in
    # HBM, against the reference's own trace (profile.cpp:126-148)
    p, v = ctx.cascade_trace(prof, golden["pred_L"][:96], golden["pred_R"][:96])
    assert same_bits(p, golden[f"trace{prof}_p"]) and same_bits(v, golden[f"trace{prof}_v"])


def test_stage_traces_vs_oracle_long(ctx, oracle):
    L, R = stereo_clip(36, 5000)
    cf = np.array([0.9899, 2e-5, 3e-6, 3e-4, 2e-4, 1e-3, 0.02, 0.9995])
    p, v = ctx.cascade_trace(2, L, R, coefs=cf)
    po, vo, _, _ = oracle.cascade_trace(2, L, R, coefs=cf)
    assert same_bits(p, po) and same_bits(v, vo)


def _spd(seed, n):
    rs = np.random.RandomState(seed)
    M = rs.randn(n, 3 * n) * 50.0
    return M @ M.T, rs.randn(n) * 1e4


def test_cholesky_kernel(ctx, oracle):
    # k_chol directly: Cholesky::Test's known answer (cholesky.cpp:3-14; the 4 x 4 system embedded in a 24 x 24 block
    # diagonal, which leaves every scalar's operation sequence unchanged), random SPD systems of both built orders in both
    # forms bit-equal to the oracle (itself pinned against the reference class: test_oracle_vs_ref.py::test_chol_solve),
    # and "matrix indefinit" (lpc.h:216-218)
    A = np.eye(24); b = np.zeros(24)
    A[:4, :4] = [[9, 3, -6, 12], [3, 26, -7, -11], [-6, -7, 9, 7], [12, -11, 7, 65]]
    b[:4] = [72, 34, 22, 326]
    x, failed = ctx.chol_solve(A, b, fast=True)
    assert not failed and np.abs(x[:4] - [2, 4, 3, 5]).sum() < 1e-14 and not x[4:].any()
    assert same_bits(x, oracle.chol_solve(A, b, 0., True)[0])
    for n, fast in ((24, False), (32, False), (24, True)):
        for seed in (1, 2):
            A, b = _spd(seed + n, n)
            got, failed = ctx.chol_solve(A, b, fast=fast)
            want, wf = oracle.chol_solve(A, b, 0. if fast else 500., fast)
            assert not failed and not wf and same_bits(got, want), (n, fast, seed)
    A, b = _spd(3, 24)
    A[5, 5] = -1e9
    assert ctx.chol_solve(A, b)[1] and oracle.chol_solve(A, b, 500., False)[1]


def _seq_sum(x):
    out = np.zeros(x.shape[0])
    with np.errstate(all="ignore"):
        for v in range(x.shape[0]):
            acc = np.float64(0.)
            for t in x[v]:
                acc = acc + t      # IEEE double add, round to nearest even: what the reference's loop does (nlms.h:110-111)
            out[v] = acc
    return out


@pytest.mark.parametrize("nterms", [1920, 384, 48, 12])
def test_mfma_serial_adder_adversarial(ctx, nterms):
    # v_mfma_f64_4x4x4 with B == 1 used as an exact serial adder, and the speculative summation built on it: bit-equal to a
    # sequential host loop on inputs chosen to break an adder that rounds differently, flushes denormals, reorders, or
    # mishandles signed zeros / infinities / NaN
    rs = np.random.RandomState(nterms)
    rows = []
    rows.append(rs.uniform(-1.5, 1.5, nterms))                                        # benign
    rows.append(rs.randn(nterms) * 10.0 ** rs.randint(-300, 300, nterms))             # 600 orders of magnitude
    r = rs.randn(nterms) * 2.0 ** rs.randint(-20, 20, nterms); r[1::2] = -r[::2] * (1 + 2.0 ** -50)   # cancellation over 40 binades
    rows.append(r)
    rows.append(rs.randn(nterms) * 5e-324 * rs.randint(1, 1 << 20, nterms))           # denormal terms and partial sums
    r = rs.randn(nterms) * 1e-310; r[::7] = 2.2250738585072014e-308                   # around the normal / denormal border
    rows.append(r)
    r = np.zeros(nterms); r[::2] = -0.0                                               # signed zeros only: (+0) + (-0) = +0
    rows.append(r)
    r = np.full(nterms, -0.0); rows.append(r.copy())                                  # all -0: 0. + (-0) = +0 from the start
    r = rs.randn(nterms); r[nterms // 2] = 1e308; r[nterms // 2 + 1] = 1e308          # overflow to +inf, stays inf
    rows.append(r)
    r = rs.randn(nterms); r[3] = np.inf; r[nterms - 2] = -np.inf                      # inf - inf = NaN
    rows.append(r)
    r = rs.randn(nterms); r[nterms // 3] = np.nan
    rows.append(r)
    r = np.ones(nterms); r[1:] = 2.0 ** -53                                           # ties: 1 + 2^-53 rounds to even every time
    rows.append(r)
    r = np.full(nterms, 2.0 ** -53); r[0] = 1 + 2.0 ** -52                            # ties the other way (odd mantissa rounds up)
    rows.append(r)
    r = rs.randn(nterms) * 1e15; r[::3] += 0.5                                        # half-ulp patterns at 2^50
    rows.append(r)
    for _ in range(6):                                                                # audio-like: products w*h of mixed scale
        rows.append(rs.randn(nterms) * np.abs(rs.randn(nterms)) * 10.0 ** rs.uniform(-6, 4))
    x = np.stack(rows)
    got = ctx.serial_sum(x)
    want = _seq_sum(x)
    for v in range(len(rows)):
        if np.isnan(want[v]):
            assert np.isnan(got[v]), v      # the payload of a generated NaN is not pinned (x86 gives the "real indefinite" -nan)
        else:
            assert got[v:v + 1].tobytes() == want[v:v + 1].tobytes(), (v, got[v], want[v])


def test_cost_golden(ctx, golden):
    e = golden["cost_e"]
    c1, s1 = ctx.cost_golomb(e[:2500], 0.0)
    c2, s2 = ctx.cost_golomb(e[2500:], s1)
    assert same_bits(np.array([c1, s1, c2, s2, ctx.cost_meanrms(e)]), golden["cost_vals"])
    assert ctx.cost_golomb(e[:0], 1.5) == (0.0, 1.5)


# ---------------------------------------------------------------- PredictStereoFrame
@pytest.mark.parametrize("prof", [0, 1, 2])
def test_predict_golden(ctx, golden, prof):
    L, R = golden["pred_L"], golden["pred_R"]
    out = ctx.predict(prof, [(L, R), (L, R)], frm=[0, 1500], n=[4096, 2000], maps=True)
    for key in ("e0", "e1", "u0", "u1", "m0", "m1"):  # m* = s2u_error_map (Remap::Map on the device)
        assert np.array_equal(out[0][key], golden[f"pred{prof}_{key}"]), key
        assert np.array_equal(out[1][key], golden[f"predw{prof}_{key}"]), key
    assert np.array_equal(out[0]["maxbpn"], golden[f"pred{prof}_maxbpn"])
    assert np.array_equal(out[1]["maxbpn"], golden[f"predw{prof}_maxbpn"])
    # without map outputs requested the map statistics are reported as -1
    out = ctx.predict(prof, [(L, R)], frm=[0], n=[512])
    assert list(out[0]["maxbpn"][2:]) == [-1, -1]


def test_predict_custom_coefs_golden(ctx, golden):
    out = ctx.predict(2, [(golden["pred_L"], golden["pred_R"])], frm=[0], n=[3000], coefs=golden["predc_coefs"][None, :])
    assert np.array_equal(out[0]["e0"], golden["predc_e0"]) and np.array_equal(out[0]["e1"], golden["predc_e1"])


@pytest.mark.parametrize("prof,seed,n", [(0, 31, 5000), (1, 32, 5000), (2, 33, 6000)])
def test_predict_vs_oracle(ctx, oracle, prof, seed, n):
    L, R = stereo_clip(seed, n)
    ref = oracle.predict(prof, L, R)
    out = ctx.predict(prof, [(L, R)])[0]
    for k in ("e0", "e1", "u0", "u1"):
        assert np.array_equal(out[k], ref[k]), k


@pytest.mark.parametrize("prof", [1, 2])
def test_predict_pipelined_long_frames(ctx, oracle, prof):
    # frames longer than two pipeline chunks (2 x 22528) take the chunk-pipelined path with
    # carried stage state; ragged lengths make the last chunks differ per frame
    frames = [stereo_clip(50, 50000), stereo_clip(51, 46001)]
    outs = ctx.predict(prof, frames, maps=True)
    for (L, R), o in zip(frames, outs):
        ref = oracle.predict(prof, L, R, maps=True)
        for k in ("e0", "e1", "u0", "u1", "m0", "m1"):
            assert np.array_equal(o[k], ref[k]), k
        assert np.array_equal(o["maxbpn"], ref["maxbpn"])


def test_predict_edge_cases(ctx, oracle):
    # empty frame, 1-sample frame, ragged batch, silence, full-scale
    L, R = stereo_clip(34, 700)
    Z = np.zeros(300, np.int32)
    rs = np.random.RandomState(5)
    Lf = (np.sign(np.sin(np.arange(900) * 0.05)) * 30000 + rs.randint(-2000, 2000, 900)).astype(np.int32)
    Rf = (-Lf // 2 + rs.randint(-500, 500, 900)).astype(np.int32)
    # 17-bit-ish samples: values beyond Remap's +-32768 table ("val too large", map.cpp:116) and
    # predictions outside it (isUsed -> false, map.cpp:134-135)
    Lw = (Lf * 3).astype(np.int32); Rw = (Rf * 3).astype(np.int32)
    frames = [(L[:0], R[:0]), (L[:1], R[:1]), (L, R), (Z, Z), (Lf, Rf), (Lw, Rw)]
    for prof in (0, 1, 2):
        outs = ctx.predict(prof, frames, maps=True)
        for (a, b), o in zip(frames, outs):
            if not len(a):
                assert len(o["e0"]) == 0 and len(o["m0"]) == 0
                continue
            ref = oracle.predict(prof, a, b, maps=True)
            for k in ("e0", "e1", "u0", "u1", "m0", "m1"):
                assert np.array_equal(o[k], ref[k]), (prof, k)
            assert np.array_equal(o["maxbpn"], ref["maxbpn"]), prof


# ---------------------------------------------------------------- GetCost / Optimize
def test_get_costs_vs_oracle(ctx, oracle):
    L, R = stereo_clip(35, 4000)
    base = sacopt.base_profile(2)[:, 2]
    cands = np.tile(base, (3, 1))
    cands[1, 1] = 3e-5
    cands[2, 7] = 0.9995
    costs, st = ctx.get_costs(2, [(L, R)], cands[None], frm=[500], n=[3000], state=[2.5])
    mean = 2.5
    for k in range(3):
        r = oracle.predict(2, L, R, frm=500, n=3000, coefs=cands[k])
        c1, mean = oracle.cost_golomb(r["e0"], mean)
        c2, mean = oracle.cost_golomb(r["e1"], mean)
        assert same_bits(np.array([(c1 + c2) / 2.]), costs[0, k:k + 1]), k
    assert same_bits(np.array([mean]), st)


def test_optimize_golden(ctx, golden):
    # FrameCoder::Optimize: identical selected parameters AND identical 84-row cost trace (bit-exact)
    frames = [(golden["opt_L"], golden["opt_R"]), (golden["opt2_L"], golden["opt2_R"])]
    best, traces, rows = ctx.optimize(2, frames)
    assert list(rows) == [84, 84]
    assert same_bits(best[0], golden["opt_best"]) and same_bits(traces[0], golden["opt_trace"])
    assert same_bits(best[1], golden["opt2_best"]) and same_bits(traces[1], golden["opt2_trace"])


def test_optimize_normal_profile_golden(ctx, golden):
    best, traces, rows = ctx.optimize(1, [(golden["opt_L"], golden["opt_R"])], ncoef_loop=6)
    assert same_bits(best[0], golden["optn_best"]) and same_bits(traces[0], golden["optn_trace"])


def test_optimize_rejects_reference_ub(ctx, golden):
    # the reference indexes coefs[0..7] for every profile (codec.cpp:389): UB for fast/normal -> error here
    with pytest.raises(sacopt.SacoptError):
        ctx.optimize(1, [(golden["opt_L"], golden["opt_R"])], ncoef_loop=8)


def test_optimize_vs_oracle_ragged(ctx, oracle):
    # a batch of frames of different lengths incl. one shorter than 8 samples (window n = 0)
    frames = [stereo_clip(40 + i, n) for i, n in enumerate([4096, 3000, 5])]
    best, traces, rows = ctx.optimize(2, frames)
    for f, (L, R) in enumerate(frames):
        ob, ot = oracle.optimize(2, L, R)
        assert same_bits(best[f], ob) and same_bits(traces[f], ot), f


def test_encode_frames_overlapped(ctx, oracle):
    # FrameCoder::Predict with Optimize enabled: optimizer rounds and the pipelined full-frame pass
    # run concurrently on the GPU; both must still match the oracle bit for bit
    frames = [stereo_clip(60, 48000), stereo_clip(61, 46000)]
    best, outs = ctx.encode(2, frames)
    for f, (L, R) in enumerate(frames):
        ob, _ = oracle.optimize(2, L, R)
        assert same_bits(best[f], ob), f
        ref = oracle.predict(2, L, R)
        assert np.array_equal(outs[f]["e0"], ref["e0"]) and np.array_equal(outs[f]["e1"], ref["e1"]), f


# ---------------------------------------------------------------- stream level (.sac frame records)
def test_frame_records_byte_identical():
    # "ref+opt": the stream is coded with the base profile, so a host that takes FrameCoder::Predict from the
    # GPU and keeps the reference's Encode()/WriteEncoded() must write the reference's bytes. The bit-plane /
    # range coder here IS the reference's (oracle/_ref, built in the container, test-only); where it is absent
    # the test is skipped. Uses the CPU's own rsqrtss (probed) on both sides, as a real encoder run would.
    import oracle_lib
    if not oracle_lib.have_ref():
        pytest.skip("oracle/_ref not built")
    ref = oracle_lib.Ref()
    c = sacopt.Context(0)  # table probed from this host's CPU = what the reference binary uses here
    try:
        frames = [stereo_clip(70, 20000), stereo_clip(71, 17001)]
        for prof in (0, 1, 2):
            best, outs = c.encode(prof, frames, maps=True)
            for (L, R), o in zip(frames, outs):
                assert oracle_lib.ref_encode_residuals(ref, prof, L, R, o) == oracle_lib.ref_encode_frame(ref, prof, L, R), prof
    finally:
        c.close()


def test_predict_many_frames_kernel_variants(ctx, oracle):
    # > 20 chains per launch use the two-channel LMS(1280,640) kernel, fewer use the per-channel one;
    # with ragged lengths the later pipeline chunks drop below 20 jobs, so a chain's carried state
    # passes from one kernel to the other. Everything must equal the one-frame-at-a-time results
    # (per-channel kernel only) and, for sampled frames, the oracle.
    lens = [16500 + 37 * i for i in range(98)] + [24600 + 11 * i for i in range(8)]
    frames = [stereo_clip(200 + i % 7, n) for i, n in enumerate(lens)]
    outs = ctx.predict(2, frames)
    for i in (0, 57, 97, 98, 105):
        L, R = frames[i]
        one = ctx.predict(2, [(L, R)])[0]
        for k in ("e0", "e1", "u0", "u1"):
            assert np.array_equal(outs[i][k], one[k]), (i, k)
    for i in (3, 101):
        L, R = frames[i]
        ref = oracle.predict(2, L, R)
        assert np.array_equal(outs[i]["e0"], ref["e0"]) and np.array_equal(outs[i]["e1"], ref["e1"]), i
    # identical inputs give identical outputs wherever they sit in the batch
    for i in range(7, 98, 7):
        assert np.array_equal(outs[i]["e0"][:16500], outs[0]["e0"][:16500])


def test_degenerate_inputs_vs_oracle(ctx, oracle):
    # constant / periodic / sparse / 24-bit inputs (see sac_amd.synth.degenerate_clips)
    clips = degenerate_clips(2500)
    for prof in (0, 1, 2):
        outs = ctx.predict(prof, list(clips.values()), maps=True)
        for (name, (L, R)), o in zip(clips.items(), outs):
            ref = oracle.predict(prof, L, R, maps=True)
            for k in ("e0", "e1", "u0", "u1", "m0", "m1", "maxbpn"):
                assert np.array_equal(o[k], ref[k]), (name, prof, k)


def test_predict_diverged_filter_beyond_int32(ctx, oracle):
    # predictions out of int32 (cvttsd2si -> INT32_MIN), residuals that wrap, s2u that wraps negative, Remap::Map walks that
    # run over the end of int32: the arithmetic of the reference's build (tests/test_oracle_vs_ref.py pins the oracle there)
    import oracle_lib
    L, R, coefs = oracle_lib.diverging_impulse_clip()
    ref = oracle.predict(2, L, R, coefs=coefs, maps=True)
    assert ref["e0"].min() == -2 ** 31 and (ref["u0"] < 0).any()
    got = ctx.predict(2, [(L, R)] * 2, coefs=np.stack([coefs, coefs]), maps=True)
    for g in got:
        for k in ("e0", "e1", "u0", "u1", "m0", "m1", "maxbpn"):
            assert np.array_equal(g[k], ref[k]), k
    # the cost functions on wrapped residuals, alone and through GetCost on a window
    for k in ("e0", "e1"):
        assert same_bits(np.array(ctx.cost_golomb(ref[k], 1.5)), np.array(oracle.cost_golomb(ref[k], 1.5))), k
        assert same_bits(np.array(ctx.cost_golomb_noalloc(ref[k], 1.5)), np.array(oracle.cost_golomb(ref[k], 1.5))), k
        assert same_bits(np.array([ctx.cost_meanrms(ref[k])]), np.array([oracle.cost_meanrms(ref[k])])), k
    costs, st = ctx.get_costs(2, [(L, R)], coefs[None, None], frm=[0], n=[len(L)], state=[0.5])
    c1, mean = oracle.cost_golomb(ref["e0"], 0.5)
    c2, mean = oracle.cost_golomb(ref["e1"], mean)
    assert same_bits(costs[0], np.array([(c1 + c2) / 2.])) and same_bits(st, np.array([mean]))
    # the coded size of such a buffer: planes maxbpn..0 of the two's-complement bits
    import ctypes as C
    sz = ctx.coded_sizes([ref["u0"], ref["u1"]], [int(ref["maxbpn"][0]), int(ref["maxbpn"][1])])
    for b, m, s in zip((ref["u0"], ref["u1"]), ref["maxbpn"][:2], sz):
        b = np.ascontiguousarray(b)
        assert s == oracle.lib.oracle_encode_mono_normal(b.ctypes.data_as(C.c_void_p), C.c_int(len(b)), C.c_int(int(m)), None, C.c_int(0))


def test_optimize_small_budget_vs_oracle(ctx, oracle):
    # BASELINE configs[1]: "small candidate budget" = coefficient 0 only (baseline + 10 GetCost calls)
    frames = [stereo_clip(80, 4096), stereo_clip(81, 3333)]
    best, traces, rows = ctx.optimize(2, frames, ncoef_loop=1)
    assert list(rows) == [11, 11]
    for f, (L, R) in enumerate(frames):
        ob, ot = oracle.optimize(2, L, R, ncoef_loop=1)
        assert same_bits(best[f], ob) and same_bits(traces[f], ot), f


# ---------------------------------------------------------------- C++ host side (sac_amd/host/framecoder.h)
def _host_example_fill(n, seed):
    # same integer generator as sac_amd/host/host_example.cpp::fill
    s = (seed * 2654435761 + 12345) & 0xffffffff
    l = np.zeros(n, np.int32); r = np.zeros(n, np.int32)
    a = 0
    for i in range(n):
        s = (s * 1664525 + 1013904223) & 0xffffffff
        noise = ((s >> 16) & 0x3ff) - 512
        a += 37 if (i // 64) & 1 else -37
        t = a * 3
        b = (abs(t) // 4) * (1 if t >= 0 else -1) + (abs(noise) // 4) * (1 if noise >= 0 else -1)   # C division truncates
        l[i] = a * 8 + noise
        r[i] = b * 8 - (abs(noise) // 2) * (1 if noise >= 0 else -1)
    return l, r


def _fnv(a):
    h = 1469598103934665603
    for v in np.ascontiguousarray(a, dtype=np.int32).view(np.uint32).tolist():
        h = ((h ^ v) * 1099511628211) & 0xffffffffffffffff
    return h


def test_cpp_host_framecoder():
    # the C++ mirror of FrameCoder (PredictBatch -> sacopt_encode_frames) run as its own program must give
    # what the ctypes path gives: residual / s2u / map checksums, maxbpn, selected parameters
    import os, subprocess
    exe = os.path.join(os.path.dirname(sacopt.LIB_PATH), "host", "host_example")
    if not os.path.exists(exe):
        pytest.skip("host_example not built")
    n, nframes = 3000, 2
    out = subprocess.run([exe, "2", str(nframes), str(n), "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [ln.split() for ln in out.stdout.strip().splitlines()]
    assert len(lines) == nframes
    c = sacopt.Context(0)
    try:
        frames = [_host_example_fill(n, 100 + b) for b in range(nframes)]
        best, outs = c.encode(2, frames, maps=True)
    finally:
        c.close()
    for b, ln in enumerate(lines):
        L, R = frames[b]
        o = outs[b]
        assert [int(x, 16) for x in ln[3:5]] == [_fnv(L), _fnv(R)], "generator mismatch"
        assert [int(x, 16) for x in ln[6:8]] == [_fnv(o["e0"]), _fnv(o["e1"])]
        assert [int(x, 16) for x in ln[9:11]] == [_fnv(o["u0"]), _fnv(o["u1"])]
        assert [int(x, 16) for x in ln[12:14]] == [_fnv(o["m0"]), _fnv(o["m1"])]
        assert [int(x) for x in ln[15:19]] == list(o["maxbpn"])
        assert same_bits(np.array([float(x) for x in ln[20:28]]), best[b])
        assert ln[28:30] == ["decode", "ok"]  # FrameCoder::Unpredict of the C++ mirror (one channel mapped) restores the samples
        assert ln[30:32] == ["apply", "ok"]   # frames coded with the SELECTED parameters, carried by sacfile.h's extension records, decode back
        # FrameCoder::EncodeBatchOn + sacfile.h's WriteEncoded: the frame record, against the ctypes path + the python container writer
        import oracle_lib
        from sac_amd import sacfile
        orc = oracle_lib.Oracle()
        with sacopt.Context(0) as c2:
            pay = c2.encode_payloads([o["u0"], o["m0"], o["u1"], o["m1"]], [o["maxbpn"][0], o["maxbpn"][2], o["maxbpn"][1], o["maxbpn"][3]],
                                     [None, orc.remap_tables(L), None, orc.remap_tables(R)])
        ch = [(pay[2 * k], False, int(o["maxbpn"][k])) if len(pay[2 * k]) < len(pay[2 * k + 1]) else (pay[2 * k + 1], True, int(o["maxbpn"][2 + k])) for k in range(2)]
        rec = np.frombuffer(sacfile.write_frame(n, ch[0], ch[1]), np.uint8)
        hsh = 1469598103934665603
        for v in rec.tolist():
            hsh = ((hsh ^ v) * 1099511628211) & 0xffffffffffffffff
        assert ln[32] == "record" and int(ln[33]) == len(rec) and int(ln[34], 16) == hsh
        assert [int(x) for x in ln[36:38]] == [int(ch[0][1]), int(ch[1][1])]


def test_cpp_host_sharder_two_contexts_two_threads():
    # sac_amd/host/sharder.h (the Codec::EncodeFile loop, codec.cpp:543-555, over several contexts): two and three contexts
    # on GPU 0, one host thread each, contiguous frame ranges, running CONCURRENTLY -- every line (input / residual / map
    # checksums, maxbpn, selected parameters, decoder and parameter-stream round trips) must equal the one-context run's
    import os, subprocess
    exe = os.path.join(os.path.dirname(sacopt.LIB_PATH), "host", "host_example")
    if not os.path.exists(exe):
        pytest.skip("host_example not built")
    runs = {}
    for nctx in (1, 2, 3):
        out = subprocess.run([exe, "2", "5", "2600", "2", str(nctx)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        runs[nctx] = out.stdout.strip().splitlines()
    assert len(runs[1]) == 5 and all(" decode ok apply ok record " in ln for ln in runs[1])
    assert runs[2] == runs[1] and runs[3] == runs[1]


# ---------------------------------------------------------------- BASELINE sizes (176400-sample frames)
def test_full_size_frames_properties(ctx, oracle):
    # Size-independent properties at the benchmark's frame size (4 s @ 44.1 kHz):
    #  * causality: the residuals of a prefix do not depend on what follows (prefix run == head of the full run),
    #    which also crosses different chunkings of the pipeline;
    #  * FrameCoder::Predict with Optimize == Optimize and PredictStereoFrame run separately;
    #  * the baseline row of the Optimize trace is GetCost(base profile) on the centre window (codec.cpp:356-386);
    #  * one full frame against the oracle (a few seconds of CPU).
    n = 176400
    frames = [stereo_clip(300, n), stereo_clip(301, n)]
    best_e, outs = ctx.encode(2, frames, maps=True)
    full = ctx.predict(2, frames, maps=True)
    for a, b in zip(outs, full):
        for k in ("e0", "e1", "u0", "u1", "m0", "m1", "maxbpn"):
            assert np.array_equal(a[k], b[k]), k
    cut = 100000
    head = ctx.predict(2, [(L[:cut], R[:cut]) for L, R in frames])
    for h, f in zip(head, full):
        assert np.array_equal(h["e0"], f["e0"][:cut]) and np.array_equal(h["e1"], f["e1"][:cut])
    best_o, traces, rows = ctx.optimize(2, frames)
    assert same_bits(best_o, best_e) and list(rows) == [84, 84]
    base = sacopt.base_profile(2)[:, 2]
    wn = n // 8; frm = (n - wn) // 2
    costs, _ = ctx.get_costs(2, frames, np.tile(base, (2, 1, 1)), frm=[frm, frm], n=[wn, wn])
    for f in range(2):
        assert traces[f][0][0] == -1 and same_bits(traces[f][0][2:3], costs[f])
    L, R = frames[0]
    ref = oracle.predict(2, L, R, maps=True)
    for k in ("e0", "e1", "u0", "u1", "m0", "m1", "maxbpn"):
        assert np.array_equal(full[0][k], ref[k]), k


def test_create_rejects_non_rsqrt_table():
    # the table is kept as 16-bit words in LDS: entries must look like rsqrtss results ([0.5, 2), low byte clear)
    bad = np.full(2048, 3.0, np.float32)
    with pytest.raises(sacopt.SacoptError):
        sacopt.Context(0, bad, 10)
    noisy = np.full(2048, np.float32(0.7071067811865476), np.float32)   # 24 significant bits
    with pytest.raises(sacopt.SacoptError):
        sacopt.Context(0, noisy, 10)
