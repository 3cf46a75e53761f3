"""CPU: the oracle (oracle/sac_oracle.c) against the golden vectors generated from the real
reference (tests/golden/make_golden.py). Bit-exact everywhere (integer residuals, and FP64
stage outputs compared as raw bits)."""
import numpy as np
import pytest

from sac_amd.synth import stereo_clip


def same_bits(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def test_synth_is_deterministic(golden):
    L, R = stereo_clip(11, 6000)
    assert np.array_equal(L, golden["pred_L"]) and np.array_equal(R, golden["pred_R"])
    L, R = stereo_clip(12, 8192)
    assert np.array_equal(L, golden["opt_L"]) and np.array_equal(R, golden["opt_R"])


def test_cholesky_kat(oracle):
    # reference: Cholesky::Test (src/math/cholesky.cpp:3-14)
    assert oracle.cholesky_kat()


def test_base_profiles(oracle, golden):
    for prof in (0, 1, 2):
        assert same_bits(oracle.base_profile(prof), golden[f"base{prof}"])


@pytest.mark.parametrize("prof", [0, 1, 2])
def test_predict_stereo_frame(oracle, golden, prof):
    L, R = golden["pred_L"], golden["pred_R"]
    d = oracle.predict(prof, L, R, frm=0, n=4096, maps=True)
    for k, v in d.items():
        assert np.array_equal(v, golden[f"pred{prof}_{k}"]), k
    d = oracle.predict(prof, L, R, frm=1500, n=2000, maps=True)
    for k, v in d.items():
        assert np.array_equal(v, golden[f"predw{prof}_{k}"]), k


def test_predict_custom_coefs(oracle, golden):
    d = oracle.predict(2, golden["pred_L"], golden["pred_R"], frm=0, n=3000, coefs=golden["predc_coefs"])
    assert np.array_equal(d["e0"], golden["predc_e0"]) and np.array_equal(d["e1"], golden["predc_e1"])


@pytest.mark.parametrize("prof", [1, 2])
def test_stage_traces(oracle, golden, prof):
    p, v, _, _ = oracle.cascade_trace(prof, golden["pred_L"][:96], golden["pred_R"][:96])
    assert same_bits(p, golden[f"trace{prof}_p"]) and same_bits(v, golden[f"trace{prof}_v"])


@pytest.mark.parametrize("S,D,m1,m2", [(1280, 640, 1e-5, 5e-6), (256, 128, 5e-4, 1e-4), (32, 16, 3e-3, 3e-3), (8, 4, 0.01, 0.01)])
def test_lms_pair(oracle, golden, S, D, m1, m2):
    p0, p1 = oracle.lms_pair(S, D, m1, m2, golden["lms_x0"], golden["lms_x1"])
    assert same_bits(p0, golden[f"lms{S}_p0"]) and same_bits(p1, golden[f"lms{S}_p1"])


def test_lpc3_pair(oracle, golden):
    p0, p1 = oracle.lpc3_pair(0.997, golden["pred_L"][:3000], golden["pred_R"][:3000])
    assert same_bits(p0, golden["lpc3_p0"]) and same_bits(p1, golden["lpc3_p1"])


def test_rls(oracle, golden):
    assert same_bits(oracle.rls_run(4, 0.998, golden["rls_pv"], golden["rls_val"]), golden["rls4_p"])
    assert same_bits(oracle.rls_run(3, 0.998, np.ascontiguousarray(golden["rls_pv"][:, :3]), golden["rls_val"]), golden["rls3_p"])


def test_cost_golomb_carried_state(oracle, golden):
    e = golden["cost_e"]
    c1, s1 = oracle.cost_golomb(e[:2500], 0.0)
    c2, s2 = oracle.cost_golomb(e[2500:], s1)
    assert same_bits(np.array([c1, s1, c2, s2, oracle.cost_meanrms(e)]), golden["cost_vals"])
    assert oracle.cost_golomb(e[:0], 1.5) == (0.0, 1.5)  # empty input: cost 0, state untouched


def test_optimize_trace(oracle, golden):
    # FrameCoder::Optimize (src/codec.cpp:354-408): 84 GetCost calls, 8 selected values
    best, trace = oracle.optimize(2, golden["opt_L"], golden["opt_R"])
    assert len(trace) == 84
    assert same_bits(trace, golden["opt_trace"]) and same_bits(best, golden["opt_best"])


def test_optimize_trace_normal_profile(oracle, golden):
    best, trace = oracle.optimize(1, golden["opt_L"], golden["opt_R"], ncoef_loop=6)
    assert same_bits(trace, golden["optn_trace"]) and same_bits(best[:6], golden["optn_best"])


def test_ragged_and_empty(oracle):
    L, R = stereo_clip(3, 50)
    d = oracle.predict(2, L, R, frm=0, n=0)
    assert len(d["e0"]) == 0
    d = oracle.predict(2, L, R, frm=49, n=1)
    assert d["e0"][0] == L[49] and d["e1"][0] == R[49]  # fresh state predicts 0


def unpredict_cases(g):
    """(name, profile, residuals, keyword arguments, expected samples) of tests/golden/golden_unpredict.npz"""
    for prof in (0, 1, 2):
        yield f"plain{prof}", prof, (g[f"e{prof}_0"], g[f"e{prof}_1"]), {}, (g[f"v{prof}_0"], g[f"v{prof}_1"])
        yield f"garbage{prof}", prof, (g["g0"], g["g1"]), {}, (g[f"vg{prof}_0"], g[f"vg{prof}_1"])
        if prof:
            yield (f"mapped{prof}", prof, (g[f"m{prof}_0"], g[f"m{prof}_1"]), dict(used0=g["used0"], used1=g["used1"]),
                   (g[f"vm{prof}_0"], g[f"vm{prof}_1"]))
    e = (g["e2_0"], g["e2_1"])
    yield "clamp2", 2, e, dict(minmax=g["mm"]), (g["vc2_0"], g["vc2_1"])
    yield "coefs2", 2, (g["g0"], g["g1"]), dict(coefs=g["coefs2"]), (g["vk2_0"], g["vk2_1"])


def test_unpredict_stereo_frame(oracle, golden_unpredict):
    # FrameCoder::UnpredictStereoFrame (codec.cpp:216-235), Remap::Unmap (map.cpp:154-168)
    g = golden_unpredict
    for name, prof, e, kw, want in unpredict_cases(g):
        v = oracle.unpredict(prof, *e, **kw)
        assert np.array_equal(v[0], want[0]) and np.array_equal(v[1], want[1]), name
    assert np.array_equal(g["v2_0"], g["L"]) and np.array_equal(g["vm1_1"], g["Rs"])  # the fixtures are round trips


def test_ols32_stage_matches_reference_class(oracle, golden):
    # LPC3(0.997, 24, 8, 8, 500): the oracle against rows written by the reference's own class (golden_config5.npz: lpc32_*)
    import os
    c5 = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_config5.npz"), allow_pickle=False))
    p0, p1 = oracle.lpc3_pair(0.997, golden["pred_L"][:3000], golden["pred_R"][:3000], S=24, D=8)
    assert p0.tobytes() == c5["lpc32_p0"].tobytes() and p1.tobytes() == c5["lpc32_p1"].tobytes()
    # the oracle-made rows of profile 3 are reproducible (they pin the GPU, not the oracle)
    d = oracle.predict(3, golden["pred_L"], golden["pred_R"], frm=0, n=4096)
    assert np.array_equal(d["e0"], c5["o32_e0"]) and np.array_equal(d["e1"], c5["o32_e1"])


def test_remap_map_forms(oracle):
    # the oracle's Remap::Map visits only the steps whose value can be in the table; the literal loop of map.cpp:141-152 takes
    # every step. Same counts: ordinary predictions / residuals, and walks that leave int32 (a few seconds: 2^31 steps each)
    import oracle_lib
    rs = np.random.RandomState(1)
    src = (rs.randint(-32768, 32768, 3000) // 7 * 7).astype(np.int32)
    pred = rs.randint(-200000, 200000, 4000).astype(np.int32); err = rs.randint(-200000, 200000, 4000).astype(np.int32)
    pred[:500] = rs.randint(-40000, 40000, 500); err[:500] = rs.randint(-300, 300, 500)
    assert np.array_equal(oracle.remap_map(src, pred, err), oracle.remap_map_literal(src, pred, err))
    src, pred, err = oracle_lib.map_wrap_cases()
    got = oracle.remap_map(src, pred, err)
    assert np.array_equal(got, oracle.remap_map_literal(src, pred, err))
    assert got[0] == 0 and got[1] < 0 and got[2] > 0   # -INT32_MIN stays negative: no steps; the wrapped walks do count
