"""GPU: sacopt_unpredict_stereo_frames (decoder-side cascade) through the C ABI against the golden
fixtures generated from the reference (tests/golden/golden_unpredict.npz) and the oracle. int32 samples, bit-exact.
Runs with the FIXTURE's rsqrtss table."""
import numpy as np
import pytest

from sac_amd import sacopt
from sac_amd.synth import stereo_clip, degenerate_clips
from test_oracle_golden import unpredict_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(rsqrt_table):
    c = sacopt.Context(0, *rsqrt_table)
    yield c
    c.close()


def _frame(e, kw):
    return dict(e0=e[0], e1=e[1], used0=kw.get("used0"), used1=kw.get("used1"), minmax=kw.get("minmax"))


def test_unpredict_golden(ctx, golden_unpredict):
    # every fixture case of FrameCoder::UnpredictStereoFrame (codec.cpp:216-235): plain, garbage residuals, mapped
    # channels (Remap::Unmap, map.cpp:154-168), the high profile's clamp (profile.cpp:163,170), custom coefficients
    for name, prof, e, kw, want in unpredict_cases(golden_unpredict):
        cf = kw.get("coefs")
        v = ctx.unpredict(prof, [_frame(e, kw)], coefs=None if cf is None else np.asarray(cf)[None, :])[0]
        assert np.array_equal(v[0], want[0]) and np.array_equal(v[1], want[1]), name


@pytest.mark.parametrize("prof", [0, 1, 2])
def test_unpredict_batch_ragged(ctx, oracle, prof):
    # several frames of different lengths in one call (one workgroup each), empty and 1-sample frames included;
    # encode on the GPU, decode on the GPU, compare with the input and with the oracle's decoder
    lens = [0, 1, 255, 256, 257, 1031, 3000]
    clips = [stereo_clip(70 + i, max(n, 1)) for i, n in enumerate(lens)]
    clips = [(L[:n], R[:n]) for (L, R), n in zip(clips, lens)]
    enc = ctx.predict(prof, [(L, R) for L, R in clips])
    dec = ctx.unpredict(prof, [dict(e0=d["e0"], e1=d["e1"]) for d in enc])
    for (L, R), d, v in zip(clips, enc, dec):
        o = oracle.unpredict(prof, d["e0"], d["e1"])
        assert np.array_equal(v[0], o[0]) and np.array_equal(v[1], o[1])
        assert np.array_equal(v[0], L) and np.array_equal(v[1], R)


@pytest.mark.parametrize("prof", [1, 2])
def test_unpredict_mapped_roundtrip(ctx, oracle, prof):
    # sparse sample values (every 7th level): the mapped residuals differ from the plain ones; one channel mapped,
    # then both; the decoder gets the tables of Remap::Analyse
    L, R = stereo_clip(81, 4000)
    L = (L // 7) * 7
    R = (R // 5) * 5
    enc = ctx.predict(prof, [(L, R)], maps=True)[0]
    s2u_inv = lambda u: np.where(u & 1, ((u.astype(np.int64) + 1) >> 1), -(u.astype(np.int64) >> 1)).astype(np.int32)  # utils.h:62-66
    m0, m1 = s2u_inv(enc["m0"]), s2u_inv(enc["m1"])
    assert not np.array_equal(m0, enc["e0"])
    u0, u1 = oracle.remap_tables(L), oracle.remap_tables(R)
    for e, kw in (((m0, enc["e1"]), dict(used0=u0)), ((enc["e0"], m1), dict(used1=u1)), ((m0, m1), dict(used0=u0, used1=u1))):
        v = ctx.unpredict(prof, [_frame(e, kw)])[0]
        o = oracle.unpredict(prof, *e, **kw)
        assert np.array_equal(v[0], o[0]) and np.array_equal(v[1], o[1])
        assert np.array_equal(v[0], L) and np.array_equal(v[1], R)


def test_unpredict_unmap_out_of_table(ctx, oracle):
    # residuals that ask for more used values than the table holds / predictions far outside +-32768
    rng = np.random.default_rng(5)
    n = 600
    e0 = rng.integers(-40000, 40000, n).astype(np.int32)
    e1 = rng.integers(-300, 300, n).astype(np.int32)
    L, R = stereo_clip(82, 2000)
    u0, u1 = oracle.remap_tables((L // 9) * 9), oracle.remap_tables(R)
    for prof in (1, 2):
        kw = dict(used0=u0, used1=u1)
        v = ctx.unpredict(prof, [_frame((e0, e1), kw)])[0]
        o = oracle.unpredict(prof, e0, e1, **kw)
        assert np.array_equal(v[0], o[0]) and np.array_equal(v[1], o[1])


def test_unpredict_degenerate(ctx, oracle):
    for name, (L, R) in degenerate_clips(1500).items():
        for prof in (0, 1, 2):
            e = oracle.predict(prof, L, R)
            v = ctx.unpredict(prof, [dict(e0=e["e0"], e1=e["e1"])])[0]
            assert np.array_equal(v[0], L) and np.array_equal(v[1], R), (name, prof)


def test_unpredict_long_high(ctx, oracle):
    # crosses several factorisation periods, IO blocks and the (1280,640) history length
    L, R = stereo_clip(83, 12000)
    e = ctx.predict(2, [(L, R)])[0]
    v = ctx.unpredict(2, [dict(e0=e["e0"], e1=e["e1"])])[0]
    assert np.array_equal(v[0], L) and np.array_equal(v[1], R)


def test_optimised_parameters_round_trip(ctx):
    # a stream that carries the optimiser's per-frame parameters -- encode each frame with the values
    # Optimize selected (different per frame), decode with the same values: samples restored; decoding with the base
    # profile instead does NOT restore them (the parameters really are in the loop)
    frames = [stereo_clip(90, 9000), stereo_clip(91, 7000)]
    best, _, _ = ctx.optimize(2, frames)
    enc = ctx.predict(2, frames, coefs=best)
    dec = ctx.unpredict(2, [dict(e0=d["e0"], e1=d["e1"]) for d in enc], coefs=best)
    for (L, R), v in zip(frames, dec):
        assert np.array_equal(v[0], L) and np.array_equal(v[1], R)
    if not np.array_equal(best[0], sacopt.base_profile(2)[:, 2]):
        wrong = ctx.unpredict(2, [dict(e0=enc[0]["e0"], e1=enc[0]["e1"])])[0]
        assert not np.array_equal(wrong[0], frames[0][0])


# ---------------------------------------------------------------- BASELINE sizes: encode -> decode round trips
def test_full_size_round_trip_44k(ctx):
    # BASELINE configs[1..3] frame size (176400 = 4 s @ 44.1 kHz), every profile: FrameCoder::Predict (optimizer off, Remap
    # outputs on) then UnpredictStereoFrame restores the samples; channel 1 goes through Remap::Map / Unmap
    n = 176400
    frames = [stereo_clip(310, n), stereo_clip(311, n)]
    for prof in (0, 1, 2):
        _, enc = ctx.encode(prof, frames, ncoef_loop=0, maps=True)
        dec_in = []
        for (L, R), d in zip(frames, enc):
            m1 = np.where(d["m1"] & 1, (d["m1"].astype(np.int64) + 1) >> 1, -(d["m1"].astype(np.int64) >> 1)).astype(np.int32)
            used1 = np.zeros(2 * 32769, np.uint8)  # Remap::Analyse of channel 1 (map.cpp:111-130)
            used1[32769 + R[(R > 0) & (R <= 32768)]] = 1
            used1[-R[(R < 0) & (R >= -32768)]] = 1
            if prof == 0:
                dec_in.append(dict(e0=d["e0"], e1=d["e1"]))  # StereoFast::Unpredict never unmaps (profile.cpp:34-43)
            else:
                dec_in.append(dict(e0=d["e0"], e1=m1, used1=used1))
        dec = ctx.unpredict(prof, dec_in)
        for (L, R), v in zip(frames, dec):
            assert np.array_equal(v[0], L) and np.array_equal(v[1], R), prof


def test_full_size_round_trip_config5(ctx):
    # BASELINE configs[4] shape: 24-bit 96 kHz, 8 channels = 4 stereo pairs, framesize 384000 (4 s): each pair is a frame
    # of the batch; high profile, encode with per-pair parameters, decode with the same
    n = 384000
    frames = [stereo_clip(320 + k, n, rate=96000, bits=24) for k in range(4)]
    assert max(int(np.abs(f[0]).max()) for f in frames) > (1 << 20)
    base = sacopt.base_profile(2)[:, 2]
    coefs = np.tile(base, (4, 1))
    coefs[1, 0] = 0.9985; coefs[2, 1] = 2e-5; coefs[3, 7] = 0.9995
    for prof in (2, 3):   # 3 = SACOPT_PROFILE_HIGH32, "32-tap OLS + long LMS cascade" (parity unpinned: no reference profile has the shape)
        enc = ctx.predict(prof, frames, coefs=coefs)
        dec = ctx.unpredict(prof, [dict(e0=d["e0"], e1=d["e1"]) for d in enc], coefs=coefs)
        for (L, R), v in zip(frames, dec):
            assert np.array_equal(v[0], L) and np.array_equal(v[1], R), prof


def test_unpredict_error_paths(ctx):
    # non-zero status + message, nothing thrown across the ABI (INTEGRATION.md "Error behaviour")
    import ctypes as C
    L = sacopt.lib()
    arr = (sacopt.DecFrame * 1)()
    arr[0].numsamples = 16  # null residual / sample pointers
    assert L.sacopt_unpredict_stereo_frames(ctx._h, 2, 1, arr, None) != 0
    assert b"bad frame" in L.sacopt_last_error(ctx._h)
    assert L.sacopt_unpredict_stereo_frames(ctx._h, 7, 1, arr, None) != 0
    assert b"unknown profile" in L.sacopt_last_error(ctx._h)
    arr[0].numsamples = -1
    assert L.sacopt_unpredict_stereo_frames(ctx._h, 2, 1, arr, None) != 0
    assert L.sacopt_unpredict_stereo_frames(ctx._h, 2, 0, None, None) == 0  # no frames: nothing to do
    assert L.sacopt_unpredict_stereo_frames(None, 2, 0, None, None) != 0    # no context
    # the context is still usable after the failures
    Ls, Rs = stereo_clip(95, 300)
    e = ctx.predict(0, [(Ls, Rs)])[0]
    v = ctx.unpredict(0, [dict(e0=e["e0"], e1=e["e1"])])[0]
    assert np.array_equal(v[0], Ls) and np.array_equal(v[1], Rs)
