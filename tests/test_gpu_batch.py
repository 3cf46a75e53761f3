"""GPU: the call shapes a rank runs at N > 1 (BASELINE configs[3]) and the library's fallback stream set, against the oracle.

sacopt_encode_frames / sacopt_optimize_frames change behaviour with the batch size (sac_amd/csrc/sacopt.cpp): more than 16
frames -> the shared stream set instead of the split one; more than 20 chains per launch -> the two-channel (1280,640)
kernel instead of k_lms1c; more than 32 lead workgroups -> the full-mask stream instead of the 32 reserved CUs; optimizer
rounds of several hundred chains; windows of >= 4096 samples are cut into chunks with carried stage state. Everything is
compared bit for bit, as in test_gpu_parity.py, with the fixture's rsqrtss table."""
import os

import numpy as np
import pytest

from sac_amd import sacopt
from sac_amd.synth import stereo_clip

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


KEYS = ("e0", "e1", "u0", "u1", "m0", "m1", "maxbpn")


def _ragged_frames(count, seed):
    # 18 000 .. 44 000 samples: every frame takes the chunk-pipelined full-frame pass (> 16 384), the windows (n / 8) of the
    # long ones (>= 32 768) are cut into chunks in the optimizer's rounds, the short ones are not
    rs = np.random.RandomState(seed)
    lens = [int(x) for x in rs.randint(18000, 30000, count - 8)] + [32768, 33001, 36000, 39999, 40960, 41000, 43210, 44000]
    rs.shuffle(lens)
    return [stereo_clip(7000 + i, n) for i, n in enumerate(lens)]


def test_large_batch_encode_and_optimize_vs_oracle(rsqrt_table, oracle):
    # FrameCoder::Predict with Optimize enabled (codec.cpp:442-461) for 52 frames in ONE call (520-624 chains per round: beyond
    # 512 the (256,128) stage takes its undivided-chain instance, beyond 48 frames the full-frame pass the two-channel kernel)
    frames = _ragged_frames(52, 3)
    with sacopt.Context(0, *rsqrt_table) as c:
        best_e, outs = c.encode(2, frames, maps=True)
        best_o, traces, rows = c.optimize(2, frames)       # 52 x 10..12 chains per round
        full = c.predict(2, frames, maps=True)
        assert list(rows) == [84] * len(frames)
        assert same_bits(best_e, best_o)
        for f in range(len(frames)):
            for k in KEYS:
                assert np.array_equal(outs[f][k], full[f][k]), (f, k)
        # the same frames in calls of one and of three (k_lms1c, split stream set, reserved CUs): identical results
        for f in (0, 17, 51):
            b1, o1 = c.encode(2, [frames[f]], maps=True)
            assert same_bits(b1[0], best_e[f]), f
            for k in KEYS:
                assert np.array_equal(o1[0][k], outs[f][k]), (f, k)
        b3, t3, _ = c.optimize(2, frames[20:23])
        assert same_bits(b3, best_o[20:23]) and all(same_bits(t3[i], traces[20 + i]) for i in range(3))
        # sampled frames against the oracle: winners, the 84-row cost trace, residuals and Remap outputs
        longest = int(np.argmax([len(L) for L, _ in frames]))
        for f in sorted({1, 22, 40, longest}):
            L, R = frames[f]
            ob, ot = oracle.optimize(2, L, R)
            assert same_bits(best_o[f], ob) and same_bits(traces[f], ot), f
            ref = oracle.predict(2, L, R, maps=True)
            for k in KEYS:
                assert np.array_equal(outs[f][k], ref[k]), (f, k)


def test_mid_batch_default_stream_set_vs_oracle(rsqrt_table, oracle):
    # 26 frames: the default stream set (17..32 frames) with 260-312 chains per round -- the lead stage's chains go out as three
    # launch sequences, two of them on idle streams of the small-batch set; the full-frame pass runs k_lms1c on sAall
    frames = _ragged_frames(26, 5)
    with sacopt.Context(0, *rsqrt_table) as c:
        best_e, outs = c.encode(2, frames, maps=True)
        best_o, traces, rows = c.optimize(2, frames)
        assert list(rows) == [84] * len(frames)
        assert same_bits(best_e, best_o)
        b3, t3, _ = c.optimize(2, frames[7:10])
        assert same_bits(b3, best_o[7:10]) and all(same_bits(t3[i], traces[7 + i]) for i in range(3))
        for f in (3, 25):
            L, R = frames[f]
            ob, ot = oracle.optimize(2, L, R)
            assert same_bits(best_o[f], ob) and same_bits(traces[f], ot), f
            ref = oracle.predict(2, L, R, maps=True)
            for k in KEYS:
                assert np.array_equal(outs[f][k], ref[k]), (f, k)


def test_large_batch_normal_profile_vs_oracle(rsqrt_table, oracle):
    # the normal profile's lead stage is (256,128): same batch logic, other kernels
    frames = _ragged_frames(40, 4)
    with sacopt.Context(0, *rsqrt_table) as c:
        best_e, outs = c.encode(1, frames, maps=True)
        best_o, traces, rows = c.optimize(1, frames)
        assert same_bits(best_e, best_o)
        for f in (0, 13, 39):
            L, R = frames[f]
            ob, ot = oracle.optimize(1, L, R, ncoef_loop=6)
            assert same_bits(best_o[f], ob[:6]) and same_bits(traces[f], ot), f
            ref = oracle.predict(1, L, R, maps=True)
            for k in KEYS:
                assert np.array_equal(outs[f][k], ref[k]), (f, k)


def test_unmasked_stream_fallback_vs_oracle(rsqrt_table, oracle):
    # SACOPT_CUMASK=0 (or a runtime without hipExtStreamCreateWithCUMask): plain streams, no sD / sE / split set / sAall
    # (sacopt_create); the pipelines then run their followers on fewer streams. Same results.
    old = os.environ.get("SACOPT_CUMASK")
    os.environ["SACOPT_CUMASK"] = "0"
    try:
        c = sacopt.Context(0, *rsqrt_table)
    finally:
        if old is None:
            del os.environ["SACOPT_CUMASK"]
        else:
            os.environ["SACOPT_CUMASK"] = old
    try:
        frames = [stereo_clip(7100, 40000), stereo_clip(7101, 36001), stereo_clip(7102, 3000)]
        best_e, outs = c.encode(2, frames, maps=True)
        best_o, traces, rows = c.optimize(2, frames)
        assert same_bits(best_e, best_o)
        for f, (L, R) in enumerate(frames):
            ob, ot = oracle.optimize(2, L, R)
            assert same_bits(best_o[f], ob) and same_bits(traces[f], ot), f
            ref = oracle.predict(2, L, R, maps=True)
            for k in KEYS:
                assert np.array_equal(outs[f][k], ref[k]), (f, k)
        # the other profiles and the decoder on the same context
        for prof in (0, 1):
            o = c.predict(prof, frames[:2], maps=True)
            for (L, R), got in zip(frames[:2], o):
                ref = oracle.predict(prof, L, R, maps=True)
                for k in KEYS:
                    assert np.array_equal(got[k], ref[k]), (prof, k)
        L, R = frames[2]
        dec = c.unpredict(2, [dict(e0=outs[2]["e0"], e1=outs[2]["e1"])])[0]
        assert np.array_equal(dec[0], L) and np.array_equal(dec[1], R)
        # many frames on the fallback set as well (two-channel kernel, no full-mask stream to move to)
        many = [stereo_clip(7200 + i, 17000 + 13 * i) for i in range(24)]
        got = c.predict(2, many)
        for f in (0, 23):
            ref = oracle.predict(2, *many[f])
            assert np.array_equal(got[f]["e0"], ref["e0"]) and np.array_equal(got[f]["e1"], ref["e1"]), f
    finally:
        c.close()


@pytest.mark.parametrize("prof,count", [(2, 5), (1, 4), (0, 3)])
def test_encode_frames_selected_vs_two_calls(rsqrt_table, oracle, prof, count):
    # sacopt_encode_frames_selected (SURVEY 8f.2 in one call: the parts of the full-frame pass are released as the optimizer fixes
    # their parameters) against Optimize + PredictStereoFrame(selected parameters) as two calls, and one frame against the oracle
    frames = _ragged_frames(8 + count, 7)[:count] + [stereo_clip(7100, 3000)]   # pipelined frames and one short frame
    with sacopt.Context(0, *rsqrt_table) as c:
        best1, outs1 = c.encode(prof, frames, maps=True, selected=True)
        best2, _, _ = c.optimize(prof, frames, ncoef_loop=sacopt.NCOEF[prof])
        outs2 = c.predict(prof, frames, coefs=best2, maps=True)
        assert same_bits(best1, best2)
        for f in range(len(frames)):
            for k in KEYS:
                assert np.array_equal(outs1[f][k], outs2[f][k]), (f, k)
        # fewer coefficients than the profile has: the rest stay at their base values
        part = min(2, sacopt.NCOEF[prof])
        b3, o3 = c.encode(prof, frames[:2], maps=True, selected=True, ncoef_loop=part)
        b4, _, _ = c.optimize(prof, frames[:2], ncoef_loop=part)
        assert same_bits(b3, b4)
        o4 = c.predict(prof, frames[:2], coefs=b4, maps=True)
        for f in range(2):
            for k in KEYS:
                assert np.array_equal(o3[f][k], o4[f][k]), (f, k)
    L, R = frames[1]
    nc = sacopt.NCOEF[prof]
    ob, _ = oracle.optimize(prof, L, R, ncoef_loop=nc)
    assert same_bits(best1[1], ob[:nc])
    ref = oracle.predict(prof, L, R, coefs=ob[:nc], maps=True)
    for k in KEYS:
        assert np.array_equal(outs1[1][k], ref[k]), k
