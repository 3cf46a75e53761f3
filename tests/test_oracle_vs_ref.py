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
