"""GPU: the entry point that allocates nothing (sacopt_cost_golomb_noalloc) against sacopt_cost_golomb, bit for bit,
through the C ABI: two calls in a row, the second with a smaller input (the static scratch keeps the first call's
contents), host and device buffers, every segment count, the size limit."""
import numpy as np
import pytest

from sac_amd import sacopt

pytestmark = pytest.mark.gpu


def _residuals(seed, n, scale):
    rs = np.random.RandomState(seed)
    e = (rs.standard_cauchy(n) * scale).clip(-2 ** 23, 2 ** 23).astype(np.int32)
    e[n // 3:n // 3 + 5000] = 0              # a silent stretch: the running mean decays below 1 (m = 1 branch)
    return e


class _DevBuf:
    """Device memory from the HIP runtime the library itself uses (no torch: its bundled runtime, initialised after the
    library's, finds no device)."""

    def __init__(self, host):
        import ctypes as C
        self.hip = C.CDLL("libamdhip64.so")
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(host.nbytes)) == 0
        assert self.hip.hipMemcpy(self.ptr, host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes), 1) == 0  # hipMemcpyHostToDevice

    def free(self):
        self.hip.hipFree(self.ptr)


def test_no_alloc_vs_oracle_and_sequential_fallback(rsqrt_table, oracle):
    # directly against the oracle's CostGolomb::Calc (codec.cpp:315-339), and the branch of k_cseg_fin where a guessed segment
    # state is wrong: >= 4096 zero residuals in front of a 16384-sample boundary make the guess exactly 0.0 while the true
    # mean is m * 0.98^k > 0, so the verification fails and one wave walks the whole buffer again
    with sacopt.Context(0, *rsqrt_table) as c:
        e = _residuals(4, 50000, 300.0)
        assert c.cost_golomb_noalloc(e, 2.0) == oracle.cost_golomb(e, 2.0)
        quiet = _residuals(5, 40000, 500.0)
        quiet[12288:16384] = 0                      # the last 4096 samples of segment 0 are silent
        for st in (0.0, 7.75):
            want = oracle.cost_golomb(quiet, st)
            assert c.cost_golomb_noalloc(quiet, st) == want
            assert c.cost_golomb(quiet, st) == want
        zeros = np.zeros(3 * 16384 + 100, np.int32)  # all silent, non-zero start state: every guess is 0.0, none is right
        want = oracle.cost_golomb(zeros, 3.25)
        assert c.cost_golomb_noalloc(zeros, 3.25) == want and c.cost_golomb(zeros, 3.25) == want
        # and the next call, smaller and well-behaved, on the scratch those left behind
        assert c.cost_golomb_noalloc(e[:9000], 0.0) == oracle.cost_golomb(e[:9000], 0.0)


def test_no_alloc_refuses_overlapping_calls_on_one_gpu(rsqrt_table):
    # every context of a GPU shares that GPU's static scratch: a call that finds it in use returns an error, it does not race
    import threading
    big = _residuals(6, 1 << 20, 200.0)
    with sacopt.Context(0, *rsqrt_table) as a, sacopt.Context(0, *rsqrt_table) as b:
        want = a.cost_golomb(big, 0.0)
        results, errors = [], []

        def work(c):
            for _ in range(6):
                try:
                    results.append(c.cost_golomb_noalloc(big, 0.0))
                except sacopt.SacoptError as ex:
                    errors.append(str(ex))

        ts = [threading.Thread(target=work, args=(c,)) for c in (a, b)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert results and all(r == want for r in results)          # whatever ran, ran alone
        assert all("already running on this GPU" in m for m in errors)
        assert len(results) + len(errors) == 12
        assert b.cost_golomb_noalloc(big[:5000], 1.0) == a.cost_golomb(big[:5000], 1.0)


def test_no_alloc_matches_cost_golomb(rsqrt_table):
    with sacopt.Context(0, *rsqrt_table) as c:
        big = _residuals(1, 300000, 900.0)     # 19 segments
        small = _residuals(2, 20001, 40.0)     # 2 segments, the second one short
        for buf, st in ((big, 0.0), (small, 3.25), (small[:777], 0.0), (big[:16384], 11.5), (big[:16385], 11.5)):
            want = c.cost_golomb(buf, st)
            got = c.cost_golomb_noalloc(buf, st)
            assert np.array([got]).tobytes() == np.array([want]).tobytes(), (len(buf), got, want)
        # carried state across calls, as one CostGolomb instance (codec.cpp:364)
        s1 = s2 = 0.0
        for part in (big[:100000], small, big[100000:]):
            c1, s1 = c.cost_golomb(part, s1)
            c2, s2 = c.cost_golomb_noalloc(part, s2)
            assert (c1, s1) == (c2, s2)
        # device buffer: nothing is staged
        d = _DevBuf(big)
        assert c.cost_golomb_noalloc((d.ptr.value, len(big)), 0.5) == c.cost_golomb(big, 0.5)
        assert c.cost_golomb_noalloc((d.ptr.value + 4 * 1000, 5000), 0.5) == c.cost_golomb(big[1000:6000], 0.5)
        d.free()
        # empty buffer: cost 0, state untouched; beyond the static scratch: a clear error
        assert c.cost_golomb_noalloc(big[:0], 1.5) == (0.0, 1.5)
        with pytest.raises(sacopt.SacoptError, match="exceeds the static scratch"):
            c.cost_golomb_noalloc(np.zeros((1 << 20) + 1, np.int32))
        full = _residuals(3, 1 << 20, 300.0)   # the maximum: 64 segments
        assert c.cost_golomb_noalloc(full, 0.0) == c.cost_golomb(full, 0.0)
        assert c.cost_golomb_noalloc(small, 0.0) == c.cost_golomb(small, 0.0)
