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
