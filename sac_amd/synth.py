"""Integer-deterministic synthetic stereo audio.

Everything is integer arithmetic (uint32 LCG noise, uint32 phase accumulators, a parabolic
integer sine, triangular envelopes), so the same seed gives bit-identical samples on every
host -- no libm, no float rounding. L/R are sums of a few partials with slow envelopes plus
partially correlated noise at about -34 dBFS, clipped to the 16- (or 24-) bit range.
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


_LCG_CACHE = None


def _lcg_tables():
    global _LCG_CACHE
    if _LCG_CACHE is None:
        B = 1 << 12
        A = 1664525
        ai = [1] * B
        si = [0] * B
        for i in range(1, B):
            ai[i] = (ai[i - 1] * A) & 0xFFFFFFFF
            si[i] = (si[i - 1] * A + 1) & 0xFFFFFFFF
        aB = np.uint64((ai[B - 1] * A) & 0xFFFFFFFF)
        sB = np.uint64((si[B - 1] * A + 1) & 0xFFFFFFFF)
        _LCG_CACHE = (np.array(ai, dtype=np.uint64), np.array(si, dtype=np.uint64), aB, sB)
    return _LCG_CACHE


def _lcg_stream(seed: int, n: int) -> np.ndarray:
    """n values of x <- 1664525*x + 1013904223 (mod 2^32), returned as int64 in [0, 2^32)."""
    # closed form per block via repeated squaring would be overkill; vectorise with the
    # jump-ahead identity x_{k} = A^k x0 + C (A^k-1)/(A-1) computed by cumulative products.
    a = np.uint64(1664525)
    c = np.uint64(1013904223)
    out = np.empty(n, dtype=np.uint64)
    x = np.uint64((seed * 2654435761 + 12345) & 0xFFFFFFFF)
    # process in blocks with doubling: build A^i and S_i = sum_{j<i} A^j for i < B
    ai, si, aB, sB = _lcg_tables()
    B = len(ai)
    pos = 0
    while pos < n:
        m = min(B, n - pos)
        out[pos:pos + m] = (ai[:m] * x + si[:m] * c) & _M32
        x = (aB * x + sB * c) & _M32
        pos += m
    return out.astype(np.int64)


def _isin(phase: np.ndarray) -> np.ndarray:
    """Parabolic integer sine of a uint32 phase (as int64): output in about [-32768, 32767]."""
    t = (phase >> 16) & 0xFFFF          # 0..65535
    u = t - 32768                       # -32768..32767
    y = (u * (32768 - np.abs(u))) >> 13  # parabola, peak ~ +-32768
    return -y


def _tri_env(n: int, period: int, offset: int) -> np.ndarray:
    """Triangular envelope in [0, 256]."""
    k = (np.arange(n, dtype=np.int64) + offset) % period
    half = period // 2
    return (256 * np.where(k < half, k, period - k)) // max(half, 1)


def stereo_clip(seed: int, n: int, rate: int = 44100, bits: int = 16):
    """Return (L, R) int32 arrays of n samples."""
    rng = _lcg_stream(seed, 64)
    idx = np.arange(n, dtype=np.int64)
    L = np.zeros(n, dtype=np.int64)
    R = np.zeros(n, dtype=np.int64)
    npart = 3 + int(rng[0] % 3)
    for p in range(npart):
        f = 60 + int(rng[1 + 4 * p] % 3000)                     # Hz
        inc = (f << 32) // rate                                 # phase increment
        ph0 = int(rng[2 + 4 * p])
        phase = (ph0 + inc * idx) & 0xFFFFFFFF
        s = _isin(phase)
        amp = 1200 + int(rng[3 + 4 * p] % 5000)                 # of 32768
        env = _tri_env(n, rate * (2 + int(rng[4 + 4 * p] % 5)), int(rng[4 + 4 * p] >> 8) % rate)
        part = (s * amp * (64 + env)) >> 23                     # ~ +-amp*1.2
        pan = int(rng[40 + p] % 257)                            # 0..256
        L += (part * pan) >> 8
        phase_r = (phase + (int(rng[48 + p]) & 0x0FFFFFFF)) & 0xFFFFFFFF
        part_r = (_isin(phase_r) * amp * (64 + env)) >> 23
        R += (part_r * (256 - pan)) >> 8
    common = (_lcg_stream(seed ^ 0x5A5A5A5A, n) >> 16) - 32768   # 16-bit white
    own_l = (_lcg_stream(seed ^ 0x13579BDF, n) >> 16) - 32768
    own_r = (_lcg_stream(seed ^ 0x2468ACE0, n) >> 16) - 32768
    # light 2-tap colouring so the noise is not perfectly white
    nl = (2 * common + own_l) >> 2
    nr = (2 * common + own_r) >> 2
    nl[1:] += nl[:-1] >> 1
    nr[1:] += nr[:-1] >> 1
    L += nl >> 5
    R += nr >> 5
    if bits > 16:
        L <<= (bits - 16)
        R <<= (bits - 16)
        L += (_lcg_stream(seed ^ 0x0F0F0F0F, n) >> (48 - bits)) - (1 << (bits - 17))
        R += (_lcg_stream(seed ^ 0x70707070, n) >> (48 - bits)) - (1 << (bits - 17))
    lim = (1 << (bits - 1)) - 1
    np.clip(L, -lim - 1, lim, out=L)
    np.clip(R, -lim - 1, lim, out=R)
    return L.astype(np.int32), R.astype(np.int32)


def degenerate_clips(n=3000):
    """Inputs that stress the OLS stage (rank-deficient covariance, "matrix indefinit" handling) and the
    adaptive stages (constant, periodic, sparse): name -> (L, R), integer-deterministic."""
    t = np.arange(n, dtype=np.int64)
    tri = np.abs((t * 37) % 2048 - 1024) * 12 - 6000          # integer triangle wave
    return {
        "dc": (np.full(n, 1000, np.int32), np.full(n, -700, np.int32)),
        "tri": (tri.astype(np.int32), np.roll(tri, 5).astype(np.int32)),
        "ramp": ((t % 64 * 100).astype(np.int32), (t % 32 * 50).astype(np.int32)),
        "impulse": (np.where(t % 500 == 0, 30000, 0).astype(np.int32), np.where(t % 333 == 0, -30000, 0).astype(np.int32)),
        "wide24": (((t * 7919) % 16000000 - 8000000).astype(np.int32), ((t * 104729) % 9000000 - 4500000).astype(np.int32)),
    }
