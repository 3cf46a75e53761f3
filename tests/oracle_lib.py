"""ctypes bindings for the test oracles (TEST INFRASTRUCTURE ONLY).

  ORACLE : oracle/_build/liboracle.so  -- the C restatement (oracle/sac_oracle.c)
  REF    : oracle/_ref/libsacref.so    -- the real reference + harness (may be absent)
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
_REF_SO = os.path.join(ROOT, "oracle", "_ref", "libsacref.so")

i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
vp = C.c_void_p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(vp)


def build_oracle():
    if not os.path.exists(_ORACLE_SO) or os.path.getmtime(_ORACLE_SO) < max(
            os.path.getmtime(os.path.join(ROOT, "oracle", f)) for f in ("sac_oracle.c", "sac_coder.c")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return _ORACLE_SO


class _Lib:
    """Common python-level API over either library (prefix 'oracle_' or 'ref_')."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.pre = prefix
        L = self.lib
        f = getattr(L, prefix + "predict_stereo_frame")
        f.restype = None if prefix == "ref_" else C.c_int
        g = getattr(L, prefix + "cost_golomb")
        g.restype = C.c_double
        g.argtypes = [i32p, C.c_int, C.POINTER(C.c_double)]
        g = getattr(L, prefix + "cost_meanrms")
        g.restype = C.c_double
        g.argtypes = [i32p, C.c_int]
        getattr(L, prefix + "rsqrt").restype = C.c_float
        getattr(L, prefix + "rsqrt").argtypes = [C.c_float]

    def base_profile(self, profile):
        out = np.zeros(24, dtype=np.float64)
        n = getattr(self.lib, self.pre + "base_profile")(C.c_int(profile), _ptr(out))
        return out[:3 * n].reshape(n, 3).copy()

    def predict(self, profile, s0, s1, frm=0, n=None, coefs=None, maps=False):
        s0 = np.ascontiguousarray(s0, dtype=np.int32)
        s1 = np.ascontiguousarray(s1, dtype=np.int32)
        total = len(s0)
        if n is None:
            n = total - frm
        e0 = np.zeros(n, np.int32); e1 = np.zeros(n, np.int32)
        u0 = np.zeros(n, np.int32); u1 = np.zeros(n, np.int32)
        m0 = np.zeros(n, np.int32) if maps else None
        m1 = np.zeros(n, np.int32) if maps else None
        mb = np.zeros(4, np.int32) if maps else None
        cf = None if coefs is None else np.ascontiguousarray(coefs, dtype=np.float64)
        f = getattr(self.lib, self.pre + "predict_stereo_frame")
        if self.pre == "ref_":
            f(C.c_int(profile), _ptr(cf), C.c_int(0 if cf is None else len(cf)), _ptr(s0), _ptr(s1),
              C.c_int(total), C.c_int(frm), C.c_int(n), _ptr(e0), _ptr(e1), _ptr(u0), _ptr(u1),
              _ptr(m0), _ptr(m1), _ptr(mb))
        else:
            f(C.c_int(profile), _ptr(cf), _ptr(s0), _ptr(s1), C.c_int(total), C.c_int(frm),
              C.c_int(n), _ptr(e0), _ptr(e1), _ptr(u0), _ptr(u1), _ptr(m0), _ptr(m1), _ptr(mb))
        return dict(e0=e0, e1=e1, u0=u0, u1=u1, m0=m0, m1=m1, maxbpn=mb)

    def cost_golomb(self, buf, state=0.0):
        buf = np.ascontiguousarray(buf, dtype=np.int32)
        st = C.c_double(state)
        r = getattr(self.lib, self.pre + "cost_golomb")(buf, len(buf), C.byref(st))
        return r, st.value

    def cost_meanrms(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.int32)
        return getattr(self.lib, self.pre + "cost_meanrms")(buf, len(buf))

    def rsqrt(self, x):
        return getattr(self.lib, self.pre + "rsqrt")(C.c_float(x))

    def lms_pair(self, S, D, mu1, mu2, x0, x1):
        x0 = np.ascontiguousarray(x0, dtype=np.float64); x1 = np.ascontiguousarray(x1, dtype=np.float64)
        n = len(x0)
        p0 = np.zeros(n); p1 = np.zeros(n)
        getattr(self.lib, self.pre + "lms_pair")(C.c_int(S), C.c_int(D), C.c_double(mu1), C.c_double(mu2),
                                                 _ptr(x0), _ptr(x1), C.c_int(n), _ptr(p0), _ptr(p1))
        return p0, p1

    def lpc3_pair(self, alpha, s0, s1, S=16, D=8):
        s0 = np.ascontiguousarray(s0, dtype=np.int32); s1 = np.ascontiguousarray(s1, dtype=np.int32)
        n = len(s0)
        p0 = np.zeros(n); p1 = np.zeros(n)
        getattr(self.lib, self.pre + "lpc3_pair")(C.c_double(alpha), C.c_int(S), C.c_int(D), _ptr(s0), _ptr(s1), C.c_int(n), _ptr(p0), _ptr(p1))
        return p0, p1

    def rls_run(self, n, alpha, pv, val):
        pv = np.ascontiguousarray(pv, dtype=np.float64); val = np.ascontiguousarray(val, dtype=np.float64)
        ln = len(val)
        p = np.zeros(ln)
        getattr(self.lib, self.pre + "rls_run")(C.c_int(n), C.c_double(alpha), _ptr(pv), _ptr(val), C.c_int(ln), _ptr(p))
        return p

    def cascade_trace(self, profile, s0, s1, coefs=None):
        s0 = np.ascontiguousarray(s0, dtype=np.int32); s1 = np.ascontiguousarray(s1, dtype=np.int32)
        n = len(s0)
        nst = 5 if profile >= 2 else 4
        p = np.zeros((n, 2, nst)); v = np.zeros((n, 2, nst))
        e0 = np.zeros(n, np.int32); e1 = np.zeros(n, np.int32)
        cf = None if coefs is None else np.ascontiguousarray(coefs, dtype=np.float64)
        f = getattr(self.lib, self.pre + "cascade_trace")
        if self.pre == "ref_":
            f(C.c_int(profile), _ptr(cf), C.c_int(0 if cf is None else len(cf)), _ptr(s0), _ptr(s1), C.c_int(n),
              _ptr(p), _ptr(v), _ptr(e0), _ptr(e1))
        else:
            f(C.c_int(profile), _ptr(cf), _ptr(s0), _ptr(s1), C.c_int(n), _ptr(p), _ptr(v), _ptr(e0), _ptr(e1))
        return p, v, e0, e1

    def unpredict(self, profile, e0, e1, coefs=None, used0=None, used1=None, minmax=None):
        """UnpredictStereoFrame: residuals (mapped where used* is given) -> samples."""
        e0 = np.ascontiguousarray(e0, dtype=np.int32); e1 = np.ascontiguousarray(e1, dtype=np.int32)
        n = len(e0)
        v0 = np.zeros(n, np.int32); v1 = np.zeros(n, np.int32)
        cf = None if coefs is None else np.ascontiguousarray(coefs, dtype=np.float64)
        u0 = None if used0 is None else np.ascontiguousarray(used0, dtype=np.uint8)
        u1 = None if used1 is None else np.ascontiguousarray(used1, dtype=np.uint8)
        mm = None if minmax is None else np.ascontiguousarray(minmax, dtype=np.int32)
        f = getattr(self.lib, self.pre + "unpredict_stereo_frame")
        f.restype = None if self.pre == "ref_" else C.c_int
        if self.pre == "ref_":
            f(C.c_int(profile), _ptr(cf), C.c_int(0 if cf is None else len(cf)), _ptr(e0), _ptr(e1), C.c_int(n),
              _ptr(u0), _ptr(u1), _ptr(mm), _ptr(v0), _ptr(v1))
        else:
            f(C.c_int(profile), _ptr(cf), _ptr(e0), _ptr(e1), C.c_int(n), _ptr(u0), _ptr(u1), _ptr(mm), _ptr(v0), _ptr(v1))
        return v0, v1

    def chol_solve(self, A, b, regp, fast=False):
        """(sol, failed) with the packing of sacopt_chol_solve."""
        A = np.asarray(A, np.float64); n = len(b)
        snap = np.ascontiguousarray(np.concatenate([np.concatenate([A[r, :r + 1] for r in range(n)]), np.asarray(b, np.float64)]))
        sol = np.zeros(n if fast else n * (n + 1) // 2)
        f = getattr(self.lib, self.pre + "chol_solve")
        f.restype = C.c_int
        rc = f(C.c_int(n), C.c_double(regp), C.c_int(int(fast)), _ptr(snap), _ptr(sol))
        return sol, bool(rc)

    def remap_map(self, src, pred, err):
        src = np.ascontiguousarray(src, dtype=np.int32)
        pred = np.ascontiguousarray(pred, dtype=np.int32); err = np.ascontiguousarray(err, dtype=np.int32)
        out = np.zeros(len(pred), np.int32)
        getattr(self.lib, self.pre + "remap_map")(_ptr(src), C.c_int(len(src)), _ptr(pred), _ptr(err),
                                                  C.c_int(len(pred)), _ptr(out))
        return out


class Oracle(_Lib):
    def __init__(self):
        super().__init__(build_oracle(), "oracle_")
        self.lib.oracle_rsqrt_native.restype = C.c_float
        self.lib.oracle_rsqrt_native.argtypes = [C.c_float]
        self._tab = None

    def set_rsqrt_table(self, tab, bits):
        """tab=None -> native rsqrtss of this CPU."""
        if tab is None:
            self._tab = None
            self.lib.oracle_set_rsqrt_table(None, 0)
        else:
            self._tab = np.ascontiguousarray(tab, dtype=np.float32)
            assert len(self._tab) == 2 << bits
            self.lib.oracle_set_rsqrt_table(_ptr(self._tab), C.c_int(bits))

    def probe_rsqrt(self, cap=1 << 16):
        tab = np.zeros(cap, np.float32)
        bits = self.lib.oracle_probe_rsqrt(_ptr(tab), C.c_int(cap))
        if bits < 0:
            return None, -1
        return tab[:2 << bits].copy(), bits

    def cholesky_kat(self):
        return bool(self.lib.oracle_cholesky_kat())

    def remap_map_literal(self, src, pred, err):
        """Remap::Map as the loop of map.cpp:141-152, every step taken (slow for large |err|)."""
        src = np.ascontiguousarray(src, dtype=np.int32)
        pred = np.ascontiguousarray(pred, dtype=np.int32); err = np.ascontiguousarray(err, dtype=np.int32)
        out = np.zeros(len(pred), np.int32)
        self.lib.oracle_remap_map_literal(_ptr(src), C.c_int(len(src)), _ptr(pred), _ptr(err), C.c_int(len(pred)), _ptr(out))
        return out

    def remap_tables(self, src):
        """Remap::Analyse of a frame as the decoder's tables: usedl[0..32768] then usedh[0..32768] (bytes)."""
        src = np.ascontiguousarray(src, dtype=np.int32)
        used = np.zeros(2 * 32769, np.uint8)
        self.lib.oracle_remap_tables(_ptr(src), C.c_int(len(src)), _ptr(used))
        return used

    def encode_channel(self, s2u, maxbpn, s2u_map=None, maxbpn_map=0, used=None, want_bytes=True):
        """FrameCoder::EncodeMonoFrame (codec.cpp:268-280) restated: both coders, the smaller kept ("normal" wins only when
        strictly smaller). Returns (payload bytes | size, enc_mapped, flag bits); without map inputs only the plain coder runs."""
        s2u = np.ascontiguousarray(s2u, dtype=np.int32)
        n = len(s2u)
        cap = 16 * n + 70000
        o1 = np.zeros(cap if want_bytes else 1, np.uint8)
        n1 = self.lib.oracle_encode_mono_normal(_ptr(s2u), C.c_int(n), C.c_int(int(maxbpn)), _ptr(o1) if want_bytes else None, C.c_int(cap if want_bytes else 0))
        if s2u_map is None:
            return (o1[:n1].tobytes() if want_bytes else n1), False, int(maxbpn)
        sm = np.ascontiguousarray(s2u_map, dtype=np.int32); us = np.ascontiguousarray(used, dtype=np.uint8)
        o2 = np.zeros(cap if want_bytes else 1, np.uint8)
        n2 = self.lib.oracle_encode_mono_mapped(_ptr(sm), C.c_int(n), C.c_int(int(maxbpn_map)), _ptr(us), _ptr(o2) if want_bytes else None, C.c_int(cap if want_bytes else 0))
        if n1 < n2:
            return (o1[:n1].tobytes() if want_bytes else n1), False, int(maxbpn)
        return (o2[:n2].tobytes() if want_bytes else n2), True, int(maxbpn_map)

    def optimize(self, profile, s0, s1, prof=None, ncoef_loop=8, max_rows=128, cost_fn=0):
        """cost_fn: 0 CostGolomb, 1 CostMeanRMS, 2 bit-plane coder size."""
        s0 = np.ascontiguousarray(s0, dtype=np.int32); s1 = np.ascontiguousarray(s1, dtype=np.int32)
        best = np.zeros(8); trace = np.zeros((max_rows, 3))
        pf = None if prof is None else np.ascontiguousarray(prof, dtype=np.float64)
        rows = self.lib.oracle_optimize_cf(C.c_int(profile), _ptr(pf), C.c_int(ncoef_loop), _ptr(s0), _ptr(s1),
                                           C.c_int(len(s0)), C.c_int(cost_fn), _ptr(best), _ptr(trace), C.c_int(max_rows))
        return best, trace[:rows].copy()


class Ref(_Lib):
    def __init__(self):
        super().__init__(_REF_SO, "ref_")

    def cholesky_kat(self):
        return bool(self.lib.ref_cholesky_test())

    def optimize(self, profile, s0, s1):
        s0 = np.ascontiguousarray(s0, dtype=np.int32); s1 = np.ascontiguousarray(s1, dtype=np.int32)
        best = np.zeros(8)
        n = self.lib.ref_optimize(C.c_int(profile), _ptr(s0), _ptr(s1), C.c_int(len(s0)), _ptr(best))
        return best[:max(n, 0)], n

    def optimize_trace(self, profile, s0, s1, ncoef_loop=8, max_rows=128, meanrms=False):
        s0 = np.ascontiguousarray(s0, dtype=np.int32); s1 = np.ascontiguousarray(s1, dtype=np.int32)
        best = np.zeros(8); trace = np.zeros((max_rows, 3))
        rows = (self.lib.ref_optimize_trace_meanrms if meanrms else self.lib.ref_optimize_trace)(C.c_int(profile), _ptr(s0), _ptr(s1), C.c_int(len(s0)),
                                           C.c_int(ncoef_loop), _ptr(best), _ptr(trace), C.c_int(max_rows))
        return best, trace[:rows].copy()


def _ref_encode(self, fn, *args, total):
    cap = 16 * total + 4096
    out = np.zeros(cap, np.uint8)
    n = fn(*args, _ptr(out), C.c_int(cap))
    assert n > 0
    return out[:n].tobytes()


def ref_encode_frame(ref, profile, s0, s1):
    """Frame record bytes (FrameCoder::WriteEncoded layout) of the reference's own Predict()+Encode()."""
    s0 = np.ascontiguousarray(s0, dtype=np.int32); s1 = np.ascontiguousarray(s1, dtype=np.int32)
    return _ref_encode(ref, ref.lib.ref_encode_frame, C.c_int(profile), _ptr(s0), _ptr(s1), C.c_int(len(s0)), total=len(s0))


def ref_encode_residuals(ref, profile, s0, s1, res):
    """Frame record bytes of the reference's Encode() fed with residuals computed elsewhere (dict u0,u1,m0,m1,maxbpn)."""
    s0 = np.ascontiguousarray(s0, dtype=np.int32); s1 = np.ascontiguousarray(s1, dtype=np.int32)
    a = [np.ascontiguousarray(res[k], dtype=np.int32) for k in ("u0", "u1", "m0", "m1", "maxbpn")]
    return _ref_encode(ref, ref.lib.ref_encode_residuals, C.c_int(profile), _ptr(s0), _ptr(s1), C.c_int(len(s0)),
                       *[_ptr(x) for x in a], total=len(s0))


def ref_encode_channels(ref, profile, s0, s1, res):
    """The reference's Encode() (both coders per channel, smaller kept: codec.cpp:268-280) fed with residuals computed
    elsewhere: [(payload bytes, enc_mapped, flag bits)] x 2 -- what a host-side container writer packs."""
    s0 = np.ascontiguousarray(s0, dtype=np.int32); s1 = np.ascontiguousarray(s1, dtype=np.int32)
    a = [np.ascontiguousarray(res[k], dtype=np.int32) for k in ("u0", "u1", "m0", "m1", "maxbpn")]
    cap = 16 * len(s0) + 4096
    out = np.zeros(cap, np.uint8)
    info = np.zeros(6, np.int32)
    n = ref.lib.ref_encode_channels(C.c_int(profile), _ptr(s0), _ptr(s1), C.c_int(len(s0)), *[_ptr(x) for x in a], _ptr(out),
                                    C.c_int(cap), _ptr(info))
    assert n >= 0
    l0, l1 = int(info[0]), int(info[3])
    return [(out[:l0].tobytes(), bool(info[1]), int(info[2])), (out[l0:l0 + l1].tobytes(), bool(info[4]), int(info[5]))]


def ref_decode_channels(ref, profile, n, ch0, ch1):
    """The reference's Decode(): payloads -> (e0, e1, used0 | None, used1 | None)."""
    b = [np.frombuffer(c[0], np.uint8) if len(c[0]) else np.zeros(1, np.uint8) for c in (ch0, ch1)]
    e0 = np.zeros(n, np.int32); e1 = np.zeros(n, np.int32)
    used = [np.zeros(2 * 32769, np.uint8) for _ in range(2)]
    ref.lib.ref_decode_channels(C.c_int(profile), C.c_int(n), _ptr(b[0]), C.c_int(len(ch0[0])), C.c_int(int(ch0[1])), C.c_int(ch0[2]),
                                _ptr(b[1]), C.c_int(len(ch1[0])), C.c_int(int(ch1[1])), C.c_int(ch1[2]), _ptr(e0), _ptr(e1),
                                _ptr(used[0]), _ptr(used[1]))
    return e0, e1, (used[0] if ch0[1] else None), (used[1] if ch1[1] else None)


def out_of_plane_buffers():
    """s2u buffers with bits ABOVE the coded planes: s2u of a residual beyond 2^30 wraps negative and MaxBPN's signed maximum
    (codec.cpp:178-211) skips it -- a diverged filter does that (found by tools/fuzz_parity.py). The coder then codes planes
    maxbpn..0 of the two's-complement bits, msb[] only ever sees the coded planes, the context comparisons shift the signed
    value (vle.cpp:32-60). [(buffer, maxbpn)]"""
    rs = np.random.RandomState(4)
    n = 700
    out = []
    for mb, mk in ((8, lambda: rs.randint(0, 1 << 13, n)),
                   (12, lambda: np.where(rs.rand(n) < 0.05, -rs.randint(1, 1 << 30, n), rs.randint(0, 1 << 13, n))),
                   (28, lambda: np.where(rs.rand(n) < 0.02, rs.randint(-2 ** 31, 2 ** 31 - 1, n), rs.randint(0, 1 << 20, n)))):
        for _ in range(2):
            out.append((mk().astype(np.int32), mb))
    return out


def diverging_impulse_clip():
    """(L, R, coefs) of a case tools/fuzz_parity.py found: sparse full-scale impulses on silence with these (in-range) high
    profile coefficients drive the cascade's predictions out of int32 -- cvttsd2si gives INT32_MIN, err = val - pred wraps
    (residuals of 3e8 .. 2^31), s2u wraps negative, and Remap::Map's walk runs over the end of int32. The reference's
    behaviour is its build's wrapping arithmetic there; oracle/Makefile says -fwrapv."""
    n = 1759
    L = np.zeros(n, np.int32); R = np.zeros(n, np.int32)
    L[[47, 49, 115, 234, 244, 376, 456, 503, 543, 716, 933, 949, 1055, 1069, 1277, 1407, 1651, 1730]] = [
        -5941, 30226, 10683, -25412, 31270, 3464, -24881, -5348, -27855, 16359, -11687, -2417, -26635, -18838, -952, -15212, 18390, -11699]
    R[[155, 266, 304, 333, 524, 849, 880, 944, 947, 1058, 1100, 1101, 1141, 1225, 1394, 1408, 1561, 1650]] = [
        -13296, 6951, 23511, -18677, 20849, 12254, 20489, 15286, -4310, 32742, 21172, 5557, 25609, 13921, -22165, -30665, -23917, -21356]
    coefs = np.array([0.9963248028931021, 1e-05, 5e-06, 0.0008946247490625482, 0.0003390631856487167, 0.0018940434635736038, 0.01,
                      0.9936731708629059])
    return L, R, coefs


def map_wrap_cases():
    """(src, pred, err) for Remap::Map where `-err` or `pred + sgn * i` leave int32 (and two ordinary ones)."""
    rs = np.random.RandomState(2)
    src = (rs.randint(-32768, 32768, 3000) // 7 * 7).astype(np.int32)
    lo, hi = -2 ** 31, 2 ** 31 - 1
    cases = [(lo, lo), (lo, lo + 5), (hi - 10, hi - 5000), (hi - 100, lo + 2100), (lo, 5), (hi, -7), (-40000, 90000), (300, -70000)]
    return src, np.array([c[0] for c in cases], np.int32), np.array([c[1] for c in cases], np.int32)


def have_ref():
    return os.path.exists(_REF_SO)
