"""ctypes binding of libsacopt.so (include/sacopt.h) -- plumbing for tests and bench.py.

The library is the product; this module only marshals numpy arrays / raw device pointers into
the C ABI. There is no fallback: if libsacopt.so is missing or no gfx950 GPU is usable, loading
or `Context()` raises.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SACOPT_LIB") or os.path.join(_HERE, "libsacopt.so")  # SACOPT_LIB: instrumented builds of tools/ only

PROFILE_FAST, PROFILE_NORMAL, PROFILE_HIGH, PROFILE_HIGH32 = 0, 1, 2, 3
COST_GOLOMB, COST_MEANRMS, COST_BITPLANE = 0, 1, 2
NCOEF = {0: 1, 1: 6, 2: 8, 3: 8}


class Frame(C.Structure):
    _fields_ = [("s0", C.c_void_p), ("s1", C.c_void_p), ("numsamples", C.c_int32)]


class Residuals(C.Structure):
    _fields_ = [("err0", C.c_void_p), ("err1", C.c_void_p), ("s2u0", C.c_void_p), ("s2u1", C.c_void_p),
                ("map0", C.c_void_p), ("map1", C.c_void_p), ("maxbpn", C.c_int32 * 4)]


class DecFrame(C.Structure):
    _fields_ = [("e0", C.c_void_p), ("e1", C.c_void_p), ("v0", C.c_void_p), ("v1", C.c_void_p), ("numsamples", C.c_int32),
                ("used0", C.c_void_p), ("used1", C.c_void_p), ("minval", C.c_int32 * 2), ("maxval", C.c_int32 * 2)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `make -C sac_amd/csrc` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        L.sacopt_last_error.restype = C.c_char_p
        L.sacopt_last_error.argtypes = [C.c_void_p]
        L.sacopt_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.sacopt_destroy.argtypes = [C.c_void_p]
        L.sacopt_destroy.restype = None
        L.sacopt_probe_rsqrt.argtypes = [C.c_void_p, C.c_int]
        L.sacopt_base_profile.argtypes = [C.c_int, C.c_void_p]
        L.sacopt_predict_stereo_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p]
        L.sacopt_get_costs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
        L.sacopt_set_cost_function.argtypes = [C.c_void_p, C.c_int]
        L.sacopt_recost_last.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.sacopt_optimize_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.sacopt_encode_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.sacopt_encode_frames_selected.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.sacopt_unpredict_stereo_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.sacopt_cost_golomb.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.sacopt_coded_sizes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sacopt_encode_payloads.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sacopt_cost_golomb_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.sacopt_cost_meanrms.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.sacopt_lms_pair.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_void_p, C.c_void_p]
        L.sacopt_lpc3_pair.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.sacopt_rls_run.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.sacopt_chol_solve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sacopt_serial_sum.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.sacopt_cascade_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sacopt_last_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


EXPORTS = ["sacopt_create", "sacopt_destroy", "sacopt_last_error", "sacopt_probe_rsqrt", "sacopt_base_profile",
           "sacopt_predict_stereo_frames", "sacopt_set_cost_function", "sacopt_get_costs", "sacopt_recost_last", "sacopt_optimize_frames", "sacopt_encode_frames", "sacopt_encode_frames_selected", "sacopt_unpredict_stereo_frames", "sacopt_cost_golomb", "sacopt_cost_golomb_noalloc", "sacopt_coded_sizes", "sacopt_encode_payloads",
           "sacopt_cost_meanrms", "sacopt_lms_pair", "sacopt_lpc3_pair", "sacopt_rls_run", "sacopt_chol_solve", "sacopt_serial_sum",
           "sacopt_cascade_trace", "sacopt_last_timing"]


def hip_runtimes_mapped():
    """Paths of the libamdhip64 files this process has mapped."""
    out = set()
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                if "libamdhip64" in ln:
                    out.add(ln.split()[-1])
    except OSError:
        pass
    return sorted(out)


def assert_single_hip_runtime():
    """torch bundles its own libamdhip64 (needed as `libamdhip64.so`), libsacopt.so needs `libamdhip64.so.7`. The dynamic loader
    gives libsacopt the runtime torch has already loaded (its soname matches) -- ONE runtime, everything works. The other
    way round (libsacopt first, then torch) the process ends up with two runtimes and torch's finds no GPU. So, in a process
    that uses both: initialise torch.cuda BEFORE the first sacopt.lib() / Context(). This raises if two runtimes are mapped."""
    r = hip_runtimes_mapped()
    if len(r) > 1:
        raise RuntimeError("two HIP runtimes in this process (%s): import torch and touch torch.cuda BEFORE sac_amd.sacopt "
                           "loads libsacopt.so" % ", ".join(r))


def _p(a):
    """numpy array -> host pointer; int -> raw (device) pointer; None -> NULL."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    return a.ctypes.data_as(C.c_void_p)


def probe_rsqrt(cap=8192):
    tab = np.zeros(cap, np.float32)
    bits = lib().sacopt_probe_rsqrt(_p(tab), cap)
    if bits < 0:
        raise RuntimeError("host rsqrtss does not fit the table model")
    return tab[:2 << bits].copy(), bits


def base_profile(profile):
    out = np.zeros(24)
    n = lib().sacopt_base_profile(profile, _p(out))
    return out[:3 * n].reshape(n, 3).copy()


class SacoptError(RuntimeError):
    pass


_live = weakref.WeakSet()   # contexts not yet closed
_finalizing = False


@atexit.register
def _close_live_contexts():
    """A context owns CU-masked streams, events and device memory: it has to go while the HIP runtime (and a profiler
    tool attached to it) is still alive, i.e. before interpreter teardown -- a sacopt_destroy from __del__ during
    finalisation ran after the runtime's own exit handlers and crashed in __cxa_finalize under rocprofv3."""
    global _finalizing
    for c in list(_live):
        c.close()
    _finalizing = True


class Context:
    """One per GPU. rsqrt_tab=None probes the CPU this process runs on. Use as a context manager or call close()."""

    def __init__(self, device=0, rsqrt_tab=None, rsqrt_bits=0):
        self._h = C.c_void_p()
        tab = None if rsqrt_tab is None else np.ascontiguousarray(rsqrt_tab, dtype=np.float32)
        rc = lib().sacopt_create(device, _p(tab), rsqrt_bits, C.byref(self._h))
        if rc:
            raise SacoptError(lib().sacopt_last_error(None).decode())
        _live.add(self)

    def close(self):
        if self._h and not _finalizing:
            lib().sacopt_destroy(self._h)
        self._h = C.c_void_p()
        _live.discard(self)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        # once interpreter finalisation has begun the handle is abandoned (the process is going away)
        try:
            if not _finalizing:
                self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise SacoptError(lib().sacopt_last_error(self._h).decode())

    @staticmethod
    def _frames(frames):
        """frames: list of (s0, s1[, numsamples]); s* numpy int32 arrays or raw device pointers."""
        arr = (Frame * max(len(frames), 1))()
        keep = []
        for i, fr in enumerate(frames):
            s0, s1 = fr[0], fr[1]
            if isinstance(s0, np.ndarray):
                s0 = np.ascontiguousarray(s0, dtype=np.int32)
                s1 = np.ascontiguousarray(s1, dtype=np.int32)
                n = len(s0) if len(fr) < 3 else fr[2]
                keep += [s0, s1]
                arr[i].s0, arr[i].s1 = s0.ctypes.data, s1.ctypes.data
            else:
                n = fr[2]
                arr[i].s0, arr[i].s1 = int(s0), int(s1)
            arr[i].numsamples = n
        return arr, keep

    @staticmethod
    def _residuals(res, i, ni, device_out, maps):
        if device_out is None:
            keys = ("e0", "e1", "u0", "u1") + (("m0", "m1") if maps else ())
            d = {k: np.zeros(max(ni, 0), np.int32) for k in keys}
            res[i].err0, res[i].err1 = d["e0"].ctypes.data, d["e1"].ctypes.data
            res[i].s2u0, res[i].s2u1 = d["u0"].ctypes.data, d["u1"].ctypes.data
            if maps:
                res[i].map0, res[i].map1 = d["m0"].ctypes.data, d["m1"].ctypes.data
        else:
            d = {}
            res[i].err0, res[i].err1 = device_out[i].get("e0"), device_out[i].get("e1")
            res[i].s2u0, res[i].s2u1 = device_out[i].get("u0"), device_out[i].get("u1")
            res[i].map0, res[i].map1 = device_out[i].get("m0"), device_out[i].get("m1")
        return d

    def predict(self, profile, frames, frm=None, n=None, coefs=None, device_out=None, maps=False):
        """Returns list of dicts (e0,e1,u0,u1[,m0,m1],maxbpn) as numpy arrays; with device_out (list of dicts
        of raw device pointers) the residuals are left on the device. maps=True also asks for the
        Remap::Map outputs (s2u_error_map, maxbpn_map)."""
        nf = len(frames)
        arr, keep = self._frames(frames)
        frm_a = None if frm is None else np.ascontiguousarray(frm, dtype=np.int32)
        n_a = None if n is None else np.ascontiguousarray(n, dtype=np.int32)
        cf = None if coefs is None else np.ascontiguousarray(coefs, dtype=np.float64)
        res = (Residuals * max(nf, 1))()
        outs = []
        for i in range(nf):
            ni = int(n_a[i]) if n_a is not None else arr[i].numsamples - (int(frm_a[i]) if frm_a is not None else 0)
            outs.append(self._residuals(res, i, ni, device_out, maps))
        self._ck(lib().sacopt_predict_stereo_frames(self._h, profile, nf, arr, _p(frm_a), _p(n_a), _p(cf), res))
        for i in range(nf):
            outs[i]["maxbpn"] = np.array(list(res[i].maxbpn), np.int32)
        return outs

    def get_costs(self, profile, frames, cand_coefs, frm=None, n=None, state=None):
        """cand_coefs: [nframes, ncand, ncoef]. Returns (costs[nframes,ncand], state[nframes])."""
        nf = len(frames)
        arr, keep = self._frames(frames)
        cc = np.ascontiguousarray(cand_coefs, dtype=np.float64)
        ncand = cc.shape[1]
        frm_a = None if frm is None else np.ascontiguousarray(frm, dtype=np.int32)
        n_a = None if n is None else np.ascontiguousarray(n, dtype=np.int32)
        costs = np.zeros((nf, ncand))
        st = np.zeros(nf) if state is None else np.ascontiguousarray(state, dtype=np.float64).copy()
        self._ck(lib().sacopt_get_costs(self._h, profile, nf, arr, _p(frm_a), _p(n_a), ncand, _p(cc), _p(costs), _p(st)))
        return costs, st

    def set_cost_function(self, which):
        """0 CostGolomb (the reference's choice), 1 CostMeanRMS, 2 bit-plane coder size."""
        self._ck(lib().sacopt_set_cost_function(self._h, which))

    def recost_last(self, nframes, ncand, state=None):
        """sacopt_recost_last: the candidates of the last get_costs call from another start state."""
        costs = np.zeros((nframes, ncand))
        st = np.zeros(nframes) if state is None else np.ascontiguousarray(state, dtype=np.float64).copy()
        self._ck(lib().sacopt_recost_last(self._h, _p(costs), _p(st)))
        return costs, st

    def optimize(self, profile, frames, prof=None, ncoef_loop=None, max_rows=128):
        """FrameCoder::Optimize for every frame. Returns (best[nframes,ncoef], trace list, rows)."""
        nf = len(frames)
        arr, keep = self._frames(frames)
        nc = NCOEF[profile]
        if ncoef_loop is None:
            ncoef_loop = nc
        pf = None if prof is None else np.ascontiguousarray(prof, dtype=np.float64)
        best = np.zeros((nf, nc))
        trace = np.zeros((nf, max_rows, 3))
        rows = np.zeros(nf, np.int32)
        self._ck(lib().sacopt_optimize_frames(self._h, profile, nf, arr, _p(pf), ncoef_loop, _p(best), _p(trace),
                                              max_rows, _p(rows)))
        return best, [trace[f, :min(rows[f], max_rows)].copy() for f in range(nf)], rows

    def encode(self, profile, frames, prof=None, ncoef_loop=None, device_out=None, maps=False, selected=False):
        """FrameCoder::Predict with Optimize enabled: returns (best[nframes,ncoef], residual dicts). selected: the residuals are
        those of PredictStereoFrame with the SELECTED parameters (sacopt_encode_frames_selected)."""
        nf = len(frames)
        arr, keep = self._frames(frames)
        nc = NCOEF[profile]
        if ncoef_loop is None:
            ncoef_loop = nc
        pf = None if prof is None else np.ascontiguousarray(prof, dtype=np.float64)
        best = np.zeros((nf, nc))
        res = (Residuals * max(nf, 1))()
        outs = []
        for i in range(nf):
            outs.append(self._residuals(res, i, arr[i].numsamples, device_out, maps))
        fn = lib().sacopt_encode_frames_selected if selected else lib().sacopt_encode_frames
        self._ck(fn(self._h, profile, nf, arr, _p(pf), ncoef_loop, _p(best), res))
        for i in range(nf):
            outs[i]["maxbpn"] = np.array(list(res[i].maxbpn), np.int32)
        return best, outs

    def unpredict(self, profile, frames, coefs=None):
        """frames: list of dicts e0, e1 (numpy int32) [, used0, used1 (uint8[2*32769] or None), minmax (min0,max0,min1,max1)].
        Returns list of (v0, v1). Device form: e0, e1, v0, v1 raw device pointers (int) and n; the samples stay on the
        device and the list holds (v0, v1) pointers."""
        nf = len(frames)
        arr = (DecFrame * max(nf, 1))()
        keep, outs = [], []
        for i, fr in enumerate(frames):
            if isinstance(fr["e0"], (int, np.integer)):
                n = int(fr["n"])
                e0p, e1p, v0, v1 = int(fr["e0"]), int(fr["e1"]), int(fr["v0"]), int(fr["v1"])
                v0p, v1p = v0, v1
            else:
                e0 = np.ascontiguousarray(fr["e0"], dtype=np.int32); e1 = np.ascontiguousarray(fr["e1"], dtype=np.int32)
                n = len(e0)
                v0 = np.zeros(n, np.int32); v1 = np.zeros(n, np.int32)
                keep += [e0, e1]
                e0p, e1p, v0p, v1p = e0.ctypes.data, e1.ctypes.data, v0.ctypes.data, v1.ctypes.data
            u = [None if fr.get(k) is None else np.ascontiguousarray(fr[k], dtype=np.uint8) for k in ("used0", "used1")]
            keep += u
            outs.append((v0, v1))
            arr[i].e0, arr[i].e1, arr[i].v0, arr[i].v1 = e0p, e1p, v0p, v1p
            arr[i].numsamples = n
            arr[i].used0 = None if u[0] is None else u[0].ctypes.data
            arr[i].used1 = None if u[1] is None else u[1].ctypes.data
            mm = fr.get("minmax")
            if mm is None:
                mm = (-2 ** 31, 2 ** 31 - 1, -2 ** 31, 2 ** 31 - 1)
            arr[i].minval[0], arr[i].maxval[0], arr[i].minval[1], arr[i].maxval[1] = [int(x) for x in mm]
        cf = None if coefs is None else np.ascontiguousarray(coefs, dtype=np.float64)
        self._ck(lib().sacopt_unpredict_stereo_frames(self._h, profile, nf, arr, _p(cf)))
        return outs

    def cost_golomb(self, buf, state=0.0):
        st = np.array([state], np.float64)
        cost = np.zeros(1)
        if isinstance(buf, np.ndarray):
            buf = np.ascontiguousarray(buf, dtype=np.int32)
            n = len(buf)
        else:
            buf, n = buf
        self._ck(lib().sacopt_cost_golomb(self._h, _p(buf), n, _p(st), _p(cost)))
        return float(cost[0]), float(st[0])

    def cost_golomb_noalloc(self, buf, state=0.0):
        """sacopt_cost_golomb_noalloc: same contract as cost_golomb, no allocation inside the library.
        buf: numpy int32 array, or (raw device pointer, n)."""
        st = np.array([state], np.float64)
        cost = np.zeros(1)
        if isinstance(buf, np.ndarray):
            buf = np.ascontiguousarray(buf, dtype=np.int32)
            n = len(buf)
        else:
            buf, n = buf
        self._ck(lib().sacopt_cost_golomb_noalloc(self._h, _p(buf), n, _p(st), _p(cost)))
        return float(cost[0]), float(st[0])

    def coded_sizes(self, bufs, maxbpn):
        """sacopt_coded_sizes: bufs = list of numpy int32 arrays (s2u residuals) or (device pointer, n) tuples."""
        nb = len(bufs)
        keep, ptrs, ns = [], (C.c_void_p * max(nb, 1))(), np.zeros(nb, np.int32)
        for i, b in enumerate(bufs):
            if isinstance(b, np.ndarray):
                a = np.ascontiguousarray(b, dtype=np.int32); keep.append(a)
                ptrs[i] = a.ctypes.data if len(a) else None; ns[i] = len(a)
            else:
                ptrs[i] = int(b[0]); ns[i] = int(b[1])
        mb = np.ascontiguousarray(maxbpn, dtype=np.int32)
        out = np.zeros(nb, np.int32)
        self._ck(lib().sacopt_coded_sizes(self._h, nb, ptrs, _p(ns), _p(mb), _p(out)))
        return out

    def encode_payloads(self, bufs, maxbpn, used=None, cap=None):
        """sacopt_encode_payloads: the payload bytes of FrameCoder::EncodeMonoFrame_Normal (used[b] None) / _Mapped (used[b] =
        uint8[2*32769] Remap tables) for s2u buffers (numpy int32 arrays or (device pointer, n) tuples). Returns a list of bytes objects."""
        nb = len(bufs)
        keep, ptrs, ns = [], (C.c_void_p * max(nb, 1))(), np.zeros(nb, np.int32)
        uptr, optr = (C.c_void_p * max(nb, 1))(), (C.c_void_p * max(nb, 1))()
        caps = np.zeros(nb, np.int32)
        outs = []
        for i, b in enumerate(bufs):
            if isinstance(b, np.ndarray):
                a = np.ascontiguousarray(b, dtype=np.int32); keep.append(a)
                ptrs[i] = a.ctypes.data if len(a) else None; ns[i] = len(a)
            else:   # (device pointer, n)
                ptrs[i] = int(b[0]); ns[i] = int(b[1])
            u = None if used is None or used[i] is None else np.ascontiguousarray(used[i], dtype=np.uint8)
            if u is not None:
                assert len(u) == 2 * 32769
                keep.append(u)
            uptr[i] = None if u is None else u.ctypes.data
            caps[i] = 4 * int(ns[i]) + 8192 + 64 if cap is None else cap   # (cap: tests of the too-small-buffer path)
            o = np.zeros(caps[i], np.uint8); outs.append(o)
            optr[i] = o.ctypes.data
        mb = np.ascontiguousarray(maxbpn, dtype=np.int32)
        sizes = np.zeros(nb, np.int32)
        self._ck(lib().sacopt_encode_payloads(self._h, nb, ptrs, _p(ns), _p(mb), uptr, optr, _p(caps), _p(sizes)))
        big = [i for i in range(nb) if sizes[i] > caps[i]]
        if big:   # incompressible data can need more than 4 bytes per sample: those buffers again, with the size the first call reported
            k = len(big)
            p2, u2, o2 = (C.c_void_p * k)(), (C.c_void_p * k)(), (C.c_void_p * k)()
            for j, i in enumerate(big):
                outs[i] = np.zeros(int(sizes[i]), np.uint8)
                p2[j], u2[j], o2[j] = ptrs[i], uptr[i], outs[i].ctypes.data
            n2 = np.ascontiguousarray(ns[big]); m2 = np.ascontiguousarray(mb[big]); c2 = np.ascontiguousarray(sizes[big]); s2 = np.zeros(k, np.int32)
            self._ck(lib().sacopt_encode_payloads(self._h, k, p2, _p(n2), _p(m2), u2, o2, _p(c2), _p(s2)))
            assert (s2 == c2).all()
        return [outs[i][:sizes[i]].tobytes() for i in range(nb)]

    def cost_meanrms(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.int32)
        cost = np.zeros(1)
        self._ck(lib().sacopt_cost_meanrms(self._h, _p(buf), len(buf), _p(cost)))
        return float(cost[0])

    def lms_pair(self, S, D, mu1, mu2, x0, x1):
        x0 = np.ascontiguousarray(x0, dtype=np.float64); x1 = np.ascontiguousarray(x1, dtype=np.float64)
        p0 = np.zeros(len(x0)); p1 = np.zeros(len(x0))
        self._ck(lib().sacopt_lms_pair(self._h, S, D, mu1, mu2, _p(x0), _p(x1), len(x0), _p(p0), _p(p1)))
        return p0, p1

    def lpc3_pair(self, alpha, s0, s1, S=16, D=8):
        s0 = np.ascontiguousarray(s0, dtype=np.int32); s1 = np.ascontiguousarray(s1, dtype=np.int32)
        p0 = np.zeros(len(s0)); p1 = np.zeros(len(s0))
        self._ck(lib().sacopt_lpc3_pair(self._h, alpha, S, D, _p(s0), _p(s1), len(s0), _p(p0), _p(p1)))
        return p0, p1

    def rls_run(self, nst, alpha, pv, val):
        pv = np.ascontiguousarray(pv, dtype=np.float64); val = np.ascontiguousarray(val, dtype=np.float64)
        p = np.zeros(len(val))
        self._ck(lib().sacopt_rls_run(self._h, nst, alpha, _p(pv), _p(val), len(val), _p(p)))
        return p

    def chol_solve(self, A, b, fast=False):
        """A: symmetric n x n (lower triangle used), b: n. Returns (sol, failed): sol = all-orders triangle (fast=False) or x."""
        A = np.asarray(A, np.float64); n = len(b)
        snap = np.concatenate([np.concatenate([A[r, :r + 1] for r in range(n)]), np.asarray(b, np.float64)])
        sol = np.zeros(n if fast else n * (n + 1) // 2)
        fail = C.c_int(0)
        self._ck(lib().sacopt_chol_solve(self._h, n, int(fast), _p(snap), _p(sol), C.byref(fail)))
        return sol, bool(fail.value)

    def serial_sum(self, x):
        """x: [count, nterms] -> strict left-to-right sums."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros(x.shape[0])
        self._ck(lib().sacopt_serial_sum(self._h, x.shape[1], _p(x), x.shape[0], _p(out)))
        return out

    def cascade_trace(self, profile, s0, s1, coefs=None):
        """Returns (p[n, 2, nst], v[n, 2, nst]) in the layout of the reference trace fixtures: p[t][ch][k] = stage prediction,
        v[t][ch][k] = stage input val[ch][k] (k = 0: the sample itself)."""
        arr, keep = self._frames([(s0, s1)])
        n = arr[0].numsamples
        nst = 5 if profile >= 2 else 4
        po = np.zeros((nst, 2, n)); vo = np.zeros((nst, 2, n))
        cf = None if coefs is None else np.ascontiguousarray(coefs, dtype=np.float64)
        self._ck(lib().sacopt_cascade_trace(self._h, profile, arr, _p(cf), _p(po), _p(vo)))
        p = np.ascontiguousarray(po.transpose(2, 1, 0))
        v = np.zeros((n, 2, nst))
        v[:, 0, 0], v[:, 1, 0] = keep[0], keep[1]
        v[:, :, 1:] = vo.transpose(2, 1, 0)[:, :, :nst - 1]
        return p, v

    def last_timing(self):
        t = C.c_double(); l = C.c_double(); k = C.c_int(); u = C.c_longlong()
        lib().sacopt_last_timing(self._h, C.byref(t), C.byref(l), C.byref(k), C.byref(u))
        return dict(total_ms=t.value, lms_ms=l.value, lms_launches=k.value, lms_units=u.value)
