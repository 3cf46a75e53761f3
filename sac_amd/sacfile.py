"""ctypes binding of libsachost.so (sac_amd/host/sacfile.h): the host-side .sac container -- header and frame records as
the reference writes them (src/file/sac.cpp:3-23, src/codec.cpp:480-522) plus the opt-in per-frame parameter extension
(SURVEY.md §8f.2). Pure host code; plumbing for tests and bench extras."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsachost.so")
PROFILE_PARAMS = 1 << 8
EXPORTS = ["sachost_canonical_wav_metadata", "sachost_write_header", "sachost_read_header", "sachost_write_frame",
           "sachost_read_frame"]
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `make -C sac_amd/csrc`")
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _u8(b):
    a = np.frombuffer(bytes(b), np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def canonical_wav_metadata(numch, rate, bits, numsamples):
    out = np.zeros(64, np.uint8)
    n = lib().sachost_canonical_wav_metadata(numch, rate, bits, numsamples, out.ctypes.data_as(C.c_void_p), 64)
    return out[:n].tobytes()


def write_header(numch, rate, bits, numsamples, profile, metadata):
    meta, mp = _u8(metadata)
    out = np.zeros(24 + len(meta), np.uint8)
    n = lib().sachost_write_header(numch, rate, bits, numsamples, profile, mp, len(meta), out.ctypes.data_as(C.c_void_p), len(out))
    assert n == len(out)
    return out.tobytes()


def read_header(blob):
    a, p = _u8(blob)
    f = (C.c_uint32 * 6)()
    used = lib().sachost_read_header(p, len(a), f)
    if not used:
        raise ValueError("not a SAC2 stream")
    return dict(numchannels=f[0], samplerate=f[1], bitspersample=f[2], numsamples=f[3], profile=f[4],
                metadata=a[24:24 + f[5]].tobytes()), used


def write_frame(numsamples, ch0, ch1, params=None):
    """ch*: (payload bytes, enc_mapped, maxbpn); params: None or (coefs, (min0, max0, min1, max1))."""
    (a0, p0), (a1, p1) = _u8(ch0[0]), _u8(ch1[0])
    chan = (C.c_int32 * 4)(int(ch0[1]), int(ch0[2]), int(ch1[1]), int(ch1[2]))
    cap = len(a0) + len(a1) + 128
    out = np.zeros(cap, np.uint8)
    if params is None:
        n = lib().sachost_write_frame(numsamples, p0, len(a0), p1, len(a1), chan, 0, 0, None, None, out.ctypes.data_as(C.c_void_p), cap)
    else:
        cf = np.ascontiguousarray(params[0], dtype=np.float64)
        mm = (C.c_int32 * 4)(*[int(x) for x in params[1]])
        n = lib().sachost_write_frame(numsamples, p0, len(a0), p1, len(a1), chan, 1, len(cf), cf.ctypes.data_as(C.c_void_p), mm,
                                      out.ctypes.data_as(C.c_void_p), cap)
    assert n > 0
    return out[:n].tobytes()


def read_frame(blob, pos, with_params=False):
    """Returns (record dict, bytes consumed): numsamples, ch = [(payload, enc_mapped, maxbpn)] x 2 [, coefs, minmax]."""
    a, _ = _u8(blob)
    sub = a[pos:]
    ns = C.c_uint32()
    chan = (C.c_int32 * 8)()
    nc = C.c_int32()
    cf = np.zeros(8)
    mm = (C.c_int32 * 4)()
    used = lib().sachost_read_frame(sub.ctypes.data_as(C.c_void_p), len(sub), int(with_params), C.byref(ns), chan, C.byref(nc),
                                    cf.ctypes.data_as(C.c_void_p), mm)
    if not used:
        raise ValueError("truncated frame record")
    ch = [(sub[chan[4 * c]:chan[4 * c] + chan[4 * c + 1]].tobytes(), bool(chan[4 * c + 2]), int(chan[4 * c + 3])) for c in range(2)]
    rec = dict(numsamples=ns.value, ch=ch)
    if with_params:
        rec["coefs"] = cf[:nc.value].copy()
        rec["minmax"] = tuple(mm)
    return rec, used
