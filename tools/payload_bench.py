"""sacopt_encode_payloads timing: both payload variants of both channels of NFR full frames of the bench clip (4 coders per frame),
in calls of 8, 60, 4*NFR buffers; and the sizes-only call beside it. usage: payload_bench.py [frames = 15]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
torch.cuda.init()
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
FRAME = 176400
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 15
L, R = stereo_clip(1000, nfr * FRAME)


def tables(x):   # Remap::Analyse (map.cpp:111-130)
    t = np.zeros(2 * 32769, np.uint8)
    t[32769 + x[(x > 0) & (x <= 32768)]] = 1
    t[-x[(x < 0) & (x >= -32768)]] = 1
    return t


with sacopt.Context(0) as ctx:
    frames = [(L[f * FRAME:(f + 1) * FRAME], R[f * FRAME:(f + 1) * FRAME]) for f in range(nfr)]
    res = ctx.predict(2, frames, maps=True)
    bufs, mb, used = [], [], []
    for (l, r_), r in zip(frames, res):
        for ch, x in enumerate((l, r_)):
            bufs += [r[f"u{ch}"], r[f"m{ch}"]]; mb += [int(r["maxbpn"][ch]), int(r["maxbpn"][2 + ch])]; used += [None, tables(x)]
    ctx.encode_payloads(bufs[:2], mb[:2], used[:2])
    tag = os.environ.get("SACOPT_LIB", "head")
    for nb in sorted({8, min(60, len(bufs)), len(bufs)}):
        t = time.perf_counter(); p = ctx.encode_payloads(bufs[:nb], mb[:nb], used[:nb]); dt = time.perf_counter() - t
        dec = sum(len(b) * (m + 1) for b, m in zip(bufs[:nb], mb[:nb])) + 65536 * sum(u is not None for u in used[:nb])
        print(f"{tag}: {nb:4d} payloads {dt * 1e3:7.1f} ms (device {ctx.last_timing()['total_ms']:7.1f} ms), {sum(len(x) for x in p)} bytes, longest chain {max(len(b) * (m + 1) for b, m in zip(bufs[:nb], mb[:nb])) / 1e6:.2f} M decisions, {dec / 1e6:.0f} M in all")
    t = time.perf_counter(); ctx.coded_sizes(bufs[:60], mb[:60]); print(f"{tag}:   sizes only, {min(60, len(bufs))} buffers: {(time.perf_counter() - t) * 1e3:.1f} ms")
