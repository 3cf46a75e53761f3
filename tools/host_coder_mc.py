#!/usr/bin/env python3
"""The part of an encode that stays on the host, measured frame-parallel: W worker processes, each takes frames of the benchmark
clip (sac_amd.synth.stereo_clip(1000, ...)), gets their residuals from the reference's own PredictStereoFrame (not timed) and
times the reference's FrameCoder::Encode() -- both BitplaneCoder variants per channel, codec.cpp:469-472 -- on them, all workers
at once. Prints one JSON object. Needs oracle/_ref (the compiled reference); started by bench.py as a child process (it
never touches the GPU). usage: host_coder_mc.py [workers] [frames]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
FRAME = 176400


def work(args):
    frames, nframes, barrier_t = args
    import ctypes as C
    import numpy as np
    import oracle_lib
    from sac_amd.synth import stereo_clip
    L, R = stereo_clip(1000, nframes * FRAME)
    lib = oracle_lib.Ref()
    prep = []
    for f in frames:
        l = np.ascontiguousarray(L[f * FRAME:(f + 1) * FRAME]); r = np.ascontiguousarray(R[f * FRAME:(f + 1) * FRAME])
        res = lib.predict(2, l, r, maps=True)
        prep.append((l, r, [np.ascontiguousarray(res[k], dtype=np.int32) for k in ("u0", "u1", "m0", "m1", "maxbpn")]))
    while time.time() < barrier_t:   # all workers start coding together
        time.sleep(0.001)
    t0 = time.perf_counter(); size = 0
    for l, r, a in prep:
        size += int(lib.lib.ref_encode_only(C.c_int(2), oracle_lib._ptr(l), oracle_lib._ptr(r), C.c_int(FRAME), *[oracle_lib._ptr(x) for x in a]))
    return time.perf_counter() - t0, size, time.time() > barrier_t + 1e-9


if __name__ == "__main__":
    import multiprocessing as mp
    import oracle_lib
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    if not oracle_lib.have_ref():
        print(json.dumps({"error": "oracle/_ref not built"})); sys.exit(0)
    w = min(w, nframes)
    share = [list(range(i, nframes, w)) for i in range(w)]
    # residual preparation takes ~2.5 s per frame on a core: the common start time leaves room for the slowest worker
    barrier_t = time.time() + 4.0 + 3.0 * max(len(s) for s in share)
    with mp.get_context("spawn").Pool(w) as pool:
        out = pool.map(work, [(s, nframes, barrier_t) for s in share])
    print(json.dumps({"seconds": max(o[0] for o in out), "cores": w, "frames": nframes, "coded_bytes": sum(o[1] for o in out),
                      "started_together": all(o[2] for o in out), "unit": "s",
                      "sample": f"{w} processes code the clip's {nframes} frames at once (FrameCoder::Encode() of the reference on the "
                                f"reference's residuals): slowest worker, {max(len(s) for s in share)} frame(s) each"}))
