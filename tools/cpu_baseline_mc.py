#!/usr/bin/env python3
"""Frame-parallel CPU baseline (SURVEY §8d ii): W worker processes, each runs the reference's (oracle/_ref, else the C
restatement's) Optimize + PredictStereoFrame on its own synthetic high-profile frame. Prints one JSON object.
Started by bench.py as a child process (it never touches the GPU). usage: cpu_baseline_mc.py [workers] [samples]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def work(seed_n):
    seed, n = seed_n
    import oracle_lib
    from sac_amd.synth import stereo_clip
    L, R = stereo_clip(seed, n)
    lib = oracle_lib.Ref() if oracle_lib.have_ref() else oracle_lib.Oracle()
    t0 = time.perf_counter()
    lib.optimize(2, L, R)
    lib.predict(2, L, R)
    return time.perf_counter() - t0


if __name__ == "__main__":
    import multiprocessing as mp
    import oracle_lib
    w = int(sys.argv[1]) if len(sys.argv) > 1 else min(16, os.cpu_count() or 1)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    oracle_lib.build_oracle()
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(w) as pool:
        per = pool.map(work, [(10 + i, n) for i in range(w)])
    dt = time.perf_counter() - t0
    print(json.dumps({"value": w * n / max(per), "unit": "samples/s", "cores": w,
                      "kind": "reference" if oracle_lib.have_ref() else "port",
                      "sample": f"{w} processes, each Optimize (84 candidates x {n // 8}-sample window) + full PredictStereoFrame of its own "
                                f"{n}-sample high-profile frame; slowest worker {max(per):.1f} s (wall incl. start-up {dt:.1f} s)"}))
