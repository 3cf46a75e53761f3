#!/usr/bin/env python3
"""bench.py -- optimize-mode encode throughput of the sac optimizer/predictor path on MI355X.

One "step" = one pass of the hot path over one batch of frames per GPU: FrameCoder::Optimize (84 candidate evaluations
per frame on the centre 1/8 window, high profile) + the final full-frame PredictStereoFrame with the base profile
("ref+opt"), i.e. FrameCoder::Predict with the optimizer enabled. Bit-plane / range coding and file I/O
are NOT in the timed region (they stay on the host by the north star): `value` is the hot path's throughput, the
`end_to_end` object prices the host coder beside it.

  N = 1  BASELINE.json configs[2]: one 60 s synthetic 16-bit 44.1 kHz stereo clip (15 frames x 176400).
  N > 1  BASELINE.json configs[3]'s shape, weak-scaled: N x T five-minute tracks (75 frames x 176400 each; configs[3] is 64
         of them = T 8 on 8 GPUs; the default is --tracks-per-gpu 1, ONE EIGHTH of configs[3]'s per-GPU work, so that a step
         stays at ~1.5 s), the global frame list cut with sac_amd.dist.frame_shard (contiguous ranges, one per rank), one
         library call of 75 frames per track, the per-frame result rows (8 selected parameters + 4 maxbpn) all-gathered in
         frame order with sac_amd.dist.gather_frame_rows (RCCL with --backend nccl) inside the timed region. No collective
         on the data path.

The timed step asks for everything the encoder's next stage consumes (src/codec.cpp:165-211): error, s2u_error,
s2u_error_map (Remap::Analyse + Map on the device) and the four maxbpn.

HIP runtime order: torch bundles its own libamdhip64; libsacopt.so binds to whichever runtime with that soname the process
has loaded first. This program initialises torch's runtime FIRST and asserts that the process maps exactly one HIP runtime
(sacopt.assert_single_hip_runtime) -- a swapped order fails there with a message instead of torch's "No HIP GPUs".

Inputs are resident in HBM before the timed region. Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME = 176400            # 4 s @ 44.1 kHz (src/codec.cpp:533)
NFRAMES = 15              # 60 s clip (configs[2])
TRACK_FRAMES = 75         # 5-minute track (configs[3])
L1C_FRAMES = 48           # sacopt.cpp: calls of up to this many frames run the full-frame pass on k_lms1c (two workgroups per frame)
FLOPS_LMS1 = 2 * (11 * 1920 + 1)   # algorithmic FP64 flops per stereo sample of LMSADA2(1280,640) x 2 ch
FLOPS_UNIT_HIGH = 58934            # per candidate stereo sample, whole high cascade
FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X datasheet vector FP64


def cpu_baseline():
    """The reference's own code (oracle/_ref, kind 'reference'; else the C restatement, 'port') on ONE host core, on a
    bounded sample of the same workload: Optimize + PredictStereoFrame of ONE real 176400-sample frame (~20-25 s), and --
    reference only -- its bit-plane/range coder Encode() (codec.cpp:469-472) on the same frame: `host_coder`, the part of
    an encode that stays on the host."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from sac_amd.synth import stereo_clip
    n = FRAME
    L, R = stereo_clip(1000, n)    # frame 0 of the benchmark clip
    have = oracle_lib.have_ref()
    lib = oracle_lib.Ref() if have else oracle_lib.Oracle()
    t0 = time.perf_counter()
    lib.optimize(2, L, R)
    res = lib.predict(2, L, R, maps=True)
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "samples/s", "cores": 1, "kind": "reference" if have else "port",
           "sample": f"Optimize (84 candidates x {n // 8}-sample window) + full PredictStereoFrame of one {n}-sample high-profile "
                     f"frame (frame 0 of the benchmark clip), {dt:.1f} s on one core"}
    if have:
        import ctypes as C
        a = [np.ascontiguousarray(res[k], dtype=np.int32) for k in ("u0", "u1", "m0", "m1", "maxbpn")]
        t0 = time.perf_counter()
        size = lib.lib.ref_encode_only(C.c_int(2), oracle_lib._ptr(L), oracle_lib._ptr(R), C.c_int(n), *[oracle_lib._ptr(x) for x in a])
        dte = time.perf_counter() - t0
        out["host_coder"] = {"value": n / dte, "unit": "samples/s", "cores": 1, "seconds_per_frame": dte, "coded_bytes": int(size),
                             "sample": "the reference's FrameCoder::Encode() (both BitplaneCoder variants per channel) on that frame's residuals"}
    return out


def decode_extra(ctx, torch, nfr, dL, dR, e0, e1):
    """Not the headline: the decoder-side cascade (sacopt_unpredict_stereo_frames) on the residuals the timed
    steps left in HBM -- one call for the clip, checked against the input samples -- and the reference's
    UnpredictStereoFrame on one host core on a bounded sample beside it."""
    from sac_amd import sacopt
    v0 = torch.empty_like(dL); v1 = torch.empty_like(dL)
    fr = [dict(e0=e0.data_ptr() + 4 * f * FRAME, e1=e1.data_ptr() + 4 * f * FRAME, v0=v0.data_ptr() + 4 * f * FRAME,
               v1=v1.data_ptr() + 4 * f * FRAME, n=FRAME) for f in range(nfr)]
    ctx.unpredict(sacopt.PROFILE_HIGH, fr)
    t = ctx.last_timing()
    out = {"workload": f"UnpredictStereoFrame of the same {nfr} frames (high profile), one workgroup per frame",
           "ms": t["total_ms"], "kernel_ms": t["lms_ms"], "value": nfr * FRAME / (t["total_ms"] * 1e-3), "unit": "samples/s",
           "round_trip_ok": bool(torch.equal(v0, dL) and torch.equal(v1, dR))}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    n = 65536
    r0 = e0[:n].cpu().numpy(); r1 = e1[:n].cpu().numpy()
    lib = oracle_lib.Ref() if oracle_lib.have_ref() else oracle_lib.Oracle()
    t0 = time.perf_counter()
    lib.unpredict(2, r0, r1)
    dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": n / dt, "unit": "samples/s", "cores": 1, "kind": "reference" if oracle_lib.have_ref() else "port",
                           "sample": f"UnpredictStereoFrame of one {n}-sample high-profile frame, {dt:.1f} s on one core"}
    return out


def make_batch(torch, dev, stereo_clip, seed, nfr):
    L, R = stereo_clip(seed, nfr * FRAME)
    dL = torch.from_numpy(L).to(dev); dR = torch.from_numpy(R).to(dev)
    o = [torch.empty_like(dL) for _ in range(6)]   # error[2], s2u_error[2], s2u_error_map[2] (codec.cpp:165-204)
    frames = [(dL.data_ptr() + 4 * f * FRAME, dR.data_ptr() + 4 * f * FRAME, FRAME) for f in range(nfr)]
    outs = [{k: o[i].data_ptr() + 4 * f * FRAME for i, k in enumerate(("e0", "e1", "u0", "u1", "m0", "m1"))} for f in range(nfr)]
    return dict(L=L, R=R, dL=dL, dR=dR, o=o, frames=frames, outs=outs)


def check_frames_against_separate_calls(ctx, sacopt, b, best, which):
    """The overlapped call's results for the frames `which` against FrameCoder::Optimize and PredictStereoFrame run as
    separate library calls on those frames alone (another batch size: other kernels and stream sets): selected parameters
    as bits, error / s2u_error / s2u_error_map and maxbpn of the whole frame."""
    import torch  # noqa: F401
    fr = [b["frames"][f] for f in which]
    best_o, _, rows = ctx.optimize(sacopt.PROFILE_HIGH, fr)
    ok = best_o.tobytes() == np.ascontiguousarray(best[which]).tobytes()
    for i, f in enumerate(which):
        chk = ctx.predict(sacopt.PROFILE_HIGH, [fr[i]], maps=True)[0]
        for j, k in enumerate(("e0", "e1", "u0", "u1", "m0", "m1")):
            ok = ok and np.array_equal(chk[k], b["o"][j][f * FRAME:(f + 1) * FRAME].cpu().numpy())
        ok = ok and np.array_equal(chk["maxbpn"], b["maxbpn"][f])
    return bool(ok), rows


def large_batch_extra(ctx, torch, dev, stereo_clip, sacopt, nfr=TRACK_FRAMES):
    """Not the headline: the same call on a 5-minute track (75 frames, BASELINE configs[3]'s unit, what every rank runs at
    N > 1) -- with more chains than CUs the chip is full and throughput is higher than on the 60 s clip, whose 15 serial
    chains bound the call."""
    b = make_batch(torch, dev, stereo_clip, 2000, nfr)
    ctx.encode(sacopt.PROFILE_HIGH, b["frames"], device_out=b["outs"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    best, res = ctx.encode(sacopt.PROFILE_HIGH, b["frames"], device_out=b["outs"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b["maxbpn"] = [r["maxbpn"] for r in res]
    ok, _ = check_frames_against_separate_calls(ctx, sacopt, b, best, [7, nfr - 1])
    return {"workload": f"{nfr} frames x {FRAME} (300 s track), same call", "ms": 1e3 * dt, "value": nfr * FRAME / dt if ok else None,
            "unit": "samples/s", "checked_against_separate_calls": ok}


def two_channel_extra(ctx, torch, dev, stereo_clip, sacopt, nch=L1C_FRAMES + 1, n=20480):
    """Not the headline: serial latency of the OTHER (1280,640) kernel, k_lms2 -- both channels of a chain in one workgroup, what
    the optimizer's rounds (more than 20 chains) and calls of more than 48 frames use -- from a full-frame pass of 49 short
    frames: its lead stage is then that kernel, one workgroup per CU beside the follower stages, and the library times those
    launches."""
    L, R = stereo_clip(3000, nch * n)
    dL = torch.from_numpy(L).to(dev); dR = torch.from_numpy(R).to(dev)
    o = [torch.empty_like(dL) for _ in range(4)]   # (this extra times one kernel: no Remap outputs)
    frames = [(dL.data_ptr() + 4 * f * n, dR.data_ptr() + 4 * f * n, n) for f in range(nch)]
    outs = [dict(e0=o[0].data_ptr() + 4 * f * n, e1=o[1].data_ptr() + 4 * f * n, u0=o[2].data_ptr() + 4 * f * n, u1=o[3].data_ptr() + 4 * f * n)
            for f in range(nch)]
    for _ in range(2):
        ctx.predict(sacopt.PROFILE_HIGH, frames, device_out=outs)
    t = ctx.last_timing()
    per_chain = t["lms_units"] / nch
    return {"kernel": "k_lms2", "workload": f"full-frame pass of {nch} frames x {n} samples", "launches": t["lms_launches"],
            "kernel_ms": t["lms_ms"], "cycles_per_sample_at_2.4GHz": t["lms_ms"] * 2.4e6 / per_chain if per_chain else None}


def device_coder_extra(ctx, torch, sacopt, b, nfr, step_s):
    """Not the headline: FrameCoder::Encode of the clip on the device as well (sacopt_encode_payloads, SURVEY 8f.4 taken to the
    bytes): both payload variants of every channel from the residuals the step left in HBM, the smaller kept as EncodeMonoFrame
    does. Bytes are checked against the oracle's coder in tests/; here the kept sizes are compared with the host coder's."""
    L = b["dL"].cpu().numpy(); R = b["dR"].cpu().numpy()

    def tables(x):   # Remap::Analyse (map.cpp:111-130): usedl[0..32768] then usedh[0..32768]
        t = np.zeros(2 * 32769, np.uint8)
        t[32769 + x[(x > 0) & (x <= 32768)]] = 1
        t[-x[(x < 0) & (x >= -32768)]] = 1
        return t
    bufs, mb, used = [], [], []
    for f in range(nfr):
        o = b["outs"][f]
        for ch, x in enumerate((L, R)):
            u = tables(x[f * FRAME:(f + 1) * FRAME])
            bufs += [(o[f"u{ch}"], FRAME), (o[f"m{ch}"], FRAME)]
            mb += [int(b["maxbpn"][f][ch]), int(b["maxbpn"][f][2 + ch])]
            used += [None, u]
    ctx.encode_payloads(bufs[:4], mb[:4], used[:4])   # warm-up (arena)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pay = ctx.encode_payloads(bufs, mb, used)
    dt = time.perf_counter() - t0
    kept = sum(min(len(pay[k]), len(pay[k + 1])) for k in range(0, len(pay), 2))
    return {"workload": f"both payload variants of both channels of {nfr} frames x {FRAME} ({len(bufs)} coders, one call)", "ms": 1e3 * dt,
            "kept_payload_bytes": kept, "mapped_kept": sum(len(pay[k + 1]) <= len(pay[k]) for k in range(0, len(pay), 2)),
            "all_device_samples_per_s": nfr * FRAME / (step_s + dt), "note": "GPU step, then the coder, nothing on the host but the container"}


def apply_params_extra(ctx, torch, sacopt, b, nfr):
    """Not the headline: SURVEY §8f.2's opt-in mode, where the stream is coded with the parameters Optimize selected: the
    full-frame pass depends on the optimizer's result. As two library calls (nothing overlaps) and as ONE
    (sacopt_encode_frames_selected: every part of the pass starts when the optimizer has fixed its parameters); the one call's
    parameters and residuals (first and last frame) are checked against the two calls'."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    best, _, _ = ctx.optimize(sacopt.PROFILE_HIGH, b["frames"])
    ctx.predict(sacopt.PROFILE_HIGH, b["frames"], coefs=best, device_out=b["outs"])
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t0
    ctx.encode(sacopt.PROFILE_HIGH, b["frames"], device_out=b["outs"], selected=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    best1, _ = ctx.encode(sacopt.PROFILE_HIGH, b["frames"], device_out=b["outs"], selected=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = best1.tobytes() == best.tobytes()
    for f in sorted({0, nfr - 1}):
        chk = ctx.predict(sacopt.PROFILE_HIGH, [b["frames"][f]], coefs=best[f:f + 1], maps=True)[0]
        for j, k in enumerate(("e0", "e1", "u0", "u1", "m0", "m1")):
            ok = ok and np.array_equal(chk[k], b["o"][j][f * FRAME:(f + 1) * FRAME].cpu().numpy())
    return {"workload": f"Optimize, then PredictStereoFrame with the selected parameters ({nfr} frames): one call, the parts of the full-frame pass "
                        "released round by round",
            "ms": 1e3 * dt, "value": nfr * FRAME / dt, "unit": "samples/s", "two_calls_ms": 1e3 * dt2, "same_as_two_calls": bool(ok)}


def pmc_traffic(full_chip=False):
    """HBM bytes per launch of the dominant kernel, (bytes, file): a STORED figure from the newest committed PMC pass of this
    same command (profiles/rNN*_pmc_hbm.json, highest round prefix; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, collected
    in rocprofv3 --pmc runs of their own) -- a bench run cannot read the counters itself. full_chip: the 75-frame call's
    two-channel kernel from profiles/rNN*_pmc_75frames.json. (None, None) if not collected."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_pmc_75frames.json" if full_chip else "r[0-9][0-9]*_pmc_hbm.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                ks = json.load(f)["kernels"]
            for name, v in ks.items():
                if full_chip and name.startswith("k_lms2"):
                    return v["hbm_bytes"] / v["dispatches"], os.path.relpath(path, ROOT)
                if not full_chip and name.startswith("k_lms1c"):
                    return v["hbm_bytes_per_dispatch"], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    return None, None


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def split_extra(ctx, torch, sacopt, b):
    """Not the headline: the two halves of the overlapped call alone (what bounds sacopt_encode_frames): FrameCoder::Optimize
    of the clip, and the full-frame PredictStereoFrame of the clip with Remap outputs."""
    out = {}
    for name, fn in (("optimize_alone_ms", lambda: ctx.optimize(sacopt.PROFILE_HIGH, b["frames"])),
                     ("full_frame_pass_alone_ms", lambda: ctx.predict(sacopt.PROFILE_HIGH, b["frames"], device_out=b["outs"]))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        out[name] = 1e3 * (time.perf_counter() - t0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU and step (default: 15 at N=1, 75 at N>1)")
    ap.add_argument("--tracks-per-gpu", type=int, default=1, help="N > 1: five-minute tracks per GPU and step, one 75-frame library call each "
                    "(BASELINE configs[3] = 64 tracks = 8 per GPU on 8 GPUs; default 1 = one eighth of that per-GPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the (untimed) decoder-side extra")
    ap.add_argument("--no-large-batch", action="store_true", help="skip the (untimed) 75-frame extra")
    ap.add_argument("--no-extras", action="store_true", help="timed steps only: no baseline, no untimed extras (the command that is profiled)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path)")
    ap.add_argument("--rehearse-one-gpu", action="store_true", help="all ranks on GPU 0 (rehearsal of the N > 1 path on a one-GPU box; needs --backend gloo)")
    args = ap.parse_args()
    if args.no_extras:
        args.no_cpu_baseline = args.no_decode = args.no_large_batch = True

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.rehearse_one_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.cuda.init()   # torch's HIP runtime first (see the module docstring); libsacopt.so is loaded after it and binds to it
    torch.zeros(1, device=dev)

    from sac_amd import sacopt
    from sac_amd.dist import frame_shard, gather_frame_rows
    from sac_amd.synth import stereo_clip

    tracks = max(args.tracks_per_gpu, 1) if world > 1 else 1
    per_call = args.frames or (NFRAMES if world == 1 else TRACK_FRAMES)   # frames per library call
    per_gpu = per_call * tracks
    total_frames = per_gpu * world
    lo, hi = frame_shard(total_frames, world, rank)      # contiguous frame range of this rank (= its own tracks at N > 1)
    nfr = hi - lo
    b = make_batch(torch, dev, stereo_clip, 1000 + rank, nfr)
    frames, outs = b["frames"], b["outs"]
    # rsqrtss table probed from this host's CPU, as the reference would see it (src/pred/nlms.h:34-57); recorded with the result
    rsq_tab, rsq_bits = sacopt.probe_rsqrt()
    ctx = sacopt.Context(local_rank, rsq_tab, rsq_bits)
    sacopt.assert_single_hip_runtime()
    import hashlib
    rsq_sha = hashlib.sha256(np.ascontiguousarray(rsq_tab, dtype=np.float32).tobytes()).hexdigest()[:16]

    lms_ms = 0.0; lms_launches = 0; lms_units = 0
    table = None

    def step(timed):
        nonlocal lms_ms, lms_launches, lms_units, table
        # FrameCoder::Predict with Optimize enabled: optimizer + final full-frame pass in one call (per track at N > 1)
        bests, res = [], []
        for c0 in range(0, nfr, per_call):
            bb, rr = ctx.encode(sacopt.PROFILE_HIGH, frames[c0:c0 + per_call], device_out=outs[c0:c0 + per_call])
            bests.append(bb); res += rr
            t1 = ctx.last_timing()
            if timed:
                lms_ms += t1["lms_ms"]
                lms_launches += t1["lms_launches"]
                lms_units += t1["lms_units"]
        best = np.concatenate(bests)
        b["maxbpn"] = [r["maxbpn"] for r in res]
        if world > 1:   # what the writer rank needs from every frame, in frame order (RCCL all-gather of 96 B per frame)
            rows = np.concatenate([best, np.stack([r["maxbpn"] for r in res]).astype(np.float64)], axis=1)
            table = gather_frame_rows(rows, total_frames, world, rank)
        return best

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best = step(True)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    samples = total_frames * FRAME * args.steps
    # ---- outside the timed region: the number of GetCost calls is measured (not assumed), and the overlapped call is
    # checked against Optimize and PredictStereoFrame run separately, so that a broken overlap cannot post a number
    if args.no_extras:
        rows, verified = np.full(1, 84), None   # profiling runs: nothing but the timed steps touches the GPU
    else:
        verified, rows = check_frames_against_separate_calls(ctx, sacopt, b, best, [0, nfr - 1] if nfr > 1 else [0])
    if world > 1:
        assert table is not None and table.shape == (total_frames, 12) and table[lo:hi, :8].tobytes() == best.tobytes()
    if rank == 0:
        cand = int(rows[0])
        cand_units = total_frames * (cand * (FRAME // 8) + FRAME)   # candidate stereo samples per step (algorithmic)
        avg_launch_s = (lms_ms / 1e3) / max(lms_launches, 1)
        units_per_launch = lms_units / max(lms_launches, 1)
        achieved = units_per_launch * FLOPS_LMS1 / avg_launch_s / 1e12 if lms_launches else 0.0
        nfr_call = min(nfr, per_call)
        one_ch = nfr_call <= L1C_FRAMES   # the full-frame pass runs k_lms1c (sacopt.cpp: L1C_FRAMES), else the two-channel k_lms2
        wgs = 2 * nfr_call if one_ch else nfr_call   # per-channel kernel: two workgroups per frame; two-channel kernel: one
        traffic, traffic_file = pmc_traffic() if nfr_call == NFRAMES else (pmc_traffic(True) if nfr_call == TRACK_FRAMES else (None, None))
        if world == 1:
            workload = (f"{nfr * 4} s synthetic 16-bit 44.1 kHz stereo clip ({nfr} frames x {FRAME}), high profile, full optimizer population "
                        f"({cand} GetCost per frame) + final full-frame predict (BASELINE configs[2]); range coding and file I/O excluded")
        else:
            workload = (f"{world * tracks} synthetic 5-minute 16-bit 44.1 kHz stereo tracks ({per_call} frames x {FRAME} each), {tracks} per GPU "
                        f"= {tracks}/8 of BASELINE configs[3]'s per-GPU work (64 tracks on 8 GPUs), one library call per track; high profile, full "
                        f"optimizer population ({cand} GetCost per frame) + final full-frame predict with Remap outputs; frames sharded with "
                        f"frame_shard, per-frame rows all-gathered over {'RCCL' if args.backend == 'nccl' else args.backend}; range coding and file I/O excluded")
        e2e_cores = 0
        line = {
            "metric": "encoded samples/sec (node) in optimize mode", "value": samples / dt if verified is not False else None, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "frames_per_gpu": nfr, "frames_per_call": nfr_call, "tracks_per_gpu": tracks, "candidates_per_frame": cand,
                       "outputs": "error, s2u_error, s2u_error_map, maxbpn, maxbpn_map per channel (codec.cpp:165-211) + 8 selected parameters per frame",
                       "rsqrt_bits": int(rsq_bits), "rsqrt_table_sha256_16": rsq_sha, "host_cpu": host_cpu_model(),
                       "parallelism": f"frames sharded x{world}, no collective on the data path" + ("" if world == 1 else f"; result rows gathered over {args.backend}"), "checked_against_separate_calls": verified},
            "algorithmic_tflops_whole_path": cand_units * args.steps * FLOPS_UNIT_HIGH / dt / 1e12,
            "roofline": {"bound": "mfma", "kernel": "k_lms1c" if one_ch else "k_lms2 (two-channel (1280,640) kernel: full-frame pass and the optimizer's heavy rounds)",
                         "note": "FP64 compute bound (78.6 TFLOP/s vector = matrix FP64 peak); the real ceiling is the dependent-add "
                                 "latency of the serial 1920-term sums, the matrix pipe is used as an exact f64 adder, not a GEMM",
                         "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_source": (traffic_file + " (stored PMC pass of this command, not this run)") if traffic_file else None,
                         "workgroups_per_launch": wgs if one_ch else None,
                         "frac_of_occupied_cus": achieved / (FP64_VECTOR_PEAK_TFLOPS * min(wgs, 256) / 256.0) if one_ch else None,
                         "avg_launch_ms": 1e3 * avg_launch_s, "launches": lms_launches,
                         "units_per_launch": units_per_launch, "flops_per_unit": FLOPS_LMS1,
                         # serial latency of the dominant kernel: launch duration over the samples one chain advances in it
                         "cycles_per_sample_at_2.4GHz": (avg_launch_s * 2.4e9 / (units_per_launch / nfr_call)) if (lms_launches and one_ch) else None},
        }
        if verified is False:
            line["error"] = "the overlapped call's results differ from FrameCoder::Optimize / PredictStereoFrame run separately: no value reported"
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cb = cpu_baseline()
            import subprocess
            try:  # frame-parallel on the box's CPU share (16 cores), as its own process: SURVEY §8d (ii)
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_mc.py"), "16", str(FRAME)],
                                     capture_output=True, text=True, timeout=300)
                line["cpu_baseline_all_cores"] = json.loads(out.stdout.strip().splitlines()[-1])
            except Exception as e:  # a reported baseline, never the measurement
                line["cpu_baseline_all_cores"] = {"error": str(e)[:200]}
            if "host_coder" in cb:
                # What an encode of this clip costs end to end when the host keeps the reference's coder: the GPU step plus
                # Encode() of the frames on `cores` host cores (frames are independent: one frame per core at a time). Derived
                # from the measured one-core rate of cpu_baseline.host_coder; serial = GPU step then coder, pipelined =
                # coder of batch k beside the GPU step of batch k+1.
                cores = 16
                gpu_s = dt / args.steps
                coder_s = cb["host_coder"]["seconds_per_frame"] * ((nfr + cores - 1) // cores)
                how = "the one-core measurement of cpu_baseline.host_coder x ceil(frames / cores)"
                measured = None
                try:  # measured: the clip's frames coded at once by `cores` processes (tools/host_coder_mc.py, its own process)
                    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_coder_mc.py"), str(cores), str(nfr)],
                                         capture_output=True, text=True, timeout=240)
                    measured = json.loads(out.stdout.strip().splitlines()[-1])
                    if "seconds" in measured:
                        coder_s = measured["seconds"]
                        how = f"measured: {measured['sample']}"
                except Exception as e:
                    measured = {"error": str(e)[:200]}
                line["end_to_end"] = {"cores": cores, "gpu_step_s": gpu_s, "host_coder_s": coder_s, "host_coder_parallel": measured,
                                      "serial_samples_per_s_derived": nfr * FRAME / (gpu_s + coder_s),
                                      "note": "host coder = the reference's FrameCoder::Encode() (bit-plane + range coder, unchanged, on the host); "
                                              "its time is " + how}
                e2e_cores = cores   # (measured after this process has released its context, below)
        if world == 1 and nfr == NFRAMES and not args.no_extras:
            line["split"] = split_extra(ctx, torch, sacopt, b)
            line["apply_selected_parameters"] = apply_params_extra(ctx, torch, sacopt, b, nfr)
        if world == 1 and nfr == NFRAMES and not args.no_extras:
            line["two_channel_kernel"] = two_channel_extra(ctx, torch, dev, stereo_clip, sacopt)
        if world == 1 and nfr == NFRAMES and not args.no_extras and "maxbpn" in b:
            ctx.encode(sacopt.PROFILE_HIGH, frames, device_out=outs)   # residuals of the clip back in the output buffers
            line["device_coder"] = device_coder_extra(ctx, torch, sacopt, b, nfr, dt / args.steps)
        if not args.no_large_batch and world == 1 and nfr == NFRAMES:
            line["large_batch"] = large_batch_extra(ctx, torch, dev, stereo_clip, sacopt)   # untimed extra, not `value`
        if not args.no_decode and world == 1:
            ctx.encode(sacopt.PROFILE_HIGH, frames, device_out=outs)   # residuals of the clip back in the output buffers
            line["decode"] = decode_extra(ctx, torch, nfr, b["dL"], b["dR"], b["o"][0], b["o"][1])   # after the timed region, not part of `value`
        if e2e_cores:
            # MEASURED end to end: the reference's own program with the batched, pipelined binding (tools/e2e_pipeline.py, its own
            # process and context). This process closes its context first: a context holds 19 hardware queues, and two
            # processes' worth on one GPU is past the point where every call gets slower -- with this
            # process's context alive the child's GPU calls took 0.95 s per batch instead of 0.42.
            ctx.close()
            try:
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_pipeline.py"), str(e2e_cores), "15"],
                                     capture_output=True, text=True, timeout=400)
                line["end_to_end"]["measured"] = json.loads(out.stdout.strip().splitlines()[-1])
            except Exception as e:
                line["end_to_end"]["measured"] = {"error": str(e)[:200]}
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if verified is False:
        sys.exit(1)


if __name__ == "__main__":
    main()
