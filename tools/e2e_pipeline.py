#!/usr/bin/env python3
"""End to end, MEASURED (second half): the reference's own program with its predictor served by libsacopt and its
Codec::EncodeFile loop batched and pipelined (oracle/_ref/sac_gpu, oracle/sac_gpu_shim.cpp: B frames per sacopt_encode_frames
call, the reference's Encode() of a batch on host threads while the GPU works on the next batch) encodes WAV files of the
benchmark's synthetic audio in optimize mode. Prints one JSON object: per file the loop's wall time (WAV read + GPU calls +
host coder + .sac write; context creation excluded) and encoded samples/s; and a byte-identity check of the batched file
against the reference binary's own (oracle/_ref/sac) on an 8 s prefix. TEST INFRASTRUCTURE (needs oracle/_ref); its own process.
usage: e2e_pipeline.py [coder_threads] [batch_frames]"""
import json, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
from make_sacfile import wav_bytes
from sac_amd.synth import stereo_clip
FRAME = 176400
REF = os.path.join(ROOT, "oracle", "_ref")


def run_gpu(wav, sac, batch, threads, device_coder=False):
    env = dict(os.environ, SAC_GPU_BATCH=str(batch), SAC_GPU_CODER_THREADS=str(threads), SAC_GPU_OPTIMIZE="1")
    if device_coder:
        env["SAC_GPU_DEVICE_CODER"] = "1"   # FrameCoder::Encode on the device too (sacopt_encode_payloads): no host coder
    t0 = time.perf_counter()
    out = subprocess.run([os.path.join(REF, "sac_gpu"), "--encode", "--high", wav, sac], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                         text=True, env=env, timeout=600)
    wall = time.perf_counter() - t0
    m = re.search(r"sac_gpu_timing (\{.*\})", out.stderr)
    if out.returncode or not m:
        raise RuntimeError(out.stderr[-300:])
    t = json.loads(m.group(1))
    t["process_wall_s"] = wall
    t["samples_per_s"] = t["samples"] / t["loop_s"]
    t["sac_bytes"] = os.path.getsize(sac)
    return t


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    if not (os.path.exists(os.path.join(REF, "sac_gpu")) and os.path.exists(os.path.join(REF, "sac"))):
        print(json.dumps({"error": "oracle/_ref programs not built"})); return
    res = {"program": "oracle/_ref/sac_gpu --encode --high (the reference's main(), WAV reader, .sac writer and bit-plane coder; Predict and the "
                      "EncodeFile loop from libsacopt / oracle/sac_gpu_shim.cpp), SAC_GPU_OPTIMIZE=1", "coder_threads": threads, "batch_frames": batch}
    with tempfile.TemporaryDirectory() as td:
        L, R = stereo_clip(1000, 75 * FRAME)
        for name, nfr in (("clip_60s", 15), ("track_300s", 75)):
            wav = os.path.join(td, name + ".wav")
            open(wav, "wb").write(wav_bytes(L[:nfr * FRAME], R[:nfr * FRAME], 44100))
            run_gpu(wav, os.path.join(td, name + ".sac"), batch, threads)            # warm-up (page cache, GPU code upload)
            res[name] = run_gpu(wav, os.path.join(td, name + ".sac"), batch, threads)
        # the whole encoder on the device: payload bytes from sacopt_encode_payloads, 75 frames per call; the file must be the same file
        name = "track_300s"
        wav = os.path.join(td, name + ".wav")
        run_gpu(wav, os.path.join(td, "dev.sac"), 75, threads, device_coder=True)
        res["track_300s_device_coder"] = run_gpu(wav, os.path.join(td, "dev.sac"), 75, threads, device_coder=True)
        res["track_300s_device_coder"]["same_file_as_host_coder"] = open(os.path.join(td, "dev.sac"), "rb").read() == open(os.path.join(td, name + ".sac"), "rb").read()
        # identity with the reference binary's own file, on a prefix it codes in a few seconds (2 frames, batches of 1 frame)
        wav = os.path.join(td, "prefix.wav")
        open(wav, "wb").write(wav_bytes(L[:2 * FRAME], R[:2 * FRAME], 44100))
        t0 = time.perf_counter()
        subprocess.run([os.path.join(REF, "sac"), "--encode", "--high", wav, os.path.join(td, "ref.sac")], check=True, stdout=subprocess.DEVNULL, timeout=600)
        ref_s = time.perf_counter() - t0
        run_gpu(wav, os.path.join(td, "gpu.sac"), 2, threads)
        res["identical_to_reference_program"] = {"samples": 2 * FRAME, "same_bytes": open(os.path.join(td, "ref.sac"), "rb").read() == open(os.path.join(td, "gpu.sac"), "rb").read(),
                                                 "reference_program_s": ref_s, "note": "the reference binary codes with its base profile and has no optimizer call; "
                                                 "in 'ref+opt' mode the stream is that same stream"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
