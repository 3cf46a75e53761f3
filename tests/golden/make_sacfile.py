"""Generate tests/golden/sacfiles.npz: whole .sac files written by the REFERENCE'S OWN PROGRAM (oracle/_ref/sac, compiled
in this container from /root/reference/src by oracle/Makefile) for seeded synthetic WAVs.

    make -C oracle ref && python tests/golden/make_sacfile.py

Fixtures are data only: per case the WAV parameters (the samples are re-generated from sac_amd.synth by seed), the
reference's .sac bytes (the tiny case) and, for every case, their SHA-256 + length, and the SHA-256 of the WAV
the reference decodes back (fast/normal round-trip; --high does not round-trip in the reference: SURVEY.md fact 3).
"""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from sac_amd.synth import stereo_clip  # noqa: E402

SAC = os.path.join(ROOT, "oracle", "_ref", "sac")
# name -> (seed, samplerate, numsamples): framesize = 4 * samplerate (codec.cpp:533), so the small rates give small frames
CASES = {"tiny": (501, 2000, 20000),        # 2.5 frames of 8000
         "mid": (502, 8000, 70001),         # 2 frames of 32000 + a ragged one of 6001
         "cd3": (503, 44100, 3 * 176400)}   # 3 full CD-rate frames (stored as SHA-256 only)


def wav_bytes(L, R, rate):
    """Canonical 44-byte-header 16-bit stereo PCM WAV."""
    n = len(L)
    pcm = np.empty(2 * n, np.int16)
    pcm[0::2] = L
    pcm[1::2] = R
    data = pcm.astype("<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, rate, rate * 4, 4, 16)
    return hdr + b"data" + struct.pack("<I", len(data)) + data


def main():
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, (seed, rate, n) in CASES.items():
            L, R = stereo_clip(seed, n, rate=rate)
            wav = os.path.join(td, name + ".wav")
            open(wav, "wb").write(wav_bytes(L, R, rate))
            out[f"{name}_params"] = np.array([seed, rate, n], np.int64)
            for prof, flag in ((0, "--fast"), (1, "--normal"), (2, "--high")):
                if name == "cd3" and prof == 2:
                    continue  # 3 x 176400 samples of StereoHigh: half a minute of CPU for nothing new
                sac = os.path.join(td, f"{name}{prof}.sac")
                subprocess.run([SAC, "--encode", flag, wav, sac], check=True, stdout=subprocess.DEVNULL)
                b = open(sac, "rb").read()
                out[f"{name}{prof}_sha"] = np.frombuffer(hashlib.sha256(b).digest(), np.uint8)
                out[f"{name}{prof}_len"] = np.int64(len(b))
                if name == "tiny":
                    out[f"{name}{prof}_sac"] = np.frombuffer(b, np.uint8)
                if prof != 2:   # the reference's own decoder restores the WAV bit for bit (fast/normal)
                    back = os.path.join(td, "back.wav")
                    subprocess.run([SAC, "--decode", sac, back], check=True, stdout=subprocess.DEVNULL)
                    assert open(back, "rb").read() == open(wav, "rb").read(), (name, prof)
                print(name, prof, len(b), flush=True)
    np.savez_compressed(os.path.join(HERE, "sacfiles.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
