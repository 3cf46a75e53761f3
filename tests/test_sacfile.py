"""CPU: the host-side .sac container (sac_amd/host/sacfile.h through libsachost.so) against files written by the
reference's own program (tests/golden/sacfiles.npz, made by tests/golden/make_sacfile.py from oracle/_ref/sac)."""
import hashlib

import numpy as np
import pytest

import oracle_lib
from sac_amd import sacfile
from sac_amd.synth import stereo_clip


@pytest.fixture(scope="module")
def sacfiles():
    import os
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sacfiles.npz"), allow_pickle=False))


def test_exports():
    L = sacfile.lib()
    for name in sacfile.EXPORTS:
        assert hasattr(L, name), name


def test_parse_and_repack_reference_files(sacfiles):
    # every byte of a reference-written file goes through ReadHeader / ReadEncoded and back through WriteHeader /
    # WriteEncoded unchanged; the header is what CanonicalWavMetadata predicts for the WAV the file was made from
    for prof in (0, 1, 2):
        blob = sacfiles[f"tiny{prof}_sac"].tobytes()
        seed, rate, n = [int(x) for x in sacfiles["tiny_params"]]
        hdr, pos = sacfile.read_header(blob)
        assert (hdr["numchannels"], hdr["samplerate"], hdr["bitspersample"], hdr["numsamples"], hdr["profile"]) == (2, rate, 16, n, prof)
        assert hdr["metadata"] == sacfile.canonical_wav_metadata(2, rate, 16, n)
        out = sacfile.write_header(2, rate, 16, n, prof, hdr["metadata"])
        left, sizes = n, []
        while left > 0:
            rec, used = sacfile.read_frame(blob, pos)
            out += sacfile.write_frame(rec["numsamples"], rec["ch"][0], rec["ch"][1])
            pos += used
            left -= rec["numsamples"]
            sizes.append(rec["numsamples"])
        assert sizes == [4 * rate, 4 * rate, n - 8 * rate] and pos == len(blob)  # framesize = 4 * samplerate (codec.cpp:533)
        assert out == blob


def test_parameter_extension_round_trip():
    coefs = np.array([0.98995, 1.009e-05, 3.007e-05, 3.007e-04, 1e-04, 1e-04, 0.0109, 0.998])
    mm = (-2 ** 31, 2 ** 31 - 1, -123, 456)
    rec = sacfile.write_frame(1234, (b"\x01\x02\x03", True, 9), (b"", False, 0), params=(coefs, mm))
    plain = sacfile.write_frame(1234, (b"\x01\x02\x03", True, 9), (b"", False, 0))
    assert len(rec) == len(plain) + 1 + 64 + 16
    got, used = sacfile.read_frame(rec + b"tail", 0, with_params=True)
    assert used == len(rec) and got["numsamples"] == 1234 and got["ch"] == [(b"\x01\x02\x03", True, 9), (b"", False, 0)]
    assert got["coefs"].tobytes() == coefs.tobytes() and got["minmax"] == mm
    with pytest.raises(ValueError):
        sacfile.read_frame(rec[:-1], 0, with_params=True)
    with pytest.raises(ValueError):
        sacfile.read_header(b"RIFFxxxxxxxxxxxxxxxxxxxxxxxxxxx")


def assemble_sac(encode_frame, seed, rate, n, prof):
    """Codec::EncodeFile (codec.cpp:530-556) with FrameCoder::Predict supplied by `encode_frame(prof, L, R) -> residual
    dict`, the reference's Encode() for the payloads, and THIS repo's container writer."""
    ref = oracle_lib.Ref()
    L, R = stereo_clip(seed, n, rate=rate)
    out = sacfile.write_header(2, rate, 16, n, prof, sacfile.canonical_wav_metadata(2, rate, 16, n))
    fs = 4 * rate
    for a in range(0, n, fs):
        l, r = L[a:a + fs], R[a:a + fs]
        ch = oracle_lib.ref_encode_channels(ref, prof, l, r, encode_frame(prof, l, r))
        out += sacfile.write_frame(len(l), ch[0], ch[1])
    return out


@pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("prof", [0, 1, 2])
def test_whole_file_from_oracle_residuals(sacfiles, oracle, prof):
    # the oracle's residuals + the reference's bit-plane coder + this repo's container writer == the reference program's file
    seed, rate, n = [int(x) for x in sacfiles["tiny_params"]]
    blob = assemble_sac(lambda p, l, r: oracle.predict(p, l, r, maps=True), seed, rate, n, prof)
    assert blob == sacfiles[f"tiny{prof}_sac"].tobytes()
    assert hashlib.sha256(blob).digest() == sacfiles[f"tiny{prof}_sha"].tobytes()


@pytest.mark.parametrize("prof", [0, 1, 2])
def test_whole_file_without_the_reference(sacfiles, oracle, prof):
    # nothing of the reference is loaded here: the oracle's predictor AND its restated bit-plane coder (oracle/sac_coder.c)
    # + this repo's container writer reproduce the file the reference's program wrote
    seed, rate, n = [int(x) for x in sacfiles["tiny_params"]]
    L, R = stereo_clip(seed, n, rate=rate)
    blob = sacfile.write_header(2, rate, 16, n, prof, sacfile.canonical_wav_metadata(2, rate, 16, n))
    fs = 4 * rate
    for a in range(0, n, fs):
        l, r = L[a:a + fs], R[a:a + fs]
        res = oracle.predict(prof, l, r, maps=True)
        ch = [oracle.encode_channel(res[f"u{c}"], res["maxbpn"][c], res[f"m{c}"], res["maxbpn"][2 + c], oracle.remap_tables(s))
              for c, s in enumerate((l, r))]
        blob += sacfile.write_frame(len(l), ch[0], ch[1])
    assert blob == sacfiles[f"tiny{prof}_sac"].tobytes()
