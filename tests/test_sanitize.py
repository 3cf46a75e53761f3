"""CPU: host code under AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile `sanitize`): the oracle's C
restatement on every profile / ragged and degenerate inputs / garbage streams, and the product's .sac container reader
(sac_amd/host/sacfile.h) on every prefix of a valid stream, byte flips and pure garbage (tests/host/*.c*)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def san_build():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True)
    if r.returncode:
        pytest.skip("sanitizer build not available here: " + r.stderr[-200:])
    return os.path.join(ROOT, "oracle", "_build")


@pytest.mark.parametrize("exe", ["san_oracle", "san_sacfile"])
def test_host_code_under_asan_ubsan(san_build, exe):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    r = subprocess.run([os.path.join(san_build, exe)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and (exe + " ok") in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
