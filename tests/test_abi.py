"""CPU: the C-ABI library loads and exports every symbol include/sacopt.h declares; the
host-only entry points behave; with no GPU the context fails loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from sac_amd import sacopt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "sacopt.h")).read()
    declared = sorted(set(re.findall(r"\b(sacopt_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    L = sacopt.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert set(declared) == set(sacopt.EXPORTS)


def test_base_profiles(golden):
    for prof in (0, 1, 2):
        assert sacopt.base_profile(prof).tobytes() == golden[f"base{prof}"].tobytes()
    assert sacopt.lib().sacopt_base_profile(7, None) == 0


def test_probe_rsqrt_host(oracle):
    # product-side probe == oracle-side probe on this CPU (both are host code; no GPU involved)
    from oracle_lib import Oracle
    tab, bits = sacopt.probe_rsqrt()
    otab, obits = Oracle().probe_rsqrt()
    assert bits == obits and tab.tobytes() == otab.tobytes()


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sacopt.SacoptError):
        sacopt.Context(0)


def test_product_library_has_no_debug_switches():
    # A/B and ablation switches (some of them break exactness) exist only in -DSACOPT_ABLATE builds of tools/:
    # the product library must not even contain the names of the environment variables that select them
    blob = open(sacopt.LIB_PATH, "rb").read()
    for name in (b"SACOPT_DBG", b"SACOPT_NOSPLIT"):
        assert name not in blob, name
