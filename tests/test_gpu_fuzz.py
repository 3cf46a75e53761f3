"""GPU: a short fixed-seed run of tools/fuzz_parity.py -- every entry point on random signals, lengths and coefficient vectors
against the oracle, bit for bit (the long campaigns: profiles/r03e_fuzz_parity.txt). Runs with the FIXTURE's rsqrtss table."""
import os
import sys

import pytest

from sac_amd import sacopt

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_fuzz_fixed_seeds(rsqrt_table, oracle):
    import fuzz_parity
    lines = []
    with sacopt.Context(0, *rsqrt_table) as ctx:
        bad = 0
        for seed in (5, 77):   # one round each: 4 profiles x 20 clips
            bad += fuzz_parity.fuzz(ctx, oracle, seed, 1, 20, log=lambda *a: lines.append(" ".join(str(x) for x in a)))
    assert bad == 0, "\n".join(lines)
