"""A short run of tools/soak.py inside the GPU suite: random table sizes, batch widths, selectivities, key ranges and aggregate
sets, every iteration checked against invariants that the exact data distribution makes bit-exact and, differentially,
against another kernel family (interpreter, global table, ring kernel, two-pass filter, a table that grows from 2^9 slots ...).
The long runs (minutes, several seeds) are recorded in DESIGN.md section 2; this one keeps the tool itself alive."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_eight_seconds_of_random_queries_keep_their_invariants():
    env = dict(os.environ, DFX_NO_TORCH="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "8", "11"], capture_output=True, text=True, timeout=600, env=env)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0 and "soak ok:" in r.stdout, tail
