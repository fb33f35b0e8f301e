"""A bounded slice of the randomised GPU parity sweep (tests/tools/fuzz_gpu.py) under pytest -m gpu:
random graphs / sources, BFS forward (claim-per-edge and binned levels) and direction-optimising,
synchronous and async-return back to back, SSSP unit and weighted -- all bit-exact against the oracle.
The direction-optimising path keeps the most state (three rotating bitmaps + visited + tiles + the
tiny-level hand-back); the one parity bug of round 1 was found by this sweep."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))

pytestmark = pytest.mark.gpu


def test_fuzz_slice(gr, gpu_ctx):
    import fuzz_gpu
    lines = []
    budget = float(os.environ.get("GRX_FUZZ_SECONDS", "45"))
    graphs, checks, bad = fuzz_gpu.sweep(budget, seed=20260924, log=lines.append, max_vertices=300_000)
    assert graphs >= 3 and checks >= 30, (graphs, checks)
    assert bad == 0, "\n".join(l for l in lines if "MISMATCH" in l or "BAD" in l)
