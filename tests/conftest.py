import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # build the C-ABI library and the oracle once per session (hipcc / gcc, no GPU needed)
    from gunrock_amd import build as _b
    _b.build()
    import oracle_lib
    oracle_lib.build_oracle()



@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "golden.npz"))


@pytest.fixture(scope="session")
def gr():
    import gunrock_amd
    return gunrock_amd


@pytest.fixture(scope="session")
def gpu_ctx(gr):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return gr.multi_context_t(0)
