"""The block-asynchronous relaxation (gunrock_amd/csrc/grx_block.hip) lives in a library of its own since round 6
(gunrock_amd/libgrx_block.so, `python -m gunrock_amd.build --with-block`): the path is opt-in (GRX_BLOCK=1) and was measured no
better than the level-synchronous kernels, so the default library and the default suite no longer pay for it.
With GRX_TEST_BLOCK=1 and the variant built, its tests -- tests/test_block_host.py (host emulation, CPU) and
tests/test_block_gpu.py + the full-size road cases of tests/test_sssp_gpu.py (GPU) -- run here in a subprocess whose GRX_LIB_PATH
points at the variant."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gunrock_amd", "libgrx_block.so")
want = pytest.mark.skipif(os.environ.get("GRX_TEST_BLOCK") != "1" or not os.path.exists(LIB),
                          reason="opt-in: GRX_TEST_BLOCK=1 and python -m gunrock_amd.build --with-block")


def _run(args, timeout):
    env = dict(os.environ, GRX_LIB_PATH=LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q"] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-500:]


@want
def test_host_emulation_of_the_block_schedule_on_the_variant_library():
    _run(["tests/test_block_host.py", "-m", "not gpu"], 1500)


@want
@pytest.mark.gpu
def test_block_asynchronous_searches_on_the_variant_library():
    _run(["tests/test_block_gpu.py", "tests/test_sssp_gpu.py", "-m", "gpu", "-k", "block or full_size_road"], 1500)
