#!/bin/bash
# round 4, call 15 (run twice): generic operators -- frontiers pre-sized like upstream, then merge path split on slots + atoms; CLI / drop-in binaries
mkdir -p gpurun_out
bash tools/bench_generic.sh > gpurun_out/r4c15_generic.log 2>&1; cat gpurun_out/generic_bfs.log
timeout 900 python -m pytest tests/test_cli.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r4c15_pytest.log; cat gpurun_out/r4c15_pytest.log
