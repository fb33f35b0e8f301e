#!/bin/bash
# round 4, call 15: generic operators with frontiers pre-sized like upstream's enactor; CLI / drop-in binaries rebuilt on the headers
mkdir -p gpurun_out
bash tools/bench_generic.sh > gpurun_out/r4c15_generic.log 2>&1; cat gpurun_out/generic_bfs.log
timeout 900 python -m pytest tests/test_cli.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r4c15_pytest.log; cat gpurun_out/r4c15_pytest.log
