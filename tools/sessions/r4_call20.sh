#!/bin/bash
# round 4, call 20: long randomised parity sweep on the final sources (BFS forward / direction-optimising, SSSP incl. forced binned relaxation and the
# block-asynchronous path, all against the oracle)
mkdir -p gpurun_out
timeout 420 python tests/tools/fuzz_gpu.py 330 > gpurun_out/r4c20_fuzz.log 2>&1; tail -3 gpurun_out/r4c20_fuzz.log
