#!/bin/bash
# round 4, call 13: rehearsal -- the 8-rank C5 test on one GPU, the default bench run
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_distributed.py -m gpu -q -k "eight_ranks" > gpurun_out/r4c13_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r4c13_pytest.log); tail -5 gpurun_out/r4c13_pytest.log
/usr/bin/time -v timeout 900 python bench.py > gpurun_out/r4c13_bench.log 2> gpurun_out/r4c13_bench.err; echo "rc $?"; tail -1 gpurun_out/r4c13_bench.log | cut -c1-4200; grep -E 'Elapsed|Maximum resident' gpurun_out/r4c13_bench.err
cp gpurun_out/bench_detail.json gpurun_out/r4c13_bench_detail.json
