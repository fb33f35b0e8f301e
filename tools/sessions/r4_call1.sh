#!/bin/bash
# round 4, call 1: new full-size target-matrix tests + hardening tests + whole GPU suite, the new bench line, bench_all rows
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_target_matrix_gpu.py -x -q -s > gpurun_out/r4c1_pytest_matrix.log 2>&1; echo "rc $?" >> gpurun_out/r4c1_pytest_matrix.log)
(timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_target_matrix_gpu.py > gpurun_out/r4c1_pytest_gpu.log 2>&1; echo "rc $?" >> gpurun_out/r4c1_pytest_gpu.log)
(timeout 600 python bench.py > gpurun_out/r4c1_bench.log 2> gpurun_out/r4c1_bench.err; echo "rc $?" >> gpurun_out/r4c1_bench.log)
cp gpurun_out/bench_detail.json gpurun_out/r4c1_bench_detail.json 2>/dev/null
(timeout 600 python tests/tools/bench_all.py ssspu_lj sssp_lj ssspu_kron sssp_kron pr_lj > gpurun_out/r4c1_bench_all.log 2>&1; echo "rc $?" >> gpurun_out/r4c1_bench_all.log)
tail -3 gpurun_out/r4c1_pytest_matrix.log; tail -3 gpurun_out/r4c1_pytest_gpu.log; tail -c 600 gpurun_out/r4c1_bench.log; tail -2 gpurun_out/r4c1_bench.err; tail -c 1500 gpurun_out/r4c1_bench_all.log
