#!/bin/bash
# round 6, call 20: the bench line on the sources of "partition tests + bench hygiene" (first full line of the round's second session)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r6_c20_bench.log 2> gpurun_out/r6_c20_bench.err; echo "rc $?" >> gpurun_out/r6_c20_bench.log
cp gpurun_out/bench_detail.json gpurun_out/r6_c20_bench_detail.json 2>/dev/null
cat gpurun_out/r6_c20_bench.log; tail -3 gpurun_out/r6_c20_bench.err
