#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/final_bench.log 2>&1; echo "rc $?" >> gpurun_out/final_bench.log
tail -2 gpurun_out/final_bench.log | cut -c1-200
