#!/bin/bash
# Round 3, GPU call 2: correctness of the DPP scans + second scatter + second sweep (full pytest -m gpu), A/B on the three
# scale-free graphs, per-kernel times of the forward search.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/c2_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c2_pytest.log
timeout 300 python tools/ab_r3.py lj 20 > gpurun_out/c2_ab_lj.log 2>&1
timeout 300 python tools/ab_r3.py kron 10 > gpurun_out/c2_ab_kron.log 2>&1
timeout 400 python tools/ab_r3.py twitter 5 > gpurun_out/c2_ab_twitter.log 2>&1
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c2_kt" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --only bfs,bfs_forward --no-cpu-baseline --steps 10 --warmup 2 > "$GRAFT_REPO_ROOT/gpurun_out/c2_kt.log" 2>&1
cd "$GRAFT_REPO_ROOT"
cp gpurun_out/c2_kt/*/p_kernel_stats.csv gpurun_out/c2_kernel_stats.csv 2>/dev/null || find gpurun_out/c2_kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/c2_kernel_stats.csv \;
find gpurun_out/c2_kt -name "*kernel_trace.csv" -exec cp {} gpurun_out/c2_kernel_trace.csv \;
rm -rf gpurun_out/c2_kt
tail -4 gpurun_out/c2_pytest.log; cat gpurun_out/c2_ab_lj.log | cut -c1-330
