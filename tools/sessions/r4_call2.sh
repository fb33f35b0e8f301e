#!/bin/bash
# round 4, call 2: the block-asynchronous path (tests, then block size x bucket width on the full road stand-in), the
# multi-source bench after the multi-level exit fix, PageRank layout experiment on the LJ stand-in
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_block_gpu.py tests/test_mid_gpu.py -x -q > gpurun_out/r4c2_pytest_block.log 2>&1; echo "rc $?" >> gpurun_out/r4c2_pytest_block.log)
tail -4 gpurun_out/r4c2_pytest_block.log
(timeout 900 python tools/ab_block.py 4894 3 > gpurun_out/r4c2_ab_block.log 2>&1; echo "rc $?" >> gpurun_out/r4c2_ab_block.log)
cat gpurun_out/r4c2_ab_block.log | cut -c1-260
(timeout 600 python -m pytest tests/test_sssp_gpu.py tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -x -q > gpurun_out/r4c2_pytest_sssp_bfs.log 2>&1; echo "rc $?" >> gpurun_out/r4c2_pytest_sssp_bfs.log)
tail -4 gpurun_out/r4c2_pytest_sssp_bfs.log
(timeout 300 python bench.py --only bfs,bfs_do,multi,pr_lj --no-cpu-baseline > gpurun_out/r4c2_bench_multi.log 2>&1; echo "rc $?" >> gpurun_out/r4c2_bench_multi.log)
python - <<'P'
import json
d = json.load(open("gpurun_out/bench_detail.json"))
print("multi", {k: d["multi_source"][k] for k in ("forward_mteps", "forward_per_source_gteps", "forward_vs_single_source", "do_mteps")})
print("pr_lj", d["pr_lj"]["ms_per_iteration"], d["pr_lj"]["first_call_ms"], d["pr_lj"]["roofline"]["frac"])
P
