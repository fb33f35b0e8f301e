#!/bin/bash
# round 4, call 3: the stable radix sort (tests, first-call times), block path after the LDS-chain rework (tests + sweep),
# weighted SSSP on the dense stand-ins (near-far widths), generic operators
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_sort_gpu.py tests/test_block_gpu.py tests/test_pr_gpu.py tests/test_cli.py -x -q > gpurun_out/r4c3_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r4c3_pytest.log)
tail -5 gpurun_out/r4c3_pytest.log
(timeout 900 python tools/ab_block.py 4894 3 > gpurun_out/r4c3_ab_block.log 2>&1; echo "rc $?" >> gpurun_out/r4c3_ab_block.log)
cat gpurun_out/r4c3_ab_block.log | cut -c1-250
(timeout 300 python tools/ab_sssp_dense.py lj > gpurun_out/r4c3_sssp_dense_lj.log 2>&1; echo "rc $?" >> gpurun_out/r4c3_sssp_dense_lj.log)
cat gpurun_out/r4c3_sssp_dense_lj.log
(timeout 400 python bench.py --only bfs,bfs_do,pr,pr_lj --no-cpu-baseline > gpurun_out/r4c3_bench.log 2>&1; echo "rc $?" >> gpurun_out/r4c3_bench.log)
python - <<'P'
import json
d = json.load(open("gpurun_out/bench_detail.json"))
print("bfs fwd first_call_ms", d["config"]["first_call_ms"], "do", d["bfs_direction_optimized"]["first_call_ms"], d["bfs_direction_optimized"]["ms_per_step"])
for k in ("pr_lj", "pr_kron"):
    print(k, d[k]["ms_per_iteration"], "first_call_ms", d[k]["first_call_ms"], d[k]["iterations"], d[k]["roofline"]["frac"])
P
(timeout 300 bash tools/bench_generic.sh > gpurun_out/r4c3_generic.log 2>&1; echo "rc $?" >> gpurun_out/r4c3_generic.log)
tail -25 gpurun_out/r4c3_generic.log
