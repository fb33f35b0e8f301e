#!/bin/bash
# round 6, call 39: the launch groups of a repeated paced search captured once and replayed as one HIP graph (GRX_PACED_GRAPH)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/ab_r6.py lj 20 "GRX_PACED_GRAPH=0" "-" "GRX_PACED_GRAPH=0" "-" 2>&1 | grep -v amdgpu.ids | cut -c1-200 > gpurun_out/r6_c39_paced_graph.txt
cat gpurun_out/r6_c39_paced_graph.txt
for cfg in "GRX_PACED_GRAPH=0" "GRX_PACED_GRAPH=1"; do
  echo "== [$cfg]"
  env $cfg timeout 600 python bench.py --only bfs,bfs_do,multi,bfs_deep,c5 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
s=j['config']['sections']
print('  step %.4f ms (rep %s) | do %.4f ms | multi fwd %.3f do %.0f | deep %.4f / %.4f | c5 %.4f / %.4f' % (j['ms_per_step'], j['config']['ms_per_step_repeated'], s['bfs_do']['ms'], s['multi_source']['forward_vs_single_source'], s['multi_source']['do_mteps'], s['bfs_deep']['fwd_ms'], s['bfs_deep']['do_ms'], s['c5_1gpu']['fwd_ms'], s['c5_1gpu']['do_ms']))"
done >> gpurun_out/r6_c39_paced_graph.txt 2>&1
tail -4 gpurun_out/r6_c39_paced_graph.txt
(timeout 900 python -m pytest tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x -k "not twitter" > gpurun_out/r6_c39_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c39_pytest.log); tail -3 gpurun_out/r6_c39_pytest.log
