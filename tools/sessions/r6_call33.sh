#!/bin/bash
# round 6, call 33: scatter + sweep of a group on a side stream beside the level kernel when the schedule is not known (GRX_TWO_STREAMS=1)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "GRX_TWO_STREAMS=0" "GRX_TWO_STREAMS=1" "GRX_TWO_STREAMS=0" "GRX_TWO_STREAMS=1"; do
  echo "== [$cfg]"
  env $cfg timeout 600 python bench.py --only bfs,multi --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
s=j['config']['sections']['multi_source']
print('  step %.4f ms | multi fwd %.0f (%.3f) cold %.0f' % (j['ms_per_step'], s['forward_mteps'], s['forward_vs_single_source'], s['forward_cold_mteps']))"
done > gpurun_out/r6_c33_two_streams.txt 2>&1
cat gpurun_out/r6_c33_two_streams.txt
(GRX_TWO_STREAMS=1 timeout 900 python -m pytest tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x -k "not twitter" > gpurun_out/r6_c33_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c33_pytest.log)
tail -3 gpurun_out/r6_c33_pytest.log
