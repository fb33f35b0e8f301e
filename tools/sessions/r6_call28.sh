#!/bin/bash
# round 6, call 28: entry thresholds of the many-levels body (GRX_MID_V / GRX_MID_E) on the deep stand-in and the 16 other sources
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "" "GRX_MID_V=32768 GRX_MID_E=65536" "GRX_MID_V=32768 GRX_MID_E=262144" "GRX_MID_V=65536 GRX_MID_E=524288 GRX_MID_EXIT_E=1048576"; do
  echo "== [$cfg]"
  env $cfg timeout 600 python bench.py --only bfs,multi,bfs_deep --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
s=j['config']['sections']
print('  step %.4f ms | multi fwd %.0f (%.3f) cold %.0f do %.0f | deep fwd %.4f ms do %.4f ms us/thin %.1f' % (j['ms_per_step'], s['multi_source']['forward_mteps'], s['multi_source']['forward_vs_single_source'], s['multi_source']['forward_cold_mteps'], s['multi_source'].get('do_mteps',0), s['bfs_deep']['fwd_ms'], s['bfs_deep']['do_ms'], s['bfs_deep']['us_per_thin_level']))"
done > gpurun_out/r6_c28_mid_thresholds.txt 2>&1
cat gpurun_out/r6_c28_mid_thresholds.txt
