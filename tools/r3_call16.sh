#!/bin/bash
# call 16: L1 / L2 / wave counters of the direction-optimising search with the second bottom-up body
mkdir -p gpurun_out
PROF_GROUPS_FILTER='^TCP_|^TCC_HIT|^SQ_WAVES|^SQ_INSTS_LDS' bash tools/profile.sh c16_do python bench.py --only bfs --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/c16_prof.log 2>&1
cp gpurun_out/prof_c16_do/summary.md gpurun_out/c16_do_summary.md
rm -rf gpurun_out/prof_c16_do
grep -n "bfs_level_kernel" -A 40 gpurun_out/c16_do_summary.md | head -120
