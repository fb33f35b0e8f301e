#!/bin/bash
# Round-3 rocprofv3 evidence: kernel-trace stats + separate PMC passes for the bench workloads, then
# profiles/r3_bench_pmc.json (tools/pmc_json.py) and the per-target summaries.
#   tools/profile_r2.sh [bfs] [ssspu] [sssp] [pr]
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=${*:-bfs ssspu sssp pr}
DIRS=""
for t in $T; do
  case $t in
    bfs)   CMD="python bench.py --only bfs --no-cpu-baseline --steps 5 --warmup 2" ;;
    ssspu) CMD="python tools/run_algo.py ssspu road 2" ;;
    sssp)  CMD="python tools/run_algo.py sssp road 2" ;;
    pr)    CMD="python tools/run_algo.py pr kron 2" ;;
  esac
  PROF_SHORT=1 bash tools/profile.sh r3_$t $CMD > gpurun_out/prof_r3_$t.log 2>&1
  DIRS="$DIRS gpurun_out/prof_r3_$t"
  cp gpurun_out/prof_r3_$t/summary.md gpurun_out/r3_${t}_rocprofv3_summary.md
  cp gpurun_out/prof_r3_$t/kt/p_kernel_stats.csv gpurun_out/r3_${t}_kernel_stats.csv 2>/dev/null
done
python tools/pmc_json.py $DIRS > gpurun_out/r3_bench_pmc.json 2> gpurun_out/r3_pmc_json.err
for t in $T; do rm -rf gpurun_out/prof_r3_$t; done  # raw per-dispatch CSVs are large: summaries + JSON are kept
head -c 1500 gpurun_out/r3_bench_pmc.json; tail -3 gpurun_out/r3_pmc_json.err
