#!/bin/bash
# Round 6: evidence on the FINAL sources -- rocprofv3 kernel stats + separate counter passes (FETCH_SIZE / WRITE_SIZE, and the SQ LDS
# group for the BFS: bank conflicts of the fat levels) for the forward / direction-optimising BFS, PageRank (kron stand-in), weighted
# SSSP on the LJ stand-in and BOTH road searches (the weighted one is a single launch now: it fits a profiler pass), then
# profiles/r6_bench_pmc.json (tools/pmc_json.py, tied to the source sha), the bench line with its detail file, smoke(), the GPU suite.
#   FINAL_TARGETS="bfs pr ssspd sssp ssspu"   FINAL_SKIP_TESTS=1
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out profiles; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
TARGETS=${FINAL_TARGETS:-bfs pr ssspd sssp ssspu}
PROF_GROUPS_FILTER='^FETCH_SIZE|^WRITE_SIZE|^SQ_INSTS_LDS|^SQ_WAVES' timeout 1500 bash tools/profile_r4.sh $TARGETS > gpurun_out/r6_profile.log 2>&1
cp gpurun_out/r4_bench_pmc.json gpurun_out/r6_bench_pmc.json; cp gpurun_out/r6_bench_pmc.json profiles/r6_bench_pmc.json
for t in $TARGETS; do
  for f in rocprofv3_summary.md kernel_stats.csv; do [ -f gpurun_out/r4_${t}_$f ] && mv gpurun_out/r4_${t}_$f gpurun_out/r6_${t}_$f; done
done
rm -f gpurun_out/r4_bench_pmc.json
el "profile: $(head -c 200 gpurun_out/r6_bench_pmc.json | tr '\n' ' ')"
timeout 900 python bench.py > gpurun_out/r6_bench.log 2> gpurun_out/r6_bench.err; echo "rc $?" >> gpurun_out/r6_bench.log
cp gpurun_out/bench_detail.json gpurun_out/r6_bench_detail.json 2>/dev/null; el "bench"
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r6_smoke.log); el "smoke"
if [ "${FINAL_SKIP_TESTS:-0}" != 1 ]; then
  (timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 > gpurun_out/r6_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_pytest_gpu.log); el "pytest"
  tail -4 gpurun_out/r6_pytest_gpu.log
fi
tail -1 gpurun_out/r6_smoke.log; head -c 1200 gpurun_out/r6_bench.log; echo; tail -c 700 gpurun_out/r6_bench.log
