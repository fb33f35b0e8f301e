#!/bin/bash
# Round 5, last session: evidence on the FINAL sources within what was left of the round's GPU budget (14 minutes).  As
# tools/final_round_r5.sh, cut down: counter passes (FETCH_SIZE / WRITE_SIZE, each its own rocprofv3 run) + kernel stats for the
# forward / direction-optimising BFS, PageRank (kron stand-in) and weighted SSSP on the LJ stand-in -- the two road-graph commands
# (thousands of launches under the profiler: minutes) are left out, their classes stay without `traffic` -- then the bench line
# with its detail file, smoke(), and the GPU suite without its five slowest cases (FINAL_HEAVY=1: all of them).
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out profiles; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
TARGETS=${FINAL_TARGETS:-bfs pr ssspd}
PROF_GROUPS_FILTER='^FETCH_SIZE|^WRITE_SIZE' timeout 420 bash tools/profile_r4.sh $TARGETS > gpurun_out/r5_profile.log 2>&1
cp gpurun_out/r4_bench_pmc.json gpurun_out/r5_bench_pmc.json; cp gpurun_out/r5_bench_pmc.json profiles/r5_bench_pmc.json
for t in $TARGETS; do
  for f in rocprofv3_summary.md kernel_stats.csv; do [ -f gpurun_out/r4_${t}_$f ] && mv gpurun_out/r4_${t}_$f gpurun_out/r5_${t}_$f; done
done
rm -f gpurun_out/r4_bench_pmc.json
el "profile: $(head -c 300 gpurun_out/r5_bench_pmc.json | tr '\n' ' ')"
timeout 400 python bench.py > gpurun_out/r5_bench.log 2> gpurun_out/r5_bench.err; echo "rc $?" >> gpurun_out/r5_bench.log
cp gpurun_out/bench_detail.json gpurun_out/r5_bench_detail.json 2>/dev/null; el "bench"
(timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r5_smoke.log); el "smoke"
DESEL="--deselect tests/test_bfs_gpu.py::test_full_size_twitter_standin_properties --deselect tests/test_distributed.py::test_c5_twitter_standin_two_ranks_one_gpu --deselect tests/test_distributed.py::test_c5_twitter_standin_eight_ranks_one_gpu --deselect tests/test_fuzz_gpu.py::test_fuzz_slice --deselect tests/test_cli.py::test_every_operator_combination_validates"
[ "${FINAL_HEAVY:-0}" = 1 ] && DESEL=""
(timeout 600 python -m pytest ${FINAL_TESTS:-tests} -m gpu -q -x --durations=10 $DESEL > gpurun_out/r5_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_pytest_gpu.log); el "pytest"
tail -4 gpurun_out/r5_pytest_gpu.log; tail -1 gpurun_out/r5_smoke.log; head -c 900 gpurun_out/r5_bench.log; echo; tail -c 300 gpurun_out/r5_bench.log
