#!/bin/bash
# Round 5: evidence on the FINAL sources -- the two counter groups bench.py's `traffic` needs + kernel stats for every class
# (profiles/r5_bench_pmc.json, source_sha inside), the bench line with its detail file, smoke(), the generic operators beside the
# engine, and the GPU test suite (FINAL_HEAVY=1: with the three 530 M-edge cases).
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out profiles; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
PROF_GROUPS_FILTER='^FETCH_SIZE|^WRITE_SIZE' timeout 400 bash tools/profile_r4.sh bfs ssspd pr ssspu sssp > gpurun_out/r5_profile.log 2>&1
cp gpurun_out/r4_bench_pmc.json gpurun_out/r5_bench_pmc.json; cp gpurun_out/r5_bench_pmc.json profiles/r5_bench_pmc.json
for t in bfs ssspd pr ssspu sssp; do
  for f in rocprofv3_summary.md kernel_stats.csv; do [ -f gpurun_out/r4_${t}_$f ] && mv gpurun_out/r4_${t}_$f gpurun_out/r5_${t}_$f; done
done
rm -f gpurun_out/r4_bench_pmc.json
el "profile: $(head -c 300 gpurun_out/r5_bench_pmc.json | tr '\n' ' ')"
timeout 400 python bench.py > gpurun_out/r5_bench.log 2> gpurun_out/r5_bench.err; echo "rc $?" >> gpurun_out/r5_bench.log
cp gpurun_out/bench_detail.json gpurun_out/r5_bench_detail.json 2>/dev/null; el "bench"
(timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r5_smoke.log); el "smoke"
timeout 300 bash tools/bench_generic.sh > gpurun_out/r5_generic.log 2>&1; el "generic"
DESEL="--deselect tests/test_bfs_gpu.py::test_full_size_twitter_standin_properties --deselect tests/test_distributed.py::test_c5_twitter_standin_two_ranks_one_gpu --deselect tests/test_distributed.py::test_c5_twitter_standin_eight_ranks_one_gpu"
[ "${FINAL_HEAVY:-0}" = 1 ] && DESEL=""
(timeout 900 python -m pytest ${FINAL_TESTS:-tests} -m gpu -q --durations=15 $DESEL > gpurun_out/r5_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_pytest_gpu.log); el "pytest"
tail -4 gpurun_out/r5_pytest_gpu.log; tail -1 gpurun_out/r5_smoke.log; head -c 700 gpurun_out/r5_bench.log; echo; tail -c 300 gpurun_out/r5_bench.log; cat gpurun_out/r5_generic.log | head -40
