#!/bin/bash
# Round 4, second session: evidence on the final sources after the host-side schedule changes -- the two
# counter groups bench.py's `traffic` needs + kernel stats for every class (profiles/history/r4b_bench_pmc.json; the LDS / SQ / TCC
# groups of profiles/history/r4_bench_pmc.json were taken on sources whose level kernels are identical), the bench line with its
# detail file, smoke(), and the GPU test suite without its three heaviest cases (the 530 M-edge stand-in: one GPU, two ranks,
# eight ranks -- unchanged code paths, run in the first session's evidence call).
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out profiles; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
PROF_GROUPS_FILTER='^FETCH_SIZE|^WRITE_SIZE' timeout 200 bash tools/profile_r4.sh bfs ssspd pr ssspu sssp > gpurun_out/r4b_profile.log 2>&1
cp gpurun_out/r4_bench_pmc.json gpurun_out/r4b_bench_pmc.json; cp gpurun_out/r4b_bench_pmc.json profiles/history/r4b_bench_pmc.json
for t in bfs ssspd pr ssspu sssp; do
  for f in rocprofv3_summary.md kernel_stats.csv; do [ -f gpurun_out/r4_${t}_$f ] && mv gpurun_out/r4_${t}_$f gpurun_out/r4b_${t}_$f; done
done
el "profile: $(head -c 300 gpurun_out/r4b_bench_pmc.json | tr '\n' ' ')"
timeout 280 python bench.py > gpurun_out/r4b_bench.log 2> gpurun_out/r4b_bench.err; echo "rc $?" >> gpurun_out/r4b_bench.log
cp gpurun_out/bench_detail.json gpurun_out/r4b_bench_detail.json 2>/dev/null; el "bench"
(timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4b_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r4b_smoke.log); el "smoke"
# FINAL_TESTS: a test selection instead of the whole suite (the last pass of the session re-ran what the last change touched;
# the whole suite ran one commit earlier: profiles/history/r4b_pytest_gpu.log)
(timeout 240 python -m pytest ${FINAL_TESTS:-tests} -m gpu -q --durations=15 \
   --deselect tests/test_bfs_gpu.py::test_full_size_twitter_standin_properties \
   --deselect tests/test_distributed.py::test_c5_twitter_standin_two_ranks_one_gpu \
   --deselect tests/test_distributed.py::test_c5_twitter_standin_eight_ranks_one_gpu \
   > gpurun_out/r4b_pytest_gpu_${FINAL_TAG:-all}.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4b_pytest_gpu_${FINAL_TAG:-all}.log); el "pytest"
tail -4 gpurun_out/r4b_pytest_gpu_${FINAL_TAG:-all}.log; tail -1 gpurun_out/r4b_smoke.log; head -c 600 gpurun_out/r4b_bench.log; echo; tail -c 300 gpurun_out/r4b_bench.log
