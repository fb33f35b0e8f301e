#!/bin/bash
# round 4, call 22 (second session; 14.7 GPU-minutes left): the host-side schedule changes -- launch groups / first batches that
# follow the previous search, reset + seed of a forward search in one launch, the source kernel writing level 1's chunk map.
# Parity tests of the new paths first, then the same-process A/B, a bench line on these sources, and -- if time is left -- the
# kernel-trace + FETCH/WRITE passes of the BFS command (profiles/history/r4_bench_pmc.json for these sources) with a kernel timeline.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 260 python -m pytest -q -x --durations=12 -m gpu \
   tests/test_bfs_gpu.py::test_launch_groups_follow_the_previous_search \
   tests/test_relax_gpu.py::test_first_batch_follows_the_previous_search \
   tests/test_pr_gpu.py::test_first_batch_follows_the_previous_run \
   tests/test_bfs_gpu.py::test_chesapeake_golden tests/test_bfs_gpu.py::test_synthetic_goldens \
   tests/test_bfs_gpu.py::test_binned_forward_levels tests/test_bfs_gpu.py::test_binned_kernels_follow_the_previous_search \
   tests/test_bfs_gpu.py::test_direction_optimizing_switches_and_matches \
   tests/test_target_matrix_gpu.py::test_async_return_labels_are_final_when_the_call_returns \
   tests/test_sssp_gpu.py::test_pattern_graph_equals_bfs_depths tests/test_pr_gpu.py::test_reference_made_goldens \
   > gpurun_out/r4c22_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4c22_pytest.log)
el "pytest: $(tail -3 gpurun_out/r4c22_pytest.log | tr '\n' ' ')"
timeout 120 python tools/ab_r4b.py lj 20 bfs,sssp,pr 2>&1 | grep -v amdgpu.ids > gpurun_out/r4c22_ab_lj.log; el "ab lj"
timeout 200 python bench.py --only bfs,bfs_do,multi > gpurun_out/r4c22_bench.log 2> gpurun_out/r4c22_bench.err; echo "rc $?" >> gpurun_out/r4c22_bench.log
cp gpurun_out/bench_detail.json gpurun_out/r4c22_bench_detail.json 2>/dev/null; el "bench"
timeout 100 python tools/ab_r4b.py kron 10 bfs,pr 2>&1 | grep -v amdgpu.ids > gpurun_out/r4c22_ab_kron.log; el "ab kron"
# kernel trace + the two counter groups bench.py's `traffic` needs, BFS command only (the other classes keep their round-4 files)
PROF_GROUPS_FILTER='^FETCH_SIZE|^WRITE_SIZE' bash tools/profile.sh r4b_bfs python bench.py --only bfs,bfs_do --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r4c22_prof.log 2>&1
python tools/pmc_json.py gpurun_out/prof_r4b_bfs > gpurun_out/r4b_bench_pmc.json 2> gpurun_out/r4c22_pmc_json.err
cp gpurun_out/prof_r4b_bfs/summary.md gpurun_out/r4b_bfs_rocprofv3_summary.md 2>/dev/null
cp gpurun_out/prof_r4b_bfs/kt/p_kernel_stats.csv gpurun_out/r4b_bfs_kernel_stats.csv 2>/dev/null
(python tools/timeline.py gpurun_out/prof_r4b_bfs/kt -4 bfs_level_bin_kernel; python tools/timeline.py gpurun_out/prof_r4b_bfs/kt -3 bfs_level_bin_kernel; python tools/timeline.py gpurun_out/prof_r4b_bfs/kt -2 "bfs_level_kernel<"; python tools/timeline.py gpurun_out/prof_r4b_bfs/kt -3 "bfs_level_kernel<") > gpurun_out/r4c22_timeline.txt 2>&1
rm -rf gpurun_out/prof_r4b_bfs; el "profile"
cat gpurun_out/r4c22_ab_lj.log; tail -4 gpurun_out/r4c22_pytest.log; tail -c 1500 gpurun_out/r4c22_bench.log
