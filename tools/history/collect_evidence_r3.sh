#!/bin/bash
# Copy what tools/final_round_r3.sh left under gpurun_out/ into the tracked profiles/ directory under stable names.
set -u
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
grep -h '^{' $G/final_bench.log | tail -1 > $P/r3_bench_line.json
cp $G/r3_bench_pmc.json $P/r3_bench_pmc.json
cp $G/r3_bfs_rocprofv3_summary.md $P/r3_bfs_rocprofv3_summary.md;   cp $G/r3_bfs_kernel_stats.csv $P/r3_bfs_kernel_stats.csv
cp $G/r3_ssspu_rocprofv3_summary.md $P/r3_sssp_unit_rocprofv3_summary.md; cp $G/r3_ssspu_kernel_stats.csv $P/r3_sssp_unit_kernel_stats.csv
cp $G/r3_sssp_rocprofv3_summary.md $P/r3_sssp_weighted_rocprofv3_summary.md; cp $G/r3_sssp_kernel_stats.csv $P/r3_sssp_weighted_kernel_stats.csv
cp $G/r3_pr_rocprofv3_summary.md $P/r3_pr_rocprofv3_summary.md;     cp $G/r3_pr_kernel_stats.csv $P/r3_pr_kernel_stats.csv
{ echo "# tools/ab_r3.py on the final sources: scatter / sweep versions, per-level kernel times (us) + head"; grep -hv amdgpu.ids $G/final_ab_lj.log $G/final_ab_kron.log $G/final_ab_twitter.log | cut -c1-420; } > $P/r3_ab_final_sources.txt
{ echo "== level 1 (89 k vertices, 31 M edges)"; grep -v amdgpu.ids $G/final_bin_debug_l1.log; echo "== level 2 (2.0 M vertices, 36 M edges)"; grep -v amdgpu.ids $G/final_bin_debug_l2.log; } | cut -c1-420 > $P/r3_binned_levels_timeline_lj.txt
grep -h '^{' $G/final_bench_all.log > $P/r3_all_configs.jsonl
cp $G/generic_bfs.log $P/r3_generic_operators_bfs_lj.txt; cp $G/generic_kernel_stats.md $P/r3_generic_operators_kernel_stats.md
grep -v amdgpu.ids $G/final_fuzz.log | tail -12 > $P/r3_fuzz_sweep.txt
cat $G/final_pytest_gpu.log | grep -E "passed|failed|rc " > $P/r3_pytest_gpu.log
tail -3 $G/final_smoke.log | grep -v amdgpu.ids > $P/r3_smoke.log
ls -la $P | grep r3_ | wc -l
{ echo "# tools/ab_bu.py on the final sources: first vs second bottom-up body, knobs of the second, other sources"; cat $G/final_ab_bu_lj.log $G/final_ab_bu_kron.log $G/final_ab_bu_twitter.log | cut -c1-420; echo; echo "# per-wave phase clocks of the second body, LJ stand-in"; cat $G/final_bu_debug_lj.log; } > $P/r3_ab_bottomup_final_sources.txt
