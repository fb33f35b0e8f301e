#!/bin/bash
# Copy what tools/final_round_r4.sh left under gpurun_out/ into the tracked profiles/ directory under stable names.
set -u
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
grep -h '^{' $G/final_bench.log | tail -1 > $P/r4_bench_line.json
cp $G/final_bench_detail.json $P/r4_bench_detail.json
cp $G/r4_bench_pmc.json $P/r4_bench_pmc.json
cp $G/r4_bfs_rocprofv3_summary.md $P/r4_bfs_rocprofv3_summary.md;   cp $G/r4_bfs_kernel_stats.csv $P/r4_bfs_kernel_stats.csv
cp $G/r4_ssspu_rocprofv3_summary.md $P/r4_sssp_unit_rocprofv3_summary.md; cp $G/r4_ssspu_kernel_stats.csv $P/r4_sssp_unit_kernel_stats.csv
cp $G/r4_sssp_rocprofv3_summary.md $P/r4_sssp_weighted_rocprofv3_summary.md; cp $G/r4_sssp_kernel_stats.csv $P/r4_sssp_weighted_kernel_stats.csv
cp $G/r4_ssspd_rocprofv3_summary.md $P/r4_sssp_weighted_dense_rocprofv3_summary.md; cp $G/r4_ssspd_kernel_stats.csv $P/r4_sssp_weighted_dense_kernel_stats.csv
cp $G/r4_pr_rocprofv3_summary.md $P/r4_pr_rocprofv3_summary.md;     cp $G/r4_pr_kernel_stats.csv $P/r4_pr_kernel_stats.csv
grep -h '^{' $G/final_bench_all.log > $P/r4_all_configs.jsonl
cp $G/final_prep_timing.log $P/r4_prep_timing_final.txt
{ echo "# tools/ab_relax.py on the final sources"; grep -hv amdgpu.ids $G/final_ab_relax_lj.log $G/final_ab_relax_kron.log | cut -c1-700; } > $P/r4_ab_relax_final_sources.txt
{ echo "# tools/ab_r4.py lj on the final sources"; grep -hv amdgpu.ids $G/final_ab_lj.log | cut -c1-420; } > $P/r4_ab_final_sources.txt
{ echo "== level 1 (89 k vertices, 31 M edges)"; grep -v amdgpu.ids $G/final_bin_debug_l1.log; echo "== level 2 (2.0 M vertices, 36 M edges)"; grep -v amdgpu.ids $G/final_bin_debug_l2.log; } | cut -c1-420 > $P/r4_binned_levels_timeline_lj.txt
cp $G/generic_bfs.log $P/r4_generic_operators_bfs_lj.txt; cp $G/generic_kernel_stats.md $P/r4_generic_operators_kernel_stats.md
grep -v amdgpu.ids $G/final_fuzz.log | tail -12 > $P/r4_fuzz_sweep.txt
cat $G/final_pytest_gpu.log | grep -E "passed|failed|rc " > $P/r4_pytest_gpu.log
tail -3 $G/final_smoke.log | grep -v amdgpu.ids > $P/r4_smoke.log
cp $G/pr_parity_c4.json $P/r4_pr_parity_c4.json 2>/dev/null
ls -la $P | grep r4_ | wc -l
