#!/bin/bash
# Copy what tools/final_round.sh left under gpurun_out/ into the tracked profiles/ directory under stable names.
set -u
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
grep -h '^{' $G/final_bench.log | tail -1 > $P/r2_bench_line.json
cp $G/r2_bench_pmc.json $P/r2_bench_pmc.json
cp $G/r2_bfs_rocprofv3_summary.md $P/r2_bfs_rocprofv3_summary.md;   cp $G/r2_bfs_kernel_stats.csv $P/r2_bfs_kernel_stats.csv
cp $G/r2_ssspu_rocprofv3_summary.md $P/r2_sssp_unit_rocprofv3_summary.md; cp $G/r2_ssspu_kernel_stats.csv $P/r2_sssp_unit_kernel_stats.csv
cp $G/r2_sssp_rocprofv3_summary.md $P/r2_sssp_weighted_rocprofv3_summary.md; cp $G/r2_sssp_kernel_stats.csv $P/r2_sssp_weighted_kernel_stats.csv
cp $G/r2_pr_rocprofv3_summary.md $P/r2_pr_rocprofv3_summary.md;     cp $G/r2_pr_kernel_stats.csv $P/r2_pr_kernel_stats.csv
grep -v amdgpu.ids $G/final_ab_lj.log > $P/r2_ab_bfs_bodies_lj.txt
grep -v amdgpu.ids $G/final_ab_kron.log > $P/r2_ab_bfs_bodies_kron.txt
grep -v amdgpu.ids $G/final_ab_mid.log > $P/r2_ab_multilevel_bodies_road.txt
{ for a in bfs ssspu sssp; do echo "== $a, GRX_LB_STRICT=1 (chunked merge-path body on every level)"; grep -h "^algo" $G/final_road_${a}_strict1.log; done; } > $P/r2_ab_road_lb_strict.txt
{ echo "== level 1 (89 k vertices, 31 M edges)"; grep -v amdgpu.ids $G/final_bin_debug_l1.log; echo "== level 2 (2.0 M vertices, 36 M edges)"; grep -v amdgpu.ids $G/final_bin_debug_l2.log; } | cut -c1-400 > $P/r2_binned_levels_timeline_lj.txt
grep -h '^{' $G/final_bench_all.log > $P/r2_all_configs.jsonl
cat $G/final_pytest_gpu.log | grep -E "passed|failed|rc " > $P/r2_pytest_gpu.log
tail -3 $G/final_smoke.log | grep -v amdgpu.ids > $P/r2_smoke.log
ls -la $P | grep r2_ | wc -l
