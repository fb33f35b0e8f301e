#!/bin/bash
# Round-4 evidence in one GPU call, on the FINAL sources: rocprofv3 passes (kernel stats + PMC classes -> r4_bench_pmc.json,
# whose source_sha bench.py checks before attaching `traffic`), the GPU test suite, smoke(), the bench line (+ its detail file),
# every BASELINE config and target-matrix cell beside the reference's GPU path, preprocessing step times, generic operators,
# the block-asynchronous path's record, a fuzz sweep.  tools/collect_evidence_r4.sh copies what is to be judged into profiles/.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out profiles
export TMPDIR=/tmp
bash tools/profile_r4.sh > gpurun_out/final_profile.log 2>&1
cp gpurun_out/r4_bench_pmc.json profiles/history/r4_bench_pmc.json
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/final_pytest_gpu.log)
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/final_smoke.log)
timeout 900 python bench.py > gpurun_out/final_bench.log 2> gpurun_out/final_bench.err; echo "rc $?" >> gpurun_out/final_bench.log
cp gpurun_out/bench_detail.json gpurun_out/final_bench_detail.json
if [ "${FINAL_SHORT:-0}" != 1 ]; then
  timeout 1200 python tests/tools/bench_all.py bfs_lj ssspu_lj sssp_lj pr_lj bfs_kron ssspu_kron sssp_kron pr_kron bfs_road ssspu_road sssp_road bfs_twitter > gpurun_out/final_bench_all.log 2>&1
  timeout 300 python tools/prep_timing.py lj kron 2>&1 | grep -v amdgpu.ids > gpurun_out/final_prep_timing.log
  timeout 300 python tools/ab_r4.py lj 20 > gpurun_out/final_ab_lj.log 2>&1
  timeout 300 python tools/ab_relax.py lj "" > gpurun_out/final_ab_relax_lj.log 2>&1
  timeout 300 python tools/ab_relax.py kron "" > gpurun_out/final_ab_relax_kron.log 2>&1
  GRX_BIN_DEBUG=1 timeout 200 python tools/bin_debug.py lj > gpurun_out/final_bin_debug_l1.log 2>&1
  GRX_BIN_DEBUG=2 timeout 200 python tools/bin_debug.py lj > gpurun_out/final_bin_debug_l2.log 2>&1
  bash tools/bench_generic.sh > gpurun_out/final_generic.log 2>&1
  timeout 300 python tests/tools/fuzz_gpu.py 120 > gpurun_out/final_fuzz.log 2>&1
fi
tail -2 gpurun_out/final_pytest_gpu.log; tail -1 gpurun_out/final_smoke.log; tail -2 gpurun_out/final_bench.log | cut -c1-400
