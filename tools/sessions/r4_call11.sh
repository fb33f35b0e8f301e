#!/bin/bash
# round 4, call 11: binned relaxation with batched slice loads in the sweep -- tests, trace, bin-count sweep
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_relax_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r4c11_pytest.log; cat gpurun_out/r4c11_pytest.log
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -- python $R/tools/ab_relax.py lj GRX_RBIN_MIN_EDGES=1048576 > $R/gpurun_out/r4c11_trace_run.log 2>&1; echo "rc $?"
python $R/tools/relax_trace.py /tmp/rt | tail -2 > $R/gpurun_out/r4c11_trace_lj.txt 2>&1; cat $R/gpurun_out/r4c11_trace_lj.txt
cd $R
for nbins in 320 448 640 900; do
  echo "== GRX_RBIN_BINS=$nbins"
  GRX_RBIN_BINS=$nbins timeout 600 python tools/ab_relax.py lj GRX_RBIN_MIN_EDGES=1048576 GRX_RBIN_MIN_EDGES=1048576,GRX_RBIN_PARTS=256 2>&1 | grep -v amdgpu.ids | cut -c1-600
done > gpurun_out/r4c11_bins_lj.log 2>&1; cat gpurun_out/r4c11_bins_lj.log
GRX_RBIN_BINS=448 timeout 600 python tools/ab_relax.py kron GRX_RBIN_MIN_EDGES=1048576 GRX_RBIN_MIN_EDGES=1048576,GRX_RBIN_PARTS=256 2>&1 | grep -v amdgpu.ids | cut -c1-600 > gpurun_out/r4c11_kron.log; cat gpurun_out/r4c11_kron.log
