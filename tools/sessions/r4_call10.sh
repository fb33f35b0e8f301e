#!/bin/bash
# round 4, call 10: kernel trace of the binned relaxation (scatter vs sweep per level), parts / threshold sweep
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/rt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -- python $R/tools/ab_relax.py lj GRX_RBIN_MIN_EDGES=1048576 > $R/gpurun_out/r4c10_trace_run.log 2>&1; echo "rc $?"
python $R/tools/relax_trace.py /tmp/rt > $R/gpurun_out/r4c10_trace_lj.txt 2>&1; tail -c 2500 $R/gpurun_out/r4c10_trace_lj.txt
cd $R
timeout 600 python tools/ab_relax.py lj GRX_RBIN_MIN_EDGES=1048576,GRX_RBIN_PARTS=64 GRX_RBIN_MIN_EDGES=1048576,GRX_RBIN_PARTS=128 GRX_RBIN_MIN_EDGES=1048576,GRX_RBIN_PARTS=256 GRX_RBIN_MIN_EDGES=524288,GRX_RBIN_PARTS=256 GRX_RBIN_MIN_EDGES=262144,GRX_RBIN_PARTS=256 GRX_RBIN_MIN_EDGES=1048576,GRX_RBIN_PARTS=256,GRX_RBIN_SWEEP_WG_PER_CU=2 2>&1 | grep -v amdgpu.ids | cut -c1-700 > gpurun_out/r4c10_sweep_lj.log; cat gpurun_out/r4c10_sweep_lj.log
