#!/bin/bash
# round 5, call 32: workgroups of the thin-level kernel per CU (4 are resident; more made thin levels slower, call 31)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for g in lj deep; do
  for k in 4 3 2 1; do KT_GRAPH=$g bash tools/kt_fat.sh ${g}_wg$k GRX_LEVEL_WG_PER_CU=$k; done
done
} > gpurun_out/r5c32_kt.log 2>&1
cut -c1-400 gpurun_out/r5c32_kt.log
