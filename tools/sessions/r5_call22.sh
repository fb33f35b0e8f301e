#!/bin/bash
# round 5, call 22: workgroups of the reset + seed launch and of the source level (forward search, LJ stand-in)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
bash tools/kt_fat.sh default
for n in 2 4 16; do bash tools/kt_fat.sh reset_wg_$n GRX_RESET_WG_PER_CU=$n; done
for n in 1 2 8; do bash tools/kt_fat.sh source_wg_$n GRX_SOURCE_WG_PER_CU=$n; done
} > gpurun_out/r5c22_kt.log 2>&1
cut -c1-330 gpurun_out/r5c22_kt.log
