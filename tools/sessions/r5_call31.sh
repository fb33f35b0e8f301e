#!/bin/bash
# round 5, call 31 (experiment build): the thin-level kernel with 15 KB of LDS instead of 33 KB (the many-levels body's staging arrays
# cut out: mode 3 is broken in that build, GRX_MID=0 keeps it unused) -- does occupancy (4 -> 7 workgroups per CU) shorten thin levels?
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for g in lj deep kron; do
  KT_GRAPH=$g bash tools/kt_fat.sh ${g}_33KB GRX_MID=0
  KT_GRAPH=$g bash tools/kt_fat.sh ${g}_15KB GRX_MID=0 GRX_LIB_PATH=gunrock_amd/libgrx_smallmid.so
done
} > gpurun_out/r5c31_kt.log 2>&1
cut -c1-400 gpurun_out/r5c31_kt.log
