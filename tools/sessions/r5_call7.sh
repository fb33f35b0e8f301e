#!/bin/bash
# round 5, call 7: per-kernel durations (rocprofv3 kernel trace) of the forward search for the bin cuts / sweep part rules
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
bash tools/kt_fat.sh default
bash tools/kt_fat.sh k15_nobalance GRX_SW2_BALANCE=0
bash tools/kt_fat.sh k15_items256_nobalance GRX_SW2_BALANCE=0 GRX_SW2_ITEMS=256
bash tools/kt_fat.sh k16 GRX_BIN_USHIFT=16
bash tools/kt_fat.sh k16_nobalance GRX_BIN_USHIFT=16 GRX_SW2_BALANCE=0
bash tools/kt_fat.sh k14 GRX_BIN_USHIFT=14
bash tools/kt_fat.sh k14_nobalance GRX_BIN_USHIFT=14 GRX_SW2_BALANCE=0
bash tools/kt_fat.sh r4cut GRX_BIN_UNIFORM=0
bash tools/kt_fat.sh r4cut_nobalance GRX_BIN_UNIFORM=0 GRX_SW2_BALANCE=0
GRX_LIB_PATH=$PWD/gunrock_amd/libgrx_r4.so bash tools/kt_fat.sh lib_r4
} > gpurun_out/r5c7_kt.log 2>&1
cat gpurun_out/r5c7_kt.log
