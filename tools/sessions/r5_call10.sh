#!/bin/bash
# round 5, call 10: the slowest items of the sweep, phase by phase, uniform bins against the balanced cut
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "" "GRX_BIN_UNIFORM=0"; do
  for lv in 1 2; do echo "== level $lv $cfg"; env $cfg GRX_BIN_DEBUG=$lv timeout 90 python tools/bin_debug.py lj 2>&1 | grep -v amdgpu.ids | grep -A9 "^claim" | grep -v "xcc [1-7]:" | cut -c1-300; done
done > gpurun_out/r5c10_slow_items.log 2>&1
cat gpurun_out/r5c10_slow_items.log
