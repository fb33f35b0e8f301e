#!/bin/bash
# call 15: per-wave phase clocks of the second bottom-up body (levels 1-3 of the LJ stand-in, level 1 of twitter)
mkdir -p gpurun_out
for l in 1 2 3; do GRX_BU_DEBUG=$l timeout 300 python tools/bu_debug.py lj 2>&1 | grep -v amdgpu.ids; done > gpurun_out/c15_bu_debug_lj.log
cat gpurun_out/c15_bu_debug_lj.log
for l in 1 2; do GRX_BU_DEBUG=$l timeout 300 python tools/bu_debug.py twitter 2>&1 | grep -v amdgpu.ids; done > gpurun_out/c15_bu_debug_twitter.log
cat gpurun_out/c15_bu_debug_twitter.log
