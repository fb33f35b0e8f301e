#!/bin/bash
# round 6, call 26: why a context cannot be created on libgrx_block.so inside pytest's subprocess
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
GRX_LIB_PATH=$PWD/gunrock_amd/libgrx_block.so python -c "
import gunrock_amd as gr, torch
print('has block', gr.has_block_async(), torch.cuda.is_available())
c = gr.multi_context_t(0); print('ctx ok')" 2>&1 | tail -3
echo "--- ldd"; ldd gunrock_amd/libgrx_block.so | grep -i "hip\|hsa" ; ldd gunrock_amd/libgrx.so | grep -i "hip\|hsa"
GRX_LIB_PATH=$PWD/gunrock_amd/libgrx_block.so timeout 600 python -m pytest tests/test_block_gpu.py -m gpu -x -q 2>&1 | tail -5
