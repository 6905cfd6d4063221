#!/bin/bash
# round 3, GPU call F: new tests only (asm kernel segments/state, generic head_dim, hd80 golden, real-size parity) with full output
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r3f; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_attn_asm_gpu.py -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -30 > $out/asm.log; tail -5 $out/asm.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "generic_head_dim or exact" 2>&1 | grep -v amdgpu.ids | tail -40 > $out/generic.log; tail -8 $out/generic.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -p no:cacheprovider -s -k "hd80 or anchored or fusion_only_n20" 2>&1 | grep -v amdgpu.ids | tail -60 > $out/e2e_hd80.log; grep -E "parity\]|passed|failed" $out/e2e_hd80.log | tail -30
timeout 1200 python -m pytest tests/test_realsize_gpu.py -q -p no:cacheprovider -s 2>&1 | grep -v amdgpu.ids | tail -80 > $out/realsize.log; grep -E "parity\]|passed|failed|Error|assert" $out/realsize.log | tail -30
