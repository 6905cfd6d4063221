#!/bin/bash
# round 3, GPU call T: the generic head_dim attention kernel after the two-query-block / prefetch rewrite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3t; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -p no:cacheprovider -k "head_dim or hd80 or generic" 2>&1 | grep -v amdgpu.ids | tail -12 > $O/pytest.log; tail -6 $O/pytest.log
timeout 300 python tools/kernel_bench.py --what attnhd --views 40 > $O/attn_hd.jsonl 2> $O/err.log; cat $O/attn_hd.jsonl
