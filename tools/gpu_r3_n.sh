#!/bin/bash
# round 3, GPU call N: partial-workgroup form of the hand-scheduled attention kernel + the sharded path after the dist.py changes; N = 101 views of 384x512
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3n; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attn_asm_gpu.py tests/test_e2e_gpu.py -q -p no:cacheprovider -k "asm or sharded or rccl" -x 2>&1 | grep -v amdgpu.ids | tail -30 > $O/pytest.log; tail -8 $O/pytest.log
timeout 600 python tools/kernel_bench.py --what attnsel --attn-dtypes fp16 --views 101 --tokens-per-view 768 --sels 2,1 > $O/attn_n101_768.jsonl 2> $O/attn.err; tail -4 $O/attn_n101_768.jsonl; tail -3 $O/attn.err
