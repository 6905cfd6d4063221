#!/bin/bash
# round 3, GPU call M: the fp32-equivalent mode on the LlamaDecoder / sharded models, general fp32 attention kernel, tensor gather of the outputs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3m; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -p no:cacheprovider -k "exact or sharded or rccl or inference_api" -x 2>&1 | grep -v amdgpu.ids | tail -30 > $O/pytest.log; tail -12 $O/pytest.log
