#!/bin/bash
mkdir -p gpurun_out/g
export TMPDIR=/tmp
O=gpurun_out/g
( timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_kernels_gpu.py -m gpu -q -rA -p no:cacheprovider -k "dino or patchify or upsample or golden" 2>&1 | tail -150 > $O/pytest.log ); grep -E "dino" $O/pytest.log | grep parity; tail -4 $O/pytest.log; grep -E "FAILED|Error" $O/pytest.log | head
