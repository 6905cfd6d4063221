#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r3g; mkdir -p $out
timeout 900 python tools/diag_modes.py 16 48 100 2>&1 | grep -v amdgpu.ids | tee $out/diag_modes.log
