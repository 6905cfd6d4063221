#!/bin/bash
# round 3, GPU call P: SQPnP as the final solve of f3r_estimate_poses
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3p; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pnp.py tests/test_focal.py -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -30 > $O/pytest.log; tail -15 $O/pytest.log
timeout 300 python tools/diag_pose.py > $O/diag_pose.log 2>&1; grep scene $O/diag_pose.log
timeout 300 python tools/kernel_bench.py --what pnp > $O/pnp_bench.jsonl 2> $O/pnp.err; tail -3 $O/pnp_bench.jsonl
