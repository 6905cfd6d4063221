"""Focal estimation (first half; reference multiview_dust3r_module.py:1081-1109 +
dust3r/post_process.py:77-142 "weiszfeld").  Parity is PINNED: tests/golden/focal_cases.pt holds inputs and outputs of the real
reference estimator (oracle/make_golden_focal.py); the oracle restatement must reproduce them (CPU), and the HIP path must match
both (GPU).  Tolerance: 5e-5 relative -- the reference sums the per-point terms in fp32 (torch's blocked order), the kernel in fp64."""
import math
import os

import pytest
import torch

from oracle import focal_oracle as FO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "focal_cases.pt")
REL_TOL = 5e-5


def cases():
    return torch.load(GOLD)


# ------------------------------------------------------------------------------------------------ CPU: oracle vs reference vectors
def test_oracle_reproduces_reference_vectors():
    cs = cases()
    assert len(cs) == 10
    for c in cs:
        f = FO.estimate_focal(c["pts3d"], c["conf"], min_conf_thr_percentile=c["percentile"])
        assert abs(f - c["focal"]) <= 1e-6 * abs(c["focal"]), (c["seed"], c["percentile"], f, c["focal"])
        assert abs(f - c["true_focal"]) < 0.02 * c["true_focal"]  # the estimator does its job on these scenes


def test_oracle_edge_cases():
    H, W = 8, 12
    pts = torch.ones(1, H, W, 3)
    conf = torch.ones(1, H, W)
    # no point survives an all-False mask -> default focal (post_process.py:102-105)
    f = FO.estimate_focal_knowing_depth_and_confidence_mask(pts, torch.tensor([[W / 2, H / 2]]), torch.zeros(1, H, W, dtype=torch.bool))
    assert abs(float(f) - max(H, W) / (2 * math.tan(math.radians(30)))) < 1e-5
    # z = 0 everywhere: every ratio is inf/nan -> 0 -> 0/0 focal = nan, like the reference arithmetic (no exception)
    pts0 = torch.zeros(1, H, W, 3)
    f0 = FO.estimate_focal_knowing_depth_and_confidence_mask(pts0, torch.tensor([[W / 2, H / 2]]), torch.ones(1, H, W, dtype=torch.bool))
    assert f0.shape == (1,)


# ------------------------------------------------------------------------------------------------ GPU: HIP vs reference / oracle
@pytest.mark.gpu
def test_hip_matches_reference_vectors(built_lib):
    from fast3r_amd import estimate_focal
    for c in cases():
        f = estimate_focal(c["pts3d"].cuda(), c["conf"].cuda(), min_conf_thr_percentile=c["percentile"])
        assert isinstance(f, float)
        assert abs(f - c["focal"]) <= REL_TOL * abs(c["focal"]), (c["seed"], c["percentile"], f, c["focal"])


@pytest.mark.gpu
def test_hip_batched_views_and_principal_point(built_lib):
    from fast3r_amd import estimate_focals
    cs = [c for c in cases() if (c["H"], c["W"]) == (64, 64)]
    g = torch.Generator().manual_seed(7)
    pts = torch.cat([cs[0]["pts3d"], cs[0]["pts3d"] * 1.0 + 0.01 * torch.randn(cs[0]["pts3d"].shape, generator=g), cs[0]["pts3d"].flip(2)])
    conf = torch.cat([cs[0]["conf"], cs[0]["conf"].flip(1), cs[0]["conf"]])
    pp = torch.tensor([30.5, 33.25])
    out = estimate_focals(pts.cuda(), conf.cuda(), pp=pp, min_conf_thr_percentile=37)
    assert out.shape == (3,) and out.dtype == torch.float32 and out.is_cuda
    for i in range(3):
        ref = FO.estimate_focal(pts[i:i + 1], conf[i:i + 1], pp=pp.view(1, 2), min_conf_thr_percentile=37)
        if math.isfinite(ref) and abs(ref) > 1e-3:
            assert abs(float(out[i]) - ref) <= REL_TOL * abs(ref), (i, float(out[i]), ref)


@pytest.mark.gpu
def test_hip_takes_cpu_tensors_and_rejects_bad_shapes(built_lib):
    """CPU inputs (what `inference()` returns after its to_cpu) are uploaded, solved on the GPU, and the result comes back on the CPU."""
    from fast3r_amd import estimate_focals
    g = torch.Generator().manual_seed(3)
    pts = torch.randn(2, 16, 16, 3, generator=g) + torch.tensor([0.0, 0.0, 4.0])
    conf = 1 + torch.rand(2, 16, 16, generator=g)
    on_cpu = estimate_focals(pts, conf)
    on_gpu = estimate_focals(pts.cuda(), conf.cuda())
    assert on_cpu.device.type == "cpu" and on_gpu.is_cuda and torch.equal(on_cpu, on_gpu.cpu())
    with pytest.raises(ValueError):
        estimate_focals(torch.zeros(1, 4, 4, 3).cuda(), torch.zeros(1, 4, 5).cuda())
