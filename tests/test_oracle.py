"""The oracle (oracle/fast3r_oracle.py) against (a) the committed reference outputs under tests/golden/ -- runs
everywhere -- and (b) the live reference, wherever /root/reference exists (the build container)."""
import copy
import warnings

import pytest
import torch

from helpers import GOLDEN_CASES, golden_model_inputs, load_golden, rel_l2
from oracle import fast3r_oracle as O
from oracle.ref_loader import reference_available

TOL = 2e-5  # fp32 vs fp32: only summation-order differences (batched vs per-view encoding)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_oracle_matches_golden(name):
    fix = load_golden(name)
    enc, dec, head, sd, views = golden_model_inputs(fix)
    torch.manual_seed(fix["rng_seed"])
    out = O.forward(views, sd, enc, dec, head)
    assert len(out) == len(fix["preds"])
    for o, g in zip(out, fix["preds"]):
        assert set(o) == set(g)
        for k in g:
            assert o[k].shape == g[k].shape
            assert rel_l2(o[k], g[k]) < TOL, (name, k, rel_l2(o[k], g[k]))


@pytest.mark.parametrize("name", ["tiny_hot_3x64", "tiny_mixed"])
def test_oracle_fused_attention_equals_the_written_out_form(name):
    """ATTN_IMPL = "sdpa" (F.scaled_dot_product_attention, blocks.py:171-179 -- what bench.py's cpu_baseline and the 20 480-token parity test
    run) is the same function as the written-out softmax (blocks.py:158-169) that produced the goldens, to fp32 summation order."""
    fix = load_golden(name)
    enc, dec, head, sd, views = golden_model_inputs(fix)
    impl = O.ATTN_IMPL
    try:
        O.ATTN_IMPL = "sdpa"
        torch.manual_seed(fix["rng_seed"])
        out = O.forward(views, sd, enc, dec, head)
    finally:
        O.ATTN_IMPL = impl
    for o, g in zip(out, fix["preds"]):
        for k in g:
            assert rel_l2(o[k], g[k]) < TOL, (name, k, rel_l2(o[k], g[k]))


def test_oracle_image_ids_match_reference_recipe():
    fix = load_golden("tiny_3x64")
    torch.manual_seed(fix["rng_seed"])
    ids = O.random_image_ids(fix["batch"], len(fix["shapes"]))
    assert torch.equal(ids, fix["image_ids"])
    assert ids[0, 0] == 0 and len(set(ids[0].tolist())) == ids.shape[1]


def test_postprocess_invariants():
    x = torch.randn(2, 4, 5, 7)
    r = O.postprocess(x, ["exp", -float("inf"), float("inf")], ["exp", 1, float("inf")])
    d = x[:, :3].norm(dim=1)
    assert torch.allclose(r["pts3d"].norm(dim=-1), torch.expm1(d), rtol=1e-5, atol=1e-6)
    assert (r["conf"] >= 1).all()


@pytest.mark.skipif(not reference_available(), reason="reference checkout only exists in the build container")
def test_oracle_matches_live_reference():
    from oracle.ref_loader import load_reference
    from fast3r_amd.synthetic import make_views, synth_state_dict, tiny_args
    warnings.filterwarnings("ignore")
    Fast3R, inference = load_reference()
    enc, dec, head = tiny_args(embed_dim=192, num_heads=3, enc_depth=2, dec_depth=10)
    m = Fast3R(copy.deepcopy(enc), copy.deepcopy(dec), copy.deepcopy(head)).eval()
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=11)
    m.load_state_dict(sd, strict=True)
    views = make_views(4, 48, 80)
    torch.manual_seed(3)
    ref = inference(copy.deepcopy(views), m, torch.device("cpu"), dtype="32", verbose=False)["preds"]
    torch.manual_seed(3)
    ora = O.forward(views, sd, enc, dec, head)
    for o, g in zip(ora, ref):
        for k in g:
            assert rel_l2(o[k], g[k]) < TOL, (k, rel_l2(o[k], g[k]))


# ------------------------------------------------------------------------------------------------ fixtures are what the reference returns
_GENERATORS = ["oracle.make_golden", "oracle.make_golden_focal", "oracle.make_golden_align", "oracle.make_golden_images",
               pytest.param("oracle.make_golden_pose", marks=pytest.mark.skipif(__import__("os").environ.get("F3R_SLOW_CHECKS") != "1",
                                                                                reason="4.5 min of numpy RANSAC: set F3R_SLOW_CHECKS=1"))]


@pytest.mark.skipif(not reference_available(), reason="reference checkout only exists in the build container")
@pytest.mark.parametrize("module", _GENERATORS)
def test_committed_fixtures_are_reproduced_by_the_live_reference(module):
    """Every fixture under tests/golden/ is the output of reference code run here (forward pass, estimate_focal, align_local_pts3d_to_global,
    load_images, estimate_camera_poses; un-installable third-party packages behind the stand-ins in oracle/): `python -m <generator>
    --check` re-runs the reference in a fresh interpreter (the stand-ins replace modules at import time) and compares bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", module, "--check"], cwd=root, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIFFERS" not in r.stdout
