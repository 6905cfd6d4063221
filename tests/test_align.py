"""align_local_pts3d_to_global (reference fast3r/models/multiview_dust3r_module.py:427-549).
Its solver lives in the un-vendored, un-pinned `roma` package (not installable: no network).  The reference METHOD is run for real around a
stand-in for it (oracle/roma_stub.py; last section of this file); the solver's pin is an INDEPENDENT closed form of the same least-squares
problem -- Horn's unit-quaternion solution in float64 (oracle/align_pin.py) -- against which both the oracle restatement (Umeyama /
SVD) and the HIP path (raw fp64 moments + Jacobi SVD) are checked, on exact, noisy, mirrored and planar clouds; plus construction
properties of the oracle (CPU, below) and HIP path == oracle on the same inputs (GPU): quantile thresholds bit-exact vs
torch.quantile, transforms / points to fp32 noise."""
import ctypes

import pytest
import torch

from oracle.align_oracle import align_local_pts3d_to_global as align_oracle
from oracle.align_oracle import rigid_points_registration


def random_rotation(g):
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    if torch.det(q) < 0:
        q[:, 0] *= -1
    return q


def make_preds(n_views, B, H, W, seed, noise=0.02, shapes=None):
    """Local pointmaps = inverse similarity of the global ones + noise, so a known (R, t, s) maps local -> global."""
    g = torch.Generator().manual_seed(seed)
    preds, truth = [], []
    for i in range(n_views):
        h, w = (H, W) if shapes is None else shapes[i]
        glob = torch.randn(B, h, w, 3, generator=g) * 2.0 + torch.tensor([0.5, -1.0, 4.0])
        R, s, t = random_rotation(g), float(0.5 + 2 * torch.rand(1, generator=g)), torch.randn(3, generator=g)
        loc = ((glob - t) @ R) / s  # global = s * R loc + t
        loc = loc + noise * torch.randn(loc.shape, generator=g)
        conf = 1 + torch.exp(torch.randn(B, h, w, generator=g))
        preds.append({"pts3d_local": loc, "conf_local": conf.clone(), "pts3d_in_other_view": glob, "conf": conf})
        truth.append((R, s, t))
    return preds, truth


# ------------------------------------------------------------------------------------------------ CPU: oracle properties
def test_oracle_recovers_exact_similarity():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(500, 3, generator=g)
    R, s, t = random_rotation(g), 1.7, torch.tensor([0.3, -2.0, 5.0])
    Rh, th, sh = rigid_points_registration(x, s * x @ R.t() + t)
    assert torch.allclose(Rh, R, atol=1e-5) and abs(float(sh) - s) < 1e-5 and torch.allclose(th, t, atol=1e-4)
    # reflection-only data still yields a proper rotation (det = +1): special Procrustes
    Rr, _, _ = rigid_points_registration(x, x * torch.tensor([1.0, 1.0, -1.0]))
    assert abs(float(torch.det(Rr)) - 1.0) < 1e-5


def _clouds():
    """(name, x, y): generic noisy, exact, mirrored (the unconstrained optimum is a reflection), planar (rank 2), strongly noisy."""
    g = torch.Generator().manual_seed(42)
    out = []
    x = torch.randn(400, 3, generator=g, dtype=torch.float64) * torch.tensor([2.0, 1.0, 0.5], dtype=torch.float64)
    R, _ = torch.linalg.qr(random_rotation(g).double())  # orthonormal to float64 accuracy
    if torch.det(R) < 0:
        R[:, 0] *= -1
    y = 1.3 * x @ R.t() + torch.tensor([0.4, -1.0, 3.0], dtype=torch.float64)
    out.append(("exact", x, y))
    out.append(("noisy", x, y + 0.05 * torch.randn(400, 3, generator=g, dtype=torch.float64)))
    out.append(("very noisy", x, y + 1.0 * torch.randn(400, 3, generator=g, dtype=torch.float64)))
    out.append(("mirrored", x, (x * torch.tensor([1.0, 1.0, -1.0], dtype=torch.float64)) @ R.t() * 0.7 + 0.01 * torch.randn(400, 3, generator=g, dtype=torch.float64)))
    xp = x.clone()
    xp[:, 2] = 0.0
    out.append(("planar", xp, 2.0 * xp @ R.t() + 0.02 * torch.randn(400, 3, generator=g, dtype=torch.float64)))
    return out


def test_oracle_is_pinned_on_an_independent_closed_form():
    """The pin of this row (the reference's solver, `roma`, is not installable): Umeyama / SVD (oracle/align_oracle.py) vs Horn's
    unit-quaternion solution (oracle/align_pin.py) -- different derivations of the same least-squares problem, no shared code -- agree to
    1e-9 in float64 on generic, mirrored and planar clouds, and both recover an exact similarity."""
    from oracle.align_pin import horn_similarity
    for name, x, y in _clouds():
        Ru, tu, su = rigid_points_registration(x, y)
        Rh, th, sh = horn_similarity(x, y)
        assert abs(float(torch.det(Ru)) - 1) < 1e-9 and abs(float(torch.det(Rh)) - 1) < 1e-9, name
        assert float((Ru - Rh).abs().max()) < 1e-9 and float((tu - th).abs().max()) < 1e-8 and abs(float(su - sh)) < 1e-9, name
        res_u = float(((su * x @ Ru.t() + tu) - y).pow(2).sum())
        # neither beats the other, and perturbing the solution only makes the residual worse (it is the optimum, not just a fixed point)
        for k in range(3):
            ax = torch.zeros(3, dtype=torch.float64)
            ax[k] = 1e-3
            K = torch.tensor([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]], dtype=torch.float64)
            Rp = torch.linalg.matrix_exp(K) @ Ru
            assert float(((su * x @ Rp.t() + tu) - y).pow(2).sum()) >= res_u - 1e-12, name
        assert float((((su * 1.001) * x @ Ru.t() + tu) - y).pow(2).sum()) >= res_u - 1e-12, name
    name, x, y = _clouds()[0]
    R, t, s = rigid_points_registration(x, y)
    assert float(((s * x @ R.t() + t) - y).abs().max()) < 1e-9


def test_oracle_control_flow_and_errors():
    preds, truth = make_preds(2, 2, 8, 12, seed=1, noise=0.0)
    views = [{}, {"valid_mask": torch.zeros(2, 8, 12, dtype=torch.bool)}]  # view 1: nothing valid -> identity (:506-510)
    out = align_oracle([dict(p) for p in preds], views, min_conf_thr_percentile=85)
    assert out[0]["pts3d_local_aligned_to_global"].shape == (2, 8, 12, 3)
    assert torch.allclose(out[0]["pts3d_local_aligned_to_global"], preds[0]["pts3d_in_other_view"], atol=1e-4)
    assert torch.equal(out[1]["pts3d_local_aligned_to_global"], preds[1]["pts3d_local"])
    bad = [dict(preds[0])]
    del bad[0]["conf"]
    with pytest.raises(ValueError):
        align_oracle(bad, [{}])


def test_product_raises_like_reference_and_has_no_cpu_path(built_lib):
    from fast3r_amd import align_local_pts3d_to_global
    from fast3r_amd._lib import F3RError
    preds, _ = make_preds(1, 1, 4, 4, seed=2)
    bad = [dict(preds[0])]
    del bad[0]["pts3d_local"]
    with pytest.raises(ValueError, match="pts3d_local"):
        align_local_pts3d_to_global(bad, [{}])
    if not torch.cuda.is_available():  # CPU preds are uploaded to the GPU when there is one; without one there is no fallback
        with pytest.raises(F3RError):
            align_local_pts3d_to_global(preds, [{}])
    assert built_lib.f3r_align_workspace_bytes(3) == 3 * 40 * 8
    assert built_lib.f3r_align_local_to_global(0x1000, 0x1000, 0x1000, None, 0x1000, 0x1000, None, 0x1000, 8, 1, 16, ctypes.c_float(1.5), None) == -1


# ------------------------------------------------------------------------------------------------ GPU: HIP path == oracle
def _to_gpu(preds):
    return [{k: v.cuda() for k, v in p.items()} for p in preds]


@pytest.mark.gpu
@pytest.mark.parametrize("pct", [0, 50, 85, 100])
def test_align_matches_oracle(built_lib, pct):
    from fast3r_amd import align_local_pts3d_to_global
    preds, truth = make_preds(4, 2, 48, 64, seed=3 + pct)
    views = [{} for _ in preds]
    ref = align_oracle([dict(p) for p in preds], views, min_conf_thr_percentile=pct)
    got, tr = align_local_pts3d_to_global(_to_gpu(preds), views, min_conf_thr_percentile=pct, return_transforms=True)
    for i, (r, o) in enumerate(zip(ref, got)):
        a, b = o["pts3d_local_aligned_to_global"].cpu(), r["pts3d_local_aligned_to_global"]
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), (i, float((a - b).abs().max()))
        for bb in range(2):  # and the solution is the planted similarity up to the noise level
            R = tr[i][bb, :9].view(3, 3).cpu()
            assert torch.allclose(R, truth[i][0], atol=5e-3) and abs(float(tr[i][bb, 12]) - truth[i][1]) < 5e-3 * truth[i][1]


@pytest.mark.gpu
def test_quantile_threshold_is_bit_exact(built_lib):
    from fast3r_amd import _lib
    g = torch.Generator().manual_seed(9)
    n_prob, npix = 5, 3001
    conf = (1 + torch.exp(torch.randn(n_prob, npix, generator=g))).cuda()
    conf[1, :100] = conf[1, 0]  # ties around the cut
    pts = torch.randn(n_prob, npix, 3, generator=g).cuda()
    out, rts, thr = torch.empty_like(pts), torch.empty(n_prob, 13).cuda(), torch.empty(n_prob).cuda()
    ws = torch.empty(n_prob * 40, dtype=torch.float64).cuda()
    for q in (0.0, 0.123, 0.5, 0.85, 0.999, 1.0):
        _lib.check(_lib.lib().f3r_align_local_to_global(conf.data_ptr(), pts.data_ptr(), pts.data_ptr(), None, out.data_ptr(), rts.data_ptr(),
                                                       thr.data_ptr(), ws.data_ptr(), ws.numel() * 8, n_prob, npix, q,
                                                       torch.cuda.current_stream().cuda_stream))
        ref = torch.quantile(conf.cpu(), q, dim=1)
        assert torch.equal(thr.cpu(), ref), (q, thr.cpu(), ref)


@pytest.mark.gpu
def test_align_masks_fallbacks_and_mixed_resolution(built_lib):
    from fast3r_amd import align_local_pts3d_to_global
    shapes = [(32, 48), (16, 16), (32, 48)]
    preds, _ = make_preds(3, 1, 0, 0, seed=21, shapes=shapes)
    g = torch.Generator().manual_seed(5)
    vm0 = torch.rand(1, 32, 48, generator=g) > 0.4
    vm2 = torch.zeros(1, 32, 48, dtype=torch.bool)
    vm2[0, 0, :2] = True  # only 2 valid points -> identity
    views = [{"valid_mask": vm0}, {}, {"valid_mask": vm2}]
    ref = align_oracle([dict(p) for p in preds], views, min_conf_thr_percentile=70)
    got = align_local_pts3d_to_global(_to_gpu(preds), [{k: v.cuda() for k, v in vw.items()} for vw in views], min_conf_thr_percentile=70)
    for r, o in zip(ref, got):
        a, b = o["pts3d_local_aligned_to_global"].cpu(), r["pts3d_local_aligned_to_global"]
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    assert torch.equal(got[2]["pts3d_local_aligned_to_global"].cpu(), preds[2]["pts3d_local"])


@pytest.mark.gpu
def test_align_after_forward_via_lit_module(built_lib):
    """The demo flow (fast3r/viz/demo.py:420-461): forward -> lit_module.align_local_pts3d_to_global(preds, views, 85)."""
    from helpers import golden_model_inputs, load_golden, views_to
    from fast3r_amd import Fast3R, MultiViewDUSt3RLitModule
    fix = load_golden("tiny_3x64")
    enc, dec, head, sd, views = golden_model_inputs(fix)
    m = Fast3R(enc, dec, head).eval()
    m.load_state_dict(sd)
    lit = MultiViewDUSt3RLitModule.load_for_inference(m.cuda())
    with torch.no_grad():
        torch.manual_seed(fix["rng_seed"])
        preds = lit(views_to(views, "cuda"))
    lit.align_local_pts3d_to_global(preds, views, min_conf_thr_percentile=85)
    ref = align_oracle([{k: v.cpu() for k, v in p.items()} for p in preds], views, min_conf_thr_percentile=85)
    for r, o in zip(ref, preds):
        a, b = o["pts3d_local_aligned_to_global"].cpu(), r["pts3d_local_aligned_to_global"]
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())


@pytest.mark.gpu
def test_readme_flow_on_the_cpu_preds_inference_returns(built_lib):
    """The reference's documented two-step use (README.md:96-125): `out = inference(...)` hands back preds on the CPU (to_cpu,
    inference_multiview.py:92), then `lit.align_local_pts3d_to_global(out['preds'], out['views'], ...)` and
    `lit.estimate_camera_poses(out['preds'], ...)` run on them.  Same calls here: CPU preds go up, results come back on the CPU,
    and equal what the same functions give on the device-resident preds."""
    from helpers import golden_model_inputs, load_golden, views_to
    from fast3r_amd import Fast3R, MultiViewDUSt3RLitModule, inference
    fix = load_golden("tiny_3x64")
    enc, dec, head, sd, views = golden_model_inputs(fix)
    m = Fast3R(enc, dec, head).eval()
    m.load_state_dict(sd)
    lit = MultiViewDUSt3RLitModule.load_for_inference(m.cuda())
    torch.manual_seed(fix["rng_seed"])
    out = inference(views, lit, torch.device("cuda"), dtype=torch.float16, verbose=False)
    assert all(v.device.type == "cpu" for p in out["preds"] for v in p.values())
    lit.align_local_pts3d_to_global(out["preds"], out["views"], min_conf_thr_percentile=85)
    with torch.no_grad():
        torch.manual_seed(fix["rng_seed"])
        dev_preds = lit(views_to(views, "cuda"))
    lit.align_local_pts3d_to_global(dev_preds, views, min_conf_thr_percentile=85)
    for c, d in zip(out["preds"], dev_preds):
        assert c["pts3d_local_aligned_to_global"].device.type == "cpu"
        assert torch.equal(c["pts3d_local_aligned_to_global"], d["pts3d_local_aligned_to_global"].cpu())
    poses_c, focals_c = lit.estimate_camera_poses(out["preds"], niter_PnP=10)
    poses_d, focals_d = lit.estimate_camera_poses(dev_preds, niter_PnP=10)
    assert len(poses_c) == 1 and len(poses_c[0]) == 3 and poses_c[0][0].shape == (4, 4)
    assert all((a == b).all() for a, b in zip(poses_c[0], poses_d[0])) and focals_c == focals_d


@pytest.mark.gpu
def test_hip_alignment_matches_the_independent_pin(built_lib):
    """f3r_align_local_to_global (raw fp64 moments + Jacobi SVD on the GPU) against Horn's quaternion solution in float64 on the same
    clouds -- exact, noisy, mirrored (reflection case of the det correction), planar (rank 2): rotation / scale / translation to
    fp32-input accuracy, and the aligned points themselves."""
    from fast3r_amd import align_local_pts3d_to_global
    from oracle.align_pin import horn_similarity
    for name, x, y in _clouds():
        n = x.shape[0]  # 400 = 20 x 20 pixels
        xf, yf = x.float(), y.float()
        preds = [{"pts3d_local": xf.view(1, 20, 20, 3).cuda(), "conf_local": torch.ones(1, 20, 20).cuda(),
                  "pts3d_in_other_view": yf.view(1, 20, 20, 3).cuda(), "conf": torch.ones(1, 20, 20).cuda()}]
        got, tr = align_local_pts3d_to_global(preds, [{}], min_conf_thr_percentile=0, return_transforms=True)
        R, t, s = horn_similarity(xf, yf)  # the pin sees exactly the fp32 inputs the kernel sees
        Rg, tg, sg = tr[0][0, :9].view(3, 3).double().cpu(), tr[0][0, 9:12].double().cpu(), float(tr[0][0, 12])
        assert float((Rg - R).abs().max()) < 2e-6, (name, float((Rg - R).abs().max()))
        assert abs(sg - float(s)) < 2e-6 * float(s) and float((tg - t).abs().max()) < 1e-5, name
        ref = (s * xf.double() @ R.t() + t).view(1, 20, 20, 3)
        assert float((got[0]["pts3d_local_aligned_to_global"].double().cpu() - ref).abs().max()) < 2e-5 * float(ref.abs().max()), name


# ------------------------------------------------------------------------------------------------ the reference's method, run for real
# tests/golden/align_cases.pt (oracle/make_golden_align.py): MultiViewDUSt3RLitModule.align_local_pts3d_to_global imported from the
# reference checkout and run unmodified -- torch.quantile threshold, `conf >= thr & valid_mask`, both fall-backs, the application to every
# pixel -- with `roma` (not installable) replaced by oracle/roma_stub.py (Horn's quaternion closed form in float64).
def _align_cases():
    import os
    return torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "align_cases.pt"), weights_only=False)["cases"]


def _max_rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def test_oracle_matches_the_reference_method():
    for c in _align_cases():
        out = align_oracle([{k: v.clone() for k, v in p.items()} for p in c["preds"]], c["views"], min_conf_thr_percentile=c["pct"])
        for v, (o, ref) in enumerate(zip(out, c["aligned"])):
            assert o["pts3d_local_aligned_to_global"].shape == ref.shape
            assert _max_rel(o["pts3d_local_aligned_to_global"], ref) <= 2e-5, (c["name"], v)
    # the fall-backs were really taken: view 3 of pct85_b2 has one valid pixel -> identity (reference :501-506)
    c = [c for c in _align_cases() if c["name"] == "pct85_b2"][0]
    assert torch.equal(c["aligned"][3], c["preds"][3]["pts3d_local"])
    assert not torch.equal(c["aligned"][2], c["preds"][2]["pts3d_local"])  # 5 valid pixels, none confident: solved on valid_mask only (:488-497)


@pytest.mark.gpu
def test_hip_matches_the_reference_method(built_lib):
    from fast3r_amd import align_local_pts3d_to_global
    for c in _align_cases():
        views = [{k: v.cuda() for k, v in vw.items()} for vw in c["views"]]
        got = align_local_pts3d_to_global(_to_gpu(c["preds"]), views, min_conf_thr_percentile=c["pct"])
        for v, (o, ref) in enumerate(zip(got, c["aligned"])):
            a = o["pts3d_local_aligned_to_global"].cpu()
            assert a.shape == ref.shape and _max_rel(a, ref) <= 3e-5, (c["name"], v, _max_rel(a, ref))
