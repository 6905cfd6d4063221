"""Camera poses (second half; reference multiview_dust3r_module.py:807-869,1038-1078 + fast_pnp,
dust3r/cloud_opt/init_im_poses.py:300-350).  The reference's solver is cv2.solvePnPRansac(flags=SOLVEPNP_SQPNP); OpenCV is not in this
image, so the named dependency's PUBLISHED algorithm is restated in fp64 (oracle/sqpnp.py: SQPnP, Terzakis & Lourakis 2020, inside
OpenCV's RANSAC structure, oracle/cv2_stub.py) and checked for what defines its result (global minimum of its cost: first section).  The
reference's WRAPPER around the solver runs for real on top of it (oracle/make_golden_pose.py -> tests/golden/pose_cases.pt).  The HIP
path finds its consensus set its own way (sampled DLT hypotheses, Gauss-Newton) and then -- since round 3 -- runs the SAME final solve,
SQPnP on the consensus set (fast3r_amd/csrc/f3r_sqpnp.h: built for the host here and checked against the restatement); it is compared
with what the reference wrapper returns at tolerances stated per scene (last section)."""

import numpy as np
import pytest
import torch

from oracle import pnp_oracle as PO


def random_rotation(g):
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    if torch.det(q) < 0:
        q[:, 0] *= -1
    return q


def make_scene(seed, H, W, f, noise, n_out, anchor=False):
    """world-frame pointmap seen by a camera (R, t: world -> camera) with focal f; returns pts (H,W,3) fp32, conf, cam_to_world."""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    z = 2 + 3 * torch.rand(H, W, generator=g, dtype=torch.float64)
    Xc = torch.stack([(xs - W / 2) * z / f, (ys - H / 2) * z / f, z], -1)
    R, t = random_rotation(g), torch.randn(3, generator=g, dtype=torch.float64)
    if anchor:  # view 0 of a Fast3R scene: the world frame IS its camera frame
        R, t = torch.eye(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64)
    Xw = (Xc - t) @ R  # Xc = R Xw + t
    Xw = Xw + noise * torch.randn(Xw.shape, generator=g, dtype=torch.float64)
    if n_out:
        idx = torch.randperm(H * W, generator=g)[:n_out]
        Xw.view(-1, 3)[idx] += torch.randn(n_out, 3, generator=g, dtype=torch.float64)
    T = torch.eye(4, dtype=torch.float64)
    T[:3, :3] = R.t()
    T[:3, 3] = -R.t() @ t
    conf = 1.0 + torch.rand(H, W, generator=g) * 4 + 1e-3
    return Xw.float(), conf, T


# ------------------------------------------------------------------------------------------------ CPU: SQPnP restatement
def _rand_problem(rng, n, noise):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    t = rng.standard_normal(3) + np.array([0.0, 0.0, 6.0])
    M = rng.standard_normal((n, 3)) * 1.5
    Xc = M @ q.T + t
    return M, Xc[:, :2] / Xc[:, 2:3] + noise * rng.standard_normal((n, 2)), q, t


def _host_sqpnp(tmp_path_factory):
    """the product's solver (f3r_sqpnp.h) compiled for the host by tests/csrc/sqpnp_host.cpp"""
    import ctypes
    import os
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path_factory.mktemp("sqpnp") / "libsqpnp_host.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(root, "tests", "csrc", "sqpnp_host.cpp")], check=True)
    lib = ctypes.CDLL(so)
    lib.sqpnp_host_solve.restype = ctypes.c_int
    dp = ctypes.POINTER(ctypes.c_double)

    def solve(M, xy, unit2=1.0):
        M, xy = np.ascontiguousarray(M, np.float64), np.ascontiguousarray(xy, np.float64)
        R, t, err = np.zeros(9), np.zeros(3), ctypes.c_double()
        ok = lib.sqpnp_host_solve(M.ctypes.data_as(dp), xy.ctypes.data_as(dp), len(M), ctypes.c_double(unit2), R.ctypes.data_as(dp),
                                  t.ctypes.data_as(dp), ctypes.byref(err))
        return (R.reshape(3, 3), t, err.value) if ok else None
    return solve


def test_product_sqpnp_equals_the_restatement(tmp_path_factory):
    """fast3r_amd/csrc/f3r_sqpnp.h (what f3r_pnp.hip runs on its consensus set), compiled for the host, against oracle/sqpnp.py on the same
    correspondences: clean, noisy and outlier-contaminated clouds of 4 .. 200 points agree to 5e-5 (both are the same published
    algorithm; only the eigen-solver and the null-space basis differ in implementation), also when the product conditions the world
    points the way the kernel does ((M - centroid) / sigma, thresholds rescaled by sigma^2)."""
    from oracle import sqpnp
    host = _host_sqpnp(tmp_path_factory)
    rng = np.random.default_rng(3)
    tol = 5e-5  # both stop their SQP iterations at a squared step of 1e-10, i.e. within ~1e-5 of the minimiser
    for trial in range(120):
        n = int(rng.integers(4, 200))
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        M = rng.standard_normal((n, 3)) * 4.5 + np.array([5.0, -2.0, 1.0])  # off-centre, not unit scale
        t = -q @ np.array([5.0, -2.0, 1.0]) + rng.standard_normal(3) + np.array([0.0, 0.0, 18.0])
        Xc = M @ q.T + t
        xy = Xc[:, :2] / Xc[:, 2:3] + (0.0 if trial % 2 == 0 else 0.01) * rng.standard_normal((n, 2))
        if trial % 7 == 0:
            k = max(1, n // 10)
            xy[:k] += rng.standard_normal((k, 2))
        want = sqpnp.solve(M, xy)
        got = host(M, xy)
        assert (want is None) == (got is None)
        if want is None:
            continue
        ts = 1 + np.abs(want[1]).max()
        assert np.abs(want[0] - got[0]).max() < tol and np.abs(want[1] - got[1]).max() < tol * ts, trial
        cen = M.mean(0)
        sig = np.sqrt(((M - cen) ** 2).sum(1).mean())
        cond = host((M - cen) / sig, xy, sig * sig)
        assert np.abs(want[0] - cond[0]).max() < tol and np.abs(want[1] - (sig * cond[1] - cond[0] @ cen)).max() < tol * ts, trial


def test_sqpnp_recovers_exact_cameras():
    """noise-free correspondences (4 .. 40 points): the restated solver returns the camera to fp64 accuracy"""
    from oracle import sqpnp
    rng = np.random.default_rng(0)
    for _ in range(40):
        M, xy, R, t = _rand_problem(rng, int(rng.integers(4, 40)), 0.0)
        Rs, ts, err = sqpnp.solve(M, xy)
        assert np.abs(Rs - R).max() < 1e-9 and np.abs(ts - t).max() < 1e-9 and err < 1e-10


def test_sqpnp_result_is_the_minimum_of_its_cost():
    """what pins an SQPnP implementation is its optimum: no rotation near the result, and none of 20 000 random rotations, has a lower
    r^T Omega r (t eliminated in closed form), and the result is a proper rotation that keeps the points in front of the camera"""
    from oracle import sqpnp
    rng = np.random.default_rng(1)
    for _ in range(6):
        M, xy, R, t = _rand_problem(rng, 30, 0.01)
        Rs, ts, err = sqpnp.solve(M, xy)
        assert abs(np.linalg.det(Rs) - 1) < 1e-12 and np.abs(Rs @ Rs.T - np.eye(3)).max() < 1e-12
        assert ((M @ Rs.T + ts)[:, 2] > 0).all()
        Om, P, _ = sqpnp.omega_matrix(M, xy)
        for scale in (1e-3, 2e-2, 0.3):
            for _k in range(500):
                w = rng.standard_normal(3) * scale
                th = np.linalg.norm(w)
                K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
                r = ((np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K) @ Rs).reshape(9)
                assert r @ Om @ r >= err - 1e-12
        q = rng.standard_normal((20000, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        a, b, c, d = q.T
        Rr = np.stack([1 - 2 * (c * c + d * d), 2 * (b * c - a * d), 2 * (b * d + a * c), 2 * (b * c + a * d), 1 - 2 * (b * b + d * d), 2 * (c * d - a * b),
                       2 * (b * d - a * c), 2 * (c * d + a * b), 1 - 2 * (b * b + c * c)], 1)
        assert (np.einsum("ni,ij,nj->n", Rr, Om, Rr) >= err - 1e-12).all()


def test_cv2_stand_in_follows_opencvs_ransac_contract():
    """solvePnPRansac of the stand-in: (ok, rvec (3,1), tvec (3,1), inliers (n,1) int32), gross outliers excluded from the inlier list, the
    final pose = SQPnP on the consensus set; fewer than 4 correspondences raise cv2.error"""
    from oracle import cv2_stub as cv2
    rng = np.random.default_rng(2)
    M, xy, R, t = _rand_problem(rng, 200, 0.0)
    K = np.array([[80.0, 0, 32], [0, 80.0, 24], [0, 0, 1]])
    uv = xy * 80.0 + np.array([32.0, 24.0])
    uv[:40] += rng.standard_normal((40, 2)) * 60  # gross outliers
    ok, rvec, tvec, inl = cv2.solvePnPRansac(M, uv, K, None, iterationsCount=100, reprojectionError=5, flags=cv2.SOLVEPNP_SQPNP)
    assert ok and rvec.shape == (3, 1) and tvec.shape == (3, 1) and inl.dtype == np.int32 and inl.shape[1] == 1
    assert (inl[:, 0] >= 40).sum() == 160 and (inl[:, 0] < 40).sum() <= 4
    assert np.abs(cv2.Rodrigues(rvec)[0] - R).max() < 1e-6 and np.abs(tvec[:, 0] - t).max() < 1e-6
    with pytest.raises(cv2.error):
        cv2.solvePnPRansac(M[:3], uv[:3], K, None)


SCENES = [(0, 48, 64, 70.0, 0.002, 300), (1, 64, 64, 55.0, 0.0, 0), (2, 40, 56, 120.0, 0.01, 500), (3, 96, 128, 100.0, 0.003, 2000)]


def grid_step(S):
    return 6.0 ** (1.0 / 99.0)  # ratio of neighbouring np.geomspace(S/2, 3S, 100) candidates


# ------------------------------------------------------------------------------------------------ CPU: the restatement does its job
@pytest.mark.parametrize("scene", SCENES)
def test_oracle_recovers_known_camera(scene):
    seed, H, W, f, noise, n_out = scene
    pts, conf, T = make_scene(*scene)
    fk, Tk = PO.fast_pnp(pts, f, conf > 1.0)
    assert fk == f and float((Tk - T).abs().max()) < 5e-3
    fs, Ts = PO.fast_pnp(pts, None, conf > 1.0)  # focal searched on the reference's grid: within one grid step of the truth
    assert f / grid_step(max(H, W)) ** 1.01 <= fs <= f * grid_step(max(H, W)) ** 1.01
    assert float((Ts[:3, :3] - T[:3, :3]).abs().max()) < 2e-2


def test_oracle_control_flow():
    pts, conf, T = make_scene(*SCENES[1])
    assert PO.fast_pnp(pts, 55.0, torch.zeros_like(conf, dtype=torch.bool)) == (None, None)  # < 4 points (:302-303)
    preds = [{"pts3d_in_other_view": pts[None], "conf": conf[None], "focal_length": 55.0},
             {"pts3d_in_other_view": pts[None], "conf": torch.ones_like(conf)[None]}]  # conf == 1 everywhere: nothing > 1.0 -> identity
    poses, focals = PO.estimate_cam_pose_one_sample(preds)
    assert float(np.abs(poses[0] - T.numpy()).max()) < 5e-3 and focals[0] == 55.0
    assert np.array_equal(poses[1], np.eye(4)) and focals[1] is None  # :1062-1064


# ------------------------------------------------------------------------------------------------ GPU: HIP vs ground truth / oracle
@pytest.mark.gpu
@pytest.mark.parametrize("scene", SCENES)
def test_hip_recovers_known_camera_and_matches_oracle(built_lib, scene):
    from fast3r_amd import estimate_poses
    seed, H, W, f, noise, n_out = scene
    pts, conf, T = make_scene(*scene)
    P, F, I = estimate_poses(pts[None].cuda(), conf[None].cuda(), focal=f)
    assert abs(float(F[0]) - f) < 1e-4 and int(I[0]) >= H * W - n_out - 50
    assert float((P[0].double().cpu() - T).abs().max()) < 5e-3
    fo, To = PO.fast_pnp(pts, f, conf > 1.0)
    assert float((P[0].double().cpu() - To).abs().max()) < 1e-4  # same algorithm incl. the final SQPnP: fp32 output of an fp64 solve
    P2, F2, _ = estimate_poses(pts[None].cuda(), conf[None].cuda())  # focal searched
    fs, Ts = PO.fast_pnp(pts, None, conf > 1.0)
    assert abs(float(F2[0]) - fs) <= 1e-4 * fs, (float(F2[0]), fs)
    assert float((P2[0].double().cpu() - Ts).abs().max()) < 1e-4


@pytest.mark.gpu
def test_hip_estimate_camera_poses_api(built_lib):
    """The reference's entry point: list of per-view pred dicts with a batch dimension -> (poses per sample per view, focals)."""
    from fast3r_amd import MultiViewDUSt3RLitModule
    scenes = [make_scene(10 + i, 48, 64, 70.0, 0.002, 100, anchor=(i == 0)) for i in range(3)]
    B = 2
    preds = [{"pts3d_in_other_view": torch.stack([s[0], s[0]]).cuda(), "conf": torch.stack([s[1], torch.ones_like(s[1])]).cuda()} for s in scenes]
    poses, focals = MultiViewDUSt3RLitModule.estimate_camera_poses(preds, niter_PnP=10, focal_length_estimation_method="individual")
    assert len(poses) == B and len(poses[0]) == 3 and poses[0][0].shape == (4, 4)
    for v, s in enumerate(scenes):
        # focal comes from the reference's 1.8 %-step grid: the rotation is unaffected, the translation absorbs f-error x depth
        assert float(np.abs(poses[0][v][:3, :3] - s[2].numpy()[:3, :3]).max()) < 1e-2 and 65.0 < focals[0][v] < 75.0
        assert float(np.abs(poses[0][v][:3, 3] - s[2].numpy()[:3, 3]).max()) < 0.15
        assert np.array_equal(poses[1][v], np.eye(4)) and focals[1][v] is None  # sample 1: conf == 1 -> no point > 1.0 -> identity
    # shared focal from view 0 of each sample (Weiszfeld, 10th percentile), then PnP with it
    preds1 = [{"pts3d_in_other_view": s[0][None].cuda(), "conf": s[1][None].cuda()} for s in scenes]
    poses1, focals1 = MultiViewDUSt3RLitModule.estimate_camera_poses(preds1, focal_length_estimation_method="first_view_from_global_head")
    assert all(abs(f - focals1[0][0]) < 1e-6 for f in focals1[0]) and abs(focals1[0][0] - 70.0) < 0.5
    for v, s in enumerate(scenes):
        assert float(np.abs(poses1[0][v] - s[2].numpy()).max()) < 3e-2
    with pytest.raises(ValueError):
        MultiViewDUSt3RLitModule.estimate_camera_poses(preds1, focal_length_estimation_method="nope")


# ------------------------------------------------------------------------------------------------ the reference's wrapper, run for real
# tests/golden/pose_cases.pt (oracle/make_golden_pose.py): the reference's MultiViewDUSt3RLitModule.estimate_camera_poses ->
# estimate_cam_pose_one_sample -> fast_pnp and its estimate_focal, imported from the reference checkout and run unmodified, with OpenCV (not
# installable here) replaced by oracle/cv2_stub.py -- OpenCV's RANSAC structure around an fp64 restatement of SQPnP (oracle/sqpnp.py), NOT
# the product's algorithm.
def _pose_cases():
    import os
    return torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pose_cases.pt"), weights_only=False)["cases"]


def test_reference_wrapper_fixture_recovers_the_known_cameras():
    """Sanity of the fixture itself: with the focal taken from view 0 (the README flow) the reference's wrapper around the stand-in solver
    finds the ground-truth cameras (the looser bound is the 40 x 56 scene, where the Weiszfeld focal is 1.5 % off and the pose absorbs it)."""
    for c in _pose_cases():
        V, B = c["scene"][1], c["scene"][2]
        ref = c["reference"]["first_view_from_global_head"]
        assert len(ref["poses"]) == B and len(ref["poses"][0]) == V
        for b in range(B):
            assert all(abs(f - ref["focals"][b][0]) < 1e-9 for f in ref["focals"][b])  # one focal per sample
            assert abs(ref["focals"][b][0] - c["scene"][5]) / c["scene"][5] < 0.03
            for v in range(V):
                assert float(np.abs(ref["poses"][b][v] - c["gt_cam2world"][v][b].numpy()).max()) < 3e-2


@pytest.mark.gpu
def test_hip_matches_the_reference_wrapper(built_lib):
    """README flow (focal_length_estimation_method='first_view_from_global_head'): same return structure, the shared focal equal to the
    reference's (its estimate_focal is the pinned Weiszfeld row), every pose within 2e-5 of what the reference's wrapper returned around
    the restated SQPnP (measured 2e-7 .. 8e-7: the fp32 rounding of the output): the two consensus sets -- OpenCV-style RANSAC over
    5-point samples there, sampled DLT hypotheses + Gauss-Newton here -- select the same points on these scenes (noise up to 5e-3 world
    units, up to 800 gross outliers, masked pixels), and on the same points the same solver (SQPnP) has one answer.  Both are 4e-3 .. 3e-2
    from the ground truth on the noisy scenes, identically.
    'individual' mode is NOT compared value by value: there the reference keeps the FIRST of its 100 focal candidates that reaches the
    maximum inlier count (`score > best[0]`, init_im_poses.py:341-342), i.e. the low end of a plateau that is wide at 5 px on small images
    (fixture: 55 for a true 70), while the product breaks ties by reprojection cost (docs/rows_f.md); only structure and failure
    handling are compared in that mode."""
    from fast3r_amd import MultiViewDUSt3RLitModule
    for c in _pose_cases():
        V, B = c["scene"][1], c["scene"][2]
        preds = [{k: v.cuda() for k, v in p.items()} for p in c["preds"]]
        poses, focals = MultiViewDUSt3RLitModule.estimate_camera_poses(preds, niter_PnP=100, focal_length_estimation_method="first_view_from_global_head")
        ref = c["reference"]["first_view_from_global_head"]
        assert len(poses) == B and all(len(p) == V for p in poses) and len(focals) == B
        for b in range(B):
            for v in range(V):
                assert isinstance(poses[b][v], np.ndarray) and poses[b][v].shape == (4, 4)
                assert abs(focals[b][v] - ref["focals"][b][v]) <= 1e-4 * ref["focals"][b][v], (focals[b][v], ref["focals"][b][v])
                d = float(np.abs(poses[b][v] - ref["poses"][b][v]).max())
                assert d < 2e-5, (c["scene"], b, v, d)
        poses_i, focals_i = MultiViewDUSt3RLitModule.estimate_camera_poses(preds, niter_PnP=100, focal_length_estimation_method="individual")
        ref_i = c["reference"]["individual"]
        assert len(poses_i) == len(ref_i["poses"]) and all(len(a) == len(b_) for a, b_ in zip(poses_i, ref_i["poses"]))
        grid = np.geomspace(max(c["scene"][3], c["scene"][4]) / 2, max(c["scene"][3], c["scene"][4]) * 3, 100)
        for b in range(B):
            for v in range(V):
                assert (focals_i[b][v] is None) == (ref_i["focals"][b][v] is None)
                assert np.abs(grid - focals_i[b][v]).min() < 1e-3 * focals_i[b][v]  # one of the reference's candidates (init_im_poses.py:313-314)
                assert float(np.abs(poses_i[b][v][:3, :3] - c["gt_cam2world"][v][b].numpy()[:3, :3]).max()) < 2e-2
