"""TEST INFRASTRUCTURE -- generates tests/golden/pose_cases.pt by running the REAL reference's camera-pose wrapper on CPU.

    python -m oracle.make_golden_pose          (build container only: needs /root/reference)

`MultiViewDUSt3RLitModule.estimate_camera_poses` (fast3r/models/multiview_dust3r_module.py:807-869) -> `estimate_cam_pose_one_sample`
(:1038-1078) -> `fast_pnp` (fast3r/dust3r/cloud_opt/init_im_poses.py:300-350) and the reference's `estimate_focal` (:1081-1109) are
imported from the reference checkout and run unmodified; the one thing replaced is OpenCV (not installable here): `cv2` resolves to
oracle/cv2_stub.py, an independent numpy RANSAC-PnP with OpenCV's call contract.  The fixture stores the synthetic scenes (pointmaps of
known cameras with noise and gross outliers, view 0 = the world frame as in a Fast3R prediction), the ground-truth cameras and what the
reference returned in both focal modes.  What this pins for the HIP path: the whole contract around the solver -- masks, focal
candidates and their selection, pose inversion, failure handling, return structure -- against code of the reference; the solver itself
can only be compared through its results (the poses), since the reference's is OpenCV's.
"""
import contextlib
import io
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cv2_stub, ref_loader  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pose_cases.pt")

# (seed, V views, B samples, H, W, focal, noise, gross outliers per view, pixels with conf == 1 (masked out by `conf > 1.0`) per view)
SCENES = [(10, 3, 1, 48, 64, 70.0, 0.002, 200, 0), (11, 4, 2, 64, 64, 55.0, 0.0, 0, 500), (12, 2, 1, 40, 56, 120.0, 0.005, 300, 100),
          (13, 3, 1, 64, 96, 90.0, 0.003, 800, 0)]


def random_rotation(g):
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    if torch.det(q) < 0:
        q[:, 0] *= -1
    return q


def make_view(g, H, W, f, noise, n_out, n_masked, anchor):
    """pointmap (H,W,3) of a pinhole camera (focal f, principal point (W/2, H/2)) expressed in the world frame, conf, cam_to_world"""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    z = 2 + 3 * torch.rand(H, W, generator=g, dtype=torch.float64)
    Xc = torch.stack([(xs - W / 2) * z / f, (ys - H / 2) * z / f, z], -1)
    if anchor:
        R, t = torch.eye(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64)
    else:
        R, t = random_rotation(g), torch.randn(3, generator=g, dtype=torch.float64)
    Xw = (Xc - t) @ R
    Xw = Xw + noise * torch.randn(Xw.shape, generator=g, dtype=torch.float64)
    if n_out:
        idx = torch.randperm(H * W, generator=g)[:n_out]
        Xw.view(-1, 3)[idx] += torch.randn(n_out, 3, generator=g, dtype=torch.float64)
    conf = 1.0 + torch.rand(H, W, generator=g) * 4 + 1e-3
    if n_masked:
        conf.view(-1)[torch.randperm(H * W, generator=g)[:n_masked]] = 1.0
    T = torch.eye(4, dtype=torch.float64)
    T[:3, :3] = R.t()
    T[:3, 3] = -R.t() @ t
    return Xw.float(), conf, T


def make_scene(seed, V, B, H, W, f, noise, n_out, n_masked):
    g = torch.Generator().manual_seed(seed)
    preds, gt = [], []
    for v in range(V):
        per_b = [make_view(g, H, W, f, noise, n_out, n_masked, anchor=(v == 0)) for _ in range(B)]
        preds.append({"pts3d_in_other_view": torch.stack([p[0] for p in per_b]), "conf": torch.stack([p[1] for p in per_b])})
        gt.append(torch.stack([p[2] for p in per_b]))
    return preds, gt


def load_reference_pose_api():
    """the reference's MultiViewDUSt3RLitModule with oracle/cv2_stub.py as cv2 (and dummies for the unrelated training-side imports)"""
    sys.modules["cv2"] = cv2_stub
    ref_loader._STUB_ROOTS = tuple(r for r in ref_loader._STUB_ROOTS if r != "cv2") + ("roma", "torchmetrics", "pl_bolts", "open3d", "rerun", "matplotlib", "trimesh", "viser", "wandb", "sklearn", "imageio")
    ref_loader.install()
    with contextlib.redirect_stdout(io.StringIO()):
        import fast3r.dust3r.cloud_opt.init_im_poses as ip
        import fast3r.models.multiview_dust3r_module as mm
    assert ip.cv2 is cv2_stub
    return mm.MultiViewDUSt3RLitModule


def main():
    warnings.filterwarnings("ignore")
    lit = load_reference_pose_api()
    cases = []
    for sc in SCENES:
        preds, gt = make_scene(*sc)
        out = {}
        for mode in ("first_view_from_global_head", "individual"):
            poses, focals = lit.estimate_camera_poses([dict(p) for p in preds], niter_PnP=100, focal_length_estimation_method=mode)
            out[mode] = dict(poses=[[np.asarray(m, dtype=np.float64) for m in s] for s in poses],
                             focals=[[None if f is None else float(f) for f in s] for s in focals])
        cases.append(dict(scene=sc, preds=preds, gt_cam2world=gt, reference=out))
        V, B = sc[1], sc[2]
        err = max(float(np.abs(out["first_view_from_global_head"]["poses"][b][v] - gt[v][b].numpy()).max()) for v in range(V) for b in range(B))
        print(sc, "max |pose - gt| =", f"{err:.2e}", "focals", [round(f, 2) for f in out["individual"]["focals"][0]])
    if "--check" in sys.argv:  # compare with the committed fixture instead of writing it
        old = torch.load(OUT, weights_only=False)["cases"]
        same = len(old) == len(cases) and all(
            all(torch.equal(pa[k], pb[k]) for pa, pb in zip(a["preds"], b["preds"]) for k in pb) and
            all(np.array_equal(x, y) for m in a["reference"] for sa, sb in zip(a["reference"][m]["poses"], b["reference"][m]["poses"]) for x, y in zip(sa, sb)) and
            all(a["reference"][m]["focals"] == b["reference"][m]["focals"] for m in a["reference"]) for a, b in zip(old, cases))
        print("pose fixture:", "bit-identical to the committed one" if same else "DIFFERS from the committed one")
        sys.exit(0 if same else 1)
    torch.save(dict(cases=cases, torch_version=torch.__version__), OUT)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB")


if __name__ == "__main__":
    main()
