"""Camera poses on the GPU: `MultiViewDUSt3RLitModule.estimate_camera_poses`
(fast3r/models/multiview_dust3r_module.py:807-869) -- the README's advertised second step (README.md:112-125).

The reference moves every prediction to the CPU and runs, per sample and per view in thread pools, `estimate_focal` (Weiszfeld) and
`fast_pnp` (cv2.solvePnPRansac / SQPnP, for each of 100 tentative focals in 'individual' mode).  Here all views of all samples are one
launch of f3r_estimate_focal (when a shared focal is asked for) and one launch of f3r_estimate_poses; the return structure is the
reference's: (poses_c2w_all, estimated_focals_all) = per sample, per view, a 4x4 numpy array and a float (None when the solve failed).
The PnP solver is not OpenCV's (different algorithm, deterministic): same contract, poses agree to reprojection accuracy, not bit for bit.
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, work_device
from .focal import estimate_focals

N_GUESSED_FOCALS = 100  # init_im_poses.py:300
CONF_THR = 1.0          # multiview_dust3r_module.py:1045


N_ITER_MAX = 32         # hypotheses the kernel scores per view (f3r_pnp.hip N_HYP); niter_PnP above it is clamped


def estimate_poses(pts3d, conf, focal=None, pp=None, conf_thr=CONF_THR, n_focals=N_GUESSED_FOCALS, n_iter=N_ITER_MAX):
    """pts3d (n, H, W, 3), conf (n, H, W), focal None / float / (n,) tensor ->
    (cam_to_world (n, 4, 4) fp32, focal (n,) fp32 with NaN where the solve failed, inliers (n,) int32) on pts3d's device (CPU inputs
    are uploaded to the current ROCm device for the kernels).  n_iter: the RANSAC iteration count of the reference's call
    (cv2.solvePnPRansac(iterationsCount=niter_PnP)) = the number of sampled hypotheses scored per view, at most 32."""
    if pts3d.dim() != 4 or pts3d.shape[-1] != 3 or tuple(conf.shape) != tuple(pts3d.shape[:3]):
        raise ValueError(f"pts3d must be (n, H, W, 3) and conf (n, H, W); got {tuple(pts3d.shape)} and {tuple(conf.shape)}")
    n, H, W, _ = pts3d.shape
    home, dev = pts3d.device, work_device(pts3d, "pts3d")
    pts3d = pts3d.to(dev).float().contiguous()
    conf = conf.to(dev).float().contiguous()
    fin = None
    if focal is not None:
        fin = torch.as_tensor(focal, dtype=torch.float32, device=dev).reshape(-1)
        fin = fin.expand(n).contiguous() if fin.numel() == 1 else fin.contiguous()
        assert fin.numel() == n
    ppx, ppy = (W / 2, H / 2) if pp is None else (float(v) for v in torch.as_tensor(pp).reshape(-1)[:2])
    poses = torch.empty((n, 4, 4), dtype=torch.float32, device=dev)
    fout = torch.empty((n,), dtype=torch.float32, device=dev)
    inl = torch.empty((n,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().f3r_estimate_poses(ptr(pts3d), ptr(conf), ptr(fin), ptr(fout), ptr(poses), ptr(inl), n, H, W, float(conf_thr),
                                            float(ppx), float(ppy), int(n_focals), max(1, int(n_iter)), stream_ptr()), "f3r_estimate_poses")
    return poses.to(home), fout.to(home), inl.to(home)


def estimate_camera_poses(preds, views=None, niter_PnP=10, focal_length_estimation_method="individual"):
    """multiview_dust3r_module.py:807-869.  preds: list over views of dicts with 'pts3d_in_other_view' (B,H,W,3) and 'conf' (B,H,W).
    `niter_PnP` is OpenCV's RANSAC iteration BOUND (init_im_poses.py:335: it stops earlier once the consensus is good enough).  The kernel
    scores its sampled hypotheses in parallel, one per thread of a 32-thread group, so fewer than 32 would only idle threads and weaken the
    consensus on outlier-heavy views: the wrapper asks for max(niter_PnP, 32) (the kernel caps at 32); `estimate_poses(n_iter=...)` is the
    knob for fewer.  Views of different resolutions are solved per resolution group (the reference loops over views, :1038-1078)."""
    if focal_length_estimation_method not in ("individual", "first_view_from_global_head", "first_view_from_local_head"):
        raise ValueError(f"Unknown focal_length_estimation_method: {focal_length_estimation_method}")  # :843
    n_views = len(preds)
    B = len(preds[0]["pts3d_in_other_view"])  # :811
    f_b = None
    if focal_length_estimation_method != "individual":  # :826-848: one focal per sample, from view 0, 10th percentile
        if focal_length_estimation_method == "first_view_from_global_head":
            p0, c0 = preds[0]["pts3d_in_other_view"], preds[0]["conf"]
        else:
            p0, c0 = preds[0]["pts3d_local_aligned_to_global"], preds[0]["conf_local"]
        f_b = estimate_focals(p0, c0.reshape(p0.shape[:3]), min_conf_thr_percentile=10)  # (B,)
    groups = {}  # (H, W) -> view indices, in view order
    for v, p in enumerate(preds):
        groups.setdefault(tuple(p["pts3d_in_other_view"].shape[1:3]), []).append(v)
    poses_all = [[None] * n_views for _ in range(B)]
    focals_all = [[None] * n_views for _ in range(B)]
    for (H, W), vs in groups.items():
        pts = torch.stack([preds[v]["pts3d_in_other_view"] for v in vs], dim=1).reshape(B * len(vs), H, W, 3)  # sample-major
        conf = torch.stack([preds[v]["conf"] for v in vs], dim=1).reshape(B * len(vs), H, W)
        focal = None if f_b is None else f_b.repeat_interleave(len(vs))
        poses, fout, _ = estimate_poses(pts, conf, focal, n_iter=max(int(niter_PnP), 32))
        poses = poses.view(B, len(vs), 4, 4).cpu().numpy().astype(np.float64)
        fout = fout.view(B, len(vs)).cpu().tolist()
        for b in range(B):
            for j, v in enumerate(vs):
                poses_all[b][v] = poses[b, j]
                focals_all[b][v] = None if math.isnan(fout[b][j]) else fout[b][j]
    return poses_all, focals_all
