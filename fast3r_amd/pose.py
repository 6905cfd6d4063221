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


def estimate_poses(pts3d, conf, focal=None, pp=None, conf_thr=CONF_THR, n_focals=N_GUESSED_FOCALS):
    """pts3d (n, H, W, 3), conf (n, H, W), focal None / float / (n,) tensor ->
    (cam_to_world (n, 4, 4) fp32, focal (n,) fp32 with NaN where the solve failed, inliers (n,) int32) on pts3d's device (CPU inputs
    are uploaded to the current ROCm device for the kernels)."""
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
                                            float(ppx), float(ppy), int(n_focals), stream_ptr()), "f3r_estimate_poses")
    return poses.to(home), fout.to(home), inl.to(home)


def estimate_camera_poses(preds, views=None, niter_PnP=10, focal_length_estimation_method="individual"):
    """multiview_dust3r_module.py:807-869.  preds: list over views of dicts with 'pts3d_in_other_view' (B,H,W,3) and 'conf' (B,H,W) on the
    GPU.  `niter_PnP` (OpenCV's RANSAC iteration count) has no counterpart in the deterministic solver and is accepted for compatibility."""
    if focal_length_estimation_method not in ("individual", "first_view_from_global_head", "first_view_from_local_head"):
        raise ValueError(f"Unknown focal_length_estimation_method: {focal_length_estimation_method}")  # :843
    n_views = len(preds)
    B = len(preds[0]["pts3d_in_other_view"])  # :811
    H, W = preds[0]["pts3d_in_other_view"].shape[1:3]
    if any(tuple(p["pts3d_in_other_view"].shape[1:3]) != (H, W) for p in preds):
        raise NotImplementedError("estimate_camera_poses: views of different resolutions -- call estimate_poses per resolution group")
    pts = torch.stack([p["pts3d_in_other_view"] for p in preds], dim=1).reshape(B * n_views, H, W, 3)  # sample-major
    conf = torch.stack([p["conf"] for p in preds], dim=1).reshape(B * n_views, H, W)
    focal = None
    if focal_length_estimation_method != "individual":  # :826-848: one focal per sample, from view 0, 10th percentile
        if focal_length_estimation_method == "first_view_from_global_head":
            p0, c0 = preds[0]["pts3d_in_other_view"], preds[0]["conf"]
        else:
            p0, c0 = preds[0]["pts3d_local_aligned_to_global"], preds[0]["conf_local"]
        f_b = estimate_focals(p0, c0.reshape(p0.shape[:3]), min_conf_thr_percentile=10)  # (B,)
        focal = f_b.repeat_interleave(n_views)
    poses, fout, _ = estimate_poses(pts, conf, focal)
    poses = poses.view(B, n_views, 4, 4).cpu().numpy().astype(np.float64)
    fout = fout.view(B, n_views).cpu().tolist()
    poses_all = [[poses[b, v] for v in range(n_views)] for b in range(B)]
    focals_all = [[(None if math.isnan(f) else f) for f in fout[b]] for b in range(B)]
    return poses_all, focals_all
