"""Focal estimation on the GPU (first half) -- the first step of the README's "estimate camera poses"
(fast3r/models/multiview_dust3r_module.py:807-869): `estimate_focal(pts3d_i, conf_i, pp=None, min_conf_thr_percentile=10)`
with the reference's signature and return type (:1081-1109), plus a batched form over all views of a scene.

The reference computes, per view on CPU tensors: torch.quantile -> boolean indexing -> 100 Weiszfeld iterations of whole-array torch
ops (fast3r/dust3r/post_process.py:77-142).  Here a view is one workgroup of one kernel launch (f3r_post.hip::focal_kernel) and all
views of a scene run side by side.  The PnP pose solve that follows in the reference (cv2.solvePnPRansac, SQPnP) is fast3r_amd/pose.py
(`MultiViewDUSt3RLitModule.estimate_camera_poses`).  CPU tensors (what `inference()` returns) are moved to the GPU for the kernel and the
result comes back on the caller's device.
"""
import math

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, work_device

N_ITER = 100  # post_process.py:131


def estimate_focals(pts3d, conf, pp=None, min_conf_thr_percentile=10, n_iter=N_ITER, min_focal=0.0, max_focal=math.inf):
    """pts3d (n, H, W, 3), conf (n, H, W) on the GPU -> (n,) fp32 focal lengths (pixels), one per view."""
    if pts3d.dim() != 4 or pts3d.shape[-1] != 3 or tuple(conf.shape) != tuple(pts3d.shape[:3]):
        raise ValueError(f"pts3d must be (n, H, W, 3) and conf (n, H, W); got {tuple(pts3d.shape)} and {tuple(conf.shape)}")
    n, H, W, _ = pts3d.shape
    home, dev = pts3d.device, work_device(pts3d, "pts3d")
    pts3d = pts3d.to(dev).float().contiguous()
    conf = conf.to(dev).float().contiguous()
    if pp is None:
        ppx, ppy = W / 2, H / 2  # multiview_dust3r_module.py:1086
    else:
        ppx, ppy = (float(v) for v in torch.as_tensor(pp).reshape(-1)[:2])
    out = torch.empty((n,), dtype=torch.float32, device=dev)
    ws_bytes = _lib.lib().f3r_focal_workspace_bytes(n, H, W)
    ws = torch.empty((max(ws_bytes, 16) // 4,), dtype=torch.float32, device=dev)
    mx = 3.0e38 if math.isinf(max_focal) else float(max_focal)
    with torch.cuda.device(dev):
        check(_lib.lib().f3r_estimate_focal(ptr(pts3d), ptr(conf), ptr(out), None, ptr(ws), ws_bytes, n, H, W,
                                            float(min_conf_thr_percentile) / 100.0, float(ppx), float(ppy), int(n_iter), float(min_focal), mx,
                                            stream_ptr()), "f3r_estimate_focal")
    return out.to(home)


def estimate_focal(pts3d_i, conf_i, pp=None, min_conf_thr_percentile=10):
    """Drop-in for the reference function (multiview_dust3r_module.py:1081-1109): pts3d_i (1, H, W, 3), conf_i (1, H, W) -> float."""
    B = pts3d_i.shape[0]
    assert B == 1  # :1083
    return float(estimate_focals(pts3d_i, conf_i.reshape(pts3d_i.shape[:3]), pp, min_conf_thr_percentile)[0])
