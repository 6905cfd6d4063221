"""`align_local_pts3d_to_global` on the GPU: the step every consumer of the forward pass runs
next (fast3r/viz/demo.py:457-461), same signature and side effect as the reference
(fast3r/models/multiview_dust3r_module.py:427-549: adds `pts3d_local_aligned_to_global` to every pred dict).

The reference loops over (view, sample) pairs in a Python thread pool, calling torch.quantile + boolean indexing +
roma.rigid_points_registration on each; here all pairs are one batched launch of three HIP kernels (f3r_post.hip).
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, work_device

_REQUIRED = ("pts3d_local", "conf_local", "pts3d_in_other_view", "conf")


def align_local_pts3d_to_global(preds, views, min_conf_thr_percentile=0, return_transforms=False):
    for pred in preds:
        for key in _REQUIRED:
            if key not in pred:
                msg = f"Key '{key}' not found in preds." if key != "conf" else "Key 'conf' (global head confidence) not found in preds."
                raise ValueError(msg)  # same type and text as the reference (:441-449)
    if len(preds) == 0:
        return preds
    home = preds[0]["pts3d_local"].device  # CPU preds (what `inference()` returns) are uploaded here and the result comes back there
    dev = work_device(preds[0]["pts3d_local"], "preds")
    B = preds[0]["pts3d_local"].shape[0]
    q = float(min_conf_thr_percentile) / 100.0
    # views of one resolution are batched into one launch; mixed resolutions give one launch per distinct (H, W)
    groups = {}
    for i, p in enumerate(preds):
        groups.setdefault(tuple(p["pts3d_local"].shape[1:3]), []).append(i)
    transforms = [None] * len(preds)
    for (H, W), idxs in groups.items():
        npix, n_prob = H * W, len(idxs) * B
        loc = torch.stack([preds[i]["pts3d_local"].float() for i in idxs]).to(dev).contiguous()          # (n, B, H, W, 3)
        glo = torch.stack([preds[i]["pts3d_in_other_view"].float() for i in idxs]).to(dev).contiguous()
        conf = torch.stack([preds[i]["conf"].float() for i in idxs]).to(dev).contiguous()                  # (n, B, H, W)
        valid = None
        if any("valid_mask" in views[i] for i in idxs):
            valid = torch.stack([views[i]["valid_mask"].to(dev) if "valid_mask" in views[i]
                                 else torch.ones((B, H, W), dtype=torch.bool, device=dev) for i in idxs]).to(torch.uint8).contiguous()
        out = torch.empty_like(loc)
        rts = torch.empty((n_prob, 13), dtype=torch.float32, device=dev)
        ws_bytes = _lib.lib().f3r_align_workspace_bytes(n_prob)
        ws = torch.empty((ws_bytes // 8,), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            check(_lib.lib().f3r_align_local_to_global(ptr(conf), ptr(loc), ptr(glo), ptr(valid), ptr(out), ptr(rts), None, ptr(ws),
                                                       ws_bytes, n_prob, npix, q, stream_ptr()), "f3r_align_local_to_global")
        rts, out = rts.view(len(idxs), B, 13).to(home), out.to(home)
        for j, i in enumerate(idxs):
            preds[i]["pts3d_local_aligned_to_global"] = out[j]
            transforms[i] = rts[j]
    if return_transforms:
        return preds, transforms
    return preds
