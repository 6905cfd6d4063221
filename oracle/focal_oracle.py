"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's focal estimation (first half):

  * `estimate_focal(pts3d_i, conf_i, pp=None, min_conf_thr_percentile=10)`, fast3r/models/multiview_dust3r_module.py:1081-1109
    (module-level function; the module itself cannot be imported here: LightningModule / torchmetrics / open3d), and
  * the "weiszfeld" branch of `estimate_focal_knowing_depth_and_confidence_mask`, fast3r/dust3r/post_process.py:77-142.

The second function IS importable from /root/reference (pure torch): oracle/make_golden_focal.py runs the real one on seeded inputs
and commits inputs + outputs as tests/golden/focal_cases.pt; tests/test_focal.py pins this restatement against those vectors, so
parity of this row is PINNED for the robust estimator and restated (8 lines of quantile + mask) for the wrapper.
"""
import math

import torch


def estimate_focal_knowing_depth_and_confidence_mask(pts3d, pp, conf_mask, min_focal=0.0, max_focal=math.inf, n_iter=100):
    """post_process.py:77-142, focal_mode="weiszfeld".  pts3d (B,H,W,3), pp (B,2) or (1,2), conf_mask (B,H,W) bool -> (1,) tensor."""
    B, H, W, _ = pts3d.shape
    # post_process.py:89-92: centred pixel grid, pixel (x=column, y=row) minus the principal point
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pixels = torch.stack([xs, ys], dim=-1).view(1, H, W, 2).to(pts3d.dtype) - pp.view(-1, 1, 1, 2)
    m = conf_mask.view(B, H, W)
    p = pts3d[m]          # :99 (N, 3)
    px = pixels.expand(B, H, W, 2)[m]  # :100 (N, 2)
    focal_base = max(H, W) / (2 * math.tan(math.radians(60) / 2))
    if p.numel() == 0:    # :102-105
        return torch.tensor([focal_base])
    xy_over_z = (p[..., :2] / p[..., 2:3]).nan_to_num(posinf=0, neginf=0)  # :121-123
    dot_xy_px = (xy_over_z * px).sum(dim=-1)   # :125
    dot_xy_xy = xy_over_z.square().sum(dim=-1)  # :126
    focal = dot_xy_px.mean() / dot_xy_xy.mean()  # :128
    for _ in range(n_iter):  # :131-136 iteratively re-weighted least squares
        dis = (px - focal * xy_over_z).norm(dim=-1)
        w = dis.clip(min=1e-8).reciprocal()
        focal = (w * dot_xy_px).sum() / (w * dot_xy_xy).sum()
    focal = focal.unsqueeze(0)
    return focal.clip(min=min_focal * focal_base, max=max_focal * focal_base)  # :140-142


def estimate_focal(pts3d_i, conf_i, pp=None, min_conf_thr_percentile=10):
    """multiview_dust3r_module.py:1081-1109.  pts3d_i (1,H,W,3), conf_i (1,H,W) -> python float."""
    B, H, W, THREE = pts3d_i.shape
    assert B == 1 and THREE == 3
    if pp is None:
        pp = torch.tensor((W / 2, H / 2)).view(1, 2)  # :1086
    conf_threshold = torch.quantile(conf_i.reshape(-1), min_conf_thr_percentile / 100.0)  # :1089-1093
    conf_mask = (conf_i >= conf_threshold).view(B, H, W)  # :1096-1097
    focal = estimate_focal_knowing_depth_and_confidence_mask(pts3d_i, pp.unsqueeze(0), conf_mask).ravel()  # :1106-1108
    return float(focal)
