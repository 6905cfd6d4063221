"""TEST INFRASTRUCTURE ONLY -- CPU restatement of `MultiViewDUSt3RLitModule.align_local_pts3d_to_global`
(fast3r/models/multiview_dust3r_module.py:427-549), the first consumer of the forward pass (fast3r/viz/demo.py:457-461).

The reference function cannot be imported here (LightningModule base, torchmetrics, open3d, pl_bolts: SURVEY.md section 8c) and its
arithmetic is in the third-party package `roma` (`roma.rigid_points_registration(x, y, compute_scaling=True)`, :509-511), which is
NOT vendored and NOT pinned (requirements.txt:26 just says `roma`) and is not installed in this image.  So this file restates
  * the reference's own control flow line by line (percentile mask :474-486, fall-backs :495-508, application :514-517), and
  * roma's published algorithm (roma/utils.py `rigid_points_registration` + `special_procrustes`: Umeyama, IEEE TPAMI 1991):
      xm, ym = mean(x), mean(y);  M = sum_k (y_k - ym)(x_k - xm)^T;  U S V^T = svd(M);
      R = U diag(1, 1, det(U) det(V)) V^T;  scale = sum(S * diag(1,1,det(U)det(V))) / sum_k |x_k - xm|^2;  t = ym - scale R xm
PARITY UNPINNED against the reference for this row (no importable reference, no golden vectors); pinned instead by construction
properties in tests/test_align.py: an exact similarity transform is recovered, and the HIP path equals this restatement.
"""
import torch


def rigid_points_registration(x: torch.Tensor, y: torch.Tensor):
    """y ~ scale * R x + t  (roma.rigid_points_registration(x, y, compute_scaling=True)); x, y: (M, 3)."""
    n = x.shape[0]
    xm, ym = x.mean(0, keepdim=True), y.mean(0, keepdim=True)
    xh, yh = x - xm, y - ym
    M = yh.t() @ xh
    U, S, Vh = torch.linalg.svd(M)
    d = torch.sign(torch.linalg.det(U) * torch.linalg.det(Vh))
    D = torch.ones(3, dtype=x.dtype)
    D[2] = d
    R = U @ torch.diag(D) @ Vh
    scale = (S * D).sum() / (xh ** 2).sum()
    t = ym[0] - scale * (R @ xm[0])
    return R, t, scale


def align_local_pts3d_to_global(preds, views, min_conf_thr_percentile=0):
    """multiview_dust3r_module.py:427-549 (single-threaded; the reference fans the (view, sample) pairs out to a thread pool)."""
    for pred in preds:  # :441-449
        for key in ("pts3d_local", "conf_local", "pts3d_in_other_view", "conf"):
            if key not in pred:
                raise ValueError(f"Key '{key}' not found in preds.")
    B = preds[0]["pts3d_local"].shape[0]
    for vi, (pred, view) in enumerate(zip(preds, views)):
        outs = []
        for b in range(B):
            pl, pg, cg = pred["pts3d_local"][b], pred["pts3d_in_other_view"][b], pred["conf"][b]
            Hc, Wc, _ = pl.shape
            valid = view["valid_mask"][b] if "valid_mask" in view else torch.ones_like(cg, dtype=torch.bool)   # :467-470
            thr = torch.quantile(cg.reshape(-1), min_conf_thr_percentile / 100.0)                                 # :476
            mask = ((cg >= thr) & valid).reshape(-1)                                                              # :479-482
            plf, pgf = pl.reshape(-1, 3), pg.reshape(-1, 3)
            x, y = plf[mask], pgf[mask]
            if x.shape[0] < 3:                                                                                    # :495-503
                mask = valid.reshape(-1)
                x, y = plf[mask], pgf[mask]
            if x.shape[0] < 3:                                                                                    # :506-510
                R, t, s = torch.eye(3, dtype=plf.dtype), torch.zeros(3, dtype=plf.dtype), 1.0
            else:
                R, t, s = rigid_points_registration(x, y)
            outs.append((s * (plf @ R.t()) + t).view(Hc, Wc, 3))                                                  # :514-517
        pred["pts3d_local_aligned_to_global"] = torch.stack(outs, dim=0)                                          # :547
    return preds
