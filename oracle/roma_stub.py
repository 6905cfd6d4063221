"""TEST INFRASTRUCTURE ONLY -- a stand-in for the one `roma` function the reference's `align_local_pts3d_to_global` calls
(fast3r/models/multiview_dust3r_module.py:509-511): `roma.rigid_points_registration(x, y, compute_scaling=True) -> (R, t, s)`.

`roma` is un-vendored and not installable here.  oracle/make_golden_align.py imports the reference's module with THIS file as `roma` and
runs the reference's own method -- the torch.quantile threshold, the `conf >= thr & valid_mask` selection, both fall-backs (< 3 points),
the application to every pixel, the thread pool, the new `pts3d_local_aligned_to_global` key -- so that everything around the solver is
the reference's code.  The solver here is Horn's unit-quaternion closed form in float64 (oracle/align_pin.py): the same least-squares
problem roma solves by SVD (`special_procrustes`), a different derivation from the Kabsch / Umeyama form of the oracle and the kernel.
Results are returned in the dtype of x, as roma does."""
import torch

from oracle.align_pin import horn_similarity


def rigid_points_registration(x, y, weights=None, compute_scaling=False):
    assert weights is None, "the reference passes no weights (multiview_dust3r_module.py:509-511)"
    R, t, s = horn_similarity(x, y)
    if not compute_scaling:  # rigid: rotation from the same quaternion, translation between the centroids
        t = y.double().mean(0) - R @ x.double().mean(0)
        return R.to(x.dtype), t.to(x.dtype)
    return R.to(x.dtype), t.to(x.dtype), s.to(x.dtype)


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    return 0
