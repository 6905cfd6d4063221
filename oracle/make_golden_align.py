"""TEST INFRASTRUCTURE -- generates tests/golden/align_cases.pt by running the REAL reference's `align_local_pts3d_to_global` on CPU.

    python -m oracle.make_golden_align          (build container only: needs /root/reference)

`MultiViewDUSt3RLitModule.align_local_pts3d_to_global` (fast3r/models/multiview_dust3r_module.py:427-549) is imported from the reference
checkout and run unmodified; `roma` (un-vendored, not installable) resolves to oracle/roma_stub.py (Horn's quaternion closed form in
float64).  Cases: several percentiles, B = 1 / 2, a `valid_mask`, a view whose confident set is too small (first fall-back: valid_mask
only), a view with fewer than 3 valid points (second fall-back: identity).  The fixture stores inputs and the reference's outputs.
"""
import contextlib
import io
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_loader, roma_stub  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "align_cases.pt")


def random_rotation(g):
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    if torch.det(q) < 0:
        q[:, 0] *= -1
    return q


def make_preds(n_views, B, H, W, seed, noise):
    g = torch.Generator().manual_seed(seed)
    preds = []
    for _ in range(n_views):
        glob = torch.randn(B, H, W, 3, generator=g) * 2.0 + torch.tensor([0.5, -1.0, 4.0])
        R, s, t = random_rotation(g), float(0.5 + 2 * torch.rand(1, generator=g)), torch.randn(3, generator=g)
        loc = ((glob - t) @ R) / s + noise * torch.randn(glob.shape, generator=g)
        conf = 1 + torch.exp(torch.randn(B, H, W, generator=g))
        preds.append({"pts3d_local": loc, "conf_local": conf.clone(), "pts3d_in_other_view": glob, "conf": conf})
    return preds


def load_reference_align():
    sys.modules["roma"] = roma_stub
    ref_loader._STUB_ROOTS = tuple(ref_loader._STUB_ROOTS) + ("torchmetrics", "pl_bolts", "open3d", "rerun", "matplotlib", "trimesh", "viser", "wandb",
                                                              "sklearn", "imageio")
    ref_loader.install()
    with contextlib.redirect_stdout(io.StringIO()):
        import fast3r.models.multiview_dust3r_module as mm
    assert mm.roma is roma_stub
    return mm.MultiViewDUSt3RLitModule.align_local_pts3d_to_global


def main():
    warnings.filterwarnings("ignore")
    align = load_reference_align()
    cases = []
    for name, (nv, B, H, W, seed, noise, pct) in {"pct0": (3, 1, 24, 32, 1, 0.02, 0), "pct85_b2": (4, 2, 32, 48, 2, 0.05, 85), "pct50": (2, 1, 40, 40, 3, 0.0, 50),
                                                   "pct100": (2, 2, 16, 24, 4, 0.03, 100)}.items():
        preds = make_preds(nv, B, H, W, seed, noise)
        views = [{} for _ in range(nv)]
        if name == "pct85_b2":
            g = torch.Generator().manual_seed(77)
            views[1]["valid_mask"] = torch.rand(B, H, W, generator=g) > 0.3           # ordinary mask
            vm = torch.zeros(B, H, W, dtype=torch.bool)
            vm[:, 0, :5] = True                                                        # 5 valid pixels, none of them confident enough:
            preds[2]["conf"][:, 0, :5] = 1.0                                           #   first fall-back (valid_mask only)
            views[2]["valid_mask"] = vm
            vm2 = torch.zeros(B, H, W, dtype=torch.bool)
            vm2[:, 3, 3] = True                                                        # 1 valid pixel: identity
            views[3]["valid_mask"] = vm2
        work = [{k: v.clone() for k, v in p.items()} for p in preds]
        align(None, work, views, min_conf_thr_percentile=pct)
        cases.append(dict(name=name, pct=pct, preds=preds, views=views, aligned=[w["pts3d_local_aligned_to_global"].clone() for w in work]))
        err = max(float((w["pts3d_local_aligned_to_global"] - p["pts3d_in_other_view"]).abs().mean()) for w, p in zip(work, preds))
        print(name, "views", nv, "B", B, "pct", pct, "max mean |aligned - global| =", f"{err:.3f}")
    if "--check" in sys.argv:  # compare with the committed fixture instead of writing it
        old = torch.load(OUT, weights_only=False)["cases"]
        same = len(old) == len(cases) and all(all(torch.equal(x, y) for x, y in zip(a["aligned"], b["aligned"])) for a, b in zip(old, cases))
        print("align fixture:", "bit-identical to the committed one" if same else "DIFFERS from the committed one")
        sys.exit(0 if same else 1)
    torch.save(dict(cases=cases, torch_version=torch.__version__), OUT)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB")


if __name__ == "__main__":
    main()
