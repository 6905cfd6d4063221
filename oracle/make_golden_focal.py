"""Generates tests/golden/focal_cases.pt by running the REAL reference estimator
(fast3r/dust3r/post_process.py::estimate_focal_knowing_depth_and_confidence_mask, focal_mode="weiszfeld") on seeded inputs.
Run in the build container (needs /root/reference):  python oracle/make_golden_focal.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_loader import load_reference  # noqa: E402


def synth_case(seed, H, W, focal, noise, outliers, degenerate):
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    z = 1.5 + torch.rand((H, W), generator=g) * 3.0
    x = (xs - W / 2) * z / focal
    y = (ys - H / 2) * z / focal
    pts = torch.stack([x, y, z], dim=-1)
    pts = pts + noise * torch.randn(pts.shape, generator=g)
    n_out = int(outliers * H * W)
    if n_out:
        idx = torch.randperm(H * W, generator=g)[:n_out]
        o = pts.view(-1, 3)[idx] + torch.randn((n_out, 3), generator=g) * 1.0  # gross errors, depth kept positive
        o[:, 2] = o[:, 2].abs() + 0.5
        pts.view(-1, 3)[idx] = o
    if degenerate:  # z = 0 -> +-inf / nan ratios, the nan_to_num path
        pts[0, :5, 2] = 0.0
        pts[1, 0] = 0.0
    conf = 1.0 + torch.rand((H, W), generator=g) * 5.0
    return pts[None].contiguous(), conf[None].contiguous()


def main():
    load_reference()
    from fast3r.dust3r.post_process import estimate_focal_knowing_depth_and_confidence_mask as ref_fn
    cases = []
    specs = [(1, 48, 64, 55.0, 0.0, 0.0, False), (2, 64, 64, 70.0, 0.01, 0.05, False), (3, 40, 56, 33.0, 0.02, 0.2, True),
             (4, 64, 48, 120.0, 0.005, 0.1, False), (5, 96, 128, 90.0, 0.01, 0.3, True)]
    for seed, H, W, f, noise, outl, deg in specs:
        pts, conf = synth_case(seed, H, W, f, noise, outl, deg)
        for pct in (10, 85):
            thr = torch.quantile(conf.reshape(-1), pct / 100.0)
            mask = (conf >= thr).view(1, H, W)
            pp = torch.tensor((W / 2, H / 2)).view(1, 2)
            with torch.no_grad():
                out = ref_fn(pts, pp.unsqueeze(0), mask, focal_mode="weiszfeld")
            cases.append(dict(seed=seed, H=H, W=W, true_focal=f, percentile=pct, pts3d=pts, conf=conf, focal=float(out.ravel()[0])))
            print(f"case seed={seed} {H}x{W} pct={pct}: true {f} reference {float(out.ravel()[0]):.6f}")
    out_path = os.path.join(ROOT, "tests", "golden", "focal_cases.pt")
    if "--check" in sys.argv:  # compare with the committed fixture instead of writing it
        old = torch.load(out_path, weights_only=False)
        same = len(old) == len(cases) and all(a["focal"] == b["focal"] and torch.equal(a["pts3d"], b["pts3d"]) and torch.equal(a["conf"], b["conf"])
                                              for a, b in zip(old, cases))
        print("focal fixture:", "bit-identical to the committed one" if same else "DIFFERS from the committed one")
        sys.exit(0 if same else 1)
    torch.save(cases, out_path)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
