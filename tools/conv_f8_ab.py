"""A/B of the DPT head's three heavy 3x3 convolution roles at the N = 320 shapes (head chunks of 25 views at 512 x 512): split "x3" (three fp16
products) against split "x3f8" (corrections on the block-scaled fp8 MFMA, include/f3r.h F3R_SPLIT_X3F8), timed with events; one JSON line per
(role, split).  Under `rocprofv3 --pmc ...` the same launches give the counters per kernel (tools/gpu_run.sh convf8pmc).

    python tools/conv_f8_ab.py [--views 25] [--reps 5] [--roles head2,head0,rcu128,rcu64]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fast3r_amd import ops  # noqa: E402

ROLES = {  # name: (H, W, Cin, Cout, residual, fused tail)
    "head2": (512, 512, 128, 128, False, True),
    "head2_nofin": (512, 512, 128, 128, False, False),
    "head0": (256, 256, 256, 128, False, False),
    "rcu128": (128, 128, 256, 256, True, False),
    "rcu128_nores": (128, 128, 256, 256, False, False),
    "rcu64": (64, 64, 256, 256, True, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=25)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--roles", default="head2,head2_nofin,head0,rcu128,rcu128_nores,rcu64")
    ap.add_argument("--splits", default="x3,x3f8,x3f8:5", help="split[:kernel_sel]; x3f8:5 = the four-phase schedule of the x3f8 kernels (measurement)")
    a = ap.parse_args()
    dev = "cuda"
    g = torch.Generator().manual_seed(3)
    for role in a.roles.split(","):
        H, W, C, N, res, fin = ROLES[role]
        B = a.views
        x32 = torch.randn((B, H, W, C), generator=g)
        w32 = torch.randn((N, C, 3, 3), generator=g) * (9 * C) ** -0.5
        bias = torch.randn(N, generator=g).to(dev)
        x_hi, x_lo = ops.split_planes(x32, torch.float16)
        x_hi, x_lo = x_hi.to(dev), x_lo.to(dev)
        x8 = ops.f8_planes(x32).to(dev)
        del x32
        wx3 = ops.pack_conv3x3_weight(w32, torch.float16, split=True).to(dev)
        w8, sc = ops.pack_conv3x3_weight_f8(w32)
        w8, sc = w8.to(dev), sc.to(dev)
        r_hi = r_lo = None
        if res:
            r_hi, r_lo = ops.split_planes(torch.randn((B, H, W, N), generator=g), torch.float16)
            r_hi, r_lo = r_hi.to(dev), r_lo.to(dev)
        finargs = ops.dpt_fin_args(torch.randn((4, N), generator=g).to(dev) * 0.05, torch.zeros(4, device=dev), ("exp", 1.0, float("inf"))) if fin else None
        for spec in a.splits.split(","):
            split, _, sel = spec.partition(":")
            kw = dict(split="x3", x_lo=x_lo) if split == "x3" else dict(split="x3f8", x_f8=x8, w_scale=sc)
            kw["kernel_sel"] = int(sel or 0)
            wt = wx3 if split == "x3" else w8

            def run():
                return ops.conv3x3(x_hi, wt, bias=bias, act="relu" if fin else None, res_lp=r_hi, res_lp_lo=r_lo, want_lo=not fin, fin=finargs, **kw)
            run()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
            ev[0].record()
            for i in range(a.reps):
                run()
                ev[i + 1].record()
            torch.cuda.synchronize()
            ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.reps))
            flop = 2.0 * B * H * W * N * 9 * C
            med = ms[len(ms) // 2]
            print(json.dumps({"role": role, "split": spec, "views": B, "shape": [H, W, C, N], "ms": round(med, 3), "ms_min": round(ms[0], 3),
                              "algorithmic_tflops": round(flop / med / 1e9, 1), "executed_units": 3 if split == "x3" else 2}), flush=True)
        del x_hi, x_lo, x8
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
