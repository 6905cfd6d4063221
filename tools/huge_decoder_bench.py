"""The reference's model_scaling_huge fusion decoder (configs/experiment/model_scaling/model_scaling_huge.yaml:12-15: embed_dim 1280, 16 heads =
head_dim 80, depth 32) behind the ViT-L encoder, one forward pass at N views of 512x512: views/s and what the fusion attention runs on --
the generated head_dim-80 kernel (kernel_sel 0) or, with --generic, the generic HIP kernel every launch took before round 4.

    python tools/huge_decoder_bench.py [--views 100] [--steps 2] [--generic] [--dtype fp16] [--precision high]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast3r_amd import Fast3R, ops  # noqa: E402
from fast3r_amd.synthetic import make_views, synth_state_dict, vit_large_args  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=100)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
ap.add_argument("--precision", default="high", choices=["fast", "high"])
ap.add_argument("--generic", action="store_true", help="force f3r_attn_args.kernel_sel = 1 (the generic HIP kernel at head_dim 80)")
args = ap.parse_args()
dev = torch.device("cuda")
enc, dec, head = vit_large_args()
dec.update(embed_dim=1280, num_heads=16, depth=32)
lp = torch.float16 if args.dtype == "fp16" else torch.bfloat16
model = Fast3R(enc, dec, head, compute_dtype=lp, precision=args.precision).eval()
model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0), strict=True)
model = model.to(dev)
if args.generic:
    real = ops.attention

    def forced(*a, **kw):
        kw["kernel_sel"] = 1
        return real(*a, **kw)
    ops.attention = forced
views = make_views(args.views, 512, 512)
for v in views:
    v["img"] = v["img"].to(dev)
with torch.no_grad():
    model(views)
    torch.cuda.synchronize()
    ops.ATTN_TIMER = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model(views)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    timer, ops.ATTN_TIMER = ops.ATTN_TIMER, None
fus = [(a.elapsed_time(b), fl, nm) for a, b, fl, _, _, nm in timer]
big = max(fl for _, fl, _ in fus)
sel = [(ms, nm) for ms, fl, nm in fus if fl == big]
avg = sum(ms for ms, _ in sel) / len(sel)
print(json.dumps({"model": "ViT-L encoder + model_scaling_huge decoder (1280 / 16 heads = head_dim 80, depth 32) + 2 DPT heads", "views": args.views,
                  "dtype": args.dtype, "precision": args.precision, "views_per_s": round(args.views / dt, 2), "ms_per_forward": round(dt * 1e3, 1),
                  "fusion_attention": {"kernel": sorted(set(nm for _, nm in sel)), "avg_launch_ms": round(avg, 3), "launches_timed": len(sel),
                                       "tflops": round(big / (avg * 1e-3) / 1e12, 1), "share_of_forward": round(avg * dec["depth"] / (dt * 1e3), 3)}}), flush=True)
