"""How long one forward pass takes in the fp32-equivalent mode (precision="exact": what inference(dtype="32") selects), ViT-L at 512x512:
    python tools/exact_mode_timing.py [--views 20,100,320]
One JSON line per view count: the second of two forwards (the first pays the allocator), beside the default format (fp16 / high)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast3r_amd import Fast3R  # noqa: E402
from fast3r_amd.synthetic import make_views, synth_state_dict, vit_large_args  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", default="20,100,320")
args = ap.parse_args()
dev = torch.device("cuda")
enc, dec, head = vit_large_args()
sd = synth_state_dict({k: tuple(v.shape) for k, v in Fast3R(enc, dec, head).state_dict().items()}, seed=0, dist="hot")
for n in [int(x) for x in args.views.split(",")]:
    views = make_views(n, 512, 512)
    for v in views:
        v["img"] = v["img"].to(dev)
    out = {"views": n}
    for precision in ("exact", "high"):
        m = Fast3R(enc, dec, head, compute_dtype=torch.float16, precision=precision).eval()
        m.load_state_dict(sd, strict=True)
        m = m.to(dev)
        with torch.no_grad():
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                m(views)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
        out[f"{precision}_ms"] = round(dt * 1e3, 1)
        del m
        torch.cuda.empty_cache()
    out["exact_over_high"] = round(out["exact_ms"] / out["high_ms"], 2)
    print(json.dumps(out), flush=True)
