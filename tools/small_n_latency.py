"""Latency of small scenes (ViT-L, 512x512, bf16): eager launches vs hipGraph replay (Fast3R.enable_graphs).
usage (GPU box): python tools/small_n_latency.py [--views 2,3,8,20]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast3r_amd import Fast3R
from fast3r_amd.synthetic import make_views, synth_state_dict, vit_large_args

ap = argparse.ArgumentParser()
ap.add_argument("--views", default="2,3,8,20")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
ap.add_argument("--precision", default="fast", choices=["fast", "high"])
ap.add_argument("--no-graph", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda")
enc, dec, head = vit_large_args()
model = Fast3R(enc, dec, head, compute_dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float16, precision=args.precision).eval()
model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0), strict=True)
model = model.to(dev)


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


with torch.no_grad():
    for n in [int(x) for x in args.views.split(",")]:
        views = make_views(n, 512, 512)
        for v in views:
            v["img"] = v["img"].to(dev)
        model.enable_graphs(False)
        eager = timed(lambda: model(views), args.iters)
        graph = float("nan")
        if not args.no_graph:
            model.enable_graphs(True, max_views=max(64, n))
            graph = timed(lambda: model(views), args.iters)
        print(json.dumps({"views": n, "dtype": args.dtype, "precision": args.precision, "eager_ms": round(eager, 2), "graph_ms": round(graph, 2), "eager_views_per_s": round(n / eager * 1e3, 1),
                          "graph_views_per_s": round(n / graph * 1e3, 1)}), flush=True)
