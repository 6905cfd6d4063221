"""TEST INFRASTRUCTURE ONLY -- anchors bench.py's `cpu_baseline.kind: "port"` on the real reference.

bench.py times oracle/fast3r_oracle.py on the GPU box's host cores because /root/reference does not exist there.  This script runs in
the build container, where it does: it times the reference itself (fast3r/dust3r/inference_multiview.py:21 `inference(..., dtype="32")`
around fast3r/models/fast3r.py:302 `Fast3R.forward`, attn_implementation="flash_attention" = torch SDPA on CPU) and the port on the SAME
model (ViT-L / ViT-L / 2 DPT heads, random init), the SAME views and the SAME thread count, interleaved, and prints port / reference.
BASELINE.md records the result.

    python -m oracle.time_reference [--views 3] [--threads 8] [--repeats 2]
"""
import argparse
import copy
import json
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast3r_amd.synthetic import make_views, synth_state_dict, vit_large_args  # noqa: E402
from oracle import fast3r_oracle as O  # noqa: E402
from oracle.ref_loader import load_reference  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--repeats", type=int, default=2)
    args = ap.parse_args()
    warnings.filterwarnings("ignore")
    torch.set_num_threads(args.threads)
    Fast3R, inference = load_reference()
    enc, dec, head = vit_large_args()
    model = Fast3R(copy.deepcopy(enc), copy.deepcopy(dec), copy.deepcopy(head)).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, 0)
    model.load_state_dict(sd, strict=True)
    views = make_views(args.views, 512, 512)
    O.ATTN_IMPL = "sdpa"

    def run_ref():
        torch.manual_seed(1234)
        t0 = time.perf_counter()
        out = inference(copy.deepcopy(views), model, torch.device("cpu"), dtype="32", verbose=False)
        return time.perf_counter() - t0, out["preds"]

    def run_port():
        torch.manual_seed(1234)
        t0 = time.perf_counter()
        with torch.no_grad():
            preds = O.forward(copy.deepcopy(views), sd, enc, dec, head)
        return time.perf_counter() - t0, preds

    _, p_ref = run_ref()      # warm-up of both (first touch of the weights, allocator, oneDNN primitive caches)
    _, p_port = run_port()
    diff = max(float((a[k].double() - b[k].double()).norm() / a[k].double().norm()) for a, b in zip(p_ref, p_port) for k in ("pts3d_in_other_view", "conf"))
    t_ref, t_port = [], []
    for _ in range(args.repeats):
        t_ref.append(run_ref()[0])
        t_port.append(run_port()[0])
    res = {"views": args.views, "threads": args.threads, "host_cores": os.cpu_count(), "reference_s": t_ref, "port_s": t_port,
           "reference_views_per_s": args.views / min(t_ref), "port_views_per_s": args.views / min(t_port),
           "port_over_reference_throughput": min(t_ref) / min(t_port), "port_vs_reference_rel_l2": diff,
           "what": "reference = inference(views, Fast3R(ViT-L, ViT-L, DPT x2), cpu, dtype='32') with attn_implementation='flash_attention' (SDPA); "
                   "port = oracle/fast3r_oracle.forward with ATTN_IMPL='sdpa'; same weights, same views of 512x512, best of the repeats"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
