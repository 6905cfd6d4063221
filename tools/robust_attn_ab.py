"""A/B of measurement builds of the three-product attention kernels on ONE box (tools/lab/build_attn_variants.sh NAME "FLAGS" ...): each variant library
is timed in its own process (F3R_LAB_LIB), the variants interleaved over several rounds so that box-to-box and minute-to-minute clock differences
cancel.   python tools/robust_attn_ab.py --variants q0,q1 [--views 100] [--planes 3]   -> one JSON line per (round, variant)"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(views, planes, iters):
    import torch
    sys.path.insert(0, ROOT)
    from fast3r_amd import ops
    T, H = views * 1024, 16
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.randn((T, 3 * H * 64), generator=g, device=dev) * 1.5
    qp, kp, vt = ops.qkv_planes(qkv, H, H, 1, T, 0.160192 * ops.LOG2E, torch.float16, planes=planes)
    st = ops.attention_state(T, H, dev)
    seg = [(kp, vt.view(H * 64, -1), T, 0, 0)]
    for _ in range(2):
        ops.attention(qp, st[0], H, 1.0, seg, q_prescaled=True, state=st, state_out=True, qk_planes=planes, kernel_sel=2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.attention(qp, st[0], H, 1.0, seg, q_prescaled=True, state=st, state_out=True, qk_planes=planes, kernel_sel=2)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(json.dumps({"ms": ms, "algorithmic_tflops": 4.0 * T * T * 64 * H / ms / 1e9}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="")
    ap.add_argument("--views", type=int, default=100)
    ap.add_argument("--planes", type=int, default=3)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--worker", action="store_true")
    a = ap.parse_args()
    if a.worker:
        worker(a.views, a.planes, a.iters)
        sys.exit(0)
    for rnd in range(a.rounds):
        for v in a.variants.split(","):
            env = dict(os.environ, F3R_LAB_LIB=os.path.join(ROOT, "tools", "lab", "var", f"libf3r_{v}.so"))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", "--views", str(a.views), "--planes", str(a.planes), "--iters", str(a.iters)],
                               env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            rec = json.loads(line[-1]) if line else {"error": p.stderr[-300:]}
            rec.update(round=rnd, variant=v, views=a.views, planes=a.planes)
            print(json.dumps(rec), flush=True)
