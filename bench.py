#!/usr/bin/env python
"""bench.py -- views/sec of the Fast3R single-forward-pass inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--views V] [--dtype fp16|bf16] [--fusion-only]

Workload (default): BASELINE.json's headline configuration -- Fast3R ViT-Large encoder + ViT-Large fusion decoder + both
DPT heads, V = 320 synthetic views of 512x512, ONE forward pass = one step -- the configuration the metric
"views/sec (512^2, ViT-L) per node at N=320" is quoted on.  It fits one GPU, so --gpus 1 runs the same 320 views on a
single MI355X and --gpus N shards them by view over N ranks (K / V^T all-gathered per fusion layer over RCCL): the
problem size is fixed, hence "scaling": "strong".  Inputs and (random-init) weights are synthetic and already resident in
HBM when the timed region starts.  `value` = V / max-over-ranks(step time).

Extra objects in the JSON line:
  roofline     fusion-attention kernel (the dominant kernel: 94.7 % of all FLOPs at N=320): algorithmic FLOPs per launch
               4*Tq*Tk*64*heads divided by the average launch duration measured live with events on the launch stream,
               against the dense 16-bit MFMA peak of 2.5 PFLOP/s (MI355X_MICROARCH.md).
  cpu_baseline the CPU oracle (oracle/fast3r_oracle.py, a port of the reference's torch-CPU fp32 path) timed on this
               box's host cores on a bounded sample (2 views of 512x512, full model), rank 0 / --gpus 1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16, MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views", type=int, default=320, help="total views N of the forward pass (BASELINE headline: 320)")
    ap.add_argument("--dtype", default="bf16", choices=["fp16", "bf16"], help="MFMA operand type (fp32 accumulate)")
    ap.add_argument("--fusion-only", action="store_true",
                    help="BASELINE configs[1]: time only the fusion decoder on frozen random encoder features")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-views", type=int, default=2)
    return ap.parse_args()


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON result.  Native libraries print there too (RCCL writes its version banner to stdout when
    # NCCL_DEBUG=VERSION is exported, as it is on the GPU boxes, and C stdio flushes it at exit, i.e. AFTER the JSON line): park the real
    # stdout, point fd 1 at stderr for everything else, and write the result to the parked descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched with `python -m torch.distributed.run --nproc-per-node N ...`")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # F3R_BENCH_FORCE_DIST=1: run the distributed code path (RCCL init, view sharding, barriers, MAX all-reduce) in a world of one
    # rank -- the only way to execute it on a one-GPU box (tools/gpu_ci.sh)
    force_dist = world == 1 and os.environ.get("F3R_BENCH_FORCE_DIST") == "1"
    distributed = world > 1 or force_dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from fast3r_amd import Fast3R, ops
    from fast3r_amd.dist import split_range
    from fast3r_amd.synthetic import make_views, synth_state_dict, vit_large_args

    lp = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    enc, dec, head = vit_large_args()
    model = Fast3R(enc, dec, head, compute_dtype=lp).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed=0)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    if distributed:
        model.shard_views()

    V = args.views
    lo, hi = split_range(V, world, rank)
    # every rank holds only ITS views in HBM (the list is indexed globally by the model)
    views = [None] * V
    for i in range(lo, hi):
        v = make_views(1, 512, 512, seed=1000 + i)[0]
        v["img"] = v["img"].to(dev)
        v["idx"] = i
        views[i] = v
    placeholder = {"img": views[lo]["img"]}
    views = [v if v is not None else placeholder for v in views]  # never read outside [lo, hi)

    if args.fusion_only:
        step_fn = make_fusion_only_step(model, V, lp, dev)
        workload = f"fusion transformer only (frozen random encoder features), N={V} views 512x512 (BASELINE configs[1] shape)"
    else:
        def step_fn():
            torch.manual_seed(1234)
            return model(views)
        workload = f"Fast3R ViT-L 512x512 end-to-end single forward pass (encoder + fusion decoder + 2 DPT heads), N={V} views"

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step_fn()
        ops.ATTN_TIMER = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        barrier()
        dt = time.perf_counter() - t0
        timer, ops.ATTN_TIMER = ops.ATTN_TIMER, None

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    # dominant kernel = the fusion attention launches (the ones whose key count is the whole scene)
    fus = [(a.elapsed_time(b), fl) for a, b, fl in timer]
    big = max(fl for _, fl in fus)
    fus = [(ms, fl) for ms, fl in fus if fl == big]
    avg_ms = sum(ms for ms, _ in fus) / len(fus)
    achieved = big / (avg_ms * 1e-3) / 1e12

    if rank == 0:
        out = {
            "metric": "views/sec (512^2, ViT-L) single forward pass at N=%d" % V,
            "value": V / (dt / args.steps), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload, "views": V, "views_per_gpu": [split_range(V, world, r)[1] - split_range(V, world, r)[0] for r in range(world)],
                       "tokens": V * 1024, "image": "512x512", "parallelism": f"view-sharded x{world}, K/V all-gather per fusion layer" if world > 1 else "single GPU",
                       "operands": f"{args.dtype} MFMA operands, fp32 accumulate / residual / LayerNorm / softmax"},
            "roofline": {"bound": "mfma", "kernel": "attn_kernel (fusion self-attention, one launch per fusion layer per rank)",
                         "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                         "flops_per_launch": big, "avg_launch_ms": avg_ms, "launches_timed": len(fus), "traffic": load_traffic(V, world),
                         "pmc": load_pmc()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, enc, dec, head, args.cpu_views)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


def make_fusion_only_step(model, V, lp, dev):
    """BASELINE configs[1]: frozen random encoder features (V,1024 tokens,1024) -> Fast3RDecoder only."""
    from fast3r_amd import ops
    pk = model._pack(dev)
    dec = model.decoder
    g = torch.Generator().manual_seed(0)
    feats = torch.randn((V * 1024, 1024), generator=g).to(lp).to(dev)
    emb = dec.image_idx_emb.to(dev)[torch.arange(V, device=dev)].contiguous()
    scale = dec.attention_scale(False)

    def step():
        x = torch.empty((V * 1024, 1024), dtype=torch.float32, device=dev)
        ops.gemm(feats, pk["de_w"], bias=pk["de_b"], rowadd=emb, rowadd_div=1024, out_f32=x)
        for pb in pk["dec"]:
            model._block(x, pb, dec.num_heads, scale, V * 1024, 1, None, None)
        w_, b_, eps = pk["dec_norm"]
        return ops.layernorm(x, w_, b_, eps, lp)
    return step


def load_traffic(V, world):
    """HBM bytes per attention launch from the committed rocprofv3 PMC pass (profiles/), if one exists for this shape."""
    path = os.path.join(ROOT, "profiles", "attn_traffic.json")
    try:
        d = json.load(open(path))
        return d.get(f"views={V},gpus={world}")
    except Exception:
        return None


def load_pmc():
    """Matrix-pipe utilisation in cycles + effective clock of the same kernel from the committed rocprofv3 PMC pass
    (profiles/r01_attn_mfma_util.json, tools/pmc_mfma_util.sh): `frac` above is this times clock / 2.4 GHz."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_attn_mfma_util.json")))
        return {"mfma_util_cycles": d["mfma_util_cycles"], "mfma_util_useful_cycles": d["mfma_util_useful_cycles"],
                "effective_clock_ghz": d["effective_clock_ghz"], "shape": "T=%d" % (1024 * d["views"])}
    except Exception:
        return None


def cpu_baseline(sd, enc, dec, head, n_views):
    """The oracle (port of the reference's CPU fp32 path) on the host cores, bounded sample of the same workload."""
    from fast3r_amd.synthetic import make_views
    from oracle import fast3r_oracle as O
    views = make_views(n_views, 512, 512)
    torch.manual_seed(1234)
    t0 = time.perf_counter()
    with torch.no_grad():
        O.forward(views, sd, enc, dec, head)
    dt = time.perf_counter() - t0
    return {"value": n_views / dt, "unit": "views/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"same model (ViT-L/ViT-L/2 DPT), {n_views} views of 512x512, one fp32 forward, {dt:.1f} s wall"}


if __name__ == "__main__":
    main()
