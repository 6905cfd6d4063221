"""Kernel micro-benchmarks on the GPU box (interleaved rounds, random data, HIP events on the launch stream): the attention kernels
f3r_attn_fwd can take (--what attnproduct / attnsel / attnhd) and the model's GEMM / conv shapes per f3r_gemm_args.kernel_sel, with the vendor
library beside them (--what gemmref).  Prints one JSON line per item.  (--what lab / labtime: ablations of the 8-wave GEMM, need
F3R_LAB_LIB=tools/lab/libf3r_hip_lab.so.)"""
import argparse
import math
import json
import sys

import torch

sys.path.insert(0, ".")
from fast3r_amd import _lib, ops  # noqa: E402

DEV = "cuda"


def time_ms(fn, rounds=5, inner=3):
    best = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / inner)
    best.sort()
    return best[len(best) // 2], best[0]


SEL_NAME = {0: "automatic", 1: "128-tile", 2: "256-tile persistent", 3: "256-tile lock-step", 4: "256x128-tile", 5: "256-tile one tile per workgroup",
            6: "hand-scheduled (asm)", 7: "compiler-scheduled kernels only"}


def bench_attn_product(dt, views, H=16):
    """The product attention kernel (no variant knob in the product library) on the fusion shape."""
    T = views * 1024
    D = H * 64
    q = torch.randn((T, D), device=DEV).to(dt)
    k = torch.randn((T, D), device=DEV).to(dt)
    vt = torch.randn((D, T), device=DEV).to(dt)
    o = torch.empty((T, D), dtype=dt, device=DEV)

    def f():
        ops.attention(q, o, H, 0.160192, [(k, vt, T, 0, 0)])
    f()
    med, mn = time_ms(f, rounds=3, inner=2)
    print(json.dumps({"kernel": "attn (product)", "dtype": str(dt).split(".")[-1], "views": views, "T": T, "ms": round(med, 3),
                      "tflops": round(4.0 * T * T * 64 * H / med / 1e9, 1)}), flush=True)


def bench_attn_sel(dt, views, sels=(1, 2), H=16, rounds=3, inner=2, tokens_per_view=1024):
    """The general HIP kernel (kernel_sel 1) next to the hand-scheduled one (kernel_sel 2) on the fusion shape, q pre-scaled."""
    T = views * tokens_per_view
    D = H * 64
    q = (torch.randn((T, D), device=DEV) * (0.160192 * 1.4426950408889634)).to(dt)
    k = torch.randn((T, D), device=DEV).to(dt)
    vt = torch.zeros((D, ops.vt_ld(T)), device=DEV, dtype=dt)
    vt[:, :T] = torch.randn((D, T), device=DEV).to(dt)
    outs = {}
    for sel in sels:
        o = torch.empty((T, D), dtype=dt, device=DEV)

        def f(sel=sel, o=o):
            ops.attention(q, o, H, 0.160192, [(k, vt, T, 0, 0)], q_prescaled=True, kernel_sel=sel)
        f()
        med, mn = time_ms(f, rounds=rounds, inner=inner)
        outs[sel] = o
        print(json.dumps({"kernel": "attn_sel", "kernel_sel": sel, "dtype": str(dt).split(".")[-1], "views": views, "T": T, "ms": round(med, 3),
                          "ms_min": round(mn, 3), "tflops": round(4.0 * T * T * 64 * H / med / 1e9, 1)}), flush=True)
    if 1 in outs and 2 in outs:
        a, b = outs[1].float(), outs[2].float()
        print(json.dumps({"kernel": "attn_sel", "views": views, "dtype": str(dt).split(".")[-1],
                          "rel_l2_asm_vs_hip": float((a - b).norm() / a.norm()), "nan": int(torch.isnan(b).sum())}), flush=True)


def bench_attn_head_dim(dt, views, hd, H=16, sels=(1, 0), T=None):
    """f3r_attn_fwd on a fusion-shaped problem with head_dim hd per f3r_attn_args.kernel_sel: 1 = the HIP kernel (the generic one unless hd = 64),
    0 = automatic (the generated kernel of csrc/asm/attn_gen.py at 64 / 80 / 128)"""
    T = views * 1024 if T is None else T
    D = H * hd
    q = (torch.randn((T, D), device=DEV) * (hd ** -0.5 * 1.4426950408889634)).to(dt)
    k = torch.randn((T, D), device=DEV).to(dt)
    vt = torch.zeros((D, ops.vt_ld(T)), device=DEV, dtype=dt)
    vt[:, :T] = torch.randn((D, T), device=DEV).to(dt)
    outs = {}
    for sel in sels:
        o = torch.empty((T, D), dtype=dt, device=DEV)

        def f():
            ops.attention(q, o, H, hd ** -0.5, [(k, vt, T, 0, 0)], q_prescaled=True, head_dim=hd, kernel_sel=sel)
        ops.ATTN_TIMER = []
        f()
        torch.cuda.synchronize()
        name = ops.ATTN_TIMER[0][5]
        ops.ATTN_TIMER = None
        med, mn = time_ms(f, rounds=3, inner=2)
        outs[sel] = o.float()
        print(json.dumps({"kernel": "attn_head_dim", "head_dim": hd, "kernel_sel": sel, "runs": name, "dtype": str(dt).split(".")[-1], "views": views, "T": T,
                          "ms": round(med, 3), "tflops": round(4.0 * T * T * hd * H / med / 1e9, 1), "tflops_best": round(4.0 * T * T * hd * H / mn / 1e9, 1)}), flush=True)
    if len(outs) == 2:
        a, b = list(outs.values())
        print(json.dumps({"kernel": "attn_head_dim", "head_dim": hd, "rel_l2_between_kernels": float((a - b).norm() / a.norm()), "nan": int(torch.isnan(b).sum())}), flush=True)


def bench_gemm(dt, M, N, K, name, act=None, res=False, out="f32", sels=(1, 2, 3), split=None):
    a = torch.randn((M, K), device=DEV).to(dt)
    a_lo = torch.randn((M, K), device=DEV).mul_(2.0 ** -11).to(dt) if split == "x3" else None
    w = ops.pack_linear_weight(torch.randn((N, K), device=DEV) * K ** -0.5, dt, split=split is not None)
    bias = torch.randn(N, device=DEV)
    x = torch.randn((M, N), device=DEV) if (res or out == "f32") else None
    olp = torch.empty((M, N), dtype=dt, device=DEV) if out == "lp" else None
    fns = {}
    for sel in sels:
        def f(sel=sel):
            if out == "f32":
                ops.gemm(a, w, bias=bias, act=act, res_f32=x if res else None, out_f32=x, kernel_sel=sel, split=split, a_lo=a_lo)
            else:
                ops.gemm(a, w, bias=bias, act=act, out_lp=olp, kernel_sel=sel, split=split, a_lo=a_lo)
        f()
        fns[sel] = f
    res_ms = {sel: [] for sel in sels}
    for _ in range(3):  # interleaved rounds
        for sel in sels:
            res_ms[sel].append(time_ms(fns[sel], rounds=3, inner=3)[0])
    mult = {None: 1, "w2": 2, "x3": 3}[split]
    for sel in sels:
        ms = sorted(res_ms[sel])[1]
        print(json.dumps({"kernel": "gemm", "name": name, "variant": SEL_NAME[sel], "split": split, "dtype": str(dt).split(".")[-1], "M": M, "N": N, "K": K,
                          "ms": round(ms, 3), "tflops_algorithmic": round(2.0 * M * N * K / ms / 1e9, 1),
                          "tflops_mfma": round(2.0 * M * N * K * mult / ms / 1e9, 1)}), flush=True)


def bench_library_matmul(dt, M, N, K, name):
    """A KNOWN-GOOD reference on the same box and the same random data (cdna_hip_programming.md rule 10 / 25): torch.matmul, i.e. the vendor
    library (hipBLASLt / rocBLAS), plain A W^T -> lowp, no epilogue.  Not a product path: a yardstick for what the chip gives this shape."""
    a = torch.randn((M, K), device=DEV).to(dt)
    w = (torch.randn((N, K), device=DEV) * K ** -0.5).to(dt)
    out = torch.empty((M, N), dtype=dt, device=DEV)
    f = lambda: torch.matmul(a, w.t(), out=out)  # noqa: E731
    f()
    ms = sorted(time_ms(f, rounds=3, inner=3)[0] for _ in range(3))[1]
    print(json.dumps({"kernel": "library matmul (torch.matmul -> hipBLASLt/rocBLAS)", "name": name, "dtype": str(dt).split(".")[-1], "M": M, "N": N, "K": K,
                      "ms": round(ms, 3), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)


LAB_BITS = {0: "lab baseline (staggered)", 1: "no LDS-DMA in loop", 2: "no fragment reads", 3: "no DMA, no reads (MFMA + barriers)", 4: "no MFMA",
            7: "barriers only", 8: "no vmcnt wait", 16: "no s_setprio", 32: "buffer_load lds", 33: "buffer_load, no DMA (sanity)",
            64: "no sched_barrier pin", 96: "buffer_load + no pin"}


def bench_lab(M, N, K):
    """Ablations of the 256-tile kernel (tools/lab/libf3r_hip_lab.so, F3R_LAB_LIB): which section of the phase costs what."""
    dt = torch.bfloat16
    a = torch.randn((M, K), device=DEV).to(dt)
    w = ops.pack_linear_weight(torch.randn((N, K), device=DEV) * K ** -0.5, dt)
    olp = torch.empty((M, N), dtype=dt, device=DEV)
    fns = {}
    for bits in LAB_BITS:
        def f(bits=bits):
            ops.gemm(a, w, out_lp=olp, kernel_sel=16 + bits)
        f()
        fns[bits] = f
    fns[-2] = lambda: ops.gemm(a, w, out_lp=olp, kernel_sel=2)
    fns[-1] = lambda: ops.gemm(a, w, out_lp=olp, kernel_sel=1)
    res_ms = {b: [] for b in fns}
    for _ in range(3):
        for b in fns:
            res_ms[b].append(time_ms(fns[b], rounds=3, inner=3)[0])
    for b in fns:
        ms = sorted(res_ms[b])[1]
        name = LAB_BITS.get(b, "product 256-tile" if b == -2 else "product 128-tile")
        print(json.dumps({"kernel": "gemm256_lab", "bits": b, "what": name, "M": M, "N": N, "K": K, "ms": round(ms, 3),
                          "tflops_nominal": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)


def bench_labtime(M, N, K, act=None, bits=128):
    """Where a 256 x 256 output tile spends its time (lab bit 128: s_memtime stamps of wave 0 of every workgroup)."""
    from fast3r_amd import _lib
    dt = torch.bfloat16
    a = torch.randn((M, K), device=DEV).to(dt)
    w = ops.pack_linear_weight(torch.randn((N, K), device=DEV) * K ** -0.5, dt)
    bias = torch.randn(N, device=DEV)
    olp = torch.empty((M, N), dtype=dt, device=DEV)
    tiles = (M // 256) * (N // 256)
    dbg = torch.zeros((tiles, 5), dtype=torch.int64, device=DEV)
    L = _lib.lib()
    orig = L.f3r_gemm

    def patched(argp, stream):
        argp._obj.rope_cos = dbg.data_ptr()  # unused by the generic epilogue: carries the stamp buffer
        return orig(argp, stream)
    L.f3r_gemm = patched
    try:
        for _ in range(3):
            ops.gemm(a, w, bias=bias, act=act, out_lp=olp, kernel_sel=16 + bits)
        torch.cuda.synchronize()
        ms = time_ms(lambda: ops.gemm(a, w, bias=bias, act=act, out_lp=olp, kernel_sel=16 + bits), rounds=3, inner=3)[0]
    finally:
        L.f3r_gemm = orig
    d = dbg.cpu().double()
    t0 = d[:, 0].min()
    span = (d[:, 4].max() - t0).item()
    seg = [(d[:, i + 1] - d[:, i]).mean().item() for i in range(4)]
    tot = (d[:, 4] - d[:, 0]).mean().item()
    rounds = -(-tiles // 256)
    print(json.dumps({"kernel": "gemm256_lab stamps", "bits": bits, "M": M, "N": N, "K": K, "act": act, "ms": round(ms, 3), "tiles": tiles, "rounds_of_256": rounds,
                      "ticks_kernel_span": span, "ticks_per_tile_mean": tot, "prologue": seg[0], "main_loop": seg[1], "epilogue_issue": seg[2],
                      "store_retire": seg[3], "ticks_per_us": round(span / (ms * 1e3), 1),
                      "frac": {k: round(v / tot, 3) for k, v in zip(("prologue", "main_loop", "epilogue_issue", "store_retire"), seg)}}), flush=True)


def bench_qkv(dt, M, D, seq, sels=(1, 2, 3)):
    a = torch.randn((M, D), device=DEV).to(dt)
    w = ops.pack_linear_weight(torch.randn((3 * D, D), device=DEV) * D ** -0.5, dt)
    bias = torch.randn(3 * D, device=DEV)
    q = torch.empty((M, D), dtype=dt, device=DEV)
    k = torch.empty((M, D), dtype=dt, device=DEV)
    vt = torch.empty((M // seq, D, seq), dtype=dt, device=DEV)
    cos, sin = ops.rope_tables(32, 100.0, DEV)
    all_sels = sels
    for rope in (None, (cos, sin, 32)):
        fns = {}
        sels = [x for x in all_sels if not (x == 6 and rope is not None)]  # the hand-scheduled kernels take QKV without rotary embedding only
        for sel in sels:
            def f(sel=sel):
                ops.gemm_qkv(a, w, bias, q, k, vt, seq, rope, kernel_sel=sel)
            f()
            fns[sel] = f
        res_ms = {sel: [] for sel in sels}
        for _ in range(3):
            for sel in sels:
                res_ms[sel].append(time_ms(fns[sel], rounds=3, inner=3)[0])
        for sel in sels:
            ms = sorted(res_ms[sel])[1]
            print(json.dumps({"kernel": "gemm_qkv", "variant": SEL_NAME[sel], "rope": rope is not None, "dtype": str(dt).split(".")[-1], "M": M, "seq": seq,
                              "ms": round(ms, 3), "tflops": round(2.0 * M * 3 * D * D / ms / 1e9, 1)}), flush=True)


def bench_conv(dt, B, H, W, Ci, Co, name, stride=1, sels=(1, 2, 3), split=None):
    x = torch.randn((B, H, W, Ci), device=DEV).to(dt)
    x_lo = torch.randn((B, H, W, Ci), device=DEV).mul_(2.0 ** -11).to(dt) if split == "x3" else None
    w = ops.pack_conv3x3_weight(torch.randn((Co, Ci, 3, 3), device=DEV) * (9 * Ci) ** -0.5, dt, split=split is not None)
    bias = torch.randn(Co, device=DEV)
    oh, ow = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    r1 = torch.randn((B, oh, ow, Co), device=DEV).to(dt)
    cases = [("a_relu (round-1 form)", dict(a_relu=True, res_lp=r1), (1,))] if split is None else []
    cases.append(("skip + relu copy", dict(res_lp=r1, want_relu=True), tuple(s_ for s_ in sels if stride == 1 and Ci % 64 == 0 and Co % 128 == 0 or s_ == 1)))
    for cname, kw, csels in cases:
        fns = {}
        for sel in csels:
            def f(sel=sel, kw=kw):
                ops.conv3x3(x, w, stride=stride, bias=bias, kernel_sel=sel, split=split, x_lo=x_lo, **kw)
            f()
            fns[sel] = f
        res_ms = {sel: [] for sel in csels}
        for _ in range(3):
            for sel in csels:
                res_ms[sel].append(time_ms(fns[sel], rounds=3, inner=3)[0])
        mult = {None: 1, "w2": 2, "x3": 3}[split]
        for sel in csels:
            ms = sorted(res_ms[sel])[1]
            fl = 2.0 * B * oh * ow * 9 * Ci * Co
            print(json.dumps({"kernel": "conv3x3", "name": name, "case": cname, "variant": SEL_NAME[sel], "split": split, "dtype": str(dt).split(".")[-1], "B": B,
                              "HW": [H, W], "Ci": Ci, "Co": Co, "ms": round(ms, 3), "tflops_algorithmic": round(fl / ms / 1e9, 1),
                              "tflops_mfma": round(fl * mult / ms / 1e9, 1)}), flush=True)


def bench_dpt_final(n_views=25, H=512, W=512):
    """head[4] 1x1 conv + postprocess over one head chunk: algorithmic bytes = npix * (Cin * 2 [* 2 planes] + 16)."""
    for dt in (torch.bfloat16, torch.float16):
        for pair in (False, True):
            x = torch.randn((n_views, H, W, 128), device=DEV).to(dt)
            lo = (torch.randn((n_views, H, W, 128), device=DEV) * 2.0 ** -11).to(dt) if pair else None
            w, b = torch.randn(4, 128, device=DEV) * 0.1, torch.randn(4, device=DEV) * 0.1
            f = lambda: ops.dpt_final(x, w, b, ["exp", 1, float("inf")], x_lo=lo)
            f()
            ms = sorted(time_ms(f, rounds=3, inner=3)[0] for _ in range(3))[1]
            nbytes = n_views * H * W * (128 * 2 * (2 if pair else 1) + 16)
            print(json.dumps({"kernel": "dpt_final", "dtype": str(dt).split(".")[-1], "planes": 2 if pair else 1, "views": n_views, "ms": round(ms, 3),
                              "GBps_algorithmic": round(nbytes / ms / 1e6, 1)}), flush=True)


def bench_align(n_views=320, H=512, W=512, pct=85):
    """align_local_pts3d_to_global at the headline shape: HBM-bound.  Algorithmic bytes per view: conf 4 B x (6 radix passes + 1)
    + local/global points 2 x 12 B (moments) + local 12 B + out 12 B (apply) = 76 B per pixel."""
    import time
    from fast3r_amd import align_local_pts3d_to_global
    from oracle.align_oracle import align_local_pts3d_to_global as align_oracle
    g = torch.Generator().manual_seed(0)
    base = {"pts3d_local": torch.randn(1, H, W, 3, generator=g), "pts3d_in_other_view": torch.randn(1, H, W, 3, generator=g),
            "conf": 1 + torch.exp(torch.randn(1, H, W, generator=g))}
    base["conf_local"] = base["conf"]
    preds = [{k: v.cuda() + 0.001 * i for k, v in base.items()} for i in range(n_views)]
    views = [{} for _ in range(n_views)]

    def f():
        align_local_pts3d_to_global(preds, views, pct)
    f()
    med, mn = time_ms(f, rounds=3, inner=1)
    nbytes = n_views * H * W * 76.0
    t0 = time.perf_counter()
    align_oracle([{k: v.cpu() for k, v in preds[i].items()} for i in range(4)], views[:4], pct)
    cpu_s = (time.perf_counter() - t0) / 4
    print(json.dumps({"kernel": "align_local_to_global", "views": n_views, "HW": [H, W], "percentile": pct, "ms": round(med, 3),
                      "views_per_s": round(n_views / med * 1e3, 1), "GBps_algorithmic": round(nbytes / med / 1e6, 1),
                      "cpu_oracle_ms_per_view": round(cpu_s * 1e3, 1), "note": "ms includes torch.stack of the inputs (3 copies)"}), flush=True)


def bench_focal(n_views=320, H=512, W=512, pct=10):
    """estimate_focal at the headline shape.  Algorithmic bytes per pixel: conf 4 B x (6 radix passes + 1) + points 12 B + workspace
    16 B written once and read 100 times (L2-resident: 4 MB per view) = 1656 B; HBM-side only the first 56 B."""
    import time
    from fast3r_amd import estimate_focals
    from oracle import focal_oracle as FO
    g = torch.Generator().manual_seed(0)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    z = 1.5 + 3 * torch.rand(H, W, generator=g)
    pts1 = torch.stack([(xs - W / 2) * z / 400.0, (ys - H / 2) * z / 400.0, z], -1) + 0.01 * torch.randn(H, W, 3, generator=g)
    conf1 = 1 + 5 * torch.rand(H, W, generator=g)
    pts = pts1[None].repeat(n_views, 1, 1, 1).cuda() * (1 + 0.001 * torch.arange(n_views).view(-1, 1, 1, 1).cuda())
    conf = conf1[None].repeat(n_views, 1, 1).cuda()
    out = estimate_focals(pts, conf, min_conf_thr_percentile=pct)
    med, mn = time_ms(lambda: estimate_focals(pts, conf, min_conf_thr_percentile=pct), rounds=3, inner=1)
    t0 = time.perf_counter()
    ref = FO.estimate_focal(pts[:1].cpu(), conf[:1].cpu(), min_conf_thr_percentile=pct)
    cpu_s = time.perf_counter() - t0
    print(json.dumps({"kernel": "estimate_focal", "views": n_views, "HW": [H, W], "percentile": pct, "ms": round(med, 3),
                      "views_per_s": round(n_views / med * 1e3, 1), "L2_GBps": round(n_views * H * W * 1656.0 / med / 1e6, 1),
                      "focal_view0": float(out[0]), "cpu_oracle_focal_view0": ref, "cpu_oracle_ms_per_view": round(cpu_s * 1e3, 1)}), flush=True)


def bench_pnp(n_views=320, H=512, W=512):
    """estimate_poses at the headline shape: ~50 passes over 16 B per pixel (L2-resident, 4 MB per view) of fp64 per-point arithmetic."""
    import time
    from fast3r_amd import estimate_poses
    from oracle import pnp_oracle as PO
    g = torch.Generator().manual_seed(0)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    z = 2 + 3 * torch.rand(H, W, generator=g)
    Xc = torch.stack([(xs - W / 2) * z / 400.0, (ys - H / 2) * z / 400.0, z], -1)
    c, s_ = math.cos(0.3), math.sin(0.3)
    R = torch.tensor([[c, 0, s_], [0, 1, 0], [-s_, 0, c]])
    t = torch.tensor([0.2, -0.1, 0.5])
    Xw = (Xc - t) @ R + 0.003 * torch.randn(H, W, 3, generator=g)
    conf1 = 1.001 + 4 * torch.rand(H, W, generator=g)
    pts = Xw[None].repeat(n_views, 1, 1, 1).cuda()
    conf = conf1[None].repeat(n_views, 1, 1).cuda()
    for mode, focal in (("focal given", 400.0), ("focal searched", None)):
        estimate_poses(pts, conf, focal)
        med, mn = time_ms(lambda: estimate_poses(pts, conf, focal), rounds=3, inner=1)
        P, F, I = estimate_poses(pts[:1], conf[:1], focal)
        t0 = time.perf_counter()
        fo, To = PO.fast_pnp(Xw, focal, conf1 > 1.0)
        cpu_s = time.perf_counter() - t0
        print(json.dumps({"kernel": "estimate_poses", "mode": mode, "views": n_views, "HW": [H, W], "ms": round(med, 3),
                          "views_per_s": round(n_views / med * 1e3, 1), "focal_view0": float(F[0]), "inliers_view0": int(I[0]),
                          "max_abs_diff_vs_cpu_restatement": float((P[0].double().cpu() - To).abs().max()),
                          "cpu_restatement_ms_per_view": round(cpu_s * 1e3, 1)}), flush=True)


def bench_image(H=3000, W=4000, size=512, n=16):
    """load_images' per-pixel work at a 12 MP camera frame: PIL LANCZOS resize + crop + ImgNorm on the host vs upload + GPU."""
    import time
    import numpy as np
    from PIL import Image
    from fast3r_amd.image import img_norm_crop, resize_u8, _resized_size
    rng = np.random.default_rng(0)
    frame = (rng.random((H, W, 3)) * 255).astype(np.uint8)
    (nw, nh), interp = _resized_size((W, H), size)
    pil = Image.fromarray(frame)
    t0 = time.perf_counter()
    for _ in range(3):
        ref = np.asarray(pil.resize((nw, nh), Image.LANCZOS))
        x = (np.transpose(ref, (2, 0, 1)).astype(np.float32) / 255 - 0.5) / 0.5
    cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
    cache = {}
    host = torch.from_numpy(frame).pin_memory()

    def f():
        u8 = host.to(DEV, non_blocking=True)
        r = resize_u8(u8, nw, nh, interp, cache)
        return img_norm_crop(r, (0, 0, nw - nw % 16, nh - nh % 16)), r
    out, r = f()
    assert np.array_equal(r.cpu().numpy(), ref)
    med, mn = time_ms(f, rounds=3, inner=n)
    u8 = host.to(DEV)
    med_k, _ = time_ms(lambda: resize_u8(u8, nw, nh, interp, cache), rounds=3, inner=n)
    print(json.dumps({"kernel": "load_images per-pixel work", "frame": [H, W], "to": [nh, nw], "gpu_ms_incl_upload": round(med, 3),
                      "gpu_ms_resize_only": round(med_k, 3), "resize_GBps": round((H * W * 3 + H * nw * 3 * 2 + nh * nw * 3) / med_k / 1e6, 1),
                      "pil_ms": round(cpu_ms, 1), "bit_exact_vs_pil": True}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="gemm,conv")
    ap.add_argument("--views", default="20,100")
    ap.add_argument("--attn-dtypes", default="bf16,fp16", help="attnproduct: which operand formats")
    ap.add_argument("--sels", default="1,2", help="attnsel: f3r_attn_args.kernel_sel values to time")
    ap.add_argument("--tokens-per-view", type=int, default=1024, help="attnsel: 768 = 384x512 images (an odd view count gives a partial last workgroup)")
    args = ap.parse_args()
    dt = torch.bfloat16
    if args.what == "gemmsmall":  # scenes of 3 - 40 views: which tile form fills 256 CUs best
        for M in (3072, 8192, 20480, 40960):
            bench_gemm(dt, M, 1024, 1024, f"proj+res M={M}", res=True, sels=(1, 2, 4))
            bench_gemm(dt, M, 1024, 4096, f"fc2+res M={M}", res=True, sels=(1, 2, 4))
            bench_gemm(dt, M, 4096, 1024, f"fc1+gelu M={M}", act="gelu", out="lp", sels=(1, 2, 4))
        sys.exit(0)
    if args.what == "gemmsmallw2":  # the same question for the product's W2 launches, with the hand-scheduled kernel (6) in the field: automatic (0) should be the fastest
        for M in [int(v) * 1024 for v in args.views.split(",")]:
            for sels in ((0, 6, 2, 4, 1),):
                bench_gemm(torch.float16, M, 1024, 1024, f"proj+res M={M}", res=True, sels=sels, split="w2")
                bench_gemm(torch.float16, M, 1024, 4096, f"fc2+res M={M}", res=True, sels=sels, split="w2")
                bench_gemm(torch.float16, M, 4096, 1024, f"fc1+gelu M={M}", act="gelu", out="lp", sels=sels, split="w2")
        sys.exit(0)
    if args.what == "attnproduct":
        for nv in [int(v) for v in args.views.split(",")]:
            for d in (args.attn_dtypes.split(",")):
                bench_attn_product({"bf16": torch.bfloat16, "fp16": torch.float16}[d], nv)
        sys.exit(0)
    if args.what == "gemmref":  # the transformer's GEMM roles at N = 320 beside the vendor library on the same data
        sels = tuple(int(x) for x in args.sels.split(","))
        for dt in (torch.float16, torch.bfloat16):
            for M in [int(v) * 1024 for v in args.views.split(",")]:
                for (n, k, nm, kw) in ((1024, 1024, "proj+res", dict(res=True)), (4096, 1024, "fc1+gelu", dict(act="gelu", out="lp")),
                                       (1024, 4096, "fc2+res", dict(res=True)), (4096, 1024, "fc1 plain lp", dict(out="lp"))):
                    bench_gemm(dt, M, n, k, f"{nm} M={M}", sels=sels, **kw)
                    bench_library_matmul(dt, M, n, k, f"{nm} M={M}")
                bench_qkv(dt, M, 1024, M, sels=sels)
                bench_library_matmul(dt, M, 3072, 1024, f"qkv M={M}")
        for M in [int(v) * 1024 for v in args.views.split(",")]:
            bench_gemm(torch.float16, M, 4096, 1024, f"fc1+gelu w2 M={M}", act="gelu", out="lp", split="w2", sels=sels)
            bench_gemm(torch.float16, M, 1024, 4096, f"fc2+res w2 M={M}", res=True, split="w2", sels=sels)
        sys.exit(0)
    if args.what == "attnsteal":  # the fusion attention (T = 1024 x views, 16 heads, fp16) with f3r_attn_args.sched_counter off / on, interleaved
        for nv in [int(x) for x in args.views.split(",")]:
            T, H = nv * 1024, 16
            q = (torch.randn((T, H * 64), device=DEV) * 0.2).half()
            k = torch.randn((T, H * 64), device=DEV).half()
            vt = torch.randn((H * 64, ops.vt_ld(T)), device=DEV).half()
            o = torch.empty_like(q)
            res = {False: [], True: []}
            ops.ATTN_COUNTERS = None
            for rnd_ in range(4):
                for steal in (False, True):
                    ops.ATTN_WORK_STEALING = steal
                    f = lambda: ops.attention(q, o, H, 1.0, [(k, vt, T, 0, 0)], q_prescaled=True, kernel_sel=2)  # noqa: E731
                    if rnd_ == 0:
                        f()
                    res[steal].append(time_ms(f, rounds=3, inner=1 if nv >= 100 else 8)[0])
            fl = 4.0 * T * T * 64 * H
            for steal in (False, True):
                ms = sorted(res[steal])[len(res[steal]) // 2]
                print(json.dumps({"kernel": "f3r_attn_asm_f16", "work_stealing": steal, "views": nv, "T": T, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 1),
                                  "frac_of_peak": round(fl / ms / 1e9 / 2500.0, 4), "rounds_ms": [round(x, 3) for x in res[steal]]}), flush=True)
            ops.ATTN_WORK_STEALING = True
        sys.exit(0)
    if args.what == "gemmf8":  # the transformer's linear roles: two fp16 weight planes (W2, today) next to the low plane in fp8 (W2F8), hand-scheduled kernels
        f16 = torch.float16
        for M in [int(v) * 1024 for v in args.views.split(",")]:
            for (n, k, nm, kw) in ((4096, 1024, "fc1+gelu", dict(act="gelu", out="lp")), (1024, 4096, "fc2+res", dict(res=True)), (1024, 1024, "proj+res", dict(res=True)),
                                   (2048, 1024, "q|k lp", dict(out="lp"))):
                a16 = torch.randn((M, k), device=DEV).to(f16)
                rows = torch.empty((M, 3 * k // 2), dtype=f16, device=DEV)
                rows[:, :k] = a16
                rows.view(torch.uint8).view(M, 3 * k)[:, 2 * k:] = a16.float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
                w32 = torch.randn((n, k), device=DEV) * k ** -0.5
                w2 = ops.pack_linear_weight(w32, f16, split=True)
                w8, ws = ops.pack_linear_weight_f8(w32)
                bias = torch.randn(n, device=DEV)
                o32 = torch.randn((M, n), device=DEV) if kw.get("res") else None
                olp = torch.empty((M, n), dtype=f16, device=DEV) if kw.get("out") == "lp" else None
                common = dict(bias=bias, act=kw.get("act"), res_f32=o32, out_f32=o32, out_lp=olp)
                fns = {"w2 (two fp16 planes), skewed start": lambda: ops.gemm(rows[:, :k], w2, split="w2", kernel_sel=9, **common),
                       "w2 (two fp16 planes)": lambda: ops.gemm(rows[:, :k], w2, split="w2", kernel_sel=6, **common),
                       "w2f8 (low plane in fp8), skewed start": lambda: ops.gemm(rows, w8, split="w2f8", w_scale=ws, kernel_sel=9, **common),
                       "w2f8 (low plane in fp8)": lambda: ops.gemm(rows, w8, split="w2f8", w_scale=ws, **common)}
                res = {nm_: [] for nm_ in fns}
                for rnd_ in range(4):
                    for nm_, f in fns.items():
                        if rnd_ == 0:
                            f()
                        res[nm_].append(time_ms(f, rounds=3, inner=3)[0])
                base = None
                for nm_ in fns:
                    ms = sorted(res[nm_])[len(res[nm_]) // 2]
                    base = base or ms
                    print(json.dumps({"kernel": "gemm", "role": nm, "form": nm_, "M": M, "N": n, "K": k, "ms": round(ms, 3), "tflops_algorithmic": round(2.0 * M * n * k / ms / 1e9, 1),
                                      "speedup_vs_w2": round(base / ms, 3)}), flush=True)
        sys.exit(0)
    if args.what == "convheads":  # the DPT-head convolutions and the encoder's QKV + RoPE-2D as the N = 320 forward runs them (fp16, X3 / W2; 25-view head chunks)
        f16 = torch.float16
        bench_conv(f16, 25, 256, 256, 256, 128, "head0 x3", split="x3", sels=(0,))
        bench_conv(f16, 25, 512, 512, 128, 128, "head2 x3", split="x3", sels=(0,))
        bench_conv(f16, 25, 256, 256, 256, 256, "refinenet1 rcu x3", split="x3", sels=(0,))
        bench_conv(f16, 25, 128, 128, 256, 256, "refinenet2 rcu x3", split="x3", sels=(0,))
        bench_conv(f16, 25, 64, 64, 256, 256, "refinenet3 rcu x3", split="x3", sels=(0,))
        bench_qkv(f16, 128 * 1024, 1024, 1024, sels=(0,))
        sys.exit(0)
    if args.what == "gemmpersist":  # persistent grid (kernel_sel 2) next to one tile per workgroup (5) on the model's shapes
        for dt in (torch.bfloat16, torch.float16):
            for M in (40960, 102400, 327680):
                bench_gemm(dt, M, 1024, 1024, f"proj+res M={M}", res=True, sels=(2, 5))
                bench_gemm(dt, M, 4096, 1024, f"fc1+gelu M={M}", act="gelu", out="lp", sels=(2, 5))
                bench_gemm(dt, M, 1024, 4096, f"fc2+res M={M}", res=True, sels=(2, 5))
                bench_qkv(dt, M, 1024, M, sels=(2, 5))
            bench_conv(dt, 8, 128, 128, 256, 256, "refinenet1 rcu", sels=(2, 5))
            bench_conv(dt, 8, 256, 256, 256, 128, "head0", sels=(4,))
            bench_conv(dt, 8, 512, 512, 128, 128, "head2", sels=(4,))
        bench_gemm(torch.float16, 102400, 4096, 1024, "fc1+gelu w2", act="gelu", out="lp", split="w2", sels=(2, 5))
        bench_conv(torch.float16, 8, 128, 128, 256, 256, "refinenet1 rcu x3", split="x3", sels=(2, 5))
        sys.exit(0)
    if args.what == "exactattn":  # precision "exact": the FMA-pipe fp32 attention against its matrix-pipe form (three-plane products)
        for T in [int(x) * 1024 for x in args.views.split(",")]:
            H = 16
            qkv = (torch.randn((T, 3 * H * 64), device=DEV) * 1.2)
            res = {}
            for name, thr in (("fma", 1 << 62), ("mfma", 0)):
                if name == "fma" and T > 40960:
                    continue  # minutes
                ops.ATTN_F32_MFMA_MIN_KEYS = thr
                f = lambda: ops.attention_f32(qkv, H, 1, T, 0.125, torch.float16, want_f32=True)  # noqa: E731
                res[name] = f()[2]
                med, mn = time_ms(f, rounds=2, inner=1)
                print(json.dumps({"kernel": "attention_f32", "form": name, "T": T, "heads": H, "ms": round(med, 2), "tflops_algorithmic": round(4.0 * T * T * 64 * H / med / 1e9, 1)}), flush=True)
            if len(res) == 2:
                print(json.dumps({"kernel": "attention_f32", "T": T, "rel_l2_between_forms": float((res["mfma"] - res["fma"]).norm() / res["fma"].norm())}), flush=True)
        sys.exit(0)
    if args.what == "attnhdsmall":  # where the generated head_dim-80 / 128 kernels overtake the generic one: short sequences (sel 1 = generic, 2 = generated)
        for hd in (80, 128):
            for T in (128, 256, 512, 1024, 2048, 4096):
                bench_attn_head_dim(torch.float16, 0, hd, sels=(1, 2), T=T)
        sys.exit(0)
    if args.what == "attnhd":  # generic head_dim kernel next to the tuned head_dim-64 path
        for hd in (64, 80, 128):
            for nv in [int(x) for x in args.views.split(",")]:
                for d in args.attn_dtypes.split(","):
                    bench_attn_head_dim(torch.float16 if d == "fp16" else torch.bfloat16, nv, hd)
        sys.exit(0)
    if args.what == "attnsel":
        for nv in [int(x) for x in args.views.split(",")]:
            for d in (args.attn_dtypes.split(",")):
                bench_attn_sel({"bf16": torch.bfloat16, "fp16": torch.float16}[d], nv, sels=tuple(int(x) for x in args.sels.split(",")), tokens_per_view=args.tokens_per_view)
        sys.exit(0)
    if args.what == "labtime":
        M = 40 * 1024
        for bits in (128, 384, 640):
            bench_labtime(M, 4096, 1024, bits=bits)
            bench_labtime(M, 4096, 1024, act="gelu", bits=bits)
            bench_labtime(M, 1024, 4096, bits=bits)
            bench_labtime(8 * M, 4096, 1024, act="gelu", bits=bits)
        sys.exit(0)
    if args.what == "labphase":  # plain timing (no stamps): baseline / 2 phases / 4 phases
        M = 40 * 1024
        dt = torch.bfloat16
        for (m, n, k, act) in ((M, 4096, 1024, None), (M, 4096, 1024, "gelu"), (M, 1024, 1024, None), (M, 1024, 4096, None), (8 * M, 4096, 1024, "gelu"), (8 * M, 1024, 1024, None)):
            a = torch.randn((m, k), device=DEV).to(dt)
            w = ops.pack_linear_weight(torch.randn((n, k), device=DEV) * k ** -0.5, dt)
            bias = torch.randn(n, device=DEV)
            olp = torch.empty((m, n), dtype=dt, device=DEV)
            for bits in (0, 256, 512):
                f = lambda: ops.gemm(a, w, bias=bias, act=act, out_lp=olp, kernel_sel=16 + bits)
                f()
                ms = sorted(time_ms(f, rounds=3, inner=3)[0] for _ in range(3))[1]
                print(json.dumps({"kernel": "gemm256_lab dephase", "bits": bits, "M": m, "N": n, "K": k, "act": act, "ms": round(ms, 3),
                                  "tflops": round(2.0 * m * n * k / ms / 1e9, 1)}), flush=True)
        sys.exit(0)
    if "lab" in args.what:
        bench_lab(40 * 1024, 4096, 1024)
        bench_lab(40 * 1024, 1024, 4096)
        sys.exit(0)
    if "gemm" in args.what:
        M = 40 * 1024
        bench_gemm(dt, M, 1024, 1024, "proj+res", res=True)
        bench_gemm(dt, M, 4096, 1024, "fc1+gelu", act="gelu", out="lp")
        bench_gemm(dt, M, 1024, 4096, "fc2+res", res=True)
        bench_gemm(dt, M, 1024, 768, "patch_embed")
        bench_gemm(dt, 8 * M, 1024, 1024, "proj+res N=320", res=True)
        bench_gemm(dt, 8 * M, 4096, 1024, "fc1+gelu N=320", act="gelu", out="lp")
        bench_qkv(dt, M, 1024, 1024)
        bench_qkv(dt, M, 1024, M)
        bench_gemm(torch.float16, M, 1024, 1024, "proj+res w2", res=True, split="w2")
        bench_gemm(torch.float16, M, 4096, 1024, "fc1+gelu w2", act="gelu", out="lp", split="w2")
        bench_gemm(torch.float16, M, 1024, 1024, "proj+res x3", res=True, split="x3")
    if "dptfinal" in args.what:
        bench_dpt_final()
    if "focal" in args.what:
        bench_focal()
    if "pnp" in args.what:
        bench_pnp()
    if "image" in args.what:
        bench_image()
    if "align" in args.what:
        bench_align()
    if "conv" in args.what:
        bench_conv(dt, 8, 128, 128, 256, 256, "refinenet1 rcu")
        bench_conv(dt, 8, 256, 256, 256, 128, "head0")
        bench_conv(dt, 8, 512, 512, 128, 128, "head2")
        bench_conv(dt, 8, 64, 64, 256, 256, "refinenet2 rcu")
        bench_conv(dt, 8, 32, 32, 768, 768, "act3 s2", stride=2)
        bench_conv(torch.float16, 8, 128, 128, 256, 256, "refinenet1 rcu x3", split="x3")
