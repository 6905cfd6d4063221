"""Run the generated GEMM kernels (fast3r_amd/csrc/asm/gemm_gen.py) in the CPU emulator against float64 on the same rounded operands.

python tools/emu_gemm.py [--dtype f16|bf16] [--role f32|lp] [--tiles MxN] [--nk1 K-tiles per segment] [--segs 1|2] [--act none|gelu|relu] [--no-bias] [--no-res]
"""
import argparse
import math
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", "fast3r_amd", "csrc", "asm"))
sys.path.insert(0, here)
import gemm_gen  # noqa: E402
from gfx950_emu import Memory, Workgroup, f32_to_half, half_to_f32, fp8_e4m3_to_f64, f64_to_fp8_e4m3  # noqa: E402

ACTS = {"none": gemm_gen.ACT_NONE, "gelu": gemm_gen.ACT_GELU, "relu": gemm_gen.ACT_RELU}


def gelu64(x):
    return 0.5 * x * (1.0 + np.vectorize(math.erf)(x / math.sqrt(2.0)))


def run_case(dtype="f16", role="f32", ntm=1, ntn=1, nk1=4, segs=1, act="none", bias=True, res=True, wgs=None, seed=0, lda_pad=0, gen_kwargs=None, grid=None):
    """one launch of (ntm x ntn) tiles, K = 64 * nk1 per segment, `segs` K segments (2 = split weights hi | lo planes in one W row); emulates the
    workgroups `wgs` (default: all) and returns the worst max-abs error relative to the output scale"""
    rng = np.random.default_rng(seed)
    M, N, K1 = 256 * ntm, 256 * ntn, 64 * nk1
    lda = K1 + lda_pad
    a = rng.standard_normal((M, lda)).astype(np.float32)
    w = (rng.standard_normal((N, K1 * segs)) * K1 ** -0.5).astype(np.float32)
    if segs == 2:  # a lo plane is ~2^-11 of its hi plane
        w[:, K1:] *= 2.0 ** -11
    ah, wh = f32_to_half(a, dtype), f32_to_half(w, dtype)
    bvec = (rng.standard_normal(N) * 0.7).astype(np.float32) if bias else None
    x = (rng.standard_normal((M, N)) * 2.0).astype(np.float32)
    mem = Memory()
    a_a, a_w = mem.alloc(ah), mem.alloc(wh)
    a_b = mem.alloc(bvec) if bias else 0
    esize = 4 if role == "f32" else 2
    if role == "f32":
        out0 = x.copy() if res else np.full((M, N), np.nan, np.float32)
        a_o = mem.alloc(out0)
        a_r = a_o if res else 0       # in place, as the model calls it (x += proj(..))
    else:
        a_o = mem.alloc(np.full((M, N), 0x7E00, np.uint16))
        a_r = 0
    karg, grid = gemm_gen.pack_args(a_a, a_w, a_b, a_r, a_o, lda * 2, K1 * segs * 2, N * 4, N * esize, nk1 * segs, nk1, ntm, ntn, ACTS[act], grid=grid)
    g = gemm_gen.GemmGen(dtype, role, **(gen_kwargs or {}))
    prog = g.build()
    problems = prog.check_hazards()
    assert not problems, "\n".join(problems[:20])
    a_arg = mem.alloc(np.frombuffer(karg, np.uint8))
    steps = 0
    n_tiles = 0
    for wg in (range(grid) if wgs is None else wgs):
        n_tiles += len(range(wg, ntm * ntn, grid))
        steps += Workgroup(prog, mem, a_arg, (wg, 0, 0), 4, g.lds_bytes, dtype).run()
    af = half_to_f32(ah, dtype).astype(np.float64)[:, :K1]
    wf = half_to_f32(wh, dtype).astype(np.float64)
    ref = af @ wf[:, :K1].T
    if segs == 2:
        ref += af @ wf[:, K1:].T
    if bias:
        ref += bvec.astype(np.float64)
    if role == "f32":
        if res:
            ref += x.astype(np.float64)
        got = mem.get(a_o, np.float32, (M, N)).astype(np.float64)
    else:
        ref = {"none": lambda v: v, "relu": lambda v: np.maximum(v, 0.0), "gelu": gelu64}[act](ref)
        got = half_to_f32(mem.get(a_o, np.uint16, (M, N)), dtype).astype(np.float64)
    if wgs is not None:  # only the emulated tiles were written: compare those (the tile map is the kernel's own; find them by what changed)
        done = ~np.isnan(got) if role == "f32" and not res else None
        if role == "lp":
            done = mem.get(a_o, np.uint16, (M, N)) != 0x7E00
        elif res:
            done = got != x.astype(np.float64)
        assert done.sum() == 65536 * n_tiles, ("tiles written", int(done.sum()), n_tiles)
        got, ref = got[done], ref[done]
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max()) / scale
    print(f"{g.name}: {ntm}x{ntn} tiles, K = {segs} x {K1}, act {act}, bias {bias}, res {res}: {steps} instructions, max err / scale = {err:.3e}, nan {int(np.isnan(got).sum())}")
    return err


def pack_rows_f8(x16, x8):
    """rows [K fp16 | K fp8] (3 K bytes) as the f8 kernels stream them"""
    M, K = x16.shape
    out = np.zeros((M, 3 * K), np.uint8)
    out[:, :2 * K] = np.ascontiguousarray(x16).view(np.uint8).reshape(M, 2 * K)
    out[:, 2 * K:] = x8
    return out


def run_case_f8(role="f32", ntm=1, ntn=1, nk16=4, act="none", bias=True, res=True, wgs=None, seed=0, grid=None, outliers=True, out8=False):
    """the split-precision GEMM with its low plane in fp8 (GemmGen(f8=True), F3R_SPLIT_W2F8): out = epilogue(A16 W_hi^T + A8 (W_lo8 2^-s_n)^T),
    A rows = [K fp16 | K fp8 (e4m3 of the same numbers, clamped to +-448)], W rows = [K fp16 hi | K fp8 e4m3((W - hi) 2^s_n)], one power-of-two
    scale per output channel handed to the kernel as E8M0 words.  Reference: float64 on exactly these planes."""
    rng = np.random.default_rng(seed)
    M, N, K = 256 * ntm, 256 * ntn, 64 * nk16
    assert K % 128 == 0
    nk8 = K // 128
    a = (rng.standard_normal((M, K)) * 2.0).astype(np.float32)
    if outliers:
        a[rng.integers(0, M, 40), rng.integers(0, K, 40)] *= 400.0      # beyond the fp8 range: clamped in the fp8 copy
    w = (rng.standard_normal((N, K)) * K ** -0.5).astype(np.float32)
    w *= np.exp2(rng.integers(-6, 3, size=(N, 1))).astype(np.float32)      # rows of very different scale: one scale per output channel
    a16, w16 = f32_to_half(a, "f16"), f32_to_half(w, "f16")
    a8 = f64_to_fp8_e4m3(half_to_f32(a16, "f16").astype(np.float64))
    wlo = w.astype(np.float64) - half_to_f32(w16, "f16").astype(np.float64)
    amax = np.maximum(np.abs(wlo).max(axis=1), 1e-30)
    s = np.floor(np.log2(224.0 / amax)).astype(np.int64)                   # max |lo| 2^s in [112, 224]
    s = np.clip(s, -100, 120)
    w8 = f64_to_fp8_e4m3(wlo * np.exp2(s)[:, None])
    e8m0 = (127 - s).astype(np.uint32)
    wsc = (e8m0 | (e8m0 << 8) | (e8m0 << 16) | (e8m0 << 24)).astype(np.uint32)
    bvec = (rng.standard_normal(N) * 0.7).astype(np.float32) if bias else None
    x = (rng.standard_normal((M, N)) * 2.0).astype(np.float32)
    mem = Memory()
    a_a, a_w, a_s = mem.alloc(pack_rows_f8(a16, a8)), mem.alloc(pack_rows_f8(w16, w8)), mem.alloc(wsc)
    a_b = mem.alloc(bvec) if bias else 0
    esize = 4 if role == "f32" else 2
    if role == "f32":
        a_o = mem.alloc(x.copy() if res else np.full((M, N), np.nan, np.float32))
        a_r = a_o if res else 0
    elif out8:   # rows [N fp16 | N fp8]: the lp role's GELU epilogue also writes the fp8 copy
        rows0 = np.zeros((M, 3 * N), np.uint8)
        rows0[:, :2 * N] = np.full((M, N), 0x7E00, np.uint16).view(np.uint8).reshape(M, 2 * N)
        rows0[:, 2 * N:] = 0x7F
        a_o = mem.alloc(rows0)
        a_r = 0
    else:
        a_o = mem.alloc(np.full((M, N), 0x7E00, np.uint16))
        a_r = 0
    nk = nk16 + nk8
    ldo_b = 3 * N if (role == "lp" and out8) else N * esize
    karg, grid = gemm_gen.pack_args(a_a, a_w, a_b, a_r, a_o, 3 * K, 3 * K, N * 4, ldo_b, nk, nk, ntm, ntn, ACTS[act], grid=grid, wscale=a_s, nk8=nk8,
                                    out8_off=2 * N if (role == "lp" and out8) else 0)
    g = gemm_gen.GemmGen("f16", role, f8=True)
    prog = g.build()
    problems = prog.check_hazards()
    assert not problems, "\n".join(problems[:20])
    a_arg = mem.alloc(np.frombuffer(karg, np.uint8))
    steps = 0
    for wg in (range(grid) if wgs is None else wgs):
        steps += Workgroup(prog, mem, a_arg, (wg, 0, 0), 4, g.lds_bytes, "f16").run(max_steps=40_000_000)
    ref = half_to_f32(a16, "f16").astype(np.float64) @ half_to_f32(w16, "f16").astype(np.float64).T
    ref += fp8_e4m3_to_f64(a8) @ (fp8_e4m3_to_f64(w8) * np.exp2(-s.astype(np.float64))[:, None]).T
    exact = a.astype(np.float64) @ w.astype(np.float64).T
    single = half_to_f32(a16, "f16").astype(np.float64) @ half_to_f32(w16, "f16").astype(np.float64).T
    if bias:
        ref += bvec.astype(np.float64)
    if role == "f32":
        if res:
            ref += x.astype(np.float64)
        got = mem.get(a_o, np.float32, (M, N)).astype(np.float64)
    else:
        ref = {"none": lambda v: v, "relu": lambda v: np.maximum(v, 0.0), "gelu": gelu64}[act](ref)
        if out8:
            rows = mem.get(a_o, np.uint8, (M, 3 * N))
            got = half_to_f32(np.ascontiguousarray(rows[:, :2 * N]).view(np.uint16).reshape(M, N), "f16").astype(np.float64)
            got8 = fp8_e4m3_to_f64(rows[:, 2 * N:])
            want8 = fp8_e4m3_to_f64(f64_to_fp8_e4m3(ref))
            bad = np.abs(got8 - want8) > 0.13 * np.maximum(np.abs(want8), 2.0 ** -9)   # one e4m3 step (fp32 GELU vs float64 may round the other way)
            assert act != "gelu" or bad.mean() < 1e-3, ("fp8 copy", float(bad.mean()))
            assert act == "gelu" or (rows[:, 2 * N:] == 0x7F).all(), "an fp8 copy without GELU"
        else:
            got = half_to_f32(mem.get(a_o, np.uint16, (M, N)), "f16").astype(np.float64)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max()) / scale
    a_exact = half_to_f32(a16, "f16").astype(np.float64) @ w.astype(np.float64).T   # what W2 is after: the weight's rounding removed
    w_err = float(np.abs((ref - (bvec.astype(np.float64) if bias else 0) - (x if (role == "f32" and res) else 0)) - a_exact).max() / np.abs(a_exact).max()) if act == "none" else None
    w_err1 = float(np.abs(single - a_exact).max() / np.abs(a_exact).max())
    print(f"{g.name}: {ntm}x{ntn} tiles, K = {K} (fp16) + {K} (fp8), act {act}, bias {bias}, res {res}: {steps} instructions, max err / scale = {err:.3e}; "
          f"weight-rounding error left {w_err if w_err is None else format(w_err, '.2e')} (single fp16 plane: {w_err1:.2e})")
    return err


def rope2d_ref(x, seq_len, rope_w, cos, sin):
    """RoPE-2D as f3r_gemm_epi.h applies it (pos_embed.py:162-183): per 64-wide head, dims i, i + 16 of the first 32 rotate by angle i of the token's
    row position, of the last 32 by its column position; x [T][D] float64"""
    T_, D = x.shape
    out = x.copy()
    pos = np.arange(T_) % seq_len
    py, px = pos // rope_w, pos % rope_w
    for h0 in range(0, D, 64):
        for half, p in ((0, py), (1, px)):
            a = x[:, h0 + 32 * half:h0 + 32 * half + 16]
            b = x[:, h0 + 32 * half + 16:h0 + 32 * half + 32]
            c, s = cos[p].astype(np.float64), sin[p].astype(np.float64)
            out[:, h0 + 32 * half:h0 + 32 * half + 16] = a * c - b * s
            out[:, h0 + 32 * half + 16:h0 + 32 * half + 32] = b * c + a * s
    return out


def run_qkv_case(dtype="f16", n_seq=2, seq_tiles=1, d_tiles=1, nk1=4, segs=1, scale=0.231, seed=0, grid=None, rope_w=0):
    """the decoder-style QKV projection as the product issues it (f3r_gemm_asm.hip): launch 1 = q | k columns of the fused weight into two
    buffers (output segments, ACT_SCALE on the q segment); launch 2 = V^T with the operand roles swapped (kernel A = the v weight rows incl.
    their lo plane, kernel W = the activations, wrapping per K segment; bias indexed by the output ROW; one output segment per sequence)"""
    rng = np.random.default_rng(seed)
    T_, D, K1 = 256 * seq_tiles * n_seq, 256 * d_tiles, 64 * nk1
    S_ = 256 * seq_tiles
    x = rng.standard_normal((T_, K1)).astype(np.float32)
    w = (rng.standard_normal((3 * D, K1 * segs)) * K1 ** -0.5).astype(np.float32)
    if segs == 2:
        w[:, K1:] *= 2.0 ** -11
    b = (rng.standard_normal(3 * D) * 0.5).astype(np.float32)
    xh, wh = f32_to_half(x, dtype), f32_to_half(w, dtype)
    mem = Memory()
    a_x, a_w, a_b = mem.alloc(xh), mem.alloc(wh), mem.alloc(b)
    a_q = mem.alloc(np.full((T_, D), 0x7E00, np.uint16))
    a_k = mem.alloc(np.full((T_, D), 0x7E00, np.uint16))
    ldvt = S_ + 64
    a_vt = mem.alloc(np.full((n_seq, D, ldvt), 0x7E00, np.uint16))
    g = gemm_gen.GemmGen(dtype, "lp")
    prog = g.build()
    assert not prog.check_hazards()
    nk = nk1 * segs
    ldw_b = K1 * segs * 2
    rope = None
    if rope_w:   # the encoder's form: RoPE-2D in the epilogue of the q | k launch (tokens of a sequence on a grid rope_w wide)
        n_pos = max(rope_w, -(-S_ // rope_w))
        ang = np.arange(n_pos)[:, None] * (100.0 ** (-np.arange(16) / 16.0))[None, :]
        cos_t, sin_t = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
        rope = (mem.alloc(cos_t), mem.alloc(sin_t), S_, rope_w)
    # launch 1: q | k
    karg, gr = gemm_gen.pack_args(a_x, a_w, a_b, 0, a_q, K1 * 2, ldw_b, 0, D * 2, nk, nk1, T_ // 256, 2 * d_tiles, gemm_gen.ACT_ROPE if rope else gemm_gen.ACT_SCALE, grid=grid,
                                  seg_stride=a_k - a_q, tps=d_tiles, scale=scale, rope=rope)
    arg = mem.alloc(np.frombuffer(karg, np.uint8))
    for wg in range(gr):
        Workgroup(prog, mem, arg, (wg, 0, 0), 4, g.lds_bytes, dtype).run()
    # launch 2: V^T = W_v X^T, roles swapped
    karg, gr = gemm_gen.pack_args(a_w + 2 * D * ldw_b, a_x, a_b + 2 * D * 4, 0, a_vt, ldw_b, K1 * 2, 0, ldvt * 2, nk, nk, d_tiles, T_ // 256, gemm_gen.ACT_NONE,
                                  grid=grid, seg_stride=D * ldvt * 2, tps=seq_tiles, flags=gemm_gen.FLAG_BIAS_ON_M, nk1_w=nk1)
    arg = mem.alloc(np.frombuffer(karg, np.uint8))
    for wg in range(gr):
        Workgroup(prog, mem, arg, (wg, 0, 0), 4, g.lds_bytes, dtype).run()
    xf = half_to_f32(xh, dtype).astype(np.float64)
    wf = half_to_f32(wh, dtype).astype(np.float64)
    ref = xf @ wf[:, :K1].T + (xf @ wf[:, K1:].T if segs == 2 else 0.0) + b.astype(np.float64)
    q = half_to_f32(mem.get(a_q, np.uint16, (T_, D)), dtype).astype(np.float64)
    k = half_to_f32(mem.get(a_k, np.uint16, (T_, D)), dtype).astype(np.float64)
    vt = half_to_f32(mem.get(a_vt, np.uint16, (n_seq, D, ldvt)), dtype).astype(np.float64)
    pad = mem.get(a_vt, np.uint16, (n_seq, D, ldvt))[:, :, S_:]
    assert (pad == 0x7E00).all(), "V^T padding columns were written"
    v = np.transpose(vt[:, :, :S_], (0, 2, 1)).reshape(T_, D)
    if rope:
        ref[:, :D] = rope2d_ref(ref[:, :D], S_, rope_w, cos_t, sin_t)
        ref[:, D:2 * D] = rope2d_ref(ref[:, D:2 * D], S_, rope_w, cos_t, sin_t)
    errs = [float(np.abs(q - ref[:, :D] * np.float32(scale)).max() / np.abs(ref[:, :D] * scale).max()), float(np.abs(k - ref[:, D:2 * D]).max() / np.abs(ref).max()),
            float(np.abs(v - ref[:, 2 * D:]).max() / np.abs(ref).max())]
    print(f"qkv {dtype}: {n_seq} x {S_} tokens, D = {D}, K = {segs} x {K1}: q {errs[0]:.3e} k {errs[1]:.3e} v {errs[2]:.3e}")
    return max(errs)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--role", default="f32")
    ap.add_argument("--tiles", default="1x1")
    ap.add_argument("--nk1", type=int, default=4)
    ap.add_argument("--grid", type=int, default=None)
    ap.add_argument("--segs", type=int, default=1)
    ap.add_argument("--act", default="none")
    ap.add_argument("--no-bias", action="store_true")
    ap.add_argument("--no-res", action="store_true")
    ap.add_argument("--wgs", default="")
    a = ap.parse_args()
    tm, tn = (int(v) for v in a.tiles.split("x"))
    run_case(a.dtype, a.role, tm, tn, a.nk1, a.segs, a.act, not a.no_bias, not a.no_res, wgs=[int(v) for v in a.wgs.split(",")] if a.wgs else None, grid=a.grid)
