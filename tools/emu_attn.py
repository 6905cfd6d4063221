"""Run the generated attention kernel (fast3r_amd/csrc/asm/attn_gen.py) in the CPU emulator against a float64 softmax reference.

python tools/emu_attn.py [--dtype f16|bf16] [--tiles N] [--heads H] [--spike] [--rowsum dot2c|add]
"""
import argparse
import os
import struct
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", "fast3r_amd", "csrc", "asm"))
sys.path.insert(0, here)
import attn_gen  # noqa: E402
from gfx950_emu import Memory, Workgroup, f32_to_half, half_to_f32  # noqa: E402


def run_case(dtype="f16", n_tiles=3, n_heads=2, wgs=((0, 1, 0),), spike=False, rowsum="add", seed=0, batch=1, kv_shift=0, q_blocks=1,
             gen_kwargs=None):
    rng = np.random.default_rng(seed)
    tq, tk = 512 * q_blocks, 64 * n_tiles
    D = n_heads * 64
    kv_heads = n_heads >> kv_shift
    Dk = kv_heads * 64
    LOG2E = 1.4426950408889634
    scale = 0.125
    q = rng.standard_normal((batch, tq, D)).astype(np.float32) * 1.5
    k = rng.standard_normal((batch, tk, Dk)).astype(np.float32) * 1.5
    v = rng.standard_normal((batch, tk, Dk)).astype(np.float32)
    if spike:  # a key far above the rest in a late tile: forces the lazy reference to move mid-stream
        k[:, tk - 40, :] = q[:, 7, :Dk] * 3.0 if kv_shift == 0 else k[:, tk - 40, :] * 6.0
    qh = f32_to_half(q * (scale * LOG2E), dtype)            # pre-scaled, as the QKV epilogue writes it
    kh = f32_to_half(k, dtype)
    ldvt = tk
    vth = np.zeros((batch, Dk, ldvt), np.uint16)
    vth[:, :, :tk] = f32_to_half(np.transpose(v, (0, 2, 1)), dtype)
    mem = Memory()
    a_q, a_k, a_vt = mem.alloc(qh), mem.alloc(kh), mem.alloc(vth)
    o = np.full((batch, tq, D), 0x7E00, np.uint16)
    a_o = mem.alloc(o)
    karg = struct.pack("<QQQQIIIIIIQQQQII", a_q, a_k, a_vt, a_o, D * 2, Dk * 2, ldvt * 2, D * 2, n_tiles, 0,
                       tq * D * 2, tk * Dk * 2, Dk * ldvt * 2, tq * D * 2, kv_shift, 0)
    assert len(karg) == attn_gen.ARG_SIZE
    a_arg = mem.alloc(np.frombuffer(karg, np.uint8))
    g = attn_gen.AttnGen(dtype, rowsum=rowsum, **(gen_kwargs or {}))
    prog = g.build()
    problems = prog.check_hazards()
    assert not problems, "\n".join(problems[:20])
    worst = 0.0
    for wg in wgs:
        w = Workgroup(prog, mem, a_arg, wg, 4, attn_gen.LDS_BYTES, dtype)
        steps = w.run()
        og = mem.get(a_o, np.uint16, (batch, tq, D))
        x, head, b = wg
        kvh = head >> kv_shift
        qf = half_to_f32(qh[b, x * 512:(x + 1) * 512, head * 64:(head + 1) * 64], dtype).astype(np.float64)
        kf = half_to_f32(kh[b, :, kvh * 64:(kvh + 1) * 64], dtype).astype(np.float64)
        vf = half_to_f32(vth[b, kvh * 64:(kvh + 1) * 64, :tk], dtype).astype(np.float64).T
        s = qf @ kf.T
        p = np.exp2(s - s.max(axis=1, keepdims=True))
        ref = (p @ vf) / p.sum(axis=1, keepdims=True)
        got = half_to_f32(og[b, x * 512:(x + 1) * 512, head * 64:(head + 1) * 64], dtype).astype(np.float64)
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        worst = max(worst, err)
        print(f"wg {wg}: {steps} instructions, rel-L2 {err:.3e}, max abs {np.abs(got - ref).max():.3e}, nan {np.isnan(got).sum()}")
    return worst


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--tiles", type=int, default=3)
    ap.add_argument("--heads", type=int, default=2)
    ap.add_argument("--spike", action="store_true")
    ap.add_argument("--rowsum", default="add")
    ap.add_argument("--cvt", default="rne")
    a = ap.parse_args()
    run_case(a.dtype, a.tiles, a.heads, spike=a.spike, rowsum=a.rowsum, gen_kwargs=dict(cvt=a.cvt))
