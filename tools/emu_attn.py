"""Run the generated attention kernel (fast3r_amd/csrc/asm/attn_gen.py) in the CPU emulator against a float64 softmax reference.

python tools/emu_attn.py [--dtype f16|bf16] [--tiles N[,N..]] [--heads H] [--spike] [--rowsum pkadd|add|dot2c] [--split-state]
  --tiles a,b,c   K/V arrive as segments of a, b, c tiles (the view-sharded layout)
  --split-state   two launches: segment 0 with state_out, the remaining segments with state_in (the multi-GPU two-launch form)
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
from gfx950_emu import Memory, Workgroup, f32_to_half, half_to_f32, fp8_e4m3_to_f64, f64_to_fp8_e4m3  # noqa: E402


def pack_args(q, o, ldq, ldk, ldvt, ldo, segs, q_bs=0, o_bs=0, kv_shift=0, flags=0, st_o=0, st_ml=0, k_bs=0, vt_bs=0, st_o_ld=0, st_ml_ld=0, tq=0, dbg=0, sched=0, n_work=0, nx=0, nxy=0, grid=0):
    """segs: [(k address, vt address, tiles)] -> the kernel argument block (byte strides)"""
    nt = sum(t for _, _, t in segs)
    b = struct.pack("<QQIIIIIIQQIIQQQQIIII", q, o, ldq, ldk, ldvt, ldo, nt, len(segs), q_bs, o_bs, kv_shift, flags, st_o, st_ml, k_bs, vt_bs,
                    st_o_ld, st_ml_ld, tq, 0)
    assert len(b) == attn_gen.ARG_SEG
    for i in range(8):
        k, vt, t = segs[i] if i < len(segs) else (0, 0, 0)
        b += struct.pack("<QQII", k, vt, t, 0)
    b += struct.pack("<Q", dbg)
    magic = lambda d: -(-(1 << 32) // d) if d > 1 else 0   # noqa: E731
    b += struct.pack("<QIIIIII", sched, n_work, nx, nxy, magic(nx), magic(nxy), grid)
    assert len(b) == attn_gen.ARG_SIZE
    return b


def run_case(dtype="f16", n_tiles=3, n_heads=2, wgs=((0, 1, 0),), spike=False, rowsum="pkadd", seed=0, batch=1, kv_shift=0, q_blocks=1,
             gen_kwargs=None, split_state=False, layout=2, tq=None, counters=None, head_dim=64, steal=0, qk_planes=1, errs=None, corr="f16", finish_state=False, qpw=None):
    """tq: number of query rows when it is not 512 * q_blocks (layout 2: the last workgroup may be partial; the buffers hold exactly tq
    rows, so a store past the end raises in the emulator's memory model).  counters: a list that receives the kernel's debug counters
    {re-base block entries, waves, tiles walked} (f3r_attn_args.dbg_counters; layout 2).  steal = G > 0: the work-stealing form -- G
    persistent workgroups share the launch's {next, done} counter (emulated one after the other: the first one takes every item, the others
    find the counter exhausted, the last one to leave zeroes it); `wgs` then lists the work items whose output is compared.
    qk_planes = 2: the three-product form (AttnGen(qk_planes=2), precision "robust"): Q and K rows hold [hi (64) | lo (64)] fp16 per head; the reference
    is float64 on hi + lo; errs (a list) also receives the distance to the reference computed from the hi planes alone (what one product gives).
    finish_state: ONE launch with state_out (what precision "robust" issues, batches included: sequence z owns state rows [z tq, (z + 1) tq)) and
    the output is computed here from the parked state, O / (l0 + l1), as f3r_attn_state_finish does.
    corr = "f8" (with qk_planes = 2): rows [hi fp16 | e4m3(hi) | e4m3(lo 2^12)] per head, the correction products on the block-scaled fp8 MFMA; the
    reference is float64 on EXACTLY those planes: q_hi k_hi + dq(q_lo8) dq(k_hi8) + dq(q_hi8) dq(k_lo8)."""
    rng = np.random.default_rng(seed)
    HD = head_dim
    g = attn_gen.AttnGen(dtype, rowsum=rowsum, head_dim=HD, qk_planes=qk_planes, corr=corr, qpw=qpw, **(gen_kwargs or {}))   # (layout: accepted for old call sites; there is one generator)
    WQ = g.WG_Q   # query rows of a workgroup (512 at head_dim 64, 256 otherwise)
    seg_tiles = list(n_tiles) if isinstance(n_tiles, (list, tuple)) else [n_tiles]
    tq, tk = (WQ * q_blocks if tq is None else tq), 64 * sum(seg_tiles)
    D = n_heads * HD
    kv_heads = n_heads >> kv_shift
    Dk = kv_heads * HD
    LOG2E = 1.4426950408889634
    scale = HD ** -0.5
    q = rng.standard_normal((batch, tq, D)).astype(np.float32) * 1.5
    k = rng.standard_normal((batch, tk, Dk)).astype(np.float32) * 1.5
    v = rng.standard_normal((batch, tk, Dk)).astype(np.float32)
    if spike:  # a key far above the rest in a late tile: forces the lazy reference to move mid-stream
        k[:, tk - 40, :] = q[:, 7, :Dk] * 3.0 if kv_shift == 0 else k[:, tk - 40, :] * 6.0
    qh = f32_to_half(q * (scale * LOG2E), dtype)            # pre-scaled, as the QKV epilogue writes it
    kh = f32_to_half(k, dtype)
    DQ, DKm = D, Dk                                          # widths of a q / k row in memory
    q_ref, k_ref = qh, kh
    if qk_planes == 2:   # rows [hi | lo] per head: lo = half(x - float(hi))
        def planes(x32, heads):
            hi = f32_to_half(x32, dtype)
            lo = f32_to_half(x32 - half_to_f32(hi, dtype), dtype)
            out = np.empty(x32.shape[:2] + (heads, 2, HD), np.uint16)
            out[:, :, :, 0] = hi.reshape(x32.shape[:2] + (heads, HD))
            out[:, :, :, 1] = lo.reshape(x32.shape[:2] + (heads, HD))
            return out.reshape(x32.shape[:2] + (heads * 2 * HD,)), hi, lo
        qmem, q_hi, q_lo = planes(q * (scale * LOG2E), n_heads)
        kmem, k_hi, k_lo = planes(k, kv_heads)
        q_ref, k_ref = (q_hi, q_lo), (k_hi, k_lo)
        if corr == "f8":   # the lo half of a head's row becomes [e4m3(hi) 64 B | e4m3(lo * 2^12) 64 B]
            def f8_rows(mem, hi, lo, heads):
                m = mem.reshape(mem.shape[:2] + (heads, 2, HD)).copy()
                hi8 = f64_to_fp8_e4m3(half_to_f32(hi, dtype).astype(np.float64)).reshape(hi.shape[:2] + (heads, HD))
                lo8 = f64_to_fp8_e4m3(half_to_f32(lo, dtype).astype(np.float64) * 4096.0).reshape(lo.shape[:2] + (heads, HD))
                both = np.concatenate([hi8, lo8], axis=-1).astype(np.uint8)              # 128 bytes
                m[:, :, :, 1] = np.ascontiguousarray(both).view(np.uint16)
                return m.reshape(mem.shape), fp8_e4m3_to_f64(hi8).reshape(hi.shape), (fp8_e4m3_to_f64(lo8) / 4096.0).reshape(lo.shape)
            qmem, q_hi8, q_lo8 = f8_rows(qmem, q_hi, q_lo, n_heads)
            kmem, k_hi8, k_lo8 = f8_rows(kmem, k_hi, k_lo, kv_heads)
        qh, kh = qmem, kmem
        DQ, DKm = 2 * D, 2 * Dk
    ldvt = 64 * max(seg_tiles)
    mem = Memory()
    a_q = mem.alloc(qh)
    segs, key0 = [], 0
    vth_all = f32_to_half(np.transpose(v, (0, 2, 1)), dtype)   # [batch][Dk][tk]
    for t in seg_tiles:
        n = 64 * t
        ks = np.ascontiguousarray(kh[:, key0:key0 + n])
        vts = np.zeros((batch, Dk, ldvt), np.uint16)
        vts[:, :, :n] = vth_all[:, :, key0:key0 + n]
        segs.append((mem.alloc(ks), mem.alloc(vts), t, n))
        key0 += n
    if batch > 1:
        assert len(segs) == 1
    o = np.full((batch, tq, D), 0x7E00, np.uint16)
    a_o = mem.alloc(o)
    st_o = mem.alloc(np.full((batch * tq, D), np.nan, np.float32))
    st_ml = mem.alloc(np.full((batch * tq, n_heads, 4), np.nan, np.float32))
    a_dbg = mem.alloc(np.zeros(56, np.uint32)) if counters is not None else 0
    nx = -(-tq // WQ)
    a_sched = mem.alloc(np.zeros(2, np.uint32)) if steal else 0
    sched_kw = dict(sched=a_sched, n_work=nx * n_heads * batch, nx=nx, nxy=nx * n_heads, grid=steal) if steal else {}
    common = dict(dbg=a_dbg, q_bs=tq * DQ * 2, o_bs=tq * D * 2, kv_shift=kv_shift, st_o=st_o, st_ml=st_ml, k_bs=segs[0][3] * DKm * 2, vt_bs=Dk * ldvt * 2,
                  st_o_ld=D * 4, st_ml_ld=n_heads * 16, tq=tq, **sched_kw)
    launches = []
    if split_state:
        assert len(segs) >= 2 and batch == 1
        launches.append(pack_args(a_q, a_o, DQ * 2, DKm * 2, ldvt * 2, D * 2, [s[:3] for s in segs[:1]], flags=attn_gen.FLAG_STATE_OUT, **common))
        launches.append(pack_args(a_q, a_o, DQ * 2, DKm * 2, ldvt * 2, D * 2, [s[:3] for s in segs[1:]], flags=attn_gen.FLAG_STATE_IN, **common))
    elif finish_state:
        launches.append(pack_args(a_q, a_o, DQ * 2, DKm * 2, ldvt * 2, D * 2, [s[:3] for s in segs], flags=attn_gen.FLAG_STATE_OUT, **common))
    else:
        launches.append(pack_args(a_q, a_o, DQ * 2, DKm * 2, ldvt * 2, D * 2, [s[:3] for s in segs], **common))
    prog = g.build()
    problems = prog.check_hazards()
    assert not problems, "\n".join(problems[:20])
    worst = 0.0
    if steal:
        for karg in launches:
            a_arg = mem.alloc(np.frombuffer(karg, np.uint8))
            for gidx in range(steal):
                Workgroup(prog, mem, a_arg, (gidx, 0, 0), 4, g.lds_bytes, dtype).run(max_steps=100_000_000)
                nxt, done = (int(v) for v in mem.get(a_sched, np.uint32, (2,)))
                last = gidx == steal - 1
                assert (nxt, done) == ((0, 0) if last else (nx * n_heads * batch + gidx + 1, gidx + 1)), (gidx, nxt, done)
    for wg in wgs:
        steps = 0
        for karg in ([] if steal else launches):
            a_arg = mem.alloc(np.frombuffer(karg, np.uint8))
            w = Workgroup(prog, mem, a_arg, wg, 4, g.lds_bytes, dtype)
            steps += w.run()
        og = mem.get(a_o, np.uint16, (batch, tq, D))
        if finish_state:   # the output from the parked state (rows z tq + r), rounded like the finishing pass rounds its hi plane
            so = mem.get(st_o, np.float32, (batch, tq, D)).astype(np.float64)
            ml = mem.get(st_ml, np.float32, (batch, tq, n_heads, 4)).astype(np.float64)
            l = np.repeat(ml[..., 1] + ml[..., 2], HD, axis=-1)
            og = f32_to_half((so / l).astype(np.float32), dtype)
        x, head, b = wg
        kvh = head >> kv_shift
        r1 = min(tq, (x + 1) * WQ)
        def f64(t, rows, hh):
            return half_to_f32(t[b, rows, hh * HD:(hh + 1) * HD], dtype).astype(np.float64)
        rows_q, all_k = slice(x * WQ, r1), slice(None)
        vf = half_to_f32(vth_all[b, kvh * HD:(kvh + 1) * HD, :], dtype).astype(np.float64).T

        def softmax_ref(qf, kf):
            s = qf @ kf.T
            p = np.exp2(s - s.max(axis=1, keepdims=True))
            return (p @ vf) / p.sum(axis=1, keepdims=True)
        if qk_planes == 2:
            qf = f64(q_ref[0], rows_q, head) + f64(q_ref[1], rows_q, head)
            kf = f64(k_ref[0], all_k, kvh) + f64(k_ref[1], all_k, kvh)
        else:
            qf, kf = f64(q_ref, rows_q, head), f64(k_ref, all_k, kvh)
        if corr == "f8":   # the scores the kernel computes, from its own planes
            hs = slice(head * HD, (head + 1) * HD)
            ks_ = slice(kvh * HD, (kvh + 1) * HD)
            s8 = (f64(q_ref[0], rows_q, head) @ f64(k_ref[0], all_k, kvh).T + q_lo8[b, rows_q, hs] @ k_hi8[b, :, ks_].T + q_hi8[b, rows_q, hs] @ k_lo8[b, :, ks_].T)
            p8 = np.exp2(s8 - s8.max(axis=1, keepdims=True))
            ref = (p8 @ vf) / p8.sum(axis=1, keepdims=True)
            if errs is not None:
                true = softmax_ref(qf, kf)
                errs.append(("f8 planes vs hi + lo", np.linalg.norm(ref - true) / np.linalg.norm(true)))
        else:
            ref = softmax_ref(qf, kf)
        got = half_to_f32(og[b, x * WQ:r1, head * HD:(head + 1) * HD], dtype).astype(np.float64)
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        if qk_planes == 2 and errs is not None and corr != "f8":
            ref1 = softmax_ref(f64(q_ref[0], rows_q, head), f64(k_ref[0], all_k, kvh))
            errs.append((err, np.linalg.norm(ref1 - ref) / np.linalg.norm(ref), np.linalg.norm(got - ref1) / np.linalg.norm(ref1)))
        worst = err if not np.isfinite(err) else max(worst, err)   # a NaN anywhere fails the case
        if not np.isfinite(worst):
            break
        print(f"wg {wg}: {steps} instructions, rel-L2 {err:.3e}, max abs {np.abs(got - ref).max():.3e}, nan {np.isnan(got).sum()}")
    if counters is not None:
        raw = mem.get(a_dbg, np.uint32, (8,))
        counters[:] = [int(raw[0]), int(raw[1])] + [int(raw[2 * i]) | (int(raw[2 * i + 1]) << 32) for i in (1, 2, 3)]   # entries, waves, tiles, cycles, ticks
    return worst


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--tiles", default="3")
    ap.add_argument("--heads", type=int, default=2)
    ap.add_argument("--spike", action="store_true")
    ap.add_argument("--rowsum", default="pkadd")
    ap.add_argument("--cvt", default="rne")
    ap.add_argument("--split-state", action="store_true")
    ap.add_argument("--layout", type=int, default=2)
    ap.add_argument("--head-dim", type=int, default=64)
    ap.add_argument("--qk-planes", type=int, default=1)
    ap.add_argument("--corr", default="f16")
    a = ap.parse_args()
    tiles = [int(x) for x in a.tiles.split(",")]
    run_case(a.dtype, tiles if len(tiles) > 1 else tiles[0], a.heads, spike=a.spike, rowsum=a.rowsum, gen_kwargs=dict(cvt=a.cvt), split_state=a.split_state, layout=a.layout, head_dim=a.head_dim, qk_planes=a.qk_planes, corr=a.corr)
