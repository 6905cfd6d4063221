"""Lane-exact CPU emulator for the subset of gfx950 instructions the generators under fast3r_amd/csrc/asm/ emit.

Test infrastructure (like oracle/): it executes the SAME instruction list that is printed for the assembler, one workgroup at a
time, so register clashes, pipeline indexing, address arithmetic and missing waits are found without a GPU.  What it encodes about
the hardware (MFMA operand layouts, LDS-DMA placement) is what the HIP kernels in fast3r_amd/csrc/ rely on and the GPU tests pin.

Asynchrony is modelled pessimistically:
  * a ds_read / global_load result reaches its registers only when an s_waitcnt retires it; until then the registers hold a poison
    pattern (a NaN), so a consumer placed before the wait produces NaNs;
  * LDS-DMA data lands in LDS only when the issuing wave's vmcnt wait retires it; other waves see it after that (waves of a
    workgroup run one after the other between barriers, so a read that is not ordered by a barrier sees stale data for some wave).
"""
import numpy as np

import sys
import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fast3r_amd", "csrc", "asm"))
from isa import Label, LabelRef, Lit, Neg, Reg, Special  # noqa: E402

POISON = np.uint32(0x7FC0DEAD)
MASK64 = (1 << 64) - 1


def f32(u):
    return u.view(np.float32)


def u32(f):
    return np.asarray(f, dtype=np.float32).view(np.uint32)


def half_to_f32(h16, dtype):
    h16 = h16.astype(np.uint16)
    if dtype == "f16":
        return h16.view(np.float16).astype(np.float32)
    return (h16.astype(np.uint32) << 16).view(np.float32)


def f32_to_half(x, dtype):
    x = np.asarray(x, dtype=np.float32)
    if dtype == "f16":
        with np.errstate(over="ignore"):
            return x.astype(np.float16).view(np.uint16)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)  # round to nearest even
    nan = np.isnan(x)
    r[nan] = 0x7FC0
    return r


class Memory:
    """flat global memory made of named numpy byte buffers at fake 64-bit addresses"""

    def __init__(self):
        self.bufs = []
        self.next = 0x7F0000000000

    def alloc(self, arr):
        b = np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()
        base = self.next
        self.next += (len(b) + 0xFFFF) & ~0xFFFF
        self.bufs.append((base, b))
        return base

    def find(self, addr, n):
        for base, b in self.bufs:
            if base <= addr and addr + n <= base + len(b):
                return b, addr - base
        raise RuntimeError(f"global access out of bounds: 0x{addr:x} (+{n})")

    def read(self, addr, n):
        b, o = self.find(int(addr), n)
        return b[o:o + n]

    def write(self, addr, data):
        b, o = self.find(int(addr), len(data))
        b[o:o + len(data)] = data

    def get(self, base, dtype, shape):
        for bb, b in self.bufs:
            if bb == base:
                n = int(np.prod(shape)) * np.dtype(dtype).itemsize
                return b[:n].view(dtype).reshape(shape)
        raise KeyError(base)


class Wave:
    def __init__(self, wg, wid, items, labels, dtype):
        self.wg, self.wid, self.items, self.labels, self.dtype = wg, wid, items, labels, dtype
        self.v = np.zeros((256, 64), np.uint32)
        self.a = np.zeros((256, 64), np.uint32)
        self.s = np.zeros(128, np.uint32)
        self.vcc = 0
        self.exec = MASK64   # honoured by vector register writes, v_cmp results, LDS / global stores and loads (not by MFMA: the matrix pipe ignores it)
        self.scc = 0
        self.m0 = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.lgkm = []   # pending LDS / SMEM results: callables applied in order
        self.vm = []     # pending VMEM operations
        self.n_exec = 0

    # ---- operand access
    def rd(self, x):
        if isinstance(x, Neg):
            return self.rd(x.r) ^ np.uint32(0x80000000)
        if isinstance(x, Reg):
            assert x.n == 1, x
            if x.kind == "v":
                return self.v[x.idx]
            if x.kind == "a":
                return self.a[x.idx]
            return np.full(64, self.s[x.idx], np.uint32)
        if isinstance(x, Lit):
            return np.full(64, x.bits & 0xFFFFFFFF, np.uint32)
        if isinstance(x, Special):
            if x.name == "m0":
                return np.full(64, self.m0, np.uint32)
            raise NotImplementedError(x.name)
        if isinstance(x, float):
            return np.full(64, np.float32(x).view(np.uint32), np.uint32)
        if isinstance(x, int):
            return np.full(64, x & 0xFFFFFFFF, np.uint32)
        raise NotImplementedError(repr(x))

    def rds(self, x):
        """scalar read"""
        if isinstance(x, Reg):
            assert x.kind == "s" and x.n == 1
            return int(self.s[x.idx])
        if isinstance(x, Lit):
            return x.bits & 0xFFFFFFFF
        if isinstance(x, Special) and x.name == "m0":
            return self.m0
        if isinstance(x, int):
            return x & 0xFFFFFFFF
        if isinstance(x, float):
            return int(np.float32(x).view(np.uint32))
        raise NotImplementedError(repr(x))

    def rds64(self, x):
        if isinstance(x, Reg) and x.kind == "s" and x.n == 2:
            return int(self.s[x.idx]) | (int(self.s[x.idx + 1]) << 32)
        if isinstance(x, Special) and x.name == "vcc":
            return self.vcc
        if isinstance(x, int):
            return x & MASK64
        raise NotImplementedError(repr(x))

    def wr(self, x, val):
        val = np.asarray(val).astype(np.uint32, copy=False) if not (isinstance(val, np.ndarray) and val.dtype == np.uint32) else val
        assert isinstance(x, Reg) and x.n == 1
        if x.kind not in ("v", "a"):
            raise AssertionError("vector write to an SGPR")
        f = self.file(x.kind)
        if self.exec == MASK64:
            f[x.idx] = val
        else:   # inactive lanes keep their value (VALU, LDS and VMEM results alike)
            f[x.idx] = np.where(self.exec_lanes(), val, f[x.idx])

    def exec_lanes(self):
        return np.array([(self.exec >> i) & 1 for i in range(64)], bool)

    def wrs(self, x, val):
        val &= 0xFFFFFFFF
        if isinstance(x, Special) and x.name == "m0":
            self.m0 = val
            return
        assert isinstance(x, Reg) and x.kind == "s" and x.n == 1, x
        self.s[x.idx] = val

    def file(self, kind):
        return self.v if kind == "v" else self.a

    def tuple_read(self, r):
        return self.file(r.kind)[r.idx:r.idx + r.n]

    def poison(self, r):
        self.file(r.kind)[r.idx:r.idx + r.n] = POISON

    # ---- MFMA (32x32 output, lane l: column l % 32; register r: row 8 (r / 4) + (r % 4) + 4 (l / 32))
    def mfma(self, ins, kdim):
        D, Aop, Bop, Cop = ins.args
        per = kdim // 2                      # k elements per lane
        nreg = per // 2
        assert Aop.n == nreg and Bop.n == nreg and D.n == 16
        lanes = np.arange(64)
        g = lanes // 32
        am = np.zeros((32, kdim), np.float32)
        bm = np.zeros((kdim, 32), np.float32)
        ar = self.tuple_read(Aop)
        br = self.tuple_read(Bop)
        for j in range(nreg):
            for half in range(2):
                kidx = per * g + 2 * j + half
                av = half_to_f32((ar[j] >> (16 * half)) & 0xFFFF, self.dtype)
                bv = half_to_f32((br[j] >> (16 * half)) & 0xFFFF, self.dtype)
                am[lanes % 32, kidx] = av
                bm[kidx, lanes % 32] = bv
        prod = am.astype(np.float64) @ bm.astype(np.float64)
        if isinstance(Cop, Reg):
            c = f32(self.tuple_read(Cop).copy())
        else:
            assert Cop == 0
            c = np.zeros((16, 64), np.float32)
        out = np.zeros((16, 64), np.float32)
        for r in range(16):
            rows = 8 * (r // 4) + (r % 4) + 4 * g
            out[r] = (prod[rows, lanes % 32] + c[r].astype(np.float64)).astype(np.float32)
        self.file(D.kind)[D.idx:D.idx + 16] = out.view(np.uint32)

    def mfma_scale_fp8(self, ins):
        """v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 e4m3 operands (cbsz = blgp = 0), op_sel 0, as measured on an MI355X (tools/ubench/
        mfma_scale_probe.py): lane (i, g) of an operand holds, in its 32 bytes, k = 16 g + 0..15 (bytes 0-15) and 32 + 16 g + 0..15 (bytes
        16-31) of row i; the scale of (row i, k block 0 = k < 32) is byte 0 of lane i's scale register, that of k block 1 byte 0 of lane
        i + 32's (E8M0: 2^(byte - 127)).  First operand -> rows of D, second -> columns; fp32 accumulate (the hardware's internal rounding
        of the 64-term sums -- ~6e-5 relative in the probe -- is not modelled: float64 products, one fp32 rounding)."""
        D, Aop, Bop, Cop, SA, SB = ins.args
        assert Aop.n == 8 and Bop.n == 8 and D.n == 16
        lanes = np.arange(64)
        g = lanes // 32

        def matrix(op, sc):
            regs = self.tuple_read(op)                                   # [8][64] uint32
            by = np.ascontiguousarray(regs.T).view(np.uint8).reshape(64, 32)  # lane-major bytes
            vals = fp8_e4m3_to_f64(by)
            m = np.zeros((32, 64))
            for lane in range(64):
                kk = [16 * g[lane] + b for b in range(16)] + [32 + 16 * g[lane] + b for b in range(16)]
                m[lane % 32, kk] = vals[lane]
            e8 = (self.rd(sc) & 0xFF).astype(np.float64) - 127.0
            for row in range(32):
                m[row, :32] *= 2.0 ** e8[row]
                m[row, 32:] *= 2.0 ** e8[row + 32]
            return m
        am, bm = matrix(Aop, SA), matrix(Bop, SB)
        prod = am @ bm.T
        if isinstance(Cop, Reg):
            c = f32(self.tuple_read(Cop).copy())
        else:
            assert Cop == 0
            c = np.zeros((16, 64), np.float32)
        out = np.zeros((16, 64), np.float32)
        for r in range(16):
            rows = 8 * (r // 4) + (r % 4) + 4 * g
            out[r] = (prod[rows, lanes % 32] + c[r].astype(np.float64)).astype(np.float32)
        self.file(D.kind)[D.idx:D.idx + 16] = out.view(np.uint32)

    # ---- waits
    def retire(self, queue, keep):
        while len(queue) > keep:
            queue.pop(0)()

    def step(self):
        it = self.items[self.pc]
        self.pc += 1
        if isinstance(it, Label):
            return
        self.n_exec += 1
        op, a = it.op, it.args
        lds = self.wg.lds
        mem = self.wg.mem
        if op.endswith("_e32") or op.endswith("_e64"):
            op = op[:-4]
        # ---------------- scalar
        if op == "s_nop" or op == "s_setprio" or op == "s_sleep":
            return
        if op == "s_endpgm":
            self.retire(self.vm, 0)
            self.done = True
            return
        if op == "s_barrier":
            self.at_barrier = True
            return
        if op == "s_waitcnt":
            txt = a[0]
            cnt = int(txt[txt.index("(") + 1:txt.index(")")])
            if txt.startswith("vmcnt"):
                self.retire(self.vm, cnt)
            elif txt.startswith("lgkmcnt"):
                self.retire(self.lgkm, cnt)
            else:
                raise NotImplementedError(txt)
            return
        if op in ("s_load_dword", "s_load_dwordx2", "s_load_dwordx4", "s_load_dwordx8", "s_load_dwordx16"):
            n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8, "s_load_dwordx16": 16}[op]
            base = self.rds64(a[1]) + self.rds(a[2])
            data = mem.read(base, 4 * n).view(np.uint32).copy()
            dst = a[0]
            self.s[dst.idx:dst.idx + n] = 0xDEADBEEF

            def land(dst=dst, data=data, n=n):
                self.s[dst.idx:dst.idx + n] = data
            self.lgkm.append(land)
            return
        if op == "s_getreg_b32":
            self.wrs(a[0], 0)   # (HW_REG_XCC_ID: the emulated workgroup runs on XCD 0)
            return
        if op in ("s_memtime", "s_memrealtime"):
            # the shader clock / the constant-rate wall clock: the emulator's stand-ins are the wave's instruction count x 4 (an issue slot
            # is ~4 cycles) and that count / 16 -- monotonic, so differences are meaningful to the kernels' own bookkeeping tests
            dst = a[0]
            t = self.n_exec * 4 if op == "s_memtime" else self.n_exec // 4
            data = np.array([t & 0xFFFFFFFF, t >> 32], np.uint32)
            self.s[dst.idx:dst.idx + 2] = 0xDEADBEEF

            def land(dst=dst, data=data):
                self.s[dst.idx:dst.idx + 2] = data
            self.lgkm.append(land)
            return
        if op == "s_mov_b32":
            self.wrs(a[0], self.rds(a[1]))
            return
        if op == "s_mov_b64" and isinstance(a[0], Special) and a[0].name == "exec":
            self.exec = self.rds64(a[1]) & MASK64
            return
        if op == "s_min_u32":
            x, y = self.rds(a[1]), self.rds(a[2])
            self.scc = int(x < y)
            self.wrs(a[0], min(x, y))
            return
        if op == "s_mov_b64":
            v = self.rds64(a[1])
            self.s[a[0].idx] = v & 0xFFFFFFFF
            self.s[a[0].idx + 1] = v >> 32
            return
        if op in ("s_add_u32", "s_addc_u32", "s_sub_u32", "s_subb_u32", "s_add_i32", "s_sub_i32"):
            x, y = self.rds(a[1]), self.rds(a[2])
            if op == "s_add_u32" or op == "s_add_i32":
                r = x + y
                self.scc = int(r > 0xFFFFFFFF)
            elif op == "s_addc_u32":
                r = x + y + self.scc
                self.scc = int(r > 0xFFFFFFFF)
            elif op == "s_subb_u32":
                r = x - y - self.scc
                self.scc = int(y + self.scc > x)
            else:
                r = x - y
                self.scc = int(y > x)
            self.wrs(a[0], r & 0xFFFFFFFF)
            return
        if op == "s_mul_i32":
            self.wrs(a[0], (self.rds(a[1]) * self.rds(a[2])) & 0xFFFFFFFF)
            return
        if op == "s_mul_hi_u32":
            self.wrs(a[0], (self.rds(a[1]) * self.rds(a[2])) >> 32)
            return
        if op in ("s_lshl_b32", "s_lshr_b32", "s_and_b32", "s_or_b32"):
            x, y = self.rds(a[1]), self.rds(a[2])
            r = {"s_lshl_b32": (x << (y & 31)), "s_lshr_b32": x >> (y & 31), "s_and_b32": x & y, "s_or_b32": x | y}[op] & 0xFFFFFFFF
            self.scc = int(r != 0)
            self.wrs(a[0], r)
            return
        if op == "s_cmp_eq_u64":
            self.scc = int(self.rds64(a[0]) == self.rds64(a[1]))
            return
        if op.startswith("s_cmp_"):
            x, y = self.rds(a[0]), self.rds(a[1])
            self.scc = int({"s_cmp_lt_u32": x < y, "s_cmp_eq_u32": x == y, "s_cmp_ge_u32": x >= y, "s_cmp_gt_u32": x > y,
                            "s_cmp_le_u32": x <= y, "s_cmp_lg_u32": x != y}[op])
            return
        if op == "s_cselect_b32":
            self.wrs(a[0], self.rds(a[1]) if self.scc else self.rds(a[2]))
            return
        if op in ("s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccnz", "s_cbranch_vccz"):
            take = {"s_branch": True, "s_cbranch_scc0": self.scc == 0, "s_cbranch_scc1": self.scc == 1,
                    "s_cbranch_vccnz": self.vcc != 0, "s_cbranch_vccz": self.vcc == 0}[op]
            if take:
                self.pc = self.labels[a[0].name]
            return
        # ---------------- vector ALU
        if op.startswith("v_mfma_f32_32x32x16"):
            self.mfma(it, 16)
            return
        if op == "v_mfma_scale_f32_32x32x64_f8f6f4":
            self.mfma_scale_fp8(it)
            return
        if op.startswith("v_mfma_f32_32x32x8"):
            self.mfma(it, 8)
            return
        if op == "v_mov_b32":
            self.wr(a[0], self.rd(a[1]).copy())
            return
        if op == "v_readfirstlane_b32":
            self.wrs(a[0], int(self.rd(a[1])[0]))
            return
        if op in ("v_accvgpr_write_b32", "v_accvgpr_read_b32"):
            self.wr(a[0], self.rd(a[1]).copy())
            return
        if op == "v_mul_hi_u32":
            x, y = self.rd(a[1]).astype(np.uint64), self.rd(a[2]).astype(np.uint64)
            self.wr(a[0], ((x * y) >> np.uint64(32)).astype(np.uint32))
            return
        if op == "v_permlane32_swap_b32":   # the upper 32 lanes of the first operand <-> the lower 32 lanes of the second (EXEC ignored, as the generators use it)
            x, y = self.rd(a[0]).copy(), self.rd(a[1]).copy()
            nx, ny = x.copy(), y.copy()
            nx[32:], ny[:32] = y[:32], x[32:]
            self.file(a[0].kind)[a[0].idx] = nx
            self.file(a[1].kind)[a[1].idx] = ny
            return
        if op in ("v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_mul_lo_u32"):
            x, y = self.rd(a[1]).astype(np.uint64), self.rd(a[2]).astype(np.uint64)
            r = {"v_and_b32": x & y, "v_or_b32": x | y, "v_xor_b32": x ^ y, "v_add_u32": x + y, "v_sub_u32": x - y,
                 "v_mul_lo_u32": x * y}[op]
            self.wr(a[0], (r & 0xFFFFFFFF).astype(np.uint32))
            return
        if op in ("v_lshlrev_b32", "v_lshrrev_b32"):
            sh, x = self.rd(a[1]) & 31, self.rd(a[2])
            r = (x.astype(np.uint64) << sh.astype(np.uint64)) if op == "v_lshlrev_b32" else (x >> sh)
            self.wr(a[0], (r & 0xFFFFFFFF).astype(np.uint32))
            return
        if op == "v_cvt_pk_fp8_f32":   # two fp32 -> two e4m3 bytes (RNE; out of range -> NaN code, as measured) into the low or (op_sel:[0,0,1]) high half of dst
            x, y = f32(self.rd(a[1])).astype(np.float64), f32(self.rd(a[2])).astype(np.float64)

            def enc(v):
                b = f64_to_fp8_e4m3(v).astype(np.uint32)
                return np.where(np.abs(v) > 464.0, np.uint32(0x7F) | (np.signbit(v).astype(np.uint32) << 7), b)   # (464 = the rounding boundary above 448)
            pair = enc(x) | (enc(y) << 8)
            old = self.rd(a[0])
            hi = "op_sel:[0,0,1]" in str(it.mods.get("text", ""))
            self.wr(a[0], ((old & np.uint32(0x0000FFFF)) | (pair << 16)) if hi else ((old & np.uint32(0xFFFF0000)) | pair))
            return
        if op in ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_max_f32", "v_min_f32"):
            x, y = f32(self.rd(a[1])), f32(self.rd(a[2]))
            with np.errstate(all="ignore"):
                r = {"v_add_f32": x + y, "v_sub_f32": x - y, "v_mul_f32": x * y, "v_max_f32": np.fmax(x, y), "v_min_f32": np.fmin(x, y)}[op]
            self.wr(a[0], u32(r))
            return
        if op == "v_fma_f32":
            x, y, z = f32(self.rd(a[1])).astype(np.float64), f32(self.rd(a[2])).astype(np.float64), f32(self.rd(a[3])).astype(np.float64)
            with np.errstate(all="ignore"):
                self.wr(a[0], u32((x * y + z).astype(np.float32)))
            return
        if op == "v_max3_f32":
            x, y, z = f32(self.rd(a[1])), f32(self.rd(a[2])), f32(self.rd(a[3]))
            self.wr(a[0], u32(np.fmax(np.fmax(x, y), z)))
            return
        if op == "v_exp_f32":
            with np.errstate(all="ignore"):
                self.wr(a[0], u32(np.exp2(f32(self.rd(a[1])).astype(np.float64)).astype(np.float32)))
            return
        if op == "v_rcp_f32":
            with np.errstate(all="ignore"):
                self.wr(a[0], u32((1.0 / f32(self.rd(a[1])).astype(np.float64)).astype(np.float32)))
            return
        if op in ("v_cvt_pk_f16_f32", "v_cvt_pk_bf16_f32"):
            dt = "f16" if "f16" in op and "bf16" not in op else "bf16"
            lo = f32_to_half(f32(self.rd(a[1])), dt).astype(np.uint32)
            hi = f32_to_half(f32(self.rd(a[2])), dt).astype(np.uint32)
            self.wr(a[0], lo | (hi << 16))
            return
        if op == "v_cvt_pkrtz_f16_f32":
            def rtz(x):
                x = np.asarray(x, dtype=np.float32)
                with np.errstate(over="ignore"):
                    h = x.astype(np.float16)
                    back = h.astype(np.float32)
                up = (np.abs(back) > np.abs(x)) & np.isfinite(x)
                hu = h.view(np.uint16).copy()
                hu[up] -= 1  # one ulp toward zero (sign-magnitude encoding)
                inf = np.isinf(back) & np.isfinite(x)
                hu[inf] = (hu[inf] & 0x8000) | 0x7BFF
                return hu
            lo = rtz(f32(self.rd(a[1]))).astype(np.uint32)
            hi = rtz(f32(self.rd(a[2]))).astype(np.uint32)
            self.wr(a[0], lo | (hi << 16))
            return
        if op in ("v_pk_add_f16", "v_pk_max_f16"):
            x, y = self.rd(a[1]), self.rd(a[2])
            if "op_sel:[0,1] op_sel_hi:[1,0]" in it.mods.get("text", ""):
                y = ((y >> 16) | (y << 16)) & 0xFFFFFFFF
            else:
                assert "text" not in it.mods
            out = np.zeros(64, np.uint32)
            for h in range(2):
                xa = half_to_f32((x >> (16 * h)) & 0xFFFF, "f16")
                ya = half_to_f32((y >> (16 * h)) & 0xFFFF, "f16")
                with np.errstate(all="ignore"):
                    r = xa + ya if op == "v_pk_add_f16" else np.fmax(xa, ya)
                out |= f32_to_half(r, "f16").astype(np.uint32) << (16 * h)
            self.wr(a[0], out)
            return
        if op == "v_fma_mix_f32":
            txt = it.mods.get("text", "")
            assert "op_sel_hi:[1,0,0]" in txt
            x = self.rd(a[1])
            xh = half_to_f32((x >> 16) & 0xFFFF if "op_sel:[1,0,0]" in txt else x & 0xFFFF, "f16")
            r = xh.astype(np.float64) * f32(self.rd(a[2])).astype(np.float64) + f32(self.rd(a[3])).astype(np.float64)
            self.wr(a[0], u32(r.astype(np.float32)))
            return
        if op == "v_cvt_f16_f32":
            self.wr(a[0], f32_to_half(f32(self.rd(a[1])), "f16").astype(np.uint32))
            return
        if op == "v_cvt_f32_f16":
            self.wr(a[0], u32(half_to_f32(self.rd(a[1]) & 0xFFFF, "f16")))
            return
        if op == "v_pack_b32_f16":
            self.wr(a[0], (self.rd(a[1]) & 0xFFFF) | ((self.rd(a[2]) & 0xFFFF) << 16))
            return
        if op in ("v_dot2c_f32_f16", "v_dot2c_f32_bf16"):
            dt = "bf16" if "bf16" in op else "f16"
            x, y = self.rd(a[1]), self.rd(a[2])
            acc = f32(self.rd(a[0])).astype(np.float64)
            for h in range(2):
                acc = acc + half_to_f32((x >> (16 * h)) & 0xFFFF, dt).astype(np.float64) * half_to_f32((y >> (16 * h)) & 0xFFFF, dt).astype(np.float64)
            with np.errstate(over="ignore"):
                self.wr(a[0], u32(acc.astype(np.float32)))
            return
        if op in ("v_cmp_eq_u32", "v_cmp_le_f32", "v_cmp_ge_f32", "v_cmp_lt_u32", "v_cmp_le_u32", "v_cmp_le_i32"):
            assert isinstance(a[0], Special) and a[0].name == "vcc"
            if op.endswith("i32"):
                x, y = self.rd(a[1]).view(np.int32), self.rd(a[2]).view(np.int32)
            elif op.endswith("u32"):
                x, y = self.rd(a[1]), self.rd(a[2])
            else:
                x, y = f32(self.rd(a[1])), f32(self.rd(a[2]))
            with np.errstate(invalid="ignore"):
                m = {"v_cmp_eq_u32": x == y, "v_cmp_lt_u32": x < y, "v_cmp_le_u32": x <= y, "v_cmp_le_i32": x <= y, "v_cmp_le_f32": x <= y, "v_cmp_ge_f32": x >= y}[op]
            self.vcc = int(sum(1 << i for i in range(64) if m[i])) & self.exec   # inactive lanes write 0
            return
        if op == "v_cndmask_b32":
            mask = self.rds64(a[3])
            sel = np.array([(mask >> i) & 1 for i in range(64)], bool)
            self.wr(a[0], np.where(sel, self.rd(a[2]), self.rd(a[1])).astype(np.uint32))
            return
        # ---------------- LDS
        if op == "ds_read_b128":
            dst, addr = a[0], self.rd(a[1]).astype(np.int64) + int(it.mods.get("offset", 0) or 0)
            assert dst.n == 4
            assert np.all(addr % 16 == 0) and addr.max() + 16 <= len(lds), "ds_read_b128 address"
            data = np.stack([lds[x:x + 16].view(np.uint32) for x in addr], axis=1).copy()  # [4][64]
            self.poison(dst)

            act = self.exec_lanes()

            def land(dst=dst, data=data, act=act):
                f = self.file(dst.kind)
                f[dst.idx:dst.idx + 4] = np.where(act[None, :], data, f[dst.idx:dst.idx + 4])
            self.lgkm.append(land)
            return
        if op == "ds_write_b128":
            addr = self.rd(a[0]).astype(np.int64) + int(it.mods.get("offset", 0) or 0)
            src = a[1]
            assert src.n == 4 and np.all(addr % 16 == 0) and addr.max() + 16 <= len(lds), "ds_write_b128 address"
            data = self.file(src.kind)[src.idx:src.idx + 4].copy()  # [4][64]
            for lane in range(64):
                if (self.exec >> lane) & 1:
                    lds[addr[lane]:addr[lane] + 16] = data[:, lane].copy().view(np.uint8)
            self.lgkm.append(lambda: None)
            return
        if op == "ds_write_b32":
            addr = self.rd(a[0]).astype(np.int64) + int(it.mods.get("offset", 0) or 0)
            data = self.rd(a[1]).copy()
            for lane in range(64):
                if (self.exec >> lane) & 1:
                    assert addr[lane] % 4 == 0 and addr[lane] + 4 <= len(lds), "ds_write_b32 address"
                    lds[addr[lane]:addr[lane] + 4] = np.array([data[lane]], np.uint32).view(np.uint8)
            self.lgkm.append(lambda: None)
            return
        if op == "ds_read_b32":
            dst, addr = a[0], self.rd(a[1]).astype(np.int64) + int(it.mods.get("offset", 0) or 0)
            assert np.all(addr % 4 == 0) and addr.max() + 4 <= len(lds), "ds_read_b32 address"
            data = np.array([lds[x:x + 4].view(np.uint32)[0] for x in addr], np.uint32)
            self.poison(dst)
            act = self.exec_lanes()

            def land(dst=dst, data=data, act=act):
                f = self.file(dst.kind)
                f[dst.idx] = np.where(act, data, f[dst.idx])
            self.lgkm.append(land)
            return
        if op == "ds_bpermute_b32":
            idx = (self.rd(a[1]) >> 2) & 63
            data = self.rd(a[2])[idx].copy()
            dst = a[0]
            self.poison(dst)

            def land(dst=dst, data=data):
                self.wr(dst, data)
            self.lgkm.append(land)
            return
        # ---------------- global memory
        if op in ("global_load_dwordx4", "global_load_dword"):
            nd = 4 if op.endswith("x4") else 1
            dst, voff, sbase = a
            base = self.rds64(sbase) + int(it.mods.get("offset", 0) or 0)
            off = self.rd(voff).astype(np.int64)
            data = np.stack([mem.read(base + o, 4 * nd).view(np.uint32) for o in off], axis=1).copy()
            self.poison(dst)

            def land(dst=dst, data=data, nd=nd):
                self.file(dst.kind)[dst.idx:dst.idx + nd] = data
            self.vm.append(land)
            return
        if op == "global_load_lds_dwordx4":
            voff, sbase = a
            base = self.rds64(sbase) + int(it.mods.get("offset", 0) or 0)
            off = self.rd(voff).astype(np.int64)
            data = [mem.read(base + o, 16).copy() for o in off]
            m0 = self.m0

            def land(data=data, m0=m0):
                for lane in range(64):
                    lds[m0 + 16 * lane:m0 + 16 * lane + 16] = data[lane]
            self.vm.append(land)
            return
        if op in ("global_store_dwordx4", "global_store_dword"):
            voff, src, sbase = a
            base = self.rds64(sbase) + int(it.mods.get("offset", 0) or 0)
            off = self.rd(voff).astype(np.int64)
            vals = self.tuple_read(src).copy()
            for lane in range(64):
                if (self.exec >> lane) & 1:
                    mem.write(base + off[lane], vals[:, lane].copy().view(np.uint8))
            self.vm.append(lambda: None)
            return
        if op == "global_atomic_add" and len(a) == 4:   # returning form (sc0): dst = the value before the add
            dst, voff, src, sbase = a
            base = self.rds64(sbase) + int(it.mods.get("offset", 0) or 0)
            off = self.rd(voff).astype(np.int64)
            vals = self.rd(src)
            old = self.rd(dst).copy()
            for lane in range(64):
                if (self.exec >> lane) & 1:
                    cur = mem.read(base + off[lane], 4).view(np.uint32)[0]
                    old[lane] = cur
                    mem.write(base + off[lane], np.array([(int(cur) + int(vals[lane])) & 0xFFFFFFFF], np.uint32).view(np.uint8))
            self.poison(dst)

            def land(dst=dst, old=old):
                self.wr(dst, old)
            self.vm.append(land)
            return
        if op == "global_atomic_add":
            voff, src, sbase = a
            base = self.rds64(sbase) + int(it.mods.get("offset", 0) or 0)
            off = self.rd(voff).astype(np.int64)
            vals = self.rd(src)
            for lane in range(64):
                if (self.exec >> lane) & 1:
                    cur = mem.read(base + off[lane], 4).view(np.uint32)[0]
                    mem.write(base + off[lane], np.array([(int(cur) + int(vals[lane])) & 0xFFFFFFFF], np.uint32).view(np.uint8))
            self.vm.append(lambda: None)
            return
        if op == "global_atomic_add_x2":
            voff, src, sbase = a
            base = self.rds64(sbase) + int(it.mods.get("offset", 0) or 0)
            off = self.rd(voff).astype(np.int64)
            vals = self.tuple_read(src)
            for lane in range(64):
                if (self.exec >> lane) & 1:
                    cur = int(mem.read(base + off[lane], 8).view(np.uint64)[0])
                    add = int(vals[0, lane]) | (int(vals[1, lane]) << 32)
                    mem.write(base + off[lane], np.array([(cur + add) & 0xFFFFFFFFFFFFFFFF], np.uint64).view(np.uint8))
            self.vm.append(lambda: None)
            return
        if op == "global_store_dwordx2":
            voff, src, sbase = a
            base = self.rds64(sbase) + int(it.mods.get("offset", 0) or 0)
            off = self.rd(voff).astype(np.int64)
            vals = self.tuple_read(src).copy()
            for lane in range(64):
                if (self.exec >> lane) & 1:
                    mem.write(base + off[lane], vals[:, lane].copy().view(np.uint8))
            self.vm.append(lambda: None)
            return
        raise NotImplementedError(it.text())


_FP8_TABLE = None


def fp8_e4m3_to_f64(b):
    """OCP e4m3 (fn: no infinities, 0x7F / 0xFF = NaN) bytes -> float64"""
    global _FP8_TABLE
    if _FP8_TABLE is None:
        t = np.zeros(256)
        for v in range(256):
            s, e, m = v >> 7, (v >> 3) & 15, v & 7
            if e == 15 and m == 7:
                x = np.nan
            elif e == 0:
                x = m / 8.0 * 2.0 ** -6
            else:
                x = (1 + m / 8.0) * 2.0 ** (e - 7)
            t[v] = -x if s else x
        _FP8_TABLE = t
    return _FP8_TABLE[np.asarray(b, np.uint8)]


def f64_to_fp8_e4m3(x):
    """round to nearest even onto the e4m3 grid after clamping to +-448 (what the producers of the fp8 planes do) -> bytes"""
    fp8_e4m3_to_f64(np.zeros(1, np.uint8))
    x = np.clip(np.asarray(x, np.float64), -448.0, 448.0)
    pos = _FP8_TABLE[:127]   # 0 .. 448 ascending (0x7F is NaN)
    a = np.abs(x)
    idx = np.clip(np.searchsorted(pos, a), 1, 126)
    lo, hi = pos[idx - 1], pos[idx]
    pick_hi = (a - lo > hi - a) | ((a - lo == hi - a) & (idx % 2 == 0))   # tie -> even code
    code = np.where(pick_hi, idx, idx - 1).astype(np.uint8)
    return (code | np.where(np.signbit(x), 0x80, 0).astype(np.uint8)).astype(np.uint8)


class Workgroup:
    def __init__(self, program, mem, kernarg_addr, wg_id, n_waves, lds_bytes, dtype):
        self.mem = mem
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.lds[:] = 0xA5  # garbage
        items = program.items
        labels = {it.name: i for i, it in enumerate(items) if isinstance(it, Label)}
        self.waves = []
        for w in range(n_waves):
            wv = Wave(self, w, items, labels, dtype)
            wv.s[0] = kernarg_addr & 0xFFFFFFFF
            wv.s[1] = kernarg_addr >> 32
            wv.s[2], wv.s[3], wv.s[4] = wg_id
            wv.v[0] = np.arange(64, dtype=np.uint32) + 64 * w
            self.waves.append(wv)

    def run(self, max_steps=10_000_000):
        steps = 0
        flip = False
        while True:
            order = self.waves[::-1] if flip else self.waves
            flip = not flip
            progressed = False
            for wv in order:
                while not wv.done and not wv.at_barrier:
                    wv.step()
                    steps += 1
                    progressed = True
                    if steps > max_steps:
                        raise RuntimeError("emulator step limit")
            live = [w for w in self.waves if not w.done]
            if not live:
                return steps
            if all(w.at_barrier for w in live):
                for w in live:
                    w.at_barrier = False
            elif not progressed:
                raise RuntimeError("deadlock")
