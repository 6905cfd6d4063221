"""Tiny gfx950 assembly IR used by the hand-scheduled kernels under fast3r_amd/csrc/asm/.

A kernel is a list of `Ins` (mnemonic + typed operands + modifier text) and labels.  The same list is
  * printed as assembler text for clang (`Program.text()`), and
  * executed lane-exactly by tools/gfx950_emu.py on the CPU (so bookkeeping bugs -- register clashes, pipeline indexing,
    address arithmetic, missing s_waitcnt -- are found without a GPU), and
  * walked by `Program.check_hazards()`: inline assembly gets no compiler-inserted wait states, so the distance rules the
    hardware does not interlock (MFMA result -> VALU/LDS/VMEM reader, VALU result -> MFMA operand, M0 write -> LDS-DMA)
    are asserted on the straight-line instruction stream.
"""
from dataclasses import dataclass, field


@dataclass(frozen=True)
class Reg:
    kind: str  # 'v' | 'a' | 's'
    idx: int
    n: int = 1

    def __str__(self):
        if self.n == 1:
            return f"{self.kind}{self.idx}"
        return f"{self.kind}[{self.idx}:{self.idx + self.n - 1}]"

    def regs(self):
        return [(self.kind, self.idx + i) for i in range(self.n)]

    def sub(self, i, n=1):
        assert 0 <= i and i + n <= self.n
        return Reg(self.kind, self.idx + i, n)


def V(i, n=1):
    return Reg("v", i, n)


def A(i, n=1):
    return Reg("a", i, n)


def S(i, n=1):
    return Reg("s", i, n)


@dataclass(frozen=True)
class Special:
    name: str  # 'vcc' | 'm0' | 'exec' | 'off' | 'scc'

    def __str__(self):
        return self.name


VCC = Special("vcc")
M0 = Special("m0")
EXEC = Special("exec")
OFF = Special("off")


@dataclass(frozen=True)
class Lit:
    """32-bit literal given as raw bits (printed in hex)."""
    bits: int

    def __str__(self):
        return f"0x{self.bits & 0xffffffff:x}"


@dataclass(frozen=True)
class Neg:
    """source operand with the VOP3 neg modifier"""
    r: Reg

    def __str__(self):
        return f"-{self.r}"


@dataclass(frozen=True)
class LabelRef:
    name: str

    def __str__(self):
        return self.name


@dataclass
class Ins:
    op: str
    args: tuple = ()
    mods: dict = field(default_factory=dict)  # offset=..., plus free text under 'text'
    comment: str = ""

    def text(self):
        parts = []
        for a in self.args:
            if isinstance(a, float):
                parts.append(repr(a))
            else:
                parts.append(str(a))
        s = self.op
        if parts:
            s += " " + ", ".join(parts)
        if "offset" in self.mods and self.mods["offset"]:
            s += f" offset:{self.mods['offset']}"
        if "text" in self.mods:
            s += " " + self.mods["text"]
        if self.comment:
            s = f"{s:<72}; {self.comment}"
        return s


@dataclass
class Label:
    name: str


def base_op(op):
    return op[:-4] if op.endswith("_e32") or op.endswith("_e64") else op


def is_mfma(op):
    return op.startswith("v_mfma")


def is_valu(op):
    return op.startswith("v_") and not is_mfma(op)


class Program:
    def __init__(self, name):
        self.name = name
        self.items = []  # Ins | Label

    def emit(self, op, *args, comment="", **mods):
        ins = Ins(op, tuple(args), dict(mods), comment)
        self.items.append(ins)
        return ins

    def label(self, name):
        self.items.append(Label(name))

    def body_text(self):
        out = []
        for it in self.items:
            if isinstance(it, Label):
                out.append(f"{it.name}:")
            else:
                out.append("\t" + it.text())
        return "\n".join(out) + "\n"

    # ---- static hazard check on the straight-line stream (labels reset nothing: distances only grow across a taken branch
    # because every branch target in these kernels is reached through at least the branch instruction itself)
    MFMA_RESULT_WAIT = 20   # issue slots between an MFMA and any non-MFMA access of its destination (8-pass XDL needs 11..18)
    VALU_TO_MFMA_WAIT = 3   # issue slots between a VALU write and an MFMA reading it as A/B/C
    def check_hazards(self):
        """straight-line check of the whole stream (an unconditional branch ends a region) plus, for every branch, the window
        [40 instructions before the branch] + [40 instructions after its target]"""
        ins = [it for it in self.items]
        problems = self._check_stream(ins, "")
        labels = {it.name: i for i, it in enumerate(ins) if isinstance(it, Label)}
        for i, it in enumerate(ins):
            if isinstance(it, Ins) and (it.op.startswith("s_cbranch") or it.op == "s_branch"):
                tgt = labels[it.args[0].name]
                lo = max(0, i - 40)
                for j in range(i - 1, lo - 1, -1):   # code above an unconditional branch does not fall through to here
                    if isinstance(ins[j], Ins) and ins[j].op in ("s_branch", "s_endpgm"):
                        lo = j + 1
                        break
                window = ins[lo:i + 1] + ins[tgt:tgt + 40]
                problems += self._check_stream(window, f"[edge {it.text().strip()}] ", branch_resets=False)
        return problems

    TRANS = ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32")

    def _check_stream(self, items, tag, branch_resets=True):
        last_mfma_write = {}   # (kind, idx) -> instruction index
        last_valu_write = {}
        last_trans_write = {}  # VGPRs written by a transcendental: a non-transcendental VALU reader needs one wait state (gfx940 forwarding hazard)
        last_valu_sgpr_write = {}  # SGPRs / VCC written by a VALU: a VALU reader needs two wait states
        last_m0_write = -10
        last_dot_write = {}
        exec_full = True       # False after `s_mov_b64 exec, <a mask>` until `s_mov_b64 exec, -1`
        n = 0
        problems = []
        for it in items:
            if isinstance(it, Label):
                exec_full = True   # (a join: every branch in these kernels is taken with all lanes active)
                continue
            op = it.op
            if op in ("s_branch", "s_endpgm"):
                exec_full = True
            # a lane mask computed under a narrowed EXEC is narrower than its compare says (v_cmp writes 0 for inactive lanes): the generators
            # always mean "all lanes" when they compare, so a v_cmp while EXEC is narrowed is a bug (round 4: the state-out epilogue)
            if op == "s_mov_b64" and it.args and isinstance(it.args[0], Special) and it.args[0].name == "exec":
                exec_full = isinstance(it.args[1], int) and it.args[1] == -1
            elif op.startswith("v_cmp") and not exec_full:
                problems.append(f"{tag}{n}: {it.text()} computes a lane mask while EXEC is narrowed (inactive lanes read as 0)")
            if op in ("s_branch", "s_endpgm") and branch_resets:
                last_mfma_write, last_valu_write, last_m0_write = {}, {}, -10
                last_trans_write, last_valu_sgpr_write, last_dot_write = {}, {}, {}
                continue
            slots = 1
            if op == "s_nop":
                slots = int(it.args[0]) + 1
            regs_r, regs_w = operand_rw(it)
            if is_mfma(op):
                for r in regs_r:
                    if r in last_valu_write and n - last_valu_write[r] < self.VALU_TO_MFMA_WAIT:
                        problems.append(f"{tag}{n}: {it.text()} reads {r} written by VALU {n - last_valu_write[r]} slots earlier")
                ab = set()
                for a in it.args[1:3]:
                    if isinstance(a, Reg):
                        ab.update(a.regs())
                for r in ab:
                    if r in last_mfma_write and n - last_mfma_write[r] < self.MFMA_RESULT_WAIT:
                        problems.append(f"{tag}{n}: {it.text()} reads MFMA result {r} as A/B after {n - last_mfma_write[r]} slots")
                for r in set(regs_w):
                    last_mfma_write[r] = n
            else:
                for r in list(regs_r) + list(regs_w):
                    if r in last_mfma_write and n - last_mfma_write[r] < self.MFMA_RESULT_WAIT:
                        problems.append(f"{tag}{n}: {it.text()} touches MFMA result {r} after {n - last_mfma_write[r]} slots")
                if is_valu(op):
                    for r in list(regs_r) + list(regs_w):   # gfx940: a dot result may be consumed at once only by the same dot opcode as its accumulator
                        if r in last_dot_write and n - last_dot_write[r][0] < 4 and not (base_op(op) == last_dot_write[r][1] and r in regs_w):
                            problems.append(f"{tag}{n}: {it.text()} touches dot result {r} after {n - last_dot_write[r][0]} slots")
                    if base_op(op).startswith("v_dot"):
                        for r in regs_w:
                            last_dot_write[r] = (n, base_op(op))
                    if base_op(op) not in self.TRANS:
                        for r in regs_r:
                            if r in last_trans_write and n - last_trans_write[r] < 2:
                                problems.append(f"{tag}{n}: {it.text()} reads transcendental result {r} with no wait state")
                    if op.startswith("v_readfirstlane") or op.startswith("v_readlane") or op.startswith("v_permlane"):
                        for r in regs_r:
                            if r in last_valu_write and n - last_valu_write[r] < 2:
                                problems.append(f"{tag}{n}: {it.text()} reads {r} written by the previous VALU instruction")
                    sread = [a for a in it.args[1:] if (isinstance(a, Special) and a.name == "vcc")]
                    sregs = [("vcc", 0)] if sread else []
                    sregs += [r for r in regs_r if r[0] == "s"]
                    for r in sregs:
                        if r in last_valu_sgpr_write and n - last_valu_sgpr_write[r] < 3:
                            problems.append(f"{tag}{n}: {it.text()} reads {r} written by a VALU {n - last_valu_sgpr_write[r]} slots earlier")
                    if base_op(op) in self.TRANS:
                        for r in regs_w:
                            last_trans_write[r] = n
                    else:
                        for r in regs_w:
                            last_trans_write.pop(r, None)
                    if op.startswith("v_cmp") and it.args and isinstance(it.args[0], Special):
                        last_valu_sgpr_write[("vcc", 0)] = n
                    if op.startswith("v_readfirstlane") or op.startswith("v_readlane"):
                        for r in regs_w:
                            last_valu_sgpr_write[r] = n
                if is_valu(op) or op.startswith("ds_read") or op.startswith("global_load_dword"):
                    for r in regs_w:
                        last_valu_write[r] = n
                if "lds_dword" in op and n - last_m0_write < 2:
                    problems.append(f"{tag}{n}: {it.text()} follows an M0 write after {n - last_m0_write} slots")
                if any(isinstance(a, Special) and a.name == "m0" for a in it.args[:1]):
                    last_m0_write = n
            n += slots
        return problems


def operand_rw(it):
    """(registers read, registers written) of an instruction, as (kind, idx) pairs.  Conservative for the ops the generators use."""
    op = it.op
    args = it.args

    def regs_of(a):
        if isinstance(a, Neg):
            a = a.r
        if isinstance(a, Reg):
            return a.regs()
        return []

    if not args:
        return [], []
    no_dst = (op.startswith("s_cmp") or op.startswith("s_cbranch") or op in ("s_branch", "s_waitcnt", "s_nop", "s_barrier",
              "s_endpgm", "s_setprio") or op.startswith("global_store") or op.startswith("global_atomic") or op.startswith("global_load_lds") or
              op.startswith("v_cmp"))
    if no_dst:
        r = []
        for a in args:
            r += regs_of(a)
        return r, []
    w = regs_of(args[0])
    r = []
    for a in args[1:]:
        r += regs_of(a)
    if op == "v_fma_mix_f32":
        pass  # the accumulator is an explicit source
    if op.startswith("v_dot2c") or op in ("v_fmac_f32",):
        r += w
    if op == "v_permlane32_swap_b32":
        r += w
        w = w + regs_of(args[1])
    return r, w
