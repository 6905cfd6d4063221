"""Tensor-level wrappers over the C ABI (include/f3r.h).  torch tensors are device memory + shape bookkeeping;
every arithmetic op below is a hand-written HIP kernel in fast3r_amd/csrc/.  No function here has a torch
fallback: CPU tensors raise F3RError.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import (AttnArgs, F3R_A_CONV3X3, F3R_A_PLAIN, F3R_ACT_GELU, F3R_ACT_NONE, F3R_ACT_RELU, F3R_EPI_CONVT,
                   F3R_EPI_GENERIC, F3R_EPI_QKV, F3R_MAX_SEG, F3R_SPLIT_NONE, F3R_SPLIT_W2, F3R_SPLIT_W2F8, F3R_SPLIT_X3, F3R_SPLIT_X3F8, GemmArgs, check, dtype_id,
                   ptr, require_gpu, stream_ptr)

ACT = {None: F3R_ACT_NONE, "none": F3R_ACT_NONE, "gelu": F3R_ACT_GELU, "relu": F3R_ACT_RELU}

# Optional per-launch timing of the attention kernel (bench.py's live roofline): when set to a list, every
# f3r_attn_fwd launch is bracketed by events on the launch stream and (start, end, flops) is appended.
ATTN_TIMER = None
# Optional device int32[56] that the hand-scheduled attention kernel adds its counters to (f3r_attn_args.dbg_counters: entries into the
# re-base block, waves, tiles walked, -, and two 64-bit sums: shader-clock cycles and constant-clock ticks the waves lived); bench.py sets
# it around the timed steps (attn_rebase, roofline.live).
ATTN_COUNTERS = None
# Work stealing of the hand-scheduled attention kernels (f3r_attn_args.sched_counter): True = hand every launch a zeroed {next, done} pair
# (one per device and stream, kept here: the kernel leaves it zero); the library uses it for launches of at least two rounds of workgroups.
ATTN_WORK_STEALING = True
_SCHED = {}
_SCHED_CAPTURE = None   # sched_scope(): the counter of the graph being captured


class sched_scope:
    """`with ops.sched_scope(counter):` -- every attention launch inside uses `counter` (int32[2] on the device) as its work-stealing pair.
    Fast3R's graph cache allocates one per captured graph OUTSIDE the capture and keeps it with the graph: graphs replayed concurrently on
    different streams then never share a pair, and the words do not live in a capture-private pool keyed by torch's shared
    capture stream."""

    def __init__(self, counter):
        assert counter.dtype == torch.int32 and counter.numel() >= 2 and counter.is_cuda
        self.counter = counter

    def __enter__(self):
        global _SCHED_CAPTURE
        self.prev, _SCHED_CAPTURE = _SCHED_CAPTURE, self.counter
        return self.counter

    def __exit__(self, *exc):
        global _SCHED_CAPTURE
        _SCHED_CAPTURE = self.prev
        return False


def _sched_counter(dev):
    """The {next, done} pair of this launch: the scope's (graph capture), else one per (device, stream).  A capture nobody opened a scope for gets
    a FRESH pair per launch (8 bytes from that graph's own pool, alive as long as the graph): never the eager pair of the capture stream.  The
    library clears the pair on the launch stream before the kernel starts, so none of them needs to be zero here."""
    if _SCHED_CAPTURE is not None and _SCHED_CAPTURE.device == dev:
        return _SCHED_CAPTURE
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(2, dtype=torch.int32, device=dev)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    t = _SCHED.get(key)
    if t is None:
        t = _SCHED[key] = torch.zeros(2, dtype=torch.int32, device=dev)
    return t


# Optional per-family timing of EVERY launch (bench.py's roofline.others): when OP_TIMER is a dict, each wrapper below brackets its C call with
# events on the launch stream and appends (start, end, algorithmic FLOP, algorithmic HBM bytes) to OP_TIMER[family]; the family is OP_FAMILY (set by
# the model around its phases: "transformer_linears", "head_convs", ...) or, for the bandwidth-bound wrappers, "elementwise".  None = no events.
OP_TIMER = None
OP_FAMILY = "other"


class _timed:
    __slots__ = ("fam", "flops", "nbytes", "e0")

    def __init__(self, flops=0.0, nbytes=0.0, family=None):
        self.fam = (family or OP_FAMILY) if OP_TIMER is not None else None
        self.flops, self.nbytes = flops, nbytes

    def __enter__(self):
        if self.fam is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.fam is not None and exc[0] is None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            OP_TIMER.setdefault(self.fam, []).append((self.e0, e1, float(self.flops), float(self.nbytes)))
        return False


class family:
    """`with ops.family("head_convs"):` -- the GEMM-like launches inside are booked under that family while OP_TIMER is on"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global OP_FAMILY
        self.prev, OP_FAMILY = OP_FAMILY, self.name

    def __exit__(self, *exc):
        global OP_FAMILY
        OP_FAMILY = self.prev
        return False


def _nb(*ts):
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------------------- weight packing (host, once)
SPLIT = {None: F3R_SPLIT_NONE, 0: F3R_SPLIT_NONE, "none": F3R_SPLIT_NONE, "w2": F3R_SPLIT_W2, "x3": F3R_SPLIT_X3, "w2f8": F3R_SPLIT_W2F8, "x3f8": F3R_SPLIT_X3F8}


def split_planes(w: torch.Tensor, lp: torch.dtype):
    """fp32 -> (hi, lo) lowp with hi = lowp(w), lo = lowp(w - hi): hi + lo carries ~2x the significand bits of one lowp number."""
    hi = w.to(lp)
    return hi, (w - hi.float()).to(lp)


def pack_linear_weight(w: torch.Tensor, lp: torch.dtype, split: bool = False) -> torch.Tensor:
    """nn.Linear / 1x1-conv weight (N, K[,1,1]) fp32 -> lowp [N][Kpad], Kpad = roundup(K, 64), zero padded.
    split: two planes per row, [N][2*Kpad] = [hi | lo] (f3r_gemm_args.split, include/f3r.h)."""
    w = w.reshape(w.shape[0], -1)
    n, k = w.shape
    kp = round_up(k, 64)
    out = torch.zeros((n, kp * (2 if split else 1)), dtype=lp, device=w.device)
    if split:
        out[:, :k], out[:, kp:kp + k] = split_planes(w, lp)
    else:
        out[:, :k] = w.to(lp)
    return out


def pack_linear_weight_f8(w: torch.Tensor):
    """nn.Linear weight (N, K) fp32 -> the operand of f3r_gemm split "w2f8" (include/f3r.h F3R_SPLIT_W2F8): rows [K fp16 hi | K fp8 e4m3((W - hi)
    2^s_n)] as a float16-typed [N][3 K / 2] tensor, and the [N] int32 scale words (E8M0 byte 127 - s_n in all four bytes): one power-of-two
    scale per output channel brings the largest |W - hi| of the row to [112, 224] (e4m3 tops out at 448; what falls below its 2^-9 subnormals
    is 2^-17 of that and contributes nothing).  K must be a multiple of 128."""
    w = w.reshape(w.shape[0], -1).float()
    n, k = w.shape
    assert k % 128 == 0, "w2f8: K must be a multiple of 128"
    hi = w.to(torch.float16)
    lo = w - hi.float()
    amax = lo.abs().amax(dim=1).clamp_min(2.0 ** -100)
    s = torch.floor(torch.log2(224.0 / amax)).clamp(-100, 120)
    lo8 = (lo * torch.exp2(s)[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    out = torch.empty((n, 3 * k), dtype=torch.uint8, device=w.device)
    out[:, :2 * k] = hi.contiguous().view(torch.uint8).view(n, 2 * k)
    out[:, 2 * k:] = lo8.view(torch.uint8)
    e8 = (127 - s).to(torch.int64)
    words = (e8 | (e8 << 8) | (e8 << 16) | (e8 << 24)).to(torch.int32)   # (e8 < 256: fits)
    return out.view(torch.float16).view(n, 3 * k // 2), words.contiguous()


def pack_conv3x3_weight(w: torch.Tensor, lp: torch.dtype, split: bool = False) -> torch.Tensor:
    """Conv2d weight (Cout, Cin, 3, 3) -> lowp [Cout][9 * Cpad], k = (ky*3 + kx) * Cpad + ci, Cpad = roundup(Cin, 64);
    split: [Cout][2][9 * Cpad] = hi plane then lo plane."""
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    cpad = round_up(ci, 64)
    taps = w.permute(0, 2, 3, 1).reshape(co, 9, ci)
    planes = split_planes(taps, lp) if split else (taps.to(lp),)
    out = torch.zeros((co, len(planes), 9, cpad), dtype=lp, device=w.device)
    for i, pl in enumerate(planes):
        out[:, i, :, :ci] = pl
    return out.reshape(co, len(planes) * 9 * cpad)


def _e4m3_rows(x: torch.Tensor):
    """[N][K] fp32 -> (e4m3 bytes of x 2^s_n, E8M0 bytes 127 - s_n): one power-of-two scale per row brings its largest magnitude to [112, 224]."""
    amax = x.abs().amax(dim=1).clamp_min(2.0 ** -100)
    s = torch.floor(torch.log2(224.0 / amax)).clamp(-100, 120)
    q = (x * torch.exp2(s)[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), (127 - s).to(torch.int64)


def pack_conv3x3_weight_f8(w: torch.Tensor):
    """Conv2d weight (Cout, Cin, 3, 3) fp32, Cin % 128 == 0 -> the operand of f3r_gemm split "x3f8" (include/f3r.h F3R_SPLIT_X3F8): rows
    [9 Cin fp16 hi | 9 Cin bytes e4m3((W - hi) 2^s_n) | 9 Cin bytes e4m3(hi 2^t_n)], k = (ky*3 + kx) * Cin + ci in every plane, as a float16-typed
    [Cout][2 * 9 Cin] tensor (the row stride of the two-fp16-plane pack), and the [Cout] int32 scale words (byte 0: 127 - s_n, byte 1: 127 - t_n)."""
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3 and ci % 128 == 0, "x3f8: 3x3 kernels, Cin a multiple of 128"
    taps = w.float().permute(0, 2, 3, 1).reshape(co, 9 * ci)
    hi = taps.to(torch.float16)
    lo8, e_lo = _e4m3_rows(taps - hi.float())
    hi8, e_hi = _e4m3_rows(hi.float())
    kp = 9 * ci
    out = torch.empty((co, 4 * kp), dtype=torch.uint8, device=w.device)
    out[:, :2 * kp] = hi.contiguous().view(torch.uint8).view(co, 2 * kp)
    out[:, 2 * kp:3 * kp] = lo8
    out[:, 3 * kp:] = hi8
    words = (e_lo | (e_hi << 8)).to(torch.int32)
    return out.view(torch.float16).view(co, 2 * kp), words.contiguous()


def f8_planes(x: torch.Tensor) -> torch.Tensor:
    """fp32 (..., C) -> the fp8 planes a "x3f8" convolution reads beside the fp16 high plane: uint8 (..., 2 C) = [e4m3(clamp(x, 448)) | e4m3(clamp((x -
    fp16(x)) 2^12, 448))] -- what f3r_gemm_args.out_f8 / f3r_interp_bilinear_f8 write on the device (tests and tools build inputs with this)."""
    hi = x.to(torch.float16).float()
    a = x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    b = ((x - hi) * 4096.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    return torch.cat([a, b], dim=-1).contiguous()


def pack_convT_weight(w: torch.Tensor, b: torch.Tensor, lp: torch.dtype, split: bool = False):
    """ConvTranspose2d (kernel == stride == s) weight (Cin, Cout, s, s) -> lowp [(dy*s+dx)*Cout + co][Kpad] and the bias
    tiled to [s*s*Cout] (every output pixel gets exactly one tap)."""
    ci, co, s, s2 = w.shape
    assert s == s2
    wp = w.permute(2, 3, 1, 0).reshape(s * s * co, ci)
    return pack_linear_weight(wp, lp, split), b.float().repeat(s * s).contiguous()


def rope_tables(n_pos: int, base: float, device, half_dim: int = 32):
    """cos/sin(pos * base^(-2i/half_dim)), i < half_dim/2, exactly as RoPE2D.get_cos_sin builds them in fp32
    (pos_embed.py:139-150) -> two [n_pos][16] fp32 tables."""
    inv_freq = 1.0 / (base ** (torch.arange(0, half_dim, 2).float() / half_dim))
    t = torch.arange(n_pos, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return freqs.cos().contiguous().to(device), freqs.sin().contiguous().to(device)


# ----------------------------------------------------------------------------------------- kernels
def patchify(img: torch.Tensor, ps: int, lp: torch.dtype, ld_out: int = 0) -> torch.Tensor:
    """(B,3,H,W) fp32 -> [B*h*w][ld_out] lowp rows for the patch-embed GEMM; ld_out = 0: 3*ps*ps, else a zero-padded row stride."""
    require_gpu(img, "img")
    assert img.dtype == torch.float32 and img.is_contiguous()
    B, C, H, W = img.shape
    assert C == 3
    ld = ld_out if ld_out else 3 * ps * ps
    out = torch.empty((B * (H // ps) * (W // ps), ld), dtype=lp, device=img.device)
    with _timed(0.0, _nb(img, out), "elementwise"):
        check(_lib.lib().f3r_patchify(ptr(img), ptr(out), B, H, W, ps, ld_out, dtype_id(lp), stream_ptr()), "f3r_patchify")
    return out


def layernorm(x, gamma, beta, eps, lp, out_lp=None, out_f32=None, want_lp=True, want_f32=False, rms=False):
    require_gpu(x, "x")
    assert x.dtype == torch.float32 and x.is_contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    if want_lp and out_lp is None:
        out_lp = torch.empty(x.shape, dtype=lp, device=x.device)
    if want_f32 and out_f32 is None:
        out_f32 = torch.empty_like(x)
    with _timed(0.0, _nb(x, out_lp, out_f32), "elementwise"):
        check(_lib.lib().f3r_layernorm(ptr(x), ptr(gamma), ptr(beta), ptr(out_lp), ptr(out_f32), rows, D, float(eps),
                                       int(rms), dtype_id(lp), stream_ptr()), "f3r_layernorm")
    return out_lp, out_f32


def layernorm_f8(x, gamma, beta, eps, out_rows=None, rms=False):
    """LayerNorm / RMSNorm -> rows [D fp16 | D fp8] (float16-typed [rows][3 D / 2]): the A operand of gemm(split="w2f8"); out_rows[:, :D] is the
    plain fp16 output (any GEMM reads it with its row stride)."""
    require_gpu(x, "x")
    assert x.dtype == torch.float32 and x.is_contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    if out_rows is None:
        out_rows = torch.empty((rows, 3 * D // 2), dtype=torch.float16, device=x.device)
    assert out_rows.dtype == torch.float16 and out_rows.stride(1) == 1 and out_rows.shape[1] == 3 * D // 2
    with _timed(0.0, _nb(x, out_rows), "elementwise"):
        check(_lib.lib().f3r_layernorm_f8(ptr(x), ptr(gamma), ptr(beta), ptr(out_rows), out_rows.stride(0), rows, D, float(eps), int(rms), stream_ptr()),
              "f3r_layernorm_f8")
    return out_rows


def _split_operand(g, a, split, a_lo):
    """Common split-precision plumbing: g.split and, for "x3", the low plane of A (same shape / strides as A)."""
    g.split = SPLIT[split]
    if g.split == F3R_SPLIT_X3:
        assert a_lo is not None and a_lo.dtype == a.dtype and a_lo.shape == a.shape and a_lo.stride() == a.stride(), "x3 split needs a_lo like a"
        g.A_lo = ptr(a_lo)


def gemm(a, w, *, K=None, bias=None, act=None, rowadd=None, rowadd_div=1, res_f32=None, res_lp=None, res_lp2=None,
         out_f32=None, out_lp=None, want_f32=False, want_lp=False, split=None, a_lo=None, res_lp_lo=None, res_lp2_lo=None,
         out_lp_lo=None, want_lo=False, kernel_sel=0, w_scale=None, out_f8_rows=False):
    """out = act(A W^T + bias) [+ rowadd[m//div]] [+ residuals].  a: lowp [M][lda>=K]; w: packed lowp [N][Kpad].
    split "w2f8" (+ w_scale): a = rows [K fp16 | K fp8] (layernorm_f8), w / w_scale from pack_linear_weight_f8; out_f8_rows (with act="gelu"): out_lp
    is [M][3 N / 2] = rows [N fp16 | N fp8], the operand of the next "w2f8" GEMM.
    split "w2" / "x3": w packed with split=True ([hi | lo] planes), "x3" also takes a_lo (f3r.h f3r_split); *_lo: low planes of the lowp
    residuals / output.  Returns (out_f32, out_lp) or, with want_lo / out_lp_lo, (out_f32, out_lp, out_lp_lo)."""
    require_gpu(a, "a")
    lp = a.dtype
    assert w.dtype == lp and a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1
    M = a.shape[0]
    N, Kpad = w.shape
    K = a.shape[1] if K is None else K
    if split == "w2f8":
        assert w_scale is not None and w_scale.dtype == torch.int32 and w_scale.numel() == N and lp == torch.float16
        K = Kpad = 2 * w.shape[1] // 3
        assert a.shape[1] * 2 >= 3 * K
    if want_f32 and out_f32 is None:
        out_f32 = torch.empty((M, N), dtype=torch.float32, device=a.device)
    if out_f8_rows:
        assert split == "w2f8" and act == "gelu"
        if out_lp is None:
            out_lp = torch.empty((M, 3 * N // 2), dtype=lp, device=a.device)
        assert out_lp.shape[1] * 2 >= 3 * N
    if (want_lp or want_lo) and out_lp is None:
        out_lp = torch.empty((M, N), dtype=lp, device=a.device)
    if want_lo and out_lp_lo is None:
        out_lp_lo = torch.empty((M, N), dtype=lp, device=a.device)
    g = GemmArgs()
    g.A, g.W, g.bias = ptr(a), ptr(w), ptr(bias)
    g.M, g.N, g.K, g.Kpad, g.lda = M, N, K, Kpad, a.stride(0)
    g.a_mode, g.epi, g.act = F3R_A_PLAIN, F3R_EPI_GENERIC, ACT[act]
    g.kernel_sel = kernel_sel
    _split_operand(g, a, split, a_lo)
    if rowadd is not None:
        g.rowadd, g.rowadd_div = ptr(rowadd), rowadd_div
    if res_f32 is not None:
        g.res_f32, g.ldr_f32 = ptr(res_f32), res_f32.stride(0)
    if res_lp is not None:
        g.res_lp, g.ldr_lp = ptr(res_lp), res_lp.stride(0)
    if res_lp2 is not None:
        g.res_lp2, g.ldr_lp2 = ptr(res_lp2), res_lp2.stride(0)
    if out_f32 is not None:
        g.out_f32, g.ldo_f32 = ptr(out_f32), out_f32.stride(0)
    if out_lp is not None:
        g.out_lp, g.ldo_lp = ptr(out_lp), out_lp.stride(0)
    if out_lp_lo is not None:
        assert out_lp is not None and out_lp_lo.stride(0) == out_lp.stride(0)
        g.out_lp_lo = ptr(out_lp_lo)
    if res_lp_lo is not None:
        assert res_lp is not None and res_lp_lo.stride(0) == res_lp.stride(0)
        g.res_lp_lo = ptr(res_lp_lo)
    if res_lp2_lo is not None:
        assert res_lp2 is not None and res_lp2_lo.stride(0) == res_lp2.stride(0)
        g.res_lp2_lo = ptr(res_lp2_lo)
    g.dtype = dtype_id(lp)
    if w_scale is not None:
        g.w_scale = ptr(w_scale)
    g.out_lp_f8 = int(bool(out_f8_rows))
    with _timed(2.0 * M * N * K):
        check(_lib.lib().f3r_gemm(ctypes.byref(g), stream_ptr()), "f3r_gemm")
    if out_lp_lo is not None:
        return out_f32, out_lp, out_lp_lo
    return out_f32, out_lp


class BlockWorkspace:
    """The intermediates of a transformer block -- LN / attention output, q, k, V^T, MLP hidden -- as views of ONE allocation laid out by
    f3r_block_workspace_bytes (include/f3r.h): made once per encoder pass / decoder sample, reused by every block of it."""

    def __init__(self, tokens, D, hidden, n_seq, seq_len, lp, device, kv_dim=None, f8_rows=False):
        """f8_rows: the pass runs GEMMs with the fp8 low plane (Fast3R.low_plane): the LayerNorm-output and hidden-state regions are sized for rows
        [w fp16 | w fp8] (rows8 / hid8) and the plain 16-bit forms (h, hid) alias their heads (f3r_block_workspace_bytes_ex) -- no second copy of
        either intermediate (~5 GB at N = 320, ~16 GB at N = 1500 on one GPU)."""
        kv_dim = D if kv_dim is None else kv_dim
        offs = (ctypes.c_size_t * 5)()
        f8_rows = bool(f8_rows) and lp == torch.float16 and D % 8 == 0 and hidden % 8 == 0
        total = _lib.lib().f3r_block_workspace_bytes_ex(tokens, D, kv_dim, hidden, n_seq, seq_len, int(f8_rows), offs)
        if total == 0:
            raise ValueError(f"f3r_block_workspace_bytes: bad sizes ({tokens=}, {D=}, {hidden=}, {n_seq=}, {seq_len=})")
        self.buf = torch.empty((total,), dtype=torch.uint8, device=device)
        ld = vt_ld(seq_len)

        def view(i, shape, wide=False):
            n = 1
            for d in shape:
                n *= d
            return self.buf[offs[i]:offs[i] + 2 * n].view(lp).view(shape)
        self.h, self.q, self.k = view(0, (tokens, D)), view(1, (tokens, D)), view(2, (tokens, kv_dim))
        self.vt, self.hid = view(3, (n_seq, kv_dim, ld)), view(4, (tokens, hidden))
        if ld != seq_len and kv_dim:
            self.vt.zero_()  # the pad columns are read by the last key tile and never written by the QKV epilogue
        self.key = (tokens, D, hidden, n_seq, seq_len, lp, str(device))
        self._rows8 = view(0, (tokens, 3 * D // 2)) if f8_rows else None
        self._hid8 = view(4, (tokens, 3 * hidden // 2)) if f8_rows else None

    def hid8(self):
        """[tokens][3 hidden / 2]: the MLP hidden state as rows [hidden fp16 | hidden fp8] (fc1's GELU epilogue writes both, fc2 reads both).  With
        f8_rows it IS the hidden-state region (self.hid aliases its head: a pass uses one form or the other)."""
        if self._hid8 is None:
            self._hid8 = torch.empty((self.hid.shape[0], 3 * self.hid.shape[1] // 2), dtype=torch.float16, device=self.buf.device)
        return self._hid8

    def rows8(self, D):
        """[tokens][3 D / 2] float16-typed rows [D fp16 | D fp8] for the LayerNorm output of the fp8-low-plane GEMMs.  With f8_rows it IS region 0:
        self.h (the attention output) aliases its head -- the LayerNorm rows are dead once the QKV / fc1 launch that reads them has run, the
        attention output is dead before the next LayerNorm writes."""
        if self._rows8 is None:
            self._rows8 = torch.empty((self.h.shape[0], 3 * D // 2), dtype=torch.float16, device=self.buf.device)
        assert self._rows8.shape[1] == 3 * D // 2
        return self._rows8


def vt_ld(seq_len: int) -> int:
    """Row stride of a V^T buffer: the attention kernel reads whole 64-key tiles, so rows are padded to 64 (zeroed)."""
    return round_up(seq_len, 64)


LOG2E = 1.4426950408889634


def gemm_qkv(a, w, bias, q, k, vt, seq_len, rope=None, q_scale=0.0, rope_mode=0, split=None, a_lo=None, kernel_sel=0, q_dim=0, w_scale=None, w_aux=None):
    """QKV projection with the attention-layout epilogue: q -> [M][Dq], k -> [M][Dkv] (optionally RoPE'd), v -> vt[M/seq][Dkv][ldvt]
    (Dq = Dkv = N/3 unless q_dim is given: grouped-query attention).
    rope = (cos, sin, tokens_per_row) or None; rope_mode 0 = RoPE-2D tables [n_pos][16], 1 = per-row-group tables [n_groups][32]
    (rope[2] = rows per group; see f3r_gemm_args.rope_mode).  vt must be zero-initialised once (its padding is never written).
    split "w2f8": a = rows [K fp16 | K fp8] (layernorm_f8); w / w_scale = pack_linear_weight_f8 of the q and k rows; w_aux = the v rows packed
    with pack_linear_weight(split=True) (f3r_gemm_args.W_aux: the V^T launch runs on two fp16 planes)."""
    require_gpu(a, "a")
    lp = a.dtype
    M = a.shape[0]
    N, Kpad = w.shape
    K = a.shape[1]
    if split == "w2f8":
        assert w_scale is not None and w_aux is not None and lp == torch.float16 and w_aux.dtype == lp
        K = Kpad = 2 * w.shape[1] // 3
        N = w.shape[0] + w_aux.shape[0]
        assert w_aux.shape[1] == 2 * K and a.shape[1] * 2 >= 3 * K and w_scale.numel() == w.shape[0]
    g = GemmArgs()
    g.A, g.W, g.bias = ptr(a), ptr(w), ptr(bias)
    g.M, g.N, g.K, g.Kpad, g.lda = M, N, K, Kpad, a.stride(0)
    if split == "w2f8":
        g.w_scale, g.W_aux = ptr(w_scale), ptr(w_aux)
    g.a_mode, g.epi, g.act = F3R_A_PLAIN, F3R_EPI_QKV, F3R_ACT_NONE
    g.q, g.k, g.vt = ptr(q), ptr(k), ptr(vt)
    g.seq_len, g.ldvt = seq_len, vt.stride(-2)
    if rope is not None:
        g.rope_cos, g.rope_sin, g.rope_w = ptr(rope[0]), ptr(rope[1]), rope[2]
        g.rope_mode = int(rope_mode)
    g.q_scale = float(q_scale)
    g.qkv_dq = int(q_dim)  # 0: three equal thirds; else the q part is q_dim columns and k, v share the rest (grouped-query attention)
    g.kernel_sel = kernel_sel
    _split_operand(g, a, split, a_lo)
    g.dtype = dtype_id(lp)
    with _timed(2.0 * M * N * K):
        check(_lib.lib().f3r_gemm(ctypes.byref(g), stream_ptr()), "f3r_gemm(qkv)")


def silu_mul(ab, hidden, out=None):
    """SwiGLU gate: ab lowp [rows][2*hidden] = [w1 x | w3 x] -> silu(w1 x) * (w3 x), lowp [rows][hidden] (llama.py:284)."""
    require_gpu(ab, "ab")
    assert ab.dim() == 2 and ab.shape[1] == 2 * hidden and ab.is_contiguous()
    if out is None:
        out = torch.empty((ab.shape[0], hidden), dtype=ab.dtype, device=ab.device)
    with _timed(0.0, _nb(ab, out), "elementwise"):
        check(_lib.lib().f3r_silu_mul(ptr(ab), ptr(out), ab.shape[0], hidden, dtype_id(ab.dtype), stream_ptr()), "f3r_silu_mul")
    return out


def rows_add(x, vec, rows):
    """x[:rows] += vec in place (fp32 residual stream; fast3r.py:957-958 `x + view0_mask * view0_embed`)."""
    require_gpu(x, "x")
    assert x.dtype == torch.float32 and x.is_contiguous() and vec.dtype == torch.float32 and vec.numel() == x.shape[-1]
    check(_lib.lib().f3r_rows_add_f32(ptr(x), ptr(vec), int(rows), x.shape[-1], stream_ptr()), "f3r_rows_add_f32")
    return x


def conv3x3(x, w, *, stride=1, bias=None, a_relu=False, act=None, res_lp=None, res_lp2=None, out=None, split=None, x_lo=None,
            res_lp_lo=None, res_lp2_lo=None, want_lo=False, want_relu=False, kernel_sel=0, x_f8=None, w_scale=None, want_f8=False,
            want_relu_lo=None, want_relu_f8=False, fin=None):
    """3x3 conv, pad 1, NHWC lowp in/out, as an implicit GEMM.  x: (B,H,W,C); w: pack_conv3x3_weight(...).
    Returns the output, or a dict {"out", "out_lo", "out_f8", "relu", "relu_lo", "relu_f8"} when low planes (want_lo), fp8 planes (want_f8: uint8
    (B,OH,OW,2N), the operand of a following split="x3f8" conv) or the pre-activated copy relu(out) (want_relu; what the next ResidualConvUnit conv
    reads; want_relu_lo [default: want_lo] / want_relu_f8 pick its planes) are asked for.
    split "x3f8": w / w_scale from pack_conv3x3_weight_f8, x_f8 = the fp8 planes of x (include/f3r.h F3R_SPLIT_X3F8).
    fin = (w4 fp32 [4][N], b4 fp32 [4], n_out, depth_mode id, conf_mode id, vmin, vmax, want_conf): the fused tail of the DPT head (f3r_gemm_args.fin_w)
    -> returns (pts3d (B,OH,OW,3) fp32, conf (B,OH,OW) fp32 or None) and writes no activation."""
    require_gpu(x, "x")
    lp = x.dtype
    assert x.is_contiguous() and x.dim() == 4
    B, H, W, C = x.shape
    OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    N, Kpad = w.shape
    g = GemmArgs()
    g.kernel_sel = kernel_sel
    if split == "x3f8":
        assert x_f8 is not None and x_f8.dtype == torch.uint8 and x_f8.shape == (B, H, W, 2 * C) and x_f8.is_contiguous() and w_scale is not None
        g.split, g.A_lo, g.w_scale = F3R_SPLIT_X3F8, ptr(x_f8), ptr(w_scale)
    else:
        _split_operand(g, x, split, x_lo)
    extra = {}
    pts = conf = None
    if fin is not None:
        fw, fb, n_out, dmode, cmode, vmin, vmax, want_conf = fin
        assert fw.dtype == torch.float32 and fw.shape == (4, N) and fw.is_contiguous() and fb.dtype == torch.float32 and fb.numel() == 4
        pts = torch.empty((B, OH, OW, 3), dtype=torch.float32, device=x.device)
        conf = torch.empty((B, OH, OW), dtype=torch.float32, device=x.device) if want_conf else None
        g.fin_w, g.fin_b, g.fin_pts, g.fin_conf = ptr(fw), ptr(fb), ptr(pts), ptr(conf)
        g.fin_n_out, g.fin_depth_mode, g.fin_conf_mode, g.fin_vmin, g.fin_vmax = n_out, dmode, cmode, vmin, vmax
        g.ldo_lp = N
    else:
        if out is None:
            out = torch.empty((B, OH, OW, N), dtype=lp, device=x.device)
        g.out_lp, g.ldo_lp = ptr(out), N
        if want_lo:
            extra["out_lo"] = torch.empty_like(out)
            g.out_lp_lo = ptr(extra["out_lo"])
        if want_f8:
            extra["out_f8"] = torch.empty((B, OH, OW, 2 * N), dtype=torch.uint8, device=x.device)
            g.out_f8 = ptr(extra["out_f8"])
        if want_relu:
            extra["relu"] = torch.empty_like(out)
            g.out_relu = ptr(extra["relu"])
            if want_lo if want_relu_lo is None else want_relu_lo:
                extra["relu_lo"] = torch.empty_like(out)
                g.out_relu_lo = ptr(extra["relu_lo"])
            if want_relu_f8:
                extra["relu_f8"] = torch.empty((B, OH, OW, 2 * N), dtype=torch.uint8, device=x.device)
                g.out_relu_f8 = ptr(extra["relu_f8"])
    if res_lp_lo is not None:
        g.res_lp_lo = ptr(res_lp_lo)
    if res_lp2_lo is not None:
        g.res_lp2_lo = ptr(res_lp2_lo)
    g.A, g.W, g.bias = ptr(x), ptr(w), ptr(bias)
    g.M, g.N, g.K, g.Kpad, g.lda = B * OH * OW, N, 0, Kpad, 0
    g.a_mode, g.a_relu = F3R_A_CONV3X3, int(a_relu)
    g.conv_H, g.conv_W, g.conv_C, g.conv_stride, g.conv_OH, g.conv_OW = H, W, C, stride, OH, OW
    g.epi, g.act = F3R_EPI_GENERIC, ACT[act]
    if res_lp is not None:
        g.res_lp, g.ldr_lp = ptr(res_lp), N
    if res_lp2 is not None:
        g.res_lp2, g.ldr_lp2 = ptr(res_lp2), N
    g.dtype = dtype_id(lp)
    with _timed(2.0 * B * OH * OW * N * 9 * C + (8.0 * B * OH * OW * N if fin is not None else 0.0)):
        check(_lib.lib().f3r_gemm(ctypes.byref(g), stream_ptr()), "f3r_gemm(conv3x3)")
    if fin is not None:
        return pts, conf
    if extra:
        extra["out"] = out
        return extra
    return out


def convT(x, w, bias_tiled, s, cout, split=None, x_lo=None, want_lo=False, kernel_sel=0):
    """ConvTranspose2d with kernel == stride == s: x (B,h,w,Cin) NHWC lowp -> (B,h*s,w*s,cout) [, its low plane with want_lo]."""
    require_gpu(x, "x")
    lp = x.dtype
    B, h, wd, cin = x.shape
    out = torch.empty((B, h * s, wd * s, cout), dtype=lp, device=x.device)
    g = GemmArgs()
    g.kernel_sel = kernel_sel
    _split_operand(g, x, split, x_lo)
    out_lo = None
    if want_lo:
        out_lo = torch.empty_like(out)
        g.out_lp_lo = ptr(out_lo)
    g.A, g.W, g.bias = ptr(x), ptr(w), ptr(bias_tiled)
    g.M, g.N, g.K, g.Kpad, g.lda = B * h * wd, s * s * cout, cin, w.shape[1], cin
    g.a_mode, g.epi, g.act = F3R_A_PLAIN, F3R_EPI_CONVT, F3R_ACT_NONE
    g.out_lp, g.ldo_lp = ptr(out), cout
    g.ct_s, g.ct_h, g.ct_w, g.ct_cout = s, h, wd, cout
    g.dtype = dtype_id(lp)
    with _timed(2.0 * B * h * wd * s * s * cout * cin):
        check(_lib.lib().f3r_gemm(ctypes.byref(g), stream_ptr()), "f3r_gemm(convT)")
    return (out, out_lo) if want_lo else out


def attention_state(tq_total: int, n_heads: int, device, head_dim: int = 64):
    """Buffers for an online softmax carried across launches: (st_o fp32 [tq][heads*head_dim], st_ml fp32 [tq][heads][4])."""
    return (torch.empty((tq_total, n_heads * head_dim), dtype=torch.float32, device=device),
            torch.empty((tq_total, n_heads, 4), dtype=torch.float32, device=device))


def attention(q, out, n_heads, scale, segments, tq=None, batch=1, q_batch_stride=0, o_batch_stride=0, q_prescaled=False,
              state=None, state_in=False, state_out=False, kv_group=1, causal=False, q_pos0=0, seg_pos0=None, kernel_sel=0, head_dim=64, qk_planes=1,
              reserve_cus=0):
    """O = softmax(scale Q K^T) V.  q/out: lowp [batch][tq][ld].  segments: list of (k, vt, seg_len, k_bstride, vt_bstride)
    with k [..][seg_len][ldk] and vt [..][kv_heads*64][ldvt].  kv_group: query heads per K / V head (grouped-query attention).
    causal: key position <= query position only, positions = q_pos0 + row / seg_pos0[s] + row (global token indices).
    kernel_sel: 0 = automatic, 1 = the general HIP kernel, 2 = the hand-scheduled kernel (include/f3r.h, f3r_attn_args.kernel_sel).
    qk_planes = 2: q and k rows hold [hi 64 | lo 64] per head (qkv_planes): three products per score block (precision "robust"; fp16, head_dim 64)."""
    require_gpu(q, "q")
    lp = q.dtype
    assert 1 <= len(segments) <= F3R_MAX_SEG
    a = AttnArgs()
    a.q, a.o = ptr(q), ptr(out)
    a.ldq, a.ldo = q.stride(-2), out.stride(-2)
    a.q_batch_stride, a.o_batch_stride = q_batch_stride, o_batch_stride
    a.tq = q.shape[-2] if tq is None else tq
    a.batch, a.n_heads, a.n_seg, a.dtype = batch, n_heads, len(segments), dtype_id(lp)
    for i, (k, vt, seg_len, kbs, vbs) in enumerate(segments):
        assert k.dtype == lp and vt.dtype == lp
        a.k_seg[i], a.vt_seg[i] = ptr(k), ptr(vt)
        a.seg_len[i], a.ldvt[i] = seg_len, vt.stride(-2)
        a.k_batch_stride[i], a.vt_batch_stride[i] = kbs, vbs
        a.ldk = k.stride(-2)
    a.scale = float(scale)
    a.q_prescaled = int(q_prescaled)
    a.kv_group, a.causal, a.q_pos0 = int(kv_group), int(bool(causal)), int(q_pos0)
    a.kernel_sel = int(kernel_sel)
    a.head_dim = int(head_dim)
    a.qk_planes = int(qk_planes)
    a.reserve_cus = int(reserve_cus)   # CUs the persistent form leaves free (f3r_attn_args.reserve_cus): a rank's local-shard launch next to the exchange
    if causal:
        pos = [0] * len(segments)
        if seg_pos0 is None:  # consecutive segments of one sequence starting at position 0
            run = 0
            for i, sgm in enumerate(segments):
                pos[i] = run
                run += int(sgm[2])
        else:
            pos = [int(v) for v in seg_pos0]
        for i, v in enumerate(pos):
            a.seg_pos0[i] = v
    if ATTN_COUNTERS is not None:
        assert ATTN_COUNTERS.dtype == torch.int32 and ATTN_COUNTERS.numel() >= 56 and ATTN_COUNTERS.data_ptr() % 8 == 0 and ATTN_COUNTERS.device == q.device
        a.dbg_counters = ptr(ATTN_COUNTERS)
    if state is not None:
        a.st_o, a.st_ml = ptr(state[0]), ptr(state[1])
        a.state_in, a.state_out = int(state_in), int(state_out)
    if ATTN_WORK_STEALING:
        a.sched_counter = ptr(_sched_counter(q.device))
    if ATTN_TIMER is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    with _timed(4.0 * a.tq * sum(int(s_[2]) for s_ in segments) * head_dim * n_heads * batch, 0.0, "attention"):   # (algorithmic: one product per score)
        check(_lib.lib().f3r_attn_fwd(ctypes.byref(a), stream_ptr()), "f3r_attn_fwd")
    if ATTN_TIMER is not None:
        e1.record()
        t_k = sum(int(s[2]) for s in segments)
        ATTN_TIMER.append((e0, e1, 4.0 * a.tq * t_k * head_dim * n_heads * batch, int(a.tq), int(t_k),
                           _lib.lib().f3r_attn_kernel_name(ctypes.byref(a)).decode()))
    return out


def qkv_planes(qkv, n_heads, kv_heads, n_seq, seq_len, q_scale, lp, planes=2):
    """precision "robust": fp32 qkv [n_seq * seq_len][Dq + 2 Dkv] (rotary embedding applied) -> (q rows [T][n_heads * 128] = per head [hi 64 | lo 64] of
    q * q_scale, k rows [T][kv_heads * 128], V^T [n_seq][kv_heads * 64][ldvt] one plane): the operands of attention(..., qk_planes=2).  planes = 3
    (fp16): the lo half of every head holds [e4m3(hi) 64 B | e4m3(lo 2^12) 64 B] instead -- attention(..., qk_planes=3), corrections on the fp8 MFMA."""
    require_gpu(qkv, "qkv")
    assert qkv.dtype == torch.float32 and qkv.dim() == 2 and qkv.stride(1) == 1 and qkv.shape[0] == n_seq * seq_len
    T = qkv.shape[0]
    qp = torch.empty((T, n_heads * 128), dtype=lp, device=qkv.device)
    kp = torch.empty((T, kv_heads * 128), dtype=lp, device=qkv.device)
    ld = vt_ld(seq_len)
    vt = torch.empty((n_seq, kv_heads * 64, ld), dtype=lp, device=qkv.device)
    with _timed(0.0, _nb(qkv, qp, kp, vt), "elementwise"):
        check(_lib.lib().f3r_qkv_planes(ptr(qkv), qkv.stride(0), n_seq, seq_len, n_heads, kv_heads, float(q_scale), ptr(qp), ptr(kp), ptr(vt), ld, dtype_id(lp),
                                        int(planes), stream_ptr()), "f3r_qkv_planes")
    return qp, kp, vt


def attention_state_finish(state, n_heads, head_dim, lp, want_lo=True, want_f32=False):
    """the parked online-softmax state of an attention launch (state_out) -> O / l as (hi, lo) lowp planes [T][n_heads * head_dim] [, fp32]"""
    st_o, st_ml = state
    require_gpu(st_o, "st_o")
    T, D = st_o.shape[0], n_heads * head_dim
    assert st_o.shape == (T, D) and st_ml.shape == (T, n_heads, 4) and st_o.is_contiguous() and st_ml.is_contiguous()
    o_hi = torch.empty((T, D), dtype=lp, device=st_o.device)
    o_lo = torch.empty((T, D), dtype=lp, device=st_o.device) if want_lo else None
    o32 = torch.empty((T, D), dtype=torch.float32, device=st_o.device) if want_f32 else None
    with _timed(0.0, _nb(st_o, st_ml, o_hi, o_lo, o32), "elementwise"):
        check(_lib.lib().f3r_attn_state_finish(ptr(st_o), ptr(st_ml), T, n_heads, head_dim, ptr(o_hi), ptr(o_lo), ptr(o32), D, dtype_id(lp), stream_ptr()),
              "f3r_attn_state_finish")
    return (o_hi, o_lo, o32) if want_f32 else (o_hi, o_lo)


def upsample2x(x, out_hw=None, x_lo=None, want_lo=False, want_f8=False):
    """bilinear x2, align_corners=True, NHWC lowp; optional crop to out_hw.  x_lo: low plane of the input; want_lo: also return the
    low plane of the output (split precision); want_f8: also its fp8 planes (see interp_bilinear)."""
    B, h, w, C = x.shape
    oh, ow = (2 * h, 2 * w) if out_hw is None else out_hw
    return interp_bilinear(x, (oh, ow), x_lo=x_lo, want_lo=want_lo, want_f8=want_f8, nominal_hw=(2 * h, 2 * w))


DEPTH_MODES = {"exp": 0, "linear": 1, "square": 2}
CONF_MODES = {"exp": 0, "sigmoid": 1}


def interp_bilinear(x, full_hw, x_lo=None, want_lo=False, want_f8=False, nominal_hw=None):
    """F.interpolate(x, size=full_hw, mode="bilinear", align_corners=True) on NHWC lowp (+ low planes), any output size (nominal_hw: the
    interpolation's own output size when full_hw is a crop of it).  want_f8: also the fp8 planes uint8 (B,oh,ow,2C) a split="x3f8" conv reads
    (f3r_interp_bilinear_f8).  Returns out, or (out, out_lo) with want_lo, or (out, out_lo | None, out_f8) with want_f8."""
    require_gpu(x, "x")
    B, h, w, C = x.shape
    oh, ow = full_hw
    fh, fw = (oh, ow) if nominal_hw is None else nominal_hw
    out = torch.empty((B, oh, ow, C), dtype=x.dtype, device=x.device)
    out_lo = torch.empty_like(out) if want_lo else None
    out_f8 = torch.empty((B, oh, ow, 2 * C), dtype=torch.uint8, device=x.device) if want_f8 else None
    with _timed(0.0, _nb(x, x_lo, out, out_lo, out_f8), "elementwise"):
        check(_lib.lib().f3r_interp_bilinear_f8(ptr(x), ptr(x_lo), ptr(out), ptr(out_lo), ptr(out_f8), B, h, w, C, fh, fw, oh, ow, dtype_id(x.dtype),
                                                stream_ptr()), "f3r_interp_bilinear")
    if want_f8:
        return out, out_lo, out_f8
    return (out, out_lo) if want_lo else out


def dpt_fin_args(w, b, conf_mode, depth_mode=("exp", -math.inf, math.inf)):
    """The `fin` argument of conv3x3 (the DPT head's tail fused into head[2]'s epilogue) from head[4]'s weight (n_out, Cin) / bias and the
    reference's (mode, vmin, vmax) triples -- the same checks as dpt_final (heads/postprocess.py:27-64)."""
    n_out, cin = w.shape
    assert b.shape == (n_out,) and n_out == (4 if conf_mode is not None else 3)
    dmode, dmin, dmax = depth_mode
    if dmode not in DEPTH_MODES:
        raise ValueError(f"bad mode={dmode!r}")  # postprocess.py:51
    assert dmin == -math.inf and dmax == math.inf  # :33-34
    cmode, vmin, vmax = 0, 1.0, math.inf
    if conf_mode is not None:
        if conf_mode[0] not in CONF_MODES:
            raise ValueError(f"bad mode={conf_mode[0]!r}")  # :64
        cmode, vmin, vmax = CONF_MODES[conf_mode[0]], float(conf_mode[1]), float(conf_mode[2])
    w4 = torch.zeros((4, cin), dtype=torch.float32, device=w.device)
    w4[:n_out] = w.float()
    b4 = torch.zeros((4,), dtype=torch.float32, device=w.device)
    b4[:n_out] = b.float()
    return (w4.contiguous(), b4, n_out, DEPTH_MODES[dmode], cmode, vmin, vmax, conf_mode is not None)


def dpt_final(x, w, b, conf_mode, x_lo=None, depth_mode=("exp", -math.inf, math.inf)):
    """x (+ x_lo) (B,H,W,Cin) NHWC lowp -> pts3d (B,H,W,3) fp32, conf (B,H,W) fp32 (or None).  w (n_out, Cin), b (n_out,), n_out = 3 + has_conf.
    depth_mode / conf_mode: the reference's (mode, vmin, vmax) triples (heads/postprocess.py:27-64)."""
    require_gpu(x, "x")
    B, H, W, Cin = x.shape
    n_out = w.shape[0]
    assert w.shape == (n_out, Cin) and b.shape == (n_out,) and n_out == (4 if conf_mode is not None else 3)
    dmode, dmin, dmax = depth_mode
    if dmode not in DEPTH_MODES:
        raise ValueError(f"bad mode={dmode!r}")  # postprocess.py:51
    assert dmin == -math.inf and dmax == math.inf  # :33-34
    pts = torch.empty((B, H, W, 3), dtype=torch.float32, device=x.device)
    conf = None
    cmode, vmin, vmax = 0, 1.0, math.inf
    if conf_mode is not None:
        if conf_mode[0] not in CONF_MODES:
            raise ValueError(f"bad mode={conf_mode[0]!r}")  # :64
        conf = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
        cmode, vmin, vmax = CONF_MODES[conf_mode[0]], float(conf_mode[1]), float(conf_mode[2])
    with _timed(2.0 * B * H * W * Cin * n_out, _nb(x, x_lo, pts, conf), "elementwise"):
        check(_lib.lib().f3r_dpt_final(ptr(x), ptr(x_lo), ptr(w), ptr(b), n_out, ptr(pts), ptr(conf), B * H * W, Cin, DEPTH_MODES[dmode], cmode,
                                       vmin, vmax, dtype_id(x.dtype), stream_ptr()), "f3r_dpt_final")
    return pts, conf


def rope_f32(qkv, n_rot_heads, seq_len, rope, rope_mode=0):
    """precision "exact": rotary embedding in place on the first n_rot_heads 64-wide column groups (q heads, then k heads) of the fp32
    qkv[T][...] buffer; rope = (cos, sin, tokens_per_row | rows per group) and rope_mode as for gemm_qkv."""
    require_gpu(qkv, "qkv")
    assert qkv.dtype == torch.float32 and qkv.dim() == 2 and qkv.stride(1) == 1
    check(_lib.lib().f3r_rope_f32(ptr(qkv), qkv.shape[0], qkv.stride(0), int(n_rot_heads), seq_len, int(rope[2]), int(rope_mode), ptr(rope[0]),
                                  ptr(rope[1]), stream_ptr()), "f3r_rope_f32")
    return qkv


def rope2d_f32(qkv, n_heads, seq_len, rope):
    """RoPE-2D on the q and k parts (first 2 * n_heads * 64 columns) of the fp32 qkv[T][3D] buffer."""
    return rope_f32(qkv, 2 * n_heads, seq_len, rope, 0)


def silu_mul_f32(ab, hidden, lp):
    """precision "exact": SwiGLU gate on fp32 ab[rows][2*hidden] -> (hi, lo) lowp planes [rows][hidden] (llama.py:284)."""
    require_gpu(ab, "ab")
    assert ab.dtype == torch.float32 and ab.dim() == 2 and ab.shape[1] == 2 * hidden and ab.is_contiguous()
    hi = torch.empty((ab.shape[0], hidden), dtype=lp, device=ab.device)
    lo = torch.empty_like(hi)
    check(_lib.lib().f3r_silu_mul_f32(ptr(ab), ptr(hi), ptr(lo), ab.shape[0], int(hidden), dtype_id(lp), stream_ptr()), "f3r_silu_mul_f32")
    return hi, lo


# precision "exact": from this many keys per sequence on, attention_f32 runs on the matrix pipe (f3r_attn_f32_mfma: three-plane products, fp32
# softmax) instead of the FMA pipe (f3r_attn_f32_ex, ~30 TFLOP/s: the reference implementation of the mode, and what every small fixture uses)
ATTN_F32_MFMA_MIN_KEYS = 8192


def attention_f32(qkv, n_heads, n_seq, seq_len, scale, lp, want_f32=False, head_dim=64, kv_group=1, causal=False, kv=None, q_pos0=0, k_pos0=0):
    """precision "exact": fp32 softmax attention -> (o_hi, o_lo) lowp planes [T][D] (the A operand of the X3 projection) [, o fp32 with
    want_f32].  qkv fp32 [T][Dq + 2 Dkv] = q | k | v column blocks (head_dim per head; Dkv = Dq / kv_group: grouped-query heads);
    kv = (k, v) fp32 [n_seq * tk][Dkv] replaces the k / v columns of qkv (a view-sharded rank attending over the gathered keys of all
    ranks); causal masks by absolute position (q_pos0 / k_pos0 = position of the first query / key)."""
    require_gpu(qkv, "qkv")
    assert qkv.dtype == torch.float32 and qkv.dim() == 2 and qkv.stride(1) == 1
    T, D = qkv.shape[0], n_heads * head_dim
    Dkv = D // max(1, kv_group)
    assert qkv.shape[1] == D + 2 * Dkv and T == n_seq * seq_len
    o_hi = torch.empty((T, D), dtype=lp, device=qkv.device)
    o_lo = torch.empty((T, D), dtype=lp, device=qkv.device)
    o32 = torch.empty((T, D), dtype=torch.float32, device=qkv.device) if want_f32 else None
    a = _lib.AttnF32Args()
    base = qkv.data_ptr()
    a.q, a.ldq = base, qkv.stride(0)
    if kv is None:
        a.k, a.v, a.ldkv, a.tk = base + D * 4, base + (D + Dkv) * 4, qkv.stride(0), seq_len
    else:
        k, v = kv
        assert k.dtype == v.dtype == torch.float32 and k.shape == v.shape and k.shape[1] == Dkv and k.is_contiguous() and v.is_contiguous()
        assert k.shape[0] % max(1, n_seq) == 0
        a.k, a.v, a.ldkv, a.tk = k.data_ptr(), v.data_ptr(), Dkv, k.shape[0] // max(1, n_seq)
    a.o_hi, a.o_lo, a.o_f32, a.ldo = ptr(o_hi), ptr(o_lo), ptr(o32), D
    a.n_seq, a.tq, a.q_pos0, a.k_pos0 = n_seq, seq_len, int(q_pos0), int(k_pos0)
    a.n_heads, a.kv_group, a.causal, a.dtype, a.head_dim, a.scale = n_heads, max(1, int(kv_group)), int(bool(causal)), dtype_id(lp), int(head_dim), float(scale)
    ws_bytes = int(_lib.lib().f3r_attn_f32_mfma_workspace(ctypes.byref(a))) if a.tk >= ATTN_F32_MFMA_MIN_KEYS else 0
    with _timed(4.0 * a.tq * a.tk * head_dim * n_heads * n_seq, 0.0, "attention"):
        if ws_bytes > 0:  # the same attention as three-plane MFMA products (f3r_exact_mfma.hip): ~1e-6 of the FMA kernel at 10x its speed
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=qkv.device)
            check(_lib.lib().f3r_attn_f32_mfma(ctypes.byref(a), ptr(ws), ws_bytes, stream_ptr()), "f3r_attn_f32_mfma")
        else:
            check(_lib.lib().f3r_attn_f32_ex(ctypes.byref(a), stream_ptr()), "f3r_attn_f32_ex")
    return (o_hi, o_lo, o32) if want_f32 else (o_hi, o_lo)


def cast_lp(x, lp, out=None, want_lo=False):
    """fp32 -> lowp [, low plane lowp(x - float(hi)) with want_lo]."""
    require_gpu(x, "x")
    assert x.dtype == torch.float32 and x.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=lp, device=x.device)
    out_lo = torch.empty(x.shape, dtype=lp, device=x.device) if want_lo else None
    with _timed(0.0, _nb(x, out, out_lo), "elementwise"):
        check(_lib.lib().f3r_cast_f32_to_lp(ptr(x), ptr(out), ptr(out_lo), x.numel(), dtype_id(lp), stream_ptr()), "f3r_cast_f32_to_lp")
    return (out, out_lo) if want_lo else out
