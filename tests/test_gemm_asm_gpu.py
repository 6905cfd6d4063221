"""Parity of the hand-scheduled GEMM kernels (csrc/asm/gemm_gen.py, f3r_gemm_args.kernel_sel 6) on a real MI355X through the C ABI.

Reference = torch fp64 on the SAME 16-bit-rounded operands (single plane) or on the unrounded fp32 weights (W2 split: hi + lo planes
recover the fp32 weight).  Every case also runs the compiler-scheduled kernels (kernel_sel 7 = automatic without the hand-scheduled one)
and the two must agree with the reference; a forced launch that the kernel cannot take must raise, never fall back silently.
Shapes: one tile, several tiles per XCD, K-tile counts that stop the five-slot ring in each unrolled copy, both split modes, the
model's own N = 100 linear layers (102 400 x 1024 / 4096).  Replaces nn.Linear of fast3r/croco/models/blocks.py:94-105,125-131."""
import pytest
import torch
import torch.nn.functional as F

from fast3r_amd import ops
from test_kernels_gpu import DEV, DTYPES, assert_close, lp_tol, rnd

pytestmark = pytest.mark.gpu
ASM, HIP = 6, 7


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (512, 256, 320), (768, 512, 384), (2048, 1024, 448), (4096, 1024, 1024), (2304, 256, 4096), (1280, 768, 704),
                                   (70 * 256, 1024, 512), (300 * 256, 256, 576)])  # the last two: more tiles than CUs -> persistent workgroups, uneven tile counts
def test_gemm_asm_roles(built_lib, dt, M, N, K):
    a, w, bias = rnd((M, K), dt, 3), rnd((N, K), dt, 4, K ** -0.5), torch.randn(N)
    wp = ops.pack_linear_weight(w.float(), dt).to(DEV)
    base = a.double() @ w.double().t() + bias.double()
    x = torch.randn(M, N)
    for sel in (ASM, HIP):
        f32, _ = ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), want_f32=True, kernel_sel=sel)
        assert_close(f32, base, 2e-5, f"gemm f32 sel={sel}")
        xg = x.clone().to(DEV)  # fp32 residual, in place (x += proj(..): blocks.py:237-238)
        ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), res_f32=xg, out_f32=xg, kernel_sel=sel)
        assert_close(xg, base + x.double(), 2e-5, f"residual in place sel={sel}")
        xo = torch.empty_like(xg)  # residual read from another buffer
        ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), res_f32=x.to(DEV), out_f32=xo, kernel_sel=sel)
        assert_close(xo, base + x.double(), 2e-5, f"residual out of place sel={sel}")
        f32, _ = ops.gemm(a.to(DEV), wp, want_f32=True, kernel_sel=sel)  # no bias
        assert_close(f32, base - bias.double(), 2e-5, f"no bias sel={sel}")
        _, y = ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), want_lp=True, kernel_sel=sel)
        assert_close(y.float(), base, lp_tol(dt), f"lowp sel={sel}")
        _, y = ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), act="gelu", want_lp=True, kernel_sel=sel)
        assert_close(y.float(), F.gelu(base), lp_tol(dt), f"gelu sel={sel}")
        _, y = ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), act="relu", want_lp=True, kernel_sel=sel)
        assert_close(y.float(), F.relu(base), lp_tol(dt), f"relu sel={sel}")


@pytest.mark.parametrize("M,N,K", [(512, 256, 128), (1024, 1024, 1024), (2048, 512, 448), (80 * 256, 1024, 256)])
def test_gemm_asm_split_weights(built_lib, M, N, K):
    """W2: A W_hi + A W_lo as two K segments of the same loop -- fp32-class weights (vs fp64 on the UNROUNDED weight)"""
    dt = torch.float16
    a = rnd((M, K), dt, 5)
    w32 = torch.randn((N, K), generator=torch.Generator().manual_seed(6)) * K ** -0.5
    bias = torch.randn(N)
    wp = ops.pack_linear_weight(w32, dt, split=True).to(DEV)
    base = a.double() @ w32.double().t() + bias.double()
    x = torch.randn(M, N)
    for sel in (ASM, HIP):
        xg = x.clone().to(DEV)
        ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), res_f32=xg, out_f32=xg, split="w2", kernel_sel=sel)
        assert_close(xg, base + x.double(), 2e-5, f"w2 residual sel={sel}")
        _, y = ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), act="gelu", want_lp=True, split="w2", kernel_sel=sel)
        assert_close(y.float(), F.gelu(base), lp_tol(dt), f"w2 gelu sel={sel}")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n_seq,S,D,K,split", [(1, 1024, 256, 256, None), (3, 512, 512, 320, None), (2, 768, 256, 256, "w2"), (1, 70 * 256, 1024, 1024, None)])
def test_gemm_asm_qkv(built_lib, dt, n_seq, S, D, K, split):
    """QKV without rotary embedding as two launches of the hand-scheduled lowp role (q | k segments with the q scale; V^T with swapped
    operand roles) against fp64 and against the compiler-scheduled kernel's QKV epilogue"""
    if split and dt == torch.bfloat16:
        pytest.skip("the split planes are an fp16 design")
    M = n_seq * S
    a = rnd((M, K), dt, 21)
    w32 = torch.randn((3 * D, K), generator=torch.Generator().manual_seed(22)) * K ** -0.5
    bias = torch.randn(3 * D) * 0.3
    wp = ops.pack_linear_weight(w32, dt, split=bool(split)).to(DEV)
    w_eff = w32 if split else w32.to(dt).float()
    ref = a.double() @ w_eff.double().t() + bias.double()
    qs = 0.160192 * ops.LOG2E
    for sel in (ASM, HIP):
        q = torch.zeros((M, D), dtype=dt, device=DEV)
        k = torch.zeros((M, D), dtype=dt, device=DEV)
        ld = ops.vt_ld(S) + 64
        vt = torch.zeros((n_seq, D, ld), dtype=dt, device=DEV)
        ops.gemm_qkv(a.to(DEV), wp, bias.to(DEV), q, k, vt, S, None, q_scale=qs, split=split, kernel_sel=sel)
        assert_close(q.float(), ref[:, :D] * qs, lp_tol(dt), f"q sel={sel}")
        assert_close(k.float(), ref[:, D:2 * D], lp_tol(dt), f"k sel={sel}")
        got_v = vt[:, :, :S].float().cpu().permute(0, 2, 1).reshape(M, D)
        assert_close(got_v, ref[:, 2 * D:], lp_tol(dt), f"v^T sel={sel}")
        assert float(vt[:, :, S:].float().abs().sum()) == 0.0  # padding untouched
    # RoPE-2D fused into the q | k launch's epilogue (round 5: ACT_ROPE; pairs (i, i + 16) meet through v_permlane32_swap): tokens of a sequence on
    # a grid rope_w wide -- against the compiler-scheduled kernel's RoPE epilogue (the reference form, test_kernels_gpu.py) and fp64
    for rope_w in (16, 12, 1):
        n_pos = max(rope_w, -(-S // rope_w))
        cos, sin = ops.rope_tables(n_pos, 100.0, DEV)
        outs = {}
        for sel in (ASM, HIP):
            q = torch.zeros((M, D), dtype=dt, device=DEV)
            k = torch.zeros((M, D), dtype=dt, device=DEV)
            vt = torch.zeros((n_seq, D, ops.vt_ld(S)), dtype=dt, device=DEV)
            ops.gemm_qkv(a.to(DEV), wp, bias.to(DEV), q, k, vt, S, (cos, sin, rope_w), q_scale=qs, split=split, kernel_sel=sel)
            outs[sel] = (q.float().cpu(), k.float().cpu(), vt.float().cpu())
        pos = torch.arange(M) % S
        py, px = pos // rope_w, pos % rope_w
        c64, s64 = cos.double().cpu(), sin.double().cpu()

        def rope(x):
            out = x.clone()
            for h0 in range(0, D, 64):
                for half, p in ((0, py), (1, px)):
                    a_, b_ = x[:, h0 + 32 * half:h0 + 32 * half + 16], x[:, h0 + 32 * half + 16:h0 + 32 * half + 32]
                    out[:, h0 + 32 * half:h0 + 32 * half + 16] = a_ * c64[p] - b_ * s64[p]
                    out[:, h0 + 32 * half + 16:h0 + 32 * half + 32] = b_ * c64[p] + a_ * s64[p]
            return out
        assert_close(outs[ASM][0], rope(ref[:, :D]) * qs, lp_tol(dt), f"rope q asm w={rope_w}")
        assert_close(outs[ASM][1], rope(ref[:, D:2 * D]), lp_tol(dt), f"rope k asm w={rope_w}")
        assert_close(outs[ASM][0], outs[HIP][0].double(), 2 * lp_tol(dt), "rope q asm vs HIP")
        assert_close(outs[ASM][1], outs[HIP][1].double(), 2 * lp_tol(dt), "rope k asm vs HIP")
        assert torch.equal(outs[ASM][2], outs[HIP][2]) or (outs[ASM][2] - outs[HIP][2]).abs().max() <= 2 * lp_tol(dt) * outs[HIP][2].abs().max()
    cosg, sing = ops.rope_tables(64, 100.0, DEV)
    with pytest.raises((ValueError, RuntimeError)):  # the per-row-group form of the LlamaDecoder stays on the compiler-scheduled kernel: forced -> error
        ops.gemm_qkv(a.to(DEV), wp, bias.to(DEV), q, k, vt, S, (torch.cat([cosg, cosg], 1).contiguous(), torch.cat([sing, sing], 1).contiguous(), S), q_scale=qs,
                     split=split, rope_mode=1, kernel_sel=ASM)


def test_gemm_asm_strided_operand_and_outputs(built_lib):
    """A as a column block of a wider matrix (lda > K), outputs as column blocks of wider buffers (ldo > N)"""
    dt = torch.float16
    M, N, K = 1024, 512, 256
    big = rnd((M, K + 192), dt, 7).to(DEV)
    a = big[:, 64:64 + K]
    w, bias = rnd((N, K), dt, 8, K ** -0.5), torch.randn(N)
    wp = ops.pack_linear_weight(w.float(), dt).to(DEV)
    base = a.cpu().double() @ w.double().t() + bias.double()
    o32 = torch.zeros((M, N + 256), dtype=torch.float32, device=DEV)
    ops.gemm(a, wp, bias=bias.to(DEV), out_f32=o32[:, 128:128 + N], kernel_sel=ASM)
    assert_close(o32[:, 128:128 + N], base, 2e-5, "strided fp32 out")
    assert float(o32[:, :128].abs().sum()) == 0.0 and float(o32[:, 128 + N:].abs().sum()) == 0.0
    olp = torch.zeros((M, N + 256), dtype=dt, device=DEV)
    ops.gemm(a, wp, bias=bias.to(DEV), out_lp=olp[:, 128:128 + N], kernel_sel=ASM)
    assert_close(olp[:, 128:128 + N].float(), base, lp_tol(dt), "strided lowp out")
    assert float(olp[:, :128].float().abs().sum()) == 0.0 and float(olp[:, 128 + N:].float().abs().sum()) == 0.0


def test_gemm_asm_refuses_what_it_cannot_take(built_lib):
    dt = torch.float16
    a = rnd((300, 256), dt, 1).to(DEV)                                       # M not a multiple of 256
    wp = ops.pack_linear_weight(rnd((256, 256), dt, 2).float(), dt).to(DEV)
    with pytest.raises((ValueError, RuntimeError)):
        ops.gemm(a, wp, want_f32=True, kernel_sel=ASM)
    a = rnd((256, 256), dt, 1).to(DEV)
    with pytest.raises((ValueError, RuntimeError)):
        ops.gemm(a, wp, want_f32=True, want_lp=True, kernel_sel=ASM)          # two outputs
    with pytest.raises((ValueError, RuntimeError)):                          # fewer than 4 K-tiles
        ops.gemm(rnd((256, 128), dt, 1).to(DEV), ops.pack_linear_weight(rnd((256, 128), dt, 2).float(), dt).to(DEV), want_f32=True, kernel_sel=ASM)
    with pytest.raises((ValueError, RuntimeError)):
        ops.gemm(a, wp, act="gelu", want_f32=True, kernel_sel=ASM)            # activation on the fp32 role
    f32, _ = ops.gemm(rnd((300, 256), dt, 1).to(DEV), wp, want_f32=True)      # automatic: the compiler-scheduled kernels take it
    assert f32.shape == (300, 256)


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_asm_at_the_n100_shapes_matches_the_compiler_scheduled_kernel(built_lib, dt):
    """the model's linear layers at BASELINE config (3) (102 400 tokens): both kernels on the same operands, sampled rows vs fp64"""
    M = 102400
    g = torch.Generator(device=DEV).manual_seed(11)
    for N, K, act in ((1024, 1024, None), (4096, 1024, "gelu"), (1024, 4096, None)):
        a = torch.randn((M, K), generator=g, device=DEV).to(dt)
        w = (torch.randn((N, K), generator=g, device=DEV) * K ** -0.5).to(dt)
        bias = torch.randn(N, generator=g, device=DEV)
        wp = ops.pack_linear_weight(w.float(), dt)
        rows = torch.tensor([0, 255, 256, 51199, 102143, 102399], device=DEV)
        ref = a[rows].double() @ w.double().t() + bias.double()
        outs = []
        for sel in (ASM, HIP):
            if act:
                _, y = ops.gemm(a, wp, bias=bias, act=act, want_lp=True, kernel_sel=sel)
                assert_close(y[rows].float().cpu(), F.gelu(ref).cpu(), lp_tol(dt), f"{N}x{K} gelu sel={sel}")
                outs.append(y.float())
            else:
                x = torch.randn((M, N), generator=torch.Generator(device=DEV).manual_seed(12), device=DEV)
                x0 = x[rows].double()
                ops.gemm(a, wp, bias=bias, res_f32=x, out_f32=x, kernel_sel=sel)
                assert_close(x[rows].cpu(), (ref + x0).cpu(), 3e-5, f"{N}x{K} residual sel={sel}")
                outs.append(x)
        d = float((outs[0] - outs[1]).abs().max())
        assert d <= (lp_tol(dt) if act else 3e-5) * float(outs[1].abs().max()), d


def _f8_operands(M, K, N, seed, outliers=True):
    """activation rows [K fp16 | K fp8] as f3r_layernorm_f8 writes them, weight rows + scale words from pack_linear_weight_f8, and the planes decoded"""
    g = torch.Generator().manual_seed(seed)
    a32 = torch.randn((M, K), generator=g) * 2.0
    if outliers:
        a32.view(-1)[torch.randint(0, M * K, (60,), generator=g)] *= 300.0      # beyond +-448: clamped in the fp8 copy
    w32 = torch.randn((N, K), generator=g) * K ** -0.5 * torch.exp2(torch.randint(-6, 3, (N, 1), generator=g).float())
    a16 = a32.to(torch.float16)
    a8 = a16.float().clamp(-448, 448).to(torch.float8_e4m3fn)
    rows = torch.empty((M, 3 * K), dtype=torch.uint8)
    rows[:, :2 * K] = a16.contiguous().view(torch.uint8).view(M, 2 * K)
    rows[:, 2 * K:] = a8.view(torch.uint8)
    wp, ws = ops.pack_linear_weight_f8(w32)
    wb = wp.view(torch.uint8).view(N, 3 * K)
    w_hi = wb[:, :2 * K].contiguous().view(torch.float16).view(N, K).double()
    w_lo = wb[:, 2 * K:].contiguous().view(torch.float8_e4m3fn).double() * torch.exp2((ws & 0xFF).double() - 127.0)[:, None]
    planes = a16.double() @ w_hi.t() + a8.double() @ w_lo.t()
    return rows.view(torch.float16).view(M, 3 * K // 2), wp, ws, planes, a16.double() @ w32.double().t(), a16.double() @ w_hi.t()


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (768, 512, 384), (2048, 1024, 1024), (2304, 256, 4096), (70 * 256, 1024, 512), (300 * 256, 256, 640)])
def test_gemm_asm_fp8_low_plane(built_lib, M, N, K):
    """F3R_SPLIT_W2F8 (round 5): the K loop runs on from K / 64 fp16 K-tiles into K / 128 fp8 K-tiles of the same two operand streams (block-scaled
    v_mfma_scale_f32_32x32x64_f8f6f4, one power-of-two scale per output channel).  (a) the kernel computes exactly its planes: fp64 on the decoded
    fp16 + fp8 operands, both roles, bias / residual / GELU, persistent workgroups crossing output tiles; (b) what the format is for: the distance
    to the product with the UNROUNDED weight is several times smaller than with a single fp16 plane."""
    rows, wp, ws, planes, exact_w, single = _f8_operands(M, K, N, 11 + K)
    bias, x = torch.randn(N), torch.randn(M, N)
    base = planes + bias.double()
    rd, wd, sd = rows.to(DEV), wp.to(DEV), ws.to(DEV)
    f32, _ = ops.gemm(rd, wd, bias=bias.to(DEV), want_f32=True, split="w2f8", w_scale=sd)
    assert_close(f32, base, 2e-5, "w2f8 f32")
    xg = x.clone().to(DEV)
    ops.gemm(rd, wd, bias=bias.to(DEV), res_f32=xg, out_f32=xg, split="w2f8", w_scale=sd)
    assert_close(xg, base + x.double(), 2e-5, "w2f8 residual in place")
    f32n, _ = ops.gemm(rd, wd, want_f32=True, split="w2f8", w_scale=sd)
    assert_close(f32n, planes, 2e-5, "w2f8 no bias")
    _, y = ops.gemm(rd, wd, bias=bias.to(DEV), act="gelu", want_lp=True, split="w2f8", w_scale=sd)
    assert_close(y.float(), F.gelu(base), lp_tol(torch.float16), "w2f8 gelu")
    _, y = ops.gemm(rd, wd, bias=bias.to(DEV), want_lp=True, split="w2f8", w_scale=sd)
    assert_close(y.float(), base, lp_tol(torch.float16), "w2f8 lowp")
    # (b) on operands without out-of-range activations (the clamp is a property of the fp8 copy, not of the weight planes)
    rows, wp, ws, planes, exact_w, single = _f8_operands(M, K, N, 12 + K, outliers=False)
    got, _ = ops.gemm(rows.to(DEV), wp.to(DEV), want_f32=True, split="w2f8", w_scale=ws.to(DEV))
    e8 = float((got.double().cpu() - exact_w).abs().max() / exact_w.abs().max())
    e1 = float((single - exact_w).abs().max() / exact_w.abs().max())
    print(f"[w2f8] {M}x{N}x{K}: weight rounding left {e8:.2e} (single fp16 plane {e1:.2e})")
    assert e8 <= e1 / 6.0


def test_gemm_fp8_low_plane_refuses_what_it_cannot_take(built_lib):
    """no second kernel reads the [fp16 | fp8] rows: an ineligible launch is an error, never a wrong answer"""
    rows, wp, ws, *_ = _f8_operands(256, 256, 256, 3)
    with pytest.raises(Exception):
        ops.gemm(rows[:200].to(DEV), wp.to(DEV), want_f32=True, split="w2f8", w_scale=ws.to(DEV))       # M not a multiple of 256


def test_layernorm_f8_rows(built_lib):
    """f3r_layernorm_f8: the fp16 part of a row is bit-identical to f3r_layernorm's output, the fp8 part is e4m3(clamp(y, +-448)) of the fp32 result"""
    g = torch.Generator().manual_seed(2)
    x = (torch.randn((777, 1024), generator=g) * 3.0)
    x[5, 17] = 4000.0
    gamma, beta = 1 + 0.1 * torch.randn(1024, generator=g), 0.1 * torch.randn(1024, generator=g)
    gamma[40] = 300.0   # a channel whose output leaves the fp8 range
    xd, gd, bd = x.to(DEV), gamma.to(DEV), beta.to(DEV)
    plain, y32 = ops.layernorm(xd, gd, bd, 1e-6, torch.float16, want_f32=True)
    rows = ops.layernorm_f8(xd, gd, bd, 1e-6)
    assert rows.shape == (777, 1536) and torch.equal(rows[:, :1024], plain)
    got8 = rows.view(torch.uint8).view(777, 3072)[:, 2048:].contiguous()
    want8 = y32.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert (got8 != want8).float().mean() < 1e-4     # (an fp32 tie may round the other way in the fused kernel's own y)
    assert not ((got8 & 0x7F) == 0x7F).any()          # no NaN codes: the clamp came before the conversion
    rms = ops.layernorm_f8(xd, gd, None, 1e-5, rms=True)
    assert torch.equal(rms[:, :1024], ops.layernorm(xd, gd, None, 1e-5, torch.float16, rms=True)[0])


def test_mlp_chain_on_fp8_rows(built_lib):
    """LayerNorm rows -> fc1 (+GELU, writes rows [N fp16 | N fp8]) -> fc2 (+fp32 residual), both with the low plane in fp8, as Fast3R._block issues
    them with low_plane = "fp8": against fp64 on the UNROUNDED weights and the kernel's own fp16 activations (what W2 computes, to the
    fp8 planes' 2^-15)"""
    g = torch.Generator().manual_seed(21)
    T, D, Hd = 1536, 512, 2048
    x = torch.randn((T, D), generator=g) * 2
    gamma, beta = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    w1, b1 = torch.randn((Hd, D), generator=g) * D ** -0.5, 0.1 * torch.randn(Hd, generator=g)
    w2, b2 = torch.randn((D, Hd), generator=g) * Hd ** -0.5, 0.1 * torch.randn(D, generator=g)
    xd = x.to(DEV)
    rows = ops.layernorm_f8(xd, gamma.to(DEV), beta.to(DEV), 1e-6)
    w1p, w1s = ops.pack_linear_weight_f8(w1)
    w2p, w2s = ops.pack_linear_weight_f8(w2)
    _, hid = ops.gemm(rows, w1p.to(DEV), bias=b1.to(DEV), act="gelu", split="w2f8", w_scale=w1s.to(DEV), out_f8_rows=True)
    assert hid.shape == (T, 3 * Hd // 2)
    h16 = hid[:, :Hd]
    ref_h = F.gelu(rows[:, :D].double().cpu() @ w1.double().t() + b1.double())
    assert_close(h16.float(), ref_h, lp_tol(torch.float16), "fc1 + GELU on fp8 rows")
    h8 = hid.view(torch.uint8).view(T, 3 * Hd)[:, 2 * Hd:].contiguous().view(torch.float8_e4m3fn).float().cpu()
    want8 = h16.float().clamp(max=448).to(torch.float8_e4m3fn).float().cpu()
    assert ((h8 - want8).abs() > 0.13 * want8.abs().clamp_min(2.0 ** -9)).float().mean() < 2e-3   # (the copy is taken from the fp32 value, not from its fp16 rounding)
    out = xd.clone()
    ops.gemm(hid, w2p.to(DEV), bias=b2.to(DEV), res_f32=out, out_f32=out, split="w2f8", w_scale=w2s.to(DEV))
    ref = x.double() + h16.double().cpu() @ w2.double().t() + b2.double()
    assert_close(out, ref, 3e-5, "fc2 + residual on fp8 rows")


@pytest.mark.parametrize("n_seq,S,D,K,rope_w", [(2, 768, 256, 256, 0), (3, 512, 512, 384, 16), (1, 70 * 256, 1024, 1024, 0), (20, 1024, 1024, 1024, 32)])
def test_gemm_asm_qkv_fp8_low_plane(built_lib, n_seq, S, D, K, rope_w):
    """F3R_EPI_QKV with F3R_SPLIT_W2F8 (round 5): the q | k launch reads rows [K fp16 | K fp8] of both operands (block-scaled fp8 MFMA for the
    weight-correction product; RoPE-2D of the encoder in its epilogue), the V^T launch runs with swapped roles on two fp16 planes (W_aux).
    Reference: fp64 on exactly those planes."""
    M = n_seq * S
    g = torch.Generator().manual_seed(31 + K)
    a32 = torch.randn((M, K), generator=g) * 2.0
    w32 = torch.randn((3 * D, K), generator=g) * K ** -0.5
    bias = torch.randn(3 * D, generator=g) * 0.3
    a16 = a32.to(torch.float16)
    a8 = a16.float().clamp(-448, 448).to(torch.float8_e4m3fn)
    rows = torch.empty((M, 3 * K), dtype=torch.uint8)
    rows[:, :2 * K] = a16.contiguous().view(torch.uint8).view(M, 2 * K)
    rows[:, 2 * K:] = a8.view(torch.uint8)
    rows = rows.view(torch.float16).view(M, 3 * K // 2)
    qk8, qks = ops.pack_linear_weight_f8(w32[:2 * D])
    v2 = ops.pack_linear_weight(w32[2 * D:], torch.float16, True)
    wb = qk8.view(torch.uint8).view(2 * D, 3 * K)
    w_hi = wb[:, :2 * K].contiguous().view(torch.float16).view(2 * D, K).double()
    w_lo = wb[:, 2 * K:].contiguous().view(torch.float8_e4m3fn).double() * torch.exp2((qks & 0xFF).double() - 127.0)[:, None]
    ref_qk = a16.double() @ w_hi.t() + a8.double() @ w_lo.t() + bias[:2 * D].double()
    ref_v = a16.double() @ (v2[:, :K].double() + v2[:, K:].double()).t() + bias[2 * D:].double()
    qs = 0.160192 * ops.LOG2E
    rope = None
    if rope_w:
        n_pos = max(rope_w, -(-S // rope_w))
        cos, sin = ops.rope_tables(n_pos, 100.0, DEV)
        rope = (cos, sin, rope_w)
        pos = torch.arange(M) % S
        py, px = pos // rope_w, pos % rope_w
        c64, s64 = cos.double().cpu(), sin.double().cpu()

        def rot(x):
            out = x.clone()
            for h0 in range(0, D, 64):
                for half, p in ((0, py), (1, px)):
                    a_, b_ = x[:, h0 + 32 * half:h0 + 32 * half + 16], x[:, h0 + 32 * half + 16:h0 + 32 * half + 32]
                    out[:, h0 + 32 * half:h0 + 32 * half + 16] = a_ * c64[p] - b_ * s64[p]
                    out[:, h0 + 32 * half + 16:h0 + 32 * half + 32] = b_ * c64[p] + a_ * s64[p]
            return out
        ref_qk = torch.cat([rot(ref_qk[:, :D]), rot(ref_qk[:, D:])], 1)
    q = torch.zeros((M, D), dtype=torch.float16, device=DEV)
    k = torch.zeros((M, D), dtype=torch.float16, device=DEV)
    ld = ops.vt_ld(S) + 64
    vt = torch.zeros((n_seq, D, ld), dtype=torch.float16, device=DEV)
    ops.gemm_qkv(rows.to(DEV), qk8.to(DEV), bias.to(DEV), q, k, vt, S, rope, q_scale=qs, split="w2f8", w_scale=qks.to(DEV), w_aux=v2.to(DEV))
    assert_close(q.float(), ref_qk[:, :D] * qs, lp_tol(torch.float16), "q (fp8 low plane)")
    assert_close(k.float(), ref_qk[:, D:], lp_tol(torch.float16), "k (fp8 low plane)")
    got_v = vt[:, :, :S].float().cpu().permute(0, 2, 1).reshape(M, D)
    assert_close(got_v, ref_v, lp_tol(torch.float16), "v^T (two fp16 planes)")
    assert float(vt[:, :, S:].float().abs().sum()) == 0.0
