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
    cos, sin = ops.rope_tables(64, 100.0, DEV)
    with pytest.raises((ValueError, RuntimeError)):  # rotary embedding stays on the compiler-scheduled kernel: forced -> error
        ops.gemm_qkv(a.to(DEV), wp, bias.to(DEV), q, k, vt, S, (cos, sin, 16), q_scale=qs, split=split, kernel_sel=ASM)


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
