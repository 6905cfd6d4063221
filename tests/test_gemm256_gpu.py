"""Parity of the 256x256-tile GEMM kernel (f3r_gemm256.hip) and of the split-precision K segments, on a real MI355X through the C ABI.

Reference = torch fp64 on CPU on the SAME 16-bit-rounded operands (single-plane cases) or on the UNROUNDED fp32 operands (split
cases: there the claim is that hi + lo planes recover fp32-class accuracy).  Every case runs the 128-tile kernel (kernel_sel 1) and
both schedules of the 256-tile kernel (2 = staggered wave rows, 3 = lock-step) plus its 256 x 128 tile form (4; by shape it takes N = odd
multiples of 128 and the launches whose 256 x 256 tiles would leave CUs idle): all must agree with the reference, and the
shapes include ragged M / N, every epilogue the model uses on this path, and K-tile counts that are odd (the loop runs two tiles per
iteration) and minimal (one tile).
"""

import pytest
import torch
import torch.nn.functional as F

from fast3r_amd import ops
from test_kernels_gpu import DEV, DTYPES, _rope_ref, assert_close, lp_tol, rnd

pytestmark = pytest.mark.gpu
SELS = [1, 2, 3, 4, 0]  # 0 = the kernel and tile form f3r_gemm picks by shape


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(4096, 1024, 1024), (2048 + 37, 512, 768), (2048, 256, 4096), (3000, 384, 64), (2304, 1280, 192),
                                   (1500, 128, 256), (777, 640, 448)])  # the last three N: odd multiples of 128 -> the 256 x 128 tile form
def test_gemm256_outputs_and_residuals(built_lib, dt, M, N, K):
    a, w, bias = rnd((M, K), dt, 3), rnd((N, K), dt, 4, K ** -0.5), torch.randn(N)
    wp = ops.pack_linear_weight(w.float(), dt).to(DEV)
    base = a.double() @ w.double().t() + bias.double()
    x = torch.randn(M, N)
    P = 512
    rowadd = torch.randn((M + P - 1) // P, N)
    for sel in SELS:
        f32, lp = ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), want_f32=True, want_lp=True, kernel_sel=sel)
        assert_close(f32, base, 2e-5, f"gemm f32 sel={sel}")
        assert_close(lp.float(), base, lp_tol(dt), f"gemm lowp sel={sel}")
        xg = x.clone().to(DEV)  # fp32 residual, in place (x += proj(..): blocks.py:237-238)
        ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), res_f32=xg, out_f32=xg, kernel_sel=sel)
        assert_close(xg, base + x.double(), 2e-5, f"residual in place sel={sel}")
        f32, _ = ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), rowadd=rowadd.to(DEV), rowadd_div=P, want_f32=True, kernel_sel=sel)
        assert_close(f32, base + rowadd.double().repeat_interleave(P, 0)[:M], 2e-5, f"rowadd sel={sel}")
        _, y = ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), act="gelu", want_lp=True, kernel_sel=sel)
        assert_close(y.float(), F.gelu(base), lp_tol(dt), f"gelu sel={sel}")
        _, y = ops.gemm(a.to(DEV), wp, want_lp=True, kernel_sel=sel)  # no bias at all (LlamaDecoder projections)
        assert_close(y.float(), base - bias.double(), lp_tol(dt), f"no bias sel={sel}")


def test_gemm256_rejects_ineligible_shapes(built_lib):
    a = rnd((64, 96), torch.float16, 1).to(DEV)  # K = 96 < Kpad = 128: the LDS-DMA staging cannot zero-fill a K tail
    wp = ops.pack_linear_weight(rnd((128, 96), torch.float16, 2).float(), torch.float16).to(DEV)
    with pytest.raises(ValueError):
        ops.gemm(a, wp, want_f32=True, kernel_sel=2)  # a forced kernel is an error on a shape it cannot take, never a silent fallback


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n_seq,gh,gw,D,K,use_rope", [(2, 32, 32, 256, 256, True), (3, 24, 30, 256, 128, True), (1, 1, 2100, 512, 320, False)])
def test_gemm256_qkv_epilogue(built_lib, dt, n_seq, gh, gw, D, K, use_rope):
    S = gh * gw
    M = n_seq * S
    a, w, bias = rnd((M, K), dt, 11), rnd((3 * D, K), dt, 12, K ** -0.5), torch.randn(3 * D) * 0.1
    wp = ops.pack_linear_weight(w.float(), dt).to(DEV)
    ref = a.double() @ w.double().t() + bias.double()
    rq, rk, rv = ref[:, :D], ref[:, D:2 * D], ref[:, 2 * D:]
    rope = None
    if use_rope:
        cos, sin = ops.rope_tables(max(gh, gw), 100.0, DEV)
        rope = (cos, sin, gw)
        p = torch.arange(M) % S
        py, px = p // gw, p % gw
        rq = _rope_ref(rq.reshape(M, D // 64, 64), py, px, cos.cpu(), sin.cpu()).reshape(M, D)
        rk = _rope_ref(rk.reshape(M, D // 64, 64), py, px, cos.cpu(), sin.cpu()).reshape(M, D)
    qs = 0.160192 * ops.LOG2E
    for sel in SELS:
        q = torch.empty((M, D), dtype=dt, device=DEV)
        k = torch.empty((M, D), dtype=dt, device=DEV)
        ld = ops.vt_ld(S)
        vt = torch.zeros((n_seq, D, ld), dtype=dt, device=DEV)
        ops.gemm_qkv(a.to(DEV), wp, bias.to(DEV), q, k, vt, S, rope, q_scale=qs, kernel_sel=sel)
        assert_close(q.float(), rq * qs, lp_tol(dt), f"q sel={sel}")
        assert_close(k.float(), rk, lp_tol(dt), f"k sel={sel}")
        got_v = vt[:, :, :S].float().cpu().permute(0, 2, 1).reshape(M, D)
        assert_close(got_v, rv, lp_tol(dt), f"v^T sel={sel}")
        assert float(vt[:, :, S:].abs().sum()) == 0.0  # padding untouched


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 32, 40, 64, 256), (1, 48, 48, 256, 256), (3, 30, 31, 128, 384), (2, 33, 29, 128, 128), (1, 20, 50, 256, 128)])
def test_conv256_with_skip_connections(built_lib, dt, B, H, W, Ci, Co):
    """3x3 conv with its operand staged by LDS-DMA (out-of-image taps from the zero line) + bias + the two lowp skip adds + the
    pre-activated second output -- ResidualConvUnit_custom inside a fusion block (dpt_block.py:133-154,208-216)."""
    x = rnd((B, H, W, Ci), dt, 13)
    w = rnd((Co, Ci, 3, 3), dt, 14, (9 * Ci) ** -0.5)
    bias = torch.randn(Co)
    r1, r2 = rnd((B, H, W, Co), dt, 15), rnd((B, H, W, Co), dt, 16)
    wp = ops.pack_conv3x3_weight(w.float(), dt).to(DEV)
    conv = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    for sel in SELS:
        y = ops.conv3x3(x.to(DEV), wp, bias=bias.to(DEV), kernel_sel=sel)
        assert_close(y.float(), conv, lp_tol(dt), f"conv sel={sel}")
        y = ops.conv3x3(x.to(DEV), wp, bias=bias.to(DEV), act="relu", kernel_sel=sel)
        assert_close(y.float(), F.relu(conv), lp_tol(dt), f"conv+relu sel={sel}")
        r = ops.conv3x3(x.to(DEV), wp, bias=bias.to(DEV), res_lp=r1.to(DEV), res_lp2=r2.to(DEV), want_relu=True, kernel_sel=sel)
        assert_close(r["out"].float(), conv + r1.double() + r2.double(), lp_tol(dt), f"conv+2 skips sel={sel}")
        assert_close(r["relu"].float(), F.relu(conv + r1.double() + r2.double()), lp_tol(dt), f"relu copy sel={sel}")
        y = ops.conv3x3(x.to(DEV), wp, kernel_sel=sel)  # scratch.layer_rn: no bias
        assert_close(y.float(), conv - bias.double(), lp_tol(dt), f"conv no bias sel={sel}")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,h,w,Ci,Co,s", [(2, 32, 32, 128, 192, 2), (1, 40, 52, 64, 96, 4)])
def test_convT256(built_lib, dt, B, h, w, Ci, Co, s):
    x = rnd((B, h, w, Ci), dt, 18)
    wt = rnd((Ci, Co, s, s), dt, 19, Ci ** -0.5)
    bias = torch.randn(Co)
    wp, bt = ops.pack_convT_weight(wt.float(), bias, dt)
    ref = F.conv_transpose2d(x.double().permute(0, 3, 1, 2), wt.double(), bias.double(), stride=s).permute(0, 2, 3, 1)
    for sel in SELS:
        y = ops.convT(x.to(DEV), wp.to(DEV), bt.to(DEV), s, Co, kernel_sel=sel)
        assert y.shape == ref.shape
        assert_close(y.float(), ref, lp_tol(dt), f"convT sel={sel}")


# ------------------------------------------------------------------------------------------------ split precision
def split_tol(dt):
    """hi + lo planes: fp16 pieces carry ~22 significand bits, bf16 pieces 16.  w2 leaves the activation single (callers pass an
    exactly representable A), x3 drops only the lo x lo product."""
    return 3e-6 if dt == torch.float16 else 2e-4


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K,sels", [(200, 192, 128, [1]), (77, 260, 200, [1]), (4096, 512, 1024, SELS), (2048 + 5, 256, 64, SELS), (1000, 128, 320, SELS)])
def test_gemm_split_precision(built_lib, dt, M, N, K, sels):
    g = torch.Generator().manual_seed(77)
    a32 = torch.randn((M, K), generator=g)
    w32 = torch.randn((N, K), generator=g) * K ** -0.5
    bias = torch.randn(N)
    a_hi, a_lo = ops.split_planes(a32, dt)
    wp2 = ops.pack_linear_weight(w32, dt, split=True).to(DEV)
    ref_w2 = a_hi.double() @ w32.double().t() + bias.double()   # weights exact, activation = its high plane
    ref_x3 = a32.double() @ w32.double().t() + bias.double()    # both exact
    for sel in sels:
        f32, _ = ops.gemm(a_hi.to(DEV), wp2, bias=bias.to(DEV), want_f32=True, split="w2", kernel_sel=sel)
        assert_close(f32, ref_w2, split_tol(dt), f"w2 sel={sel}")
        f32, hi, lo = ops.gemm(a_hi.to(DEV), wp2, bias=bias.to(DEV), want_f32=True, want_lo=True, split="x3", a_lo=a_lo.to(DEV), kernel_sel=sel)
        assert_close(f32, ref_x3, split_tol(dt), f"x3 sel={sel}")
        assert_close(hi.float().double() + lo.float().double(), ref_x3, split_tol(dt), f"x3 hi+lo output planes sel={sel}")
    # the single-plane product on the same operands is orders of magnitude coarser: the test would notice a kernel that ignored the
    # low planes
    err_x3 = float((f32.double().cpu() - ref_x3).abs().max())
    single, _ = ops.gemm(a_hi.to(DEV), ops.pack_linear_weight(w32, dt).to(DEV), bias=bias.to(DEV), want_f32=True)
    err_single = float((single.double().cpu() - ref_x3).abs().max())
    assert err_single > 8 * err_x3, (err_single, err_x3)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,H,W,Ci,Co,sels", [(1, 9, 7, 96, 64, [1]), (2, 32, 40, 64, 256, SELS), (1, 40, 40, 128, 128, SELS)])
def test_conv_split_precision(built_lib, dt, B, H, W, Ci, Co, sels):
    g = torch.Generator().manual_seed(78)
    x32 = torch.randn((B, H, W, Ci), generator=g)
    w32 = torch.randn((Co, Ci, 3, 3), generator=g) * (9 * Ci) ** -0.5
    bias = torch.randn(Co)
    r32 = torch.randn((B, H, W, Co), generator=g)
    x_hi, x_lo = ops.split_planes(x32, dt)
    r_hi, r_lo = ops.split_planes(r32, dt)
    wp2 = ops.pack_conv3x3_weight(w32, dt, split=True).to(DEV)
    ref = F.conv2d(x32.double().permute(0, 3, 1, 2), w32.double(), bias.double(), padding=1).permute(0, 2, 3, 1) + r32.double()
    for sel in sels:
        r = ops.conv3x3(x_hi.to(DEV), wp2, bias=bias.to(DEV), split="x3", x_lo=x_lo.to(DEV), res_lp=r_hi.to(DEV), res_lp_lo=r_lo.to(DEV),
                        want_lo=True, want_relu=True, kernel_sel=sel)
        assert_close(r["out"].float().double() + r["out_lo"].float().double(), ref, split_tol(dt), f"conv x3 sel={sel}")
        assert_close(r["relu"].float().double() + r["relu_lo"].float().double(), F.relu(ref), split_tol(dt), f"conv x3 relu planes sel={sel}")


@pytest.mark.parametrize("dt", DTYPES)
def test_convT_split_precision(built_lib, dt):
    B, h, w, Ci, Co, s = 2, 32, 32, 128, 192, 2
    g = torch.Generator().manual_seed(79)
    x32 = torch.randn((B, h, w, Ci), generator=g)
    wt = torch.randn((Ci, Co, s, s), generator=g) * Ci ** -0.5
    bias = torch.randn(Co)
    x_hi, x_lo = ops.split_planes(x32, dt)
    wp, bt = ops.pack_convT_weight(wt, bias, dt, split=True)
    ref = F.conv_transpose2d(x32.double().permute(0, 3, 1, 2), wt.double(), bias.double(), stride=s).permute(0, 2, 3, 1)
    for sel in SELS:
        hi, lo = ops.convT(x_hi.to(DEV), wp.to(DEV), bt.to(DEV), s, Co, split="x3", x_lo=x_lo.to(DEV), want_lo=True, kernel_sel=sel)
        assert_close(hi.float().double() + lo.float().double(), ref, split_tol(dt), f"convT x3 sel={sel}")


def test_gemm256_many_launches_are_deterministic(built_lib):
    """Race screen for the counted-vmcnt / barrier schedule: the same launch repeated must give bit-identical results (an LDS-DMA
    tile read before it landed shows up as run-to-run differences long before it shows up against a tolerance)."""
    dt = torch.bfloat16
    M, N, K = 8192, 1024, 1024
    a, w = rnd((M, K), dt, 5).to(DEV), ops.pack_linear_weight(rnd((N, K), dt, 6, K ** -0.5).float(), dt).to(DEV)
    for sel in (2, 3):
        first = None
        for _ in range(20):
            f32, _ = ops.gemm(a, w, want_f32=True, kernel_sel=sel)
            if first is None:
                first = f32.clone()
            else:
                assert torch.equal(first, f32), f"kernel_sel={sel} is not deterministic"
