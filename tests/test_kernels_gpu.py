"""Per-kernel parity on a real MI355X, through the C ABI.  Reference = plain torch fp32/fp64 on CPU applied to the SAME
16-bit-rounded operands, so the only differences are fp32 accumulation order and the final rounding of lowp outputs.
Tolerances: fp32 outputs 2e-4 of the output scale; lowp outputs 2^-9 (fp16) / 2^-6 (bf16) relative to the scale.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from fast3r_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [torch.float16, torch.bfloat16]


def lp_tol(dt):
    return 2.0 ** -9 if dt == torch.float16 else 2.0 ** -6


def rnd(shape, dt, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dt)


def assert_close(got, ref, tol, what=""):
    got, ref = got.detach().double().cpu(), ref.double()
    scale = float(ref.abs().max().clamp_min(1e-6))
    err = float((got - ref).abs().max())
    assert math.isfinite(err) and err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol:.1e})"


# ------------------------------------------------------------------------------------------------ elementwise
@pytest.mark.parametrize("dt", DTYPES)
def test_cast_and_patchify_exact(built_lib, dt):
    x = torch.randn(3, 1000 * 8)
    assert torch.equal(ops.cast_lp(x.to(DEV), dt).cpu(), x.to(dt))
    img = torch.rand(2, 3, 48, 80) * 2 - 1
    got = ops.patchify(img.to(DEV), 16, dt).cpu()
    ref = F.unfold(img, kernel_size=16, stride=16).transpose(1, 2).reshape(-1, 768).to(dt)  # (c,dy,dx) order = Conv2d weight order
    assert torch.equal(got, ref)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("rows,D,eps", [(7, 1024, 1e-6), (130, 128, 1e-5), (5, 1280, 1e-6), (3, 4096, 1e-5)])
def test_layernorm(built_lib, dt, rows, D, eps):
    x = torch.randn(rows, D) * 3 + 0.5
    g, b = torch.randn(D), torch.randn(D)
    lp, f32 = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), eps, dt, want_f32=True)
    ref = F.layer_norm(x.double(), (D,), g.double(), b.double(), eps)
    assert_close(f32, ref, 2e-5, "ln f32")
    assert_close(lp.float(), ref, lp_tol(dt), "ln lowp")
    rms, _ = ops.layernorm(x.to(DEV), g.to(DEV), None, eps, dt, rms=True)
    ref_rms = x.double() * torch.rsqrt(x.double().pow(2).mean(-1, keepdim=True) + eps) * g.double()
    assert_close(rms.float(), ref_rms, lp_tol(dt), "rmsnorm")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,h,w,C,crop", [(2, 16, 16, 256, None), (1, 4, 5, 64, (7, 10)), (3, 7, 10, 128, None), (1, 1, 1, 8, None)])
def test_upsample2x(built_lib, dt, B, h, w, C, crop):
    x = rnd((B, h, w, C), dt, 1)
    got = ops.upsample2x(x.to(DEV), crop)
    ref = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    if crop:
        ref = ref[:, :crop[0], :crop[1]]
    assert got.shape == ref.shape
    assert_close(got.float(), ref, lp_tol(dt), "upsample")


@pytest.mark.parametrize("dt", DTYPES)
def test_dpt_final(built_lib, dt):
    x = rnd((2, 9, 13, 128), dt, 2)
    w, b = torch.randn(4, 128) * 0.1, torch.randn(4) * 0.1
    pts, conf = ops.dpt_final(x.to(DEV), w.to(DEV), b.to(DEV), ["exp", 1, float("inf")])
    y = x.double() @ w.double().t() + b.double()
    d = y[..., :3].norm(dim=-1, keepdim=True)
    ref_p = y[..., :3] / d.clip(min=1e-8) * torch.expm1(d)
    assert_close(pts, ref_p, 1e-5, "pts3d")
    assert_close(conf, 1 + y[..., 3].exp(), 1e-5, "conf")
    assert (conf >= 1).all()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cin,pair", [(128, True), (64, False), (64, True), (256, False)])
def test_dpt_final_planes_and_widths(built_lib, dt, cin, pair):
    """Cin == 128 runs the LDS-DMA staged kernel (with / without the low plane of a split-precision activation), every other width the
    one-pixel-per-lane kernel; 1000 pixels = 15 full groups of 64 + a ragged one."""
    x32 = torch.randn((1, 25, 40, cin), generator=torch.Generator().manual_seed(5))
    hi = x32.to(dt)
    lo = (x32 - hi.float()).to(dt) if pair else None
    w, b = torch.randn(4, cin) * 0.1, torch.randn(4) * 0.1
    pts, conf = ops.dpt_final(hi.to(DEV), w.to(DEV), b.to(DEV), ["exp", 1, float("inf")], x_lo=None if lo is None else lo.to(DEV))
    xin = hi.double() + (lo.double() if pair else 0.0)
    y = xin @ w.double().t() + b.double()
    d = y[..., :3].norm(dim=-1, keepdim=True)
    assert_close(pts, y[..., :3] / d.clip(min=1e-8) * torch.expm1(d), 1e-5, "pts3d")
    assert_close(conf, 1 + y[..., 3].exp(), 1e-5, "conf")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("depth_mode,conf_mode", [("linear", ("sigmoid", 0.5, 3.0)), ("square", ("exp", 1, 20.0)), ("exp", None), ("square", ("sigmoid", 0.0, 1.0))])
def test_dpt_final_other_modes(built_lib, dt, depth_mode, conf_mode):
    """heads/postprocess.py:27-64: depth 'linear' / 'square' / 'exp', conf 'exp' with a finite vmax (clip) / 'sigmoid' / no confidence channel
    at all (3-channel head[4]: the kernel must not read a 4th weight row)."""
    inf = float("inf")
    x = rnd((2, 9, 13, 128), dt, 2)
    n_out = 4 if conf_mode is not None else 3
    w, b = torch.randn(n_out, 128) * 0.1, torch.randn(n_out) * 0.1
    if conf_mode is not None and conf_mode[0] == "exp":
        w[3] *= 5.0  # make exp(c) cross vmax - vmin for some pixels
    pts, conf = ops.dpt_final(x.to(DEV), w.to(DEV), b.to(DEV), conf_mode, depth_mode=(depth_mode, -inf, inf))
    y = x.double() @ w.double().t() + b.double()
    xyz = y[..., :3]
    d = xyz.norm(dim=-1, keepdim=True)
    ref_p = xyz if depth_mode == "linear" else xyz / d.clip(min=1e-8) * (d.square() if depth_mode == "square" else torch.expm1(d))
    assert_close(pts, ref_p, 1e-5, "pts3d " + depth_mode)
    if conf_mode is None:
        assert conf is None
    else:
        cm, vmin, vmax = conf_mode
        ref_c = vmin + y[..., 3].exp().clip(max=vmax - vmin) if cm == "exp" else (vmax - vmin) * torch.sigmoid(y[..., 3]) + vmin
        assert_close(conf, ref_c, 1e-5, "conf " + cm)
        if cm == "exp":
            assert float(conf.max()) <= vmax + 1e-5 and float((conf >= vmax - 1e-4).float().mean()) > 0.0  # the clip is exercised


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 96, 1024), (77, 260, 200), (1024, 512, 768), (2048, 1024, 4096)])
def test_gemm_bias_f32_and_lowp(built_lib, dt, M, N, K):
    a, w, bias = rnd((M, K), dt, 3), rnd((N, K), dt, 4, K ** -0.5), torch.randn(N)
    wp = ops.pack_linear_weight(w.float(), dt).to(DEV)
    f32, lp = ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), want_f32=True, want_lp=True)
    ref = a.double() @ w.double().t() + bias.double()
    assert_close(f32, ref, 2e-5, "gemm f32")
    assert_close(lp.float(), ref, lp_tol(dt), "gemm lowp")


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_epilogues(built_lib, dt):
    M, N, K, P = 200, 192, 128, 50
    a, w, bias = rnd((M, K), dt, 5), rnd((N, K), dt, 6, K ** -0.5), torch.randn(N)
    wp = ops.pack_linear_weight(w.float(), dt).to(DEV)
    base = a.double() @ w.double().t() + bias.double()
    # GELU (erf), lowp out
    _, y = ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), act="gelu", want_lp=True)
    assert_close(y.float(), F.gelu(base), lp_tol(dt), "gelu")
    # image-id row add + fp32 residual, in place on the residual buffer
    rowadd = torch.randn(M // P, N)
    x = torch.randn(M, N)
    xg = x.clone().to(DEV)
    ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), rowadd=rowadd.to(DEV), rowadd_div=P, res_f32=xg, out_f32=xg)
    assert_close(xg, base + rowadd.double().repeat_interleave(P, 0) + x.double(), 2e-5, "rowadd+res in place")
    # relu + two lowp residuals
    r1, r2 = rnd((M, N), dt, 7), rnd((M, N), dt, 8)
    _, y = ops.gemm(a.to(DEV), wp, bias=bias.to(DEV), act="relu", res_lp=r1.to(DEV), res_lp2=r2.to(DEV), want_lp=True)
    assert_close(y.float(), F.relu(base) + r1.double() + r2.double(), lp_tol(dt), "relu+2res")
    # strided A (lda > K) and K tail (K=96 -> Kpad 128)
    abig = rnd((M, 160), dt, 9)
    w96 = rnd((N, 96), dt, 10, 0.1)
    f32, _ = ops.gemm(abig.to(DEV)[:, :96], ops.pack_linear_weight(w96.float(), dt).to(DEV), K=96, want_f32=True)
    assert_close(f32, abig[:, :96].double() @ w96.double().t(), 2e-5, "strided A, K tail")


def _rope_ref(t, pos_y, pos_x, cos, sin):
    """pos_embed.py:162-183 on (M, H, 64) fp64."""
    def r1(h, p):
        c, s = cos[p][:, None, :].double(), sin[p][:, None, :].double()
        a, b = h[..., :16], h[..., 16:]
        return torch.cat([a * c - b * s, b * c + a * s], -1)
    return torch.cat([r1(t[..., :32], pos_y), r1(t[..., 32:], pos_x)], -1)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n_seq,gh,gw,D,use_rope", [(3, 8, 8, 128, True), (2, 7, 10, 256, True), (1, 1, 300, 128, False), (1, 1, 70, 128, False)])
def test_gemm_qkv_epilogue(built_lib, dt, n_seq, gh, gw, D, use_rope):
    S = gh * gw
    M, K = n_seq * S, 128
    a, w, bias = rnd((M, K), dt, 11), rnd((3 * D, K), dt, 12, K ** -0.5), torch.randn(3 * D) * 0.1
    wp = ops.pack_linear_weight(w.float(), dt).to(DEV)
    q = torch.empty((M, D), dtype=dt, device=DEV)
    k = torch.empty((M, D), dtype=dt, device=DEV)
    ld = ops.vt_ld(S)
    vt = torch.zeros((n_seq, D, ld), dtype=dt, device=DEV)
    rope = None
    if use_rope:
        cos, sin = ops.rope_tables(max(gh, gw), 100.0, DEV)
        rope = (cos, sin, gw)
    ops.gemm_qkv(a.to(DEV), wp, bias.to(DEV), q, k, vt, S, rope)
    ref = a.double() @ w.double().t() + bias.double()
    rq, rk, rv = ref[:, :D], ref[:, D:2 * D], ref[:, 2 * D:]
    if use_rope:
        p = torch.arange(M) % S
        py, px = p // gw, p % gw
        rq = _rope_ref(rq.reshape(M, D // 64, 64), py, px, cos.cpu(), sin.cpu()).reshape(M, D)
        rk = _rope_ref(rk.reshape(M, D // 64, 64), py, px, cos.cpu(), sin.cpu()).reshape(M, D)
    assert_close(q.float(), rq, lp_tol(dt), "q")
    assert_close(k.float(), rk, lp_tol(dt), "k")
    got_v = vt[:, :, :S].float().cpu().permute(0, 2, 1).reshape(M, D)
    assert_close(got_v, rv, lp_tol(dt), "v^T")
    assert float(vt[:, :, S:].abs().sum()) == 0.0  # padding untouched


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("rows_per_group,n_groups,D,sel", [(24, 5, 128, 0), (1, 5, 128, 0), (24, 30, 256, 2), (1, 30, 256, 2), (24, 30, 256, 1)])
def test_gemm_qkv_rope_by_row_group(built_lib, dt, rows_per_group, n_groups, D, sel):
    """rope_mode 1 (LlamaDecoder): one angle set per row group, complex pairs (2j, 2j+1) of the reference (llama.py:96-122) realised
    as a row permutation of the q / k weights + the half-split rotation of the epilogue.  Scores q . k must be those of the reference."""
    from fast3r_amd.fast3r import _ROPE_PERM
    K = 128  # (D = 256, sel = 2: the 256-tile kernel, interior and edge sub-tiles of M = 720)
    M = n_groups * 24
    a = rnd((M, K), dt, 21)
    wq, wk, wv = (rnd((D, K), dt, 22 + i, K ** -0.5) for i in range(3))
    perm = torch.tensor([h * 64 + d for h in range(D // 64) for d in _ROPE_PERM])
    wp = ops.pack_linear_weight(torch.cat([wq.float()[perm], wk.float()[perm], wv.float()]), dt).to(DEV)
    n_tab = M // rows_per_group
    ang = torch.rand(n_tab, 32, generator=torch.Generator().manual_seed(5)) * 6.0
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    q = torch.empty((M, D), dtype=dt, device=DEV)
    k = torch.empty((M, D), dtype=dt, device=DEV)
    vt = torch.zeros((1, D, ops.vt_ld(M)), dtype=dt, device=DEV)
    ops.gemm_qkv(a.to(DEV), wp, None, q, k, vt, M, (cos.to(DEV), sin.to(DEV), rows_per_group), rope_mode=1, kernel_sel=sel)

    def ref_rot(w):  # reference arithmetic in fp64: view_as_complex on adjacent pairs, times cis(angle of the row's group)
        x = (a.double() @ w.double().t()).reshape(M, D // 64, 32, 2)
        g = torch.arange(M) // rows_per_group
        c, s_ = cos.double()[g][:, None, :], sin.double()[g][:, None, :]
        return torch.stack([x[..., 0] * c - x[..., 1] * s_, x[..., 0] * s_ + x[..., 1] * c], dim=-1).reshape(M, D)
    rq, rk = ref_rot(wq), ref_rot(wk)
    assert_close(q.float().cpu()[:, torch.argsort(perm)], rq, lp_tol(dt), "q (un-permuted)")
    assert_close(k.float().cpu()[:, torch.argsort(perm)], rk, lp_tol(dt), "k (un-permuted)")
    nh = D // 64
    sc = (q.float().cpu().reshape(M, nh, 64).transpose(0, 1) @ k.float().cpu().reshape(M, nh, 64).transpose(0, 1).transpose(1, 2))
    sc_ref = rq.reshape(M, nh, 64).transpose(0, 1) @ rk.reshape(M, nh, 64).transpose(0, 1).transpose(1, 2)
    assert_close(sc, sc_ref, 4 * lp_tol(dt), "scores are permutation invariant")
    assert_close(vt[0, :, :M].float().cpu().t(), a.double() @ wv.double().t(), lp_tol(dt), "v^T")


@pytest.mark.parametrize("dt", DTYPES)
def test_silu_mul_and_rows_add_and_rmsnorm(built_lib, dt):
    rows, hidden = 37, 192
    ab = rnd((rows, 2 * hidden), dt, 31, 2.0)
    out = ops.silu_mul(ab.to(DEV), hidden)
    ref = F.silu(ab[:, :hidden].double()) * ab[:, hidden:].double()
    assert_close(out.float(), ref, lp_tol(dt), "silu_mul")
    x = torch.randn(50, 128)
    vec = torch.randn(128)
    xg = x.clone().to(DEV)
    ops.rows_add(xg, vec.to(DEV), 13)
    exp = x.clone()
    exp[:13] += vec
    assert torch.equal(xg.cpu(), exp)
    w = 1 + 0.1 * torch.randn(128)
    y, _ = ops.layernorm(xg, w.to(DEV), None, 1e-5, dt, rms=True)
    refn = exp.double() * torch.rsqrt(exp.double().pow(2).mean(-1, keepdim=True) + 1e-5) * w.double()
    assert_close(y.float(), refn, lp_tol(dt), "rmsnorm")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,H,W,Ci,Co,stride", [(2, 8, 8, 256, 256, 1), (1, 16, 12, 96, 256, 1), (2, 9, 7, 768, 128, 2), (1, 32, 32, 128, 128, 1), (3, 5, 5, 192, 64, 2)])
def test_conv3x3(built_lib, dt, B, H, W, Ci, Co, stride):
    x = rnd((B, H, W, Ci), dt, 13)
    w = rnd((Co, Ci, 3, 3), dt, 14, (9 * Ci) ** -0.5)
    bias = torch.randn(Co)
    wp = ops.pack_conv3x3_weight(w.float(), dt).to(DEV)
    y = ops.conv3x3(x.to(DEV), wp, stride=stride, bias=bias.to(DEV))
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), stride=stride, padding=1).permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    assert_close(y.float(), ref, lp_tol(dt), "conv3x3")


@pytest.mark.parametrize("dt", DTYPES)
def test_conv3x3_rcu_fusions(built_lib, dt):
    """pre-ReLU on the operand + bias + two residual adds = ResidualConvUnit_custom inside a fusion block."""
    B, H, W, C = 2, 10, 6, 256
    x, extra = rnd((B, H, W, C), dt, 15), rnd((B, H, W, C), dt, 16)
    w = rnd((C, C, 3, 3), dt, 17, (9 * C) ** -0.5)
    bias = torch.randn(C) * 0.1
    wp = ops.pack_conv3x3_weight(w.float(), dt).to(DEV)
    y = ops.conv3x3(x.to(DEV), wp, bias=bias.to(DEV), a_relu=True, res_lp=x.to(DEV), res_lp2=extra.to(DEV))
    ref = F.conv2d(F.relu(x.double()).permute(0, 3, 1, 2), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1) + x.double() + extra.double()
    assert_close(y.float(), ref, lp_tol(dt), "rcu conv")
    y = ops.conv3x3(x.to(DEV), wp, bias=bias.to(DEV), act="relu")
    ref = F.relu(F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), padding=1)).permute(0, 2, 3, 1)
    assert_close(y.float(), ref, lp_tol(dt), "conv+relu")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,h,w,Ci,Co,s", [(2, 4, 4, 96, 96, 4), (1, 7, 10, 192, 192, 2), (3, 2, 3, 96, 96, 4)])
def test_conv_transpose(built_lib, dt, B, h, w, Ci, Co, s):
    x = rnd((B, h, w, Ci), dt, 18)
    wt = rnd((Ci, Co, s, s), dt, 19, Ci ** -0.5)
    bias = torch.randn(Co)
    wp, bt = ops.pack_convT_weight(wt.float(), bias, dt)
    y = ops.convT(x.to(DEV), wp.to(DEV), bt.to(DEV), s, Co)
    ref = F.conv_transpose2d(x.double().permute(0, 3, 1, 2), wt.double(), bias.double(), stride=s).permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    assert_close(y.float(), ref, lp_tol(dt), "convT")


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, H, scale):
    """q (Tq, H*64), k/v (Tk, H*64) lowp -> fp64 softmax(q k^T scale) v."""
    Tq, Tk = q.shape[0], k.shape[0]
    qh, kh, vh = (t.double().reshape(-1, H, 64).transpose(0, 1) for t in (q, k, v))
    a = ((qh @ kh.transpose(1, 2)) * scale).softmax(-1)
    return (a @ vh).transpose(0, 1).reshape(Tq, H * 64)


def _vt_of(v, H):
    """(T, H*64) -> zero padded V^T [H*64][ld]"""
    T = v.shape[0]
    vt = torch.zeros((H * 64, ops.vt_ld(T)), dtype=v.dtype)
    vt[:, :T] = v.t()
    return vt


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("T,H,scale", [(16, 2, 0.125), (64, 2, 0.160192), (100, 1, 0.125), (256, 2, 0.125), (1000, 2, 0.160192), (3072, 1, 0.125)])
def test_attention_single_segment(built_lib, dt, T, H, scale):
    q, k, v = rnd((T, H * 64), dt, 20), rnd((T, H * 64), dt, 21), rnd((T, H * 64), dt, 22)
    o = torch.empty((T, H * 64), dtype=dt, device=DEV)
    ops.attention(q.to(DEV), o, H, scale, [(k.to(DEV), _vt_of(v, H).to(DEV), T, 0, 0)])
    assert_close(o.float(), _attn_ref(q, k, v, H, scale), 2 * lp_tol(dt), f"attn T={T}")


@pytest.mark.parametrize("dt", DTYPES)
def test_attention_segments_equal_concatenation(built_lib, dt):
    """Uneven K/V segments (the view-sharded multi-GPU layout, incl. a padded and an empty one) == one segment."""
    H, Tq = 2, 200
    lens = [130, 0, 64, 257]
    q = rnd((Tq, H * 64), dt, 23)
    ks = [rnd((max(n, 1), H * 64), dt, 30 + i) for i, n in enumerate(lens)]
    vs = [rnd((max(n, 1), H * 64), dt, 40 + i) for i, n in enumerate(lens)]
    segs = []
    for kk, vv, n in zip(ks, vs, lens):
        pad_k = torch.cat([kk[:n], torch.full((7, H * 64), float("nan"), dtype=dt)])  # rows past seg_len must never be read as keys
        segs.append((pad_k.to(DEV), _vt_of(vv[:n], H).to(DEV) if n else _vt_of(vv[:1] * 0, H).to(DEV), n, 0, 0))
    o = torch.empty((Tq, H * 64), dtype=dt, device=DEV)
    ops.attention(q.to(DEV), o, H, 0.125, segs)
    kcat = torch.cat([kk[:n] for kk, n in zip(ks, lens)])
    vcat = torch.cat([vv[:n] for vv, n in zip(vs, lens)])
    assert_close(o.float(), _attn_ref(q, kcat, vcat, H, 0.125), 2 * lp_tol(dt), "segments")


@pytest.mark.parametrize("dt", DTYPES)
def test_attention_batched_sequences(built_lib, dt):
    """Encoder layout: batch of independent sequences, strided batch access."""
    nb, S, H = 3, 80, 2
    D = H * 64
    q, k, v = rnd((nb * S, D), dt, 50), rnd((nb * S, D), dt, 51), rnd((nb * S, D), dt, 52)
    ld = ops.vt_ld(S)
    vt = torch.zeros((nb, D, ld), dtype=dt)
    for b in range(nb):
        vt[b, :, :S] = v[b * S:(b + 1) * S].t()
    o = torch.empty((nb * S, D), dtype=dt, device=DEV)
    ops.attention(q.to(DEV), o, H, 0.125, [(k.to(DEV), vt.to(DEV), S, S * D, D * ld)], tq=S, batch=nb, q_batch_stride=S * D, o_batch_stride=S * D)
    ref = torch.cat([_attn_ref(q[b * S:(b + 1) * S], k[b * S:(b + 1) * S], v[b * S:(b + 1) * S], H, 0.125) for b in range(nb)])
    assert_close(o.float(), ref, 2 * lp_tol(dt), "batched attn")


@pytest.mark.parametrize("dt", DTYPES)
def test_attention_online_softmax_rescale_is_forced(built_lib, dt):
    """A key far above the rest, placed in a LATE tile: the running max must jump and every earlier contribution be
    rescaled (the rare data-dependent branch gets its own test)."""
    T, H = 640, 1
    q, k, v = rnd((T, 64), dt, 60), rnd((T, 64), dt, 61, 0.3), rnd((T, 64), dt, 62)
    k[500] = (q[7].float() * 3.0).to(dt)  # q7 . k500 >> everything else, in tile 7
    k[130] = (q[300].float() * 2.0).to(dt)
    o = torch.empty((T, 64), dtype=dt, device=DEV)
    ops.attention(q.to(DEV), o, H, 0.125, [(k.to(DEV), _vt_of(v, H).to(DEV), T, 0, 0)])
    assert_close(o.float(), _attn_ref(q, k, v, H, 0.125), 2 * lp_tol(dt), "forced rescale")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("pattern", ["drift_up", "stairs", "outlier_first", "all_equal"])
def test_attention_lazy_reference_patterns(built_lib, dt, pattern):
    """The product kernel keeps a LAZY softmax reference (no per-tile max; it re-bases when a tile's row sum says P has
    grown): score sequences built to sit on that trigger.  Keys are multiples of one direction u, queries too, so the
    score of (query i, key j) is a_i * b_j * |u|^2 * scale and each pattern is a choice of b over the 10 key tiles."""
    T, H, scale = 640, 1, 0.125
    g = torch.Generator().manual_seed(90)
    u = torch.randn(64, generator=g)
    u = u / u.norm()
    a = torch.linspace(0.5, 8.0, T)  # per-query gain (incl. rows whose logits span > 60 in exp2 units)
    tile = torch.arange(T) // 64
    if pattern == "drift_up":      # every tile a little above the previous one: many small re-bases / rows riding below the trigger
        b = 0.35 * tile.float() + 0.01 * torch.randn(T, generator=g)
    elif pattern == "stairs":      # long plateaus, two big jumps (one inside a tile)
        b = torch.where(torch.arange(T) < 200, 0.0, torch.where(torch.arange(T) < 500, 6.0, 15.0)) + 0.05 * torch.randn(T, generator=g)
    elif pattern == "outlier_first":  # the maximum is key 3; everything later is far below it (P underflows to 0 in both)
        b = -4.0 + 0.5 * torch.randn(T, generator=g)
        b[3] = 12.0
    else:                          # identical scores: every P equals 1 against an exact reference, row sum 32 per lane and tile
        b = torch.full((T,), 1.5)
    q = (a[:, None] * u[None, :] * 8.0).to(dt)
    k = (b[:, None] * u[None, :]).to(dt)
    v = rnd((T, 64), dt, 91)
    o = torch.empty((T, 64), dtype=dt, device=DEV)
    ops.attention(q.to(DEV), o, H, scale, [(k.to(DEV), _vt_of(v, H).to(DEV), T, 0, 0)])
    assert torch.isfinite(o.float()).all()
    assert_close(o.float(), _attn_ref(q, k, v, H, scale), 2 * lp_tol(dt), f"lazy reference {pattern}")


@pytest.mark.parametrize("dt", DTYPES)
def test_attention_state_carried_across_launches(built_lib, dt):
    """Launch 1 over the local segment parks (m, l, O); launch 2 resumes over the remote segments: must equal ONE launch
    over [local, remote...] bit for bit (same tiles in the same order, state kept in fp32), and the fp64 reference."""
    H, Tq = 2, 300
    lens = [192, 100, 257]
    q = rnd((Tq, H * 64), dt, 70)
    ks = [rnd((n, H * 64), dt, 71 + i) for i, n in enumerate(lens)]
    vs = [rnd((n, H * 64), dt, 81 + i) for i, n in enumerate(lens)]
    segs = [(kk.to(DEV), _vt_of(vv, H).to(DEV), n, 0, 0) for kk, vv, n in zip(ks, vs, lens)]
    one = torch.empty((Tq, H * 64), dtype=dt, device=DEV)
    ops.attention(q.to(DEV), one, H, 0.125, segs)
    two = torch.empty((Tq, H * 64), dtype=dt, device=DEV)
    state = ops.attention_state(Tq, H, DEV)
    two.fill_(float("nan"))
    ops.attention(q.to(DEV), two, H, 0.125, segs[:1], state=state, state_out=True)
    assert torch.isnan(two.float()).all()  # the first launch must not write the output
    ops.attention(q.to(DEV), two, H, 0.125, segs[1:], state=state, state_in=True)
    assert torch.equal(one, two)
    assert_close(two.float(), _attn_ref(q, torch.cat(ks), torch.cat(vs), H, 0.125), 2 * lp_tol(dt), "state carry")


# ------------------------------------------------------------------------------------------------ head_dim != 64 (f3r_attn_generic.hip)
def _attn_ref_hd(q, k, v, H, Hkv, hd, scale):
    Tq = q.shape[0]
    qh = q.double().reshape(Tq, H, hd).transpose(0, 1)
    kh = k.double().reshape(-1, Hkv, hd).transpose(0, 1).repeat_interleave(H // Hkv, 0)
    vh = v.double().reshape(-1, Hkv, hd).transpose(0, 1).repeat_interleave(H // Hkv, 0)
    return (((qh @ kh.transpose(1, 2)) * scale).softmax(-1) @ vh).transpose(0, 1).reshape(Tq, H * hd)


def _vt_hd(v, ld=None):
    T = v.shape[0]
    vt = torch.zeros((v.shape[1], ops.vt_ld(T) if ld is None else ld), dtype=v.dtype)
    vt[:, :T] = v.t()
    return vt


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("hd,T,H,Hkv", [(80, 200, 2, 2), (80, 1024, 3, 3), (128, 333, 2, 1), (32, 130, 4, 4), (96, 64, 1, 1), (16, 77, 2, 2)])
def test_attention_generic_head_dim(built_lib, dt, hd, T, H, Hkv):
    """The reference's Attention takes any dim // num_heads (blocks.py:113-143; model_scaling_huge.yaml: 80): the generic kernel vs fp64,
    incl. partial key tiles, ragged query blocks and grouped-query heads."""
    scale = hd ** -0.5
    q, k, v = rnd((T, H * hd), dt, 120), rnd((T, Hkv * hd), dt, 121), rnd((T, Hkv * hd), dt, 122)
    o = torch.full((T, H * hd), float("nan"), dtype=dt, device=DEV)
    ops.attention(q.to(DEV), o, H, scale, [(k.to(DEV), _vt_hd(v).to(DEV), T, 0, 0)], head_dim=hd, kv_group=H // Hkv)
    assert_close(o.float(), _attn_ref_hd(q, k, v, H, Hkv, hd, scale), 2 * lp_tol(dt), f"attn head_dim {hd}")


@pytest.mark.parametrize("dt", DTYPES)
def test_attention_generic_head_dim_segments_state_batch(built_lib, dt):
    """head_dim 80: K/V segments (incl. an empty one) == their concatenation; two launches carrying (m, l, O) == one; batched sequences."""
    hd, H, Tq = 80, 2, 300
    lens = [130, 0, 64, 257]
    scale = hd ** -0.5
    q = rnd((Tq, H * hd), dt, 130)
    ks = [rnd((max(n, 1), H * hd), dt, 131 + i) for i, n in enumerate(lens)]
    vs = [rnd((max(n, 1), H * hd), dt, 141 + i) for i, n in enumerate(lens)]
    segs = [(kk.to(DEV), _vt_hd(vv[:max(n, 1)] * (1 if n else 0)).to(DEV), n, 0, 0) for kk, vv, n in zip(ks, vs, lens)]
    kcat = torch.cat([kk[:n] for kk, n in zip(ks, lens)])
    vcat = torch.cat([vv[:n] for vv, n in zip(vs, lens)])
    ref = _attn_ref_hd(q, kcat, vcat, H, H, hd, scale)
    one = torch.empty((Tq, H * hd), dtype=dt, device=DEV)
    ops.attention(q.to(DEV), one, H, scale, segs, head_dim=hd)
    assert_close(one.float(), ref, 2 * lp_tol(dt), "segments")
    two = torch.full((Tq, H * hd), float("nan"), dtype=dt, device=DEV)
    state = ops.attention_state(Tq, H, DEV, head_dim=hd)
    ops.attention(q.to(DEV), two, H, scale, segs[:1], state=state, state_out=True, head_dim=hd)
    assert torch.isnan(two.float()).all()
    ops.attention(q.to(DEV), two, H, scale, segs[1:], state=state, state_in=True, head_dim=hd)
    assert torch.equal(one, two)
    nb, S = 3, 90
    D = H * hd
    qb, kb, vb = rnd((nb * S, D), dt, 150), rnd((nb * S, D), dt, 151), rnd((nb * S, D), dt, 152)
    ld = ops.vt_ld(S)
    vt = torch.zeros((nb, D, ld), dtype=dt)
    for b in range(nb):
        vt[b, :, :S] = vb[b * S:(b + 1) * S].t()
    o = torch.empty((nb * S, D), dtype=dt, device=DEV)
    ops.attention(qb.to(DEV), o, H, scale, [(kb.to(DEV), vt.to(DEV), S, S * D, D * ld)], tq=S, batch=nb, q_batch_stride=S * D, o_batch_stride=S * D,
                  head_dim=hd)
    refb = torch.cat([_attn_ref_hd(qb[b * S:(b + 1) * S], kb[b * S:(b + 1) * S], vb[b * S:(b + 1) * S], H, H, hd, scale) for b in range(nb)])
    assert_close(o.float(), refb, 2 * lp_tol(dt), "batched head_dim 80")
    with pytest.raises(ValueError, match="causal"):
        ops.attention(q.to(DEV), one, H, scale, segs, head_dim=hd, causal=True)


# ------------------------------------------------------------------------------------------------ grouped-query / causal attention
def _attn_ref_general(q, k, v, H, Hkv, scale, causal=False, q_pos0=0, k_pos=None):
    """q (Tq, H*64), k / v (Tk, Hkv*64) -> fp64 softmax(q k^T scale [+ causal mask]) v with repeat_kv (llama.py:125-134)."""
    Tq, Tk = q.shape[0], k.shape[0]
    rep = H // Hkv
    qh = q.double().reshape(Tq, H, 64).transpose(0, 1)
    kh = k.double().reshape(Tk, Hkv, 64).transpose(0, 1).repeat_interleave(rep, dim=0)
    vh = v.double().reshape(Tk, Hkv, 64).transpose(0, 1).repeat_interleave(rep, dim=0)
    sc = (qh @ kh.transpose(1, 2)) * scale
    if causal:
        kp = torch.arange(Tk) if k_pos is None else k_pos
        qp = q_pos0 + torch.arange(Tq)
        sc = sc.masked_fill(kp[None, :] > qp[:, None], float("-inf"))
    return (sc.softmax(-1) @ vh).transpose(0, 1).reshape(Tq, H * 64)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("T,H,Hkv", [(200, 4, 2), (640, 4, 1), (130, 6, 2)])
def test_attention_grouped_query(built_lib, dt, T, H, Hkv):
    q, k, v = rnd((T, H * 64), dt, 120), rnd((T, Hkv * 64), dt, 121), rnd((T, Hkv * 64), dt, 122)
    o = torch.empty((T, H * 64), dtype=dt, device=DEV)
    ops.attention(q.to(DEV), o, H, 0.125, [(k.to(DEV), _vt_of(v, Hkv).to(DEV), T, 0, 0)], kv_group=H // Hkv)
    assert_close(o.float(), _attn_ref_general(q, k, v, H, Hkv, 0.125), 2 * lp_tol(dt), "gqa")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("T,H,Hkv", [(64, 2, 2), (200, 2, 2), (700, 4, 2), (1000, 2, 1)])
def test_attention_causal(built_lib, dt, T, H, Hkv):
    """is_causal=True (llama.py:239): every tile left of the diagonal is fully visible, the diagonal tiles are masked per row, the tiles
    to the right are masked whole; ragged last tile included."""
    q, k, v = rnd((T, H * 64), dt, 130), rnd((T, Hkv * 64), dt, 131), rnd((T, Hkv * 64), dt, 132)
    o = torch.empty((T, H * 64), dtype=dt, device=DEV)
    ops.attention(q.to(DEV), o, H, 0.125, [(k.to(DEV), _vt_of(v, Hkv).to(DEV), T, 0, 0)], kv_group=H // Hkv, causal=True)
    assert_close(o.float(), _attn_ref_general(q, k, v, H, Hkv, 0.125, causal=True), 2 * lp_tol(dt), "causal")


@pytest.mark.parametrize("dt", DTYPES)
def test_attention_causal_over_shards(built_lib, dt):
    """The view-sharded layout of causal attention: this rank's queries sit at global positions [300, 500); launch 1 attends over the
    local shard (same positions) and parks the state, launch 2 over the two remote shards [0, 300) (all visible) and [500, 730) (all
    hidden).  Must equal causal attention of those 200 query rows over the whole 730-token sequence."""
    H, lens, pos = 2, [300, 200, 230], [0, 300, 500]
    ks = [rnd((n, H * 64), dt, 140 + i) for i, n in enumerate(lens)]
    vs = [rnd((n, H * 64), dt, 150 + i) for i, n in enumerate(lens)]
    q = rnd((200, H * 64), dt, 160)
    segs = [(kk.to(DEV), _vt_of(vv, H).to(DEV), n, 0, 0) for kk, vv, n in zip(ks, vs, lens)]
    o = torch.empty((200, H * 64), dtype=dt, device=DEV)
    state = ops.attention_state(200, H, DEV)
    ops.attention(q.to(DEV), o, H, 0.125, [segs[1]], state=state, state_out=True, causal=True, q_pos0=300, seg_pos0=[300])
    ops.attention(q.to(DEV), o, H, 0.125, [segs[0], segs[2]], state=state, state_in=True, causal=True, q_pos0=300, seg_pos0=[0, 500])
    ref = _attn_ref_general(q, torch.cat(ks), torch.cat(vs), H, H, 0.125, causal=True, q_pos0=300)
    assert_close(o.float(), ref, 2 * lp_tol(dt), "causal over shards")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,Hq,Hkv,K,sels", [(300, 4, 2, 128, [1]), (2304, 8, 4, 256, [1, 2, 3]), (77, 6, 2, 192, [1])])
def test_gemm_qkv_grouped_query_split(built_lib, dt, M, Hq, Hkv, K, sels):
    """QKV epilogue with unequal parts (f3r_gemm_args.qkv_dq): q -> [M][Hq*64], k -> [M][Hkv*64], v -> V^T [Hkv*64][ld]."""
    Dq, Dkv = Hq * 64, Hkv * 64
    a = rnd((M, K), dt, 170)
    w = rnd((Dq + 2 * Dkv, K), dt, 171, K ** -0.5)
    wp = ops.pack_linear_weight(w.float(), dt).to(DEV)
    ref = a.double() @ w.double().t()
    for sel in sels:
        q = torch.empty((M, Dq), dtype=dt, device=DEV)
        k = torch.empty((M, Dkv), dtype=dt, device=DEV)
        vt = torch.zeros((1, Dkv, ops.vt_ld(M)), dtype=dt, device=DEV)
        ops.gemm_qkv(a.to(DEV), wp, None, q, k, vt, M, None, q_dim=Dq, kernel_sel=sel)
        assert_close(q.float(), ref[:, :Dq], lp_tol(dt), f"q sel={sel}")
        assert_close(k.float(), ref[:, Dq:Dq + Dkv], lp_tol(dt), f"k sel={sel}")
        assert_close(vt[0, :, :M].float().cpu().t(), ref[:, Dq + Dkv:], lp_tol(dt), f"v^T sel={sel}")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n_seq,gh,gw,H", [(2, 16, 16, 2), (3, 7, 10, 4), (1, 1, 300, 1)])
def test_exact_mode_rope_and_fp32_attention(built_lib, dt, n_seq, gh, gw, H):
    """precision "exact": f3r_rope2d_f32 in place on the q / k parts of an fp32 [T][3D] buffer and f3r_attn_f32 (fp32 softmax attention, hi +
    lo output planes) vs fp64; sequences of 256 / 70 (ragged key tile and query block) / 300 tokens."""
    S, D = gh * gw, H * 64
    T = n_seq * S
    g = torch.Generator().manual_seed(7)
    qkv = torch.randn((T, 3 * D), generator=g)
    cos, sin = ops.rope_tables(max(gh, gw), 100.0, DEV)
    buf = qkv.clone().to(DEV)
    ops.rope2d_f32(buf, H, S, (cos, sin, gw))
    p = torch.arange(T) % S
    py, px = p // gw, p % gw
    rq = _rope_ref(qkv[:, :D].double().reshape(T, H, 64), py, px, cos.cpu(), sin.cpu()).reshape(T, D)
    rk = _rope_ref(qkv[:, D:2 * D].double().reshape(T, H, 64), py, px, cos.cpu(), sin.cpu()).reshape(T, D)
    assert_close(buf[:, :D], rq, 2e-6, "rope q")
    assert_close(buf[:, D:2 * D], rk, 2e-6, "rope k")
    assert torch.equal(buf[:, 2 * D:].cpu(), qkv[:, 2 * D:])  # v untouched
    scale = 0.125
    o_hi, o_lo, o32 = ops.attention_f32(buf, H, n_seq, S, scale, dt, want_f32=True)
    q64, k64, v64 = (t.reshape(n_seq, S, H, 64).transpose(1, 2) for t in (rq, rk, qkv[:, 2 * D:].double()))
    ref = (torch.softmax(q64 @ k64.transpose(-1, -2) * scale, dim=-1) @ v64).transpose(1, 2).reshape(T, D)
    assert_close(o32, ref, 3e-6, "fp32 attention")
    # hi + lo planes carry the fp32 result to 2^-21 (fp16) / 2^-16 (bf16)
    assert_close(o_hi.float().cpu().double() + o_lo.float().cpu().double(), ref, 3e-6 if dt == torch.float16 else 3e-5, "hi + lo planes")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H,kv_group,causal,tq,tk,q_pos0", [(4, 2, True, 300, 300, 0), (4, 4, False, 70, 333, 0), (2, 1, True, 130, 520, 260),
                                                            (4, 2, True, 64, 200, 100)])
def test_exact_mode_general_attention_rope_and_swiglu(built_lib, dt, H, kv_group, causal, tq, tk, q_pos0):
    """The general forms of the fp32-equivalent mode (LlamaDecoder and view-sharded models): f3r_attn_f32_ex with grouped-query heads, a causal
    mask on absolute positions and keys that are not the queries' own rows (a rank's queries over the gathered keys of all ranks);
    f3r_rope_f32 in the per-view form of the Llama blocks; f3r_silu_mul_f32 -- all vs fp64."""
    D, Dkv = H * 64, H // kv_group * 64
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn((tq, D + 2 * Dkv), generator=g)
    k_all, v_all = torch.randn((tk, Dkv), generator=g), torch.randn((tk, Dkv), generator=g)
    scale = 0.125
    own = tq == tk and q_pos0 == 0   # the queries' own keys (columns of qkv) or an external K / V
    kv = None if own else (k_all.to(DEV), v_all.to(DEV))
    o_hi, o_lo, o32 = ops.attention_f32(qkv.to(DEV), H, 1, tq, scale, dt, want_f32=True, kv_group=kv_group, causal=causal, kv=kv, q_pos0=q_pos0)
    kk, vv = (qkv[:, D:D + Dkv], qkv[:, D + Dkv:]) if own else (k_all, v_all)
    q64 = qkv[:, :D].double().reshape(tq, H, 64).transpose(0, 1)
    k64 = kk.double().reshape(tk, H // kv_group, 64).transpose(0, 1).repeat_interleave(kv_group, dim=0)
    v64 = vv.double().reshape(tk, H // kv_group, 64).transpose(0, 1).repeat_interleave(kv_group, dim=0)
    sc = q64 @ k64.transpose(-1, -2) * scale
    if causal:
        vis = torch.arange(tk)[None, :] <= (q_pos0 + torch.arange(tq))[:, None]
        sc = sc.masked_fill(~vis[None], float("-inf"))
    ref = (torch.softmax(sc, dim=-1) @ v64).transpose(0, 1).reshape(tq, D)
    assert_close(o32, ref, 3e-6, "fp32 attention (general)")
    assert_close(o_hi.float().cpu().double() + o_lo.float().cpu().double(), ref, 3e-6 if dt == torch.float16 else 3e-5, "hi + lo planes")
    # per-view rotary embedding (rope_mode 1): rows of view v rotate by table row v; dims [0,32) pair (i, i+16) with columns 0-15, [32,64) with 16-31
    n_views, per_view = 5, 12
    T = n_views * per_view
    x = torch.randn((T, D + 2 * Dkv), generator=g)
    ang = torch.rand((n_views, 32), generator=g) * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    buf = x.clone().to(DEV)
    ops.rope_f32(buf, H + H // kv_group, T, (cos.to(DEV), sin.to(DEV), per_view), rope_mode=1)
    xr = x[:, :D + Dkv].double().reshape(T, -1, 2, 2, 16)            # [T][head][half][pair member][i]
    c = cos.double().reshape(n_views, 2, 16).repeat_interleave(per_view, dim=0)[:, None]   # [T][1][half][i]
    s_ = sin.double().reshape(n_views, 2, 16).repeat_interleave(per_view, dim=0)[:, None]
    a_, b_ = xr[:, :, :, 0], xr[:, :, :, 1]
    rot = torch.stack([a_ * c - b_ * s_, b_ * c + a_ * s_], dim=3).reshape(T, D + Dkv)
    assert_close(buf[:, :D + Dkv], rot, 2e-6, "per-view rope")
    assert torch.equal(buf[:, D + Dkv:].cpu(), x[:, D + Dkv:])
    hidden = 96
    ab = torch.randn((37, 2 * hidden), generator=g) * 3
    hi, lo = ops.silu_mul_f32(ab.to(DEV), hidden, dt)
    want = F.silu(ab[:, :hidden].double()) * ab[:, hidden:].double()
    assert_close(hi.float().cpu().double() + lo.float().cpu().double(), want, 3e-6 if dt == torch.float16 else 3e-5, "fp32 swiglu planes")


@pytest.mark.parametrize("dt", DTYPES)
def test_patchify_any_patch_size_and_general_bilinear(built_lib, dt):
    """DINOv2's patch 14: im2col rows zero-padded to a row stride that is a multiple of 8; the head's Interpolate(scale_factor=14/8)."""
    img = torch.rand(2, 3, 28, 42) * 2 - 1
    got = ops.patchify(img.to(DEV), 14, dt, ld_out=640).cpu()
    ref = F.unfold(img, kernel_size=14, stride=14).transpose(1, 2).reshape(-1, 588).to(dt)
    assert got.shape == (2 * 2 * 3, 640) and torch.equal(got[:, :588], ref) and float(got[:, 588:].abs().sum()) == 0.0
    x = rnd((2, 16, 24, 64), dt, 7)
    up = ops.interp_bilinear(x.to(DEV), (28, 42))
    want = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=1.75, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    assert up.shape == want.shape
    assert_close(up.float(), want, lp_tol(dt), "bilinear x1.75")
