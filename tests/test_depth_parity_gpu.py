"""Parity where it can actually fail ("Next round" item 1).

(a) STRESS weights at ViT-L DEPTH against the CPU oracle: `synth_state_dict(..., dist="hot")` (N(0, 1/fan_in): attention logits with
    std ~1.3, softmax far from uniform, noise-amplifying heads) through the full ViT-L / ViT-L / 2-DPT model (24 + 24 blocks) at
    N = 3 and N = 8 views of 512^2 -- the benchmarked format (fp16 / high) must be within 1e-3 rel-L2 of the fp32 oracle on every
    output; fp16 / fast and bf16 / fast are printed beside it (no claim: DESIGN.md section 3 (Precision modes)).
    Reference path: fast3r/models/fast3r.py:302-497 at BASELINE configs (1), (3)-lite.
(b) the same weights at N = 100 (BASELINE config 3's size) and (d) at N = 320 (the benchmarked configuration itself): fp16 / high against
    the on-device fp32-equivalent mode.
(c) every GEMM / conv ROLE of the model at its N = 320 shape (M = 327 680 rows and up) against fp64 on 4096 SAMPLED output elements
    (rows drawn from every tile incl. the first, the last and tile edges), computed from the same rounded operands -- an independent
    witness at M >= 102 400, where until now only the exact mode (which shares these kernels) and one checksum test looked.
    Single-plane operands and both split modes (W2 = weights hi + lo: transformer linears; X3 = both operands: head convolutions).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import rel_l2, views_to
from fast3r_amd import Fast3R, ops
from fast3r_amd.synthetic import make_views, synth_state_dict, vit_large_args
from oracle import fast3r_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-3
_CACHE = {}


def _vitl_hot():
    if "sd" not in _CACHE:
        enc, dec, head = vit_large_args()
        shapes = {k: tuple(v.shape) for k, v in Fast3R(enc, dec, head).state_dict().items()}
        _CACHE["args"] = (enc, dec, head)
        _CACHE["sd"] = synth_state_dict(shapes, 0, dist="hot")
    return _CACHE["args"], _CACHE["sd"]


def _build(dt, precision, low_plane="fp8"):
    (enc, dec, head), sd = _vitl_hot()
    m = Fast3R(enc, dec, head, compute_dtype=dt, precision=precision).eval()
    m.low_plane = low_plane              # "+fp16" variants: EVERY correction product on fp16 planes (the MLPs' A W_lo, and since round 6 the two
    m.head_corrections = low_plane      # correction products of the heads' 3x3 convolutions) -- the format before the fp8 MFMA was used anywhere
    m.load_state_dict(sd, strict=True)
    return m.to(DEV)


def _worst(out, ref):
    worst = {}
    for o, g in zip(out, ref):
        for k in g:
            assert torch.isfinite(o[k]).all(), k
            worst[k] = max(worst.get(k, 0.0), rel_l2(o[k].cpu(), g[k]))
    return worst


@pytest.mark.parametrize("n_views", [3, 8])
def test_vit_large_depth_with_stress_weights_vs_cpu_oracle(built_lib, n_views):
    (enc, dec, head), sd = _vitl_hot()
    views = make_views(n_views, 512, 512)
    O.ATTN_IMPL = "sdpa"
    try:
        with torch.no_grad():
            torch.manual_seed(1234)
            ref = O.forward(views, sd, enc, dec, head)
    finally:
        O.ATTN_IMPL = "naive"
    gv = views_to(views, DEV)
    report = {}
    for dt, precision in ((torch.float16, "high"), (torch.float16, "high+fp16"), (torch.float16, "fast"), (torch.bfloat16, "fast"), (torch.float16, "exact")):
        m = _build(dt, precision.split("+")[0], low_plane="fp16" if precision.endswith("+fp16") else "fp8")
        with torch.no_grad():
            torch.manual_seed(1234)
            out = m(gv)
        report[(str(dt).replace("torch.", ""), precision)] = w = _worst(out, ref)
        print(f"[parity] ViT-L HOT N={n_views} 512^2 {dt} {precision} vs CPU oracle: " + ", ".join(f"{k}={v:.2e}" for k, v in w.items()))
        del m, out
        torch.cuda.empty_cache()
    assert max(report[("float16", "high")].values()) <= TOL, report
    # "high" = the defaults Fast3R.low_plane = head_corrections = "fp8" (the MLPs' and the heads' correction products on the block-scaled fp8 MFMA);
    # "high+fp16" = fp16 planes everywhere: the same bar
    assert max(report[("float16", "high+fp16")].values()) <= TOL, report
    assert max(report[("float16", "exact")].values()) <= 2e-5, report   # the fp32-equivalent mode stays an anchor at depth 48


def test_vit_large_depth_with_heavy_tailed_weights_vs_cpu_oracle(built_lib):
    """a SECOND stress distribution (VERDICT r4 weak #1c; synthetic.py dist="heavy"): Student-t (nu = 4) weights scaled to variance 1 / fan_in
    and per-channel LayerNorm gains log-uniform in [0.2, 5], ViT-L / ViT-L / 2 DPT heads at N = 3 views of 512^2 against the CPU oracle; all
    four operand formats are printed.

    MEASURED (MI355X, round 5): fp16 / high 2.5e-3, fp16 / fast 4.3e-3, bf16 / fast 4.6e-2, exact 1.6e-5 -- on THIS distribution no 16-bit
    operand format holds the 1e-3 bar.  oracle/precision_study.py on the same weights (tiny model) says why: the network itself amplifies
    perturbations here -- rounding ONLY the attention operands to fp16, or ONLY the linear layers' activations, each alone gives ~5e-3, and
    even the fp32 path's own rounding noise comes out at 1.6e-5 instead of 3e-7; removing any single 16-bit rounding site buys nothing, only
    the fp32-equivalent mode (precision="exact", inference(dtype="32")) holds.  So the 1e-3 parity of the benchmarked format is a statement
    about the default-init protocol and the N(0, 1/fan_in) stress set (asserted elsewhere), NOT about every weight distribution; what IS
    asserted here: the exact mode stays at fp32 noise, the split planes still improve on single fp16, and the benchmarked format stays
    within 5e-3 (a regression guard at twice the measured distance)."""
    (enc, dec, head), _ = _vitl_hot()
    shapes = {k: tuple(v.shape) for k, v in Fast3R(enc, dec, head).state_dict().items()}
    sd = synth_state_dict(shapes, 0, dist="heavy")
    views = make_views(3, 512, 512)
    O.ATTN_IMPL = "sdpa"
    try:
        with torch.no_grad():
            torch.manual_seed(1234)
            ref = O.forward(views, sd, enc, dec, head)
    finally:
        O.ATTN_IMPL = "naive"
    gv = views_to(views, DEV)
    report = {}
    for dt, precision in ((torch.float16, "high"), (torch.float16, "robust"), (torch.float16, "robust+fp16"), (torch.float16, "robust+encoder"), (torch.float16, "fast"),
                          (torch.bfloat16, "fast"), (torch.float16, "exact")):
        m = Fast3R(enc, dec, head, compute_dtype=dt, precision=precision.split("+")[0]).eval()
        m.robust_corrections = "fp16" if precision.endswith("+fp16") else "fp8"   # where the score corrections run: fp8 MFMA (default) or two more fp16 products
        m.robust_encoder_attention = "planes" if precision.endswith("+encoder") else "fp32"   # the encoder's attention on the three-product kernel too (faster, less margin)
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV)
        with torch.no_grad():
            torch.manual_seed(1234)
            out = m(gv)
        report[(str(dt).replace("torch.", ""), precision)] = w = _worst(out, ref)
        print(f"[parity] ViT-L HEAVY-TAILED N=3 512^2 {dt} {precision} vs CPU oracle: " + ", ".join(f"{k}={v:.2e}" for k, v in w.items()))
        del m, out
        torch.cuda.empty_cache()
    hi, fast, exact, robust = (max(report[("float16", p)].values()) for p in ("high", "fast", "exact", "robust"))
    assert exact <= 3e-5, report
    assert hi <= fast and hi <= 5e-3, report
    # round 6: precision "robust" (linear layers X3, Q K^T from hi + lo planes on f3r_attn_asm_qk3_f16) is the 16-bit-operand tier that DOES hold the
    # bar on this distribution (CPU emulation of the same operand set: 3.6e-4, oracle/precision_study.py --study robust_vitl)
    assert robust <= TOL and max(report[("float16", "robust+fp16")].values()) <= TOL and max(report[("float16", "robust+encoder")].values()) <= TOL, report
    assert robust <= 3e-4, report   # the default (fp32 attention in the encoder) keeps a wide margin: measured 7.8e-5


def test_vit_large_n100_stress_weights_vs_fp32_equivalent_path(built_lib):
    views = views_to(make_views(100, 512, 512), DEV)
    m = _build(torch.float16, "exact")
    with torch.no_grad():
        torch.manual_seed(4321)
        ref = [{k: v.cpu() for k, v in o.items()} for o in m(views)]
    del m
    torch.cuda.empty_cache()
    report = {}
    for dt, precision in ((torch.float16, "high"), (torch.float16, "high+fp16"), (torch.float16, "fast")):
        m = _build(dt, precision.split("+")[0], low_plane="fp16" if precision.endswith("+fp16") else "fp8")
        with torch.no_grad():
            torch.manual_seed(4321)
            out = m(views)
        report[(str(dt), precision)] = w = _worst(out, ref)
        print(f"[parity] ViT-L HOT N=100 512^2 {dt} {precision} vs exact: " + ", ".join(f"{k}={v:.2e}" for k, v in w.items()))
        del m, out
        torch.cuda.empty_cache()
    assert max(report[(str(torch.float16), "high")].values()) <= TOL, report
    assert max(report[(str(torch.float16), "high+fp16")].values()) <= TOL, report


def test_vit_large_n320_stress_weights_vs_fp32_equivalent_path(built_lib):
    """(d) THE BENCHMARKED CONFIGURATION (BASELINE.json metric: N = 320 views of 512^2, ViT-L, one GPU) on the stress weights: every output
    of all 320 views of the fp16 / high forward within 1e-3 rel-L2 of the fp32-equivalent mode.  Feasible inside a test since round 4:
    from 8192 keys on the exact mode's attention runs on the matrix pipe (f3r_attn_f32_mfma: three-plane products, fp32 softmax; ~1 s per
    fusion layer at 327 680 tokens instead of ~15 s), a form tests/test_exact_mfma_gpu.py ties to the FMA kernel and to float64 and the
    N = 8 case above (8192 fusion tokens) ties to the CPU oracle through all 48 blocks."""
    views = views_to(make_views(320, 512, 512), DEV)
    m = _build(torch.float16, "exact")
    with torch.no_grad():
        torch.manual_seed(4321)
        ref = [{k: v.cpu() for k, v in o.items()} for o in m(views)]
    del m
    torch.cuda.empty_cache()
    m = _build(torch.float16, "high")
    taps = []
    m.kv_tap = lambda k, vt: taps.append((k.clone(), vt.clone()))
    with torch.no_grad():
        torch.manual_seed(4321)
        out = m(views)
    m.kv_tap = None
    w = _worst(out, ref)
    print("[parity] ViT-L HOT N=320 512^2 fp16 high vs exact: " + ", ".join(f"{k}={v:.2e}" for k, v in w.items()))
    assert max(w.values()) <= TOL, w
    # BASELINE configs[3] (N = 320 as 8 shards of 40 views) against the SAME checker, not only against the unsharded fp16 forward
    # (tests/test_realsize_gpu.py): one GPU runs rank 3's exact work -- its 40 views, the local launch parking the softmax state, the
    # remote launch over the 7 other shards' K / V^T as the unsharded forward produced them -- on the stress weights
    del out
    world, rank, N = 8, 3, 320
    per = N * 1024 // world

    def kv_source(layer, r, k_out, vt_out):
        k, vt = taps[layer]
        vt = vt.view(vt.shape[-2], vt.shape[-1])
        k_out.copy_(k[r * per:(r + 1) * per])
        vt_out.copy_(vt[:, r * per:(r + 1) * per])
    m.emulate_rank(rank, world, kv_source)
    with torch.no_grad():
        torch.manual_seed(4321)
        mine = m(views)
    m.emulate_rank(None, 0)
    lo = rank * (N // world)
    assert len(mine) == N // world
    w8 = _worst(mine, ref[lo:lo + N // world])
    print(f"[parity] ViT-L HOT N=320, rank {rank} of {world} (40 views, two-launch attention) fp16 high vs exact: " + ", ".join(f"{k}={v:.2e}" for k, v in w8.items()))
    assert max(w8.values()) <= TOL, w8


def test_model_scaling_huge_decoder_stress_weights_vs_fp32_equivalent_path(built_lib):
    """(e) the one reference configuration whose heads are not 64 wide (configs/experiment/model_scaling/model_scaling_huge.yaml:12-15: fusion
    decoder 1280 / 16 heads = head_dim 80, depth 32) behind the ViT-L encoder, 40 views of 512^2 (40 960 fusion tokens), stress weights:
    fp16 / high -- whose fusion attention is the generated kernel f3r_attn_asm_d80_f16, asserted from the launch records -- against the
    fp32-equivalent mode (head_dim 80 there: the FMA-pipe attention), every output <= 1e-3."""
    enc, dec, head = vit_large_args()
    dec = dict(dec, embed_dim=1280, num_heads=16, depth=32)
    shapes = {k: tuple(v.shape) for k, v in Fast3R(enc, dec, head).state_dict().items()}
    sd = synth_state_dict(shapes, 1, dist="hot")
    views = views_to(make_views(40, 512, 512), DEV)

    def run(precision):
        m = Fast3R(enc, dec, head, compute_dtype=torch.float16, precision=precision).eval()
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV)
        ops.ATTN_TIMER = []
        try:
            with torch.no_grad():
                torch.manual_seed(99)
                out = [{k: v.cpu() for k, v in o.items()} for o in m(views)]
            names = {rec[5] for rec in ops.ATTN_TIMER}
        finally:
            ops.ATTN_TIMER = None
        del m
        torch.cuda.empty_cache()
        return out, names
    ref, _ = run("exact")
    out, names = run("high")
    assert any("f3r_attn_asm_d80_f16" in n for n in names), names
    w = _worst(out, ref)
    print("[parity] ViT-L encoder + model_scaling_huge decoder, HOT, N=40 512^2 fp16 high vs exact: " + ", ".join(f"{k}={v:.2e}" for k, v in w.items()))
    assert max(w.values()) <= TOL, w


# ------------------------------------------------------------------------------------------------ (c) GEMM / conv roles at N = 320 shapes
M320 = 327680
NS = 4096


def _dev_randn(shape, seed, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(shape, generator=g, device=DEV) * scale


def _sample_rows(M, n, seed, tile=256):
    """row indices drawn uniformly + the edges of the first / a middle / the last 256-row tile"""
    g = torch.Generator().manual_seed(seed)
    r = torch.randint(0, M, (n,), generator=g)
    edges = []
    for t0 in (0, (M // tile // 2) * tile, ((M - 1) // tile) * tile):
        edges += [t0, t0 + 1, t0 + 127, t0 + 128, min(M - 1, t0 + tile - 1)]
    edges += [M - 1, M - 2, M - 129]
    r[:len(edges)] = torch.tensor(edges).clamp_(0, M - 1)
    return r.to(DEV)


def _sample_cols(N, n, seed):
    g = torch.Generator().manual_seed(seed + 1)
    c = torch.randint(0, N, (n,), generator=g)
    c[:6] = torch.tensor([0, 1, 127, 128, N - 1, N - 2]).clamp_(0, N - 1)
    return c.to(DEV)


def _planes(x32, dt, split):
    """the operand(s) a kernel multiplies: (hi,) or (hi, lo)"""
    hi = x32.to(dt)
    return (hi, (x32 - hi.float()).to(dt)) if split else (hi,)


def _rowdot(a_rows, w_rows):
    return (a_rows.double() * w_rows.double()).sum(-1)


def _check(got, ref, tol, what):
    got, ref = got.double().cpu(), ref.double().cpu()
    scale = float(ref.abs().max().clamp_min(1e-6))
    err = float((got - ref).abs().max())
    assert math.isfinite(err) and err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol:.1e})"


def _lin_case(dt, split, K, N, seed):
    """A lowp [M][K] (single plane: the transformer's activations are single in every mode), W fp32 (N, K) packed per `split`"""
    a = _dev_randn((M320, K), seed, 1.0).to(dt)
    w32 = _dev_randn((N, K), seed + 1, K ** -0.5)
    wp = ops.pack_linear_weight(w32, dt, split=bool(split))
    w_eff = w32 if split else w32.to(dt).float()  # W2 recovers the fp32 weight to ~2^-22; single plane multiplies the rounded one
    bias = _dev_randn((N,), seed + 2, 0.5)
    return a, w32, wp, w_eff, bias


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("split", [None, "w2"])
def test_role_proj_and_fc2_residual_at_n320(built_lib, dt, split):
    """x += A W^T + b in place, fp32 (blocks.py:237-238): proj K = 1024, fc2 K = 4096."""
    if split and dt == torch.bfloat16:
        pytest.skip("the split planes are an fp16 design (DESIGN.md section 3)")
    for K, seed in ((1024, 100), (4096, 110)):
        a, w32, wp, w_eff, bias = _lin_case(dt, split, K, 1024, seed)
        x = _dev_randn((M320, 1024), seed + 3, 2.0)
        rows, cols = _sample_rows(M320, NS, seed), _sample_cols(1024, NS, seed)
        ref = _rowdot(a[rows], w_eff[cols]) + bias[cols].double() + x[rows, cols].double()
        ops.gemm(a, wp, bias=bias, res_f32=x, out_f32=x, split=split)
        _check(x[rows, cols], ref, 3e-5 if not split else 2e-5, f"residual role K={K} {dt} split={split}")
        del a, w32, wp, x
        torch.cuda.empty_cache()


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("split", [None, "w2"])
def test_role_fc1_gelu_at_n320(built_lib, dt, split):
    if split and dt == torch.bfloat16:
        pytest.skip("the split planes are an fp16 design")
    a, w32, wp, w_eff, bias = _lin_case(dt, split, 1024, 4096, 200)
    rows, cols = _sample_rows(M320, NS, 200), _sample_cols(4096, NS, 200)
    ref = F.gelu(_rowdot(a[rows], w_eff[cols]) + bias[cols].double())
    _, y = ops.gemm(a, wp, bias=bias, act="gelu", want_lp=True, split=split)
    _check(y[rows, cols], ref, 2.0 ** -9 if dt == torch.float16 else 2.0 ** -6, f"fc1+GELU {dt} split={split}")


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("split", [None, "w2"])
@pytest.mark.parametrize("use_rope", [False, True])
def test_role_qkv_at_n320(built_lib, dt, split, use_rope):
    """QKV + (encoder: RoPE-2D) + q scale + V^T layout (blocks.py:138-143, pos_embed.py:162-183) at 320 sequences of 1024 tokens."""
    if split and dt == torch.bfloat16:
        pytest.skip("the split planes are an fp16 design")
    D, S, n_seq = 1024, 1024, 320
    a, w32, wp, w_eff, bias = _lin_case(dt, split, 1024, 3 * D, 300)
    q = torch.empty((M320, D), dtype=dt, device=DEV)
    k = torch.empty((M320, D), dtype=dt, device=DEV)
    seq_len = S if use_rope else M320  # decoder: ONE sequence of all tokens; encoder: one per view
    ns = n_seq if use_rope else 1
    vt = torch.zeros((ns, D, ops.vt_ld(seq_len)), dtype=dt, device=DEV)
    rope = None
    if use_rope:
        cos, sin = ops.rope_tables(32, 100.0, DEV)
        rope = (cos, sin, 32)
    qs = 0.160192 * ops.LOG2E
    ops.gemm_qkv(a, wp, bias, q, k, vt, seq_len, rope, q_scale=qs, split=split)
    rows = _sample_rows(M320, NS, 300)
    g = torch.Generator().manual_seed(301)
    heads = torch.randint(0, 16, (NS,), generator=g).to(DEV)
    dims = torch.randint(0, 64, (NS,), generator=g).to(DEV)
    tol = 2.0 ** -9 if dt == torch.float16 else 2.0 ** -6

    def lin(part, h, d):  # column of the fused projection
        c = part * D + h * 64 + d
        return _rowdot(a[rows], w_eff[c]) + bias[c].double()

    for part, got_t in ((0, q), (1, k)):
        v = lin(part, heads, dims)
        if use_rope:
            # pos_embed.py:162-183: dims [0,32) rotate by the row position, [32,64) by the column position; pairs (i, i+16) inside a half
            pos = rows % S
            py, px = pos // 32, pos % 32
            half, i = dims // 32, dims % 32
            p = torch.where(half == 0, py, px)
            j = i % 16
            partner = half * 32 + (i + 16) % 32
            vp = lin(part, heads, partner)
            c_, s_ = cos[p, j].double(), sin[p, j].double()
            v = torch.where(i < 16, v * c_ - vp * s_, v * c_ + vp * s_)
        if part == 0:
            v = v * qs
        _check(got_t[rows, heads * 64 + dims], v, tol, f"qkv part {part} rope={use_rope} {dt} split={split}")
    v = lin(2, heads, dims)
    got = vt[rows // seq_len, heads * 64 + dims, rows % seq_len]
    _check(got, v, tol, f"v^T rope={use_rope} {dt} split={split}")


def _conv_case(dt, split, B, H, W, Ci, Co, seed):
    x32 = _dev_randn((B, H, W, Ci), seed, 1.0)
    w32 = _dev_randn((Co, Ci, 3, 3), seed + 1, (9 * Ci) ** -0.5)
    bias = _dev_randn((Co,), seed + 2, 0.5)
    xs = _planes(x32, dt, split)
    wp = ops.pack_conv3x3_weight(w32, dt, split=bool(split))
    x_eff = x32 if split else xs[0].float()
    w_eff = w32 if split else w32.to(dt).float()
    return xs, wp, x_eff, w_eff, bias


def _conv_ref(x_eff, w_eff, bias, b, oy, ox, co):
    xp = F.pad(x_eff, (0, 0, 1, 1, 1, 1))  # NHWC: pad W and H by one
    dy = torch.arange(3, device=DEV).view(1, 3, 1)
    dx = torch.arange(3, device=DEV).view(1, 1, 3)
    patch = xp[b.view(-1, 1, 1), oy.view(-1, 1, 1) + dy, ox.view(-1, 1, 1) + dx]          # (S, 3, 3, C)
    wsel = w_eff[co].permute(0, 2, 3, 1)                                                   # (S, 3, 3, C)
    return (patch.double() * wsel.double()).sum((1, 2, 3)) + bias[co].double()


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("split", [None, "x3"])
@pytest.mark.parametrize("shape", [(20, 128, 128, 256, 256), (2, 512, 512, 128, 128)])
def test_role_conv3x3_at_n320(built_lib, dt, split, shape):
    """refinenet RCU conv 256 -> 256 at 128^2 (M = 20 views x 16 384 = 327 680 rows) and head conv 128 -> 128 at 512^2 (M = 524 288;
    dpt_block.py:133-154,365-382), + bias + ReLU, incl. image-border pixels (zero padding through the zero line)."""
    if split and dt == torch.bfloat16:
        pytest.skip("the split planes are an fp16 design")
    B, H, W, Ci, Co = shape
    xs, wp, x_eff, w_eff, bias = _conv_case(dt, split, B, H, W, Ci, Co, 400 + Ci)
    r = ops.conv3x3(xs[0], wp, bias=bias, act="relu", split=split, x_lo=xs[1] if split else None, want_lo=bool(split))
    y = r["out"].float() + r["out_lo"].float() if split else r.float()
    g = torch.Generator().manual_seed(500 + Ci)
    b = torch.randint(0, B, (NS,), generator=g)
    oy = torch.randint(0, H, (NS,), generator=g)
    ox = torch.randint(0, W, (NS,), generator=g)
    co = torch.randint(0, Co, (NS,), generator=g)
    # corners and borders of the first and the last image
    b[:8] = torch.tensor([0, 0, 0, 0, B - 1, B - 1, B - 1, B - 1])
    oy[:8] = torch.tensor([0, 0, H - 1, H - 1, 0, 0, H - 1, H - 1])
    ox[:8] = torch.tensor([0, W - 1, 0, W - 1, 0, W - 1, 0, W - 1])
    b, oy, ox, co = b.to(DEV), oy.to(DEV), ox.to(DEV), co.to(DEV)
    ref = F.relu(_conv_ref(x_eff, w_eff, bias, b, oy, ox, co))
    tol = 2e-5 if split else (2.0 ** -9 if dt == torch.float16 else 2.0 ** -6)
    _check(y[b, oy, ox, co], ref, tol, f"conv3x3 {shape} {dt} split={split}")
