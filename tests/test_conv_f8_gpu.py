"""The DPT head's 3x3 convolutions with fp8 correction products (f3r_gemm split "x3f8", include/f3r.h F3R_SPLIT_X3F8) and the head's tail fused into
head[2]'s epilogue (f3r_gemm_args.fin_w), on a real MI355X through the C ABI (reference: fast3r/croco/models/dpt_block.py:133-154,365-382,
fast3r/dust3r/heads/postprocess.py:16-64).

Two kinds of checks, as for the fp8 low plane of the linear layers (tests/test_gemm_asm_gpu.py):
  (a) the kernel computes EXACTLY its planes: float64 on the decoded operand planes (fp16 hi, e4m3 hi8 / lo8 with their power-of-two scales) --
      a wrong lane / byte / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 cannot pass this at 1e-5;
  (b) against the UNROUNDED fp32 operands the result is fp32-class (what the two-fp16-plane X3 form gives), orders of magnitude better than a
      single plane.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from fast3r_amd import ops
from test_kernels_gpu import DEV, assert_close

pytestmark = pytest.mark.gpu
H16 = torch.float16


def _dec8(b):
    return b.view(torch.float8_e4m3fn).double()


def _decode_weight_f8(wp, sc, co, ci):
    """pack_conv3x3_weight_f8 rows -> (w_hi, w_lo8 decoded, w_hi8 decoded) as float64 (Cout, Cin, 3, 3)"""
    kp = 9 * ci
    raw = wp.view(torch.uint8).view(co, 4 * kp)
    hi = raw[:, :2 * kp].contiguous().view(torch.float16).double()
    e_lo = (sc.long() & 0xff).double()
    e_hi = ((sc.long() >> 8) & 0xff).double()
    lo8 = _dec8(raw[:, 2 * kp:3 * kp].contiguous()) * torch.exp2(e_lo - 127)[:, None]
    hi8 = _dec8(raw[:, 3 * kp:].contiguous()) * torch.exp2(e_hi - 127)[:, None]
    cv = lambda t: t.view(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous()
    return cv(hi), cv(lo8), cv(hi8)


def _conv64(x_nhwc, w, stride=1):
    return F.conv2d(x_nhwc.double().permute(0, 3, 1, 2), w.double(), None, padding=1, stride=stride).permute(0, 2, 3, 1)


def _planes_ref(x32, wp, sc, co, ci):
    """what split "x3f8" computes, in float64, from the planes the kernel reads"""
    x_hi = x32.to(H16)
    p8 = ops.f8_planes(x32)
    a_hi8 = _dec8(p8[..., :ci].contiguous())
    a_lo8 = _dec8(p8[..., ci:].contiguous()) / 4096.0
    w_hi, w_lo8, w_hi8 = _decode_weight_f8(wp, sc, co, ci)
    return _conv64(x_hi, w_hi) + _conv64(a_hi8, w_lo8) + _conv64(a_lo8, w_hi8), x_hi, p8


@pytest.mark.parametrize("B,H,W,Ci,Co,res", [(2, 48, 40, 128, 128, False), (1, 64, 64, 256, 128, False), (2, 40, 48, 256, 256, True),
                                             (3, 31, 33, 256, 256, False), (1, 33, 17, 128, 256, True)])
def test_conv_x3f8_planes_and_accuracy(built_lib, B, H, W, Ci, Co, res):
    g = torch.Generator().manual_seed(7 + Ci + Co)
    x32 = torch.randn((B, H, W, Ci), generator=g) * 1.5
    x32[0, 0, 0, :8] = torch.tensor([700.0, -900.0, 448.0, -448.0, 1e-4, -3e-6, 0.0, 2000.0])  # beyond e4m3's range / below its subnormals
    w32 = torch.randn((Co, Ci, 3, 3), generator=g) * (9 * Ci) ** -0.5
    w32[1] *= 1e-3  # per-output-channel scales matter
    w32[2] *= 50.0
    bias = torch.randn(Co, generator=g)
    r32 = torch.randn((B, H, W, Co), generator=g)
    wp, sc = ops.pack_conv3x3_weight_f8(w32)
    ref_planes, x_hi, p8 = _planes_ref(x32, wp, sc, Co, Ci)
    ref_planes = ref_planes + bias.double()
    ref_true = _conv64(x32, w32) + bias.double()
    kw = {}
    if res:
        r_hi, r_lo = ops.split_planes(r32, H16)
        kw = dict(res_lp=r_hi.to(DEV), res_lp_lo=r_lo.to(DEV))
        ref_planes = ref_planes + r_hi.double() + r_lo.double()
        ref_true = ref_true + r32.double()
    r = ops.conv3x3(x_hi.to(DEV), wp.to(DEV), bias=bias.to(DEV), split="x3f8", x_f8=p8.to(DEV), w_scale=sc.to(DEV), want_lo=True, want_relu=True,
                    want_f8=True, want_relu_f8=True, **kw)
    got = r["out"].float().double().cpu() + r["out_lo"].float().double().cpu()
    # the out-of-range elements (clamped in the fp8 planes, as documented) only touch the outputs of pixel (0, 0, 0)'s neighbourhood in (b);
    # the error scale is the largest output AWAY from them
    far = torch.ones_like(ref_true, dtype=torch.bool)
    far[0, :2, :2] = False
    scale = float((ref_true.abs() * far).max())
    e_planes = float(((got - ref_planes).abs() * far).max()) / scale
    e_planes_near = float(((got - ref_planes).abs() / ref_planes.abs().clamp_min(scale)).max())  # (relative to the large outputs beside the big inputs)
    e_true = float((got - ref_true).abs().max()) / scale
    e_true_far = float(((got - ref_true).abs() * far).max()) / scale
    print(f"x3f8 conv {B}x{H}x{W} {Ci}->{Co} res={res}: vs planes {e_planes:.2e}, vs fp32 operands {e_true_far:.2e} (incl. clamped pixel {e_true:.2e}; planes there {e_planes_near:.2e})")
    assert e_planes < 2e-6 and e_planes_near < 4e-6, (e_planes, e_planes_near)
    assert e_true_far < 3e-5, e_true_far
    # a single fp16 plane on the same operands is far coarser: the test notices a kernel that dropped the corrections
    single = ops.conv3x3(x_hi.to(DEV), ops.pack_conv3x3_weight(w32, H16).to(DEV), bias=bias.to(DEV), want_lo=True, **kw)
    e_single = float(((single["out"].float().double().cpu() + single["out_lo"].float().double().cpu() - ref_true).abs() * far).max()) / scale
    assert e_single > 8 * e_true_far, (e_single, e_true_far)
    # the fp8 planes the epilogue writes for the NEXT conv: hi8 within one e4m3 step (2^-3 relative after clamping) of the value, lo8 of the fp16 rest
    for name, val in (("out", got), ("relu", F.relu(got))):
        p = r[name + "_f8"].cpu()
        assert p.shape == (B, H, W, 2 * Co) and p.dtype == torch.uint8
        hi8, lo8 = _dec8(p[..., :Co].contiguous()), _dec8(p[..., Co:].contiguous()) / 4096.0
        v = val.clamp(-448.0, 448.0)
        worst = float(((hi8 - v).abs() - (v.abs() * 2.0 ** -4 + 2.0 ** -10)).max())
        assert worst <= 0.0, (name, worst, float((hi8 - v).abs().max()), float(v.abs().max()))
        hi_plane = r[name].float().double().cpu()   # (the rest is taken against the kernel's OWN fp16 rounding: at a tie hi + lo rounds the other way)
        rest = (val - hi_plane).clamp(-448.0 / 4096, 448.0 / 4096)
        assert float(((lo8 - rest).abs() - (rest.abs() * 2.0 ** -4 + 2.0 ** -22)).max()) <= 1e-9, name
        assert float((hi_plane - val).abs().max()) <= float(val.abs().max()) * 2.0 ** -10


@pytest.mark.parametrize("split", ["x3", "x3f8"])
def test_conv_stride_2_on_the_256_tile_kernel(built_lib, split):
    """act_postprocess[3][1] (Conv2d 3x3, stride 2, dpt_block.py:466-481): since round 6 the 256-tile kernel takes strided convolutions too (the lane's
    pixel is the input pixel under the centre tap; taps and padding as before) -- every kernel form against float64 on the unrounded operands"""
    B, H, W, Ci, Co = 2, 33, 40, 128, 256
    g = torch.Generator().manual_seed(21)
    x32 = torch.randn((B, H, W, Ci), generator=g)
    w32 = torch.randn((Co, Ci, 3, 3), generator=g) * (9 * Ci) ** -0.5
    bias = torch.randn(Co, generator=g)
    ref = _conv64(x32, w32, stride=2) + bias.double()
    x_hi, x_lo = ops.split_planes(x32, H16)
    if split == "x3f8":
        wp, sc = ops.pack_conv3x3_weight_f8(w32)
        kw = dict(split="x3f8", x_f8=ops.f8_planes(x32).to(DEV), w_scale=sc.to(DEV))
        sels = [0]
    else:
        wp = ops.pack_conv3x3_weight(w32, H16, split=True)
        kw = dict(split="x3", x_lo=x_lo.to(DEV))
        sels = [1, 2, 4, 0]
    outs = []
    for sel in sels:
        r = ops.conv3x3(x_hi.to(DEV), wp.to(DEV), stride=2, bias=bias.to(DEV), want_lo=True, kernel_sel=sel, **kw)
        assert r["out"].shape == (B, 17, 20, Co)
        assert_close(r["out"].float().double().cpu() + r["out_lo"].float().double().cpu(), ref, 2e-5, f"stride-2 conv {split} sel={sel}")
        outs.append(r["out"])


def test_conv_x3f8_needs_the_256_tile_kernel(built_lib):
    """no second kernel reads this layout: a launch the 256-tile kernel cannot take is an error, never a fallback"""
    x32 = torch.randn((1, 8, 8, 128))
    w32 = torch.randn((128, 128, 3, 3)) * 0.03
    wp, sc = ops.pack_conv3x3_weight_f8(w32)
    w96 = torch.randn((96, 128, 3, 3)) * 0.03
    wp96, sc96 = ops.pack_conv3x3_weight_f8(w96)
    with pytest.raises(ValueError):  # 96 output channels: not a multiple of the 128-wide tile half
        ops.conv3x3(x32.to(H16).to(DEV), wp96.to(DEV), split="x3f8", x_f8=ops.f8_planes(x32).to(DEV), w_scale=sc96.to(DEV))
    xb = x32.to(torch.bfloat16)
    with pytest.raises(ValueError):  # bf16 planes
        ops.conv3x3(xb.to(DEV), wp.view(torch.bfloat16).to(DEV), split="x3f8", x_f8=ops.f8_planes(x32).to(DEV), w_scale=sc.to(DEV))


@pytest.mark.parametrize("B,h,w,C,oh,ow", [(2, 16, 24, 128, 32, 48), (1, 9, 9, 256, 17, 15)])
def test_interp_bilinear_f8_planes(built_lib, B, h, w, C, oh, ow):
    g = torch.Generator().manual_seed(5)
    x32 = torch.randn((B, h, w, C), generator=g) * 3.0
    x_hi, x_lo = ops.split_planes(x32, H16)
    plain = ops.interp_bilinear(x_hi.to(DEV), (oh, ow), x_lo=x_lo.to(DEV), want_lo=True, nominal_hw=(2 * h, 2 * w))
    out, out_lo, p8 = ops.interp_bilinear(x_hi.to(DEV), (oh, ow), x_lo=x_lo.to(DEV), want_lo=True, want_f8=True, nominal_hw=(2 * h, 2 * w))
    assert torch.equal(out, plain[0]) and torch.equal(out_lo, plain[1])
    val = out.float().double().cpu() + out_lo.float().double().cpu()
    p = p8.cpu()
    hi8, lo8 = _dec8(p[..., :C].contiguous()), _dec8(p[..., C:].contiguous()) / 4096.0
    assert float(((hi8 - val).abs() - (val.abs() * 2.0 ** -4 + 2.0 ** -10)).max()) <= 0.0
    rest = out_lo.float().double().cpu()
    assert float(((lo8 - rest).abs() - (rest.abs() * 2.0 ** -4 + 2.0 ** -22)).max()) <= 1e-9
    only8 = ops.interp_bilinear(x_hi.to(DEV), (oh, ow), x_lo=x_lo.to(DEV), want_f8=True, nominal_hw=(2 * h, 2 * w))
    assert only8[1] is None and torch.equal(only8[2], p8) and torch.equal(only8[0], out)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("split", [None, "x3", "x3f8"])
@pytest.mark.parametrize("B,H,W,conf", [(2, 64, 64, True), (1, 100, 52, True), (3, 33, 47, False)])
def test_fused_head_tail_matches_the_separate_kernels(built_lib, dt, split, B, H, W, conf):
    """head[2] (+ ReLU) -> head[4] -> postprocess in ONE launch (fin) against conv3x3 -> f3r_dpt_final, and against float64 on the same planes"""
    if dt == torch.bfloat16 and split == "x3f8":
        pytest.skip("the fp8 correction planes go with fp16 high planes")
    bf = dt == torch.bfloat16   # (hi + lo bf16 planes carry 16 significand bits: the separate path's output planes are that coarse)
    C = N = 128
    g = torch.Generator().manual_seed(11)
    x32 = torch.randn((B, H, W, C), generator=g)
    w32 = torch.randn((N, C, 3, 3), generator=g) * (9 * C) ** -0.5
    bias = torch.randn(N, generator=g) * 0.1
    n_out = 4 if conf else 3
    w4 = torch.randn((n_out, N), generator=g) * 0.08
    b4 = torch.randn(n_out, generator=g) * 0.1
    conf_mode = ("exp", 1.0, math.inf) if conf else None
    x_hi, x_lo = ops.split_planes(x32, dt)
    kw = {}
    if split == "x3f8":
        wp, sc = ops.pack_conv3x3_weight_f8(w32)
        kw = dict(split="x3f8", x_f8=ops.f8_planes(x32).to(DEV), w_scale=sc.to(DEV))
    elif split == "x3":
        wp = ops.pack_conv3x3_weight(w32, dt, split=True)
        kw = dict(split="x3", x_lo=x_lo.to(DEV))
    else:
        wp = ops.pack_conv3x3_weight(w32, dt)
    fin = ops.dpt_fin_args(w4.to(DEV), b4.to(DEV), conf_mode)
    pts, cf = ops.conv3x3(x_hi.to(DEV), wp.to(DEV), bias=bias.to(DEV), act="relu", fin=fin, **kw)
    sep = ops.conv3x3(x_hi.to(DEV), wp.to(DEV), bias=bias.to(DEV), act="relu", want_lo=True, **kw)
    pts2, cf2 = ops.dpt_final(sep["out"], w4.to(DEV), b4.to(DEV), conf_mode, x_lo=sep["out_lo"])
    assert pts.shape == (B, H, W, 3) and (cf is None) == (not conf)
    # the separate path rounds head[2]'s output to hi + lo planes (2^-22); the fused one keeps fp32: equal to that rounding
    assert_close(pts, pts2.double().cpu(), 1e-4 if bf else 2e-5, "fused pts3d vs separate kernels")
    if conf:
        assert_close(cf, cf2.double().cpu(), 1e-4 if bf else 2e-5, "fused conf vs separate kernels")
    if split != "x3f8":  # float64 on the operands the kernel reads
        xin = (x_hi.double() + x_lo.double()) if split == "x3" else x_hi.double()
        wq = w32.double() if split == "x3" else w32.to(dt).double()
        y = F.relu(_conv64(xin, wq) + bias.double())
        z = y @ w4.double().t() + b4.double()
        d = z[..., :3].norm(dim=-1, keepdim=True)
        ref = z[..., :3] / d.clamp_min(1e-8) * torch.expm1(d)
        assert_close(pts, ref, (2e-4 if bf else 3e-5) if split == "x3" else 2e-5, "fused pts3d vs float64")
        if conf:
            assert_close(cf, 1.0 + torch.exp(z[..., 3]), 2e-4 if bf and split == "x3" else 3e-5, "fused conf vs float64")
