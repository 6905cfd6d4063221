"""TEST INFRASTRUCTURE ONLY -- CPU emulation of operand-rounding designs for the HIP path (decides DESIGN.md section 3 (Precision modes)).

Runs the oracle (oracle/fast3r_oracle.py) with its GEMM-like functionals wrapped so that operands (and, optionally, stored
activations) are rounded the way a candidate kernel design would round them, and prints the end-to-end rel-L2 of every design
against the exact fp32 oracle on the stress fixture (`hot` weights) and the default one.

    python oracle/precision_study.py

Zones: "tr" = encoder + fusion transformer, "hd" = DPT heads.  A design is {zone: mode}:
    "f32"     exact
    "bf16" / "f16"   operands of every GEMM / conv rounded once (fp32 accumulate), outputs kept fp32
    "+store"  suffix: the op's OUTPUT is also rounded to that type (activations live in HBM as 16-bit)
    "f16x2"   activations split hi + lo (two fp16 numbers, 22 bits), weights single fp16
    "f16w2"   weights split hi + lo, activations single fp16
    "f16x3"   both split: a_hi w_hi + a_lo w_hi + a_hi w_lo
    "bf16x3"  same with bf16 pieces (3 x 8 bits; the classic bf16x3 ~ fp32-ish emulation)
    "f16w2_8" / "f16x3_8"   like f16w2 / f16x3, but the CORRECTION products (a w_lo, a_lo w_hi) have both operands rounded to fp8 e4m3 with a
              per-tensor power-of-two scale -- what a block-scaled FP8 MFMA (2x the fp16 rate on gfx950) would compute: a candidate for cutting
              the cost of precision="high" from 2x / 3x to 1.5x / 2x of the 16-bit GEMMs
"""
import math
import os
import sys
import types

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fast3r_amd.synthetic import make_views, synth_state_dict, tiny_args  # noqa: E402
from oracle import fast3r_oracle as O  # noqa: E402

DT = {"bf16": torch.bfloat16, "f16": torch.float16}


def rnd(x, dt):
    return x.to(dt).float()


def rnd8(x):
    """fp8 e4m3 with a per-tensor power-of-two scale into its range (max 448; subnormals below 2^-6 of the scaled value)."""
    m = float(x.abs().max())
    if m == 0.0:
        return x
    sc = 2.0 ** math.floor(math.log2(240.0 / m))
    return (x * sc).to(torch.float8_e4m3fn).float() / sc


def split2(x, dt):
    hi = rnd(x, dt)
    return hi, rnd(x - hi, dt)


class Zone:
    mode = "f32"
    roles = {}   # per-role override inside the transformer zone: {"fc1" | "fc2" | "qkv" | "proj": mode} (a linear layer's role is read off its weight shape)
    conv_roles = {}  # per-convolution override inside a DPT head: {call index inside dpt_forward: mode}; 0-6 act_postprocess, 7-10 layer_rn,
    conv_idx = 0     # 11-13 refinenet4, 14-18 refinenet3, 19-23 refinenet2, 24-28 refinenet1 (4 RCU convs + out_conv), 29 head.0, 30 head.2, 31 head.4


def linear_role(w):
    n, k = w.shape
    return "fc1" if n == 4 * k else "fc2" if k == 4 * n else "qkv" if n == 3 * k else "proj" if n == k else "other"


ZONE = Zone()


def _apply(fn, a, w, *rest, **kw):
    """fn(a, w, ...) bilinear in (a, w): emulate the operand rounding of ZONE.mode."""
    mode = ZONE.mode
    if fn is F.linear and ZONE.roles and w.dim() == 2:
        mode = ZONE.roles.get(linear_role(w), mode)
    if fn is not F.linear and w.dim() == 4:
        mode = ZONE.conv_roles.get(ZONE.conv_idx, mode)
        ZONE.conv_idx += 1
    store = mode.endswith("+store")
    base = mode.replace("+store", "")
    bias = rest[0] if rest else kw.pop("bias", None)
    rest = rest[1:]
    if base == "f32":
        out = fn(a, w, None, *rest, **kw)
    elif base in DT:
        out = fn(rnd(a, DT[base]), rnd(w, DT[base]), None, *rest, **kw)
    elif base in ("f16x2", "bf16x2"):
        dt = DT[base[:-2]]
        ah, al = split2(a, dt)
        wh = rnd(w, dt)
        out = fn(ah, wh, None, *rest, **kw) + fn(al, wh, None, *rest, **kw)
    elif base in ("f16w2", "bf16w2"):  # weights split, activations single
        dt = DT[base[:-2]]
        ah = rnd(a, dt)
        wh, wl = split2(w, dt)
        out = fn(ah, wh, None, *rest, **kw) + fn(ah, wl, None, *rest, **kw)
    elif base in ("f16w2_8", "f16x3_8"):
        dt = torch.float16
        ah, al = split2(a, dt)
        wh, wl = split2(w, dt)
        a1 = ah if base == "f16x3_8" else rnd(a, dt)
        out = fn(a1, wh, None, *rest, **kw) + fn(rnd8(a1), rnd8(wl), None, *rest, **kw)
        if base == "f16x3_8":
            out = out + fn(rnd8(al), rnd8(wh), None, *rest, **kw)
    elif base == "f16x3_8c":
        # what f3r_gemm split "x3f8" computes (round 6, DPT-head convolutions): a_hi w_hi on fp16; a_hi8 w_lo8 with one power-of-two scale per OUTPUT
        # CHANNEL on w_lo (largest |w_lo| of the row -> [112, 224]) and a_hi8 = e4m3(clamp(a_hi, 448)) unscaled; a_lo8 w_hi8 with the fixed scale 2^12 on
        # a_lo (clamped) and w_hi8 = e4m3(w_hi) scaled per output channel like the weights' low plane
        ah, al = split2(a, torch.float16)
        wh, wl = split2(w, torch.float16)
        def f8(x):
            return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()
        def per_row(x):
            amax = x.abs().flatten(1).amax(dim=1).clamp_min(2.0 ** -100)
            sc = torch.exp2(torch.floor(torch.log2(224.0 / amax))).view(-1, *([1] * (x.dim() - 1)))
            return f8(x * sc) / sc
        if fn is F.conv_transpose2d:
            out = fn(ah, wh, None, *rest, **kw) + fn(al, wh, None, *rest, **kw) + fn(ah, wl, None, *rest, **kw)
        else:
            out = fn(ah, wh, None, *rest, **kw) + fn(f8(ah), per_row(wl), None, *rest, **kw) + fn(f8(al * 4096.0) / 4096.0, per_row(wh), None, *rest, **kw)
    elif base in ("f16x3", "bf16x3"):
        dt = DT[base[:-2]]
        ah, al = split2(a, dt)
        wh, wl = split2(w, dt)
        out = fn(ah, wh, None, *rest, **kw) + fn(al, wh, None, *rest, **kw) + fn(ah, wl, None, *rest, **kw)
    else:
        raise ValueError(mode)
    if bias is not None:
        out = out + bias.view(1, -1, *([1] * (out.dim() - 2))) if fn is not F.linear else out + bias
    if store:
        out = rnd(out, DT[base])
    return out


class FProxy(types.SimpleNamespace):
    pass


def make_proxy():
    p = FProxy()
    for name in dir(F):
        if not name.startswith("_"):
            setattr(p, name, getattr(F, name))
    p.linear = lambda a, w, b=None: _apply(F.linear, a, w, b)
    p.conv2d = lambda a, w, b=None, **kw: _apply(F.conv2d, a, w, b, **kw)
    p.conv_transpose2d = lambda a, w, b=None, **kw: _apply(F.conv_transpose2d, a, w, b, **kw)
    return p


def attention_emul(x, sd, pre, num_heads, scale, xpos=None, rope_base=None, q_chunk=2048):
    """oracle.attention with q, k, v, P rounded like the attention kernel does when the transformer zone is 16-bit."""
    mode = ZONE.mode.replace("+store", "")
    B, S, C = x.shape
    qkv = O.F.linear(x, sd[pre + "qkv.weight"], sd.get(pre + "qkv.bias"))
    qkv = qkv.reshape(B, S, 3, num_heads, C // num_heads).transpose(1, 3)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    if rope_base is not None and xpos is not None:
        q = O.rope2d(q, xpos, rope_base)
        k = O.rope2d(k, xpos, rope_base)
    qk_dt = ATTN.get("qk", mode)
    pv_dt = ATTN.get("pv", mode)
    if qk_dt in DT:
        q, k = rnd(q * scale, DT[qk_dt]) / scale, rnd(k, DT[qk_dt])
        a = (q @ k.transpose(-2, -1)) * scale
    elif qk_dt == "f16x3_8":   # hi . hi on fp16, the two correction products from fp8 operands (block-scaled fp8 MFMA)
        qh, ql = split2(q * scale, torch.float16)
        kh, kl = split2(k, torch.float16)
        a = qh @ kh.transpose(-2, -1) + rnd8(ql) @ rnd8(kh).transpose(-2, -1) + rnd8(qh) @ rnd8(kl).transpose(-2, -1)
    elif qk_dt in ("f16x3", "f16q2", "f16k2"):   # round 6: Q and / or K as hi + lo fp16 planes (q_hi k_hi + q_lo k_hi + q_hi k_lo on the matrix pipe)
        qh, ql = split2(q * scale, torch.float16)
        kh, kl = split2(k, torch.float16)
        a = qh @ kh.transpose(-2, -1)
        if qk_dt in ("f16x3", "f16q2"):
            a = a + ql @ kh.transpose(-2, -1)
        if qk_dt in ("f16x3", "f16k2"):
            a = a + qh @ kl.transpose(-2, -1)
    else:
        a = (q @ k.transpose(-2, -1)) * scale
    a = a - a.amax(dim=-1, keepdim=True)
    p = a.exp()
    l = p.sum(-1, keepdim=True)
    if pv_dt in DT:
        p = rnd(p, DT[pv_dt])
        v = rnd(v, DT[pv_dt])
        o = p @ v
    elif pv_dt in ("f16x3_8", "f16v2_8"):
        ph, pl = split2(p, torch.float16)
        vh, vl = split2(v, torch.float16)
        o = ph @ vh + rnd8(ph) @ rnd8(vl)
        if pv_dt == "f16x3_8":
            o = o + rnd8(pl) @ rnd8(vh)
    elif pv_dt in ("f16x3", "f16p2", "f16v2"):   # round 6: P and / or V as hi + lo planes
        ph, pl = split2(p, torch.float16)
        vh, vl = split2(v, torch.float16)
        o = ph @ vh
        if pv_dt in ("f16x3", "f16p2"):
            o = o + pl @ vh
        if pv_dt in ("f16x3", "f16v2"):
            o = o + ph @ vl
    else:
        o = p @ v
    o = o / l
    o = o.transpose(1, 2).reshape(B, S, C)
    if ZONE.mode in DT or ZONE.mode.endswith("+store"):
        pass
    return O.F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


ATTN = {}


def run_design(views, sd, args, tr, hd, attn=None, roles=None, conv_roles=None):
    enc, dec, head = args
    ZONE.roles = dict(roles or {})
    ZONE.conv_roles = dict(conv_roles or {})
    ATTN.clear()
    if attn:
        ATTN.update(attn)
    saveF, saveA = O.F, O.attention
    O.F = make_proxy()
    O.attention = attention_emul
    orig_dpt = O.dpt_forward

    def dpt_zone(tokens4, *a, **k):
        prev = ZONE.mode
        ZONE.mode = hd
        ZONE.conv_idx = 0
        if hd != "f32":  # the hooks reach the head as 16-bit rows (they are GEMM operands) unless the head is exact
            pass
        try:
            return orig_dpt(tokens4, *a, **k)
        finally:
            ZONE.mode = prev
    O.dpt_forward = dpt_zone
    ZONE.mode = tr
    try:
        torch.manual_seed(1234)
        return O.forward(views, sd, enc, dec, head)
    finally:
        O.F, O.attention, O.dpt_forward = saveF, saveA, orig_dpt
        ZONE.mode = "f32"
        ZONE.roles = {}
        ZONE.conv_roles = {}


def main():
    from fast3r_amd import Fast3R
    args = tiny_args()
    shp = {k: tuple(v.shape) for k, v in Fast3R(*args).state_dict().items()}
    designs = [
        ("bf16", "bf16+store", None), ("bf16", "bf16", None), ("bf16", "f16", None), ("bf16", "f16x2", None),
        ("bf16", "f16x3", None), ("bf16", "bf16x3", None), ("bf16", "f32", None),
        ("f16", "f16+store", None), ("f16", "f16", None), ("f16", "f16x2", None), ("f16", "f16x3", None), ("f16", "f32", None),
        ("f32", "f16", None), ("f32", "bf16", None),
        ("bf16", "f16x3", {"qk": "f16"}), ("bf16", "f32", {"qk": "f16"}), ("bf16", "f32", {"qk": "f16", "pv": "f16"}),
        # the product's "high" mode and its variations: attention format per product, correction terms in fp8
        ("f16w2", "f16x3", {"qk": "f16", "pv": "f16"}), ("f16w2", "f16x3", {"qk": "f16", "pv": "bf16"}), ("f16w2", "f16x3", {"qk": "bf16", "pv": "f16"}),
        ("f16w2_8", "f16x3", {"qk": "f16", "pv": "f16"}), ("f16w2", "f16x3_8", {"qk": "f16", "pv": "f16"}), ("f16w2_8", "f16x3_8", {"qk": "f16", "pv": "f16"}),
    ]
    for dist in ("hot", "default"):
        sd = synth_state_dict(shp, 0, dist)
        views = make_views(3, 64, 64)
        torch.manual_seed(1234)
        ref = O.forward(views, sd, *args)
        for tr, hd, attn in designs:
            out = run_design(views, sd, args, tr, hd, attn)
            worst = {}
            for o, r in zip(out, ref):
                for k in r:
                    worst[k] = max(worst.get(k, 0.0), O.rel_l2(o[k], r[k]))
            print(f"{dist:8s} tr={tr:6s} hd={hd:11s} attn={attn}: " + "  ".join(f"{k.replace('pts3d_', 'p_')}={v:.2e}" for k, v in worst.items()), flush=True)


STUDIES = {
    # round 5.  Each entry: (weight distribution, [(label, transformer mode, head mode, attention formats, per-role overrides, per-conv overrides)])
    "heads": ("hot", [   # can the last convolutions of the DPT head (29 = head.0, 30 = head.2: 2/3 of the heads' FLOPs) carry fewer planes?  -> no
        ("x3 everywhere (the product)", "f16w2", "f16x3", None, None, {}),
        ("head.0 single fp16", "f16w2", "f16x3", None, None, {29: "f16"}), ("head.2 single fp16", "f16w2", "f16x3", None, None, {30: "f16"}),
        ("head.0 + head.2 single fp16", "f16w2", "f16x3", None, None, {29: "f16", 30: "f16"}),
        ("head.0 + head.2 activations split only", "f16w2", "f16x3", None, None, {29: "f16x2", 30: "f16x2"}),
        ("head.0 + head.2 weights split only", "f16w2", "f16x3", None, None, {29: "f16w2", 30: "f16w2"}),
        ("every conv: activations split only", "f16w2", "f16x3", None, None, {i: "f16x2" for i in range(32)})]),
    # round 6: the correction products of the heads' 3x3 convolutions from fp8 operands (f3r_gemm split "x3f8"): 11-13 / 14-18 / 19-23 /
    # 24-28 = refinenet4..1 (4 RCU convs + the 1x1 out_conv, which stays on fp16 planes), 29 = head.0, 30 = head.2
    "heads_f8": ("hot", [
        ("x3 everywhere (the product before)", "f16w2", "f16x3", None, None, {}),
        ("RCU convs + head.0 + head.2 with fp8 corrections", "f16w2", "f16x3", None, None,
         {i: "f16x3_8c" for i in (11, 12, 14, 15, 16, 17, 19, 20, 21, 22, 24, 25, 26, 27, 29, 30)}),
        ("head.0 + head.2 only", "f16w2", "f16x3", None, None, {29: "f16x3_8c", 30: "f16x3_8c"}),
        ("every 3x3 / 1x1 conv of the heads", "f16w2", "f16x3_8c", None, None, {})]),
    "fp8": ("hot", [     # the weight-correction product of the transformer's linear layers from fp8 operands (Fast3R.low_plane = "fp8")
        ("w2 everywhere (two fp16 planes)", "f16w2", "f16x3", None, None, {}), ("low plane in fp8 everywhere", "f16w2_8", "f16x3", None, None, {}),
        ("low plane in fp8 on qkv + fc1 + fc2 (the product)", "f16w2", "f16x3", None, {"qkv": "f16w2_8", "fc1": "f16w2_8", "fc2": "f16w2_8"}, {}),
        ("single fp16 plane (precision fast)", "f16", "f16x3", None, None, {})]),
    "heavy": ("heavy", [  # Student-t weights + LayerNorm gains in [0.2, 5]: which 16-bit rounding site is to blame?  -> every one of them alone
        ("the product: tr w2, attention fp16, heads x3", "f16w2", "f16x3", None, None, {}),
        ("attention operands exact, rest as the product", "f16w2", "f16x3", {"qk": "f32", "pv": "f32"}, None, {}),
        ("q k^T exact only", "f16w2", "f16x3", {"qk": "f32", "pv": "f16"}, None, {}), ("p v exact only", "f16w2", "f16x3", {"qk": "f16", "pv": "f32"}, None, {}),
        ("linear layers x3, attention fp16", "f16x3", "f16x3", None, None, {}), ("linear layers x3 AND attention exact", "f16x3", "f16x3", {"qk": "f32", "pv": "f32"}, None, {}),
        ("transformer exact, attention fp16, heads x3", "f32", "f16x3", None, None, {}), ("heads exact, rest as the product", "f16w2", "f32", None, None, {}),
        ("precision fast", "f16", "f16", None, None, {})]),
    # round 6: price a "robust" tier between high and exact.  Linear layers x3 (both operands as planes) is the known floor of the
    # transformer side; which attention operands must ALSO carry a low plane?  "f16x3" = hi + lo on both operands of the product (3 MFMA products),
    # "f16q2" / "f16k2" / "f16p2" / "f16v2" = a low plane on that one operand only (2 products)
    "robust": ("heavy", [
        ("linear x3, attention fp16 (r5 row)", "f16x3", "f16x3", None, None, {}),
        ("linear x3, QK^T 3 products, PV fp16", "f16x3", "f16x3", {"qk": "f16x3", "pv": "f16"}, None, {}),
        ("linear x3, QK^T fp16, PV 3 products", "f16x3", "f16x3", {"qk": "f16", "pv": "f16x3"}, None, {}),
        ("linear x3, QK^T 3 products, PV 3 products", "f16x3", "f16x3", {"qk": "f16x3", "pv": "f16x3"}, None, {}),
        ("linear x3, QK^T exact, PV fp16", "f16x3", "f16x3", {"qk": "f32", "pv": "f16"}, None, {}),
        ("linear x3, QK^T fp16, PV exact", "f16x3", "f16x3", {"qk": "f16", "pv": "f32"}, None, {}),
        ("linear x3, Q lo only + PV 3", "f16x3", "f16x3", {"qk": "f16q2", "pv": "f16x3"}, None, {}),
        ("linear x3, K lo only + PV 3", "f16x3", "f16x3", {"qk": "f16k2", "pv": "f16x3"}, None, {}),
        ("linear x3, QK 3 + P lo only", "f16x3", "f16x3", {"qk": "f16x3", "pv": "f16p2"}, None, {}),
        ("linear x3, QK 3 + V lo only", "f16x3", "f16x3", {"qk": "f16x3", "pv": "f16v2"}, None, {}),
        ("linear w2, QK 3 + PV 3", "f16w2", "f16x3", {"qk": "f16x3", "pv": "f16x3"}, None, {}),
        ("linear x2 (activations split, weights single), QK 3 + PV 3", "f16x2", "f16x3", {"qk": "f16x3", "pv": "f16x3"}, None, {}),
        ("linear x3 with fp8 corrections, QK 3 + PV 3", "f16x3_8", "f16x3", {"qk": "f16x3", "pv": "f16x3"}, None, {}),
        ("linear x3 on qkv + proj only (MLP w2), QK 3 + PV 3", "f16w2", "f16x3", {"qk": "f16x3", "pv": "f16x3"}, {"qkv": "f16x3", "proj": "f16x3"}, {}),
        ("linear x3 on fc1 + fc2 only (qkv, proj w2), QK 3 + PV 3", "f16w2", "f16x3", {"qk": "f16x3", "pv": "f16x3"}, {"fc1": "f16x3", "fc2": "f16x3"}, {}),
        ("linear x3 on qkv only, QK 3 + PV 3", "f16w2", "f16x3", {"qk": "f16x3", "pv": "f16x3"}, {"qkv": "f16x3"}, {}),
        ("linear x3_8, QK 3_8 + PV 3_8 (every correction in fp8)", "f16x3_8", "f16x3", {"qk": "f16x3_8", "pv": "f16x3_8"}, None, {}),
        ("linear x3_8, QK 3_8 + V lo in fp8 (no P lo)", "f16x3_8", "f16x3", {"qk": "f16x3_8", "pv": "f16v2_8"}, None, {}),
        ("linear x3, QK 3_8 + PV 3_8", "f16x3", "f16x3", {"qk": "f16x3_8", "pv": "f16x3_8"}, None, {}),
    ]),
    # the same question on the model the GPU test measures (--model vitl: ViT-L / ViT-L / 2 DPT heads, N = 3 views of 512^2; ~2 min per row on 8 cores)
    "robust_vitl": ("heavy", [
        ("the product (high): linear w2, attention fp16", "f16w2", "f16x3", None, None, {}),
        ("linear x3, attention fp16", "f16x3", "f16x3", None, None, {}),
        ("linear w2, QK 3 + PV 3", "f16w2", "f16x3", {"qk": "f16x3", "pv": "f16x3"}, None, {}),
        ("linear x3, QK 3, PV fp16", "f16x3", "f16x3", {"qk": "f16x3", "pv": "f16"}, None, {}),
        ("linear x3, QK 3 + V lo only", "f16x3", "f16x3", {"qk": "f16x3", "pv": "f16v2"}, None, {}),
        ("linear x3, QK 3 + PV 3", "f16x3", "f16x3", {"qk": "f16x3", "pv": "f16x3"}, None, {}),
        ("linear x3_8, QK 3_8 + PV 3_8 (every correction in fp8)", "f16x3_8", "f16x3", {"qk": "f16x3_8", "pv": "f16x3_8"}, None, {}),
        ("linear x3_8, QK 3_8 + V lo in fp8 (no P lo)", "f16x3_8", "f16x3", {"qk": "f16x3_8", "pv": "f16v2_8"}, None, {}),
        ("linear w2, QK 3_8 + PV 3_8", "f16w2", "f16x3", {"qk": "f16x3_8", "pv": "f16x3_8"}, None, {}),
        ("linear x2 (activations split, weights single), QK 3 + PV 3", "f16x2", "f16x3", {"qk": "f16x3", "pv": "f16x3"}, None, {}),
    ]),
}


def study(name):
    """python oracle/precision_study.py --study heads | fp8 | heavy  (the round-5 questions; results: profiles/r05_precision_study_*.txt)"""
    from fast3r_amd import Fast3R
    dist, cases = STUDIES[name]
    if name.endswith("_vitl"):
        from fast3r_amd.synthetic import vit_large_args
        args = vit_large_args(attn_implementation="pytorch_naive")
        size = 512
    else:
        args = tiny_args()
        size = 64
    shp = {k: tuple(v.shape) for k, v in Fast3R(*args).state_dict().items()}
    sd = synth_state_dict(shp, 0, dist)
    views = make_views(3, size, size)
    torch.manual_seed(1234)
    ref = O.forward(views, sd, *args)
    for label, tr, hd, attn, roles, convs in cases:
        out = run_design(views, sd, args, tr, hd, attn or {"qk": "f16", "pv": "f16"}, roles=roles, conv_roles=convs)
        worst = {}
        for o, r in zip(out, ref):
            for k in r:
                worst[k] = max(worst.get(k, 0.0), O.rel_l2(o[k], r[k]))
        print(f"{dist:8s} {label:52s} " + "  ".join(f"{k.replace('pts3d_', 'p_')}={v:.2e}" for k, v in worst.items()), flush=True)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--study", default="", choices=[""] + sorted(STUDIES))
    a = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        if a.study:
            study(a.study)
        else:
            main()
