"""CPU tests of the host side: state_dict layout, image-id RNG recipe, weight packing layouts (checked by emulating
the kernels' index maps with torch on CPU), collate / inference plumbing, error behaviour without a GPU."""
import copy
import time
import warnings

import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden
from fast3r_amd import Fast3R, MultiViewDUSt3RLitModule, inference, ops
from fast3r_amd._lib import F3RError
from fast3r_amd.dist import split_range
from fast3r_amd.inference_multiview import collate_with_cat, to_cpu
from fast3r_amd.synthetic import make_views, tiny_args, vit_large_args
from oracle.ref_loader import reference_available


def test_state_dict_layout_matches_golden_shapes():
    fix = load_golden("tiny_3x64")
    m = Fast3R(*tiny_args(**fix["tiny_kwargs"]))
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in fix["state_shapes"].items()}
    assert "decoder.image_idx_emb" not in sd  # non-persistent buffer (fast3r.py:691-697)
    s = m.downstream_head.dpt.scratch
    assert s.layer_rn[2].weight is s.layer3_rn.weight  # aliases, both key sets present (dpt_block.py:79-86)
    assert "downstream_head.dpt.scratch.layer_rn.2.weight" in sd and "downstream_head.dpt.scratch.layer3_rn.weight" in sd


def test_vit_large_param_count():
    with torch.device("meta"):  # shapes only: the real 647 M-parameter init costs a minute of CPU for nothing
        m = Fast3R(*vit_large_args())
    sd = m.state_dict()
    assert len(sd) == 720
    n_enc = sum(p.numel() for p in m.encoder.parameters())
    n_dec = sum(p.numel() for p in m.decoder.parameters())
    assert abs(n_enc / 1e6 - 303.10) < 0.05 and abs(n_dec / 1e6 - 303.36) < 0.05


@pytest.mark.skipif(not reference_available(), reason="reference checkout only exists in the build container")
def test_strict_load_of_reference_state_dict():
    from oracle.ref_loader import load_reference
    warnings.filterwarnings("ignore")
    R, _ = load_reference()
    enc, dec, head = tiny_args()
    ref = R(copy.deepcopy(enc), copy.deepcopy(dec), copy.deepcopy(head))
    m = Fast3R(enc, dec, head)
    res = m.load_state_dict(ref.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(m.decoder.image_idx_emb, ref.decoder.image_idx_emb)
    assert abs(m.decoder.attention_scale(False) - ref.decoder.dec_blocks[0].attn.attn_bias_scale) < 1e-12
    assert m.decoder.attention_scale(True) == ref.decoder.dec_blocks[0].attn.scale


def test_image_id_recipe_matches_reference_draw():
    fix = load_golden("tiny_3x64")
    m = Fast3R(*tiny_args(**fix["tiny_kwargs"]))
    torch.manual_seed(fix["rng_seed"])
    ids = m.decoder.draw_image_ids(fix["batch"], len(fix["shapes"]))
    assert torch.equal(ids, fix["image_ids"])
    # exactly one draw of the global RNG (fast3r.py:706)
    torch.manual_seed(5)
    m.decoder.draw_image_ids(1, 4)
    after = torch.rand(1)
    torch.manual_seed(5)
    torch.randint(0, 2 ** 32, (1,))
    assert torch.equal(after, torch.rand(1))
    m2 = Fast3R(*tiny_args(random_image_idx_embedding=False))
    assert m2.decoder.draw_image_ids(2, 3).tolist() == [[0, 1, 2], [0, 1, 2]]
    with pytest.raises(ValueError):
        m.decoder.draw_image_ids(1, 1001)  # the reference fails here too


def test_constructor_errors_match_reference_types():
    enc, dec, head = tiny_args()
    with pytest.raises(ValueError):
        Fast3R(dict(enc, encoder_type="nope"), dec, head)  # fast3r.py:85
    with pytest.raises(ValueError):
        Fast3R(enc, dict(dec, decoder_type="nope"), head)  # fast3r.py:98
    with pytest.raises(NotImplementedError):
        Fast3R(enc, dec, dict(head, head_type="linear"))  # fast3r.py:157
    with pytest.raises(ValueError):
        Fast3R(dict(enc, attn_implementation="nope"), dec, head)  # blocks.py:192


def test_the_scaling_ablation_decoder_builds():
    """configs/experiment/model_scaling/model_scaling_huge.yaml:13-15: fusion decoder 1280 wide, 16 heads (head_dim 80), depth 32 -- the
    reference's Attention takes any dim // num_heads (blocks.py:113-143); widths that are not a multiple of 16 are refused by name."""
    from fast3r_amd.fast3r import Fast3RDecoder
    dec = Fast3RDecoder(True, 1024, embed_dim=1280, num_heads=16, depth=2)  # (depth 32 in the config: the layer count is free)
    assert dec.embed_dim // dec.num_heads == 80 and abs(dec.attention_scale(True) - 80 ** -0.5) < 1e-12
    sd = dec.state_dict()
    assert sd["dec_blocks.0.attn.qkv.weight"].shape == (3 * 1280, 1280) and sd["decoder_embed.weight"].shape == (1280, 1024)
    enc, decargs, head = tiny_args(dec_embed_dim=320, dec_num_heads=4)
    m = Fast3R(enc, decargs, head)
    assert m.decoder.embed_dim // m.decoder.num_heads == 80
    with pytest.raises(ValueError, match="multiple of 16"):
        Fast3RDecoder(True, 128, embed_dim=200, num_heads=2, depth=12)


def test_no_cpu_fallback():
    m = Fast3R(*tiny_args()).eval()
    with pytest.raises(F3RError):
        m(make_views(2, 32, 32))
    with pytest.raises(F3RError):
        ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), 1e-6, torch.float16)
    lit = MultiViewDUSt3RLitModule.load_for_inference(m)
    assert not lit.training and lit.net is m
    with pytest.raises(F3RError):
        inference(make_views(2, 32, 32), lit, "cpu", "32", verbose=False)


def test_inference_dtype_argument_selects_the_operand_format():
    """fast3r_amd/inference_multiview.py::_operand_format against the reference's table (inference_multiview.py:41-52):
    "32" is the only true-fp32 spelling there -> precision "exact" here; torch.float32 is NOT fp32 there (default autocast dtype)."""
    import warnings
    from fast3r_amd.inference_multiview import _operand_format
    m = Fast3R(*tiny_args(), compute_dtype=torch.bfloat16, precision="fast")
    assert _operand_format("16-mixed", m) == (torch.float16, "fast") and _operand_format(torch.float16, m) == (torch.float16, "fast")
    assert _operand_format("bf16-mixed", m) == (torch.bfloat16, "fast") and _operand_format(torch.bfloat16, m) == (torch.bfloat16, "fast")
    assert _operand_format("32", m) == (torch.float16, "exact") and _operand_format(32, m) == (torch.float16, "exact")
    assert _operand_format(None, m) == (torch.bfloat16, "fast")  # anything else: the model's own format
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert _operand_format(torch.float32, m) == (torch.float16, "high")
        llama = Fast3R(*tiny_args(decoder_type="llama"))
        assert _operand_format("32", llama) == (torch.float16, "exact")  # since round 3 the fp32-equivalent mode covers the LlamaDecoder too
    with pytest.raises(ValueError, match="precision"):
        Fast3R(*tiny_args(), precision="fp32")


def test_split_range_is_contiguous_and_balanced():
    for n in (0, 1, 7, 8, 320, 1500):
        for w in (1, 2, 3, 8):
            rs = [split_range(n, w, r) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    assert [split_range(1500, 8, r)[1] - split_range(1500, 8, r)[0] for r in range(8)] == [188] * 4 + [187] * 4


def test_collate_and_to_cpu_structure():
    views = make_views(3, 32, 48)
    batch = collate_with_cat([tuple(views)])
    assert isinstance(batch, list) and len(batch) == 3 and batch[0]["img"].shape == (1, 3, 32, 48)
    assert batch[1]["true_shape"].tolist() == [[32, 48]]
    res = dict(views=batch, preds=[{"a": torch.ones(1, 2)} for _ in range(3)], loss=None)
    out = collate_with_cat([to_cpu(res)], lists=False)
    assert out["loss"] is None and len(out["preds"]) == 3 and out["preds"][0]["a"].shape == (1, 2)


# ---- packing layouts: emulate the kernels' index maps on CPU with torch and compare with the torch op they replace
def test_pack_conv3x3_layout():
    torch.manual_seed(0)
    co, ci, H, W = 8, 24, 5, 6
    w = torch.randn(co, ci, 3, 3)
    x = torch.randn(2, ci, H, W)
    wp = ops.pack_conv3x3_weight(w, torch.float32)  # fp32 "lowp" just to test the index map
    cpad = 64
    assert wp.shape == (co, 9 * cpad)
    xn = x.permute(0, 2, 3, 1)  # NHWC
    xp = F.pad(xn, (0, cpad - ci, 1, 1, 1, 1))
    cols = torch.stack([xp[:, dy:dy + H, dx:dx + W, :] for dy in range(3) for dx in range(3)], dim=3)  # (B,H,W,9,cpad)
    y = cols.reshape(2, H, W, 9 * cpad) @ wp.t()
    ref = F.conv2d(x, w, padding=1).permute(0, 2, 3, 1)
    assert torch.allclose(y, ref, atol=1e-4)


def test_pack_conv3x3_f8_layout_and_planes():
    """ops.pack_conv3x3_weight_f8 / ops.f8_planes (f3r.h F3R_SPLIT_X3F8): emulate on the CPU what the kernel sums -- A_hi W_hi + A_hi8 W_lo8 + A_lo8 W_hi8
    with the E8M0 scales applied -- from the packed bytes alone and compare with the fp32 convolution: fp32-class, and the fp8 rounding of the two
    correction products alone is what separates it from exact planes (dpt_block.py:133-154; oracle/precision_study.py mode f16x3_8c is the same sum)."""
    torch.manual_seed(0)
    co, ci, H, W = 16, 128, 6, 7
    w = torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5
    w[1] *= 1e-3
    w[2] *= 40.0
    x = torch.randn(2, H, W, ci) * 2.0
    x[0, 0, 0, :4] = torch.tensor([900.0, -500.0, 1e-5, 0.0])   # beyond e4m3's +-448 (clamped in the fp8 planes), and below its subnormals
    wp, sc = ops.pack_conv3x3_weight_f8(w)
    kp = 9 * ci
    assert wp.dtype == torch.float16 and wp.shape == (co, 2 * kp) and sc.dtype == torch.int32 and sc.shape == (co,)
    raw = wp.view(torch.uint8).view(co, 4 * kp)
    dec8 = lambda b: b.contiguous().view(torch.float8_e4m3fn).double()
    w_hi = raw[:, :2 * kp].contiguous().view(torch.float16).double()
    e_lo, e_hi = (sc.long() & 0xff).double(), ((sc.long() >> 8) & 0xff).double()
    w_lo8 = dec8(raw[:, 2 * kp:3 * kp]) * torch.exp2(e_lo - 127)[:, None]
    w_hi8 = dec8(raw[:, 3 * kp:]) * torch.exp2(e_hi - 127)[:, None]
    taps = w.permute(0, 2, 3, 1).reshape(co, kp).double()          # k = (ky*3 + kx) * Cin + ci in every plane
    assert torch.equal(w_hi, taps.to(torch.float16).double())
    # the scaled planes use e4m3's range: largest magnitude of every row in [112, 448], relative error of a plane element <= 2^-4 of the row's scale
    assert (raw[:, 2 * kp:3 * kp].view(torch.float8_e4m3fn).float().abs().amax(dim=1) >= 112).all()
    assert ((w_lo8 - (taps - w_hi)).abs().amax(dim=1) <= (taps - w_hi).abs().amax(dim=1) * 2.0 ** -4).all()
    assert ((w_hi8 - w_hi).abs().amax(dim=1) <= w_hi.abs().amax(dim=1) * 2.0 ** -4).all()
    p8 = ops.f8_planes(x)
    assert p8.dtype == torch.uint8 and p8.shape == (2, H, W, 2 * ci)
    a_hi = x.to(torch.float16).double()
    a_hi8, a_lo8 = dec8(p8[..., :ci]), dec8(p8[..., ci:]) / 4096.0
    assert float(a_hi8.abs().max()) == 448.0 and float((a_lo8 - (x.double() - a_hi).clamp(-448 / 4096, 448 / 4096)).abs().max()) <= 2.0 ** -4 * 448 / 4096

    def conv(a, wk):   # NHWC x [co][kp] -> NHWC, through the same im2col index map as test_pack_conv3x3_layout
        xp = F.pad(a, (0, 0, 1, 1, 1, 1))
        cols = torch.stack([xp[:, dy:dy + H, dx:dx + W, :] for dy in range(3) for dx in range(3)], dim=3)
        return cols.reshape(2, H, W, kp) @ wk.t()
    got = conv(a_hi, w_hi) + conv(a_hi8, w_lo8) + conv(a_lo8, w_hi8)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    far = torch.ones_like(ref, dtype=torch.bool)
    far[0, :2, :2] = False                                            # (the clamped inputs touch only their 3x3 neighbourhood)
    scale = float((ref.abs() * far).max())
    err = float(((got - ref).abs() * far).max()) / scale
    single = float(((conv(a_hi, w_hi) - ref).abs() * far).max()) / scale
    assert err < 3e-5 and single > 10 * err, (err, single)


def test_pack_convT_layout():
    torch.manual_seed(0)
    ci, co, s, h, w_ = 16, 8, 4, 3, 5
    wt = torch.randn(ci, co, s, s)
    b = torch.randn(co)
    x = torch.randn(2, ci, h, w_)
    wp, bt = ops.pack_convT_weight(wt, b, torch.float32)
    assert wp.shape == (s * s * co, 64) and bt.shape == (s * s * co,)
    y = x.permute(0, 2, 3, 1).reshape(-1, ci) @ wp[:, :ci].t() + bt  # rows (b,y,x), cols (dy,dx,co)
    y = y.view(2, h, w_, s, s, co).permute(0, 1, 3, 2, 4, 5).reshape(2, h * s, w_ * s, co)
    ref = F.conv_transpose2d(x, wt, b, stride=s).permute(0, 2, 3, 1)
    assert torch.allclose(y, ref, atol=1e-4)


def test_pack_linear_pads_k_to_64():
    w = torch.randn(12, 96)
    p = ops.pack_linear_weight(w, torch.float16)
    assert p.shape == (12, 128) and p.dtype == torch.float16
    assert torch.equal(p[:, :96], w.half()) and p[:, 96:].abs().sum() == 0


def test_rope_tables_match_reference_formula():
    cos, sin = ops.rope_tables(32, 100.0, "cpu")
    i = torch.arange(16).float()
    th = torch.arange(32).float()[:, None] * (100.0 ** (-i / 16))[None]
    assert cos.shape == (32, 16) and torch.allclose(cos, th.cos(), atol=1e-6) and torch.allclose(sin, th.sin(), atol=1e-6)


def test_hub_mixin_round_trip(tmp_path):
    """The reference loads the released model with Fast3R.from_pretrained (PyTorchModelHubMixin, fast3r.py:44-48): a local snapshot
    directory (config.json with the three *_args dicts + model.safetensors) must build the same module and load the same tensors."""
    import json
    from fast3r_amd import Fast3R
    from fast3r_amd.synthetic import synth_state_dict, tiny_args
    enc, dec, head = tiny_args()
    m = Fast3R(enc, dec, head).eval()
    m.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 3))
    m.save_pretrained(tmp_path)
    cfg = json.load(open(tmp_path / "config.json"))
    assert set(cfg) >= {"encoder_args", "decoder_args", "head_args"} and (tmp_path / "model.safetensors").exists()
    m2 = Fast3R.from_pretrained(tmp_path)
    assert m2.encoder_args == m.encoder_args and m2.decoder_args == m.decoder_args and m2.head_args == m.head_args
    assert list(m2.state_dict()) == list(m.state_dict())
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    # the two arguments the reference does not have travel with the snapshot: a model saved as bf16 / "fast" does not come back as the default
    m3 = Fast3R(enc, dec, head, compute_dtype=torch.bfloat16, precision="fast")
    m3.save_pretrained(tmp_path / "b")
    cfg = json.load(open(tmp_path / "b" / "config.json"))
    assert cfg["precision"] == "fast" and cfg["compute_dtype"] == "torch.bfloat16"
    m4 = Fast3R.from_pretrained(tmp_path / "b")
    assert m4.precision == "fast" and m4.compute_dtype == torch.bfloat16
    assert m2.precision == "high" and m2.compute_dtype == torch.float16  # the constructor default = the benchmarked format


def test_load_from_dust3r_checkpoint(tmp_path):
    """fast3r.py:162-234: encoder + first head of a DUSt3R checkpoint are mapped in; decoder / second head keys are ignored."""
    enc, dec, head = tiny_args()
    src = Fast3R(enc, dec, head)
    sd = src.state_dict()
    dust3r = {}
    for k, v in sd.items():
        if k.startswith("encoder."):
            dust3r[k[len("encoder."):]] = v + 1.0
        elif k.startswith("downstream_head."):
            dust3r[k.replace("downstream_head.", "downstream_head1.", 1)] = v - 1.0
    dust3r["dec_blocks.0.attn.qkv.weight"] = torch.zeros(3, 3)       # DUSt3R's own decoder: not ours
    dust3r["downstream_head2.dpt.head.4.bias"] = torch.zeros(4)
    path = tmp_path / "dust3r.pth"
    torch.save({"model": dust3r}, path)
    m = Fast3R(enc, dec, head)
    loaded, skipped = m.load_from_dust3r_checkpoint(str(path))
    assert skipped == {"dec_blocks.0.attn.qkv.weight", "downstream_head2.dpt.head.4.bias"}
    new = m.state_dict()
    for k, v in sd.items():
        if k.startswith("encoder."):
            assert torch.equal(new[k], v + 1.0), k
        elif k.startswith("downstream_head."):
            assert torch.equal(new[k], v - 1.0), k
    m2 = Fast3R(enc, dec, dict(head, skip_load_pretrained_head=True))
    before = {k: v.clone() for k, v in m2.downstream_head.state_dict().items()}
    m2.load_from_dust3r_checkpoint(str(path))
    assert all(torch.equal(v, m2.downstream_head.state_dict()[k]) for k, v in before.items())


def test_half_and_bfloat16_modules_still_pack():
    """SURVEY.md section 8b "Modes": model.half() / .bfloat16() must keep working -- the packed operands are rebuilt from whatever
    precision the parameters hold, biases / norm weights go back to fp32 for the epilogues."""
    for cast in ("half", "bfloat16"):
        m = Fast3R(*tiny_args()).eval()
        getattr(m, cast)()
        assert m._packed is None  # _apply invalidates the cache
        pk = m._pack(torch.device("cpu"))
        blk = pk["dec"][0]
        assert blk.qkv_w.dtype == m.compute_dtype and blk.qkv_b.dtype == torch.float32 and blk.n1w.dtype == torch.float32
        assert torch.isfinite(blk.qkv_w.float()).all() and pk["head"] is not None


def test_the_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under fast3r_amd/ imports it, importing the package (and building a model) loads no oracle
    module, and bench.py reaches it only inside its cpu_baseline leg."""
    import ast
    import glob
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        hits = []
        for node in ast.walk(ast.parse(open(path).read())):
            if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                hits.append(node.lineno)
            if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                hits.append(node.lineno)
        return hits
    for f in glob.glob(os.path.join(root, "fast3r_amd", "**", "*.py"), recursive=True):
        assert oracle_imports(f) == [], f
    code = ("import sys; import fast3r_amd; from fast3r_amd.synthetic import tiny_args; fast3r_amd.Fast3R(*tiny_args()); "
            "print([m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')])")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("[]"), r.stdout + r.stderr
    # bench.py: the only import of the oracle sits inside cpu_baseline()
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        inner = [n for n in ast.walk(fn) if isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle"]
        assert (len(inner) > 0) == (fn.name == "cpu_baseline"), fn.name
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n)]
    assert top == []


def test_a_missing_or_stale_library_fails_loudly(tmp_path, monkeypatch):
    """No silent fallback: without libf3r_hip.so (or with one built for an older ABI) the first kernel call raises F3RError."""
    from fast3r_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libf3r_hip.so"))
    with pytest.raises(F3RError, match="not found"):
        _lib.lib()
    monkeypatch.undo()
    real = _lib.lib()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", real.f3r_version() + 1)
    with pytest.raises(F3RError, match="rebuild"):
        _lib.lib()


def test_fp8_low_plane_weight_pack_on_cpu():
    """ops.pack_linear_weight_f8 (f3r.h F3R_SPLIT_W2F8): rows [K fp16 hi | K fp8 e4m3((W - hi) 2^s_n)] + one E8M0 word per output channel; decoding the
    planes recovers the weight ~16x closer than one fp16 plane, whatever the row's scale; no NaN code is ever produced"""
    from fast3r_amd import ops
    g = torch.Generator().manual_seed(0)
    w = torch.randn((512, 256), generator=g) * 0.05 * torch.exp2(torch.randint(-8, 4, (512, 1), generator=g).float())
    w[7] = 0.0   # an all-zero row must not break the scale search
    p, s = ops.pack_linear_weight_f8(w)
    assert p.shape == (512, 384) and p.dtype == torch.float16 and s.shape == (512,) and s.dtype == torch.int32
    raw = p.view(torch.uint8).view(512, 768)
    hi = raw[:, :512].contiguous().view(torch.float16).view(512, 256).double()
    lo_b = raw[:, 512:].contiguous()
    assert not ((lo_b & 0x7F) == 0x7F).any()
    e = (s & 0xFF).double() - 127.0
    assert torch.equal(s & 0xFF, (s >> 8) & 0xFF) and torch.equal(s & 0xFF, (s >> 24) & 0xFF)   # the byte in every position of the word
    rec = hi + lo_b.view(torch.float8_e4m3fn).double() * torch.exp2(e)[:, None]
    rows = w.abs().amax(1) > 0
    err8 = ((rec - w.double()).abs().amax(1) / w.abs().amax(1).clamp_min(1e-30))[rows]
    err1 = ((hi - w.double()).abs().amax(1) / w.abs().amax(1).clamp_min(1e-30))[rows]
    assert float(err8.max()) <= 2.0 ** -14 and float((err8 / err1.clamp_min(1e-12)).median()) <= 1 / 12
    assert torch.equal(rec[7], torch.zeros(256, dtype=torch.float64))
    with pytest.raises(AssertionError):
        ops.pack_linear_weight_f8(torch.randn(256, 192))   # K must be a multiple of 128


def test_bench_live_roofline_arithmetic():
    """bench.live_roofline turns the attention kernel's own counters (f3r_attn_args.dbg_counters, ABI 330) into utilisation / clock / implied rate:
    checked on counters constructed for a known answer (head_dim 64: 64 MFMAs of 32 cycles per 64-key tile and wave)"""
    import bench
    from fast3r_amd import _lib
    waves, tiles_per_wave = 4096, 5120
    cycles_per_wave = int(32 * 64 * tiles_per_wave / 0.75)        # utilisation 0.75 by construction
    ticks_per_wave = int(cycles_per_wave / 1.7e9 * 1e8)            # 1.7 GHz shader clock against the 100 MHz constant clock
    per_xcd = [[cycles_per_wave * waves // 8, ticks_per_wave * waves // 8 * (1.04 if x == 3 else 1.0), waves // 8] for x in range(8)]
    counters = [waves + 7, waves, tiles_per_wave * waves, cycles_per_wave * waves, ticks_per_wave * waves, per_xcd]
    saved = _lib.lib
    _lib.lib = lambda: type("L", (), {"f3r_wall_clock_khz": staticmethod(lambda: 100000)})()
    try:
        live = bench.live_roofline(counters, avg_launch_ms=300.0, achieved_tflops=1300.0, head_dim=64, power={"source": None})
    finally:
        _lib.lib = saved
    assert abs(live["mfma_util_cycles"] - 0.75) < 1e-3 and abs(live["effective_clock_ghz"] - 1.7) < 2e-3
    assert abs(live["implied_tflops"] - 0.75 * 1.7 * 256 * 4 * 1024 / 1e3) < 2.0
    assert abs(live["achieved_over_implied"] - 1300.0 / live["implied_tflops"]) < 1e-9
    assert abs(live["per_xcd"]["slowest_over_mean"] - 1.04 / (1 + 0.04 / 8)) < 1e-3 and live["per_xcd"]["waves"] == [waves // 8] * 8
    assert live["per_xcd"]["dealing"] == "static" and abs(live["per_xcd"]["busiest_over_mean"] - live["per_xcd"]["slowest_over_mean"]) < 1e-9
    # work stealing: the slow XCD (x = 3: 4 % longer waves) takes 4 % fewer items, the busy times are level
    for x in range(8):
        w = int(round(waves // 8 / (1.04 if x == 3 else 1.0)))
        per_xcd[x] = [cycles_per_wave * w, ticks_per_wave * w * (1.04 if x == 3 else 1.0), w]
    _lib.lib = lambda: type("L", (), {"f3r_wall_clock_khz": staticmethod(lambda: 100000)})()
    try:
        live = bench.live_roofline(counters, avg_launch_ms=300.0, achieved_tflops=1300.0, head_dim=64, power={"source": None})
    finally:
        _lib.lib = saved
    assert live["per_xcd"]["dealing"] == "work stealing" and live["per_xcd"]["busiest_over_mean"] < 1.003 < 1.03 < live["per_xcd"]["slowest_over_mean"]
    empty = bench.live_roofline([0, 0, 0, 0, 0, [[0, 0, 0]] * 8], 1.0, 1.0, 64, {"source": None})
    assert "note" in empty


def test_bench_power_sampler_never_raises_without_a_gpu():
    """the 2 Hz sampler of bench.py (amdsmi, else rocm-smi --json): on a box with neither it reports source None and an error text, and costs nothing"""
    import bench
    s = bench.PowerSampler(0, period=0.05)
    with s:
        time.sleep(0.2)
    out = s.summary()
    assert set(out) >= {"source", "samples", "power_w_mean", "sclk_mhz_mean", "error"}
    assert out["samples"] == 0 or out["source"] in ("amdsmi", "rocm-smi")


def test_magic_number_division_is_exact_under_the_bound_the_launchers_check():
    """The kernels divide by multiplying with ceil(2^32 / d) and keeping the high word (attention work items -> (query block, head, batch):
    f3r_attn_asm.hip takes the work-stealing form only while n_work * nxy < 2^32; GEMM tile map: f3r_gemm_asm.hip tile_map_exact; RoPE token ->
    (y, x) in the q | k epilogue).  The claim behind those checks: for d >= 2 and a * d < 2^32, (a * ceil(2^32 / d)) >> 32 == a // d -- and just
    above the bound it fails, so the bound is not decoration."""
    from hypothesis import given, settings, strategies as st

    def magic(d):
        return -(-(1 << 32) // d)

    @settings(max_examples=3000, deadline=None)
    @given(st.integers(2, 1 << 20), st.data())
    def exact(d, data):
        a = data.draw(st.integers(0, ((1 << 32) - 1) // d))
        assert a * d < (1 << 32) and magic(d) < (1 << 32)
        assert (a * magic(d)) >> 32 == a // d
    exact()
    # edge of the bound, dense: every a for a few divisors, incl. the largest a the bound admits
    for d in (2, 3, 5, 7, 40, 320, 640, 1023, 1025, 65535):
        top = ((1 << 32) - 1) // d
        for a in list(range(0, 2000)) + list(range(max(0, top - 2000), top + 1)):
            assert (a * magic(d)) >> 32 == a // d, (a, d)
    # beyond it the trick does break: d = 3, a = 2^31 (a * d = 1.5 x 2^32) comes out one too high
    assert ((1 << 31) * magic(3)) >> 32 == (1 << 31) // 3 + 1


def test_uncalibrated_checkpoint_warning_logic():
    """round 6: `inference()` warns ONCE when a LOADED checkpoint runs a 16-bit operand tier that Fast3R.calibrate_precision() never measured on
    those weights (the 1e-3 parity of fp16 / "high" is a statement about default-init-like weights: DESIGN.md section 3); a freshly constructed
    model, the fp32-equivalent mode and a calibrated model stay silent; new weights make the calibration stale."""
    import warnings
    from fast3r_amd import inference_multiview as im
    m = Fast3R(*tiny_args())
    assert m.weights_loaded is False and m.calibration is None and "robust" in m.PRECISION_TIERS
    assert Fast3R(*tiny_args(), precision="robust").precision == "robust"

    def warned(net):
        im._warned_uncalibrated = False
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            im._warn_if_uncalibrated(net)
        return any("calibrate_precision" in str(x.message) for x in w)
    assert not warned(m)                                   # default init: nothing was loaded
    m.load_state_dict(m.state_dict())
    assert m.weights_loaded and warned(m)                  # a checkpoint, never calibrated
    assert not im._warned_uncalibrated is False and not _second_warning(im, m)   # ... once per process
    m.precision = "exact"
    assert not warned(m)                                   # the fp32-equivalent mode needs no calibration
    m.precision = "high"
    m.calibration = dict(recommended="high", params_version=m._params_version())
    assert m.precision_is_calibrated() and not warned(m)
    m.calibration = dict(recommended="robust", params_version=m._params_version())
    assert not m.precision_is_calibrated() and warned(m)   # calibrated, but running a cheaper tier than recommended
    m.precision = "robust"
    assert m.precision_is_calibrated()
    with torch.no_grad():
        next(m.parameters()).add_(1.0)                     # the weights changed: the report is about other weights
    assert not m.precision_is_calibrated()
    m.load_state_dict(m.state_dict())
    assert m.calibration is None


def _second_warning(im, net):
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        im._warn_if_uncalibrated(net)
    return len(w) > 0
