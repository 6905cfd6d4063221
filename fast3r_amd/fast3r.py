"""Fast3R on MI355X: the reference's model API (fast3r/models/fast3r.py) executed by hand-written HIP kernels.

Drop-in surface kept from the reference (file:line = /root/reference/fast3r/models/fast3r.py):
  Fast3R(encoder_args, decoder_args, head_args, freeze="none")       :50-70
  Fast3R.forward(views, profiling=False) -> list[dict] | (list, dict) :302-497
  Fast3R.set_max_parallel_views_for_head(int)                         :298-300
  attributes encoder_args / decoder_args / head_args, .encoder / .decoder / .downstream_head[_local]
  state_dict(): identical keys and shapes, so `load_state_dict(ref.state_dict(), strict=True)`
  works, including the duplicated `scratch.layer_rn.{i}` aliases (croco/models/dpt_block.py:79-86).

What is different by design: torch.nn modules are used ONLY as parameter containers (names, shapes, device moves);
none of their forward() methods is ever called.  All arithmetic goes through fast3r_amd.ops -> libf3r_hip.so.  The
residual stream, LayerNorm statistics, softmax and post-processing are fp32; GEMM / conv / attention operands are
16-bit (`compute_dtype`, fp16 by default, bf16 selectable) with fp32 MFMA accumulation.  There is no CPU path.
"""
import math
import time
from collections import OrderedDict
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import F3RError
from .dist import ViewSharding, split_range


# ======================================================================================= parameter containers
class _Params(nn.Module):
    """A module that only owns parameters; calling it is a bug (the compute lives in the HIP engine)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise F3RError("fast3r_amd parameter containers are not callable; use Fast3R.forward")


class _Attention(_Params):
    def __init__(self, dim, qkv_bias=True):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)  # blocks.py:125
        self.proj = nn.Linear(dim, dim)                    # blocks.py:128


class _Mlp(_Params):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)  # blocks.py:94
        self.fc2 = nn.Linear(hidden, dim)  # blocks.py:97


class _Block(_Params):
    def __init__(self, dim, mlp_ratio, eps, qkv_bias=True):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)  # blocks.py:214
        self.attn = _Attention(dim, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)  # blocks.py:227
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _PatchEmbed(_Params):
    def __init__(self, patch_size, dim):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(3, dim, kernel_size=patch_size, stride=patch_size)  # blocks.py:412-414
        self.norm = nn.Identity()


class CroCoEncoder(_Params):
    """fast3r.py:499-559.  RoPE-2D (freq from 'RoPE<freq>'), LayerNorm eps 1e-6."""

    def __init__(self, img_size=512, patch_size=16, patch_embed_cls="ManyAR_PatchEmbed", embed_dim=768, num_heads=12,
                 depth=12, mlp_ratio=4, pos_embed="RoPE100", attn_implementation="pytorch_naive"):
        super().__init__()
        assert patch_embed_cls in ["PatchEmbedDust3R", "ManyAR_PatchEmbed"]  # patch_embed.py:19
        if not pos_embed.startswith("RoPE"):
            raise NotImplementedError("Unknown pos_embed " + pos_embed)  # fast3r.py:533
        if attn_implementation not in ("pytorch_naive", "flash_attention", "pytorch_auto"):
            raise ValueError(f"Unknown attn_implementation: {attn_implementation}")  # blocks.py:192
        if embed_dim % num_heads != 0 or embed_dim // num_heads != 64:
            raise ValueError("fast3r_amd kernels are built for head_dim 64 (ViT-B/L/H family)")
        self.patch_embed_cls = patch_embed_cls
        self.patch_size, self.embed_dim, self.num_heads, self.depth = patch_size, embed_dim, num_heads, depth
        self.pos_embed = pos_embed
        self.rope_freq = float(pos_embed[len("RoPE"):])
        self.patch_embed = _PatchEmbed(patch_size, embed_dim)
        self.enc_blocks = nn.ModuleList([_Block(embed_dim, mlp_ratio, 1e-6) for _ in range(depth)])
        self.enc_norm = nn.LayerNorm(embed_dim, eps=1e-6)


class _LayerScale(_Params):
    def __init__(self, dim, init_values=1.0):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class _DinoBlock(_Params):
    """DINOv2 `Block` (dinov2/layers/block.py): x + ls1(attn(norm1(x))); x + ls2(mlp(norm2(x)))."""

    def __init__(self, dim, mlp_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, qkv_bias=True)
        self.ls1 = _LayerScale(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.ls2 = _LayerScale(dim)


class _DinoViT(_Params):
    """Parameter layout of DINOv2's `DinoVisionTransformer` as torch.hub's `dinov2_vitl14` builds it (facebookresearch/dinov2
    models/vision_transformer.py: img_size 518, patch 14, no register tokens, LayerScale, MLP ffn, block_chunks = 0): the keys under
    `encoder.model.` are the hub checkpoint's own."""

    def __init__(self, embed_dim, depth, num_heads, mlp_ratio, patch_size, pos_grid):
        super().__init__()
        self.embed_dim, self.num_heads, self.patch_size, self.pos_grid = embed_dim, num_heads, patch_size, pos_grid
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + pos_grid * pos_grid, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))  # in the checkpoint; unused at inference
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        self.patch_embed = _PatchEmbed(patch_size, embed_dim)
        self.blocks = nn.ModuleList([_DinoBlock(embed_dim, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)


class DinoEncoder(_Params):
    """fast3r.py:561-651: DINOv2 ViT-L/14 patch tokens (`forward_features(...)['x_norm_patchtokens']`), portrait samples encoded upright
    and their tokens put back in the stored (landscape) order.  The reference builds the backbone with torch.hub.load (network); here
    the same architecture is built locally (random init; a checkpoint's `encoder.model.*` keys load as they are).  The size arguments
    exist only so that tests can build a small one: the reference class is always ViT-L/14."""

    def __init__(self, patch_size=14, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, pos_grid=37, **kwargs):
        super().__init__()
        assert patch_size == 14, "DINOv2 model must have patch size 14"  # fast3r.py:570
        if embed_dim // num_heads != 64:
            raise ValueError("fast3r_amd kernels are built for head_dim 64")
        self.patch_size, self.embed_dim, self.num_heads, self.depth = patch_size, embed_dim, num_heads, depth
        self.patch_embed_cls = "dino"
        self.model = _DinoViT(embed_dim, depth, num_heads, mlp_ratio, patch_size, pos_grid)


def sincos_1d_table(embed_dim, n_pos):
    """get_1d_sincos_pos_embed_from_grid (croco/models/pos_embed.py:58-76): [sin | cos], float64 -> float32."""
    omega = np.arange(embed_dim // 2, dtype=float)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", np.arange(n_pos).reshape(-1).astype(float), omega)
    return torch.from_numpy(np.concatenate([np.sin(out), np.cos(out)], axis=1)).float()


class Fast3RDecoder(_Params):
    """fast3r.py:654-808.  No RoPE; additive image-index embedding; block LayerNorm eps 1e-5, dec_norm 1e-6."""

    def __init__(self, random_image_idx_embedding, enc_embed_dim, embed_dim=768, num_heads=12, depth=12, mlp_ratio=4.0,
                 qkv_bias=True, drop=0.0, attn_drop=0.0, attn_implementation="pytorch_naive",
                 attn_bias_for_inference_enabled=True, max_image_idx=1000):
        super().__init__()
        if attn_implementation not in ("pytorch_naive", "flash_attention", "pytorch_auto"):
            raise ValueError(f"Unknown attn_implementation: {attn_implementation}")
        hd = embed_dim // num_heads
        if embed_dim % num_heads != 0 or hd % 16 != 0 or not 16 <= hd <= 128:
            # the reference takes any dim // num_heads (blocks.py:113-143); 64 runs the tuned attention kernels, the other multiples of
            # 16 up to 128 (model_scaling_huge.yaml: 1280 / 16 = 80) the generic one (f3r_attn_generic.hip)
            raise ValueError(f"fast3r_amd attention kernels are built for head_dim = a multiple of 16 up to 128 (got {embed_dim} / {num_heads})")
        if embed_dim % 64 != 0:
            raise ValueError(f"fast3r_amd: the fused QKV epilogue splits q / k / v on 64-column groups: embed_dim must be a multiple of 64 (got {embed_dim})")
        self.embed_dim, self.num_heads, self.depth = embed_dim, num_heads, depth
        self.random_image_idx_embedding = random_image_idx_embedding
        self.attn_bias_for_inference_enabled = attn_bias_for_inference_enabled
        self.decoder_embed = nn.Linear(enc_embed_dim, embed_dim, bias=True)
        self.dec_blocks = nn.ModuleList([_Block(embed_dim, mlp_ratio, 1e-5, qkv_bias) for _ in range(depth)])
        # The reference table has 1000 rows (fast3r.py:691-697) and therefore fails for N > 1000 views
        # (SURVEY.md section 0.7).  Same formula, more rows when asked for: ids < 1000 are bit-identical.
        self.register_buffer("image_idx_emb", sincos_1d_table(embed_dim, max_image_idx), persistent=False)
        self.dec_norm = nn.LayerNorm(embed_dim, eps=1e-6)

    def attention_scale(self, training: bool) -> float:
        """blocks.py:116-124,151-154."""
        hd = self.embed_dim // self.num_heads
        if (not training) and self.attn_bias_for_inference_enabled:
            return hd ** -0.5 * (1.0 * math.log(137) / math.log(20)) ** 0.5
        return hd ** -0.5

    def draw_image_ids(self, batch_size, num_views, rank=0):
        """fast3r.py:702-743 (_generate_per_rank_generator + _get_random_image_pos), or 0..N-1 (fast3r.py:339-348,794-796).
        Consumes exactly one value of the global torch CPU RNG when random ids are on, like the reference."""
        if not self.random_image_idx_embedding:
            return torch.arange(num_views)[None].repeat(batch_size, 1)
        max_image_idx = self.image_idx_emb.shape[0] - 1
        if num_views - 1 > max_image_idx:
            raise ValueError(f"{num_views} views need an image-index table of at least {num_views} rows "
                             f"(have {max_image_idx + 1}); build the decoder with max_image_idx >= {num_views}")
        seed = torch.randint(0, 2 ** 32, (1,)).item()
        g = torch.Generator()
        g.manual_seed(seed + rank)
        ids = torch.zeros(batch_size, num_views, dtype=torch.long)
        for b in range(batch_size):
            ids[b, 1:] = torch.randperm(max_image_idx, generator=g)[: num_views - 1] + 1
        return ids


class _RMSNorm(_Params):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))  # llama.py:150-153


class _LlamaAttention(_Params):
    def __init__(self, dim, n_heads, n_kv_heads):
        super().__init__()
        hd = dim // n_heads
        self.wq = nn.Linear(dim, n_heads * hd, bias=False)     # llama.py:195-198
        self.wk = nn.Linear(dim, n_kv_heads * hd, bias=False)
        self.wv = nn.Linear(dim, n_kv_heads * hd, bias=False)
        self.wo = nn.Linear(n_heads * hd, dim, bias=False)


class _LlamaFeedForward(_Params):
    def __init__(self, dim, hidden_dim, multiple_of, ffn_dim_multiplier):
        super().__init__()
        hidden_dim = int(2 * hidden_dim / 3)                   # llama.py:273-277
        if ffn_dim_multiplier is not None:
            hidden_dim = int(ffn_dim_multiplier * hidden_dim)
        hidden_dim = multiple_of * ((hidden_dim + multiple_of - 1) // multiple_of)
        self.w1 = nn.Linear(dim, hidden_dim, bias=False)
        self.w2 = nn.Linear(hidden_dim, dim, bias=False)
        self.w3 = nn.Linear(dim, hidden_dim, bias=False)


class _LlamaBlock(_Params):
    def __init__(self, dim, n_heads, n_kv_heads, multiple_of, ffn_dim_multiplier, norm_eps):
        super().__init__()
        self.attention = _LlamaAttention(dim, n_heads, n_kv_heads)                       # llama.py:323-326
        self.feed_forward = _LlamaFeedForward(dim, 4 * dim, multiple_of, ffn_dim_multiplier)  # llama.py:327-332
        self.attention_norm = _RMSNorm(dim, norm_eps)
        self.ffn_norm = _RMSNorm(dim, norm_eps)


class LlamaDecoder(_Params):
    """fast3r.py:810-968 (the `llama_dec` experiment, configs/experiment/llama_dec/llama_dec.yaml): pre-norm RMSNorm blocks with SwiGLU,
    bias-free projections, rotary embedding of q / k by the IMAGE id of a token's view (all patches of a view share one angle set), a
    learnable embedding added to the tokens of view 0 before every layer, final RMSNorm.  Bidirectional (the released config) or
    causal attention; grouped-query attention with any n_kv_heads that divides n_heads (incl. 1: multi-query); head_dim must be 64."""

    def __init__(self, random_image_idx_embedding, enc_embed_dim, embed_dim=4096, n_layers=32, n_heads=32, n_kv_heads=None,
                 multiple_of=256, ffn_dim_multiplier=None, norm_eps=1e-5, rope_theta=10000, max_seq_len=1000, is_causal=False,
                 depth_init=True, **kwargs):
        super().__init__()
        if embed_dim % n_heads != 0 or embed_dim // n_heads != 64:
            raise ValueError("fast3r_amd kernels are built for head_dim 64")
        n_kv_heads = n_heads if n_kv_heads is None else int(n_kv_heads)
        if n_heads % n_kv_heads != 0:
            raise ValueError(f"n_heads ({n_heads}) must be a multiple of n_kv_heads ({n_kv_heads})")  # repeat_kv, llama.py:125-134,196
        self.embed_dim, self.num_heads, self.depth = embed_dim, n_heads, n_layers
        self.n_kv_heads, self.is_causal = n_kv_heads, bool(is_causal)
        self.random_image_idx_embedding = random_image_idx_embedding
        self.rope_theta, self.norm_eps = rope_theta, norm_eps
        self.view0_embed = nn.Parameter(torch.zeros(embed_dim))                          # fast3r.py:841-842
        nn.init.normal_(self.view0_embed, mean=0.0, std=0.02)
        self.decoder_embed = nn.Linear(enc_embed_dim, embed_dim, bias=True)              # :845
        self.layers = nn.ModuleList([_LlamaBlock(embed_dim, n_heads, n_kv_heads, multiple_of, ffn_dim_multiplier, norm_eps)
                                     for _ in range(n_layers)])                          # :848-852
        self.norm = _RMSNorm(embed_dim, norm_eps)                                        # :854
        # precompute_freqs_cis (llama.py:41-60) as [cos (32) | sin (32)] per position instead of complex64; a plain attribute, not a
        # buffer, like the reference's precomputed_freqs_cis (fast3r.py:837)
        hd = embed_dim // n_heads
        freqs = 1.0 / (rope_theta ** (torch.arange(0, hd, 2)[: hd // 2].float() / hd))
        ang = torch.outer(torch.arange(max_seq_len).float(), freqs).float()
        self.image_idx_emb = torch.cat([ang.cos(), ang.sin()], dim=1)

    def attention_scale(self, training: bool) -> float:
        return (self.embed_dim // self.num_heads) ** -0.5  # F.scaled_dot_product_attention default (llama.py:239)

    draw_image_ids = Fast3RDecoder.draw_image_ids  # same RNG recipe (fast3r.py:856-897 == :702-743)


# within a 64-wide head: destination position -> source dim, so that the reference's complex pairs (2j, 2j+1) land where the QKV
# epilogue rotates (rope_mode 1: dims [0,32) pair i with i+16 using table columns 0-15, dims [32,64) likewise with columns 16-31).
# The same permutation on q and k leaves every q . k unchanged.
_ROPE_PERM = [2 * j for j in range(16)] + [2 * j + 1 for j in range(16)] + [2 * j for j in range(16, 32)] + [2 * j + 1 for j in range(16, 32)]


class _RCU(_Params):
    def __init__(self, f):
        super().__init__()
        self.conv1 = nn.Conv2d(f, f, 3, padding=1)  # dpt_block.py:105-123
        self.conv2 = nn.Conv2d(f, f, 3, padding=1)


class _Fusion(_Params):
    def __init__(self, f):
        super().__init__()
        self.out_conv = nn.Conv2d(f, f, 1)  # dpt_block.py:180-188
        self.resConfUnit1 = _RCU(f)
        self.resConfUnit2 = _RCU(f)


class _DPT(_Params):
    """Parameter layout of DPTOutputAdapter_fix (heads/dpt_head.py:28-40; croco/models/dpt_block.py:315-490)."""

    def __init__(self, num_channels, feature_dim, last_dim, hooks, dim_tokens, patch_size, layer_dims=(96, 192, 384, 768)):
        super().__init__()
        self.hooks, self.patch_size, self.num_channels = hooks, patch_size, num_channels
        self.feature_dim, self.last_dim, self.layer_dims = feature_dim, last_dim, list(layer_dims)
        ld = layer_dims
        scratch = _Params()
        for i in range(4):
            setattr(scratch, f"layer{i + 1}_rn", nn.Conv2d(ld[i], feature_dim, 3, padding=1, bias=False))
        scratch.layer_rn = nn.ModuleList([getattr(scratch, f"layer{i + 1}_rn") for i in range(4)])  # aliases
        for i in range(1, 5):
            setattr(scratch, f"refinenet{i}", _Fusion(feature_dim))
        self.scratch = scratch
        self.head = nn.Sequential(nn.Conv2d(feature_dim, feature_dim // 2, 3, padding=1), nn.Identity(),
                                  nn.Conv2d(feature_dim // 2, last_dim, 3, padding=1), nn.Identity(),
                                  nn.Conv2d(last_dim, num_channels, 1))
        self.act_postprocess = nn.ModuleList([
            nn.Sequential(nn.Conv2d(dim_tokens[0], ld[0], 1), nn.ConvTranspose2d(ld[0], ld[0], 4, stride=4)),
            nn.Sequential(nn.Conv2d(dim_tokens[1], ld[1], 1), nn.ConvTranspose2d(ld[1], ld[1], 2, stride=2)),
            nn.Sequential(nn.Conv2d(dim_tokens[2], ld[2], 1)),
            nn.Sequential(nn.Conv2d(dim_tokens[3], ld[3], 1), nn.Conv2d(ld[3], ld[3], 3, stride=2, padding=1)),
        ])


class PixelwiseTaskWithDPT(_Params):
    """heads/dpt_head.py:93-129 (parameters under `.dpt`)."""

    def __init__(self, *, hooks_idx, dim_tokens, num_channels, feature_dim, last_dim, patch_size, depth_mode, conf_mode):
        super().__init__()
        self.depth_mode, self.conf_mode = depth_mode, conf_mode
        self.dpt = _DPT(num_channels, feature_dim, last_dim, hooks_idx, dim_tokens, patch_size)


# ======================================================================================= packed (device) weights
class _PackedBlock:
    __slots__ = ("n1w", "n1b", "n2w", "n2b", "eps", "qkv_w", "qkv_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                 "rms", "rope_mode", "swiglu_hidden", "q_dim", "kv_dim", "kv_group", "causal", "head_dim", "fc1_split",
                 "fc1_w8", "fc1_ws", "fc2_w8", "fc2_ws",   # the MLP weights with their low plane in fp8 (Fast3R.low_plane = "fp8"), or None
                 "qk_w8", "qk_ws", "v_w2")                 # ... and the q | k rows of the QKV weight; its v rows as two fp16 planes

    def __init__(self):
        self.fc1_w8 = self.fc1_ws = self.fc2_w8 = self.fc2_ws = self.qk_w8 = self.qk_ws = self.v_w2 = None
        self.rms, self.rope_mode, self.swiglu_hidden = False, 0, 0
        self.fc1_split = None  # None: like every other projection of the block
        self.head_dim = 64  # the Fast3R fusion decoder may have another width (model_scaling_huge.yaml: 80)
        self.q_dim, self.kv_dim, self.kv_group, self.causal = 0, None, 1, False  # grouped-query / causal attention (LlamaDecoder only)


def _f32(t):
    return None if t is None else t.detach().float().contiguous()


def _pack_block(blk: _Block, lp, split=False, head_dim=64, fc1_split=None, f8=False):
    p = _PackedBlock()
    p.head_dim = head_dim
    p.n1w, p.n1b, p.n2w, p.n2b = _f32(blk.norm1.weight), _f32(blk.norm1.bias), _f32(blk.norm2.weight), _f32(blk.norm2.bias)
    p.eps = blk.norm1.eps
    p.qkv_w, p.qkv_b = ops.pack_linear_weight(blk.attn.qkv.weight.detach().float(), lp, split), _f32(blk.attn.qkv.bias)
    p.proj_w, p.proj_b = ops.pack_linear_weight(blk.attn.proj.weight.detach().float(), lp, split), _f32(blk.attn.proj.bias)
    # fc1_split False = fc1 single-plane inside "high" (Fast3R.high_fc1_planes = False): measured +3.5 % views/s at N = 100 for TWICE the
    # distance to the fp32 path on the real-size stress model (4.8e-4 / 6.1e-4 against 2.4e-4 / 2.8e-4), so it is an experiment knob, not
    # the default (oracle/precision_study.py per-role run, DESIGN.md section 3 (Precision modes))
    p.fc1_split = bool(split) if fc1_split is None else bool(fc1_split)
    p.fc1_w, p.fc1_b = ops.pack_linear_weight(blk.mlp.fc1.weight.detach().float(), lp, p.fc1_split), _f32(blk.mlp.fc1.bias)
    p.fc2_w, p.fc2_b = ops.pack_linear_weight(blk.mlp.fc2.weight.detach().float(), lp, split), _f32(blk.mlp.fc2.bias)
    # Fast3R.low_plane = "fp8" (precision "high", fp16): fc1's weight a second time as rows [K fp16 hi | K fp8 low plane] + one scale per output
    # channel (f3r.h F3R_SPLIT_W2F8); the block uses it whenever its token count is a multiple of 256, the two-fp16-plane pack otherwise
    w1 = blk.mlp.fc1.weight.detach().float()
    w2_ = blk.mlp.fc2.weight.detach().float()
    if f8 and split and lp == torch.float16 and p.fc1_split and w1.shape[1] % 128 == 0 and w1.shape[0] % 256 == 0 and w2_.shape[0] % 256 == 0:
        p.fc1_w8, p.fc1_ws = ops.pack_linear_weight_f8(w1)
        p.fc2_w8, p.fc2_ws = ops.pack_linear_weight_f8(w2_)
    wq = blk.attn.qkv.weight.detach().float()
    Dm = wq.shape[0] // 3
    if f8 and split and lp == torch.float16 and wq.shape[1] % 128 == 0 and Dm % 256 == 0 and head_dim == 64:
        p.qk_w8, p.qk_ws = ops.pack_linear_weight_f8(wq[:2 * Dm])
        p.v_w2 = ops.pack_linear_weight(wq[2 * Dm:], lp, True)
    return p


def _pack_dino_block(blk: _DinoBlock, lp, split=False):
    """DINOv2 block -> the packed fields of a ViT block.  LayerScale is folded into the projection that precedes it:
    gamma * (W a + b) = (gamma[:, None] * W) a + gamma * b -- exact in real arithmetic, no epilogue change."""
    p = _PackedBlock()
    p.n1w, p.n1b, p.n2w, p.n2b = _f32(blk.norm1.weight), _f32(blk.norm1.bias), _f32(blk.norm2.weight), _f32(blk.norm2.bias)
    p.eps = blk.norm1.eps
    g1, g2 = blk.ls1.gamma.detach().float(), blk.ls2.gamma.detach().float()
    p.qkv_w, p.qkv_b = ops.pack_linear_weight(blk.attn.qkv.weight.detach().float(), lp, split), _f32(blk.attn.qkv.bias)
    p.proj_w = ops.pack_linear_weight(g1[:, None] * blk.attn.proj.weight.detach().float(), lp, split)
    p.proj_b = (g1 * blk.attn.proj.bias.detach().float()).contiguous()
    p.fc1_w, p.fc1_b = ops.pack_linear_weight(blk.mlp.fc1.weight.detach().float(), lp, split), _f32(blk.mlp.fc1.bias)
    p.fc2_w = ops.pack_linear_weight(g2[:, None] * blk.mlp.fc2.weight.detach().float(), lp, split)
    p.fc2_b = (g2 * blk.mlp.fc2.bias.detach().float()).contiguous()
    return p


def _pack_llama_block(blk: _LlamaBlock, n_heads, lp, split=False, n_kv_heads=None, causal=False):
    """LlamaDecoder layer -> the same packed fields as a ViT block: [wq; wk; wv] as one QKV matrix (q / k rows permuted per head, see
    _ROPE_PERM), [w1; w3] stacked for one up-projection GEMM, no biases, RMSNorm weights."""
    p = _PackedBlock()
    p.rms, p.rope_mode = True, 1
    p.n1w, p.n1b, p.n2w, p.n2b = _f32(blk.attention_norm.weight), None, _f32(blk.ffn_norm.weight), None
    p.eps = blk.attention_norm.eps
    n_kv_heads = n_heads if n_kv_heads is None else n_kv_heads
    perm = torch.tensor([h * 64 + d for h in range(n_heads) for d in _ROPE_PERM])
    perm_kv = torch.tensor([h * 64 + d for h in range(n_kv_heads) for d in _ROPE_PERM])
    wq, wk, wv = (m.weight.detach().float() for m in (blk.attention.wq, blk.attention.wk, blk.attention.wv))
    p.qkv_w, p.qkv_b = ops.pack_linear_weight(torch.cat([wq[perm], wk[perm_kv], wv], dim=0), lp, split), None
    if n_kv_heads != n_heads:
        p.q_dim, p.kv_dim, p.kv_group = n_heads * 64, n_kv_heads * 64, n_heads // n_kv_heads
    p.causal = bool(causal)
    p.proj_w, p.proj_b = ops.pack_linear_weight(blk.attention.wo.weight.detach().float(), lp, split), None
    w1, w3 = blk.feed_forward.w1.weight.detach().float(), blk.feed_forward.w3.weight.detach().float()
    p.swiglu_hidden = w1.shape[0]
    p.fc1_w, p.fc1_b = ops.pack_linear_weight(torch.cat([w1, w3], dim=0), lp, split), None
    p.fc2_w, p.fc2_b = ops.pack_linear_weight(blk.feed_forward.w2.weight.detach().float(), lp, split), None
    return p


class _PackedHead:
    pass


class _ConvW:
    """A packed 3x3 convolution weight: `w` for split None / "x3" (ops.pack_conv3x3_weight) and, when the head runs its correction products on the
    fp8 MFMA (Fast3R.head_corrections = "fp8") and Cin % 128 == 0, `w8` / `scale` for split "x3f8" (ops.pack_conv3x3_weight_f8)."""
    __slots__ = ("w", "w8", "scale")

    def __init__(self, w, w8=None, scale=None):
        self.w, self.w8, self.scale = w, w8, scale


def _pack_head(head: PixelwiseTaskWithDPT, lp, split=False, f8=False):
    d = head.dpt
    h = _PackedHead()
    ap = d.act_postprocess
    lin = lambda w: ops.pack_linear_weight(w.detach().float(), lp, split)

    def c33(w):
        w = w.detach().float()
        if f8 and split and lp == torch.float16 and w.shape[1] % 128 == 0:
            return _ConvW(ops.pack_conv3x3_weight(w, lp, split), *ops.pack_conv3x3_weight_f8(w))
        return _ConvW(ops.pack_conv3x3_weight(w, lp, split))
    h.a_w = [lin(ap[i][0].weight) for i in range(4)]
    h.a_b = [_f32(ap[i][0].bias) for i in range(4)]
    # The transposed convolutions write their output with the channel count rounded up to 64 (layer_dims[0] = 96 -> 128 zero-weight / zero-bias
    # channels, matched by zero input channels in layer1_rn's weight: exact): the 3x3 convolution behind them then has a channel count the
    # 256-tile kernel's LDS-DMA staging takes (a ragged one ran on the 128-tile kernel through registers: 13.8 of 126 ms of head convolutions at N = 100).
    def convT_padded(m):
        w, b = m.weight.detach().float(), m.bias.detach().float()
        pad = ops.round_up(w.shape[1], 64) - w.shape[1]
        if pad:
            w, b = nn.functional.pad(w, (0, 0, 0, 0, 0, pad)), nn.functional.pad(b, (0, pad))
        return ops.pack_convT_weight(w, b, lp, split) + (w.shape[1],)

    def rn(i):
        w = getattr(d.scratch, f"layer{i + 1}_rn").weight.detach().float()
        pad = ops.round_up(w.shape[1], 64) - w.shape[1] if i < 2 else 0   # (levels 0 / 1 read the transposed convolutions' padded outputs)
        return c33(nn.functional.pad(w, (0, 0, 0, 0, 0, pad)) if pad else w)
    h.t0_w, h.t0_b, h.t0_cout = convT_padded(ap[0][1])
    h.t1_w, h.t1_b, h.t1_cout = convT_padded(ap[1][1])
    h.c3_w, h.c3_b = c33(ap[3][1].weight), _f32(ap[3][1].bias)
    h.rn_w = [rn(i) for i in range(4)]
    h.ref = []
    for i in range(1, 5):
        r = getattr(d.scratch, f"refinenet{i}")
        h.ref.append(dict(
            u1c1=(c33(r.resConfUnit1.conv1.weight), _f32(r.resConfUnit1.conv1.bias)),
            u1c2=(c33(r.resConfUnit1.conv2.weight), _f32(r.resConfUnit1.conv2.bias)),
            u2c1=(c33(r.resConfUnit2.conv1.weight), _f32(r.resConfUnit2.conv1.bias)),
            u2c2=(c33(r.resConfUnit2.conv2.weight), _f32(r.resConfUnit2.conv2.bias)),
            out=(lin(r.out_conv.weight), _f32(r.out_conv.bias))))
    h.h0_w, h.h0_b = c33(d.head[0].weight), _f32(d.head[0].bias)
    h.h2_w, h.h2_b = c33(d.head[2].weight), _f32(d.head[2].bias)
    h.h4_w = d.head[4].weight.detach().float().reshape(d.num_channels, -1).contiguous()
    h.h4_b = _f32(d.head[4].bias)
    h.dims = d.layer_dims
    h.feature_dim, h.last_dim, h.num_channels, h.patch_size = d.feature_dim, d.last_dim, d.num_channels, d.patch_size
    h.depth_mode, h.conf_mode = head.depth_mode, head.conf_mode
    h.fin = None
    if d.last_dim == 128 and h.h4_w.shape[1] == 128:  # the fused tail (f3r_gemm_args.fin_w): head[4] as a zero-padded [4][128] matrix
        try:
            h.fin = ops.dpt_fin_args(h.h4_w, h.h4_b, h.conf_mode, tuple(h.depth_mode))
        except (ValueError, AssertionError):
            h.fin = None  # an unsupported mode raises where the reference raises: in the head's own call (ops.dpt_final)
    return h


# ======================================================================================= the model
class _GraphCache:
    """Captured forwards by scene shape (see Fast3R.enable_graphs); least-recently-used shapes are dropped beyond `max_entries`
    (every entry pins a private memory pool the size of a forward's activations)."""

    def __init__(self):
        self.max_views = 64
        self.max_entries = 8
        self.entries = OrderedDict()
        self.seen = set()

    def clear(self):
        self.entries.clear()
        self.seen.clear()

    def run(self, model, views, host_outputs=False):
        imgs = [v["img"] for v in views]
        dev = imgs[0].device
        shapes = {tuple(i.shape) for i in imgs}
        if len(views) > self.max_views or len(shapes) != 1 or any(i.dtype != torch.float32 for i in imgs):
            return None
        if model._orientation_plan(views)["any_portrait"]:  # also the eager path's argument checks (utils/misc.py:69), done on the host
            return None
        # (the operand-format knobs are part of the key: a graph captured with another low_plane / high_fc1_planes would keep replaying the old
        # packed weights after the model re-packed)
        key = (len(views), tuple(imgs[0].shape), str(dev), model.compute_dtype, model.precision, model.max_parallel_views_for_head,
               model.training, model._params_version(), model.low_plane, model.high_fc1_planes, model.head_corrections,
               model.head_f8_min_rows, model.head_tail_fused_min_rows)
        dec = model.decoder
        B = imgs[0].shape[0]
        if key not in self.entries:
            if key not in self.seen:  # first sight of a shape: plain eager forward (also warms the packed weights / RoPE tables)
                self.seen.add(key)
                return None
            static_imgs = [torch.empty_like(i) for i in imgs]
            static_emb = torch.zeros((B, len(views), dec.embed_dim), dtype=torch.float32, device=dev)
            static_views = [{"img": t} for t in static_imgs]
            for t, i in zip(static_imgs, imgs):
                t.copy_(i)
            graph = torch.cuda.CUDAGraph()
            sched = torch.zeros(2, dtype=torch.int32, device=dev)   # this graph's own work-stealing pair (ops.sched_scope), allocated outside the capture
            torch.cuda.synchronize()
            with torch.cuda.graph(graph), ops.sched_scope(sched):
                outs = model._forward_eager(static_views, False, _emb_rows=static_emb)
            self.entries[key] = (graph, static_imgs, static_emb, outs, sched)
            while len(self.entries) > self.max_entries:
                self.entries.popitem(last=False)
        self.entries.move_to_end(key)
        graph, static_imgs, static_emb, outs, _sched = self.entries[key]
        for t, i in zip(static_imgs, imgs):
            t.copy_(i, non_blocking=True)
        ids = dec.draw_image_ids(B, len(views))
        static_emb.copy_(dec.image_idx_emb.to(dev)[ids.to(dev)])
        graph.replay()
        if host_outputs:  # inference(): the replayed (static) outputs go to the host through the same sink as the eager path's head chunks
            sink = _HostSink(dev, B)
            sink.push([(i, slice(None), k, v) for i, r in enumerate(outs) for k, v in r.items()], [])
            sink.finish()   # before the next replay may overwrite the static outputs
            return [{k: sink.result(i, k) for k in r} for i, r in enumerate(outs)]
        return [{k: v.clone() for k, v in r.items()} for r in outs]


try:  # the reference's loading API (fast3r.py:44-48): Fast3R.from_pretrained(repo id or local snapshot dir) / save_pretrained
    import huggingface_hub as _hf
    _HubMixin = _hf.PyTorchModelHubMixin
except Exception:  # pragma: no cover  (huggingface_hub is a requirement of the reference; without it only from_pretrained is missing)
    class _HubMixin:
        def __init_subclass__(cls, **kw):
            super().__init_subclass__()


class _HostSink:
    """Device -> host leg of `inference()` (fast3r/dust3r/inference_multiview.py:92-93 + utils/device.py:17-53 `to_cpu`): one pinned host
    tensor per (view, output) -- torch's caching host allocator recycles them between calls, and a tensor the caller still holds is
    never handed out again -- filled by non_blocking copies on a side stream that waits for the event recorded behind the head chunk that
    produced the data.  The copies of chunk c overlap the heads of chunk c + 1; finish() waits for the last one.

    The returned tensors are PAGE-LOCKED for as long as the caller holds them (~8.4 MB per view: 2.7 GB at N = 320).  PINNED_LIMIT_BYTES
    bounds what one call locks (default 8 GiB; 0 = never pin); past the limit, or when the host refuses a page-locked allocation, the
    remaining outputs land in ordinary pageable tensors (the copy is then synchronous, as the reference's `to_cpu` is) -- never an error."""
    PINNED_LIMIT_BYTES = 8 << 30

    def __init__(self, dev, batch):
        self.dev, self.B = dev, batch
        self.stream = torch.cuda.Stream(device=dev)
        self.bufs = {}
        self.pinned_bytes = 0
        self.pageable = 0   # outputs that fell back to pageable memory

    def _alloc(self, shape, dtype):
        nbytes = int(torch.empty((), dtype=dtype).element_size())
        for s_ in shape:
            nbytes *= int(s_)
        if self.pinned_bytes + nbytes <= self.PINNED_LIMIT_BYTES:
            try:
                t = torch.empty(shape, dtype=dtype, pin_memory=True)
                self.pinned_bytes += nbytes
                return t
            except RuntimeError:   # hipHostMalloc refused (locked-memory limit of the host / container): from here on, pageable
                self.PINNED_LIMIT_BYTES = 0
        self.pageable += 1
        return torch.empty(shape, dtype=dtype)

    def push(self, items, keep_alive):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        for t in keep_alive:
            if t is not None:
                t.record_stream(self.stream)  # the caching allocator must not recycle the chunk's tensors before the side stream has read them
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            for i, b, name, src in items:
                key = (i, name)
                if key not in self.bufs:
                    shape = tuple(src.shape) if isinstance(b, slice) else (self.B,) + tuple(src.shape)
                    self.bufs[key] = self._alloc(shape, src.dtype)
                self.bufs[key][b].copy_(src, non_blocking=True)   # (a pageable destination makes this copy synchronous: still correct)

    def result(self, i, name):
        return self.bufs[(i, name)]

    def finish(self):
        self.stream.synchronize()


# compute_dtype travels through config.json as text ("torch.float16"): the constructor accepts the string back
_DTYPE_CODER = {torch.dtype: (lambda d: str(d), lambda s_: getattr(torch, str(s_).replace("torch.", "")))}


class Fast3R(nn.Module, _HubMixin, repo_url="https://github.com/facebookresearch/fast3r", tags=["image-to-3d"], coders=_DTYPE_CODER):
    """`Fast3R.from_pretrained("jedyang97/Fast3R_ViT_Large_512")` (or a local directory holding config.json + model.safetensors) works as
    in the reference: the mixin reads the three *_args dicts from config.json, builds the model and loads the state dict (identical keys).

    Two arguments the reference does not have (both optional):
      compute_dtype   the 16-bit MFMA operand type (torch.float16 default, torch.bfloat16);
      precision       default "high" with fp16 operands: the one format that is within 1e-3 rel-L2 of the fp32 reference on EVERY
                      fixture, the stress fixture included (DESIGN.md section 3 (Precision modes)) -- what bench.py measures;
                      "fast": every GEMM / conv operand is ONE 16-bit number (within 1e-3 on default-init weights only).
                      "high": split-precision operands (f3r.h f3r_split) -- transformer weights as hi + lo planes (2 MFMA passes per
                      GEMM), both operands of the DPT heads as hi + lo planes (3 passes, activations kept as two planes in HBM);
                      attention unchanged.  With fp16 this brings the stress fixture inside 1e-3 of the fp32 reference (DESIGN.md
                      section 4, oracle/precision_study.py) at ~1.15x the time of "fast" at N=320.
                      "exact": what `inference(dtype="32")` means in the reference (no autocast): BOTH operands of every GEMM / conv as
                      hi + lo planes (~22 significand bits each, 3 MFMA passes), the attention core in plain fp32 (f3r_attn_f32).  A
                      validation mode for scenes of tens of views; CroCo / DINOv2 encoder + Fast3R decoder on one GPU only."""

    def __init__(self, encoder_args: dict, decoder_args: dict, head_args: dict, freeze="none",
                 compute_dtype: torch.dtype = torch.float16, precision: str = "high"):
        super().__init__()
        if isinstance(compute_dtype, str):  # config.json round trip stores the dtype as text
            compute_dtype = getattr(torch, compute_dtype.replace("torch.", ""))
        if precision not in ("fast", "high", "robust", "exact"):
            raise ValueError(f"precision must be 'fast', 'high', 'robust' or 'exact', got {precision!r}")
        self.encoder_args = dict(encoder_args)
        self.build_encoder(encoder_args)
        self.decoder_args = dict(decoder_args)
        self.build_decoder(decoder_args)
        self.head_args = dict(head_args)
        self.build_head(head_args)
        self.max_parallel_views_for_head = 25  # fast3r.py:68
        self.max_parallel_views_for_encoder = 128  # the reference chunks at 400 (fast3r.py:250); bounds the workspace
        self.compute_dtype = compute_dtype
        self.precision = precision
        self.sharding = None  # set by shard_views(): view-sharded multi-GPU execution (fast3r_amd/dist.py)
        self.debug_taps = None  # set to a dict to capture the lowp DPT inputs (hooks 0, L/2, 3L/4, L) per sample
        self.kv_tap = None  # callable(k, vt): sees every fusion layer's K [T][D] / V^T [D][ld] of an UNSHARDED forward (tests)
        self.use_graphs = False  # enable_graphs(): hipGraph replay of small scenes
        self._graphs = _GraphCache()
        self._packed = None
        self._rope_cache = {}
        self.weights_loaded = False   # load_state_dict / from_pretrained sets it: a checkpoint, not this constructor's default init
        self.calibration = None       # calibrate_precision()'s report: which precision tier THESE weights need for 1e-3
        self.set_freeze(freeze)

    # ---------------------------------------------------------------- construction (fast3r.py:72-157)
    def build_encoder(self, encoder_args):
        if encoder_args["encoder_type"] == "croco":
            a = deepcopy(dict(encoder_args))
            a.pop("encoder_type")
            self.encoder = CroCoEncoder(**a)
        elif encoder_args["encoder_type"] == "dino_v2":
            a = deepcopy(dict(encoder_args))
            a.pop("encoder_type")
            self.encoder = DinoEncoder(**a)  # fast3r.py:80-83
        else:
            raise ValueError(f"Unsupported encoder type: {encoder_args['encoder_type']}")

    def build_decoder(self, decoder_args):
        dt = decoder_args.get("decoder_type", "fast3r")
        self.decoder_args["decoder_type"] = dt
        if dt == "fast3r":
            a = deepcopy(dict(decoder_args))
            a.pop("decoder_type", None)
            self.decoder = Fast3RDecoder(**a)
        elif dt == "llama":
            a = deepcopy(dict(decoder_args))
            a.pop("decoder_type", None)
            self.decoder = LlamaDecoder(**a)
        else:
            raise ValueError(f"Unsupported decoder type: {dt}")

    def build_head(self, head_args):
        self.output_mode, self.head_type = head_args["output_mode"], head_args["head_type"]
        self.depth_mode, self.conf_mode = head_args["depth_mode"], head_args["conf_mode"]
        mk = lambda: self.head_factory(head_args["head_type"], head_args["output_mode"], has_conf=bool(head_args["conf_mode"]),
                                       patch_size=head_args["patch_size"])
        self.downstream_head = mk()
        self.downstream_head_local = mk() if head_args.get("with_local_head", False) else None
        # transpose_to_landscape(head, activate=landscape_only) (fast3r.py:123-130, dust3r/utils/misc.py:61-106): see _orientation_plan
        self.landscape_only = bool(head_args.get("landscape_only", False))

    # `.head` / `.local_head` of the reference are closures over the two heads (utils/misc.py:61-106), not modules:
    # expose the names without registering the heads a second time (state_dict keys must not change).
    @property
    def head(self):
        return self.downstream_head

    @property
    def local_head(self):
        return self.downstream_head_local

    def head_factory(self, head_type, output_mode, has_conf=False, patch_size=16):
        if head_type == "dpt" and output_mode == "pts3d":
            assert self.decoder_args["depth"] > 9  # fast3r.py:137
            l2 = self.decoder_args["depth"]
            ed, dd = self.encoder_args["embed_dim"], self.decoder_args["embed_dim"]
            if tuple(self.head_args["depth_mode"])[0] not in ops.DEPTH_MODES:
                raise ValueError(f"bad mode={tuple(self.head_args['depth_mode'])[0]!r}")  # postprocess.py:51 (raised at the first forward there)
            if self.head_args["conf_mode"] and tuple(self.head_args["conf_mode"])[0] not in ops.CONF_MODES:
                raise ValueError(f"bad mode={tuple(self.head_args['conf_mode'])[0]!r}")   # :64
            return PixelwiseTaskWithDPT(num_channels=3 + has_conf, feature_dim=256, last_dim=128,
                                        hooks_idx=[0, l2 * 2 // 4, l2 * 3 // 4, l2], dim_tokens=[ed, dd, dd, dd],
                                        patch_size=patch_size, depth_mode=self.head_args["depth_mode"],
                                        conf_mode=self.head_args["conf_mode"])
        raise NotImplementedError(f"unexpected {head_type=} and {output_mode=}")  # fast3r.py:157

    def load_from_dust3r_checkpoint(self, dust3r_checkpoint_path):
        """fast3r.py:162-234: initialise from a DUSt3R checkpoint ({'model': state_dict}): `patch_embed.* | enc_blocks.* | enc_norm.*`
        go to `encoder.*`, `downstream_head1.*` to `downstream_head.*` (skipped with head_args['skip_load_pretrained_head'], and
        reverted if the shapes do not fit); everything else in the file is ignored.  Returns (loaded_keys, not_loaded_keys)."""
        checkpoint = torch.load(dust3r_checkpoint_path, weights_only=False)["model"]
        enc_sd, head_sd, loaded = {}, {}, set()
        for key, value in checkpoint.items():
            if key.startswith(("patch_embed", "enc_blocks", "enc_norm")):
                enc_sd["encoder." + key] = value
                loaded.add(key)
            elif key.startswith("downstream_head1"):
                head_sd[key.replace("downstream_head1", "downstream_head")] = value
                loaded.add(key)
        res = self.load_state_dict(enc_sd, strict=False)
        bad = set(res.unexpected_keys)
        loaded -= {k[len("encoder."):] for k in bad}
        if not self.head_args.get("skip_load_pretrained_head", False):
            backup = {k: v.clone() for k, v in self.downstream_head.state_dict().items()}
            try:
                res = self.load_state_dict(head_sd, strict=False)
                loaded -= {k.replace("downstream_head", "downstream_head1", 1) for k in res.unexpected_keys}
            except RuntimeError:  # size mismatch: keep the head as it was (:213-222)
                self.downstream_head.load_state_dict(backup)
                loaded -= {k for k in checkpoint if k.startswith("downstream_head1")}
        else:
            loaded -= {k for k in checkpoint if k.startswith("downstream_head1")}
        return loaded, set(checkpoint.keys()) - loaded

    def set_freeze(self, freeze):
        self.freeze = freeze
        todo = {"none": [], "encoder": [self.encoder], "sandwich": [self.encoder, self.downstream_head]}[freeze]
        for m in todo:
            for p in m.parameters():
                p.requires_grad = False

    def set_max_parallel_views_for_head(self, n):
        self.max_parallel_views_for_head = n

    def shard_views(self, process_group=None, exchange="allgather", p2p_channels=3, reserve_cus=32):
        """Enable the view-sharded multi-GPU path: this rank encodes / decodes / regresses only its contiguous range of
        views and exchanges K / V^T per fusion layer over RCCL (fast3r_amd/dist.py).  exchange: "allgather" (one collective per tensor
        and layer, one remote attention launch), "p2p" (pairwise rounds dealt onto p2p_channels communicators, one remote launch per
        arrived shard) or "auto" (three fusion layers with each on the first forward, then the one that exposed less)."""
        self.sharding = ViewSharding(process_group, exchange=exchange, p2p_channels=p2p_channels, reserve_cus=reserve_cus)
        return self

    def emulate_rank(self, rank, world, kv_source=None, exchange="allgather", reserve_cus=0):
        """ONE GPU runs exactly rank `rank`'s share of a `world`-rank view-sharded forward -- its views through the encoder, the fusion
        layers as local launch (parking the softmax state) + remote launch over world - 1 K / V^T segments, its heads -- with no
        collective (dist.EmulatedSharding).  kv_source fills the remote segments per layer (parity test); without it they hold random
        operands (bench.py --emulate-rank: a per-rank step time, clearly not a multi-GPU measurement).  `emulate_rank(None, 0)` undoes it."""
        from .dist import EmulatedSharding
        self.sharding = None if rank is None else EmulatedSharding(world, rank, kv_source, exchange=exchange)
        if self.sharding is not None:
            self.sharding.reserve_cus = int(reserve_cus)   # what ViewSharding.reserve_cus costs the local launch, measurable on one GPU
        return self

    # ---------------------------------------------------------------- packed weights
    def load_state_dict(self, state_dict, strict=True, assign=False):
        self.invalidate_packed_weights()
        self.weights_loaded, self.calibration = True, None
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    # ---------------------------------------------------------------- which precision tier do these weights need?
    PRECISION_TIERS = ("fast", "high", "robust", "exact")   # cheapest first (N = 320: 8.3 / 8.9 / ~19 / 36 s on one MI355X)

    @torch.no_grad()
    def calibrate_precision(self, views, tol=1e-3, tiers=("fast", "high", "robust"), max_views=8, seed=1234):
        """Measure, on up to `max_views` of the caller's own views, how far every 16-bit operand tier is from the fp32-equivalent mode
        (precision="exact", 3e-7 of the reference on its golden outputs) WITH THE WEIGHTS THE MODEL HOLDS NOW, and say which tier to use.

        Why: the distance of a 16-bit tier to the fp32 reference depends on the checkpoint.  Default-init-like and N(0, 1 / fan_in) weights keep
        "high" (the default) within 3e-4; a noise-amplifying set (heavy-tailed weights, LayerNorm gains spread over a decade) measured 2.5e-3 with
        "high" and 3.6e-4 with "robust" (DESIGN.md section 3).  Nobody can test the released `Fast3R.from_pretrained("jedyang97/Fast3R_ViT_Large_512")`
        weights offline, so the model measures itself: ~2 s for 8 views of 512 x 512 at ViT-L size (the exact pass is most of it).

        Returns (and keeps in `self.calibration`) {"per_tier": {tier: {output: rel-L2 vs exact}}, "worst": {tier: max}, "recommended": the cheapest
        tier within `tol` (or "exact"), "tol", "n_views", ...}.  Does not change `self.precision`: set it from the answer, or pass
        `inference(..., dtype=...)` as before.  Reference context: fast3r/croco/models/blocks.py:158-190 (bf16 autocast there), README.md:84."""
        views = list(views)[:max(1, int(max_views))]
        if not views:
            raise ValueError("calibrate_precision needs at least one view")
        if self.sharding is not None:
            raise ValueError("calibrate_precision runs on one GPU: call it before shard_views()")
        dev = next(self.parameters()).device
        views = [dict(v, img=v["img"].to(dev)) for v in views]
        saved = (self.precision, self.use_graphs)
        rng = torch.get_rng_state()
        per_tier, ms = {}, {}
        try:
            self.use_graphs = False

            def run(p):
                self.precision = p
                torch.manual_seed(seed)   # the image ids consume the global CPU RNG (fast3r.py:702-743): the same ids for every tier
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                out = self._forward_eager(views)
                torch.cuda.synchronize(dev)
                ms[p] = (time.perf_counter() - t0) * 1e3
                return out
            ref = run("exact")
            for p in tiers:
                if p not in self.PRECISION_TIERS[:-1]:
                    raise ValueError(f"unknown precision tier {p!r}")
                out = run(p)
                worst = {}
                for o, r in zip(out, ref):
                    for k in r:
                        e = float((o[k].double() - r[k].double()).norm() / r[k].double().norm().clamp_min(1e-30))
                        worst[k] = max(worst.get(k, 0.0), e)
                per_tier[p] = worst
                del out
            del ref
        finally:
            self.precision, self.use_graphs = saved
            torch.set_rng_state(rng)
        worst = {p: max(v.values()) for p, v in per_tier.items()}
        order = [p for p in self.PRECISION_TIERS if p in worst]
        ok = [p for p in order if worst[p] <= tol and math.isfinite(worst[p])]
        self.calibration = dict(per_tier=per_tier, worst=worst, recommended=ok[0] if ok else "exact", tol=tol, n_views=len(views),
                                ms_incl_weight_packing=ms, compute_dtype=str(self.compute_dtype), params_version=self._params_version(),
                                checker="precision='exact' on the same views and weights")
        return self.calibration

    def precision_is_calibrated(self):
        """True when calibrate_precision() ran on the weights the model holds now and the tier in use is at least the recommended one."""
        c = self.calibration
        if c is None or c.get("params_version") != self._params_version():
            return False
        t = self.PRECISION_TIERS
        return t.index(self.precision) >= t.index(c["recommended"])

    def _apply(self, fn, *a, **k):
        self.invalidate_packed_weights()
        return super()._apply(fn, *a, **k)

    def invalidate_packed_weights(self):
        self._packed = self._packed_alt = None
        if getattr(self, "_graphs", None) is not None:
            self._graphs.clear()  # captured graphs hold pointers into the packed weights

    def _params_version(self):
        """Sum of the in-place version counters of all parameters: `p.copy_` under no_grad, a submodule's own load_state_dict or an
        optimizer step change it, so packed (device, 16-bit) weights and captured graphs built from older values are never reused.
        Edits made through `p.data` (`p.data.copy_`, `p.data.add_`) do NOT bump the counter (`.data` is a detached alias with its own
        version): call `invalidate_packed_weights()` after them."""
        return sum(p._version for p in self.parameters())

    @property
    def _hp(self):
        """split (hi + lo) weight planes and head activations: "high", "robust" and "exact"."""
        return self.precision in ("high", "robust", "exact")

    @property
    def _x3(self):
        """both operands of every GEMM / conv as hi + lo planes (three MFMA products): "robust" and "exact"."""
        return self.precision in ("robust", "exact")

    # "fp8" (the default; precision "high" with fp16 operands): the correction products A W_lo of the transformer's MLPs (fc1, fc2) run on the
    # block-scaled fp8 MFMA (f3r.h F3R_SPLIT_W2F8: 1.4x the matrix-pipe rate of a second fp16 plane; LayerNorm and fc1's GELU epilogue write the
    # fp8 copies beside their fp16 outputs) for every pass whose token count is a multiple of 256; "fp16": two fp16 planes everywhere.  Measured
    # on ViT-L stress weights (profiles/r05_parity_low_plane_fp8_vs_fp16_vit_large_hot.txt): the same distance to the fp32 path (2.34e-4 vs
    # 2.39e-4 at N = 3, 2.76e-4 both at N = 100), +0.7 % views/s at N = 320, +1.7 % at N = 100, +3.9 % fusion-only at N = 20.  Changing it
    # needs invalidate_packed_weights().
    low_plane = "fp8"
    # precision "robust": where the two correction products of the fusion attention's scores (q_lo k_hi + q_hi k_lo) run -- "fp8": the block-scaled
    # fp8 MFMA (f3r_attn_asm_qk3f8_f16: 256 matrix-pipe cycles per score block; the fp8 copies move the softmax by ~2e-5), "fp16": two more fp16
    # products (f3r_attn_asm_qk3_f16: 384 cycles).  Read per forward: no re-packing needed.
    robust_corrections = "fp8"
    # precision "robust": the ENCODER's per-view attention (1024 keys; 0.6 % of the step's attention FLOPs at N = 320) -- "fp32": the fp32 attention of
    # precision "exact" (0.8 s of a 14.2 s step on the FMA pipe), "planes": the same three-product kernel as the fusion layers, one launch over the
    # batch of views (59 ms).  Measured on the heavy-tailed ViT-L set (N = 3 vs the CPU oracle): 7.8e-5 with "fp32", 5.2e-4 with "planes" -- both
    # inside 1e-3, the default keeps the larger margin (a tier people turn to because their checkpoint amplifies noise); N = 320: 14.2 s / 13.7 s.
    robust_encoder_attention = "fp32"
    # DPT heads in precision "high" with fp16 planes: "fp8" (the default, round 6) runs the two correction products of every 3x3 convolution with
    # Cin % 128 == 0 and at least head_f8_min_rows output pixels PER VIEW (from the 64 x 64 level on at 512 x 512: refinenet2 / 1, head[0],
    # head[2]: ~88 % of the heads' FLOPs) on the block-
    # scaled fp8 MFMA (f3r.h F3R_SPLIT_X3F8: two matrix-pipe units instead of X3's three; the producers write the fp8 planes beside the fp16 high
    # plane instead of an fp16 low plane nobody else reads); "fp16": three fp16 products everywhere.  oracle/precision_study.py --study heads_f8:
    # the stress fixture moves from 6.94e-4 to 6.93e-4.  Changing it needs invalidate_packed_weights().
    head_corrections = "fp8"
    head_f8_min_rows = 4096
    # head[2] + ReLU -> head[4] (1x1 conv to 3 / 4 channels) -> postprocess in ONE launch (f3r_gemm_args.fin_w) whenever head[2] has at least
    # this many output pixels per view and the head's last_dim is 128: the 128-channel activation at full resolution (512 B per pixel with planes) is never
    # written or read back.  A very large value restores conv -> f3r_dpt_final.
    head_tail_fused_min_rows = 4096
    high_fc1_planes = True   # False: fc1 weights single-plane in precision "high" (see _pack_block); changing it needs invalidate_packed_weights()

    @property
    def _head_f8(self):
        return self.head_corrections == "fp8" and self.precision == "high" and self.compute_dtype == torch.float16

    @property
    def _fc1_split(self):
        return None if (self.high_fc1_planes or self.precision != "high") else False

    @property
    def _sp(self):
        """split mode of the transformer's linear layers: weights only ("w2") in "high", both operands ("x3") in "exact"."""
        return {"fast": None, "high": "w2", "robust": "x3", "exact": "x3"}[self.precision]

    def _pair(self, x_f32):
        """fp32 -> (hi, lo) lowp planes."""
        return ops.cast_lp(x_f32, self.compute_dtype, want_lo=True)

    def _pack(self, device):
        lp = self.compute_dtype
        f8 = self.low_plane == "fp8" and self.precision == "high" and lp == torch.float16
        key = (lp, self.precision, str(device), self._params_version(), f8, bool(self.high_fc1_planes), self._head_f8)
        if self._packed is not None and self._packed["key"] == key:
            return self._packed
        alt = getattr(self, "_packed_alt", None)
        if alt is not None and alt["key"] == key:  # two packs are kept, so alternating inference(dtype=...) calls do not re-pack
            self._packed, self._packed_alt = alt, self._packed
            return self._packed
        if self._packed is not None and self._packed["key"][3] != key[3]:  # parameters changed: both packs are stale
            self._packed = self._packed_alt = None
            self._graphs.clear()
        if alt is not None:  # a third format evicts the older pack: captured graphs may hold pointers into its weight tensors
            self._graphs.clear()
        self._packed_alt = self._packed
        enc, dec = self.encoder, self.decoder
        hp = self._hp
        pk = dict(key=key)
        if isinstance(enc, DinoEncoder):
            vit = enc.model
            pk["pe_w"] = ops.pack_linear_weight(vit.patch_embed.proj.weight.detach().float(), lp, hp)  # (D, 3*14*14 = 588) -> Kpad 640
            pk["pe_b"] = _f32(vit.patch_embed.proj.bias)
            pk["enc"] = [_pack_dino_block(b, lp, hp) for b in vit.blocks]
            pk["enc_norm"] = (_f32(vit.norm.weight), _f32(vit.norm.bias), vit.norm.eps)
            pk["dino_pos"] = {}  # (h, w) -> interpolated position rows, filled on demand (_dino_pos)
        else:
            pk["pe_w"] = ops.pack_linear_weight(enc.patch_embed.proj.weight.detach().float(), lp, hp)
            pk["pe_b"] = _f32(enc.patch_embed.proj.bias)
            pk["enc"] = [_pack_block(b, lp, hp, fc1_split=self._fc1_split, f8=f8) for b in enc.enc_blocks]
            pk["enc_norm"] = (_f32(enc.enc_norm.weight), _f32(enc.enc_norm.bias), enc.enc_norm.eps)
        pk["de_w"] = ops.pack_linear_weight(dec.decoder_embed.weight.detach().float(), lp, hp)
        pk["de_b"] = _f32(dec.decoder_embed.bias)
        if isinstance(dec, LlamaDecoder):
            pk["dec"] = [_pack_llama_block(b, dec.num_heads, lp, hp, dec.n_kv_heads, dec.is_causal) for b in dec.layers]
            pk["dec_norm"] = (_f32(dec.norm.weight), None, dec.norm.eps)
            pk["view0"] = _f32(dec.view0_embed)
        else:
            pk["dec"] = [_pack_block(b, lp, hp, dec.embed_dim // dec.num_heads, fc1_split=self._fc1_split, f8=f8) for b in dec.dec_blocks]
            pk["dec_norm"] = (_f32(dec.dec_norm.weight), _f32(dec.dec_norm.bias), dec.dec_norm.eps)
        pk["head"] = _pack_head(self.downstream_head, lp, hp, self._head_f8)
        pk["head_local"] = _pack_head(self.downstream_head_local, lp, hp, self._head_f8) if self.downstream_head_local is not None else None
        self._packed = pk
        return pk

    def _rope(self, n_pos, device):
        key = (n_pos, str(device))
        if key not in self._rope_cache:
            self._rope_cache[key] = ops.rope_tables(n_pos, self.encoder.rope_freq, device)
        return self._rope_cache[key]

    # ---------------------------------------------------------------- transformer block on the HIP kernels
    def _block_ws(self, pb, T, D, n_seq, seq_len, dev, external_kv=False):
        """Workspace of the blocks of one encoder pass / decoder sample: ONE allocation shared by all of its layers.  external_kv: K / V^T
        are written into the buffers of a KVExchange (view-sharded path), the workspace keeps no room for them."""
        hidden = pb.fc1_w.shape[0]  # rows of the (possibly stacked [w1; w3]) up-projection
        f8_rows = (pb.fc1_w8 is not None or pb.qk_w8 is not None) and T % 256 == 0   # the pass may take the fp8-low-plane GEMMs (_block decides per GEMM)
        return ops.BlockWorkspace(T, D, hidden, n_seq, seq_len, self.compute_dtype, dev, kv_dim=0 if external_kv else pb.kv_dim, f8_rows=f8_rows)

    def _block(self, x, pb, n_heads, scale, seq_len, n_seq, rope, kv_exchange=None, ws=None, kv_tap=None):
        """x: fp32 residual stream [n_seq*seq_len][D], updated in place.  blocks.py:236-239.  precision "high": every projection runs
        with split weights (hi + lo planes, split="w2"); the activations (LN output, attention output, MLP hidden) stay single.
        ws: the pass's BlockWorkspace (made here when absent): no per-layer allocation."""
        lp = self.compute_dtype
        if self._x3:
            return self._block_exact(x, pb, n_heads, scale, seq_len, n_seq, rope, kv_exchange)
        sp = "w2" if self.precision == "high" else None
        D = x.shape[1]
        T = x.shape[0]
        if ws is None:
            ws = self._block_ws(pb, T, D, n_seq, seq_len, x.device)
        q = ws.q
        if kv_exchange is None:
            k, vt = ws.k, ws.vt
            ldvt = vt.shape[-1]
        else:  # view-sharded: write K / V^T straight into the (padded, persistent) send buffers of the exchange
            k, vt = kv_exchange.k_loc, kv_exchange.vt_loc
        # the q | k launch with its low plane in fp8 (Fast3R.low_plane) where its 256 x 256 tiles fill the chip (from 8192 tokens on; the encoder's
        # RoPE-2D rides in that launch's epilogue since round 5), V^T on two fp16 planes; sequences must be multiples of 256 tokens for the V^T tiles
        f8_qkv = (pb.qk_w8 is not None and T % 256 == 0 and seq_len % 256 == 0 and pb.rope_mode == 0 and pb.q_dim == 0
                  and (T // 256) * (pb.qk_w8.shape[0] // 256) >= 256)
        if f8_qkv:
            rows1 = ops.layernorm_f8(x, pb.n1w, pb.n1b, pb.eps, out_rows=ws.rows8(D), rms=pb.rms)
            ops.gemm_qkv(rows1, pb.qk_w8, pb.qkv_b, q, k, vt, seq_len, rope, q_scale=scale * ops.LOG2E, split="w2f8", w_scale=pb.qk_ws, w_aux=pb.v_w2)
            h = ws.h
        else:
            h, _ = ops.layernorm(x, pb.n1w, pb.n1b, pb.eps, lp, out_lp=ws.h, rms=pb.rms)
            ops.gemm_qkv(h, pb.qkv_w, pb.qkv_b, q, k, vt, seq_len, rope, q_scale=scale * ops.LOG2E, rope_mode=pb.rope_mode, split=sp, q_dim=pb.q_dim)
        if kv_tap is not None and kv_exchange is None:
            kv_tap(k, vt)  # the K / V^T of this fusion layer (capture for the per-rank emulation test)
        o = h  # LN output is dead: reuse as the attention output buffer
        gqa = dict(kv_group=pb.kv_group, causal=pb.causal, head_dim=pb.head_dim)
        if kv_exchange is None:
            Dkv = k.shape[1]
            ops.attention(q, o, n_heads, scale, [(k, vt, seq_len, seq_len * Dkv, Dkv * ldvt)], tq=seq_len, batch=n_seq,
                          q_batch_stride=seq_len * D, o_batch_stride=seq_len * D, q_prescaled=True, **gqa)
        else:
            # view-sharded: the all-gather of the remote K / V^T runs while the kernel attends over the local shard; the
            # online-softmax state (m, l, O) is parked in fp32 and resumed over the remote segments once they have landed
            sharding = self.sharding
            sharding.begin_layer(kv_exchange)  # exchange="auto": which form this layer uses (probe, then the winner)
            kv_exchange.start()
            if seq_len == 0:
                kv_exchange.finish()  # a rank without tokens still takes part in the collective
            elif kv_exchange.has_remote:
                pos = kv_exchange.positions()  # global token index of the first row of every rank's shard (causal attention only)
                # (reserve_cus: the persistent launch leaves a few CUs to the kernels that move the other ranks' shards meanwhile -- ViewSharding.reserve_cus)
                ops.attention(q, o, n_heads, scale, [kv_exchange.local_segment()], tq=seq_len, q_prescaled=True,
                              state=kv_exchange.state, state_out=True, q_pos0=pos[kv_exchange.rank], seg_pos0=[pos[kv_exchange.rank]],
                              reserve_cus=getattr(sharding, "reserve_cus", 0), **gqa)
                kv_exchange.mark_local_done()
                # the remote shards, as ONE group once the all-gathers have landed or ("p2p" exchange) shard by shard in arrival order
                groups = [(w, sg) for w, sg in kv_exchange.remote_groups()]
                live = [i for i, (_, sg) in enumerate(groups) if sg]
                for i, (wait, segs) in enumerate(groups):
                    wait()
                    if live and i == live[0]:
                        kv_exchange.mark_remote_start()  # what the local launch did not hide = the gap on this stream up to here
                    if not segs:
                        continue
                    ops.attention(q, o, n_heads, scale, segs, tq=seq_len, q_prescaled=True, state=kv_exchange.state, state_in=True,
                                  state_out=(i != live[-1]), q_pos0=pos[kv_exchange.rank],
                                  seg_pos0=[kv_exchange.remote_position_of(sg) for sg in segs], **gqa)
            else:
                kv_exchange.finish()
                ops.attention(q, o, n_heads, scale, [kv_exchange.local_segment()], tq=seq_len, q_prescaled=True, **gqa)
            sharding.end_layer(kv_exchange)
        ops.gemm(o, pb.proj_w, bias=pb.proj_b, res_f32=x, out_f32=x, split=sp)
        # low plane in fp8 (Fast3R.low_plane): only the hand-scheduled 256 x 256-tile kernels read the [fp16 | fp8] rows, so a GEMM takes that form
        # when its tiles fill the chip at least once (256 CUs): fc1 (4 D wide) from 4096 tokens on, fc2 from 16 384; below that the scene is
        # launch- and occupancy-bound and the compiler-scheduled small-tile kernels win (N = 3: 118 us against 36 us per fc2, round-5 profile)
        f8_fc1 = pb.fc1_w8 is not None and T % 256 == 0 and not pb.swiglu_hidden and (T // 256) * (pb.fc1_w8.shape[0] // 256) >= 256
        if f8_fc1:
            f8_fc2 = (T // 256) * (pb.fc2_w8.shape[0] // 256) >= 256
            rows = ops.layernorm_f8(x, pb.n2w, pb.n2b, pb.eps, out_rows=ws.rows8(D), rms=pb.rms)
            if f8_fc2:   # fc1's GELU epilogue writes rows [4 D fp16 | 4 D fp8], fc2 runs on them
                _, hid = ops.gemm(rows, pb.fc1_w8, bias=pb.fc1_b, act="gelu", out_lp=ws.hid8(), split="w2f8", w_scale=pb.fc1_ws, out_f8_rows=True)
                ops.gemm(hid, pb.fc2_w8, bias=pb.fc2_b, res_f32=x, out_f32=x, split="w2f8", w_scale=pb.fc2_ws)
            else:
                _, hid = ops.gemm(rows, pb.fc1_w8, bias=pb.fc1_b, act="gelu", out_lp=ws.hid, split="w2f8", w_scale=pb.fc1_ws)
                ops.gemm(hid, pb.fc2_w, bias=pb.fc2_b, res_f32=x, out_f32=x, split=sp)
            return x
        h2, _ = ops.layernorm(x, pb.n2w, pb.n2b, pb.eps, lp, out_lp=o, rms=pb.rms)
        if pb.swiglu_hidden:  # LlamaDecoder FeedForward: w2(silu(w1 x) * w3 x) (llama.py:284)
            _, ab = ops.gemm(h2, pb.fc1_w, out_lp=ws.hid, split=sp)
            hid = ops.silu_mul(ab, pb.swiglu_hidden)
        else:
            _, hid = ops.gemm(h2, pb.fc1_w, bias=pb.fc1_b, act="gelu", out_lp=ws.hid, split=sp if pb.fc1_split in (None, True) else None)
        ops.gemm(hid, pb.fc2_w, bias=pb.fc2_b, res_f32=x, out_f32=x, split=sp)
        return x

    def _block_exact(self, x, pb, n_heads, scale, seq_len, n_seq, rope, kv_exchange):
        """precision "exact": the same block with every operand as hi + lo planes and fp32 attention.  LayerNorm / RMSNorm -> fp32 -> planes;
        QKV through the generic epilogue into an fp32 [T][Dq + 2 Dkv] buffer (+ rotary embedding in place, 2-D or per-view); f3r_attn_f32_ex
        (grouped-query heads, causal by absolute position) -> planes; proj / fc1 (+GELU | SwiGLU gate, planes out) / fc2 as X3 GEMMs with the
        fp32 residual epilogue.  View-sharded: the fp32 K and V rows of all ranks are all-gathered (blocking: this is the validation mode)
        and the rank's queries attend over them in one launch."""
        lp = self.compute_dtype
        T = x.shape[0]
        _, hf = ops.layernorm(x, pb.n1w, pb.n1b, pb.eps, lp, want_lp=False, want_f32=True, rms=pb.rms)
        h, hl = self._pair(hf)
        qkv, _ = ops.gemm(h, pb.qkv_w, bias=pb.qkv_b, want_f32=True, split="x3", a_lo=hl)
        n_kv = n_heads // pb.kv_group
        if rope is not None and T:
            ops.rope_f32(qkv, n_heads + n_kv, max(1, seq_len), rope, pb.rope_mode)
        gqa = dict(head_dim=pb.head_dim, kv_group=pb.kv_group, causal=pb.causal)
        if (kv_exchange is None and self.precision == "robust" and self._qk3_ok(pb, n_seq, seq_len)
                and (n_seq == 1 or self.robust_encoder_attention == "planes")):
            # precision "robust": Q and K as hi + lo fp16 planes, three products per score block on the hand-scheduled kernel (f3r_attn_args.qk_planes),
            # P and V single fp16; the launch parks its softmax state and a small pass turns it into the planes of the attention output
            # (robust_corrections "fp8": the two correction products on the block-scaled fp8 MFMA -- rows [hi fp16 | e4m3(hi) | e4m3(lo 2^12)])
            planes = 3 if self.robust_corrections == "fp8" else 2
            qp, kp, vt = ops.qkv_planes(qkv, n_heads, n_kv, n_seq, seq_len, scale * ops.LOG2E, lp, planes=planes)
            state = ops.attention_state(T, n_heads, x.device, pb.head_dim)
            ldvt_ = vt.shape[-1]
            ops.attention(qp, state[0], n_heads, scale, [(kp, vt.view(n_seq * n_kv * 64, ldvt_), seq_len, seq_len * n_kv * 128, n_kv * 64 * ldvt_)], tq=seq_len,
                          batch=n_seq, q_batch_stride=seq_len * n_heads * 128, o_batch_stride=seq_len * n_heads * 64, q_prescaled=True, state=state,
                          state_out=True, kv_group=pb.kv_group, head_dim=64, qk_planes=planes, kernel_sel=2)
            o, ol = ops.attention_state_finish(state, n_heads, 64, lp)
            del qp, kp, vt, state
        elif kv_exchange is None:
            o, ol = ops.attention_f32(qkv, n_heads, n_seq, seq_len, scale, lp, **gqa)
        else:
            Dq, Dkv = n_heads * pb.head_dim, n_kv * pb.head_dim
            k_all, v_all = kv_exchange.gather_rows_f32(qkv[:, Dq:Dq + Dkv], qkv[:, Dq + Dkv:])
            if T:
                o, ol = ops.attention_f32(qkv, n_heads, 1, seq_len, scale, lp, kv=(k_all, v_all), q_pos0=kv_exchange.positions()[kv_exchange.rank], **gqa)
            else:
                o = ol = torch.empty((0, Dq), dtype=lp, device=x.device)
        ops.gemm(o, pb.proj_w, bias=pb.proj_b, res_f32=x, out_f32=x, split="x3", a_lo=ol)
        _, hf = ops.layernorm(x, pb.n2w, pb.n2b, pb.eps, lp, want_lp=False, want_f32=True, out_f32=hf, rms=pb.rms)
        h, hl = self._pair(hf)
        if pb.swiglu_hidden:  # LlamaDecoder FeedForward: w2(silu(w1 x) * w3 x) (llama.py:284)
            ab, _ = ops.gemm(h, pb.fc1_w, want_f32=True, split="x3", a_lo=hl)
            hid, hid_lo = ops.silu_mul_f32(ab, pb.swiglu_hidden, lp)
        else:
            _, hid, hid_lo = ops.gemm(h, pb.fc1_w, bias=pb.fc1_b, act="gelu", want_lp=True, want_lo=True, split="x3", a_lo=hl)
        ops.gemm(hid, pb.fc2_w, bias=pb.fc2_b, res_f32=x, out_f32=x, split="x3", a_lo=hid_lo)
        return x

    def _qk3_ok(self, pb, n_seq, seq_len):
        """can this block's attention take the three-product kernels (f3r_attn_asm_qk3{,f8}_f16)?  head_dim 64, fp16, no causal mask, whole 64-key
        tiles per sequence, at least one wave of queries, a power-of-two head group; any number of sequences (the fusion decoder's one, the
        encoder's one per view: the kernels park their softmax state per sequence).  Everything else in precision "robust" -- odd token counts
        (DINOv2's class token), causal Llama decoders, sharded models -- runs the fp32 attention of precision "exact" (more exact, slower)."""
        g = pb.kv_group
        return (self.compute_dtype == torch.float16 and pb.head_dim == 64 and not pb.causal and n_seq >= 1 and seq_len >= 64 and seq_len % 64 == 0
                and n_seq < 65536 and g >= 1 and (g & (g - 1)) == 0)

    def _planes(self, x_f32):
        """fp32 -> the operand format of the DPT heads: (hi, lo) planes in "high" precision, (lowp, None) otherwise."""
        if self._hp:
            return ops.cast_lp(x_f32, self.compute_dtype, want_lo=True)
        return ops.cast_lp(x_f32, self.compute_dtype), None

    def _dino_pos(self, pk, h, w, dev):
        """DINOv2 `interpolate_pos_encoding` (models/vision_transformer.py) for an h x w token grid -> (cls row + pos[0] (D,), patch rows
        (h*w, D)) fp32, cached per grid.  Parameter preprocessing on the host side of the boundary (like the RoPE tables): the trained
        (pos_grid x pos_grid) table is resized bicubically with the hub models' settings (interpolate_offset 0.1, no antialias)."""
        key = (h, w)
        if key not in pk["dino_pos"]:
            vit = self.encoder.model
            pe = vit.pos_embed.detach().float().to(dev)
            M, D = vit.pos_grid, vit.embed_dim
            patch = pe[:, 1:]
            if not (h * w == M * M and h == w):
                patch = torch.nn.functional.interpolate(patch.reshape(1, M, M, D).permute(0, 3, 1, 2), mode="bicubic", antialias=False,
                                                        scale_factor=(float(h + 0.1) / M, float(w + 0.1) / M))
                assert tuple(patch.shape[-2:]) == (h, w)
                patch = patch.permute(0, 2, 3, 1).reshape(1, h * w, D)
            cls_row = (vit.cls_token.detach().float().to(dev)[0, 0] + pe[0, 0]).contiguous()
            pk["dino_pos"][key] = (cls_row, patch[0].contiguous())
        return pk["dino_pos"][key]

    def _encode_dino(self, imgs, pk):
        """DINOv2 forward_features -> x_norm_patchtokens (DinoEncoder._process_images, fast3r.py:636-651) for a batch of same-size,
        upright images: patch embedding (k = s = 14) + [cls] + resized position table, the ViT blocks (no RoPE; LayerScale folded into
        the packed weights), final LayerNorm, patch tokens only.  Same return convention as _encode."""
        lp = self.compute_dtype
        hp, sp = self._hp, self._sp
        vit = self.encoder.model
        NV, _, H, W = imgs.shape
        ps, D = vit.patch_size, vit.embed_dim
        assert H % ps == 0 and W % ps == 0, f"Input image size ({H}, {W}) is not a multiple of the patch size ({ps})"
        h, w = H // ps, W // ps
        P = h * w
        cls_row, pos_rows = self._dino_pos(pk, h, w, imgs.device)
        out = torch.empty((NV * P, D), dtype=lp, device=imgs.device)
        out_lo = torch.empty_like(out) if hp else None
        kpad = pk["pe_w"].shape[1] // (2 if hp else 1)
        step = max(1, self.max_parallel_views_for_encoder)
        for v0 in range(0, NV, step):
            v1 = min(NV, v0 + step)
            n = v1 - v0
            a = ops.patchify(imgs[v0:v1].contiguous(), ps, lp, ld_out=kpad)
            tok, _ = ops.gemm(a, pk["pe_w"], bias=pk["pe_b"], res_f32=pos_rows.repeat(n, 1), want_f32=True, split=sp,
                              a_lo=self._patch_lo(imgs[v0:v1], ps, kpad))
            x = torch.empty((n, 1 + P, D), dtype=torch.float32, device=imgs.device)  # [cls | patches] per image
            x[:, 0] = cls_row
            x[:, 1:] = tok.view(n, P, D)
            x = x.view(n * (1 + P), D)
            ws = self._block_ws(pk["enc"][0], x.shape[0], D, n, 1 + P, imgs.device)
            for pb in pk["enc"]:
                self._block(x, pb, vit.num_heads, 64 ** -0.5, 1 + P, n, None, ws=ws)
            w_, b_, eps = pk["enc_norm"]
            if hp:
                _, y = ops.layernorm(x, w_, b_, eps, lp, want_lp=False, want_f32=True)
                hi, lo = ops.cast_lp(y, lp, want_lo=True)
                out[v0 * P:v1 * P].view(n, P, D).copy_(hi.view(n, 1 + P, D)[:, 1:])
                out_lo[v0 * P:v1 * P].view(n, P, D).copy_(lo.view(n, 1 + P, D)[:, 1:])
            else:
                y, _ = ops.layernorm(x, w_, b_, eps, lp)
                out[v0 * P:v1 * P].view(n, P, D).copy_(y.view(n, 1 + P, D)[:, 1:])
        return out, out_lo, P, (h, w)

    def _patch_lo(self, imgs, ps, ld_out):
        """precision "exact": the low plane of the im2col rows = the im2col rows of the image's low plane (patchify only moves and rounds:
        its output is lowp(img), so lowp(img - lowp(img)) patchified is exactly the remainder plane); None otherwise."""
        if not self._x3:
            return None
        _, lo = ops.cast_lp(imgs.contiguous(), self.compute_dtype, want_lo=True)
        return ops.patchify(lo.float(), ps, self.compute_dtype, ld_out=ld_out)

    def _encode(self, imgs, pk):
        """CroCoEncoder.forward (fast3r.py:549-559) for a batch of same-size images -> enc_norm output [NV*P][D] as (lowp, low plane or
        None).  The high plane alone feeds decoder_embed; both planes are hook 0 of the heads in "high" precision."""
        if isinstance(self.encoder, DinoEncoder):
            return self._encode_dino(imgs, pk)
        lp = self.compute_dtype
        hp, sp = self._hp, self._sp
        enc = self.encoder
        NV, _, H, W = imgs.shape
        ps = enc.patch_size
        assert H % ps == 0, f"Input image height ({H}) is not a multiple of patch size ({ps})."  # patch_embed.py:27-32
        assert W % ps == 0, f"Input image width ({W}) is not a multiple of patch size ({ps})."
        h, w = H // ps, W // ps
        P = h * w
        rope = self._rope(max(h, w), imgs.device) + (w,)
        out = torch.empty((NV * P, enc.embed_dim), dtype=lp, device=imgs.device)
        out_lo = torch.empty_like(out) if hp else None
        step = max(1, self.max_parallel_views_for_encoder)
        for v0 in range(0, NV, step):
            v1 = min(NV, v0 + step)
            a = ops.patchify(imgs[v0:v1].contiguous(), ps, lp)
            x, _ = ops.gemm(a, pk["pe_w"], bias=pk["pe_b"], want_f32=True, split=sp, a_lo=self._patch_lo(imgs[v0:v1], ps, 0))
            ws = self._block_ws(pk["enc"][0], x.shape[0], enc.embed_dim, v1 - v0, P, imgs.device)
            for pb in pk["enc"]:
                self._block(x, pb, enc.num_heads, (enc.embed_dim // enc.num_heads) ** -0.5, P, v1 - v0, rope, ws=ws)
            w_, b_, eps = pk["enc_norm"]
            if hp:
                _, y = ops.layernorm(x, w_, b_, eps, lp, want_lp=False, want_f32=True)
                hi, lo = ops.cast_lp(y, lp, want_lo=True)
                out[v0 * P:v1 * P].copy_(hi)
                out_lo[v0 * P:v1 * P].copy_(lo)
            else:
                ops.layernorm(x, w_, b_, eps, lp, out_lp=out[v0 * P:v1 * P])
        return out, out_lo, P, (h, w)

    # ---------------------------------------------------------------- fusion decoder on the HIP kernels
    def _decode_sample(self, pk, enc_hi, enc_lo, Ps, emb_rows, v_lo, kvx, f32_hooks=False):
        """Fast3RDecoder.forward / LlamaDecoder.forward (fast3r.py:768-808 / :924-966) for ONE sample: enc_hi [T_loc][Denc] lowp (the
        local views' encoder tokens, view after view; enc_lo = their low plane or None), Ps = tokens per local view, emb_rows
        (N_total, D) fp32 = the image-id rows of ALL views (this rank's are [v_lo, v_lo + len(Ps))).  Returns the 4 hooked
        outputs (fast3r.py:148) as (plane, low plane or None) pairs; f32_hooks: as (fp32 tensor, None) instead, i.e. before the rounding
        to the heads' operand format (decode_tokens: parity of the decoder alone)."""
        dec = self.decoder
        lp = self.compute_dtype
        sp = self._sp
        a_lo = enc_lo if sp == "x3" else None  # "exact": decoder_embed reads both planes of the encoder tokens
        dev = enc_hi.device
        L = dec.depth
        llama = isinstance(dec, LlamaDecoder)
        hd_ = int(self.decoder_args["depth"]) if llama else L  # the heads read decoder_args["depth"] for both decoder types (fast3r.py:137-148)
        hooks = [0, hd_ * 2 // 4, hd_ * 3 // 4, hd_]
        scale = dec.attention_scale(self.training)
        D = dec.embed_dim
        T_loc, n_loc = sum(Ps), len(Ps)
        x = torch.empty((T_loc, D), dtype=torch.float32, device=dev)
        ws = self._block_ws(pk["dec"][0], T_loc, D, 1, T_loc, dev, external_kv=kvx is not None)  # one allocation for all L blocks
        planes = (lambda t: (t.clone(), None)) if f32_hooks else self._planes
        want_f32_norm = f32_hooks or self._hp
        if llama:
            # embed; per layer add view0_embed to the tokens of view 0, then the block with the rotary angles of each token's view;
            # outputs[0] = embedded tokens, outputs[n_layers] = final RMSNorm
            ops.gemm(enc_hi, pk["de_w"], bias=pk["de_b"], out_f32=x, split=sp, a_lo=a_lo)
            rows = emb_rows[v_lo:v_lo + n_loc]                              # (n_loc, 64) = [cos (32) | sin (32)] of each view's id
            if len(set(Ps)) == 1:
                rope = (rows[:, :32].contiguous(), rows[:, 32:].contiguous(), Ps[0])
            else:  # mixed resolutions: one table row per token
                per_tok = rows.repeat_interleave(torch.tensor(Ps, device=dev), dim=0)
                rope = (per_tok[:, :32].contiguous(), per_tok[:, 32:].contiguous(), 1)
            view0_rows = Ps[0] if v_lo == 0 else 0                          # view 0 lives on the rank that owns the first views
            taps = {}
            if 0 in hooks:
                taps[0] = planes(x)
            for li, pb in enumerate(pk["dec"]):
                ops.rows_add(x, pk["view0"], view0_rows)
                self._block(x, pb, dec.num_heads, scale, T_loc, 1, rope, kvx, ws=ws, kv_tap=self.kv_tap)
                if (li + 1) in hooks and (li + 1) != L:
                    taps[li + 1] = planes(x)
            if L in hooks:
                w_, b_, eps = pk["dec_norm"]
                if want_f32_norm:
                    _, y = ops.layernorm(x, w_, None, eps, lp, want_lp=False, want_f32=True, rms=True)
                    taps[L] = planes(y)
                else:
                    taps[L] = (ops.layernorm(x, w_, None, eps, lp, rms=True)[0], None)
        else:
            if len(set(Ps)) == 1:
                ops.gemm(enc_hi, pk["de_w"], bias=pk["de_b"], rowadd=emb_rows[v_lo:v_lo + n_loc].contiguous(), rowadd_div=Ps[0], out_f32=x, split=sp,
                         a_lo=a_lo)
            else:
                r0 = 0
                for i in range(n_loc):
                    ops.gemm(enc_hi[r0:r0 + Ps[i]], pk["de_w"], bias=pk["de_b"], rowadd=emb_rows[v_lo + i:v_lo + i + 1].contiguous(),
                             rowadd_div=Ps[i], out_f32=x[r0:r0 + Ps[i]], split=sp, a_lo=None if a_lo is None else a_lo[r0:r0 + Ps[i]])
                    r0 += Ps[i]
            taps = {0: (enc_hi.float(), None) if f32_hooks else (enc_hi, enc_lo)}
            for li, pb in enumerate(pk["dec"]):
                self._block(x, pb, dec.num_heads, scale, T_loc, 1, None, kvx, ws=ws, kv_tap=self.kv_tap)
                if (li + 1) in hooks[1:3]:
                    taps[li + 1] = planes(x)
            w_, b_, eps = pk["dec_norm"]
            if want_f32_norm:
                _, y = ops.layernorm(x, w_, b_, eps, lp, want_lp=False, want_f32=True)
                taps[L] = planes(y)
            else:
                taps[L] = (ops.layernorm(x, w_, b_, eps, lp)[0], None)
        return [taps[hk] for hk in hooks]

    @torch.no_grad()
    def decode_tokens(self, enc_tokens, tokens_per_view, image_ids, return_f32=False, enc_tokens_lo=None):
        """The fusion decoder alone (BASELINE configs[1]: "fusion transformer only, frozen random encoder"): enc_tokens lowp
        [sum(tokens_per_view)][enc_embed_dim] on the GPU, image_ids (N,) or (1, N) long -> the 4 hooked outputs [T][D], lowp as the
        heads read them, or fp32 before that rounding with return_f32.  enc_tokens_lo: optional low plane of the tokens (split-precision
        modes: tokens = enc_tokens + enc_tokens_lo), default zero."""
        dev = enc_tokens.device
        if dev.type != "cuda":
            raise F3RError(f"fast3r_amd.Fast3R runs only on a ROCm GPU (tokens are on {dev}); there is no CPU fallback")
        with torch.cuda.device(dev):
            pk = self._pack(dev)
            ids = torch.as_tensor(image_ids).reshape(-1).to(dev)
            assert ids.numel() == len(tokens_per_view)
            emb_rows = self.decoder.image_idx_emb.to(dev)[ids]
            enc_lo = None
            if self._hp and not isinstance(self.decoder, LlamaDecoder):
                enc_lo = torch.zeros_like(enc_tokens) if enc_tokens_lo is None else enc_tokens_lo.contiguous()
            with ops.family("transformer_linears"):
                out = self._decode_sample(pk, enc_tokens.contiguous(), enc_lo, list(tokens_per_view), emb_rows, 0, None, f32_hooks=return_f32)
            return [t[0] for t in out]

    # ---------------------------------------------------------------- DPT head on the HIP kernels
    def _dpt(self, hk, toks, nv, gh, gw):
        """DPTOutputAdapter_fix.forward + postprocess (heads/dpt_head.py:42-129) for nv same-size views.
        toks: the 4 hooked token matrices as (lowp [nv*gh*gw][C], low plane or None) pairs.  Activations stay NHWC lowp (two planes
        each in "high" precision, every conv then runs split="x3"); accumulation fp32.  A ResidualConvUnit reads relu(x) for its first
        conv and x for its skip add (dpt_block.py:143-154): the PRODUCER of x writes both (f3r_gemm_args.out_relu), so no conv
        pre-activates its operand while staging it and all of them can use LDS-DMA."""
        hp = self._hp
        sp = "x3" if hp else None
        ld = hk.dims
        # Activations are (hi, lo, f8) triples: the fp16 / bf16 high plane, the 16-bit low plane (None = not carried) and the fp8 planes (None = not
        # carried).  Which of them a producer writes follows from its CONSUMERS: a residual add, a 1x1 conv or an upsample read hi + lo; a 3x3
        # convolution that runs split "x3f8" reads hi + f8 and needs no lo.
        f8_on = self._head_f8

        def use_f8(rows, cin, w):
            # decided by the pixels of ONE view (rows / nv), so that a forward computes the same numbers whatever max_parallel_views_for_head is;
            # the operand of a launch must stay below 4 GiB (32-bit lane offsets of the 256-tile kernel: 63 views of 512 x 512 x 128 channels)
            return bool(f8_on and w.w8 is not None and cin % 128 == 0 and rows // nv >= self.head_f8_min_rows and rows * cin * 2 < (1 << 32))

        def c1(i):  # act_postprocess[i][0]: 1x1 conv on the tokens
            r = ops.gemm(toks[i][0], hk.a_w[i], bias=hk.a_b[i], want_lp=True, want_lo=hp, split=sp, a_lo=toks[i][1])
            return tuple(t.view(nv, gh, gw, ld[i]) for t in r[1:]) + (None,) if hp else (r[1].view(nv, gh, gw, ld[i]), None, None)

        def convT(xp, w, b, s, cout):
            r = ops.convT(xp[0], w, b, s, cout, split=sp, x_lo=xp[1], want_lo=hp)
            return r + (None,) if hp else (r, None, None)

        def conv(xp, w, bias=None, stride=1, act=None, res=None, res2=None, relu_copy=False, x_f8=False, relu_f8=False, x_lo=True, fin=None):
            """-> {"x": (out, lo, f8), "relu": (relu(out), lo, f8)}; x_f8 / relu_f8: the consumer of that copy is an "x3f8" convolution (fp8 planes
            instead of the low plane; x_lo=False drops the output's own low plane too).  fin: the fused head tail -> (pts3d, conf)."""
            kw = dict(split=sp, x_lo=xp[1])
            wt = w.w
            if xp[2] is not None:  # the producer wrote fp8 planes because this launch takes them
                kw = dict(split="x3f8", x_f8=xp[2], w_scale=w.scale)
                wt = w.w8
            r = ops.conv3x3(xp[0], wt, stride=stride, bias=bias, act=act,
                            res_lp=None if res is None else res[0], res_lp_lo=None if res is None else res[1],
                            res_lp2=None if res2 is None else res2[0], res_lp2_lo=None if res2 is None else res2[1],
                            want_lo=hp and x_lo, want_relu=relu_copy, want_f8=x_f8, want_relu_lo=hp and not relu_f8, want_relu_f8=relu_f8, fin=fin, **kw)
            if fin is not None:
                return r
            if not isinstance(r, dict):
                return {"x": (r, None, None)}
            return {"x": (r["out"], r.get("out_lo"), r.get("out_f8")), "relu": (r.get("relu"), r.get("relu_lo"), r.get("relu_f8"))}

        def rows_of(t):
            return t.shape[0] * t.shape[1] * t.shape[2]

        l0 = convT(c1(0), hk.t0_w, hk.t0_b, 4, hk.t0_cout)                             # dpt_block.py:416-434 (channels padded to 64: _pack_head)
        l1 = convT(c1(1), hk.t1_w, hk.t1_b, 2, hk.t1_cout)                             # :436-454
        l2 = c1(2)                                                                      # :456-464
        l3 = conv(c1(3), hk.c3_w, bias=hk.c3_b, stride=2)["x"]                          # :466-481
        # scratch.layer_rn (no bias); the relu copy feeds the first RCU convolution of its level
        ls = [conv(l, hk.rn_w[i], relu_copy=True, relu_f8=use_f8(rows_of(l[0]), hk.feature_dim, hk.ref[i]["u2c1" if i == 3 else "u1c1"][0]))
              for i, l in enumerate((l0, l1, l2, l3))]
        del l0, l1, l2, l3

        def rcu(x, c1w, c2w, extra=None, relu_copy=False, relu_f8=False):
            # x + conv2(relu(conv1(relu(x)))) (+ extra); x = {"x": planes, "relu": planes of relu(x)}
            f8 = use_f8(rows_of(x["x"][0]), hk.feature_dim, c2w[0])
            t = conv(x["relu"], c1w[0], bias=c1w[1], act="relu", x_f8=f8, x_lo=not f8)["x"]
            return conv(t, c2w[0], bias=c2w[1], res=x["x"], res2=extra, relu_copy=relu_copy, relu_f8=relu_f8)

        def fusion(r, path, skip=None, crop=None, out_f8=False):
            # out_conv(up2(RCU2(path + RCU1(skip)))); the 1x1 out_conv commutes exactly with the bilinear
            # interpolation (both linear, weights sum to 1), so it runs BEFORE the upsample on 4x fewer pixels.
            if skip is not None:
                path = rcu(skip, r["u1c1"], r["u1c2"], extra=path, relu_copy=True, relu_f8=use_f8(rows_of(skip["x"][0]), hk.feature_dim, r["u2c1"][0]))
            y = rcu(path, r["u2c1"], r["u2c2"])["x"]
            B, hh, ww, C = y[0].shape
            g = ops.gemm(y[0].view(B * hh * ww, C), r["out"][0], bias=r["out"][1], want_lp=True, want_lo=hp, split=sp,
                         a_lo=None if y[1] is None else y[1].view(B * hh * ww, C))
            z = (g[1].view(B, hh, ww, C), g[2].view(B, hh, ww, C) if hp else None)
            if out_f8:
                return ops.upsample2x(z[0], crop, x_lo=z[1], want_lo=False, want_f8=True)
            u = ops.upsample2x(z[0], crop, x_lo=z[1], want_lo=hp)
            return u + (None,) if hp else (u, None, None)

        p4 = fusion(hk.ref[3], ls[3], None, crop=(ls[2]["x"][0].shape[1], ls[2]["x"][0].shape[2]))  # dpt_head.py:69-71
        p3 = fusion(hk.ref[2], p4, ls[2])
        p2 = fusion(hk.ref[1], p3, ls[1])
        B1, h1, w1 = ls[0]["x"][0].shape[:3]
        p1 = fusion(hk.ref[0], p2, ls[0], out_f8=use_f8(B1 * 4 * h1 * w1, hk.feature_dim, hk.h0_w))
        del ls, p4, p3, p2
        y = conv(p1, hk.h0_w, bias=hk.h0_b)["x"]                                        # head[0]
        # head[1]: Interpolate(scale_factor = patch_size / 8, bilinear, align_corners=True) (dpt_block.py:374): x2 for patch 16, x1.75 for 14
        full = (y[0].shape[1] * hk.patch_size // 8, y[0].shape[2] * hk.patch_size // 8)
        rows2 = y[0].shape[0] * full[0] * full[1]
        if use_f8(rows2, hk.last_dim, hk.h2_w):
            u = ops.interp_bilinear(y[0], full, x_lo=y[1], want_lo=False, want_f8=True)
        else:
            u = ops.interp_bilinear(y[0], full, x_lo=y[1], want_lo=hp)
            u = u + (None,) if hp else (u, None, None)
        # head[2], head[3] (+ head[4] and postprocess in the same launch when the 256 x 128 tile kernel takes it)
        if (hk.fin is not None and rows2 // nv >= self.head_tail_fused_min_rows and hk.last_dim % 64 == 0 and rows2 * hk.last_dim * 2 < (1 << 32)):
            return conv(u, hk.h2_w, bias=hk.h2_b, act="relu", fin=hk.fin)
        y = conv(u, hk.h2_w, bias=hk.h2_b, act="relu")["x"]
        return ops.dpt_final(y[0], hk.h4_w, hk.h4_b, hk.conf_mode, x_lo=y[1], depth_mode=tuple(hk.depth_mode))  # head[4] + postprocess

    # ---------------------------------------------------------------- forward
    def forward(self, views, profiling=False, host_outputs=False):
        """fast3r.py:302-497.  views: list[N] of dicts with 'img' (B,3,H,W) on a ROCm device.
        host_outputs (fast3r_amd.inference sets it): return the per-view tensors on the HOST instead of the device -- every head chunk's
        pointmaps / confidences are copied into pinned host memory by a side stream as soon as the chunk is done, overlapped with the heads
        of the next chunk (what the reference does after the forward with a blocking `to_cpu`, dust3r/inference_multiview.py:92-93)."""
        if len(views) == 0:
            return ([], {}) if profiling else []
        if (self.use_graphs and not profiling and self.sharding is None and self.debug_taps is None and not isinstance(self.decoder, LlamaDecoder)
                and not self._x3):  # (the plane modes allocate per layer: they always run eagerly)
            dev = views[0]["img"].device
            if dev.type == "cuda":  # (anything else: the eager path raises its F3RError)
                with torch.cuda.device(dev):  # capture and replay on the tensors' device, whatever the caller's current device is
                    out = self._graphs.run(self, views, host_outputs=host_outputs)
                if out is not None:
                    return out
        return self._forward_eager(views, profiling, host_outputs=host_outputs)

    @staticmethod
    def _head_chunk_views(limit, gh, gw, cus=256):
        """Views per head launch: the count in (limit / 2, limit] whose convolution launches fill the chip's rounds best.  max_parallel_views_for_head
        (the reference's 25, fast3r.py:68) bounds the head's memory; the results do not depend on the chunking (tested bit for bit), so inside that
        bound the chunk is chosen for the 256-row output tiles of the big 3x3 convolutions: at 512 x 512, 25 views are 1 600 tiles = 6.25 rounds of 256
        workgroups at the 128 x 128 level and 1.56 at 64 x 64 (11 % / 22 % of those launches idle), 16 views are 4 and 1 round exactly.  Cost model:
        sum over the levels of (FLOPs per pixel) x rounds(c) / c."""
        if limit <= 2:
            return limit
        t = gh * gw   # pixels per view at the token grid; levels at 1/4, 1, 4, 16 (RCU convolutions), 64 (head[0]), 256 (head[2]) times that
        levels = [(t // 4, 2 * 256 * 256), (t, 4 * 256 * 256), (4 * t, 4 * 256 * 256), (16 * t, 4 * 256 * 256), (64 * t, 256 * 128), (256 * t, 128 * 128)]
        best, best_cost = limit, None
        for c in range(limit, max(1, limit // 2), -1):
            cost = sum(w * (-(-(-(-c * max(px, 1) // 256)) // cus)) for px, w in levels) / c
            if best_cost is None or cost < best_cost * 0.995:   # (ties and near-ties: the larger chunk = fewer launches)
                best, best_cost = c, cost
        return best

    def enable_graphs(self, on=True, max_views=64):
        """Opt-in hipGraph replay for launch-bound scenes (a forward of N <= max_views same-size views is a chain of
        ~400 + 30 N short launches; below N of about 20 the host cannot issue them as fast as the GPU retires them).  The
        first forward of a new (N, B, H, W) shape runs eagerly once and is then captured; later forwards of that shape copy
        the images into the captured input buffers, draw the image ids on the host exactly like the eager path and replay.
        Same kernels, same order, same results."""
        self.use_graphs = bool(on)
        self._graphs.max_views = max_views
        return self

    def _orientation_plan(self, views):
        """Per (view, sample): is the image stored transposed (a portrait picture kept as a landscape tensor)?

        The reference handles aspect ratio in two places: `ManyAR_PatchEmbed` (dust3r/patch_embed.py:41-105) swaps the
        axes of the samples whose `true_shape` is portrait before the patch convolution, and `transpose_to_landscape(head,
        activate=landscape_only)` (dust3r/utils/misc.py:61-106) runs the head of those samples on the (W, H) grid and swaps the output
        back.  With PatchEmbedDust3R / landscape_only=False (what the inference loaders set, utils/checkpoint_utils.py:37-38) neither
        happens and `true_shape` must be identical for all samples of a view (misc.py:69).  Returns, per view, `enc_swap` (B bools),
        `head_hw` (B (H, W) pairs: the image size the head predicts at) and `head_swap` (B bools)."""
        many_ar = self.encoder.patch_embed_cls in ("ManyAR_PatchEmbed", "dino")  # DinoEncoder.forward also encodes portraits upright (:592-599)
        plan, any_p = [], False
        for v in views:
            img = v["img"]
            B, H, W = img.shape[0], img.shape[-2], img.shape[-1]
            ts = v.get("true_shape", None)
            ts = torch.tensor([[H, W]] * B) if ts is None else torch.as_tensor(ts).cpu().reshape(B, 2).long()
            portrait = (ts[:, 1] < ts[:, 0]).tolist()
            if self.encoder.patch_embed_cls == "ManyAR_PatchEmbed":
                assert W >= H, f"img should be in landscape mode, but got {W=} {H=}"  # patch_embed.py:62
            enc_swap = [bool(p) and many_ar for p in portrait]
            if self.landscape_only:  # wrapper_yes: by definition the batch is stored in landscape mode, W >= H (misc.py:77-80)
                Ht, Wt = int(ts.min()), int(ts.max())
                head_hw = [(Wt, Ht) if p else (Ht, Wt) for p in portrait]
                head_swap = [bool(p) for p in portrait]
            else:                    # wrapper_no
                if not bool((ts[0:1] == ts).all()):
                    raise AssertionError("true_shape must be all identical")  # utils/misc.py:69
                head_hw = [tuple(ts[0].tolist())] * B
                head_swap = [False] * B
            any_p = any_p or any(enc_swap) or any(head_swap) or any(tuple(hw) != (H, W) for hw in head_hw)
            plan.append(dict(enc_swap=enc_swap, head_hw=head_hw, head_swap=head_swap))
        return dict(views=plan, any_portrait=any_p)

    def _forward_eager(self, views, profiling=False, _emb_rows=None, host_outputs=False):
        dev = views[0]["img"].device
        if dev.type != "cuda":
            raise F3RError(f"fast3r_amd.Fast3R runs only on a ROCm GPU (views are on {dev}); there is no CPU fallback")
        if any(v["img"].device != dev for v in views) or next(self.parameters()).device != dev:
            raise F3RError("fast3r_amd.Fast3R: the model and every view must live on the same device")
        with torch.cuda.device(dev):  # kernels launch on the CURRENT device / stream: make that the tensors' device
            return self._forward_on_device(views, profiling, _emb_rows, dev, host_outputs)

    def _forward_on_device(self, views, profiling, _emb_rows, dev, host_outputs=False):
        prof = {} if profiling else None
        lp = self.compute_dtype
        pk = self._pack(dev)
        enc, dec = self.encoder, self.decoder
        ps = enc.patch_size
        sh = self.sharding
        N_total = len(views)
        v_lo, v_hi = (0, N_total) if sh is None else sh.my_range(N_total)
        if sh is not None and N_total < sh.world:
            raise ValueError(f"view sharding needs at least one view per rank ({N_total} views, {sh.world} ranks)")
        my_views = views[v_lo:v_hi]
        n_loc = len(my_views)
        B = views[0]["img"].shape[0]

        # ---- encode (fast3r.py:250-296): every (view, sample) image, grouped by its (possibly un-transposed) size
        t0 = time.time()
        plan = self._orientation_plan(views) if _emb_rows is None else dict(views=[None] * N_total, any_portrait=False)
        Ps, feats = [], [[None] * B for _ in range(n_loc)]  # feats[i][b] = (hi [P][D], lo or None)
        if not plan["any_portrait"] and all(tuple(v["img"].shape[-2:]) == tuple(my_views[0]["img"].shape[-2:]) for v in my_views) and n_loc > 0:
            imgs = torch.cat([v["img"] for v in my_views], dim=0).float()
            with ops.family("transformer_linears"):   # (bench.py roofline.others: which family a GEMM-like launch is booked under)
                f, flo, P, grid = self._encode(imgs, pk)
            f = f.view(n_loc, B, P, -1)
            flo = None if flo is None else flo.view(n_loc, B, P, -1)
            for i in range(n_loc):
                for b in range(B):
                    feats[i][b] = (f[i, b], None if flo is None else flo[i, b])
            Ps = [P] * n_loc
            head_grid = [[grid] * B for _ in range(n_loc)]
            head_swap = [[False] * B for _ in range(n_loc)]
        else:
            groups = {}  # image size as encoded -> [(i, b, image (3, H', W'))]
            head_grid, head_swap = [], []
            for i, v in enumerate(my_views):
                pl = plan["views"][v_lo + i]
                img = v["img"].float()
                P_view = (img.shape[-2] // ps) * (img.shape[-1] // ps)
                Ps.append(P_view)
                grids_i = []
                for b in range(B):
                    im = img[b].transpose(-1, -2) if pl["enc_swap"][b] else img[b]
                    groups.setdefault(tuple(im.shape[-2:]), []).append((i, b, im))
                    hh, ww = pl["head_hw"][b]
                    assert hh % ps == 0 and ww % ps == 0 and (hh // ps) * (ww // ps) == P_view, \
                        f"true_shape {pl['head_hw'][b]} does not describe the {tuple(img.shape[-2:])} image of view {v_lo + i}"
                    grids_i.append((hh // ps, ww // ps))
                head_grid.append(grids_i)
                head_swap.append(list(pl["head_swap"]))
            untranspose = isinstance(enc, DinoEncoder)  # its portrait tokens go back to the stored (landscape) order (fast3r.py:601-622)
            for _, items in groups.items():
                with ops.family("transformer_linears"):
                    f, flo, P, (gh_, gw_) = self._encode(torch.stack([im for _, _, im in items]).contiguous(), pk)
                f = f.view(len(items), P, -1)
                flo = None if flo is None else flo.view(len(items), P, -1)
                for j, (i, b, _) in enumerate(items):
                    fj, fl = f[j], (None if flo is None else flo[j])
                    if untranspose and plan["views"][v_lo + i]["enc_swap"][b]:
                        fj = fj.view(gh_, gw_, -1).transpose(0, 1).reshape(P, -1)
                        fl = None if fl is None else fl.view(gh_, gw_, -1).transpose(0, 1).reshape(P, -1)
                    feats[i][b] = (fj, fl)
        if profiling:
            torch.cuda.synchronize()
            prof["encode_images_time"] = time.time() - t0

        # ---- image ids (fast3r.py:339-348 / 702-743): drawn once for all N views (rank 0 decides when sharded)
        t1 = time.time()
        if _emb_rows is not None:  # graph capture / replay: the rows sit in a captured buffer the caller fills before each replay
            emb_rows = _emb_rows
        else:
            ids = dec.draw_image_ids(B, N_total)
            if sh is not None:
                ids = sh.broadcast_ids(ids, dev)
            emb_rows = dec.image_idx_emb.to(dev)[ids.to(dev)]  # (B, N_total, D) fp32 rows of the table
        if profiling:
            prof["pos_emb_time"] = time.time() - t1

        # ---- fusion decoder (fast3r.py:768-808) per sample; local tokens are queries, K/V cover all views
        if profiling:
            torch.cuda.synchronize()
        t2 = time.time()
        D = dec.embed_dim
        T_loc = sum(Ps)
        hook_toks = []  # per sample: [4] (plane [T_loc][C], low plane or None)
        for b in range(B):
            enc_hi = torch.cat([feats[i][b][0] for i in range(n_loc)], dim=0) if n_loc > 1 else feats[0][b][0]
            enc_hi = enc_hi.contiguous()
            enc_lo = None
            if feats[0][b][1] is not None:
                enc_lo = (torch.cat([feats[i][b][1] for i in range(n_loc)], dim=0) if n_loc > 1 else feats[0][b][1]).contiguous()
            kv_dim = getattr(pk["dec"][0], "kv_dim", None) or D
            kvx = None
            if sh is not None:
                # every rank holds the same list of views, so it knows every rank's token count without asking (no per-forward collective)
                tok = [(v["img"].shape[-2] // ps) * (v["img"].shape[-1] // ps) for v in views]
                t_all = [sum(tok[slice(*split_range(N_total, sh.world, r))]) for r in range(sh.world)]
                assert t_all[sh.rank] == T_loc
                kvx = sh.make_kv_exchange(T_loc, kv_dim, lp, dev, n_heads=dec.num_heads, q_dim=D, t_all=t_all)
            with ops.family("transformer_linears"):
                hook_toks.append(self._decode_sample(pk, enc_hi, enc_lo, Ps, emb_rows[b], v_lo, kvx))
            if self.debug_taps is not None:
                self.debug_taps.setdefault("hooks", []).append([t[0].float().cpu() for t in hook_toks[-1]])
        del feats
        if profiling:
            torch.cuda.synchronize()
            prof["decoder_time"] = time.time() - t2
            prof["head_prepare_input_time"] = 0.0  # hooks are consumed in place: no rearrange step (fast3r.py:385-398)

        # ---- heads (fast3r.py:407-485), per sample in chunks of max_parallel_views_for_head consecutive views with one token grid
        t3 = time.time()
        results = [{} for _ in range(n_loc)]
        per_view = [[None] * B for _ in range(n_loc)]  # per (view, sample): {name: tensor (H, W[, 3])}
        offs = np.concatenate([[0], np.cumsum(Ps)]).tolist()
        step = max(1, self.max_parallel_views_for_head)
        heads = [("pts3d_in_other_view", "conf", pk["head"])]
        if pk["head_local"] is not None:
            heads.append(("pts3d_local", "conf_local", pk["head_local"]))
        sink = _HostSink(dev, B) if (host_outputs and sh is None) else None
        for b in range(B):
            i0 = 0
            while i0 < n_loc:
                i1 = i0 + 1
                gh, gw = head_grid[i0][b]
                chunk = self._head_chunk_views(step, gh, gw)
                while i1 < n_loc and i1 - i0 < chunk and head_grid[i1][b] == head_grid[i0][b]:
                    i1 += 1
                toks = [(t[0][offs[i0]:offs[i1]], None if t[1] is None else t[1][offs[i0]:offs[i1]]) for t in hook_toks[b]]
                for i in range(i0, i1):
                    per_view[i][b] = {}
                for pname, cname, hk in heads:
                    with ops.family("head_convs"):
                        pts, conf = self._dpt(hk, toks, i1 - i0, gh, gw)
                    for i in range(i0, i1):
                        sw = head_swap[i][b]
                        per_view[i][b][pname] = pts[i - i0].swapaxes(0, 1) if sw else pts[i - i0]  # transposed(): misc.py:105-106
                        if conf is not None:
                            per_view[i][b][cname] = conf[i - i0].swapaxes(0, 1) if sw else conf[i - i0]
                    if sink is not None:  # this chunk's outputs start their way to the host while the next chunk's heads run
                        sink.push([(i, b, nm, per_view[i][b][nm]) for i in range(i0, i1) for nm in (pname, cname) if nm in per_view[i][b]], (pts, conf))
                i0 = i1
        for i in range(n_loc):
            for name in per_view[i][0]:
                if sink is not None:
                    results[i][name] = sink.result(i, name)
                elif B == 1:   # one sample: the (1, H, W[, 3]) result IS the head chunk's row (no copy: 4 N device-to-device copies per forward otherwise)
                    results[i][name] = per_view[i][0][name].unsqueeze(0)
                else:
                    results[i][name] = torch.stack([per_view[i][b][name] for b in range(B)], dim=0)
        if sink is not None:
            sink.finish()
        if sh is not None:
            results = sh.gather_results(results, N_total, dev)
        if profiling:
            torch.cuda.synchronize()
            prof["head_forward_time"] = time.time() - t3
            prof["total_time"] = time.time() - t0
            return results, prof
        return results
