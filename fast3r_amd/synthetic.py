"""Deterministic synthetic weights / views / model configs.

There is no network (no released checkpoint, no dataset), so the benchmark, the
smoke test and the parity tests all run on synthetic data of the real shapes
(BASELINE.md section 3).  Everything here is a pure function of its seed so the
build container (where the reference can be imported) and the GPU box (where it
cannot) regenerate bit-identical tensors: torch's CPU generator is the only
source of randomness.
"""
import math
import zlib

import torch


def vit_large_args(img_size=512, attn_implementation="flash_attention", random_image_idx_embedding=True,
                   attn_bias_for_inference_enabled=True, max_image_idx=1000):
    """The (inferred) released `Fast3R_ViT_Large_512` constructor args.  max_image_idx > 1000 adds the
    fast3r_amd-only decoder argument that extends the image-index table past the reference's 1000 rows (BASELINE configs[4], N=1500)."""
    encoder_args = dict(encoder_type="croco", img_size=img_size, patch_size=16, patch_embed_cls="PatchEmbedDust3R",
                        embed_dim=1024, num_heads=16, depth=24, mlp_ratio=4, pos_embed="RoPE100",
                        attn_implementation=attn_implementation)
    decoder_args = dict(decoder_type="fast3r", random_image_idx_embedding=random_image_idx_embedding,
                        enc_embed_dim=1024, embed_dim=1024, num_heads=16, depth=24, mlp_ratio=4.0, qkv_bias=True,
                        drop=0.0, attn_drop=0.0, attn_implementation=attn_implementation,
                        attn_bias_for_inference_enabled=attn_bias_for_inference_enabled)
    if max_image_idx != 1000:
        decoder_args["max_image_idx"] = int(max_image_idx)
    head_args = dict(head_type="dpt", output_mode="pts3d", landscape_only=False,
                     depth_mode=["exp", -float("inf"), float("inf")], conf_mode=["exp", 1, float("inf")],
                     patch_size=16, with_local_head=True)
    return encoder_args, decoder_args, head_args


def tiny_args(embed_dim=128, num_heads=2, enc_depth=2, dec_depth=12, img_size=64, with_local_head=True,
              random_image_idx_embedding=True, attn_implementation="pytorch_naive",
              attn_bias_for_inference_enabled=True, decoder_type="fast3r", llama_layers=12,
              patch_embed_cls="PatchEmbedDust3R", landscape_only=False, llama_kv_heads=None, llama_causal=False,
              dec_embed_dim=None, dec_num_heads=None):
    """Small model of the same family (head_dim 64 unless dec_embed_dim / dec_num_heads give the fusion decoder another width, as in
    configs/experiment/model_scaling/model_scaling_huge.yaml; decoder depth must be > 9, fast3r.py:137)."""
    encoder_args = dict(encoder_type="croco", img_size=img_size, patch_size=16, patch_embed_cls=patch_embed_cls,
                        embed_dim=embed_dim, num_heads=num_heads, depth=enc_depth, mlp_ratio=4, pos_embed="RoPE100",
                        attn_implementation=attn_implementation)
    decoder_args = dict(decoder_type="fast3r", random_image_idx_embedding=random_image_idx_embedding,
                        enc_embed_dim=embed_dim, embed_dim=dec_embed_dim or embed_dim, num_heads=dec_num_heads or num_heads, depth=dec_depth,
                        mlp_ratio=4.0, qkv_bias=True, drop=0.0, attn_drop=0.0,
                        attn_implementation=attn_implementation,
                        attn_bias_for_inference_enabled=attn_bias_for_inference_enabled)
    if patch_embed_cls == "dino":  # DinoEncoder (fast3r.py:561-651) scaled down: the reference class is always ViT-L/14 with a 37 x 37 position grid
        encoder_args = dict(encoder_type="dino_v2", patch_size=14, embed_dim=embed_dim, depth=enc_depth, num_heads=num_heads, mlp_ratio=4, pos_grid=5)
    if decoder_type == "llama":
        # configs/experiment/llama_dec/llama_dec.yaml:52-66 merged over configs/model/fast3r.yaml: the base keys stay in the dict (they
        # disappear into LlamaDecoder's **kwargs) and `depth` -- not n_layers -- is what the heads read (fast3r.py:137-148)
        decoder_args = dict(decoder_type="llama", random_image_idx_embedding=random_image_idx_embedding, enc_embed_dim=embed_dim,
                            embed_dim=embed_dim, n_layers=llama_layers, n_heads=num_heads, n_kv_heads=llama_kv_heads, multiple_of=64,
                            ffn_dim_multiplier=None, norm_eps=1e-5, rope_theta=10000, max_seq_len=1000, is_causal=llama_causal,
                            depth_init=True, depth=dec_depth, num_heads=num_heads, mlp_ratio=4.0, qkv_bias=True)
    head_args = dict(head_type="dpt", output_mode="pts3d", landscape_only=landscape_only,
                     depth_mode=["exp", -float("inf"), float("inf")], conf_mode=["exp", 1, float("inf")],
                     patch_size=14 if patch_embed_cls == "dino" else 16, with_local_head=with_local_head)
    return encoder_args, decoder_args, head_args


def synth_tensor(key: str, shape, seed: int = 0, dist: str = "default") -> torch.Tensor:
    """One tensor of a synthetic state_dict; a pure function of (key, shape, seed, dist).

    dist="default": the distribution of the reference's own random init (torch defaults, which is all the reference
        applies on this path): Linear / Conv2d / ConvTranspose2d weight and bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)).
        This is the protocol of BASELINE.md section 3 / SURVEY.md section 8c (the 1e-3 rel-L2 yardstick).
    dist="hot": weights ~ N(0, 1/fan_in) (3x the variance): attention logits have std ~1.3 instead of ~0.15, so the
        softmax is far from uniform and rounding noise is amplified instead of averaged away -- a stress distribution.
    dist="heavy" (round 5, a second stress distribution): weights ~ Student-t with 4 degrees of freedom scaled to variance 1/fan_in
        (outlier weights: a few products dominate a dot product, which is where 16-bit operand rounding shows), LayerNorm gains
        log-uniform in [0.2, 5] per channel (channels of very different scale in every GEMM operand).
    LayerNorm gamma ~ 1 + 0.1 N(0,1), beta ~ 0.02 N(0,1) in the first two (so gamma/beta mistakes cannot hide).
    """
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
    shape = tuple(shape)
    is_norm = ".norm" in key or key.endswith("_norm.weight") or key.endswith("_norm.bias") or key.endswith(".gamma")  # LayerScale ~ 1 + noise
    if key.endswith("pos_embed") or key.endswith("cls_token"):  # DINOv2 position table / class token: O(1) so that they matter
        return 0.5 * torch.randn(shape, generator=g)
    if is_norm:
        if key.endswith("weight"):
            if dist == "heavy" and not key.endswith(".gamma"):
                return torch.exp((torch.rand(shape, generator=g) * 2 - 1) * math.log(5.0))
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.02 * torch.randn(shape, generator=g)
    if len(shape) >= 2:
        if key.endswith("act_postprocess.0.1.weight") or key.endswith("act_postprocess.1.1.weight"):
            fan_in = shape[1] * shape[2] * shape[3]  # ConvTranspose2d (Cin, Cout, k, k): torch's fan_in = size(1) * k * k
            eff = shape[0]                            # but each output pixel sums over Cin only
        else:
            fan_in = eff = math.prod(shape[1:])
        if dist == "hot":
            gain = 0.1 if key.endswith("dpt.head.4.weight") else 1.0  # keeps |xyz| = O(1) ahead of expm1 / exp
            return torch.randn(shape, generator=g) * (gain / math.sqrt(eff))
        if dist == "heavy":
            gain = 0.1 if key.endswith("dpt.head.4.weight") else 1.0
            z = torch.randn(shape, generator=g)
            chi2 = sum(torch.randn(shape, generator=g) ** 2 for _ in range(4))
            t4 = z / torch.sqrt(chi2 / 4.0)                  # Student-t, nu = 4: variance nu / (nu - 2) = 2
            return t4 * (gain / math.sqrt(2.0 * eff))
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    if dist in ("hot", "heavy"):
        return 0.02 * torch.randn(shape, generator=g)
    # bias: U(+-1/sqrt(fan_in)); fan_in is not recoverable from the bias shape alone, use the layer width as proxy
    return (torch.rand(shape, generator=g) * 2 - 1) * (1.0 / math.sqrt(max(shape[0], 1)))


def synth_state_dict(shapes: dict, seed: int = 0, dist: str = "default") -> dict:
    """shapes: {key: shape} (e.g. from `model.state_dict()`) -> deterministic fp32 state_dict.

    `scratch.layer_rn.{i}` aliases `scratch.layer{i+1}_rn` (dpt_block.py:79-86); aliases get equal values.
    """
    out = {}
    for k, shp in shapes.items():
        canon = k
        for i in range(4):
            canon = canon.replace(f"scratch.layer_rn.{i}.", f"scratch.layer{i + 1}_rn.")
        out[k] = synth_tensor(canon, shp, seed, dist)
    return out


def make_views(n_views: int, height: int = 512, width: int = 512, batch: int = 1, seed: int = 1000):
    """BASELINE.md section 3 inputs: img ~ U(-1,1) from Generator(seed+i); true_shape int32 [[H,W]]."""
    views = []
    for i in range(n_views):
        g = torch.Generator().manual_seed(seed + i)
        img = torch.rand((batch, 3, height, width), generator=g) * 2.0 - 1.0
        views.append(dict(img=img, true_shape=torch.tensor([[height, width]] * batch, dtype=torch.int32),
                          idx=i, instance=str(i), dataset="syn", label=f"syn/{i}"))
    return views
