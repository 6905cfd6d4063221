"""TEST INFRASTRUCTURE ONLY -- a stand-in for `torch.hub.load('facebookresearch/dinov2', 'dinov2_vitl14')` (fast3r.py:569).

The reference's DinoEncoder downloads its backbone; there is no network here and the DINOv2 code is not vendored in the reference.  To run
the REAL reference around it -- `DinoEncoder.forward` (landscape / portrait split, token transposes, fast3r.py:574-651), the decoder on
`enc_embed_dim` features, the DPT heads at patch size 14 (`Interpolate(scale_factor=14/8)`), postprocess -- oracle/make_golden.py swaps
`torch.hub.load` for `DinoStub`: an nn.Module with DINOv2's parameter names (`DinoVisionTransformer`: cls_token, pos_embed, mask_token,
patch_embed.proj, blocks.N.{norm1, attn.qkv, attn.proj, ls1.gamma, norm2, mlp.fc1, mlp.fc2, ls2.gamma}, norm) whose
`forward_features` is the restatement in oracle/fast3r_oracle.py::dino_vit_patch_tokens.  What such a golden pins: everything of the
`dino_v2` configuration EXCEPT the inside of the backbone, which stays "parity unpinned" (restated from the published model code).
"""
import torch
import torch.nn as nn

from oracle import fast3r_oracle as O


class _LS(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim, mlp_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim)
        self.ls1 = _LS(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.ls2 = _LS(dim)


class _PatchEmbed(nn.Module):
    def __init__(self, dim, ps):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, ps, ps)


class DinoStub(nn.Module):
    def __init__(self, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, pos_grid=37, patch_size=14):
        super().__init__()
        self.embed_dim, self.num_heads, self.depth, self.patch_size = embed_dim, num_heads, depth, patch_size
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + pos_grid * pos_grid, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        self.patch_embed = _PatchEmbed(embed_dim, patch_size)
        self.blocks = nn.ModuleList([_Block(embed_dim, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)

    def forward_features(self, x):
        sd = {"encoder.model." + k: v for k, v in self.state_dict().items()}
        args = dict(patch_size=self.patch_size, num_heads=self.num_heads, depth=self.depth)
        toks, _ = O.dino_vit_patch_tokens(x, sd, args)
        return {"x_norm_patchtokens": toks}
