"""Shared test helpers: fixture loading, model construction from a golden fixture, comparators."""
import os

import torch

from fast3r_amd.synthetic import make_views, synth_state_dict, tiny_args

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = ["tiny_3x64", "tiny_mixed", "tiny_b2_seqids", "tiny_oddgrid", "tiny_hot_3x64", "tiny_llama_3x64", "tiny_llama_seqids_b2",
                "tiny_portrait_b2", "tiny_llama_gqa_causal", "tiny_dino_portrait_b2", "tiny_hd80_3x64", "tiny_llama_mqa"]


def load_golden(name):
    return torch.load(os.path.join(GOLDEN_DIR, name + ".pt"), weights_only=False)


def golden_views(fix, seed=1000):
    vs = []
    for i, (h, w) in enumerate(fix["shapes"]):
        v = make_views(1, h, w, fix["batch"], seed=seed + i)[0]
        v["idx"], v["instance"], v["label"] = i, str(i), f"syn/{i}"
        if fix.get("true_shapes") is not None:
            v["true_shape"] = torch.tensor(fix["true_shapes"][i], dtype=torch.int32)
        vs.append(v)
    return vs


def golden_model_inputs(fix):
    """-> (encoder_args, decoder_args, head_args, state_dict, views)"""
    enc, dec, head = tiny_args(**fix["tiny_kwargs"])
    sd = synth_state_dict(fix["state_shapes"], fix["weight_seed"], fix["weight_dist"])
    return enc, dec, head, sd, golden_views(fix)


def rel_l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def views_to(views, device):
    out = []
    for v in views:
        v = dict(v)
        v["img"] = v["img"].to(device)
        out.append(v)
    return out
