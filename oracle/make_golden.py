"""TEST INFRASTRUCTURE -- generates tests/golden/*.pt by running the REAL reference on CPU (true fp32, dtype="32").

Run in the build container only (needs /root/reference):   python -m oracle.make_golden
The reference ships no golden vectors, so these are "outputs of the reference itself run
here": each fixture stores the constructor args, the input recipe (seeds; inputs and weights are regenerated
bit-identically by fast3r_amd/synthetic.py), the image ids the reference drew, and the reference outputs.
Weights are NOT stored (the DPT heads alone are 2 x 20.5 M parameters): both sides rebuild them from
synth_state_dict(shapes, seed).
"""
import copy
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast3r_amd.synthetic import make_views, synth_state_dict, tiny_args  # noqa: E402
from oracle.ref_loader import load_reference  # noqa: E402

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name -> (tiny_args kwargs, list of (H, W) per view, batch, weight seed, rng seed, weight distribution)
# "default" = the reference's own random-init distribution (the BASELINE.md section 3 protocol the 1e-3 bar is stated on);
# "hot" = N(0, 1/fan_in) stress weights (sharp attention, noise-amplifying heads), see fast3r_amd/synthetic.py.
CASES = {
    "tiny_3x64": (dict(), [(64, 64)] * 3, 1, 0, 1234, "default"),
    "tiny_mixed": (dict(enc_depth=1), [(64, 64), (48, 64), (64, 80)], 1, 1, 99, "default"),
    "tiny_b2_seqids": (dict(random_image_idx_embedding=False, with_local_head=False), [(32, 48)] * 4, 2, 2, 7, "default"),
    "tiny_oddgrid": (dict(enc_depth=1, attn_bias_for_inference_enabled=False), [(112, 160)] * 2, 1, 3, 5, "default"),
    "tiny_hot_3x64": (dict(), [(64, 64)] * 3, 1, 0, 1234, "hot"),
    "tiny_llama_3x64": (dict(decoder_type="llama"), [(64, 64)] * 3, 1, 4, 31, "default"),
    "tiny_llama_seqids_b2": (dict(decoder_type="llama", random_image_idx_embedding=False, enc_depth=1, llama_layers=14), [(32, 48)] * 4, 2, 5, 3, "default"),
    # grouped-query (4 query heads on 2 K/V heads) + causal LlamaDecoder (components/llama.py:195-198,229-239): not the released config
    "tiny_llama_gqa_causal": (dict(decoder_type="llama", embed_dim=256, num_heads=4, llama_kv_heads=2, llama_causal=True, enc_depth=1,
                                   llama_layers=12), [(48, 64)] * 3, 1, 7, 17, "default"),
    # multi-query attention: 4 query heads on ONE K/V head (repeat_kv accepts any divisor, components/llama.py:125-134)
    "tiny_llama_mqa": (dict(decoder_type="llama", embed_dim=256, num_heads=4, llama_kv_heads=1, enc_depth=1, llama_layers=12), [(48, 64)] * 3, 1, 10, 23, "default"),
    # the training-config pair ManyAR_PatchEmbed + landscape_only=True (configs/model/fast3r.yaml:55,77): images are STORED landscape
    # (48 x 64) and `true_shape` says which samples are portrait pictures: view 0 all landscape, view 1 all portrait, view 2 mixed
    "tiny_portrait_b2": (dict(enc_depth=1, patch_embed_cls="ManyAR_PatchEmbed", landscape_only=True), [(48, 64)] * 3, 2, 6, 11, "default"),
    # encoder_type 'dino_v2' (fast3r.py:79-83,561-651) with the backbone that torch.hub would download replaced by oracle/dino_stub.py (DINOv2's
    # parameter names, forward_features = our restatement): the reference's DinoEncoder.forward incl. the portrait split, its decoder on
    # enc_embed_dim features and its DPT heads at patch size 14 (Interpolate 14/8) run for real; the inside of the backbone stays unpinned
    "tiny_dino_portrait_b2": (dict(patch_embed_cls="dino", enc_depth=2, landscape_only=True), [(56, 70)] * 3, 2, 8, 3, "default"),
    # a fusion decoder whose heads are not 64 wide (configs/experiment/model_scaling/model_scaling_huge.yaml:13-15: 1280 / 16 = 80):
    # encoder 128 / 2 heads, decoder 320 / 4 heads (the QKV epilogue splits q / k / v on 64-column groups: widths are multiples of 64,
    # as 1280 is)
    "tiny_hd80_3x64": (dict(enc_depth=1, dec_embed_dim=320, dec_num_heads=4), [(64, 64), (48, 64), (64, 64)], 1, 9, 21, "default"),
}
TRUE_SHAPES = {"tiny_portrait_b2": [[[48, 64], [48, 64]], [[64, 48], [64, 48]], [[64, 48], [48, 64]]],
               "tiny_dino_portrait_b2": [[[56, 70], [56, 70]], [[70, 56], [70, 56]], [[70, 56], [56, 70]]]}


def views_for(shapes, batch, seed=1000, true_shapes=None):
    vs = []
    for i, (h, w) in enumerate(shapes):
        vs.append(make_views(1, h, w, batch, seed=seed + i)[0])
        vs[-1]["idx"], vs[-1]["instance"], vs[-1]["label"] = i, str(i), f"syn/{i}"
        if true_shapes is not None:
            vs[-1]["true_shape"] = torch.tensor(true_shapes[i], dtype=torch.int32)
    return vs


def main(only=None, check=False):
    """check: regenerate in memory and compare with the committed fixtures instead of writing (exit code 1 on a difference)."""
    warnings.filterwarnings("ignore")
    Fast3R, inference = load_reference()
    os.makedirs(OUT_DIR, exist_ok=True)
    bad = []
    for name, (kw, shapes, batch, wseed, rseed, wdist) in CASES.items():
        if only and name not in only:
            continue
        enc, dec, head = tiny_args(**kw)
        hub_load = torch.hub.load
        if enc["encoder_type"] == "dino_v2":  # no network: hand the reference a local backbone instead of the download (oracle/dino_stub.py)
            from oracle.dino_stub import DinoStub
            torch.hub.load = lambda *a, **k: DinoStub(embed_dim=enc["embed_dim"], depth=enc["depth"], num_heads=enc["num_heads"],
                                                      mlp_ratio=enc["mlp_ratio"], pos_grid=enc["pos_grid"])
        try:
            model = Fast3R(copy.deepcopy(enc), copy.deepcopy(dec), copy.deepcopy(head)).eval()
        finally:
            torch.hub.load = hub_load
        shp = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(synth_state_dict(shp, wseed, wdist), strict=True)
        views = views_for(shapes, batch, true_shapes=TRUE_SHAPES.get(name))
        # capture the ids the reference draws (fast3r.py:740-743) without touching its code path
        torch.manual_seed(rseed)
        ids = None
        if dec["random_image_idx_embedding"]:
            seed = torch.randint(0, 2 ** 32, (1,)).item()
            g = torch.Generator().manual_seed(seed)
            ids = torch.zeros(batch, len(shapes), dtype=torch.long)
            for b in range(batch):
                ids[b, 1:] = torch.randperm(999, generator=g)[: len(shapes) - 1] + 1
        torch.manual_seed(rseed)
        out = inference(copy.deepcopy(views), model, torch.device("cpu"), dtype="32", verbose=False)
        preds = out["preds"]
        fix = dict(name=name, tiny_kwargs=kw, shapes=shapes, batch=batch, weight_seed=wseed, weight_dist=wdist, rng_seed=rseed,
                   state_shapes=shp, image_ids=ids, true_shapes=TRUE_SHAPES.get(name),
                   preds=[{k: v.clone() for k, v in p.items()} for p in preds],
                   torch_version=torch.__version__)
        path = os.path.join(OUT_DIR, name + ".pt")
        if check:
            old = torch.load(path, weights_only=False)
            same = old["state_shapes"] == shp and all(torch.equal(a[k], b[k]) for a, b in zip(old["preds"], fix["preds"]) for k in b) and \
                (old["image_ids"] is None) == (ids is None) and (ids is None or torch.equal(old["image_ids"], ids))
            print(name, "bit-identical to the committed fixture" if same else "DIFFERS from the committed fixture")
            if not same:
                bad.append(name)
            continue
        torch.save(fix, path)
        print(name, os.path.getsize(path) // 1024, "KB", {k: tuple(v.shape) for k, v in preds[0].items()})
    if bad:
        sys.exit(1)


if __name__ == "__main__":
    names = set(a for a in sys.argv[1:] if not a.startswith("--"))
    main(names, check="--check" in sys.argv)  # optional: only the named cases (the others stay as committed)
