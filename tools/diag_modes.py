"""Where do the operand formats part ways at a given scene size?  ViT-L, synthetic weights: outputs and the four hooked decoder states
of precision fast / high / exact (and high with the general HIP attention kernel forced) against each other."""
import sys

import torch

sys.path.insert(0, ".")
from fast3r_amd import Fast3R, ops  # noqa: E402
from fast3r_amd.synthetic import make_views, synth_state_dict, vit_large_args  # noqa: E402


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


enc, dec, head = vit_large_args()
shapes = {k: tuple(v.shape) for k, v in Fast3R(enc, dec, head).state_dict().items()}
sd = synth_state_dict(shapes, 0)
orig_attention = ops.attention


def run(N, precision, force_hip=False):
    m = Fast3R(enc, dec, head, compute_dtype=torch.float16, precision=precision).eval()
    m.load_state_dict(sd)
    m = m.to("cuda")
    m.debug_taps = {}
    if force_hip:
        ops.attention = lambda *a, **k: orig_attention(*a, **dict(k, kernel_sel=1))
    try:
        views = [dict(v, img=v["img"].cuda()) for v in make_views(N, 512, 512)]
        with torch.no_grad():
            torch.manual_seed(4321)
            out = m(views)
    finally:
        ops.attention = orig_attention
    res = [{k: v.cpu() for k, v in o.items()} for o in out]
    taps = m.debug_taps["hooks"][0]
    del m
    torch.cuda.empty_cache()
    return res, taps


for N in [int(x) for x in (sys.argv[1:] or ["16", "48", "100"])]:
    runs = {"exact": run(N, "exact"), "high": run(N, "high"), "high_hip": run(N, "high", True), "fast": run(N, "fast")}
    for a, b in (("high", "exact"), ("high_hip", "exact"), ("fast", "exact"), ("high", "high_hip"), ("high", "fast")):
        (oa, ta), (ob, tb) = runs[a], runs[b]
        worst = {}
        for x, y in zip(oa, ob):
            for k in y:
                worst[k] = max(worst.get(k, 0.0), rel(x[k], y[k]))
        hooks = [rel(x, y) for x, y in zip(ta, tb)]
        print(f"N={N} {a} vs {b}: outputs " + ", ".join(f"{k}={v:.2e}" for k, v in worst.items()) + " | hooks 0,L/2,3L/4,L " + ", ".join(f"{h:.2e}" for h in hooks), flush=True)
