"""Fast3R.calibrate_precision on the real-size model (ViT-L / ViT-L / 2 DPT heads, 8 views of 512 x 512) for three synthetic checkpoints -- the default
init, the N(0, 1 / fan_in) stress set and the heavy-tailed one -- with wall-clock times: what a user of a real checkpoint would see.
    python tools/calibrate_demo.py > profiles/r06_calibrate_precision_vit_large.json"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast3r_amd import Fast3R  # noqa: E402
from fast3r_amd.synthetic import make_views, synth_state_dict, vit_large_args  # noqa: E402

enc, dec, head = vit_large_args()
shapes = {k: tuple(v.shape) for k, v in Fast3R(enc, dec, head).state_dict().items()}
views = [dict(v, img=v["img"].cuda()) for v in make_views(8, 512, 512)]
out = {"what": "Fast3R.calibrate_precision(views[:8]) at ViT-L size on one MI355X: every 16-bit tier against precision='exact' on the same views and weights",
       "checkpoints": {}}
for dist in ("default", "hot", "heavy"):
    m = Fast3R(enc, dec, head).eval()
    m.load_state_dict(synth_state_dict(shapes, 0, dist=dist), strict=True)
    m = m.cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rep = m.calibrate_precision(views)
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    rep = m.calibrate_precision(views)   # weights of two formats stay packed: a repeat shows the forwards alone
    again = time.perf_counter() - t0
    out["checkpoints"][dist] = {"recommended": rep["recommended"], "worst_rel_l2": rep["worst"], "per_tier": rep["per_tier"],
                                "seconds_first_call_incl_weight_packing": first, "seconds_second_call": again, "forward_ms": rep["ms_incl_weight_packing"]}
    del m
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
