"""Per-stage parity breakdown on the GPU box (diagnostic, not a test): where does the end-to-end rel-L2 come from?
For each (model, weight distribution, operand dtype): DPT-input taps vs the oracle's, final outputs vs the oracle's, and
head-only error (product heads fed the oracle's exact hooks)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from fast3r_amd import Fast3R  # noqa: E402
from fast3r_amd.synthetic import make_views, synth_state_dict, tiny_args, vit_large_args  # noqa: E402
from oracle import fast3r_oracle as O  # noqa: E402

DEV = "cuda"


def run(tag, args, n_views, hw, dist):
    enc, dec, head = args
    shp = {k: tuple(v.shape) for k, v in Fast3R(enc, dec, head).state_dict().items()}
    sd = synth_state_dict(shp, 0, dist)
    views = make_views(n_views, hw, hw)
    t0 = time.time()
    with torch.no_grad():
        torch.manual_seed(1234)
        ref, taps = O.forward(views, sd, enc, dec, head, return_taps=True)
    t_or = time.time() - t0
    for dt in (torch.float16, torch.bfloat16):
        m = Fast3R(enc, dec, head, compute_dtype=dt).eval()
        m.load_state_dict(sd)
        m = m.to(DEV)
        m.debug_taps = {}
        gv = [dict(v, img=v["img"].to(DEV)) for v in views]
        with torch.no_grad():
            torch.manual_seed(1234)
            out = m(gv)
            row = {"case": tag, "dist": dist, "dtype": str(dt).split(".")[-1], "oracle_s": round(t_or, 1)}
            for i, name in enumerate(("hook0_enc", "hook_mid", "hook_3q", "hook_last")):
                row[name] = O.rel_l2(m.debug_taps["hooks"][0][i], taps["hooks"][i][0])
            for k in ref[0]:
                row[k] = max(O.rel_l2(o[k].cpu(), r[k]) for o, r in zip(out, ref))
            # head-only: exact oracle hooks (rounded once to the operand type) through the product heads
            pk = m._pack(torch.device(DEV))
            P = taps["hooks"][0].shape[1] // n_views
            g = hw // 16
            toks = [t[0].to(dt).to(DEV).contiguous() for t in taps["hooks"]]
            pts, conf = m._dpt(pk["head"], [(t, None) for t in toks], n_views, g, g)
            ref_pts = torch.cat([r["pts3d_in_other_view"] for r in ref])
            ref_conf = torch.cat([r["conf"] for r in ref])
            row["headonly_pts"] = O.rel_l2(pts.cpu(), ref_pts)
            row["headonly_conf"] = O.rel_l2(conf.cpu(), ref_conf)
        print(json.dumps(row), flush=True)
        del m


if __name__ == "__main__":
    for dist in ("default", "hot"):
        run("tiny_3x64", tiny_args(), 3, 64, dist)
    for dist in ("default", "hot"):
        run("vitl_3x256", vit_large_args(), 3, 256, dist)
