"""One small launch of the hand-scheduled attention kernel with everything printed (first-light / debugging on the GPU box)."""
import math
import sys

import torch

sys.path.insert(0, ".")
from fast3r_amd import ops  # noqa: E402

dt = torch.float16 if len(sys.argv) < 2 or sys.argv[1] == "fp16" else torch.bfloat16
Tq, Tk, H = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (512, 128, 1)
g = torch.Generator().manual_seed(0)
qs = (torch.randn((Tq, H * 64), generator=g) * 0.27).to(dt)
k = (torch.randn((Tk, H * 64), generator=g) * 1.5).to(dt)
v = torch.randn((Tk, H * 64), generator=g).to(dt)
vt = torch.zeros((H * 64, ops.vt_ld(Tk)), dtype=dt)
vt[:, :Tk] = v.t()
o = torch.full((Tq, H * 64), float("nan"), dtype=dt, device="cuda")
print("launching", Tq, Tk, H, dt, flush=True)
ops.attention(qs.cuda(), o, H, 1.0, [(k.cuda(), vt.cuda(), Tk, 0, 0)], q_prescaled=True, kernel_sel=2)
torch.cuda.synchronize()
print("launched", flush=True)
qh = qs.double().reshape(Tq, H, 64).transpose(0, 1)
kh = k.double().reshape(Tk, H, 64).transpose(0, 1)
vh = v.double().reshape(Tk, H, 64).transpose(0, 1)
ref = (((qh @ kh.transpose(1, 2)) * math.log(2.0)).softmax(-1) @ vh).transpose(0, 1).reshape(Tq, H * 64)
got = o.double().cpu()
err = (got - ref).abs()
print("nan", int(torch.isnan(got).sum()), "max err", float(err.nan_to_num(9.0).max()), "rel-l2", float((got - ref).nan_to_num(9.0).norm() / ref.norm()))
if err.nan_to_num(9.0).max() > 1e-2:
    bad = (err.nan_to_num(9.0) > 1e-2)
    rows = bad.any(1).nonzero().flatten()
    print("bad rows", rows[:40].tolist(), "count", len(rows))
    cols = bad.any(0).nonzero().flatten()
    print("bad cols", cols[:70].tolist(), "count", len(cols))
    r = int(rows[0])
    print("row", r, "got", got[r, :8].tolist(), "ref", ref[r, :8].tolist())
