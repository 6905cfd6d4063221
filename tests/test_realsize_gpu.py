"""Parity at the sizes BASELINE.json names, on the GPU, where the CPU oracle cannot run inside a test budget.

The checker is the product's own fp32-equivalent path (precision="exact": every GEMM / conv operand as an fp16 hi + lo pair, the
attention core in plain fp32 -- f3r_exact.hip -- or, from 8192 keys on, as three-plane MFMA products with an fp32 softmax --
f3r_exact_mfma.hip, tied to the plain form and to float64 by tests/test_exact_mfma_gpu.py), which is itself pinned on the reference: to 3e-7 on the reference's golden outputs
(test_e2e_gpu.py::test_exact_mode_matches_reference_golden_to_fp32_noise), to 6e-7 on ViT-L at N=3 vs the CPU oracle, and
(test_e2e_gpu.py::test_exact_mode_is_anchored_on_the_oracle_at_20k_tokens) at 20 480 tokens of the fusion decoder vs the CPU oracle.

  configs[2]  N=100 views 512^2, ViT-L encoder + fusion + heads: fp16/high and bf16/fast vs exact, every output <= 1e-3 rel-L2
  configs[3]  N=320 sharded 40 views per GPU: ONE GPU runs rank r's exact work (Fast3R.emulate_rank: 40 local views, local launch
              parking the softmax state + remote launch over the 7 other shards, whose K / V^T per layer are the ones an unsharded
              forward produced) and must reproduce the unsharded outputs of its views to <= 2e-4 (fp32 summation order only)
"""
import pytest
import torch

from helpers import rel_l2, views_to
from fast3r_amd import Fast3R
from fast3r_amd.synthetic import make_views, synth_state_dict, vit_large_args

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-3
_CACHE = {}


def _vitl():
    if "sd" not in _CACHE:
        enc, dec, head = vit_large_args()
        shapes = {k: tuple(v.shape) for k, v in Fast3R(enc, dec, head).state_dict().items()}
        _CACHE["args"] = (enc, dec, head)
        _CACHE["sd"] = synth_state_dict(shapes, 0)
    return _CACHE["args"], _CACHE["sd"]


def _build(dt, precision):
    (enc, dec, head), sd = _vitl()
    m = Fast3R(enc, dec, head, compute_dtype=dt, precision=precision).eval()
    m.load_state_dict(sd, strict=True)
    return m.to(DEV)


def test_vit_large_n100_matches_the_fp32_equivalent_path(built_lib):
    """BASELINE configs[2] at its real size: 100 views of 512^2 end to end."""
    views = views_to(make_views(100, 512, 512), DEV)
    m = _build(torch.float16, "exact")
    with torch.no_grad():
        torch.manual_seed(4321)  # (after the model is built: parameter initialisation draws from the same generator as the image ids)
        ref = [{k: v.cpu() for k, v in o.items()} for o in m(views)]
    del m
    torch.cuda.empty_cache()
    report = {}
    for dt, precision in ((torch.float16, "high"), (torch.bfloat16, "fast")):
        m = _build(dt, precision)
        with torch.no_grad():
            torch.manual_seed(4321)
            out = m(views)
        worst = {}
        for o, g in zip(out, ref):
            for k in g:
                worst[k] = max(worst.get(k, 0.0), rel_l2(o[k].cpu(), g[k]))
        report[(str(dt), precision)] = worst
        print(f"[parity] ViT-L N=100 512^2 {dt} {precision} vs exact: " + ", ".join(f"{k}={v:.2e}" for k, v in worst.items()))
        del m, out
        torch.cuda.empty_cache()
    for key, worst in report.items():
        assert max(worst.values()) <= TOL, (key, worst)


def test_rank_of_eight_at_n320_reproduces_the_unsharded_forward(built_lib):
    """BASELINE configs[3] (N=320, 40 views per GPU), the per-rank work through the MODEL on one GPU: the unsharded forward records
    every fusion layer's K / V^T (kv_tap: 24 x 1.3 GB); rank 3 of 8 is then emulated -- its 40 views, the two-launch attention over its
    own shard + the 7 others' captured K / V^T -- and must give the unsharded pointmaps of views 120..159."""
    N, world, rank = 320, 8, 3
    views = views_to(make_views(N, 512, 512), DEV)
    m = _build(torch.float16, "high")
    taps = []
    m.kv_tap = lambda k, vt: taps.append((k.clone(), vt.clone()))
    with torch.no_grad():
        torch.manual_seed(99)
        full = m(views)
    m.kv_tap = None
    assert len(taps) == 24
    T = N * 1024
    per = T // world

    def kv_source(layer, r, k_out, vt_out):
        k, vt = taps[layer]
        vt = vt.view(vt.shape[-2], vt.shape[-1])  # [1][D][ld] of the one decoder sample
        k_out.copy_(k[r * per:(r + 1) * per])
        vt_out.copy_(vt[:, r * per:(r + 1) * per])
    m.emulate_rank(rank, world, kv_source)
    with torch.no_grad():
        torch.manual_seed(99)
        mine = m(views)
    m.emulate_rank(None, 0)
    lo, hi = rank * (N // world), (rank + 1) * (N // world)
    assert len(mine) == hi - lo
    worst = {}
    for i in (0, 7, 19, 39):  # sampled views of the shard
        for k in full[lo + i]:
            worst[k] = max(worst.get(k, 0.0), rel_l2(mine[i][k], full[lo + i][k]))
    print(f"[parity] rank {rank}/{world} of N={N} vs unsharded: " + ", ".join(f"{k}={v:.2e}" for k, v in worst.items()))
    assert max(worst.values()) <= 2e-4, worst
