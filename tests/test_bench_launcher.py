"""bench.py as the driver starts it: `python bench.py --gpus N` must bring up N ranks by itself.  Exercised here on CPU
through the F3R_BENCH_DRYRUN switch (gloo backend, no kernels): the launcher, the per-rank view split, the collectives and the
single JSON line on rank 0's stdout are the real code; only the GPU work is replaced by a sleep."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["F3R_BENCH_DRYRUN"] = "1"
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout  # exactly one line on stdout: the result
    return json.loads(lines[0])


def test_gpus_2_self_launches_two_ranks():
    out = _run(["--gpus", "2", "--views", "9", "--steps", "1", "--warmup", "0"])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["rccl_ranks_seen"] == 2
    assert out["config"]["views_per_gpu"] == [5, 4] and out["steps"] == 1


def test_gpus_1_runs_in_process():
    out = _run(["--views", "3", "--steps", "1", "--warmup", "0"])
    assert out["n_gpus"] == 1 and out["config"]["views_per_gpu"] == [3]


def test_views_1500_builds_the_extended_id_table():
    """BASELINE configs[4] (N=1500): the reference's 1000-row image-id table cannot index it (fast3r.py:691-697,742-743); bench.py
    --views 1500 builds the decoder with max_image_idx = 1500 (rows < 1000 identical to the reference's)."""
    from fast3r_amd.fast3r import Fast3RDecoder, sincos_1d_table
    from fast3r_amd.synthetic import vit_large_args
    _, dec, _ = vit_large_args(max_image_idx=1500)
    assert dec["max_image_idx"] == 1500
    d = Fast3RDecoder(random_image_idx_embedding=True, enc_embed_dim=64, embed_dim=128, num_heads=2, depth=1, max_image_idx=1500)
    assert d.image_idx_emb.shape == (1500, 128) and torch.equal(d.image_idx_emb[:1000], sincos_1d_table(128, 1000))
    torch.manual_seed(0)
    ids = d.draw_image_ids(1, 1500)
    assert ids.shape == (1, 1500) and ids[0, 0] == 0 and len(set(ids[0].tolist())) == 1500 and int(ids.max()) == 1499
    _, dec1000, _ = vit_large_args()
    assert "max_image_idx" not in dec1000  # the reference's constructor arguments stay untouched at N <= 1000
