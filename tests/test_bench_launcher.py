"""bench.py as the driver starts it: `python bench.py --gpus N` must bring up N ranks by itself.  Exercised here on CPU
through the F3R_BENCH_DRYRUN switch (gloo backend, no kernels): the launcher, the per-rank view split, the collectives and the
single JSON line on rank 0's stdout are the real code; only the GPU work is replaced by a sleep."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, expect_rc0=True):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["F3R_BENCH_DRYRUN"] = "1"
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    assert (p.returncode == 0) == expect_rc0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout  # exactly one JSON line on stdout: the result
    out = json.loads(lines[0])
    out["_stderr"] = p.stderr
    return out


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


def test_a_failing_exchange_form_falls_back_to_allgather_on_every_rank():
    """the first multi-GPU run must not be able to fail silently.  A fault injected into ONE rank's warm-up while the per-peer
    exchange is in use is agreed on over the gloo control group, every rank restarts its process groups and retries with the all-gather form, and
    the one JSON line says so (world 2 and 3; "auto" counts as an untried form too)."""
    out = _run(["--gpus", "2", "--views", "6", "--steps", "1", "--warmup", "1", "--exchange", "p2p"], {"F3R_BENCH_INJECT_FAULT": "1:p2p", "F3R_BENCH_PG_TIMEOUT_S": "8"})
    ex = out["exchange"]
    assert ex["requested"] == "p2p" and ex["in_use"] == "allgather" and "injected fault" in ex["fallback_reason"] and "rank 1" in ex["fallback_reason"]
    assert out["rccl_ranks_seen"] == 2 and len(out["ms_per_step_per_rank"]) == 2 and all(m > 0 for m in out["ms_per_step_per_rank"])
    out = _run(["--gpus", "3", "--views", "7", "--steps", "1", "--warmup", "1", "--exchange", "auto"], {"F3R_BENCH_INJECT_FAULT": "2:auto", "F3R_BENCH_PG_TIMEOUT_S": "8"})
    assert out["exchange"]["in_use"] == "allgather" and "rank 2" in out["exchange"]["fallback_reason"] and len(out["ms_per_step_per_rank"]) == 3
    # no fault: nothing falls back
    out = _run(["--gpus", "2", "--views", "6", "--steps", "1", "--warmup", "1", "--exchange", "p2p"])
    assert out["exchange"]["in_use"] == "p2p" and out["exchange"]["fallback_reason"] is None


def test_a_failing_allgather_is_fatal_but_still_prints_one_line():
    out = _run(["--gpus", "2", "--views", "6", "--steps", "1", "--warmup", "1"], {"F3R_BENCH_INJECT_FAULT": "0:allgather", "F3R_BENCH_PG_TIMEOUT_S": "8"}, expect_rc0=False)
    assert out["value"] is None and "injected fault" in out["error"] and out["n_gpus"] == 2


def test_the_watchdog_reports_a_stall_without_killing_the_run():
    """a warm-up that outlives F3R_BENCH_WATCHDOG_S gets its Python stacks dumped to stderr; the run itself goes on"""
    out = _run(["--views", "3", "--steps", "1", "--warmup", "0"], {"F3R_BENCH_WATCHDOG_S": "0.0"})
    assert out["n_gpus"] == 1   # (a zero-second watchdog may or may not fire before the block ends: the run must be unaffected either way)


def test_a_rank_that_cannot_start_its_process_group_still_ends_the_job_with_one_line():
    """the failure nobody can agree on over a control group, because there is none yet: rank 1 dies before init_process_group.  It prints the line
    itself (first_to_fail: one rank per launcher), the launcher ends the other rank, the exit code is non-zero"""
    out = _run(["--gpus", "2", "--views", "6", "--steps", "1", "--warmup", "1"], {"F3R_BENCH_INJECT_FAULT": "1:startup", "F3R_BENCH_PG_TIMEOUT_S": "8"}, expect_rc0=False)
    assert out["value"] is None and "rank 1 at start-up" in out["error"] and "injected fault" in out["error"] and out["n_gpus"] == 2


def test_a_stale_failure_marker_does_not_silence_the_next_job(monkeypatch, tmp_path):
    """the O_EXCL marker that elects the rank which prints the failure line is never removed; a later job with the same parent pid, port
    and run id found it and printed nothing.  A marker older than FAIL_MARKER_STALE_S is taken over; a fresh one (a sibling rank of THIS job got
    there first) still means "stay silent"."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv("MASTER_PORT", "1")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", f"stale{os.getpid()}")
    path = f"/tmp/f3r_bench_fail_{os.getppid()}_1_stale{os.getpid()}"
    try:
        assert bench.first_to_fail() is True          # first rank of the job: creates the marker
        assert bench.first_to_fail() is False         # its sibling, seconds later: silent
        old = time.time() - bench.FAIL_MARKER_STALE_S - 5
        os.utime(path, (old, old))
        assert bench.first_to_fail() is True          # a new job long after: the stale marker is taken over
        assert bench.first_to_fail() is False
    finally:
        if os.path.exists(path):
            os.unlink(path)
