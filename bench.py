#!/usr/bin/env python
"""bench.py -- views/sec of the Fast3R single-forward-pass inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--views V] [--dtype fp16|bf16] [--precision fast|high] [--no-alt] [--fusion-only]

Workload (default): BASELINE.json's headline configuration -- Fast3R ViT-Large encoder + ViT-Large fusion decoder + both
DPT heads, V = 320 synthetic views of 512x512, ONE forward pass = one step -- the configuration the metric
"views/sec (512^2, ViT-L) per node at N=320" is quoted on.  It fits one GPU, so --gpus 1 runs the same 320 views on a
single MI355X and --gpus N shards them by view over N ranks (K / V^T all-gathered per fusion layer over RCCL): the
problem size is fixed, hence "scaling": "strong".  Inputs and (random-init) weights are synthetic and already resident in
HBM when the timed region starts.  `value` = V / max-over-ranks(step time).

Launch: `python bench.py --gpus N` with N > 1 and no torchrun environment re-executes ITSELF through
`python -m torch.distributed.run --standalone --nproc-per-node N` (one rank per GPU over RCCL); started by torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.  F3R_BENCH_DRYRUN=1 swaps the GPU work for a no-op on the gloo backend so the
launcher, the view split and the rank-0 JSON plumbing can be exercised on a CPU box (tests/test_bench_launcher.py).

Operand format: the default is fp16 MFMA operands with precision "high" (split hi + lo planes for the GEMM weights and the DPT
heads, DESIGN.md section 3 (Precision modes)) -- the format whose pointmaps are within 1e-3 rel-L2 of the fp32 reference on the stress fixture as well
as on the default-init protocol.  `alt_format` in the same line is the same workload measured right after in bf16 / "fast" (one
16-bit number per operand: the round-1 headline format, faster, 2e-2 on the stress fixture); --no-alt skips it.

Extra objects in the JSON line:
  roofline     fusion-attention kernel (the dominant kernel: 94.7 % of all FLOPs at N=320): algorithmic FLOPs per launch
               4*Tq*Tk*64*heads divided by the average launch duration measured live with events on the launch stream,
               against the dense 16-bit MFMA peak of 2.5 PFLOP/s (MI355X_MICROARCH.md).  `e2e` = all algorithmic FLOPs of the
               forward pass / step time / peak.  `live` IS measured in this run: every wave of the timed fusion-attention launches
               brackets itself with s_memtime / s_memrealtime and counts its tiles (effective shader clock, matrix-pipe utilisation in
               cycles, cycles per launch), a sampler thread reads socket power and sclk at 2 Hz.  `traffic` / `reference_pmc` are NOT
               measured in this run: they are read from the committed rocprofv3 PMC summaries under profiles/ and carry their `source`.
  n100, fusion_only_n20   BASELINE configs[2] / configs[1] measured right after the headline (1 warm-up + 2 steps each).
  parity       rel-L2 of this dtype / precision on the stress fixture tests/golden/tiny_hot_3x64.pt (reference outputs), run
               through the same model class right here.
  cpu_baseline the CPU oracle (oracle/fast3r_oracle.py, a port of the reference's torch-CPU fp32 path, SDPA attention) on this
               box's host cores: ascending thread-count sweep on one view (stops when more threads get slower), then 1 warm-up + 2 timed forwards
               of 3 views at the best count.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16, MI355X_MICROARCH.md "Chip-level parameters"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views", type=int, default=320, help="total views N of the forward pass (BASELINE headline: 320)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"], help="MFMA operand type (fp32 accumulate)")
    ap.add_argument("--precision", default="high", choices=["fast", "high", "robust"],
                    help="high: split-precision GEMM operands (weights hi+lo in the transformer, both operands in the heads), see DESIGN.md section 3 (Precision modes); "
                         "the default pair (fp16, high) is the operand format that meets the 1e-3 parity bar on the stress fixture")
    ap.add_argument("--head-corrections", default="fp8", choices=["fp16", "fp8"],
                    help="precision high: where the correction products of the DPT heads' 3x3 convolutions run -- two more fp16 products, or the "
                         "block-scaled fp8 MFMA (Fast3R.head_corrections; f3r.h F3R_SPLIT_X3F8)")
    ap.add_argument("--no-fused-tail", action="store_true", help="head[2] -> f3r_dpt_final as two launches instead of the fused epilogue (f3r_gemm_args.fin_w)")
    ap.add_argument("--low-plane", default="fp8", choices=["fp16", "fp8"],
                    help="precision high: where the MLPs' correction products A W_lo run -- a second fp16 plane, or the block-scaled fp8 MFMA "
                         "(Fast3R.low_plane; f3r.h F3R_SPLIT_W2F8)")
    ap.add_argument("--no-alt", action="store_true", help="skip the second measurement in the other operand format (bf16 / fast)")
    ap.add_argument("--fusion-only", action="store_true",
                    help="BASELINE configs[1]: time only the fusion decoder on frozen random encoder features")
    ap.add_argument("--weights", default="default", choices=["default", "hot", "heavy"],
                    help="synthetic weight distribution (fast3r_amd/synthetic.py): default = the reference's own random init (near-uniform softmax); "
                         "hot = N(0, 1/fan_in): sharp attention, the lazy softmax reference of the attention kernel really moves")
    ap.add_argument("--no-hot", action="store_true", help="skip the short second measurement on the hot weights (default-weights runs only)")
    ap.add_argument("--parity-exact", action="store_true",
                    help="also run the same views through precision='exact' (the on-device fp32-equivalent path) and report the rel-L2 of the timed format "
                         "against it (any N on one GPU: from 8192 keys on the exact attention runs on the matrix pipe, 36 s at N = 320)")
    ap.add_argument("--no-inference", action="store_true", help="skip timing the same forward through fast3r_amd.inference() (host in, host out)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the two bounded extra objects of the default line: n100 (BASELINE configs[2]: N = 100 end to end) and fusion_only_n20 "
                         "(configs[1]: N = 20, fusion transformer only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-families", action="store_true", help="skip the extra forward that times every launch per kernel family (roofline.others)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-views", type=int, default=3)
    ap.add_argument("--emulate-rank", type=int, default=None,
                    help="EMULATION, not a multi-GPU measurement: this one GPU runs exactly the work of rank R of --of W view-sharded ranks (its "
                         "views, local + remote attention launches over pre-filled K/V segments, parked softmax state) with no collective")
    ap.add_argument("--of", type=int, default=8, help="world size of the emulated job (--emulate-rank)")
    ap.add_argument("--exchange", default="allgather", choices=["allgather", "auto", "p2p"],
                    help="K/V exchange of the view-sharded path: one all-gather per tensor and layer + ONE remote attention launch (default: the form "
                         "whose every RCCL call has run on an MI355X, in a world of one rank), or pairwise rounds (dealt onto --p2p-channels "
                         "communicators) + one remote launch per arrived shard; auto: three fusion layers with each form in the first warm-up "
                         "forward, then the one that exposed less (fast3r_amd/dist.py; the per-peer form has only ever run over gloo)")
    ap.add_argument("--p2p-channels", type=int, default=3)
    ap.add_argument("--reserve-cus", type=int, default=32,
                    help="view-sharded runs (and --emulate-rank): CUs the persistent local-shard attention launch leaves to the kernels of the K / V^T "
                         "exchange (f3r_attn_args.reserve_cus; profiles/r06_exchange_under_persistent_attention.json)")
    return ap.parse_args(argv)


class Watchdog:
    """`with Watchdog(what, seconds):` -- if the block has not finished after `seconds`, every thread's Python stack and the tail of this
    rank's RCCL log (NCCL_DEBUG_FILE, set by main() for multi-rank runs) go to stderr, once.  It does not kill anything: the process-group
    timeout (F3R_BENCH_PG_TIMEOUT_S) turns a hung collective into an exception, which the exchange fallback below handles."""

    def __init__(self, what, seconds=120.0, rank=0):
        import threading
        self.what, self.seconds, self.rank = what, seconds, rank
        self._done = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self.fired = False

    def _run(self):
        if self._done.wait(self.seconds):
            return
        self.fired = True
        import faulthandler
        print(f"[bench watchdog] rank {self.rank}: '{self.what}' still running after {self.seconds:.0f} s; Python stacks:", file=sys.stderr, flush=True)
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        log = os.environ.get("NCCL_DEBUG_FILE", "").replace("%p", str(os.getpid())).replace("%h", os.uname().nodename)
        if log and os.path.exists(log):
            try:
                tail = open(log, errors="replace").read()[-4000:]
                print(f"[bench watchdog] rank {self.rank}: tail of {log}:\n{tail}", file=sys.stderr, flush=True)
            except OSError:
                pass

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._done.set()


class PowerSampler:
    """Socket power and shader clock of one GPU, sampled at ~2 Hz by a thread while the timed steps run (roofline.live): amdsmi when the
    module initialises, else `rocm-smi --json`.  Never raises: a box without either reports {"source": None}."""

    def __init__(self, index=0, period=0.5):
        import threading
        self.index, self.period = index, period
        self.samples, self.source, self.error = [], None, None
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._smi = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self._handle = amdsmi.amdsmi_get_processor_handles()[index]
            self._smi, self.source = amdsmi, "amdsmi"
        except Exception as exc:  # noqa: BLE001
            self.error = f"amdsmi: {type(exc).__name__}"

    @staticmethod
    def _num(x):
        try:
            v = float(x)
            return v if v == v and v > 0 and v < 1e7 else None   # amdsmi marks absent fields with N/A or 0xFFFF...
        except Exception:  # noqa: BLE001
            return None

    def _read(self):
        if self._smi is not None:
            p = clk = None
            try:
                m = self._smi.amdsmi_get_gpu_metrics_info(self._handle)
                p = self._num(m.get("current_socket_power")) or self._num(m.get("average_socket_power"))
                cl = m.get("current_gfxclks") or []
                cl = [self._num(c) for c in (cl if isinstance(cl, (list, tuple)) else [cl])]
                cl = [c for c in cl if c]
                clk = sum(cl) / len(cl) if cl else self._num(m.get("current_gfxclk"))
            except Exception:  # noqa: BLE001
                pass
            if p is None:
                try:
                    pi = self._smi.amdsmi_get_power_info(self._handle)
                    p = self._num(pi.get("current_socket_power")) or self._num(pi.get("average_socket_power")) or self._num(pi.get("socket_power"))
                except Exception:  # noqa: BLE001
                    pass
            if clk is None:
                try:
                    clk = self._num(self._smi.amdsmi_get_clock_info(self._handle, self._smi.AmdSmiClkType.GFX).get("clk"))
                except Exception:  # noqa: BLE001
                    pass
            return p, clk
        try:
            txt = subprocess.run(["rocm-smi", "-d", str(self.index), "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            card = next(iter(json.loads(txt).values()))
            p = clk = None
            for k, v in card.items():
                kl = k.lower()
                if "power" in kl and "socket" in kl and p is None:
                    p = self._num(v)
                if "sclk" in kl and "clock" in kl and clk is None:
                    clk = self._num(str(v).strip("()").lower().replace("mhz", ""))
            self.source = "rocm-smi"
            return p, clk
        except Exception as exc:  # noqa: BLE001
            self.error = (self.error or "") + f"; rocm-smi: {type(exc).__name__}"
            return None, None

    def _run(self):
        while not self._stop.is_set():
            p, clk = self._read()
            if p is not None or clk is not None:
                self.samples.append((p, clk))
            elif self._smi is None and self.source is None:
                return   # neither source works on this box
            self._stop.wait(self.period)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=10)

    def summary(self):
        ps = [p for p, _ in self.samples if p is not None]
        cs = [c for _, c in self.samples if c is not None]
        return {"source": self.source if self.samples else None, "samples": len(self.samples), "period_s": self.period,
                "power_w_mean": sum(ps) / len(ps) if ps else None, "power_w_max": max(ps) if ps else None,
                "sclk_mhz_mean": sum(cs) / len(cs) if cs else None, "error": None if self.samples else self.error}


def fail_line(fd, args, world, ranks_seen, exchange_state, exc):
    out = {"metric": "views/sec (512^2, ViT-L) single forward pass at N=%d" % args.views, "value": None, "unit": "views/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "error": f"{type(exc).__name__}: {exc}"[:2000], "rccl_ranks_seen": ranks_seen,
           "exchange": dict(exchange_state)}
    os.write(fd, (json.dumps(out) + "\n").encode())


def first_to_fail():
    """True for exactly one rank of a job whose ranks fail before they can talk to each other (the ranks of one launcher share its pid as
    parent): that rank prints the line."""
    run_id = "".join(c for c in os.environ.get("TORCHELASTIC_RUN_ID", "") if c.isalnum())[:40]
    path = f"/tmp/f3r_bench_fail_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{run_id}"
    for attempt in range(2):
        try:
            os.close(os.open(path, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o600))
            return True
        except FileExistsError:
            # The marker of an EARLIER job with the same parent pid / port / run id (a shell loop re-running the bench, a reused pid) must not
            # silence this one: the ranks of one job fail within seconds of each other (or together at the process-group timeout), so a marker
            # older than FAIL_MARKER_STALE_S belongs to a job that is gone -- take it over
            try:
                age = time.time() - os.stat(path).st_mtime
            except OSError:
                continue   # it vanished between the two calls: try to create it again
            if attempt == 0 and age > FAIL_MARKER_STALE_S:
                try:
                    os.unlink(path)
                except OSError:
                    pass
                continue
            return False
        except OSError:
            break
    return int(os.environ.get("RANK", "0")) == 0


FAIL_MARKER_STALE_S = 300.0


def flops_forward(V, P=1024, D=1024, L_enc=24, L_dec=24, heads=2):
    """Algorithmic FLOPs of one forward pass at 512x512: GEMMs 2mnk, attention 4 T^2 D per layer, DPT heads."""
    T = V * P
    blk = 2 * T * D * (3 * D + D + 4 * D + 4 * D)            # qkv, proj, fc1, fc2
    enc = L_enc * (blk + V * 4.0 * P * P * D) + 2.0 * T * 768 * D
    dec = L_dec * (blk + 4.0 * T * T * D) + 2.0 * T * D * D
    head = heads * V * 2.45e11                                # per view and head: act_postprocess + 4 refinenets + output convs
    return enc + dec + head


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start N ranks of this script on this node and relay rank 0's JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["F3R_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0 and args.gpus > 1:
        sys.exit(self_launch(args))
    world = max(world, 1)
    # stdout carries exactly ONE line, the JSON result.  Native libraries print there too (RCCL writes its version banner to stdout when
    # NCCL_DEBUG=VERSION is exported, as it is on the GPU boxes, and C stdio flushes it at exit, i.e. AFTER the JSON line): park the real
    # stdout, point fd 1 at stderr for everything else, and write the result to the parked descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    dry = os.environ.get("F3R_BENCH_DRYRUN") == "1"

    import torch
    import torch.distributed as dist
    from fast3r_amd.dist import split_range

    # F3R_BENCH_FORCE_DIST=1: run the distributed code path (RCCL init, view sharding, barriers, MAX all-reduce) in a world of one
    # rank -- the only way to execute it on a one-GPU box (tools/gpu_ci.sh)
    force_dist = world == 1 and os.environ.get("F3R_BENCH_FORCE_DIST") == "1"
    distributed = world > 1 or force_dist
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    import datetime
    pg_timeout = datetime.timedelta(seconds=float(os.environ.get("F3R_BENCH_PG_TIMEOUT_S", "300")))
    wd_seconds = float(os.environ.get("F3R_BENCH_WATCHDOG_S", "120"))
    exchange_state = {"requested": args.exchange, "in_use": args.exchange, "fallback_reason": None}
    ranks_seen = 1
    ctl = {"group": None}   # gloo control plane beside RCCL: failure flags and per-rank timings travel here, so a broken communicator cannot hide them

    def init_groups(attempt=0):
        # a restart (attempt > 0) must not meet the first start's rendezvous keys in the launcher's store (same group names -> stale socket
        # addresses): it brings its own TCPStore, hosted by rank 0 on the next port
        kw = {}
        if attempt > 0:
            store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + attempt, world, rank == 0,
                                  timeout=pg_timeout + datetime.timedelta(seconds=60))
            kw = dict(store=store, rank=rank, world_size=world)
        if dry:
            dist.init_process_group("gloo", timeout=pg_timeout, **kw)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout, **kw)
        # (three times the data plane's timeout: a rank that failed at once waits here for the ranks that first have to run into theirs)
        # Single node, rendezvous on 127.0.0.1: let gloo use the loopback interface instead of resolving the host name (which a container
        # may not be able to do).  A control group that cannot be made is not fatal: flags and timings then travel over the data-plane group.
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        try:
            ctl["group"] = dist.new_group(backend="gloo", timeout=3 * pg_timeout + datetime.timedelta(seconds=30))
        except Exception as exc:  # noqa: BLE001
            print(f"[bench] no gloo control group ({type(exc).__name__}: {exc}): using the default process group for control traffic", file=sys.stderr, flush=True)
            ctl["group"] = None

    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if world > 1 and not dry:   # the first multi-GPU run must explain itself if it stalls: RCCL's own log per rank, read back by the watchdog
            # one RCCL workgroup per channel must find a free CU while the persistent local-shard attention launch runs: the sharded model leaves
            # ViewSharding.reserve_cus (32) CUs free, so RCCL is capped at as many channels (profiles/r06_exchange_under_persistent_attention.json)
            os.environ.setdefault("NCCL_MAX_NCHANNELS", str(max(1, args.reserve_cus)) if args.reserve_cus > 0 else "32")
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_FILE", f"/tmp/f3r_bench_rccl_rank{rank}_%p.log")
        try:
            with Watchdog("process-group start-up + first collective", wd_seconds, rank):
                if os.environ.get("F3R_BENCH_INJECT_FAULT") == f"{rank}:startup":
                    raise RuntimeError("injected fault (F3R_BENCH_INJECT_FAULT=%d:startup)" % rank)
                init_groups()
                t = torch.ones(1, dtype=torch.float64, device=dev)   # every rank adds 1: the sum is the number of ranks that really took part in a collective
                dist.all_reduce(t)
                ranks_seen = int(t.item())
        except Exception as exc:  # noqa: BLE001  -- a job that cannot even start still ends with ONE line (printed by whichever rank failed first:
            import traceback        # rank 0 may be the one that is stuck) and a non-zero exit code
            traceback.print_exc(file=sys.stderr)
            if first_to_fail():
                fail_line(real_stdout, args, world, ranks_seen, exchange_state, RuntimeError(f"rank {rank} at start-up: {type(exc).__name__}: {exc}"))
            os._exit(1)

    V = args.views
    emu = args.emulate_rank is not None
    if emu:
        assert world == 1 and not distributed, "--emulate-rank runs on ONE GPU"
        assert 0 <= args.emulate_rank < args.of <= 8
    lo, hi = split_range(V, args.of, args.emulate_rank) if emu else split_range(V, world, rank)
    views_per_gpu = [split_range(V, world, r)[1] - split_range(V, world, r)[0] for r in range(world)]

    def barrier():
        if distributed:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        if distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def per_rank(x):
        """[x of rank 0, x of rank 1, ...] over the gloo control group (a list of one without ranks)"""
        if not distributed:
            return [float(x)]
        cdev = torch.device("cpu") if (ctl["group"] is not None or dry) else dev   # (without the gloo group: device tensors over RCCL)
        out = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(out, torch.tensor([float(x)], dtype=torch.float64, device=cdev), group=ctl["group"])
        return [float(t.item()) for t in out]

    def shutdown():
        """leave the process groups behind without letting a destructor hang the job: after a fallback the first set of communicators was
        abandoned mid-collective (their threads may never join), so the process ends right after its output is flushed; otherwise a normal
        destroy, bounded by a timer"""
        sys.stdout.flush()
        sys.stderr.flush()
        if not distributed:
            return
        if exchange_state["fallback_reason"]:
            os._exit(0)
        import threading
        t = threading.Timer(30.0, lambda: os._exit(0))
        t.daemon = True
        t.start()
        try:
            dist.destroy_process_group()
        except Exception as exc:  # noqa: BLE001
            print(f"[bench] destroy_process_group: {exc}", file=sys.stderr)
        t.cancel()

    def with_exchange_fallback(setup_and_warm):
        """setup_and_warm(exchange) builds the sharded path with that exchange form and runs the warm-up steps.  An exception on ANY rank
        (also: a collective that ran into the process-group timeout) is agreed on through the gloo control group; if the form was "auto" or
        "p2p" -- the forms that have never met RCCL with more than one rank -- every rank tears the process groups down, starts them
        again and retries ONCE with "allgather"; the JSON line then carries exchange.fallback_reason.  A failure of "allgather" itself is fatal
        (rank 0 still prints one line, with `error`)."""
        for attempt in range(2):
            exc_txt = None
            try:
                with Watchdog(f"warm-up with exchange={exchange_state['in_use']}", wd_seconds, rank):
                    setup_and_warm(exchange_state["in_use"])
            except Exception as exc:  # noqa: BLE001
                import traceback
                exc_txt = f"rank {rank}: {type(exc).__name__}: {exc}"
                traceback.print_exc(file=sys.stderr)
            if not distributed:
                if exc_txt:
                    raise RuntimeError(exc_txt)
                return
            flags = [None] * world
            try:
                dist.all_gather_object(flags, exc_txt, group=ctl["group"])
            except Exception as exc:  # noqa: BLE001  (the control plane itself is gone: nothing left to agree on)
                raise RuntimeError(f"control group failed after: {exc_txt}: {exc}")
            bad = [f for f in flags if f]
            if not bad:
                return
            if exchange_state["in_use"] == "allgather" or attempt == 1:
                raise RuntimeError("; ".join(bad))
            exchange_state["fallback_reason"] = f"exchange={exchange_state['in_use']} failed during warm-up ({'; '.join(bad)[:600]}): retried with allgather"
            print("[bench] " + exchange_state["fallback_reason"], file=sys.stderr, flush=True)
            exchange_state["in_use"] = "allgather"
            try:
                dist.destroy_process_group()
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] destroy_process_group: {exc}", file=sys.stderr)
            with Watchdog("process-group restart for the allgather retry", wd_seconds, rank):
                init_groups(attempt + 1)

    if dry:
        # launcher / plumbing check only: no kernels (the product path has no CPU fallback), no measurement claimed.  The exchange fallback is
        # the real code: F3R_BENCH_INJECT_FAULT="<rank>:<exchange>" makes that rank's warm-up raise while that exchange form is in use.
        inject = os.environ.get("F3R_BENCH_INJECT_FAULT", "")

        def dry_setup_and_warm(exchange):
            if inject and inject.split(":") == [str(rank), exchange]:
                raise RuntimeError(f"injected fault (F3R_BENCH_INJECT_FAULT={inject})")
            t = torch.ones(1)
            dist.all_reduce(t) if distributed else None   # the form's first collective
        try:
            with_exchange_fallback(dry_setup_and_warm)
        except Exception as exc:  # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            if rank == 0:
                fail_line(real_stdout, args, world, ranks_seen, exchange_state, exc)
            os._exit(1)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.01 * (hi - lo))
        mine = time.perf_counter() - t0
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        ms_ranks = [x / args.steps * 1e3 for x in per_rank(mine)]
        if rank == 0:
            out = {"metric": "DRY RUN (no GPU work)", "dry_run": True, "value": None, "unit": "views/s", "n_gpus": world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "ms_per_step_per_rank": ms_ranks, "rccl_ranks_seen": ranks_seen, "backend": "gloo",
                   "config": {"views": V, "views_per_gpu": views_per_gpu},
                   "exchange": {"requested": args.exchange, "p2p_channels": args.p2p_channels, "in_use": exchange_state["in_use"] if distributed else None,
                                "fallback_reason": exchange_state["fallback_reason"], "exposed_ms_per_layer": None,
                                "note": "filled by a real run: the form the start-up probe chose and, per fusion layer, how long the compute stream sat "
                                        "between the local and the first remote attention launch (max over ranks)"}}
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        shutdown()
        return

    from fast3r_amd import Fast3R, ops
    from fast3r_amd.synthetic import make_views, synth_state_dict, vit_large_args

    enc, dec, head = vit_large_args(max_image_idx=max(1000, V))
    shapes = {k: tuple(v.shape) for k, v in Fast3R(enc, dec, head).state_dict().items()}
    sds = {}

    def state_dict_for(weights):
        if weights not in sds:
            sds[weights] = synth_state_dict(shapes, seed=0, dist=weights)
        return sds[weights]
    sd = state_dict_for(args.weights)

    # every rank holds only ITS views in HBM (the list is indexed globally by the model)
    views = [None] * V
    for i in range(lo, hi):
        v = make_views(1, 512, 512, seed=1000 + i)[0]
        v["img"] = v["img"].to(dev)
        v["idx"] = i
        views[i] = v
    placeholder = {"img": views[lo]["img"]}
    views = [v if v is not None else placeholder for v in views]  # never read outside [lo, hi)

    def measure(dtype_name, precision, steps=None, warmup=None, weights=None, parity_exact=False, time_inference=False, n_views=None, fusion_only=None,
                parity=True, families=False):
        """W warm-up + K timed steps of one operand format -> the measured fields of the JSON line.  n_views / fusion_only: the bounded extra
        objects of the default line (the first n_views of the resident views; BASELINE configs[1] / [2]) -- single-GPU runs only."""
        V = args.views if n_views is None else n_views
        fo = args.fusion_only if fusion_only is None else fusion_only
        vlist = views if V == args.views else views[:V]
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        weights = args.weights if weights is None else weights
        lp = torch.float16 if dtype_name == "fp16" else torch.bfloat16
        model = Fast3R(enc, dec, head, compute_dtype=lp, precision=precision).eval()
        model.low_plane = args.low_plane
        model.head_corrections = args.head_corrections
        if args.no_fused_tail:
            model.head_tail_fused_min_rows = 1 << 62
        if os.environ.get("F3R_ROBUST_ENCODER"):      # measurement: "planes" = the encoder's attention on the three-product kernel too
            model.robust_encoder_attention = os.environ["F3R_ROBUST_ENCODER"]
        if os.environ.get("F3R_ROBUST_CORR"):         # measurement: "fp16" = the score corrections as two more fp16 products
            model.robust_corrections = os.environ["F3R_ROBUST_CORR"]
        model.load_state_dict(state_dict_for(weights), strict=True)
        model = model.to(dev)
        if emu:
            model.emulate_rank(args.emulate_rank, args.of, exchange="allgather" if args.exchange == "auto" else args.exchange, reserve_cus=args.reserve_cus)
        if fo:
            step_fn = make_fusion_only_step(model, V, lp, dev)
        else:
            def step_fn():
                torch.manual_seed(1234)
                return model(vlist)
        def setup_and_warm(exchange):
            if distributed:
                model.shard_views(exchange=exchange, p2p_channels=args.p2p_channels, reserve_cus=args.reserve_cus)
            with torch.no_grad():
                for _ in range(warmup):
                    step_fn()
                if distributed and not dry:
                    torch.cuda.synchronize()   # a device-side failure of the exchange surfaces here, inside the guarded region
        with_exchange_fallback(setup_and_warm)
        with torch.no_grad():
            ops.ATTN_TIMER = []
            ops.ATTN_COUNTERS = torch.zeros(56, dtype=torch.int32, device=dev)
            sampler = PowerSampler(local_rank)
            if distributed:  # exposed exchange per layer over the timed steps (events on the compute stream, read after the last barrier)
                model.sharding.time_exchange = True
                for kvx in model.sharding._kvx_cache.values():
                    kvx.timing = []
            barrier()
            with sampler:
                t0 = time.perf_counter()
                for _ in range(steps):
                    last_out = step_fn()
                if not dry:
                    torch.cuda.synchronize()
                mine = time.perf_counter() - t0
                barrier()
                dt = time.perf_counter() - t0
            timer, ops.ATTN_TIMER = ops.ATTN_TIMER, None
            u32, ops.ATTN_COUNTERS = [int(c) & 0xFFFFFFFF for c in ops.ATTN_COUNTERS.tolist()], None
            # f3r_attn_args.dbg_counters (ABI 330): u32 entries, u32 waves, u64 tiles, u64 shader-clock cycles, u64 constant-clock ticks
            counters = [u32[0], u32[1], u32[2] | (u32[3] << 32), u32[4] | (u32[5] << 32), u32[6] | (u32[7] << 32)]
            counters.append([[u32[8 + 6 * x + 2 * i] | (u32[9 + 6 * x + 2 * i] << 32) for i in range(3)] for x in range(8)])  # per XCD: cycles, ticks, waves
        dt = max_over_ranks(dt)
        exch = None
        if distributed:
            ms = [x for kvx in model.sharding._kvx_cache.values() for x in kvx.exposed_ms()]
            per_layer = max_over_ranks(sum(ms) / max(1, len(ms)))
            exch = {"requested": args.exchange, "in_use": model.sharding.exchange_in_use, "fallback_reason": exchange_state["fallback_reason"],
                    "p2p_channels": args.p2p_channels,
                    "exposed_ms_per_layer": per_layer, "layers_timed_on_rank0": len(ms), "probe": {k: v for k, v in model.sharding._probe.items() if k != "layer"},
                    "what": "mean gap on the compute stream between the end of the local-shard attention launch and the start of the first remote one "
                            "(max over ranks): the part of the K / V^T exchange the local launch did not hide"}
        # dominant kernel = the fusion attention launches (the ones whose key count is the whole scene)
        if emu:
            # two launches per fusion layer: queries = the rank's tokens, keys = its own shard (local) / the other ranks' shards (remote)
            t_loc = (hi - lo) * 1024
            pair = [(a.elapsed_time(b), fl) for a, b, fl, tq, tk, _ in timer if tq == t_loc and tk >= t_loc]
            kname = sorted(set(nm for _, _, _, tq, tk, nm in timer if tq == t_loc and tk >= t_loc))
            n_layers = int(dec["depth"]) * steps
            big = sum(fl for _, fl in pair) / n_layers
            avg_ms = sum(ms for ms, _ in pair) / n_layers
            fus = pair
        else:
            fus = [(a.elapsed_time(b), fl, nm) for a, b, fl, _, _, nm in timer]
            big = max(fl for _, fl, _ in fus)
            kname = sorted(set(nm for _, fl, nm in fus if fl == big))
            fus = [(ms, fl) for ms, fl, _ in fus if fl == big]
            avg_ms = sum(ms for ms, _ in fus) / len(fus)
        achieved = big / (avg_ms * 1e-3) / 1e12
        prec = {"fast": "", "robust": "; every GEMM / conv with both operands as hi+lo planes, fusion attention with Q and K as hi+lo planes (three products per "
                                      "score block, f3r_attn_asm_qk3_f16), the encoder's attention in fp32"}.get(
            precision, "; split-precision GEMM operands (weights hi+lo in the transformer, both operands hi+lo in the heads)")
        e2e = None if (fo or emu) else flops_forward(V) / (dt / steps) / 1e12 / world
        if emu:
            kvx = model.sharding.last_exchange
            res_emu = {"comm_bytes_per_layer_into_this_gpu": kvx.comm_bytes_per_layer, "fusion_layers": int(dec["depth"]),
                       "xgmi_link_budget": "7 links x ~153 GB/s per GPU (MI355X_MICROARCH / SURVEY section 5)",
                       "allgather_ms_per_layer_all_links": kvx.comm_bytes_per_layer / (7 * 153e9) * 1e3,
                       "exchange": args.exchange, "reserve_cus": args.reserve_cus, "attention_launches_per_layer": len(fus) // max(1, n_layers),
                       "attention_ms_per_layer": avg_ms}
        # how often the lazy softmax reference of the hand-scheduled kernel moved (f3r_attn_args.dbg_counters, summed over every launch of
        # the timed steps that took that kernel): per wave and 64-key tile, the forced first re-base of each wave not counted
        entries, waves = counters[0], counters[1]
        tiles_per_wave = None if waves == 0 else counters[2] / waves   # counted by the kernel (64-bit sum, ABI 330)
        rebase = None if waves == 0 else {"rebases_per_wave": (entries - waves) / waves,
                                          "rebases_per_tile": None if tiles_per_wave is None else (entries - waves) / waves / tiles_per_wave,
                                          "waves": waves, "tiles_per_wave": tiles_per_wave,
                                          "note": "summed over the timed steps; the forced first re-base of each wave is excluded"}
        live = live_roofline(counters, avg_ms, achieved, int(dec["embed_dim"]) // int(dec["num_heads"]), sampler.summary(),
                             qk_products=3 if any("qk3" in n for n in kname) else 1, fp8_corrections=any("qk3f8" in n for n in kname),
                             q256=any(n.startswith("f3r_attn_asm_q256") for n in kname))   # (a launch that only runs its LAST round on 256-query items counts as the 512-query form)
        res = {"value": V / (dt / steps), "ms_per_step": dt / steps * 1e3, "ms_per_step_per_rank": [x / steps * 1e3 for x in per_rank(mine)], "steps": steps, "warmup": warmup, "dtype": dtype_name, "precision": precision,
               "weights": weights, "attn_rebase": rebase,
               "operands": f"{dtype_name} MFMA operands, fp32 accumulate / residual / LayerNorm / softmax" + prec,
               "roofline": {"bound": "mfma", "kernel": " / ".join(kname) + " -- the fusion self-attention launches of f3r_attn_fwd",
                            "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                            "flops_per_launch": big, "avg_launch_ms": avg_ms, "launches_timed": len(fus), "live": live,
                            "e2e": None if e2e is None else {"flops_per_forward": flops_forward(V), "achieved_per_gpu": e2e, "frac": e2e / MFMA_PEAK_TFLOPS}}}
        if emu:
            res["emulation"] = res_emu
        if exch is not None:
            res["exchange"] = exch
        if time_inference and not (emu or distributed or fo):
            # the function users call (fast3r/dust3r/inference_multiview.py:70-99): host images in, everything back on the host.  Same model,
            # same views (as host tensors, like load_images returns them); 1 warm-up (pinned buffers of the output leg) + 2 timed calls.
            # A failure here (e.g. no pinned memory left on the host) must not cost the headline measurement: it is reported instead.
            if parity_exact:
                last_out = [{k: v.float().cpu() for k, v in o.items()} for o in last_out]  # kept for the parity leg below, off the device
            else:
                last_out = None
            try:
                from fast3r_amd import inference as f3r_inference
                host_views = [dict(v, img=v["img"].cpu()) for v in vlist]
                times = []
                for it in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    torch.manual_seed(1234)
                    r = f3r_inference(host_views, model, dev, dtype="16-mixed" if dtype_name == "fp16" else "bf16-mixed", verbose=False)
                    torch.cuda.synchronize()
                    times.append(time.perf_counter() - t0)
                    assert r["preds"][0]["pts3d_in_other_view"].device.type == "cpu"
                    del r
                inf_ms = min(times[1:]) * 1e3
                step_ms = dt / steps * 1e3
                res["inference"] = {"what": "fast3r_amd.inference(host views, model, device, dtype): upload + forward + every output on the host (pinned, copied by a "
                                            "side stream per head chunk) -- the call the reference's users make (inference_multiview.py:70-99)",
                                    "inference_ms": inf_ms, "ms_per_step": step_ms, "extra_ms": inf_ms - step_ms, "extra_frac_of_step": (inf_ms - step_ms) / step_ms,
                                    "calls_ms": [t * 1e3 for t in times]}
                del host_views
            except Exception as exc:  # noqa: BLE001
                res["inference"] = {"error": f"{type(exc).__name__}: {exc}"}
        if families and not (emu or distributed):
            # roofline.others: ONE more forward with an event pair around every launch, summed per kernel family.  Outside the
            # timed steps (the events cost a few microseconds per launch); the families are exactly the ones the rocprofv3 kernel stats group.
            try:
                res["roofline"]["others"] = family_rooflines(ops, step_fn, steps=1, fusion_flops=big)
            except Exception as exc:  # noqa: BLE001  (an extra: never at the price of the headline line)
                res["roofline"]["others"] = {"error": f"{type(exc).__name__}: {exc}"}
        if rank == 0 and not args.no_parity and not emu and parity:
            res["parity"] = parity_on_stress_fixture(lp, precision, dev)
        if parity_exact and not (emu or distributed or fo):
            keep = [{k: v.float().cpu() for k, v in o.items()} for o in last_out]  # (.cpu() of a host tensor is the tensor itself)
            del model, last_out
            torch.cuda.empty_cache()
            mx = Fast3R(enc, dec, head, compute_dtype=torch.float16, precision="exact").eval()
            mx.load_state_dict(state_dict_for(weights), strict=True)
            mx = mx.to(dev)
            with torch.no_grad():
                torch.manual_seed(1234)
                ref = mx(vlist)
            worst = {}
            for o, g in zip(keep, ref):
                for k in g:
                    a, b = o[k].double().flatten(), g[k].double().flatten().cpu()
                    worst[k] = max(worst.get(k, 0.0), float((a - b).norm() / b.norm()))
            res["parity_vs_exact"] = {"checker": "precision='exact' on the same views and weights (fp32-equivalent path, 3e-7 of the reference on its golden outputs)",
                                      "rel_l2": max(worst.values()), "per_output": worst, "bar": 1e-3}
            model = mx
            del ref
        last_out = None
        del model
        torch.cuda.empty_cache()
        return res

    if args.fusion_only:
        workload = f"fusion transformer only (frozen random encoder features), N={V} views 512x512 (BASELINE configs[1] shape)"
    else:
        workload = f"Fast3R ViT-L 512x512 end-to-end single forward pass (encoder + fusion decoder + 2 DPT heads), N={V} views"

    try:
        main_res = measure(args.dtype, args.precision, parity_exact=args.parity_exact, time_inference=not args.no_inference, families=not args.no_families)
    except Exception as exc:  # noqa: BLE001  -- a failed run still ends with ONE line on rank 0's stdout, and a non-zero exit code
        import traceback
        traceback.print_exc(file=sys.stderr)
        if rank == 0:
            fail_line(real_stdout, args, world, ranks_seen, exchange_state, exc)
        os._exit(1)   # (not sys.exit: a rank stuck in a collective's destructor must not keep the job alive)
    if emu:
        out = {"metric": "EMULATED per-rank step: ONE GPU runs rank %d of %d of the view-sharded forward at N=%d (no collectives, remote K/V segments "
                         "pre-filled) -- NOT a multi-GPU measurement" % (args.emulate_rank, args.of, V),
               "emulation": True, "value": None, "unit": "views/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": main_res["ms_per_step"], "per_rank_step_ms": main_res["ms_per_step"],
               "projected_views_per_s_if_comm_is_hidden": V / (main_res["ms_per_step"] * 1e-3),
               "dtype": main_res["dtype"], "precision": main_res["precision"], "data": "synthetic",
               "config": {"workload": workload, "views": V, "views_of_this_rank": hi - lo, "rank": args.emulate_rank, "world": args.of, "exchange": args.exchange},
               "roofline": main_res["roofline"], "exchange": main_res["emulation"]}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        return
    # The same workload in the other operand format, measured in the same process: the default (fp16 operands, precision "high") is the
    # format that meets the 1e-3 parity bar on the stress fixture; bf16 / "fast" is the round-1 headline format (parity 2e-2 there).
    alt_fmt = ("bf16", "fast") if (args.dtype, args.precision) != ("bf16", "fast") else ("fp16", "high")
    # (bounded: at most 3 timed steps after at most 1 warm-up, whatever --steps / --warmup ask of the main measurement)
    # (an extra like the ones below: never at the price of the headline line.  With ranks, a failed extra leaves the communicators in an unknown
    # state, so the remaining extras are skipped)
    alt_res, extras_ok = None, True
    if not args.no_alt:
        try:
            alt_res = measure(*alt_fmt, steps=min(args.steps, 3), warmup=min(args.warmup, 1))
        except Exception as exc:  # noqa: BLE001
            print(f"alt-format measurement failed: {type(exc).__name__}: {exc}", file=sys.stderr)
            extras_ok = not distributed

    # The dominant kernel on inputs that exercise it: with the default-init weights the softmax is near-uniform and the
    # kernel's re-base branch is never taken in the timed region; the hot weights attend sharply.  Bounded: 1 warm-up + at most 2 steps.
    hot_res = None
    if args.weights == "default" and not args.no_hot and not emu and not args.fusion_only and extras_ok:
        try:
            hot_res = measure(args.dtype, args.precision, steps=min(args.steps, 2), warmup=min(args.warmup, 1), weights="hot")
        except Exception as exc:  # noqa: BLE001  (an extra: never at the price of the headline line)
            print(f"hot-weights measurement failed: {type(exc).__name__}: {exc}", file=sys.stderr)

    # The robust tier in the driver's record (round 6): the same N = 100 workload with precision "robust" (1 warm-up + 1 step, ~4 s) and what the tier is
    # FOR -- the heavy-tailed tiny model (4 views of 64 x 64: a whole key tile, so the three-product kernel runs), high and robust against the exact mode
    # on the device.  Single-GPU default runs only; never at the price of the headline line.
    robust_extra = None
    if (world == 1 and not distributed and not emu and not args.fusion_only and not args.no_extra_configs and V >= 100 and args.dtype == "fp16"
            and args.precision != "robust"):
        try:
            r = measure("fp16", "robust", steps=1, warmup=1, parity=False, n_views=100)
            robust_extra = {"what": "precision='robust' (every GEMM / conv with both operands as hi + lo planes; fusion attention scores from hi + lo planes of Q "
                                    "and K, the corrections on the block-scaled fp8 MFMA) on BASELINE configs[2]: N = 100 views 512x512 end to end",
                            "value": r["value"], "unit": "views/s", "ms_per_step": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"],
                            "roofline": {k: r["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "launches_timed")},
                            "note": "roofline.achieved counts ONE product per score (algorithmic), the kernel executes three (two of them at the fp8 rate)"}
            robust_extra["heavy_tailed_tiny_model_vs_exact"] = tier_distances_on_heavy_tailed_tiny_model(dev)
        except Exception as exc:  # noqa: BLE001
            print(f"robust-tier measurement failed: {type(exc).__name__}: {exc}", file=sys.stderr)
            robust_extra = {"error": f"{type(exc).__name__}: {exc}"}

    # BASELINE configs[2] and configs[1] in the same line: the sizes where the non-attention work shows.  Bounded
    # (1 warm-up + 2 steps each, ~10 s together), single-GPU default runs only, never at the price of the headline line.
    extra = {}
    if world == 1 and not distributed and not emu and not args.fusion_only and not args.no_extra_configs and V >= 100:
        for key, kw in (("n100", dict(n_views=100)), ("fusion_only_n20", dict(n_views=20, fusion_only=True))):
            try:
                r = measure(args.dtype, args.precision, steps=2, warmup=1, parity=False, families=not args.no_families, **kw)
                extra[key] = {"what": "BASELINE configs[2]: N = 100 views 512x512, full encoder + fusion + heads" if key == "n100" else
                                      "BASELINE configs[1]: N = 20 views 512x512, fusion transformer only on frozen random encoder features",
                              "value": r["value"], "unit": "views/s", "ms_per_step": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"],
                              "dtype": r["dtype"], "precision": r["precision"],
                              "roofline": {k: r["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "launches_timed", "e2e")}}
                if "others" in r["roofline"]:
                    extra[key]["roofline"]["others"] = r["roofline"]["others"]
                extra[key]["roofline"]["live"] = {k: v for k, v in r["roofline"]["live"].items() if k not in ("source", "power", "per_xcd")}
                extra[key]["attention_share_of_step"] = r["roofline"]["avg_launch_ms"] * int(dec["depth"]) / r["ms_per_step"]
            except Exception as exc:  # noqa: BLE001
                print(f"{key} measurement failed: {type(exc).__name__}: {exc}", file=sys.stderr)
                extra[key] = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        out = {
            "metric": "views/sec (512^2, ViT-L) single forward pass at N=%d" % V,
            "value": main_res["value"], "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"], "ms_per_step_per_rank": main_res["ms_per_step_per_rank"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None,
            "dtype": main_res["dtype"], "precision": main_res["precision"], "data": "synthetic", "rccl_ranks_seen": ranks_seen,
            "config": {"workload": workload, "views": V, "views_per_gpu": views_per_gpu,
                       "tokens": V * 1024, "image": "512x512", "parallelism": f"view-sharded x{world}, K/V all-gather per fusion layer" if world > 1 else "single GPU",
                       "operands": main_res["operands"], "low_plane": args.low_plane, "head_corrections": args.head_corrections,
                       "head_tail_fused": not args.no_fused_tail},
            # `live` = this run's own clock / utilisation / power record; `traffic` and `reference_pmc` are read from committed rocprofv3
            # PMC passes of other runs (they carry their source file) and are there to be compared with `live`, not to stand in for it
            "roofline": dict(main_res["roofline"], traffic=load_traffic(V, world), reference_pmc=load_pmc(main_res["dtype"])),
        }
        out.update(extra)
        if robust_extra is not None:
            out["robust_tier"] = robust_extra
        if "exchange" in main_res:
            out["exchange"] = main_res["exchange"]
        out["weights"] = main_res["weights"]
        out["attn_rebase"] = main_res["attn_rebase"]
        for k in ("parity", "parity_vs_exact", "inference"):
            if k in main_res:
                out[k] = main_res[k]
        if hot_res is not None:
            out["hot_weights"] = {k: hot_res[k] for k in ("weights", "value", "ms_per_step", "steps", "warmup", "dtype", "precision", "attn_rebase")}
            out["hot_weights"]["roofline"] = {k: hot_res["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "e2e")}
            out["hot_weights"]["frac_vs_default_weights"] = hot_res["roofline"]["frac"] / main_res["roofline"]["frac"]
        if alt_res is not None:
            out["alt_format"] = {k: alt_res[k] for k in ("dtype", "precision", "value", "ms_per_step", "steps", "warmup", "operands") if k in alt_res}
            out["alt_format"]["roofline"] = {k: alt_res["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "e2e")}
            if "parity" in alt_res:
                out["alt_format"]["parity"] = alt_res["parity"]
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(sd, enc, dec, head, args.cpu_views)
            except Exception as exc:  # noqa: BLE001
                print(f"CPU baseline failed: {type(exc).__name__}: {exc}", file=sys.stderr)
                out["cpu_baseline"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    shutdown()


def live_roofline(counters, avg_launch_ms, achieved_tflops, head_dim, power, qk_products=1, fp8_corrections=False, q256=False):
    """What the timed fusion-attention launches THEMSELVES recorded (f3r_attn_args.dbg_counters, ABI 330): every wave of the hand-scheduled
    kernel brackets its life with s_memtime (shader clock) and s_memrealtime (constant clock), and counts the 64-key tiles it walked.  One wave
    per SIMD, so a wave's cycles are its SIMD's cycles: matrix-pipe utilisation = 32 cycles x MFMAs issued / cycles lived; the effective shader
    clock = cycles / (ticks / clock rate).  util x clock x (256 CUs x 4 SIMDs x 1024 FLOP per cycle) is the rate the counters imply; `achieved`
    (events around the launches) sits just below it by the launch overhead and the idle tail of the last round of workgroups."""
    entries, waves, tiles, cycles, ticks, per_xcd = counters
    if waves == 0 or cycles == 0 or ticks == 0:
        return {"note": "no launch of the hand-scheduled attention kernel in the timed steps", "power": power}
    from fast3r_amd import _lib
    khz = int(_lib.lib().f3r_wall_clock_khz()) or 100000
    qpw = 4 if (head_dim == 64 and qk_products == 1 and not q256) else 2   # 32-query blocks per wave (f3r_attn_asm_q256_*: two)
    # Q K^T k-steps (x 3 for the three-product kernels of precision "robust") + P V blocks of one 64-key tile, per wave, in units of ONE 32-cycle
    # MFMA (the two block-scaled fp8 MFMAs that replace eight fp16 k-steps in f3r_attn_asm_qk3f8_f16 take 64 cycles each = 4 units per half tile)
    nk, ndb = head_dim // 16, (head_dim + 31) // 32
    qk_units = nk * qk_products if not fp8_corrections else nk + 4
    mfma_per_tile = qpw * (2 * qk_units + 4 * ndb)
    clock_ghz = cycles / (ticks / (khz * 1e3)) / 1e9
    util = 32.0 * mfma_per_tile * tiles / cycles
    implied = util * clock_ghz * 256 * 4 * 1024 / 1e3   # TFLOP/s EXECUTED on the matrix pipe
    if qk_products != 1:   # `achieved` counts one product per score (algorithmic): scale the implied rate to the same unit
        implied *= (2 * nk + 4 * ndb) / (2 * qk_units + 4 * ndb)
    # per XCD: with one item per workgroup id the ids are dealt round-robin over the 8 XCDs (equal work each) and the launch lasts as long as the
    # slowest XCD needs; with work stealing (f3r_attn_args.sched_counter) an XCD takes items as fast as its clock lets it, and the XCDs' busy
    # times (the tick sums) come out equal instead of their wave counts
    xcd = None
    if all(w > 0 and t > 0 for _, t, w in per_xcd):
        mean_us = [t / w / (khz * 1e3) * 1e6 for _, t, w in per_xcd]
        busy = [t for _, t, _ in per_xcd]
        stealing = len({w for _, _, w in per_xcd}) > 1
        xcd = {"wave_time_us_mean": mean_us, "clock_ghz": [c / (t / (khz * 1e3)) / 1e9 for c, t, _ in per_xcd], "waves": [w for _, _, w in per_xcd],
               "slowest_over_mean": max(mean_us) / (sum(mean_us) / 8), "busiest_over_mean": max(busy) / (sum(busy) / 8),
               "dealing": "work stealing" if stealing else "static",
               "note": ("work stealing: persistent workgroups take items from one counter, so a faster XCD walks more of them (waves) and the busy "
                        "times (wave time x waves) are level: busiest_over_mean, not slowest_over_mean, is what bounds the launch" if stealing else
                        "equal work per XCD (static round-robin of workgroup ids): the slowest XCD bounds the launch; achieved_over_implied below "
                        "1 / slowest_over_mean is idle time inside the XCDs (tail of the last round, dispatch gaps)")}
    return {"source": "s_memtime / s_memrealtime brackets + tile counts written by every wave of the timed launches (f3r_attn_args.dbg_counters)",
            "effective_clock_ghz": clock_ghz, "mfma_util_cycles": util, "cycles_per_launch": avg_launch_ms * 1e-3 * clock_ghz * 1e9,
            "wave_cycles_mean": cycles / waves, "waves": waves, "mfma_per_wave_mean": mfma_per_tile * tiles / waves, "wall_clock_khz": khz,
            "implied_tflops": implied, "implied_frac": implied / MFMA_PEAK_TFLOPS, "achieved_over_implied": achieved_tflops / implied,
            "per_xcd": xcd, "power": power}


HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured with a float4 copy)


def family_rooflines(ops, step_fn, steps, fusion_flops):
    """One forward with ops.OP_TIMER on: every launch of the C ABI is bracketed by an event pair on its stream and booked under its kernel family --
    transformer_linears (patch embedding, QKV, proj, fc1, fc2, decoder_embed), head_convs (every 1x1 / 3x3 / transposed convolution of the two DPT
    heads), attention (split into the fusion launches = the roofline kernel, and the encoder's), elementwise (LayerNorm, casts, bilinear upsampling,
    final conv + postprocess, patchify).  MFMA families: ALGORITHMIC FLOP (one product per output, whatever planes the precision mode executes)
    / summed launch time / 2.5 PFLOP/s; elementwise: tensor bytes in + out / time / 8 TB/s.  Launch gaps are in nobody's sum (`sum_ms` vs the step)."""
    import torch
    ops.OP_TIMER = {}
    try:
        with torch.no_grad():
            for _ in range(steps):
                step_fn()
        torch.cuda.synchronize()
        rec = ops.OP_TIMER
    finally:
        ops.OP_TIMER = None
    fams = {}
    att = rec.pop("attention", [])
    rec["attention_fusion"] = [r for r in att if r[2] >= 0.99 * fusion_flops]
    rec["attention_other"] = [r for r in att if r[2] < 0.99 * fusion_flops]
    out, total = [], 0.0
    for fam in ("attention_fusion", "transformer_linears", "head_convs", "attention_other", "elementwise", "other"):
        rs = rec.get(fam, [])
        if not rs:
            continue
        ms = sum(a.elapsed_time(b) for a, b, _, _ in rs) / steps
        fl = sum(r[2] for r in rs) / steps
        by = sum(r[3] for r in rs) / steps
        total += ms
        e = {"family": fam, "ms_per_forward": ms, "launches_per_forward": len(rs) // steps}
        if fam == "elementwise":
            e.update({"bound": "hbm", "algorithmic_gbytes": by / 1e9, "achieved_gb_s": by / (ms * 1e-3) / 1e9, "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        else:
            e.update({"bound": "mfma", "algorithmic_tflop": fl / 1e12, "algorithmic_tflops": fl / (ms * 1e-3) / 1e12, "frac": fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS})
        out.append(e)
    return {"what": "one extra forward with an event pair around every launch (fast3r_amd.ops.OP_TIMER), summed per kernel family; algorithmic work "
                    "(one product per output; the split-precision planes a mode executes are NOT counted) over the summed launch time",
            "families": out, "sum_ms": total}


def make_fusion_only_step(model, V, lp, dev):
    """BASELINE configs[1]: frozen random encoder features (V,1024 tokens,1024) -> Fast3RDecoder only."""
    import torch
    g = torch.Generator().manual_seed(0)
    feats = torch.randn((V * 1024, 1024), generator=g).to(lp).to(dev)
    ids = torch.arange(V)[None]

    def step():
        return model.decode_tokens(feats, [1024] * V, ids)[-1]
    return step


def parity_on_stress_fixture(lp, precision, dev):
    """The 1e-3 bar of BASELINE.json's north_star, measured for THIS dtype / precision on the stress fixture (tests/golden: outputs
    of the real reference on N(0, 1/fan_in) weights -- sharp attention, noise-amplifying heads)."""
    import torch
    from fast3r_amd import Fast3R
    from fast3r_amd.synthetic import make_views, synth_state_dict, tiny_args
    name = "tiny_hot_3x64"
    fix = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)
    enc, dec, head = tiny_args(**fix["tiny_kwargs"])
    sd = synth_state_dict(fix["state_shapes"], fix["weight_seed"], fix["weight_dist"])
    m = Fast3R(enc, dec, head, compute_dtype=lp, precision=precision).eval()
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    views = []
    for i, (h, w) in enumerate(fix["shapes"]):
        v = make_views(1, h, w, fix["batch"], seed=1000 + i)[0]
        v["img"] = v["img"].to(dev)
        views.append(v)
    with torch.no_grad():
        torch.manual_seed(fix["rng_seed"])
        out = m(views)
    worst = {}
    for o, g in zip(out, fix["preds"]):
        for k in g:
            a, b = o[k].double().flatten().cpu(), g[k].double().flatten()
            worst[k] = max(worst.get(k, 0.0), float((a - b).norm() / b.norm()))
    return {"fixture": f"tests/golden/{name}.pt (reference outputs, stress weights)", "rel_l2": max(worst.values()), "per_output": worst,
            "bar": 1e-3}


def tier_distances_on_heavy_tailed_tiny_model(dev):
    """rel-L2 of precision "high" and "robust" against the fp32-equivalent mode ON THE DEVICE for the tiny model with heavy-tailed weights
    (fast3r_amd/synthetic.py dist="heavy": the distribution on which one 16-bit number per operand does not hold 1e-3), 4 views of 64 x 64 = one whole
    key tile in the fusion layers.  HIP against HIP: the exact mode is pinned on the reference's goldens (3e-7) by the test suite."""
    import torch
    from fast3r_amd import Fast3R
    from fast3r_amd.synthetic import make_views, synth_state_dict, tiny_args
    enc, dec, head = tiny_args()
    shapes = {k: tuple(v.shape) for k, v in Fast3R(enc, dec, head).state_dict().items()}
    m = Fast3R(enc, dec, head).eval()
    m.load_state_dict(synth_state_dict(shapes, 0, dist="heavy"), strict=True)
    m = m.to(dev)
    views = [dict(v, img=v["img"].to(dev)) for v in make_views(4, 64, 64)]
    rep = m.calibrate_precision(views, tiers=("high", "robust"))
    return {"high": rep["worst"]["high"], "robust": rep["worst"]["robust"], "recommended": rep["recommended"], "bar": 1e-3,
            "checker": "precision='exact' on the same views and weights (Fast3R.calibrate_precision)"}


def load_traffic(V, world):
    """HBM bytes per attention launch from the committed rocprofv3 PMC pass (profiles/), if one exists for this shape."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "attn_traffic.json"))).get(f"views={V},gpus={world}")
    except Exception:
        return None
    if t is None:
        return None
    out = {"source": "profiles/attn_traffic.json (rocprofv3 --pmc, not measured in this run)", "shape": f"views={V},gpus={world}"}
    out.update(t if isinstance(t, dict) else {"bytes": t})
    return out


def load_pmc(dtype_name="fp16"):
    """Matrix-pipe utilisation in cycles + effective clock of the same kernel from the newest committed rocprofv3 PMC pass
    (profiles/r*_attn_mfma_util.json, tools/pmc_r03_attn.sh): `frac` above is this times clock / 2.4 GHz."""
    try:
        import glob
        cands = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_attn_mfma_util.json")))
        d = json.load(open(cands[-1]))
        d = d.get("formats", {}).get(dtype_name, d)  # round 2+: one entry per operand format
        return {"source": f"profiles/{os.path.basename(cands[-1])} (rocprofv3 --pmc, not measured in this run)", "shape": "T=%d" % (1024 * d["views"]),
                "operands": dtype_name if "avg_dispatch_ms" in d else "bf16",
                "mfma_util_cycles": d["mfma_util_cycles"], "mfma_util_useful_cycles": d["mfma_util_useful_cycles"],
                "effective_clock_ghz": d["effective_clock_ghz"]}
    except Exception:
        return None


def cpu_baseline(sd, enc, dec, head, n_views):
    """The oracle (port of the reference's CPU fp32 path, SDPA attention like `attn_implementation="flash_attention"`) on the host
    cores: a thread-count sweep on ONE view picks the count, then 1 warm-up + 2 timed forwards of n_views views at that count."""
    import torch
    from fast3r_amd.synthetic import make_views
    from oracle import fast3r_oracle as O
    O.ATTN_IMPL = "sdpa"
    nproc = os.cpu_count() or 1
    saved = torch.get_num_threads()
    one = make_views(1, 512, 512)
    sweep = {}
    with torch.no_grad():
        torch.set_num_threads(min(8, nproc))
        O.forward(one, sd, enc, dec, head)  # first touch of the weights / allocator
        # ascending thread counts, stopping at the first one that is slower than its predecessor: past the sweet spot torch's
        # intra-op pool oversubscribes badly on these hosts (256 hardware threads: 239 s per view against 2.2 s at 32)
        for th in sorted({t for t in (8, 16, 32, 64, 128) if t <= nproc} | {min(8, nproc)}):
            torch.set_num_threads(th)
            t0 = time.perf_counter()
            torch.manual_seed(1234)
            O.forward(one, sd, enc, dec, head)
            sweep[th] = time.perf_counter() - t0
            if len(sweep) > 1 and sweep[th] > 1.15 * min(v for k, v in sweep.items() if k != th):
                break
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        views = make_views(n_views, 512, 512)
        torch.manual_seed(1234)
        O.forward(views, sd, enc, dec, head)  # warm-up
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            torch.manual_seed(1234)
            O.forward(views, sd, enc, dec, head)
            times.append(time.perf_counter() - t0)
    torch.set_num_threads(saved)
    O.ATTN_IMPL = "naive"
    dt = min(times)
    anchor = None
    try:   # what `kind: "port"` rests on: the oracle timed beside the imported reference where both can run (the build container; /root/reference
        a = json.load(open(os.path.join(ROOT, "profiles", "r03_cpu_port_vs_reference.json")))   # does not exist on the GPU box)
        anchor = {"source": "profiles/r03_cpu_port_vs_reference.json (oracle/time_reference.py, build container, %d threads)" % a["threads"],
                  "port_over_reference_throughput": a["port_over_reference_throughput"], "port_vs_reference_rel_l2": a["port_vs_reference_rel_l2"],
                  "reference_views_per_s_there": a["reference_views_per_s"], "port_views_per_s_there": a["port_views_per_s"]}
    except Exception:  # noqa: BLE001
        pass
    return {"value": n_views / dt, "unit": "views/s", "cores": best, "kind": "port", "port_over_reference": anchor,
            "sample": f"same model (ViT-L/ViT-L/2 DPT), {n_views} views of 512x512, fp32, SDPA attention; 1 warm-up + 2 timed forwards "
                      f"({', '.join('%.1f s' % t for t in times)}; best reported) at {best} threads of {nproc}",
            "thread_sweep_s_per_view": {str(k): round(v, 2) for k, v in sweep.items()}}


if __name__ == "__main__":
    main()
