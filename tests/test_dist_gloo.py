"""world_size-2 gloo test of the view-sharded path's host logic on CPU: range split, image-id broadcast, token-count
all-gather, the per-layer K / V^T all-gather with uneven shards, and -- with the oracle's attention maths standing in
for the HIP kernel as the checker -- that attention over the gathered segments equals unsharded attention."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _segment_attention(q, segs, scale):
    """Checker: softmax over the concatenation of the segments (k [T][D], vt [D][ld], len), 2 heads of 64."""
    k = torch.cat([s[0][: s[2]] for s in segs], dim=0).float()
    v = torch.cat([s[1][:, : s[2]].t() for s in segs], dim=0).float()
    H = q.shape[1] // 64
    out = torch.empty_like(q, dtype=torch.float32)
    for h in range(H):
        sl = slice(h * 64, h * 64 + 64)
        a = ((q[:, sl].float() @ k[:, sl].t()) * scale).softmax(-1)
        out[:, sl] = a @ v[:, sl]
    return out


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fast3r_amd.dist import ViewSharding
        sh = ViewSharding()
        n_views, P, D = 5, 24, 128  # uneven: rank 0 owns 3 views, rank 1 owns 2
        lo, hi = sh.my_range(n_views)
        assert (lo, hi) == ((0, 3) if rank == 0 else (3, 5))
        ids = torch.arange(n_views)[None] * (rank + 1)  # ranks disagree until the broadcast
        ids = sh.broadcast_ids(ids, torch.device("cpu"))
        assert ids.tolist() == [list(range(n_views))]
        t_loc = (hi - lo) * P
        try:
            sh.make_kv_exchange(t_loc, D, torch.float32, torch.device("cpu"), t_all=[3 * P, 2 * P])
            raise AssertionError("n_heads must be given (no guess from the width)")
        except ValueError:
            pass
        kvx = sh.make_kv_exchange(t_loc, D, torch.float32, torch.device("cpu"), n_heads=2)
        assert kvx.t_all == [3 * P, 2 * P] and kvx.ldvt % 64 == 0 and kvx.ldvt >= 3 * P
        # full (unsharded) problem, identical on both ranks
        g = torch.Generator().manual_seed(0)
        T = n_views * P
        qf, kf, vf = (torch.randn(T, D, generator=g) for _ in range(3))
        r0 = lo * P
        for layer in range(2):  # buffers are reused across layers
            kvx.k_loc[:t_loc] = kf[r0:r0 + t_loc] + layer
            kvx.vt_loc[0, :, :t_loc] = (vf[r0:r0 + t_loc] + layer).t()
            segs = kvx.exchange()
            assert [s[2] for s in segs] == [3 * P, 2 * P]
            out = _segment_attention(qf[r0:r0 + t_loc], segs, 0.16)
            full = _segment_attention(qf, [(kf + layer, (vf + layer).t().contiguous(), T, 0, 0)], 0.16)
            assert torch.allclose(out, full[r0:r0 + t_loc], atol=1e-5)
            assert float(kvx.vt_loc[0, :, t_loc:].abs().sum()) == 0.0  # padding stays zero
            # split form used by the model: local shard first (while the gather is in flight), then the remote segments
            kvx.start()
            loc, rem = kvx.local_segment(), kvx.finish()
            assert kvx.has_remote and loc[2] == t_loc and [s[2] for s in rem] == [2 * P if rank == 0 else 3 * P]
            out2 = _segment_attention(qf[r0:r0 + t_loc], [loc] + rem, 0.16)  # softmax is order independent
            assert torch.allclose(out2, full[r0:r0 + t_loc], atol=1e-5)
        # precision "exact": the fp32 K / V rows of all ranks in global token order
        k_all, v_all = kvx.gather_rows_f32(kf[r0:r0 + t_loc, :64].contiguous(), vf[r0:r0 + t_loc, :64].contiguous())
        assert torch.equal(k_all, kf[:, :64]) and torch.equal(v_all, vf[:, :64])
        # the model passes the token counts it derives from the (shared) list of views: same exchange object, no collective
        assert sh.make_kv_exchange(t_loc, D, torch.float32, torch.device("cpu"), t_all=[3 * P, 2 * P], n_heads=2) is kvx
        try:
            sh.gather_outputs = True
            sh.gather_results([{"idx": torch.arange(3)} for _ in range(lo, hi)], n_views, torch.device("cpu"))
            raise AssertionError("integer outputs must be refused by the fp32 transport")
        except TypeError:
            sh.gather_outputs = False
        res = sh.gather_results([{"x": torch.full((1, 2), float(i))} for i in range(lo, hi)], n_views, torch.device("cpu"))
        assert len(res) == hi - lo  # outputs stay sharded by default
        sh.gather_outputs = True
        mine = [{"x": torch.full((1, 2 + i), float(i)), "c": torch.arange(3 * (i + 1), dtype=torch.float32).view(3, i + 1)} for i in range(lo, hi)]
        res = sh.gather_results(mine, n_views, torch.device("cpu"))   # ragged shapes per view, two tensors per view
        assert [float(r["x"][0, 0]) for r in res] == [0.0, 1.0, 2.0, 3.0, 4.0]
        assert all(r["x"].shape == (1, 2 + i) and torch.equal(r["c"], torch.arange(3 * (i + 1), dtype=torch.float32).view(3, i + 1)) for i, r in enumerate(res))
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_view_sharding_gloo_world2():
    mp.set_start_method("spawn", force=True)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [mp.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {0: "ok", 1: "ok"}


def _worker_p2p(rank, world, port, ret):
    """the per-peer form: pairwise rounds in ring-distance order, the remote shards handed out one by one in ARRIVAL order"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fast3r_amd.dist import ViewSharding
        sh = ViewSharding(exchange="p2p", p2p_channels=2)   # the two rounds of a world of three run on two communicators
        assert len(sh.p2p_groups) == 2
        n_views, P, D = 7, 24, 128  # uneven: 3 + 2 + 2 views
        lo, hi = sh.my_range(n_views)
        t_loc = (hi - lo) * P
        kvx = sh.make_kv_exchange(t_loc, D, torch.float32, torch.device("cpu"), n_heads=2)
        assert kvx.mode == "p2p" and kvx.t_all == [3 * P, 2 * P, 2 * P]
        g = torch.Generator().manual_seed(0)
        T = n_views * P
        qf, kf, vf = (torch.randn(T, D, generator=g) for _ in range(3))
        r0 = lo * P
        for layer in range(2):
            kvx.k_loc[:t_loc] = kf[r0:r0 + t_loc] + layer
            kvx.vt_loc[0, :, :t_loc] = (vf[r0:r0 + t_loc] + layer).t()
            kvx.start()
            groups = kvx.remote_groups()
            assert len(groups) == world - 1
            arrived, order = [kvx.local_segment()], []
            for wait, segs in groups:
                wait()
                assert len(segs) == 1
                order.append(kvx.remote_position_of(segs[0]))
                arrived += segs
            pos = kvx.positions()
            assert order == [pos[(rank - d) % world] for d in range(1, world)]  # round d brings the shard of rank - d
            out = _segment_attention(qf[r0:r0 + t_loc], arrived, 0.16)  # softmax is order independent
            full = _segment_attention(qf, [(kf + layer, (vf + layer).t().contiguous(), T, 0, 0)], 0.16)
            assert torch.allclose(out, full[r0:r0 + t_loc], atol=1e-5)
            assert [s[2] for s in kvx.finish()] == [t for r, t in enumerate(kvx.t_all) if r != rank]  # everything at once still works
        # ---- exchange="auto": one untimed warm-up layer with each form, then PROBE_LAYERS timed layers with each (alternating), every rank
        # times what the local launch did not hide, the maxima over ranks pick the winner -- the same one on every rank -- and later
        # layers use it; a new geometry probes again
        sa = ViewSharding(exchange="auto", p2p_channels=2)
        kva = sa.make_kv_exchange(t_loc, D, torch.float32, torch.device("cpu"), n_heads=2)
        used = []
        n_probe = sa.PROBE_WARM + 2 * sa.PROBE_LAYERS
        for layer in range(n_probe + 2):
            kva.k_loc[:t_loc] = kf[r0:r0 + t_loc] + layer
            kva.vt_loc[0, :, :t_loc] = (vf[r0:r0 + t_loc] + layer).t()
            sa.begin_layer(kva)
            used.append(kva.mode)
            kva.start()
            kva.mark_local_done()
            arrived = [kva.local_segment()]
            first = True
            for wait, segs in kva.remote_groups():
                wait()
                if first and segs:
                    kva.mark_remote_start()
                    first = False
                arrived += segs
            sa.end_layer(kva)
            out = _segment_attention(qf[r0:r0 + t_loc], arrived, 0.16)
            full = _segment_attention(qf, [(kf + layer, (vf + layer).t().contiguous(), T, 0, 0)], 0.16)
            assert torch.allclose(out, full[r0:r0 + t_loc], atol=1e-5)
        assert used[:n_probe] == ["allgather", "p2p"] * (n_probe // 2) and sa.exchange_in_use in ("allgather", "p2p")
        assert used[n_probe] == used[n_probe + 1] == sa.exchange_in_use
        votes = [None] * world
        dist.all_gather_object(votes, sa.exchange_in_use)
        assert len(set(votes)) == 1
        kvb = sa.make_kv_exchange(t_loc, 2 * D, torch.float32, torch.device("cpu"), n_heads=4)   # another geometry: the choice is open again
        assert sa.exchange_in_use == "auto (probing)" and kvb is not kva
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_view_sharding_per_peer_exchange_gloo_world3():
    mp.set_start_method("spawn", force=True)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [mp.Process(target=_worker_p2p, args=(r, 3, port, ret)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {0: "ok", 1: "ok", 2: "ok"}
