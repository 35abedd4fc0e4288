"""The N > 1 path on CPU: world_size-2 gloo processes run the row-sharded lookup protocol
(armnet_hip/sharded.py) with a numpy test double for the two device kernels, and check that
rows[perm] reproduces table[ids] on every rank — i.e. routing, split matrix, both all-to-alls and
the inverse permutation are right.  The device kernels themselves are covered by the -m gpu tests."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyShardOps:
    """Test double with the same contract as HipShardOps (armnet_shard_route_ids / gather)."""

    def route(self, ids_flat, R, nfeat, dedup=False):
        ids = ids_flat.numpy().astype(np.int64)
        if dedup:       # every distinct id once, grouped by owner, sorted by local row index
            L = (nfeat + R - 1) // R
            pos = (ids % R) * L + ids // R
            upos, inv = np.unique(pos, return_inverse=True)
            counts = np.bincount(upos // L, minlength=R).astype(np.int32)
            send_local = np.zeros(ids.size, np.int32)
            send_local[: upos.size] = (upos % L).astype(np.int32)
            return torch.from_numpy(counts), torch.from_numpy(send_local), torch.from_numpy(inv.astype(np.int32))
        owner = ids % R
        order = np.argsort(owner, kind="stable")
        perm = np.empty(ids.size, np.int32)
        perm[order] = np.arange(ids.size, dtype=np.int32)
        counts = np.bincount(owner, minlength=R).astype(np.int32)
        send_local = (ids[order] // R).astype(np.int32)
        return torch.from_numpy(counts), torch.from_numpy(send_local), torch.from_numpy(perm)

    def gather(self, local_idx, table_local, out=None):
        got = table_local[local_idx.long()].contiguous()
        if out is not None:
            out.copy_(got)
            return out
        return got

    def direct_perm(self, ids_flat, R, nfeat, id_status=None):
        """contract of armnet_shard_direct_perm: address of the row in the all-gathered (padded) shards"""
        ids = ids_flat.numpy().astype(np.int64)
        L = (nfeat + R - 1) // R
        return torch.from_numpy(((ids % R) * L + ids // R).astype(np.int32))

    def route_fixed(self, ids_flat, R, nfeat, cap, dedup, overflow, id_status=None, hot_rows=0):
        """contract of armnet_shard_route_fixed(_hot): the slots directly; positions inside a slot are ANY unique assignment
        (here: reversed arrival order without de-duplication, sorted by local index with it); ids < hot_rows take no slot
        and point at row R * cap + id"""
        ids = ids_flat.numpy().astype(np.int64)
        owner, local = ids % R, ids // R
        send_pad = np.zeros(R * cap, np.int32)
        perm_pad = np.empty(ids.size, np.int32)
        hot = ids < hot_rows
        perm_pad[hot] = (R * cap + ids[hot]).astype(np.int32)
        for o in range(R):
            sel = np.nonzero((owner == o) & ~hot)[0]
            if dedup:
                u, inv = np.unique(local[sel], return_inverse=True)
                slot = inv
            else:
                u = local[sel][::-1]
                slot = np.arange(sel.size)[::-1]
            k = min(u.size, cap)
            send_pad[o * cap: o * cap + k] = u[:k]
            if u.size > cap:
                overflow |= 1
            perm_pad[sel] = o * cap + np.where(slot < cap, slot, 0)
        return torch.from_numpy(send_pad), torch.from_numpy(perm_pad)

    def pad_route(self, counts, send_local, perm, R, cap, overflow):
        """contract of armnet_shard_pad_route: R equal slots of cap indices, index 0 in the unused entries"""
        c = counts.numpy().astype(np.int64)
        start = np.concatenate([[0], np.cumsum(c)])
        send_pad = np.zeros(R * cap, np.int32)
        for o in range(R):
            k = min(int(c[o]), cap)
            send_pad[o * cap: o * cap + k] = send_local.numpy()[start[o]: start[o] + k]
        p = perm.numpy().astype(np.int64)
        owner = np.searchsorted(start[1:], p, side="right")
        owner = np.minimum(owner, R - 1)
        slot = p - start[owner]
        if (c > cap).any():
            overflow |= 1
        perm_pad = (owner * cap + np.where(slot < cap, slot, 0)).astype(np.int32)
        return torch.from_numpy(send_pad), torch.from_numpy(perm_pad)


def _worker(rank, world, port, nfeat, E, B, F, q, dedup=False, protocol="exact", capacity_factor=1.25):
    sys.path.insert(0, os.path.join(ROOT, "arm-net_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from armnet_hip.sharded import RowShardedTable, shard_rows
        g = torch.Generator().manual_seed(7)
        table = torch.randn(nfeat, E, generator=g)                       # same full table on every rank
        g2 = torch.Generator().manual_seed(100 + rank)
        ids = torch.randint(0, nfeat, (B, F), generator=g2)
        ids[0, :3] = torch.tensor([0, nfeat - 1, 1])                     # boundary rows, both owners
        ids[1, :] = ids[1, 0]                                            # duplicates in one sample
        shard = RowShardedTable(shard_rows(table, rank, world), nfeat, None, ops=NumpyShardOps(), dedup=dedup,
                                protocol=protocol, capacity_factor=capacity_factor)
        rows, perm = shard.lookup(ids)
        got = rows[perm.long()].view(B, F, E)
        ok = bool(torch.equal(got, table[ids]))
        over = shard.overflowed() if protocol == "fixed" else False
        if over:                                       # what sharded_arm_block does: redo with the exact protocol
            rows2, perm2 = shard.lookup(ids, protocol="exact")
            ok = bool(torch.equal(rows2[perm2.long()].view(B, F, E), table[ids]))
        q.put((rank, ok, int(rows.shape[0]), over))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dedup", [(2, False), (3, False), (2, True)])
def test_sharded_lookup_protocol_gloo(world, dedup):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + world + (10 if dedup else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, 1001, 8, 37, 5, q, dedup)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert [r[1] for r in res] == [True] * world, res
    assert all((r[2] == 37 * 5) if not dedup else (r[2] <= 37 * 5) for r in res)


def _run(world, port, args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port) + args[:4] + (q,) + args[4:]) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return sorted(q.get(timeout=5) for _ in range(world))


@pytest.mark.parametrize("world,dedup", [(2, False), (3, False), (2, True)])
def test_fixed_capacity_protocol_without_host_sync_gloo(world, dedup):
    """the no-sync protocol: equal-split exchanges of padded slots; rows[perm_pad] reproduces table[ids] on every rank,
    the received buffer has world * cap rows on every rank, nothing overflows at the default slack"""
    port = 31500 + os.getpid() % 2000 + world + (10 if dedup else 0)
    res = _run(world, port, (1001, 8, 37, 5, dedup, "fixed", 1.25))
    assert [r[1] for r in res] == [True] * world, res
    assert len({r[2] for r in res}) == 1 and res[0][2] % world == 0      # equal slots on every rank
    assert not any(r[3] for r in res)


@pytest.mark.parametrize("world", [2, 3])
def test_whole_shard_exchange_when_the_batch_covers_the_table_gloo(world):
    """1 500 lookups of a 1 001-row table: the de-duplicated slot would be the whole shard, so the owners all-gather
    their shards (padded to ceil(nfeat / R) rows: 501 + 500, 334 + 334 + 333) and perm is the direct address"""
    port = 35500 + os.getpid() % 2000 + world
    res = _run(world, port, (1001, 8, 300, 5, True, "fixed", 1.25))
    assert [r[1] for r in res] == [True] * world, res
    assert all(r[2] == world * ((1001 + world - 1) // world) for r in res), res     # R * L rows received
    assert not any(r[3] for r in res)


def test_fixed_capacity_overflow_is_flagged_on_every_rank_and_repeated_exactly():
    """a slot far too small: the device flag is raised, the all-reduce makes every rank see it, the exact protocol
    then gives the right rows"""
    port = 33500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    # skewed ids: every id is owned by rank 0 -> its slot overflows whatever the slack
    procs = [ctx.Process(target=_skew_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert all(r[1] for r in res) and all(r[2] for r in res), res


def _skew_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "arm-net_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from armnet_hip.sharded import RowShardedTable, shard_rows
        nfeat, E, B, F = 1000, 4, 64, 6
        table = torch.randn(nfeat, E, generator=torch.Generator().manual_seed(7))
        ids = torch.randint(0, nfeat // 2, (B, F), generator=torch.Generator().manual_seed(rank)) * 2   # all even
        if rank == 1:
            ids = ids + 0                                              # rank 1 also only asks rank 0
        shard = RowShardedTable(shard_rows(table, rank, world), nfeat, None, ops=NumpyShardOps(), dedup=False,
                                protocol="fixed", capacity_factor=1.0)
        rows, perm = shard.lookup(ids)
        over = shard.overflowed()
        rows2, perm2 = shard.lookup(ids, protocol="exact")
        ok = bool(torch.equal(rows2[perm2.long()].view(B, F, E), table[ids]))
        q.put((rank, ok, over))
    finally:
        dist.destroy_process_group()


def test_numpy_double_matches_route_contract():
    """counts / send_local / perm invariants the HIP kernel is tested against on the GPU."""
    ops = NumpyShardOps()
    ids = torch.randint(0, 1000, (999,), generator=torch.Generator().manual_seed(3))
    counts, send_local, perm = ops.route(ids, 4, 1000)
    assert int(counts.sum()) == ids.numel()
    assert sorted(perm.tolist()) == list(range(ids.numel()))
    owner_of_pos = np.repeat(np.arange(4), counts.numpy())
    assert np.array_equal(owner_of_pos[perm.numpy()], ids.numpy() % 4)
    assert np.array_equal(send_local.numpy()[perm.numpy()], ids.numpy() // 4)


# ---- round 3: a re-cut shard must drop the whole-shard cache; agreed slot sizes; the collective error poll ----------

def _recut_worker(rank, world, port, q, whole):
    sys.path.insert(0, os.path.join(ROOT, "arm-net_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from armnet_hip.sharded import RowShardedTable, shard_rows
        nfeat, E, B, F = 1001, 8, 300, 5
        old = torch.randn(nfeat, E, generator=torch.Generator().manual_seed(7))
        new = torch.randn(nfeat, E, generator=torch.Generator().manual_seed(8))
        ids = torch.randint(0, nfeat, (B, F), generator=torch.Generator().manual_seed(100 + rank))
        shard = RowShardedTable(shard_rows(old, rank, world), nfeat, None, ops=NumpyShardOps(), dedup=True,
                                protocol="fixed")
        shard.whole_shard = whole
        rows, perm = shard.lookup(ids)
        ok_old = bool(torch.equal(rows[perm.long()].view(B, F, E), old[ids]))
        path = shard.last_path
        shard.table_local = shard_rows(new, rank, world)        # what ArmNetBase._refresh_shard does after a weight update
        rows, perm = shard.lookup(ids)
        got = rows[perm.long()].view(B, F, E)
        q.put((rank, ok_old, bool(torch.equal(got, new[ids])), bool(torch.equal(got, old[ids])), path))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("whole", ["auto", False])
def test_recut_shard_serves_the_new_rows_gloo(whole):
    """round-2 verdict, weak 1: after the shard is re-cut (load_state_dict / optimizer step / invalidate_folded) the
    whole-shard exchange kept all-gathering its cached copy of the OLD shard"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + os.getpid() % 2000 + (whole is False)
    procs = [ctx.Process(target=_recut_worker, args=(r, 2, port, q, whole)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    for rank, ok_old, ok_new, stale, path in res:
        assert ok_old and ok_new and not stale, res
        assert path == ("whole_shards" if whole == "auto" else "fixed")


def _ragged_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "arm-net_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from armnet_hip.sharded import RowShardedTable, shard_rows
        nfeat, E, F = 5000, 4, 6
        table = torch.randn(nfeat, E, generator=torch.Generator().manual_seed(7))
        shard = RowShardedTable(shard_rows(table, rank, world), nfeat, None, ops=NumpyShardOps(), dedup=False,
                                protocol="fixed")
        oks, sizes = [], []
        # step 0: unequal batches (the ranks agree on the larger one); step 1: a ragged last batch on one rank only (7 vs
        # 64 samples: local sizes would give different slot sizes, and 8 * n >= nfeat a different de-duplication choice);
        # step 2: a LARGER batch than agreed on rank 1 only — its slots may overflow; poll() raises the agreed size
        import warnings
        grown = []
        for step, B in enumerate([(64, 50), (7, 64), (64, 400)]):
            ids = torch.randint(0, nfeat, (B[rank], F), generator=torch.Generator().manual_seed(10 * step + rank))
            with warnings.catch_warnings(record=True) as wl:
                warnings.simplefilter("always")
                rows, perm = shard.lookup(ids)
            grown.append(any("exceeds the agreed step size" in str(w.message) for w in wl))
            sizes.append(int(rows.shape[0]))
            over, _ = shard.poll(None)
            if over:
                rows, perm = shard.lookup(ids, protocol="exact")
            oks.append(bool(torch.equal(rows[perm.long()].view(B[rank], F, E), table[ids])))
        q.put((rank, oks, sizes, shard.slot_lookups, grown))
    finally:
        dist.destroy_process_group()


def test_fixed_protocol_with_unequal_batches_agrees_on_one_slot_size_gloo():
    """round-2 advisor finding (medium): slot size and exchange path were derived from the LOCAL batch, so ranks with
    different batch sizes ran mismatched equal-split exchanges.  They now come from one agreed step size."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert all(all(r[1]) for r in res), res
    assert res[0][2] == res[1][2], res                     # the same receive-buffer size on both ranks, every step
    assert res[0][3] == res[1][3] == 400 * 6, res          # the larger step raised the agreed size on BOTH ranks
    # round-3 advisor finding: the rank whose step outgrew the agreed size is told so (its slots may overflow until the
    # next poll: an unverified caller must not find out from wrong rows) — and only that rank, only at that step
    assert res[0][4] == [False, False, False] and res[1][4] == [False, False, True], res


def _poll_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "arm-net_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from armnet_hip.sharded import RowShardedTable, shard_rows
        table = torch.randn(100, 4)
        shard = RowShardedTable(shard_rows(table, rank, world), 100, None, ops=NumpyShardOps())
        status = torch.tensor([1 if rank == 1 else 0], dtype=torch.int32)      # only rank 1 saw a bad id
        q.put((rank,) + shard.poll(status))
    finally:
        dist.destroy_process_group()


def test_bad_id_on_one_rank_is_seen_by_every_rank_gloo():
    """round-2 advisor finding (medium): the IndexError used to be raised from the rank-local flag BEFORE the collective,
    leaving the other ranks blocked in it"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 38500 + os.getpid() % 2000
    procs = [ctx.Process(target=_poll_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res == [(0, False, True), (1, False, True)], res


# ---- round 5: hot-row replication (SURVEY.md 8e's third lever) -----------------------------------------------------------

def _zipf_ids(nfeat, shape, seed):
    """bench.py's skewed id stream: log-uniform over [0, nfeat) — the head of the id space carries most lookups"""
    u = torch.rand(*shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
    return (nfeat ** u - 1).clamp_(0, nfeat - 1).to(torch.int64)


def _hot_worker(rank, world, port, q, dedup, hot_rows):
    sys.path.insert(0, os.path.join(ROOT, "arm-net_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from armnet_hip.sharded import RowShardedTable, shard_rows
        nfeat, E, B, F = 20011, 8, 97, 6
        old = torch.randn(nfeat, E, generator=torch.Generator().manual_seed(7))
        new = torch.randn(nfeat, E, generator=torch.Generator().manual_seed(8))
        ids = _zipf_ids(nfeat, (B, F), 100 + rank)
        ids[0, :4] = torch.tensor([0, hot_rows - 1, hot_rows, nfeat - 1])          # both sides of the hot boundary
        out = {}
        for name, hr in (("cold", 0), ("hot", hot_rows)):
            shard = RowShardedTable(shard_rows(old, rank, world), nfeat, None, ops=NumpyShardOps(), dedup=dedup,
                                    protocol="fixed", hot_rows=hr)
            shard.whole_shard = False
            rows, perm = shard.lookup(ids)
            ok = bool(torch.equal(rows[perm.long()].view(B, F, E), old[ids]))
            over = shard.overflowed()
            cap_rows = int(rows.shape[0]) - hr                                     # R * cap: what the row exchange carries
            # a step with FEWER hot ids than the first one: may overflow (flagged on every rank), repaired exactly
            ids2 = torch.randint(0, nfeat, (B, F), generator=torch.Generator().manual_seed(200 + rank))
            rows2, perm2 = shard.lookup(ids2)
            over2 = shard.overflowed()
            if over2:
                rows2, perm2 = shard.lookup(ids2, protocol="exact")
            ok2 = bool(torch.equal(rows2[perm2.long()].view(B, F, E), old[ids2]))
            # re-cut shard (weight update): the hot copy must follow
            shard.table_local = shard_rows(new, rank, world)
            rows3, perm3 = shard.lookup(ids)
            ok3 = bool(torch.equal(rows3[perm3.long()].view(B, F, E), new[ids]))
            out[name] = (ok, over, cap_rows, ok2, ok3, shard.last_path)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dedup", [(2, False), (2, True), (3, False)])
def test_hot_rows_are_served_locally_and_shrink_the_exchange_gloo(world, dedup):
    """RowShardedTable(hot_rows=N): rows[perm] still reproduces table[ids] bit for bit (ids on both sides of the hot
    boundary, duplicates, a colder second step, a re-cut shard), the hot ids take no slot, and the slots — what the row
    exchange carries — shrink with the cold fraction of the skewed stream"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 38500 + os.getpid() % 2000 + 3 * world + dedup
    procs = [ctx.Process(target=_hot_worker, args=(r, world, port, q, dedup, 2048)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    for rank, out in res:
        for name in ("cold", "hot"):
            ok, over, cap_rows, ok2, ok3, path = out[name]
            assert ok and ok2 and ok3 and not over and path in ("fixed", "exact"), (rank, name, out)
    assert len({out["hot"][2] for _, out in res}) == 1                  # the same slot size on every rank
    cold_rows, hot_rows_ = res[0][1]["cold"][2], res[0][1]["hot"][2]
    # without de-duplication the slots shrink with the cold fraction of the lookups (>= 3x here); a de-duplicating route
    # already sizes its slots by DISTINCT ids (round 5), of which the hot head is a smaller share (>= 2x)
    assert hot_rows_ * (2 if dedup else 3) <= cold_rows, (cold_rows, hot_rows_)


def test_hot_rows_on_one_rank_without_a_process_group():
    """world 1, no torch.distributed: the gather writes straight into the front of the consumer's buffer"""
    sys.path.insert(0, os.path.join(ROOT, "arm-net_amd"))
    from armnet_hip.sharded import RowShardedTable
    nfeat, E = 5003, 4
    table = torch.randn(nfeat, E, generator=torch.Generator().manual_seed(1))
    ids = _zipf_ids(nfeat, (50, 7), 3)
    for dedup in (False, True):
        shard = RowShardedTable(table.clone(), nfeat, None, ops=NumpyShardOps(), dedup=dedup, hot_rows=512)
        shard.whole_shard = False
        rows, perm = shard.lookup(ids)
        assert torch.equal(rows[perm.long()].view(50, 7, E), table[ids])
        assert torch.equal(rows[-512:], table[:512]) and not shard.overflowed()
        # later steps drawn from the same stream have a few more or fewer cold / distinct ids than the first one: the slots
        # are sized from the first step's ESTIMATE plus slack and must not be clamped to it (round 5: they were, on one rank)
        for seed in range(4, 10):
            ids2 = _zipf_ids(nfeat, (50, 7), seed)
            rows2, perm2 = shard.lookup(ids2)
            assert torch.equal(rows2[perm2.long()].view(50, 7, E), table[ids2])
        assert not shard.overflowed()
    with pytest.raises(ValueError):
        RowShardedTable(table, nfeat, None, ops=NumpyShardOps(), hot_rows=nfeat + 1)


def test_empty_slice_with_hot_rows_joins_the_exchanges_and_a_colder_stream_relearns_its_slots():
    """round-5 advisor findings (medium x 2), one rank with the numpy double.  (1) hot_rows > 0 and an EMPTY lookup (a ragged
    rank, the empty last micro-batch): used to raise on that rank alone while its peers blocked in all_to_all_single; it now
    sends filler requests and returns the hot rows behind an empty permutation.  (2) The cold fraction that sizes the slots
    beside the hot rows was measured once: a stream that turned colder overflowed on EVERY later step (poll() re-learned the
    distinct count only).  The poll() that sees the overflow now drops the fraction and the agreed cold step size; the next
    lookup measures both again, so the step after the repair fits."""
    sys.path.insert(0, os.path.join(ROOT, "arm-net_amd"))
    from armnet_hip.sharded import RowShardedTable
    nfeat, E, N = 5003, 4, 512
    table = torch.randn(nfeat, E, generator=torch.Generator().manual_seed(1))
    ids = _zipf_ids(nfeat, (50, 7), 3)
    for dedup in (False, True):
        shard = RowShardedTable(table.clone(), nfeat, None, ops=NumpyShardOps(), dedup=dedup, hot_rows=N)
        shard.whole_shard = False
        rows, perm = shard.lookup(ids)                                             # sizes the slots
        rows0, perm0 = shard.lookup(ids[:0])                                       # (1)
        assert perm0.numel() == 0 and torch.equal(rows0[-N:], table[:N]) and rows0.shape[0] == rows.shape[0]
        assert not shard.overflowed()
    # (2): first step all hot -> cold fraction 0, slots of one lookup; then a cold stream
    shard = RowShardedTable(table.clone(), nfeat, None, ops=NumpyShardOps(), dedup=False, hot_rows=N)
    shard.whole_shard = False
    hot_ids = torch.randint(0, N, (50, 7), generator=torch.Generator().manual_seed(5))
    rows, perm = shard.lookup(hot_ids)
    assert torch.equal(rows[perm.long()].view(50, 7, E), table[hot_ids]) and not shard.overflowed()
    small = rows.shape[0] - N
    cold_ids = torch.randint(N, nfeat, (50, 7), generator=torch.Generator().manual_seed(6))
    shard.lookup(cold_ids)
    assert shard.overflowed()                                                      # flagged, sizes dropped
    overflows = 0
    for seed in range(7, 11):                                                      # four more cold steps: they fit now
        ids2 = torch.randint(N, nfeat, (50, 7), generator=torch.Generator().manual_seed(seed))
        rows2, perm2 = shard.lookup(ids2)
        if shard.overflowed():
            overflows += 1
        else:
            assert torch.equal(rows2[perm2.long()].view(50, 7, E), table[ids2])
    assert overflows == 0 and rows2.shape[0] - N > 10 * small


def test_ops_object_without_out_support_on_the_one_rank_hot_row_path():
    """round-5 advisor finding (low): an ops object whose gather() has no `out` parameter used to raise TypeError before the
    documented fallback copy was reached"""
    sys.path.insert(0, os.path.join(ROOT, "arm-net_amd"))
    from armnet_hip.sharded import RowShardedTable

    class NoOut(NumpyShardOps):
        def gather(self, idx, table):
            return NumpyShardOps.gather(self, idx, table)

    nfeat, E = 3001, 4
    table = torch.randn(nfeat, E, generator=torch.Generator().manual_seed(2))
    ids = _zipf_ids(nfeat, (40, 5), 9)
    shard = RowShardedTable(table.clone(), nfeat, None, ops=NoOut(), dedup=False, hot_rows=256)
    shard.whole_shard = False
    rows, perm = shard.lookup(ids)
    assert torch.equal(rows[perm.long()].view(40, 5, E), table[ids])


def test_slot_capacity_clamps_only_to_a_true_upper_bound():
    sys.path.insert(0, os.path.join(ROOT, "arm-net_amd"))
    from armnet_hip.sharded import fixed_ingress_bytes, slot_capacity
    assert slot_capacity(1000, 1, 10**6, False, 1.25, upper=1000) == 1000          # all lookups of the step: nobody can ask for more
    assert slot_capacity(1000, 1, 10**6, False, 1.25) > 1250                          # an estimate: slack on top, no clamp
    assert slot_capacity(10**6, 8, 10**6, True, 1.25) == 125000                       # de-duplicated: never more than the shard
    assert fixed_ingress_bytes(2_555_904, 8, 10**6, 16, dedup=False) == slot_capacity(2_555_904, 8, 10**6, False) * 68 * 7
    assert fixed_ingress_bytes(2_555_904, 8, 10**6, 16, dedup=True, n_distinct=430_000) < fixed_ingress_bytes(2_555_904, 8, 10**6, 16, dedup=True)
