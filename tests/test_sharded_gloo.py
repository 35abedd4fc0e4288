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

    def gather(self, local_idx, table_local):
        return table_local[local_idx.long()].contiguous()


def _worker(rank, world, port, nfeat, E, B, F, q, dedup=False):
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
        shard = RowShardedTable(shard_rows(table, rank, world), nfeat, None, ops=NumpyShardOps(), dedup=dedup)
        rows, perm = shard.lookup(ids)
        got = rows[perm.long()].view(B, F, E)
        ok = bool(torch.equal(got, table[ids]))
        q.put((rank, ok, int(rows.shape[0])))
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
