"""The row-sharded lookup with the REAL device kernels at world size 2 on one GPU: two processes share cuda:0,
the exchanges go through gloo (staged through host memory, sharded.py), everything else — HIP routing with and
without de-duplication, owner-side gather, the fused kernel over (received rows, perm) — is the product path.
Each rank's result must be bit-equal to the replicated-table result for its own samples."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, dedup, micro, q, protocol="exact", whole="auto"):
    for p in (ROOT, os.path.join(ROOT, "arm-net_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from golden_util import load
        from model_util import build_model
        dev = "cuda:0"
        meta, sd, _, _, _ = load("g2_criteo_1h_a2.0_stress")
        c = meta["ctor"]
        g = torch.Generator().manual_seed(50 + rank)
        # exact protocol: ragged (the ranks hold different batch sizes); fixed-capacity protocol: equal shapes
        B = 333 + 7 * rank if protocol == "exact" else 333
        ids = torch.randint(0, c["nfeat"], (B, c["nfield"]), generator=g)
        ids[0, :3] = torch.tensor([0, c["nfeat"] - 1, 1])     # boundary rows, both owners
        ids[1, :] = ids[1, 0]                                 # duplicates in one sample
        vals = torch.rand(B, c["nfield"], generator=g)
        m = build_model(meta, sd, dev)
        with torch.no_grad():
            want = m.arm_block(ids.to(dev), vals.clone().to(dev))
            m.shard_embedding()
            m._shard.dedup = dedup
            m._shard.protocol = protocol
            m._shard.micro_batches = micro
            m._shard.whole_shard = whole
            assert m._shard.world == world and m._shard._via_host
            got = m.arm_block(ids.to(dev), vals.clone().to(dev))
            assert not (protocol == "fixed" and m._shard.overflowed())
        q.put((rank, bool(torch.equal(got, want)), float((got - want).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dedup,micro,protocol,whole", [
    (False, 1, "exact", "auto"), (True, 1, "exact", "auto"), (True, 3, "exact", "auto"), (False, 500, "exact", "auto"),
    (False, 1, "fixed", "auto"), (True, 1, "fixed", "auto"), (True, 1, "fixed", False), (False, 3, "fixed", "auto"),
    (True, 50, "fixed", "auto")])
def test_two_ranks_on_one_gpu_are_bit_equal_to_replicated(dedup, micro, protocol, whole):
    """micro > 1: the lookups of slice m+1 run on a side stream beside the fused kernel of slice m; 500 slices for
    333 / 340 samples also exercises empty slices (every rank still takes part in every exchange).  (True, 1, "fixed"):
    13 k lookups of a smaller table -> the whole-shard exchange (all-gather + direct addresses) when `whole` is "auto",
    the de-duplicated request lists when it is off"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = (29621 + 2 * int(dedup) + (micro > 1) + 4 * (micro > 100) + 8 * (protocol == "fixed") + 16 * (micro == 50)
            + 32 * (whole is False))
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dedup, micro, q, protocol, whole)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, err in res:
        assert ok, f"rank {rank}: sharded result differs from replicated by {err}"


def _worker_in_flight(rank, world, port, q, whole):
    for p in (ROOT, os.path.join(ROOT, "arm-net_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from golden_util import load
        from model_util import build_model
        dev = "cuda:0"
        meta, sd, _, _, _ = load("g2_criteo_1h_a2.0_stress")
        c = meta["ctor"]
        g = torch.Generator().manual_seed(90 + rank)
        B, K = 4096, 6                                         # 160k lookups of a 5k-row table: de-duplication on
        batches = [(torch.randint(0, c["nfeat"], (B, c["nfield"]), generator=g).to(dev),
                    torch.rand(B, c["nfield"], generator=g).to(dev)) for _ in range(K)]
        m = build_model(meta, sd, dev)
        m.check_ids = False                                    # no host sync inside a step
        with torch.no_grad():
            want = [m.arm_block(i, v.clone()) for i, v in batches]
            m.shard_embedding()
            m._shard.dedup, m._shard.protocol, m._shard.whole_shard = True, "fixed", whole
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            for s in streams:
                s.wait_stream(torch.cuda.current_stream())
            got = []
            for k, (i, v) in enumerate(batches):               # K steps enqueued back to back, two in flight
                with torch.cuda.stream(streams[k % 2]):
                    got.append(m.arm_block(i, v.clone()))
            torch.cuda.synchronize()
            assert not m._shard.overflowed()
        q.put((rank, all(bool(torch.equal(a, b)) for a, b in zip(got, want)), 0.0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("whole", ["auto", False])
def test_steps_in_flight_on_two_streams_are_bit_equal_to_replicated(whole):
    """bench.py --in-flight 2: consecutive steps alternate between two streams; nothing but the overflow flag is shared
    between steps (the de-duplication workspace is per stream)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_in_flight, args=(r, 2, 29677 + (whole is False), q, whole)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, _ in res:
        assert ok, f"rank {rank}: a step in flight differs from the replicated result"
