"""The workload bench.py measures: its flags, the model of a weight regime, the synthetic batches."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--alpha", type=float, default=2.0)
    ap.add_argument("--regime", choices=["fresh", "stress", "both"], default="both",
                    help="fresh = the reference's initialisers (what `value` reports); stress = SURVEY §8c "
                         "sparse-support weights; both = measure both, `value` from fresh")
    ap.add_argument("--settle-ms", type=float, default=2500.0,
                    help="untimed run of the step before the warm-up steps, so that the device clocks have settled")
    ap.add_argument("--rotate", type=int, default=4,
                    help="distinct (ids, vals, out) batches cycled through by the steps (working set > 256 MiB MALL)")
    ap.add_argument("--batch", type=int, default=65536, help="samples per GPU per step")
    ap.add_argument("--nfield", type=int, default=39)
    ap.add_argument("--nfeat", type=int, default=1_000_000)
    ap.add_argument("--nemb", type=int, default=16)
    ap.add_argument("--nhid", type=int, default=32)
    ap.add_argument("--nhead", type=int, default=1, help=">1 selects models.armnet (multi-head)")
    ap.add_argument("--ensemble", action="store_true",
                    help="build the model with the DNN ensemble branch (BASELINE.json configs[4]); it only enters full_forward")
    ap.add_argument("--ids", choices=["uniform", "zipf"], default="uniform")
    ap.add_argument("--shard", choices=["replicate", "rows", "both"], default=None,
                    help="replicate = every rank holds the table (no collective); rows = table row-sharded over "
                         "the ranks, RCCL all-to-all lookup (SURVEY §8e); both (default when N > 1) = value from "
                         "replicate plus a row_sharded object measured in the same run")
    ap.add_argument("--dedup", choices=["auto", "on", "off"], default="auto",
                    help="row-sharded variant: per-rank id de-duplication before the exchange")
    ap.add_argument("--protocol", choices=["fixed", "exact"], default="fixed",
                    help="row-sharded variant: fixed-capacity equal-split exchanges (no host sync) or exact splits")
    ap.add_argument("--micro-batches", type=int, default=1,
                    help="row-sharded variant: slices per step whose exchanges overlap the previous slice's kernel")
    ap.add_argument("--whole-shard", choices=["auto", "off"], default="auto",
                    help="row-sharded variant: all-gather the shards when the batch covers the table (auto) or always "
                         "answer request lists (off)")
    ap.add_argument("--stream-communicators", action="store_true",
                    help="row-sharded steps in flight get one process group (RCCL communicator) per stream instead of sharing one "
                         "(RowShardedTable.data_groups; no gain through RCCL at world size 1 — unmeasured on several GPUs, so off)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="row-sharded variant: steps kept in flight on alternating streams (1 = one stream)")
    ap.add_argument("--config4-capacity-factor", type=float, default=1.06,
                    help="slot slack of the fixed-capacity exchange in the configs[3] measurement")
    ap.add_argument("--hot-rows", type=int, default=0,
                    help="row-sharded variant: rows [0, N) of the (frequency-ordered) id space are replicated on every rank "
                         "and never routed (SURVEY §8e's hot-row lever; meaningful with --ids zipf)")
    ap.add_argument("--no-config4", action="store_true",
                    help="N > 1: skip the extra row-sharded measurement of BASELINE.json configs[3] (nfeat 100 M, nemb 64)")
    ap.add_argument("--no-config5", action="store_true",
                    help="N > 1: skip the data-parallel measurement of BASELINE.json configs[4] (armnet + DNN ensemble, Avazu shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-alphas", action="store_true", help="skip the alpha = 1.7 / 1.5 measurements reported beside `value`")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def build_model(a, device, rank=0, world=1, regime=None):
    regime = regime or a.regime
    torch.manual_seed(2025)                     # the reference's default seed (train.py:47)
    # row-sharded runs never materialise the full table: the module gets a 16-row placeholder and the
    # rank's shard is generated directly on the device below
    nfeat_mod = 16 if a.shard == "rows" else a.nfeat
    if a.nhead == 1:
        from models.armnet_1h import ARMNetModel
        m = ARMNetModel(a.nfield, nfeat_mod, a.nemb, a.alpha, a.nhid, a.nemb, 2, 256, 0.0, bool(getattr(a, "ensemble", False)), 2, 256)
    else:
        from models.armnet import ARMNetModel
        m = ARMNetModel(a.nfield, nfeat_mod, a.nemb, a.nhead, a.alpha, a.nhid, 2, 256, 0.0, bool(getattr(a, "ensemble", False)), 2, 256)
    if regime == "stress":
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            w = m.embedding.embedding.weight
            w.copy_(torch.randn(w.shape, generator=g) * 0.5)
            m.attn_layer.query.mul_(4.0)
            m.arm_bn.running_mean.copy_(torch.rand(m.arm_bn.running_mean.shape, generator=g) + 0.5)
            m.arm_bn.running_var.copy_(torch.rand(m.arm_bn.running_var.shape, generator=g) * 1.5 + 0.5)
    m.eval()
    # round 6: the default product path is what is timed — the in-kernel id range test is LIVE and its report deferred
    # (block.IdStatus: a pinned host word, no host sync per step).  The row-sharded step's check is an all-reduce + host
    # sync per call (every rank must raise together): off there, as in a serving loop that polls per N steps.
    m.check_ids = a.shard != "rows"
    m = m.to(device)
    if a.shard == "rows":
        from armnet_hip.sharded import RowShardedTable
        n_local = (a.nfeat - rank + world - 1) // world
        bound = (6.0 / (a.nfeat + a.nemb)) ** 0.5 if regime == "fresh" else 0.87    # xavier-uniform / stress
        gdev = torch.Generator(device=device).manual_seed(2025 + rank)
        shard = (torch.rand(n_local, a.nemb, device=device, generator=gdev) * 2 - 1) * bound
        m._shard = RowShardedTable(shard, a.nfeat, None, protocol=a.protocol,
                                   dedup={"auto": "auto", "on": True, "off": False}[a.dedup],
                                   hot_rows=int(getattr(a, "hot_rows", 0)))
        m._shard.micro_batches = a.micro_batches
        m._shard.whole_shard = "auto" if getattr(a, "whole_shard", "auto") == "auto" else False
        m.nfeat = a.nfeat
    return m


def make_batch(a, rank, device, k=0):
    """batch k of rank `rank`: ids uniform (or Zipf) over nfeat, vals ~ U[0,1)"""
    g = torch.Generator().manual_seed(2025 + 1000 * rank + 7919 * k)
    if a.ids == "uniform":
        ids = torch.randint(0, a.nfeat, (a.batch, a.nfield), generator=g, dtype=torch.int64)
    else:                                       # Zipf(1.05)-like skew, reported separately
        u = torch.rand(a.batch, a.nfield, generator=g, dtype=torch.float64)
        ids = (a.nfeat ** u - 1).clamp_(0, a.nfeat - 1).to(torch.int64)
    vals = torch.rand(a.batch, a.nfield, generator=g)
    return ids.to(device), vals.to(device), ids, vals
