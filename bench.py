#!/usr/bin/env python3
"""bench.py — ARM-Net forward hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the fused block (clamp, embedding gather*value,
folded gates, alpha-entmax, value weighting, interaction, exp, eval-BN) over one synthetic
Criteo-shaped batch resident in HBM:  armnet_1h, nfield=39, nemb=16, nhid=32, nfeat=1M, B=65536 per
GPU (BASELINE.json configs[1]); alpha=2.0 is the reference's own Criteo setting (run.sh:18-19).
`value` = samples/s of that block over all ranks (weak scaling: every rank owns B samples; eval-mode
samples are independent, so there is no data-path collective with a replicated 64 MB table).
The same JSON line also carries
  full_forward  the whole ARMNetModel.forward to logits (adds the MLP head on hipBLASLt),
  roofline      algorithmic HBM bytes / measured kernel time against the 8 TB/s peak,
  cpu_baseline  the CPU oracle (C restatement of the reference's op chain, OpenMP) on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--alpha", type=float, default=2.0)
    ap.add_argument("--regime", choices=["fresh", "stress"], default="fresh",
                    help="fresh = the reference's initialisers; stress = SURVEY §8c sparse-support weights")
    ap.add_argument("--batch", type=int, default=65536, help="samples per GPU per step")
    ap.add_argument("--nfield", type=int, default=39)
    ap.add_argument("--nfeat", type=int, default=1_000_000)
    ap.add_argument("--nemb", type=int, default=16)
    ap.add_argument("--nhid", type=int, default=32)
    ap.add_argument("--nhead", type=int, default=1, help=">1 selects models.armnet (multi-head)")
    ap.add_argument("--ids", choices=["uniform", "zipf"], default="uniform")
    ap.add_argument("--shard", choices=["replicate", "rows", "both"], default=None,
                    help="replicate = every rank holds the table (no collective); rows = table row-sharded over "
                         "the ranks, RCCL all-to-all lookup (SURVEY §8e); both (default when N > 1) = value from "
                         "replicate plus a row_sharded object measured in the same run")
    ap.add_argument("--micro-batches", type=int, default=1,
                    help="row-sharded variant: slices per step whose exchanges overlap the previous slice's kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def build_model(a, device, rank=0, world=1):
    torch.manual_seed(2025)                     # the reference's default seed (train.py:47)
    # row-sharded runs never materialise the full table: the module gets a 16-row placeholder and the
    # rank's shard is generated directly on the device below
    nfeat_mod = 16 if a.shard == "rows" else a.nfeat
    if a.nhead == 1:
        from models.armnet_1h import ARMNetModel
        m = ARMNetModel(a.nfield, nfeat_mod, a.nemb, a.alpha, a.nhid, a.nemb, 2, 256, 0.0, False, 2, 256)
    else:
        from models.armnet import ARMNetModel
        m = ARMNetModel(a.nfield, nfeat_mod, a.nemb, a.nhead, a.alpha, a.nhid, 2, 256, 0.0, False, 2, 256)
    if a.regime == "stress":
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            w = m.embedding.embedding.weight
            w.copy_(torch.randn(w.shape, generator=g) * 0.5)
            m.attn_layer.query.mul_(4.0)
            m.arm_bn.running_mean.copy_(torch.rand(m.arm_bn.running_mean.shape, generator=g) + 0.5)
            m.arm_bn.running_var.copy_(torch.rand(m.arm_bn.running_var.shape, generator=g) * 1.5 + 0.5)
    m.eval()
    m.check_ids = False                         # no host sync inside the timed region
    m = m.to(device)
    if a.shard == "rows":
        from armnet_hip.sharded import RowShardedTable
        n_local = (a.nfeat - rank + world - 1) // world
        bound = (6.0 / (a.nfeat + a.nemb)) ** 0.5 if a.regime == "fresh" else 0.87    # xavier-uniform / stress
        gdev = torch.Generator(device=device).manual_seed(2025 + rank)
        shard = (torch.rand(n_local, a.nemb, device=device, generator=gdev) * 2 - 1) * bound
        m._shard = RowShardedTable(shard, a.nfeat, None)
        m._shard.micro_batches = a.micro_batches
        m.nfeat = a.nfeat
    return m


def make_batch(a, rank, device):
    g = torch.Generator().manual_seed(2025 + 1000 * rank)
    if a.ids == "uniform":
        ids = torch.randint(0, a.nfeat, (a.batch, a.nfield), generator=g, dtype=torch.int64)
    else:                                       # Zipf(1.05)-like skew, reported separately
        u = torch.rand(a.batch, a.nfield, generator=g, dtype=torch.float64)
        ids = (a.nfeat ** u - 1).clamp_(0, a.nfeat - 1).to(torch.int64)
    vals = torch.rand(a.batch, a.nfield, generator=g)
    return ids.to(device), vals.to(device), ids, vals


def timed(fn, steps, sync_all):
    """Exactly `steps` calls bracketed by barrier + synchronize; wall ms and HIP-event ms."""
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    sync_all()
    return wall_ms, ev0.elapsed_time(ev1)


def cpu_baseline(a, model, ids_cpu, vals_cpu):
    """The CPU oracle (oracle/armnet_oracle.c, kind "port") on this host's cores, bounded sample."""
    from oracle import armnet_oracle as orc
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    threads = orc.effective_cpus()              # affinity and cgroup CPU quota, not just the core count
    orc.set_threads(threads)
    variant = "1h" if a.nhead == 1 else "mh"
    n = min(a.batch, 65536)
    ids = ids_cpu[:n].numpy()
    done, t_used, passes = 0, 0.0, 0
    while t_used < a.cpu_seconds and passes < 3:
        v = vals_cpu[:n].numpy().copy()
        t0 = time.perf_counter()
        orc.arm_block(variant, ids, v, sd, a.alpha)
        t_used += time.perf_counter() - t0
        done += n
        passes += 1
        if passes == 1 and t_used > a.cpu_seconds / 2:
            break
    # one-thread line: a 2048-sample slice through the same entry point
    orc.set_threads(1)
    n1 = min(n, 2048)
    v = vals_cpu[:n1].numpy().copy()
    t0 = time.perf_counter()
    orc.arm_block(variant, ids[:n1], v, sd, a.alpha)
    t1 = time.perf_counter() - t0
    orc.set_threads(threads)
    return {"value": done / t_used, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{passes} pass(es) of the first {n} samples of the same batch through "
                      f"oracle_arm_block (50-step bisection, OpenMP, {threads} threads)",
            "one_thread": {"value": n1 / t1, "unit": "samples/s", "sample": f"{n1} samples, 1 thread"}}


def pmc_traffic(a):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), when they were taken on
    exactly this workload; None otherwise (counters cannot be read from inside the timed process)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            t = json.load(f)
        key = f"nfield={a.nfield} nemb={a.nemb} nhid={a.nhid} nhead={a.nhead} B={a.batch} alpha={a.alpha} ids={a.ids}"
        return t["traffic_bytes_per_launch"] if t["workload"] == key else None
    except Exception:
        return None


def main():
    a = parse()
    # stdout carries exactly ONE line (the JSON): RCCL prints a version banner to fd 1 when the first communicator
    # is created, so everything else written to fd 1 by any library goes to stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.shard is None:
        a.shard = "both" if world > 1 else "replicate"
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or bool(os.environ.get("ARMNET_BENCH_FORCE_DIST"))   # FORCE: exercise RCCL with 1 rank
    # developer knobs for a 1-GPU box: ARMNET_BENCH_BACKEND=gloo ARMNET_BENCH_DEVICE=0 run the N > 1 flow with
    # several ranks sharing one device (the exchanges of the row-sharded variant are then staged through the host)
    backend = os.environ.get("ARMNET_BENCH_BACKEND", "nccl")
    if "ARMNET_BENCH_DEVICE" in os.environ:
        local = int(os.environ["ARMNET_BENCH_DEVICE"])
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    model = build_model(a, dev, rank, world)
    ids, vals, ids_cpu, vals_cpu = make_batch(a, rank, dev)
    O = a.nhead * a.nhid

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step_block():
        with torch.no_grad():
            return model.arm_block(ids, vals)

    def step_full():
        with torch.no_grad():
            return model({"id": ids, "value": vals})

    for _ in range(a.warmup):
        step_block()
        step_full()
    wall_ms, ev_ms = timed(step_block, a.steps, sync_all)
    full_wall_ms, _ = timed(step_full, a.steps, sync_all)

    t = torch.tensor([wall_ms, ev_ms, full_wall_ms], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_ms, ev_ms, full_wall_ms = t.tolist()

    # Row-sharded variant: same model, same batch, the table row-sharded over the ranks and fetched by all-to-all.
    # It runs AFTER the headline numbers are final and under a watchdog: whatever happens in there (an exception on
    # one rank, a collective that never completes) must not cost the main line.
    sharded = {"ms": float("nan"), "err": None, "done": False}
    if a.shard == "both":
        import threading

        def run_sharded():
            try:
                torch.cuda.set_device(local)
                model.shard_embedding()
                model._shard.micro_batches = a.micro_batches
                for _ in range(a.warmup):
                    step_block()
                ms, _ = timed(step_block, a.steps, sync_all)
                ts = torch.tensor([ms], device=dev, dtype=torch.float64)
                if use_dist:
                    dist.all_reduce(ts, op=dist.ReduceOp.MAX)
                sharded["ms"] = float(ts.item())
            except Exception as e:  # noqa: BLE001
                sharded["err"] = f"{type(e).__name__}: {e}"
            sharded["done"] = True

        th = threading.Thread(target=run_sharded, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("ARMNET_BENCH_SHARDED_TIMEOUT", "180")))
        if not sharded["done"]:
            sharded["err"] = "timeout: the row-sharded measurement did not complete (collective hang?)"
    sharded_ms, sharded_err = sharded["ms"], sharded["err"]
    if sharded_err is None and a.shard == "both":
        model._shard = None

    if rank == 0:
        ms_per_step = wall_ms / a.steps
        kernel_ms = ev_ms / a.steps                       # back-to-back launches of ONE kernel per step
        value = world * a.batch * a.steps / (wall_ms * 1e-3)
        read_b = a.nfield * (8 + 4 + 4 * a.nemb)          # ids int64 + vals + F rows      
        write_b = 4 * O * a.nemb                          # post-BN activations
        alg_bytes = (read_b + write_b) * a.batch
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "samples/sec, ARM-Net forward (fused embedding + ARM interaction block), "
                      "Criteo nfield=39 nemb=16 B=65536",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"armnet{'_1h' if a.nhead == 1 else ''} fused block a2..a9, nfield={a.nfield} "
                                   f"nfeat={a.nfeat} nemb={a.nemb} nhid={a.nhid} nhead={a.nhead} "
                                   f"alpha={a.alpha} B={a.batch}/GPU, ids {a.ids} int64, weights {a.regime}-init, "
                                   f"eval mode", "global_batch": world * a.batch,
                       "parallelism": (f"dp{world} (table replicated, no collective)" if a.shard != "rows" else
                                       f"dp{world} x row-sharded table (mod {world}), RCCL all-to-all lookup")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(a),
                         "kernel": "armnet::fused_mfma_kernel", "kernel_ms": kernel_ms,
                         "alg_bytes_per_sample": read_b + write_b,
                         "folded_tflops": 4 * O * a.nfield * a.nemb * a.batch / (kernel_ms * 1e-3) / 1e12},
            "full_forward": {"value": world * a.batch * a.steps / (full_wall_ms * 1e-3), "unit": "samples/s",
                             "ms_per_step": full_wall_ms / a.steps,
                             "note": "fused block + MLP head 2x256 (torch/hipBLASLt fp32) to logits"},
        }
        if a.shard == "both" and sharded_err is not None:
            line["row_sharded"] = {"error": sharded_err}
        elif a.shard == "both":
            line["row_sharded"] = {
                "value": world * a.batch * a.steps / (sharded_ms * 1e-3), "unit": "samples/s",
                "ms_per_step": sharded_ms / a.steps,
                "note": f"same block with the table row-sharded (row i on rank i mod {world}): HIP routing with per-rank "
                        f"id de-duplication (direct-address mark + scan), all_to_all_single of int32 row indices, "
                        f"owner-side gather, all_to_all_single of {a.nemb * 4}-byte rows (one per DISTINCT id: at most "
                        f"{a.batch * a.nfield * a.nemb * 4 / 1e6:.0f} MB per rank per step, {(world - 1) / world:.0%} of it "
                        f"across xGMI), fused kernel over (rows, perm)"}
        if world == 1 and not a.no_cpu_baseline and a.shard == "replicate":
            line["cpu_baseline"] = cpu_baseline(a, model, ids_cpu, vals_cpu)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if a.shard == "both" and not sharded["done"]:
        os._exit(0)                 # a stuck collective: leave without tearing the process group down
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
