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
Every step works on the NEXT of --rotate (default 4) distinct batches — own ids, vals and output buffer,
4 x 165 MB = 660 MB per rotation — so that nothing but the 64 MB table (the model's parameter, re-read by every
step of a real serving loop too) can be served from the 256 MiB Infinity Cache.
The same JSON line also carries
  regimes       the same measurement under both weight regimes: "fresh" (the reference's own initialisers:
                random-init weights of the architecture, what `value` is) and "stress" (sparse
                supports, several solver iterations per row — what trained weights look like),
  full_forward  the whole ARMNetModel.forward to logits (adds the MLP head),
  roofline      algorithmic HBM bytes / measured kernel time against the 8 TB/s peak, plus the flop-side
                fraction, the fraction of the access pattern's own measured ceiling and the binding limit,
  cpu_baseline  the CPU oracle (C restatement of the reference's op chain, OpenMP) on this host.
"""
import argparse
import hashlib
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
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 runs at the fp32 vector rate

# the parts live in benchlib/ (importable, testable); re-exported here so that `import bench` keeps its surface
from benchlib.workload import parse, build_model, make_batch  # noqa: E402,F401
from benchlib.timing import AGREE, SETTLE_NOISE, settle_clocks, median, timed  # noqa: E402,F401
from benchlib.baselines import cpu_baseline, host_twin, aten_chain_block, cpu_baseline_aten  # noqa: E402,F401
from benchlib.evidence import kernel_src_sha, committed_measurements, live_pattern_ceiling  # noqa: E402,F401
from benchlib.errors import ERROR_RC, error_line  # noqa: E402,F401


def main():
    a = parse()
    # stdout carries exactly ONE line (the JSON): RCCL prints a version banner to fd 1 when the first communicator
    # is created, so everything else written to fd 1 by any library goes to stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def fail(what, msg, world=None, print_line=True):
        """leave with the error line (rank 0 / the launcher only: the line is ONE) and the kind's exit code, without
        tearing a possibly wedged process group down"""
        sys.stderr.write(f"bench.py: {what}: {msg}\n")
        if print_line:
            os.write(json_fd, (json.dumps(error_line(a, what, msg, world)) + "\n").encode())
        os._exit(ERROR_RC[what])

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU over RCCL), exactly the
        # command the driver would use; rank 0 of the child job prints the JSON line on the inherited stdout.  Under a
        # deadline: a child job that wedges (an RCCL collective that never returns takes its watchdogs' os._exit paths,
        # but a rank stuck INSIDE the runtime may not) is killed as a process group and reported as an error line.
        import signal
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        limit = float(os.environ.get("ARMNET_BENCH_LAUNCH_TIMEOUT", "1800"))
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, start_new_session=True)
        try:
            out, _ = child.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            os.killpg(child.pid, signal.SIGKILL)
            child.wait()
            fail("launch_timeout", f"the {a.gpus}-rank job did not finish within {limit:g} s and was killed")
        lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.strip().startswith("{")]
        if lines:
            os.write(json_fd, (lines[-1] + "\n").encode())
            sys.exit(child.returncode)
        fail("ranks", f"the {a.gpus}-rank job exited with code {child.returncode} without printing a line (see stderr)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.shard is None:
        a.shard = "both" if world > 1 else "replicate"
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or bool(os.environ.get("ARMNET_BENCH_FORCE_DIST"))   # FORCE: exercise RCCL with 1 rank
    if world == 1 and use_dist:                     # a plain `python bench.py` has no rendezvous in its environment
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29531")):
            os.environ.setdefault(k, v)
    # developer knobs for a 1-GPU box: ARMNET_BENCH_BACKEND=gloo ARMNET_BENCH_DEVICE=0 run the N > 1 flow with
    # several ranks sharing one device (the exchanges of the row-sharded variant are then staged through the host)
    backend = os.environ.get("ARMNET_BENCH_BACKEND", "nccl")
    if "ARMNET_BENCH_DEVICE" in os.environ:
        local = int(os.environ["ARMNET_BENCH_DEVICE"])
    if world != a.gpus:
        fail("ranks", f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run", print_line=rank == 0)
    # the whole run under a deadline (N > 1): whatever wedges — a barrier of the replicated windows, a collective outside
    # the watched row-sharded sections — ends as an error line and a non-zero exit code instead of a hang
    if use_dist:
        import threading
        deadline = float(os.environ.get("ARMNET_BENCH_DEADLINE", "1500"))
        dl = threading.Timer(deadline, lambda: fail("deadline", f"rank {rank}: the run did not finish within {deadline:g} s "
                                                               "(a collective that never returned?)", world, rank == 0))
        dl.daemon = True
        dl.start()
    try:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        torch.zeros(1, device=dev)
    except Exception as e:  # noqa: BLE001
        fail("device", f"rank {rank}: cuda:{local} is not usable: {type(e).__name__}: {e}", world, rank == 0)
    ranks_seen = 1
    if use_dist:
        import datetime
        try:
            tmo = datetime.timedelta(seconds=float(os.environ.get("ARMNET_BENCH_INIT_TIMEOUT", "300")))
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=tmo)
            else:
                dist.init_process_group(backend, timeout=tmo)
            # the first collective (RCCL builds its communicator here): how many ranks are really in the job
            t = torch.ones(1, device=dev if backend == "nccl" else "cpu", dtype=torch.int32)
            dist.all_reduce(t)
            ranks_seen = int(t.item())
        except Exception as e:  # noqa: BLE001
            fail("init", f"rank {rank}: {backend} process group over {world} ranks: {type(e).__name__}: {e}", world, rank == 0)
        if ranks_seen != world:
            fail("ranks", f"the first all-reduce saw {ranks_seen} ranks, WORLD_SIZE is {world}", world, rank == 0)
    O = a.nhead * a.nhid
    NB = max(1, a.rotate)

    # NB distinct batches, each with its own output buffer: step i works on batch i mod NB
    batches = [make_batch(a, rank, dev, k) for k in range(NB)]
    ids_cpu, vals_cpu = batches[0][2], batches[0][3]
    outs = [torch.empty(a.batch, O, a.nemb, device=dev) for _ in range(NB)] if a.shard != "rows" else [None] * NB

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist:
        def agree(done):
            t = torch.tensor([1 if done else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())
        AGREE[0] = agree

    # forward clamps x['value'] IN PLACE (armnet_1h.py:81), so a value buffer that has been through the block once is
    # already clamped and the kernel's write-back would never fire again.  SURVEY §8d asks for a fresh copy of the values
    # for every iteration: every step of a window gets its own buffer from a pool that is restored from pristine copies,
    # untimed, before the window starts (0.1 % of U[0,1) values are below the clamp's 1e-3: the write-back is live in
    # every timed step).
    POOL = min(a.warmup + a.steps, 256)
    pristine = [batches[k][1].clone() for k in range(NB)]
    vals_pool = [pristine[j % NB].clone() for j in range(POOL)]
    turn = [0]

    def restore_vals():
        torch._foreach_copy_(vals_pool, [pristine[j % NB] for j in range(POOL)])
        turn[0] = 0

    def make_steps(model):
        def step_block():
            j = turn[0] % POOL
            turn[0] += 1
            with torch.no_grad():
                return model.arm_block(batches[j % NB][0], vals_pool[j], out=outs[j % NB])

        def step_full():
            j = turn[0] % POOL
            turn[0] += 1
            with torch.no_grad():
                return model({"id": batches[j % NB][0], "value": vals_pool[j]})
        return step_block, step_full

    def window(fn):
        """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize: (wall ms, HIP-event ms)"""
        restore_vals()
        for _ in range(a.warmup):
            fn()
        return timed(fn, a.steps, sync_all)

    def windows(fn, n):
        return [window(fn) for _ in range(n)]

    regimes = ["fresh", "stress"] if a.regime == "both" else [a.regime]
    head = regimes[0]                                     # `value` regime: fresh (random-init weights) unless --regime
    res, models, meas = {}, {}, {}
    # `value` = the MEDIAN of several windows of K steps each, spread over the life of the process (round-3 verdict,
    # item 1: one 2 ms window right after start-up measured the clock ramp on the driver's box): 3 back to back after the
    # clocks settled, then one after every other section of this run (full forward, batches in flight, the other weight
    # regime, the other alphas).  All of them are in the line (`value_windows_ms`) with their spread.
    head_windows, settle_info = [], {}
    head_step = [None]

    def head_window(after):
        """one more window of the headline step, after another measurement section changed what the device was doing"""
        settle_clocks(head_step[0], min(a.settle_ms, 30.0), cap_ms=200.0)
        w = window(head_step[0])
        head_windows.append((after,) + w)

    for regime in regimes:
        model = models[regime] = build_model(a, dev, rank, world, regime)
        step_block, step_full = make_steps(model)
        m = meas[regime] = {"cold": 0.0, "fl_block": 0.0, "fl_full": 0.0}
        if regime == head:
            head_step[0] = step_block
            if a.settle_ms > 0:
                # the same W + K steps from a cold device, for the record (`cold_start` in the line)
                m["cold"] = window(step_block)[0]
            n_chunks, flat = settle_clocks(step_block, a.settle_ms, cap_ms=a.settle_ms + 3000.0)
            settle_info.update(chunks_of_64_steps=n_chunks, plateau_reached=flat, cap_ms=a.settle_ms + 3000.0,
                               spread_of_last_8_chunks=SETTLE_NOISE[0])
            head_windows.append(("clock settle",) + window(step_block))
            m["block"] = None                             # filled from head_windows at the end
        else:
            settle_clocks(step_block, min(a.settle_ms, 300.0), cap_ms=1000.0)   # (the device is warm: the head regime ran)
            m["block"] = windows(step_block, 3)
        settle_clocks(step_full, min(a.settle_ms, 50.0), cap_ms=300.0)
        m["full"] = windows(step_full, 3)
        if regime == head:
            head_window("full_forward")
        # the same steps with `in_flight` batches on alternating streams (a serving loop that does not wait for batch i
        # before it enqueues batch i+1): consecutive launches overlap, which hides the gap between dependent launches of
        # one stream, the block prologue and the tail.  Reported beside `value`, which stays the one-stream number.
        if a.in_flight > 1 and a.shard != "rows":
            streams = [torch.cuda.Stream(device=dev) for _ in range(a.in_flight)]
            for s_ in streams:
                s_.wait_stream(torch.cuda.current_stream())
            for key, base in (("fl_block", step_block), ("fl_full", step_full)):
                def step_fl(base=base):
                    with torch.cuda.stream(streams[turn[0] % len(streams)]):
                        return base()
                settle_clocks(step_fl, min(a.settle_ms, 50.0), cap_ms=300.0)
                m[key] = median([w[0] for w in windows(step_fl, 3)])
            if regime == head:
                head_window("batches_in_flight")
        elif regime != head:
            pass
        if regime != head:
            head_window(f"regime {regime}")

    def reduce_results():
        """MAX over the ranks of every window (each window is bracketed by barriers), then the medians"""
        flat = [x for w in head_windows for x in w[1:]]
        for regime in regimes:
            m = meas[regime]
            for key in ("block", "full"):
                flat += [x for w in (m[key] or []) for x in w]
            flat += [m["cold"], m["fl_block"], m["fl_full"]]
        t = torch.tensor(flat, device=dev, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        it = iter(t.tolist())
        hw = [(w[0], next(it), next(it)) for w in head_windows]
        for regime in regimes:
            m = meas[regime]
            got = {}
            for key in ("block", "full"):
                got[key] = [(next(it), next(it)) for _ in (m[key] or [])]
            blk = [(w[1], w[2]) for w in hw] if regime == head else got["block"]
            cold, flb, flf = next(it), next(it), next(it)
            res[regime] = [median([w[0] for w in blk]), median([w[1] for w in blk]),
                           median([w[0] for w in got["full"]]), cold, flb, flf]
        return hw

    res_provisional = reduce_results()       # the row-sharded pre-run below sizes itself from the replicated step time
    model = models[head]
    sharded_overflow = None

    # Row-sharded variant: same model, same batches, the table row-sharded over the ranks and fetched by all-to-all.
    # It runs AFTER the headline numbers are final and under a watchdog: whatever happens in there (an exception on
    # one rank, a collective that never completes) must not cost the main line.
    sharded = {"ms": float("nan"), "err": None, "done": False, "by_in_flight": {}, "in_flight": 1, "modes": {}}
    if a.shard == "both":
        import threading

        def measure_mode(whole):
            """one exchange of the row-sharded step (whole = "auto": whole-shard all-gather when the batch covers the
            table | False: the request-list all-to-all protocol north_star names), with 1 and a.in_flight steps in flight"""
            sh = model._shard
            sh.whole_shard = whole
            sh.slot_lookups = None                           # agreed again (all ranks arrive here together)
            mode = {"by_in_flight": {}}
            # `in_flight` steps on alternating streams (a serving loop with that many batches in flight): the row
            # exchange of step i+1 overlaps the fused kernel of step i.  The collectives stay in issue order on the
            # process group's own stream; the timed region ends with a device-wide synchronize.  Measured with one
            # step in flight first, then with a.in_flight: the faster one is the mode's number, both are reported
            # (and a failure of the second keeps the first).
            n_settle = int(min(2000, max(0.0, a.settle_ms) / max(1e-3, 2.0 * res[head][0] / a.steps)))
            if backend != "nccl":
                n_settle = min(n_settle, 10)                 # host-staged developer path: a step takes tens of ms
            for nfl in sorted({1, max(1, a.in_flight)}):
                try:
                    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
                    for s_ in streams:
                        s_.wait_stream(torch.cuda.current_stream())
                    turn = [0]

                    def step_sharded():
                        k = turn[0] % NB
                        s_ = streams[turn[0] % nfl]
                        turn[0] += 1
                        with torch.no_grad(), torch.cuda.stream(s_):
                            return model.arm_block(batches[k][0], batches[k][1])

                    # clocks: every step holds collectives, so the untimed pre-run is a step COUNT that is the same
                    # on every rank (from the all-reduced replicated step time), not a time budget
                    for _ in range(n_settle):
                        step_sharded()
                    for _ in range(a.warmup):
                        step_sharded()
                    ms, _ = timed(step_sharded, a.steps, sync_all)
                    ts = torch.tensor([ms], device=dev, dtype=torch.float64)
                    if use_dist:
                        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
                    # the fixed-capacity protocol never looked at a count on the host: check its overflow flag ONCE,
                    # after the timed steps (a set flag means some lookups read a wrong row: the number would be void)
                    if bool(sh.overflowed()):
                        mode["overflow"] = True
                    mode["by_in_flight"][str(nfl)] = float(ts.item())
                except Exception as e:  # noqa: BLE001
                    if not mode["by_in_flight"]:
                        raise
                    mode["in_flight_err"] = f"in_flight={nfl}: {type(e).__name__}: {e}"
            mode["exchange"] = getattr(sh, "last_path", None)
            # the lookup half of the step ALONE on one stream — routing, both exchanges, the owner-side gather — so that the
            # curve explains itself: step time against (this + the kernel's time with the table local)
            try:
                lt = [0]

                def lookup_only():
                    k = lt[0] % NB
                    lt[0] += 1
                    with torch.no_grad():
                        return sh.lookup(batches[k][0])

                for _ in range(a.warmup):
                    lookup_only()
                ms_l, _ = timed(lookup_only, a.steps, sync_all)
                tl = torch.tensor([ms_l], device=dev, dtype=torch.float64)
                if use_dist:
                    dist.all_reduce(tl, op=dist.ReduceOp.MAX)
                sh.overflowed()                              # (resets the flag; these steps' rows were not used)
                mode["lookup_ms"] = float(tl.item())
            except Exception as e:  # noqa: BLE001
                mode["lookup_err"] = f"{type(e).__name__}: {e}"
            best = min(mode["by_in_flight"], key=mode["by_in_flight"].get)
            mode["ms"], mode["in_flight"] = mode["by_in_flight"][best], int(best)
            # bytes one rank RECEIVES from the other ranks per step, and what that means per xGMI link (R - 1 peers, one
            # link each): a link-bound result is then legible in the line
            R, E4 = world, a.nemb * 4
            n_lk = a.batch * a.nfield
            if mode["exchange"] == "whole_shards":
                per_peer = ((a.nfeat + R - 1) // R) * E4
            else:
                n_slot = sh.slot_lookups or n_lk
                dd = (8 * n_slot >= a.nfeat) if sh.dedup == "auto" else bool(sh.dedup)
                cap = sh.capacity(min(sh.slot_distinct or n_slot, n_slot) if dd else n_slot, dd,
                                  upper=None if sh.hot_rows else n_slot)
                per_peer = cap * (E4 + 4)                    # the rows it asked for + the request list it answers
                mode["slot_rows"] = cap
            mode["ingress_bytes_per_rank_per_step"] = per_peer * (R - 1)
            if R > 1:
                mode["implied_gb_per_s_per_link"] = per_peer / (mode["ms"] / a.steps * 1e-3) / 1e9
            return mode

        def run_sharded():
            try:
                torch.cuda.set_device(local)
                # the row-sharded step's id check is one all-reduce + host read per call (every rank has to raise together):
                # off inside the timed steps, as in a serving loop that polls every N steps (`overflowed()` is checked after them)
                model.check_ids = False
                # one communicator per step in flight: the exchanges of consecutive steps then run on different RCCL streams
                dgs = [dist.new_group() for _ in range(max(1, a.in_flight))] if (use_dist and a.in_flight > 1 and a.stream_communicators) else None
                model.shard_embedding(hot_rows=a.hot_rows, data_groups=dgs)
                sharded["communicators"] = len(dgs) if dgs else 1
                model._shard.micro_batches = a.micro_batches
                model._shard.protocol = a.protocol
                model._shard.dedup = {"auto": "auto", "on": True, "off": False}[a.dedup]
                plan = ["auto", False] if a.whole_shard == "auto" else [False]
                for whole in plan:
                    m_ = measure_mode(whole)
                    if m_["exchange"] in sharded["modes"]:
                        continue                             # "auto" resolved to the request lists anyway
                    sharded["modes"][m_["exchange"]] = m_
                ok = {k: v for k, v in sharded["modes"].items() if not v.get("overflow")}
                if not ok:
                    sharded["overflow"] = True
                    ok = sharded["modes"]
                # `value` at N > 1 is the protocol north_star names — the request-list all-to-all ("fixed") — whenever it
                # ran; the whole-shard exchange and the replicated table are reported beside it (round-3 verdict, item 9)
                best = "fixed" if "fixed" in ok else min(ok, key=lambda k: ok[k]["ms"])
                sharded.update(ms=ok[best]["ms"], in_flight=ok[best]["in_flight"], exchange=best,
                               by_in_flight=ok[best]["by_in_flight"], in_flight_err=ok[best].get("in_flight_err"))
            except Exception as e:  # noqa: BLE001
                sharded["err"] = f"{type(e).__name__}: {e}"
            model.check_ids = True
            sharded["done"] = True

        th = threading.Thread(target=run_sharded, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("ARMNET_BENCH_SHARDED_TIMEOUT", "180")))
        if not sharded["done"]:
            sharded["err"] = "timeout: the row-sharded measurement did not complete (collective hang?)"
    sharded_ms, sharded_err = sharded["ms"], sharded["err"]
    if sharded_err is None and a.shard == "both":
        if sharded.get("overflow"):
            sharded_err = "a slot of the fixed-capacity exchange overflowed during the timed steps"
        model._shard = None
    elif a.shard == "rows":
        sharded_overflow = bool(model._shard.overflowed())

    # BASELINE.json configs[3] — nfeat = 100 M, nemb = 64: the table that is row-sharded because of its size (25.6 GB),
    # 2.5 M lookups per rank and step into it (request lists, no de-duplication to speak of).  Measured beside the
    # headline at N > 1, under its own watchdog; never touches `value`.
    big = {"done": False, "err": None}
    if a.shard == "both" and world > 1 and sharded["done"] and not a.no_config4 and a.nhead == 1:
        import threading

        def run_big():
            try:
                torch.cuda.set_device(local)
                a4 = argparse.Namespace(**vars(a))
                a4.nemb, a4.nfeat, a4.shard, a4.regime = 64, 100_000_000, "rows", "fresh"
                m4 = build_model(a4, dev, rank, world, "fresh")
                # uniform synthetic ids: an owner's share of the 2.5 M lookups is n/R +- 0.2 % (binomial), and this step
                # is bound by the links, so the slots carry 6 % slack instead of the default 25 % (which is sized for
                # skewed production ids); an overflow would void the number and is checked after the timed steps
                m4._shard.capacity_factor = a.config4_capacity_factor
                if a.in_flight > 1 and a.stream_communicators:
                    m4._shard.data_groups = [dist.new_group() for _ in range(a.in_flight)]
                b4 = [make_batch(a4, rank, dev, k)[:2] for k in range(NB)]
                nfl = max(1, a.in_flight)
                streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
                for s_ in streams:
                    s_.wait_stream(torch.cuda.current_stream())
                turn = [0]

                def step():
                    k = turn[0]
                    turn[0] += 1
                    with torch.no_grad(), torch.cuda.stream(streams[k % nfl]):
                        return m4.arm_block(b4[k % NB][0], b4[k % NB][1])

                for _ in range(3 + (25 if a.settle_ms > 0 else 0)):     # fixed count (collectives inside): ~50-100 ms
                    step()
                steps4 = min(a.steps, 20)
                ms, _ = timed(step, steps4, sync_all)
                ts = torch.tensor([ms], device=dev, dtype=torch.float64)
                dist.all_reduce(ts, op=dist.ReduceOp.MAX)
                big.update(ms=float(ts.item()), steps=steps4, overflow=bool(m4._shard.overflowed()),
                           exchange=m4._shard.last_path, in_flight=nfl,
                           shard_gb=m4._shard.table_local.numel() * 4 / 1e9)
            except Exception as e:  # noqa: BLE001
                big["err"] = f"{type(e).__name__}: {e}"
            big["done"] = True

        th4 = threading.Thread(target=run_big, daemon=True)
        th4.start()
        th4.join(timeout=float(os.environ.get("ARMNET_BENCH_SHARDED_TIMEOUT", "120")))
        if not big["done"]:
            big["err"] = "timeout: the configs[3] measurement did not complete"

    # BASELINE.json configs[4] — armnet + DNN ensemble (full models/armnet.py), Avazu shape (nfield 22, nfeat 2 M, nemb 32,
    # 4 heads x 32), B = 131 072 GLOBAL, data-parallel: tables replicated (2 x 256 MB), the batch split over the ranks, no
    # collective on the data path.  The whole forward to logits (fused block, second lookup, both heads with the ensemble
    # Linear folded in).  N > 1 only (N = 1 stays configs[1]); never touches `value`.
    c5 = {}
    if a.shard == "both" and world > 1 and not a.no_config5 and sharded["done"] and (big["done"] or not big["err"]):
        try:
            a5 = argparse.Namespace(**vars(a))
            a5.nhead, a5.nemb, a5.nfield, a5.nfeat, a5.ensemble, a5.shard = 4, 32, 22, 2_000_000, True, "replicate"
            a5.batch = 131072 // world
            m5 = build_model(a5, dev, rank, world, "fresh")
            b5 = [make_batch(a5, rank, dev, k)[:2] for k in range(NB)]
            t5 = [0]

            def step5():
                k = t5[0] % NB
                t5[0] += 1
                with torch.no_grad():
                    return m5({"id": b5[k][0], "value": b5[k][1]})

            settle_clocks(step5, min(a.settle_ms, 50.0), cap_ms=300.0)
            w5 = []
            for _ in range(3):
                for _ in range(a.warmup):
                    step5()
                w5.append(timed(step5, a.steps, sync_all)[0])
            t = torch.tensor(w5, device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c5 = {"ms": median(t.tolist()), "batch_per_rank": a5.batch}
            del m5, b5
        except Exception as e:  # noqa: BLE001
            c5 = {"err": f"{type(e).__name__}: {e}"}

    # "pin alpha = 2.0 for the headline and report alpha = 1.7 (train.py's default) and 1.5 beside it": the same
    # block, batches and timing loop with the other sparse maps (N = 1 only; never `value`)
    other_alphas = {}
    if world == 1 and a.shard == "replicate" and not a.no_other_alphas:
        for al in (1.7, 1.5):
            if abs(al - a.alpha) < 1e-9:
                continue
            a2 = argparse.Namespace(**vars(a))
            a2.alpha = al
            for regime in regimes:
                m2 = build_model(a2, dev, rank, world, regime)
                step2 = make_steps(m2)[0]
                settle_clocks(step2, min(a.settle_ms, 50.0), cap_ms=300.0)
                w2 = median([w[0] for w in windows(step2, 3)])
                other_alphas.setdefault(str(al), {})[regime] = {"value": a.batch * a.steps / (w2 * 1e-3), "unit": "samples/s",
                                                                "ms_per_step": w2 / a.steps}
                del m2
            head_window(f"alpha {al:g}")
    # ... and three more at the very end: most of the windows lie late in the life of the process, when a freshly leased
    # device has long finished whatever it does in its first seconds
    stuck = (a.shard == "both" and not sharded["done"]) or (big["err"] and not big["done"])   # a collective never returned
    unchecked_ms = None
    if not stuck and a.shard != "rows":
        # the same step WITHOUT the in-kernel id range test (what rounds 1-5 timed as `value`), beside the checked `value`
        models[head].check_ids = False
        settle_clocks(head_step[0], min(a.settle_ms, 30.0), cap_ms=200.0)
        unchecked_ms = median([w[0] for w in windows(head_step[0], 3)])
        models[head].check_ids = True
    for i in range(0 if stuck else 3):
        head_window("end of run" if i == 0 else "previous window")
    ids_ok = None
    if not stuck and a.shard != "rows":
        try:
            models[head].poll()                           # the deferred report of every step this process ran
            ids_ok = True
        except IndexError:
            ids_ok = False
    head_windows_red = res_provisional if stuck else reduce_results()
    live_ceiling = [None]
    if rank == 0 and world == 1 and a.shard == "replicate":
        torch.cuda.synchronize()
        live_ceiling[0] = live_pattern_ceiling(a)
    if rank == 0:
        read_b = a.nfield * (8 + 4 + 4 * a.nemb)          # ids int64 + vals + F rows      
        write_b = 4 * O * a.nemb                          # post-BN activations
        alg_bytes = (read_b + write_b) * a.batch
        flops = 4 * O * a.nfield * a.nemb * a.batch       # folded formulation: the two contractions

        def roof(regime):
            w_ms, e_ms, f_ms = res[regime][:3]
            k_ms = e_ms / a.steps                         # back-to-back launches of ONE kernel per step
            achieved = alg_bytes / (k_ms * 1e-3) / 1e9
            tfl = flops / (k_ms * 1e-3) / 1e12
            traffic, tag, ceil_us, ceil_src = committed_measurements(a, regime)
            if live_ceiling[0] is not None:
                ceil_us, ceil_src = live_ceiling[0], ("tools/ubench/gather_stream run in this process' session, right after "
                                                      "the timed steps (the kernel's memory traffic and nothing else)")
            fr = {"hbm": achieved / HBM_PEAK_GBS, "mfma_fp32": tfl / FP32_MFMA_PEAK_TFLOPS,
                  "access_pattern_ceiling": (ceil_us * 1e-3 / k_ms) if ceil_us else None,
                  # the bytes the fabric really moves (PMC, 128 bytes per read request: a random 64-byte row costs a
                  # whole 128-byte line) over the same kernel time, against the same 8 TB/s
                  "hbm_traffic": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None}
            # requests the L2 sends to the fabric per launch, modelled: one 128-byte line fill per 128 bytes of a random row (a
            # 64-byte row costs a whole one), ids / values as streams of line fills, output as 64-byte write requests — the
            # counters of profiles/r05_*_rocprof_summary.txt agree to 3 % (headline: 2.72 M + 2.10 M measured).  A gather-only
            # kernel sustains 47 G such requests per second from HBM, 55 G from the Infinity Cache
            n_req = (a.batch * a.nfield * ((a.nemb * 4 + 127) // 128) + a.batch * a.nfield * 12 // 128
                     + a.batch * write_b // 64)
            return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "fabric_requests": {"per_launch_model": n_req, "giga_per_s": n_req / (k_ms * 1e-3) / 1e9,
                                        "gather_only_kernel_giga_per_s": {"hbm": 47.0, "infinity_cache": 55.0}},
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tag,
                    # PMC counters cannot be read from inside the timed process: `traffic` is the committed rocprofv3
                    # --pmc pass for this workload AND these kernel sources (null otherwise), not a measurement of this run
                    "traffic_measured_in_run": False,
                    "kernel": "armnet::fused_mfma_kernel", "kernel_ms": k_ms,
                    "alg_bytes_per_sample": read_b + write_b, "alg_bytes_per_launch": alg_bytes,
                    "folded_tflops": tfl, "fractions": fr,
                    "access_pattern_ceiling_us": ceil_us, "access_pattern_ceiling_source": ceil_src,
                    # which limit binds: the memory system's REQUEST rate and instruction issue, together.
                    # A random 64-byte row costs one 128-byte fabric request — 47-55 G requests/s whatever the payload
                    # (profiles/r03_fetch_size_calibration_gather_rows.txt) — so this launch's ~3.9 M requests cannot
                    # finish in under ~70 us (`access_pattern_ceiling`); on the issue side fp32 MFMA cycles ADD to the
                    # VALU cycles of a SIMD (~63 us of pure issue).  Neither byte nor flop roofline is reached.
                    "binding_limit": "memory-request rate (one 128-byte request per random 64-byte row) + instruction "
                                     "issue (fp32 MFMA and VALU cycles add on a SIMD); see fractions"}

        def regime_obj(regime):
            w_ms, e_ms, f_ms = res[regime][:3]
            r = roof(regime)
            return {"value": world * a.batch * a.steps / (w_ms * 1e-3), "unit": "samples/s",
                    "ms_per_step": w_ms / a.steps, "kernel_ms": r["kernel_ms"], "roofline_frac_hbm": r["frac"],
                    "roofline_frac_mfma_fp32": r["fractions"]["mfma_fp32"],
                    "full_forward_samples_per_s": world * a.batch * a.steps / (f_ms * 1e-3)}

        wall_ms, ev_ms, full_wall_ms, cold_wall_ms, fl_block_ms, fl_full_ms = res[head]
        ms_per_step = wall_ms / a.steps
        value = world * a.batch * a.steps / (wall_ms * 1e-3)
        win_ms = [w[1] / a.steps for w in head_windows_red]
        replicated_value, replicated_ms = value, ms_per_step
        # every window of K steps of the replicated-table step this process timed, in order, with the section that ran
        # before it; value / ms_per_step are the MEDIAN window (MAX over ranks per window first)
        win_obj = {"value_windows_ms": win_ms, "value_windows_after": [w[0] for w in head_windows_red],
                   "value_spread": (max(win_ms) - min(win_ms)) / median(win_ms),
                   "value_best": world * a.batch / (min(win_ms) * 1e-3), "clock_settle": settle_info}
        value_is_sharded = a.shard == "both" and sharded_err is None
        parallelism = (f"dp{world} (table replicated, no collective)" if a.shard != "rows" else
                       f"dp{world} x row-sharded table (mod {world}), RCCL all-to-all lookup")
        if a.shard == "both" and sharded_err is None:
            # N > 1: the headline is the row-sharded table north_star asks for; the replicated-table number (what 288 GB
            # of HBM per GPU makes possible for every BASELINE.json table) is reported beside it
            value = world * a.batch * a.steps / (sharded_ms * 1e-3)
            ms_per_step = sharded_ms / a.steps
            parallelism = (f"dp{world} x embedding table row-sharded over the {world} ranks (row i on rank i mod {world}), "
                           f"RCCL all-to-all lookup per step; `replicated` = the same step with the table on every rank")
        ws_mb = NB * (a.batch * a.nfield * 12 + a.batch * O * a.nemb * 4) / 1e6
        line = {
            "metric": "samples/sec, ARM-Net forward (fused embedding + ARM interaction block), "
                      "Criteo nfield=39 nemb=16 B=65536",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("ROW-SHARDED LOOKUP PATH on this rank count (routing + owner gather + fused block from "
                                    "rows), " if a.shard == "rows" else "") +
                                   f"armnet{'_1h' if a.nhead == 1 else ''} fused block a2..a9, nfield={a.nfield} "
                                   f"nfeat={a.nfeat} nemb={a.nemb} nhid={a.nhid} nhead={a.nhead} "
                                   f"alpha={a.alpha} B={a.batch}/GPU, ids {a.ids} int64, weights {head}-init "
                                   f"(random-init), eval mode; steps rotate over {NB} distinct batches "
                                   f"(ids+vals+out = {ws_mb:.0f} MB per rotation > 256 MiB Infinity Cache; only "
                                   f"the {a.nfeat * a.nemb * 4 / 1e6:.0f} MB table is re-read); every step of a window gets a "
                                   f"pristine (unclamped) copy of its values from a pool of {POOL} buffers restored before the "
                                   f"window, so the in-place clamp's write-back is live in the timed steps; the in-kernel id range "
                                   f"test is live (the module's default check_ids: report deferred to the next call / poll(), no "
                                   f"host sync per step)"
                                   + ("" if a.warmup + a.steps <= POOL else f" (first {POOL} steps of a window only)")
                                   + ("; no clock-settling pre-run; " if a.settle_ms <= 0 else
                                      f"; device clocks settled to a plateau by >= {a.settle_ms:g} ms of the same step, untimed; "
                                      if settle_info.get("plateau_reached") else
                                      f"; the clock-settling pre-run hit its {a.settle_ms + 3000.0:g} ms cap WITHOUT reaching a "
                                      f"plateau (see clock_settle); ") +
                                   f"value = median of {len(win_ms)} windows of {a.steps} steps spread over the process",
                       "global_batch": world * a.batch, "parallelism": parallelism, "clock_settle_ms": a.settle_ms,
                       "ids": a.ids},
            # every window of K steps of the headline step this process timed, in order, with the section that ran before
            # it; `value` / `ms_per_step` are the MEDIAN window (MAX over ranks per window first)
            **({} if value_is_sharded else win_obj),
            "roofline": roof(head),
            "regimes": {r: regime_obj(r) for r in regimes},
            "full_forward": {"value": world * a.batch * a.steps / (full_wall_ms * 1e-3), "unit": "samples/s",
                             "ms_per_step": full_wall_ms / a.steps,
                             "note": f"fused block + MLP head 2x256 to logits ({model.mlp.eval_path()})"
                                     + (" + DNN ensemble branch (second table lookup, deep MLP 2x256 on the HIP head, "
                                        "ensemble Linear)" if a.ensemble else "")},
        }
        if unchecked_ms is not None:
            line["id_check"] = {"in_value": True, "all_ids_in_range_at_poll": ids_ok,
                                "unchecked": {"value": world * a.batch * a.steps / (unchecked_ms * 1e-3), "unit": "samples/s",
                                              "ms_per_step": unchecked_ms / a.steps},
                                "note": "`value` runs the module's default path: armnet_fused_fwd_f32 with the id range test on, "
                                        "its flag a pinned host word read at the next call / model.poll() (the reference's GPU "
                                        "behaviour: an asynchronous device-side assert, layers.py:20); `unchecked` = "
                                        "check_ids = False, what rounds 1-5 timed"}
        if other_alphas:
            line["other_alphas"] = dict(other_alphas, note="the same fused block, batches and timing loop with alpha = 1.7 (the "
                                        "reference's argparse default, train.py:33) and 1.5; `value` stays alpha = "
                                        f"{a.alpha:g} (the reference's Criteo setting, run.sh:18-19)")
        if cold_wall_ms:
            line["cold_start"] = {
                "value": world * a.batch * a.steps / (cold_wall_ms * 1e-3), "unit": "samples/s",
                "ms_per_step": cold_wall_ms / a.steps,
                "note": f"the same {a.warmup} + {a.steps} steps taken right after process start, before the {a.settle_ms:g} ms "
                        f"untimed pre-run that lets the device clocks settle (`value` is the steady state of a serving "
                        f"loop; a device coming out of idle runs the same kernel about 20 % slower for its first ~50 ms)"}
        if fl_block_ms:
            line["batches_in_flight"] = {
                "n": a.in_flight, "value": world * a.batch * a.steps / (fl_block_ms * 1e-3), "unit": "samples/s",
                "ms_per_step": fl_block_ms / a.steps,
                "full_forward_samples_per_s": world * a.batch * a.steps / (fl_full_ms * 1e-3),
                "note": f"the same {a.steps} steps enqueued on {a.in_flight} alternating streams: consecutive launches "
                        f"overlap (launch gap, block prologue and tail hidden); `value`, `roofline` and `regimes` are the "
                        f"one-stream numbers"}
        if use_dist:
            line["rccl_ranks_seen"] = ranks_seen
            line["backend"] = backend
        if a.shard == "both":
            line["replicated"] = {"value": replicated_value, "unit": "samples/s", "ms_per_step": replicated_ms,
                                  "note": "table on every rank, batch split, no data-path collective",
                                  **(win_obj if value_is_sharded else {})}
        if a.shard == "both" and sharded_err is not None:
            line["row_sharded"] = {"error": sharded_err,
                                   "note": "`value` falls back to the replicated-table number for this line"}
        elif a.shard == "both":
            def mode_obj(m_):
                o = {"value": world * a.batch * a.steps / (m_["ms"] * 1e-3), "unit": "samples/s",
                     "ms_per_step": m_["ms"] / a.steps, "steps_in_flight": m_["in_flight"],
                     "samples_per_s_by_steps_in_flight": {k: world * a.batch * a.steps / (v * 1e-3)
                                                          for k, v in m_["by_in_flight"].items()},
                     "ingress_bytes_per_rank_per_step": m_["ingress_bytes_per_rank_per_step"],
                     "ingress_bytes_per_rank": m_["ingress_bytes_per_rank_per_step"],
                     "implied_gb_per_s_per_link": m_.get("implied_gb_per_s_per_link"),
                     # the halves of a step, each measured alone: routing + both exchanges + owner gather | the fused kernel
                     # with the table local (the replicated step's HIP-event time)
                     "exchange_ms": (m_["lookup_ms"] / a.steps) if m_.get("lookup_ms") else None,
                     "kernel_ms": ev_ms / a.steps}
                if o["exchange_ms"] and world > 1:
                    o["exchange_gb_per_s_per_link"] = (m_["ingress_bytes_per_rank_per_step"] / (world - 1)) / (o["exchange_ms"] * 1e-3) / 1e9
                for k in ("slot_rows", "overflow", "in_flight_err", "lookup_err"):
                    if m_.get(k) is not None:
                        o[k] = m_[k]
                return o

            line["row_sharded"] = {
                "value": world * a.batch * a.steps / (sharded_ms * 1e-3), "unit": "samples/s",
                "ms_per_step": sharded_ms / a.steps, "steps_in_flight": sharded["in_flight"],
                "exchange": sharded.get("exchange"), "ids": a.ids, "communicators": sharded.get("communicators", 1),
                "by_exchange": {k: mode_obj(v) for k, v in sharded["modes"].items()},
                "note": f"(= `value`: by_exchange.fixed, the request-list protocol north_star names, whenever it ran — NOT the "
                        f"faster of the two exchanges) the block with the table row-sharded (row i "
                        f"on rank i mod {world}), no host synchronisation in the step; ids {a.ids}.  by_exchange.fixed = the "
                        f"request-list protocol north_star names: HIP routing with per-rank id de-duplication "
                        f"(direct-address mark + scan), fixed-capacity slots, equal-split all_to_all_single of int32 row "
                        f"indices, owner-side gather, equal-split all_to_all_single of {a.nemb * 4}-byte rows (one per "
                        f"DISTINCT id), fused kernel over (rows, perm); overflow flag checked after the timed steps.  "
                        f"by_exchange.whole_shards (only when the batch covers the table: {a.batch * a.nfield} lookups of "
                        f"{a.nfeat} rows per rank): the owners ship their shards as they are — one all_gather_into_tensor "
                        f"of {a.nfeat * a.nemb * 4 / 1e6:.0f} MB per rank and step into a transient buffer, direct row "
                        f"addresses — instead of answering request lists.  ingress_bytes_per_rank_per_step = what one rank "
                        f"receives from its {world - 1} peers; implied_gb_per_s_per_link = that per peer / step time; exchange_ms = "
                        f"the lookup half of a step alone (routing, both exchanges, owner-side gather), kernel_ms = the fused "
                        f"kernel with the table local, exchange_gb_per_s_per_link = the ingress per peer / exchange_ms.  "
                        f"steps_in_flight > 1: consecutive steps alternate between that many streams, so the row exchange "
                        f"of one step runs under the fused kernel of the previous one"}
        if big["done"] or big["err"]:
            if big["err"] or big.get("overflow"):
                line["config4_row_sharded"] = {"error": big["err"] or "a slot of the fixed-capacity exchange overflowed"}
            else:
                line["config4_row_sharded"] = {
                    "value": world * a.batch * big["steps"] / (big["ms"] * 1e-3), "unit": "samples/s",
                    "ms_per_step": big["ms"] / big["steps"], "steps": big["steps"], "exchange": big["exchange"],
                    "steps_in_flight": big["in_flight"], "shard_gb_per_rank": big["shard_gb"],
                    "slot_capacity_factor": a.config4_capacity_factor,
                    "note": f"BASELINE.json configs[3]: armnet_1h nfield={a.nfield} nfeat=100000000 nemb=64 nhid={a.nhid} "
                            f"B={a.batch}/GPU, the 25.6 GB table row-sharded over the {world} ranks (never materialised "
                            f"whole), fused block per step; every sample needs {(world - 1) / world:.0%} of its "
                            f"{a.nfield} x 256-byte rows from other ranks"}
        if c5:
            line["config5_data_parallel"] = ({"error": c5["err"]} if "err" in c5 else {
                "value": world * c5["batch_per_rank"] * a.steps / (c5["ms"] * 1e-3), "unit": "samples/s",
                "ms_per_step": c5["ms"] / a.steps, "global_batch": world * c5["batch_per_rank"],
                "note": f"BASELINE.json configs[4]: armnet (4 heads x 32) + DNN ensemble, nfield=22 nfeat=2000000 nemb=32, "
                        f"B=131072 global = {c5['batch_per_rank']}/GPU, whole forward to logits, tables replicated, no data-path "
                        f"collective; median of 3 windows of {a.steps} steps, MAX over ranks"})
        if a.shard == "rows":
            line["row_sharded_overflow"] = sharded_overflow
            line["row_sharded_path"] = getattr(model._shard, "last_path", None)
            # what the links would carry at 8 ranks for THIS id stream, with and without the hot rows replicated: the slots
            # of the fixed-capacity exchange are sized by the routed (cold) lookups, so the cut is the cold fraction
            from armnet_hip.sharded import fixed_ingress_bytes
            n_lk = a.batch * a.nfield
            hot_n = int(a.hot_rows) or 65536
            cold = int((ids_cpu >= hot_n).sum())
            dd = {"auto": "auto", "on": True, "off": False}[a.dedup]
            u_all = int(torch.unique(ids_cpu).numel())
            u_cold = int(torch.unique(ids_cpu[ids_cpu >= hot_n]).numel())
            w_o, w_h = (fixed_ingress_bytes(n, 8, a.nfeat, a.nemb, dd, n_distinct=u) for n, u in ((n_lk, u_all), (max(cold, 1), max(u_cold, 1))))
            line["hot_rows"] = {"rows": int(a.hot_rows), "priced_for_rows": hot_n, "cold_fraction": cold / n_lk,
                                "replicated_bytes_per_rank": hot_n * a.nemb * 4, "distinct_ids": u_all, "distinct_cold_ids": u_cold,
                                "ingress_bytes_per_rank_per_step_at_8_ranks": {"without": w_o, "with": w_h, "cut": w_o / w_h},
                                "note": f"ids {a.ids}: rows [0, {hot_n}) of the id space (frequency-ordered for skewed click logs) "
                                        "replicated on every rank; such ids take no slot and cross no link "
                                        "(armnet_shard_route_fixed_hot); slots sized by the cold lookups of the first step"}
        if world == 1 and not a.no_cpu_baseline and a.shard == "replicate":
            a_head = argparse.Namespace(**vars(a))
            line["cpu_baseline"] = cpu_baseline(a_head, model, ids_cpu, vals_cpu)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if (a.shard == "both" and not sharded["done"]) or (big["err"] and not big["done"]):
        os._exit(0)                 # a stuck collective: leave without tearing the process group down
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
