"""The CPU baselines bench.py reports beside the GPU number (rank 0, N = 1): the C port of the oracle and the product
module's own host-tensor branch (the reference's ATen op chain)."""
import time
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402


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
    out = {"value": done / t_used, "unit": "samples/s", "cores": threads, "kind": "port",
           "sample": f"{passes} pass(es) of the first {n} samples of the same batch through "
                     f"oracle_arm_block (50-step bisection, OpenMP, {threads} threads)",
           "one_thread": {"value": n1 / t1, "unit": "samples/s", "sample": f"{n1} samples, 1 thread"}}
    # the reference's own ATen op chain on the same host threads, beside the C port (ii) above
    try:
        out["aten_chain"] = cpu_baseline_aten(a, model, ids_cpu, vals_cpu, threads)
    except Exception as e:  # noqa: BLE001
        out["aten_chain"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def host_twin(a, model):
    """the same module, never moved to the GPU: constructor arguments of `build_model`, the device model's state_dict"""
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    nfeat = sd["embedding.embedding.weight"].shape[0]
    ens = bool(getattr(a, "ensemble", False))
    if a.nhead == 1:
        from models.armnet_1h import ARMNetModel
        m = ARMNetModel(a.nfield, nfeat, a.nemb, a.alpha, a.nhid, a.nemb, 2, 256, 0.0, ens, 2, 256)
    else:
        from models.armnet import ARMNetModel
        m = ARMNetModel(a.nfield, nfeat, a.nemb, a.nhead, a.alpha, a.nhid, 2, 256, 0.0, ens, 2, 256)
    m.load_state_dict(sd, strict=True)
    m.allow_host = True                         # the host branch is the point here (no "forgotten .cuda()" warning)
    return m.eval()


def aten_chain_block(host_model, ids, vals):
    """SURVEY §8d CPU baseline (i): the reference's ATen OP CHAIN for rows a2..a9 on CPU tensors — what `train.py:117`
    executes when the model sits on the host: the product module itself, never moved to the GPU, called with host
    tensors (`armnet_hip/host_ops.py`: in-place clamp, embedding x value, key projection, gates, 50-step bisection entmax
    with tensor-tensor `pow` (softmax when alpha == 1), value weighting, einsum + exp; then the eval-mode BatchNorm1d
    module).  Held to the golden vectors by tests/test_host_tensors.py and tests/test_bench_contract.py."""
    return host_model.arm_block(ids, vals)


def cpu_baseline_aten(a, model, ids_cpu, vals_cpu, threads):
    """the op chain above on `threads` host threads, on the first n samples of the same batch (about a.cpu_seconds)"""
    host = host_twin(a, model)
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        n = min(a.batch, 2048)
        with torch.no_grad():
            t0 = time.perf_counter()
            aten_chain_block(host, ids_cpu[:n], vals_cpu[:n].clone())
            t_probe = time.perf_counter() - t0
            # size the sample so that it takes about cpu_seconds / 2 (the softmax branch is ~15x faster than bisection)
            n = int(max(n, min(a.batch, n * (a.cpu_seconds / 2) / max(t_probe, 1e-6)))) // 1024 * 1024 or n
            t0 = time.perf_counter()
            aten_chain_block(host, ids_cpu[:n], vals_cpu[:n].clone())
            t = time.perf_counter() - t0
    finally:
        torch.set_num_threads(old)
    return {"value": n / t, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"1 pass of the first {n} samples of the same batch through the reference's ATen op chain — the product "
                      f"module's own host-tensor branch (armnet_hip/host_ops.py: embedding, Linear/einsum, "
                      f"{'softmax' if a.alpha == 1.0 else '50-step bisection entmax with tensor pow'}, "
                      f"einsum, exp, BatchNorm1d) on CPU tensors, torch.set_num_threads({threads})"}
