"""bench.py's output contract: exactly ONE JSON line on stdout with the driver's keys, the roofline and cpu_baseline
objects (N = 1) and the row-sharded headline beside the replicated number (N > 1; two ranks sharing the one GPU, exchanges
staged through gloo — the RCCL path needs more GPUs than this box has)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _one_json_line(out):
    lines = [ln for ln in out.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {lines[:3]}"
    return json.loads(lines[0])


def test_defaults_are_the_contract_defaults():
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    sys.argv = ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = old
    assert a.gpus == 1 and a.steps >= 1 and a.warmup >= 1
    assert (a.nfield, a.nfeat, a.nemb, a.nhid, a.nhead, a.batch) == (39, 1_000_000, 16, 32, 1, 65536)   # configs[1]
    assert len(bench.kernel_src_sha()) == 16


@pytest.mark.gpu
def test_single_gpu_line_has_the_contract_keys_and_both_objects():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2",
                        "--cpu-seconds", "1"], cwd=ROOT, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = _one_json_line(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "regimes", "full_forward"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "samples/s" and d["value"] > 1e6 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    assert set(r["fractions"]) == {"hbm", "mfma_fp32", "access_pattern_ceiling", "hbm_traffic"}
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "samples/s" and c["sample"]
    assert set(d["regimes"]) == {"fresh", "stress"}
    # steady state is `value`; the cold-start run of the same steps and the two-batches-in-flight variant ride along
    assert d["config"]["clock_settle_ms"] > 0 and d["cold_start"]["value"] > 0
    oa = d["other_alphas"]                                             # alpha = 1.7 and 1.5 beside the headline
    assert set(oa) == {"1.7", "1.5", "note"} and oa["1.7"]["fresh"]["value"] > 1e6 and oa["1.5"]["stress"]["value"] > 1e6
    fl = d["batches_in_flight"]
    assert fl["n"] == 2 and fl["value"] > 0 and fl["full_forward_samples_per_s"] > 0


def _check_two_rank_line(d):
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    rs, rep = d["row_sharded"], d["replicated"]
    assert "error" not in rs, rs
    assert d["value"] == rs["value"] and rep["value"] > 0
    # three numbers, always: replicated, the request-list all-to-all protocol ("fixed") and the whole-shard exchange
    by = rs["by_exchange"]
    assert set(by) == {"whole_shards", "fixed"} and rs["exchange"] in by and rs["ids"] == "uniform"
    assert rs["value"] == max(v["value"] for v in by.values())
    for v in by.values():
        assert set(v["samples_per_s_by_steps_in_flight"]) == {"1", "2"}
        assert v["ingress_bytes_per_rank_per_step"] > 0 and v["implied_gb_per_s_per_link"] > 0
    assert by["whole_shards"]["ingress_bytes_per_rank_per_step"] == 500_000 * 64     # the other rank's shard
    assert "row-sharded" in d["config"]["parallelism"]
    assert "cpu_baseline" not in d                                     # rank 0 at N = 1 only


@pytest.mark.gpu
def test_two_rank_line_reports_the_row_sharded_step_as_value():
    env = dict(os.environ, ARMNET_BENCH_BACKEND="gloo", ARMNET_BENCH_DEVICE="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--regime", "fresh", "--no-config4"],
                       cwd=ROOT, env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    _check_two_rank_line(_one_json_line(p.stdout))


@pytest.mark.gpu
def test_plain_python_launch_with_gpus_2_starts_its_own_ranks():
    """round-2 verdict, missing 2: `python bench.py --gpus N` without torch.distributed.run used to die on an assert;
    it now launches the N ranks itself and rank 0 prints the one line"""
    env = dict(os.environ, ARMNET_BENCH_BACKEND="gloo", ARMNET_BENCH_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--regime", "fresh", "--no-config4"], cwd=ROOT, env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    _check_two_rank_line(_one_json_line(p.stdout))
