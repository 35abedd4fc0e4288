"""The ONE stdout line of a bench.py run that could not produce a number, and its exit codes."""

ERROR_RC = {"launch_timeout": 3, "init": 4, "device": 5, "deadline": 6, "ranks": 7}


def error_line(a, what, msg, world=None):
    """the ONE stdout line of a run that could not produce a number (round-5 verdict, next 5c: an RCCL failure must be a
    line with "error" and a non-zero exit code, not a hang): the contract's keys with value = null"""
    return {"metric": "samples/sec, ARM-Net forward (fused embedding + ARM interaction block), Criteo nfield=39 nemb=16 B=65536",
            "value": None, "unit": "samples/s", "n_gpus": int(world if world is not None else a.gpus), "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "config": {"workload": "not measured"},
            "error": f"{what}: {msg}", "error_kind": what, "rc": ERROR_RC[what]}
