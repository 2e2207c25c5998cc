"""bench.py end to end on the GPU at a small size: the driver's contract (ONE JSON line on stdout, the keys it reads,
the roofline and cpu_baseline objects) and the forced-sharded line through RCCL at world 1."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _run(args, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must hold exactly one line, got {len(lines)}: {p.stdout[:500]}"
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    d = _run(["--gpus", "1", "--steps", "12", "--warmup", "4", "--freq", "40", "--cpu-steps", "1"])
    for k in CONTRACT_KEYS:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 4 and d["unit"] == "steps/s"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] - 1e3) < 1e-6 * 1e3
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert d["steps_accepted"] >= 1 and d["line_search_trials"] >= d["steps_accepted"]
    assert d["accepted_steps_per_s"] > 0 and d["evaluations_per_s"] > 0


def test_forced_sharded_line_goes_through_rccl_at_world_one():
    d = _run(["--steps", "10", "--warmup", "4", "--freq", "40", "--cpu-steps", "0", "--no-roofline"],
             {"MS_BENCH_FORCE_SHARDED": "1"})
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["rccl_ranks"] == 1 and d["exchanges"] >= 10 and d["exchanges_per_step"] >= 1.0
    assert "library (ms_shard_step" in d["config"]["parallelism"]
