"""bench.py as the driver calls it, on one MI355X: the plain N = 1 call, and the N = 2 island model through both
hosts started from the plain command line (`--share-gpu`: both islands on device 0; gloo for the process host,
since RCCL refuses two ranks on one GPU).  Checks the contract of the ONE JSON line, not the numbers.

Run with: python -m pytest tests -m gpu
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "12", "--warmup", "3", "--sustain", "0.05", "--settle-ms", "5", "--event-steps", "8"]


def run_bench(args, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), f"stdout must be ONE JSON line, got {len(lines)}: {out.stdout[:500]}"
    return json.loads(lines[0])


def check_contract(d, n_gpus):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["steps"] == 12 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["launches"] >= 8
    assert "per-kernel timing off" in d["config"]["timed_region"]
    # candidates/s = P x islands x steps / time
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / (d["config"]["islands"] * population(d)) - 1.0) < 1e-6


def population(d):
    w = d["config"]["workload"]
    return int(w.split("pop=")[-1].split(" ")[0])


def test_single_gpu_line_has_roofline_cpu_baseline_and_the_full_sort_leg():
    d = run_bench(SMALL + ["--parents", "4096", "--offspring", "12288"])
    check_contract(d, 1)
    assert d["scaling"] == "weak" and d["config"]["islands"] == 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    fs = d["full_sort"]
    assert fs["steps"] == 12 and fs["value"] > 0 and "sortPopulation" in fs["kernels"]
    assert d["sustained"]["value"] > 0 and d["settle"]["generations"] >= 32
    # the event pass saw every kernel of the fused loop
    assert {"synthesise", "window+FFT+fitness", "sortPopulation"} <= set(d["kernels"])


@pytest.mark.parametrize("host,extra", [("group", []), ("group", ["--sync-migration"]), ("process", ["--backend", "gloo"])])
def test_two_islands_from_the_plain_command_line(host, extra):
    """`python bench.py --gpus 2 ...` without any rank environment: the process host starts its two ranks itself, the
    group host drives both islands from one process; either way one line with n_gpus = 2."""
    d = run_bench(SMALL + ["--gpus", "2", "--share-gpu", "--host", host, "--parents", "2048", "--offspring", "6144",
                           "--no-cpu-baseline"] + extra)
    check_contract(d, 2)
    assert d["config"]["islands"] == 2 and d["config"]["host"] == host and d["config"]["elites_per_island"] == 16
    assert d["config"]["shared_gpu_rehearsal"] is True
    assert ("sots_group" in d["config"]["parallelism"]) == (host == "group")
    assert d["config"]["migration"].startswith("same generation" if "--sync-migration" in extra else "overlapped")
    assert "cpu_baseline" not in d
    assert d["best_fitness_sse"] >= 0
