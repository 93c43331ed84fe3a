"""bench.py's CPU-side pieces (no GPU): the cpu_baseline leg is bounded in time and in threads,
and the roofline accounting tables cover every fused kernel."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_baseline_is_time_bounded_and_reports_both_legs():
    bench = importlib.import_module("bench")
    from oracle import oracle as O
    pmax, tparams = bench.VOICES["2op"]
    target = O.synth(O.SYNTH_2OP, tparams, [0.0] * 4, pmax, 1024)
    t0 = time.perf_counter()
    cb = bench.cpu_baseline("2op", 10, target, budget_s=0.5)
    assert time.perf_counter() - t0 < 30.0
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "candidates/s" and cb["value"] > 0
    assert "sample" in cb
    if "all_cores" in cb:  # hosts with one CPU have no second leg
        assert 1 < cb["all_cores"]["cores"] <= 16 and cb["all_cores"]["value"] > 0


def test_oracle_results_do_not_depend_on_the_thread_count():
    bench = importlib.import_module("bench")
    from oracle import oracle as O
    pmax, tparams = bench.VOICES["2op"]
    target = O.synth(O.SYNTH_2OP, tparams, [0.0] * 4, pmax, 512)
    pops = []
    for threads in (1, 4):
        es = O.OracleES(32, 96, O.SYNTH_2OP, 9, None, pmax, seed=7, recomb_block=32)
        es.set_target_audio(target)
        es.init_population(0)
        O.set_threads(threads)
        try:
            for _ in range(3):
                es.generation()
        finally:
            O.set_threads(1)
        pops.append(es.read_population())
    for a, b in zip(*pops):
        assert np.array_equal(a, b)


def test_b_alg_shares_add_up_to_the_survey_figure():
    bench = importlib.import_module("bench")
    for n in (1024, 4096):
        evaluation = sum(bench.B_ALG_SHARE[k](n, 4) for k in ("synthesise", "window+FFT+fitness"))
        assert evaluation == 24 * n + 16  # B_alg without the O(D) population traffic


def test_config_presets_name_the_baseline_workloads():
    """bench.py --config: BASELINE.json configs[2..4] as stated; the default is
    config 2 on one GPU and config 4's fixed total sharded over N > 1 GPUs."""
    import json
    bench = importlib.import_module("bench")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    pa, off, synth, log2n, elites, scaling, label, cfg = bench.resolve_workload(None, 1)
    assert (pa + off, synth, log2n, cfg, scaling) == (65536, "2op", 10, 2, "weak") and "configs[2]" in label
    pa, off, synth, log2n, elites, scaling, label, cfg = bench.resolve_workload(None, 8)
    assert (pa + off, synth, log2n, cfg, scaling) == (131072, "2op", 10, 4, "strong") and "configs[4]" in label
    assert (pa + off) * 8 == 1048576
    pa, off, synth, log2n, elites, scaling, label, cfg = bench.resolve_workload(3, 8)
    assert (pa + off, synth, log2n, elites) == (32768, "4op_series", 12, 16) and "configs[3]" in label
    pa, off, *_ = bench.resolve_workload(3, 1)
    assert pa + off == 262144
    pa, off, *_ = bench.resolve_workload(3, 1, shard_of=8)
    assert pa + off == 32768
    # the preset texts are BASELINE.json's own strings up to typography
    for i, c in bench.BASELINE_CONFIGS.items():
        norm = lambda t: t.replace("×", "x").replace("—", "-")
        assert norm(base[i]) == c["text"], (base[i], c["text"])
    # a population must stay a multiple of the recombination block
    for g in (1, 2, 4, 8):
        for cfg in (2, 3, 4):
            pa, off, *_ = bench.resolve_workload(cfg, g)
            assert pa % 32 == 0 and off % 32 == 0


def test_plain_command_with_several_gpus_starts_the_ranks_before_touching_torch_or_hip(tmp_path):
    """`python bench.py --gpus N` (no rank environment): the N ranks are started as CHILD processes by
    torch.distributed.run before this process has imported torch or opened libsots_hip (never
    re-exec or fork a process that has touched the GPU)."""
    import json
    import subprocess
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import json, os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "os.environ.pop('WORLD_SIZE', None); os.environ.pop('RANK', None)\n"
        "import bench\n"
        "seen = {}\n"
        "def fake_run(cmd, env=None, **kw):\n"
        "    seen['cmd'] = cmd\n"
        "    seen['torch_imported'] = 'torch' in sys.modules\n"
        "    seen['hip_library_open'] = any('libsots_hip' in l for l in open('/proc/self/maps'))\n"
        "    seen['ipc'] = (env or {}).get('HSA_ENABLE_IPC_MODE_LEGACY')\n"
        "    class R: returncode = 7; stdout = 'gloo banner\\n{\\\"relayed\\\": 1}\\n'\n"
        "    return R()\n"
        "bench.subprocess.run = fake_run\n"
        "try:\n"
        "    bench.main(['--gpus', '4', '--steps', '3', '--warmup', '1', '--config', '3'])\n"
        "except SystemExit as e:\n"
        "    seen['exit'] = e.code\n"
        "print(json.dumps(seen))\n")
    out = subprocess.run([sys.executable, str(probe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    seen = json.loads(out.stdout.strip().splitlines()[-1])
    # only the ranks' JSON line is relayed on stdout; what else they print (gloo's banner) goes to stderr
    assert '{"relayed": 1}' in out.stdout and "gloo banner" not in out.stdout and "gloo banner" in out.stderr
    assert seen["torch_imported"] is False and seen["hip_library_open"] is False
    assert seen["exit"] == 7  # the ranks' return code is ours
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "3"][-7:]
    assert seen["ipc"] == "0"


def test_launcher_is_not_used_by_ranks_the_group_host_or_one_gpu():
    import argparse
    bench = importlib.import_module("bench")
    ns = lambda **kw: argparse.Namespace(**kw)
    old = os.environ.pop("WORLD_SIZE", None)
    try:
        assert bench.self_launch_command(ns(gpus=1, host="process"), []) is None
        assert bench.self_launch_command(ns(gpus=8, host="group"), []) is None
        assert bench.self_launch_command(ns(gpus=8, host="process"), ["--gpus", "8"]) is not None
        os.environ["WORLD_SIZE"] = "8"
        assert bench.self_launch_command(ns(gpus=8, host="process"), ["--gpus", "8"]) is None
    finally:
        os.environ.pop("WORLD_SIZE", None)
        if old is not None:
            os.environ["WORLD_SIZE"] = old


class _FakeClock:
    D = 4

    def __init__(self, log):
        self.log = log

    def timing_enable(self, on=True):
        self.log.append(("timing", bool(on)))

    def timing_reset(self):
        self.log.append(("reset",))

    def stage_time_ms(self, stage):
        return 1.0, 8


class _FakeHost:
    name = "fake"

    def __init__(self):
        self.log = []
        self.clock = _FakeClock(self.log)

    def init_population(self):
        self.log.append(("init",))

    def set_sort_mode(self, mode):
        self.log.append(("sort_mode", mode))

    def run(self, n):
        self.log.append(("run", n))

    def fence(self):
        self.log.append(("fence",))

    def all_ranks_agree(self, flag):
        return True

    def finish(self):
        self.log.append(("finish",))

    def best_fitness(self):
        return 0.5


def test_timed_region_runs_with_per_kernel_timing_off():
    """the metric is the un-instrumented loop.  The K timed steps sit between two fences with nothing
    else in between, timing was switched off before them, and the event pass comes afterwards."""
    import argparse
    bench = importlib.import_module("bench")
    host = _FakeHost()
    args = argparse.Namespace(settle_ms=0.0, warmup=5, steps=20, sustain=0.0, event_steps=8, full_sort_steps=20, full_sort=False)
    out = bench.measure(host, args, (("synthesise", 10, 4096),), 1)
    log = host.log
    i = log.index(("run", 20))
    assert log[i - 1] == ("fence",) and log[i + 1] == ("fence",)          # exactly K steps between two fences
    assert ("timing", True) not in log[:i + 2]                             # nothing switched timing on before or inside
    assert ("timing", False) in log[:i]
    j = log.index(("timing", True))
    assert j > i and log[j + 1] == ("run", 8)                              # the disclosed event pass, afterwards
    # the reference's full sort beside the headline, also timed with timing off
    k = log.index(("sort_mode", 1))
    r = log.index(("run", 20), k)
    assert log[r - 1] == ("fence",) and log[r + 1] == ("fence",) and ("timing", True) not in log[k:r + 2]
    assert out["kernels"]["synthesise"]["launches"] == 8 and out["full_sort"]["steps"] == 20


def test_pmc_traffic_is_dropped_when_the_kernel_source_changed(tmp_path, monkeypatch):
    import json
    bench = importlib.import_module("bench")
    (tmp_path / "profiles").mkdir()
    now = bench.kernels_source_sha16()
    assert now and len(now) == 16
    table = {"workloads": {"fresh": {"kernels_sha16": now, "synthesise": 123}, "old": {"kernels_sha16": "0" * 16, "synthesise": 456},
                           "untagged": {"synthesise": 789}}}
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps(table))
    sha = bench.kernels_source_sha16
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernels_source_sha16", sha.__wrapped__ if hasattr(sha, "__wrapped__") else (lambda: now))
    entry, src = bench.load_pmc_traffic("fresh")
    assert entry["synthesise"] == 123 and "stale" not in src
    for key in ("old", "untagged"):
        entry, src = bench.load_pmc_traffic(key)
        assert entry == {} and src.startswith("stale")
    assert bench.load_pmc_traffic("missing") == ({}, None)
