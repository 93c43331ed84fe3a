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
