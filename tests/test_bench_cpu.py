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
