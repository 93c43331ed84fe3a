"""Host overhead of the island group (sots_group_*) on ONE GPU: G islands sharing device 0.

For each population size: the G islands run n generations (a) as plain contexts launched back to back from one
thread without any exchange - the GPU-bound floor of G islands on one device -, (b) as a group exchanging elites
every generation with one call of n, (c) as a group with n calls of one generation (the C++ class's
executeGeneration); each with pack and inject inside the sort kernel (default) and as launches of their own
(`_unfused`).  The GPU executes the islands' kernels one after another either way (every kernel fills the
chip), so (b) - (a) and (c) - (a) are what the island threads, the host barrier and the exchange add per
generation.  Writes one JSON line per shape.

    python tools/group_overhead.py [--islands 2 3] [--pops 8192 32768 65536] [--gens 300]
    python tools/group_overhead.py --config 3 --islands 8 --gens 60     # BASELINE configs[3] at its stated size on one GPU
    python tools/group_overhead.py --config 4 --islands 8 --gens 100    # BASELINE configs[4] likewise (8 x 131072)
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd"
# voice -> (kind name, log2 N, paramMaxs, target parameters); --config 3 / 4 = the per-GPU shards of BASELINE configs[3] / [4]
VOICE = {"2op": ("2op", 10, [3520.0, 8.0, 3520.0, 1.0], [1450 / 3520, 3 / 8, 200 / 3520, 1.0]),
         "4op_series": ("4op_series", 12, [3520.0, 8.0] * 4, [0.3, 0.25, 0.85, 0.19, 0.89, 0.125, 0.5, 0.1])}
CONFIG = {3: ("4op_series", 262144), 4: ("2op", 1048576)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--islands", type=int, nargs="+", default=[1, 2, 3],
                    help="1 = ONE island with the RCCL backend forced (a one-rank communicator): what an island of a multi-GPU "
                         "group pays on its own device for pack, collective launch and inject")
    ap.add_argument("--pops", type=int, nargs="+", default=[8192, 32768, 65536])
    ap.add_argument("--gens", type=int, default=300)
    ap.add_argument("--elites", type=int, default=16)
    ap.add_argument("--config", type=int, default=None, choices=[3, 4],
                    help="BASELINE configs[3] (4-op, N = 4096, 262144 in total) or configs[4] (2-op, N = 1024, 1048576 in total), "
                         "sharded over --shards islands-to-be; --pops is ignored")
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--quick", action="store_true", help="only the default schedule (overlapped, fused, host-gated) and the same-generation one")
    args = ap.parse_args()
    voice = "2op"
    if args.config:
        voice, total = CONFIG[args.config]
        args.pops = [total // args.shards]
    kname, log2n, PMAX, tvals = VOICE[voice]
    pkg = importlib.import_module(PKG)
    c = pkg.capi
    # the target from the HIP synthesiser itself (no oracle on this path)
    kind = c.SYNTH_NAMES[kname]
    es = pkg.HipES(32, 32, kind, log2n, None, PMAX, seed=1)
    v = np.tile(np.asarray(tvals, np.float32), (es.P, 1))
    es.write_population(v, np.full_like(v, 0.1), None)
    es.synthesise()
    target = es.read_audio()[0].copy()
    es.close()

    def best_of(f, reps=3):
        return min(f() for _ in range(reps))

    for P in args.pops:
        parents, offspring = P // 4, P - P // 4
        for G in args.islands:
            n = args.gens
            # (a) plain contexts, one thread, back to back, own streams, no exchange
            ctxs = [pkg.HipES(parents, offspring, kind, log2n, None, PMAX, seed=0x5EED0001, gid_base=r * P) for r in range(G)]
            for e in ctxs:
                e.set_target_audio(target)
                e.init_population(0)
                e.execute_generations(50)
            for e in ctxs:
                e.synchronize()

            def plain():
                t0 = time.perf_counter()
                for _ in range(n):
                    for e in ctxs:
                        e.execute_generations(1)
                for e in ctxs:
                    e.synchronize()
                return (time.perf_counter() - t0) / n * 1e6

            t_plain = best_of(plain)
            for e in ctxs:
                e.close()
            # one island alone: what G islands cost if nothing but the GPU's time were spent (G x this)
            solo = pkg.HipES(parents, offspring, kind, log2n, None, PMAX, seed=0x5EED0001)
            solo.set_target_audio(target)
            solo.init_population(0)
            solo.execute_generations(50)
            solo.synchronize()

            def alone():
                t0 = time.perf_counter()
                solo.execute_generations(n)
                solo.synchronize()
                return (time.perf_counter() - t0) / n * 1e6

            t_solo = best_of(alone)
            solo.close()
            row = {"voice": kname, "N": 1 << log2n, "P_per_island": P, "islands": G, "candidates_in_total": P * G, "generations": n,
                   "single_island_us_per_generation": t_solo, "islands_x_single_us": G * t_solo,
                   "plain_contexts_us_per_generation": t_plain}
            if args.config:
                row["baseline_config"] = args.config
            shapes = ((False, False, False), (True, False, False), (True, False, True), (False, True, False), (True, True, False))
            if args.quick:
                shapes = shapes[:2]
            for overlap, unfused, event_waits in shapes:
                g = pkg.HipGroup([0] * G, args.elites, parents, offspring, kind, log2n, None, PMAX, seed=0x5EED0001,
                                 migration_interval=1, overlap=overlap, unfused=unfused, force_rccl=(G == 1), event_waits=event_waits)
                g.set_target_audio(target)
                g.init_population(0)
                g.execute_generations(50)
                g.synchronize()

                def one_call():
                    t0 = time.perf_counter()
                    g.execute_generations(n)
                    g.synchronize()
                    return (time.perf_counter() - t0) / n * 1e6

                def many_calls():
                    t0 = time.perf_counter()
                    for _ in range(n):
                        g.execute_generations(1)
                    g.synchronize()
                    return (time.perf_counter() - t0) / n * 1e6

                a, b = best_of(one_call), best_of(many_calls)
                key = ("overlapped" if overlap else "same_generation") + ("_unfused" if unfused else "") + ("_event_waits" if event_waits else "")
                row[key] = {"one_call_us": a, "calls_of_one_us": b, "us_per_generation_per_island": a / G,
                            "vs_islands_x_single_pct": 100.0 * (a - G * t_solo) / (G * t_solo),
                            "overhead_one_call_pct": 100.0 * (a - t_plain) / t_plain,
                            "overhead_calls_of_one_pct": 100.0 * (b - t_plain) / t_plain}
                g.close()
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
