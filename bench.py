#!/usr/bin/env python3
"""bench.py -- candidates evaluated per second of the evolutionary FM-matching loop.

A "step" is one generation (recombine+mutate | synthesise+window | FFT+fitness | sort |
rotate, plus the elite all-gather when N > 1) over one island's whole population.

Workloads (`--config`, BASELINE.json configs[] index; synthetic targets, fp32):
  2  pop = 65536 (16384 parents + 49152 offspring) PER GPU, 2-operator FM, 1024-pt FFT:
     the single-GPU configuration the metric is quoted on; N > 1 = one such island per GPU (weak)
  3  pop = 262144 IN TOTAL, 4-operator FM, 4096-pt FFT, 16 elites per island: 262144/N per GPU (strong)
  4  pop = 1048576 IN TOTAL, 2-operator FM, 1024-pt FFT: 1048576/N per GPU (strong)
Default: config 2 at --gpus 1, config 4 at --gpus N > 1 (north_star: one MI355X at pop = 65536,
the 8-GPU island model at pop = 1048576).  One process per GPU, elites all-gathered over RCCL
every generation.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4}]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd"

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
# shares of the stage-separated B_alg = 24N+16 bytes per candidate that fall to each fused kernel
B_ALG_SHARE = {
    "recombine+mutate": lambda n, d: 16 * d,
    "synthesise": lambda n, d: 4 * n,                      # synth write
    "window+FFT+fitness": lambda n, d: 20 * n + 16,        # window r+w 8N, FFT read 4N + write 8(N/2+1), fitness read 8(N/2+1)
    "sortPopulation": lambda n, d: 16 + 8 * (2 * d + 1),
}
# voice -> (synth kind name, paramMaxs, target parameters in the unit cube)
VOICES = {
    "2op": ([3520.0, 8.0, 3520.0, 1.0], [1450.0 / 3520.0, 3.0 / 8.0, 200.0 / 3520.0, 1.0]),
    "3op_series": ([3520.0, 8.0, 3520.0, 8.0, 3520.0, 8.0],
                   [3078 / 3520.0, 2.0 / 8.0, 3015 / 3520.0, 1.5 / 8.0, 3141 / 3520.0, 1.0 / 8.0]),
    "4op_series": ([3520.0, 8.0, 3520.0, 8.0, 3520.0, 8.0, 3520.0, 8.0], [0.3, 0.25, 0.85, 0.19, 0.89, 0.125, 0.5, 0.1]),
    "triple_parallel": ([3520.0, 8.0, 3520.0, 1.0], [0.41, 0.375, 0.057, 1.0, 0.2, 0.5, 0.11, 0.7, 0.6, 0.1, 0.3, 0.4]),
}

BASELINE_CONFIGS = {
    2: dict(total=None, per_gpu=65536, synth="2op", log2n=10, elites=16, scaling="weak",
            text="pop=65536, 2-op FM, 1024-pt FFT, 1xMI355X - HBM-bound run, rocprof GB/s vs roofline"),
    3: dict(total=262144, per_gpu=None, synth="4op_series", log2n=12, elites=16, scaling="strong",
            text="pop=262144, 4-op FM, 4096-pt FFT, island-sharded across 8xMI355X with RCCL elite allgather over xGMI"),
    4: dict(total=1048576, per_gpu=None, synth="2op", log2n=10, elites=16, scaling="strong",
            text="pop=1,048,576, 2-op FM, 1024-pt FFT, 8xMI355X island model, per-GPU counter-based PRNG, 10k generations"),
}


def resolve_workload(config, gpus, parents=None, offspring=None, synth=None, log2n=None, elites=None, shard_of=None):
    """(parents, offspring, synth, log2n, elites, scaling, label) of one island for --config / --gpus;
    explicit --parents/--offspring/--synth/--log2n override the preset (the label then says so)."""
    if config is None:
        config = 2 if gpus == 1 else 4
    c = BASELINE_CONFIGS[config]
    shards = shard_of or gpus  # --shard-of G: this run's islands are G-GPU shards (profiling one shard on one GPU)
    per_gpu = c["per_gpu"] if c["total"] is None else c["total"] // shards
    if c["total"] is not None and c["total"] % (shards * 128) != 0:
        raise SystemExit(f"--config {config}: {c['total']} candidates do not shard over {gpus} GPUs in blocks of 128")
    custom = any(x is not None for x in (parents, offspring, synth, log2n))
    if parents is None and offspring is None:
        parents, offspring = per_gpu // 4, per_gpu - per_gpu // 4     # the 1 : 3 split of configs[2]
    elif parents is None or offspring is None:
        raise SystemExit("--parents and --offspring go together")
    synth = synth or c["synth"]
    log2n = log2n or c["log2n"]
    elites = c["elites"] if elites is None else elites
    p = parents + offspring
    label = (f"custom: " if custom else f"BASELINE configs[{config}]: {c['text']} -> ")
    if shard_of and shard_of != gpus and c["total"] is not None:
        label += f"[the per-GPU shard of a {shard_of}-GPU run] "
    label += (f"pop={p} ({parents}+{offspring}) per GPU x {gpus} island{'s' if gpus > 1 else ''} = {p * gpus}, "
              f"{synth} FM, {1 << log2n}-sample / {1 << log2n}-pt FFT, fp32")
    return parents, offspring, synth, log2n, elites, (c["scaling"] if not custom else "weak"), label, config


def make_target(pkg, voice, log2n, device):
    """Target audio from the HIP synthesiser itself (the oracle is not used on the product path)."""
    pmax, tparams = VOICES[voice]
    es = pkg.HipES(32, 32, pkg.capi.SYNTH_NAMES[voice], log2n, None, pmax, seed=1, workgroup_size=32, device=device)
    v = np.tile(np.asarray(tparams, np.float32), (es.P, 1))
    es.write_population(v, np.full_like(v, 0.1), None)
    es.synthesise()
    audio = es.read_audio()[0].copy()
    es.close()
    return audio


def cpu_baseline(voice, log2n, target_audio, budget_s=10.0):
    """The CPU oracle (oracle/sots_oracle.c, a single-threaded port of the reference's
    Evolutionary_Strategy_CPU path) timed on this host on a bounded sample of the workload.
    Both legs stop on the clock, so the sample size adapts to the host."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # idle OpenMP threads sleep instead of spinning
    from oracle import oracle as O
    parents, offspring = 512, 1536
    kind = {"2op": O.SYNTH_2OP, "3op_series": O.SYNTH_3OP_SERIES, "4op_series": O.SYNTH_4OP_SERIES,
            "triple_parallel": O.SYNTH_TRIPLE_PAR}[voice]
    ref = O.OracleES(parents, offspring, kind, log2n, None, VOICES[voice][0], seed=0x5EED0001, recomb_block=32)
    ref.set_target_audio(target_audio)
    ref.init_population(0)
    p = parents + offspring

    def timed(budget):
        ref.generation()  # warm
        gens, t0 = 0, time.perf_counter()
        while True:
            ref.generation()
            gens += 1
            dt = time.perf_counter() - t0
            if dt >= budget or gens >= 5000:
                return gens, dt

    gens, dt = timed(budget_s)
    out = {"value": p * gens / dt, "unit": "candidates/s", "cores": 1, "kind": "port",
           "sample": f"pop={p} x {gens} generations, {voice} FM, N={1 << log2n}, fp64 built-in FFT "
                     f"(FFTW unavailable), {dt:.1f} s on 1 core"}
    # the reference's CPU path is single-threaded (the faithful baseline above); additionally
    # the evaluation loop (synthesis + FFT + fitness, independent per individual) on the box's CPU share
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, 16)  # a one-GPU box owns 16 CPUs whatever the affinity mask says
    if cores > 1:
        O.set_threads(cores)
        try:
            gens_mt, dt_mt = timed(0.5 * budget_s)
        finally:
            O.set_threads(1)
        out["all_cores"] = {"value": p * gens_mt / dt_mt, "unit": "candidates/s", "cores": cores,
                            "sample": f"same workload, evaluation loop under OpenMP on {cores} threads "
                                      f"(variation and sort stay serial), {gens_mt} generations in {dt_mt:.1f} s"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=None, choices=sorted(BASELINE_CONFIGS),
                    help="BASELINE.json configs[] index (default: 2 at --gpus 1, 4 at --gpus N > 1)")
    ap.add_argument("--parents", type=int, default=None)
    ap.add_argument("--offspring", type=int, default=None)
    ap.add_argument("--log2n", type=int, default=None)
    ap.add_argument("--elites", type=int, default=None)
    ap.add_argument("--synth", default=None, choices=sorted(VOICES))
    ap.add_argument("--shard-of", type=int, default=None,
                    help="size the islands as the per-GPU shards of a G-GPU run of configs 3/4 (e.g. --config 3 --shard-of 8 "
                         "on one GPU = one 32768-candidate island)")
    ap.add_argument("--sustain", type=float, default=1.0,
                    help="seconds the loop keeps running after the headline region for the `sustained` record (0 = skip)")
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="untimed generations on the same context BEFORE the W warm-up steps, for this long, after which the "
                         "population is re-initialised: an MI355X that was idle runs its first ~50 ms about 20 %% below its settled "
                         "clocks, and the driver's 5 warm-up + 20 timed steps (3.4 ms) would all fall in there.  The metric "
                         "(SURVEY 8(d)) is the steady-state rate.  Reported as `settle`; 0 switches it off")
    ap.add_argument("--sync-migration", action="store_true",
                    help="inject elites inside the generation that gathered them (default: the all-gather "
                         "overlaps the next generation and its rows arrive one generation later)")
    ap.add_argument("--full-sort", action="store_true",
                    help="sort all P rows every generation as the reference does (SOTS_SORT_FULL); default: the rows "
                         "recombination reads are placed each generation, the rest of the order when it is read")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + --share-gpu rehearses N > 1 on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="map every rank to cuda:0 (rehearsal only)")
    ap.add_argument("--timing-every", type=int, default=None,
                    help="record per-kernel HIP events on every k-th timed step (default max(1, steps // 8): "
                         "at least 8 launches behind every per-kernel figure; 0 = never)")
    args = ap.parse_args()
    (args.parents, args.offspring, args.synth, args.log2n, args.elites, scaling, workload, config_id) = resolve_workload(
        args.config, args.gpus, args.parents, args.offspring, args.synth, args.log2n, args.elites, args.shard_of)
    if args.timing_every is None:
        args.timing_every = max(1, args.steps // 8)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    pkg = importlib.import_module(PKG)
    P = args.parents + args.offspring
    N = 1 << args.log2n
    target = make_target(pkg, args.synth, args.log2n, local_rank)
    es = pkg.HipES(args.parents, args.offspring, pkg.capi.SYNTH_NAMES[args.synth], args.log2n, None, VOICES[args.synth][0],
                   seed=0x5EED0001, workgroup_size=32, device=local_rank, gid_base=rank * P,
                   num_generations=args.steps)
    stream = torch.cuda.Stream(device=device)
    es.set_stream(stream.cuda_stream)
    es.set_target_audio(target)
    if args.full_sort:
        es.set_sort_mode(pkg.capi.SORT_FULL)
    island = pkg.island.IslandExchange(rank, world, args.elites, es.D, device, overlap=not args.sync_migration)

    def step():
        es.execute_generations(1)
        island.migrate_device(es)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    c = pkg.capi
    KERNELS = (
        # Algorithmic bytes per candidate of the FUSED loop's kernels: what each
        # kernel must move, not what the reference's stage-separated pipeline moves
        # (B_alg = 24N+16 with the window round trip and the spectrum written out).
        ("recombine+mutate", c.STAGE_FUSED_VARIATION, 16 * es.D),
        ("synthesise", c.STAGE_FUSED_SYNTH, 4 * N + 4 * es.D),       # parameters in, audio row out
        ("window+FFT+fitness", c.STAGE_FUSED_SPECTRAL, 4 * N + 4),   # audio row in, fitness out
        ("sortPopulation", c.STAGE_SORT, 16 + 8 * (2 * es.D + 1)))

    def harvest():
        """per-kernel device time since the last timing_reset (HIP events on the launch stream)"""
        ks = {}
        for name, stage, alg_bytes in KERNELS:
            ms, cnt = es.stage_time_ms(stage)
            if cnt:
                ks[name] = {"avg_us": 1e3 * ms / cnt, "launches": int(cnt), "alg_bytes_per_candidate": alg_bytes}
        if "recombine+mutate" not in ks and "synthesise" in ks:
            # large 4-gene populations: the synthesis kernel makes its own individuals
            ks["synthesise"]["alg_bytes_per_candidate"] += 16 * es.D
            ks["synthesise"]["includes"] = "recombine+mutate"
        return ks

    settle = None
    with torch.cuda.stream(stream):
        es.init_population(0)
        if args.settle_ms > 0:
            # clocks and caches settle (untimed, disclosed in the JSON line); the run proper starts from a fresh population
            t_s, gens = time.perf_counter(), 0
            while (time.perf_counter() - t_s) * 1e3 < args.settle_ms:
                es.execute_generations(32)
                gens += 32
                torch.cuda.synchronize(device)
            settle = {"ms": (time.perf_counter() - t_s) * 1e3, "generations": gens,
                      "what": "untimed generations before the warm-up steps (device clocks settle), then the population is re-initialised"}
            es.init_population(0)
        for _ in range(args.warmup):
            step()
        es.timing_reset()
        fence()
        t0 = time.perf_counter()
        for k in range(args.steps):
            es.timing_enable(args.timing_every > 0 and k % args.timing_every == 0)
            step()
        fence()
        dt = time.perf_counter() - t0
        es.timing_enable(False)
        kernels = harvest()
        # `sustained`: the same loop keeps going for >= --sustain seconds (headline fields above are
        # not touched by it): what a long run settles at once clocks and caches have
        sustained = None
        if args.sustain > 0:
            es.timing_reset()
            chunk = max(8, args.steps)
            every = max(1, chunk // 8)
            done, t1 = 0, time.perf_counter()
            stop = torch.zeros(1, dtype=torch.int32, device=device)
            while True:
                for k in range(chunk):
                    es.timing_enable(k % every == 0)
                    step()
                done += chunk
                torch.cuda.synchronize(device)
                stop[0] = 1 if time.perf_counter() - t1 >= args.sustain else 0
                if world > 1:
                    dist.all_reduce(stop, op=dist.ReduceOp.MAX)  # every rank leaves after the same chunk
                if int(stop.item()):
                    break
            fence()
            dt_s = time.perf_counter() - t1
            es.timing_enable(False)
            sustained = {"steps": done, "seconds": dt_s, "kernels": harvest()}
        island.finish()
    es.timing_enable(False)

    dt_t = torch.tensor([dt, sustained["seconds"] if sustained else 0.0], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt_max = float(dt_t[0].item())
    if sustained:
        sustained["seconds"] = float(dt_t[1].item())
        sustained["value"] = P * world * sustained["steps"] / sustained["seconds"]
        sustained["unit"] = "candidates/s"
        sustained["ms_per_step"] = 1e3 * sustained["seconds"] / sustained["steps"]

    fitness = es.read_fitness()
    best = float(np.nanmin(fitness))  # immigrants injected after the last sort sit in the parent tail, so row 0 need not be the best
    # What crossing the boundary with HOST buffers would cost (never part of `value`): a blocking
    # read of the whole population (values, steps, fitness) through the C-ABI, as a caller would do
    # that inspects every generation; the product path itself keeps the population in HBM.
    t_rb = time.perf_counter()
    for _ in range(5):
        es.read_population()
    readback_ms = (time.perf_counter() - t_rb) / 5 * 1e3

    if rank == 0:
        value = P * world * args.steps / dt_max
        if not kernels:
            raise SystemExit("bench.py: no per-kernel events were recorded (--timing-every 0): the roofline record needs them")
        dom = max(kernels, key=lambda k: kernels[k]["avg_us"])
        dk = kernels[dom]
        achieved = dk["alg_bytes_per_candidate"] * P / (dk["avg_us"] * 1e-6) / 1e9
        # measured HBM bytes per launch from the committed rocprofv3 PMC passes (same workload);
        # only valid for the configuration they were collected on
        pmc = {}
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        wkey = f"P{P}_N{N}_{args.synth}"
        if os.path.exists(tpath):
            try:
                pmc = json.load(open(tpath)).get("workloads", {}).get(wkey, {})
            except Exception:
                pmc = {}
        traffic = pmc.get(dom)
        per_kernel = {}
        for name, k in kernels.items():
            a = k["alg_bytes_per_candidate"] * P / (k["avg_us"] * 1e-6) / 1e9
            per_kernel[name] = {"achieved_GBs_alg": a, "frac_alg": a / HBM_PEAK_GBS, "traffic": pmc.get(name),
                                "measured_GBs": (pmc[name] / (k["avg_us"] * 1e-6) / 1e9) if name in pmc else None}
        b_alg = 24 * N + 16
        out = {
            "metric": "candidates evaluated/sec (pop x gens / s)",
            "value": value,
            "unit": "candidates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "baseline_config": config_id,
                       "islands": world, "elites_per_island": args.elites if world > 1 else 0,
                       "migration_interval": 1,
                       "migration": "none" if world == 1 else ("same generation" if args.sync_migration else "overlapped, arrives one generation later"),
                       "parallelism": f"island x{world}",
                       "sortPopulation": "all P rows every generation (reference behaviour)" if args.full_sort else
                                         "the rows the next recombination reads, in order, every generation; the rest of the "
                                         "order when the population is read (DESIGN.md 4.1)"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "basis": "bytes this FUSED kernel has to move (DESIGN.md 3.2), not its share of SURVEY 8(d)'s "
                                  "stage-separated B_alg = 24N+16; that basis is given in achieved_b_alg_share / frac_b_alg_share "
                                  "and is an effective bandwidth that exceeds 1 once stages are fused",
                         "achieved_b_alg_share": B_ALG_SHARE[dom](N, es.D) * P / (dk["avg_us"] * 1e-6) / 1e9,
                         "frac_b_alg_share": B_ALG_SHARE[dom](N, es.D) * P / (dk["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "traffic_source": f"profiles/pmc_traffic.json[{wkey}] (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, separate passes)" if traffic else None,
                         "launches": dk["launches"],
                         "avg_kernel_us": dk["avg_us"],
                         "alg_bytes_per_launch": dk["alg_bytes_per_candidate"] * P},
            "pipeline_effective": {"b_alg_unfused_bytes_per_candidate": b_alg,
                                   "effective_GBs_per_gpu": value / world * b_alg / 1e9,
                                   "frac_of_hbm_peak": value / world * b_alg / 1e9 / HBM_PEAK_GBS,
                                   "note": "SURVEY 8(d) prices the reference's stage-separated pipeline at B_alg = 24N+16 bytes per "
                                           "candidate; the fused loop applies the window on the FFT kernel's load and never "
                                           "materialises the spectrum, so it moves about 8N. This entry is the whole-loop rate "
                                           "priced at the unfused B_alg (an effective figure that can exceed 1); `roofline` prices "
                                           "the dominant kernel at the bytes that kernel itself has to move"},
            "pcie_inclusive": {"population_readback_ms": readback_ms,
                               "bytes": P * (2 * es.D + 1) * 4,
                               "candidates_per_s_if_read_back_every_generation": P * world / (dt_max / args.steps + readback_ms * 1e-3),
                               "note": "not `value`: sots_execute_generations keeps every buffer in HBM; hosts cross PCIe only for "
                                       "the target (4N bytes in) and the final population"},
            "kernels": kernels,
            "roofline_per_kernel": per_kernel,
            "sustained": sustained,
            "settle": settle,
            "best_fitness_sse": best,
            "best_fitness_mse": best / (N // 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.synth, args.log2n, target)
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    es.close()


if __name__ == "__main__":
    main()
