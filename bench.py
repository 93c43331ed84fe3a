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
the 8-GPU island model at pop = 1048576).

Two hosts for the N > 1 island model, same JSON line:
  --host process  one process per GPU, elites all-gathered over torch.distributed (nccl = RCCL).  Started by
                  `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or by plain
                  `python bench.py --gpus N`: bench.py then starts those N ranks itself, as fresh child processes
                  and before this process has touched torch or the GPU, relays rank 0's line and exits with
                  their return code.
  --host group    ONE process drives every GPU through the library's island group (sots_group_*: one persistent
                  host thread per island, ncclCommInitAll + ncclAllGather on the islands' streams).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4}] [--host {process,group}]

The K timed steps run with per-kernel timing OFF (the un-instrumented loop); the per-kernel
events behind `roofline` / `kernels` come from a separate, disclosed pass after the timed region.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
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


def cpu_baseline(voice, log2n, target_audio, budget_s=10.0, max_threads=16):
    """The CPU oracle (oracle/sots_oracle.c, a single-threaded port of the reference's
    Evolutionary_Strategy_CPU path) timed on this host on a bounded sample of the workload: the three population
    sizes of BASELINE.md section 3 (P = 64 - BASELINE configs[0] -, 1024, 4096; the workload's voice and N), every
    leg stopped by the clock, so the sample adapts to the host.  `value` is the P = 4096 leg on one core."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # idle OpenMP threads sleep instead of spinning
    from oracle import oracle as O
    kind = {"2op": O.SYNTH_2OP, "3op_series": O.SYNTH_3OP_SERIES, "4op_series": O.SYNTH_4OP_SERIES,
            "triple_parallel": O.SYNTH_TRIPLE_PAR}[voice]

    def timed(ref, budget):
        ref.generation()  # warm
        gens, t0 = 0, time.perf_counter()
        while True:
            ref.generation()
            gens += 1
            dt = time.perf_counter() - t0
            if dt >= budget or gens >= 20000:
                return gens, dt

    sizes, ref = [], None
    for parents, offspring, share in ((16, 48, 0.25), (256, 768, 0.25), (1024, 3072, 0.5)):
        ref = O.OracleES(parents, offspring, kind, log2n, None, VOICES[voice][0], seed=0x5EED0001, recomb_block=min(32, parents))
        ref.set_target_audio(target_audio)
        ref.init_population(0)
        p = parents + offspring
        gens, dt = timed(ref, share * budget_s)
        sizes.append({"pop": p, "value": p * gens / dt, "generations": gens, "seconds": dt})
    big = sizes[-1]
    out = {"value": big["value"], "unit": "candidates/s", "cores": 1, "kind": "port",
           "sample": f"pop={big['pop']} x {big['generations']} generations, {voice} FM, N={1 << log2n}, fp64 built-in FFT "
                     f"(FFTW unavailable), {big['seconds']:.1f} s on 1 core; `sizes`: the same at pop = 64 (BASELINE configs[0]) and 1024",
           "sizes": sizes}
    # the reference's CPU path is single-threaded (the faithful baseline above); additionally
    # the evaluation loop (synthesis + FFT + fitness, independent per individual) on the CPUs this process
    # may run on (its affinity mask), at most --cpu-threads of them
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    cores = max(1, min(allowed, max_threads))
    if cores > 1:
        O.set_threads(cores)
        try:
            gens_mt, dt_mt = timed(ref, 0.5 * budget_s)
        finally:
            O.set_threads(1)
        out["all_cores"] = {"value": big["pop"] * gens_mt / dt_mt, "unit": "candidates/s", "cores": cores,
                            "sample": f"pop={big['pop']}, evaluation loop under OpenMP on {cores} threads "
                                      f"(affinity mask: {allowed} CPUs, --cpu-threads {max_threads}; variation and sort stay "
                                      f"serial), {gens_mt} generations in {dt_mt:.1f} s"}
    return out


def kernels_source_sha16():
    """Identifies the kernel source the committed PMC traffic figures were collected on."""
    path = os.path.join(ROOT, PKG, "csrc", "sots_kernels.hip")
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def load_pmc_traffic(wkey):
    """(per-kernel HBM bytes per launch, source text) from profiles/pmc_traffic.json - only while the entry was
    collected on THIS kernel source (its `kernels_sha16`); a stale entry yields no traffic figure."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        entry = json.load(open(tpath)).get("workloads", {}).get(wkey)
    except (OSError, ValueError):
        entry = None
    if not entry:
        return {}, None
    have, now = entry.get("kernels_sha16"), kernels_source_sha16()
    if have is None or have != now:
        return {}, (f"stale: profiles/pmc_traffic.json[{wkey}] was collected on kernel source {have}, this is {now} "
                    f"(re-run tools/final_profile.sh + tools/pmc_traffic.py)")
    return entry, (f"profiles/pmc_traffic.json[{wkey}] (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, separate passes, "
                   f"kernel source {have})")


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch_command(args, argv):
    """`python bench.py --gpus N` with N > 1, the process host and no rank environment: the command that starts the N
    ranks (None when this process IS a rank, runs the group host, or N = 1)."""
    if args.gpus <= 1 or args.host != "process" or "WORLD_SIZE" in os.environ:
        return None
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(cmd):
    """Runs the ranks as CHILD processes (this process has not imported torch or opened the HIP library, and never
    replaces itself: a process that has touched the GPU must not exec); their stdout is ours, so rank 0's JSON
    line is the one line this command prints."""
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout.splitlines():
        # rank 0's JSON line is the ONE line of this command; anything else the ranks print (gloo's connection
        # banner, for one) goes to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
    return proc.returncode


# ---- the two hosts of the island model -------------------------------------------------------------------
class ProcessHost:
    """One process per GPU: this rank's island + torch.distributed for the elite exchange."""

    name = "process"

    def __init__(self, pkg, args, target, rank, world, device, stream):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.rank, self.world, self.device = torch, dist, rank, world, device
        P = args.parents + args.offspring
        self.es = pkg.HipES(args.parents, args.offspring, pkg.capi.SYNTH_NAMES[args.synth], args.log2n, None,
                            VOICES[args.synth][0], seed=0x5EED0001, workgroup_size=32, device=device.index,
                            gid_base=rank * P, num_generations=args.steps)
        self.es.set_stream(stream.cuda_stream)
        self.es.set_target_audio(target)
        self.island = pkg.island.IslandExchange(rank, world, args.elites, self.es.D, device, overlap=not args.sync_migration)
        self.clock = self.es  # the context whose per-kernel events are reported

    def set_sort_mode(self, mode):
        self.es.set_sort_mode(mode)

    def init_population(self):
        self.island.restart()
        self.es.init_population(0)

    def run(self, n):
        for _ in range(n):
            self.island.generation(self.es)  # the generation + its elite exchange (pack and inject inside the sort kernel)

    def fence(self):
        self.torch.cuda.synchronize(self.device)
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.device)

    def all_ranks_agree(self, flag):
        """every rank leaves a clock-bounded loop after the same chunk"""
        if self.world == 1:
            return bool(flag)
        t = self.torch.tensor([1 if flag else 0], dtype=self.torch.int32, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(int(t.item()))

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return list(seconds)
        t = self.torch.tensor(list(seconds), dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    def finish(self):
        self.island.finish()

    def best_fitness(self):
        return float(np.nanmin(self.es.read_fitness()))

    def readback_ms(self):
        t = time.perf_counter()
        for _ in range(5):
            self.es.read_population()
        return (time.perf_counter() - t) / 5 * 1e3

    def close(self):
        self.es.close()


class GroupHost:
    """One process, every island inside the library (sots_group_*): a persistent host thread per island, RCCL between
    distinct devices, event-ordered copies between islands that share one."""

    name = "group"

    def __init__(self, pkg, args, target, devices):
        self.world = len(devices)
        self.rank = 0
        self.g = pkg.HipGroup(devices, args.elites, args.parents, args.offspring, pkg.capi.SYNTH_NAMES[args.synth], args.log2n,
                              None, VOICES[args.synth][0], seed=0x5EED0001, workgroup_size=32, gid_base=0,
                              migration_interval=1, overlap=not args.sync_migration)
        self.g.set_target_audio(target)
        self.islands = [self.g.island(i) for i in range(self.world)]
        self.es = self.clock = self.islands[0]

    def set_sort_mode(self, mode):
        for es in self.islands:
            es.set_sort_mode(mode)

    def init_population(self):
        self.g.init_population(0)

    def run(self, n):
        self.g.execute_generations(n)  # returns once everything is enqueued (executeAllGenerations)

    def fence(self):
        self.g.synchronize()

    def all_ranks_agree(self, flag):
        return bool(flag)

    def max_over_ranks(self, seconds):
        return list(seconds)

    def finish(self):
        self.g.synchronize()

    def best_fitness(self):
        return float(self.g.best()[1])

    def readback_ms(self):
        t = time.perf_counter()
        for _ in range(5):
            for es in self.islands:
                es.read_population()
        return (time.perf_counter() - t) / 5 * 1e3

    def close(self):
        self.g.close()


def timed_steps(host, steps):
    """EXACTLY `steps` generations between two fences, nothing else in between: no per-kernel events, no host reads."""
    host.fence()
    t0 = time.perf_counter()
    host.run(steps)
    host.fence()
    return time.perf_counter() - t0


def measure(host, args, kernel_table, sort_full_mode):
    """settle -> W warm-up steps -> K timed steps (timing off) -> `sustained` (timing off) -> per-kernel event pass
    (timing on, disclosed) -> the reference's every-generation full sort (timing off)."""
    clock = host.clock

    def harvest():
        """per-kernel device time since the last timing_reset (HIP events on the launch stream)"""
        ks = {}
        for name, stage, alg_bytes in kernel_table:
            ms, cnt = clock.stage_time_ms(stage)
            if cnt:
                ks[name] = {"avg_us": 1e3 * ms / cnt, "launches": int(cnt), "alg_bytes_per_candidate": alg_bytes}
        if "recombine+mutate" not in ks and "synthesise" in ks:
            # large 4-gene populations: the synthesis kernel makes its own individuals
            ks["synthesise"]["alg_bytes_per_candidate"] += 16 * clock.D
            ks["synthesise"]["includes"] = "recombine+mutate"
        return ks

    out = {"settle": None, "sustained": None, "full_sort": None}
    clock.timing_enable(False)
    host.init_population()
    if args.settle_ms > 0:
        # clocks and caches settle (untimed, disclosed in the JSON line); the run proper starts from a fresh population
        t_s, gens = time.perf_counter(), 0
        while True:
            host.run(32)
            gens += 32
            host.fence()
            if host.all_ranks_agree((time.perf_counter() - t_s) * 1e3 >= args.settle_ms):
                break
        out["settle"] = {"ms": (time.perf_counter() - t_s) * 1e3, "generations": gens,
                         "what": "untimed generations before the warm-up steps (device clocks settle), then the population is re-initialised"}
        host.init_population()
    host.run(args.warmup)
    out["seconds"] = timed_steps(host, args.steps)  # the figure of record: per-kernel timing is off
    # `sustained`: the same un-instrumented loop keeps going for >= --sustain seconds: what a long run settles at
    if args.sustain > 0:
        chunk = max(8, args.steps)
        done, t1 = 0, time.perf_counter()
        while True:
            host.run(chunk)
            done += chunk
            host.fence()
            if host.all_ranks_agree(time.perf_counter() - t1 >= args.sustain):
                break
        out["sustained"] = {"steps": done, "seconds": time.perf_counter() - t1}
    # per-kernel events: a separate pass with timing ON for every launch (costs a few per cent, which is why it is
    # not the timed region); `roofline` and `kernels` come from here
    if args.event_steps > 0:
        clock.timing_reset()
        clock.timing_enable(True)
        host.run(args.event_steps)
        host.fence()
        clock.timing_enable(False)
        out["kernels"] = harvest()
    else:
        out["kernels"] = {}
    host.finish()
    out["best"] = host.best_fitness()
    # the reference's schedule beside the headline: every generation sorts all P rows (ocl_program.cl:664-711)
    if args.full_sort_steps > 0 and not args.full_sort:
        host.set_sort_mode(sort_full_mode)
        host.run(max(2, args.warmup))
        dt_f = timed_steps(host, args.full_sort_steps)
        clock.timing_reset()
        clock.timing_enable(True)
        host.run(min(16, args.full_sort_steps))
        host.fence()
        clock.timing_enable(False)
        out["full_sort"] = {"steps": args.full_sort_steps, "seconds": dt_f, "kernels": harvest()}
        host.finish()
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=None, choices=sorted(BASELINE_CONFIGS),
                    help="BASELINE.json configs[] index (default: 2 at --gpus 1, 4 at --gpus N > 1)")
    ap.add_argument("--host", default="process", choices=["process", "group"],
                    help="N > 1: one process per GPU over torch.distributed (bench.py starts the ranks itself when it is not "
                         "already one), or one process driving every GPU through the library's island group (sots_group_*)")
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
                    help="the HEADLINE region sorts all P rows every generation as the reference does (SOTS_SORT_FULL); default: "
                         "the rows recombination reads are placed each generation, the rest of the order when it is read, and "
                         "the full-sort schedule is timed beside it (`full_sort`)")
    ap.add_argument("--full-sort-steps", type=int, default=None,
                    help="timed generations of the `full_sort` record (default: --steps; 0 = skip)")
    ap.add_argument("--event-steps", type=int, default=32,
                    help="generations of the separate per-kernel event pass behind `roofline` / `kernels` (timing ON; >= 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="upper bound of the cpu_baseline's all-cores leg (it never exceeds this process's affinity mask)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + --share-gpu rehearses N > 1 on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="map every rank / island to device 0 (rehearsal only)")
    args = ap.parse_args(argv)
    (args.parents, args.offspring, args.synth, args.log2n, args.elites, scaling, workload, config_id) = resolve_workload(
        args.config, args.gpus, args.parents, args.offspring, args.synth, args.log2n, args.elites, args.shard_of)
    if args.full_sort_steps is None:
        args.full_sort_steps = args.steps
    if 0 < args.event_steps < 8:
        args.event_steps = 8

    # N > 1 from a plain command line: start the ranks FIRST - nothing above has imported torch or opened libsots_hip
    cmd = self_launch_command(args, argv)
    if cmd is not None:
        raise SystemExit(self_launch(cmd))

    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "WARN"  # the image sets VERSION: RCCL then prints a banner on STDOUT, beside the one JSON line
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.host == "group":
        world = args.gpus
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--host group is ONE process for all GPUs: do not start it under torch.distributed.run")
        rank = local_rank = 0
    else:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    if args.host == "group" and not args.share_gpu and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"--host group --gpus {args.gpus}: only {torch.cuda.device_count()} devices (use --share-gpu to rehearse)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if args.host == "process" and world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    pkg = importlib.import_module(PKG)
    P = args.parents + args.offspring
    N = 1 << args.log2n
    target = make_target(pkg, args.synth, args.log2n, local_rank)
    if args.host == "group":
        host = GroupHost(pkg, args, target, [0] * world if args.share_gpu else list(range(world)))
        with_rccl = host.g.uses_rccl
    else:
        stream = torch.cuda.Stream(device=device)
        host = ProcessHost(pkg, args, target, rank, world, device, stream)
        with_rccl = world > 1 and args.backend == "nccl"
    D = host.es.D
    if args.full_sort:
        host.set_sort_mode(pkg.capi.SORT_FULL)

    c = pkg.capi
    KERNELS = (
        # Algorithmic bytes per candidate of the FUSED loop's kernels: what each
        # kernel must move, not what the reference's stage-separated pipeline moves
        # (B_alg = 24N+16 with the window round trip and the spectrum written out).
        ("recombine+mutate", c.STAGE_FUSED_VARIATION, 16 * D),
        ("synthesise", c.STAGE_FUSED_SYNTH, 4 * N + 4 * D),       # parameters in, audio row out
        ("window+FFT+fitness", c.STAGE_FUSED_SPECTRAL, 4 * N + 4),   # audio row in, fitness out
        ("sortPopulation", c.STAGE_SORT, 16 + 8 * (2 * D + 1)))

    if args.host == "process":
        with torch.cuda.stream(stream):
            m = measure(host, args, KERNELS, c.SORT_FULL)
    else:
        m = measure(host, args, KERNELS, c.SORT_FULL)

    secs = [m["seconds"], m["sustained"]["seconds"] if m["sustained"] else 0.0, m["full_sort"]["seconds"] if m["full_sort"] else 0.0]
    dt_max, dt_sus, dt_full = host.max_over_ranks(secs)
    kernels, sustained, full_sort = m["kernels"], m["sustained"], m["full_sort"]
    if sustained:
        sustained.update(seconds=dt_sus, value=P * world * sustained["steps"] / dt_sus, unit="candidates/s",
                         ms_per_step=1e3 * dt_sus / sustained["steps"],
                         what="the timed loop kept running (per-kernel timing off)")
    if full_sort:
        full_sort.update(seconds=dt_full, value=P * world * full_sort["steps"] / dt_full, unit="candidates/s",
                         ms_per_step=1e3 * dt_full / full_sort["steps"],
                         what="SOTS_SORT_FULL: every generation sorts all P rows, the reference's schedule "
                              "(ocl_program.cl:664-711); same population, continued; per-kernel timing off")
    best = m["best"]  # immigrants injected after the last sort sit in the parent tail, so row 0 need not be the best
    # What crossing the boundary with HOST buffers would cost (never part of `value`): a blocking
    # read of the whole population (values, steps, fitness) through the C-ABI, as a caller would do
    # that inspects every generation; the product path itself keeps the population in HBM.
    readback_ms = host.readback_ms()

    if rank == 0:
        value = P * world * args.steps / dt_max
        if not kernels:
            raise SystemExit("bench.py: no per-kernel events were recorded (--event-steps 0): the roofline record needs them")
        dom = max(kernels, key=lambda k: kernels[k]["avg_us"])
        dk = kernels[dom]
        achieved = dk["alg_bytes_per_candidate"] * P / (dk["avg_us"] * 1e-6) / 1e9
        # measured HBM bytes per launch from the committed rocprofv3 PMC passes (same workload, same kernel source)
        wkey = f"P{P}_N{N}_{args.synth}"
        pmc, traffic_source = load_pmc_traffic(wkey)
        traffic = pmc.get(dom)
        per_kernel = {}
        for name, k in kernels.items():
            a = k["alg_bytes_per_candidate"] * P / (k["avg_us"] * 1e-6) / 1e9
            per_kernel[name] = {"achieved_GBs_alg": a, "frac_alg": a / HBM_PEAK_GBS, "traffic": pmc.get(name),
                                "measured_GBs": (pmc[name] / (k["avg_us"] * 1e-6) / 1e9) if pmc.get(name) else None}
        b_alg = 24 * N + 16
        parallelism = (f"island x{world}" if world == 1 else
                       f"island x{world}, host = one process per GPU, elites over torch.distributed ({args.backend})" if args.host == "process" else
                       f"island x{world}, host = one process (sots_group_*: a thread per island), elites over "
                       f"{'RCCL ncclAllGather' if with_rccl else 'event-ordered device copies (islands share a GPU)'}")
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
                       "parallelism": parallelism, "host": host.name,
                       "shared_gpu_rehearsal": bool(args.share_gpu and world > 1),
                       "timed_region": "per-kernel timing off (SURVEY 8(d): un-instrumented loop)",
                       "sortPopulation": "all P rows every generation (reference behaviour)" if args.full_sort else
                                         "the rows the next recombination reads, in order, every generation; the rest of the "
                                         "order when the population is read (DESIGN.md 4.1); `full_sort` = the reference's schedule"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "basis": "bytes this FUSED kernel has to move (DESIGN.md 3.2), not its share of SURVEY 8(d)'s "
                                  "stage-separated B_alg = 24N+16; that basis is given in achieved_b_alg_share / frac_b_alg_share "
                                  "and is an effective bandwidth that exceeds 1 once stages are fused",
                         "achieved_b_alg_share": B_ALG_SHARE[dom](N, D) * P / (dk["avg_us"] * 1e-6) / 1e9,
                         "frac_b_alg_share": B_ALG_SHARE[dom](N, D) * P / (dk["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "traffic_source": traffic_source,
                         "events": f"separate pass of {args.event_steps} generations with a HIP event pair per launch on the launch "
                                   f"stream, after the timed region (island 0 of {world})",
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
                               "bytes": P * (2 * D + 1) * 4 * (world if args.host == "group" else 1),
                               "candidates_per_s_if_read_back_every_generation": P * world / (dt_max / args.steps + readback_ms * 1e-3),
                               "note": "not `value`: sots_execute_generations keeps every buffer in HBM; hosts cross PCIe only for "
                                       "the target (4N bytes in) and the final population"},
            "kernels": kernels,
            "roofline_per_kernel": per_kernel,
            "sustained": sustained,
            "full_sort": full_sort,
            "settle": m["settle"],
            "best_fitness_sse": best,
            "best_fitness_mse": best / (N // 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.synth, args.log2n, target, max_threads=args.cpu_threads)
        # a RECORD, not a measurement of this run: what the reference's own device kernels (kernels/ocl_program.cl compiled as
        # it stands, tests/ref_kernels_time.py) took on an MI355X at configs[2] - the reference publishes no numbers
        rec = os.path.join(ROOT, "profiles", "r04_reference_kernels.json")
        if config_id == 2 and os.path.exists(rec):
            with open(rec) as f:
                r = json.load(f)
            out["reference_kernels_record"] = {"source": "profiles/r04_reference_kernels.json (recorded, not re-run here)",
                                               "reference_generation_us_without_its_fft": r["reference_generation_us_without_fft"],
                                               "kernels_us": r["kernels_us"],
                                               "this_run_generation_us": dt_max / args.steps * 1e6}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    host.close()


if __name__ == "__main__":
    main()
