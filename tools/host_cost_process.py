"""Host-side cost per generation of the one-process-per-GPU island exchange (island.IslandExchange.generation), measured
on ONE GPU with a one-rank NCCL process group: the ctypes calls, the event records, the side-stream all-gather issued
through torch.distributed - everything except the other ranks.  If this exceeds the GPU time of a generation the process
host is host-bound.
    python tools/host_cost_process.py [--parents 16384 --offspring 49152]"""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd"

ap = argparse.ArgumentParser()
ap.add_argument("--parents", type=int, default=16384)
ap.add_argument("--offspring", type=int, default=49152)
ap.add_argument("--gens", type=int, default=400)
args = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
    os.environ["NCCL_DEBUG"] = "WARN"
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
pkg = importlib.import_module(PKG)
import bench
target = bench.make_target(pkg, "2op", 10, 0)
es = pkg.HipES(args.parents, args.offspring, pkg.capi.SYNTH_2OP, 10, None, bench.VOICES["2op"][0], seed=1)
stream = torch.cuda.Stream(device=device)
es.set_stream(stream.cuda_stream)
es.set_target_audio(target)
E = 16
ex = pkg.island.IslandExchange(0, 1, E, es.D, device, overlap=True)
# a one-rank world never exchanges; borrow the overlapped schedule's moving parts by hand (world = 1: nothing is injected)
ex.world = 1
side = torch.cuda.Stream(device=device)
mine = [torch.empty(E, ex.width, device=device) for _ in range(2)]
allb = [torch.empty(E, ex.width, device=device) for _ in range(2)]
packed = [torch.cuda.Event(), torch.cuda.Event()]
arrived = [torch.cuda.Event(), torch.cuda.Event()]
VARIANT = 0


def generation(x):
    idx = x & 1
    es.fuse_exchange_next_sort(mine[idx].data_ptr(), E, None, 1, 0, E, arrived[idx ^ 1].cuda_event if x > 0 else None)
    es.execute_generations(1)
    if VARIANT == 0:
        packed[idx].record()
        with torch.cuda.stream(side):
            side.wait_event(packed[idx])
            dist.all_gather_into_tensor(allb[idx], mine[idx])
            arrived[idx].record()
    else:
        # the collective's stream waits for the island's stream directly (torch records the event), the side stream only
        # for the collective: one cross-stream hop less before `arrived`
        work = dist.all_gather_into_tensor(allb[idx], mine[idx], async_op=True)
        with torch.cuda.stream(side):
            work.wait()
            arrived[idx].record()


with torch.cuda.stream(stream):
    es.init_population(0)
    es.execute_generations(50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x in range(args.gens):
        es.execute_generations(1)
    t_enq_plain = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_plain = time.perf_counter() - t0
    print(f"P={es.P}: plain loop {1e6 * t_plain / args.gens:.1f} us per generation (host enqueue {1e6 * t_enq_plain / args.gens:.1f})")
    for VARIANT in (0, 1, 0, 1):
        for x in range(20):
            generation(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for x in range(args.gens):
            generation(x)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"  one-rank exchange, {'side stream waits for the island, collective under it' if VARIANT == 0 else 'collective waits for the island directly'}: "
              f"{1e6 * t_all / args.gens:.1f} us per generation (host {1e6 * t_enq / args.gens:.1f}, includes the host gate)")
dist.destroy_process_group()
