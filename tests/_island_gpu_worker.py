"""Worker for test_gpu_islands.py: one island per process, every rank on cuda:0, torch.distributed
over gloo (RCCL refuses two ranks on one GPU).  The island runs the HIP library; the exchange is
island.IslandExchange.migrate_device - the code path bench.py drives over RCCL: pack on the
island's stream, all-gather of device tensors, sots_inject_gathered_device with the rank * elites skip."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd"


def main():
    out_dir, gens, elites, overlap = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) != 0
    parents, offspring = int(sys.argv[5]), int(sys.argv[6])
    fused = len(sys.argv) > 7 and int(sys.argv[7]) != 0  # pack + inject inside the sort kernel (IslandExchange.generation)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module(PKG)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    pmax = [3520.0, 8.0, 3520.0, 1.0]
    P = parents + offspring
    es = pkg.HipES(parents, offspring, pkg.capi.SYNTH_2OP, 10, None, pmax, seed=0x5EED0001, workgroup_size=32,
                   device=0, gid_base=rank * P)
    target = np.load(os.path.join(out_dir, "target.npy"))
    stream = torch.cuda.Stream(device=device)
    es.set_stream(stream.cuda_stream)
    es.set_target_audio(target)
    ex = pkg.island.IslandExchange(rank, world, elites, es.D, device, overlap=overlap)
    reinit = len(sys.argv) > 8 and int(sys.argv[8]) != 0  # a run of 3 generations first, then init_population WITHOUT restart()
    with torch.cuda.stream(stream):
        if reinit:
            es.init_population(0)
            for _ in range(3):
                if fused:
                    ex.generation(es)
                else:
                    es.execute_generations(1)
                    ex.migrate_device(es)
        es.init_population(0)
        for _ in range(gens):
            if fused:
                ex.generation(es)
            else:
                es.execute_generations(1)
                ex.migrate_device(es)
        ex.finish()
    torch.cuda.synchronize(device)
    v, s, f = es.read_population()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), v=v, s=s, f=f)
    es.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
