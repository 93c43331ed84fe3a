"""Worker for test_islands_gloo.py: one island per process over torch.distributed (gloo)."""
import importlib
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd"


def main():
    out_dir, gens, elites, overlap = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) != 0
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    island_mod = importlib.import_module(PKG + ".island")
    from oracle import oracle as O

    pmax = [3520.0, 8.0, 3520.0, 1.0]
    parents, offspring = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (32, 96)
    P = parents + offspring
    es = O.OracleES(parents, offspring, O.SYNTH_2OP, 10, None, pmax, seed=0x5EED0001, recomb_block=32, gid_base=rank * P)
    es.set_target_audio(O.synth(0, [1450 / 3520, 3 / 8, 200 / 3520, 1.0], [0.0] * 4, pmax, 1024))
    es.init_population(0)
    ex = island_mod.IslandExchange(rank, world, elites, es.D, "cpu", overlap=overlap)
    sent = []
    for _ in range(gens):
        es.generation()
        sent.append(es.pack_elites(elites))
        ex.migrate_host(es.pack_elites, es.inject)
    ex.finish()
    v, s, f = es.read_population()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), v=v, s=s, f=f, sent=np.stack(sent))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
