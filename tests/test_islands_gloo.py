"""The N > 1 path on CPU: world_size-2, -3 and -8 island runs over gloo, one process per island,
checked against a single-process simulation of the same islands.  The evaluator is the CPU
oracle; the exchange code (island.IslandExchange) is the one bench.py uses over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def simulate(O, world, gens, elites, overlap, parents=32, offspring=96):
    pmax = [3520.0, 8.0, 3520.0, 1.0]
    tgt = O.synth(0, [1450 / 3520, 3 / 8, 200 / 3520, 1.0], [0.0] * 4, pmax, 1024)
    isl = []
    for r in range(world):
        es = O.OracleES(parents, offspring, O.SYNTH_2OP, 10, None, pmax, seed=0x5EED0001, recomb_block=32,
                        gid_base=r * (parents + offspring))
        es.set_target_audio(tgt)
        es.init_population(0)
        isl.append(es)
    in_flight = None  # overlap schedule: elites gathered after generation g arrive after g+1
    for _ in range(gens):
        for es in isl:
            es.generation()
        if overlap and in_flight is not None:
            for r, es in enumerate(isl):
                es.inject(np.concatenate([in_flight[q] for q in range(world) if q != r]))
        packs = [es.pack_elites(elites) for es in isl]
        if overlap:
            in_flight = packs
        else:
            for r, es in enumerate(isl):
                es.inject(np.concatenate([packs[q] for q in range(world) if q != r]))
    return [es.read_population() for es in isl]


# (80, 176): numParents is not a multiple of the recombination block of 32, so the rows recombination reads are the two
# whole parent blocks (64 rows) and the immigrants sit at THEIR tail, rows 64 - n .. 63
# world 8 (round 4): the rank count of BASELINE configs[3] / [4] - 7 x 4 immigrants fit the 32 rows recombination reads
@pytest.mark.parametrize("world,overlap,parents,offspring", [(2, 0, 32, 96), (3, 0, 32, 96), (2, 1, 32, 96), (3, 1, 32, 96), (2, 0, 80, 176),
                                                             (8, 0, 32, 96), (8, 1, 32, 96)])
def test_island_exchange_over_gloo(tmp_path, O, world, overlap, parents, offspring):
    gens, elites = 4, 4
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_RANK=str(rank), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_island_worker.py"),
                                       str(tmp_path), str(gens), str(elites), str(overlap), str(parents), str(offspring)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()
    want = simulate(O, world, gens, elites, overlap, parents, offspring)
    got = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in range(world):
        assert np.array_equal(got[r]["v"], want[r][0])
        assert np.array_equal(got[r]["s"], want[r][1])
        assert np.array_equal(got[r]["f"], want[r][2])
    # every island's parent tail holds the other islands' elites, in rank order: the last ones sent,
    # or with the overlap schedule those of the generation before
    for r in range(world):
        others = np.concatenate([got[q]["sent"][-2 if overlap else -1] for q in range(world) if q != r])
        n = others.shape[0]
        breeding = max(1, parents // 32) * 32
        assert np.array_equal(got[r]["f"][breeding - n:breeding], others[:, 0])
        assert np.array_equal(got[r]["v"][breeding - n:breeding], others[:, 1:5])
    # islands are distinct streams (global individual ids differ)
    assert not np.array_equal(got[0]["v"], got[1]["v"])


def test_single_island_is_a_no_op(O, pkg):
    for overlap in (False, True):
        ex = pkg.island.IslandExchange(0, 1, 4, 4, "cpu", overlap=overlap)
        calls = []
        ex.migrate_host(lambda n: calls.append(n), lambda rows: calls.append(rows))
        ex.finish()
        assert calls == [] and ex.num_immigrants == 0
