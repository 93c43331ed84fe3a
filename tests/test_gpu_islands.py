"""The multi-rank DEVICE path on one MI355X: two (and three) fresh processes, one island each, all
on cuda:0, exchanging elites with island.IslandExchange.migrate_device (device tensors, the
island's own stream, double-buffered mine/all tensors in the overlapped schedule) - compared bit
for bit with a single-process run of the same islands that exchange through the blocking HOST
calls sots_pack_elites_host / sots_inject_immigrants_host.  Pins the stream ordering of
work.wait(), the buffer reuse and the rank * elites skip of sots_inject_gathered_device before
RCCL ever sees 8 ranks."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMAX = [3520.0, 8.0, 3520.0, 1.0]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def simulate(pkg, target, world, gens, elites, overlap, parents, offspring):
    P = parents + offspring
    isl = []
    for r in range(world):
        es = pkg.HipES(parents, offspring, pkg.capi.SYNTH_2OP, 10, None, PMAX, seed=0x5EED0001, workgroup_size=32,
                       device=0, gid_base=r * P)
        es.set_target_audio(target)
        es.init_population(0)
        isl.append(es)
    in_flight = None
    for _ in range(gens):
        for es in isl:
            es.execute_generations(1)
        if overlap and in_flight is not None:
            for r, es in enumerate(isl):
                es.inject_immigrants(np.concatenate([in_flight[q] for q in range(world) if q != r]))
        packs = [es.pack_elites(elites) for es in isl]
        if overlap:
            in_flight = packs
        else:
            for r, es in enumerate(isl):
                es.inject_immigrants(np.concatenate([packs[q] for q in range(world) if q != r]))
    out = [es.read_population() for es in isl]
    for es in isl:
        es.close()
    return out


# 2048 + 6144: the six-launch loop (variation kernel + cut synthesis); 49152 + 16384 with world 2 would be
# the fused-variation loop but is left to bench.py's rehearsal - the exchange code is the same
# fused = 1: IslandExchange.generation (pack and inject inside the generation's sort kernel, what bench.py runs);
# 4096 + 12288 takes the selection kernels, the small ones k_sort_small, 2048 + 6144 the tile sort + rank scatter
# reinit = 1: the ranks first run three generations, then call init_population again WITHOUT IslandExchange.restart():
# the exchange object must drop the old population's rows in flight by itself
@pytest.mark.parametrize("world,overlap,parents,offspring,fused,reinit", [
    (2, 0, 2048, 6144, 0, 0), (2, 1, 2048, 6144, 0, 0), (3, 1, 96, 160, 0, 0), (3, 0, 80, 176, 0, 0),
    (2, 0, 2048, 6144, 1, 0), (2, 1, 2048, 6144, 1, 0), (3, 1, 96, 160, 1, 0), (3, 0, 80, 176, 1, 0), (2, 1, 4096, 12288, 1, 0),
    (3, 0, 4096, 12288, 1, 0), (2, 1, 2048, 6144, 1, 1), (2, 1, 96, 160, 0, 1)])
def test_migrate_device_across_processes_equals_host_exchange(tmp_path, pkg, O, world, overlap, parents, offspring, fused, reinit):
    gens, elites = 6, 16
    target = O.synth(0, [1450 / 3520, 3 / 8, 200 / 3520, 1.0], [0.0] * 4, PMAX, 1024)
    np.save(tmp_path / "target.npy", target)
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(rank), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_island_gpu_worker.py"), str(tmp_path),
                                       str(gens), str(elites), str(overlap), str(parents), str(offspring), str(fused), str(reinit)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out.decode()
    want = simulate(pkg, target, world, gens, elites, overlap, parents, offspring)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["v"], want[r][0]), f"rank {r} values"
        assert np.array_equal(got["s"], want[r][1]), f"rank {r} steps"
        assert np.array_equal(got["f"], want[r][2], equal_nan=True), f"rank {r} fitness"
    # islands are distinct streams
    assert not np.array_equal(want[0][0], want[1][0])
    # same-generation schedule: the run ends with sort -> pack -> inject, so every island's breeding tail
    # holds the other islands' current elites in rank order.  With 80 parents and blocks of 32 the breeding
    # rows are the two whole parent blocks, so the tail is rows 64 - n .. 63, not 80 - n .. 79.
    if not overlap:
        n = (world - 1) * elites
        breeding = max(1, parents // 32) * 32
        for r in range(world):
            others_f = np.concatenate([want[q][2][:elites] for q in range(world) if q != r])
            others_v = np.concatenate([want[q][0][:elites] for q in range(world) if q != r])
            assert np.array_equal(want[r][2][breeding - n:breeding], others_f)
            assert np.array_equal(want[r][0][breeding - n:breeding], others_v)


# ---- the island group inside the library (sots_group_*, csrc/sots_group.hip) ----------------------------------
def test_one_device_group_is_the_plain_context(pkg, O):
    """numDevices = 1 degrades to one island that exchanges nothing: bit-identical to a plain context; and the
    same island with the RCCL backend forced (a one-rank communicator: librccl is opened, ncclCommInitAll and
    ncclAllGather run on the island's stream) injects nothing and is bit-identical too."""
    target = O.synth(0, [1450 / 3520, 3 / 8, 200 / 3520, 1.0], [0.0] * 4, PMAX, 1024)
    es = pkg.HipES(2048, 6144, pkg.capi.SYNTH_2OP, 10, None, PMAX, seed=0x5EED0001, workgroup_size=32)
    es.set_target_audio(target)
    es.init_population(0)
    es.execute_generations(6)
    want = es.read_population()
    es.close()
    # (True, True, 1 | 2): the overlapped, host-gated schedule over RCCL - side-stream ncclAllGather, the island's thread
    # waiting for `arrived` before the sort that takes the rows, with and without generations between two exchanges.
    # One rank only: more than one DISTINCT device has not been available to these tests.
    for force, overlap, interval in ((False, False, 1), (True, False, 1), (True, True, 1), (True, True, 2)):
        g = pkg.HipGroup([0], 16, 2048, 6144, pkg.capi.SYNTH_2OP, 10, None, PMAX, seed=0x5EED0001, force_rccl=force,
                         overlap=overlap, migration_interval=interval)
        assert g.size == 1 and g.uses_rccl == force
        g.set_target_audio(target)
        g.init_population(0)
        g.execute_generations(4)
        g.execute_generations(2)
        g.synchronize()
        got = g.island(0).read_population()
        for x, y in zip(got, want):
            assert np.array_equal(x, y), f"force_rccl={force} overlap={overlap} interval={interval}"
        island, fit = g.best()
        assert island == 0 and fit == want[2].min()
        g.close()


# unfused: 0 = pack and inject inside the sort kernel; in the overlapped schedule the HOST waits for the previous exchange's
# `arrived` event before it enqueues the sort that takes its rows (host-gated, the default);
# 1 = pack and inject as launches of their own; 2 = fused, but the compute stream waits for the side stream's event
# (hipStreamWaitEvent) instead of the host
@pytest.mark.parametrize("parents,offspring,unfused", [(2048, 6144, 0), (2048, 6144, 1), (4096, 12288, 0), (96, 160, 0), (4096, 12288, 2)])
@pytest.mark.parametrize("world,overlap,interval", [(2, False, 1), (2, True, 1), (3, False, 2), (3, True, 1), (3, True, 2)])
def test_group_of_islands_sharing_the_gpu_equals_host_exchange(pkg, O, world, overlap, interval, parents, offspring, unfused):
    """Several islands of one group on device 0 (event-ordered device-to-device copies stand in for RCCL, which
    refuses two ranks on one GPU): island threads, double-buffered exchange and both schedules against the
    single-threaded host-exchange simulation."""
    gens, elites = 6, 16
    target = O.synth(0, [1450 / 3520, 3 / 8, 200 / 3520, 1.0], [0.0] * 4, PMAX, 1024)
    g = pkg.HipGroup([0] * world, elites, parents, offspring, pkg.capi.SYNTH_2OP, 10, None, PMAX, seed=0x5EED0001,
                     migration_interval=interval, overlap=overlap, unfused=unfused == 1, event_waits=unfused == 2)
    assert g.size == world and not g.uses_rccl
    g.set_target_audio(target)
    g.init_population(0)
    g.execute_generations(4)
    g.execute_generations(gens - 4)      # the schedule carries over calls
    g.synchronize()
    got = [g.island(r).read_population() for r in range(world)]
    # reference: the same islands, exchanging through blocking host calls every `interval` generations
    P = parents + offspring
    isl = []
    for r in range(world):
        es = pkg.HipES(parents, offspring, pkg.capi.SYNTH_2OP, 10, None, PMAX, seed=0x5EED0001, workgroup_size=32, gid_base=r * P)
        es.set_target_audio(target)
        es.init_population(0)
        isl.append(es)
    in_flight = None
    for gen in range(1, gens + 1):
        for es in isl:
            es.execute_generations(1)
        if gen % interval:
            continue
        if overlap and in_flight is not None:
            for r, es in enumerate(isl):
                es.inject_immigrants(np.concatenate([in_flight[q] for q in range(world) if q != r]))
        packs = [es.pack_elites(elites) for es in isl]
        if overlap:
            in_flight = packs
        else:
            for r, es in enumerate(isl):
                es.inject_immigrants(np.concatenate([packs[q] for q in range(world) if q != r]))
    for r in range(world):
        want = isl[r].read_population()
        for x, y in zip(got[r], want):
            assert np.array_equal(x, y, equal_nan=True), f"island {r}"
    island, fit = g.best()
    assert fit == min(float(np.nanmin(p[2])) for p in got)
    for es in isl:
        es.close()
    g.close()


def host_exchange_reference(pkg, target, world, gens, elites, overlap, interval, parents, offspring, kind, log2n, pmax):
    """The same islands as plain contexts driven from ONE thread, exchanging through the blocking host calls
    (sots_pack_elites_host / sots_inject_immigrants_host) every `interval` generations."""
    P = parents + offspring
    isl = []
    for r in range(world):
        es = pkg.HipES(parents, offspring, kind, log2n, None, pmax, seed=0x5EED0001, workgroup_size=32, gid_base=r * P)
        es.set_target_audio(target)
        es.init_population(0)
        isl.append(es)
    in_flight = None
    for gen in range(1, gens + 1):
        for es in isl:
            es.execute_generations(1)
        if gen % interval:
            continue
        if overlap and in_flight is not None:
            for r, es in enumerate(isl):
                es.inject_immigrants(np.concatenate([in_flight[q] for q in range(world) if q != r]))
        packs = [es.pack_elites(elites) for es in isl]
        if overlap:
            in_flight = packs
        else:
            for r, es in enumerate(isl):
                es.inject_immigrants(np.concatenate([packs[q] for q in range(world) if q != r]))
    out = [es.read_population() for es in isl]
    for es in isl:
        es.close()
    return out


# BASELINE configs[3] and configs[4] AT THEIR STATED SIZE: eight islands (the 8-GPU run's shards) inside one group, all on
# device 0 - the reference is single-device (Evolutionary_Strategy_OpenCL.hpp:194-226), so the host-exchange simulation is
# the only other witness of this path.  8 x 0.55 GB of audio per side.
FULL_SIZE = {3: (8192, 24576, 3, 12, [3520.0, 8.0] * 4, [0.3, 0.25, 0.85, 0.19, 0.89, 0.125, 0.5, 0.1]),
             4: (32768, 98304, 0, 10, PMAX, [1450 / 3520, 3 / 8, 200 / 3520, 1.0])}


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("config", [3, 4])
def test_world8_group_at_full_baseline_size_equals_host_exchange(pkg, O, config, overlap):
    """configs[3]: 8 x 32768 = 262144 candidates, 4-op, N = 4096; configs[4]: 8 x 131072 = 1048576 candidates, 2-op,
    N = 1024.  Three generations with an exchange of 16 elites per island after each: every island bit-identical to the
    single-thread simulation, the immigrants sit where recombination reads them, every island ends sorted."""
    parents, offspring, kind, log2n, pmax, tvals = FULL_SIZE[config]
    world, gens, elites = 8, 3, 16
    P = parents + offspring
    target = O.synth(kind, tvals, [0.0] * len(tvals), pmax, 1 << log2n)
    g = pkg.HipGroup([0] * world, elites, parents, offspring, kind, log2n, None, pmax, seed=0x5EED0001,
                     migration_interval=1, overlap=overlap)
    assert g.size == world and not g.uses_rccl
    g.set_target_audio(target)
    g.init_population(0)
    g.execute_generations(2)
    g.execute_generations(gens - 2)
    g.synchronize()
    got = [g.island(r).read_population() for r in range(world)]
    island, fit = g.best()
    assert fit == min(float(np.nanmin(p[2])) for p in got)
    g.close()
    want = host_exchange_reference(pkg, target, world, gens, elites, overlap, 1, parents, offspring, kind, log2n, pmax)
    for r in range(world):
        for name, x, y in zip("vsf", got[r], want[r]):
            assert np.array_equal(x, y, equal_nan=True), f"config {config} island {r} {name}"
    assert sum(p[2].size for p in got) == world * P == (262144 if config == 3 else 1048576)
    n = (world - 1) * elites
    chunks = {}  # source island -> its elite rows as each other island received them
    for r in range(world):
        v, s, f = got[r]
        assert not np.array_equal(v, got[(r + 1) % world][0]), "islands are distinct streams"
        # the last (world-1)*16 of the rows recombination reads hold the other islands' elites in rank order (of the
        # generation just finished in the same-generation schedule, of the one before in the overlapped one: they were
        # taken by generation 3's sort); every other row is in order
        assert np.all(np.diff(f[:parents - n]) >= 0) and np.all(np.diff(f[parents:]) >= 0), f"island {r} not sorted"
        sources = [q for q in range(world) if q != r]
        for i, q in enumerate(sources):
            lo = parents - n + i * elites
            chunks.setdefault(q, []).append((f[lo:lo + elites], v[lo:lo + elites], s[lo:lo + elites]))
    for q, seen in chunks.items():
        assert len(seen) == world - 1
        for c in seen[1:]:
            assert all(np.array_equal(x, y) for x, y in zip(c, seen[0])), f"island {q}'s elites arrived differently"
        assert np.all(np.diff(seen[0][0]) >= 0)
        if not overlap:
            assert np.array_equal(seen[0][0], got[q][2][:elites]) and np.array_equal(seen[0][1], got[q][0][:elites])


def test_group_rejects_bad_arguments(pkg, O):
    with pytest.raises(pkg.SotsError):
        pkg.HipGroup([], 16, 64, 192, pkg.capi.SYNTH_2OP, 10, None, PMAX)
    with pytest.raises(pkg.SotsError):
        pkg.HipGroup([0, 0, 0, 0], 32, 64, 192, pkg.capi.SYNTH_2OP, 10, None, PMAX)   # 96 immigrants > 64 parents
    with pytest.raises(pkg.SotsError):
        pkg.HipGroup([0, 99], 4, 64, 192, pkg.capi.SYNTH_2OP, 10, None, PMAX)          # no such device
