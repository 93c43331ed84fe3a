"""State handling of the C-ABI context on one MI355X: the lazily completed sort order (enum sots_sort_mode),
re-initialisation, and checkpoint / resume over the reference's save / restore surface
(readPopulationData / writePopulationData, Evolutionary_Strategy.hpp:642-649, plus sots_set_generation:
the counter-based PRNG is keyed by the generation, so a resumed run must carry it along).

Run with: python -m pytest tests -m gpu
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PMAX = [3520.0, 8.0, 3520.0, 1.0]
SEED = 0x5EED0001


def make(pkg, O, parents, offspring, log2n=10, **kw):
    es = pkg.HipES(parents, offspring, pkg.capi.SYNTH_2OP, log2n, None, PMAX, seed=SEED, workgroup_size=32, **kw)
    es.set_target_audio(O.synth(0, [1450 / 3520, 3 / 8, 200 / 3520, 1.0], [0.0] * 4, PMAX, 1 << log2n))
    return es


# 4096 + 12288: the selection applies (P > 1024, S <= P/2), so the fused loop leaves a pending tail behind
@pytest.mark.parametrize("parents,offspring", [(4096, 12288), (64, 192)])
def test_init_population_after_a_run_is_a_fresh_population(pkg, O, parents, offspring):
    """the lazy-sort state of the LAST run must not be completed into the rows of a freshly
    initialised population."""
    a, b = make(pkg, O, parents, offspring), make(pkg, O, parents, offspring)
    a.init_population(0)
    a.execute_generations(3)
    a.init_population(1)
    b.init_population(1)
    for x, y in zip(a.read_population(), b.read_population()):
        assert np.array_equal(x, y)
    # ... and the two then evolve alike, whichever loop runs
    a.execute_generations(2)
    b.execute_generation()
    b.execute_generation()
    for x, y in zip(a.read_population(), b.read_population()):
        assert np.array_equal(x, y)
    assert a.generation == b.generation == 2
    a.close()
    b.close()


def test_top_only_state_ends_where_rows_are_written(pkg, O):
    """SOTS_SORT_TOP_ONLY never produces rows S..P-1; the pending state must not outlive the half it would be
    completed from: a stage that overwrites it ends the state, leaving the mode with the half intact
    completes it, and packing more elites than were placed is an error."""
    parents, offspring = 4096, 12288
    S = parents
    full = make(pkg, O, parents, offspring)
    full.init_population(0)
    full.execute_generations(2)
    want = full.read_population()

    es = make(pkg, O, parents, offspring)
    es.set_sort_mode(pkg.capi.SORT_TOP_ONLY)
    es.init_population(0)
    es.execute_generations(2)
    got = es.read_population()  # a pure read: the state stays pending
    for x, y in zip(got, want):
        assert np.array_equal(x[:S], y[:S])
    with pytest.raises(pkg.SotsError) as e:
        es.pack_elites(S + 1)
    assert e.value.code == -5  # SOTS_ERR_STATE
    assert np.array_equal(es.pack_elites(16)[:, 0], want[2][:16])
    # leaving the mode while the unsorted half is intact: the rest of the order appears
    es.set_sort_mode(pkg.capi.SORT_LAZY_TAIL)
    for x, y in zip(es.read_population(), want):
        assert np.array_equal(x, y)

    # a stage that writes rows ends the state: recombine overwrites the unsorted half, so switching modes
    # afterwards must NOT "complete" anything from it
    es.set_sort_mode(pkg.capi.SORT_TOP_ONLY)
    es.execute_generations(1)
    top = [x[:S].copy() for x in es.read_population()]
    es.recombine()  # current half <- recombination of rows 0..S-1 (all it reads)
    rec = es.read_population()
    es.set_sort_mode(pkg.capi.SORT_LAZY_TAIL)
    for x, y in zip(es.read_population(), rec):
        assert np.array_equal(x, y, equal_nan=True)
    full.execute_generations(1)
    fv, fs, ff = full.read_population()
    assert np.array_equal(top[0], fv[:S]) and np.array_equal(top[2], ff[:S])
    full.recombine()
    for x, y in zip(full.read_population()[:2], rec[:2]):
        assert np.array_equal(x, y)  # recombination reads whole parent blocks only: the same offspring either way
    es.close()
    full.close()


@pytest.mark.parametrize("parents,offspring,mode", [(4096, 12288, "lazy"), (4096, 12288, "full"), (256, 768, "lazy"),
                                                    (16384, 49152, "lazy")])
@pytest.mark.parametrize("loop", ["fused", "staged"])
def test_resume_reproduces_the_trajectory(pkg, O, parents, offspring, mode, loop):
    """2k generations straight == k generations -> read_population -> FRESH context -> write_population +
    set_generation(k) -> k generations, bit for bit (SURVEY 5 'Checkpoint / resume')."""
    k = 4
    if loop == "staged" and parents + offspring > 20000:
        pytest.skip("the staged loop at this size is covered by the fused == staged tests")

    def run(es, n):
        if loop == "fused":
            es.execute_generations(n)
        else:
            for _ in range(n):
                es.execute_generation()

    def fresh():
        es = make(pkg, O, parents, offspring)
        if mode == "full":
            es.set_sort_mode(pkg.capi.SORT_FULL)
        return es

    straight = fresh()
    straight.init_population(0)
    run(straight, 2 * k)
    want = straight.read_population()
    straight.close()

    first = fresh()
    first.init_population(0)
    run(first, k)
    v, s, f = first.read_population()
    assert first.generation == k
    first.close()

    second = fresh()
    second.write_population(v, s, f)
    second.generation = k
    run(second, k)
    got = second.read_population()
    assert second.generation == 2 * k
    second.close()
    for x, y in zip(got, want):
        assert np.array_equal(x, y, equal_nan=True)


def test_group_calls_of_one_generation_equal_one_call(pkg, O):
    """The island threads are persistent: n calls of sots_group_execute_generations(1) (what the C++ class's
    executeGeneration does) walk the same exchange schedule as one call of n."""
    parents, offspring, elites, gens = 2048, 6144, 16, 7
    target = O.synth(0, [1450 / 3520, 3 / 8, 200 / 3520, 1.0], [0.0] * 4, PMAX, 1024)
    pops = []
    for overlap in (False, True):
        for chunks in ([gens], [1] * gens, [2, 1, 3, 1]):
            g = pkg.HipGroup([0, 0, 0], elites, parents, offspring, pkg.capi.SYNTH_2OP, 10, None, PMAX, seed=SEED,
                             migration_interval=2, overlap=overlap)
            g.set_target_audio(target)
            g.init_population(0)
            for n in chunks:
                g.execute_generations(n)
            g.synchronize()
            pops.append([g.island(r).read_population() for r in range(3)])
            g.close()
        for other in pops[1:]:
            for isl_a, isl_b in zip(pops[0], other):
                for x, y in zip(isl_a, isl_b):
                    assert np.array_equal(x, y, equal_nan=True), f"overlap={overlap}"
        pops.clear()


def test_group_rejects_immigrants_beyond_the_breeding_rows(pkg, O):
    """80 parents with blocks of 32: recombination reads two whole blocks (64 rows).  72 immigrants fit the 80
    parents but not the rows they must land in - refused when the group is made, not at the first exchange
    inside the island threads."""
    with pytest.raises(pkg.SotsError):
        pkg.HipGroup([0, 0, 0], 36, 80, 176, pkg.capi.SYNTH_2OP, 10, None, PMAX)
    g = pkg.HipGroup([0, 0, 0], 32, 80, 176, pkg.capi.SYNTH_2OP, 10, None, PMAX)  # 64 immigrants: the whole breeding rows
    g.close()


# kind, parents, offspring, world, elites, sort mode: k_sort_small (P <= 1024, rows of 25 floats; immigrant and elite ranges
# overlapping), tile sort + rank scatter, the selection kernels (direct and merged tiles), the two-level full sort
@pytest.mark.parametrize("kind,parents,offspring,world,elites,full", [
    (0, 32, 96, 2, 24, False), (2, 64, 192, 3, 16, False), (3, 256, 768, 2, 16, False), (0, 2048, 6144, 4, 16, False),
    (3, 4096, 12288, 8, 16, False), (2, 4096, 12288, 2, 16, True), (0, 32768, 98304, 8, 16, False), (0, 65536, 196608, 8, 16, False)])
def test_exchange_inside_the_sort_kernel_equals_the_two_launches(pkg, O, kind, parents, offspring, world, elites, full):
    """sots_fuse_exchange_next_sort: the generation's sort kernel takes the immigrant rows from the gathered buffer and
    writes the best rows to the send buffer - the same population and the same packed rows as
    sots_inject_gathered_device + sots_pack_elites_device after the generation, for every kernel that moves sorted rows."""
    import torch
    pmax = {0: PMAX, 2: PMAX + [0.0] * 8, 3: [3520.0, 8.0] * 4}[kind]
    D = {0: 4, 2: 12, 3: 8}[kind]
    rank = world - 1 if world > 2 else 0
    rng = np.random.default_rng(parents + world)
    gathered = torch.from_numpy(rng.random((world * elites, 2 * D + 1), dtype=np.float32)).cuda()
    target = O.synth(kind, list(rng.random(D)), [0.0] * D, pmax, 512)
    pops, packs = [], []
    for fused in (False, True):
        es = pkg.HipES(parents, offspring, kind, 9, None, pmax, seed=SEED, workgroup_size=32)
        es.set_target_audio(target)
        if full:
            es.set_sort_mode(pkg.capi.SORT_FULL)
        es.init_population(0)
        es.execute_generations(2)
        sink = torch.full((elites, 2 * D + 1), -3.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        if fused:
            es.fuse_exchange_next_sort(sink.data_ptr(), elites, gathered.data_ptr(), world, rank, elites)
            es.execute_generations(1)
        else:
            es.execute_generations(1)
            es.inject_gathered_device(gathered.data_ptr(), world, rank, elites)
            es.pack_elites_device(sink.data_ptr(), elites)
        es.synchronize()
        packs.append(sink.cpu().numpy())
        es.execute_generations(1)  # the immigrants take part in the next recombination
        pops.append(es.read_population())
        es.close()
    assert np.array_equal(packs[0], packs[1], equal_nan=True)
    for x, y in zip(*pops):
        assert np.array_equal(x, y, equal_nan=True)
    # the exchange is used once
    assert not np.all(packs[1] == -3.0)
