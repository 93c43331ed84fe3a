"""CPU tests of the oracle (oracle/sots_oracle.c) against the committed golden vectors
(tests/golden/golden_v1.npz, produced by the independent NumPy restatement in
tests/golden/make_golden.py), the Random123 Philox known answers, a naive DFT and the
self-match property.  The reference ships no fixtures for this path; these pins are this
repository's own.  The pins that come from the reference itself - outputs of its device kernels, compiled as they stand
and run on an MI355X - are in tests/test_ocl_reference.py.
"""
import os

import numpy as np
import pytest

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
PMAX = {0: [3520.0, 8.0, 3520.0, 1.0],
        1: [3520.0, 8.0, 3520.0, 8.0, 3520.0, 8.0],
        2: [3520.0, 8.0, 3520.0, 1.0],
        3: [3520.0, 8.0, 3520.0, 8.0, 3520.0, 8.0, 3520.0, 8.0]}


def test_philox_known_answers(O):
    # Random123 kat_vectors, philox4x32-10
    kat = [([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
           ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
           ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
            [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1])]
    for ctr, key, want in kat:
        assert list(O.philox(ctr, key)) == want


def test_wavetable(O):
    t = O.wavetable()
    # Evolutionary_Strategy.hpp:328-331: period W-1, so the last entry is sin(2 pi) in fp32
    assert t[0] == 0.0 and t[8192] == 1.0
    assert t[32767] == np.float32(1.7484555e-07)
    assert np.float32(32768) / np.float32(44100) == np.float32(0.74303854)
    # host sinf vs a correctly rounded sine: at most one ulp anywhere
    g = GOLD["wavetable"]
    assert np.abs(t - g).max() <= np.spacing(np.float32(1.0))
    assert (t != g).mean() < 0.05


def test_window(O):
    w, wf = O.window(1024)
    assert np.array_equal(w, GOLD["window_1024"])
    assert wf == GOLD["window_factor_1024"][0] == np.float32(1.0)
    n = np.arange(1024)
    assert np.abs(w - (1 - np.cos(2 * np.pi * n / 1024))).max() < 1e-11
    assert O.window(4096)[1] == GOLD["window_factor_4096"][0]


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_synth_spectrum_fitness_golden(O, kind):
    tab = GOLD["wavetable"]  # same table on both sides: the synthesis algorithm is what is pinned
    params = GOLD[f"k{kind}_n1024_params"]
    audio = GOLD[f"k{kind}_n1024_audio"]
    mags = GOLD[f"k{kind}_n1024_mag"]
    fit = GOLD[f"k{kind}_n1024_fitness_vs_row0"]
    for i, pv in enumerate(params):
        a = O.synth(kind, pv, [0.0] * len(PMAX[kind]), PMAX[kind], 1024, table=tab)
        assert np.array_equal(a, audio[i]), f"kind {kind} row {i}"
        m = O.spectrum(a)
        # fp64 FFTs agree to ~1e-13; the final hypotf/scale is fp32, so allow one ulp of the peak
        np.testing.assert_allclose(m, mags[i], rtol=0, atol=np.spacing(np.float32(mags[i].max())))
        f = O.fitness(m, O.spectrum(audio[0]))
        np.testing.assert_allclose(f, fit[i], rtol=1e-5, atol=1e-12)
    assert fit[0] == 0.0


def test_synth_4096_and_param_min(O):
    tab = GOLD["wavetable"]
    a = O.synth(3, GOLD["k3_n4096_params"][0], [0.0] * 8, PMAX[3], 4096, table=tab)
    assert np.array_equal(a, GOLD["k3_n4096_audio"][0])
    m = O.spectrum(a)
    np.testing.assert_allclose(m, GOLD["k3_n4096_mag"][0], rtol=0, atol=np.spacing(np.float32(m.max())))
    b = O.synth(0, [0.9, 1.0, 0.02, 0.5], GOLD["k0_pmin"], PMAX[0], 1024, table=tab)
    assert np.array_equal(b, GOLD["k0_pmin_audio"][0])


def test_rfft_against_naive_dft_and_numpy(O):
    rng = np.random.default_rng(1)
    for n in (512, 1024, 4096):
        a = rng.standard_normal(n).astype(np.float32)
        w, _ = O.window(n)
        x = O.rfft(a, w)
        ref = np.fft.rfft(a.astype(np.float64) * w)
        assert np.abs(x - ref).max() < 1e-10 * np.abs(ref).max()
        if n <= 1024:
            assert np.abs(x - O.rfft(a, w, naive=True)).max() < 1e-9 * np.abs(ref).max()
        # Parseval on the windowed signal
        xw = a.astype(np.float64) * w
        full = np.abs(x[0]) ** 2 + np.abs(x[-1]) ** 2 + 2 * np.sum(np.abs(x[1:-1]) ** 2)
        assert abs(full / n - np.sum(xw ** 2)) < 1e-9 * np.sum(xw ** 2)


def test_self_match_known_answer(O):
    # the reference's debug constants (ocl_program.cl:247-250): 1450 Hz, I=3, 200 Hz, A=1
    vals = [0.411931818, 0.375, 0.0568181818, 1.0]
    a = O.synth(0, vals, [0.0] * 4, PMAX[0], 1024)
    m = O.spectrum(a)
    assert O.fitness(m, m) == 0.0
    b = O.synth(0, [0.5, 0.375, 0.0568181818, 1.0], [0.0] * 4, PMAX[0], 1024)
    assert O.fitness(O.spectrum(b), m) > 1e-4
    # shipped target of parameters.json:39 with the 3-op voice
    t3 = [3078 / 3520, 2 / 8, 3015 / 3520, 1.5 / 8, 3141 / 3520, 1 / 8]
    a3 = O.synth(1, t3, [0.0] * 6, PMAX[1], 2048)
    assert np.isfinite(a3).all() and np.abs(a3).max() <= 8.0 * 3520 * 0 + 1.0 * 8 * 3520  # bounded by m3


def test_init_recombine_mutate_golden(O):
    seed, gid_base, chunk, parents, block, gen = [int(x) for x in GOLD["meta_seed_gid_chunk_parents_block_gen"]]
    v, s = O.init_population(64, 6, seed, gid_base, chunk)
    assert np.array_equal(v, GOLD["init_values"]) and np.array_equal(s, GOLD["init_steps"])
    rv, rs = O.recombine(v, s, parents, block)
    assert np.array_equal(rv, GOLD["recombine_values"]) and np.array_equal(rs, GOLD["recombine_steps"])
    mv, ms = O.mutate(rv, GOLD["mutate_in_steps"], seed, gid_base, gen)
    assert np.array_equal(mv, GOLD["mutate_values"])
    np.testing.assert_allclose(ms, GOLD["mutate_steps"], rtol=1e-6, atol=0)


def test_recombine_is_a_per_gene_permutation_of_the_parent_block(O):
    rng = np.random.default_rng(2)
    v = rng.random((256, 4), dtype=np.float32)
    s = rng.random((256, 4), dtype=np.float32)
    rv, rs = O.recombine(v, s, 64, 32)
    for b in range(8):
        pb = b % 2
        for g in range(4):
            assert np.array_equal(np.sort(rv[b * 32:(b + 1) * 32, g]), np.sort(v[pb * 32:(pb + 1) * 32, g]))
            assert np.array_equal(np.sort(rs[b * 32:(b + 1) * 32, g]), np.sort(s[pb * 32:(pb + 1) * 32, g]))
    # gene 0 is never shifted (shift = g*(b+1) = 0)
    assert np.array_equal(rv[:32, 0], v[:32, 0])


def test_mutate_statistics(O):
    v = np.full((4096, 4), 0.5, np.float32)
    s = np.full((4096, 4), 0.1, np.float32)
    mv, ms = O.mutate(v, s, 1234, 0, 0)
    g = (mv - v) / (s * 1.4)  # |Ek*gauss| <= gauss * alpha
    assert abs(g.mean()) < 0.01
    # mean of 12 uniforms on [-1,1]: sigma = 1/6
    assert 0.1 < (mv - v).std() / 0.1 < 0.25
    assert np.all(ms > 0)
    mv2, _ = O.mutate(v, s, 1234, 0, 1)
    assert not np.array_equal(mv, mv2)  # the generation is part of the counter
    mv3, _ = O.mutate(v, s, 1234, 0, 0)
    assert np.array_equal(mv, mv3)      # and the stream is reproducible


def test_sort_golden_and_edge_cases(O):
    f = GOLD["sort_fitness"]
    assert np.array_equal(O.sort_perm(f), GOLD["sort_perm"])
    assert list(O.sort_perm(np.array([1.0], np.float32))) == [0]
    assert list(O.sort_perm(np.array([2.0, 1.0], np.float32))) == [1, 0]
    same = np.zeros(33, np.float32)
    assert list(O.sort_perm(same)) == list(range(33))


def test_oracle_generation_and_islands(O):
    es = O.OracleES(32, 96, 0, 10, None, PMAX[0], seed=7, recomb_block=32)
    tgt = O.synth(0, [0.411931818, 0.375, 0.0568181818, 1.0], [0.0] * 4, PMAX[0], 1024)
    es.set_target_audio(tgt)
    es.init_population(0)
    first = None
    for _ in range(6):
        es.generation()
        _, _, f = es.read_population()
        assert np.all(np.diff(f) >= 0)
        first = f[0] if first is None else first
    rows = es.pack_elites(4)
    v, s, f = es.read_population()
    assert np.array_equal(rows[:, 0], f[:4]) and np.array_equal(rows[:, 1:5], v[:4]) and np.array_equal(rows[:, 5:], s[:4])
    imm = np.arange(2 * 9, dtype=np.float32).reshape(2, 9)
    es.inject(imm)
    v, s, f = es.read_population()
    assert np.array_equal(v[30:32], imm[:, 1:5]) and np.array_equal(f[30:32], imm[:, 0])
