"""GPU parity tests: every HIP stage, called through the C-ABI, against the CPU oracle on
the same seeded inputs.  Bit-exact for integer/index work and for every fp32 stage whose
arithmetic is order-fixed (init, recombine, mutate values, synthesis, window, sort);
stated tolerances for the fp32 FFT/fitness against the oracle's fp64 FFT.

Run with: python -m pytest tests -m gpu
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PMAX = {0: [3520.0, 8.0, 3520.0, 1.0],
        1: [3520.0, 8.0, 3520.0, 8.0, 3520.0, 8.0],
        2: [3520.0, 8.0, 3520.0, 1.0] + [0.0] * 8,
        3: [3520.0, 8.0, 3520.0, 8.0, 3520.0, 8.0, 3520.0, 8.0]}
SEED = 0x5EED0001

# fp32 FFT vs fp64 FFT: per-row max |dX| <= FFT_TOL * max|X| (log2(N) rounding layers of ~6e-8)
FFT_TOL = 3e-6
# north_star: fitness within 1e-4 relative of the CPU reference; the absolute floor covers
# near-perfect matches, where the sum of squares itself is at the rounding level of the
# spectra: FIT_ATOL_REL times the target's spectral energy sum(target^2)
FIT_RTOL, FIT_ATOL_REL = 1e-4, 1e-9


def fit_atol(O, tgt_audio):
    m = O.spectrum(tgt_audio).astype(np.float64)
    return FIT_ATOL_REL * float(np.sum(m * m)) + 1e-12


def make_pair(pkg, O, parents, offspring, kind=0, log2n=10, block=32, gid_base=0, seed=SEED, pmin=None, pmax=None):
    kw = dict(synth_kind=kind, audio_log2=log2n, param_min=pmin, param_max=pmax if pmax is not None else PMAX[kind],
              seed=seed, gid_base=gid_base)
    es = pkg.HipES(parents, offspring, workgroup_size=block, **kw)
    ref = O.OracleES(parents, offspring, recomb_block=block, **kw)
    return es, ref


def target_audio(O, kind, n):
    vals = {0: [1450.0 / 3520.0, 3.0 / 8.0, 200.0 / 3520.0, 1.0],
            1: [3078 / 3520.0, 2.0 / 8.0, 3015 / 3520.0, 1.5 / 8.0, 3141 / 3520.0, 1.0 / 8.0],
            2: [0.41, 0.375, 0.057, 1.0, 0.2, 0.5, 0.11, 0.7, 0.6, 0.1, 0.3, 0.4],
            3: [0.3, 0.25, 0.85, 0.19, 0.89, 0.125, 0.5, 0.1]}[kind]
    return O.synth(kind, vals, [0.0] * len(vals), PMAX[kind], n), vals


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_init_population_bitexact(pkg, O, kind):
    es, ref = make_pair(pkg, O, 64, 192, kind, gid_base=1000)
    for chunk in (0, 3):
        es.init_population(chunk)
        ref.init_population(chunk)
        v, s, f = es.read_population()
        rv, rs, rf = ref.read_population()
        assert np.array_equal(v, rv)
        assert np.array_equal(s, rs)
        assert np.array_equal(f, rf)
        assert v.min() >= 0.0 and v.max() <= 1.0
    es.close()


@pytest.mark.parametrize("parents,offspring,block,kind", [
    (64, 192, 32, 0), (16, 16, 32, 1), (256, 768, 64, 0), (96, 160, 32, 2), (128, 384, 128, 3), (3, 5, 1, 0)])
def test_recombine_bitexact(pkg, O, parents, offspring, block, kind):
    es, ref = make_pair(pkg, O, parents, offspring, kind, block=block)
    rng = np.random.default_rng(7)
    P, D = es.P, es.D
    v = rng.random((P, D), dtype=np.float32)
    s = rng.random((P, D), dtype=np.float32)
    es.write_population(v, s, np.zeros(P, np.float32))
    es.recombine()
    gv, gs, _ = es.read_population()
    ev, es_ = O.recombine(v, s, parents, block)
    assert np.array_equal(gv, ev)
    assert np.array_equal(gs, es_)
    # the source half is untouched (out-of-place)
    ov, os_, _ = es.read_population(other=True)
    assert np.array_equal(ov, v) and np.array_equal(os_, s)
    es.close()


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_mutate_values_bitexact_steps_close(pkg, O, kind):
    es, ref = make_pair(pkg, O, 64, 192, kind, gid_base=77)
    rng = np.random.default_rng(11)
    P, D = es.P, es.D
    v = rng.random((P, D), dtype=np.float32)
    s = (rng.random((P, D), dtype=np.float32) * 0.5 + 0.01).astype(np.float32)
    s[0, :] = 3.0  # large steps force the reflect branch (ocl_program.cl:176-182)
    for gen in (0, 1, 1234):
        es.write_population(v, s, None)
        es.generation = gen
        es.mutate()
        gv, gs, _ = es.read_population()
        ev, es_ = O.mutate(v, s, SEED, 77, gen)
        assert np.array_equal(gv, ev), f"values differ at generation {gen}"
        # exp/pow are libm on the CPU and ocml on the device: a few ulp
        np.testing.assert_allclose(gs, es_, rtol=2e-6, atol=0)
    es.close()


# N = 256 (k_fft_x with two complex points per lane) and N = 16384 (a workgroup per row for the transform): round 4, the
# reference takes any audioLengthLog2 (main.cpp:90) and renders 2^14 samples of its best match (main.cpp:273)
@pytest.mark.parametrize("kind,log2n", [(0, 10), (1, 11), (2, 10), (3, 12), (0, 9), (0, 13), (0, 8), (1, 8), (3, 8), (2, 8), (0, 14), (3, 14), (1, 15)])
def test_synthesise_bitexact(pkg, O, kind, log2n):
    es, ref = make_pair(pkg, O, 64, 192, kind, log2n)
    es.init_population(0)
    ref.init_population(0)
    v, s, _ = es.read_population()
    # plant the reference's known parameter sets and the corners of the unit cube
    _, tv = target_audio(O, kind, es.N)
    v[0] = tv
    v[1] = 0.0
    v[2] = 1.0
    es.write_population(v, s, None)
    ref.write_population(v, s, None)
    es.synthesise()
    ref.evaluate()
    assert np.array_equal(es.read_audio(), ref.audio())
    es.close()


@pytest.mark.parametrize("kind,log2n", [(0, 10), (1, 11), (2, 10), (3, 11)])
@pytest.mark.parametrize("parents,offspring", [(288, 896), (32, 64), (512, 1536)])
def test_synthesise_time_in_the_lanes_ragged_workgroups_bitexact(pkg, O, kind, log2n, parents, offspring):
    """A few individuals per CU: k_synth_tp (a scanning and an evaluating wavefront per operator, lane = sample in the
    evaluation).  1184 individuals are five per workgroup with four in the last (one batch of four plus one, and a partly
    filled batch), 96 one per workgroup, 2048 eight (two full batches; the triple voice, at most five per CU, runs
    k_synth there): every row bit for bit against the oracle, both from stored genes and - fused loop - from genes made
    inside the kernel (one thread per gene)."""
    es, ref = make_pair(pkg, O, parents, offspring, kind, log2n)
    es.init_population(0)
    ref.init_population(0)
    v, s, _ = es.read_population()
    _, tv = target_audio(O, kind, es.N)
    v[0], v[1], v[2] = tv, 0.0, 1.0
    v[es.P - 1] = tv
    es.write_population(v, s, None)
    ref.write_population(v, s, None)
    es.synthesise()
    ref.evaluate()
    assert np.array_equal(es.read_audio(), ref.audio())
    es.close()
    # the fused loop makes the individuals inside the kernel: same audio and same population as the separate stages
    a, _ = make_pair(pkg, O, parents, offspring, kind, log2n)
    b, _ = make_pair(pkg, O, parents, offspring, kind, log2n)
    tgt, _ = target_audio(O, kind, a.N)
    for e in (a, b):
        e.set_target_audio(tgt)
        e.init_population(0)
    a.execute_generations(1)
    b.recombine(); b.mutate(); b.synthesise()
    assert np.array_equal(a.read_audio(), b.read_audio())
    b.window(); b.fft(); b.fitness(); b.sort(); b.rotate()
    for x, y in zip(a.read_population(), b.read_population()):
        assert np.array_equal(x, y)
    a.close(); b.close()


def test_synthesise_two_op_large_population_uses_lane_per_individual_kernel(pkg, O):
    """Above 128 individuals per CU the 2-op voice runs k_synth (one lane per individual, four
    wavefronts per workgroup, looping over tiles) instead of the chain cut into two wavefronts; rows are
    spot-checked against the oracle."""
    parents, offspring = 16640, 49920          # P = 66560 = 260 per CU on 256 CUs
    es, _ = make_pair(pkg, O, parents, offspring, 0, 10)
    es.init_population(0)
    v, s, _ = es.read_population()
    es.synthesise()
    audio = es.read_audio()
    rng = np.random.default_rng(17)
    rows = np.concatenate([[0, 1, 63, 64, 319, 320, es.P - 1], rng.choice(es.P, 200, replace=False)])
    for r in rows:
        assert np.array_equal(audio[r], O.synth(0, v[r], [0.0] * 4, PMAX[0], es.N)), f"row {r}"
    es.close()


# k_synth_ol (round 4): the 4-operator voice from 65 individuals per CU (BASELINE configs[3]'s shard: 128) and the 3-operator voice at
# 65 ... 240 run with the OPERATORS in the lanes - a row of 16 lanes holds 4 individuals, hand-over by DPP row shifts, no LDS links,
# no barriers; tiles of up to 240 rows, larger shares in several equal tiles.
# Sizes: configs[3]'s shard; the same with a partly filled last workgroup; ragged populations (recombination blocks of 1) in wild
# parameter boxes (negative and beyond-table increments: operator 0 has the first wrap only, the index clamp acts on both sides - the
# clamped path; the reference's box takes the clamp-free one); 313 per CU (two tiles of 160) and 274 per CU of the 3-operator voice
# (k_synth there); the 2-operator voice (k_synth; with -DSOTS_OL_ALL the same kernel with two lanes per individual) keeps its cases.
@pytest.mark.parametrize("kind,log2n,parents,offspring,wild", [
    (3, 12, 8192, 24576, False), (3, 9, 8192, 24416, False), (3, 9, 5000, 15011, True), (3, 10, 4352, 13056, False),
    (0, 10, 4096, 12288, False), (0, 9, 3333, 9987, True), (0, 10, 16640, 49920 + 32 * 700, False),
    (1, 11, 3072, 9216, False), (1, 9, 3100, 9311, True), (1, 9, 20000, 50016, False), (1, 11, 8192, 24576, False), (1, 9, 6000, 18013, True),
    (3, 9, 3111, 9332, True), (3, 9, 20000, 60032, False)])
def test_synthesise_operators_in_the_lanes_bitexact(pkg, O, kind, log2n, parents, offspring, wild):
    pmin, pmax = WILD[kind] if wild else (None, PMAX[kind])
    block = 32 if (parents + offspring) % 32 == 0 else 1
    es, _ = make_pair(pkg, O, parents, offspring, kind, log2n, block=block, pmin=pmin, pmax=pmax)
    es.init_population(0)
    v, s, _ = es.read_population()
    _, tv = target_audio(O, kind, es.N)
    v[0], v[1], v[2], v[es.P - 1] = tv, 0.0, 1.0, tv
    es.write_population(v, s, None)
    es.synthesise()
    audio = es.read_audio()
    rng = np.random.default_rng(41 + kind)
    share = -(-es.P // 256)
    rows = np.unique(np.concatenate([np.arange(0, 70), np.arange(es.P - 70, es.P), np.arange(share - 40, share + 40) % es.P,
                                     np.arange(255, 265) * share % es.P, rng.choice(es.P, 120, replace=False)]))
    lo = [0.0] * es.D if pmin is None else pmin
    for r in rows:
        assert np.array_equal(audio[r], O.synth(kind, v[r], lo, pmax, es.N)), f"row {r}"
    es.close()
    if wild:
        return
    # the fused loop (the workgroup makes its individuals, a thread per gene, where the variation is folded in): same audio
    # and same population as the separate stages
    a, _ = make_pair(pkg, O, parents, offspring, kind, log2n, block=block)
    b, _ = make_pair(pkg, O, parents, offspring, kind, log2n, block=block)
    tgt, _ = target_audio(O, kind, a.N)
    for e in (a, b):
        e.set_target_audio(tgt)
        e.init_population(0)
    a.execute_generations(1)
    b.recombine(); b.mutate(); b.synthesise()
    assert np.array_equal(a.read_audio(), b.read_audio())
    b.window(); b.fft(); b.fitness(); b.sort(); b.rotate()
    for x, y in zip(a.read_population(), b.read_population()):
        assert np.array_equal(x, y)
    a.close(); b.close()


def test_synthesise_nonzero_param_min(pkg, O):
    pmin = [100.0, 0.5, 50.0, 0.1]
    es, ref = make_pair(pkg, O, 32, 32, 0, 10, pmin=pmin)
    es.init_population(0)
    ref.init_population(0)
    es.synthesise()
    ref.evaluate()
    assert np.array_equal(es.read_audio(), ref.audio())
    es.close()


# Parameter boxes far outside the reference's: negative and > Nyquist frequencies, so that phase
# increments are negative or exceed the table length, phases leave [0, W) for good, the index
# clamp acts on both sides and the free-running operators cannot take the clamp-free path.
WILD = {0: ([-2000.0, -4.0, -30000.0, -1.0], [60000.0, 12.0, 90000.0, 2.0]),
        1: ([-2000.0, -50000.0, -3.0, -20000.0, -4.0, -20000.0], [9000.0, 70000.0, 9.0, 40000.0, 6.0, 30000.0]),
        2: ([-2000.0, -4.0, -30000.0, -1.0], [60000.0, 12.0, 90000.0, 2.0]),
        3: ([-2000.0, -50000.0, -3.0, -20000.0, -4.0, -20000.0, -2.0, -9000.0],
            [9000.0, 70000.0, 9.0, 40000.0, 6.0, 30000.0, 5.0, 20000.0])}


@pytest.mark.parametrize("kind,log2n,parents,offspring", [(0, 10, 64, 192), (0, 9, 30, 35), (1, 10, 64, 64),
                                                          (2, 9, 64, 192), (3, 11, 32, 96)])
def test_synthesise_wild_parameter_ranges_bitexact(pkg, O, kind, log2n, parents, offspring):
    """The branch-free phase wraps (unsigned-minimum forms) and the index clamp must equal the
    oracle's conditional wraps for every phase value, not only for in-range ones."""
    pmin, pmax = WILD[kind]
    es, ref = make_pair(pkg, O, parents, offspring, kind, log2n, block=1 if (parents + offspring) % 32 else 32,
                        pmin=pmin, pmax=pmax)
    es.init_population(0)
    ref.init_population(0)
    v, s, _ = es.read_population()
    v[0] = 0.0
    v[1] = 1.0
    es.write_population(v, s, None)
    ref.write_population(v, s, None)
    es.synthesise()
    ref.evaluate()
    a, r = es.read_audio(), ref.audio()
    assert np.isfinite(r).all()
    assert np.array_equal(a, r)
    es.close()


def test_synthesise_wild_parameter_ranges_large_population(pkg, O):
    """Same, on the lane-per-individual kernel (P above 128 per CU), where wavefronts whose
    modulators all stay in range take the clamp-free index path and the others do not."""
    pmin, pmax = [-300.0, 0.0, -30000.0, -1.0], [45000.0, 12.0, 90000.0, 2.0]
    es, _ = make_pair(pkg, O, 16640, 49920, 0, 9, pmin=pmin, pmax=pmax)
    es.init_population(0)
    v, s, _ = es.read_population()
    v[:64, 0] = np.linspace(0.01, 0.9, 64, dtype=np.float32)    # first wavefront: every modulator in range
    v[64:128, 0] = 0.0                                           # second: negative increments
    es.write_population(v, s, None)
    es.synthesise()
    audio = es.read_audio()
    rng = np.random.default_rng(23)
    rows = np.concatenate([np.arange(0, 130), rng.choice(es.P, 150, replace=False), [es.P - 1]])
    for r in rows:
        assert np.array_equal(audio[r], O.synth(0, v[r], pmin, pmax, es.N)), f"row {r}"
    es.close()


def test_window_bitexact(pkg, O):
    es, _ = make_pair(pkg, O, 32, 96, 0, 10)
    rng = np.random.default_rng(3)
    a = (rng.random((es.P, es.N), dtype=np.float32) * 2 - 1).astype(np.float32)
    es.write_audio(a)
    es.window()
    w64, wf = O.window(es.N)
    expect = a * w64.astype(np.float32)[None, :]
    assert np.array_equal(es.read_audio(), expect)
    assert wf == np.float32(1.0)
    es.close()


@pytest.mark.parametrize("log2n", [8, 9, 10, 11, 12, 13, 14, 15])
def test_fft_against_fp64(pkg, O, log2n):
    es, _ = make_pair(pkg, O, 16, 48, 0, log2n)
    rng = np.random.default_rng(log2n)
    n = es.N
    a = (rng.standard_normal((es.P, n))).astype(np.float32)
    a[0] = 0.0
    a[1] = 1.0                      # DC only
    a[2] = np.cos(2 * np.pi * 5 * np.arange(n) / n)   # single bin
    a[3] = (-1.0) ** np.arange(n)   # Nyquist only
    a[4] = 0.0; a[4, 0] = 1.0       # impulse: flat spectrum
    es.write_audio(a)
    es.fft()
    spec = es.read_spectrum()
    assert spec.shape == (es.P, n // 2 + 4)
    ones = np.ones(n)
    for i in range(es.P):
        ref = O.rfft(a[i], ones)
        got = spec[i, : n // 2 + 1].astype(np.complex128)
        scale = max(np.abs(ref).max(), 1e-30)
        assert np.abs(got - ref).max() <= FFT_TOL * scale, f"row {i}"
        assert np.all(spec[i, n // 2 + 1:] == 0)  # padding bins stay zero
    es.close()


@pytest.mark.parametrize("kind,log2n", [(0, 10), (1, 11), (3, 12), (2, 10), (0, 8), (3, 8), (0, 14), (1, 15)])
def test_fitness_staged_and_fused_against_oracle(pkg, O, kind, log2n):
    es, ref = make_pair(pkg, O, 64, 192, kind, log2n)
    tgt, tv = target_audio(O, kind, es.N)
    es.set_target_audio(tgt)
    ref.set_target_audio(tgt)
    np.testing.assert_allclose(es.read_target(), O.spectrum(tgt), rtol=1e-6, atol=1e-9)
    es.init_population(0)
    ref.init_population(0)
    v, s, _ = es.read_population()
    v[5] = tv  # self-match row
    es.write_population(v, s, None)
    ref.write_population(v, s, None)
    ref.evaluate()
    _, _, rf = ref.read_population()
    # staged: synth -> window -> fft -> fitness
    es.synthesise(); es.window(); es.fft(); es.fitness()
    f_staged = es.read_fitness()
    atol = fit_atol(O, tgt)
    np.testing.assert_allclose(f_staged, rf, rtol=FIT_RTOL, atol=atol)
    # magnitudes of the materialised spectrum against the oracle's
    spec = es.read_spectrum()[:, : es.N // 2]
    mag = np.abs(spec.astype(np.complex128)) / es.N
    np.testing.assert_allclose(mag, ref.spectrum(), rtol=0, atol=3e-6 * max(1.0, np.abs(ref.spectrum()).max()))
    # self-match KAT (ocl_program.cl:247-250): fitness of the true parameters is ~0
    assert rf[5] == 0.0
    assert f_staged[5] <= atol
    es.close()


def sort_case(P, rng):
    f = rng.random(P, dtype=np.float32)
    if P >= 16:
        f[3] = f[11]                # ties
        f[5] = f[7] = f[9]
        f[2] = np.nan
        f[P - 1] = np.nan
        f[4] = np.inf
        f[6] = 0.0
        f[8] = -0.0
        f[10] = 0.0
    return f


# covers every sort plan: one launch / one workgroup (<= 1024: 1, 2, 4, 8, 16 runs of 64 keys, with and without
# padding keys), 1024-key tiles + one rank level (2048 .. 65536, with and without padding keys), 1024-key tiles
# merged into runs of 8192 + a rank level over the runs (131072 to 1M, with and without padding keys), global
# bitonic fallback (> 1M)
@pytest.mark.parametrize("parents,offspring,block", [
    (16, 16, 32), (64, 192, 32), (24, 72, 32), (32, 96, 32), (128, 384, 32), (248, 744, 32), (256, 768, 32),
    (512, 1536, 32), (704, 2304, 32), (1024, 3072, 32), (2048, 6144 + 32, 32),
    (16384, 49152, 32), (32768, 98304, 32), (40000, 110016, 32), (65536, 196608, 32), (262144, 786432, 32),
    (262144, 786432 + 32, 32)])
def test_sort_matches_stable_oracle(pkg, O, parents, offspring, block):
    es, _ = make_pair(pkg, O, parents, offspring, 0, 10, block=block)
    rng = np.random.default_rng(parents)
    P, D = es.P, es.D
    f = sort_case(P, rng)
    if P > 4096:
        f[::97] = f[5]              # many duplicates across tiles
    v = rng.random((P, D), dtype=np.float32)
    s = rng.random((P, D), dtype=np.float32)
    es.write_population(v, s, f)
    es.sort(); es.rotate()
    gv, gs, gf = es.read_population()
    perm = O.sort_perm(f)
    assert np.array_equal(gf, f[perm], equal_nan=True)
    assert np.array_equal(gv, v[perm])
    assert np.array_equal(gs, s[perm])
    es.close()


@pytest.mark.parametrize("kind,log2n", [(1, 10), (3, 10), (2, 10)])
def test_small_sort_moves_wide_rows(pkg, O, kind, log2n):
    """the one-launch sort moves four genes per trip: voices with 6, 8 and 12 genes, a population with padding keys"""
    es, _ = make_pair(pkg, O, 96, 288, kind, log2n)
    rng = np.random.default_rng(kind)
    P, D = es.P, es.D
    f = sort_case(P, rng)
    v = rng.random((P, D), dtype=np.float32)
    s = rng.random((P, D), dtype=np.float32)
    es.write_population(v, s, f)
    es.sort(); es.rotate()
    gv, gs, gf = es.read_population()
    perm = O.sort_perm(f)
    assert np.array_equal(gf, f[perm], equal_nan=True) and np.array_equal(gv, v[perm]) and np.array_equal(gs, s[perm])
    es.close()


def fitness_pattern(name, P, rng):
    """Fitness vectors that stress the selection: the tiles of 1024 consecutive rows it sorts first may be
    alike or wildly different, values may repeat across tiles, NaN / inf / signed zeros may sit anywhere."""
    if name == "random":
        f = rng.random(P, dtype=np.float32)
    elif name == "ascending":           # the best rows are all in the first tiles
        f = np.arange(P, dtype=np.float32) / P
    elif name == "descending":          # ... in the last tiles
        f = (P - np.arange(P, dtype=np.float32)) / P
    elif name == "constant":            # every key ties: the order is the index order
        f = np.full(P, 0.25, np.float32)
    elif name == "few_values":          # massive ties across tiles
        f = rng.integers(0, 5, P).astype(np.float32)
    elif name == "tile_skew":           # good and bad tiles alternate, as offspring of good and bad parent blocks do
        f = (rng.random(P, dtype=np.float32) * np.where((np.arange(P) // 1024) % 4 == 0, 0.05, 1.0)).astype(np.float32)
    elif name == "converged":           # tiny values in a narrow band + exact zeros
        f = (1e-12 * (1 + 1e-3 * rng.random(P))).astype(np.float32)
        f[rng.choice(P, P // 50, replace=False)] = 0.0
    elif name == "clones":              # a converged population: most individuals are copies of a few
        f = rng.choice(np.array([1e-13, 1.5e-13, 2e-13, 7e-10], np.float32), P, p=[0.55, 0.25, 0.15, 0.05]).astype(np.float32)
        f[rng.choice(P, P // 100, replace=False)] = rng.random(P // 100, dtype=np.float32)
    elif name == "specials":
        f = sort_case(P, rng)
        f[::97] = f[5]
        f[rng.choice(P, P // 3, replace=False)] = np.nan    # more NaN than non-selected rows can hide
        f[rng.choice(P, 50, replace=False)] = -np.inf
        f[rng.choice(P, 50, replace=False)] = -1.5
    else:
        raise ValueError(name)
    return f


SELECT_CASES = [
    # parents, offspring, kind, pattern
    (1024, 3072, 0, "random"), (2048, 6144, 1, "tile_skew"), (512, 1536, 0, "random"), (512, 1536, 2, "specials"), (288, 896, 0, "few_values"),
    (768, 2304, 3, "clones"), (1024, 1024, 0, "random"), (1056, 1024, 0, "random"), (6000 - 6000 % 32, 18000 - 18000 % 32, 0, "random"),
    (16384, 49152, 0, "random"), (16384, 49152, 0, "ascending"), (16384, 49152, 0, "descending"),
    (16384, 49152, 0, "constant"), (16384, 49152, 0, "few_values"), (16384, 49152, 0, "tile_skew"),
    (16384, 49152, 0, "converged"), (16384, 49152, 2, "specials"), (64, 65472, 0, "random"), (32768, 32768, 0, "tile_skew"),
    (8192, 24576, 3, "tile_skew"), (32768, 98304, 0, "tile_skew"), (32768, 98304, 0, "few_values"),
    (65536, 196608, 0, "random"), (40000, 110016, 0, "specials"), (16384, 49152, 0, "clones"), (32768, 98304, 0, "clones"),
]


@pytest.mark.parametrize("parents,offspring,kind,pattern", SELECT_CASES)
def test_select_places_exactly_the_rows_recombination_reads(pkg, O, parents, offspring, kind, pattern):
    """sots_stage_select (the fused loop's sortPopulation): rows 0..S-1 equal the stable full sort's bit for
    bit and NOTHING else is written (sentinel rows stay); the first reader then gets the whole order."""
    es, _ = make_pair(pkg, O, parents, offspring, kind, 9)
    rng = np.random.default_rng(parents + len(pattern))
    P, D = es.P, es.D
    S = max(parents, max(1, parents // 32) * 32)
    f = fitness_pattern(pattern, P, rng)
    v = rng.random((P, D), dtype=np.float32)
    s = rng.random((P, D), dtype=np.float32)
    sentinel_v = np.full((P, D), -7.0, np.float32)
    es.set_sort_mode(pkg.capi.SORT_TOP_ONLY)
    es.write_population(sentinel_v, sentinel_v, np.full(P, -7.0, np.float32))  # the half the selection writes into
    es.rotate()
    es.write_population(v, s, f)
    es.select(); es.rotate()
    gv, gs, gf = es.read_population()
    perm = O.sort_perm(f)
    assert np.array_equal(gf[:S], f[perm][:S], equal_nan=True)
    assert np.array_equal(gv[:S], v[perm][:S]) and np.array_equal(gs[:S], s[perm][:S])
    if 1024 < P <= 131072 and 2 * S <= P:    # up to 128 tiles of 1024 keys; outside that the population is sorted in full
        assert np.all(gf[S:] == -7.0) and np.all(gv[S:] == -7.0) and np.all(gs[S:] == -7.0), "rows beyond S were written"
    # the rest of the order on demand, from the untouched unsorted half
    es.set_sort_mode(pkg.capi.SORT_LAZY_TAIL)
    gv, gs, gf = es.read_population()
    assert np.array_equal(gf, f[perm], equal_nan=True) and np.array_equal(gv, v[perm]) and np.array_equal(gs, s[perm])
    ov, os_, of = es.read_population(other=True)
    assert np.array_equal(ov, v) and np.array_equal(os_, s) and np.array_equal(of, f, equal_nan=True)
    es.close()


def test_lazy_tail_keeps_immigrants_and_full_sort_mode_matches(pkg, O):
    """After a fused generation only the breeding rows are in place; immigrants injected into them must survive
    the completion of the order, and SOTS_SORT_FULL (the reference's every-generation full sort) must give the
    same populations as the default mode."""
    a, _ = make_pair(pkg, O, 4096, 12288, 0, 10)
    b, _ = make_pair(pkg, O, 4096, 12288, 0, 10)
    b.set_sort_mode(pkg.capi.SORT_FULL)
    tgt, _ = target_audio(O, 0, a.N)
    rng = np.random.default_rng(4)
    imm = rng.random((48, 2 * a.D + 1), dtype=np.float32)
    for es in (a, b):
        es.set_target_audio(tgt)
        es.init_population(0)
        es.timing_enable(True)
        es.execute_generations(3)
        es.inject_immigrants(imm)
    assert a.stage_time_ms(pkg.capi.STAGE_SORT_TAIL)[1] == 0, "nobody has looked at the tail yet"
    pa, pb = a.read_population(), b.read_population()
    assert a.stage_time_ms(pkg.capi.STAGE_SORT_TAIL)[1] == 1 and b.stage_time_ms(pkg.capi.STAGE_SORT_TAIL)[1] == 0
    for x, y in zip(pa, pb):
        assert np.array_equal(x, y)
    assert np.array_equal(pa[2][4096 - 48:4096], imm[:, 0]) and np.array_equal(pa[0][4096 - 48:4096], imm[:, 1:5])
    assert np.all(np.diff(pa[2][4096:]) >= 0)
    # and the runs continue identically
    for es in (a, b):
        es.execute_generations(2)
    for x, y in zip(a.read_population(), b.read_population()):
        assert np.array_equal(x, y)
    a.close(); b.close()


# the last case is large enough (>= 192 individuals per CU, 4 genes) for the fused loop to make its
# individuals inside the synthesis kernel, with a partly filled last tile
# (the last three: the cut kernels whose helper wavefronts make the genes - one group of 64 per CU, two groups, 3-op)
@pytest.mark.parametrize("kind,log2n,parents,offspring", [(0, 10, 64, 192), (1, 11, 16, 16), (3, 12, 32, 96), (2, 10, 32, 96),
                                                          (0, 9, 16416, 49152), (0, 10, 4096, 12288), (0, 10, 8192, 24576 - 32),
                                                          (1, 10, 4096 + 32, 12288), (0, 8, 64, 192), (0, 8, 4096, 12288), (3, 8, 4352, 13056),
                                                          (0, 14, 32, 96), (3, 14, 300, 724), (1, 15, 16, 48)])
def test_fused_generation_equals_staged(pkg, O, kind, log2n, parents, offspring):
    a, _ = make_pair(pkg, O, parents, offspring, kind, log2n)
    b, _ = make_pair(pkg, O, parents, offspring, kind, log2n)
    tgt, _ = target_audio(O, kind, a.N)
    for es in (a, b):
        es.set_target_audio(tgt)
        es.init_population(0)
    for _ in range(3):
        a.execute_generation()
    b.execute_generations(3)
    assert a.generation == b.generation == 3
    for x, y in zip(a.read_population(), b.read_population()):
        assert np.array_equal(x, y)
    a.close(); b.close()


@pytest.mark.parametrize("log2n,parents,offspring", [(11, 2048 + 32, 6144 + 68), (12, 2048, 6144 + 100), (13, 1024 + 32, 3072 + 68)])
def test_long_rows_several_per_wavefront(pkg, O, log2n, parents, offspring):
    """k_fft_x (N >= 2048) with more rows than resident wavefronts (every wavefront loops, the next row's loads go
    out while the current one is split) and a last workgroup that is not full: the fused generation equals the
    stage-separated one bit for bit, and rows of the materialised spectrum - the first, a middle one, the last
    ones - are the transform of the windowed audio."""
    a, _ = make_pair(pkg, O, parents, offspring, 0, log2n, block=4)  # P = 4 mod 8: no whole number of 8- or 16-row workgroups
    b, _ = make_pair(pkg, O, parents, offspring, 0, log2n, block=4)
    assert a.P % 8 == 4
    tgt, _ = target_audio(O, 0, a.N)
    for es in (a, b):
        es.set_target_audio(tgt)
        es.init_population(0)
    a.execute_generation()
    b.execute_generations(1)
    for x, y in zip(a.read_population(), b.read_population()):
        assert np.array_equal(x, y)
    # the stage-separated path left windowed audio and its spectrum behind
    n, P = a.N, a.P
    a.synthesise(); a.window(); a.fft()
    audio, spec = a.read_audio(), a.read_spectrum()
    ones = np.ones(n)
    for i in (0, 1, P // 2 + 7, P - 17, P - 2, P - 1):
        ref = O.rfft(audio[i], ones)
        got = spec[i, : n // 2 + 1].astype(np.complex128)
        assert np.abs(got - ref).max() <= FFT_TOL * max(np.abs(ref).max(), 1e-30), f"row {i}"
    a.close(); b.close()


@pytest.mark.parametrize("kind,log2n,parents,offspring", [(0, 10, 64, 192), (1, 11, 16, 16), (0, 9, 30, 35), (3, 12, 32, 96)])
def test_audio_after_fused_loop_is_the_raw_synthesis(pkg, O, kind, log2n, parents, offspring):
    """The fused loop keeps the audio in its own (tiled) layout and leaves the window to the FFT
    kernel: sots_read_synth must still return dense, un-windowed rows."""
    block = 1 if (parents + offspring) % 32 else 32
    a, _ = make_pair(pkg, O, parents, offspring, kind, log2n, block=block)
    b, _ = make_pair(pkg, O, parents, offspring, kind, log2n, block=block)
    tgt, _ = target_audio(O, kind, a.N)
    for es in (a, b):
        es.set_target_audio(tgt)
        es.init_population(0)
    a.execute_generations(1)
    b.recombine(); b.mutate(); b.synthesise()
    assert np.array_equal(a.read_audio(), b.read_audio())
    a.close(); b.close()


def test_config2_trajectory_per_generation_parity(pkg, O):
    """BASELINE config 2: P=1024 (256+768), 2-op, N=1024, 100 generations.  Each
    generation the oracle is re-synchronised to the device's pre-generation state, runs the same
    generation, and both results must agree stage by stage."""
    es, ref = make_pair(pkg, O, 256, 768, 0, 10)
    tgt, _ = target_audio(O, 0, es.N)
    es.set_target_audio(tgt)
    ref.set_target_audio(tgt)
    es.init_population(0)
    atol = fit_atol(O, tgt)
    best = []
    for gen in range(100):
        v, s, f = es.read_population()
        ref.write_population(v, s, f)
        ref.set_generation(gen)
        es.recombine(); es.mutate()
        ref.recombine(); ref.mutate()
        gv, gs, _ = es.read_population()
        rv, rs, _ = ref.read_population()
        assert np.array_equal(gv, rv), f"gen {gen}"
        np.testing.assert_allclose(gs, rs, rtol=2e-6, atol=0)
        ref.write_population(gv, gs, None)  # carry the device's steps so values stay bit-exact
        es.synthesise(); es.window(); es.fft(); es.fitness()
        ref.evaluate()
        gf = es.read_fitness()
        _, _, rf = ref.read_population()
        np.testing.assert_allclose(gf, rf, rtol=FIT_RTOL, atol=atol)
        es.sort(); es.rotate()
        sv, ss, sf = es.read_population()
        perm = O.sort_perm(gf)
        assert np.array_equal(sf, gf[perm]) and np.array_equal(sv, gv[perm]) and np.array_equal(ss, gs[perm])
        best.append(float(sf[0]))
    assert es.generation == 100
    # (mu + lambda)-style selection on a re-evaluated population is not monotone, but the search
    # must make progress on this easy target
    assert min(best) < best[0]
    es.close()


def full_size_properties(pkg, O, parents, offspring, kind, log2n):
    """Size-independent properties of one island at a BASELINE shard size: the fused loop equals the
    stage-separated one bit for bit, the result is sorted, sorting is a permutation (checksum of
    checksums), a planted optimum comes out first, 64 random rows match the CPU oracle."""
    es, _ = make_pair(pkg, O, parents, offspring, kind, log2n)
    twin, _ = make_pair(pkg, O, parents, offspring, kind, log2n)
    tgt, tv = target_audio(O, kind, es.N)
    for e in (es, twin):
        e.set_target_audio(tgt)
        e.init_population(0)
    es.execute_generations(2)
    twin.execute_generation(); twin.execute_generation()
    v, s, f = es.read_population()
    for x, y in zip((v, s, f), twin.read_population()):
        assert np.array_equal(x, y), "fused loop differs from the stage-separated one"
    twin.close()
    assert np.all(np.diff(f) >= 0), "population not sorted by fitness"
    assert np.all(np.isfinite(f)) and f[0] >= 0
    # one more generation staged, with a planted perfect individual: it must come out first
    es.recombine(); es.mutate()
    v, s, _ = es.read_population()
    plant = es.P // 2 + 1234
    v[plant] = tv
    es.write_population(v, s, None)
    es.synthesise(); es.window(); es.fft(); es.fitness()
    fu = es.read_fitness()
    es.sort(); es.rotate()
    v2, s2, f2 = es.read_population()
    atol = fit_atol(O, tgt)
    assert f2[0] <= atol and np.array_equal(v2[0], v[plant])
    # sorted output is a permutation of the input rows (checksum of checksums)
    assert np.array_equal(np.sort(fu), f2)
    w = np.array([1.0, 3.0, 7.0, 11.0, 13.0, 17.0, 19.0, 23.0, 29.0, 31.0, 37.0, 41.0])[: es.D]
    assert np.array_equal(np.sort(v.astype(np.float64) @ w), np.sort(v2.astype(np.float64) @ w))
    assert np.array_equal(np.sort(s.astype(np.float64) @ w), np.sort(s2.astype(np.float64) @ w))
    # spot-check 64 random rows of the full-size evaluation against the oracle
    rng = np.random.default_rng(5)
    rows = rng.choice(es.P, 64, replace=False)
    tgt_mag = O.spectrum(tgt)
    for r in rows:
        a = O.synth(kind, v[r], [0.0] * es.D, PMAX[kind], es.N)
        want = O.fitness(O.spectrum(a), tgt_mag)
        assert abs(fu[r] - want) <= FIT_RTOL * want + atol, f"row {r}: {fu[r]} vs {want}"
    es.close()


@pytest.mark.parametrize("parents,offspring,log2n", [(8192, 24576, 10), (8160, 24512, 9), (4128, 12352, 10)])
def test_synthesise_four_op_one_wavefront_per_operator_bitexact(pkg, O, parents, offspring, log2n):
    """65 ... 128 individuals per CU (BASELINE configs[3]'s shard: 32768 on 256 CUs): the 4-operator chain is cut in
    front of every operator - k_synth<4OP, 1, false, 2, 3>: four wavefronts per 64 individuals, eight per CU, three
    hand-over links of 8-sample blocks, 16-sample tiles that leave as half lines.  Rows are spot-checked bit for bit
    against the oracle, the partly filled last tile (P = 32672: 32 rows in the last workgroup) and the
    second group of a workgroup included; the variation folded into the same launch gives the same audio."""
    kind = 3
    es, _ = make_pair(pkg, O, parents, offspring, kind, log2n)
    es.init_population(0)
    v, s, _ = es.read_population()
    _, tv = target_audio(O, kind, es.N)
    v[0], v[1], v[2] = tv, 0.0, 1.0
    es.write_population(v, s, None)
    es.synthesise()
    audio = es.read_audio()
    rng = np.random.default_rng(23)
    rows = np.unique(np.concatenate([[0, 1, 2, 63, 64, 65, 127, 128, 191, 192, es.P - 33, es.P - 32, es.P - 1],
                                     rng.choice(es.P, 100, replace=False)]))
    for r in rows:
        assert np.array_equal(audio[r], O.synth(kind, v[r], [0.0] * es.D, PMAX[kind], es.N)), f"row {r}"
    es.close()


@pytest.mark.parametrize("parents,offspring,log2n", [(1024, 3072, 10), (4096 - 32, 12288, 9), (16, 16, 11)])
def test_synthesise_triple_voice_one_wavefront_per_chain_bitexact(pkg, O, parents, offspring, log2n):
    """Up to 64 individuals per CU the voice of three parallel 2-operator chains runs a wavefront per chain
    (k_synth<TRIPLE, -1>): chains 1 and 2 hand their gain * table products to the wavefront of chain 0, which adds them in
    the reference's order (Evolutionary_Strategy.hpp:493).  Spot rows bit for bit against the oracle, partly filled tiles
    included."""
    kind = 2
    es, _ = make_pair(pkg, O, parents, offspring, kind, log2n)
    es.init_population(0)
    v, s, _ = es.read_population()
    _, tv = target_audio(O, kind, es.N)
    v[0], v[1], v[2] = tv, 0.0, 1.0
    es.write_population(v, s, None)
    es.synthesise()
    audio = es.read_audio()
    rng = np.random.default_rng(29)
    rows = np.unique(np.concatenate([[0, 1, 2, min(63, es.P - 1), es.P - 1], rng.choice(es.P, min(es.P, 80), replace=False)]))
    for r in rows:
        assert np.array_equal(audio[r], O.synth(kind, v[r], [0.0] * 4, PMAX[kind], es.N)), f"row {r}"
    es.close()


def test_full_size_properties_config2(pkg, O):
    """BASELINE configs[2]: P = 65536 (16384 + 49152), 2-op, N = 1024, one GPU."""
    full_size_properties(pkg, O, 16384, 49152, 0, 10)


def test_full_size_properties_config3_shard(pkg, O):
    """BASELINE configs[3], the per-GPU shard of the 8-GPU run: P = 32768 (8192 + 24576), 4-op series,
    N = 4096 - the regime where k_synth<3,2> is cut into two wavefronts and the transform is k_fft_x<12>."""
    full_size_properties(pkg, O, 8192, 24576, 3, 12)


def test_full_size_properties_config4_shard(pkg, O):
    """BASELINE configs[4], the per-GPU shard of the 8-GPU run: P = 131072 (32768 + 98304), 2-op, N = 1024."""
    full_size_properties(pkg, O, 32768, 98304, 0, 10)


def test_full_size_properties_ragged_rows_dealt_inside_the_workgroup(pkg, O):
    """From four rows per wavefront k_fft runs as ONE workgroup of twelve wavefronts per CU whose rows are dealt out as the
    wavefronts ask for them (an LDS counter): a population that is a multiple of the recombination block only (12320 rows: the workgroups own
    48 or 49 rows, the last takes run past the end) - every row evaluated exactly once, 64 of them against the oracle."""
    full_size_properties(pkg, O, 3104, 9216, 0, 10)


def test_island_rows_roundtrip(pkg, O):
    es, ref = make_pair(pkg, O, 64, 192, 0, 10)
    rng = np.random.default_rng(9)
    v = rng.random((es.P, es.D), dtype=np.float32)
    s = rng.random((es.P, es.D), dtype=np.float32)
    f = np.sort(rng.random(es.P, dtype=np.float32))
    es.write_population(v, s, f)
    ref.write_population(v, s, f)
    rows = es.pack_elites(16)
    assert np.array_equal(rows, ref.pack_elites(16))
    imm = rng.random((48, 2 * es.D + 1), dtype=np.float32)
    es.inject_immigrants(imm)
    ref.inject(imm)
    for x, y in zip(es.read_population(), ref.read_population()):
        assert np.array_equal(x, y)
    # the gathered form: world 4, this island is rank 2 -> blocks 0, 1, 3 are injected
    import torch
    allrows = rng.random((4 * 16, 2 * es.D + 1), dtype=np.float32)
    dev = torch.from_numpy(allrows).cuda()
    es.inject_gathered_device(dev.data_ptr(), 4, 2, 16)
    es.synchronize()
    ref.inject(np.concatenate([allrows[:32], allrows[48:]]))
    for x, y in zip(es.read_population(), ref.read_population()):
        assert np.array_equal(x, y)
    es.close()


def test_stage_timers_and_errors(pkg, O):
    es, _ = make_pair(pkg, O, 64, 192, 0, 10)
    with pytest.raises(pkg.SotsError) as e:
        es.execute_generation()     # no target yet
    assert e.value.code == -5
    tgt, _ = target_audio(O, 0, es.N)
    es.set_target_audio(tgt)
    es.init_population(0)
    es.timing_enable(True)
    es.execute_generation()
    es.execute_generations(2)
    for st in (pkg.capi.STAGE_RECOMBINE, pkg.capi.STAGE_SYNTHESISE, pkg.capi.STAGE_FFT, pkg.capi.STAGE_FITNESS):
        ms, cnt = es.stage_time_ms(st)
        assert cnt == 1 and ms > 0
    ms, cnt = es.stage_time_ms(pkg.capi.STAGE_SORT)
    assert cnt == 3
    ms, cnt = es.stage_time_ms(pkg.capi.STAGE_FUSED_SPECTRAL)
    assert cnt == 2 and ms > 0
    with pytest.raises(pkg.SotsError) as e:
        es.set_target_spectrum(np.zeros(7, np.float32))
    assert e.value.code == -4
    es.close()
