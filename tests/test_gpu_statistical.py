"""Whole-run (statistical) parity: a FREE-RUNNING HIP evolution against a FREE-RUNNING CPU-oracle
evolution (SURVEY 4 last paragraph; Evolutionary_Strategy_CPU.hpp:353-418 is the loop restated).

Stage-by-stage parity (test_gpu_parity.py) re-synchronises the oracle every generation.  Left
alone, the two runs cannot stay on one trajectory: mutation step sizes differ by an ulp or two
(device expf/powf against libm), fitness by up to 1e-4 relative (fp32 against fp64 transform),
near-ties then sort differently, and the search is chaotic - a CPU experiment that multiplies the
ORACLE's own steps by 1 +- 1e-7 sends 72 % of the seeds to a different trajectory by generation 10.
What must agree is the distribution of outcomes over seeds.  BASELINE configs[1] size: P = 1024
(256 + 768), 2-op, N = 1024, the reference's self-match target (ocl_program.cl:247-250),
SEEDS seeds x 100 generations.  At this population size most runs end in one of three local optima
(SSE 0.135, 0.108, 0.0708) and about a fifth reach the target, in both implementations.

Stated tolerances, calibrated on oracle-vs-perturbed-oracle runs where the same
statistics came out at 0.035 dex, 0.28 and 0.03:
  * generation 1, before any divergence: every seed's best fitness within 2e-4 relative;
  * generations 10, 50, 100: medians of log10(best fitness) within MEDIAN_DEX = 0.15 (a factor 1.41);
    two-sample Kolmogorov-Smirnov distance of log10(best fitness) <= KS_MAX = 0.41 (the alpha = 0.01
    critical value for 32 + 32 samples is 0.407);
  * generation 100: the fractions of seeds below 1e-3 (converged to the target's basin) within 6/32.
The second test below runs the same comparison against the REFERENCE'S OWN generation loop (its kernels, compiled as
they stand; numpy in place of clFFT)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PMAX = [3520.0, 8.0, 3520.0, 1.0]
SEEDS, GENS = 32, 100
CHECK = (10, 50, 100)
MEDIAN_DEX, KS_MAX, FRACTION_TOL = 0.15, 0.41, 6 / 32


def ks_distance(a, b):
    pts = np.sort(np.concatenate([a, b]))
    return max(abs((a <= x).mean() - (b <= x).mean()) for x in pts)


def test_free_running_hip_and_oracle_agree_in_distribution(pkg, O):
    target = O.synth(0, [1450 / 3520, 3 / 8, 200 / 3520, 1.0], [0.0] * 4, PMAX, 1024)
    marks = (1,) + CHECK
    hip = np.zeros((SEEDS, len(marks)))
    cpu = np.zeros((SEEDS, len(marks)))
    try:
        O.set_threads(16)  # the oracle's evaluation loop; results do not depend on the thread count
        for k in range(SEEDS):
            seed = 0x5EED0001 + k
            es = pkg.HipES(256, 768, pkg.capi.SYNTH_2OP, 10, None, PMAX, seed=seed, workgroup_size=32)
            ref = O.OracleES(256, 768, O.SYNTH_2OP, 10, None, PMAX, seed=seed, recomb_block=32)
            es.set_target_audio(target)
            ref.set_target_audio(target)
            es.init_population(0)
            ref.init_population(0)
            done = 0
            for j, g in enumerate(marks):
                es.execute_generations(g - done)          # free-running: nothing is copied across
                for _ in range(g - done):
                    ref.generation()
                done = g
                hip[k, j] = es.read_fitness()[0]
                cpu[k, j] = ref.read_population()[2][0]
            es.close()
    finally:
        O.set_threads(1)
    # generation 1: one variation + evaluation from the identical initial population
    np.testing.assert_allclose(hip[:, 0], cpu[:, 0], rtol=2e-4)
    lh, lc = np.log10(hip + 1e-30), np.log10(cpu + 1e-30)
    report = []
    for j, g in enumerate(marks[1:], start=1):
        dm = abs(np.median(lh[:, j]) - np.median(lc[:, j]))
        ks = ks_distance(lh[:, j], lc[:, j])
        report.append(f"gen {g}: median log10 best HIP {np.median(lh[:, j]):.3f} oracle {np.median(lc[:, j]):.3f}, KS {ks:.3f}")
        assert dm <= MEDIAN_DEX, report[-1]
        assert ks <= KS_MAX, report[-1]
    frac_h, frac_c = (hip[:, -1] < 1e-3).mean(), (cpu[:, -1] < 1e-3).mean()
    report.append(f"fraction of seeds below 1e-3 at generation {GENS}: HIP {frac_h:.3f} oracle {frac_c:.3f}")
    print("\n".join(report))
    assert abs(frac_h - frac_c) <= FRACTION_TOL, report[-1]
    # the search works: the median improves, and some seed of each implementation finds the target
    assert np.median(lh[:, -1]) < np.median(lh[:, 0]) - 0.3 and np.median(lc[:, -1]) < np.median(lc[:, 0]) - 0.3
    assert hip[:, -1].min() < 1e-3 and cpu[:, -1].min() < 1e-3


def test_free_running_hip_and_the_reference_kernels_agree_in_distribution(pkg, O):
    """The same statistics against the REFERENCE'S OWN generation loop: its kernels (kernels/ocl_program.cl compiled as it
    stands, oracle/build_ref_ocl.py) launched in its host's order through the HIP module API, numpy standing in for clFFT
    (tests/_ocl_ref.py RefGenerationLoop).  The two runs share nothing but the problem: the reference draws from MWC64X
    states (seeded here per run; its host seeds them from the clock), recombines in place (offspring blocks may read
    parents that are already overwritten), windows with an fp32 cosine and advances phases with a double ratio
    (tests/test_ocl_reference.py has each of these measured)."""
    import os
    import _ocl_ref as R
    tag = "2op_n1024_p1024_wg32"
    if not os.path.exists(R.code_object(tag, "exact")):
        pytest.skip("oracle/_ref holds no code objects (python oracle/build_ref_ocl.py needs /root/reference)")
    target = O.synth(0, [1450 / 3520, 3 / 8, 200 / 3520, 1.0], [0.0] * 4, PMAX, 1024)
    marks = CHECK
    hip = np.zeros((SEEDS, len(marks)))
    ref = np.zeros((SEEDS, len(marks)))
    loop = R.RefGenerationLoop(tag, "exact", 32, 4, 10, 256, 768, [0.0] * 4, PMAX, O.wavetable(), O.spectrum(target))
    try:
        for k in range(SEEDS):
            es = pkg.HipES(256, 768, pkg.capi.SYNTH_2OP, 10, None, PMAX, seed=0x0C10000 + k, workgroup_size=32)
            es.set_target_audio(target)
            es.init_population(0)
            loop.init(np.random.default_rng(977 + k).integers(1, 2 ** 32 - 1, size=(1024, 2), dtype=np.uint64).astype(np.uint32))
            done = 0
            for j, g in enumerate(marks):
                es.execute_generations(g - done)
                for _ in range(g - done):
                    loop.generation()
                done = g
                hip[k, j] = es.read_fitness()[0]
                ref[k, j] = loop.best_fitness()
            es.close()
    finally:
        loop.close()
    lh, lr = np.log10(hip + 1e-30), np.log10(ref + 1e-30)
    report = []
    for j, g in enumerate(marks):
        dm = abs(np.median(lh[:, j]) - np.median(lr[:, j]))
        ks = ks_distance(lh[:, j], lr[:, j])
        report.append(f"gen {g}: median log10 best HIP {np.median(lh[:, j]):.3f} reference kernels {np.median(lr[:, j]):.3f}, KS {ks:.3f}")
        # the reference's in-place recombination makes ITS runs differ from one execution to the next, so this is a
        # two-sample test proper: KS at alpha = 0.001 (0.49 for 32 + 32 samples; measured 0.125 / 0.25 / 0.25), and the
        # medians within the gap between two neighbouring local optima (0.18 dex; measured 0.002 / 0.11 / 0.09)
        assert ks <= 0.49, report[-1]
        assert dm <= 0.25, report[-1]
    frac_h, frac_r = (hip[:, -1] < 1e-3).mean(), (ref[:, -1] < 1e-3).mean()
    report.append(f"fraction of seeds below 1e-3 at generation {GENS}: HIP {frac_h:.3f} reference kernels {frac_r:.3f}")
    print("\n".join(report))
    assert abs(frac_h - frac_r) <= 8 / 32, report[-1]
    # both end in the same places: the target's basin or one of the three local optima the oracle's runs find too
    # (SSE 0.0708, 0.1078, 0.1349) - measured: 9 / 6 / 9 / 6 HIP runs, 8 / 15 / 3 / 4 reference runs of 32 (chi-square 7.3 on 3 degrees of freedom, p = 0.06)
    for runs in (hip[:, -1], ref[:, -1]):
        known = (runs < 1e-3) | (np.abs(runs - 0.0708) < 2e-3) | (np.abs(runs - 0.1078) < 2e-3) | (np.abs(runs - 0.1349) < 2e-3)
        assert known.mean() >= 0.75
    assert hip[:, -1].min() < 1e-3 and ref[:, -1].min() < 1e-3
