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
Parity stays "unpinned" in the sense of DESIGN.md 6 (the oracle is this repository's restatement)."""
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
