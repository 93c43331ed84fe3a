#!/usr/bin/env python3
"""Times the REFERENCE's own device kernels on this GPU at BASELINE configs[2] (P = 16384 + 49152, 2-op, N = 1024), launched
as the reference's host launches them (global size P, workgroup size 32 - parameters.json:31; the window kernel: global size N,
Evolutionary_Strategy_OpenCL.hpp:471-533), and the product's generation beside them.  The reference publishes no numbers
(BASELINE.md); this is what its kernels take on an MI355X, WITHOUT its FFT (clFFT is not in the image: the reference's
generation is at least the sum below).

A measurement script, not a test (it lives under tests/ because it uses oracle/_ref, which only tests may):
  gpurun -- 'python tests/ref_kernels_time.py gpurun_out/r04_reference_kernels.json'   -> profiles/
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ocl_ref as R  # noqa: E402
from oracle import oracle as O  # noqa: E402

TAG, WG, D, LOG2N, PARENTS, OFFSPRING = "2op_n1024_p65536_wg32", 32, 4, 10, 16384, 49152


def main():
    dst = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04_reference_kernels.json"
    import importlib
    pkg = importlib.import_module("survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd")
    p, n = PARENTS + OFFSPRING, 1 << LOG2N
    rng = np.random.default_rng(1)
    out = {"config": "BASELINE configs[2]: P = %d + %d, 2-op, N = %d" % (PARENTS, OFFSPRING, n), "workgroup_size": WG, "kernels_us": {}}
    for flavour in ("asrun",):
        prog = R.RefProgram(TAG, flavour)
        rot = R.DeviceBuffer(np.zeros(1, np.uint32))
        values = R.DeviceBuffer(rng.random((2, p, D), dtype=np.float32))
        steps = R.DeviceBuffer(np.full((2, p, D), 0.1, np.float32))
        fitness = R.DeviceBuffer(rng.random((2, p), dtype=np.float32))
        states = R.DeviceBuffer(rng.integers(1, 2 ** 32 - 1, size=(p, 2), dtype=np.uint64).astype(np.uint32))
        audio = R.DeviceBuffer(nbytes=p * n * 4)
        spectrum = R.DeviceBuffer(nbytes=p * (n + 8) * 4)
        target = R.DeviceBuffer(np.zeros(n // 2 + 8, np.float32))
        pmin, pmax = R.DeviceBuffer(np.zeros(D, np.float32)), R.DeviceBuffer(np.array([3520, 8, 3520, 1], np.float32))
        table = R.DeviceBuffer(np.concatenate([O.wavetable(), np.zeros(64, np.float32)]))
        k = out["kernels_us"]
        k["recombinePopulation"] = 1e3 * prog.time_ms("recombinePopulation", p, WG, [values, steps, rot])
        k["mutatePopulation"] = 1e3 * prog.time_ms("mutatePopulation", p, WG, [values, steps, states, rot])
        # (mutation leaves the values outside [0, 1] after a few rounds: fresh ones for the synthesis, as a generation would have)
        values.free()
        values = R.DeviceBuffer(rng.random((2, p, D), dtype=np.float32))
        k["synthesisePopulation"] = 1e3 * prog.time_ms("synthesisePopulation", p, WG, [audio, values, pmin, pmax, rot, table])
        k["applyWindowPopulation"] = 1e3 * prog.time_ms("applyWindowPopulation", n, WG, [audio], launches=2)
        k["fitnessPopulation"] = 1e3 * prog.time_ms("fitnessPopulation", p, WG, [fitness, spectrum, target, rot])
        fitness.free()
        fitness = R.DeviceBuffer(rng.random((2, p), dtype=np.float32))
        k["sortPopulation"] = 1e3 * prog.time_ms("sortPopulation", p, WG, [values, steps, fitness, rot], launches=2)
        for b in (rot, values, steps, fitness, states, audio, spectrum, target, pmin, pmax, table):
            b.free()
        prog.unload()
    out["reference_generation_us_without_fft"] = sum(out["kernels_us"].values())
    # the product's generation on the same box, un-instrumented
    es = pkg.HipES(PARENTS, OFFSPRING, synth_kind=0, audio_log2=LOG2N, param_max=[3520.0, 8.0, 3520.0, 1.0])
    tgt = O.synth(0, [1450.0 / 3520.0, 3.0 / 8.0, 200.0 / 3520.0, 1.0], [0.0] * 4, [3520.0, 8.0, 3520.0, 1.0], n)
    es.set_target_audio(tgt)
    es.init_population(0)
    es.execute_generations(50)
    es.synchronize()
    t0 = time.perf_counter()
    es.execute_generations(400)
    es.synchronize()
    out["product_generation_us_with_fft"] = 1e6 * (time.perf_counter() - t0) / 400
    es.close()
    out["ratio"] = out["reference_generation_us_without_fft"] / out["product_generation_us_with_fft"]
    out["note"] = ("the reference's kernels (kernels/ocl_program.cl compiled as it stands, six-decimal macros as its host passes them) by HIP events, "
                   "mean of 2-3 launches after a warm-up; its FFT (clFFT) and its eight host synchronisations per generation are not in the sum")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
