#!/usr/bin/env python3
"""Golden vectors from the REFERENCE's own kernels: runs kernels/ocl_program.cl - compiled as it stands by
oracle/build_ref_ocl.py (build container) - on an MI355X through the HIP module API (tests/_ocl_ref.py) and writes
inputs and outputs to tests/golden/ocl_ref_v1.npz.

Run on a GPU box:  gpurun -- 'python tests/golden/make_ocl_golden.py gpurun_out/ocl_ref_v1.npz'  and copy the file here.
Nothing of the reference travels: the code objects are what the build container compiled; this script holds inputs.

Per configuration (oracle/build_ref_ocl.py CONFIGS); the exact flavour keeps everything, the asrun flavour (six-decimal
macro values, as the reference's host formats them) only the outputs those macros change - mutate, window, fitness:
  recombine   random values/steps -> recombinePopulation (three runs must agree: the kernel reads parent blocks that
              other workgroups overwrite in place)
  init        MWC64X states -> initPopulation values, steps, advanced states
  mutate      random values, steps 0.1..0.3, MWC64X states -> mutatePopulation
  synth       parameter rows (reference KAT row, corners, random) -> the configuration's synthesis kernel, first rows kept
  window      rows of ones and of noise -> applyWindowPopulation
  fitness     random complex spectra (bins >= N/2 zero), target -> fitnessPopulation
  sort        tie-free random fitness + rows -> sortPopulation (other rotation half)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _ocl_ref as R  # noqa: E402
from oracle import build_ref_ocl as B  # noqa: E402
from oracle import oracle as O  # noqa: E402

SYNTH_KERNEL = {4: "synthesisePopulation", 6: "synthesisePopulationDoubleSeries", 12: "synthesisePopulationTripleParallel"}
PMAX = {4: [3520, 8, 3520, 1], 6: [3520, 8, 3520, 8, 8, 8], 12: [3520, 8, 3520, 1] * 3}
W = 32768


def synth_rows(d, rows, rng):
    """parameter rows in [0, 1]: the reference's debug constants (ocl_program.cl:247-250), corners, random.
    3-op: values[4] == values[5] with equal maxima, so that the OpenCL kernel's offset params[4] (ocl_program.cl:368)
    and the CPU path's params[5] (Evolutionary_Strategy.hpp:430) are the same number."""
    v = rng.random((rows, d), dtype=np.float32)
    kat = np.array([0.411931818, 0.375, 0.0568181818, 1.0], dtype=np.float32)
    v[0, :4] = kat
    if d == 12:
        v[0, 4:8] = np.array([0.69602272727, 0.375, 0.0568181818, 1.0], dtype=np.float32)
        v[0, 8:12] = np.array([0.98011363636, 0.375, 0.0568181818, 1.0], dtype=np.float32)
    v[1] = 0.0
    v[2] = 1.0
    v[3] = 0.5
    if d == 6:
        v[:, 5] = v[:, 4]
    return v


def run_config(tag, wg, d, log2n, parents, offspring, flavour, out, table):
    n, p = 1 << log2n, parents + offspring
    key = "%s/%s/" % (tag, flavour)
    rng = np.random.default_rng(int.from_bytes(tag.encode(), "little") % (2 ** 31))  # same inputs for both flavours
    full = flavour == "exact"  # the asrun flavour keeps only the kernels its six-decimal macros change: mutate, window, fitness
    def put(name, array, always=False):
        if full or always:
            out[key + name] = np.ascontiguousarray(array)

    prog = R.RefProgram(tag, flavour)
    rot0 = R.DeviceBuffer(np.zeros(1, np.uint32))
    try:
        # ---- recombine -------------------------------------------------------------------------------------
        vin = rng.random((2, p, d), dtype=np.float32)
        sin = rng.random((2, p, d), dtype=np.float32)
        runs = []
        for _ in range(3):
            bv, bs = R.DeviceBuffer(vin), R.DeviceBuffer(sin)
            prog.launch("recombinePopulation", p, wg, [bv, bs, rot0])
            runs.append((bv.read(np.float32, (2, p, d)), bs.read(np.float32, (2, p, d))))
            bv.free(), bs.free()
        put("recombine_deterministic", np.array([all(np.array_equal(runs[0][0], r[0]) and np.array_equal(runs[0][1], r[1]) for r in runs[1:])]))
        put("recombine_in_values", vin[0]), put("recombine_in_steps", sin[0])
        put("recombine_out_values", runs[0][0][0]), put("recombine_out_steps", runs[0][1][0])
        assert np.array_equal(runs[0][0][1], vin[1]) and np.array_equal(runs[0][1][1], sin[1]), "rotation half 1 was touched"

        # ---- init + mutate (MWC64X states in, advanced states out) -----------------------------------------
        states = rng.integers(1, 2 ** 32 - 1, size=(p, 2), dtype=np.uint64).astype(np.uint32)
        bst = R.DeviceBuffer(states)
        bv, bs, bf = R.DeviceBuffer(np.zeros((2, p, d), np.float32)), R.DeviceBuffer(np.zeros((2, p, d), np.float32)), R.DeviceBuffer(np.zeros((2, p), np.float32))
        prog.launch("initPopulation", p, wg, [bv, bs, bf, bst, rot0])
        put("init_states", states)
        put("init_values", bv.read(np.float32, (2, p, d))[0]), put("init_steps", bs.read(np.float32, (2, p, d))[0])
        put("init_states_after", bst.read(np.uint32, (p, 2)))
        bv.free(), bs.free(), bf.free(), bst.free()
        mv = rng.random((2, p, d), dtype=np.float32)
        mv[0, : p // 8] *= 0.02            # near the lower edge: the reflect branch (ocl_program.cl:176-182) fires
        mv[0, p // 8: p // 4] = 1.0 - mv[0, p // 8: p // 4] * 0.02
        ms = (0.1 + 0.2 * rng.random((2, p, d), dtype=np.float32)).astype(np.float32)
        states2 = rng.integers(1, 2 ** 32 - 1, size=(p, 2), dtype=np.uint64).astype(np.uint32)
        bv, bs, bst = R.DeviceBuffer(mv), R.DeviceBuffer(ms), R.DeviceBuffer(states2)
        prog.launch("mutatePopulation", p, wg, [bv, bs, bst, rot0])
        put("mutate_states", states2)   # (inputs are the same for both flavours: kept once)
        put("mutate_in_values", mv[0]), put("mutate_in_steps", ms[0])
        put("mutate_out_values", bv.read(np.float32, (2, p, d))[0], True), put("mutate_out_steps", bs.read(np.float32, (2, p, d))[0], True)
        put("mutate_states_after", bst.read(np.uint32, (p, 2)))
        bv.free(), bs.free(), bst.free()

        # ---- synthesis: the first `rows` individuals (a launch of `rows` work-items) -------------------------
        rows, kept = wg, (8 if n > 1024 else 24)  # individuals launched / rows of audio kept
        pv = np.zeros((2, p, d), np.float32)
        pv[0, :rows] = synth_rows(d, rows, rng)
        pmin, pmax = np.zeros(d, np.float32), np.array(PMAX[d], np.float32)
        baud = R.DeviceBuffer(nbytes=p * n * 4)
        bv, bmin, bmax = R.DeviceBuffer(pv), R.DeviceBuffer(pmin), R.DeviceBuffer(pmax)
        btab = R.DeviceBuffer(np.concatenate([table, np.zeros(64, np.float32)]))  # (uint)pos can reach W: DESIGN 6, deviation 8
        prog.launch(SYNTH_KERNEL[d], rows, wg, [baud, bv, bmin, bmax, rot0, btab])
        put("synth_values", pv[0, :kept]), put("synth_pmin", pmin), put("synth_pmax", pmax)
        put("synth_audio", baud.read(np.float32, (p, n))[:kept])
        bv.free(), bmin.free(), bmax.free(), btab.free()

        # ---- window: row 0 all ones (the window itself), rows 1..2 noise; the kernel walks POPULATION_COUNT rows ----
        aud = np.zeros((p, n), np.float32)
        aud[0] = 1.0
        aud[1:3] = rng.standard_normal((2, n), dtype=np.float32)
        baud2 = R.DeviceBuffer(aud)
        prog.launch("applyWindowPopulation", n, wg, [baud2])
        put("window_in", aud[:3]), put("window_out", baud2.read(np.float32, (p, n))[:3], True)
        baud.free(), baud2.free()

        # ---- fitness on materialised spectra: [rows][N + 8] interleaved complex, bins >= N/2 zero --------------
        frows, fkept = wg, 8
        spec = np.zeros((p, n + 8), np.float32)
        spec[:frows, :n] = (rng.standard_normal((frows, n), dtype=np.float32) * np.float32(0.25 * n)).astype(np.float32)
        target = np.zeros(n // 2 + 8, np.float32)
        target[: n // 2] = np.abs(rng.standard_normal(n // 2, dtype=np.float32)) * np.float32(0.3)
        bsp, btg, bfit = R.DeviceBuffer(spec), R.DeviceBuffer(target), R.DeviceBuffer(np.zeros((2, p), np.float32))
        prog.launch("fitnessPopulation", frows, wg, [bfit, bsp, btg, rot0])
        put("fitness_spectrum", spec[:fkept]), put("fitness_target", target[: n // 2])
        put("fitness_out", bfit.read(np.float32, (2, p))[0, :fkept], True)
        bsp.free(), btg.free(), bfit.free()

        # ---- sort: tie-free fitness -------------------------------------------------------------------------
        fit = np.zeros((2, p), np.float32)
        fit[0] = rng.permutation(p).astype(np.float32) * np.float32(0.37) + rng.random(p, dtype=np.float32) * np.float32(0.1)
        assert len(np.unique(fit[0])) == p
        sv, ss = rng.random((2, p, d), dtype=np.float32), rng.random((2, p, d), dtype=np.float32)
        bv, bs, bf = R.DeviceBuffer(sv), R.DeviceBuffer(ss), R.DeviceBuffer(fit)
        prog.launch("sortPopulation", p, wg, [bv, bs, bf, rot0])
        put("sort_in_values", sv[0]), put("sort_in_steps", ss[0]), put("sort_in_fitness", fit[0])
        put("sort_out_values", bv.read(np.float32, (2, p, d))[1]), put("sort_out_steps", bs.read(np.float32, (2, p, d))[1])
        put("sort_out_fitness", bf.read(np.float32, (2, p))[1])
        bv.free(), bs.free(), bf.free()
    finally:
        rot0.free()
        prog.unload()


def main():
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "ocl_ref_v1.npz")
    table = O.wavetable()
    out = {}
    for tag, wg, d, log2n, parents, offspring in B.CONFIGS:
        for flavour in ("asrun", "exact"):
            run_config(tag, wg, d, log2n, parents, offspring, flavour, out, table)
            print("ran", tag, flavour, flush=True)
            out["%s/%s/macros" % (tag, flavour)] = np.array([" ".join("%s=%s" % kv for kv in B.macros(wg, d, log2n, parents, offspring, flavour))])
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
