"""Diagnostic: start and end of every wavefront of the fused spectral kernel k_fft (N <= 1024) on the chip-wide 100 MHz clock and
the rows each took, grouped by the wavefront's age on its SIMD (needs a -DSOTS_STAMP_ENDS build: light stamps, product registers).
usage: SOTS_LIB_PATH=variants/ends.so python tools/fft_ends_probe.py [P]"""
import ctypes as C, importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd")
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pmax, tp = bench.VOICES["2op"]
es = pkg.HipES(P // 4, P - P // 4, pkg.capi.SYNTH_2OP, 10, None, pmax, seed=1)
es.set_target_audio(bench.make_target(pkg, "2op", 10, 0))
es.init_population()
es.execute_generations(300)
es.synchronize()
L = es.L
L.sots_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
L.sots_debug_clear_stamps()
es.execute_generations(1); es.synchronize()
buf = (C.c_ulonglong * (2 * 16384))()
L.sots_debug_stamps(buf, 2 * 16384)
a = np.frombuffer(buf, dtype=np.uint64)[:16384].reshape(-1, 4)
a = a[a[:, 2] > 0]
b, e, rows, hw = a[:, 0].astype(np.float64), a[:, 1].astype(np.float64), a[:, 2].astype(np.float64), a[:, 3]
t0 = b.min()
q = lambda v: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(v, [0, 10, 50, 90, 100]))
print(f"P={P}: {len(a)} wavefronts, {rows.sum():.0f} rows")
print("  first row starts (us)", q((b - t0) / 100))
print("  end (us)             ", q((e - t0) / 100))
print("  rows                 ", q(rows))
simd = (hw >> 4) & 3
W = 12
if len(a) % W == 0:
    by = lambda v: np.array2string(np.median(v.reshape(-1, W), axis=0), precision=1, floatmode="fixed", max_line_width=200)
    print("  by wavefront of the workgroup (median over workgroups):")
    print("   SIMD     ", by(simd.astype(np.float64)))
    print("   rows     ", by(rows))
    print("   end (us) ", by((e - t0) / 100))
    wg_end = ((e - t0) / 100).reshape(-1, W).max(axis=1)
    print("  workgroup ends (us)  ", q(wg_end))
    wg_start = ((b - t0) / 100).reshape(-1, W).min(axis=1)
    print("  workgroup starts (us)", q(wg_start), " correlation with its end: %.2f" % np.corrcoef(wg_start, wg_end)[0, 1])
    n = len(wg_end)
    print("  by XCD (workgroup index mod 8): median start / end / rows of the workgroup")
    wg_rows = rows.reshape(-1, W).sum(axis=1)
    for x in range(8):
        m = np.arange(n) % 8 == x
        print(f"   {x}: {np.median(wg_start[m]):5.1f} {np.median(wg_end[m]):5.1f} (max {wg_end[m].max():5.1f}) {np.median(wg_rows[m]):.0f}")
    order = np.argsort(wg_end)
    print("  the eight last workgroups:", [(int(i), round(float(wg_start[i]), 1), round(float(wg_end[i]), 1)) for i in order[-8:]])
    print("  the eight first workgroups:", [(int(i), round(float(wg_start[i]), 1), round(float(wg_end[i]), 1)) for i in order[:8]])
es.close()
