"""Diagnostic: when the wavefronts of the lane-exchange spectral kernel (k_fft_x, fused form) start and end, and how long
they wait for their rows (needs a -DSOTS_STAMP build; the first 2048 wavefronts are stamped).
usage: SOTS_LIB_PATH=variants/stamp.so python tools/fftx_probe.py [P] [log2n]"""
import ctypes as C, importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd")
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
pmax, tp = bench.VOICES["2op"]
es = pkg.HipES(P // 4, P - P // 4, pkg.capi.SYNTH_2OP, log2n, None, pmax, seed=1)
es.set_target_audio(bench.make_target(pkg, "2op", log2n, 0))
es.init_population()
es.execute_generations(100)
es.synchronize()
L = es.L
L.sots_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
L.sots_debug_clear_stamps()
es.execute_generations(1); es.synchronize()
buf = (C.c_ulonglong * (2 * 16384))()
L.sots_debug_stamps(buf, 2 * 16384)
a = np.frombuffer(buf, dtype=np.uint64)[:16384].reshape(-1, 8).astype(np.float64)
a = a[a[:, 3] > 0]
b, tb, e, rows, wait, split, clk = (a[:, j] for j in range(7))
t0 = b.min()
q = lambda v: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(v, [0, 10, 50, 90, 100]))
print(f"P={P} N={1 << log2n}: {len(a)} wavefronts stamped, {rows.mean():.2f} rows each")
print("  start (us)        ", q((b - t0) / 100))
print("  tables ready (us) ", q((tb - t0) / 100))
print("  end (us)          ", q((e - t0) / 100))
print("  residence (us)    ", q((e - b) / 100))
ghz = np.median(clk / (e - b)) * 100 / 1e3
print(f"  shader clock {ghz:.2f} GHz;  per row (cycles): wait for the row {np.median(wait / rows):.0f}, split {np.median(split / rows):.0f}, "
      f"everything {np.median(clk / rows):.0f}")
print("  waiting share of residence", q(wait / clk * 100), "%")
W = 16 if log2n <= 12 else 8
n = len(a) // W * W
by = lambda v: np.array2string(np.median(v[:n].reshape(-1, W), axis=0), precision=1, floatmode="fixed", max_line_width=200)
print("  by wavefront of the workgroup (median over workgroups):")
print("   rows        ", by(rows))
print("   end (us)    ", by((e - t0) / 100))
print("   cycles/row  ", by(clk / rows))
print("   wait/row    ", by(wait / rows))
es.close()
