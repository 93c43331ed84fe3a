"""Diagnostic: where a wavefront of the fused spectral kernel spends its cycles per row (needs a -DSOTS_STAMP build):
waiting for the row to land, the transform (passes + LDS exchanges), the tail (split, error, reduction).
usage: SOTS_LIB_PATH=variants/libsots_stamp.so python tools/fft_probe.py [P] [log2n]"""
import ctypes as C, importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd")
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pmax, tp = bench.VOICES["2op"]
es = pkg.HipES(P // 4, P - P // 4, pkg.capi.SYNTH_2OP, log2n, None, pmax, seed=1)
es.set_target_audio(bench.make_target(pkg, "2op", log2n, 0))
es.init_population()
es.execute_generations(300)          # clocks settled
es.synchronize()
L = es.L
L.sots_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
L.sots_debug_clear_stamps()
es.execute_generations(1); es.synchronize()
buf = (C.c_ulonglong * (2 * 16384))()
L.sots_debug_stamps(buf, 2 * 16384)
a = np.frombuffer(buf, dtype=np.uint64)[3 * 8192:].reshape(-1, 16).astype(np.float64)
a = a[a[:, 3] > 0]
rows = a[:, 3]
print(f"P={P} N={1 << log2n}: {len(a)} wavefronts stamped, {rows.mean():.1f} rows each; cycles per row (median over wavefronts):")
for j, name in enumerate(["wait for the row", "window + transform", "split + error + reduce"]):
    print(f"  {name:24s} {np.median(a[:, j] / rows):8.0f}")
print(f"  total                    {np.median((a[:, 0] + a[:, 1] + a[:, 2]) / rows):8.0f}")
b, e = a[:, 4], a[:, 5]
t0 = b.min()
print(f"  workgroups 0..{len(a) - 1}: first starts at 0, last starts at {(b.max() - t0) / 100:.1f} us, ends between {(e.min() - t0) / 100:.1f} and {(e.max() - t0) / 100:.1f} us; "
      f"median residence {np.median(e - b) / 100:.1f} us")
print(f"  shader clock while resident: {np.median(a[:, 6] / (e - b)) * 100 / 1e3:.2f} GHz (cycles / 100 MHz ticks)")
print(f"  prologue (tables, twiddles, first requests) until the first row starts: {np.median(a[:, 7]):.0f} cycles; resident {np.median(a[:, 6]):.0f} cycles")
W = 12  # wavefronts per workgroup of the wide kernel (k_fft<10, ., ., 12>); slots are workgroup * W + wavefront
if len(a) % W == 0 and P >= 4 * W * 256:
    by = lambda v: np.array2string(np.median(v.reshape(-1, W), axis=0), precision=1, floatmode="fixed", max_line_width=200)
    print("  by wavefront of the workgroup (median over workgroups):")
    print("   rows       ", by(rows))
    print("   end (us)   ", by((e - t0) / 100))
    print("   cycles/row ", by(a[:, 6] / rows))
es.close()
