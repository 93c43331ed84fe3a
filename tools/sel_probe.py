"""Diagnostic: phase stamps (shader cycles since workgroup start) of the selection kernels; needs a
-DSOTS_STAMP build.  usage: SOTS_LIB_PATH=variants/libsots_stamp.so python tools/sel_probe.py [P] [pattern]"""
import ctypes as C, importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pattern = sys.argv[2] if len(sys.argv) > 2 else "random"
es = pkg.HipES(P // 4, P - P // 4, pkg.capi.SYNTH_2OP, 9, None, [3520.0, 8.0, 3520.0, 1.0], seed=1)
rng = np.random.default_rng(0)
f = rng.random(P, dtype=np.float32)
if pattern == "skew":
    f *= np.where((np.arange(P) // 1024) % 4 == 0, 0.05, 1.0).astype(np.float32)
v = rng.random((P, es.D), dtype=np.float32)
es.set_sort_mode(pkg.capi.SORT_TOP_ONLY)
es.write_population(v, v, f)
L = es.L
L.sots_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
for _ in range(200):
    es.select(); es.rotate(); es.rotate()
es.synchronize()
L.sots_debug_clear_stamps()
es.select(); es.synchronize()
buf = (C.c_ulonglong * (2 * 16384))()
L.sots_debug_stamps(buf, 2 * 16384)
a = np.frombuffer(buf, dtype=np.uint64)[2 * 8192:].reshape(-1, 16).astype(np.float64)
names = ["samples in LDS", "v* selected", "off[] scanned", "own keys requested", "copies issued", "copies landed",
         "searched", "rows moved", "T: fitness loaded", "T: runs sorted", "T: ranked", "T: written"]
rank_wgs = a[(a[:, 7] > 0)]
tile_wgs = a[(a[:, 11] > 0)]
print(f"P={P} pattern={pattern}: cycles since workgroup start, median / max over workgroups")
for j, nme in enumerate(names):
    src = rank_wgs if j < 8 else tile_wgs
    if len(src):
        print(f"  {nme:22s} {np.median(src[:, j]):9.0f} {src[:, j].max():9.0f}")
es.close()
