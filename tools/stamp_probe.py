"""Diagnostic: in-kernel shader cycles of the synthesis loop (needs a -DSOTS_STAMP build).
usage: SOTS_LIB_PATH=variants/libsots_stamp.so [SOTS_SYNTH_CUT=0|1] python tools/stamp_probe.py P [log2n] [voice]"""
import ctypes as C, importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd")
P = int(sys.argv[1]); log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
import bench
voice = sys.argv[3] if len(sys.argv) > 3 else "2op"
pmax, _ = bench.VOICES[voice]
es = pkg.HipES(P // 4, P - P // 4, pkg.capi.SYNTH_NAMES[voice], log2n, None, pmax, seed=1)
es.init_population()
L = es.L
t0 = time.time()
while time.time() - t0 < 2.0:           # let the clock settle under load
    for _ in range(200): es.synthesise()
    es.synchronize()
L.sots_debug_clear_stamps(); 
es.synthesise(); es.synchronize()
buf = (C.c_ulonglong * (2 * 16384))()
L.sots_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
L.sots_debug_stamps(buf, 2 * 16384)
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
waves_per_wg = int(os.environ.get("SOTS_PROBE_WAVES", "0"))
if waves_per_wg:
    # by wavefront of the workgroup (slot = workgroup * waves + wavefront) and by workgroup
    nwg = int((a[:, 0] > 0).sum()) // waves_per_wg
    g = a[:nwg * waves_per_wg, 0].reshape(nwg, waves_per_wg) / (1 << log2n)
    print("   by wavefront of the workgroup (median cycles/sample):", " ".join(f"{x:.1f}" for x in np.median(g, axis=0)))
    wmax = g.max(axis=1)
    print(f"   slowest wavefront per workgroup: min {wmax.min():.1f} median {np.median(wmax):.1f} max {wmax.max():.1f};  fastest per workgroup: median {np.median(g.min(axis=1)):.1f}")
a = a[a[:, 0] > 0]
n = 1 << log2n
clk = a[:, 0] / a[:, 1] * 100e6
print(f"{os.environ.get('SOTS_LIB_PATH', 'default'):24s} {voice} P={P} waves stamped={len(a)} cycles/sample median={np.median(a[:,0])/n:.1f} min={a[:,0].min()/n:.1f} max={a[:,0].max()/n:.1f}  "
      f"clock median={np.median(clk)/1e9:.3f} GHz  loop time median={np.median(a[:,1])/100:.1f} us")
