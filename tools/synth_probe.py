"""Diagnostic: where k_synth's time outside its sample loop goes (needs a -DSOTS_STAMP build):
shader cycles since workgroup start at "individuals made", "table in LDS", "samples stored".
usage: SOTS_LIB_PATH=variants/libsots_stamp.so python tools/synth_probe.py [P]"""
import ctypes as C, importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd")
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pmax, tv = bench.VOICES["2op"]
tgt = bench.make_target(pkg, "2op", 10, 0)
es = pkg.HipES(P // 4, P - P // 4, pkg.capi.SYNTH_2OP, 10, None, pmax, seed=1)
es.set_target_audio(tgt)
es.init_population()
es.execute_generations(400)
es.synchronize()
L = es.L
L.sots_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
L.sots_debug_clear_stamps()
es.timing_reset(); es.timing_enable(True)
es.execute_generations(1); es.synchronize()
es.timing_enable(False)
ms, n = es.stage_time_ms(pkg.capi.STAGE_FUSED_SYNTH)
buf = (C.c_ulonglong * (2 * 16384))()
L.sots_debug_stamps(buf, 2 * 16384)
a = np.frombuffer(buf, dtype=np.uint64)
ph = a[2 * 8192:2 * 8192 + 512 * 16].reshape(-1, 16).astype(np.float64)
loop = a[:2 * 8192].reshape(-1, 2).astype(np.float64)
loop = loop[loop[:, 0] > 0]
clk = np.median(loop[:, 0] / loop[:, 1]) * 100e6 if len(loop) else 2.3e9
ph = ph[ph[:, 14] > 0]
print(f"P={P}: k_synth by its events {1e3 * ms / max(n, 1):.1f} us; shader clock {clk / 1e9:.2f} GHz; {len(ph)} workgroups stamped")
for j, name in ((12, "individuals made"), (13, "table in LDS"), (14, "samples stored (issued)")):
    print(f"  {name:26s} median {np.median(ph[:, j]) / clk * 1e6:6.2f} us   max {ph[:, j].max() / clk * 1e6:6.2f} us   ({np.median(ph[:, j]):.0f} cycles)")
t0, t1 = ph[64:, 10], ph[64:, 11]  # (workgroups 0..63 share these two slots with k_sel_tiles' phases)
print(f"  workgroups start over {(t0.max() - t0.min()) / 100:.2f} us, end over {(t1.max() - t1.min()) / 100:.2f} us; first start -> last end "
      f"{(t1.max() - t0.min()) / 100:.2f} us (the kernel's events / rocprofv3 add the launch in front and the cache write-back behind)")
if len(loop):
    print(f"  sample loop per wavefront  median {np.median(loop[:, 1]) / 100:6.2f} us")
es.close()
