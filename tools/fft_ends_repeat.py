import ctypes as C, importlib, sys, os
import numpy as np
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd")
import bench
P = 65536
pmax, tp = bench.VOICES["2op"]
es = pkg.HipES(P // 4, P - P // 4, pkg.capi.SYNTH_2OP, 10, None, pmax, seed=1)
es.set_target_audio(bench.make_target(pkg, "2op", 10, 0))
es.init_population(); es.execute_generations(300); es.synchronize()
L = es.L
L.sots_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
runs = []
for rep in range(4):
    L.sots_debug_clear_stamps()
    es.execute_generations(1); es.synchronize()
    buf = (C.c_ulonglong * (2 * 16384))()
    L.sots_debug_stamps(buf, 2 * 16384)
    a = np.frombuffer(buf, dtype=np.uint64)[:16384].reshape(-1, 4)
    a = a[a[:, 2] > 0]
    b, e = a[:, 0].astype(np.float64), a[:, 1].astype(np.float64)
    t0 = b.min()
    wg_end = ((e - t0) / 100).reshape(-1, 12).max(axis=1)
    runs.append(wg_end)
    print("launch", rep, "workgroup ends: min %.1f median %.1f max %.1f" % (wg_end.min(), np.median(wg_end), wg_end.max()))
r = np.corrcoef(np.array(runs))
print("correlation of the per-workgroup end times between launches:\n", np.round(r, 2))
es.close()
m = np.mean(np.array(runs), axis=0)
n = len(m)
print("mean end per workgroup over the launches: min %.1f median %.1f max %.1f, std across workgroups %.2f (of single launches: %.2f)" % (m.min(), np.median(m), m.max(), m.std(), np.mean([r.std() for r in runs])))
for mod in (8, 16, 32, 64):
    g = np.array([m[np.arange(n) % mod == k].mean() for k in range(mod)])
    print(f"  by workgroup index mod {mod}: spread of the group means {g.max() - g.min():.2f} us", np.round(g, 1) if mod <= 16 else "")
g = np.array([m[(np.arange(n) // 8) % 32 == k].mean() for k in range(32)])
print("  by (index // 8) (the workgroup's turn on its XCD):", np.round(g, 1))
print("  slowest:", np.argsort(m)[-12:], " fastest:", np.argsort(m)[:12])
