import importlib, sys, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
pkg = importlib.import_module("survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd")
from oracle import oracle as O
import test_gpu_parity as T
for (parents, offspring, pattern) in [(32768,32768,"random"),(32768,32768,"tile_skew"),(32768,98304,"tile_skew"),(16384,49152,"tile_skew")]:
    es = pkg.HipES(parents, offspring, 0, 9, None, T.PMAX[0], seed=1)
    rng = np.random.default_rng(parents + len(pattern))
    P, D = es.P, es.D
    S = parents
    f = T.fitness_pattern(pattern, P, rng)
    v = rng.random((P, D), dtype=np.float32); s = rng.random((P, D), dtype=np.float32)
    es.set_sort_mode(2)
    sent = np.full((P, D), -7.0, np.float32)
    es.write_population(sent, sent, np.full(P, -7.0, np.float32)); es.rotate(); es.write_population(v, s, f)
    es.select(); es.rotate()
    gv, gs, gf = es.read_population()
    perm = O.sort_perm(f)
    want = f[perm][:S]
    bad = np.nonzero(gf[:S] != want)[0]
    print(parents, offspring, pattern, "mismatches", len(bad), "first", bad[:10], "unwritten", int((gf[:S] == -7.0).sum()), "tail written", int((gf[S:] != -7.0).sum()))
    if len(bad):
        i = bad[0]
        print("  got", gf[i-2:i+6], "want", want[i-2:i+6])
        # where did the wanted rows go
        src_rank = {}
    es.close()
