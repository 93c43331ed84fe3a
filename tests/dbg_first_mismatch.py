"""Diagnostic (run by hand on a GPU box, not collected by pytest): first mismatching samples of a few rows of each series voice
against the oracle.  Lives under tests/ because it uses the oracle (test infrastructure)."""
import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module(bench.PKG)
from oracle import oracle as O  # (diagnostic tool: the oracle is the checker here, as in tests/)
for kind, name, log2n in ((0, "2op", 10), (3, "4op_series", 10), (1, "3op_series", 10)):
    pmax = bench.VOICES[name][0]
    es = pkg.HipES(4096, 12288, kind, log2n, None, pmax, seed=1)
    es.init_population(0)
    v, s, _ = es.read_population()
    es.synthesise()
    a = es.read_audio()
    for r in (0, 1, 17, 5000):
        ref = O.synth(kind, v[r], [0.0] * es.D, pmax, es.N)
        bad = np.nonzero(a[r] != ref)[0]
        print(name, "row", r, "mismatches", len(bad), "first", bad[:6], "got", a[r][bad[:3]], "want", ref[bad[:3]])
    es.close()
