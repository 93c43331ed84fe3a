"""Diagnostic: device time of synthesisePopulation alone (stage-separated launch, stored genes), per library variant.
usage: python tools/synth_time.py lib.so[,lib2.so...] [--synth 4op_series --log2n 12 --pop 32768] [--reps 40]"""
import argparse, importlib, os, subprocess, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--synth", default="4op_series"); ap.add_argument("--log2n", type=int, default=12)
    ap.add_argument("--pop", type=int, default=32768); ap.add_argument("--reps", type=int, default=40)
    a, _ = ap.parse_known_args()
    pkg = importlib.import_module(bench.PKG)
    pmax, _t = bench.VOICES[a.synth]
    es = pkg.HipES(a.pop // 4, a.pop - a.pop // 4, pkg.capi.SYNTH_NAMES[a.synth], a.log2n, None, pmax, seed=1)
    es.init_population(0)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(20): es.synthesise()
        es.synchronize()
    best = 1e9
    for _ in range(3):
        es.synchronize(); t = time.perf_counter()
        for _ in range(a.reps): es.synthesise()
        es.synchronize(); best = min(best, (time.perf_counter() - t) / a.reps * 1e6)
    print(f"{best:8.1f} us per launch")


if __name__ == "__main__":
    if os.environ.get("SOTS_SYNTH_TIME_CHILD"):
        one()
    else:
        libs = sys.argv[1].split(",")
        for rep in range(2):
            for lib in libs:
                env = dict(os.environ, SOTS_LIB_PATH=lib, SOTS_SYNTH_TIME_CHILD="1")
                out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
                print(f"{lib:40s} {' '.join(sys.argv[2:])}  {out.stdout.strip() or out.stderr.strip()[-300:]}", flush=True)
