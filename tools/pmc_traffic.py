"""profiles/pmc_traffic.json (HBM bytes per launch of each fused-loop kernel, keyed by workload) from a
pmc_summary.json of tools/pmc_collect.sh:
    python tools/pmc_traffic.py <pmc_summary.json> <workload key, e.g. P65536_N1024_2op> [round tag]"""
import hashlib
import json
import os
import sys

d = json.load(open(sys.argv[1]))
key = sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else ""


def nbytes(k):
    # FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); both in KiB
    return (2 * d[k].get("FETCH_SIZE", 0) + d[k].get("WRITE_SIZE", 0)) * 1024


def pick(prefix):
    ks = [k for k in d if k.startswith(prefix)]
    return max(ks, key=lambda k: d[k].get("dispatches_seen", 0))


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd"
# bench.py reports these figures only while the kernel source is the one they were collected on
sha16 = hashlib.sha256(open(os.path.join(ROOT, PKG, "csrc", "sots_kernels.hip"), "rb").read()).hexdigest()[:16]
entry = {"_round": tag,
         "kernels_sha16": sha16,
         "synthesise": round(nbytes(pick("k_synth"))),
         "window+FFT+fitness": round(nbytes(pick("k_fft"))),
         "recombine+mutate": round(nbytes("k_recombine_mutate")) if "k_recombine_mutate" in d else None,
         # the fused loop's sortPopulation = the two selection kernels (k_sort_* only run when the order is read)
         "sortPopulation": round(sum(nbytes(k) for k in d if k.startswith("k_sel")))}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
allw = json.load(open(path)) if os.path.exists(path) else {}
allw.setdefault("_source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_collect.sh), mean per dispatch; "
                           "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B "
                           "requests at 64 B)")
allw.setdefault("workloads", {})[key] = entry
json.dump(allw, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
