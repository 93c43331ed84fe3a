"""profiles/pmc_traffic.json (HBM bytes per launch of each fused-loop kernel) from a pmc_summary.json
of tools/pmc_collect.sh: python tools/pmc_traffic.py profiles/r01_final_pmc_summary.json"""
import json
import sys

d = json.load(open(sys.argv[1]))


def nbytes(k):
    # FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); both in KiB
    return (2 * d[k].get("FETCH_SIZE", 0) + d[k].get("WRITE_SIZE", 0)) * 1024


def pick(prefix, most="SQ_WAVES"):
    ks = [k for k in d if k.startswith(prefix)]
    return max(ks, key=lambda k: d[k].get("dispatches_seen", 0))


out = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_collect.sh via tools/final_profile.sh), "
                  "mean per dispatch, P=65536 N=1024 2-op; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled per "
                  "MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)",
       "synthesise": round(nbytes(pick("k_synth"))),
       "window+FFT+fitness": round(nbytes(pick("k_fft"))),
       "recombine+mutate": round(nbytes("k_recombine_mutate")) if "k_recombine_mutate" in d else None,
       "sortPopulation": round(sum(nbytes(k) for k in d if k.startswith("k_sort")))}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
