"""Print the throughput and per-kernel averages of bench.py JSON lines."""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        if line.startswith("{"):
            d = json.loads(line)
            ks = {k: round(v["avg_us"], 1) for k, v in d.get("kernels", {}).items()}
            print(f'{d["value"] / 1e6:.1f} Mcand/s  {d["ms_per_step"] * 1e3:.0f} us/gen  {ks}  roofline {d["roofline"]["kernel"]} {d["roofline"]["frac"]:.3f}')
