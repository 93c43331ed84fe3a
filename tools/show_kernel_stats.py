"""Print a rocprofv3 --stats kernel_stats.csv compactly: python tools/show_kernel_stats.py <dir-or-csv>"""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)[0]
for r in csv.DictReader(open(path)):
    n = r["Name"]
    n = n.split("::")[2].split("(")[0] if "anonymous namespace" in n else n.split("(")[0]
    print(f'{n:36s} calls {r["Calls"]:>6s}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  min {float(r["MinNs"]) / 1e3:9.1f}  max {float(r["MaxNs"]) / 1e3:9.1f}  {float(r["Percentage"]):6.2f} %')
