"""Raw per-kernel means of every counter found under the rocprofv3 output directories given (one line per directory and kernel),
with the kernel's mean duration from the kernel trace.  Usage: python tools/pmc_raw.py DIR [DIR ...] [--match substring[,substring...]]"""
import csv
import glob
import os
import sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
args = [a for a in args if a != match]
for d in args:
    vals = defaultdict(lambda: defaultdict(list))
    times = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            vals[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            times[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k in sorted(vals):
        if match and not any(m in k for m in match.split(",")):
            continue
        ms = sum(times[k]) / max(1, len(times[k]))
        print(f"{d} | {k} | {ms:.3f} ms x{len(times[k])} | " + ", ".join(f"{c}={sum(x) / len(x):.4g}" for c, x in sorted(vals[k].items())))
