"""Per kernel name and per block of 6 consecutive launches (dispatch order): mean of every counter + mean duration."""
import csv, glob, os, sys
from collections import defaultdict
for d in sys.argv[1:]:
    rows = defaultdict(lambda: defaultdict(dict))   # kernel -> dispatch id -> counter -> value
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[r["Kernel_Name"][:50]][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    times = defaultdict(dict)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            times[r["Kernel_Name"][:50]][int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k in sorted(rows):
        if not any(m in k for m in ("gemm", "Cijk")):
            continue
        ids = sorted(rows[k])
        for blk in range(0, len(ids), 6):
            sel = ids[blk:blk + 6]
            cs = sorted({c for i in sel for c in rows[k][i]})
            ms = [times[k].get(i) for i in sel if times[k].get(i) is not None]
            print(f"{os.path.basename(d)} | {k} | block {blk // 6} | {sum(ms) / max(1, len(ms)):.3f} ms | " +
                  ", ".join(f"{c}={sum(rows[k][i].get(c, 0) for i in sel) / len(sel):.4g}" for c in cs))
