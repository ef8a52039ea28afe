#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -- python3 scratch/pmc_gemm.py "$@" > $OUT/p1.log 2>&1 || { tail $OUT/p1.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
d = "gpurun_out/pmc_gemm/p1"
v = collections.defaultdict(lambda: collections.defaultdict(list)); t = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        v[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        t[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in sorted(v):
    if "gemm" not in k.lower() and "Cijk" not in k:
        continue
    m = {c: sum(x[2:]) / max(1, len(x[2:])) for c, x in v[k].items()}      # skip the first two launches (warm-up)
    ms = sum(t[k][2:]) / max(1, len(t[k][2:]))
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print(f"{k:72s} {ms:7.3f} ms  busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * gui):.3f}  clock {gui / (ms * 1e6):.2f} GHz  "
          f"wait/wave {m.get('SQ_WAIT_ANY', 0) / max(1, m.get('SQ_WAVE_CYCLES', 1)):.2f}")
PY
