#!/bin/bash
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_conv
rm -rf $OUT; mkdir -p $OUT
cd $R
python3 scratch/pmc_conv.py 5 | tee $OUT/plain.log
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 scratch/pmc_conv.py 5 > $OUT/$name.log 2>&1; echo "$name done"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - <<'PY'
import csv, glob, collections, os
root = "gpurun_out/pmc_conv"
v = collections.defaultdict(lambda: collections.defaultdict(list)); t = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        v[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        t[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(root + "/summary.md", "w") as o:
    for k in sorted(v):
        if "gemm16" not in k and "conv_rows" not in k and "conv_wide" not in k: continue
        m = {c: sum(x) / len(x) for c, x in v[k].items()}
        ms = sum(t[k]) / len(t[k]); gui = m.get("GRBM_GUI_ACTIVE", 0) / 8; wc = m.get("SQ_WAVE_CYCLES", 1); sb = m.get("SQ_BUSY_CYCLES", 1)
        line = (f"{k}: {ms:.2f} ms busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * gui):.3f} clock {gui / (ms * 1e6):.2f} GHz "
                f"wait/stall/active {m.get('SQ_WAIT_ANY', 0) / wc:.2f}/{m.get('SQ_WAIT_INST_ANY', 0) / wc:.2f}/{m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} "
                f"lds_wait {m.get('SQ_WAIT_INST_LDS', 0) / wc:.2f} lds_active {m.get('SQ_LDS_IDX_ACTIVE', 0) / sb:.2f} bank_conf {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, m.get('SQ_LDS_IDX_ACTIVE', 1)):.3f} "
                f"HBM {(2 * m.get('FETCH_SIZE', 0) + m.get('WRITE_SIZE', 0)) * 1024 / 1e9:.2f} GB (fetch {2 * m.get('FETCH_SIZE', 0) * 1024 / 1e9:.2f}) "
                f"L2 hit {m.get('TCC_HIT_sum', 0) / max(1, m.get('TCC_REQ_sum', 1)):.3f} req {m.get('TCC_REQ_sum', 0) / 1e6:.1f}M "
                f"valu/lds insts {m.get('SQ_INSTS_VALU', 0) / 1e6:.0f}M/{m.get('SQ_INSTS_LDS', 0) / 1e6:.0f}M")
        print(line); o.write(line + "\n\n")
PY
