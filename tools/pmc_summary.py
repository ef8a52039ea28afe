"""Summarise rocprofv3 --pmc CSV output (one directory per pass) per kernel: mean counter value per dispatch, kernel time from
the kernel trace, and a few derived figures (MFMA-busy, effective clock, LDS-busy, HBM bytes with the gfx950 FETCH_SIZE x2)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("attn_fwd_pipe_kernel", "attn_fwd_kernel", "attn_bwd_dkv2_kernel", "attn_bwd_dq2_kernel", "attn_bwd_delta_kernel"):
        if key in name:
            return key + (name[name.find("<"):name.find(">") + 1] if key == "attn_fwd_kernel" and "<" in name else "")
    return None


def main(root):
    vals = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> values
    times = defaultdict(list)
    for d in sorted(glob.glob(os.path.join(root, "*/"))):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r.get("Kernel_Name", ""))
                if k:
                    vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r.get("Kernel_Name", ""))
                if k:
                    times[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print("| kernel | launches | ms (profiled passes, mean) | MFMA-busy | clock GHz | wave-cycles split wait / issue-stall / active | LDS busy (IDX_ACTIVE / BUSY) | LDS bank-conflict share | HBM bytes per launch (2 x FETCH + WRITE) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for k in sorted(vals):
        v = {c: sum(x) / len(x) for c, x in vals[k].items()}
        ms = sum(times[k]) / max(len(times[k]), 1)
        gui = v.get("GRBM_GUI_ACTIVE")
        mfma = v.get("SQ_VALU_MFMA_BUSY_CYCLES")
        busy = f"{mfma / (1024 * gui / 8):.2f}" if gui and mfma else "-"
        clk = f"{gui / 8 / (ms * 1e6):.2f}" if gui and ms else "-"
        wc = v.get("SQ_WAVE_CYCLES")
        split = (f"{v.get('SQ_WAIT_ANY', 0) / wc:.2f} / {v.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} / {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}" if wc else "-")
        sb = v.get("SQ_BUSY_CYCLES")
        lds = f"{v.get('SQ_LDS_IDX_ACTIVE', 0) / sb:.2f}" if sb else "-"
        bc = f"{v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}" if "SQ_LDS_IDX_ACTIVE" in v else "-"
        hbm = f"{(2 * v.get('FETCH_SIZE', 0) + v.get('WRITE_SIZE', 0)) * 1024 / 1e9:.2f} GB (FETCH {v.get('FETCH_SIZE', 0) / 1e6:.2f} M KB, WRITE {v.get('WRITE_SIZE', 0) / 1e6:.2f} M KB)" if "FETCH_SIZE" in v else "-"
        print(f"| `{k}` | {len(times[k])} | {ms:.2f} | {busy} | {clk} | {split} | {lds} | {bc} | {hbm} |")
    print()
    print("raw means:")
    for k in sorted(vals):
        print(" ", k, {c: round(sum(x) / len(x), 1) for c, x in sorted(vals[k].items())})


main(sys.argv[1])
