"""Per-loop instruction census of a hipcc `-S` file: MFMAs, vector instructions, LDS reads, scratch (spill) traffic, waits.
Usage: python tools/asm_loops.py file.s [substring of the mangled kernel name]"""
import re
import sys

PATS = {"mfma": r"v_mfma", "scratch": r"scratch_", "exp": r"v_exp", "valu": r"^\s+v_(?!mfma)", "ds_read": r"ds_read",
        "ds_write": r"ds_write", "lds_dma": r"global_load_lds|buffer_load.* lds", "vmem": r"global_(load|store)|buffer_(load|store)",
        "waitcnt": r"s_waitcnt", "barrier": r"s_barrier", "v_mov": r"v_mov", "accvgpr": r"accvgpr", "nop": r"s_nop"}


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for f in re.split(r"\n(?=_Z\S*:)", s):
        name = f.split(":")[0]
        if not name.startswith("_Z") or want not in name:
            continue
        lines = f.split("\n")
        meta = re.search(r"\.vgpr_count:\s+(\d+)", s[s.find(name + ".kd") if (name + ".kd") in s else 0:])
        print(name, "lines", len(lines))
        labels = {}
        for i, l in enumerate(lines):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        for i, l in enumerate(lines):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                body = lines[labels[m.group(1)]:i]
                c = {k: sum(1 for x in body if re.search(p, x)) for k, p in PATS.items()}
                if c["mfma"] or c["valu"] > 20:
                    print(f"  loop {m.group(1)} [{labels[m.group(1)]}-{i}]: " + ", ".join(f"{k} {v}" for k, v in c.items() if v))


main()
