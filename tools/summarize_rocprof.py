#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats` output dir into a small committed summary (profiles/)."""
import csv
import json
import sys
from pathlib import Path


def main(src: str, dst: str, note: str = ""):
    f = next(Path(src).rglob("*kernel_stats.csv"))
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = [f"# rocprofv3 --kernel-trace --stats summary ({note})", "",
           "| kernel | calls | total ms | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|---|"]
    for r in rows[:25]:
        out.append(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | "
                   f"{float(r['AverageNs'])/1e6:.4f} | {float(r['MinNs'])/1e6:.4f} | {float(r['MaxNs'])/1e6:.4f} | "
                   f"{100*float(r['TotalDurationNs'])/tot:.2f} |")
    out.append("")
    out.append(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
    Path(dst).write_text("\n".join(out) + "\n")
    print("\n".join(out[:14]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
