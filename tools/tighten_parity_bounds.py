#!/usr/bin/env python3
"""Rewrites the literal bounds of `assert rel_l2(..., bound=B) < B` in tests/test_gpu_*.py to 1.5 x the largest value that assert
site measured in a full `pytest -m gpu` run (the record tests/conftest.py writes: gpurun_out/kernel_parity.json), wherever the
bound in the source is looser than 2 x that value.  Sites that measured exactly 0 (bitwise-equal code paths) and bounds that are
expressions are left alone.  Usage: python tools/tighten_parity_bounds.py gpurun_out/kernel_parity.json [--apply]"""
import collections
import json
import math
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
FLOOR = 1e-6          # fp32-vs-fp32 sites sit at 1e-7 .. 1e-9: summation-order noise, not a property worth 1.5 x


def lit(x: float) -> str:
    e = math.floor(math.log10(x))
    m = math.ceil(x / 10 ** e * 10) / 10          # round UP to two significant digits
    if m >= 10:
        m, e = 1.0, e + 1
    return f"{m:.1f}e{e:d}"


def main():
    rec = json.load(open(sys.argv[1]))
    apply = "--apply" in sys.argv
    site = collections.defaultdict(list)
    for recs in rec.values():
        for r in recs:
            if r.get("bound") is not None:
                site[(r["line"], float(r["bound"]))].append(r["rel_l2"])
    by_file = collections.defaultdict(dict)
    for (where, b), v in site.items():
        f, ln = where.split(":")
        by_file[f][(int(ln), b)] = max(v)
    pat = re.compile(r"bound=([0-9.eE+-]+)\)\s*<\s*([0-9.eE+-]+)(?=\s*(?:$|and\b|,|\)|#))")
    changed = 0
    for f, sites in sorted(by_file.items()):
        path = ROOT / "tests" / f
        lines = path.read_text().split("\n")
        for (ln, b), mx in sorted(sites.items()):
            if mx <= 0 or b <= 2 * max(mx, FLOOR / 1.5):
                continue
            new = lit(max(1.5 * mx, FLOOR))
            if float(new) >= b:
                continue

            def rep(m):
                if float(m.group(1)) == b and float(m.group(2)) == b:
                    return f"bound={new}) < {new}"
                return m.group(0)
            out = pat.sub(rep, lines[ln - 1])
            if out != lines[ln - 1]:
                print(f"{f}:{ln}: {b:g} -> {new}   (measured max {mx:.3e})")
                lines[ln - 1] = out
                changed += 1
        if apply:
            path.write_text("\n".join(lines))
    print(f"{changed} bounds {'rewritten' if apply else 'would be rewritten'}")


if __name__ == "__main__":
    main()
