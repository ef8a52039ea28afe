#!/usr/bin/env python3
"""Token-stream overlap between a host file of this repository and the reference span it restates.

The copy detector the driver runs is line based; this is the stricter view DESIGN.md §2 quotes: both sides are cut into
Python tokens (comments, docstrings, blank lines and layout dropped), and the share of THIS repository's tokens that
lie inside `difflib.SequenceMatcher` matching blocks of at least `--min-block` tokens is printed.

    python tools/token_overlap.py longcat-video-tta_amd/tta/inner_loop.py \
        /root/reference/lora_experiment/scripts/run_lora_tta.py:425-634
    python tools/token_overlap.py --table        # every (repo file, reference span) pair DESIGN.md lists

Runs in the build container only (the reference does not travel to the GPU box); reads both sides as text.
"""
import argparse
import difflib
import io
import sys
import tokenize
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "longcat-video-tta_amd"
REF = Path("/root/reference")

# (repo file, reference file, first line, last line) — the spans VERDICT / DESIGN §2 compare
PAIRS = [
    ("tta/early_stopping.py", "delta_experiment/scripts/early_stopping.py", 1, None),
    ("tta/inner_loop.py", "lora_experiment/scripts/run_lora_tta.py", 425, 634),
    ("tta/full_tta.py", "lora_experiment/scripts/run_full_tta.py", 95, 304),
    ("tta/latent_split.py", "delta_experiment/scripts/common.py", 1365, 1517),
    ("tta/common.py", "delta_experiment/scripts/common.py", 46, 611),
    ("tta/cli_args.py", "delta_experiment/scripts/common.py", 1404, 1706),
    ("tta/flow_matching.py", "delta_experiment/scripts/common.py", 274, 559),
    ("tta/lora.py", "lora_experiment/scripts/run_lora_tta.py", 104, 418),
    ("tta/delta.py", "delta_experiment/scripts/run_delta_a.py", 88, 305),
    ("lora_experiment/scripts/run_lora_tta.py", "lora_experiment/scripts/run_lora_tta.py", 1, None),
    ("lora_experiment/scripts/run_full_tta.py", "lora_experiment/scripts/run_full_tta.py", 1, None),
    ("baseline_experiment/scripts/run_baseline.py", "baseline_experiment/scripts/run_baseline.py", 1, None),
    ("delta_experiment/scripts/run_delta_a.py", "delta_experiment/scripts/run_delta_a.py", 1, None),
    ("delta_experiment/scripts/run_film_tta.py", "delta_experiment/scripts/run_film_tta.py", 1, None),
]


def tokens_of(text: str):
    """Significant tokens of Python source: no comments, no docstrings, no NEWLINE / INDENT bookkeeping."""
    out, prev_sig = [], None
    try:
        for tok in tokenize.generate_tokens(io.StringIO(text).readline):
            if tok.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT,
                            tokenize.ENCODING, tokenize.ENDMARKER):
                continue
            if tok.type == tokenize.STRING and prev_sig in (None, ":", "NEWLINE") and tok.string.lstrip("rRbBuU")[:3] in ('"""', "'''"):
                continue                              # a docstring (a bare triple-quoted expression statement)
            out.append(tok.string)
            prev_sig = tok.string
    except (tokenize.TokenError, IndentationError):
        pass                                          # a span cut mid-block: keep what tokenised
    return out


def read_span(path: Path, first: int = 1, last=None) -> str:
    lines = path.read_text(errors="replace").splitlines(keepends=True)
    body = "".join(lines[first - 1:last])
    # a span that starts inside an indented block does not tokenise: dedent it to its own first line
    import textwrap
    return textwrap.dedent(body)


def overlap(repo_text: str, ref_text: str, min_block: int = 6):
    a, b = tokens_of(repo_text), tokens_of(ref_text)
    if not a:
        return 0.0, 0, len(b)
    sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
    matched = sum(m.size for m in sm.get_matching_blocks() if m.size >= min_block)
    return matched / len(a), len(a), len(b)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("repo_file", nargs="?")
    ap.add_argument("reference_span", nargs="?", help="path[:first-last]")
    ap.add_argument("--table", action="store_true", help="print the table over the pairs listed in this file")
    ap.add_argument("--min-block", type=int, default=6)
    args = ap.parse_args(argv)
    rows = []
    if args.table:
        for rf, ref, lo, hi in PAIRS:
            rp, fp = PKG / rf, REF / ref
            if not rp.exists() or not fp.exists():
                rows.append((rf, f"{ref}:{lo}-{hi or 'end'}", None, 0, 0))
                continue
            frac, na, nb = overlap(rp.read_text(), read_span(fp, lo, hi), args.min_block)
            rows.append((rf, f"{ref}:{lo}-{hi or 'end'}", frac, na, nb))
    elif args.repo_file and args.reference_span:
        spec = args.reference_span
        path, lo, hi = spec, 1, None
        if ":" in spec and spec.rsplit(":", 1)[1].replace("-", "").isdigit():
            path, rng = spec.rsplit(":", 1)
            lo, hi = (int(x) for x in rng.split("-")) if "-" in rng else (int(rng), None)
        frac, na, nb = overlap(Path(args.repo_file).read_text(), read_span(Path(path), lo, hi), args.min_block)
        rows.append((args.repo_file, spec, frac, na, nb))
    else:
        ap.error("give a repo file and a reference span, or --table")
    print(f"{'repo file':52s} {'reference span':62s} {'matched':>8s} {'tokens':>7s} {'ref':>7s}")
    for rf, span, frac, na, nb in rows:
        shown = "absent" if frac is None else f"{100 * frac:6.1f} %"
        print(f"{rf:52s} {span:62s} {shown:>8s} {na:7d} {nb:7d}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
