#!/usr/bin/env python3
"""Mechanical screen for the hand-issued MFMAs of this library (hipcc `-S` text in, findings out).

Several kernels issue their MFMAs from `asm volatile` so that the accumulator register class and the program order are what the
source says (csrc/attn_fwd_w64.hip, attn_fwd_pipe.hip, gemm4k.h).  hipcc's hazard recogniser does not look
inside an asm statement: when the register allocator puts a copy of an MFMA operand (`v_mov_b32`, `v_accvgpr_write_b32`,
`v_accvgpr_mov_b32`, any VALU result) right in front of such a statement, nothing inserts the wait states the matrix pipe needs
between a vector write and an MFMA read of the same register, and the MFMA reads the OLD value — four distinct wrong-result bugs
of round 3 were exactly this (profiles/r03_attn_bwd_lab.md), one of them visible only for an odd tile count.  The sources guard
the places where the compiler may do that with `s_nop 3` in the same asm statement.  This scan checks the compiler's output:

  H1  for every `v_mfma*` that is not directly preceded by an `s_nop`, look at the previous instructions for a vector
      instruction that writes a register inside the MFMA's A, B or C operand ranges (MFMA -> MFMA accumulator chains are
      interlocked by the hardware and are not findings).  Pairs the compiler generated BOTH halves of are its hazard
      recogniser's business and are skipped; of the rest,
        * a COMPILER-placed write (a register copy the allocator put there) within LOOKBACK = 4 instructions of an asm-issued
          MFMA is a finding: the author did not choose that distance, the next compiler release may shrink it, and this is the
          exact shape of the round-3 bugs (their guards are `s_nop 3` = four wait states);
        * an author-placed (asm) write is a finding when fewer than MIN_STATES = 2 instructions stand between it and the MFMA,
          the wait states the matrix pipe needs after a vector write of one of its sources (what the compiler itself inserts
          between a VALU write and an MFMA read on gfx90a and later);
  H2  no scratch (spill) access inside the steady-state loops: the innermost loops with the kernel's largest MFMA count (shorter
      MFMA loops are the peeled / ragged tail forms, which run once or twice per workgroup and may reload a spilled pointer);
  H3  outside asm statements the compiler itself does not touch `m0` in kernels whose asm clobbers it (the LDS-DMA requests
      set m0 per piece; a compiler use of m0 across them would be silently corrupted).

Usage: python tools/isa_hazards.py file.s [file.s ...]     (exit code 1 when there are findings)
"""
import re
import sys
from typing import List, NamedTuple, Optional, Set, Tuple

LOOKBACK = 4
MIN_STATES = 2
_REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+))")


class Inst(NamedTuple):
    line: int
    text: str
    op: str
    args: str
    in_asm: bool


def _regs(operand: str) -> Set[Tuple[str, int]]:
    out = set()
    for m in _REG.finditer(operand):
        lo = int(m.group(2) if m.group(2) is not None else m.group(4))
        hi = int(m.group(3) if m.group(3) is not None else m.group(4))
        out.update((m.group(1), r) for r in range(lo, hi + 1))
    return out


def _split_operands(args: str) -> List[str]:
    parts, depth, cur = [], 0, ""
    for ch in args:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


def parse_kernels(text: str):
    """-> {kernel name: [Inst, ...]} (labels are kept as Inst with op ':label')."""
    kernels, cur, name, in_asm = {}, None, None, False
    for n, raw in enumerate(text.split("\n"), 1):
        line = raw.split("//")[0].rstrip()
        if not line.strip():
            continue
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if s.startswith(";") or s.startswith("."):
            lab = s.split(";")[0].strip()
            if lab.startswith(".LBB") and lab.endswith(":") and cur is not None:
                cur.append(Inst(n, lab, ":label", lab[:-1], False))
            if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
                cur, name = None, None
            continue
        m = re.match(r"^([A-Za-z_][\w$.]*):", s)
        if m and not raw.startswith((" ", "\t")):
            name = m.group(1)
            cur = kernels.setdefault(name, [])
            continue
        if cur is None:
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        op, _, args = s.partition(" ")
        cur.append(Inst(n, s, op, args.strip(), in_asm))
    return {k: v for k, v in kernels.items() if any(i.op.startswith("v_mfma") for i in v)}


def _vector_write(i: Inst) -> Set[Tuple[str, int]]:
    """Registers a NON-MFMA vector instruction writes (its first operand when that is a v / a register)."""
    if not i.op.startswith("v_") or i.op.startswith("v_mfma") or i.op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()
    ops = _split_operands(i.args)
    return _regs(ops[0]) if ops and re.match(r"^[va](\[|\d)", ops[0]) else set()


def scan_mfma_hazards(insts: List[Inst]) -> List[str]:
    findings = []
    real = [i for i in insts if i.op != ":label"]
    for k, i in enumerate(real):
        if not i.op.startswith("v_mfma"):
            continue
        if k > 0 and real[k - 1].op == "s_nop":
            continue
        ops = _split_operands(i.args)
        reads = set()
        for o in ops[1:4]:
            reads |= _regs(o)
        states = 0                                       # wait states between the candidate writer and the MFMA
        for back in range(1, LOOKBACK + 1):
            if k - back < 0:
                break
            p = real[k - back]
            hit = _vector_write(p) & reads
            if hit and (p.in_asm or i.in_asm):
                limit = MIN_STATES if p.in_asm else LOOKBACK
                if states < limit:
                    r = sorted(hit)[0]
                    who = "asm-placed" if p.in_asm else "COMPILER-placed"
                    findings.append(f"H1 line {i.line}: `{i.text}` reads {r[0]}{r[1]}, written by the {who} `{p.text}` (line {p.line}) "
                                    f"with {states} wait state(s) in between (needs {limit}) and no s_nop in front of the MFMA")
                    break
            states += (int(p.args or 0) + 1) if p.op == "s_nop" else 1
    return findings


def innermost_mfma_loops(insts: List[Inst]) -> List[Tuple[int, int]]:
    """(first, last) instruction index of every innermost backward-branch loop that contains an MFMA."""
    label_at = {i.args: k for k, i in enumerate(insts) if i.op == ":label"}
    loops = []
    for k, i in enumerate(insts):
        if i.op.startswith(("s_cbranch", "s_branch")):
            tgt = label_at.get(i.args.split()[-1] if i.args else "")
            if tgt is not None and tgt < k and any(x.op.startswith("v_mfma") for x in insts[tgt:k]):
                loops.append((tgt, k))
    return [(a, b) for (a, b) in loops if not any((c > a or d < b) and c >= a and d <= b for (c, d) in loops if (c, d) != (a, b))]


def scan_kernel(name: str, insts: List[Inst], check_m0: bool = True) -> List[str]:
    out = [f"{name}: {f}" for f in scan_mfma_hazards(insts)]
    loops = innermost_mfma_loops(insts)
    count = lambda ab: sum(x.op.startswith("v_mfma") for x in insts[ab[0]:ab[1]])
    most = max((count(ab) for ab in loops), default=0)
    for a, b in [ab for ab in loops if count(ab) == most]:
        sc = [i for i in insts[a:b] if i.op.startswith("scratch_")]
        if sc:
            out.append(f"{name}: H2 {len(sc)} scratch access(es) inside the MFMA loop at lines {insts[a].line}-{insts[b].line}, "
                       f"first `{sc[0].text}` (line {sc[0].line})")
    if check_m0 and any(i.in_asm and re.search(r"\bm0\b", i.args) for i in insts):
        for i in insts:
            if not i.in_asm and i.op != ":label" and re.search(r"\bm0\b", i.args):
                out.append(f"{name}: H3 compiler-generated use of m0 outside asm: `{i.text}` (line {i.line})")
    return out


def scan_text(text: str) -> Tuple[List[str], dict]:
    findings, stats = [], {}
    for name, insts in parse_kernels(text).items():
        findings += scan_kernel(name, insts)
        mf = [i for i in insts if i.op.startswith("v_mfma")]
        stats[name] = {"mfma": len(mf), "mfma_from_asm": sum(i.in_asm for i in mf), "mfma_loops": len(innermost_mfma_loops(insts)),
                       "scratch": sum(i.op.startswith("scratch_") for i in insts)}
    return findings, stats


def main(argv: Optional[List[str]] = None) -> int:
    bad = 0
    for path in (argv or sys.argv[1:]):
        findings, stats = scan_text(open(path).read())
        for k, v in stats.items():
            print(f"{path}: {k}: {v}")
        for f in findings:
            print("FINDING", f)
        bad += len(findings)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
