"""CPU: the compiler's output for every source that issues MFMAs from `asm` passes the mechanical hazard screen of
tools/isa_hazards.py — and the screen demonstrably fails on un-guarded forms (kept below as fixture strings, the way
tests/test_first_contact_guards.py keeps bent stand-ins).

Why: 59 % of the headline step runs on `attn_fwd_w64_kernel`, whose MFMAs hipcc's hazard recogniser cannot see.  Its correctness
depends on where the register allocator puts copies relative to those asm statements; that placement changed four times during
round 3 and each time produced wrong results that only a shape-parametrised parity test caught (profiles/r03_attn_bwd_lab.md).
A ROCm upgrade must fail HERE, on the CPU, not as a silent numeric drift on the GPU box.
"""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT / "longcat-video-tta_amd"))
import isa_hazards as H  # noqa: E402

ASM_MFMA_SOURCES = ["attn_fwd_w64.hip", "attn_fwd_pipe.hip", "gemm.hip"]   # gemm.hip includes gemm4k.h


def _sources_with_asm_mfma():
    csrc = ROOT / "longcat-video-tta_amd" / "csrc"
    hits = set()
    for f in list(csrc.glob("*.hip")) + list(csrc.glob("*.h")):
        t = f.read_text()
        if "asm volatile(\"v_mfma" in t or "\\tv_mfma" in t or "s_nop 3\\n\\tv_mfma" in t:
            hits.add(f.name)
    return hits


def test_the_list_of_asm_mfma_sources_is_complete():
    hits = _sources_with_asm_mfma()
    assert hits == {"attn_fwd_w64.hip", "attn_fwd_pipe.hip", "gemm4k.h"}, hits


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    """hipcc --cuda-device-only -S with the library's own flags (lcv_hip/build.py) for the four sources, in parallel."""
    from lcv_hip import build as B
    out = tmp_path_factory.mktemp("isa")

    def one(name):
        dst = out / (name + ".s")
        cmd = [B.HIPCC, *B.FLAGS, *B.EXTRA.get(name, []), "--cuda-device-only", "-S", str(B.CSRC / name), "-o", str(dst)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return name, dst.read_text()
    with ThreadPoolExecutor(max_workers=3) as ex:
        return dict(ex.map(one, ASM_MFMA_SOURCES))


def test_no_hazard_findings_in_the_shipped_kernels(isa):
    seen_asm_mfma = 0
    for name, text in isa.items():
        findings, stats = H.scan_text(text)
        assert not findings, f"{name}:\n" + "\n".join(findings)
        seen_asm_mfma += sum(s["mfma_from_asm"] for s in stats.values())
        if name == "attn_fwd_w64.hip":
            (k, s), = stats.items()
            assert "attn_fwd_w64_kernel" in k and s["mfma"] == s["mfma_from_asm"] > 300 and s["mfma_loops"] >= 2
    assert seen_asm_mfma > 700           # the scan really walked the asm-issued MFMAs (w64 416, gemm4k 3 x 128, pipe 6)


# ---------------------------------------------------------------------------------------------------------------- bent stand-ins
_HEAD = "_Z4bentv:\n"
_TAIL = "\ts_endpgm\n.Lfunc_end0:\n"
_MFMA = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 v[0:15], v[100:103], a[4:7], v[0:15]\n\t;;#ASMEND\n"
_GUARDED = "\t;;#ASMSTART\n\ts_nop 3\n\tv_mfma_f32_32x32x16_bf16 v[0:15], v[100:103], a[4:7], v[0:15]\n\t;;#ASMEND\n"
BENT = {
    # the allocator's copy of an A-operand register directly in front of the asm MFMA (round-3 bug class 2)
    "vmov_into_A": "\tv_mov_b32_e32 v101, v7\n" + _MFMA,
    # an AGPR B operand materialised right before its first use (bug class 1)
    "accvgpr_write_into_B": "\tv_accvgpr_write_b32 a5, v9\n\ts_add_u32 s4, s4, 1\n" + _MFMA,
    # an accumulator copied between code paths, three instructions before the MFMA that reads it as SrcC
    "accumulator_copy_three_back": "\tv_mov_b32_e32 v3, v40\n\ts_add_u32 s4, s4, 1\n\ts_cmp_lt_u32 s4, s5\n" + _MFMA,
    # the author's own pack one instruction before the MFMA that reads it: fewer than two wait states
    "asm_pack_too_close": "\t;;#ASMSTART\n\tv_cvt_pk_bf16_f32 v100, v1, v2\n\t;;#ASMEND\n" + _MFMA,
}
GOOD = {
    "guarded": "\tv_mov_b32_e32 v101, v7\n" + _GUARDED,
    "copy_five_back": "\tv_mov_b32_e32 v101, v7\n" + "\ts_add_u32 s4, s4, 1\n" * 4 + _MFMA,
    "copy_of_an_unrelated_register": "\tv_mov_b32_e32 v50, v7\n" + _MFMA,
    "asm_pack_two_states_away": "\t;;#ASMSTART\n\tv_cvt_pk_bf16_f32 v100, v1, v2\n\t;;#ASMEND\n\ts_add_u32 s4, s4, 1\n\ts_add_u32 s4, s4, 1\n" + _MFMA,
    "compiler_pair_is_the_compilers_business": "\tv_mov_b32_e32 v101, v7\n\tv_mfma_f32_32x32x16_bf16 v[0:15], v[100:103], a[4:7], v[0:15]\n",
    "mfma_chain_is_interlocked": _MFMA + _MFMA,
}


@pytest.mark.parametrize("name", sorted(BENT))
def test_the_scan_fails_on_an_unguarded_form(name):
    findings, _ = H.scan_text(_HEAD + BENT[name] + _TAIL)
    assert len(findings) == 1 and "H1" in findings[0], findings


@pytest.mark.parametrize("name", sorted(GOOD))
def test_the_scan_accepts_the_guarded_and_the_harmless_forms(name):
    findings, _ = H.scan_text(_HEAD + GOOD[name] + _TAIL)
    assert not findings, findings


def test_the_scan_flags_spills_in_the_steady_loop_and_compiler_uses_of_m0():
    loop = (".LBB0_1:\n" + _MFMA * 4 + "\tscratch_load_dword v60, off, off offset:8\n\ts_cmp_lt_u32 s4, s5\n\ts_cbranch_scc1 .LBB0_1\n")
    tail = (".LBB0_2:\n" + _MFMA + "\tscratch_load_dword v61, off, off\n\ts_cbranch_scc1 .LBB0_2\n")      # a shorter MFMA loop: tolerated
    findings, _ = H.scan_text(_HEAD + loop + tail + _TAIL)
    assert len(findings) == 1 and "H2" in findings[0] and "offset:8" in findings[0]
    clean = loop.replace("\tscratch_load_dword v60, off, off offset:8\n", "")
    assert not H.scan_text(_HEAD + clean + tail + _TAIL)[0]
    m0 = "\t;;#ASMSTART\n\ts_mov_b32 m0, s7\n\t;;#ASMEND\n" + _GUARDED + "\ts_mov_b32 m0, s9\n"
    findings, _ = H.scan_text(_HEAD + m0 + _TAIL)
    assert len(findings) == 1 and "H3" in findings[0]
