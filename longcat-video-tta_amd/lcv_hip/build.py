"""Build liblcv_hip.so (gfx950 only) from csrc/*.hip with hipcc, in-tree.

Objects are rebuilt only when their source (or a header) is newer, so repeated
calls are cheap.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR.parent / "csrc"
INCLUDE = PKG_DIR.parents[1] / "include"
BUILD = CSRC / "build"
LIB = PKG_DIR / "liblcv_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value",
         "-fno-gpu-rdc", "-I", str(INCLUDE)]


def _newest_header():
    hs = list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))
    return max(h.stat().st_mtime for h in hs)


# per-file extras: the attention kernels never see NaNs (finite inputs, -inf only as the mask value), so their max
# chains may drop IEEE sNaN quieting (bare v_max3_f32 instead of a canonicalising v_max per MFMA output)
_ATTN_FLAGS = ["-fno-honor-nans", "-mno-amdgpu-ieee", "-fno-slp-vectorize"]
EXTRA = {"attn_fwd.hip": _ATTN_FLAGS, "attn_fwd_pipe.hip": _ATTN_FLAGS, "attn_bwd_dq2.hip": _ATTN_FLAGS, "attn_bwd_dkv2.hip": _ATTN_FLAGS,
         "attn_fwd_w64.hip": _ATTN_FLAGS}


def _compile(src: Path, hdr_mtime: float) -> Path:
    obj = BUILD / (src.stem + ".o")
    if obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_mtime):
        return obj
    cmd = [HIPCC, *FLAGS, *EXTRA.get(src.name, []), "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(verbose: bool = False) -> Path:
    BUILD.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError(f"no HIP sources under {CSRC}")
    hdr = _newest_header()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, hdr), srcs))
    if LIB.exists() and all(LIB.stat().st_mtime > o.stat().st_mtime for o in objs):
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
