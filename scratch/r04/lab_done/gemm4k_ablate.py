"""lab: gemm4k ablations (scratch/r04/liblcv_hip_lab.so built with -DLCV_GEMM4K_LAB; results wrong on purpose except lab 0 / 32)"""
import os, sys, shutil, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
import lcv_hip.lib as L
from pathlib import Path
L._LIB_PATH = Path("scratch/r04/liblcv_hip_lab.so").resolve()
from lcv_hip import ops
dev = "cuda"; bf = torch.bfloat16
def rn(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(bf).to(dev)
def timeit(fn, n=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
M = 93600
NAMES = {"0": "full", "32": "dma spread 1/8", "16": "raw epi", "2": "no dma", "18": "no dma + raw epi", "48": "spread + raw epi",
         "96": "spread + nt stores", "160": "spread + no drain", "288": "spread + xcd skew", "480": "spread+nt+nodrain+skew"}
for (N, K, name) in ((12288, 4096, "qkv"), (4096, 4096, "proj"), (22016, 4096, "w13")):
    a = rn(M, K, seed=11); w = rn(N, K, seed=12, scale=0.02); b = rn(N, seed=13)
    out = torch.empty(M, N, device=dev, dtype=bf)
    fl = 2 * M * N * K
    os.environ["LCV_GEMM_TILE"] = "9"
    ms = timeit(lambda: ops.gemm_nt(a, w, b, out=out)); print(f"{name} [9]: {ms:.3f} ms {fl / ms / 1e9:.0f}", flush=True)
    os.environ["LCV_GEMM_TILE"] = "k"
    for lab in ("0", "32", "96", "160", "288", "480", "48", "32"):
        os.environ["LCV_GEMM_LAB"] = lab
        ms = timeit(lambda: ops.gemm_nt(a, w, b, out=out)); print(f"{name} [k lab {lab:>2} {NAMES[lab]:<20}]: {ms:.3f} ms {fl / ms / 1e9:.0f}", flush=True)
    ms = timeit(lambda: torch.nn.functional.linear(a, w, b)); print(f"{name} [hipblaslt]: {ms:.3f} ms {fl / ms / 1e9:.0f}", flush=True)
    del a, w, out; torch.cuda.empty_cache()
