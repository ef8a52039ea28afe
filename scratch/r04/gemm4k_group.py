"""lab: gemm4k / 8p vs LCV_GEMM_GROUP_M on qkv and w13"""
import os, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
import lcv_hip.lib as L
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
for (N, K, name) in ((12288, 4096, "qkv"), (22016, 4096, "w13"), (4096, 4096, "proj")):
    a = rn(M, K, seed=11); w = rn(N, K, seed=12, scale=0.02); b = rn(N, seed=13)
    fl = 2 * M * N * K
    for t in ("k", "9"):
        L.set_knob("LCV_GEMM_TILE", t)
        row = []
        for gm in ("1", "2", "3", "4", "6", "8", "12", "16", "32", "64"):
            L.set_knob("LCV_GEMM_GROUP_M", gm)
            ms = timeit(lambda: ops.gemm_nt(a, w, b)); row.append(f"g{gm}: {fl / ms / 1e9:.0f}")
        print(f"{name} [{t}]: " + " | ".join(row), flush=True)
    del a, w; torch.cuda.empty_cache()
