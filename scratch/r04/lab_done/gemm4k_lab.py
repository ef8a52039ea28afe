"""lab: gemm4k with LCV_GEMM_LAB start-skew values, qkv + w13 + w2 shapes, vs tile 9 and hipBLASLt"""
import os, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
dev = "cuda"; bf = torch.bfloat16
def rn(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(bf).to(dev)
def timeit(fn, n=8, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
M = 93600
labs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "2", "3", "4", "6"]
for (N, K, name) in ((12288, 4096, "qkv"), (22016, 4096, "w13"), (4096, 11008, "w2")):
    a = rn(M, K, seed=11); w = rn(N, K, seed=12, scale=0.02); b = rn(N, seed=13)
    fl = 2 * M * N * K
    row = []
    os.environ["LCV_GEMM_TILE"] = "9"
    ms = timeit(lambda: ops.gemm_nt(a, w, b)); row.append(f"[9] {ms:.3f} {fl / ms / 1e9:.0f}")
    os.environ["LCV_GEMM_TILE"] = "k"
    for lab in labs:
        os.environ["LCV_GEMM_LAB"] = lab
        ms = timeit(lambda: ops.gemm_nt(a, w, b)); row.append(f"[k lab {lab}] {ms:.3f} {fl / ms / 1e9:.0f}")
    ms = timeit(lambda: torch.nn.functional.linear(a, w, b)); row.append(f"[hipblaslt] {ms:.3f} {fl / ms / 1e9:.0f}")
    print(f"{name}: " + " | ".join(row), flush=True)
    del a, w; torch.cuda.empty_cache()
