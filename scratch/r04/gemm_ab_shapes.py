"""Round 4: same-process A/B of the 8-phase default (LCV_GEMM_TILE=9) and gemm4k (k) at the GEMM shapes the DiT really runs, WITH their
real epilogues: K3 denoise (M = 93 600), K3-TTA (M = 25 200, + a rank-8 LoRA K tile), K2 (40 560), the 480p reference point (12 480),
K5 (96 720).  Prints ms and TF/s per (shape, kernel) and hipBLASLt for the plain-bias shapes."""
import os, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
import lcv_hip.lib as L
from lcv_hip.lib import LCV_EPI_GATE_RESIDUAL, LCV_EPI_SWIGLU
dev = "cuda"; bf = torch.bfloat16
def rn(*s, seed=0, scale=1.0, dtype=bf):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(dtype).to(dev)
def timeit(fn, n=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
Ms = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [93600, 25200, 12480]
lora = len(sys.argv) > 2 and sys.argv[2] == "lora"
for M in Ms:
    T = 13 if M % 13 == 0 else 7
    for (name, N, K, kind) in (("qkv", 12288, 4096, "bias"), ("proj", 4096, 4096, "gate"), ("xq", 4096, 4096, "bias"),
                               ("w13", 22016, 4096, "swiglu"), ("w2", 4096, 11008, "gate")):
        a = rn(M, K, seed=1); w = rn(N, K, seed=2, scale=0.02); b = rn(N, seed=3) if kind != "swiglu" else None
        kw = {}
        if kind == "gate":
            kw.update(epilogue=LCV_EPI_GATE_RESIDUAL, resid=rn(M, N, seed=6), mod=rn(1, T, 6 * N, seed=7, dtype=torch.float32), gate_idx=2, rows_per_frame=M // T)
        if kind == "swiglu":
            kw.update(epilogue=LCV_EPI_SWIGLU)
        if lora and kind != "swiglu":
            kw.update(a2=rn(M, 64, seed=8), w2=rn(N, 64, seed=9, scale=0.05))
        out = torch.empty(M, N // 2 if kind == "swiglu" else N, device=dev, dtype=bf)
        fl = 2 * M * N * K
        row = []
        res = {}
        for rnd in range(2):
            for t in ("9", "k"):
                L.set_knob("LCV_GEMM_TILE", t)
                ms = timeit(lambda: ops.gemm_nt(a, w, b, out=out, **kw))
                row.append(f"[{t}] {ms:.3f} ms {fl / ms / 1e9:.0f}")
                res[t] = out.clone()
        same = torch.equal(res["9"], res["k"])
        if kind == "bias" and not lora:
            ms = timeit(lambda: torch.nn.functional.linear(a, w, b)); row.append(f"[hipblaslt] {ms:.3f} ms {fl / ms / 1e9:.0f}")
        print(f"M={M} {name} N={N} K={K} {kind}{' +lora' if lora and kind != 'swiglu' else ''}: " + " | ".join(row) + f" | bitwise equal: {same}", flush=True)
        del a, w, out, res; torch.cuda.empty_cache()
