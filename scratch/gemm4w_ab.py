"""Four-wave 128x128 GEMM (LCV_GEMM_TILE=4, csrc/gemm4w.h) against the 8-phase kernel (=9) and hipBLASLt in one process:
bit-equality (same fp32 accumulation order per output element: K ascending in 32-deep steps... checked, not assumed) and time."""
import os, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
dev = "cuda"; bf = torch.bfloat16
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"

def timeit(fn, n=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def run(tile, fn):
    os.environ["LCV_GEMM_TILE"] = tile
    try:
        return fn()
    finally:
        os.environ.pop("LCV_GEMM_TILE", None)

g = torch.Generator(device=dev).manual_seed(0)
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev, generator=g) * scale).to(bf)

# ---- correctness: edge rows, LoRA K tiles, epilogues, small grids ----
cases = [
    ("plain 300x512x256 (few tiles)", dict(M=300, N=512, K=256)),
    ("edge M 4099, N 768, K 1024 + bias", dict(M=4099, N=768, K=1024, bias=True)),
    ("N edge 2048x(256+64)x512", dict(M=2048, N=320, K=512, bias=True)),
    ("lora K2=64, 5000x1024x512", dict(M=5000, N=1024, K=512, K2=64, bias=True)),
    ("lora K2=128 odd nk, 2500x512x192", dict(M=2500, N=512, K=192, K2=128)),
    ("gate-residual 4800x1024x1024", dict(M=4800, N=1024, K=1024, bias=True, epi="gate")),
    ("swiglu 4100x2048x512", dict(M=4100, N=2048, K=512, epi="swiglu")),
    ("gelu f32 out 2100x512x512", dict(M=2100, N=512, K=512, bias=True, epi="gelu", f32=True)),
    ("many tiles 70000x1024x256", dict(M=70000, N=1024, K=256, bias=True)),
]
bad = 0
for name, c in cases:
    M, N, K = c["M"], c["N"], c["K"]
    a = rnd(M, K); w = rnd(N, K, scale=0.05)
    kw = {}
    if c.get("bias"): kw["bias"] = rnd(N)
    if c.get("K2"):
        kw["a2"] = rnd(M, c["K2"]); kw["w2"] = rnd(N, c["K2"], scale=0.05)
    epi = c.get("epi")
    if epi == "gate":
        kw.update(epilogue=ops.LCV_EPI_GATE_RESIDUAL, resid=rnd(M, N), mod=torch.randn(M // 1200 + 1, 6 * N, device=dev, generator=g), gate_idx=2, rows_per_frame=1200)
    elif epi == "swiglu":
        kw.update(epilogue=ops.LCV_EPI_SWIGLU)
    elif epi == "gelu":
        kw.update(epilogue=ops.LCV_EPI_GELU_TANH)
    if c.get("f32"): kw["out_f32"] = True
    bias = kw.pop("bias", None)
    ref = run("9", lambda: ops.gemm_nt(a, w, bias, **kw))
    out = run(os.environ.get("G4_MODE", "4"), lambda: ops.gemm_nt(a, w, bias, **kw))
    torch.cuda.synchronize()
    eq = torch.equal(out, ref)
    d = (out.float() - ref.float()).abs().max().item()
    nan = bool(torch.isnan(out.float()).any())
    print(f"{name:45s} equal {eq}  max|diff| {d:.3e}  nan {nan}", flush=True)
    bad += (not eq)
print("MISMATCHES", bad, flush=True)
if quick or bad:
    sys.exit(1 if bad else 0)

# ---- time at the K3 shapes ----
for (M, N, K, name) in ((93600, 12288, 4096, "qkv"), (93600, 4096, 4096, "proj"), (93600, 22016, 4096, "w13"), (93600, 4096, 11008, "w2")):
    a = rnd(M, K); w = rnd(N, K, scale=0.02); b = rnd(N)
    for tile in ("9", "3", "9", "3"):
        ms = run(tile, lambda: timeit(lambda: ops.gemm_nt(a, w, b)))
        print(f"  {name:5s} tile {tile}: {ms:7.2f} ms  {2 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
    ms = timeit(lambda: torch.nn.functional.linear(a, w, b))
    print(f"  {name:5s} hipBLASLt: {ms:7.2f} ms  {2 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
    del a, w
