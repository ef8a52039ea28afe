"""A/B in one process: the phase-ordered forward kernel (LCV_ATTN_PIPE=0) vs the software-pipelined one, unit-scale path."""
import math, os, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
dev = "cuda"; bf = torch.bfloat16
def timeit(fn, n=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
shapes = [(1, 32, 46800, 46800), (2, 32, 10800, 25200), (1, 32, 6240, 12480)] if len(sys.argv) < 2 else [tuple(int(x) for x in sys.argv[1].split(","))]
for (B, H, Nq, Nk) in shapes:
    D = 128
    g = torch.Generator(device=dev).manual_seed(0)
    def rmsn(t): return t * torch.rsqrt(t.float().pow(2).mean(-1, keepdim=True) + 1e-6)
    q = (rmsn(torch.randn(B, Nq, H, D, device=dev, generator=g)) * (D ** -0.5 * math.log2(math.e))).to(bf)
    k = rmsn(torch.randn(B, Nk, H, D, device=dev, generator=g)).to(bf)
    v = torch.randn(B, Nk, H, D, device=dev, generator=g).to(bf)
    VARS = ("0", "1", "1p1", "1p2", "1p3")
    outs = {}; res = {v: [] for v in VARS}
    for rep in range(3):
        for var in VARS:
            os.environ["LCV_ATTN_PIPE"] = var[0]
            os.environ["LCV_ATTN_PIPE_PRIO"] = var[2] if len(var) > 1 else "0"
            o = torch.empty(B, Nq, H, D, device=dev, dtype=bf)
            res[var].append(timeit(lambda: ops.attention(q, k, v, math.log(2.0), out=o)))
            outs[var] = o
    fl = 4.0 * B * H * Nq * Nk * D
    for var in VARS:
        print(f"B{B} H{H} Nq{Nq} Nk{Nk} pipe={var}: best {min(res[var]):.2f} ms = {fl/min(res[var])/1e9:.0f} TF/s  all {[round(x,2) for x in res[var]]}", flush=True)
    d = outs["0"].float() - outs["1"].float()
    print("   rel_l2 pipe vs phase-ordered:", (d.norm() / outs["0"].float().norm()).item(), "max abs", d.abs().max().item(), "finite", bool(torch.isfinite(outs["1"].float()).all()), flush=True)
