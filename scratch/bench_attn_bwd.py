"""Attention backward at the K3-TTA shape (cond 14 400 + noise 10 800 queries over 25 200 keys, 32 heads), one process:
A/B of environment knobs (LCV_ATTN_XCD, ...).  Algorithmic work 10 * C * (Nc^2 + Nn * N) = 19.6 TF per layer."""
import math, os, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
dev = "cuda"; bf = torch.bfloat16
H, D, N, nc = 32, 128, 25200, 14400
g = torch.Generator(device=dev).manual_seed(0)
def rmsn(t): return t * torch.rsqrt(t.float().pow(2).mean(-1, keepdim=True) + 1e-6)
# the product's layouts: q, k = slots of the roped [B, N, 2, H, D] buffer, v = slot 2 of the packed qkv GEMM output [B, N, 3, H, D]
qk_buf = torch.empty(1, N, 2, H, D, device=dev, dtype=bf); qkv_buf = torch.empty(1, N, 3, H, D, device=dev, dtype=bf)
qk_buf[:, :, 0] = (rmsn(torch.randn(1, N, H, D, device=dev, generator=g)) * (D ** -0.5 * math.log2(math.e))).to(bf)
qk_buf[:, :, 1] = rmsn(torch.randn(1, N, H, D, device=dev, generator=g)).to(bf)
qkv_buf[:, :, 2] = torch.randn(1, N, H, D, device=dev, generator=g).to(bf)
q, k, v = qk_buf[:, :, 0], qk_buf[:, :, 1], qkv_buf[:, :, 2]
o = torch.empty(1, N, H, D, device=dev, dtype=bf)
_, l1 = ops.attention(q[:, :nc], k[:, :nc], v[:, :nc], math.log(2.0), out=o[:, :nc], need_lse=True)
_, l2 = ops.attention(q[:, nc:], k, v, math.log(2.0), out=o[:, nc:], need_lse=True)
do = torch.randn(1, N, H, D, device=dev, generator=g).to(bf)
def run(dq, dk, dv):
    ops.attention_bwd(q[:, :nc], k[:, :nc], v[:, :nc], o[:, :nc], do[:, :nc], l1, dq[:, :nc], dk[:, :nc], dv[:, :nc], math.log(2.0), accumulate_kv=False)
    ops.attention_bwd(q[:, nc:], k, v, o[:, nc:], do[:, nc:], l2, dq[:, nc:], dk, dv, math.log(2.0), accumulate_kv=True)
def timeit(fn, n=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
knob = sys.argv[1] if len(sys.argv) > 1 else "LCV_ATTN_XCD"
vals = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1"]
flops = 10.0 * H * D * (nc * nc + (N - nc) * N)
res = {x: [] for x in vals}; outs = {}
for rep in range(3):
    for x in vals:
        os.environ[knob] = x
        dq = torch.zeros(1, N, H, D, device=dev, dtype=bf); dk = torch.zeros_like(dq); dv = torch.zeros_like(dq)
        res[x].append(timeit(lambda: run(dq, dk, dv)))
        outs[x] = (dq, dk, dv)
for x in vals:
    print(f"{knob}={x}: best {min(res[x]):.2f} ms per layer = {flops / min(res[x]) / 1e9:.0f} TF/s algorithmic ({flops / min(res[x]) / 1e9 / 2500:.3f} of peak)  all {[round(t, 2) for t in res[x]]}", flush=True)
a, b_ = outs[vals[0]], outs[vals[-1]]
for nme, x, y in zip(("dq", "dk", "dv"), a, b_):
    print(f"   {nme}: rel-L2 between {vals[0]} and {vals[-1]}: {((x.float() - y.float()).norm() / x.float().norm()).item():.2e}", flush=True)
