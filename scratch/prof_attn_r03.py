"""Round-3 attention kernels for rocprofv3 counter passes: the K3 forward (B = 2 CFG launch, 46 800 x 46 800) with the pipelined
two-waves-per-SIMD kernel (LCV_ATTN_FWD_W64=0) and the 64-rows-per-wave kernel (default), and the backward at K3-TTA."""
import math, os, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
dev = "cuda"; bf = torch.bfloat16
H, D = 32, 128
g = torch.Generator(device=dev).manual_seed(0)
def rmsn(t): return t * torch.rsqrt(t.float().pow(2).mean(-1, keepdim=True) + 1e-6)
def qkv(B, N):
    q = (rmsn(torch.randn(B, N, H, D, device=dev, generator=g)) * (D ** -0.5 * math.log2(math.e))).to(bf)
    k = rmsn(torch.randn(B, N, H, D, device=dev, generator=g)).to(bf)
    v = torch.randn(B, N, H, D, device=dev, generator=g).to(bf)
    return q, k, v
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
what = sys.argv[2] if len(sys.argv) > 2 else "fwd,bwd"
if "fwd" in what:
    q, k, v = qkv(2, 46800)
    o = torch.empty_like(q)
    for w64 in ("0", "1"):
        os.environ["LCV_ATTN_FWD_W64"] = w64
        for _ in range(reps):
            ops.attention(q, k, v, math.log(2.0), out=o)
    torch.cuda.synchronize()
    del q, k, v, o
if "bwd" in what:
    N, nc = 25200, 14400
    q, k, v = qkv(1, N)
    o = torch.empty_like(q)
    _, l1 = ops.attention(q[:, :nc], k[:, :nc], v[:, :nc], math.log(2.0), out=o[:, :nc], need_lse=True)
    _, l2 = ops.attention(q[:, nc:], k, v, math.log(2.0), out=o[:, nc:], need_lse=True)
    do = torch.randn(1, N, H, D, device=dev, generator=g).to(bf)
    dq = torch.zeros_like(q); dk = torch.zeros_like(k); dv = torch.zeros_like(v)
    for _ in range(reps):
        ops.attention_bwd(q[:, :nc], k[:, :nc], v[:, :nc], o[:, :nc], do[:, :nc], l1, dq[:, :nc], dk[:, :nc], dv[:, :nc], math.log(2.0), accumulate_kv=False)
        ops.attention_bwd(q[:, nc:], k, v, o[:, nc:], do[:, nc:], l2, dq[:, nc:], dk, dv, math.log(2.0), accumulate_kv=True)
    torch.cuda.synchronize()
print("done")
