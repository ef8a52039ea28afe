"""Text cross-attention backward at the K3-TTA shape (10 800 noise queries x 77 keys x 32 heads, general-scale kernels)."""
import sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
dev = "cuda"; bf = torch.bfloat16
for Nq in (10800, 25200):
    q = torch.randn(1, Nq, 32, 128, device=dev).to(bf); k = torch.randn(1, 77, 32, 128, device=dev).to(bf); v = torch.randn(1, 77, 32, 128, device=dev).to(bf)
    do = torch.randn(1, Nq, 32, 128, device=dev).to(bf)
    o, lse = ops.attention(q, k, v, 128 ** -0.5, need_lse=True)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    for _ in range(3): ops.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, 128 ** -0.5)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, 128 ** -0.5)
    e.record(); torch.cuda.synchronize()
    print(f"Nq={Nq}: {s.elapsed_time(e) / 20 * 1e3:.1f} us per backward", flush=True)
