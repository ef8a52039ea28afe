"""tn_skinny (LoRA parameter gradients) at the K3-TTA shapes: time and GB/s of the streamed operand."""
import sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
dev = "cuda"; bf = torch.bfloat16
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (M, K) in ((25200, 4096), (25200, 12288), (6240, 4096), (6240, 12288)):
    g = torch.randn(M, 64, device=dev).to(bf); x = torch.randn(M, K, device=dev).to(bf)
    ref = (g[:, :8].float().t() @ x.float())
    out = ops.tn_skinny(g, x, 8)
    err = ((out - ref).norm() / ref.norm()).item()
    ms = timeit(lambda: ops.tn_skinny(g, x, 8))
    print(f"M={M} K={K}: {ms*1e3:7.1f} us  {M*K*2/ms/1e6:7.1f} GB/s  rel err {err:.2e}", flush=True)
