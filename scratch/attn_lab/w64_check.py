"""attn_fwd_w64_kernel (LCV_ATTN_FWD_W64=1) against attn_fwd_pipe_kernel on the same inputs: bits, rel-L2, time."""
import os, sys, math, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "longcat-video-tta_amd"))
from lcv_hip import ops
D = 128
def run(q, k, v, which):
    os.environ["LCV_ATTN_FWD_W64"] = which
    return ops.attention(q, k, v, ops.LN2, need_lse=True)
def timed(q, k, v, which, n=5):
    os.environ["LCV_ATTN_FWD_W64"] = which
    for _ in range(2): ops.attention(q, k, v, ops.LN2)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.attention(q, k, v, ops.LN2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(1, 2, 600, 700), (2, 3, 257, 577), (1, 1, 64, 1024), (1, 4, 1000, 3000), (1, 2, 333, 641)]
if len(sys.argv) > 1 and sys.argv[1] == "big": shapes = []
for (B, H, Nq, Nk) in shapes:
    g = torch.Generator().manual_seed(Nq)
    q = (torch.randn(B, Nq, H, D, generator=g) * (D ** -0.5 * math.log2(math.e))).bfloat16().cuda()
    k = torch.randn(B, Nk, H, D, generator=g).bfloat16().cuda()
    v = torch.randn(B, Nk, H, D, generator=g).bfloat16().cuda()
    (o0, l0), (o1, l1) = run(q, k, v, "0"), run(q, k, v, "1")
    ref = torch.softmax((q.float().permute(0, 2, 1, 3) @ k.float().permute(0, 2, 3, 1)) * math.log(2), -1) @ v.float().permute(0, 2, 1, 3)
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    print(f"B{B} H{H} Nq{Nq} Nk{Nk}: w64 vs pipe bits equal {torch.equal(o0, o1)} rel {rel(o1, o0):.2e}; lse equal {torch.equal(l0, l1)}; "
          f"vs fp32: pipe {rel(o0.permute(0, 2, 1, 3), ref):.2e} w64 {rel(o1.permute(0, 2, 1, 3), ref):.2e}", flush=True)
for (B, H, N) in [(1, 32, 14400), (2, 32, 46800)]:
    g = torch.Generator().manual_seed(N)
    q = (torch.randn(B, N, H, D, generator=g) * (D ** -0.5 * math.log2(math.e))).bfloat16().cuda()
    k = torch.randn(B, N, H, D, generator=g).bfloat16().cuda()
    v = torch.randn(B, N, H, D, generator=g).bfloat16().cuda()
    (o0, l0), (o1, l1) = run(q, k, v, "0"), run(q, k, v, "1")
    print(f"B{B} H{H} N{N}: bits equal {torch.equal(o0, o1)} rel {float((o1.float() - o0.float()).norm() / o0.float().norm()):.2e}", flush=True)
    for rep in range(2):
        t0, t1 = timed(q, k, v, "0"), timed(q, k, v, "1")
        fl = 4.0 * B * H * N * N * D
        print(f"   pipe {t0:.3f} ms = {fl / t0 / 1e9:.0f} TF/s ({fl / t0 / 1e9 / 2500:.3f});  w64 {t1:.3f} ms = {fl / t1 / 1e9:.0f} TF/s ({fl / t1 / 1e9 / 2500:.3f})", flush=True)
