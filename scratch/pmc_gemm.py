"""For rocprofv3 --pmc: the K3 output projection (93 600 x 4 096 x 4 096) on the default 8-phase kernel, the four-wave stream
kernel, its 8-wave form and hipBLASLt (torch), a few launches each."""
import os, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
dev = "cuda"; bf = torch.bfloat16
M, N, K = 93600, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
a = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.02).to(bf); b = torch.randn(N, device=dev).to(bf)
for tile in ("9", "4", "5"):
    os.environ["LCV_GEMM_TILE"] = tile
    for _ in range(6):
        ops.gemm_nt(a, w, b)
os.environ.pop("LCV_GEMM_TILE")
for _ in range(6):
    torch.nn.functional.linear(a, w, b)
torch.cuda.synchronize()
print("done")
