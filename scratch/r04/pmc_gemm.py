"""For rocprofv3 --pmc (round 4): the K3 projection shapes at M = 93 600 on the 8-phase kernel, gemm4k (per tile-group
setting in PMC_GROUPS) and hipBLASLt (torch), a few launches each.  argv: shape name (proj|qkv|w13|w2)."""
import os, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
import lcv_hip.lib as L
dev = "cuda"; bf = torch.bfloat16
SH = {"proj": (4096, 4096), "qkv": (12288, 4096), "w13": (22016, 4096), "w2": (4096, 11008)}
name = sys.argv[1] if len(sys.argv) > 1 else "proj"
N, K = SH[name]
M = 93600
a = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.02).to(bf); b = torch.randn(N, device=dev).to(bf)
# launches are told apart by their order in the kernel trace: 6 x 8-phase, then 6 x gemm4k per entry of GROUPS (tile rows per group)
GROUPS = os.environ.get("PMC_GROUPS", "3").split(",")
L.set_knob("LCV_GEMM_TILE", "9")
for _ in range(6):
    ops.gemm_nt(a, w, b)
L.set_knob("LCV_GEMM_TILE", "k")
for g in GROUPS:
    L.set_knob("LCV_GEMM_GROUP_M", g)
    for _ in range(6):
        ops.gemm_nt(a, w, b)
L.set_knob("LCV_GEMM_GROUP_M", None)
L.set_knob("LCV_GEMM_TILE", None)
for _ in range(6):
    torch.nn.functional.linear(a, w, b)
torch.cuda.synchronize()
print("done")
