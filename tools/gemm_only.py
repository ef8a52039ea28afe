#!/usr/bin/env python3
"""A few K3-shaped GEMM and attention launches for PMC collection under rocprofv3."""
import sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "longcat-video-tta_amd"))
import torch
from lcv_hip import ops
M, N, K = 46800, 12288, 4096
a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm_nt(a, w, b)
y = torch.nn.functional.linear(a, w, b)
qkv = torch.randn(1, M, 3, 32, 128, device="cuda", dtype=torch.bfloat16)
o = torch.empty(1, M, 32, 128, device="cuda", dtype=torch.bfloat16)
for _ in range(2):
    ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 128 ** -0.5, out=o)
torch.cuda.synchronize()
print("done")
