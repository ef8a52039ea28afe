"""What launch configuration does hipBLASLt pick for the K3 projection shapes?  Run under
`rocprofv3 --kernel-trace --output-format csv` and read Workgroup_Size / LDS / VGPR / Accum_VGPR of the Cijk kernel."""
import torch
a = torch.randn(46800, 4096, device="cuda", dtype=torch.bfloat16)
w = torch.randn(12288, 4096, device="cuda", dtype=torch.bfloat16) * 0.02
b = torch.randn(12288, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    y = torch.nn.functional.linear(a, w, b)
torch.cuda.synchronize()
