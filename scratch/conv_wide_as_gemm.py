"""conv_wide.h (256 x 192 tile, 8 MFMA + 8 loader waves) timed as a plain GEMM: a 1 x 1 x 1 convolution over M "pixels"."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "longcat-video-tta_amd"))
import torch
from lcv_hip.lib import call
from lcv_hip import ops, lib
dev = "cuda"
zero = torch.zeros(256, dtype=torch.bfloat16, device=dev)
for (M, N, K) in ((46800, 12288, 4096), (46800, 4224, 4096), (46800, 22080, 4096), (46800, 4224, 11008 // 64 * 64)):
    x = torch.randn((1, 1, 1, M, K), device=dev, dtype=torch.bfloat16)
    w = torch.randn((N, K), device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    out = torch.empty((1, 1, 1, M, N), device=dev, dtype=torch.bfloat16)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 5
    for i in range(n + 2):
        if i == 2: ev[0].record()
        call("lcv_causal_conv3d", x.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), zero.data_ptr(), 1, 1, 1, M, K, N, N, 1, 1, 1, 0, ops._stream())
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / n
    ref = ops.gemm_nt(x.view(M, K), w, b)
    for i in range(n + 2):
        if i == 2: ev[0].record()
        ops.gemm_nt(x.view(M, K), w, b)
    ev[1].record(); torch.cuda.synchronize()
    ms2 = ev[0].elapsed_time(ev[1]) / n
    same = torch.equal(ref, out.view(M, N))
    print(f"M {M} N {N} K {K}: {lib.load().lcv_conv3d_last_kernel().decode()} {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.0f} TF/s | gemm_nt {ms2:.3f} ms {2.0 * M * N * K / ms2 / 1e9:.0f} TF/s | equal {same}", flush=True)
