"""Two VAE-decoder convolutions alone, for rocprofv3 counter passes: the 96->96 3x3x3 at 720p (T frames) and 192->192 at 360x640."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "longcat-video-tta_amd"))
import torch
from lcv_hip.lib import call
from lcv_hip import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda"
zero = torch.zeros(256, dtype=torch.bfloat16, device=dev)
def run(H, W, cin, cp, cout, n, k=(3, 3, 3), up=0):
    x = torch.randn((1, T, H, W, cp), device=dev, dtype=torch.bfloat16)
    x[..., cin:] = 0
    taps = k[0] * k[1] * k[2]
    w = torch.randn((cout, taps, cp), device=dev, dtype=torch.bfloat16) * 0.02
    w[..., cin:] = 0
    w = w.reshape(cout, taps * cp).contiguous()
    b = torch.zeros(cout, device=dev, dtype=torch.bfloat16)
    ldc = (cout + 63) // 64 * 64
    ldc = cout if cout == 3 else ldc
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    out = torch.zeros((1, T, Ho, Wo, ldc), device=dev, dtype=torch.bfloat16)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(n + 1):
        if i == 1: ev[0].record()
        call("lcv_causal_conv3d", x.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), zero.data_ptr(),
             1, T, H, W, cin, cout, ldc, k[0], k[1], k[2], up, ops._stream())
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / n
    real = 2.0 * T * Ho * Wo * cout * taps * cin
    from lcv_hip import lib as _l
    print(f"conv {cin}->{cout} k{k} up{up} {T}x{Ho}x{Wo}: {ms:.3f} ms  real {real / ms / 1e9:.0f} TF/s  [{_l.load().lcv_conv3d_last_kernel().decode()}]", flush=True)
run(720, 1280, 96, 128, 96, 3)
run(720, 1280, 96, 128, 3, 3)
run(360, 640, 192, 192, 96, 3, k=(1, 3, 3), up=1)
run(360, 640, 192, 192, 192, 3)
run(180, 320, 384, 384, 384, 3)
