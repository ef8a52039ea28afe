"""Time the HIP UMT5-XXL text encoder (24 layers, d_model 4096, 64 heads, d_ff 10240; synthetic weights) on one prompt of
512 padded tokens — the once-per-video step of delta_experiment/scripts/common.py:228-255.  usage: python tools/umt5_time.py [layers]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "longcat-video-tta_amd"))
import torch  # noqa: E402

from longcat_video.modules.umt5_encoder import UMT5EncoderModel  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 24
enc = UMT5EncoderModel(device="cuda", num_layers=layers).init_synthetic_()
ids = torch.randint(0, 256384, (1, 512), device="cuda")
mask = torch.zeros(1, 512, dtype=torch.long, device="cuda"); mask[:, :77] = 1
for _ in range(2):
    out = enc(ids, mask).last_hidden_state
torch.cuda.synchronize()
t0 = time.time()
n = 5
for _ in range(n):
    out = enc(ids, mask).last_hidden_state
torch.cuda.synchronize()
dt = (time.time() - t0) / n
flops = layers * 2 * 512 * (4 * 4096 * 4096 + 3 * 4096 * 10240) + layers * 4 * 512 * 512 * 4096
print(f"UMT5 encoder, {layers} layers, 512 tokens: {dt * 1e3:.1f} ms per prompt ({flops / dt / 1e12:.0f} TFLOP/s over the GEMM+attention flops), "
      f"finite={bool(torch.isfinite(out.float()).all())}, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
