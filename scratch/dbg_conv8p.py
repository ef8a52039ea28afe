import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "longcat-video-tta_amd"))
import torch
from longcat_video.modules.vae_wan import AutoencoderKLWan, _Conv
BF16 = torch.bfloat16
vae = AutoencoderKLWan(base_dim=16, z_dim=4, device="cuda", dtype=BF16)
g = torch.Generator().manual_seed(0)
for (ci, co, k, up, shape) in ((64, 192, (3, 3, 3), False, (3, 5, 7)), (192, 384, (1, 1, 1), False, (3, 5, 7)), (128, 256, (3, 3), True, (3, 5, 7)), (64, 192, (3, 3, 3), False, (3, 40, 50)), (64, 192, (1, 1, 1), False, (1, 16, 16)), (128, 192, (1, 1, 1), False, (1, 16, 16))):
    conv = _Conv(ci, co, k, device="cuda", dtype=BF16)
    with torch.no_grad():
        conv.weight.copy_(torch.randn((co, ci) + k, generator=g) * (ci * 9) ** -0.5); conv.bias.zero_()
    x = torch.randn((1,) + shape + (ci,), generator=g).to(BF16).cuda()
    os.environ.pop("LCV_CONV_8P", None)
    a = vae._conv(x, conv, up2x=up).float()
    os.environ["LCV_CONV_8P"] = "0"
    b = vae._conv(x, conv, up2x=up).float()
    d = (a - b).abs()
    bad = (d > 1e-3).nonzero()
    print(ci, co, k, up, shape, "max diff", d.max().item(), "n bad", len(bad), "first bad", bad[:3].tolist(), "last bad", bad[-2:].tolist())
    if len(bad):
        chan = torch.unique(bad[:, -1]); print("   bad channels", chan[:10].tolist(), "...", chan[-3:].tolist(), "bad pixel rows (flattened)", torch.unique(bad[:, 1] * shape[1] * shape[2] * (4 if up else 1) + bad[:, 2] * shape[2] * (2 if up else 1) + bad[:, 3])[:12].tolist())
