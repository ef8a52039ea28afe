"""Build container only: the REFERENCE's `load_video_frames` and the ground-truth leg of `evaluate_generation_metrics`
(delta_experiment/scripts/common.py:103-155, 663-731) run over tests/fake_av.py's seeded clips -> tests/golden/video_io.pt."""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE)); sys.path.insert(0, str(HERE.parent))
import fake_av  # noqa: E402
import _ref_loader as RL  # noqa: E402

fake_av.install()
RL.stub_longcat()
RL.add_reference_paths()
import common as ref  # noqa: E402

CASES = [  # (path, num_frames, height, width, start_frame)
    ("fake://11/9/20x28", 5, 24, 40, 0),       # upscale, a window from the start
    ("fake://12/9/20x28", 5, 12, 16, 3),       # downscale, skipped prefix
    ("fake://13/4/16x16", 7, 16, 16, 1),       # short clip: the last frame repeated
    ("fake://14/6/18x30", 6, 18, 30, 0),       # identity size
]
out = {"frames": []}
for path, n, h, w, s in CASES:
    t = ref.load_video_frames(path, n, height=h, width=w, start_frame=s)
    out["frames"].append({"path": path, "num_frames": n, "height": h, "width": w, "start_frame": s, "pixels": t.clone()})
    print(path, tuple(t.shape), float(t.min()), float(t.max()))
# ground-truth leg: PSNR / SSIM of a seeded "generated" clip against frames gen_start .. of the fake video, resized LANCZOS
g = np.random.RandomState(5).rand(3 + 4, 24, 32, 3).astype(np.float32)
m = ref.evaluate_generation_metrics(g, "fake://21/12/20x28", num_cond_frames=3, num_gen_frames=4, gen_start_frame=6, device="cpu")
out["gt"] = {"path": "fake://21/12/20x28", "gen_seed": 5, "gen_shape": [7, 24, 32, 3], "num_cond_frames": 3, "num_gen_frames": 4,
             "gen_start_frame": 6, "psnr": m["psnr"], "ssim": m["ssim"]}
m2 = ref.evaluate_generation_metrics(g, "fake://22/8/20x28", num_cond_frames=3, num_gen_frames=4, gen_start_frame=6, device="cpu")
out["gt_short"] = {"path": "fake://22/8/20x28", "psnr": m2["psnr"], "ssim": m2["ssim"]}       # only 2 ground-truth frames left
print(out["gt"], out["gt_short"])
torch.save(out, HERE / "video_io.pt")
