#!/usr/bin/env python3
"""Mint golden vectors for the evaluation row from the REFERENCE's own functions (build container only):
`compute_psnr`, `_ssim_single` (the branch that runs when torchmetrics is absent, as it is in this image — the fixture
records that) and `OnlineFrechetAccumulator._accumulate / compute` (delta_experiment/scripts/common.py:617-622, 760-776,
2210-2231, 2316-2326, 2386-2428).  Only DATA is written.  Re-run: python tests/golden/make_eval_golden.py"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden as mg  # noqa: E402

OUT = Path(__file__).resolve().parent


def main():
    mg._stub_longcat()
    sys.path.insert(0, str(mg.REF / "delta_experiment" / "scripts"))
    import common as ref
    g = torch.Generator().manual_seed(77)
    N, H, W = 3, 20, 28
    gt_u8 = torch.randint(0, 256, (N, H, W, 3), generator=g, dtype=torch.uint8)
    gt = (gt_u8.numpy() / 255.0).astype(np.float32)                       # common.py:712-715
    gen = np.clip(gt + 0.1 * torch.randn((N, H, W, 3), generator=g).numpy().astype(np.float32), 0, 1).astype(np.float32)
    gen[2] = gt[2]                                                        # identical frame: the mse < 1e-10 -> 50 dB branch
    out = {"gen": torch.from_numpy(gen), "gt_u8": gt_u8, "psnr_compute_psnr": [], "ssim_single": [],
           "ssim_impl": None}
    try:
        import torchmetrics  # noqa: F401
        out["ssim_impl"] = "torchmetrics"
    except ImportError:
        out["ssim_impl"] = "fallback_global_statistics"
    for i in range(N):
        p = torch.from_numpy(gen[i]).permute(2, 0, 1).unsqueeze(0).float()
        q = torch.from_numpy(gt[i]).permute(2, 0, 1).unsqueeze(0).float()
        out["psnr_compute_psnr"].append(ref.compute_psnr(p, q))
        out["ssim_single"].append(ref._ssim_single(p, q))
    # Frechet accumulator: 6 "videos" of 400-d features each for generated / reference
    acc = ref.OnlineFrechetAccumulator(device="cpu", compute_fid=False, min_videos=4)
    rng = np.random.RandomState(5)
    gf = rng.randn(6, 400) * 0.7 + 0.1
    rf = rng.randn(6, 400)
    for a, b in zip(gf, rf):
        acc._gen_sum, acc._gen_cov = acc._accumulate(a, acc._gen_sum, acc._gen_cov)
        acc._ref_sum, acc._ref_cov = acc._accumulate(b, acc._ref_sum, acc._ref_cov)
        acc._count += 1
    out["fvd_gen_feats"], out["fvd_ref_feats"] = torch.from_numpy(gf), torch.from_numpy(rf)
    out["fvd_result"] = acc.compute()
    acc.min_videos = 256
    out["fvd_result_warn"] = acc.compute()
    one = ref.OnlineFrechetAccumulator(device="cpu")
    out["fvd_result_too_few"] = one.compute()
    # matrix-rows form of _accumulate (the FID path) against the vector form
    s, c = ref.OnlineFrechetAccumulator._accumulate(gf, np.zeros(400), np.zeros((400, 400)))
    out["rows_sum"], out["rows_cov_trace"] = torch.from_numpy(s), float(np.trace(c))
    out["frechet_direct"] = ref._compute_frechet_distance(acc._gen_sum, acc._gen_cov, 6, acc._ref_sum, acc._ref_cov, 6)
    torch.save(out, OUT / "eval_metrics.pt")
    print({k: v for k, v in out.items() if not torch.is_tensor(v)})


if __name__ == "__main__":
    main()
