"""CPU: pin oracle/eval_oracle.py and the host mirror tta/eval_metrics.py (Frechet accumulator, aggregation) against
tests/golden/eval_metrics.pt — minted from the reference's own compute_psnr / _ssim_single / OnlineFrechetAccumulator by
tests/golden/make_eval_golden.py — and cross-check the two third-party SSIM restatements against independent float64
evaluations (torchmetrics / skimage are absent offline: those two stay "parity unpinned", see the oracle header)."""
import math
from pathlib import Path

import numpy as np
import pytest
import torch

FX = torch.load(Path(__file__).resolve().parent / "golden" / "eval_metrics.pt")
GEN = FX["gen"].numpy()
GT = (FX["gt_u8"].numpy() / 255.0).astype(np.float32)


def test_psnr_loop_matches_reference_compute_psnr():
    from oracle import eval_oracle as E
    got = E.frame_psnr(GEN, GT)
    for g, ref in zip(got, FX["psnr_compute_psnr"]):
        if math.isinf(ref):
            assert g == 50.0          # identical frames: evaluate_generation_metrics caps at 50 dB where compute_psnr says inf
        else:
            assert abs(g - ref) < 1e-4   # numpy float32 pairwise mean vs torch mse_loss: a few fp32 ulps of the mse
    base = E.baseline_psnr(GEN, GT)
    assert base[2] == 60.0 and abs(base[0] - FX["psnr_compute_psnr"][0]) < 1e-4


def test_ssim_global_statistics_branch_matches_reference():
    from oracle import eval_oracle as E
    assert FX["ssim_impl"] == "fallback_global_statistics"
    for i, ref in enumerate(FX["ssim_single"]):
        p = torch.from_numpy(GEN[i]).permute(2, 0, 1).unsqueeze(0)
        g = torch.from_numpy(GT[i]).permute(2, 0, 1).unsqueeze(0)
        assert abs(E.ssim_global_statistics(p, g) - ref) < 1e-6


def _gaussian_ssim_float64(x, y):
    """The textbook definition evaluated independently (scipy separable correlation, float64)."""
    from scipy.ndimage import correlate1d
    d = np.arange(-5, 6, dtype=np.float64)
    t = np.exp(-(d / 1.5) ** 2 / 2); t /= t.sum()
    f = lambda a: correlate1d(correlate1d(a, t, axis=0, mode="reflect"), t, axis=1, mode="reflect")
    vals = []
    for c in range(x.shape[2]):
        X, Y = x[..., c].astype(np.float64), y[..., c].astype(np.float64)
        mx, my = f(X), f(Y)
        sx, sy, sxy = np.maximum(f(X * X) - mx * mx, 0), np.maximum(f(Y * Y) - my * my, 0), f(X * Y) - mx * my
        S = ((2 * mx * my + 1e-4) * (2 * sxy + 9e-4)) / ((mx * mx + my * my + 1e-4) * (sx + sy + 9e-4))
        vals.append(S[5:-5, 5:-5])
    return float(np.mean(vals))


def test_gaussian_ssim_restatement_against_independent_float64():
    from oracle import eval_oracle as E
    got = E.frame_ssim(GEN, GT)
    for i in range(GEN.shape[0]):
        assert abs(got[i] - _gaussian_ssim_float64(GEN[i], GT[i])) < 2e-6
    assert got[2] == pytest.approx(1.0, abs=1e-6)
    # symmetric in its arguments, and a constant offset lowers it
    assert E.frame_ssim(GT, GEN)[0] == pytest.approx(got[0], abs=1e-6)
    assert E.frame_ssim(np.clip(GEN + 0.2, 0, 1).astype(np.float32), GT)[0] < got[0]


def test_uniform7_ssim_properties():
    from oracle import eval_oracle as E
    a = E.ssim_uniform7(GEN[0], GT[0])
    assert 0.0 < a < 1.0 and E.ssim_uniform7(GT[0], GT[0]) == pytest.approx(1.0, abs=1e-12)
    assert E.ssim_uniform7(GT[0], GEN[0]) == pytest.approx(a, abs=1e-12)
    # direct evaluation of one interior window (row 9, col 12, channel 1) from the definition
    X, Y = GEN[0][6:13, 9:16, 1].astype(np.float64), GT[0][6:13, 9:16, 1].astype(np.float64)
    ux, uy = X.mean(), Y.mean()
    vx, vy, vxy = X.var(ddof=1), Y.var(ddof=1), ((X - ux) * (Y - uy)).sum() / 48
    s = ((2 * ux * uy + 1e-4) * (2 * vxy + 9e-4)) / ((ux * ux + uy * uy + 1e-4) * (vx + vy + 9e-4))
    from scipy.ndimage import uniform_filter
    f = lambda a_: uniform_filter(a_, size=7)
    Xc, Yc = GEN[0][..., 1].astype(np.float64), GT[0][..., 1].astype(np.float64)
    mx, my = f(Xc), f(Yc)
    S = ((2 * mx * my + 1e-4) * (2 * 49 / 48 * (f(Xc * Yc) - mx * my) + 9e-4)) / \
        ((mx ** 2 + my ** 2 + 1e-4) * (49 / 48 * (f(Xc * Xc) - mx * mx) + 49 / 48 * (f(Yc * Yc) - my * my) + 9e-4))
    assert S[9, 12] == pytest.approx(s, rel=1e-9)


def test_frechet_accumulator_matches_reference():
    from oracle import eval_oracle as E
    from tta.eval_metrics import OnlineFrechetAccumulator, frechet_distance
    gf, rf = FX["fvd_gen_feats"].numpy(), FX["fvd_ref_feats"].numpy()
    o = E.FrechetAccumulator()
    acc = OnlineFrechetAccumulator(min_videos=4)
    assert acc.compute() == FX["fvd_result_too_few"]
    for a, b in zip(gf, rf):
        o.update(a, b)
        acc.update_features(a, b)
    assert acc.compute() == FX["fvd_result"]                      # same keys, same rounded value
    assert o.compute()["fvd"] == FX["fvd_result"]["fvd"]
    acc.min_videos = 256
    assert acc.compute() == FX["fvd_result_warn"]
    assert frechet_distance(acc._gen_sum, acc._gen_cov, 6, acc._ref_sum, acc._ref_cov, 6) == pytest.approx(FX["frechet_direct"], rel=1e-12)
    s, c = OnlineFrechetAccumulator._accumulate(gf, np.zeros(400), np.zeros((400, 400)))   # rows form (the FID path)
    assert np.allclose(s, FX["rows_sum"].numpy()) and np.trace(c) == pytest.approx(FX["rows_cov_trace"], rel=1e-12)
    with pytest.raises(RuntimeError, match="I3D"):
        acc.update(torch.zeros(4, 8, 8, 3), torch.zeros(4, 8, 8, 3), 1, 2)


def test_aggregate_quality_metrics():
    from tta.eval_metrics import aggregate_quality_metrics
    s = {"results": [{"success": True, "psnr": 20.0, "ssim": 0.5, "lpips": None},
                     {"success": True, "psnr": 22.0, "ssim": 0.7, "lpips": None},
                     {"success": False, "psnr": 99.0}]}
    aggregate_quality_metrics(s)
    assert s["psnr"] == 21.0 and s["ssim"] == 0.6 and s["lpips"] is None
