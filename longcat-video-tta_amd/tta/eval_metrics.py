"""Evaluation after generation, on the device.

Host-side mirror of the reference's `evaluate_generation_metrics`, `aggregate_quality_metrics`,
`OnlineFrechetAccumulator` and `finalize_online_eval` (delta_experiment/scripts/common.py:663-757, 2234-2431, 2453-2530).
The reference copies every generated clip to the host and loops over frames in numpy / torch-CPU; here the frames stay
where the VAE decoder wrote them and PSNR / SSIM are two kernels of liblcv_hip.so (`lcv_frame_sqerr`, `lcv_frame_ssim`).

Kept from the reference: frame slicing `[num_cond : num_cond + num_gen]`, `n_compare = min(gen, gt)`, the `mse < 1e-10 ->
50 dB` rule, per-frame-then-mean averaging, NaN for a metric whose model is unavailable (LPIPS needs AlexNet weights and
FVD the I3D TorchScript file: neither exists offline), the result keys and the Frechet accumulator's float64 sums.
Not mirrored: decoding the source video with PyAV and the LANCZOS resize of the ground truth (host image IO, outside the
path): the caller hands ground-truth frames already at the output resolution.
"""
import math
from typing import Any, Callable, Dict, Optional

import numpy as np
import torch

from lcv_hip import ops

I3D_FEATURE_DIM = 400            # common.py:2143
DEFAULT_MIN_FVD_VIDEOS = 256     # common.py:2146
COV_EPS = 1e-6                   # common.py:2147


def evaluate_generation_metrics(gen_output: torch.Tensor, gt_frames: torch.Tensor, num_cond_frames: int,
                                num_gen_frames: int, flavour: str = "tta") -> Dict[str, float]:
    """gen_output: fp32 GPU [N,H,W,3] in [0,1], the full pipeline output (conditioning frames first);
    gt_frames: GPU uint8 or fp32 [>=n,H,W,3], the ground truth of the GENERATED frames at the output resolution.
    Returns {"psnr", "ssim", "lpips"}.  flavour "tta": common.py:663-757 (50 dB cap, torchmetrics' Gaussian SSIM);
    "baseline": run_baseline.py:124-136, 436-441 (60 dB cap, skimage's 7x7 uniform SSIM)."""
    if flavour not in ("tta", "baseline"):
        raise ValueError(f"unknown metric flavour {flavour!r}")
    gen = gen_output[num_cond_frames:num_cond_frames + num_gen_frames]
    n = min(int(gen.shape[0]), int(gt_frames.shape[0]))
    if n == 0:
        return {"psnr": float("nan"), "ssim": float("nan"), "lpips": float("nan")}
    mse, ssim = ops.frame_metrics(gen[:n].float(), gt_frames[:n], ssim="gaussian11" if flavour == "tta" else "uniform7")
    cap = 50.0 if flavour == "tta" else 60.0
    psnr = [cap if m < 1e-10 else float(10.0 * math.log10(1.0 / m)) for m in mse.tolist()]
    return {"psnr": float(np.mean(psnr)), "ssim": float(np.mean(ssim.tolist())), "lpips": float("nan")}


def aggregate_quality_metrics(summary: dict) -> None:
    """Average per-video PSNR / SSIM / LPIPS into the summary (common.py:2453-2458)."""
    ok = [r for r in summary.get("results", []) if r.get("success")]
    for key in ("psnr", "ssim", "lpips"):
        vals = [r[key] for r in ok if r.get(key) is not None]
        summary[key] = round(float(np.mean(vals)), 6) if vals else None


def frechet_distance(sum_a, cov_sum_a, n_a, sum_b, cov_sum_b, n_b, eps: float = COV_EPS) -> float:
    """Frechet distance from running sums, float64 (common.py:2210-2231)."""
    from scipy.linalg import sqrtm
    mu_a, mu_b = sum_a / n_a, sum_b / n_b
    sigma_a = cov_sum_a / n_a - np.outer(mu_a, mu_a)
    sigma_b = cov_sum_b / n_b - np.outer(mu_b, mu_b)
    sigma_a += eps * np.eye(sigma_a.shape[0])
    sigma_b += eps * np.eye(sigma_b.shape[0])
    diff = mu_a - mu_b
    covmean = sqrtm(sigma_a @ sigma_b)
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(diff @ diff + np.trace(sigma_a + sigma_b - 2 * covmean))


class OnlineFrechetAccumulator:
    """Running first / second moments of clip features for online FVD (common.py:2234-2431).  The feature extractor
    (I3D Kinetics-400 TorchScript in the reference) is a constructor argument: `feature_fn(clip [T,H,W,3] fp32 GPU
    tensor in [0,1]) -> 400 floats`.  Without one, `update` raises — there is no stand-in network."""

    def __init__(self, feature_fn: Optional[Callable] = None, min_videos: int = DEFAULT_MIN_FVD_VIDEOS,
                 dim: int = I3D_FEATURE_DIM):
        self.feature_fn, self.min_videos, self.dim = feature_fn, min_videos, dim
        self._gen_sum, self._gen_cov = np.zeros(dim), np.zeros((dim, dim))
        self._ref_sum, self._ref_cov = np.zeros(dim), np.zeros((dim, dim))
        self._count = 0

    @staticmethod
    def _accumulate(feats: np.ndarray, feat_sum: np.ndarray, cov_sum: np.ndarray):
        feats = np.asarray(feats, dtype=np.float64)
        if feats.ndim == 1:
            feat_sum += feats
            cov_sum += np.outer(feats, feats)
        else:
            feat_sum += feats.sum(axis=0)
            cov_sum += feats.T @ feats
        return feat_sum, cov_sum

    def update_features(self, gen_feat: np.ndarray, ref_feat: np.ndarray) -> None:
        self._gen_sum, self._gen_cov = self._accumulate(gen_feat, self._gen_sum, self._gen_cov)
        self._ref_sum, self._ref_cov = self._accumulate(ref_feat, self._ref_sum, self._ref_cov)
        self._count += 1

    def update(self, gen_output: torch.Tensor, gt_frames: torch.Tensor, num_cond_frames: int, num_gen_frames: int) -> None:
        if self.feature_fn is None:
            raise RuntimeError("online FVD needs the I3D Kinetics-400 TorchScript feature extractor (common.py:2271-2279); "
                               "it is a network download and does not exist offline — pass feature_fn=")
        gen = gen_output[num_cond_frames:num_cond_frames + num_gen_frames]
        n = min(int(gen.shape[0]), int(gt_frames.shape[0]))
        if n == 0:
            return
        gt = gt_frames[:n].float() / 255.0 if gt_frames.dtype == torch.uint8 else gt_frames[:n].float()
        self.update_features(np.asarray(self.feature_fn(gen[:n].float()), dtype=np.float64),
                             np.asarray(self.feature_fn(gt), dtype=np.float64))

    def compute(self) -> Dict[str, Any]:
        result: Dict[str, Any] = {}
        if self._count < 2:
            result["fvd"] = None
            result["fvd_num_videos"] = self._count
            result["fvd_error"] = "Need at least 2 videos for FVD"
            return result
        fvd = frechet_distance(self._gen_sum, self._gen_cov, self._count, self._ref_sum, self._ref_cov, self._count)
        result["fvd"] = round(fvd, 6)
        result["fvd_num_videos"] = self._count
        result["fvd_feature_extractor"] = "i3d_kinetics400_torchscript"
        result["fvd_feature_dim"] = self.dim
        if self._count < self.min_videos:
            result["fvd_sample_size_warning"] = (f"FVD computed with {self._count} videos (recommended >= {self.min_videos}). "
                                                 "Covariance estimate may be unreliable.")
        return result


def finalize_online_eval(accumulator: Optional[OnlineFrechetAccumulator], summary: dict) -> None:
    """Merge FVD into the summary next to PSNR / SSIM / LPIPS (common.py:2461-2478; VBench++ is not part of this build)."""
    if accumulator is not None:
        summary.update(accumulator.compute())
