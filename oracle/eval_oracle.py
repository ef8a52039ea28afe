"""CPU oracle for the evaluation row (TEST INFRASTRUCTURE — never imported by the product).

Restates, in numpy / torch-CPU, what the reference computes after every generation
(delta_experiment/scripts/common.py, paths relative to /root/reference):

* `frame_psnr`            — the per-frame PSNR loop of evaluate_generation_metrics (:719-729): float32 frames,
                            mse < 1e-10 -> 50.0 dB, mean over frames.  Pinned against the reference's `compute_psnr`
                            (:617-622) through tests/golden/eval_metrics.pt (same formula without the 50 dB clamp).
* `ssim_global_statistics`— the branch of `_ssim_single` (:765-776) that runs when torchmetrics is absent.  Pinned by
                            the same fixture (minted here, where torchmetrics IS absent).
* `ssim_gaussian`         — the branch that runs in the reference's real environment (torchmetrics is installed by
                            env_setup/01_setup_longcat_env.sbatch:226): `StructuralSimilarityIndexMeasure(data_range=1.0)`.
                            torchmetrics is a third-party dependency that is absent offline and not pinned by the
                            reference (no version in the sbatch), so this function restates the published algorithm of
                            its `_ssim_update` — 11x11 Gaussian (sigma 1.5) depthwise filter over reflect-padded inputs,
                            variances clamped at 0, c1=(0.01 R)^2, c2=(0.03 R)^2, the padded border cropped away, mean
                            over the rest.  **Parity unpinned** for this one function; tests cross-check it against an
                            independent scipy float64 evaluation of the textbook definition.
* `baseline_psnr`, `ssim_uniform7` — the baseline runner's own metrics (baseline_experiment/scripts/run_baseline.py:124-136,
                            436-441): float64, 60 dB cap, and skimage's `structural_similarity(channel_axis=2,
                            data_range=1.0)` defaults.  skimage is absent offline too; its published algorithm is restated
                            with the same scipy.ndimage.uniform_filter calls it makes (7x7, sample covariance 49/48,
                            border of 3 cropped, mean over channels).  **Parity unpinned** likewise.
* `FrechetAccumulator`    — `OnlineFrechetAccumulator._accumulate / compute` + `_compute_frechet_distance`
                            (:2210-2231, :2316-2326, :2386-2428) in float64.  Pinned by the fixture.
"""
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F


def frame_psnr(gen: np.ndarray, gt: np.ndarray) -> List[float]:
    """gen, gt float32 [N,H,W,3] in [0,1] -> per-frame PSNR list (common.py:719-728)."""
    out = []
    for i in range(gen.shape[0]):
        mse = np.mean((gen[i] - gt[i]) ** 2)
        out.append(50.0 if mse < 1e-10 else float(10.0 * np.log10(1.0 / mse)))
    return out


def ssim_global_statistics(p: torch.Tensor, g: torch.Tensor) -> float:
    """[1,C,H,W] in [0,1]; per-channel global means / variances (common.py:765-776)."""
    mu_p, mu_g = p.mean(dim=[2, 3], keepdim=True), g.mean(dim=[2, 3], keepdim=True)
    sp = ((p - mu_p) ** 2).mean(dim=[2, 3], keepdim=True)
    sg = ((g - mu_g) ** 2).mean(dim=[2, 3], keepdim=True)
    spg = ((p - mu_p) * (g - mu_g)).mean(dim=[2, 3], keepdim=True)
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu_p * mu_g + c1) * (2 * spg + c2)) / ((mu_p ** 2 + mu_g ** 2 + c1) * (sp + sg + c2))
    return m.mean().item()


def gaussian_taps(kernel_size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    dist = torch.arange((1 - kernel_size) / 2, (1 + kernel_size) / 2, 1, dtype=torch.float32)
    gauss = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    return (gauss / gauss.sum()).unsqueeze(0)                       # [1, k]


def ssim_gaussian(p: torch.Tensor, g: torch.Tensor, data_range: float = 1.0, dtype=torch.float32) -> torch.Tensor:
    """[B,C,H,W] -> per-image SSIM [B] (torchmetrics' gaussian-kernel path, see the module header).  torchmetrics works
    in the input dtype (fp32 in the reference); `dtype=torch.float64` evaluates the same map without the fp32 cancellation
    in E[x^2] - mu^2, which is what two fp32 implementations with different summation orders are both compared to."""
    k, sigma = 11, 1.5
    C = p.shape[1]
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    t = gaussian_taps(k, sigma).to(dtype)
    p, g = p.to(dtype), g.to(dtype)
    kernel = torch.matmul(t.t(), t).expand(C, 1, k, k)               # outer product, one copy per channel
    pad = (k - 1) // 2
    pp = F.pad(p, (pad, pad, pad, pad), mode="reflect")
    gp = F.pad(g, (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat((pp, gp, pp * pp, gp * gp, pp * gp))           # [5B, C, H+10, W+10]
    o = F.conv2d(stack, kernel, groups=C)
    B = p.shape[0]
    mu_p, mu_g, e_pp, e_gg, e_pg = (o[i * B:(i + 1) * B] for i in range(5))
    s_p = torch.clamp(e_pp - mu_p ** 2, min=0.0)
    s_g = torch.clamp(e_gg - mu_g ** 2, min=0.0)
    s_pg = e_pg - mu_p * mu_g
    full = ((2 * mu_p * mu_g + c1) * (2 * s_pg + c2)) / ((mu_p ** 2 + mu_g ** 2 + c1) * (s_p + s_g + c2))
    return full[..., pad:-pad, pad:-pad].reshape(B, -1).mean(-1)


def frame_ssim(gen: np.ndarray, gt: np.ndarray) -> List[float]:
    """Per-frame loop of evaluate_generation_metrics (:731-737) with the torchmetrics branch of _ssim_single."""
    out = []
    for i in range(gen.shape[0]):
        p = torch.from_numpy(gen[i]).permute(2, 0, 1).unsqueeze(0).float()
        g = torch.from_numpy(gt[i]).permute(2, 0, 1).unsqueeze(0).float()
        out.append(ssim_gaussian(p, g).mean().item())
    return out


def baseline_psnr(gen: np.ndarray, gt: np.ndarray) -> List[float]:
    """run_baseline.py:124-129 per frame (float64, 60 dB cap)."""
    out = []
    for i in range(gen.shape[0]):
        mse = np.mean((gen[i].astype(np.float64) - gt[i].astype(np.float64)) ** 2)
        out.append(60.0 if mse < 1e-10 else float(10.0 * np.log10(1.0 / mse)))
    return out


def ssim_uniform7(p: np.ndarray, g: np.ndarray, data_range: float = 1.0) -> float:
    """One [H,W,C] frame pair, float64: skimage.metrics.structural_similarity with its defaults (win 7, uniform filter,
    use_sample_covariance=True, K1 .01, K2 .03), channel_axis=2 -> mean of the per-channel means."""
    from scipy.ndimage import uniform_filter
    win, NP = 7, 49
    cov_norm = NP / (NP - 1)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    pad = (win - 1) // 2
    vals = []
    for c in range(p.shape[2]):
        X, Y = p[..., c].astype(np.float64), g[..., c].astype(np.float64)
        ux, uy = uniform_filter(X, size=win), uniform_filter(Y, size=win)
        uxx, uyy, uxy = uniform_filter(X * X, size=win), uniform_filter(Y * Y, size=win), uniform_filter(X * Y, size=win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        vals.append(S[pad:-pad, pad:-pad].mean(dtype=np.float64))
    return float(np.mean(vals))


def frechet_distance(sum_a, cov_sum_a, n_a, sum_b, cov_sum_b, n_b, eps: float = 1e-6) -> float:
    """common.py:2210-2231 (float64, scipy sqrtm)."""
    from scipy.linalg import sqrtm
    mu_a, mu_b = sum_a / n_a, sum_b / n_b
    sa = cov_sum_a / n_a - np.outer(mu_a, mu_a) + eps * np.eye(len(mu_a))
    sb = cov_sum_b / n_b - np.outer(mu_b, mu_b) + eps * np.eye(len(mu_b))
    d = mu_a - mu_b
    covmean = sqrtm(sa @ sb)
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(d @ d + np.trace(sa + sb - 2 * covmean))


class FrechetAccumulator:
    """Running sums of 400-d clip features for generated / reference videos (common.py:2234-2262, 2316-2326)."""

    def __init__(self, dim: int = 400):
        self.gs, self.gc = np.zeros(dim), np.zeros((dim, dim))
        self.rs, self.rc = np.zeros(dim), np.zeros((dim, dim))
        self.n = 0

    def update(self, gen_feat: np.ndarray, ref_feat: np.ndarray):
        self.gs += gen_feat; self.gc += np.outer(gen_feat, gen_feat)
        self.rs += ref_feat; self.rc += np.outer(ref_feat, ref_feat)
        self.n += 1

    def compute(self) -> Dict:
        if self.n < 2:
            return {"fvd": None, "fvd_num_videos": self.n, "fvd_error": "Need at least 2 videos for FVD"}
        return {"fvd": round(frechet_distance(self.gs, self.gc, self.n, self.rs, self.rc, self.n), 6),
                "fvd_num_videos": self.n}
