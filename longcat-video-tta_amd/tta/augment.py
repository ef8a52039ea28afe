"""Augmented variants of the TTA clip: what the inner loop draws one of per step (run_lora_tta.py:470-500).

Mirrors `delta_experiment/scripts/common.py:1161-1362` (`parse_speed_factors`, `build_augmented_pixel_variants`,
`build_augmented_latent_variants`) and the runner's use of them (`run_lora_tta.py:1100-1126`): the variants are made at the
PIXEL level — horizontal flip, fixed and random rotations, temporal speed changes — and every variant but the original is then
encoded by the VAE (here: the HIP encoder).  Pixel preparation is host-side tensor indexing and one bilinear warp;
it is not on the hot path and uses plain torch ops.

Rotation: the reference calls `torchvision.transforms.functional.affine(frame, angle, translate=[0, 0], scale, shear=[0, 0],
BILINEAR, fill=0)` per frame.  torchvision is not installed in this image, so `rotate_clip` restates its published tensor path
(inverse affine matrix about the image centre, pixel-centre sampling grid, `grid_sample(bilinear, zeros, align_corners=False)`)
on all frames at once.  **Parity unpinned for the rotation only**; the zoom factor, the angle draws (`torch.arange` options +
`torch.randint`), flip, speed indices and names are pinned by `tests/golden/augment.pt`, minted from the reference's functions.
"""
import math
from typing import Any, Dict, Iterable, List, Optional

import torch


def parse_speed_factors(raw: str) -> List[float]:
    """'0.5,2.0' -> [0.5, 2.0] (common.py:1164-1169)."""
    if not raw:
        return []
    return [float(p) for p in (q.strip() for q in raw.split(",")) if p]


def _rotation_scale(h: int, w: int, degrees: float) -> float:
    """Zoom that keeps a rotated h x w frame free of empty corners (common.py:1172-1179; fp32 sin / cos as there)."""
    rad = abs(degrees) * (3.141592653589793 / 180.0)
    cos_a = abs(torch.cos(torch.tensor(rad)).item())
    sin_a = abs(torch.sin(torch.tensor(rad)).item())
    return max((w * cos_a + h * sin_a) / w, (h * cos_a + w * sin_a) / h)


def rotate_clip(pixel_frames: torch.Tensor, degrees: float, zoom: bool = False) -> torch.Tensor:
    """[1, C, T, H, W] rotated about the frame centre by `degrees` (torchvision's `affine` convention), bilinear, empty
    area = 0, optionally zoomed to fill the canvas (common.py:1182-1216)."""
    _, C, T, H, W = pixel_frames.shape
    scale = _rotation_scale(H, W, degrees) if zoom else 1.0
    rot = math.radians(degrees)
    # inverse map (output pixel -> input pixel), centre at the origin, no shear / translation:
    #   x_in = ( cos x + sin y) / scale,   y_in = (-sin x + cos y) / scale        (x right, y down, pixel centres)
    m00, m01, m10, m11 = math.cos(rot) / scale, math.sin(rot) / scale, -math.sin(rot) / scale, math.cos(rot) / scale
    dev, dt = pixel_frames.device, torch.float32
    xs = torch.arange(W, device=dev, dtype=dt) - (W - 1) * 0.5
    ys = torch.arange(H, device=dev, dtype=dt) - (H - 1) * 0.5
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    x_in = m00 * xx + m01 * yy
    y_in = m10 * xx + m11 * yy
    grid = torch.stack([x_in / (0.5 * W), y_in / (0.5 * H)], dim=-1)          # normalised, align_corners=False
    frames = pixel_frames[0].permute(1, 0, 2, 3).to(dt)                      # [T, C, H, W]
    # torchvision samples a ones channel along with the image and blends towards `fill` by it (fill = 0: image x coverage)
    frames = torch.cat([frames, torch.ones_like(frames[:, :1])], dim=1)
    out = torch.nn.functional.grid_sample(frames, grid.unsqueeze(0).expand(T, -1, -1, -1), mode="bilinear",
                                          padding_mode="zeros", align_corners=False)
    out = out[:, :-1] * out[:, -1:]
    return out.to(pixel_frames.dtype).permute(1, 0, 2, 3).unsqueeze(0)


def _random_angles(lo, hi, count, step) -> List[float]:
    """`count` rotation angles: from the `step` grid over [lo, hi] by one `torch.randint` draw, or - step <= 0 - uniform.
    The draw consumes the global torch RNG exactly as common.py:1280-1290 does (bounds swapped if given in reverse)."""
    lo, hi = (0.0 if lo is None else lo), (0.0 if hi is None else hi)
    lo, hi = min(lo, hi), max(lo, hi)
    if step and step > 0:
        grid = torch.arange(lo, hi + 1e-6, step)
        grid = grid if len(grid) else torch.tensor([lo])
        return grid[torch.randint(0, len(grid), (count,))].tolist()
    return torch.empty(count).uniform_(lo, hi).tolist()


def _retimed(t_len: int, factor: float, device):
    """(name, frame indices) of a speed variant: factor > 1 keeps every round(factor)-th frame (a SHORTER clip), factor < 1
    repeats every frame round(1 / factor) times and keeps the first t_len (common.py:1293-1312); factor 1 is no variant."""
    if factor == 1.0:
        return None
    if factor > 1.0:
        n = max(2, int(round(factor)))
        return f"speed_{n}x", torch.arange(0, t_len, step=n, device=device)
    n = max(2, int(round(1.0 / factor)))
    return f"slow_{n}x", torch.arange(t_len, device=device).repeat_interleave(n)[:t_len]


def build_augmented_pixel_variants(pixel_frames: torch.Tensor, *, enable_flip: bool = False, rotate_deg: float = 0.0,
                                   rotate_random_min: float = 5.0, rotate_random_max: float = 15.0,
                                   rotate_random_count: int = 2, rotate_random_step: float = 1.0, rotate_zoom: bool = True,
                                   speed_factors: Optional[Iterable[float]] = None) -> List[Dict[str, Any]]:
    """[{pixel_frames [1, C, T', H, W], name}, ...]: the original first, then flip, the fixed rotation pair (-deg, +deg), the
    random rotations, the speed variants - the reference's order and names (common.py:1219-1314)."""
    made = [("orig", pixel_frames)]
    if enable_flip:
        made.append(("flip_h", pixel_frames.flip(dims=[4])))
    turns = []                                                          # (name, angle)
    if rotate_deg and rotate_deg > 0:
        turns += [(f"rotate_{a:+.1f}", a) for a in (-rotate_deg, rotate_deg)]
    if rotate_random_count and rotate_random_count > 0:
        drawn = _random_angles(rotate_random_min, rotate_random_max, rotate_random_count, rotate_random_step)
        turns += [(f"rotate_rand_{float(a):+.1f}", float(a)) for a in drawn if abs(a) >= 1e-6]
    made += [(name, rotate_clip(pixel_frames, angle, zoom=rotate_zoom)) for name, angle in turns]
    for factor in (speed_factors or ()):
        spec = _retimed(int(pixel_frames.shape[2]), factor, pixel_frames.device)
        if spec is not None:
            made.append((spec[0], pixel_frames[:, :, spec[1]]))
    return [{"pixel_frames": px, "name": name} for name, px in made]


def _encode(vae, pixel_frames: torch.Tensor) -> torch.Tensor:
    from .common import encode_video
    keep = 1 + 4 * ((pixel_frames.shape[2] - 1) // 4)          # the causal encoder takes 1 + 4k frames
    return encode_video(vae, pixel_frames[:, :, :keep].to(vae.dtype), normalize=True)


def build_augmented_latent_variants(pixel_frames: torch.Tensor, base_latents: torch.Tensor, vae, **kw) -> List[Dict[str, Any]]:
    """{latents, name} per pixel variant; 'orig' reuses `base_latents` (common.py:1317-1362)."""
    out = []
    for item in build_augmented_pixel_variants(pixel_frames, **kw):
        lat = base_latents if item["name"] == "orig" else _encode(vae, item["pixel_frames"])
        out.append({"latents": lat, "name": item["name"]})
    return out


def build_train_latents_variants(vae, pixel_frames: torch.Tensor, cond_latents: torch.Tensor, train_latents: torch.Tensor,
                                 args) -> List[Dict[str, Any]]:
    """The runner's block (run_lora_tta.py:1100-1126): every pixel variant encoded and cut to the training window
    [T_cond, T_cond + T_train) of the latent clip.  One deviation: a variant whose encoded clip does not cover that window (a
    `speed_Nx` clip has 1/N of the frames) is skipped with a note - the reference would hand the loop an empty or short slice."""
    pix = build_augmented_pixel_variants(
        pixel_frames, enable_flip=args.aug_flip, rotate_deg=args.aug_rotate_deg, rotate_random_min=args.aug_rotate_random_min,
        rotate_random_max=args.aug_rotate_random_max, rotate_random_count=args.aug_rotate_random_count,
        rotate_random_step=args.aug_rotate_random_step, rotate_zoom=args.aug_rotate_zoom,
        speed_factors=parse_speed_factors(args.aug_speed_factors))
    t0 = cond_latents.shape[2]
    t1 = t0 + train_latents.shape[2]
    out = []
    for pv in pix:
        if pv["name"] == "orig":
            out.append({"latents": train_latents, "name": "orig"})
            continue
        lat = _encode(vae, pv["pixel_frames"])
        if lat.shape[2] < t1:
            print(f"  augmentation variant {pv['name']}: {lat.shape[2]} latent frames do not cover the training window "
                  f"[{t0}, {t1}) - skipped")
            continue
        out.append({"latents": lat[:, :, t0:t1].to(train_latents.dtype), "name": pv["name"]})
    return out
