"""Raw video in, the runners' per-video blob out: the host-side step in front of the hot path (SURVEY §8(f) rank 1 "per-video
pre-encode") for datasets that were not pre-encoded to `<data-dir>/latents/*.pt`.

Frame windows and pixel conventions are the reference's:
  * `load_video_frames` — delta_experiment/scripts/common.py:103-155: PyAV decode, `start_frame` decoded frames skipped, the last
    frame repeated when the clip is short, uint8 / 255, trilinear `F.interpolate` to (T, height, width) without corner alignment,
    then [-1, 1]; [1, 3, T, H, W] fp32.  The runners call it at 480 x 832 whatever `--resolution` says (Appendix B).
  * TTA window = `tta_total_frames` frames ending at `gen_start_frame`; conditioning window = `num_cond_frames` frames ending
    there (run_lora_tta.py:1084-1096, 1196-1207: both loaded at 480 x 832, the conditioning frames quantised to uint8 images);
  * ground truth = the frames from `gen_start_frame` on, `frame.to_image()` resized with PIL LANCZOS to the OUTPUT size
    (common.py:698-715).
PyAV is needed for any of it and is imported where it is used: an image without it (this one) raises a plain ImportError that
names the alternative (pre-encoded clips).  Nothing here touches the GPU except through the VAE / text encoder handed in."""
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


def _av():
    try:
        import av
    except ImportError as e:
        raise ImportError("raw-video datasets need PyAV (`import av`) for decoding; without it pre-encode the clips to "
                          "<data-dir>/latents/<stem>.pt (tta/runner_common.py::load_entry describes the blob)") from e
    return av


def load_video_frames(video_path: str, num_frames: int, height: int = 480, width: int = 832, start_frame: int = 0) -> torch.Tensor:
    av = _av()
    container = av.open(video_path)
    frames: List[np.ndarray] = []
    seen = 0
    try:
        for frame in container.decode(video=0):
            if seen < start_frame:
                seen += 1
                continue
            if len(frames) >= num_frames:
                break
            frames.append(frame.to_ndarray(format="rgb24"))
            seen += 1
    finally:
        container.close()
    if not frames:
        raise ValueError(f"No frames decoded from {video_path}")
    frames += [frames[-1]] * (num_frames - len(frames))
    x = torch.from_numpy(np.stack(frames[:num_frames], axis=0)).permute(3, 0, 1, 2).float() / 255.0      # [3, T, H0, W0]
    x = F.interpolate(x.unsqueeze(0), size=(x.shape[1], height, width), mode="trilinear", align_corners=False)
    return x * 2.0 - 1.0


def load_ground_truth_frames(video_path: str, gen_start_frame: int, num_frames: int, out_h: int, out_w: int) -> Optional[np.ndarray]:
    """uint8 [n, out_h, out_w, 3], n <= num_frames (fewer when the clip ends); None when there is none."""
    from PIL import Image
    av = _av()
    container = av.open(video_path)
    images, seen = [], 0
    try:
        for frame in container.decode(video=0):
            if seen < gen_start_frame:
                seen += 1
                continue
            if len(images) >= num_frames:
                break
            images.append(frame.to_image())
            seen += 1
    finally:
        container.close()
    if not images:
        return None
    return np.stack([np.array(im.resize((out_w, out_h), Image.LANCZOS)) for im in images], axis=0)


def frames_to_uint8(pixel_frames: torch.Tensor) -> torch.Tensor:
    """[1, 3, T, H, W] in [-1, 1] -> uint8 [T, H, W, 3], the quantisation of run_lora_tta.py:1202-1207 (truncating cast)."""
    pf = ((pixel_frames.squeeze(0).float() + 1.0) / 2.0).clamp(0, 1)
    return (pf.permute(1, 2, 3, 0) * 255).to(torch.uint8)


def prepare_video_entry(entry: Dict, args, pipe, device, total_frames: Optional[int] = None) -> Dict:
    """The blob `load_entry` returns for a raw-video entry: TTA-window pixels and their normalised latents, the caption's (and the
    empty negative prompt's) UMT5 embeddings, the conditioning frames of the continuation and the ground truth of its generated
    frames.  Needs `pipe.vae`, `pipe.tokenizer` and `pipe.text_encoder` (a real checkpoint directory provides all three)."""
    from tta.common import encode_prompt, encode_video
    for need in ("vae", "tokenizer", "text_encoder"):
        if getattr(pipe, need, None) is None:
            raise RuntimeError(f"raw-video input needs pipe.{need} (load a checkpoint directory that has it), or pre-encoded clips")
    path, caption = entry["path"], entry.get("caption", "")
    total = total_frames if total_frames is not None else args.tta_total_frames
    gen_start = int(getattr(args, "gen_start_frame", total))
    pix = load_video_frames(path, total, height=480, width=832, start_frame=max(0, gen_start - total)).to(device, torch.bfloat16)
    latents = encode_video(pipe.vae, pix, normalize=True).to(torch.bfloat16)
    pe, pm = encode_prompt(pipe.tokenizer, pipe.text_encoder, caption, device=device, dtype=torch.bfloat16)
    ne, nm = encode_prompt(pipe.tokenizer, pipe.text_encoder, "", device=device, dtype=torch.bfloat16)
    blob = dict(latents=latents, pixel_frames=pix, prompt_embeds=pe, prompt_mask=pm, negative_embeds=ne, negative_mask=nm, caption=caption)
    if not getattr(args, "skip_generation", False):
        H, W = {"480p": (480, 832), "720p": (720, 1280)}[args.resolution]
        cond = load_video_frames(path, args.num_cond_frames, height=480, width=832, start_frame=max(0, gen_start - args.num_cond_frames))
        cond_u8 = frames_to_uint8(cond)
        if (H, W) != (480, 832):
            # the reference hands 480 x 832 images to pipe.generate_vc, whose preprocessing resizes them to the output bucket
            # [assumed-from-upstream: diffusers' VideoProcessor default, PIL LANCZOS; spec/dit.md lists it with the other assumptions]
            from PIL import Image
            cond_u8 = torch.from_numpy(np.stack([np.array(Image.fromarray(f.numpy()).resize((W, H), Image.LANCZOS)) for f in cond_u8.cpu()], 0))
        blob["cond_frames"] = cond_u8
        n_gen = int(args.num_frames) - int(args.num_cond_frames)
        gt = load_ground_truth_frames(path, gen_start, n_gen, H, W)
        if gt is not None:
            blob["gt_frames"] = torch.from_numpy(gt)
    return blob
