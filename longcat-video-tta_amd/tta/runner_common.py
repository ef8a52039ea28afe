"""What the per-method runner scripts share: the common CLI block, model / data loading (real
checkpoints or `synthetic[:depth[:hidden[:caption]]]`; `latents/*.pt` or `synthetic:N` data), one-process-per-GPU data
parallelism over videos, the KV-cached CFG continuation, and the reference's `checkpoint.json` / `summary.json` schemas
(delta_experiment/scripts/run_delta_a.py:763-935, run_delta_b.py:915-960, run_delta_c.py:640-703;
`save_checkpoint` common.py:2055-2059).  The LoRA runner has its own main (it also writes `config.json`)."""
import json
import os
import sys
import time
from pathlib import Path
from typing import Callable, Dict

import numpy as np
import torch

from longcat_video.parallel import data_parallel as dp
from tta import cli_args as C
from tta.early_stopping import add_early_stopping_args, build_early_stopper_from_args
from tta.latent_split import _estimate_latent_len, num_frames_valid, split_tta_latents


def add_common_args(p):
    p.add_argument("--checkpoint-dir", type=str, required=True, help="checkpoint dir, or synthetic[:depth[:hidden[:caption]]]")
    p.add_argument("--data-dir", type=str, required=True)
    p.add_argument("--output-dir", type=str, required=True)
    p.add_argument("--max-videos", type=int, default=100)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--num-cond-frames", type=int, default=2)
    p.add_argument("--num-frames", type=int, default=16)
    p.add_argument("--gen-start-frame", type=int, default=32)
    p.add_argument("--num-inference-steps", type=int, default=50)
    p.add_argument("--guidance-scale", type=float, default=4.0)
    p.add_argument("--resolution", type=str, default="480p")
    p.add_argument("--skip-generation", action="store_true")
    p.add_argument("--no-save-videos", action="store_true")


def add_shared_groups(p, clip_gate: bool = True):
    add_early_stopping_args(p)
    C.add_augmentation_args(p)
    C.add_tta_frame_args(p)
    C.add_caption_guard_args(p)
    C.add_caption_override_args(p)
    C.add_feature_frame_guard_args(p)
    C.add_online_eval_args(p)
    if clip_gate:
        C.add_clip_gate_args(p)


def setup_distributed(args):
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")
        dp.host_group()          # the gloo group of the end-of-job merge is created while every rank is still here
    device = f"cuda:{local_rank}" if args.device.startswith("cuda") else args.device
    return rank, world, device


def load_components(args, device):
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
    from longcat_video.pipeline_longcat_video import LongCatVideoPipeline
    ck = args.checkpoint_dir
    if ck.startswith("synthetic"):
        kw = {}
        if ":" in ck:  # synthetic:depth[:hidden[:caption]] — reduced sizes are for plumbing tests only
            parts = ck.split(":")[1:]
            kw["depth"] = int(parts[0])
            if len(parts) > 1:
                kw.update(hidden_size=int(parts[1]), num_heads=int(parts[1]) // 128)
            if len(parts) > 2:
                kw.update(caption_channels=int(parts[2]))
        dit = LongCatVideoTransformer3DModel(device=device, dtype=torch.bfloat16, **kw).init_synthetic_(1234)
        from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
        sched, vae = FlowMatchEulerDiscreteScheduler(), AutoencoderKLWan(device=device).init_synthetic_()
    else:
        dit = LongCatVideoTransformer3DModel.from_pretrained(ck, subfolder="dit", cp_split_hw=[1, 1],
                                                             enable_flashattn2=True, torch_dtype=torch.bfloat16).to(device)
        sched = FlowMatchEulerDiscreteScheduler.from_pretrained(ck, subfolder="scheduler")
        from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
        vae = AutoencoderKLWan.from_pretrained(ck, subfolder="vae", torch_dtype=torch.bfloat16).to(device)
    tokenizer = text_encoder = None
    if not ck.startswith("synthetic") and os.path.isdir(os.path.join(ck, "text_encoder")):
        # common.py:59-64: AutoTokenizer (host-side SentencePiece, from transformers) + the UMT5 encoder, here on the HIP kernels
        from longcat_video.modules.umt5_encoder import UMT5EncoderModel
        text_encoder = UMT5EncoderModel.from_pretrained(ck, subfolder="text_encoder", torch_dtype=torch.bfloat16).to(device)
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(ck, subfolder="tokenizer")
    pipe = LongCatVideoPipeline(tokenizer=tokenizer, text_encoder=text_encoder, vae=vae, scheduler=sched, dit=dit)
    pipe.device = torch.device(device)
    return dit, pipe


def list_eval_entries(args, dit):
    """The clips of this run, in order - then, as every reference runner does right after its listing (run_lora_tta.py:911-925):
    `--fixed-caption` applied and the caption guard run over them (mode `fail` raises on a listing whose captions are mostly empty,
    identical or a placeholder; listings under 20 clips are never judged)."""
    from .datasets import apply_fixed_caption, validate_caption_quality
    entries = _list_entries(args, dit)
    if entries and entries[0]["kind"] != "synthetic" and any("caption" in e for e in entries):
        entries = apply_fixed_caption(entries, getattr(args, "fixed_caption", None), context="eval")
        validate_caption_quality(entries, mode=getattr(args, "caption_guard_mode", "fail"),
                                 min_nonempty_ratio=getattr(args, "caption_guard_min_nonempty_ratio", 0.95),
                                 min_unique_ratio=getattr(args, "caption_guard_min_unique_ratio", 0.10),
                                 max_top1_ratio=getattr(args, "caption_guard_max_top1_ratio", 0.50),
                                 max_generic_top1_ratio=getattr(args, "caption_guard_max_generic_top1_ratio", 0.20),
                                 top_k=getattr(args, "caption_guard_topk", 5), context="eval")
    return entries


def _list_entries(args, dit):
    d = args.data_dir
    if d.startswith("synthetic"):
        n = int(d.split(":")[1]) if ":" in d else 4
        return [{"kind": "synthetic", "name": f"synthetic_{i:04d}", "path": f"synthetic://{i}", "seed": 1000 + i}
                for i in range(min(n, args.max_videos))]
    # WHICH clips and in WHICH order: the reference's own sampler (tta/datasets.py mirrors load_ucf101_video_list, the one every
    # runner calls: run_lora_tta.py:911-913 - RandomState(seed), stratified by class) over the video files / metadata.csv of the
    # data directory, so that a run here and a run of the reference over the same --data-dir --max-videos --seed adapt the same clips
    from .datasets import load_ucf101_video_list
    lat = Path(d) / "latents"
    try:
        picked = load_ucf101_video_list(d, max_videos=args.max_videos, seed=getattr(args, "seed", 42), validate_decodable=not lat.is_dir())
    except FileNotFoundError:
        picked = None
    if lat.is_dir():
        if picked is None:      # pre-encoded clips only, no video files beside them: nothing for the sampler to order
            files = sorted(lat.glob("*.pt"))[: args.max_videos]
            return [{"kind": "latents", "name": f.stem, "path": str(f)} for f in files]
        out, missing = [], []
        for e in picked:
            f = lat / (Path(e["video_path"]).stem + ".pt")
            (out if f.exists() else missing).append({"kind": "latents", "name": f.stem, "path": str(f), "caption": e["caption"],
                                                      "class_name": e["class_name"]})
        if missing:
            print(f"  {len(missing)} selected video(s) have no pre-encoded file under {lat} and are skipped: "
                  + ", ".join(m["name"] for m in missing[:5]))
        return out
    if picked is None:
        raise FileNotFoundError(f"No video files found in {d}")
    return [{"kind": "video", "name": Path(e["video_path"]).stem, "path": e["video_path"], "caption": e["caption"],
             "class_name": e["class_name"]} for e in picked]


def load_entry(entry, args, dit, device, total_frames=None, pipe=None):
    h, w = {"480p": (60, 104), "720p": (90, 160)}[args.resolution]
    if entry["kind"] == "synthetic":
        T = _estimate_latent_len(total_frames if total_frames is not None else args.tta_total_frames)
        g = torch.Generator(device=device).manual_seed(entry["seed"])
        cy = dit.config.caption_channels
        lat = torch.randn((1, dit.config.in_channels, T, h, w), generator=g, device=device).to(torch.bfloat16)
        pe = torch.randn((1, 1, 512, cy), generator=g, device=device).to(torch.bfloat16)
        pm = torch.zeros((1, 512), dtype=torch.int64, device=device); pm[:, :77] = 1
        return dict(latents=lat, prompt_embeds=pe, prompt_mask=pm, negative_embeds=torch.zeros_like(pe), negative_mask=pm,
                    caption="synthetic")
    if entry["kind"] == "latents":
        blob = torch.load(entry["path"], map_location=device)
        blob.setdefault("caption", "")
        blob["latents"] = blob["latents"].to(torch.bfloat16)     # the TTA loss works on bf16 latents (common.py:463-466)
        fixed = getattr(args, "fixed_caption", None)
        if fixed is not None and entry.get("caption") is not None and entry["caption"] != blob["caption"]:
            # --fixed-caption over a pre-encoded clip: its stored embeddings belong to another caption
            if pipe is None or getattr(pipe, "tokenizer", None) is None or getattr(pipe, "text_encoder", None) is None:
                raise RuntimeError(f"--fixed-caption {entry['caption']!r}: {entry['path']} was pre-encoded with the caption "
                                   f"{blob['caption']!r} and this run has no tokenizer / text encoder to encode the override")
            from tta.common import encode_prompt
            blob["prompt_embeds"], blob["prompt_mask"] = encode_prompt(pipe.tokenizer, pipe.text_encoder, entry["caption"], device=device)
            blob["caption"] = entry["caption"]
        return blob
    # a raw video file: PyAV decode of the reference's frame windows, VAE encode, UMT5 prompt encode (tta/video_io.py)
    from .video_io import prepare_video_entry
    if pipe is None:
        raise RuntimeError("load_entry: a raw-video entry needs the pipeline (vae, tokenizer, text_encoder): pass pipe=")
    return prepare_video_entry(entry, args, pipe, device, total_frames)


def train_latents_variants_for(args, pipe, blob, entry, cond, train, device):
    """`--aug-enabled` (run_lora_tta.py:1100-1126): pixel-level variants of the TTA clip, each encoded by the VAE and cut to the
    training window.  Pixels come from the entry (`pixel_frames` [1, 3, T, H, W] in [-1, 1] of a pre-encoded clip); a synthetic
    entry gets seeded random pixels AND takes its latents from their encode, so that 'orig' and the variants describe one clip
    (returns the possibly replaced (cond, train, variants))."""
    if not getattr(args, "aug_enabled", False):
        return cond, train, None
    if pipe.vae is None:
        raise RuntimeError("--aug-enabled needs the VAE encoder (checkpoint without vae/)")
    from tta.augment import build_train_latents_variants
    px = blob.get("pixel_frames")
    if px is None and entry is not None and entry.get("kind") == "synthetic":
        H, W = {"480p": (480, 832), "720p": (720, 1280)}[args.resolution]
        n_lat = cond.shape[2] + train.shape[2]
        g = torch.Generator(device=device).manual_seed(entry["seed"] + 31)
        px = torch.rand((1, 3, 1 + 4 * (n_lat - 1), H, W), generator=g, device=device) * 2 - 1
        from tta.common import encode_video
        lat = encode_video(pipe.vae, px.to(pipe.vae.dtype), normalize=True).to(torch.bfloat16)
        cond, train = lat[:, :, :cond.shape[2]], lat[:, :, cond.shape[2]:n_lat]
    if px is None:
        raise RuntimeError(f"--aug-enabled: entry {entry.get('name') if entry else '?'} carries no `pixel_frames`")
    variants = build_train_latents_variants(pipe.vae, torch.as_tensor(px).to(device), cond, train, args)
    return cond, train, variants


def continuation_cond_latents(pipe, blob, entry, args, device):
    """Clean conditioning latents of the continuation, the way the reference obtains them: the `num_cond_frames` PIXEL frames
    that end at `gen_start_frame` are VAE-encoded ON THEIR OWN (lora_experiment/scripts/run_lora_tta.py:1197-1217 ->
    `generate_video_continuation` -> `pipe.generate_vc`, delta_experiment/scripts/common.py:566-611).  With a causal VAE that is
    NOT a slice of the longer TTA-window encode: the standalone clip gets its own first-frame treatment and spans exactly
    1 + 4k frames.  Sources, in order:
      `cond_latents`  normalised latents of that standalone encode, prepared with the entry;
      `cond_frames`   uint8 [n, H, W, 3] (or float [-1, 1] [1, 3, n, H, W]) pixels of those frames: encoded here with the HIP
                      encoder (posterior mode, normalised), the last 1 + 4k of them;
      synthetic entry seeded random pixels through the same encode (plumbing: the path is the real one);
      otherwise       the last latents of the TTA window — a DECLARED deviation (`cond_source = "sliced_tta_window"`): such an
                      entry's PSNR / SSIM are not comparable with the reference's tables.
    Returns (latents fp32 [1, C, n_lat, h, w], source)."""
    ncl = _estimate_latent_len(args.num_cond_frames)
    if blob.get("cond_latents") is not None:
        z = blob["cond_latents"].to(device).float()
        return z[:, :, -ncl:], "cond_latents"
    frames = blob.get("cond_frames")
    source = "cond_frames"
    if frames is None and entry is not None and entry.get("kind") == "synthetic" and pipe.vae is not None:
        H, W = {"480p": (480, 832), "720p": (720, 1280)}[args.resolution]
        g = torch.Generator(device=device).manual_seed(entry["seed"] + 104729)
        frames = torch.randint(0, 256, (args.num_cond_frames, H, W, 3), generator=g, device=device, dtype=torch.uint8)
        source = "synthetic_cond_frames"
    if frames is not None and pipe.vae is not None:
        f = torch.as_tensor(frames).to(device)
        if f.dtype == torch.uint8:
            f = (f.float() / 255.0 * 2.0 - 1.0).permute(3, 0, 1, 2).unsqueeze(0)          # [1, 3, n, H, W] in [-1, 1]
        n = f.shape[2]
        keep = 1 + 4 * ((n - 1) // 4)                                                     # the encoder takes 1 + 4k frames
        dist = pipe.vae.encode(f[:, :, n - keep:].to(pipe.vae.dtype)).latent_dist
        mean = torch.tensor(pipe.vae.config.latents_mean, device=device, dtype=torch.float32).view(1, -1, 1, 1, 1)
        std = torch.tensor(pipe.vae.config.latents_std, device=device, dtype=torch.float32).view(1, -1, 1, 1, 1)
        return (dist.mode().float() - mean) / std, source
    return blob["latents"][:, :, -ncl:].float(), "sliced_tta_window"


def generate_continuation(pipe, blob, args, idx, device, num_frames=None, entry=None):
    """KV-cached CFG continuation from the clean conditioning latents (`generate_video_continuation`, common.py:566-611).
    The conditioning encode is inside the timed region, as in the reference.  Returns (denoised latents, seconds); the
    origin of the conditioning latents is left in `blob["_cond_source"]` for the per-video result."""
    torch.cuda.synchronize()
    t0 = time.time()
    n_valid = num_frames_valid(num_frames if num_frames is not None else args.num_frames)
    T_lat = _estimate_latent_len(n_valid)
    cond, source = continuation_cond_latents(pipe, blob, entry, args, device)
    blob["_cond_source"] = source
    ncl = cond.shape[2]
    g = torch.Generator(device=device).manual_seed(args.seed + idx)
    lat = torch.randn((1, cond.shape[1], T_lat) + tuple(cond.shape[3:]), generator=g, device=device, dtype=torch.float32)
    lat[:, :, :ncl] = cond
    out = pipe.denoise(lat, blob["prompt_embeds"], blob["prompt_mask"], blob.get("negative_embeds"), blob.get("negative_mask"),
                       num_cond_latents=ncl, num_inference_steps=args.num_inference_steps,
                       guidance_scale=args.guidance_scale, use_kv_cache=True)
    torch.cuda.synchronize()
    return out, time.time() - t0


_VIDEO_WRITER_NOTE = [False]


def save_frames(pipe, latents, path_noext: str, frames: torch.Tensor = None, fps: int = 24):
    """The generated clip as `<name>.mp4` exactly as the reference writes it (`save_video_from_numpy`, run_lora_tta.py:641-647:
    `imageio.mimwrite(path, uint8 frames, fps=24, codec="libx264", quality=9)`) when `imageio` (+ its ffmpeg plugin) is importable;
    in an image without it (this one: no video encoder of any kind) the same uint8 frame stack goes to `<name>.npy` and one line
    says so.  Host-side file output only: nothing of the compute path depends on which of the two was written."""
    frames = pipe.decode_to_frames(latents) if frames is None else frames
    frames_u8 = (frames * 255).to(torch.uint8).cpu().numpy()
    try:
        import imageio
    except ImportError:
        imageio = None
    if imageio is not None:
        imageio.mimwrite(path_noext + ".mp4", frames_u8, fps=fps, codec="libx264", quality=9)
        return path_noext + ".mp4"
    if not _VIDEO_WRITER_NOTE[0]:
        _VIDEO_WRITER_NOTE[0] = True
        print("  (imageio is not installed: generated clips are saved as uint8 .npy frame stacks instead of .mp4)")
    np.save(path_noext + ".npy", frames_u8)
    return path_noext + ".npy"


def ground_truth_frames(blob, entry, shape, device):
    """uint8 [n, H, W, 3] ground truth of the generated frames at the output resolution: `gt_frames` of a pre-encoded
    entry (the PyAV decode + LANCZOS resize of common.py:698-715 happen when the entry is prepared), seeded noise for a
    synthetic one (plumbing: the numbers mean nothing, the path is the real one)."""
    gt = blob.get("gt_frames")
    if gt is not None:
        return torch.as_tensor(gt).to(device)
    if entry["kind"] == "synthetic":
        g = torch.Generator(device=device).manual_seed(entry["seed"] + 7919)
        return torch.randint(0, 256, tuple(shape), generator=g, device=device, dtype=torch.uint8)
    return None


def score_generation(frames: torch.Tensor, blob, entry, args, num_frames=None, flavour: str = "tta") -> Dict:
    """PSNR / SSIM / LPIPS of the generated frames against the ground truth, on the device (common.py:1233-1243)."""
    from tta.eval_metrics import evaluate_generation_metrics
    num_gen = (num_frames if num_frames is not None else args.num_frames) - args.num_cond_frames
    n_have = max(0, min(num_gen, frames.shape[0] - args.num_cond_frames))
    gt = ground_truth_frames(blob, entry, (n_have,) + tuple(frames.shape[1:]), frames.device)
    if gt is None or n_have == 0:
        return {"psnr": None, "ssim": None, "lpips": None}
    m = evaluate_generation_metrics(frames, gt, args.num_cond_frames, num_gen, flavour=flavour)
    return {k: (None if v != v else v) for k, v in m.items()}     # NaN (no LPIPS network offline) -> null in the JSON


def clip_gate_summary(args) -> Dict:
    return {"clip_gate_enabled": args.clip_gate_enabled, "clip_gate_threshold": args.clip_gate_threshold,
            "clip_gate_backend": args.clip_gate_backend, "clip_gate_model": args.clip_gate_model,
            "clip_gate_sample_frames": args.clip_gate_sample_frames, "clip_gate_aggregation": args.clip_gate_aggregation,
            "clip_gate_sampling_mode": "late_only" if args.clip_gate_late_only else args.clip_gate_sampling_mode,
            "clip_gate_late_fraction": args.clip_gate_late_fraction, "clip_gate_log_only": args.clip_gate_log_only,
            "clip_gate_fail_open": args.clip_gate_fail_open,
            "clip_gate_stats": {"skip_rate": 0.0, "num_skipped": 0, "num_scored": 0}}


def run_delta_method(args, method: str, make_wrapper: Callable, optimize_fn: Callable, params_of: Callable,
                     result_extra: Callable, summary_head: Dict, file_suffix: str, cleanup: Callable = None):
    """The per-video loop of the delta runners: fresh wrapper -> (anchored early stopping) -> optimise on the conditioning
    window -> continuation with the wrapper's hooks installed -> checkpoint after every video -> summary."""
    C.normalize_tta_frame_args(args)
    C.validate_tta_feature_budget(args, context=method)
    C.reject_out_of_scope(args)
    if getattr(args, "batch_videos", 1) != 1:
        raise NotImplementedError("retrieval-augmented batch TTA needs the sentence-transformer pool (SURVEY §2 #16)")
    rank, world, device = setup_distributed(args)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    os.makedirs(args.output_dir, exist_ok=True)
    dp.begin_job(args.output_dir, rank)
    videos_dir = os.path.join(args.output_dir, "videos"); os.makedirs(videos_dir, exist_ok=True)
    prior = dp.load_checkpoint(args.output_dir, rank if world > 1 else None)
    all_results = prior["results"] if prior else []
    done = {r["idx"] for r in all_results}
    dit, pipe = load_components(args, device)
    for p in dit.parameters():
        p.requires_grad = False
    entries = list_eval_entries(args, dit)
    my_idx = [i for i in dp.shard_indices(len(entries), rank, world) if i not in done]
    early_stopper = build_early_stopper_from_args(args)
    n_ctx_lat = _estimate_latent_len(args.tta_context_frames)
    from tta.inner_loop import choose_gradient_checkpointing

    for idx in my_idx:
        e = entries[idx]
        wrapper = None
        try:
            torch.manual_seed(dp.seed_for_video(args.seed, idx))
            blob = load_entry(e, args, dit, device, pipe=pipe)
            cond, train, val = split_tta_latents(blob["latents"], n_ctx_lat, args.es_holdout_fraction)
            n_tok = (cond.shape[2] + train.shape[2]) * (cond.shape[3] // 2) * (cond.shape[4] // 2)
            choose_gradient_checkpointing(dit, n_tok)
            wrapper = make_wrapper(dit).to(device)
            pe, pm = blob["prompt_embeds"], blob["prompt_mask"]
            # variants first (as in the LoRA / full runners): a synthetic entry's cond / train latents are REPLACED by the encode
            # of its own pixels, and the stopper's resident anchor batch must be built from the clip the loop trains on
            cond, train, variants = train_latents_variants_for(args, pipe, blob, e, cond, train, device)
            es = early_stopper if (early_stopper is not None and val is not None) else None
            if es is not None:
                # the wrapper is called directly, so the stopper scores its whole anchor set in one batched forward
                es.setup(model=wrapper, cond_latents=cond, val_latents=val, prompt_embeds=pe, prompt_mask=pm, device=device,
                         dtype=torch.bfloat16, video_id=e["name"])
            t0 = time.time()
            opt = optimize_fn(wrapper, cond, train, pe, pm, device, es, variants)
            torch.cuda.synchronize()
            train_time = time.time() - t0
            result = {"idx": idx, "video_name": e["name"], "video_path": e["path"], "caption": blob.get("caption", ""),
                      "train_time": train_time, "es_check_time": opt.get("es_check_time", 0.0),
                      "final_loss": opt["losses"][-1] if opt["losses"] else None, "batch_size": 1, "num_neighbors": 0,
                      "early_stopping_info": opt.get("early_stopping_info"), "success": True}
            if variants is not None:
                result["aug_variants"] = [v["name"] for v in variants]
            result.update(result_extra(opt))
            gen_time = 0.0
            if not args.skip_generation:
                wrapper.apply_to_dit()
                try:
                    out, gen_time = generate_continuation(pipe, blob, args, idx, device, entry=e)
                finally:
                    wrapper.remove_from_dit()
                if pipe.vae is not None:
                    t1 = time.time()
                    frames = pipe.decode_to_frames(out)
                    torch.cuda.synchronize()
                    gen_time += time.time() - t1              # the reference's gen_time includes the decode (generate_vc)
                    result.update(score_generation(frames, blob, e, args))
                    if not args.no_save_videos:
                        result["output_path"] = save_frames(pipe, out, os.path.join(videos_dir, f"{e['name']}_{file_suffix}"),
                                                            frames=frames)
                result["gen_time"] = gen_time
            result["total_time"] = train_time + gen_time
            print(f"  [{idx}] {e['name']}: train {train_time:.1f}s loss {result['final_loss']}"
                  + (f" gen {gen_time:.1f}s" if not args.skip_generation else ""))
            all_results.append(result)
        except Exception as ex:  # recorded and skipped, like the reference (run_delta_a.py:881-893)
            import traceback
            print(f"  ERROR: {ex}")
            traceback.print_exc()
            all_results.append({"idx": idx, "video_name": e["name"], "video_path": e["path"], "error": str(ex),
                                "success": False})
            if getattr(ex, "fatal", False):   # a failed launch / device error: the HIP context may be dead — stop here
                dp.write_checkpoint(args.output_dir, idx, all_results, rank=rank if world > 1 else None)
                raise
        finally:
            if cleanup is not None and wrapper is not None:
                cleanup(wrapper)   # e.g. norm tuning: put the job's original weights back before the next video
        dp.write_checkpoint(args.output_dir, idx + world, all_results, rank=rank if world > 1 else None)

    # end-of-job merge: file rendezvous with a bounded wait (a wedged peer must not park this rank for a day)
    took = [r.get("total_time") or r.get("train_time") or 0.0 for r in all_results if r.get("success")]
    wait_s = dp.merge_wait_seconds(max(took) if took else 0.0, len(all_results))
    merged = (dp.gather_results(all_results, output_dir=args.output_dir, wait_s=wait_s) if world > 1
              else dp.merge_results([all_results]))
    if rank == 0:
        ok = [r for r in merged if r.get("success", False)]
        mean = lambda k: float(np.mean([r.get(k, 0.0) or 0.0 for r in ok])) if ok else 0
        summary = {"method": method}
        summary.update(summary_head)
        summary.update({"num_cond_frames": args.num_cond_frames, "num_frames": args.num_frames,
                        "gen_start_frame": args.gen_start_frame, "num_videos": len(merged), "num_successful": len(ok),
                        "avg_train_time": mean("train_time"), "avg_clip_gate_eval_time": 0,
                        "avg_es_check_time": mean("es_check_time"), "avg_gen_time": mean("gen_time"),
                        "avg_total_time": mean("total_time")})
        if hasattr(args, "clip_gate_enabled"):
            summary.update(clip_gate_summary(args))
        summary["results"] = merged
        from tta.eval_metrics import aggregate_quality_metrics
        aggregate_quality_metrics(summary)
        dp.write_checkpoint(args.output_dir, dp.contiguous_next_idx(merged), merged)
        with open(os.path.join(args.output_dir, "summary.json"), "w") as f:
            json.dump(summary, f, indent=2, default=str)
        print(f"{method} complete: {len(ok)}/{len(merged)} videos")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
        if rank == 0 and dp.exit_code_after_merge():
            sys.exit(dp.exit_code_after_merge())      # summary.json is written, but a peer never delivered its final rows
