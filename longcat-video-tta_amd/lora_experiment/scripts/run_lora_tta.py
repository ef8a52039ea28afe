#!/usr/bin/env python3
"""LoRA test-time adaptation runner on MI355X — same relative path, CLI flags and artifact schemas as the reference's
`lora_experiment/scripts/run_lora_tta.py` (flags :654-741; `config.json` :855-908; per-video result keys :1174-1248;
`checkpoint.json` = {next_idx, results}; `summary.json` :1276-1324), so `sweep_experiment/sbatch/run_sweep.sbatch:375-438`
drives it unchanged.

Inputs.  The hot path starts at latents: `--data-dir` holds `latents/*.pt` files
({"latents": [1,16,T,h,w] normalised VAE latents of the conditioning window, "prompt_embeds": [1,1,L,4096],
"prompt_mask": [1,L], optional "negative_embeds"/"negative_mask", "caption"}), or is `synthetic:N` for seeded synthetic
videos (plumbing / benchmarking).  Raw-video decoding, the VAE encoder and the UMT5 text encoder are the caller-side rows
that come next; pointing `--data-dir` at raw videos raises a per-video error that is recorded exactly the
way the reference records failures (:1264-1271).

Multi-GPU.  Launched under `torch.distributed.run` the videos are sharded `idx = rank (mod world)` (one video per
GPU, no data-path collective), each rank writes `checkpoint.rank{r}.json`, rank 0 merges into the reference-format
`checkpoint.json` / `summary.json`.  sigma / eps are re-seeded per video with `seed + idx` (the reference's single RNG
stream is sequential across videos; declared deviation).
"""
import argparse
import functools
import json
import os
import sys
import time
from pathlib import Path

_HERE = Path(__file__).resolve()
_PKG = _HERE.parents[2]
if str(_PKG) not in sys.path:
    sys.path.insert(0, str(_PKG))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from longcat_video.parallel import data_parallel as dp  # noqa: E402
from tta import cli_args as C  # noqa: E402
from tta.early_stopping import add_early_stopping_args, build_early_stopper_from_args  # noqa: E402
from tta.inner_loop import choose_gradient_checkpointing, finetune_lora_on_conditioning  # noqa: E402
from tta.latent_split import _estimate_latent_len, num_frames_valid, split_tta_latents  # noqa: E402
from tta.runner_common import (list_eval_entries, load_components, load_entry, save_frames,  # noqa: E402,F401
                               score_generation)
from tta.lora import (count_builtin_lora_parameters, get_builtin_lora_parameters, inject_builtin_lora_into_dit,  # noqa: E402
                      reset_builtin_lora_weights, save_builtin_lora_weights)
from tta.lora import (count_lora_parameters, get_lora_parameters, inject_lora_into_dit, reset_lora_weights,  # noqa: E402
                      save_lora_weights)


def build_parser():
    p = argparse.ArgumentParser(description="LoRA TTA for LongCat-Video (MI355X)")
    p.add_argument("--checkpoint-dir", type=str, required=True, help="checkpoint dir, or synthetic[:depth] for random init")
    p.add_argument("--data-dir", type=str, required=True)
    p.add_argument("--output-dir", type=str, required=True)
    p.add_argument("--restart", action="store_true")
    p.add_argument("--lora-rank", type=int, default=8)
    p.add_argument("--lora-alpha", type=float, default=16.0)
    p.add_argument("--lora-dropout", type=float, default=0.0)
    p.add_argument("--target-ffn", action="store_true")
    p.add_argument("--target-modules", type=str, default="qkv,proj")
    p.add_argument("--lora-target-blocks", type=str, default="all")
    p.add_argument("--use-builtin-lora", action="store_true")
    p.add_argument("--save-lora-weights", action="store_true")
    p.add_argument("--learning-rate", type=float, default=2e-4)
    p.add_argument("--num-steps", type=int, default=20)
    p.add_argument("--warmup-steps", type=int, default=3)
    p.add_argument("--weight-decay", type=float, default=0.01)
    p.add_argument("--max-grad-norm", type=float, default=1.0)
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
    p.add_argument("--batch-videos", type=int, default=1)
    p.add_argument("--batch-method", type=str, default="similarity", choices=["similarity", "sequential"])
    p.add_argument("--retrieval-pool-dir", type=str, default=None)
    add_early_stopping_args(p)
    C.add_augmentation_args(p)
    C.add_tta_frame_args(p)
    C.add_caption_guard_args(p)
    C.add_caption_override_args(p)
    C.add_feature_frame_guard_args(p)
    C.add_online_eval_args(p)
    C.add_clip_gate_args(p)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    C.normalize_tta_frame_args(args)
    C.validate_tta_feature_budget(args, context="lora_tta")
    C.reject_out_of_scope(args)
    if args.batch_videos != 1:
        raise NotImplementedError("retrieval-augmented batch TTA needs the sentence-transformer pool (SURVEY §2 #16)")

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")
        dp.host_group()          # the gloo group of the end-of-job merge is created while every rank is still here
    device = f"cuda:{local_rank}" if args.device.startswith("cuda") else args.device
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    os.makedirs(args.output_dir, exist_ok=True)
    dp.begin_job(args.output_dir, rank)
    videos_dir = os.path.join(args.output_dir, "videos"); os.makedirs(videos_dir, exist_ok=True)
    lora_dir = os.path.join(args.output_dir, "lora_weights")
    if args.save_lora_weights:
        os.makedirs(lora_dir, exist_ok=True)

    prior = None if args.restart else dp.load_checkpoint(args.output_dir, rank if world > 1 else None)
    all_results = prior["results"] if prior else []
    done = {r["idx"] for r in all_results}

    dit, pipe = load_components(args, device)
    # run_lora_tta.py:806-811 switches block checkpointing on unconditionally; here it is decided per video from the
    # token count (LCV_TTA_CHECKPOINT=on reproduces the reference): see tta.inner_loop.choose_gradient_checkpointing
    for p in dit.parameters():                             # :813-815
        p.requires_grad = False
    target_modules = [m.strip() for m in args.target_modules.split(",") if m.strip()]
    use_builtin = bool(args.use_builtin_lora)                  # run_lora_tta.py:783, 818-846
    lora_impl = "builtin" if use_builtin else "custom"
    if use_builtin:
        lora_modules = inject_builtin_lora_into_dit(dit, rank=args.lora_rank, alpha=args.lora_alpha, target_modules=target_modules,
                                                    target_ffn=args.target_ffn, target_blocks=args.lora_target_blocks)
        counts = count_builtin_lora_parameters(lora_modules)
        get_params = lambda: get_builtin_lora_parameters(lora_modules)
        reset_adapters = lambda: reset_builtin_lora_weights(lora_modules)
        save_adapters = save_builtin_lora_weights
    else:
        lora_modules = inject_lora_into_dit(dit, rank=args.lora_rank, alpha=args.lora_alpha, dropout=args.lora_dropout,
                                            target_modules=target_modules, target_ffn=args.target_ffn,
                                            target_blocks=args.lora_target_blocks)
        counts = count_lora_parameters(lora_modules)
        get_params = lambda: get_lora_parameters(lora_modules)
        reset_adapters = lambda: reset_lora_weights(lora_modules)
        save_adapters = save_lora_weights
    if rank == 0:
        exp_config = {
            "method": f"lora_tta_{lora_impl}",
            "lora": {"implementation": lora_impl, "rank": args.lora_rank, "alpha": args.lora_alpha,
                     "dropout": args.lora_dropout, "target_modules": target_modules,
                     "target_blocks": args.lora_target_blocks, "target_ffn": args.target_ffn,
                     "num_modules": len(lora_modules), "trainable_params": counts["trainable"]},
            "training": {"learning_rate": args.learning_rate, "num_steps": args.num_steps,
                         "warmup_steps": args.warmup_steps, "weight_decay": args.weight_decay,
                         "max_grad_norm": args.max_grad_norm},
            "generation": {"num_cond_frames": args.num_cond_frames, "num_frames": args.num_frames,
                           "num_inference_steps": args.num_inference_steps, "guidance_scale": args.guidance_scale,
                           "resolution": args.resolution},
            "seed": args.seed, "max_videos": args.max_videos, "clip_gate_enabled": args.clip_gate_enabled,
            "clip_gate_threshold": args.clip_gate_threshold, "clip_gate_backend": args.clip_gate_backend,
            "clip_gate_model": args.clip_gate_model, "clip_gate_sample_frames": args.clip_gate_sample_frames,
            "clip_gate_aggregation": args.clip_gate_aggregation,
            "clip_gate_sampling_mode": "late_only" if args.clip_gate_late_only else args.clip_gate_sampling_mode,
            "clip_gate_late_fraction": args.clip_gate_late_fraction, "clip_gate_log_only": args.clip_gate_log_only,
            "clip_gate_fail_open": args.clip_gate_fail_open,
            "runtime": {"backend": "mi355x-hip", "world_size": world},
        }
        with open(os.path.join(args.output_dir, "config.json"), "w") as f:
            json.dump(exp_config, f, indent=2)

    entries = list_eval_entries(args, dit)
    my_idx = [i for i in dp.shard_indices(len(entries), rank, world) if i not in done]
    early_stopper = build_early_stopper_from_args(args)
    n_ctx_lat = _estimate_latent_len(args.tta_context_frames)

    for idx in my_idx:
        e = entries[idx]
        try:
            torch.manual_seed(dp.seed_for_video(args.seed, idx))
            blob = load_entry(e, args, dit, device, pipe=pipe)
            cond, train, val = split_tta_latents(blob["latents"], n_ctx_lat, args.es_holdout_fraction)
            n_tok = (cond.shape[2] + train.shape[2]) * (cond.shape[3] // 2) * (cond.shape[4] // 2)
            choose_gradient_checkpointing(dit, n_tok)
            from tta.runner_common import train_latents_variants_for
            cond, train, variants = train_latents_variants_for(args, pipe, blob, e, cond, train, device)   # --aug-enabled
            reset_adapters()
            es = early_stopper if (early_stopper is not None and val is not None) else None
            if es is not None:
                es.setup(dit, cond, val, blob["prompt_embeds"], blob["prompt_mask"], device=device, dtype=torch.bfloat16,
                         video_id=e["name"])
            tr = finetune_lora_on_conditioning(dit, lora_modules, cond, train, blob["prompt_embeds"], blob["prompt_mask"],
                                               num_steps=args.num_steps, lr=args.learning_rate,
                                               warmup_steps=args.warmup_steps, weight_decay=args.weight_decay,
                                               max_grad_norm=args.max_grad_norm, device=device, dtype=torch.bfloat16,
                                               early_stopper=es, lora_param_fn=get_params, train_latents_variants=variants)
            result = {"idx": idx, "video_name": e["name"], "video_path": e["path"], "caption": blob.get("caption", ""),
                      "train_time": tr["train_time"], "es_check_time": tr.get("es_check_time", 0.0),
                      "final_loss": tr["losses"][-1] if tr["losses"] else None, "num_train_steps": len(tr["losses"]),
                      "batch_size": 1, "num_neighbors": 0, "early_stopping_info": tr.get("early_stopping_info"),
                      "success": True}
            if variants is not None:
                result["aug_variants"] = [v["name"] for v in variants]
            gen_time = 0.0
            if not args.skip_generation:
                from tta.runner_common import generate_continuation
                out, gen_only = generate_continuation(pipe, blob, args, idx, device, entry=e)   # cond encode + denoise
                result["cond_source"] = blob.get("_cond_source")
                t0 = time.time() - gen_only
                frames = pipe.decode_to_frames(out) if pipe.vae is not None else None
                torch.cuda.synchronize()
                gen_time = time.time() - t0
                result["gen_time"] = gen_time
                if frames is not None:
                    result.update(score_generation(frames, blob, e, args))     # run_lora_tta.py:1233-1243
                    if not args.no_save_videos:
                        result["output_path"] = save_frames(pipe, out, os.path.join(videos_dir, f"{e['name']}_lora"),
                                                            frames=frames)
            result["total_time"] = tr["train_time"] + gen_time
            if args.save_lora_weights:
                save_adapters(lora_modules, os.path.join(lora_dir, f"{e['name']}_lora.pt"))
            print(f"  [{idx}] {e['name']}: train {tr['train_time']:.1f}s loss {result['final_loss']}"
                  + (f" gen {gen_time:.1f}s" if not args.skip_generation else ""))
            all_results.append(result)
        except Exception as ex:  # recorded and skipped, like the reference (:1264-1271)
            import traceback
            print(f"  ERROR: {ex}")
            traceback.print_exc()
            all_results.append({"idx": idx, "video_name": e["name"], "video_path": e["path"], "error": str(ex),
                                "success": False})
            if getattr(ex, "fatal", False):   # a failed launch / device error: the HIP context may be dead — stop here
                dp.write_checkpoint(args.output_dir, idx, all_results, rank=rank if world > 1 else None)
                raise
        dp.write_checkpoint(args.output_dir, idx + world, all_results, rank=rank if world > 1 else None)

    # end-of-job merge: file rendezvous with a bounded wait (a wedged peer must not park this rank for a day)
    took = [r.get("total_time") or r.get("train_time") or 0.0 for r in all_results if r.get("success")]
    wait_s = dp.merge_wait_seconds(max(took) if took else 0.0, len(all_results))
    merged = (dp.gather_results(all_results, output_dir=args.output_dir, wait_s=wait_s) if world > 1
              else dp.merge_results([all_results]))
    if rank == 0:
        ok = [r for r in merged if r.get("success", False)]
        mean = lambda k: float(np.mean([r.get(k, 0.0) or 0.0 for r in ok])) if ok else 0
        summary = {"method": "lora_tta", "lora_rank": args.lora_rank, "lora_alpha": args.lora_alpha,
                   "learning_rate": args.learning_rate, "num_steps": args.num_steps,
                   "num_cond_frames": args.num_cond_frames, "num_frames": args.num_frames,
                   "gen_start_frame": args.gen_start_frame, "batch_videos": args.batch_videos,
                   "retrieval_pool_dir": args.retrieval_pool_dir, "num_videos": len(merged), "num_successful": len(ok),
                   "num_failed": len(merged) - len(ok), "avg_train_time": mean("train_time"),
                   "avg_clip_gate_eval_time": 0, "avg_es_check_time": mean("es_check_time"),
                   "avg_gen_time": mean("gen_time"), "avg_total_time": mean("total_time"),
                   "avg_final_loss": (lambda v: float(np.mean(v)) if v else None)(
                       [r["final_loss"] for r in ok if r.get("final_loss") is not None]),
                   "clip_gate_enabled": False, "clip_gate_stats": {"skip_rate": 0.0, "num_skipped": 0, "num_scored": 0},
                   "results": merged}
        from tta.eval_metrics import aggregate_quality_metrics
        aggregate_quality_metrics(summary)                                     # run_lora_tta.py:1315 (common.py:2453-2458)
        dp.write_checkpoint(args.output_dir, dp.contiguous_next_idx(merged), merged)
        with open(os.path.join(args.output_dir, "summary.json"), "w") as f:
            json.dump(summary, f, indent=2, default=str)
        print(f"LoRA TTA complete: {len(ok)}/{len(merged)} videos")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
        if rank == 0 and dp.exit_code_after_merge():
            sys.exit(dp.exit_code_after_merge())      # summary.json is written, but a peer never delivered its final rows


if __name__ == "__main__":
    main()
