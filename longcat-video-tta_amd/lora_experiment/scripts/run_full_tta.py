#!/usr/bin/env python3
"""Full-model TTA runner on MI355X — same relative path, CLI flags and artifact schemas as the reference's
`lora_experiment/scripts/run_full_tta.py` (flags :337-376 + shared groups; `config.json` :466-510, `checkpoint.json`,
`summary.json` :864-905).  All DiT parameters are unfrozen (:452-453), block checkpointing is switched on (:444-449), the
base weights are snapshotted once and restored before every video (:462, :222-228 — on the device here), the inner loop is
`tta.full_tta.finetune_full_on_conditioning` (SGD by default, `--optimizer adamw` optional), generation and scoring are
the LoRA runner's.  One process per GPU under torch.distributed.run shards the videos (no collective on the path).
"""
import argparse
import functools
import json
import os
import sys
import time
from pathlib import Path

_PKG = Path(__file__).resolve().parents[2]
if str(_PKG) not in sys.path:
    sys.path.insert(0, str(_PKG))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from longcat_video.parallel import data_parallel as dp  # noqa: E402
from tta import cli_args as C  # noqa: E402
from tta import runner_common as R  # noqa: E402
from tta.early_stopping import build_early_stopper_from_args  # noqa: E402
from tta.full_tta import finetune_full_on_conditioning, reset_dit_weights, snapshot_base_state  # noqa: E402
from tta.latent_split import _estimate_latent_len, split_tta_latents  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="Full-model TTA for LongCat-Video (MI355X)")
    R.add_common_args(p)
    p.add_argument("--restart", action="store_true")
    p.add_argument("--learning-rate", type=float, default=1e-5)
    p.add_argument("--num-steps", type=int, default=10)
    p.add_argument("--warmup-steps", type=int, default=2)
    p.add_argument("--weight-decay", type=float, default=0.01)
    p.add_argument("--max-grad-norm", type=float, default=1.0)
    p.add_argument("--optimizer", type=str, default="sgd", choices=["sgd", "adamw"])
    p.add_argument("--batch-videos", type=int, default=1)
    p.add_argument("--batch-method", type=str, default="similarity", choices=["similarity", "sequential"])
    p.add_argument("--retrieval-pool-dir", type=str, default=None)
    R.add_shared_groups(p, clip_gate=True)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    C.normalize_tta_frame_args(args)
    C.validate_tta_feature_budget(args, context="full_tta")
    C.reject_out_of_scope(args)
    if args.batch_videos != 1:
        raise NotImplementedError("retrieval-augmented batch TTA needs the sentence-transformer pool (SURVEY §2 #16)")
    rank, world, device = R.setup_distributed(args)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    os.makedirs(args.output_dir, exist_ok=True)
    dp.begin_job(args.output_dir, rank)
    videos_dir = os.path.join(args.output_dir, "videos"); os.makedirs(videos_dir, exist_ok=True)
    prior = None if args.restart else dp.load_checkpoint(args.output_dir, rank if world > 1 else None)
    all_results = prior["results"] if prior else []
    done = {r["idx"] for r in all_results}

    dit, pipe = R.load_components(args, device)
    from torch.utils.checkpoint import checkpoint as _ckpt_fn
    dit.gradient_checkpointing = True                                                   # :444-449
    dit._gradient_checkpointing_func = functools.partial(_ckpt_fn, use_reentrant=False)
    for p in dit.parameters():                                                          # :452-453
        p.requires_grad = True
    total_params = sum(p.numel() for p in dit.parameters())
    trainable_params = sum(p.numel() for p in dit.parameters() if p.requires_grad)
    base_state = snapshot_base_state(dit)                                               # :462 (device-resident here)
    if rank == 0:
        gate = R.clip_gate_summary(args)
        exp_config = {
            "method": "full_tta",
            "training": {"learning_rate": args.learning_rate, "num_steps": args.num_steps, "warmup_steps": args.warmup_steps,
                         "weight_decay": args.weight_decay, "max_grad_norm": args.max_grad_norm, "optimizer": args.optimizer,
                         "total_params": total_params, "trainable_params": trainable_params},
            "generation": {"num_cond_frames": args.num_cond_frames, "num_frames": args.num_frames,
                           "num_inference_steps": args.num_inference_steps, "guidance_scale": args.guidance_scale,
                           "resolution": args.resolution},
            "seed": args.seed, "max_videos": args.max_videos,
            **{k: v for k, v in gate.items() if k != "clip_gate_stats"},
            "clip_gate": {"enabled": args.clip_gate_enabled, "threshold": args.clip_gate_threshold,
                          "backend": args.clip_gate_backend, "model": args.clip_gate_model,
                          "sample_frames": args.clip_gate_sample_frames, "aggregation": args.clip_gate_aggregation,
                          "sampling_mode": gate["clip_gate_sampling_mode"], "late_fraction": args.clip_gate_late_fraction,
                          "log_only": args.clip_gate_log_only, "fail_open": args.clip_gate_fail_open},
            "runtime": {"backend": "mi355x-hip", "world_size": world},
        }
        with open(os.path.join(args.output_dir, "config.json"), "w") as f:
            json.dump(exp_config, f, indent=2)

    entries = R.list_eval_entries(args, dit)
    my_idx = [i for i in dp.shard_indices(len(entries), rank, world) if i not in done]
    early_stopper = build_early_stopper_from_args(args)
    n_ctx_lat = _estimate_latent_len(args.tta_context_frames)

    for idx in my_idx:
        e = entries[idx]
        try:
            torch.manual_seed(dp.seed_for_video(args.seed, idx))
            reset_dit_weights(dit, base_state)                                          # :640 per-video reset
            blob = R.load_entry(e, args, dit, device, pipe=pipe)
            cond, train, val = split_tta_latents(blob["latents"], n_ctx_lat, args.es_holdout_fraction)
            cond, train, variants = R.train_latents_variants_for(args, pipe, blob, e, cond, train, device)   # --aug-enabled (:699-722)
            pe, pm = blob["prompt_embeds"], blob["prompt_mask"]
            es = early_stopper if (early_stopper is not None and val is not None) else None
            if es is not None:                                                          # :724-742 (the stopper snapshots the model itself)
                es.setup(model=dit, cond_latents=cond, val_latents=val, prompt_embeds=pe, prompt_mask=pm, device=device,
                         dtype=torch.bfloat16, video_id=e["name"])
            tr = finetune_full_on_conditioning(dit, cond, train, pe, pm, num_steps=args.num_steps, lr=args.learning_rate,
                                               warmup_steps=args.warmup_steps, weight_decay=args.weight_decay,
                                               max_grad_norm=args.max_grad_norm, device=device, dtype=torch.bfloat16,
                                               early_stopper=es, optimizer_type=args.optimizer, train_latents_variants=variants)
            result = {"idx": idx, "video_name": e["name"], "video_path": e["path"], "caption": blob.get("caption", ""),
                      "train_time": tr["train_time"], "es_check_time": tr.get("es_check_time", 0.0),
                      "final_loss": tr["losses"][-1] if tr["losses"] else None, "num_train_steps": len(tr["losses"]),
                      "batch_size": 1, "num_neighbors": 0, "early_stopping_info": tr.get("early_stopping_info"), "success": True}
            gen_time = 0.0
            if not args.skip_generation:
                out, gen_time = R.generate_continuation(pipe, blob, args, idx, device, entry=e)
                if pipe.vae is not None:
                    t1 = time.time()
                    frames = pipe.decode_to_frames(out)
                    torch.cuda.synchronize()
                    gen_time += time.time() - t1
                    result.update(R.score_generation(frames, blob, e, args))
                    if not args.no_save_videos:
                        result["output_path"] = R.save_frames(pipe, out, os.path.join(videos_dir, f"{e['name']}_full"), frames=frames)
                result["gen_time"] = gen_time
            result["total_time"] = tr["train_time"] + gen_time
            print(f"  [{idx}] {e['name']}: train {tr['train_time']:.1f}s loss {result['final_loss']}"
                  + (f" gen {gen_time:.1f}s" if not args.skip_generation else ""))
            all_results.append(result)
        except Exception as ex:  # recorded and skipped, like the reference (:853-861)
            import traceback
            print(f"  ERROR: {ex}")
            traceback.print_exc()
            all_results.append({"idx": idx, "video_name": e["name"], "video_path": e["path"], "error": str(ex), "success": False})
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
        summary = {"method": "full_tta", "learning_rate": args.learning_rate, "num_steps": args.num_steps,
                   "num_cond_frames": args.num_cond_frames, "num_frames": args.num_frames,
                   "gen_start_frame": args.gen_start_frame, "batch_videos": args.batch_videos,
                   "retrieval_pool_dir": args.retrieval_pool_dir, "total_params": total_params, "num_videos": len(merged),
                   "num_successful": len(ok), "num_failed": len(merged) - len(ok), "avg_train_time": mean("train_time"),
                   "avg_clip_gate_eval_time": 0, "avg_es_check_time": mean("es_check_time"), "avg_gen_time": mean("gen_time"),
                   "avg_total_time": mean("total_time"),
                   "avg_final_loss": (lambda v: float(np.mean(v)) if v else None)(
                       [r["final_loss"] for r in ok if r.get("final_loss") is not None])}
        summary.update(R.clip_gate_summary(args))
        summary["results"] = merged
        from tta.eval_metrics import aggregate_quality_metrics
        aggregate_quality_metrics(summary)
        dp.write_checkpoint(args.output_dir, dp.contiguous_next_idx(merged), merged)
        with open(os.path.join(args.output_dir, "summary.json"), "w") as f:
            json.dump(summary, f, indent=2, default=str)
        print(f"Full TTA complete: {len(ok)}/{len(merged)} videos")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
        if rank == 0 and dp.exit_code_after_merge():
            sys.exit(dp.exit_code_after_merge())      # summary.json is written, but a peer never delivered its final rows


if __name__ == "__main__":
    main()
