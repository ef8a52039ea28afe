#!/usr/bin/env python3
"""No-TTA baseline continuation on MI355X — same relative path, CLI flags and `summary.json` / `per_video_metrics.csv`
schemas as the reference's `baseline_experiment/scripts/run_baseline.py` (flags :235-262, summary :515-550: the
"metrics" + "timing.per_video_inference_s" form `export_all_results.py:132-166` recognises as the baseline variant).
Inputs are those of the TTA runners (`latents/*.pt` or `synthetic:N`); pixel metrics (PSNR / SSIM / LPIPS) are the
on-device-eval row that comes after the path and are reported as null.  Under `torch.distributed.run`
the videos are sharded over ranks (the reference launches it with torchrun too, :76-79 — there for context parallelism)."""
import argparse
import csv
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
from tta import runner_common as R  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="LongCat-Video baseline inference (MI355X)")
    p.add_argument("--checkpoint-dir", type=str, required=True)
    p.add_argument("--data-dir", type=str, required=True)
    p.add_argument("--output-dir", type=str, required=True)
    p.add_argument("--num-cond-frames", type=int, default=2)
    p.add_argument("--num-gen-frames", type=int, default=14)
    p.add_argument("--gen-start-frame", type=int, default=32)
    p.add_argument("--resolution", type=str, default="480p", choices=["480p", "720p"])
    p.add_argument("--num-inference-steps", type=int, default=50)
    p.add_argument("--guidance-scale", type=float, default=4.0)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--max-videos", type=int, default=100)
    p.add_argument("--save-videos", action="store_true")
    p.add_argument("--device", type=str, default="cuda")
    return p


def _stats(vals):
    """run_baseline.py:547-554: mean / std / min / max to 4 places, {} when nothing was scored."""
    if not vals:
        return {}
    return {"mean": round(float(np.mean(vals)), 4), "std": round(float(np.std(vals)), 4), "min": round(min(vals), 4),
            "max": round(max(vals), 4)}


def main(argv=None):
    args = build_parser().parse_args(argv)
    wall_start = time.time()
    rank, world, device = R.setup_distributed(args)
    torch.manual_seed(args.seed)
    out_dir = Path(args.output_dir); out_dir.mkdir(parents=True, exist_ok=True)
    dp.begin_job(str(out_dir), rank)
    t0 = time.time()
    dit, pipe = R.load_components(args, device)
    model_load_time = time.time() - t0
    num_frames = args.num_cond_frames + args.num_gen_frames           # run_baseline.py:300
    args.num_frames = num_frames
    entries = R.list_eval_entries(args, dit)
    rows = []
    t_inf = time.time()
    for idx in dp.shard_indices(len(entries), rank, world):
        e = entries[idx]
        try:
            blob = R.load_entry(e, args, dit, device, total_frames=args.num_cond_frames, pipe=pipe)
            out, dt = R.generate_continuation(pipe, blob, args, idx, device, num_frames=num_frames, entry=e)
            row = {"idx": idx, "index": idx, "filename": e["name"], "caption": blob.get("caption", ""), "psnr": None,
                   "ssim": None, "lpips": None, "resolution": args.resolution, "inference_time_s": round(dt, 2)}
            if pipe.vae is not None:
                t1 = time.time()
                frames = pipe.decode_to_frames(out)
                torch.cuda.synchronize()
                dt += time.time() - t1                          # the reference times generate_vc, decode included
                row["inference_time_s"] = round(dt, 2)
                row["resolution"] = f"{frames.shape[1]}x{frames.shape[2]}"
                # run_baseline.py:436-455: its own float64 PSNR (60 dB cap) and skimage-default SSIM, rounded to 4 places
                m = R.score_generation(frames, blob, e, args, num_frames=num_frames, flavour="baseline")
                row.update({k: (None if v is None else round(v, 4)) for k, v in m.items()})
                if args.save_videos:
                    (out_dir / "videos").mkdir(exist_ok=True)
                    R.save_frames(pipe, out, str(out_dir / "videos" / e["name"]), frames=frames)
            print(f"  [{idx}] {e['name']}: {dt:.1f}s")
        except Exception as ex:  # recorded and skipped (run_baseline.py:492-501)
            import traceback
            print(f"  ERROR: {ex}")
            traceback.print_exc()
            row = {"idx": idx, "index": idx, "filename": e["name"], "caption": "", "psnr": None, "ssim": None, "lpips": None,
                   "error": str(ex)}
        rows.append(row)
    total_inference_time = time.time() - t_inf
    took = [r.get("inference_time_s", 0.0) for r in rows]
    merged = (dp.gather_results(rows, output_dir=str(out_dir), wait_s=dp.merge_wait_seconds(max(took) if took else 0.0, len(rows)))
              if world > 1 else dp.merge_results([rows]))
    if rank == 0:
        times = [r["inference_time_s"] for r in merged if "inference_time_s" in r]
        with open(out_dir / "per_video_metrics.csv", "w", newline="") as f:
            wr = csv.DictWriter(f, fieldnames=["index", "filename", "caption", "psnr", "ssim", "lpips", "resolution",
                                               "inference_time_s"], extrasaction="ignore")
            wr.writeheader()
            wr.writerows(merged)
        summary = {
            "experiment": "baseline_inference", "model": "LongCat-Video", "checkpoint_dir": args.checkpoint_dir,
            "resolution": args.resolution, "num_cond_frames": args.num_cond_frames, "num_gen_frames": args.num_gen_frames,
            "gen_start_frame": args.gen_start_frame, "num_frames_total": num_frames,
            "num_inference_steps": args.num_inference_steps, "guidance_scale": args.guidance_scale, "seed": args.seed,
            "num_videos": len(merged), "num_successful": len(times),
            "timing": {"model_load_s": round(model_load_time, 2), "total_inference_s": round(total_inference_time, 2),
                       "wall_total_s": round(time.time() - wall_start, 2),
                       "per_video_inference_s": {"mean": round(float(np.mean(times)), 2) if times else None,
                                                 "std": round(float(np.std(times)), 2) if times else None,
                                                 "min": round(min(times), 2) if times else None,
                                                 "max": round(max(times), 2) if times else None}},
            "metrics": {k: _stats([r[k] for r in merged if r.get(k) is not None]) for k in ("psnr", "ssim", "lpips")},
            "runtime": {"backend": "mi355x-hip", "world_size": world},
        }
        with open(out_dir / "summary.json", "w") as f:
            json.dump(summary, f, indent=2)
        print(f"baseline complete: {len(times)}/{len(merged)} videos")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
        if rank == 0 and dp.exit_code_after_merge():
            sys.exit(dp.exit_code_after_merge())      # summary.json is written, but a peer never delivered its final rows


if __name__ == "__main__":
    main()
