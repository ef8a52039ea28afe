#!/usr/bin/env python3
"""ONE TTA video end to end at BASELINE.json's headline size, through the PRODUCT RUNNER (the reference's per-video loop,
lora_experiment/scripts/run_lora_tta.py:974-1273; timing keys :1208-1248): 48 blocks with synthetic weights, a synthetic 720p
entry, LoRA r = 8 on qkv + proj of all blocks, 20 inner steps with the anchored early stopper on, then the 50-step KV-cached CFG
continuation to 49 frames, VAE decode and the on-device PSNR / SSIM.  Writes the run's own `summary.json` timing keys plus the
wall clock around the whole `main()` (model build excluded / included) to the JSON named on the command line.

usage (GPU box): python tools/tta_video_end_to_end.py gpurun_out/r04_tta_video_k3_end_to_end.json"""
import importlib.util
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "longcat-video-tta_amd"


def main(out_json):
    import torch
    out_dir = Path(out_json).with_suffix("")
    path = PKG / "lora_experiment/scripts/run_lora_tta.py"
    spec = importlib.util.spec_from_file_location("runner_lora", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    # 13 conditioning frames -> 4 latent frames; `--num-frames` is the clip's TOTAL length (run_lora_tta.py:1209-1213): 49 frames -> 13
    # latent frames at 720p (46 800 tokens, 9 of the 13 frames
    # denoised against the cached conditioning K / V); TTA window 32 frames, 13 of context -> split 4 / 3 / 1: the 25 200-token
    # training sequence bench.py times, with one held-out latent frame for the early stopper
    argv = ["--checkpoint-dir", "synthetic", "--data-dir", "synthetic:1", "--output-dir", str(out_dir), "--resolution", "720p",
            "--num-cond-frames", "13", "--num-frames", "49", "--gen-start-frame", "32", "--tta-total-frames", "32",
            "--tta-context-frames", "13", "--num-steps", "20", "--lora-rank", "8", "--lora-alpha", "16", "--num-inference-steps", "50",
            "--guidance-scale", "4.0", "--no-save-videos", "--max-videos", "1"]
    t0 = time.time()
    m.main(argv)
    torch.cuda.synchronize()
    wall = time.time() - t0
    s = json.loads((out_dir / "summary.json").read_text())
    r = s["results"][0]
    rec = {"what": "one TTA video through lora_experiment/scripts/run_lora_tta.py at 720p: LoRA r=8 qkv+proj x 48 blocks, 20 inner steps, "
                   "early stopper on, 50-step CFG continuation to 49 frames (13 conditioning + 36 generated: 4 cached + 9 denoised latent frames), decode, PSNR / SSIM",
           "argv": argv, "wall_s_of_main_including_model_build": round(wall, 2),
           "per_video": {k: r.get(k) for k in ("train_time", "es_check_time", "gen_time", "total_time", "final_loss", "psnr", "ssim", "success")},
           "early_stopping_info": {k: (r.get("early_stopping_info") or {}).get(k) for k in ("stopped_early", "best_step", "num_checks")},
           "summary": {k: s.get(k) for k in ("avg_train_time", "avg_es_check_time", "avg_gen_time", "avg_total_time", "num_successful")},
           "device": torch.cuda.get_device_name(0)}
    Path(out_json).write_text(json.dumps(rec, indent=1))
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04_tta_video_k3_end_to_end.json")
