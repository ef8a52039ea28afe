"""CPU: the runner at the reference's relative path accepts the flag set `run_sweep.sbatch` passes for METHOD=lora
(sweep_experiment/sbatch/run_sweep.sbatch:375-438) and applies the reference's post-parse normalisation."""
import importlib.util
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
RUNNER = ROOT / "longcat-video-tta_amd" / "lora_experiment" / "scripts" / "run_lora_tta.py"


def _load():
    spec = importlib.util.spec_from_file_location("run_lora_tta_amd", RUNNER)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_runner_parses_the_sbatch_flag_set():
    m = _load()
    argv = ["--checkpoint-dir", "/ckpt", "--data-dir", "/data", "--output-dir", "/out", "--max-videos", "100",
            "--seed", "42", "--lora-rank", "8", "--lora-alpha", "16.0", "--learning-rate", "2e-4", "--num-steps", "20",
            "--warmup-steps", "3", "--weight-decay", "0.01", "--max-grad-norm", "1.0", "--num-cond-frames", "14",
            "--num-frames", "28", "--gen-start-frame", "32", "--num-inference-steps", "50", "--guidance-scale", "4.0",
            "--resolution", "480p", "--target-modules", "qkv,proj", "--lora-target-blocks", "last_4",
            "--tta-total-frames", "48", "--tta-context-frames", "14", "--es-check-every", "5", "--es-patience", "3",
            "--es-anchor-sigmas", "0.25,0.5,0.75", "--es-noise-draws", "2", "--es-strategy", "patience",
            "--es-holdout-fraction", "0.25", "--caption-guard-mode", "warn", "--feature-frame-guard-mode", "fail",
            "--clip-gate-threshold", "0.0", "--clip-gate-sample-frames", "4", "--clip-gate-fail-closed",
            "--aug-rotate-deg", "10.0", "--min-fvd-videos", "256", "--no-save-videos", "--skip-generation",
            "--save-lora-weights", "--target-ffn", "--restart", "--batch-videos", "1", "--batch-method", "similarity"]
    args = m.build_parser().parse_args(argv)
    from tta import cli_args as C
    C.normalize_tta_frame_args(args)
    assert args.tta_total_frames == 32            # clamped to gen_start_frame (GT-leak guard)
    assert args.tta_context_frames == 14 and args.clip_gate_fail_open is False
    info = C.validate_tta_feature_budget(args, context="t")
    assert info["split_budget"] == {"total_latents": 8, "cond_latents": 4, "train_latents": 3, "val_latents": 1}
    # template default TTA_TOTAL_FRAMES = NUM_COND_FRAMES = 14 -> split 3/1/0 -> ES cannot run -> 'fail' mode raises
    args2 = m.build_parser().parse_args(["--checkpoint-dir", "c", "--data-dir", "d", "--output-dir", "o",
                                         "--num-cond-frames", "14"])
    C.normalize_tta_frame_args(args2)
    assert (args2.tta_total_frames, args2.tta_context_frames) == (14, 14)
    with pytest.raises(RuntimeError, match="val_latents=0"):
        C.validate_tta_feature_budget(args2, context="t")
    args2.es_disable = True
    C.validate_tta_feature_budget(args2, context="t")
    args2.clip_gate_enabled = True
    with pytest.raises(NotImplementedError):
        C.reject_out_of_scope(args2)
