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


# --------------------------------------------------------------------------- the other runners at the reference's paths
_PKG = ROOT / "longcat-video-tta_amd"
_COMMON = ["--checkpoint-dir", "/ckpt", "--data-dir", "/data", "--output-dir", "/out", "--max-videos", "100",
           "--num-cond-frames", "14", "--num-frames", "28", "--gen-start-frame", "32", "--tta-total-frames", "32",
           "--tta-context-frames", "14", "--num-inference-steps", "50", "--guidance-scale", "4.0", "--resolution", "480p",
           "--seed", "42", "--no-save-videos", "--caption-guard-mode", "warn", "--feature-frame-guard-mode", "fail",
           "--clip-gate-threshold", "0.0", "--min-fvd-videos", "256", "--es-check-every", "5", "--es-patience", "3"]


def _load_script(rel):
    path = _PKG / rel
    assert path.is_file(), f"runner missing at the reference's relative path: {rel}"
    spec = importlib.util.spec_from_file_location("runner_" + path.stem, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("rel,extra,check", [
    ("delta_experiment/scripts/run_delta_a.py",
     ["--delta-steps", "20", "--delta-lr", "1e-3", "--batch-videos", "1", "--batch-method", "similarity"],
     lambda a: a.delta_steps == 20 and a.delta_lr == 1e-3 and a.retrieval_pool_dir is None),
    ("delta_experiment/scripts/run_delta_b.py",
     ["--delta-steps", "20", "--delta-lr", "1e-3", "--num-groups", "4", "--delta-target", "hidden", "--delta-target-blocks",
      "last_4", "--delta-dim", "64"],
     lambda a: a.num_groups == 4 and a.delta_target == "hidden" and a.delta_dim == 64 and a.delta_target_blocks == "last_4"),
    ("delta_experiment/scripts/run_delta_c.py", ["--delta-steps", "10", "--delta-lr", "1e-2", "--delta-mode", "per_channel"],
     lambda a: a.delta_mode == "per_channel" and a.delta_steps == 10),
])
def test_delta_runners_parse_the_sbatch_flag_sets(rel, extra, check):
    """sweep_experiment/sbatch/run_sweep.sbatch:440-560 (METHOD=delta_a|delta_b|delta_c)."""
    m = _load_script(rel)
    args = m.build_parser().parse_args(_COMMON + extra)
    assert check(args) and args.num_cond_frames == 14 and args.es_check_every == 5 and args.clip_gate_enabled is False


def test_film_runner_parses_its_flags_and_has_no_clip_gate_group():
    """run_film_tta.py:348-373 adds no CLIP-gate group and run_sweep.sbatch:586-591 passes none."""
    m = _load_script("delta_experiment/scripts/run_film_tta.py")
    common = [a for a in _COMMON if not a.startswith("--clip-gate")]
    i = _COMMON.index("--clip-gate-threshold")
    common = _COMMON[:i] + _COMMON[i + 2:]
    a = m.build_parser().parse_args(common + ["--film-steps", "20", "--film-lr", "1e-3", "--num-groups", "4", "--film-mode",
                                              "shift_scale"])
    assert a.film_mode == "shift_scale" and a.num_groups == 4 and not hasattr(a, "clip_gate_enabled")
    with pytest.raises(SystemExit):
        m.build_parser().parse_args(common + ["--clip-gate-enabled"])


def test_norm_tune_runner_parses_its_flags():
    """run_norm_tune_tta.py:292-319 (no CLIP-gate group either)."""
    m = _load_script("delta_experiment/scripts/run_norm_tune_tta.py")
    i = _COMMON.index("--clip-gate-threshold")
    common = _COMMON[:i] + _COMMON[i + 2:]
    a = m.build_parser().parse_args(common + ["--norm-steps", "20", "--norm-lr", "1e-3", "--norm-target", "qk_norm"])
    assert a.norm_target == "qk_norm" and a.norm_steps == 20 and a.also_tune_delta is False and not hasattr(a, "clip_gate_enabled")


def test_baseline_runner_parses_the_reference_flags():
    """baseline_experiment/scripts/run_baseline.py:235-262."""
    m = _load_script("baseline_experiment/scripts/run_baseline.py")
    a = m.build_parser().parse_args(["--checkpoint-dir", "c", "--data-dir", "d", "--output-dir", "o", "--num-cond-frames", "2",
                                     "--num-gen-frames", "14", "--gen-start-frame", "32", "--resolution", "720p",
                                     "--num-inference-steps", "50", "--guidance-scale", "4.0", "--seed", "42", "--max-videos",
                                     "100", "--save-videos"])
    assert a.num_gen_frames == 14 and a.resolution == "720p" and a.save_videos


def test_full_tta_runner_parses_the_sbatch_flag_set():
    """METHOD=full: sweep_experiment/sbatch/run_sweep.sbatch:344-372 -> lora_experiment/scripts/run_full_tta.py."""
    path = ROOT / "longcat-video-tta_amd" / "lora_experiment" / "scripts" / "run_full_tta.py"
    spec = importlib.util.spec_from_file_location("run_full_tta_amd", path)
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    argv = ["--checkpoint-dir", "/ckpt", "--data-dir", "/data", "--output-dir", "/out", "--max-videos", "100",
            "--learning-rate", "1e-5", "--num-steps", "10", "--warmup-steps", "2", "--weight-decay", "0.01", "--max-grad-norm", "1.0",
            "--optimizer", "sgd", "--batch-videos", "1", "--num-cond-frames", "14", "--num-frames", "28", "--gen-start-frame", "32",
            "--tta-total-frames", "32", "--tta-context-frames", "14", "--num-inference-steps", "50", "--guidance-scale", "4.0",
            "--resolution", "480p", "--seed", "42", "--es-check-every", "5", "--es-patience", "3", "--es-anchor-sigmas", "0.25,0.5,0.75",
            "--es-noise-draws", "2", "--es-strategy", "patience", "--es-holdout-fraction", "0.25", "--clip-gate-threshold", "0.0",
            "--no-save-videos", "--restart"]
    args = m.build_parser().parse_args(argv)
    assert args.optimizer == "sgd" and args.learning_rate == 1e-5 and args.num_steps == 10 and args.warmup_steps == 2
    d = m.build_parser().parse_args(["--checkpoint-dir", "c", "--data-dir", "d", "--output-dir", "o"])
    assert (d.learning_rate, d.num_steps, d.warmup_steps, d.weight_decay, d.max_grad_norm, d.optimizer) == (1e-5, 10, 2, 0.01, 1.0, "sgd")
    with pytest.raises(SystemExit):
        m.build_parser().parse_args(argv + ["--optimizer", "lion"])


def test_nvidia_smi_shim_keeps_the_sbatch_template_alive():
    """run_sweep.sbatch:24, 254: `nvidia-smi --query-gpu=name,memory.free --format=csv` runs under `set -euo pipefail`; on an
    AMD node the stand-in under tools/shims must exit 0 and print a CSV header plus one row per GPU."""
    import os
    import subprocess
    env = dict(os.environ, PATH=str(ROOT / "tools" / "shims") + os.pathsep + os.environ.get("PATH", ""))
    r = subprocess.run(["bash", "-c", "set -euo pipefail; nvidia-smi --query-gpu=name,memory.free --format=csv"], env=env,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "name, memory.free [MiB]" and len(lines) >= 2 and lines[1].endswith("MiB")
