"""GPU: the LoRA-TTA runner end to end (synthetic 2-block model, synthetic latents): split -> adapter reset -> ES setup ->
inner loop -> KV-cached CFG denoise, with the reference's artifact schemas and resume semantics."""
import importlib.util
import json

import torch  # noqa: F401
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
RUNNER = ROOT / "longcat-video-tta_amd" / "lora_experiment" / "scripts" / "run_lora_tta.py"


def test_runner_end_to_end(tmp_path):
    spec = importlib.util.spec_from_file_location("run_lora_tta_amd", RUNNER)
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    out = tmp_path / "run"
    argv = ["--checkpoint-dir", "synthetic:2:256:64", "--data-dir", "synthetic:3", "--output-dir", str(out),
            "--num-cond-frames", "5", "--num-frames", "13", "--gen-start-frame", "40", "--tta-total-frames", "33",
            "--tta-context-frames", "9", "--num-steps", "6", "--es-check-every", "2", "--es-patience", "1",
            "--num-inference-steps", "3", "--lora-rank", "4", "--lora-alpha", "8", "--save-lora-weights"]
    m.main(argv)
    cfg = json.loads((out / "config.json").read_text())
    assert cfg["method"] == "lora_tta_custom" and cfg["lora"]["rank"] == 4 and cfg["lora"]["num_modules"] == 10
    assert set(cfg) >= {"lora", "training", "generation", "seed", "max_videos", "clip_gate_enabled"}
    ck = json.loads((out / "checkpoint.json").read_text())
    assert ck["next_idx"] == 3 and len(ck["results"]) == 3
    s = json.loads((out / "summary.json").read_text())
    assert s["method"] == "lora_tta" and s["num_successful"] == 3 and s["num_failed"] == 0
    for r in s["results"]:
        assert r["success"] and {"idx", "video_name", "train_time", "gen_time", "total_time", "final_loss",
                                 "num_train_steps", "early_stopping_info", "es_check_time"} <= set(r)
        assert r["early_stopping_info"]["total_checks"] >= 2 and 1 <= r["num_train_steps"] <= 6
        # on-device evaluation of the decoded continuation: the keys export_all_results.py:168-171 reads
        assert 0 < r["psnr"] < 50 and -1 <= r["ssim"] <= 1 and r["lpips"] is None and r["output_path"].endswith("_lora.npy")
    assert s["psnr"] == pytest.approx(sum(r["psnr"] for r in s["results"]) / 3, abs=1e-5) and s["lpips"] is None
    assert len(list((out / "lora_weights").glob("*_lora.pt"))) == 3
    # resume: nothing left to do, results preserved
    m.main(argv)
    assert len(json.loads((out / "summary.json").read_text())["results"]) == 3


def test_runner_with_augmentation_variants(tmp_path):
    """`--aug-enabled` (run_lora_tta.py:1100-1126): the TTA clip's pixel variants - flip, a fixed rotation pair,
    one random rotation, a slowed copy - are each encoded by the HIP VAE encoder, cut to the training window, and the inner
    loop draws one per step; a `speed_2x` copy is too short for the window and is skipped (tta/augment.py)."""
    spec = importlib.util.spec_from_file_location("run_lora_tta_amd_aug", RUNNER)
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    out = tmp_path / "run_aug"
    argv = ["--checkpoint-dir", "synthetic:2:256:64", "--data-dir", "synthetic:2", "--output-dir", str(out),
            "--num-cond-frames", "5", "--num-frames", "13", "--tta-total-frames", "33", "--tta-context-frames", "9",
            "--num-steps", "4", "--num-inference-steps", "2", "--lora-rank", "4", "--lora-alpha", "8", "--skip-generation",
            "--aug-enabled", "--aug-flip", "--aug-rotate-deg", "8", "--aug-rotate-random-count", "1", "--aug-speed-factors", "0.5,2.0"]
    m.main(argv)
    s = json.loads((out / "summary.json").read_text())
    assert s["num_successful"] == 2
    for r in s["results"]:
        names = r["aug_variants"]
        assert names[:4] == ["orig", "flip_h", "rotate_-8.0", "rotate_+8.0"] and names[4].startswith("rotate_rand_")
        assert names[5:] == ["slow_2x"] and r["num_train_steps"] == 4 and r["final_loss"] == r["final_loss"]


def test_runner_builtin_lora_path(tmp_path):
    """`--use-builtin-lora`: upstream-native LoRAModule adapters behind patched forwards, same loop and artifacts
    (config.json method `lora_tta_builtin`, run_lora_tta.py:856)."""
    spec = importlib.util.spec_from_file_location("run_lora_tta_amd", RUNNER)
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    out = tmp_path / "builtin"
    m.main(["--checkpoint-dir", "synthetic:2:256:64", "--data-dir", "synthetic:2", "--output-dir", str(out), "--num-cond-frames", "5",
            "--num-frames", "13", "--gen-start-frame", "40", "--tta-total-frames", "33", "--tta-context-frames", "9", "--num-steps", "4",
            "--es-check-every", "2", "--es-patience", "1", "--num-inference-steps", "2", "--lora-rank", "4", "--lora-alpha", "8",
            "--use-builtin-lora", "--save-lora-weights", "--no-save-videos"])
    cfg = json.loads((out / "config.json").read_text())
    assert cfg["method"] == "lora_tta_builtin" and cfg["lora"]["implementation"] == "builtin" and cfg["lora"]["num_modules"] == 10
    # rank-4 adapters with n_seperate = 3 / 2 on the fused qkv / kv projections: more down-projection rows than the custom path
    assert cfg["lora"]["trainable_params"] == 2 * ((3 * 4 * 256 + 768 * 4) + 3 * (4 * 256 + 256 * 4) + (2 * 4 * 256 + 512 * 4))
    s = json.loads((out / "summary.json").read_text())
    assert s["num_successful"] == 2 and all(r["final_loss"] > 0 and r["psnr"] > 0 for r in s["results"])
    w = torch.load(next((out / "lora_weights").glob("*_lora.pt")))
    assert w["lora_0.down"].shape == (12, 256) and w["lora_0.up"].shape == (768, 4)


def _run(rel, argv):
    path = ROOT / "longcat-video-tta_amd" / rel
    spec = importlib.util.spec_from_file_location("runner_" + path.stem, path)
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    m.main(argv)


@pytest.mark.parametrize("rel,method,extra,key", [
    ("delta_experiment/scripts/run_delta_a.py", "delta_a", ["--delta-steps", "4", "--delta-lr", "1e-2"], "delta_norm"),
    ("delta_experiment/scripts/run_delta_b.py", "delta_b", ["--delta-steps", "4", "--delta-lr", "1e-2", "--num-groups", "2"], "delta_norms"),
    ("delta_experiment/scripts/run_delta_c.py", "delta_c", ["--delta-steps", "4", "--delta-lr", "1e-2"], "delta_out_norm"),
    ("delta_experiment/scripts/run_film_tta.py", "film_adapter", ["--film-steps", "4", "--film-lr", "1e-2", "--num-groups", "2",
                                                                  "--film-mode", "shift_scale"], "correction_norm"),
    ("delta_experiment/scripts/run_norm_tune_tta.py", "norm_tune", ["--norm-steps", "4", "--norm-lr", "1e-2", "--norm-target",
                                                                    "all_norm"], "norm_param_drift"),
    # norm weights (bf16) + a delta-A vector (fp32) in one optimizer under one clip (run_norm_tune_tta.py:380-391)
    ("delta_experiment/scripts/run_norm_tune_tta.py", "norm_tune", ["--norm-steps", "4", "--norm-lr", "1e-2", "--norm-target",
                                                                    "qk_norm", "--also-tune-delta"], "delta_norm"),
])
def test_delta_runners_end_to_end(tmp_path, rel, method, extra, key):
    """delta wrapper -> anchored ES -> optimise -> hooks installed for the KV-cached continuation -> reference schemas."""
    out = tmp_path / method
    argv = ["--checkpoint-dir", "synthetic:2:256:64", "--data-dir", "synthetic:2", "--output-dir", str(out),
            "--num-cond-frames", "5", "--num-frames", "13", "--gen-start-frame", "40", "--tta-total-frames", "33",
            "--tta-context-frames", "9", "--es-check-every", "2", "--es-patience", "1", "--num-inference-steps", "2"] + extra
    _run(rel, argv)
    s = json.loads((out / "summary.json").read_text())
    assert s["method"] == method and s["num_videos"] == 2 and s["num_successful"] == 2 and not (out / "config.json").exists()
    assert {"avg_train_time", "avg_es_check_time", "avg_gen_time", "avg_total_time"} <= set(s)
    assert ("clip_gate_enabled" in s) == (method not in ("film_adapter", "norm_tune"))
    assert s["psnr"] is not None and s["ssim"] is not None
    for r in s["results"]:
        assert r["success"] and key in r and r["gen_time"] > 0 and r["final_loss"] is not None and r["psnr"] > 0
        assert r["early_stopping_info"]["total_checks"] >= 1
    norms = [sum(r[key]) if isinstance(r[key], list) else r[key] for r in s["results"]]
    assert all(n > 0 for n in norms)                      # the delta moved: gradients reach it through the frozen DiT
    ck = json.loads((out / "checkpoint.json").read_text())
    assert ck["next_idx"] == 2 and len(ck["results"]) == 2


def test_baseline_runner_end_to_end(tmp_path):
    out = tmp_path / "base"
    _run("baseline_experiment/scripts/run_baseline.py",
         ["--checkpoint-dir", "synthetic:2:256:64", "--data-dir", "synthetic:2", "--output-dir", str(out), "--num-cond-frames", "5",
          "--num-gen-frames", "8", "--num-inference-steps", "2"])
    s = json.loads((out / "summary.json").read_text())
    # the shape export_all_results.py:132-166 recognises as the no-TTA baseline
    assert "metrics" in s and "results" not in s and s["num_successful"] == 2 and s["num_frames_total"] == 13
    assert s["timing"]["per_video_inference_s"]["mean"] is not None
    assert set(s["metrics"]["psnr"]) == {"mean", "std", "min", "max"} and s["metrics"]["ssim"]["mean"] is not None
    assert s["metrics"]["lpips"] == {}
    assert (out / "per_video_metrics.csv").read_text().splitlines()[0].startswith("index,filename,caption,psnr")


def test_no_tta_control_num_steps_zero(tmp_path):
    """The reference's no-TTA control is the TTA runner with `num_steps: 0` (sweep_experiment/configs/ucf101_no_tta.yaml:8-15):
    the adapters stay at their zero-effect initialisation, the run succeeds, final_loss is null."""
    spec = importlib.util.spec_from_file_location("run_lora_tta_amd", RUNNER)
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    out = tmp_path / "ctl"
    m.main(["--checkpoint-dir", "synthetic:2:256:64", "--data-dir", "synthetic:1", "--output-dir", str(out), "--num-cond-frames", "5",
            "--num-frames", "13", "--gen-start-frame", "40", "--tta-total-frames", "33", "--tta-context-frames", "9", "--num-steps", "0",
            "--es-disable", "--num-inference-steps", "2"])
    s = json.loads((out / "summary.json").read_text())
    r = s["results"][0]
    assert s["num_successful"] == 1 and r["num_train_steps"] == 0 and r["final_loss"] is None and r["gen_time"] > 0


@pytest.mark.parametrize("optimizer", ["sgd", "adamw"])
def test_full_tta_runner_end_to_end(tmp_path, optimizer):
    """Every DiT parameter trainable -> checkpointed backward -> fused clip + SGD / AdamW -> early stopping on the stopper's
    own snapshot -> continuation -> scoring; reference schemas (run_full_tta.py:466-510, 864-905)."""
    out = tmp_path / f"full_{optimizer}"
    _run("lora_experiment/scripts/run_full_tta.py",
         ["--checkpoint-dir", "synthetic:2:256:64", "--data-dir", "synthetic:2", "--output-dir", str(out), "--num-cond-frames", "5",
          "--num-frames", "13", "--gen-start-frame", "40", "--tta-total-frames", "33", "--tta-context-frames", "9", "--num-steps", "4",
          "--learning-rate", "1e-4", "--optimizer", optimizer, "--es-check-every", "2", "--es-patience", "1",
          "--num-inference-steps", "2", "--no-save-videos"])
    cfg = json.loads((out / "config.json").read_text())
    assert cfg["method"] == "full_tta" and cfg["training"]["trainable_params"] == cfg["training"]["total_params"] > 1e6
    assert cfg["training"]["optimizer"] == optimizer and cfg["clip_gate"]["enabled"] is False
    s = json.loads((out / "summary.json").read_text())
    assert s["method"] == "full_tta" and s["num_successful"] == 2 and s["num_failed"] == 0 and s["total_params"] == cfg["training"]["total_params"]
    assert s["avg_final_loss"] > 0 and s["psnr"] is not None
    for r in s["results"]:
        assert r["success"] and 1 <= r["num_train_steps"] <= 4 and r["final_loss"] > 0 and r["gen_time"] > 0 and r["psnr"] > 0
        assert r["early_stopping_info"]["total_checks"] >= 1
    assert json.loads((out / "checkpoint.json").read_text())["next_idx"] == 2


def test_full_tta_training_moves_the_weights_and_reset_restores_them():
    import torch
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    from tta.full_tta import finetune_full_on_conditioning, reset_dit_weights, snapshot_base_state
    dit = LongCatVideoTransformer3DModel(device="cuda", dtype=torch.bfloat16, depth=2, hidden_size=256, num_heads=2,
                                         caption_channels=64).init_synthetic_(5)
    for p in dit.parameters():
        p.requires_grad = True
    base = snapshot_base_state(dit)
    g = torch.Generator(device="cuda").manual_seed(0)
    lat = torch.randn((1, 16, 4, 16, 16), generator=g, device="cuda").to(torch.bfloat16)
    pe = torch.randn((1, 1, 32, 64), generator=g, device="cuda").to(torch.bfloat16)
    pm = torch.ones((1, 32), dtype=torch.int64, device="cuda")
    tr = finetune_full_on_conditioning(dit, lat[:, :, :2], lat[:, :, 2:], pe, pm, num_steps=3, lr=1e-2, warmup_steps=1,
                                       max_grad_norm=1.0, optimizer_type="sgd")
    assert len(tr["losses"]) == 3 and all(l == l for l in tr["losses"])
    moved = [n for n, p in dit.named_parameters() if not torch.equal(p.detach(), base[n])]
    # every family of parameters got an update (norm weights sit at 1.0, where a clipped bf16 step is below half an ulp)
    assert len(moved) >= 0.75 * len(list(dit.named_parameters())), len(moved)
    for fam in ("x_embedder.proj.weight", "t_embedder.mlp.0.weight", "y_embedder.y_proj.2.weight", "blocks.0.adaLN_modulation.1.weight",
                "blocks.1.attn.qkv.weight", "blocks.0.attn.proj.bias", "blocks.1.cross_attn.kv_linear.weight", "blocks.0.ffn.w2.weight",
                "final_layer.linear.weight", "final_layer.adaLN_modulation.1.bias"):
        assert fam in moved, fam
    assert all(p.grad is None for p in dit.parameters())                          # gradients released before the continuation
    reset_dit_weights(dit, base)
    assert all(torch.equal(p.detach(), base[n]) for n, p in dit.named_parameters())
