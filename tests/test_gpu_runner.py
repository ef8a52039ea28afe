"""GPU: the LoRA-TTA runner end to end (synthetic 2-block model, synthetic latents): split -> adapter reset -> ES setup ->
inner loop -> KV-cached CFG denoise, with the reference's artifact schemas and resume semantics."""
import importlib.util
import json
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
    assert len(list((out / "lora_weights").glob("*_lora.pt"))) == 3
    # resume: nothing left to do, results preserved
    m.main(argv)
    assert len(json.loads((out / "summary.json").read_text())["results"]) == 3
