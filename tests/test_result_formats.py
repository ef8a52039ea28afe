"""CPU: the on-disk result formats as the reference's analysis scripts read them.

tests/golden/result_tree/ is a PROJECT_ROOT-shaped tree written by THIS build's runners on an MI355X
(tools/make_result_tree.py: every runner on the synthetic 2-block model, 3 videos).  In the build container the
reference's own sweep_experiment/scripts/export_all_results.py and export_loss_curves.py were pointed at it
(PROJECT_ROOT=<tree>): 9 complete runs collected, 8 matched to the cond5_gen8 baseline with dPSNR/dSSIM, loss curves
extracted for every run with early stopping — see DESIGN.md §2.  Those scripts cannot travel, so this test restates the
reads they perform (export_all_results.py:104-268, export_loss_curves.py:79-151) and applies them to the same tree."""
import json
import math
import statistics
from pathlib import Path

import pytest

TREE = Path(__file__).resolve().parent / "golden" / "result_tree"
SERIES = TREE / "sweep_experiment" / "results" / "series_amd_plumbing"
TTA_RUNS = {"L1": "lora_tta", "L0_no_tta": "lora_tta", "F_full1": "full_tta", "DA1": "delta_a", "DB1": "delta_b", "DC1": "delta_c",
            "F1": "film_adapter", "N1": "norm_tune"}


def _extract_run(run_dir: Path) -> dict:
    """The reads of export_all_results.py:extract_run, restated."""
    s = json.loads((run_dir / "summary.json").read_text())
    rec = {"series": run_dir.parent.name, "run_id": run_dir.name, "method": s.get("method", "")}
    is_baseline = "metrics" in s and "results" not in s
    results = s.get("results", [])
    ok = [r for r in results if r.get("success", False)]
    if is_baseline:
        rec["n_ok"], rec["n_total"] = s.get("num_successful", 0), s.get("num_videos", 0)
        for key in ("psnr", "ssim", "lpips"):
            m = s.get("metrics", {})
            if key in m and m[key]:
                rec[f"{key}_mean"], rec[f"{key}_std"] = m[key].get("mean"), m[key].get("std")
        t = s.get("timing", {}).get("per_video_inference_s", {})
        rec["gen_time_mean"] = t.get("mean")
    else:
        rec["n_ok"], rec["n_total"] = len(ok), s.get("num_videos", len(results))
        for key in ("psnr", "ssim", "lpips"):
            vals = [r[key] for r in ok if key in r and r[key] is not None and not math.isnan(r[key])]
            rec[f"{key}_mean"] = statistics.mean(vals) if vals else None
        rec["train_time_mean"] = statistics.mean([r["train_time"] for r in ok if "train_time" in r])
        rec["gen_time_mean"] = statistics.mean([r["gen_time"] for r in ok if "gen_time" in r])
        rec["total_time_mean"] = statistics.mean([r.get("total_time", r.get("train_time", 0) + r.get("gen_time", 0)) for r in ok])
        losses = [r["final_loss"] for r in ok if r.get("final_loss") is not None and not math.isnan(r["final_loss"])]
        rec["final_loss_mean"] = statistics.mean(losses) if losses else None
        es = [r.get("early_stopping_info") for r in ok if r.get("early_stopping_info")]
        if es:
            rec["es_stopped_count"] = len([e for e in es if e.get("stopped_early", False)])
            rec["es_best_step_mean"] = statistics.mean([e["best_step"] for e in es if "best_step" in e])
    for k in ("delta_steps", "delta_lr", "num_groups", "learning_rate", "num_steps", "lora_rank", "lora_alpha", "norm_target",
              "norm_steps", "film_mode", "film_steps", "num_cond_frames", "num_frames", "gen_start_frame", "clip_gate_enabled"):
        if k in s and s[k] is not None:
            rec[k] = s[k]
    cfgp = run_dir / "config.json"
    if cfgp.exists():
        for k, v in json.loads(cfgp.read_text()).items():
            for k2, v2 in (v.items() if isinstance(v, dict) else [(k, v)]):
                if k2 not in rec or rec[k2] is None:
                    rec[k2] = v2
    cs = s.get("clip_gate_stats")
    if isinstance(cs, dict):
        rec["clip_skip_rate"] = cs.get("skip_rate")
    return rec


@pytest.mark.parametrize("run,method", sorted(TTA_RUNS.items()))
def test_tta_run_is_readable_by_the_exporter(run, method):
    rec = _extract_run(SERIES / run)
    assert rec["method"] == method and rec["n_ok"] == rec["n_total"] == 3
    assert rec["psnr_mean"] > 0 and -1 <= rec["ssim_mean"] <= 1 and rec["lpips_mean"] is None   # no LPIPS network offline
    assert rec["gen_time_mean"] > 0 and rec["total_time_mean"] >= rec["gen_time_mean"]
    assert rec["num_cond_frames"] == 5 and rec["num_frames"] == 13                              # -> cond5 / gen8 matching key
    if run == "L0_no_tta":
        assert rec["final_loss_mean"] is None and rec["num_steps"] == 0
    else:
        assert rec["final_loss_mean"] > 0 and rec["es_best_step_mean"] >= 0
    if method == "lora_tta":                          # config.json is flattened into the record (:231-245)
        assert rec["rank"] in (4, 8) and rec["trainable_params"] > 0 and rec["implementation"] == "custom"
    if method == "full_tta":
        assert rec["trainable_params"] == rec["total_params"] > 1e6 and rec["optimizer"] == "sgd" and rec["num_steps"] == 4
    if method in ("film_adapter", "norm_tune"):
        assert "clip_gate_enabled" not in rec         # those two runners do not add the CLIP-gate group
    else:
        assert rec["clip_gate_enabled"] is False and rec["clip_skip_rate"] == 0.0
    ck = json.loads((SERIES / run / "checkpoint.json").read_text())
    assert ck["next_idx"] == 3                        # the in-progress view (:110-120) reads next_idx


def test_baseline_run_is_recognised_and_matched():
    d = TREE / "baseline_experiment" / "results" / "cond5_gen8"       # run_id parsed by `cond(\\d+)_gen(\\d+)` (:358-360)
    rec = _extract_run(d)
    assert rec["method"] == "" and rec["n_ok"] == 3 and rec["psnr_mean"] > 0 and rec["ssim_std"] is not None
    assert rec["gen_time_mean"] > 0 and rec["num_cond_frames"] == 5
    head = (d / "per_video_metrics.csv").read_text().splitlines()[0]
    assert head == "index,filename,caption,psnr,ssim,lpips,resolution,inference_time_s"


@pytest.mark.parametrize("run", ["L1", "F_full1", "DA1", "DB1", "DC1", "F1", "N1"])
def test_loss_curves_are_extractable(run):
    """export_loss_curves.py:79-151: loss_history [[step, loss], ...] per video, aggregated per step; it also takes the
    mean PSNR of the run (and raises when no video has one — the reason the runners score every generation)."""
    s = json.loads((SERIES / run / "summary.json").read_text())
    ok = [r for r in s["results"] if r.get("success", False)]
    curves = [r["early_stopping_info"]["loss_history"] for r in ok if r.get("early_stopping_info", {}).get("loss_history")]
    assert len(curves) == 3
    for lh in curves:
        steps = [p[0] for p in lh]
        assert steps == sorted(steps) and steps[0] == 0 and all(isinstance(p[1], float) and math.isfinite(p[1]) for p in lh)
    assert statistics.mean([r["psnr"] for r in ok if r.get("psnr") and not math.isnan(r["psnr"])]) > 0
    assert (s.get("delta_steps") or s.get("num_steps") or 0) >= 0


def test_generated_clip_goes_through_imageio_like_the_reference_when_it_is_installed(tmp_path, monkeypatch):
    """`save_frames` mirrors `save_video_from_numpy` (run_lora_tta.py:641-647): uint8 frames, fps 24, libx264, quality 9, `.mp4`;
    without imageio (this image) the same frames land in a `.npy` stack."""
    import sys
    import types
    import numpy as np
    import torch
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "longcat-video-tta_amd"))
    from tta import runner_common as R
    frames = torch.rand(3, 8, 8, 3)
    monkeypatch.setitem(sys.modules, "imageio", None)          # import imageio -> ImportError
    out = R.save_frames(None, None, str(tmp_path / "clip_lora"), frames=frames)
    assert out.endswith("clip_lora.npy")
    got = np.load(out)
    assert got.dtype == np.uint8 and got.shape == (3, 8, 8, 3) and np.array_equal(got, (frames * 255).to(torch.uint8).numpy())
    calls = []
    fake = types.ModuleType("imageio")
    fake.mimwrite = lambda path, fr, **kw: calls.append((path, fr.dtype, fr.shape, kw))
    monkeypatch.setitem(sys.modules, "imageio", fake)
    out = R.save_frames(None, None, str(tmp_path / "clip_lora"), frames=frames)
    assert out.endswith("clip_lora.mp4")
    assert calls == [(out, np.dtype("uint8"), (3, 8, 8, 3), {"fps": 24, "codec": "libx264", "quality": 9})]
