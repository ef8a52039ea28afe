"""Run every runner on the synthetic 2-block model into a PROJECT_ROOT-shaped tree (sweep_experiment/results/<series>/<run>,
baseline_experiment/results/<run>) so the reference's own analysis scripts (export_all_results.py, export_loss_curves.py)
can be pointed at it: SURVEY §8(f) row 2.  usage: python tools/make_result_tree.py <out_root>"""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "longcat-video-tta_amd"


def run(rel, argv):
    path = PKG / rel
    spec = importlib.util.spec_from_file_location("runner_" + path.stem, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.main(argv)


def main(out_root):
    out_root = Path(out_root)
    common = ["--checkpoint-dir", "synthetic:2:256:64", "--data-dir", "synthetic:3", "--num-cond-frames", "5", "--num-frames", "13",
              "--gen-start-frame", "40", "--tta-total-frames", "33", "--tta-context-frames", "9", "--es-check-every", "2",
              "--es-patience", "1", "--num-inference-steps", "2", "--no-save-videos"]
    sweep = out_root / "sweep_experiment" / "results" / "series_amd_plumbing"
    run("lora_experiment/scripts/run_lora_tta.py", common + ["--output-dir", str(sweep / "L1"), "--num-steps", "6", "--lora-rank", "4",
                                                             "--lora-alpha", "8"])
    run("lora_experiment/scripts/run_lora_tta.py", common + ["--output-dir", str(sweep / "L0_no_tta"), "--num-steps", "0", "--es-disable"])
    run("lora_experiment/scripts/run_full_tta.py", common + ["--output-dir", str(sweep / "F_full1"), "--num-steps", "4", "--learning-rate", "1e-4"])
    run("delta_experiment/scripts/run_delta_a.py", common + ["--output-dir", str(sweep / "DA1"), "--delta-steps", "4"])
    run("delta_experiment/scripts/run_delta_b.py", common + ["--output-dir", str(sweep / "DB1"), "--delta-steps", "4", "--num-groups", "2"])
    run("delta_experiment/scripts/run_delta_c.py", common + ["--output-dir", str(sweep / "DC1"), "--delta-steps", "4"])
    run("delta_experiment/scripts/run_film_tta.py", common + ["--output-dir", str(sweep / "F1"), "--film-steps", "4", "--num-groups", "2"])
    run("delta_experiment/scripts/run_norm_tune_tta.py", common + ["--output-dir", str(sweep / "N1"), "--norm-steps", "4"])
    run("baseline_experiment/scripts/run_baseline.py",
        ["--checkpoint-dir", "synthetic:2:256:64", "--data-dir", "synthetic:3", "--output-dir",
         str(out_root / "baseline_experiment" / "results" / "base_5c8g"), "--num-cond-frames", "5", "--num-gen-frames", "8",
         "--num-inference-steps", "2"])
    print("tree written under", out_root)


if __name__ == "__main__":
    main(sys.argv[1])
