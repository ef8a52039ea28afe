"""argparse groups shared by the TTA runners — flag names and defaults of the reference
(delta_experiment/scripts/common.py:1404-1485, 1601-1706, 2438-2450; early_stopping.py:33-51), so
`sweep_experiment/sbatch/run_sweep.sbatch` can pass its flag set unchanged.  Flags of subsystems outside the hot path
(CLIP gate, caption guard, online FVD) are parsed and recorded; enabling one raises a clear error.  Augmentation is built
(tta/augment.py: the variants' pre-encode is row (f)1 of SURVEY §8)."""
import argparse
from typing import Any, Dict, List

from .latent_split import estimate_tta_split_budget


def add_tta_frame_args(parser):
    g = parser.add_argument_group("TTA frame split")
    g.add_argument("--tta-total-frames", type=int, default=None)
    g.add_argument("--tta-context-frames", type=int, default=None)
    return parser


def add_augmentation_args(parser):
    g = parser.add_argument_group("Augmentation")
    g.add_argument("--aug-enabled", action="store_true", default=False)
    g.add_argument("--aug-flip", action="store_true", default=False)
    g.add_argument("--aug-rotate-deg", type=float, default=10.0)
    g.add_argument("--aug-rotate-random-min", type=float, default=5.0)
    g.add_argument("--aug-rotate-random-max", type=float, default=15.0)
    g.add_argument("--aug-rotate-random-count", type=int, default=2)
    g.add_argument("--aug-rotate-random-step", type=float, default=1.0)
    g.add_argument("--no-aug-rotate-zoom", action="store_false", dest="aug_rotate_zoom")      # common.py:1697-1703
    parser.set_defaults(aug_rotate_zoom=True)
    g.add_argument("--aug-speed-factors", type=str, default="")
    return parser


def add_caption_guard_args(parser):
    g = parser.add_argument_group("Caption guard")
    g.add_argument("--caption-guard-mode", type=str, default="fail", choices=["fail", "warn", "off"])
    g.add_argument("--caption-guard-min-nonempty-ratio", type=float, default=0.95)
    g.add_argument("--caption-guard-min-unique-ratio", type=float, default=0.10)
    g.add_argument("--caption-guard-max-top1-ratio", type=float, default=0.50)
    g.add_argument("--caption-guard-max-generic-top1-ratio", type=float, default=0.20)
    g.add_argument("--caption-guard-topk", type=int, default=5)
    return parser


def add_caption_override_args(parser):
    parser.add_argument_group("Caption override").add_argument("--fixed-caption", type=str, default=None)
    return parser


def add_feature_frame_guard_args(parser):
    parser.add_argument_group("Feature frame guard").add_argument(
        "--feature-frame-guard-mode", type=str, default="fail", choices=["fail", "warn", "off"])
    return parser


def add_online_eval_args(parser):
    g = parser.add_argument_group("Online evaluation")
    g.add_argument("--compute-fvd", action="store_true", default=False)
    g.add_argument("--compute-fid", action="store_true", default=False)
    g.add_argument("--compute-vbench", action="store_true", default=False)
    g.add_argument("--min-fvd-videos", type=int, default=256)
    return parser


def add_clip_gate_args(parser):
    g = parser.add_argument_group("CLIP gate")
    g.add_argument("--clip-gate-enabled", action="store_true", default=False)
    g.add_argument("--clip-gate-threshold", type=float, default=0.0)
    g.add_argument("--clip-gate-backend", type=str, default="clip", choices=["clip", "xclip"])
    g.add_argument("--clip-gate-model", type=str, default=None)
    g.add_argument("--clip-gate-sample-frames", type=int, default=4)
    g.add_argument("--clip-gate-aggregation", type=str, default="mean", choices=["mean", "min", "max"])
    g.add_argument("--clip-gate-sampling-mode", type=str, default="full_window", choices=["full_window", "late_only"])
    g.add_argument("--clip-gate-late-fraction", type=float, default=0.4)
    g.add_argument("--clip-gate-late-only", action="store_true", default=False)
    g.add_argument("--clip-gate-fail-open", dest="clip_gate_fail_open", action="store_true", default=True)
    g.add_argument("--clip-gate-fail-closed", dest="clip_gate_fail_open", action="store_false")
    g.add_argument("--clip-gate-log-only", action="store_true", default=False)
    return parser


def normalize_tta_frame_args(args):
    """Post-parse normalisation of lora_experiment/scripts/run_lora_tta.py:743-758 (GT-leak clamp included)."""
    if args.tta_total_frames is None:
        args.tta_total_frames = args.num_cond_frames
    if args.tta_context_frames is None or args.tta_context_frames > args.tta_total_frames:
        args.tta_context_frames = args.num_cond_frames
    if args.tta_total_frames > args.gen_start_frame:
        print(f"[WARN] tta_total_frames ({args.tta_total_frames}) exceeds gen_start_frame ({args.gen_start_frame}); "
              "clamping to avoid GT leakage.")
        args.tta_total_frames = args.gen_start_frame
    if args.tta_context_frames > args.tta_total_frames:
        args.tta_context_frames = args.tta_total_frames
    return args


def validate_tta_feature_budget(args, context: str = "") -> Dict[str, Any]:
    """ES needs at least one held-out latent (common.py:1533-1598); CLIP-gate budget is not checked (gate out of scope)."""
    mode = str(getattr(args, "feature_frame_guard_mode", "fail")).lower()
    if mode not in {"fail", "warn", "off"}:
        mode = "fail"
    prefix = f"[feature_budget:{context}]" if context else "[feature_budget]"
    tta_total = int(getattr(args, "tta_total_frames", 0) or 0)
    tta_context = int(getattr(args, "tta_context_frames", 0) or 0)
    holdout = float(getattr(args, "es_holdout_fraction", 0.25) or 0.25)
    split = estimate_tta_split_budget(tta_total, tta_context, holdout_fraction=holdout)
    issues: List[str] = []
    if not bool(getattr(args, "es_disable", False)) and split["val_latents"] < 1:
        issues.append("ES is enabled but estimated val_latents=0 "
                      f"(tta_total_frames={tta_total}, tta_context_frames={tta_context}, holdout={holdout}). "
                      "Increase tta_total_frames and/or reduce tta_context_frames.")
    if mode != "off":
        print(f"{prefix} split(total={split['total_latents']}, cond={split['cond_latents']}, "
              f"train={split['train_latents']}, val={split['val_latents']})")
    if issues:
        msg = f"{prefix} " + " | ".join(issues)
        if mode == "warn":
            print(f"WARNING: {msg}")
        elif mode == "fail":
            raise RuntimeError(msg)
    return {"split_budget": split}


def reject_out_of_scope(args):
    """Subsystems outside the hot path are parsed for CLI compatibility but cannot be switched on here."""
    if getattr(args, "clip_gate_enabled", False):
        raise NotImplementedError("the CLIP gate is outside the denoise-and-adapt hot path (SURVEY §2 #15)")
    if any(getattr(args, k, False) for k in ("compute_fvd", "compute_fid", "compute_vbench")):
        raise NotImplementedError("online FVD/FID/VBench evaluation is outside the hot path (SURVEY §2 #17)")
