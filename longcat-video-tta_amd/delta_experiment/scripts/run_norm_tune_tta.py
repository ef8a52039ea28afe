#!/usr/bin/env python3
"""Norm-tuning TTA (unfreeze the cross-attention pre-norm affine and / or the q/k RMS-norm weights) on MI355X — same relative
path, CLI flags and artifact schemas as the reference's `delta_experiment/scripts/run_norm_tune_tta.py` (flags :292-319 — no
CLIP-gate group; summary :631-655).  The tuned weights live in the DiT for the video's continuation and are restored before the
next video.  `--also-tune-delta` (:380-391) adds a delta-A vector (fp32) on the t_embedder output to the same optimizer: one fused
AdamW per dtype, tied by a joint clip coefficient (`FusedAdamWClip.joint_clip_grad_norm_`)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import _runner  # noqa: F401,E402

import torch  # noqa: E402

from tta import runner_common as R  # noqa: E402
from tta.delta import NormTuneForward, optimize_norm_params  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="Norm-tuning TTA for LongCat-Video (MI355X)")
    R.add_common_args(p)
    p.add_argument("--norm-steps", type=int, default=20)
    p.add_argument("--norm-lr", type=float, default=1e-3)
    p.add_argument("--norm-target", type=str, default="all_norm", choices=["cross_attn_norm", "qk_norm", "all_norm"])
    p.add_argument("--also-tune-delta", action="store_true")
    R.add_shared_groups(p, clip_gate=False)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    R.run_delta_method(
        args, "norm_tune",
        make_wrapper=lambda dit: NormTuneForward(dit, args.norm_target, also_tune_delta=args.also_tune_delta),
        optimize_fn=lambda w, cond, train, pe, pm, device, es, tv=None: optimize_norm_params(
            w, w.tuned_params, cond, train, pe, pm, num_steps=args.norm_steps, lr=args.norm_lr, device=device, dtype=torch.bfloat16,
            early_stopper=es, train_latents_variants=tv),
        params_of=lambda w: w.tuned_params,
        result_extra=lambda opt: {k: opt[k] for k in ("norm_param_drift", "delta_norm") if k in opt},
        summary_head={"norm_target": args.norm_target, "norm_steps": args.norm_steps, "norm_lr": args.norm_lr},
        file_suffix="norm_tune", cleanup=lambda w: w.restore())


if __name__ == "__main__":
    main()
