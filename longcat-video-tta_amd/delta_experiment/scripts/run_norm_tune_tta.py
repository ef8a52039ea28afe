#!/usr/bin/env python3
"""Norm-tuning TTA (unfreeze the cross-attention pre-norm affine and / or the q/k RMS-norm weights) on MI355X — same relative
path, CLI flags and artifact schemas as the reference's `delta_experiment/scripts/run_norm_tune_tta.py` (flags :292-319 — no
CLIP-gate group; summary :631-655).  The tuned weights live in the DiT for the video's continuation and are restored before the
next video.  `--also-tune-delta` (a delta-A vector in the same optimizer) is parsed but not built: the fused clip + AdamW takes
one dtype per parameter list (bf16 norms vs the fp32 delta)."""
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
    if args.also_tune_delta:
        raise NotImplementedError("--also-tune-delta mixes an fp32 delta with bf16 norm weights in one optimizer; run "
                                  "run_delta_a.py and run_norm_tune_tta.py separately")
    R.run_delta_method(
        args, "norm_tune",
        make_wrapper=lambda dit: NormTuneForward(dit, args.norm_target),
        optimize_fn=lambda w, cond, train, pe, pm, device, es: optimize_norm_params(
            w, cond, train, pe, pm, num_steps=args.norm_steps, lr=args.norm_lr, device=device, dtype=torch.bfloat16,
            early_stopper=es),
        params_of=lambda w: w.norm_params,
        result_extra=lambda opt: {"norm_param_drift": opt["norm_param_drift"]},
        summary_head={"norm_target": args.norm_target, "norm_steps": args.norm_steps, "norm_lr": args.norm_lr},
        file_suffix="norm_tune", cleanup=lambda w: w.restore())


if __name__ == "__main__":
    main()
