#!/usr/bin/env python3
"""Delta-C TTA (a per-channel output offset; no gradient flows into the DiT) on MI355X — same relative path, CLI flags and
artifact schemas as the reference's `delta_experiment/scripts/run_delta_c.py` (flags :253-285, summary :673-703)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import _runner  # noqa: F401,E402

import torch  # noqa: E402

from tta import runner_common as R  # noqa: E402
from tta.delta import DeltaCWrapper, optimize_delta_c  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="Delta-C TTA for LongCat-Video (MI355X)")
    R.add_common_args(p)
    p.add_argument("--delta-steps", type=int, default=20)
    p.add_argument("--delta-lr", type=float, default=1e-3)
    p.add_argument("--delta-mode", type=str, default="per_channel")
    R.add_shared_groups(p)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    R.run_delta_method(
        args, "delta_c",
        make_wrapper=lambda dit: DeltaCWrapper(dit, mode=args.delta_mode, out_channels=dit.config.out_channels),
        optimize_fn=lambda w, cond, train, pe, pm, device, es, tv=None: optimize_delta_c(
            w, cond, train, pe, pm, num_steps=args.delta_steps, lr=args.delta_lr, device=device, dtype=torch.bfloat16,
            early_stopper=es, train_latents_variants=tv),
        params_of=lambda w: [w.delta_out],
        result_extra=lambda opt: {"delta_out_norm": opt["delta_out_norm"]},
        summary_head={"delta_mode": args.delta_mode, "delta_steps": args.delta_steps, "delta_lr": args.delta_lr},
        file_suffix="delta_c")


if __name__ == "__main__":
    main()
