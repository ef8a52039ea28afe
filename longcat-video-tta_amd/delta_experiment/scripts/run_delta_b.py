#!/usr/bin/env python3
"""Delta-B TTA (per-block-group vectors on the timestep embedding or the hidden stream) on MI355X — same relative path, CLI
flags and artifact schemas as the reference's `delta_experiment/scripts/run_delta_b.py` (flags :452-492, summary :917-960)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import _runner  # noqa: F401,E402

import torch  # noqa: E402

from tta import runner_common as R  # noqa: E402
from tta.delta import DeltaBWrapper, optimize_delta_b  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="Delta-B TTA for LongCat-Video (MI355X)")
    R.add_common_args(p)
    p.add_argument("--delta-steps", type=int, default=20)
    p.add_argument("--delta-lr", type=float, default=1e-3)
    p.add_argument("--num-groups", type=int, default=4)
    p.add_argument("--delta-target", type=str, default="timestep", choices=["timestep", "hidden"])
    p.add_argument("--delta-dim", type=int, default=None)
    p.add_argument("--delta-target-blocks", type=str, default="all")
    R.add_shared_groups(p)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    R.run_delta_method(
        args, "delta_b",
        make_wrapper=lambda dit: DeltaBWrapper(dit, num_groups=args.num_groups, adaln_tembed_dim=dit.config.adaln_tembed_dim,
                                               hidden_size=dit.config.hidden_size, delta_target=args.delta_target,
                                               delta_dim=args.delta_dim, target_blocks=args.delta_target_blocks),
        optimize_fn=lambda w, cond, train, pe, pm, device, es, tv=None: optimize_delta_b(
            w, cond, train, pe, pm, num_steps=args.delta_steps, lr=args.delta_lr, device=device, dtype=torch.bfloat16,
            early_stopper=es, train_latents_variants=tv),
        params_of=lambda w: list(w.deltas) + ([w.delta_final] if w.delta_final is not None else []),
        result_extra=lambda opt: {"delta_norms": opt["delta_norms"]},
        summary_head={"delta_target": args.delta_target, "delta_target_blocks": args.delta_target_blocks,
                      "num_groups": args.num_groups, "delta_steps": args.delta_steps, "delta_lr": args.delta_lr},
        file_suffix="delta_b")


if __name__ == "__main__":
    main()
