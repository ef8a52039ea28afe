#!/usr/bin/env python3
"""Delta-A TTA (one learned vector added to the timestep embedding) on MI355X — same relative path, CLI flags and
`checkpoint.json` / `summary.json` schemas as the reference's `delta_experiment/scripts/run_delta_a.py` (flags :372-403, result
keys :763-776, summary :903-935; it writes no config.json), so `run_sweep.sbatch:440-470` drives it unchanged.  Inputs and the
multi-GPU sharding are those of the LoRA runner (see its header): `latents/*.pt` or `synthetic:N`, one video per GPU."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import _runner  # noqa: F401,E402

import torch  # noqa: E402

from tta import runner_common as R  # noqa: E402
from tta.delta import DeltaAWrapper, optimize_delta_a  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="Delta-A TTA for LongCat-Video (MI355X)")
    R.add_common_args(p)
    p.add_argument("--delta-steps", type=int, default=20)
    p.add_argument("--delta-lr", type=float, default=1e-3)
    p.add_argument("--batch-videos", type=int, default=1)
    p.add_argument("--batch-method", type=str, default="similarity", choices=["similarity", "sequential"])
    p.add_argument("--retrieval-pool-dir", type=str, default=None)
    R.add_shared_groups(p)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    R.run_delta_method(
        args, "delta_a",
        make_wrapper=lambda dit: DeltaAWrapper(dit, adaln_tembed_dim=dit.config.adaln_tembed_dim),
        optimize_fn=lambda w, cond, train, pe, pm, device, es, tv=None: optimize_delta_a(
            w, cond, train, pe, pm, num_steps=args.delta_steps, lr=args.delta_lr, device=device, dtype=torch.bfloat16,
            early_stopper=es, train_latents_variants=tv),
        params_of=lambda w: [w.delta],
        result_extra=lambda opt: {"delta_norm": opt["delta_norm"]},
        summary_head={"delta_steps": args.delta_steps, "delta_lr": args.delta_lr, "batch_videos": args.batch_videos,
                      "retrieval_pool_dir": args.retrieval_pool_dir},
        file_suffix="delta_a")


if __name__ == "__main__":
    main()
