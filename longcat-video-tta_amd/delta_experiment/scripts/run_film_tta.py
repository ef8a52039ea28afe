#!/usr/bin/env python3
"""FiLM-adapter TTA (per-block-group additive corrections to the adaLN output) on MI355X — same relative path, CLI flags and
artifact schemas as the reference's `delta_experiment/scripts/run_film_tta.py` (flags :348-373 — no CLIP-gate group, as in the
reference; summary :676-700)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import _runner  # noqa: F401,E402

import torch  # noqa: E402

from tta import runner_common as R  # noqa: E402
from tta.delta import FiLMAdapterWrapper, optimize_film_adapter  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="FiLM adapter TTA for LongCat-Video (MI355X)")
    R.add_common_args(p)
    p.add_argument("--film-steps", type=int, default=20)
    p.add_argument("--film-lr", type=float, default=1e-3)
    p.add_argument("--num-groups", type=int, default=4)
    p.add_argument("--film-mode", type=str, default="full", choices=["full", "shift_scale", "scale_only"])
    R.add_shared_groups(p, clip_gate=False)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    R.run_delta_method(
        args, "film_adapter",
        make_wrapper=lambda dit: FiLMAdapterWrapper(dit, num_groups=args.num_groups, hidden_size=dit.config.hidden_size,
                                                    film_mode=args.film_mode),
        optimize_fn=lambda w, cond, train, pe, pm, device, es, tv=None: optimize_film_adapter(
            w, cond, train, pe, pm, num_steps=args.film_steps, lr=args.film_lr, device=device, dtype=torch.bfloat16,
            early_stopper=es, train_latents_variants=tv),
        params_of=lambda w: list(w.corrections),
        result_extra=lambda opt: {"correction_norm": opt["correction_norm"]},
        summary_head={"film_mode": args.film_mode, "num_groups": args.num_groups, "film_steps": args.film_steps,
                      "film_lr": args.film_lr},
        file_suffix="film")


if __name__ == "__main__":
    main()
