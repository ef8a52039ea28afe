#!/usr/bin/env python3
"""Golden vectors for the UNCONDITIONED flow-matching losses (`compute_flow_matching_loss`,
`compute_flow_matching_loss_fixed`, delta_experiment/scripts/common.py:274-407 — defined and imported by every runner but
called by none), minted from the reference's own functions on the toy DiT of make_golden.py with the sigma / noise draws
injected.  Only DATA is written.  Re-run: python tests/golden/make_uncond_golden.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden as mg  # noqa: E402

OUT = Path(__file__).resolve().parent


def main():
    mg._stub_longcat()
    sys.path.insert(0, str(mg.REF / "delta_experiment" / "scripts"))
    import common as C
    torch.manual_seed(0)
    dit = mg.ToyDiT().eval()
    lat = torch.randn(1, 16, 4, 4, 6).to(torch.bfloat16)
    sig_u = torch.tensor([0.62109375])
    eps = torch.randn(1, 16, 4, 4, 6).to(torch.bfloat16)
    real_rand, real_randn_like = torch.rand, torch.randn_like
    torch.rand = lambda *a, **k: sig_u.clone()
    torch.randn_like = lambda t, **k: eps.clone()
    try:
        loss = C.compute_flow_matching_loss(dit, lat, None, None, device="cpu", dtype=torch.bfloat16)
    finally:
        torch.rand, torch.randn_like = real_rand, real_randn_like
    seen = dit.seen[-1]
    out = {"toy_dit_state": {k: v.clone() for k, v in dit.state_dict().items()},
           "uncond": dict(latents=lat, sig_u=sig_u, eps=eps, hidden_states=seen["hidden_states"], timestep=seen["timestep"],
                          num_cond_latents=seen["num_cond_latents"], loss=loss.detach())}
    n0 = len(dit.seen)
    lf = C.compute_flow_matching_loss_fixed(dit, lat, None, None, [0.25, 0.75], noise_draws=2, device="cpu", dtype=torch.bfloat16)
    calls = dit.seen[n0:]
    out["uncond_fixed"] = dict(latents=lat, sigmas=[0.25, 0.75], noise_draws=2, loss=float(lf),
                               hidden_states=torch.stack([c["hidden_states"] for c in calls]),
                               timesteps=torch.stack([c["timestep"] for c in calls]))
    torch.save(out, OUT / "uncond_loss.pt")
    print("loss", float(loss), "fixed", float(lf), "calls", len(calls))


if __name__ == "__main__":
    main()
