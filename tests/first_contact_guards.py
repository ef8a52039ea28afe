#!/usr/bin/env python3
"""First-contact guards: one function per row A1-A20 of `spec/dit.md` (TEST INFRASTRUCTURE - uses `oracle/`).

The DiT, VAE, scheduler, pipeline and `LoRAModule` this repository replaces live in `meituan-longcat/LongCat-Video`, which is
absent offline; `spec/dit.md` §A lists what the build ASSUMES about them.  The day a `LongCat-Video/` checkout is visible -
the directory the reference puts on `sys.path` at delta_experiment/scripts/common.py:27-39 - this script turns those twenty
assumptions into twenty PASS / FAIL lines with the measured difference:

    python tools/first_contact_guards.py --upstream /path/to/LongCat-Video [--checkpoint /path/to/weights] [--device cuda]

How it works.  Upstream's package and this repository's drop-in share the name `longcat_video`, so the upstream side runs in a
CHILD process (`--probe`) with the checkout first on `sys.path`: it builds upstream's classes at a toy size (hidden 256, 2 heads
of 128, 2 blocks; random weights of `oracle.dit_oracle.make_params`, loaded by name), evaluates a fixed list of probes and
writes the results to a file.  The parent compares them with the ORACLE (`oracle/dit_oracle.py`, `pipeline_oracle.py`,
`vae_oracle.py`: the CPU restatement every parity test of this repository is anchored on) and, with `--device cuda` on an
MI355X, with the PRODUCT (`longcat-video-tta_amd/`) as well.  A probe that cannot run (a constructor argument that differs, a
module that needs flash-attn on the CPU, a guard that needs real weights) is reported INCONCLUSIVE with the exception text -
never silently skipped.

Until upstream is visible, `tests/upstream_standin/` (an "upstream" that behaves exactly as the spec assumes, built on the oracle)
stands in: `tests/test_first_contact_guards.py` runs every guard against it (all PASS) and against bent copies
(`STANDIN_BREAK`: the matching guard must FAIL), so the script is known to execute and to have teeth.
"""
import argparse
import inspect
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
TOY = dict(in_channels=16, out_channels=16, hidden_size=256, depth=2, num_heads=2, caption_channels=64, mlp_ratio=4,
           adaln_tembed_dim=64, frequency_embedding_size=256, patch_size=(1, 2, 2), text_tokens_zero_pad=False)
GRID = (3, 4, 5)          # latent grid of the probes (T, h/2, w/2): three different extents so a swapped axis shows


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# =============================================================================================== shared inputs
def toy_inputs():
    g = torch.Generator().manual_seed(1234)
    T, H, W = GRID[0], GRID[1] * 2, GRID[2] * 2
    return dict(
        hs=torch.randn(1, 16, T, H, W, generator=g).to(torch.bfloat16),
        hs_noise_perturbed=None,
        y=torch.randn(1, 1, 12, TOY["caption_channels"], generator=g).to(torch.bfloat16),
        y2=torch.randn(1, 1, 12, TOY["caption_channels"], generator=g).to(torch.bfloat16),
        mask=torch.tensor([[1] * 9 + [0] * 3]),
        ts=torch.tensor([[0.0, 371.5, 371.5]]).to(torch.bfloat16),
        x_tok=torch.randn(1, GRID[0] * GRID[1] * GRID[2], TOY["hidden_size"], generator=g).to(torch.bfloat16),
        t_emb=torch.randn(1, GRID[0], TOY["adaln_tembed_dim"], generator=g),
        y_tok=torch.randn(1, 9, TOY["hidden_size"], generator=g).to(torch.bfloat16),
        rms_x=torch.randn(2, 7, 2, 128, generator=g).to(torch.bfloat16),
        rms_w=(1 + 0.1 * torch.randn(128, generator=g)).to(torch.bfloat16),
        cfg_c=torch.randn(1, 16, 2, 4, 4, generator=g), cfg_u=torch.randn(1, 16, 2, 4, 4, generator=g),
        z=torch.randn(1, 16, 3, 4, 4, generator=g).to(torch.bfloat16),
    )


def toy_state():
    from oracle import dit_oracle as O
    cfg = O.small_config(hidden_size=TOY["hidden_size"], depth=TOY["depth"], num_heads=TOY["num_heads"],
                         caption_channels=TOY["caption_channels"])
    P = O.make_params(cfg, seed=21, std=0.05)
    g = torch.Generator().manual_seed(5)
    for k in P:                       # norm weights away from 1 so that their placement matters
        if k.endswith("norm.weight"):
            P[k] = (1 + 0.1 * torch.randn(P[k].shape, generator=g)).to(torch.bfloat16)
        if k.endswith("pre_crs_attn_norm.bias"):
            P[k] = (0.1 * torch.randn(P[k].shape, generator=g)).to(torch.bfloat16)
    return cfg, P


# =============================================================================================== the upstream side (child)
def probe_upstream(upstream: str, checkpoint: str, out_file: str):
    """Runs with `upstream` first on sys.path.  Every probe is independent; a failure is recorded, not raised."""
    sys.path.insert(0, upstream)
    sys.path.insert(1, str(ROOT))
    res, err = {}, {}

    def attempt(name, fn):
        try:
            res[name] = fn()
        except Exception as ex:                                            # noqa: BLE001 - the text goes into the report
            err[name] = f"{type(ex).__name__}: {ex}"

    import importlib
    mods = {}
    for key, path in (("dit", "longcat_video.modules.longcat_video_dit"), ("sched", "longcat_video.modules.scheduling_flow_match_euler_discrete"),
                      ("vae", "longcat_video.modules.autoencoder_kl_wan"), ("pipe", "longcat_video.pipeline_longcat_video"),
                      ("lora", "longcat_video.modules.lora_utils")):
        try:
            mods[key] = importlib.import_module(path)
        except Exception as ex:                                            # noqa: BLE001
            err["import." + key] = f"{type(ex).__name__}: {ex}"
    inp = toy_inputs()
    cfg, P = toy_state()
    dit = None
    if "dit" in mods:
        def build():
            kw = dict(TOY)
            try:
                m = mods["dit"].LongCatVideoTransformer3DModel(enable_flashattn2=False, cp_split_hw=[1, 1], **kw)
            except TypeError:
                m = mods["dit"].LongCatVideoTransformer3DModel(**kw)
            own = m.state_dict()
            res["A20.keys_upstream"] = sorted(own.keys())
            res["A9.ffn_shapes"] = [list(m.blocks[0].ffn.w1.weight.shape), list(m.blocks[0].ffn.w2.weight.shape), list(m.blocks[0].ffn.w3.weight.shape)]
            # tensors whose shape differs from the build's layout cannot be loaded: they keep upstream's own initialisation (the
            # guards that need equal weights then FAIL, which is the right answer) and are listed
            fit = {k: v for k, v in P.items() if k in own and tuple(own[k].shape) == tuple(v.shape)}
            res["A20.shape_mismatch"] = sorted(k for k in P if k in own and k not in fit)
            m.load_state_dict(fit, strict=False)
            return m.eval()
        try:
            dit = build()
        except Exception as ex:                                            # noqa: BLE001
            err["dit.build"] = f"{type(ex).__name__}: {ex}"
    if dit is not None:
        blk = dit.blocks[0]
        attempt("A4.eps", lambda: [float(blk.mod_norm_attn.eps), float(blk.mod_norm_ffn.eps), float(blk.pre_crs_attn_norm.eps),
                                   float(dit.final_layer.norm_final.eps)])
        attempt("A3.rmsnorm", lambda: _rms(blk.attn.q_norm, inp))
        attempt("A3.eps", lambda: float(blk.attn.q_norm.eps))
        attempt("A5.t_embedder", lambda: dit.float().t_embedder(torch.tensor([0.0, 371.5, 999.0]), dtype=torch.float32).float())
        attempt("A6.repr", lambda: repr(dit.y_embedder.y_proj))
        attempt("A6.y_embedder", lambda: dit.float().y_embedder(inp["y"].float()).float())
        shape = GRID
        S = GRID[1] * GRID[2]

        def block_run(ncl, y_tok=None, x_tok=None):
            d32 = dit.float()
            x = (inp["x_tok"] if x_tok is None else x_tok).float()
            y = (inp["y_tok"] if y_tok is None else y_tok).float()
            with torch.no_grad():
                return d32.blocks[0](x, y, inp["t_emb"], [9], shape, num_cond_latents=ncl).float()
        attempt("A1.block_fp32_ncl0", lambda: block_run(0))
        attempt("A7.block_fp32_ncl1", lambda: block_run(1))

        def cond_rows_vs_noise():
            x2 = inp["x_tok"].clone(); x2[:, S:] += 1.0                    # perturb every NOISE token
            a, b = block_run(1), block_run(1, x_tok=x2)
            return float((a[:, :S] - b[:, :S]).abs().max())
        attempt("A7.cond_rows_moved_by_noise", cond_rows_vs_noise)

        def cond_rows_vs_prompt():
            a, b = block_run(1), block_run(1, y_tok=inp["y_tok"] * 3.0 + 1.0)
            return [float((a[:, :S] - b[:, :S]).abs().max()), float((a[:, S:] - b[:, S:]).abs().max())]
        attempt("A8.cond_rows_moved_by_prompt", cond_rows_vs_prompt)

        def model_run(dtype, ncl):
            m = dit.to(dtype)
            with torch.no_grad():
                return m(hidden_states=inp["hs"].to(dtype), timestep=inp["ts"].to(dtype), encoder_hidden_states=inp["y"].to(dtype),
                         encoder_attention_mask=inp["mask"], num_cond_latents=ncl).float()
        attempt("A10.model_fp32_ncl1", lambda: model_run(torch.float32, 1))
        attempt("A10.model_bf16_ncl1", lambda: model_run(torch.bfloat16, 1))
        attempt("A15.model_out_dtype", lambda: str(dit.to(torch.bfloat16)(hidden_states=inp["hs"], timestep=inp["ts"], encoder_hidden_states=inp["y"],
                                                                         encoder_attention_mask=inp["mask"], num_cond_latents=1).dtype))

        def kv_equiv():
            m = dit.float()
            hs = inp["hs"].float()
            with torch.no_grad():
                pinned = m(hidden_states=hs, timestep=inp["ts"].float(), encoder_hidden_states=inp["y"].float(),
                           encoder_attention_mask=inp["mask"], num_cond_latents=1)
                _, kv = m(hidden_states=hs[:, :, :1], timestep=inp["ts"][:, :1].float() * 0, encoder_hidden_states=inp["y"].float()[:, :, :4] * 0,
                          return_kv=True, skip_crs_attn=True)
                cached = m(hidden_states=hs[:, :, 1:], timestep=inp["ts"][:, 1:].float(), encoder_hidden_states=inp["y"].float(),
                           encoder_attention_mask=inp["mask"], num_cond_latents=1, kv_cache_dict=kv)
            return rel_l2(cached.float(), pinned[:, :, 1:].float())
        attempt("A2.cached_vs_pinned", kv_equiv)
    if "sched" in mods:
        def sched():
            S_ = mods["sched"].FlowMatchEulerDiscreteScheduler
            out = {}
            for shift in (1.0, 3.0):
                s = S_(num_train_timesteps=1000, shift=shift)
                s.set_timesteps(8, sigmas=torch.linspace(1, 0.001, 8).numpy())
                out[str(shift)] = [s.sigmas.float().cpu(), s.timesteps.float().cpu()]
            return out
        attempt("A14.scheduler", sched)
    if "pipe" in mods:
        attempt("A16.retrieve_latents_default", lambda: str(inspect.signature(mods["pipe"].retrieve_latents).parameters["sample_mode"].default))
        attempt("A11.pipeline_source_negates", lambda: _grep(mods["pipe"], ("noise_pred = -noise_pred", "noise_pred = -", "= -noise_pred")))
        attempt("A12.pipeline_source_zero_star", lambda: _grep(mods["pipe"], ("optimized_scale", "st_star", "zero_star", "zero-star")))

        def grid():
            P_ = mods["pipe"].LongCatVideoPipeline
            try:
                p = P_(tokenizer=None, text_encoder=None, vae=None, scheduler=None, dit=None)
            except Exception:                                              # noqa: BLE001
                p = P_.__new__(P_)
            return p.get_timesteps_sigmas(50).float().cpu()
        attempt("A13.sigma_grid", grid)

        def arithmetic():                                                  # only a pipeline that exposes the two pieces (the stand-in does)
            P_ = mods["pipe"].LongCatVideoPipeline
            p = P_.__new__(P_)
            v = p.combine_cfg(inp["cfg_c"], inp["cfg_u"], 4.0)
            return [v, p.euler(inp["cfg_c"], v, -0.02)]
        attempt("A12.step_arithmetic", arithmetic)
    if "lora" in mods:
        def lora():
            import torch.nn as nn
            m = mods["lora"].LoRAModule("x", nn.Linear(8, 24), 1.0, 4, 8.0, n_seperate=3)
            ups = m.lora_up.blocks if hasattr(m.lora_up, "blocks") else [m.lora_up]
            return dict(attrs=sorted(a for a in ("lora_down", "lora_up", "multiplier", "alpha_scale", "use_lora") if hasattr(m, a)),
                        down=list(m.lora_down.weight.shape), ups=[list(u.weight.shape) for u in ups], alpha_scale=float(m.alpha_scale),
                        up_is_zero=all(float(u.weight.abs().max()) == 0.0 for u in ups))
        attempt("A19.lora_module", lora)
    if "vae" in mods:
        def vae():
            from oracle import vae_oracle as V
            cfg_v = V.default_config(base_dim=32)
            m = mods["vae"].AutoencoderKLWan(base_dim=32, z_dim=16)
            Pv = dict(V.make_params(cfg_v, seed=3)); Pv.update(V.make_encoder_params(cfg_v, seed=4))
            m.load_state_dict({k: v.float() for k, v in Pv.items()}, strict=False)
            with torch.no_grad():
                out = m.decode(inp["z"].float(), return_dict=False)[0].float()
            return dict(keys=sorted(m.state_dict().keys()), decode=out)
        attempt("A18.vae", vae)
    if checkpoint:
        def real_cfg():
            with open(os.path.join(checkpoint, "dit", "config.json")) as f:
                return json.load(f)
        attempt("A9.real_dit_config", real_cfg)

        def real_sched():
            with open(os.path.join(checkpoint, "scheduler", "scheduler_config.json")) as f:
                return json.load(f)
        attempt("A14.real_scheduler_config", real_sched)

        def real_keys():
            from safetensors import safe_open
            keys = []
            d = os.path.join(checkpoint, "dit")
            for fn in sorted(os.listdir(d)):
                if fn.endswith(".safetensors"):
                    with safe_open(os.path.join(d, fn), "pt") as f:
                        keys += list(f.keys())
            return sorted(keys)
        attempt("A20.real_checkpoint_keys", real_keys)
    torch.save({"res": res, "err": err}, out_file)


def _rms(mod, inp):
    import copy
    mod = copy.deepcopy(mod)                      # the probe's weight must not leak into the model the later probes run
    with torch.no_grad():
        mod.weight.data = inp["rms_w"].to(mod.weight.dtype)
        return mod(inp["rms_x"]).float()


def _grep(module, needles):
    src = inspect.getsource(module)
    return [n for n in needles if n in src]


# =============================================================================================== the judging side (parent)
class Report:
    def __init__(self):
        self.rows = []

    def add(self, gid, status, text):
        self.rows.append((gid, status, text))
        print(f"{gid:4s} {status:12s} {text}", flush=True)


def _need(up, name):
    if name in up["res"]:
        return up["res"][name]
    raise LookupError(up["err"].get(name) or next((v for k, v in up["err"].items() if k.startswith(("import.", "dit.build"))), "probe did not run"))


def judge(up, device="cpu"):
    """One function per spec row; each returns (status, text).  `up` = {"res": ..., "err": ...} from the child."""
    from oracle import dit_oracle as O
    from oracle import pipeline_oracle as PO
    inp = toy_inputs()
    cfg, P = toy_state()
    P32 = {k: v.float() for k, v in P.items()}
    S = GRID[1] * GRID[2]
    rep = Report()

    def oracle_block(ncl, orc=O):
        return orc.block_forward(P32, "blocks.0.", inp["x_tok"].float(), inp["y_tok"].float(), inp["t_emb"], [9], GRID, ncl, TOY["num_heads"])

    def run(gid, fn):
        try:
            status, text = fn()
        except LookupError as ex:
            status, text = "INCONCLUSIVE", f"upstream probe failed: {ex}"
        except Exception as ex:                                            # noqa: BLE001
            status, text = "ERROR", f"{type(ex).__name__}: {ex}"
        rep.add(gid, status, text)

    def a1():
        got = _need(up, "A1.block_fp32_ncl0")
        e = rel_l2(got, oracle_block(0))
        # which alternative would upstream match if the assumed split is wrong?  (diagnostic only)
        alts = {}
        for name, dims in (("64|32|32", (64, 32, 32)), ("32|48|48", (32, 48, 48))):
            alts[name] = rel_l2(got, oracle_block(0, _oracle_with_rope_split(dims)))
        alts["half-split rotation"] = rel_l2(got, oracle_block(0, _oracle_with_half_rotation()))
        txt = f"block (no cond), grid {GRID}: upstream vs oracle [44|42|42 interleaved] rel-L2 {e:.1e}; alternatives " + \
            ", ".join(f"{k} {v:.1e}" for k, v in alts.items())
        return ("PASS" if e < 1e-4 else "FAIL"), txt

    def a2():
        e = _need(up, "A2.cached_vs_pinned")
        return ("PASS" if e < 1e-4 else "FAIL"), f"upstream KV-cached step vs conditioning frames pinned in the sequence: rel-L2 {e:.1e} (equal => RoPE positions continue after the cached frames)"

    def a3():
        got = _need(up, "A3.rmsnorm")
        ref = O.rmsnorm_fp32(inp["rms_x"], inp["rms_w"], rnd=O.bf16_round)
        eq = bool(torch.equal(got, ref))
        alt = O.bf16_round(O.rmsnorm_fp32(inp["rms_x"], inp["rms_w"]))          # one rounding after the weight multiply
        eps = up["res"].get("A3.eps")
        return ("PASS" if eq else "FAIL"), f"RMSNorm_FP32 on bf16 input: bit-equal to normalise -> bf16 -> * weight -> bf16: {eq} " \
                                           f"(max |d| {float((got - ref).abs().max()):.1e}; vs single-rounding form {float((got - alt).abs().max()):.1e}); eps {eps}"

    def a4():
        eps = _need(up, "A4.eps")
        ok = all(abs(e - 1e-6) < 1e-12 for e in eps)
        return ("PASS" if ok else "FAIL"), f"LayerNorm eps (mod_norm_attn, mod_norm_ffn, pre_crs_attn_norm, norm_final) = {eps}; assumed 1e-6"

    def a5():
        got = _need(up, "A5.t_embedder")
        ref = O.t_embedder(P32, torch.tensor([0.0, 371.5, 999.0]))
        e = rel_l2(got, ref)
        return ("PASS" if e < 1e-5 else "FAIL"), f"t_embedder([0, 371.5, 999]) fp32: rel-L2 {e:.1e} (a sin|cos swap gives ~1)"

    def a6():
        got = _need(up, "A6.y_embedder")
        e = rel_l2(got, O.y_embedder(P32, inp["y"].float()))
        return ("PASS" if e < 1e-5 else "FAIL"), f"y_embedder fp32: rel-L2 {e:.1e}; modules: {' '.join(str(up['res'].get('A6.repr', '?')).split())[:160]}"

    def a7():
        got = _need(up, "A7.block_fp32_ncl1")
        e = rel_l2(got, oracle_block(1))
        moved = _need(up, "A7.cond_rows_moved_by_noise")
        ok = e < 1e-4 and moved == 0.0
        return ("PASS" if ok else "FAIL"), f"block with 1 conditioning frame: rel-L2 {e:.1e}; conditioning rows moved by a noise-token perturbation: {moved:.1e} (must be 0)"

    def a8():
        moved = _need(up, "A8.cond_rows_moved_by_prompt")
        ok = moved[0] == 0.0 and moved[1] > 0.0
        return ("PASS" if ok else "FAIL"), f"prompt changed: conditioning rows moved {moved[0]:.1e} (must be 0), noise rows moved {moved[1]:.1e} (must be > 0)"

    def a9():
        shapes = _need(up, "A9.ffn_shapes")
        Fh = O.ffn_hidden_dim(TOY["hidden_size"])
        ok = shapes == [[Fh, TOY["hidden_size"]], [TOY["hidden_size"], Fh], [Fh, TOY["hidden_size"]]]
        txt = f"toy FFN shapes {shapes}; assumed hidden = 256 * ceil(floor(2 * 4C / 3) / 256) = {Fh}"
        real = up["res"].get("A9.real_dit_config")
        if real is not None:
            txt += f"; real dit/config.json: hidden_size {real.get('hidden_size')}, mlp_ratio {real.get('mlp_ratio')} -> {O.ffn_hidden_dim(real.get('hidden_size', 4096), real.get('mlp_ratio', 4))} (the build uses 11008)"
        return ("PASS" if ok else "FAIL"), txt

    def a10():
        got32 = _need(up, "A10.model_fp32_ncl1")
        ref32 = O.dit_forward(P32, cfg, inp["hs"], inp["ts"], inp["y"], inp["mask"], 1, bf16=False)
        e32 = rel_l2(got32, ref32)
        txt = f"whole model (1 conditioning frame) fp32: rel-L2 {e32:.1e}"
        ok = e32 < 1e-4
        if "A10.model_bf16_ncl1" in up["res"]:
            refb = O.dit_forward(P, cfg, inp["hs"], inp["ts"], inp["y"], inp["mask"], 1, bf16=True)
            eb = rel_l2(up["res"]["A10.model_bf16_ncl1"], refb)
            txt += f"; bf16 at the oracle's rounding points: {eb:.1e}"
            ok = ok and eb < 1e-2
        if device == "cuda":
            txt += "; product on the MI355X vs upstream bf16: " + _product_model_vs(up, P, inp)
        return ("PASS" if ok else "FAIL"), txt

    def a11():
        hits = up["res"].get("A11.pipeline_source_negates")
        arith = up["res"].get("A12.step_arithmetic")
        if arith is not None:
            ref = PO.euler_update(inp["cfg_c"], arith[0], -0.02, negate=True)
            e = rel_l2(arith[1], ref)
            return ("PASS" if e < 1e-6 else "FAIL"), f"Euler step on a fixed prediction vs x + dt * (-v): rel-L2 {e:.1e}; source markers {hits}"
        if hits is None:
            raise LookupError(up["err"].get("A11.pipeline_source_negates", "pipeline module not importable"))
        return ("PASS" if hits else "FAIL"), f"pipeline source contains {hits or 'no negation of noise_pred'} (assumed: `noise_pred = -noise_pred` before scheduler.step)"

    def a12():
        arith = up["res"].get("A12.step_arithmetic")
        hits = up["res"].get("A12.pipeline_source_zero_star")
        if arith is not None:
            e = rel_l2(arith[0], PO.cfg_zero_star(inp["cfg_c"], inp["cfg_u"], 4.0))
            plain = rel_l2(arith[0], inp["cfg_u"] + 4.0 * (inp["cfg_c"] - inp["cfg_u"]))
            return ("PASS" if e < 1e-6 else "FAIL"), f"CFG combination vs zero-star: rel-L2 {e:.1e} (vs plain CFG {plain:.1e}); source markers {hits}"
        if hits is None:
            raise LookupError(up["err"].get("A12.pipeline_source_zero_star", "pipeline module not importable"))
        return ("PASS" if hits else "FAIL"), f"pipeline source markers of CFG-zero-star: {hits or 'none'}"

    def a13():
        got = _need(up, "A13.sigma_grid")
        ref = torch.linspace(1, 0.001, 50, dtype=torch.float32)
        ok = got.shape == ref.shape and bool(torch.allclose(got, ref, atol=1e-7, rtol=0))
        return ("PASS" if ok else "FAIL"), f"get_timesteps_sigmas(50): first {float(got[0]):.4f}, last {float(got[-1]):.4f}, n {len(got)}; assumed linspace(1, 0.001, 50)"

    def a14():
        got = _need(up, "A14.scheduler")
        worst = 0.0
        for shift in (1.0, 3.0):
            ts, sig = PO.sigma_grid(8, shift)
            worst = max(worst, float((got[str(shift)][0] - sig).abs().max()), float((got[str(shift)][1] - ts).abs().max()) / 1000.0)
        txt = f"scheduler.set_timesteps(sigmas=grid) at shift 1 and 3: max |d sigma| {worst:.1e}"
        if "A14.real_scheduler_config" in up["res"]:
            txt += f"; real scheduler_config.json: {up['res']['A14.real_scheduler_config']}"
        return ("PASS" if worst < 1e-6 else "FAIL"), txt

    def a15():
        dt = _need(up, "A15.model_out_dtype")
        return "PASS" if dt == "torch.float32" else "FAIL", f"a bf16 model called with a bf16 timestep returns {dt} (assumed: bf16 timestep accepted, fp32 prediction returned)"

    def a16():
        d = _need(up, "A16.retrieve_latents_default")
        return ("PASS" if d == "sample" else "FAIL"), f"retrieve_latents(sample_mode=...) default = {d!r}; assumed 'sample'"

    def a17():
        e = _need(up, "A2.cached_vs_pinned")
        return ("PASS" if e < 1e-4 else "FAIL"), f"the (return_kv, skip_crs_attn, kv_cache_dict) protocol exists upstream and equals the pinned sequence: rel-L2 {e:.1e}"

    def a18():
        from oracle import vae_oracle as V
        got = _need(up, "A18.vae")
        cfg_v = V.default_config(base_dim=32)
        Pv = dict(V.make_params(cfg_v, seed=3)); Pv.update(V.make_encoder_params(cfg_v, seed=4))
        ref = V.decode_full({k: v.float() for k, v in Pv.items()}, cfg_v, inp["z"].float())
        e = rel_l2(got["decode"], ref)
        keys_ok = set(got["keys"]) == set(Pv)
        return ("PASS" if e < 1e-3 and keys_ok else "FAIL"), f"decode of [1,16,3,4,4] (base 32): rel-L2 {e:.1e}, output {tuple(got['decode'].shape)} (expected [1,3,9,32,32]); state-dict keys equal: {keys_ok}"

    def a19():
        got = _need(up, "A19.lora_module")
        ok = (got["attrs"] == ["alpha_scale", "lora_down", "lora_up", "multiplier", "use_lora"] and got["down"] == [12, 8]
              and got["ups"] == [[8, 4]] * 3 and abs(got["alpha_scale"] - 2.0) < 1e-12 and got["up_is_zero"])
        return ("PASS" if ok else "FAIL"), f"LoRAModule('x', Linear(8, 24), 1.0, 4, 8.0, n_seperate=3): {got}"

    def a20():
        keys = _need(up, "A20.keys_upstream")
        mine = set(P)
        extra, missing = sorted(set(keys) - mine), sorted(mine - set(keys))
        bad_shape = up["res"].get("A20.shape_mismatch", [])
        txt = f"toy state-dict: {len(keys)} upstream keys, {len(extra)} not in the build's layout {extra[:4]}, {len(missing)} of the build's missing upstream {missing[:4]}, {len(bad_shape)} with another shape {bad_shape[:4]}"
        ok = not missing and not bad_shape and not [k for k in extra if not k.endswith(("num_batches_tracked",))]
        real = up["res"].get("A20.real_checkpoint_keys")
        if real is not None:
            sys.path.insert(0, str(ROOT / "longcat-video-tta_amd"))
            from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
            full = set(LongCatVideoTransformer3DModel(device="meta").state_dict().keys())
            txt += f"; real checkpoint: {len(real)} tensors, {len(set(real) - full)} unknown to the drop-in, {len(full - set(real))} of the drop-in's absent"
            ok = ok and set(real) >= full
        return ("PASS" if ok else "FAIL"), txt

    for gid, fn in (("A1", a1), ("A2", a2), ("A3", a3), ("A4", a4), ("A5", a5), ("A6", a6), ("A7", a7), ("A8", a8), ("A9", a9),
                    ("A10", a10), ("A11", a11), ("A12", a12), ("A13", a13), ("A14", a14), ("A15", a15), ("A16", a16), ("A17", a17),
                    ("A18", a18), ("A19", a19), ("A20", a20)):
        run(gid, fn)
    if up["err"]:
        print("\nupstream probes that raised:", flush=True)
        for k, v in sorted(up["err"].items()):
            print(f"  {k}: {v}", flush=True)
    return rep


def _oracle_with_rope_split(dims):
    import types
    from oracle import dit_oracle as O
    M = types.ModuleType("alt_oracle"); M.__dict__.update(O.__dict__)

    def rope_angles_3d(grid, head_dim=128, base=10000.0, device=None):
        T, H, W = grid

        def axis(n, dim):
            freqs = 1.0 / (base ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
            return torch.outer(torch.arange(n, dtype=torch.float32), freqs).repeat_interleave(2, dim=-1)
        ft, fh, fw = axis(T, dims[0]), axis(H, dims[1]), axis(W, dims[2])
        ang = torch.cat([ft[:, None, None, :].expand(T, H, W, dims[0]), fh[None, :, None, :].expand(T, H, W, dims[1]),
                         fw[None, None, :, :].expand(T, H, W, dims[2])], dim=-1)
        return ang.reshape(T * H * W, head_dim).to(device)
    M.rope_angles_3d = rope_angles_3d
    return _rebind(M, O)


def _oracle_with_half_rotation():
    import types
    from oracle import dit_oracle as O
    M = types.ModuleType("alt_oracle"); M.__dict__.update(O.__dict__)

    def rotate_half(x):
        x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
        return torch.cat((-x2, x1), dim=-1)
    M.rotate_half = rotate_half
    return _rebind(M, O)


def _rebind(M, O):
    import types
    for name in ("apply_rope", "self_attention", "block_forward"):
        fn = O.__dict__[name]
        new = types.FunctionType(fn.__code__, M.__dict__, fn.__name__, fn.__defaults__, fn.__closure__)
        new.__kwdefaults__ = fn.__kwdefaults__
        M.__dict__[name] = new
    return M


def _product_model_vs(up, P, inp) -> str:
    """`--device cuda`: the drop-in on the MI355X against upstream's own bf16 output (and thereby A1 / A3 / A5-A10 end to end)."""
    if "A10.model_bf16_ncl1" not in up["res"]:
        return "upstream bf16 run unavailable"
    sys.path.insert(0, str(ROOT / "longcat-video-tta_amd"))
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    kw = {k: v for k, v in TOY.items()}
    m = LongCatVideoTransformer3DModel(device="cuda", dtype=torch.bfloat16, **kw)
    m.load_state_dict(P, strict=False)
    with torch.no_grad():
        out = m.eval()(hidden_states=inp["hs"].cuda(), timestep=inp["ts"].cuda(), encoder_hidden_states=inp["y"].cuda(),
                       encoder_attention_mask=inp["mask"].cuda(), num_cond_latents=1).float().cpu()
    return f"rel-L2 {rel_l2(out, up['res']['A10.model_bf16_ncl1']):.1e}"


def run_guards(upstream: str, checkpoint: str = "", device: str = "cpu", env=None):
    """Probe `upstream` in a child process, judge here.  Returns the Report."""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "upstream_probe.pt")
        cmd = [sys.executable, str(Path(__file__).resolve()), "--probe", "--upstream", upstream, "--out", out]
        if checkpoint:
            cmd += ["--checkpoint", checkpoint]
        child_env = dict(os.environ)
        child_env.update(env or {})
        child_env["PYTHONPATH"] = upstream + os.pathsep + str(ROOT) + os.pathsep + child_env.get("PYTHONPATH", "")
        r = subprocess.run(cmd, env=child_env, capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(out):
            raise RuntimeError(f"the upstream probe process failed (rc {r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")
        up = torch.load(out, weights_only=False)
    sys.path.insert(0, str(ROOT))
    return judge(up, device)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--upstream", required=True, help="the LongCat-Video checkout (the directory that contains `longcat_video/`)")
    ap.add_argument("--checkpoint", default="", help="weights directory with dit/ vae/ scheduler/ (optional: A9, A14, A20 read it)")
    ap.add_argument("--device", default="cpu", choices=["cpu", "cuda"], help="cuda: also run the product on the MI355X (A10)")
    ap.add_argument("--probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--out", default="", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    if args.probe:
        probe_upstream(args.upstream, args.checkpoint, args.out)
        return 0
    rep = run_guards(args.upstream, args.checkpoint, args.device)
    n = {s: sum(1 for _, st, _ in rep.rows if st == s) for s in ("PASS", "FAIL", "INCONCLUSIVE", "ERROR")}
    print(f"\n{n['PASS']} PASS, {n['FAIL']} FAIL, {n['INCONCLUSIVE']} INCONCLUSIVE, {n['ERROR']} ERROR of {len(rep.rows)} guards")
    return 0 if n["FAIL"] == 0 and n["ERROR"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
