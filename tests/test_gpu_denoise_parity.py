"""GPU parity of the multi-step denoise loop (`LongCatVideoPipeline.denoise`, the arithmetic behind the reference's
`generate_video_continuation`, delta_experiment/scripts/common.py:566-611, and the inference-only baseline,
baseline_experiment/scripts/run_baseline.py:409-420) against `oracle/pipeline_oracle.py`, at BASELINE.json's configs:

  K1  16x256x256 -> latents [1,16,5,32,32] (1 280 tokens), 4 Euler steps, CFG 4, FULL width (4096 / 32 heads), depth 2:
      the HIP path vs the CPU oracle after every step, with no conditioning frames, with the conditioning-frame KV cache and
      with the conditioning frames pinned in the sequence;
  depth sweep at K1 (2 ... 48 blocks, full width): how the gap to the oracle grows with depth, next to the oracle's own
      bf16-vs-fp32 gap and to north_star's 1e-3 figure;
  K2  49x480p -> [1,16,13,60,104] (20 280 tokens), ALL 48 blocks, one CFG step: whole prediction and Euler update vs the
      oracle, KV-cached == pinned conditioning;
  K3  49x720p -> [1,16,13,90,160] (46 800 tokens), all 48 blocks, one forward vs the oracle;
  K5  121x480p -> [1,16,31,60,104] (48 360 tokens, BASELINE.json config 5), all 48 blocks, 4 conditioning frames, one forward.

At K2 / K3 the oracle cannot run on host cores in seconds (1.5 PFLOP of fp32 per CFG step), so the SAME oracle code is
evaluated with plain PyTorch fp32 ops on the card (`device="cuda"`; `test_oracle_is_device_independent` checks that this
changes nothing beyond fp32 summation order).  bf16 has eps = 7.8e-3, so "1e-3 rel" (north_star) is read as the order of the
relative-L2 gap, not an element-wise bound.

ONE criterion everywhere (round 3; the round-2 file mixed it with absolute bounds 3-5 x above what was measured):
  (a) the HIP path is no further from the fp32 oracle than the oracle's own bf16-rounding-point evaluation is:
      hip_vs_fp32 < 1.5 * oracle_bf16_vs_fp32 + 1e-3;
  (b) the HIP path stays within 1.5 x of ITS OWN measured distance to the bf16-point oracle (`_MEASURED_R2` below: the values of
      gpurun_out/denoise_parity.json at the end of round 2) - a regression that doubled an error fails.
The three numbers per config are printed and written to gpurun_out/denoise_parity.json."""
import json
import os
from pathlib import Path

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda"
_REPORT = {}
# rel-L2 of the HIP path vs the bf16-point oracle as measured at the end of round 2 (MI355X, deterministic kernels)
_MEASURED_R2 = {
    "k1_ncond0_kv1": {"latents": 4.87e-3, "update": 4.43e-3},
    "k1_ncond2_kv1": {"latents": 4.85e-3, "update": 4.40e-3},
    "k1_ncond2_kv0": {"latents": 5.81e-3, "update": 6.11e-3},
    "depth_sweep_k1": {2: 4.06e-3, 4: 5.38e-3, 8: 6.35e-3, 16: 7.44e-3, 48: 1.029e-2},
    "k2_depth48_cfg_step": {"pred": 1.053e-2, "update": 1.293e-2, "latents": 1.118e-3},
    "k3_depth48_forward": 1.031e-2,
    "zero_pad_branch": 2.2e-3,
    "oracle_cuda_vs_cpu_bf16": 3.4e-3,
}


def _record(key, value):
    _REPORT[key] = value
    out = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parents[1])) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        (out / "denoise_parity.json").write_text(json.dumps(_REPORT, indent=1))
    except OSError:
        pass


def _full_width_cfg(depth):
    from oracle import dit_oracle as D
    cfg = D.small_config(hidden_size=4096, depth=depth, num_heads=32, caption_channels=4096)
    cfg["adaln_tembed_dim"] = 512
    return cfg


def _model(depth, seed=1234):
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    m = LongCatVideoTransformer3DModel(device=DEV, dtype=BF16, depth=depth).init_synthetic_(seed)
    return m.eval()


def _pipe(dit):
    from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
    from longcat_video.pipeline_longcat_video import LongCatVideoPipeline
    p = LongCatVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(), dit=dit)
    p.device = torch.device(DEV)
    return p


def _text(L=512, valid=77, neg_valid=5, seed=43):
    g = torch.Generator().manual_seed(seed)
    pe = torch.randn(1, 1, L, 4096, generator=g).to(BF16)
    ne = torch.randn(1, 1, L, 4096, generator=g).to(BF16)
    pm = torch.zeros(1, L, dtype=torch.int64); pm[:, :valid] = 1
    nm = torch.zeros(1, L, dtype=torch.int64); nm[:, :neg_valid] = 1
    return pe, pm, ne, nm


def _P(m, device):
    return {k: v.detach().to(device) for k, v in m.state_dict().items()}


@pytest.fixture(scope="module")
def dit2():
    return _model(2)


@pytest.fixture(scope="module")
def dit48():
    m = _model(48)
    yield m
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("ncond,use_kv,steps", [(0, True, 4), (2, True, 4), (2, False, 2)])
def test_k1_denoise_matches_cpu_oracle_every_step(dit2, ncond, use_kv, steps):
    """BASELINE config 1 (16x256x256, 4-step) at full width, depth 2: latents after EVERY Euler step vs the CPU oracle."""
    from oracle import pipeline_oracle as PO
    cfg = _full_width_cfg(2)
    P = _P(dit2, "cpu")
    pipe = _pipe(dit2)
    lat = torch.randn(1, 16, 5, 32, 32, generator=torch.Generator().manual_seed(42))
    pe, pm, ne, nm = _text()
    got, ref, ref32 = [], [], []
    pipe.denoise(lat.to(DEV), pe.to(DEV), pm.to(DEV), ne.to(DEV), nm.to(DEV), num_cond_latents=ncond,
                 num_inference_steps=steps, guidance_scale=4.0, use_kv_cache=use_kv,
                 step_callback=lambda i, x: got.append(x.detach().float().cpu().clone()))
    PO.denoise(P, cfg, lat, pe, pm, ne, nm, num_cond_latents=ncond, num_inference_steps=steps, guidance_scale=4.0,
               use_kv_cache=use_kv, bf16=True, step_callback=lambda i, x: ref.append(x.clone()))
    # the fp32 ground truth (no rounding points), evaluated with fp32 torch ops on the card (== the CPU to 1e-5)
    PO.denoise(_P(dit2, DEV), cfg, lat.to(DEV), pe.to(DEV), pm.to(DEV), ne.to(DEV), nm.to(DEV), num_cond_latents=ncond,
               num_inference_steps=steps, guidance_scale=4.0, use_kv_cache=use_kv, bf16=False,
               step_callback=lambda i, x: ref32.append(x.cpu().clone()))
    assert len(got) == len(ref) == len(ref32) == steps
    errs, errs32, own = [], [], []
    for i in range(steps):
        g_i, r_i, f_i = got[i], ref[i], ref32[i]
        if g_i.shape[2] != r_i.shape[2]:          # the KV-cached product loop hands its callback the noise frames only
            r_i, f_i = r_i[:, :, -g_i.shape[2]:], f_i[:, :, -g_i.shape[2]:]
        errs.append(rel_l2(g_i, r_i)); errs32.append(rel_l2(g_i, f_i)); own.append(rel_l2(r_i, f_i))
    n = got[-1].shape[2]
    upd = rel_l2(got[-1] - lat[:, :, -n:], ref[-1][:, :, -n:] - lat[:, :, -n:])   # the update, not the (large, shared) starting noise
    key = f"k1_ncond{ncond}_kv{int(use_kv)}"
    print(f"K1 ncond={ncond} kv={use_kv}: per step HIP-vs-oracle(bf16 points) {['%.2e' % e for e in errs]}, HIP-vs-fp32 "
          f"{['%.2e' % e for e in errs32]}, oracle bf16-vs-fp32 {['%.2e' % e for e in own]}; accumulated update rel-L2 {upd:.2e}")
    _record(key, {"latents_rel_l2_per_step": errs, "hip_vs_fp32_per_step": errs32, "oracle_bf16_vs_fp32_per_step": own,
                  "update_rel_l2": upd})
    m = _MEASURED_R2[key]
    for i in range(steps):
        assert errs32[i] < 1.5 * own[i] + 1e-3, (i, errs32, own)                      # (a)
    assert max(errs) < 1.5 * m["latents"] and upd < 1.5 * m["update"], (errs, upd)    # (b)


@pytest.mark.parametrize("ncond,use_kv", [(0, True), (2, True), (2, False)])
def test_graph_replayed_denoise_equals_the_eager_loop(dit2, ncond, use_kv, monkeypatch):
    """LCV_DENOISE_GRAPH=1: the DiT forward of a denoise step is captured into a hipGraph after the first (eager) step and
    replayed with the step's latents / timestep copied into static buffers.  Same kernels, same order, same buffers' contents:
    the latents after every step must equal the eager loop's bit for bit (K1 shape, full width, CFG, all three
    conditioning forms).  Also prints the step times: at 1 280 tokens the eager step is launch-bound."""
    import time
    pipe = _pipe(dit2)
    lat = torch.randn(1, 16, 5, 32, 32, generator=torch.Generator().manual_seed(7)).to(DEV)
    pe, pm, ne, nm = (t.to(DEV) for t in _text())

    def run(flag, steps=8):
        monkeypatch.setenv("LCV_DENOISE_GRAPH", flag)
        seen = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            out = pipe.denoise(lat, pe, pm, ne, nm, num_cond_latents=ncond, num_inference_steps=steps, guidance_scale=4.0,
                               use_kv_cache=use_kv, step_callback=lambda i, x: seen.append(x.clone()))
        torch.cuda.synchronize()
        return out, seen, (time.perf_counter() - t0) / steps
    run("0", 3)                                            # warm-up of both code paths' caches
    eager, seen_e, t_e = run("0")
    graph, seen_g, t_g = run("1")
    assert len(seen_e) == len(seen_g) == 8
    for i, (a, b) in enumerate(zip(seen_e, seen_g)):
        assert torch.equal(a, b), f"step {i}"
    assert torch.equal(eager, graph)
    print(f"K1 depth 2, ncond={ncond} kv={use_kv}: eager {t_e * 1e3:.2f} ms / step, graph (incl. capture) {t_g * 1e3:.2f} ms / step")
    _record(f"graph_k1_ncond{ncond}_kv{int(use_kv)}", {"eager_ms_per_step": t_e * 1e3, "graph_ms_per_step_incl_capture": t_g * 1e3})


def test_oracle_is_device_independent(dit2):
    """The oracle evaluated with torch fp32 ops on the card equals the oracle on host cores up to summation order: this is
    what lets the K2 / K3 tests below use it at sizes the host cannot finish."""
    from oracle import dit_oracle as D
    cfg = _full_width_cfg(2)
    g = torch.Generator().manual_seed(3)
    hs = torch.randn(1, 16, 2, 16, 16, generator=g).to(BF16)
    pe, pm, _, _ = _text(L=64, valid=20)
    ts = torch.tensor([[0.0, 640.0]]).to(BF16)
    a = D.dit_forward(_P(dit2, "cpu"), cfg, hs, ts, pe, pm, 1, bf16=True)
    b = D.dit_forward(_P(dit2, DEV), cfg, hs.to(DEV), ts.to(DEV), pe.to(DEV), pm.to(DEV), 1, bf16=True)
    e = rel_l2(b, a)
    print(f"oracle cuda-vs-cpu rel-L2 {e:.2e}")
    # measured 3.4e-3: bf16 rounding points flip on fp32 last-bit differences between the two devices' summation orders —
    # the oracle's OWN sensitivity, and the scale against which the HIP-vs-oracle gaps below (4e-3 at depth 2) are read;
    # the fp32 mode is the tight check
    _record("oracle_cuda_vs_cpu_bf16", e)
    assert e < 1.5 * _MEASURED_R2["oracle_cuda_vs_cpu_bf16"]
    a32 = D.dit_forward(_P(dit2, "cpu"), cfg, hs, ts, pe, pm, 1, bf16=False)
    b32 = D.dit_forward(_P(dit2, DEV), cfg, hs.to(DEV), ts.to(DEV), pe.to(DEV), pm.to(DEV), 1, bf16=False)
    assert rel_l2(b32, a32, bound=4.0e-6) < 4.0e-6


def test_depth_sweep_k1_full_width(dit48):
    """How the HIP-vs-oracle gap grows with depth at full width (K1 tokens): printed next to the oracle's own
    bf16-vs-fp32 gap (what bf16 storage costs ANY implementation) and north_star's 1e-3."""
    from oracle import dit_oracle as D
    g = torch.Generator().manual_seed(42)
    hs = torch.randn(1, 16, 5, 32, 32, generator=g).to(BF16)
    pe, pm, _, _ = _text()
    ts = torch.full((1, 5), 700.0).to(BF16)
    P = _P(dit48, DEV)
    rows = {}
    blocks = dit48.blocks
    for depth in (2, 4, 8, 16, 48):
        cfg = _full_width_cfg(depth)
        dit48.blocks = blocks[:depth]
        try:
            with torch.no_grad():
                got = dit48(hidden_states=hs.to(DEV), timestep=ts.to(DEV), encoder_hidden_states=pe.to(DEV),
                            encoder_attention_mask=pm.to(DEV), num_cond_latents=0)
        finally:
            dit48.blocks = blocks
        ref = D.dit_forward(P, cfg, hs.to(DEV), ts.to(DEV), pe.to(DEV), pm.to(DEV), 0, bf16=True)
        ref32 = D.dit_forward(P, cfg, hs.to(DEV), ts.to(DEV), pe.to(DEV), pm.to(DEV), 0, bf16=False)
        rows[depth] = {"hip_vs_oracle_bf16pts": rel_l2(got, ref), "hip_vs_oracle_fp32": rel_l2(got, ref32),
                       "oracle_bf16pts_vs_fp32": rel_l2(ref, ref32)}
        print(f"depth {depth:2d}: hip-vs-oracle(bf16 points) {rows[depth]['hip_vs_oracle_bf16pts']:.2e}  "
              f"hip-vs-fp32 {rows[depth]['hip_vs_oracle_fp32']:.2e}  oracle bf16-vs-fp32 {rows[depth]['oracle_bf16pts_vs_fp32']:.2e}"
              f"  (north_star: 1e-3)")
    _record("depth_sweep_k1", rows)
    for depth, r in rows.items():
        # the HIP path must sit no further from the fp32 truth than ~the oracle's own bf16 emulation does (x1.5 + 1e-3)
        assert r["hip_vs_oracle_fp32"] < 1.5 * r["oracle_bf16pts_vs_fp32"] + 1e-3, (depth, r)              # (a)
        assert r["hip_vs_oracle_bf16pts"] < 1.5 * _MEASURED_R2["depth_sweep_k1"][depth], (depth, r)          # (b)


def test_k2_full_depth_cfg_step_vs_oracle_and_cached_equals_pinned(dit48):
    """BASELINE config 2 (49x480p, 20 280 tokens) at ALL 48 blocks: one CFG step (B = 2 pass + zero-star + Euler)."""
    from oracle import dit_oracle as D
    from oracle import pipeline_oracle as PO
    cfg = _full_width_cfg(48)
    pipe = _pipe(dit48)
    lat = torch.randn(1, 16, 13, 60, 104, generator=torch.Generator().manual_seed(42)).to(DEV)
    pe, pm, ne, nm = (t.to(DEV) for t in _text())
    out = pipe.denoise(lat, pe, pm, ne, nm, num_cond_latents=0, num_inference_steps=50, guidance_scale=4.0, stop_step=1)
    assert out.shape == lat.shape and torch.isfinite(out).all()
    P = _P(dit48, DEV)
    seen = []
    # the oracle's first step of the same 50-step schedule
    ts, sig = PO.sigma_grid(50)
    x_in = D.bf16_round(lat).expand(2, -1, -1, -1, -1)
    t_in = D.bf16_round(torch.full((2, 13), float(ts[0]), device=DEV))
    pred = D.dit_forward(P, cfg, x_in, t_in, torch.cat([ne, pe]), torch.cat([nm, pm]), 0, bf16=True)
    v = PO.cfg_zero_star(pred[1:2], pred[0:1], 4.0)
    ref = PO.euler_update(lat, v, float(sig[1]) - float(sig[0]))
    pred32 = D.dit_forward(P, cfg, x_in, t_in, torch.cat([ne, pe]), torch.cat([nm, pm]), 0, bf16=False)   # fp32 ground truth
    with torch.no_grad():
        got_pred = dit48(hidden_states=x_in.to(BF16), timestep=t_in.to(BF16), encoder_hidden_states=torch.cat([ne, pe]),
                         encoder_attention_mask=torch.cat([nm, pm]), num_cond_latents=0)
    e_pred = rel_l2(got_pred, pred)
    e_pred32, own = rel_l2(got_pred, pred32), rel_l2(pred, pred32)
    e_upd = rel_l2(out - lat, ref - lat)
    e_lat = rel_l2(out, ref)
    print(f"K2 depth 48: prediction HIP-vs-oracle(bf16 points) {e_pred:.2e}, HIP-vs-fp32 {e_pred32:.2e}, oracle bf16-vs-fp32 {own:.2e}; "
          f"step update rel-L2 {e_upd:.2e}, latents rel-L2 {e_lat:.2e}")
    _record("k2_depth48_cfg_step", {"pred_rel_l2": e_pred, "pred_hip_vs_fp32": e_pred32, "pred_oracle_bf16_vs_fp32": own,
                                    "update_rel_l2": e_upd, "latents_rel_l2": e_lat})
    assert torch.equal(lat, torch.randn(1, 16, 13, 60, 104, generator=torch.Generator().manual_seed(42)).to(DEV))  # input untouched
    m = _MEASURED_R2["k2_depth48_cfg_step"]
    assert e_pred32 < 1.5 * own + 1e-3                                                                       # (a)
    assert e_pred < 1.5 * m["pred"] and e_upd < 1.5 * m["update"] and e_lat < 1.5 * m["latents"]             # (b)
    del pred, pred32, v, ref, got_pred, x_in
    torch.cuda.empty_cache()
    # conditioning frames: KV cache of one t = 0 pass == frames pinned in the sequence (4 cond + 9 noise latent frames)
    a = pipe.denoise(lat, pe, pm, ne, nm, num_cond_latents=4, num_inference_steps=50, use_kv_cache=True, stop_step=1)
    b = pipe.denoise(lat, pe, pm, ne, nm, num_cond_latents=4, num_inference_steps=50, use_kv_cache=False, stop_step=1)
    assert torch.equal(a[:, :, :4], lat[:, :, :4]) and torch.equal(b[:, :, :4], lat[:, :, :4])
    e_kv = rel_l2(a[:, :, 4:] - lat[:, :, 4:], b[:, :, 4:] - lat[:, :, 4:])
    print(f"K2 depth 48: KV-cached vs pinned conditioning, step update rel-L2 {e_kv:.2e}")
    _record("k2_depth48_cached_vs_pinned_update_rel_l2", e_kv)
    assert e_kv < 7.5e-7        # the same kernels on the same rows: bit-identical (0.0, end of round 2) or a split-K tail's summation order (5.0e-7, mid round 2)


def test_k3_full_depth_forward_vs_oracle(dit48):
    """The headline config (49x720p, 46 800 tokens, 48 blocks): one forward vs the oracle at the bf16 rounding points."""
    from oracle import dit_oracle as D
    cfg = _full_width_cfg(48)
    hs = torch.randn(1, 16, 13, 90, 160, generator=torch.Generator().manual_seed(42)).to(BF16).to(DEV)
    pe, pm, _, _ = (t.to(DEV) for t in _text())
    ts = torch.full((1, 13), 999.0).to(BF16).to(DEV)
    with torch.no_grad():
        got = dit48(hidden_states=hs, timestep=ts, encoder_hidden_states=pe, encoder_attention_mask=pm, num_cond_latents=0)
    P = _P(dit48, DEV)
    ref = D.dit_forward(P, cfg, hs, ts, pe, pm, 0, bf16=True)
    ref32 = D.dit_forward(P, cfg, hs, ts, pe, pm, 0, bf16=False)
    e, e32, own = rel_l2(got, ref), rel_l2(got, ref32), rel_l2(ref, ref32)
    print(f"K3 depth 48: prediction HIP-vs-oracle(bf16 points) {e:.2e}, HIP-vs-fp32 {e32:.2e}, oracle bf16-vs-fp32 {own:.2e}")
    _record("k3_depth48_forward_rel_l2", e)
    _record("k3_depth48_forward", {"hip_vs_oracle_bf16pts": e, "hip_vs_oracle_fp32": e32, "oracle_bf16pts_vs_fp32": own})
    assert torch.isfinite(got).all()
    assert e32 < 1.5 * own + 1e-3                                     # (a)
    assert e < 1.5 * _MEASURED_R2["k3_depth48_forward"]               # (b)


def test_k5_full_depth_forward_vs_oracle(dit48):
    """BASELINE.json config 5 (Panda70M 121-frame clip, 121x480p -> latents [1,16,31,60,104], 48 360 tokens, 48 blocks) on ONE
    GPU: one forward with 4 conditioning latent frames pinned (the continuation form the long-clip runs use) vs the oracle —
    the size the sequence-parallel path shards; before round 4 only its shard arithmetic was tested."""
    from oracle import dit_oracle as D
    cfg = _full_width_cfg(48)
    T, ncond = 31, 4
    hs = torch.randn(1, 16, T, 60, 104, generator=torch.Generator().manual_seed(45)).to(BF16).to(DEV)
    pe, pm, _, _ = (t.to(DEV) for t in _text())
    ts = torch.zeros(1, T); ts[:, ncond:] = 800.0
    ts = ts.to(BF16).to(DEV)
    with torch.no_grad():
        got = dit48(hidden_states=hs, timestep=ts, encoder_hidden_states=pe, encoder_attention_mask=pm, num_cond_latents=ncond)
    P = _P(dit48, DEV)
    ref = D.dit_forward(P, cfg, hs, ts, pe, pm, ncond, bf16=True)
    ref32 = D.dit_forward(P, cfg, hs, ts, pe, pm, ncond, bf16=False)
    e, e32, own = rel_l2(got, ref), rel_l2(got, ref32), rel_l2(ref, ref32)
    print(f"K5 depth 48 (48 360 tokens, 4 cond frames): prediction HIP-vs-oracle(bf16 points) {e:.2e}, HIP-vs-fp32 {e32:.2e}, "
          f"oracle bf16-vs-fp32 {own:.2e}")
    _record("k5_depth48_forward", {"hip_vs_oracle_bf16pts": e, "hip_vs_oracle_fp32": e32, "oracle_bf16pts_vs_fp32": own})
    assert got.shape == (1, 16, T, 60, 104) and torch.isfinite(got).all()
    assert e32 < 1.5 * own + 1e-3                                     # (a)


def test_text_tokens_zero_pad_branch_matches_oracle():
    """`dit.text_tokens_zero_pad = True` (run_delta_a.py:170-178): embedded text multiplied by the mask, mask set to ones,
    so all L tokens — zeros included — take part in the cross-attention softmax."""
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    from oracle import dit_oracle as D
    cfg = D.small_config(hidden_size=256, depth=2, num_heads=2, caption_channels=64)
    cfg["text_tokens_zero_pad"] = True
    P = D.make_params(cfg, seed=17, std=0.05)
    m = LongCatVideoTransformer3DModel(device=DEV, dtype=BF16, hidden_size=256, depth=2, num_heads=2, caption_channels=64,
                                       adaln_tembed_dim=64, text_tokens_zero_pad=True)
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not missing and not unexpected and m.text_tokens_zero_pad is True
    g = torch.Generator().manual_seed(1)
    hs = torch.randn(2, 16, 3, 8, 12, generator=g).to(BF16)
    y = torch.randn(2, 1, 20, 64, generator=g).to(BF16)
    mask = torch.zeros(2, 20, dtype=torch.int64); mask[0, :13] = 1; mask[1, :20] = 1
    ts = torch.tensor([[0.0, 371.5, 371.5], [0.0, 902.25, 902.25]]).to(BF16)
    with torch.no_grad():
        got = m.eval()(hidden_states=hs.to(DEV), timestep=ts.to(DEV), encoder_hidden_states=y.to(DEV),
                       encoder_attention_mask=mask.to(DEV), num_cond_latents=1)
    ref = D.dit_forward(P, cfg, hs, ts, y, mask, 1, bf16=True)
    cfg_off = dict(cfg, text_tokens_zero_pad=False)
    ref_off = D.dit_forward(P, cfg_off, hs, ts, y, mask, 1, bf16=True)
    e = rel_l2(got, ref)
    print(f"zero-pad branch rel-L2 {e:.2e}; distance between the two branches {rel_l2(ref, ref_off):.2e}")
    _record("zero_pad_branch_rel_l2", e)
    assert e < 1.5 * _MEASURED_R2["zero_pad_branch"] and rel_l2(ref, ref_off) > 5 * e      # the branch is really taken (packing would give ref_off)
