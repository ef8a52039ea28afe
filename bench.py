#!/usr/bin/env python3
"""Headline benchmark: denoised latent frames / s on the 49x720p, 50-step, CFG denoise of the LongCat-Video DiT
(13.6 B parameters, 48 blocks) — BASELINE.json's metric on workload K3 (latents [1,16,13,90,160], 46 800 tokens).

A "step" is one denoise step of the hot path: a batched (uncond | cond) DiT forward over all tokens followed by the
fused CFG-zero-star + Euler update.  `value` = T_lat / (num_inference_steps * seconds_per_step) = finished latent
frames per second of the 50-step job, whole-job aggregate over ranks (data-parallel: one independent video per GPU,
no data-path collective => "weak" scaling).  Inputs are resident in HBM before the timed region.

Extra objects on the JSON line: `roofline` (flash-attention forward, the dominant kernel: algorithmic flops per launch
/ its average launch duration measured with HIP events on the launch stream inside the timed region) and
`cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "longcat-video-tta_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

WORKLOADS = {  # name: (T_lat, h, w, description)
    "K1": (5, 32, 32, "16x256x256 clip (17 frames -> 5 latent frames, 1 280 tokens)"),
    "K2": (13, 60, 104, "49x480p (13 latent frames, 20 280 tokens)"),
    "K3": (13, 90, 160, "49x720p (13 latent frames, 46 800 tokens)"),
    "K3p": (49, 90, 160, "north_star's literal 49x90x160 latents (49 latent frames = 193 frames at 720p, 176 400 tokens)"),
    "K5": (31, 60, 104, "121x480p long clip (31 latent frames, 48 360 tokens)"),
}
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md


_T0 = time.perf_counter()


def progress(msg: str):
    """One line per stage on stderr (stdout carries exactly one JSON line): a default run takes a few minutes and a
    silent one looks hung to whoever launched it."""
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="K3", choices=sorted(WORKLOADS))
    ap.add_argument("--num-inference-steps", type=int, default=50)
    ap.add_argument("--guidance-scale", type=float, default=4.0)
    ap.add_argument("--depth", type=int, default=48, help="debug only: anything but 48 is not the named model")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras (VAE decode and LoRA-TTA inner-loop step times for the per-video wall clock)")
    ap.add_argument("--full-tta-leg", action="store_true",
                    help="also time full-model TTA at the 480p operating point (SURVEY §2.1 #9 marks it out of scope: opt-in since round 3)")
    ap.add_argument("--no-k3p", action="store_true", help="skip the one secondary CFG step at the literal 49x90x160 size (~45 s)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N > 1 and no torch.distributed environment: print the launcher command and exit")
    ap.add_argument("--parallelism", default="dp", choices=["dp", "sp"],
                    help="dp: one independent video per GPU (weak scaling, default); sp: ONE video, latent frames sharded "
                         "over the GPUs with an RCCL K/V all-gather per attention layer (strong scaling, config K5)")
    return ap.parse_args()


def cpu_baseline(num_inference_steps: int) -> dict:
    """The CPU oracle (oracle/pipeline_oracle.py on oracle/dit_oracle.py, kind 'port') timed on this box's host cores as
    SURVEY §8(d) / BASELINE.md §3 prescribe: config K1 (16x256x256 -> latents [1,16,5,32,32], 1 280 tokens, 512 text tokens
    of which 77 valid), full width, 4 Euler steps with CFG off and then 4 with CFG on; in each run the first step is the
    warm-up and the other three are timed (median).  All 48 blocks in fp32 are 54 GB of weights and ~40 s per forward on a
    16-core share, so depth 2 is timed and multiplied by 24 (embedders / head < 0.5 %): a BOUNDED sample, ~25 s of CPU work.
    Threads = min(affinity mask, 16: the CPU share of a one-GPU job), stated in `cores`."""
    from oracle import dit_oracle as orc
    from oracle import pipeline_oracle as porc
    # the affinity mask of a shared GPU host can list every core of the machine while the CPU share of a one-GPU job is 16:
    # more threads than that only thrash (a 128-thread run of this function did not finish in 7 minutes)
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(cores)
    cfg = dict(hidden_size=4096, depth=2, num_heads=32, in_channels=16, out_channels=16, adaln_tembed_dim=512,
               caption_channels=4096, patch_size=(1, 2, 2), ffn_hidden=orc.ffn_hidden_dim(4096),
               frequency_embedding_size=256, text_tokens_zero_pad=False)
    P = {k: v.float() for k, v in orc.make_params(cfg, seed=1234).items()}  # bf16-valued weights, widened once
    T, h, w, _ = WORKLOADS["K1"]
    g = torch.Generator().manual_seed(42)
    lat = torch.randn(1, 16, T, h, w, generator=g)
    g = torch.Generator().manual_seed(43)
    pe = torch.randn(1, 1, 512, 4096, generator=g).to(torch.bfloat16)
    ne = torch.randn(1, 1, 512, 4096, generator=g).to(torch.bfloat16)
    mask = torch.zeros(1, 512, dtype=torch.int64)
    mask[:, :77] = 1

    def timed_steps(cfg_on: bool):
        marks = [time.perf_counter()]
        with torch.no_grad():
            porc.denoise(P, cfg, lat, pe, mask, ne if cfg_on else None, mask if cfg_on else None, num_cond_latents=0,
                         num_inference_steps=4, guidance_scale=4.0 if cfg_on else 1.0, bf16=True,
                         step_callback=lambda i, x: marks.append(time.perf_counter()))
        steps = sorted(b - a for a, b in zip(marks[1:-1], marks[2:]))     # steps 1..3 (step 0 = warm-up)
        return steps[len(steps) // 2]
    step_off = timed_steps(False)
    progress(f"CPU baseline: CFG-off step {step_off:.2f} s")
    step_on = timed_steps(True)
    progress(f"CPU baseline: CFG-on step {step_on:.2f} s; TTA steps ...")
    sec_per_step48 = step_on * 24            # 2 -> 48 blocks
    # One LoRA-TTA inner step on the same sample: forward + LoRA-only backward through the same 2 blocks
    # with rank-8 adapters on qkv/proj folded in as W + s*B*A (torch autograd over the oracle), 1 warm-up + 1 timed, x24.
    x = lat.to(torch.bfloat16)
    ts = torch.full((1, T), 500.0).to(torch.bfloat16)
    rnd = orc.bf16_round
    with torch.no_grad():
        xe = orc.x_embedder(P, x.float(), (1, 2, 2), rnd)
        t = orc.t_embedder(P, ts.float().flatten()).reshape(1, T, -1)
        ye, lens = orc.pack_text(orc.y_embedder(P, pe.float(), rnd), mask)
    tta = []
    for _ in range(2):
        P2 = dict(P)
        for i in range(2):
            for n_ in ("attn.qkv", "attn.proj"):
                W_ = P[f"blocks.{i}.{n_}.weight"]
                A_ = (torch.randn(8, W_.shape[1], generator=g) * 0.02).requires_grad_(True)
                B_ = torch.zeros(W_.shape[0], 8, requires_grad=True)
                P2[f"blocks.{i}.{n_}.weight"] = W_ + 2.0 * (B_ @ A_)
        t0 = time.perf_counter()
        out = xe
        for i in range(2):
            out = orc.block_forward(P2, f"blocks.{i}.", out, ye, t, lens, (T, h // 2, w // 2), 0, 32)
        out.square().mean().backward()
        tta.append(time.perf_counter() - t0)
    return {"value": T / (num_inference_steps * sec_per_step48), "unit": "denoised latent frames/s", "cores": cores,
            "kind": "port", "k1_step_s_cfg_off_depth2": round(step_off, 3), "k1_step_s_cfg_on_depth2": round(step_on, 3),
            "k1_4step_cfg_on_s_extrapolated_48_blocks": round(4 * sec_per_step48, 1),
            "tta_inner_step_s": round(tta[-1] * 24, 1),
            "sample": f"K1 {WORKLOADS['K1'][3]}, full width, 2 of 48 blocks: 4 Euler steps CFG off, then 4 with CFG 4.0 "
                      "(step 0 of each run = warm-up, median of steps 1-3), oracle/pipeline_oracle.py at the bf16 rounding "
                      f"points in fp32 math on {cores} threads (min(affinity mask, 16)); value = 5 latent frames / "
                      f"({num_inference_steps} steps x CFG-on step time x 24 [2 -> 48 blocks]); tta_inner_step_s = one forward + "
                      "LoRA-only backward (rank 8 on qkv+proj) of the same 2 blocks, second of two runs, x24"}


def measure_reference_point(dit, dev, pe, pm, ne, nm) -> dict:
    """The ONE operating point the reference publishes numbers for: 480p, 14 conditioning + 14
    generated frames (29-frame pipeline -> 8 latent frames: 4 clean + 4 noised = 6 240 noise tokens attending 12 480),
    50 steps, CFG 4.0, KV cache -> 80.4 s / video (experimental_report.md:102); LoRA-TTA inner loop, 20 steps, r=8 on
    qkv+proj of all 48 blocks, Tc=3 + Tt=1 latent frames (6 240 tokens) -> 84.2-85.1 s (experimental_report.md:325-329)."""
    import gc
    from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
    from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
    from longcat_video.pipeline_longcat_video import LongCatVideoPipeline
    from tta.inner_loop import choose_gradient_checkpointing, finetune_lora_on_conditioning
    from tta.lora import inject_lora_into_dit, remove_lora_from_dit
    out = {}
    h, w = 60, 104
    vae = AutoencoderKLWan(device=dev).init_synthetic_()
    pipe = LongCatVideoPipeline(vae=vae, scheduler=FlowMatchEulerDiscreteScheduler(), dit=dit)
    pipe.device = dev
    g = torch.Generator(device=dev).manual_seed(11)
    frames = torch.rand((1, 3, 13, 480, 832), generator=g, device=dev) * 2 - 1      # 13 of the 14 cond frames: 1 + 4k
    lat = torch.randn((1, 16, 8, h, w), generator=g, device=dev, dtype=torch.float32)

    def gen():
        cond = vae.encode(frames.to(torch.bfloat16)).latent_dist.mode().float()      # [1,16,4,60,104]
        x = lat.clone(); x[:, :, :4] = cond
        z = pipe.denoise(x, pe, pm, ne, nm, num_cond_latents=4, num_inference_steps=50, guidance_scale=4.0, use_kv_cache=True)
        return pipe._decode_to_numpy(z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    video = gen()
    torch.cuda.synchronize(); out["gen_s"] = time.perf_counter() - t0
    out["gen_frames"] = int(video.shape[0])
    out["gen_s_reference_h200"] = 80.4
    del vae, pipe, video
    gc.collect(); torch.cuda.empty_cache()
    for p_ in dit.parameters():
        p_.requires_grad = False
    mods = inject_lora_into_dit(dit, rank=8, alpha=16.0, target_modules=["qkv", "proj"])
    cond = torch.randn((1, 16, 3, h, w), generator=g, device=dev).to(torch.bfloat16)
    train = torch.randn((1, 16, 1, h, w), generator=g, device=dev).to(torch.bfloat16)
    choose_gradient_checkpointing(dit, 4 * (h // 2) * (w // 2))
    kw = dict(lr=2e-4, warmup_steps=3, device=str(dev), dtype=torch.bfloat16)
    finetune_lora_on_conditioning(dit, mods, cond, train, pe, pm, num_steps=1, **kw)   # warm-up (W^T copies)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    finetune_lora_on_conditioning(dit, mods, cond, train, pe, pm, num_steps=20, **kw)
    torch.cuda.synchronize(); out["tta20_s"] = time.perf_counter() - t0
    out["tta20_s_reference_h200"] = [84.2, 85.1]
    out["per_video_s"] = out["gen_s"] + out["tta20_s"]
    out["per_video_s_reference_h200"] = 167.6
    remove_lora_from_dit(dit)
    # delta-A / AdaSteer-1 (one delta in R^512 on the timestep embedding; backward through all 48 frozen blocks down to t):
    # 20 steps -> 82.8 s on 1x H200 (experimental_report.md:230-232)
    from tta.delta import DeltaAWrapper, optimize_delta_a
    wrap = DeltaAWrapper(dit, dit.config.adaln_tembed_dim).to(dev)
    kw = dict(lr=1e-3, device=str(dev), dtype=torch.bfloat16)
    optimize_delta_a(wrap, cond, train, pe, pm, num_steps=1, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    optimize_delta_a(wrap, cond, train, pe, pm, num_steps=20, **kw)
    torch.cuda.synchronize(); out["delta_a20_s"] = time.perf_counter() - t0
    out["delta_a20_s_reference_h200"] = 82.8
    wrap.remove_from_dit()
    return out


def measure_full_tta_reference_point(dit, dev, pe, pm) -> dict:
    """Full-model TTA at the same operating point (SGD, 5 steps -> 24.0 s on 1x H200,
    experimental_report.md:182-184): all 13.6 B parameters trainable, block checkpointing on, fused clip + SGD.
    Runs LAST: it moves the weights."""
    import functools
    import gc
    from torch.utils.checkpoint import checkpoint
    from lcv_hip import autograd_ops
    from tta.full_tta import finetune_full_on_conditioning
    autograd_ops.clear_weight_caches()                      # the LoRA legs' resident W^T copies (27 GB)
    for b in dit.blocks:
        b.ffn._w13 = None
    gc.collect(); torch.cuda.empty_cache()
    h, w = 60, 104
    g = torch.Generator(device=dev).manual_seed(13)
    cond = torch.randn((1, 16, 3, h, w), generator=g, device=dev).to(torch.bfloat16)
    train = torch.randn((1, 16, 1, h, w), generator=g, device=dev).to(torch.bfloat16)
    dit.gradient_checkpointing = True
    dit._gradient_checkpointing_func = functools.partial(checkpoint, use_reentrant=False)
    for p_ in dit.parameters():
        p_.requires_grad = True
    kw = dict(lr=1e-5, warmup_steps=2, device=str(dev), dtype=torch.bfloat16, optimizer_type="sgd")
    finetune_full_on_conditioning(dit, cond, train, pe, pm, num_steps=2, **kw)   # warm-up: allocator sizes, weight-transpose buffers
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = finetune_full_on_conditioning(dit, cond, train, pe, pm, num_steps=5, **kw)
    torch.cuda.synchronize()
    out = {"full_tta5_s": time.perf_counter() - t0, "full_tta5_s_reference_h200": 24.0,
           "full_tta_losses": [round(x, 4) for x in res["losses"]]}
    for p_ in dit.parameters():
        p_.requires_grad = False
        p_.grad = None
    dit.gradient_checkpointing = False
    return out


def measure_extras(dit, dev, T, h, w, pe, pm) -> dict:
    """VAE decode of the finished clip and the LoRA-TTA inner-loop step (lora_experiment config: r=8, alpha 16,
    qkv+proj on all 48 blocks, Tc=4 clean + Tt=3 noised latent frames at the bench resolution)."""
    import gc
    from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
    from tta.inner_loop import choose_gradient_checkpointing, finetune_lora_on_conditioning
    from tta.lora import inject_lora_into_dit, remove_lora_from_dit
    out = {}
    vae = AutoencoderKLWan(device=dev).init_synthetic_()
    z = torch.randn((1, 16, T, h, w), device=dev).to(torch.bfloat16)
    vae.decode(z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vae.decode(z)
    torch.cuda.synchronize(); out["vae_decode_s"] = time.perf_counter() - t0
    work = vae.decode_work(T, h, w)              # algorithmic conv / attention MACs of the decoder graph (true channel counts)
    out["vae_decode_tflop"] = round(work["flops"] / 1e12, 1)
    out["vae_decode_frac_of_mfma_peak"] = round(work["flops"] / out["vae_decode_s"] / 1e12 / MFMA_PEAK_TFLOPS, 3)
    del vae, z
    gc.collect(); torch.cuda.empty_cache()
    for p in dit.parameters():
        p.requires_grad = False
    mods = inject_lora_into_dit(dit, rank=8, alpha=16.0, target_modules=["qkv", "proj"])
    g = torch.Generator(device=dev).manual_seed(7)
    cond = torch.randn((1, 16, 4, h, w), generator=g, device=dev).to(torch.bfloat16)
    train = torch.randn((1, 16, 3, h, w), generator=g, device=dev).to(torch.bfloat16)
    ckpt = choose_gradient_checkpointing(dit, 7 * (h // 2) * (w // 2))
    kw = dict(lr=2e-4, warmup_steps=3, device=str(dev), dtype=torch.bfloat16)
    finetune_lora_on_conditioning(dit, mods, cond, train, pe, pm, num_steps=1, **kw)   # warm-up: builds the W^T copies
    from lcv_hip import ops
    ops.PROFILE_BWD = []      # HIP events around every self-attention backward of the 20 steps (the dominant kernels of this leg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = finetune_lora_on_conditioning(dit, mods, cond, train, pe, pm, num_steps=20, **kw)   # BASELINE config 3: 20 iterations
    torch.cuda.synchronize()
    out["tta20_s"] = time.perf_counter() - t0          # all 20 inner steps, measured
    prof_b, ops.PROFILE_BWD = ops.PROFILE_BWD, None
    big = [(s.elapsed_time(e), f) for (s, e, f, nq, nk) in prof_b if nk > 512]        # self-attention regions (not the 77-key text)
    if big:
        ms = sum(m for m, _ in big); fl = sum(f for _, f in big)
        out["tta_attn_bwd_roofline"] = {"kernels": "attn_bwd_delta + attn_bwd_dkv2 + attn_bwd_dq2 (two passes, no atomics)",
                                        "bound": "mfma", "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": MFMA_PEAK_TFLOPS,
                                        "unit": "TFLOP/s (algorithmic: 10 B H Nq Nk D)", "frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 3),
                                        "ms_per_layer": round(ms / (20 * len(dit.blocks)), 2), "calls": len(big)}
    out["tta_step_s"] = out["tta20_s"] / 20
    out["tta_tokens"] = 7 * (h // 2) * (w // 2)
    out["tta_block_checkpointing"] = bool(ckpt)
    out["tta_losses"] = [round(x, 4) for x in res["losses"][:2]] + ["..."] + [round(res["losses"][-1], 4)]
    # the same 20 steps WITH the anchored early stopper ("a second run ES on"): 1 held-out latent frame, 3 sigmas x 2
    # noise draws scored as one resident batch at setup and every 5 steps (patience 3: a synthetic loss keeps falling, so all
    # 20 steps and all 5 checks run)
    from tta.early_stopping import AnchoredEarlyStopper
    from tta.lora import reset_lora_weights
    reset_lora_weights(mods)
    val = torch.randn((1, 16, 1, h, w), generator=g, device=dev).to(torch.bfloat16)
    es = AnchoredEarlyStopper(check_every=5, patience=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    es.setup(dit, cond, val, pe, pm, device=str(dev), dtype=torch.bfloat16, video_id="bench")
    res_es = finetune_lora_on_conditioning(dit, mods, cond, train, pe, pm, num_steps=20, early_stopper=es, **kw)
    torch.cuda.synchronize()
    out["tta20_es_s"] = time.perf_counter() - t0
    out["tta20_es_steps_run"] = len(res_es["losses"])
    out["tta20_es_check_time_s"] = round(res_es.get("es_check_time", 0.0), 2)
    out["tta20_es_checks"] = (res_es.get("early_stopping_info") or {}).get("total_checks")
    remove_lora_from_dit(dit)
    return out


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with no torch.distributed environment: start N ranks of this same file under
    torch.distributed.run (one process per GPU over RCCL) as a CHILD process, relay its output — rank 0 prints the one JSON
    line — and return its exit code.  This process never touches the GPU (importing torch does not initialise it), and nothing
    that has is ever replaced by another program."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    passthrough = [a for a in sys.argv[1:] if a != "--dry-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *passthrough]
    if args.dry_launch:
        print(" ".join(cmd), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool's host driver
    progress(f"no torch.distributed environment: launching {args.gpus} ranks: {' '.join(cmd)}")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        progress("CPU baseline: oracle K1 depth 2, 4 + 4 Euler steps and 2 TTA steps on the host cores ...")
        cpu_base = cpu_baseline(args.num_inference_steps)
        progress(f"CPU baseline done: {cpu_base['k1_step_s_cfg_on_depth2']} s per CFG step at depth 2")

    from lcv_hip import lib, ops
    lib.call("lcv_device_check")
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
    from longcat_video.pipeline_longcat_video import LongCatVideoPipeline

    T, h, w, desc = WORKLOADS[args.workload]
    dit = LongCatVideoTransformer3DModel(device=dev, dtype=torch.bfloat16, depth=args.depth).eval()
    dit.init_synthetic_(seed=1234)
    progress(f"DiT built ({args.depth} blocks, synthetic weights); warm-up: {args.warmup} step(s) of {args.workload}")
    pipe = LongCatVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(), dit=dit)
    pipe.device = dev
    sp = args.parallelism == "sp" and world > 1
    if sp:
        dit.enable_sequence_parallel(None)

    # synthetic inputs: one independent video per rank (data parallel), seeded per rank
    g = torch.Generator(device=dev).manual_seed(42 + (0 if sp else rank))  # sp: every rank holds the same video
    latents = torch.randn((1, 16, T, h, w), generator=g, device=dev, dtype=torch.float32)
    g2 = torch.Generator(device=dev).manual_seed(43)
    pe = torch.randn((1, 1, 512, 4096), generator=g2, device=dev, dtype=torch.float32).to(torch.bfloat16)
    ne = torch.randn((1, 1, 512, 4096), generator=g2, device=dev, dtype=torch.float32).to(torch.bfloat16)
    pm = torch.zeros((1, 512), dtype=torch.int64, device=dev)
    pm[:, :77] = 1
    nm = torch.zeros((1, 512), dtype=torch.int64, device=dev)
    nm[:, :77] = 1

    def run(start, stop, x):
        return pipe.denoise(x, pe, pm, ne, nm, num_cond_latents=0, num_inference_steps=args.num_inference_steps,
                            guidance_scale=args.guidance_scale, start_step=start, stop_step=stop)

    x = latents
    x = run(0, args.warmup, x)  # W untimed warm-up steps (also builds the fused-weight copies)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    progress(f"timed region: {args.steps} CFG denoise step(s)")
    ops.PROFILE = []  # (start_event, end_event, flops, Nq, Nk, kernel name) per attention launch, recorded on the launch stream
    t0 = time.perf_counter()
    x = run(args.warmup, args.warmup + args.steps, x)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    rccl_ranks = 1
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                      # the group size as a real RCCL all-reduce sees it
        rccl_ranks = int(ones.item())
    assert torch.isfinite(x).all().item(), "non-finite latents"

    sec_per_step = elapsed / args.steps
    progress(f"timed region done: {sec_per_step:.3f} s per step")

    # ---- secondary timings (untimed region): one CFG denoise step at BASELINE configs 2, 1 and 5 (K2 49x480p, K1 16x256x256, K5 121x480p) ----
    secondary = {}
    if world == 1 and not args.no_extras and args.workload == "K3":
        for name in ("K2", "K1", "K5"):     # BASELINE.json configs 2, 1 and 5 (the 121-frame clip, on one GPU)
            T2, h2, w2, _ = WORKLOADS[name]
            lat2 = torch.randn((1, 16, T2, h2, w2), generator=g, device=dev, dtype=torch.float32)
            run2 = lambda a, b_, x_: pipe.denoise(x_, pe, pm, ne, nm, num_cond_latents=0, num_inference_steps=args.num_inference_steps,
                                                  guidance_scale=args.guidance_scale, start_step=a, stop_step=b_)
            x2 = run2(0, 1, lat2)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            x2 = run2(1, 3, x2)
            torch.cuda.synchronize()
            secondary[f"{name}_cfg_step_s"] = round((time.perf_counter() - t1) / 2, 4)
            progress(f"secondary {name}: {secondary[f'{name}_cfg_step_s']} s per CFG step")
            assert torch.isfinite(x2).all().item()
        if not args.no_k3p and args.depth == 48:
            # north_star's literal shape, once and for the record (the headline stays K3): ONE CFG step at 176 400 tokens, no
            # separate warm-up (kernels and weight copies are warm; a second ~45 s step would push the default run past its budget)
            T2, h2, w2, _ = WORKLOADS["K3p"]
            lat2 = torch.randn((1, 16, T2, h2, w2), generator=g, device=dev, dtype=torch.float32)
            progress("secondary K3p: one CFG step at 176 400 tokens (~45 s) ...")
            torch.cuda.synchronize(); t1 = time.perf_counter()
            x2 = pipe.denoise(lat2, pe, pm, ne, nm, num_cond_latents=0, num_inference_steps=args.num_inference_steps,
                              guidance_scale=args.guidance_scale, start_step=0, stop_step=1)
            torch.cuda.synchronize()
            secondary["K3p_cfg_step_s"] = round(time.perf_counter() - t1, 3)
            secondary["K3p_tokens"] = T2 * (h2 // 2) * (w2 // 2)
            progress(f"secondary K3p: {secondary['K3p_cfg_step_s']} s per CFG step")
            assert torch.isfinite(x2).all().item()
            del lat2, x2
            torch.cuda.empty_cache()

    # ---- extras (outside the timed region, rank 0 of a 1-GPU run): the other two legs of "wall-clock per TTA video" ----
    extras = dict(secondary)
    if world == 1 and not args.no_extras and args.depth == 48:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):   # stdout carries exactly one JSON line
            progress("extras 1/2: the reference's 480p operating point (generation + LoRA TTA)")
            ref_point = measure_reference_point(dit, dev, pe, pm, ne, nm)
            progress("extras 2/2: VAE decode, then 1 + 20 LoRA-TTA inner steps at the bench resolution, then 20 more with early stopping")
            extras.update(measure_extras(dit, dev, T, h, w, pe, pm))
            if args.full_tta_leg:
                progress("extras (opt-in): full-model TTA at the 480p operating point")
                ref_point.update(measure_full_tta_reference_point(dit, dev, pe, pm))
            progress("extras done")
        extras["reference_operating_point_480p_14c14g"] = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ref_point.items()}
        # 20 measured inner steps + (50 denoise steps at the measured step time) + measured decode
        extras["wall_clock_per_tta_video_s"] = (
            extras["tta20_s"] + args.num_inference_steps * sec_per_step + extras["vae_decode_s"])
    per_gpu = T / (args.num_inference_steps * sec_per_step)
    value = per_gpu if sp else per_gpu * world

    # roofline of the dominant kernel (self-attention forward launches only: Nq = Nk = all tokens)
    fmax = max((f for (_, _, f, _, _, _) in prof), default=0.0)   # the self-attention launches (largest flops)
    big = [(s.elapsed_time(e), f) for (s, e, f, nq, nk, kn) in prof if f == fmax]
    kernels = sorted({kn for (_, _, f, _, _, kn) in prof if f == fmax})   # what the library actually launched for them
    avg_ms = sum(m for m, _ in big) / max(len(big), 1)
    flops = big[0][1] if big else 0.0
    achieved = flops / (avg_ms * 1e-3) / 1e12 if big else 0.0
    traffic = None
    pmc = ROOT / "profiles" / "attn_fwd_hbm_traffic.json"
    if pmc.exists():
        try:
            traffic = json.loads(pmc.read_text()).get(args.workload)
        except Exception:
            traffic = None
    line = {
        "metric": "denoised latent frames/sec (49x720p, 50-step CFG denoise)", "value": value,
        "unit": "latent frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sec_per_step * 1e3, "higher_is_better": True, "scaling": "strong" if sp else "weak", "vs_baseline": None,
        "rccl_ranks": rccl_ranks,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {desc}; DiT 48 blocks x 4096 hidden, CFG {args.guidance_scale}, "
                               f"{args.num_inference_steps}-step flow-match Euler, no conditioning frames",
                   "depth": args.depth, "tokens": T * (h // 2) * (w // 2), "parallelism": f"{'sp' if sp else 'dp'}{world}",
                   "latent_frame_steps_per_s": T * (1 if sp else world) / sec_per_step,
                   "wall_clock_per_video_s_extrapolated": sec_per_step * args.num_inference_steps, **extras},
        "roofline": {"kernel": " | ".join(kernels) if kernels else "none", "bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": traffic,
                     "avg_launch_ms": avg_ms, "launches": len(big), "flops_per_launch": flops},
    }
    if cpu_base is not None:
        line["cpu_baseline"] = cpu_base
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
