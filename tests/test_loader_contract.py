"""CPU: the loader contract of SURVEY §8 row a17 — the five `from_pretrained(checkpoint_dir, subfolder=...)` calls of the
reference's `load_longcat_components` (delta_experiment/scripts/common.py:46-96) against a TINY checkpoint directory written
here in the diffusers / transformers layout the real checkpoint has (`<ckpt>/{tokenizer,text_encoder,vae,scheduler,dit}/`,
common.py:59-74; download recipe baseline_experiment/sbatch/download_model.sbatch:40-42):

  dit/config.json with `_class_name` / `_diffusers_version`, LIST-valued `patch_size`, keys this build does not know;
  dit/ weights in TWO *.safetensors shards + `diffusion_pytorch_model.safetensors.index.json`;
  vae/config.json + weights; scheduler/scheduler_config.json with a `shift`; text_encoder/ (UMT5 config + weights);
  tokenizer/ (a `tokenizers` file the installed transformers can open).

No kernel runs (modules are built on the CPU and only their parameters are inspected); the forward paths over loaded weights
are GPU-tested elsewhere (test_gpu_dit.py loads the same state-dict names from the oracle's parameter set)."""
import json
import os

import pytest
import torch

BF16 = torch.bfloat16


def _save_sharded(state, folder, n_shards=2, stem="diffusion_pytorch_model"):
    from safetensors.torch import save_file
    keys = sorted(state)
    per = (len(keys) + n_shards - 1) // n_shards
    weight_map = {}
    for i in range(n_shards):
        part = {k: state[k].contiguous() for k in keys[i * per:(i + 1) * per]}
        name = f"{stem}-{i + 1:05d}-of-{n_shards:05d}.safetensors"
        save_file(part, os.path.join(folder, name))
        weight_map.update({k: name for k in part})
    with open(os.path.join(folder, f"{stem}.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": sum(v.numel() * v.element_size() for v in state.values())},
                   "weight_map": weight_map}, f)
    return weight_map


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    from longcat_video.modules.umt5_encoder import UMT5EncoderModel
    from safetensors.torch import save_file
    root = tmp_path_factory.mktemp("ckpt")
    for sub in ("dit", "vae", "scheduler", "text_encoder", "tokenizer"):
        os.makedirs(root / sub)
    g = torch.Generator().manual_seed(0)
    # --- dit: fp32 on disk (the loader casts to torch_dtype), two shards + index
    dit_cfg = dict(hidden_size=256, depth=2, num_heads=2, caption_channels=64, adaln_tembed_dim=64, in_channels=16, out_channels=16,
                   mlp_ratio=4, frequency_embedding_size=256, text_tokens_zero_pad=False)
    src = LongCatVideoTransformer3DModel(device="cpu", dtype=torch.float32, **dit_cfg)
    dit_state = {k: torch.randn(v.shape, generator=g) * 0.02 for k, v in src.state_dict().items()}
    _save_sharded(dit_state, str(root / "dit"))
    with open(root / "dit" / "config.json", "w") as f:
        json.dump({"_class_name": "LongCatVideoTransformer3DModel", "_diffusers_version": "0.35.1", **dit_cfg,
                   "patch_size": [1, 2, 2], "enable_flashattn3": False, "enable_xformers": False, "enable_bsa": False,
                   "bsa_params": None, "cp_split_hw": None, "a_key_from_a_future_upstream": 1}, f)
    # --- vae
    vae_cfg = dict(base_dim=16, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[], temperal_downsample=[False, True, True],
                   dropout=0.0, latents_mean=[0.1 * i for i in range(16)], latents_std=[1.0 + 0.05 * i for i in range(16)])
    vsrc = AutoencoderKLWan(device="cpu", dtype=torch.float32, **vae_cfg)
    vae_state = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in vsrc.state_dict().items()}
    save_file(vae_state, str(root / "vae" / "diffusion_pytorch_model.safetensors"))
    with open(root / "vae" / "config.json", "w") as f:
        json.dump({"_class_name": "AutoencoderKLWan", "_diffusers_version": "0.35.1", **vae_cfg}, f)
    # --- scheduler
    with open(root / "scheduler" / "scheduler_config.json", "w") as f:
        json.dump({"_class_name": "FlowMatchEulerDiscreteScheduler", "_diffusers_version": "0.35.1", "num_train_timesteps": 1000,
                   "shift": 7.0, "use_dynamic_shifting": False}, f)
    # --- text encoder (transformers layout: config.json + model.safetensors)
    te_cfg = dict(vocab_size=64, d_model=64, d_kv=64, d_ff=128, num_layers=1, num_heads=1)
    tsrc = UMT5EncoderModel(device="cpu", dtype=torch.float32, **te_cfg)
    te_state = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in tsrc.state_dict().items()}
    save_file(te_state, str(root / "text_encoder" / "model.safetensors"))
    with open(root / "text_encoder" / "config.json", "w") as f:
        json.dump({"architectures": ["UMT5EncoderModel"], "model_type": "umt5", **te_cfg, "feed_forward_proj": "gated-gelu",
                   "relative_attention_num_buckets": 32, "relative_attention_max_distance": 128, "layer_norm_epsilon": 1e-6}, f)
    # --- tokenizer: a word-level `tokenizers` file
    from tokenizers import Tokenizer, models, pre_tokenizers
    vocab = {"<pad>": 0, "</s>": 1, "<unk>": 2, **{w: 3 + i for i, w in enumerate("a red kite flies over the sea".split())}}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.save(str(root / "tokenizer" / "tokenizer.json"))
    with open(root / "tokenizer" / "tokenizer_config.json", "w") as f:
        json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "pad_token": "<pad>", "eos_token": "</s>", "unk_token": "<unk>",
                   "model_max_length": 512}, f)
    return dict(root=str(root), dit_state=dit_state, vae_state=vae_state, te_state=te_state, dit_cfg=dit_cfg, vae_cfg=vae_cfg)


def test_load_longcat_components_reads_the_reference_layout(ckpt):
    from tta import common as C
    comp = C.load_longcat_components(ckpt["root"], device="cpu", dtype=BF16)
    assert set(comp) == {"tokenizer", "text_encoder", "vae", "scheduler", "dit", "pipe"}      # common.py:89-96
    dit, vae, sched, te, pipe = comp["dit"], comp["vae"], comp["scheduler"], comp["text_encoder"], comp["pipe"]
    # dit: config surface the reference reads (common.py:262-271, run_delta_a.py:147-149, 475), dtype, every tensor from the shards
    assert dit.config.patch_size == (1, 2, 2) and dit.patch_size == (1, 2, 2)
    assert (dit.config.hidden_size, dit.config.adaln_tembed_dim, dit.config.out_channels, len(dit.blocks)) == (256, 64, 16, 2)
    assert not hasattr(dit.config, "a_key_from_a_future_upstream")
    got = dit.state_dict()
    assert set(got) == set(ckpt["dit_state"])
    for k, v in ckpt["dit_state"].items():
        assert got[k].dtype == BF16 and torch.equal(got[k], v.to(BF16)), k
    assert dit.x_embedder.proj.weight.dtype == BF16 and dit.text_tokens_zero_pad is False
    # vae: per-channel statistics the latent (de)normalisation uses (common.py:177-206), dtype
    assert vae.config.z_dim == 16 and vae.config.latents_mean == ckpt["vae_cfg"]["latents_mean"]
    assert vae.config.latents_std == ckpt["vae_cfg"]["latents_std"] and vae.dtype == BF16
    vgot = vae.state_dict()
    for k, v in ckpt["vae_state"].items():
        assert torch.equal(vgot[k], v.to(BF16)), k
    # scheduler: the static shift of the checkpoint reaches the sigma grid
    assert sched.shift == 7.0 and sched.config.num_train_timesteps == 1000
    sched.set_timesteps(4, sigmas=pipe.get_timesteps_sigmas(4))
    base = torch.linspace(1, 0.001, 4)
    assert torch.allclose(sched.sigmas[:4].cpu(), 7 * base / (1 + 6 * base)) and float(sched.sigmas[4]) == 0.0
    # text encoder + tokenizer
    tgot = te.state_dict()
    for k, v in ckpt["te_state"].items():
        assert torch.equal(tgot[k], v.to(BF16)), k
    ids = comp["tokenizer"](["a red kite"], padding="max_length", max_length=16, truncation=True, return_tensors="pt")
    assert ids.input_ids.shape == (1, 16) and int(ids.attention_mask.sum()) >= 3
    assert pipe.dit is dit and pipe.vae is vae and pipe.scheduler is sched and pipe.text_encoder is te


def test_dit_from_pretrained_keyword_surface_and_error_paths(ckpt, tmp_path):
    import shutil
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel as DiT
    from safetensors.torch import load_file, save_file
    # the exact call of common.py:71-74
    m = DiT.from_pretrained(ckpt["root"], subfolder="dit", cp_split_hw=[1, 1], enable_flashattn2=True, torch_dtype=BF16)
    assert m.config.cp_split_hw == (1, 1) and next(m.parameters()).dtype == BF16
    with pytest.raises(NotImplementedError):            # a spatial context-parallel split is not this build's parallelism
        DiT.from_pretrained(ckpt["root"], subfolder="dit", cp_split_hw=[2, 1], torch_dtype=BF16)
    # a checkpoint that lacks tensors is refused by name, not half-loaded
    broken = tmp_path / "broken"
    shutil.copytree(os.path.join(ckpt["root"], "dit"), broken / "dit")
    shard = sorted(f for f in os.listdir(broken / "dit") if f.endswith(".safetensors"))[0]
    st = load_file(str(broken / "dit" / shard))
    dropped = sorted(st)[0]
    st.pop(dropped)
    save_file(st, str(broken / "dit" / shard))
    with pytest.raises(RuntimeError, match="missing"):
        DiT.from_pretrained(str(broken), subfolder="dit", torch_dtype=BF16)
    # an index that names a shard which is not there
    os.remove(broken / "dit" / shard)
    with pytest.raises(FileNotFoundError):
        DiT.from_pretrained(str(broken), subfolder="dit", torch_dtype=BF16)
    # no weights at all
    empty = tmp_path / "empty" / "dit"
    os.makedirs(empty)
    shutil.copy(os.path.join(ckpt["root"], "dit", "config.json"), empty / "config.json")
    with pytest.raises(FileNotFoundError):
        DiT.from_pretrained(str(tmp_path / "empty"), subfolder="dit", torch_dtype=BF16)


def test_vae_and_scheduler_loaders_alone(ckpt):
    from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
    from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
    vae = AutoencoderKLWan.from_pretrained(ckpt["root"], subfolder="vae", torch_dtype=BF16)          # common.py:65-67
    assert vae.dtype == BF16 and vae._has_encoder
    s = FlowMatchEulerDiscreteScheduler.from_pretrained(ckpt["root"], subfolder="scheduler", torch_dtype=BF16)   # :68-70
    assert s.shift == 7.0
    s0 = FlowMatchEulerDiscreteScheduler()                      # no checkpoint: shift 1 (the bench's synthetic runs)
    assert s0.shift == 1.0
