"""CPU oracle for the WAN-style causal 3-D VAE decoder and encoder (TEST INFRASTRUCTURE — never imported by the product).

**Parity unpinned**: `AutoencoderKLWan` belongs to the un-vendored upstream package; the reference
only fixes the contract around it — `vae.decode(z.to(vae.dtype), return_dict=False)[0]` in [-1, 1],
`(v + 1) / 2` then clamp, `[1,16,T,h,w] -> [1,3,1+4(T-1),8h,8w]` (delta_experiment/scripts/common.py:209-221) and the
latent (de)normalisation (:177-206).  The graph below restates the published WAN-2.1 decoder
[assumed-from-upstream]: base 96, dim_mult [1,2,4,4], 2(+1) residual blocks per stage, single-head per-frame mid
attention, causal 3x3x3 convs, channel RMS norm, nearest-exact 2x upsample + 3x3 conv, temporal 2x upsample by a
(3,1,1) causal conv to 2C channels whose halves become consecutive frames — with the FIRST latent frame exempt from
temporal upsampling (it decodes to 1 frame, every later latent frame to 4).

The encoder (second half of this file) restates the published WAN-2.1 encoder the same way [assumed-from-upstream]:
conv_in 3 -> 96, per stage 2 residual blocks then a downsampler (ZeroPad2d((0,1,0,1)) + 3x3 stride-2 conv per frame; for
"downsample3d" followed by a (3,1,1) stride-2 temporal conv from which the FIRST frame is exempt), mid block, RMS norm + SiLU,
conv_out -> 2 z_dim, 1x1x1 quant_conv, posterior mean = first z_dim channels (the reference takes the mode:
`retrieve_latents(vae.encode(x))`, delta_experiment/scripts/common.py:158-174).  `[1,3,1+4k,8h,8w] -> [1,16,1+k,h,w]`.

Two formulations are given and must agree (tests/test_vae_oracle.py), for the decoder and for the encoder:
  * `decode_chunked`  — frame-by-frame with feature caches, the way the upstream module streams it;
  * `decode_full`     — the whole sequence at once with causal zero padding (what the HIP decoder implements; a
                        288 GB device has no need for the chunk loop).
"""
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

CACHE_T = 2


def default_config(base_dim=96, z_dim=16):
    return dict(base_dim=base_dim, z_dim=z_dim, dim_mult=[1, 2, 4, 4], num_res_blocks=2,
                temperal_downsample=[False, True, True])


def decoder_plan(cfg):
    """[(in_dim, out_dim, n_res, upsample_mode)] per up block + the channel list."""
    dim, mult = cfg["base_dim"], cfg["dim_mult"]
    dims = [dim * u for u in [mult[-1]] + mult[::-1]]
    t_up = cfg["temperal_downsample"][::-1]
    plan = []
    for i, (i_d, o_d) in enumerate(zip(dims[:-1], dims[1:])):
        if i > 0:
            i_d = i_d // 2
        mode = None
        if i != len(mult) - 1:
            mode = "upsample3d" if t_up[i] else "upsample2d"
        plan.append((i_d, o_d, cfg["num_res_blocks"] + 1, mode))
    return dims, plan


def make_params(cfg, seed=0, std=0.05, dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    P = {}

    def conv(name, co, ci, *k):
        fan = ci
        for kk in k:
            fan *= kk
        P[name + ".weight"] = (torch.randn(co, ci, *k, generator=g) * (1.0 / fan ** 0.5)).to(dtype)
        P[name + ".bias"] = (torch.randn(co, generator=g) * std).to(dtype)

    def norm(name, c, images=False):
        shape = (c, 1, 1) if images else (c, 1, 1, 1)
        P[name + ".gamma"] = (1.0 + 0.1 * torch.randn(shape, generator=g)).to(dtype)

    def res(name, ci, co):
        norm(name + ".norm1", ci); conv(name + ".conv1", co, ci, 3, 3, 3)
        norm(name + ".norm2", co); conv(name + ".conv2", co, co, 3, 3, 3)
        if ci != co:
            conv(name + ".conv_shortcut", co, ci, 1, 1, 1)

    z = cfg["z_dim"]
    dims, plan = decoder_plan(cfg)
    conv("post_quant_conv", z, z, 1, 1, 1)
    conv("decoder.conv_in", dims[0], z, 3, 3, 3)
    res("decoder.mid_block.resnets.0", dims[0], dims[0])
    norm("decoder.mid_block.attentions.0.norm", dims[0], images=True)
    conv("decoder.mid_block.attentions.0.to_qkv", 3 * dims[0], dims[0], 1, 1)
    conv("decoder.mid_block.attentions.0.proj", dims[0], dims[0], 1, 1)
    res("decoder.mid_block.resnets.1", dims[0], dims[0])
    for i, (ci, co, n_res, mode) in enumerate(plan):
        c = ci
        for j in range(n_res):
            res(f"decoder.up_blocks.{i}.resnets.{j}", c, co)
            c = co
        if mode is not None:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.resample.1", co // 2, co, 3, 3)
            if mode == "upsample3d":
                conv(f"decoder.up_blocks.{i}.upsamplers.0.time_conv", 2 * co, co, 3, 1, 1)
    norm("decoder.norm_out", dims[-1])
    conv("decoder.conv_out", 3, dims[-1], 3, 3, 3)
    return P


# --------------------------------------------------------------------------- primitives (NCDHW, fp32 math)
def rms_norm(x, gamma, channel_dim=1):
    c = x.shape[channel_dim]
    return F.normalize(x, dim=channel_dim) * (c ** 0.5) * gamma.float()


def causal_conv3d(x, w, b, cache=None):
    """WanCausalConv3d: 2*pad_t frames in front (from `cache` when given, zeros otherwise), symmetric spatial pad."""
    kt, kh, kw = w.shape[2:]
    pad_t = kt - 1
    if cache is not None and pad_t > 0:
        x = torch.cat([cache, x], dim=2)
        pad_t -= cache.shape[2]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, pad_t, 0))
    return F.conv3d(x, w.float(), b.float())


def _bf(t, rnd):
    return t.to(torch.bfloat16).float() if rnd else t


def res_block_full(P, name, x, rnd):
    h = x
    if name + ".conv_shortcut.weight" in P:
        h = _bf(causal_conv3d(x, P[name + ".conv_shortcut.weight"], P[name + ".conv_shortcut.bias"]), rnd)
    y = _bf(F.silu(rms_norm(x, P[name + ".norm1.gamma"])), rnd)
    y = _bf(causal_conv3d(y, P[name + ".conv1.weight"], P[name + ".conv1.bias"]), rnd)
    y = _bf(F.silu(rms_norm(y, P[name + ".norm2.gamma"])), rnd)
    y = _bf(causal_conv3d(y, P[name + ".conv2.weight"], P[name + ".conv2.bias"]), rnd)
    return _bf(y + h, rnd)


def mid_attention(P, name, x, rnd):
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = _bf(rms_norm(y, P[name + ".norm.gamma"]), rnd)
    qkv = _bf(F.conv2d(y, P[name + ".to_qkv.weight"].float(), P[name + ".to_qkv.bias"].float()), rnd)
    qkv = qkv.reshape(b * t, 1, c * 3, -1).permute(0, 1, 3, 2)
    q, k, v = qkv.chunk(3, dim=-1)
    s = (q @ k.transpose(-1, -2)) * (c ** -0.5)
    o = _bf(torch.softmax(s, dim=-1) @ v, rnd)
    o = o.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
    o = _bf(F.conv2d(o, P[name + ".proj.weight"].float(), P[name + ".proj.bias"].float()), rnd)
    o = o.view(b, t, c, h, w).permute(0, 2, 1, 3, 4)
    return _bf(o + x, rnd)


def spatial_up_conv(P, name, x, rnd):
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.interpolate(y, scale_factor=(2.0, 2.0), mode="nearest-exact")
    y = _bf(F.conv2d(y, P[name + ".resample.1.weight"].float(), P[name + ".resample.1.bias"].float(), padding=1), rnd)
    return y.view(b, t, y.shape[1], 2 * h, 2 * w).permute(0, 2, 1, 3, 4)


def _interleave(y, c):
    b, _, t, h, w = y.shape
    y = y.reshape(b, 2, c, t, h, w)
    return torch.stack((y[:, 0], y[:, 1]), 3).reshape(b, c, 2 * t, h, w)


def decode_full(P, cfg, z, rnd=False):
    """Whole-sequence decode: z [B, z_dim, T, h, w] -> [B, 3, 1 + 4(T-1), 8h, 8w], clamped to [-1, 1]."""
    dims, plan = decoder_plan(cfg)
    x = _bf(z.float(), rnd)
    x = _bf(causal_conv3d(x, P["post_quant_conv.weight"], P["post_quant_conv.bias"]), rnd)
    x = _bf(causal_conv3d(x, P["decoder.conv_in.weight"], P["decoder.conv_in.bias"]), rnd)
    x = res_block_full(P, "decoder.mid_block.resnets.0", x, rnd)
    x = mid_attention(P, "decoder.mid_block.attentions.0", x, rnd)
    x = res_block_full(P, "decoder.mid_block.resnets.1", x, rnd)
    for i, (ci, co, n_res, mode) in enumerate(plan):
        for j in range(n_res):
            x = res_block_full(P, f"decoder.up_blocks.{i}.resnets.{j}", x, rnd)
        if mode == "upsample3d":
            name = f"decoder.up_blocks.{i}.upsamplers.0"
            first, rest = x[:, :, :1], x[:, :, 1:]
            if rest.shape[2] > 0:  # the first latent frame is exempt from temporal upsampling
                y = _bf(causal_conv3d(rest, P[name + ".time_conv.weight"], P[name + ".time_conv.bias"]), rnd)
                x = torch.cat([first, _interleave(y, co)], dim=2)
            x = spatial_up_conv(P, name, x, rnd)
        elif mode == "upsample2d":
            x = spatial_up_conv(P, f"decoder.up_blocks.{i}.upsamplers.0", x, rnd)
    x = _bf(F.silu(rms_norm(x, P["decoder.norm_out.gamma"])), rnd)
    x = _bf(causal_conv3d(x, P["decoder.conv_out.weight"], P["decoder.conv_out.bias"]), rnd)
    return x.clamp(-1.0, 1.0)


# --------------------------------------------------------------------------- chunked (feature-cache) formulation
def _conv_cached(P, name, x, cache: list, idx: list):
    """A causal conv fed one chunk at a time: its cache holds the last CACHE_T input frames seen so far."""
    i = idx[0]
    cache_x = x[:, :, -CACHE_T:].clone()
    if cache_x.shape[2] < 2 and cache[i] is not None:
        cache_x = torch.cat([cache[i][:, :, -1:], cache_x], dim=2)
    y = causal_conv3d(x, P[name + ".weight"], P[name + ".bias"], cache[i])
    cache[i] = cache_x
    idx[0] += 1
    return y


def _res_block_chunk(P, name, x, cache, idx):
    h = x
    if name + ".conv_shortcut.weight" in P:
        h = causal_conv3d(x, P[name + ".conv_shortcut.weight"], P[name + ".conv_shortcut.bias"])
    y = F.silu(rms_norm(x, P[name + ".norm1.gamma"]))
    y = _conv_cached(P, name + ".conv1", y, cache, idx)
    y = F.silu(rms_norm(y, P[name + ".norm2.gamma"]))
    y = _conv_cached(P, name + ".conv2", y, cache, idx)
    return y + h


def decode_chunked(P, cfg, z):
    """One latent frame per call through the decoder with per-conv feature caches (fp32, no bf16 rounding)."""
    dims, plan = decoder_plan(cfg)
    cache: List[Optional[object]] = [None] * 64
    x_all = causal_conv3d(z.float(), P["post_quant_conv.weight"], P["post_quant_conv.bias"])
    outs = []
    for f in range(z.shape[2]):
        idx = [0]
        x = x_all[:, :, f:f + 1]
        x = _conv_cached(P, "decoder.conv_in", x, cache, idx)
        x = _res_block_chunk(P, "decoder.mid_block.resnets.0", x, cache, idx)
        x = mid_attention(P, "decoder.mid_block.attentions.0", x, False)
        x = _res_block_chunk(P, "decoder.mid_block.resnets.1", x, cache, idx)
        for i, (ci, co, n_res, mode) in enumerate(plan):
            for j in range(n_res):
                x = _res_block_chunk(P, f"decoder.up_blocks.{i}.resnets.{j}", x, cache, idx)
            if mode == "upsample3d":
                name = f"decoder.up_blocks.{i}.upsamplers.0"
                k = idx[0]
                if cache[k] is None:
                    cache[k] = "Rep"  # first chunk: no temporal upsampling, nothing cached
                    idx[0] += 1
                else:
                    cache_x = x[:, :, -CACHE_T:].clone()
                    if cache_x.shape[2] < 2 and not isinstance(cache[k], str):
                        cache_x = torch.cat([cache[k][:, :, -1:], cache_x], dim=2)
                    if cache_x.shape[2] < 2 and isinstance(cache[k], str):
                        cache_x = torch.cat([torch.zeros_like(cache_x), cache_x], dim=2)
                    prev = None if isinstance(cache[k], str) else cache[k]
                    y = causal_conv3d(x, P[name + ".time_conv.weight"], P[name + ".time_conv.bias"], prev)
                    cache[k] = cache_x
                    idx[0] += 1
                    x = _interleave(y, co)
                x = spatial_up_conv(P, name, x, False)
            elif mode == "upsample2d":
                x = spatial_up_conv(P, f"decoder.up_blocks.{i}.upsamplers.0", x, False)
        x = F.silu(rms_norm(x, P["decoder.norm_out.gamma"]))
        x = _conv_cached(P, "decoder.conv_out", x, cache, idx)
        outs.append(x)
    return torch.cat(outs, dim=2).clamp(-1.0, 1.0)


# =========================================================================== encoder
def encoder_plan(cfg):
    """Flat list of ("res", in_dim, out_dim) / ("down2d" | "down3d", dim) entries = upstream `encoder.down_blocks`."""
    dim, mult = cfg["base_dim"], cfg["dim_mult"]
    dims = [dim * u for u in [1] + mult]
    t_down = cfg["temperal_downsample"]
    plan = []
    for i, (i_d, o_d) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(cfg["num_res_blocks"]):
            plan.append(("res", i_d, o_d))
            i_d = o_d
        if i != len(mult) - 1:
            plan.append(("down3d" if t_down[i] else "down2d", o_d))
    return dims, plan


def make_encoder_params(cfg, seed=1, std=0.05, dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    P = {}

    def conv(name, co, ci, *k):
        fan = ci
        for kk in k:
            fan *= kk
        P[name + ".weight"] = (torch.randn(co, ci, *k, generator=g) * (1.0 / fan ** 0.5)).to(dtype)
        P[name + ".bias"] = (torch.randn(co, generator=g) * std).to(dtype)

    def norm(name, c, images=False):
        shape = (c, 1, 1) if images else (c, 1, 1, 1)
        P[name + ".gamma"] = (1.0 + 0.1 * torch.randn(shape, generator=g)).to(dtype)

    def res(name, ci, co):
        norm(name + ".norm1", ci); conv(name + ".conv1", co, ci, 3, 3, 3)
        norm(name + ".norm2", co); conv(name + ".conv2", co, co, 3, 3, 3)
        if ci != co:
            conv(name + ".conv_shortcut", co, ci, 1, 1, 1)

    z = cfg["z_dim"]
    dims, plan = encoder_plan(cfg)
    conv("encoder.conv_in", dims[0], 3, 3, 3, 3)
    for k, item in enumerate(plan):
        name = f"encoder.down_blocks.{k}"
        if item[0] == "res":
            res(name, item[1], item[2])
        else:
            conv(name + ".resample.1", item[1], item[1], 3, 3)
            if item[0] == "down3d":
                conv(name + ".time_conv", item[1], item[1], 3, 1, 1)
    top = dims[-1]
    res("encoder.mid_block.resnets.0", top, top)
    norm("encoder.mid_block.attentions.0.norm", top, images=True)
    conv("encoder.mid_block.attentions.0.to_qkv", 3 * top, top, 1, 1)
    conv("encoder.mid_block.attentions.0.proj", top, top, 1, 1)
    res("encoder.mid_block.resnets.1", top, top)
    norm("encoder.norm_out", top)
    conv("encoder.conv_out", 2 * z, top, 3, 3, 3)
    conv("quant_conv", 2 * z, 2 * z, 1, 1, 1)
    return P


def spatial_down_conv(P, name, x, rnd):
    """ZeroPad2d((0, 1, 0, 1)) + Conv2d(3x3, stride 2) on every frame."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.pad(y, (0, 1, 0, 1))
    y = _bf(F.conv2d(y, P[name + ".resample.1.weight"].float(), P[name + ".resample.1.bias"].float(), stride=2), rnd)
    return y.view(b, t, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)


def temporal_down_full(P, name, x, rnd):
    """Whole-sequence form of the chunked temporal downsample: frame 0 passes, frame t >= 1 = conv(x[2t-2 .. 2t])."""
    if x.shape[2] == 1:
        return x
    w, b = P[name + ".time_conv.weight"].float(), P[name + ".time_conv.bias"].float()
    y = _bf(F.conv3d(x, w, b, stride=(2, 1, 1)), rnd)   # windows start at frames 0, 2, 4, ...
    return torch.cat([x[:, :, :1], y], dim=2)


def encode_full(P, cfg, video, rnd=False):
    """Whole-sequence encode: video [B, 3, 1+4k, H, W] in [-1, 1] -> posterior mean [B, z_dim, 1+k, H/8, W/8]."""
    dims, plan = encoder_plan(cfg)
    x = _bf(video.float(), rnd)
    x = _bf(causal_conv3d(x, P["encoder.conv_in.weight"], P["encoder.conv_in.bias"]), rnd)
    for k, item in enumerate(plan):
        name = f"encoder.down_blocks.{k}"
        if item[0] == "res":
            x = res_block_full(P, name, x, rnd)
        else:
            x = spatial_down_conv(P, name, x, rnd)
            if item[0] == "down3d":
                x = temporal_down_full(P, name, x, rnd)
    x = res_block_full(P, "encoder.mid_block.resnets.0", x, rnd)
    x = mid_attention(P, "encoder.mid_block.attentions.0", x, rnd)
    x = res_block_full(P, "encoder.mid_block.resnets.1", x, rnd)
    x = _bf(F.silu(rms_norm(x, P["encoder.norm_out.gamma"])), rnd)
    x = _bf(causal_conv3d(x, P["encoder.conv_out.weight"], P["encoder.conv_out.bias"]), rnd)
    x = _bf(causal_conv3d(x, P["quant_conv.weight"], P["quant_conv.bias"]), rnd)
    return x[:, : cfg["z_dim"]]


def encode_chunked(P, cfg, video):
    """The upstream streaming order: frame 0 alone, then 4 frames per call, per-conv feature caches (fp32)."""
    dims, plan = encoder_plan(cfg)
    cache: List[Optional[object]] = [None] * 64
    T = video.shape[2]
    outs = []
    for it in range(1 + (T - 1) // 4):
        idx = [0]
        x = video[:, :, :1].float() if it == 0 else video[:, :, 1 + 4 * (it - 1):1 + 4 * it].float()
        x = _conv_cached(P, "encoder.conv_in", x, cache, idx)
        for k, item in enumerate(plan):
            name = f"encoder.down_blocks.{k}"
            if item[0] == "res":
                x = _res_block_chunk(P, name, x, cache, idx)
            else:
                x = spatial_down_conv(P, name, x, False)
                if item[0] == "down3d":
                    j = idx[0]
                    if cache[j] is None:
                        cache[j] = x.clone()          # first chunk: cached, not convolved
                    else:
                        last = x[:, :, -1:].clone()
                        xin = torch.cat([cache[j][:, :, -1:], x], dim=2)
                        x = F.conv3d(xin, P[name + ".time_conv.weight"].float(), P[name + ".time_conv.bias"].float(),
                                     stride=(2, 1, 1))
                        cache[j] = last
                    idx[0] += 1
        x = _res_block_chunk(P, "encoder.mid_block.resnets.0", x, cache, idx)
        x = mid_attention(P, "encoder.mid_block.attentions.0", x, False)
        x = _res_block_chunk(P, "encoder.mid_block.resnets.1", x, cache, idx)
        x = F.silu(rms_norm(x, P["encoder.norm_out.gamma"]))
        x = _conv_cached(P, "encoder.conv_out", x, cache, idx)
        outs.append(x)
    x = torch.cat(outs, dim=2)
    x = causal_conv3d(x, P["quant_conv.weight"], P["quant_conv.bias"])
    return x[:, : cfg["z_dim"]]
