"""WAN-style causal 3-D VAE (decoder and encoder) on the gfx950 kernels (whole sequence at once, channels-last).

Contract kept for the reference (delta_experiment/scripts/common.py:65-67, 158-221): `AutoencoderKLWan.from_pretrained(dir,
subfolder="vae", torch_dtype=)`, `.config.{z_dim, latents_mean, latents_std}`, `.dtype`, `.decode(z, return_dict=False)[0]`
-> `[B, 3, 1+4(T-1), 8h, 8w]` in [-1, 1], `.encode(video).latent_dist.{mode(), sample(generator)}` with video
`[B, 3, 1+4k, 8h, 8w]` in [-1, 1] -> `[B, z_dim, 1+k, h, w]` (common.py:158-174; pipeline `generate_vc` encodes the
conditioning frames with it).  Parameter names follow the diffusers module tree (`decoder.conv_in`,
`decoder.mid_block.resnets.0.norm1.gamma`, `decoder.up_blocks.i.upsamplers.0.{resample.1,time_conv}`, `post_quant_conv`)
so a published checkpoint loads by name.  [assumed-from-upstream] graph: see oracle/vae_oracle.py header.

MI355X-first choices: the upstream module streams one latent frame at a time through per-conv feature caches to fit small
GPUs; on 288 GB the whole clip is decoded in one pass (identical arithmetic: tests/test_vae_oracle.py), every causal conv is
an implicit GEMM on the MFMA core with its taps gathered by LDS-DMA (zero page for padding), the nearest 2x upsample is
folded into the following conv's gather (no upsampled tensor exists), the residual add rides in the conv epilogue, and
activations stay channels-last so a pixel row IS a GEMM row.
"""
import json
import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from lcv_hip import ops
from lcv_hip.lib import LCV_EPI_GATE_RESIDUAL, call

BF16 = torch.bfloat16

# WAN 2.1 latent statistics (diffusers AutoencoderKLWan defaults) [assumed-from-upstream]
_LATENTS_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632,
                 -0.1922, -0.9497, 0.2503, -0.2921]
_LATENTS_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382,
                1.1253, 2.8251, 1.9160]


def _pad64(c: int) -> int:
    return (c + 63) // 64 * 64


class _Conv(nn.Module):
    """Parameter holder in torch conv layout ([Cout, Cin, (kt,) kh, kw]) + a cached kernel-layout copy."""

    def __init__(self, cin, cout, k, device=None, dtype=BF16):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, tuple(k)
        self.weight = nn.Parameter(torch.empty((cout, cin) + self.k, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(cout, device=device, dtype=dtype), requires_grad=False)
        self._packed = None
        self._ver = None

    def packed(self) -> torch.Tensor:
        """[Cout, taps * Cin_pad] with K ordered (dt, dh, dw, cin), zero-padded channels."""
        if self._packed is None or self._ver != self.weight._version:
            w = self.weight.detach()
            if w.dim() == 4:
                w = w.unsqueeze(2)
            co, ci, kt, kh, kw = w.shape
            cp = _pad64(ci)
            wp = torch.zeros((co, kt, kh, kw, cp), dtype=BF16, device=w.device)
            wp[..., :ci] = w.permute(0, 2, 3, 4, 1).to(BF16)
            self._packed = wp.reshape(co, kt * kh * kw * cp).contiguous()
            self._ver = self.weight._version
        return self._packed


class _Norm(nn.Module):
    def __init__(self, dim, images=False, device=None, dtype=BF16):
        super().__init__()
        shape = (dim, 1, 1) if images else (dim, 1, 1, 1)
        self.dim = dim
        self.gamma = nn.Parameter(torch.ones(shape, device=device, dtype=dtype), requires_grad=False)

    def padded(self) -> torch.Tensor:
        """gamma zero-padded to the kernels' channel pitch; cached (two tiny launches per norm call otherwise)."""
        key = (self.gamma._version, self.gamma.data_ptr())
        if getattr(self, "_pad_key", None) != key:
            g = torch.zeros(_pad64(self.dim), dtype=BF16, device=self.gamma.device)
            g[: self.dim] = self.gamma.detach().flatten().to(BF16)
            self._padded, self._pad_key = g, key
        return self._padded


class _Res(nn.Module):
    def __init__(self, ci, co, **kw):
        super().__init__()
        self.norm1 = _Norm(ci, **kw)
        self.conv1 = _Conv(ci, co, (3, 3, 3), **kw)
        self.norm2 = _Norm(co, **kw)
        self.conv2 = _Conv(co, co, (3, 3, 3), **kw)
        self.conv_shortcut = _Conv(ci, co, (1, 1, 1), **kw) if ci != co else nn.Identity()


class _Attn(nn.Module):
    def __init__(self, dim, **kw):
        super().__init__()
        self.norm = _Norm(dim, images=True, **kw)
        self.to_qkv = _Conv(dim, 3 * dim, (1, 1), **kw)
        self.proj = _Conv(dim, dim, (1, 1), **kw)


class _Mid(nn.Module):
    def __init__(self, dim, **kw):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(dim, dim, **kw), _Res(dim, dim, **kw)])
        self.attentions = nn.ModuleList([_Attn(dim, **kw)])


class _Resample(nn.Module):
    def __init__(self, dim, mode, **kw):
        super().__init__()
        self.mode = mode
        self.resample = nn.ModuleList([nn.Identity(), _Conv(dim, dim // 2, (3, 3), **kw)])  # index 1 = the conv
        if mode == "upsample3d":
            self.time_conv = _Conv(dim, 2 * dim, (3, 1, 1), **kw)


class _DownResample(nn.Module):
    """upstream WanResample("downsample2d" | "downsample3d"): ZeroPad2d((0,1,0,1)) + 3x3 stride-2 conv per frame, then for 3d a
    (3,1,1) stride-2 temporal conv from which the first frame is exempt."""

    def __init__(self, dim, mode, **kw):
        super().__init__()
        self.mode = mode
        self.resample = nn.ModuleList([nn.Identity(), _Conv(dim, dim, (3, 3), **kw)])
        if mode == "downsample3d":
            self.time_conv = _Conv(dim, dim, (3, 1, 1), **kw)


class _Encoder(nn.Module):
    def __init__(self, dim, z_dim, dim_mult, num_res_blocks, temperal_downsample, **kw):
        super().__init__()
        dims = [dim * u for u in [1] + dim_mult]
        self.conv_in = _Conv(3, dims[0], (3, 3, 3), **kw)
        blocks = []
        for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
            for _ in range(num_res_blocks):
                blocks.append(_Res(ci, co, **kw))
                ci = co
            if i != len(dim_mult) - 1:
                blocks.append(_DownResample(co, "downsample3d" if temperal_downsample[i] else "downsample2d", **kw))
        self.down_blocks = nn.ModuleList(blocks)   # flat, as upstream names them
        self.mid_block = _Mid(dims[-1], **kw)
        self.norm_out = _Norm(dims[-1], **kw)
        self.conv_out = _Conv(dims[-1], 2 * z_dim, (3, 3, 3), **kw)


class DiagonalGaussianDistribution:
    """The posterior object the reference reads (`retrieve_latents`: `.mode()`; `.sample(generator)` kept for parity of the
    surface): parameters [B, 2 z, T, h, w] = (mean | logvar), logvar clamped to [-30, 20]."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def mode(self) -> torch.Tensor:
        return self.mean

    def sample(self, generator=None) -> torch.Tensor:
        eps = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * eps


class _UpBlock(nn.Module):
    def __init__(self, ci, co, n_res, mode, **kw):
        super().__init__()
        rs, c = [], ci
        for _ in range(n_res):
            rs.append(_Res(c, co, **kw))
            c = co
        self.resnets = nn.ModuleList(rs)
        self.upsamplers = nn.ModuleList([_Resample(co, mode, **kw)]) if mode else None


class _Decoder(nn.Module):
    def __init__(self, dim, z_dim, dim_mult, num_res_blocks, temperal_upsample, **kw):
        super().__init__()
        dims = [dim * u for u in [dim_mult[-1]] + dim_mult[::-1]]
        self.conv_in = _Conv(z_dim, dims[0], (3, 3, 3), **kw)
        self.mid_block = _Mid(dims[0], **kw)
        ups = []
        for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
            if i > 0:
                ci = ci // 2
            mode = None
            if i != len(dim_mult) - 1:
                mode = "upsample3d" if temperal_upsample[i] else "upsample2d"
            ups.append(_UpBlock(ci, co, num_res_blocks + 1, mode, **kw))
        self.up_blocks = nn.ModuleList(ups)
        self.norm_out = _Norm(dims[-1], **kw)
        self.conv_out = _Conv(dims[-1], 3, (3, 3, 3), **kw)


class AutoencoderKLWan(nn.Module):
    def __init__(self, base_dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, attn_scales=(),
                 temperal_downsample=(False, True, True), latents_mean=None, latents_std=None, device=None,
                 dtype=BF16, **unused):
        super().__init__()
        self.config = SimpleNamespace(base_dim=base_dim, z_dim=z_dim, dim_mult=list(dim_mult),
                                      num_res_blocks=num_res_blocks, temperal_downsample=list(temperal_downsample),
                                      latents_mean=list(latents_mean or _LATENTS_MEAN[:z_dim]),
                                      latents_std=list(latents_std or _LATENTS_STD[:z_dim]))
        kw = dict(device=device, dtype=dtype)
        self.post_quant_conv = _Conv(z_dim, z_dim, (1, 1, 1), **kw)
        self.decoder = _Decoder(base_dim, z_dim, list(dim_mult), num_res_blocks, list(temperal_downsample)[::-1], **kw)
        self.quant_conv = _Conv(2 * z_dim, 2 * z_dim, (1, 1, 1), **kw)
        self.encoder = _Encoder(base_dim, z_dim, list(dim_mult), num_res_blocks, list(temperal_downsample), **kw)
        self._has_encoder = True
        self._zero = None

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    @classmethod
    def from_pretrained(cls, checkpoint_dir, subfolder: Optional[str] = None, torch_dtype=BF16, device=None, **kw):
        path = os.path.join(checkpoint_dir, subfolder) if subfolder else checkpoint_dir
        with open(os.path.join(path, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        m = cls(device=device or "cpu", dtype=torch_dtype, **cfg)
        from safetensors.torch import load_file
        state = {}
        for s in sorted(x for x in os.listdir(path) if x.endswith(".safetensors")):
            state.update(load_file(os.path.join(path, s)))
        keep = {k: v for k, v in state.items() if k.startswith(("decoder.", "post_quant_conv.", "encoder.", "quant_conv."))}
        missing, _ = m.load_state_dict(keep, strict=False)
        dec_missing = [k for k in missing if k.startswith(("decoder.", "post_quant_conv."))]
        if dec_missing:
            raise RuntimeError(f"VAE checkpoint is missing {len(dec_missing)} decoder tensors, e.g. {dec_missing[:5]}")
        m._has_encoder = not any(k.startswith(("encoder.", "quant_conv.")) for k in missing)  # decoder-only checkpoints load
        return m

    @torch.no_grad()
    def init_synthetic_(self, seed: int = 4321):
        dev = self.post_quant_conv.weight.device
        g = torch.Generator(device=dev).manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith("gamma"):
                p.fill_(1.0)
            elif name.endswith("bias"):
                p.zero_()
            else:
                fan = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32) * fan ** -0.5)
        return self

    # ------------------------------------------------------------------ kernels
    def _zero_page(self, dev):
        if self._zero is None or self._zero.device != dev:
            self._zero = torch.zeros(256, dtype=BF16, device=dev)
        return self._zero

    def _conv(self, x, conv: _Conv, resid=None, up2x=False, pad_out=True):
        """x [B,T,H,W,Cpad] -> [B,T,H',W',pad64(Cout)] (Cout=3 stays unpadded)."""
        B, T, H, W, Cp = x.shape
        k = conv.k if len(conv.k) == 3 else (1,) + conv.k
        Ho, Wo = (2 * H, 2 * W) if up2x else (H, W)
        ldc = _pad64(conv.cout) if pad_out else conv.cout
        alloc = torch.zeros if ldc != conv.cout else torch.empty
        out = alloc((B, T, Ho, Wo, ldc), dtype=BF16, device=x.device)
        # the real channel count when the library can use it (96-channel stages: row-tile kernel), else the padded pitch
        cin = conv.cin if (conv.cin % 32 == 0 and _pad64(conv.cin) == Cp) else Cp
        call("lcv_causal_conv3d", x.data_ptr(), conv.packed().data_ptr(), conv.bias.data_ptr(),
             None if resid is None else resid.data_ptr(), out.data_ptr(), self._zero_page(x.device).data_ptr(),
             B, T, H, W, cin, conv.cout, ldc, k[0], k[1], k[2], 1 if up2x else 0, ops._stream())
        return out

    def _norm(self, x, norm: _Norm, silu=True):
        Cp = x.shape[-1]
        y = torch.empty_like(x)
        call("lcv_vae_rmsnorm_silu", x.data_ptr(), norm.padded().data_ptr(), y.data_ptr(), x.numel() // Cp, norm.dim, Cp,
             1 if silu else 0, ops._stream())
        return y

    def _res(self, x, r: _Res):
        h = x if isinstance(r.conv_shortcut, nn.Identity) else self._conv(x, r.conv_shortcut)
        y = self._conv(self._norm(x, r.norm1), r.conv1)
        return self._conv(self._norm(y, r.norm2), r.conv2, resid=h)

    def _attn(self, x, a: _Attn):
        """Per-frame single-head attention over h*w tokens, head_dim = C (384): scores and PV on the MFMA GEMM."""
        B, T, H, W, C = x.shape
        n = H * W
        npad = _pad64(n)
        xn = self._norm(x, a.norm, silu=False).view(B * T, n, C)
        out = torch.empty_like(xn)
        wq = a.to_qkv.packed()
        for f in range(B * T):
            qkv = ops.gemm_nt(xn[f], wq, a.to_qkv.bias)                      # [n, 3C]
            s = ops.gemm_nt(qkv[:, :C], qkv[:, C:2 * C], None, out_f32=True)  # [n, n] fp32
            p = torch.empty((n, npad), dtype=BF16, device=x.device)
            call("lcv_softmax_rows", s.data_ptr(), p.data_ptr(), n, n, n, npad, float(C) ** -0.5, ops._stream())
            vt = torch.zeros((C, npad), dtype=BF16, device=x.device)
            vt[:, :n] = qkv[:, 2 * C:].t()
            o = ops.gemm_nt(p, vt, None)                                      # [n, C]
            ops.gemm_nt(o, a.proj.packed(), a.proj.bias, epilogue=LCV_EPI_GATE_RESIDUAL,
                        resid=x.view(B * T, n, C)[f], out=out[f])
        return out.view(B, T, H, W, C)

    def _upsample(self, x, u: _Resample):
        B, T, H, W, C = x.shape
        if u.mode == "upsample3d" and T > 1:
            parts = []
            for b in range(B):  # the first latent frame is exempt from temporal upsampling
                rest = x[b:b + 1, 1:].contiguous()
                y = self._conv(rest, u.time_conv)                            # [1, T-1, H, W, 2C]
                y = y.view(1, T - 1, H, W, 2, C).permute(0, 1, 4, 2, 3, 5).reshape(1, 2 * (T - 1), H, W, C)
                parts.append(torch.cat([x[b:b + 1, :1], y], dim=1))
            x = torch.cat(parts, dim=0) if B > 1 else parts[0]
        return self._conv(x.contiguous(), u.resample[1], up2x=True)

    def _conv_strided(self, x, conv: _Conv, stride, out_thw):
        """x [B,T,H,W,Cpad] -> [B,T',H',W',pad64(Cout)], no front padding, zero taps past the input (lcv_conv3d_strided)."""
        B, T, H, W, Cp = x.shape
        k = conv.k if len(conv.k) == 3 else (1,) + conv.k
        To, Ho, Wo = out_thw
        ldc = _pad64(conv.cout)
        alloc = torch.zeros if ldc != conv.cout else torch.empty
        out = alloc((B, To, Ho, Wo, ldc), dtype=BF16, device=x.device)
        call("lcv_conv3d_strided", x.data_ptr(), conv.packed().data_ptr(), conv.bias.data_ptr(), out.data_ptr(),
             self._zero_page(x.device).data_ptr(), B, T, H, W, Cp, conv.cout, ldc, k[0], k[1], k[2], stride[0], stride[1],
             stride[2], To, Ho, Wo, ops._stream())
        return out

    def _downsample(self, x, d: "_DownResample"):
        B, T, H, W, C = x.shape
        x = self._conv_strided(x, d.resample[1], (1, 2, 2), (T, H // 2, W // 2))
        if d.mode == "downsample3d" and T > 1:
            if (T - 1) % 2:
                raise ValueError(f"temporal downsample needs an odd frame count at this stage, got {T}")
            parts = []
            for b in range(B):  # frame 0 is exempt; frame t >= 1 = conv(x[2t-2 .. 2t]) (chunked upstream form, whole sequence)
                xb = x[b:b + 1]
                y = self._conv_strided(xb, d.time_conv, (2, 1, 1), ((T - 1) // 2, H // 2, W // 2))
                parts.append(torch.cat([xb[:, :1], y], dim=1))
            x = torch.cat(parts, dim=0) if B > 1 else parts[0]
        return x

    @torch.no_grad()
    def encode(self, video: torch.Tensor, return_dict: bool = True):
        """video [B, 3, 1+4k, 8h, 8w] in [-1, 1] -> posterior over [B, z_dim, 1+k, h, w] (`.latent_dist`)."""
        if not self._has_encoder:
            raise RuntimeError("this VAE checkpoint carries no encoder tensors (decoder-only)")
        B, C, T, H, W = video.shape
        if C != 3 or (T - 1) % 4 or H % 8 or W % 8:
            raise ValueError(f"encode expects [B, 3, 1+4k, 8h, 8w], got {tuple(video.shape)}")
        e = self.encoder
        x = torch.zeros((B, T, H, W, 64), dtype=BF16, device=video.device)
        x[..., :3] = video.to(BF16).permute(0, 2, 3, 4, 1)
        x = self._conv(x, e.conv_in)
        for blk in e.down_blocks:
            x = self._res(x, blk) if isinstance(blk, _Res) else self._downsample(x, blk)
        x = self._res(x, e.mid_block.resnets[0])
        x = self._attn(x, e.mid_block.attentions[0])
        x = self._res(x, e.mid_block.resnets[1])
        x = self._conv(self._norm(x, e.norm_out), e.conv_out)
        x = self._conv(x, self.quant_conv)
        zz = 2 * self.config.z_dim
        params = x[..., :zz].permute(0, 4, 1, 2, 3).contiguous().to(self.dtype)
        dist = DiagonalGaussianDistribution(params)
        if return_dict:
            return SimpleNamespace(latent_dist=dist)
        return (dist,)

    def decode_work(self, T: int, h: int, w: int) -> dict:
        """ALGORITHMIC work of `decode` on latents [1, z, T, h, w] (true channel counts, no padding): `flops` = 2 x MACs of every
        convolution and of the mid-block attention; `norm_bytes` = 4 B per element through the channel RMS norms (bf16 in + out).
        Walks the same graph as `decode` ("count from the restated decoder graph")."""
        d = self.decoder
        fl = nb = 0

        def conv(c, t, hh, ww):
            taps = 1
            for k in c.k:
                taps *= k
            return 2 * t * hh * ww * c.cout * c.cin * taps

        def res(r, t, hh, ww):
            f = conv(r.conv1, t, hh, ww) + conv(r.conv2, t, hh, ww)
            if not isinstance(r.conv_shortcut, nn.Identity):
                f += conv(r.conv_shortcut, t, hh, ww)
            return f, 4 * t * hh * ww * (r.conv1.cin + r.conv2.cin)

        fl += conv(self.post_quant_conv, T, h, w) + conv(d.conv_in, T, h, w)
        t, hh, ww = T, h, w
        for r in d.mid_block.resnets:
            f, b = res(r, t, hh, ww); fl += f; nb += b
        a = d.mid_block.attentions[0]
        n, C = hh * ww, a.proj.cin
        fl += t * (2 * n * 3 * C * C + 4 * n * n * C + 2 * n * C * C); nb += 4 * t * n * C
        for ub in d.up_blocks:
            for r in ub.resnets:
                f, b = res(r, t, hh, ww); fl += f; nb += b
            if ub.upsamplers is not None:
                u = ub.upsamplers[0]
                if u.mode == "upsample3d" and t > 1:
                    fl += conv(u.time_conv, t - 1, hh, ww)
                    t = 1 + 2 * (t - 1)
                hh, ww = 2 * hh, 2 * ww
                fl += conv(u.resample[1], t, hh, ww)
        fl += conv(d.conv_out, t, hh, ww); nb += 4 * t * hh * ww * d.conv_out.cin
        return {"flops": fl, "norm_bytes": nb, "frames": t, "height": hh, "width": ww}

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = False):
        """z [B, z_dim, T, h, w] -> video [B, 3, 1+4(T-1), 8h, 8w] in [-1, 1]."""
        B, Cz, T, h, w = z.shape
        d = self.decoder
        x = torch.zeros((B, T, h, w, _pad64(Cz)), dtype=BF16, device=z.device)
        x[..., :Cz] = z.to(BF16).permute(0, 2, 3, 4, 1)
        x = self._conv(x, self.post_quant_conv)
        x = self._conv(x, d.conv_in)
        x = self._res(x, d.mid_block.resnets[0])
        x = self._attn(x, d.mid_block.attentions[0])
        x = self._res(x, d.mid_block.resnets[1])
        for ub in d.up_blocks:
            for r in ub.resnets:
                x = self._res(x, r)
            if ub.upsamplers is not None:
                x = self._upsample(x, ub.upsamplers[0])
        x = self._conv(self._norm(x, d.norm_out), d.conv_out, pad_out=False)  # [B, T', H', W', 3]
        video = x.permute(0, 4, 1, 2, 3).float().clamp_(-1.0, 1.0).to(self.dtype)
        if return_dict:
            return SimpleNamespace(sample=video)
        return (video,)
