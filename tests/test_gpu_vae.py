"""GPU parity of the HIP VAE decoder against the CPU oracle (oracle/vae_oracle.decode_full at the same bf16 rounding
points).  Tolerance: relative L2 <= 2e-2 on the clamped video after ~30 bf16-rounded conv/norm stages (the oracle's own
bf16-vs-fp32 gap is printed for scale)."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _build(cfg, P):
    from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
    vae = AutoencoderKLWan(base_dim=cfg["base_dim"], z_dim=cfg["z_dim"], device="cuda", dtype=BF16)
    missing, unexpected = vae.load_state_dict(P, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return vae


def test_conv_kernel_against_conv3d():
    import torch.nn.functional as F
    from longcat_video.modules.vae_wan import AutoencoderKLWan, _Conv
    vae = AutoencoderKLWan(base_dim=16, z_dim=4, device="cuda", dtype=BF16)
    g = torch.Generator().manual_seed(0)
    for (ci, co, k, up) in ((64, 96, (3, 3, 3), False), (128, 64, (3, 3), True), (64, 128, (3, 1, 1), False), (64, 3, (3, 3, 3), False)):
        conv = _Conv(ci, co, k, device="cuda", dtype=BF16)
        w = torch.randn((co, ci) + k, generator=g) * (ci * 9) ** -0.5
        b = torch.randn(co, generator=g) * 0.1
        with torch.no_grad():
            conv.weight.copy_(w); conv.bias.copy_(b)
        x = torch.randn(1, 3, 5, 7, ci, generator=g).to(BF16)
        got = vae._conv(x.cuda(), conv, up2x=up, pad_out=(co != 3))[..., :co].float().cpu()
        xn = x.float().permute(0, 4, 1, 2, 3)
        wf, bf = conv.weight.float().cpu(), conv.bias.float().cpu()
        if len(k) == 2:
            y = xn.permute(0, 2, 1, 3, 4).reshape(3, ci, 5, 7)
            if up:
                y = F.interpolate(y, scale_factor=(2.0, 2.0), mode="nearest-exact")
            ref = F.conv2d(y, wf, bf, padding=1).view(1, 3, co, y.shape[2], y.shape[3]).permute(0, 1, 3, 4, 2)
        else:
            y = F.pad(xn, (k[2] // 2, k[2] // 2, k[1] // 2, k[1] // 2, k[0] - 1, 0))
            ref = F.conv3d(y, wf, bf).permute(0, 2, 3, 4, 1)
        assert rel_l2(got, ref) < 3e-3, (ci, co, k, up)


@pytest.mark.parametrize("T", [1, 3])
def test_vae_decode_matches_oracle(T):
    from oracle import vae_oracle as V
    cfg = V.default_config(base_dim=16, z_dim=4)
    P = V.make_params(cfg, seed=3)
    vae = _build(cfg, P)
    g = torch.Generator().manual_seed(4)
    z = torch.randn(1, 4, T, 6, 10, generator=g).to(BF16)
    got = vae.decode(z.cuda(), return_dict=False)[0]
    ref = V.decode_full({k: v.float() for k, v in P.items()}, cfg, z, rnd=True)
    ref32 = V.decode_full({k: v.float() for k, v in P.items()}, cfg, z, rnd=False)
    assert got.shape == ref.shape == (1, 3, 1 + 4 * (T - 1), 48, 80)
    e = rel_l2(got, ref)
    print(f"vae decode rel_l2 hip-vs-oracle = {e:.2e}; oracle bf16-vs-fp32 = {rel_l2(ref, ref32):.2e}")
    assert e < 2e-2
    assert got.min() >= -1 and got.max() <= 1
