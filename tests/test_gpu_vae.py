"""GPU parity of the HIP VAE decoder and encoder against the CPU oracle (oracle/vae_oracle.{decode,encode}_full at the same
bf16 rounding points).  Tolerance: relative L2 <= 2e-2 on the clamped video after ~30 bf16-rounded conv/norm stages (the oracle's own
bf16-vs-fp32 gap is printed for scale)."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _build(cfg, P):
    from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
    vae = AutoencoderKLWan(base_dim=cfg["base_dim"], z_dim=cfg["z_dim"], device="cuda", dtype=BF16)
    from oracle import vae_oracle as V
    full = dict(P)
    for k, v in V.make_encoder_params(cfg, seed=5).items():
        full.setdefault(k, v)
    for k, v in V.make_params(cfg, seed=3).items():
        full.setdefault(k, v)
    missing, unexpected = vae.load_state_dict(full, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return vae


def test_conv_kernel_against_conv3d():
    import torch.nn.functional as F
    from longcat_video.modules.vae_wan import AutoencoderKLWan, _Conv
    vae = AutoencoderKLWan(base_dim=16, z_dim=4, device="cuda", dtype=BF16)
    g = torch.Generator().manual_seed(0)
    # Cout >= 192: the wide tiles (see the next test for many tiles)
    for (ci, co, k, up) in ((64, 96, (3, 3, 3), False), (128, 64, (3, 3), True), (64, 128, (3, 1, 1), False), (64, 3, (3, 3, 3), False),
                            (64, 192, (3, 3, 3), False), (128, 256, (3, 3), True), (192, 384, (1, 1, 1), False)):
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
        assert rel_l2(got, ref, bound=3e-3) < 3e-3, (ci, co, k, up)


@pytest.mark.parametrize("ci,co,k,up,W,resid", [
    (96, 96, (3, 3, 3), False, 200, False),     # one partial 256-pixel tile per image row
    (96, 96, (3, 3, 3), False, 400, True),      # two tiles, the second partial; residual tail
    (96, 3, (3, 3, 3), False, 256, False),      # the 3-channel head (16-column tile), exact tile
    (192, 96, (3, 3), True, 200, False),        # folded 2x upsample: output rows of 400 pixels, two channel slices per tap
    (192, 96, (1, 1, 1), False, 210, False),    # pointwise shortcut
])
def test_row_tile_conv_kernel(ci, co, k, up, W, resid, monkeypatch):
    """conv_rows.h (96-channel stages) against torch's convolution, and against the implicit-GEMM kernel it replaces there
    (LCV_CONV_ROWS=0): same products, another summation order."""
    import torch.nn.functional as F
    from longcat_video.modules.vae_wan import AutoencoderKLWan, _Conv
    vae = AutoencoderKLWan(base_dim=16, z_dim=4, device="cuda", dtype=BF16)
    g = torch.Generator().manual_seed(11)
    T, H = 3, 5
    conv = _Conv(ci, co, k, device="cuda", dtype=BF16)
    taps = 1
    for v in k:
        taps *= v
    with torch.no_grad():
        conv.weight.copy_(torch.randn((co, ci) + k, generator=g) * (ci * taps) ** -0.5)
        conv.bias.copy_(torch.randn(co, generator=g) * 0.1)
    cp = (ci + 63) // 64 * 64
    x = torch.zeros(1, T, H, W, cp, dtype=BF16)
    x[..., :ci] = torch.randn(1, T, H, W, ci, generator=g).to(BF16)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    r = None
    if resid:
        r = torch.zeros(1, T, Ho, Wo, (co + 63) // 64 * 64, dtype=BF16)
        r[..., :co] = torch.randn(1, T, Ho, Wo, co, generator=g).to(BF16)
    run = lambda: vae._conv(x.cuda(), conv, resid=None if r is None else r.cuda(), up2x=up, pad_out=(co != 3))
    from lcv_hip import lib as _lib
    got_full = run()
    assert _lib.load().lcv_conv3d_last_kernel().decode().startswith("conv_rows")
    got = got_full[..., :co].float().cpu()
    assert float(got_full[..., co:].abs().max() if got_full.shape[-1] > co else 0) == 0       # padding channels stay zero
    monkeypatch.setenv("LCV_CONV_ROWS", "0")
    old = run()[..., :co].float().cpu()
    assert _lib.load().lcv_conv3d_last_kernel().decode().startswith("conv16_igemm")
    monkeypatch.delenv("LCV_CONV_ROWS")
    xn = x[..., :ci].float().permute(0, 4, 1, 2, 3)
    wf, bf = conv.weight.float().cpu(), conv.bias.float().cpu()
    if len(k) == 2:
        y = xn.permute(0, 2, 1, 3, 4).reshape(T, ci, H, W)
        if up:
            y = F.interpolate(y, scale_factor=(2.0, 2.0), mode="nearest-exact")
        ref = F.conv2d(y, wf, bf, padding=1).view(1, T, co, Ho, Wo).permute(0, 1, 3, 4, 2)
    else:
        y = F.pad(xn, (k[2] // 2, k[2] // 2, k[1] // 2, k[1] // 2, k[0] - 1, 0))
        ref = F.conv3d(y, wf, bf).permute(0, 2, 3, 4, 1)
    if resid:
        ref = r[..., :co].float() + ref.to(BF16).float()
    assert rel_l2(got, ref, bound=3e-3) < 3e-3
    assert rel_l2(got, old, bound=5.5e-5) < 5.5e-5 and not torch.equal(got, torch.zeros_like(got))


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


@pytest.mark.parametrize("ci,co,k,up,resid", [(192, 192, (3, 3, 3), False, True), (384, 192, (3, 3), True, False), (64, 320, (3, 3, 3), False, False)])
def test_wide_conv_kernel_many_tiles(ci, co, k, up, resid, monkeypatch):
    """The wide implicit GEMMs (Cout >= 192) over several hundred 256-pixel tiles with image borders inside tiles: the default
    (192-column tiles with loader waves when Cout is a multiple of 192; the same tile without them and on a three-buffer ring), the opt-in 8-phase form (LCV_CONV_8P=1; more than one tile per CU of its
    persistent grid) and the 256-column two-stage kernel - the same products in the same K order, so bit for bit - and torch."""
    import torch.nn.functional as F
    from longcat_video.modules.vae_wan import AutoencoderKLWan, _Conv
    from lcv_hip import lib as _lib
    vae = AutoencoderKLWan(base_dim=16, z_dim=4, device="cuda", dtype=BF16)
    g = torch.Generator().manual_seed(21)
    T, H, W = 3, 150, 203                       # 91350 pixels = 357 tiles (x 4 with the upsample)
    conv = _Conv(ci, co, k, device="cuda", dtype=BF16)
    taps = 1
    for v in k:
        taps *= v
    with torch.no_grad():
        conv.weight.copy_(torch.randn((co, ci) + k, generator=g) * (ci * taps) ** -0.5)
        conv.bias.copy_(torch.randn(co, generator=g) * 0.1)
    x = torch.randn(1, T, H, W, ci, generator=g).to(BF16)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    r = None
    if resid:
        r = torch.zeros(1, T, Ho, Wo, (co + 63) // 64 * 64, dtype=BF16)
        r[..., :co] = torch.randn(1, T, Ho, Wo, co, generator=g).to(BF16)
    run = lambda: vae._conv(x.cuda(), conv, resid=None if r is None else r.cuda(), up2x=up)
    got_full = run()                                                    # the default: 192- or 256-column tiles
    assert _lib.load().lcv_conv3d_last_kernel().decode() == ("conv_wide<256x192>" if co % 192 == 0 else "conv16_igemm<256x256>")
    if co % 192 == 0:
        for knob, name in (("2", "conv16_igemm<256x192>"), ("3", "conv16_igemm<192x192x3>")):   # no loader waves; three-buffer ring
            monkeypatch.setenv("LCV_CONV_N192", knob)
            other = run()
            assert _lib.load().lcv_conv3d_last_kernel().decode() == name
            monkeypatch.delenv("LCV_CONV_N192")
            assert torch.equal(other, got_full)
    monkeypatch.setenv("LCV_CONV_8P", "1")
    p8 = run()
    assert _lib.load().lcv_conv3d_last_kernel().decode().startswith("conv8p")
    monkeypatch.delenv("LCV_CONV_8P")
    monkeypatch.setenv("LCV_CONV_N192", "0")
    old = run()
    assert _lib.load().lcv_conv3d_last_kernel().decode() == "conv16_igemm<256x256>"
    monkeypatch.delenv("LCV_CONV_N192")
    assert torch.equal(got_full, old) and torch.equal(p8, old)
    got = got_full[..., :co].float().cpu()
    xn = x.float().permute(0, 4, 1, 2, 3)
    wf, bf = conv.weight.float().cpu(), conv.bias.float().cpu()
    if len(k) == 2:
        y = xn.permute(0, 2, 1, 3, 4).reshape(T, ci, H, W)
        if up:
            y = F.interpolate(y, scale_factor=(2.0, 2.0), mode="nearest-exact")
        ref = F.conv2d(y, wf, bf, padding=1).view(1, T, co, Ho, Wo).permute(0, 1, 3, 4, 2)
    else:
        y = F.pad(xn, (k[2] // 2, k[2] // 2, k[1] // 2, k[1] // 2, k[0] - 1, 0))
        ref = F.conv3d(y, wf, bf).permute(0, 2, 3, 4, 1)
    if resid:
        ref = r[..., :co].float() + ref.to(BF16).float()
    assert rel_l2(got, ref, bound=3e-3) < 3e-3


def test_strided_conv_kernel_against_torch():
    """The encoder's two downsampling convs: ZeroPad2d((0,1,0,1)) + 3x3 stride 2 per frame, and (3,1,1) stride 2 in time."""
    import torch.nn.functional as F
    from longcat_video.modules.vae_wan import AutoencoderKLWan, _Conv
    vae = AutoencoderKLWan(base_dim=16, z_dim=4, device="cuda", dtype=BF16)
    g = torch.Generator().manual_seed(1)
    for (H, W, ci, co) in ((6, 10, 64, 64), (7, 9, 64, 64), (32, 45, 128, 192)):     # the last: wide tiles, several of them
        conv = _Conv(ci, co, (3, 3), device="cuda", dtype=BF16)
        with torch.no_grad():
            conv.weight.copy_(torch.randn((co, ci, 3, 3), generator=g) * (ci * 9) ** -0.5); conv.bias.copy_(torch.randn(co, generator=g) * 0.1)
        x = torch.randn(1, 3, H, W, ci, generator=g).to(BF16)
        got = vae._conv_strided(x.cuda(), conv, (1, 2, 2), (3, H // 2, W // 2)).float().cpu()
        y = F.pad(x.float().permute(0, 1, 4, 2, 3).reshape(3, ci, H, W), (0, 1, 0, 1))
        ref = F.conv2d(y, conv.weight.float().cpu(), conv.bias.float().cpu(), stride=2)
        assert ref.shape[-2:] == (H // 2, W // 2)
        assert rel_l2(got, ref.view(1, 3, co, H // 2, W // 2).permute(0, 1, 3, 4, 2), bound=3e-3) < 3e-3
    ci = co = 64
    conv = _Conv(ci, co, (3, 1, 1), device="cuda", dtype=BF16)
    with torch.no_grad():
        conv.weight.copy_(torch.randn((co, ci, 3, 1, 1), generator=g) * (ci * 3) ** -0.5); conv.bias.copy_(torch.randn(co, generator=g) * 0.1)
    x = torch.randn(1, 9, 4, 5, ci, generator=g).to(BF16)
    got = vae._conv_strided(x.cuda(), conv, (2, 1, 1), (4, 4, 5)).float().cpu()
    ref = F.conv3d(x.float().permute(0, 4, 1, 2, 3), conv.weight.float().cpu(), conv.bias.float().cpu(), stride=(2, 1, 1))
    assert rel_l2(got, ref.permute(0, 2, 3, 4, 1), bound=3e-3) < 3e-3


@pytest.mark.parametrize("k", [0, 2])
def test_vae_encode_matches_oracle(k):
    from oracle import vae_oracle as V
    cfg = V.default_config(base_dim=16, z_dim=4)
    P = V.make_encoder_params(cfg, seed=5)
    vae = _build(cfg, P)
    g = torch.Generator().manual_seed(6)
    video = (torch.rand(1, 3, 1 + 4 * k, 32, 48, generator=g) * 2 - 1).to(BF16)
    post = vae.encode(video.cuda()).latent_dist
    got = post.mode()
    Pf = {n: v.float() for n, v in P.items()}
    ref = V.encode_full(Pf, cfg, video, rnd=True)
    ref32 = V.encode_full(Pf, cfg, video, rnd=False)
    assert got.shape == ref.shape == (1, 4, 1 + k, 4, 6)
    e = rel_l2(got, ref)
    print(f"vae encode rel_l2 hip-vs-oracle = {e:.2e}; oracle bf16-vs-fp32 = {rel_l2(ref, ref32):.2e}")
    assert e < 2e-2
    # the posterior surface the reference reads: mode() is the mean; sample() is mean + std * eps from the given generator
    from longcat_video.pipeline_longcat_video import retrieve_latents
    assert torch.equal(retrieve_latents(vae.encode(video.cuda()), sample_mode="argmax"), got)
    # the default draws from the posterior (mean + std * eps with the given generator): reproducible and centred on the mode
    g1 = torch.Generator(device="cuda").manual_seed(3)
    g2 = torch.Generator(device="cuda").manual_seed(3)
    enc = vae.encode(video.cuda())
    s1, s2 = retrieve_latents(enc, generator=g1), retrieve_latents(enc, generator=g2)
    assert torch.equal(s1, s2) and not torch.equal(s1, got)
    assert ((s1.float() - got.float()).abs() <= 6.0 * enc.latent_dist.std.float() + 0.02 * s1.float().abs() + 1e-2).all()   # + bf16 rounding of the sum
    g1 = torch.Generator(device="cuda").manual_seed(7); g2 = torch.Generator(device="cuda").manual_seed(7)
    assert torch.equal(post.sample(g1), post.sample(g2)) and post.logvar.max() <= 20 and post.logvar.min() >= -30


def test_vae_encode_rejects_bad_shapes_and_decoder_only_checkpoints():
    from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
    vae = AutoencoderKLWan(base_dim=16, z_dim=4, device="cuda", dtype=BF16).init_synthetic_()
    with pytest.raises(ValueError):
        vae.encode(torch.zeros(1, 3, 4, 32, 32, device="cuda", dtype=BF16))   # T != 1 + 4k
    with pytest.raises(ValueError):
        vae.encode(torch.zeros(1, 3, 5, 30, 32, device="cuda", dtype=BF16))   # H % 8
    vae._has_encoder = False
    with pytest.raises(RuntimeError):
        vae.encode(torch.zeros(1, 3, 5, 32, 32, device="cuda", dtype=BF16))
