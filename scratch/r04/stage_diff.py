"""Where do the HIP DiT and the oracle at the kernels' rounding points (dit_forward(bf16="kernel")) part?  Records the hidden stream
after the embedders, after every stage of every block (hooks on the product's modules make it take its unfused paths; the fused
epilogues round where the unfused kernels do) and after the final layer, and prints rel-L2 per stage."""
import sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from oracle import dit_oracle as orc
from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
BF16 = torch.bfloat16
cfg = orc.small_config(hidden_size=256, depth=2, num_heads=2, caption_channels=64)
P = orc.make_params(cfg, seed=7, std=0.05)
m = LongCatVideoTransformer3DModel(device="cuda", dtype=BF16, hidden_size=256, depth=2, num_heads=2, caption_channels=64,
                                   adaln_tembed_dim=cfg["adaln_tembed_dim"])
m.load_state_dict(P, strict=False); m.eval()
g = torch.Generator().manual_seed(0)
B, T, H, W, L = 1, 3, 8, 12, 20
hs = torch.randn(B, 16, T, H, W, generator=g).to(BF16); y = torch.randn(B, 1, L, 64, generator=g).to(BF16)
mask = torch.zeros(B, L, dtype=torch.int64); mask[0, :13] = 1
ts = torch.zeros(B, T); ts[:, 0:] = 371.5
ncond = 0
mode = sys.argv[1] if len(sys.argv) > 1 else "kernel"
# ---- oracle side: record the outputs of the stage functions in call order
rec_o = []
def wrap(name):
    f = getattr(orc, name)
    def w(*a, **k):
        r = f(*a, **k)
        rec_o.append((name, (r[0] if isinstance(r, tuple) else r).detach().float().cpu()))
        return r
    setattr(orc, name, w)
for n in ("x_embedder", "y_embedder", "t_embedder", "adaln_table", "modulate_fp32", "self_attention", "layernorm_fp32", "cross_attention", "ffn", "block_forward",
          "final_layer"):
    wrap(n)
ref = orc.dit_forward(P, cfg, hs, ts.to(BF16), y, mask, ncond, bf16=(mode if mode == "kernel" else True))
# ---- product side
rec_h = []
def hook(name):
    def h(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        rec_h.append((name, o.detach().float().cpu()))
    return h
m.x_embedder.register_forward_hook(hook("x_embedder")); m.y_embedder.register_forward_hook(hook("y_embedder"))
m.t_embedder.register_forward_hook(hook("t_embedder")); m.final_layer.register_forward_hook(hook("final_layer"))
for i, b in enumerate(m.blocks):
    b.adaLN_modulation.register_forward_hook(hook("adaln_table")); b.attn.register_forward_hook(hook("self_attention"))
    b.pre_crs_attn_norm.register_forward_hook(hook("layernorm_fp32")); b.cross_attn.register_forward_hook(hook("cross_attention"))
    b.ffn.register_forward_hook(hook("ffn")); b.register_forward_hook(hook("block_forward"))
from longcat_video.modules import layers as LY
_fork = LY.A.adaln_modulate_fork
def fork_rec(x, mod, *a, **k):
    r = _fork(x, mod, *a, **k)
    rec_h.append(("modulate_fp32", r[0].detach().float().cpu()))
    return r
LY.A.adaln_modulate_fork = fork_rec
with torch.no_grad():
    got = m(hidden_states=hs.cuda(), timestep=ts.to(BF16).cuda(), encoder_hidden_states=y.cuda(), encoder_attention_mask=mask.cuda(),
            num_cond_latents=ncond)
rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
print("mode", mode, " final:", rel(got.float().cpu(), ref))
oi = list(rec_o)
names_h = [n for n, _ in rec_h]
j = 0
for n, t in rec_h:
    while j < len(oi) and oi[j][0] != n: j += 1
    if j >= len(oi): break
    to = oi[j][1]; j += 1
    if to.numel() != t.numel():
        print(f"{n:18s} shapes differ {tuple(t.shape)} vs {tuple(to.shape)}"); continue
    print(f"{n:18s} rel-L2 {rel(t.reshape(-1), to.reshape(-1)):.2e}   exact-equal fraction {(t.reshape(-1) == to.reshape(-1)).float().mean().item():.4f}")
