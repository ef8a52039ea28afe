"""Self-attention of block 0, stage by stage, HIP ops vs the oracle pieces on IDENTICAL inputs (each stage is fed the HIP path's own
previous output, so a figure is that stage's own disagreement, not an inherited one)."""
import math, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from oracle import dit_oracle as orc
from lcv_hip import ops
from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
BF16 = torch.bfloat16
cfg = orc.small_config(hidden_size=256, depth=2, num_heads=2, caption_channels=64)
P = orc.make_params(cfg, seed=7, std=0.05)
m = LongCatVideoTransformer3DModel(device="cuda", dtype=BF16, hidden_size=256, depth=2, num_heads=2, caption_channels=64,
                                   adaln_tembed_dim=cfg["adaln_tembed_dim"])
m.load_state_dict(P, strict=False); m.eval()
att = m.blocks[0].attn
g = torch.Generator().manual_seed(0)
B, T, Hh, Ww = 1, 3, 4, 6
N, C, H, D = T * Hh * Ww, 256, 2, 128
x = torch.randn(B, N, C, generator=g).to(BF16)
rel = lambda a, b: ((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm()).item()
eq = lambda a, b: (a.float().cpu() == b.float().cpu()).float().mean().item()
rnd = orc.bf16_round
with torch.no_grad():
    qkv = att.qkv(x.cuda()).view(B, N, 3, H, D)
    qkv_o = orc._lin(P, "blocks.0.attn.qkv", x, rnd).view(B, N, 3, H, D)
    print(f"qkv GEMM          rel {rel(qkv, qkv_o):.2e} equal {eq(qkv, qkv_o):.4f}")
    cs = att.rope_3d.table((T, Hh, Ww), "cuda")
    q, k, v = qkv[:, :, 0].clone(), qkv[:, :, 1].clone(), qkv[:, :, 2]
    qin, kin = q.clone(), k.clone()
    ops.qknorm_rope(q, k, None, q, k, None, att.q_norm.weight, att.k_norm.weight, cs, 0, att.q_norm.eps, q_scale=ops.log2_qscale(att.scale))
    ang = orc.rope_angles_3d((T, Hh, Ww), D)
    qo = orc.rmsnorm_fp32(qin.cpu().permute(0, 2, 1, 3), P["blocks.0.attn.q_norm.weight"], rnd=rnd)
    ko = orc.rmsnorm_fp32(kin.cpu().permute(0, 2, 1, 3), P["blocks.0.attn.k_norm.weight"], rnd=rnd)
    (qo2, sc), ko2 = orc._rope_q(qo, ang, D ** -0.5, orc.bf16_round_kernel_attention), orc.apply_rope(ko, ang, rnd)
    print(f"q' (norm+rope*c)  rel {rel(q.permute(0, 2, 1, 3), qo2):.2e} equal {eq(q.permute(0, 2, 1, 3), qo2):.4f}")
    print(f"k  (norm+rope)    rel {rel(k.permute(0, 2, 1, 3), ko2):.2e} equal {eq(k.permute(0, 2, 1, 3), ko2):.4f}")
    o, _ = ops.attention(q, k, v, ops.LN2)
    oo = orc.sdpa_at_kernel_rounding(q.cpu().permute(0, 2, 1, 3), k.cpu().permute(0, 2, 1, 3), v.cpu().permute(0, 2, 1, 3), math.log(2.0))
    print(f"attention         rel {rel(o.permute(0, 2, 1, 3), oo):.2e} equal {eq(o.permute(0, 2, 1, 3), oo):.4f}   (Nk = {N})")
    pr = att.proj(o.reshape(B, N, C))
    pro = orc._lin(P, "blocks.0.attn.proj", o.cpu().reshape(B, N, C), rnd)
    print(f"proj GEMM         rel {rel(pr, pro):.2e} equal {eq(pr, pro):.4f}")
