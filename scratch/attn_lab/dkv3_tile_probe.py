"""Which query tile of the small failing shape does attn_bwd_dkv3 get wrong?  dO is non-zero for one 32-query tile at a time."""
import os, sys, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "longcat-video-tta_amd"))
from lcv_hip import ops
B, H, Nq, Nk, D = 1, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 128, 128
g = torch.Generator().manual_seed(1)
q = (torch.randn(B, Nq, H, D, generator=g) * 0.3 * D ** -0.5 * math.log2(math.e)).bfloat16().cuda()
k = (torch.randn(B, Nk, H, D, generator=g) * 0.3).bfloat16().cuda()
v = torch.randn(B, Nk, H, D, generator=g).bfloat16().cuda()
do_full = torch.randn(B, Nq, H, D, generator=g).bfloat16().cuda()
o, lse = ops.attention(q, k, v, ops.LN2, need_lse=True)
def run(do, which):
    os.environ["LCV_ATTN_BWD_DKV"] = which
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, ops.LN2)
    return dk.float(), dv.float()
for t in list(range((Nq + 31) // 32)) + [-1]:
    do = do_full.clone()
    if t >= 0:
        do.zero_(); do[:, 32 * t:32 * t + 32] = do_full[:, 32 * t:32 * t + 32]
    (k2, v2), (k3, v3) = run(do, "2"), run(do, "3")
    e = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    print(f"tile {t}: dk {e(k3, k2):.3e} dv {e(v3, v2):.3e}  per-wave dk " + " ".join(f"{e(k3[:, 32*w:32*w+32], k2[:, 32*w:32*w+32]):.2e}" for w in range(min(4, Nk // 32))))
# query by query inside the last tile
t = (Nq + 31) // 32 - 1
bad = []
for r in range(32 * t, Nq):
    do = torch.zeros_like(do_full); do[:, r] = do_full[:, r]
    (k2, v2), (k3, v3) = run(do, "2"), run(do, "3")
    err = float((k3 - k2).norm() / k2.norm().clamp_min(1e-30))
    if err > 0: bad.append((r - 32 * t, round(err, 3)))
print("last tile, single-query dO: rows with a dk difference:", bad)
# what did row 20 of the last tile get?  fp32 restatements of dk for one query under hypotheses about its D = dP - delta
r = 32 * t + 20
do = torch.zeros_like(do_full); do[:, r] = do_full[:, r]
(k2, _), (k3, _) = run(do, "2"), run(do, "3")
qf, kf, vf, dof, of = (x[0, :, 0].float() for x in (q, k, v, do, o))
s = qf[r] @ kf.T                       # log2 units
p_ = torch.exp2(s - torch.logsumexp(s * math.log(2), 0) / math.log(2))
dp = dof[r] @ vf.T
delta = (dof[r] * of[r]).sum()
rel = lambda a, b: float((a - b).norm() / b.norm())
for name, dmat in (("dP - delta", dp - delta), ("dP", dp), ("-delta", -delta.expand_as(dp)), ("dP(first 112 dims) - delta", dof[r, :112] @ vf[:, :112].T - delta),
                   ("2 dP - delta", 2 * dp - delta), ("dP - 2 delta", dp - 2 * delta)):
    dk_h = ((p_ * dmat)[:, None] * qf[r][None, :]) * math.log(2)
    print(f"  hypothesis D = {name}: dkv3 off by {rel(k3[0, :, 0], dk_h):.3e}, dkv2 off by {rel(k2[0, :, 0], dk_h):.3e}")
a3, a2 = k3[0, :, 0], k2[0, :, 0]
ratio = (a3 * a2).sum(1) / (a2 * a2).sum(1)                     # dS3[key] / dS2[key]
resid = ((a3 - ratio[:, None] * a2).norm(dim=1) / a2.norm(dim=1))
print("  per key dS3/dS2:", " ".join(f"{x:.2f}" for x in ratio.tolist()))
print("  max residual (not a multiple of q):", float(resid.max()))
ds2 = (a2 * qf[r][None]).sum(1) / (qf[r] ** 2).sum() / math.log(2); ds3 = (a3 * qf[r][None]).sum(1) / (qf[r] ** 2).sum() / math.log(2)
print("  (dS3 - dS2) / P per key:", " ".join(f"{x:.3f}" for x in ((ds3 - ds2) / p_).tolist()))
print("  delta of this row:", float(delta), " dP range", float(dp.min()), float(dp.max()))
Qt = qf[32 * t:32 * t + 32]                                   # [32, 128]
coef = torch.linalg.lstsq(Qt.T, (a3 / math.log(2)).T).solution  # [32 rows, keys]: dS3[row, key]
coef2 = torch.linalg.lstsq(Qt.T, (a2 / math.log(2)).T).solution
print("  |dS3| per row of the last tile (mean over keys):", " ".join(f"{x:.3f}" for x in coef.abs().mean(1).tolist()))
print("  |dS2| per row of the last tile (mean over keys):", " ".join(f"{x:.3f}" for x in coef2.abs().mean(1).tolist()))
fit = (Qt.T @ coef).T
print("  fit residual:", float((fit - a3 / math.log(2)).norm() / (a3 / math.log(2)).norm()))
