"""GPU parity of the HIP UMT5 text encoder (longcat_video/modules/umt5_encoder.py + csrc/text_encoder.hip) against
oracle/umt5_oracle.py, which tests/test_umt5_oracle.py pins to transformers.UMT5EncoderModel.

Tolerances: single kernels against the oracle at the same bf16 rounding points — bitwise for the gather, <= 1 bf16 ulp
on < 1 % of elements for the norm / gated GELU (tanhf vs torch.tanh), rel-L2 <= 4e-3 for the attention (probabilities are
rounded to bf16 before the PV sum on both sides; the exp differs in the last bits); the whole encoder rel-L2 <= 2e-2 on
the valid tokens (the bf16 model is itself ~1e-2 from the fp32 one at these depths — printed for scale)."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _build(cfg, P):
    from longcat_video.modules.umt5_encoder import UMT5EncoderModel
    m = UMT5EncoderModel(device="cuda", dtype=BF16, **cfg)
    sd = dict(P); sd["encoder.embed_tokens.weight"] = P["shared.weight"]       # as a Hugging Face checkpoint carries it
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m.eval()


def test_gather_norm_geglu_kernels():
    from lcv_hip import ops
    from oracle import umt5_oracle as O
    g = torch.Generator().manual_seed(0)
    table = torch.randn(500, 256, generator=g).to(BF16)
    ids = torch.randint(0, 500, (3, 41), generator=g)
    assert torch.equal(ops.gather_rows(table.cuda(), ids.cuda()).cpu(), table[ids.view(-1)])
    for C in (256, 4096):
        x = (torch.randn(37, C, generator=g) * 3).to(BF16); w = (1 + 0.2 * torch.randn(C, generator=g)).to(BF16)
        got = ops.t5_rmsnorm(x.cuda(), w.cuda(), 1e-6).cpu().float()
        ref = O.rms_norm(x.float(), w.float(), 1e-6, O.bf16_round)
        bad = (got != ref)
        assert bad.float().mean() < 0.01 and ((got - ref).abs() <= ref.abs() * 2 ** -7 + 1e-30).all()
    gu = (torch.randn(33, 2 * 512, generator=g) * 2).to(BF16).cuda()
    got = ops.geglu_tanh(gu[:, :512], gu[:, 512:]).cpu().float()
    ref = O.bf16_round(O.bf16_round(O.gelu_new(gu[:, :512].cpu().float())) * gu[:, 512:].cpu().float())
    assert (got != ref).float().mean() < 0.01 and ((got - ref).abs() <= ref.abs() * 2 ** -6 + 1e-30).all()


@pytest.mark.parametrize("B,S,H,nvalid", [(1, 100, 4, 100), (2, 512, 8, 77), (1, 64, 2, 5), (1, 333, 3, 200)])
def test_attention_kernel_matches_oracle(B, S, H, nvalid):
    from lcv_hip import ops
    from oracle import umt5_oracle as O
    g = torch.Generator().manual_seed(S + H)
    inner = H * 64
    qkv = (torch.randn(B, S, 3 * inner, generator=g) * 0.6).to(BF16)
    relb = torch.randn(32, H, generator=g).to(BF16)
    mask = torch.zeros(B, S, dtype=torch.int32); mask[:, :nvalid] = 1
    if B > 1:
        mask[1, :] = 1                                                     # one fully valid row in the batch
    d = torch.arange(-(S - 1), S)
    bias_by_dist = relb[O.relative_position_bucket(d)].t().float().contiguous()
    got = ops.t5_attention(qkv.cuda(), H, bias_by_dist.cuda(), mask.cuda()).cpu().float()
    r = O.bf16_round
    q, k, v = (qkv.float()[..., i * inner:(i + 1) * inner].view(B, S, H, 64).transpose(1, 2) for i in range(3))
    pos = torch.arange(S)
    bias = relb.float()[O.relative_position_bucket(pos[None, :] - pos[:, None])].permute(2, 0, 1)[None]
    s = r(r(q @ k.transpose(2, 3)) + bias)
    s = s.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
    ref = r(r(torch.softmax(s, -1)) @ v).transpose(1, 2).reshape(B, S, inner)
    assert rel_l2(got, ref, bound=4.7e-5) < 4.7e-5, rel_l2(got, ref)


@pytest.mark.parametrize("cfg,S,nvalid", [
    (dict(vocab_size=1000, d_model=256, d_kv=64, d_ff=512, num_layers=3, num_heads=4), 96, 61),
    (dict(vocab_size=2048, d_model=4096, d_kv=64, d_ff=10240, num_layers=2, num_heads=64), 512, 77),   # UMT5-XXL width
])
def test_encoder_matches_oracle(cfg, S, nvalid):
    from oracle import umt5_oracle as O
    P = O.make_params(cfg, seed=7)
    m = _build(cfg, P)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, cfg["vocab_size"], (1, S), generator=g)
    mask = torch.zeros(1, S, dtype=torch.long); mask[:, :nvalid] = 1
    out = m(ids.cuda(), mask.cuda()).last_hidden_state
    assert out.shape == (1, S, cfg["d_model"]) and out.dtype == BF16
    Pf = {k: v.float() for k, v in P.items()}
    ref = O.encoder_forward(Pf, cfg, ids, mask, bf16=True)
    exact = O.encoder_forward(Pf, cfg, ids, mask, bf16=False)
    e, scale = rel_l2(out[0, :nvalid], ref[0, :nvalid]), rel_l2(ref[0, :nvalid], exact[0, :nvalid])
    print(f"HIP vs bf16 oracle {e:.2e}; bf16 oracle vs fp32 oracle {scale:.2e}")
    assert e < 2e-2
    assert rel_l2(out[0, :nvalid], exact[0, :nvalid], bound=2.5) < 2.5 * scale + 5e-3


def test_encoder_matches_transformers_bf16_cpu():
    """The class the reference instantiates, run in bf16 on the CPU with the same weights."""
    transformers = pytest.importorskip("transformers")
    from oracle import umt5_oracle as O
    cfg = dict(vocab_size=1000, d_model=256, d_kv=64, d_ff=512, num_layers=3, num_heads=4)
    P = O.make_params(cfg, seed=11)
    m = _build(cfg, P)
    c = transformers.UMT5Config(vocab_size=1000, d_model=256, d_kv=64, d_ff=512, num_layers=3, num_heads=4,
                                feed_forward_proj="gated-gelu", dropout_rate=0.0)
    hf = transformers.UMT5EncoderModel(c).eval()
    sd = {k: v.float() for k, v in P.items()}; sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    hf.load_state_dict(sd, strict=False)
    ids = torch.randint(0, 1000, (2, 80), generator=torch.Generator().manual_seed(5))
    mask = torch.ones(2, 80, dtype=torch.long); mask[1, 50:] = 0
    with torch.no_grad():
        ref32 = hf(ids, mask).last_hidden_state
        ref16 = hf.to(BF16)(ids, mask).last_hidden_state.float()
    out = m(ids.cuda(), mask.cuda()).last_hidden_state
    keep = mask.bool()
    e16, e32, scale = rel_l2(out[keep], ref16[keep]), rel_l2(out[keep], ref32[keep]), rel_l2(ref16[keep], ref32[keep])
    print(f"HIP vs HF bf16 {e16:.2e}; HIP vs HF fp32 {e32:.2e}; HF bf16 vs HF fp32 {scale:.2e}")
    assert e32 < 1.5 * scale + 3e-3                                     # no further from the exact model than HF's own bf16 run


def test_pipeline_encode_prompt_with_the_hip_encoder():
    from longcat_video.modules.umt5_encoder import UMT5EncoderModel
    from longcat_video.pipeline_longcat_video import LongCatVideoPipeline

    class Tok:                                                            # the call the pipeline makes on a HF tokenizer
        def __call__(self, texts, padding=None, max_length=512, truncation=True, add_special_tokens=True,
                     return_attention_mask=True, return_tensors="pt"):
            n = min(len(texts[0].split()) + 1, max_length)
            ids = torch.zeros(1, max_length, dtype=torch.long); ids[0, :n] = torch.arange(1, n + 1)
            m = torch.zeros(1, max_length, dtype=torch.long); m[0, :n] = 1
            from types import SimpleNamespace
            return SimpleNamespace(input_ids=ids, attention_mask=m)

    enc = UMT5EncoderModel(device="cuda", vocab_size=512, d_model=256, d_kv=64, d_ff=512, num_layers=2, num_heads=4).init_synthetic_()
    pipe = LongCatVideoPipeline(tokenizer=Tok(), text_encoder=enc)
    pipe.device = torch.device("cuda")
    emb, mask = pipe.encode_prompt("a cat walks across the frame")
    assert emb.shape == (1, 1, 512, 256) and emb.dtype == BF16 and int(mask.sum()) == 7
    assert torch.isfinite(emb.float()).all()
