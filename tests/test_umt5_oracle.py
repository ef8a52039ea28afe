"""CPU: pin oracle/umt5_oracle.py against the real `transformers.UMT5EncoderModel` (the class the reference calls,
delta_experiment/scripts/common.py:62-64, 250) on random small encoders; and the bucket table against transformers' own."""
import pytest
import torch

transformers = pytest.importorskip("transformers")


def _hf(cfg, P, dtype):
    from transformers import UMT5Config, UMT5EncoderModel
    c = UMT5Config(vocab_size=cfg["vocab_size"], d_model=cfg["d_model"], d_kv=cfg["d_kv"], d_ff=cfg["d_ff"],
                   num_layers=cfg["num_layers"], num_heads=cfg["num_heads"], feed_forward_proj="gated-gelu",
                   dropout_rate=0.0, relative_attention_num_buckets=32, relative_attention_max_distance=128)
    m = UMT5EncoderModel(c).eval()
    sd = {k: v.float() for k, v in P.items()}
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("embed_tokens" in k for k in missing), (missing, unexpected)
    return m.to(dtype)


@pytest.mark.parametrize("S,npad", [(24, 0), (40, 13)])
def test_oracle_matches_transformers_fp32(S, npad):
    from oracle import umt5_oracle as O
    cfg = dict(vocab_size=300, d_model=64, d_kv=16, d_ff=96, num_layers=3, num_heads=4)
    P = O.make_params(cfg, seed=S)
    ids = torch.randint(0, 300, (2, S), generator=torch.Generator().manual_seed(1))
    mask = torch.ones(2, S, dtype=torch.long); mask[1, S - npad:] = 0
    with torch.no_grad():
        ref = _hf(cfg, P, torch.float32)(ids, mask).last_hidden_state
    got = O.encoder_forward({k: v.float() for k, v in P.items()}, cfg, ids, mask)
    keep = mask.bool()
    assert (got - ref)[keep].abs().max() < 2e-5 * ref.abs().max()
    assert (got - ref).abs().max() < 2e-5 * ref.abs().max()          # padded query rows too (they only ignore padded keys)


def test_oracle_bf16_rounding_points_track_transformers_bf16():
    from oracle import umt5_oracle as O
    cfg = dict(vocab_size=300, d_model=64, d_kv=16, d_ff=96, num_layers=2, num_heads=4)
    P = O.make_params(cfg, seed=9)
    ids = torch.randint(0, 300, (1, 32), generator=torch.Generator().manual_seed(2))
    mask = torch.ones(1, 32, dtype=torch.long); mask[0, 25:] = 0
    with torch.no_grad():
        ref = _hf(cfg, P, torch.bfloat16)(ids, mask).last_hidden_state.float()
    got = O.encoder_forward({k: v.float() for k, v in P.items()}, cfg, ids, mask, bf16=True)
    exact = O.encoder_forward({k: v.float() for k, v in P.items()}, cfg, ids, mask)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    print("bf16 oracle vs HF bf16:", rel(got, ref), " HF bf16 vs fp32:", rel(ref, exact))
    assert rel(got, ref) < 1.5 * rel(ref, exact) + 2e-3             # both are bf16 noise away from the exact result


def test_bucket_table_matches_transformers():
    from oracle import umt5_oracle as O
    from transformers import UMT5Config
    from transformers.models.umt5.modeling_umt5 import UMT5Attention
    att = UMT5Attention(UMT5Config(d_model=64, d_kv=16, num_heads=4, d_ff=64, num_layers=1, vocab_size=10),
                        has_relative_attention_bias=True)
    d = torch.arange(-600, 601)
    assert torch.equal(O.relative_position_bucket(d), att._relative_position_bucket(d))


def test_hip_encoder_loads_a_huggingface_checkpoint_unchanged(tmp_path):
    """`UMT5EncoderModel.from_pretrained(ckpt, subfolder="text_encoder")` (common.py:62-64) on a directory written by
    transformers' own `save_pretrained`: config keys, safetensors shards and parameter names must line up (loading only:
    the forward needs the GPU)."""
    from transformers import UMT5Config, UMT5EncoderModel as HF
    from longcat_video.modules.umt5_encoder import UMT5EncoderModel
    c = UMT5Config(vocab_size=320, d_model=128, d_kv=64, d_ff=192, num_layers=2, num_heads=2, feed_forward_proj="gated-gelu",
                   dropout_rate=0.0)
    hf = HF(c).eval().to(torch.bfloat16)
    hf.save_pretrained(tmp_path / "text_encoder", safe_serialization=True)
    m = UMT5EncoderModel.from_pretrained(str(tmp_path), subfolder="text_encoder", torch_dtype=torch.bfloat16)
    assert (m.config.vocab_size, m.config.d_model, m.config.d_ff, m.config.num_layers, m.config.num_heads) == (320, 128, 192, 2, 2)
    mine, theirs = m.state_dict(), hf.state_dict()
    assert set(mine) == {k for k in theirs if k != "encoder.embed_tokens.weight"}
    for k, v in mine.items():
        assert torch.equal(v, theirs[k]), k
    assert m.dtype == torch.bfloat16
