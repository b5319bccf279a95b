"""FLAN-T5 encoder on the HIP engine vs `transformers.T5EncoderModel` (the class the reference
instantiates, models.py:98-100) with the same random-init weights, fp32 both sides.  Covers: relative position buckets up to
distance 199 (all 32 buckets), ragged attention masks, gated tanh-GELU feed-forward, RMS norms with non-trivial gains, the
tied-embedding key alias, and the `text_encoder="engine"` route of the drop-in API."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tango_amd import weights as W  # noqa: E402
from tango_amd.text_encoder import T5EncoderOnEngine  # noqa: E402


def hf_encoder(cfg, seed, hot=False):
    from transformers import T5Config, T5EncoderModel
    torch.manual_seed(seed)
    c = T5Config(vocab_size=cfg["vocab_size"], d_model=cfg["d_model"], d_kv=cfg["d_kv"], d_ff=cfg["d_ff"], num_layers=cfg["num_layers"],
                 num_heads=cfg["num_heads"], relative_attention_num_buckets=cfg["relative_attention_num_buckets"],
                 relative_attention_max_distance=cfg["relative_attention_max_distance"], layer_norm_epsilon=cfg["layer_norm_epsilon"],
                 feed_forward_proj="gated-gelu", tie_word_embeddings=False, dropout_rate=0.0)
    m = T5EncoderModel(c).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "layer_norm" in n:
                p.copy_(1.0 + 0.3 * torch.randn_like(p))                 # non-trivial RMSNorm gains
            elif "relative_attention_bias" in n:
                p.copy_(torch.randn_like(p))                              # make the position bias matter
            elif p.dim() == 2 and "shared" not in n and "embed" not in n:
                p.mul_(3.0)                                               # default init is tiny: scale up so attention is not uniform
        if hot:
            # FLAN-T5's residual stream is LARGE (hundreds to thousands by the last blocks: the reason its fp16 inference
            # overflows) -- random init never gets there.  Emulate it: embeddings x40 and the two residual-writing matrices
            # (attention `o`, feed-forward `wo`) x4, so the stream grows block over block and every RMSNorm divides a big
            # number: the regime in which error growth with depth would show.
            m.shared.weight.mul_(40.0)
            for n, p in m.named_parameters():
                if n.endswith("SelfAttention.o.weight") or n.endswith("DenseReluDense.wo.weight"):
                    p.mul_(4.0)
    return m


CFGS = {
    "tiny": dict(vocab_size=100, d_model=128, d_kv=64, num_heads=2, d_ff=256, num_layers=2, relative_attention_num_buckets=32,
                 relative_attention_max_distance=128, layer_norm_epsilon=1e-6),
    "large4": dict(W.T5_CONFIG_LARGE, vocab_size=512, num_layers=4),     # flan-t5-large widths, 4 of its 24 blocks
    "large24": dict(W.T5_CONFIG_LARGE, vocab_size=512),                  # flan-t5-large: all 24 blocks at full width
    "xl24": dict(W.T5_CONFIG_XL, vocab_size=512),                        # flan-t5-xl (configs/diffusion_model_xl_config.json's encoder):
}                                                                        # d_model 2048, 32 heads, d_ff 5120, 24 blocks


@pytest.mark.parametrize("name,B,L", [("tiny", 3, 37), ("tiny", 1, 200), ("large4", 2, 64), ("large4", 2, 130)])
def test_t5_encoder_matches_transformers(name, B, L):
    cfg = CFGS[name]
    ref = hf_encoder(cfg, 7)
    enc = T5EncoderOnEngine(cfg)
    enc.load_state_dict(ref.state_dict())          # includes the tied encoder.embed_tokens.weight alias
    g = torch.Generator().manual_seed(L)
    ids = torch.randint(2, cfg["vocab_size"], (B, L), generator=g)
    am = torch.ones(B, L, dtype=torch.long)
    if B > 1:
        am[1, L // 3:] = 0                          # ragged batch: padded keys must not be attended to
        ids[1, L // 3:] = 0
    with torch.no_grad():
        want = ref(input_ids=ids, attention_mask=am)[0]
    got = enc(input_ids=ids.cuda(), attention_mask=am.cuda())[0].cpu()
    assert got.shape == want.shape == (B, L, cfg["d_model"])
    valid = am.bool()
    err = ((got - want).abs()[valid].max() / want.abs()[valid].max()).item()
    print("T5 encoder on engine (%s, B=%d, L=%d): rel err vs transformers fp32 %.3e" % (name, B, L, err))
    assert err <= 2e-4
    # no mask at all (models.py never does this, the API allows it)
    if B == 1:
        got2 = enc(input_ids=ids.cuda())[0].cpu()
        assert ((got2 - want).abs().max() / want.abs().max()).item() <= 2e-4


@pytest.mark.parametrize("name,B,L,hot", [("large24", 2, 64, False), ("large24", 2, 64, True), ("xl24", 1, 64, True)])
def test_t5_encoder_full_depth(name, B, L, hot):
    """All 24 blocks at the real widths: error growth with depth, with a FLAN-like large residual stream
    (`hot`), at the large AND the XL widths (BASELINE config 5 names the FLAN-T5-XL encoder)."""
    cfg = CFGS[name]
    ref = hf_encoder(cfg, 11, hot=hot)
    enc = T5EncoderOnEngine(cfg)
    enc.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(2, cfg["vocab_size"], (B, L), generator=g)
    am = torch.ones(B, L, dtype=torch.long)
    if B > 1:
        am[1, 23:] = 0
        ids[1, 23:] = 0
    with torch.no_grad():
        out = ref(input_ids=ids, attention_mask=am, output_hidden_states=True)
    want = out[0]
    peak = max(h.abs().max().item() for h in out.hidden_states)
    del ref
    got = enc(input_ids=ids.cuda(), attention_mask=am.cuda())[0].cpu()
    valid = am.bool()
    err = ((got - want).abs()[valid].max() / want.abs()[valid].max()).item()
    print("T5 encoder on engine (%s%s, 24 blocks, B=%d, L=%d): rel err vs transformers fp32 %.3e; peak |hidden state| %.0f"
          % (name, " hot" if hot else "", B, L, err, peak))
    assert torch.isfinite(got).all() and err <= 5e-4
    if hot:
        assert peak > 500.0, "the 'hot' weights must actually produce a FLAN-like residual stream"


def test_t5_state_dict_errors():
    cfg = CFGS["tiny"]
    ref = hf_encoder(cfg, 3)
    enc = T5EncoderOnEngine(cfg)
    sd = dict(ref.state_dict())
    sd.pop("encoder.block.1.layer.1.DenseReluDense.wo.weight")
    with pytest.raises(RuntimeError, match="Missing key"):
        enc.load_state_dict(sd)
    sd = dict(ref.state_dict())
    sd["encoder.block.9.layer.0.layer_norm.weight"] = torch.ones(128)
    with pytest.raises(RuntimeError, match="unexpected"):
        enc.load_state_dict(sd)
    assert W.t5_config_from_state_dict({"text_encoder." + k: v for k, v in ref.state_dict().items()}) == cfg
