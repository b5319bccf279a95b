"""Parameter inventories (reference state-dict key names + shapes) and seeded synthetic weights.

The engine ingests exactly the tensors the reference checkpoints carry:
  * `pytorch_model_main.bin`  -> keys `unet.*`           (AudioDiffusion.state_dict, tango.py:24,28)
  * `pytorch_model_vae.bin`   -> keys `post_quant_conv.* decoder.* vocoder.*` (+ unused encoder.*)
Key patterns: SURVEY.md Appendix B; module construction order follows the fork's
UNet2DConditionModel (mustango/diffusers/src/diffusers/models/unet_2d_condition.py:116-330),
audioldm Decoder (audioldm/variational_autoencoder/modules.py:546-648) and HiFi-GAN Generator
(audioldm/hifigan/models.py:112-147).

No real checkpoint is reachable offline, so benchmarks and parity tests use *seeded synthetic*
weights: every tensor is drawn from its own generator keyed by (seed, crc32(name)), so the CPU
oracle and the HIP engine can materialise identical tensors independently, one at a time.
"""
import zlib
from collections import OrderedDict
from typing import Dict, Iterator, Tuple

import torch

Shapes = "OrderedDict[str, Tuple[int, ...]]"


def _resnet(out, p, cin, cout, temb):
    out[p + ".norm1.weight"] = (cin,)
    out[p + ".norm1.bias"] = (cin,)
    out[p + ".conv1.weight"] = (cout, cin, 3, 3)
    out[p + ".conv1.bias"] = (cout,)
    out[p + ".time_emb_proj.weight"] = (cout, temb)
    out[p + ".time_emb_proj.bias"] = (cout,)
    out[p + ".norm2.weight"] = (cout,)
    out[p + ".norm2.bias"] = (cout,)
    out[p + ".conv2.weight"] = (cout, cout, 3, 3)
    out[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        out[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
        out[p + ".conv_shortcut.bias"] = (cout,)


def _transformer(out, p, c, cross):
    out[p + ".norm.weight"] = (c,)
    out[p + ".norm.bias"] = (c,)
    out[p + ".proj_in.weight"] = (c, c)
    out[p + ".proj_in.bias"] = (c,)
    b = p + ".transformer_blocks.0"
    out[b + ".attn1.to_q.weight"] = (c, c)
    out[b + ".attn1.to_k.weight"] = (c, c)
    out[b + ".attn1.to_v.weight"] = (c, c)
    out[b + ".attn1.to_out.0.weight"] = (c, c)
    out[b + ".attn1.to_out.0.bias"] = (c,)
    out[b + ".ff.net.0.proj.weight"] = (8 * c, c)
    out[b + ".ff.net.0.proj.bias"] = (8 * c,)
    out[b + ".ff.net.2.weight"] = (c, 4 * c)
    out[b + ".ff.net.2.bias"] = (c,)
    out[b + ".attn2.to_q.weight"] = (c, c)
    out[b + ".attn2.to_k.weight"] = (c, cross)
    out[b + ".attn2.to_v.weight"] = (c, cross)
    out[b + ".attn2.to_out.0.weight"] = (c, c)
    out[b + ".attn2.to_out.0.bias"] = (c,)
    for n in ("norm1", "norm2", "norm3"):
        out[b + "." + n + ".weight"] = (c,)
        out[b + "." + n + ".bias"] = (c,)
    out[p + ".proj_out.weight"] = (c, c)
    out[p + ".proj_out.bias"] = (c,)


def unet_param_shapes(cfg: dict, prefix: str = "") -> Shapes:
    """All UNet tensors for a Tango-style config (686 tensors for configs/diffusion_model_config.json; 1518 for Mustango's
    mustango/configs/music_diffusion_model_config.json, whose *Music blocks carry `attentions2` / `attentions3` next to
    `attentions` at every cross-attention site)."""
    music = any(str(t).endswith("Music") for t in list(cfg["down_block_types"]) + list(cfg["up_block_types"]))

    def _site(out, p, j, c, cross):
        for name in (("attentions", "attentions2", "attentions3") if music else ("attentions",)):
            _transformer(out, f"{p}.{name}.{j}", c, cross)
    ch = list(cfg["block_out_channels"])
    cross = cfg["cross_attention_dim"]
    lpb = cfg.get("layers_per_block", 2)
    temb = ch[0] * 4
    out: Shapes = OrderedDict()
    P = prefix
    out[P + "conv_in.weight"] = (ch[0], cfg["in_channels"], 3, 3)
    out[P + "conv_in.bias"] = (ch[0],)
    out[P + "time_embedding.linear_1.weight"] = (temb, ch[0])
    out[P + "time_embedding.linear_1.bias"] = (temb,)
    out[P + "time_embedding.linear_2.weight"] = (temb, temb)
    out[P + "time_embedding.linear_2.bias"] = (temb,)
    c_prev = ch[0]
    for i, bt in enumerate(cfg["down_block_types"]):
        for j in range(lpb):
            cin = c_prev if j == 0 else ch[i]
            _resnet(out, f"{P}down_blocks.{i}.resnets.{j}", cin, ch[i], temb)
            if bt in ("CrossAttnDownBlock2D", "CrossAttnDownBlock2DMusic"):
                _site(out, f"{P}down_blocks.{i}", j, ch[i], cross)
        if i != len(ch) - 1:
            out[f"{P}down_blocks.{i}.downsamplers.0.conv.weight"] = (ch[i], ch[i], 3, 3)
            out[f"{P}down_blocks.{i}.downsamplers.0.conv.bias"] = (ch[i],)
        c_prev = ch[i]
    cm = ch[-1]
    _resnet(out, P + "mid_block.resnets.0", cm, cm, temb)
    _site(out, P + "mid_block", 0, cm, cross)
    _resnet(out, P + "mid_block.resnets.1", cm, cm, temb)
    rch = list(reversed(ch))
    prev_out = rch[0]
    for i, bt in enumerate(cfg["up_block_types"]):
        outc = rch[i]
        inc = rch[min(i + 1, len(ch) - 1)]
        for j in range(lpb + 1):
            skip = inc if j == lpb else outc
            rin = prev_out if j == 0 else outc
            _resnet(out, f"{P}up_blocks.{i}.resnets.{j}", rin + skip, outc, temb)
            if bt in ("CrossAttnUpBlock2D", "CrossAttnUpBlock2DMusic"):
                _site(out, f"{P}up_blocks.{i}", j, outc, cross)
        if i != len(ch) - 1:
            out[f"{P}up_blocks.{i}.upsamplers.0.conv.weight"] = (outc, outc, 3, 3)
            out[f"{P}up_blocks.{i}.upsamplers.0.conv.bias"] = (outc,)
        prev_out = outc
    out[P + "conv_norm_out.weight"] = (ch[0],)
    out[P + "conv_norm_out.bias"] = (ch[0],)
    out[P + "conv_out.weight"] = (cfg["out_channels"], ch[0], 3, 3)
    out[P + "conv_out.bias"] = (cfg["out_channels"],)
    return out


def _vae_res(out, p, cin, cout):
    out[p + ".norm1.weight"] = (cin,)
    out[p + ".norm1.bias"] = (cin,)
    out[p + ".conv1.weight"] = (cout, cin, 3, 3)
    out[p + ".conv1.bias"] = (cout,)
    out[p + ".norm2.weight"] = (cout,)
    out[p + ".norm2.bias"] = (cout,)
    out[p + ".conv2.weight"] = (cout, cout, 3, 3)
    out[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        out[p + ".nin_shortcut.weight"] = (cout, cin, 1, 1)
        out[p + ".nin_shortcut.bias"] = (cout,)


def vae_encoder_param_shapes(cfg: dict, prefix: str = "") -> Shapes:
    """encoder.* + quant_conv (the encode-side subset of pytorch_model_vae.bin; reference modules.py:419-517,
    autoencoder.py:38).  `attn_resolutions` is empty in the Tango VAE config, so only mid.attn_1 exists."""
    ch, mult, nrb = cfg["ch"], list(cfg["ch_mult"]), cfg["num_res_blocks"]
    zc, ed = cfg["z_channels"], cfg.get("embed_dim", 8)
    out: Shapes = OrderedDict()
    E = prefix + "encoder."
    out[E + "conv_in.weight"] = (ch, cfg.get("in_channels", 1), 3, 3)
    out[E + "conv_in.bias"] = (ch,)
    bi = ch
    for lvl in range(len(mult)):
        bo = ch * mult[lvl]
        for b in range(nrb):
            _vae_res(out, f"{E}down.{lvl}.block.{b}", bi, bo)
            bi = bo
        if lvl != len(mult) - 1:
            out[f"{E}down.{lvl}.downsample.conv.weight"] = (bi, bi, 3, 3)
            out[f"{E}down.{lvl}.downsample.conv.bias"] = (bi,)
    _vae_res(out, E + "mid.block_1", bi, bi)
    a = E + "mid.attn_1"
    out[a + ".norm.weight"] = (bi,)
    out[a + ".norm.bias"] = (bi,)
    for n in ("q", "k", "v", "proj_out"):
        out[f"{a}.{n}.weight"] = (bi, bi, 1, 1)
        out[f"{a}.{n}.bias"] = (bi,)
    _vae_res(out, E + "mid.block_2", bi, bi)
    out[E + "norm_out.weight"] = (bi,)
    out[E + "norm_out.bias"] = (bi,)
    out[E + "conv_out.weight"] = (2 * zc, bi, 3, 3)
    out[E + "conv_out.bias"] = (2 * zc,)
    out[prefix + "quant_conv.weight"] = (2 * ed, 2 * zc, 1, 1)
    out[prefix + "quant_conv.bias"] = (2 * ed,)
    return out


def vae_decoder_param_shapes(cfg: dict, prefix: str = "") -> Shapes:
    """post_quant_conv + decoder.* (the decode-side subset of pytorch_model_vae.bin)."""
    ch, mult, nrb = cfg["ch"], list(cfg["ch_mult"]), cfg["num_res_blocks"]
    zc, ed = cfg["z_channels"], cfg.get("embed_dim", 8)
    out: Shapes = OrderedDict()
    P = prefix
    out[P + "post_quant_conv.weight"] = (zc, ed, 1, 1)
    out[P + "post_quant_conv.bias"] = (zc,)
    D = P + "decoder."
    bi = ch * mult[-1]
    out[D + "conv_in.weight"] = (bi, zc, 3, 3)
    out[D + "conv_in.bias"] = (bi,)
    _vae_res(out, D + "mid.block_1", bi, bi)
    a = D + "mid.attn_1"
    out[a + ".norm.weight"] = (bi,)
    out[a + ".norm.bias"] = (bi,)
    for n in ("q", "k", "v", "proj_out"):
        out[f"{a}.{n}.weight"] = (bi, bi, 1, 1)
        out[f"{a}.{n}.bias"] = (bi,)
    _vae_res(out, D + "mid.block_2", bi, bi)
    for lvl in reversed(range(len(mult))):
        bo = ch * mult[lvl]
        for b in range(nrb + 1):
            _vae_res(out, f"{D}up.{lvl}.block.{b}", bi, bo)
            bi = bo
        if lvl != 0:
            out[f"{D}up.{lvl}.upsample.conv.weight"] = (bi, bi, 3, 3)
            out[f"{D}up.{lvl}.upsample.conv.bias"] = (bi,)
    out[D + "norm_out.weight"] = (bi,)
    out[D + "norm_out.bias"] = (bi,)
    out[D + "conv_out.weight"] = (cfg["out_ch"], bi, 3, 3)
    out[D + "conv_out.bias"] = (cfg["out_ch"],)
    return out


def hifigan_param_shapes(cfg: dict, prefix: str = "vocoder.") -> Shapes:
    """vocoder.* of pytorch_model_vae.bin (weight-norm removed, hifigan/utilities.py:67-73)."""
    c0 = cfg["upsample_initial_channel"]
    out: Shapes = OrderedDict()
    P = prefix
    out[P + "conv_pre.weight"] = (c0, cfg["num_mels"], 7)
    out[P + "conv_pre.bias"] = (c0,)
    nk = len(cfg["resblock_kernel_sizes"])
    ch = c0
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        out[f"{P}ups.{i}.weight"] = (c0 // 2 ** i, c0 // 2 ** (i + 1), k)   # ConvTranspose1d [in,out,k]
        out[f"{P}ups.{i}.bias"] = (c0 // 2 ** (i + 1),)
    for i in range(len(cfg["upsample_rates"])):
        ch = c0 // 2 ** (i + 1)
        for j, ks in enumerate(cfg["resblock_kernel_sizes"]):
            rp = f"{P}resblocks.{i * nk + j}"
            for m in range(len(cfg["resblock_dilation_sizes"][j])):
                out[f"{rp}.convs1.{m}.weight"] = (ch, ch, ks)
                out[f"{rp}.convs1.{m}.bias"] = (ch,)
            for m in range(len(cfg["resblock_dilation_sizes"][j])):
                out[f"{rp}.convs2.{m}.weight"] = (ch, ch, ks)
                out[f"{rp}.convs2.{m}.bias"] = (ch,)
    out[P + "conv_post.weight"] = (1, ch, 7)
    out[P + "conv_post.bias"] = (1,)
    return out


#: google/flan-t5-large and -xl encoder hyper-parameters (their config.json; d_kv is 64 in both)
T5_CONFIG_LARGE = dict(vocab_size=32128, d_model=1024, d_kv=64, num_heads=16, d_ff=2816, num_layers=24,
                       relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6)
T5_CONFIG_XL = dict(T5_CONFIG_LARGE, d_model=2048, num_heads=32, d_ff=5120)


def t5_encoder_param_shapes(cfg: dict, prefix: str = "text_encoder.") -> Shapes:
    """transformers T5EncoderModel.state_dict() for a gated-gelu (v1.1 / FLAN) config, minus the tied `encoder.embed_tokens.weight`."""
    d, inner, dff = cfg["d_model"], cfg["num_heads"] * cfg["d_kv"], cfg["d_ff"]
    out: Shapes = {}
    out[prefix + "shared.weight"] = (cfg["vocab_size"], d)
    out[prefix + "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] = (cfg["relative_attention_num_buckets"], cfg["num_heads"])
    for i in range(cfg["num_layers"]):
        b = "%sencoder.block.%d" % (prefix, i)
        for n in ("q", "k", "v"):
            out["%s.layer.0.SelfAttention.%s.weight" % (b, n)] = (inner, d)
        out["%s.layer.0.SelfAttention.o.weight" % b] = (d, inner)
        out["%s.layer.0.layer_norm.weight" % b] = (d,)
        out["%s.layer.1.DenseReluDense.wi_0.weight" % b] = (dff, d)
        out["%s.layer.1.DenseReluDense.wi_1.weight" % b] = (dff, d)
        out["%s.layer.1.DenseReluDense.wo.weight" % b] = (d, dff)
        out["%s.layer.1.layer_norm.weight" % b] = (d,)
    out[prefix + "encoder.final_layer_norm.weight"] = (d,)
    return out


def stft_param_shapes(cfg: dict, prefix: str = "") -> Shapes:
    """buffers of audioldm/audio/stft.py TacotronSTFT the front-end reads (pytorch_model_stft.bin; `stft_fn.inverse_basis` is
    not on the wave -> mel path)"""
    cutoff = cfg["filter_length"] // 2 + 1
    return {prefix + "stft_fn.forward_basis": (2 * cutoff, 1, cfg["filter_length"]), prefix + "mel_basis": (cfg["n_mel_channels"], cutoff)}


def t5_config_from_state_dict(sd, prefix: str = "text_encoder.") -> dict:
    """Recover the encoder hyper-parameters from the tensors themselves (a checkpoint carries no config.json for them)."""
    d = sd[prefix + "shared.weight"].shape[1] if prefix + "shared.weight" in sd else sd[prefix + "encoder.embed_tokens.weight"].shape[1]
    vocab = (sd.get(prefix + "shared.weight", sd.get(prefix + "encoder.embed_tokens.weight"))).shape[0]
    rel = sd[prefix + "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
    inner = sd[prefix + "encoder.block.0.layer.0.SelfAttention.q.weight"].shape[0]
    layers = 1 + max(int(k[len(prefix):].split(".")[2]) for k in sd if k.startswith(prefix + "encoder.block."))
    if prefix + "encoder.block.0.layer.1.DenseReluDense.wi_0.weight" not in sd:
        raise ValueError("only gated-gelu T5 (v1.1 / FLAN-T5) encoders are supported")
    return dict(vocab_size=vocab, d_model=d, d_kv=inner // rel.shape[1], num_heads=rel.shape[1],
                d_ff=sd[prefix + "encoder.block.0.layer.1.DenseReluDense.wi_0.weight"].shape[0], num_layers=layers,
                relative_attention_num_buckets=rel.shape[0], relative_attention_max_distance=128, layer_norm_epsilon=1e-6)


def _is_norm(name: str) -> bool:
    leaf = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else name
    return leaf.startswith("norm") or leaf in ("conv_norm_out", "norm_out")


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 1234) -> torch.Tensor:
    """One seeded synthetic fp32 CPU tensor.  Matrices/filters ~ U(-b, b), b = gain/sqrt(fan_in)
    (PyTorch's default Linear/Conv bound, gain 1; vocoder filters use gain sqrt(3) so the waveform
    keeps a usable amplitude through the leaky-relu stack); norm gains ~ 1 + 0.1 N(0,1), norm biases
    and layer biases ~ 0.1-scaled so that every gamma/beta/bias path is exercised by parity tests."""
    g = torch.Generator()
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    if _is_norm(name):
        t = torch.randn(shape, generator=g)
        return (1.0 + 0.1 * t) if name.endswith(".weight") else 0.1 * t
    if len(shape) == 1:
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.05
    if ".ups." in name and len(shape) == 3:            # ConvTranspose1d [in, out, k]: fan_in = in*k/stride-ish
        fan_in = shape[0] * shape[2] / 4.0
    else:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
    gain = 1.0
    if name.startswith("vocoder.") or ".vocoder." in name:
        gain = 3.0 ** 0.5
    b = gain / (fan_in ** 0.5)
    return (torch.rand(shape, generator=g) * 2 - 1) * b


def synth_state_dict(shapes: Shapes, seed: int = 1234) -> Dict[str, torch.Tensor]:
    return OrderedDict((k, synth_tensor(k, s, seed)) for k, s in shapes.items())


def iter_synth(shapes: Shapes, seed: int = 1234) -> Iterator[Tuple[str, torch.Tensor]]:
    for k, s in shapes.items():
        yield k, synth_tensor(k, s, seed)
