"""Module-level parity through the C ABI: HIP engine vs the CPU oracle on the same seeded inputs.

Stated tolerances: fp32 engine, one UNet forward: max-abs <= 1e-3 * max|y|; short loop
with injected identical noise: latents max-abs <= 1e-2; int16 wave: <= 1 LSB on >= 99.9 % of samples
given the same mel (fp32).  fp16 engine (fp16 storage / MFMA operands, fp32 accumulate + statistics):
one UNet forward <= 3e-2 * max|y|."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tango_oracle as O  # noqa: E402  (checker only)
from tango_amd import weights as W  # noqa: E402
from tango_amd.engine import Engine  # noqa: E402
from tango_amd.scheduler import DDIMScheduler, DDPMScheduler, SD21_SCHEDULER_CONFIG  # noqa: E402

UTOL = {"fp32": 1e-3, "fp16": 3e-2, "bf16": 2e-1}
_cache = {}


def unet_engine(name, dtype):
    key = ("unet", name, dtype)
    if key not in _cache:
        cfg = {"tiny": O.UNET_CONFIG_TINY, "large": O.UNET_CONFIG_LARGE}[name]
        e = Engine(unet=cfg, dtype=dtype)
        e.load_synthetic(1234)
        _cache[key] = e
    return _cache[key]


def unet_sd(name):
    key = ("unet_sd", name)
    if key not in _cache:
        cfg = {"tiny": O.UNET_CONFIG_TINY, "large": O.UNET_CONFIG_LARGE}[name]
        _cache[key] = W.synth_state_dict(W.unet_param_shapes(cfg, "unet."), 1234)
    return _cache[key]


def text_inputs(B2, L, d, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    enc = torch.randn(B2, L, d, generator=g)
    mask = torch.ones(B2, L, dtype=torch.bool)
    if ragged:
        mask[0, 1:] = False            # the uncond row of T5("") attends to token 0 only (models.py:282-289)
        if B2 > 2:
            mask[2, L // 2:] = False
    return enc, mask


def relerr(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-9)).item()


@pytest.mark.parametrize("dtype", ["fp32", "fp16", "bf16"])
def test_unet_forward_tiny(dtype):
    cfg = O.UNET_CONFIG_TINY
    e = unet_engine("tiny", dtype)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 8, 256, 16, generator=g)
    enc, mask = text_inputs(4, 13, cfg["cross_attention_dim"], 12)
    ref = O.unet_forward(unet_sd("tiny"), cfg, x, 801, enc, mask, prefix="unet.")
    out = e.unet_forward(x.cuda(), 801, enc.cuda(), mask.cuda()).cpu()
    err = relerr(out, ref)
    print("unet tiny %s rel err %.3e" % (dtype, err))
    assert err <= UTOL[dtype]
    # no-mask path and another timestep
    ref2 = O.unet_forward(unet_sd("tiny"), cfg, x[:2], 5, enc[:2], None, prefix="unet.")
    out2 = e.unet_forward(x[:2].cuda(), 5, enc[:2].cuda(), None).cpu()
    assert relerr(out2, ref2) <= UTOL[dtype]


def test_mis_shaped_inputs_raise_instead_of_reading_out_of_bounds():
    """the C ABI takes raw pointers and sizes the reads from its plan, so the shim must reject tensors of any other shape"""
    cfg = O.UNET_CONFIG_TINY
    e = unet_engine("tiny", "fp32")
    d = cfg["cross_attention_dim"]
    x = torch.randn(2, 8, 256, 16).cuda()
    enc, mask = text_inputs(2, 13, d, 12)
    enc, mask = enc.cuda(), mask.cuda()
    for bad in (x[:, :, :128], x[:, :4], x[0]):
        with pytest.raises(ValueError):
            e.unet_forward(bad, 5, enc, mask)
    with pytest.raises(ValueError):
        e.unet_forward(x, 5, enc[..., : d - 8].contiguous(), mask)          # wrong embedding width
    with pytest.raises(ValueError):
        e.unet_forward(x, 5, enc[:1], mask[:1])                              # rows != batch
    with pytest.raises(ValueError):
        e.unet_forward(x, 5, enc, mask[:, :5])                               # mask of another length
    sch = DDPMScheduler.from_config({k: SD21_SCHEDULER_CONFIG[k] for k in
                                     ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample", "variance_type")})
    sch.set_timesteps(2)
    lat = torch.randn(1, 8, 256, 16).cuda()
    args = (sch.timesteps.numpy(), sch.coef_table(), 3.0)
    with pytest.raises(ValueError):
        e.denoise(lat, enc[:1], mask[:1], *args)                             # CFG needs [uncond; cond] = 2 rows per latent
    with pytest.raises(ValueError):
        e.denoise(lat, enc, mask, *args, noise=torch.randn(2, 1, 8, 128, 16).cuda())
    with pytest.raises(ValueError):
        e.denoise(torch.randn(1, 8, 128, 16).cuda(), enc, mask, *args)
    e.denoise(lat, enc, mask, *args, noise=torch.randn(2, 1, 8, 256, 16).cuda())   # the well-formed call still runs


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_unet_forward_large(dtype):
    """Full Tango UNet (866 M params, configs/diffusion_model_config.json), one CFG pair, L = 64."""
    cfg = O.UNET_CONFIG_LARGE
    e = unet_engine("large", dtype)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 8, 256, 16, generator=g)
    enc, mask = text_inputs(2, 64, 1024, 22)
    ref = O.unet_forward(unet_sd("large"), cfg, x, 995, enc, mask, prefix="unet.")
    out = e.unet_forward(x.cuda(), 995, enc.cuda(), mask.cuda()).cpu()
    err = relerr(out, ref)
    print("unet large %s rel err %.3e" % (dtype, err))
    assert err <= UTOL[dtype]
    _cache.pop(("unet", "large", dtype), None)   # free 3.5/1.7 GB of packed weights


@pytest.mark.parametrize("name", ["tiny", "large"])
def test_unet_matches_reference_golden(name):
    """HIP engine (fp32) vs the committed outputs of the REAL reference UNet (fork UNet2DConditionModel run in the
    build container, tests/golden/unet_ref.npz) -- no oracle in the loop."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "unet_ref.npz"))
    cfg = {"tiny": O.UNET_CONFIG_TINY, "large": O.UNET_CONFIG_LARGE}[name]
    B2, L, t, seed = [int(v) for v in z[name + "/meta"]]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B2, 8, 256, 16, generator=g)
    enc = torch.randn(B2, L, cfg["cross_attention_dim"], generator=g)
    mask = torch.ones(B2, L, dtype=torch.bool)
    mask[0, 1:] = False
    if B2 > 2:
        mask[2, L // 2:] = False
    # the fixture used un-prefixed reference key names for the per-tensor seeds (oracle/make_golden.py)
    e = Engine(unet=cfg, dtype="fp32")
    e.load_state_dict({"unet." + k: v for k, v in W.iter_synth(W.unet_param_shapes(cfg), 1234)})
    e.finalize()
    out = e.unet_forward(x.cuda(), t, enc.cuda(), mask.cuda()).cpu()
    ref = z[name + "/slice"]
    err = np.abs(out[:, :, ::37, ::5].numpy() - ref).max() / np.abs(ref).max()
    print("engine fp32 vs reference golden (%s): rel err %.3e" % (name, err))
    assert err <= 1e-4
    a = out.double()
    cs = np.asarray([float(a.sum()), float(a.abs().sum()), float((a * a).sum())])
    assert np.allclose(cs, z[name + "/checksum"], rtol=1e-4, atol=1e-1)


def test_vae_vocoder_match_reference_golden():
    """HIP engine (fp32) vs committed outputs of the reference AutoencoderKL (decode_first_stage / decode_to_waveform)"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vae_voc_ref.npz"))
    e = Engine(vae=O.VAE_CONFIG, hifigan=O.HIFIGAN_CONFIG, dtype="fp32")
    e.load_synthetic(1234)
    g = torch.Generator().manual_seed(41)
    lat = torch.randn(2, 8, 256, 16, generator=g)
    mel = e.vae_decode(lat.cuda())
    ref = z["mel_slice"]
    assert np.abs(mel[:, 0, ::41, ::3].cpu().numpy() - ref).max() / np.abs(ref).max() <= 1e-4
    wav = e.vocode(mel).cpu().numpy()
    d = np.abs(wav[:, :4096].astype(np.int32) - z["wav_head"].astype(np.int32))
    assert (d <= 1).mean() >= 0.995 and d.max() <= 3, (d.max(), (d <= 1).mean())


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("sched_name", ["ddpm", "ddim"])
def test_denoise_loop_tiny(dtype, sched_name):
    """models.py:224-249 with injected noise, 4 steps, guidance 3: engine loop (hipGraph and eager) vs oracle."""
    cfg = O.UNET_CONFIG_TINY
    e = unet_engine("tiny", dtype)
    B, L, N = 2, 9, 4
    enc, mask = text_inputs(2 * B, L, cfg["cross_attention_dim"], 31)
    g = torch.Generator().manual_seed(32)
    lat0 = torch.randn(B, 8, 256, 16, generator=g)
    noises = torch.randn(N, B, 8, 256, 16, generator=g)
    if sched_name == "ddpm":
        osch = O.DDPMOracle(**O.SD21_SCHEDULER)
        keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample", "variance_type")
        sch = DDPMScheduler.from_config({k: SD21_SCHEDULER_CONFIG[k] for k in keys})
    else:
        osch = O.DDIMOracle(**dict(O.SD21_SCHEDULER, set_alpha_to_one=False, steps_offset=1))
        keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample", "set_alpha_to_one", "steps_offset")
        sch = DDIMScheduler.from_config({k: SD21_SCHEDULER_CONFIG[k] for k in keys})
    if sched_name == "ddpm":
        ref = O.denoise_loop(unet_sd("tiny"), cfg, osch, enc, mask, lat0.clone(), N, 3.0, noises=list(noises), prefix="unet.")
    else:
        osch.set_timesteps(N)
        lat = lat0.clone()
        for t in osch.timesteps:
            out = O.unet_forward(unet_sd("tiny"), cfg, torch.cat([lat] * 2), t, enc, mask, prefix="unet.")
            u, c = out.chunk(2)
            lat = osch.step(u + 3.0 * (c - u), t, lat)
        ref = lat
    sch.set_timesteps(N)
    assert sch.timesteps.tolist() == osch.timesteps.tolist()
    outs = []
    for use_graph in (True, False):
        lat = lat0.clone().cuda()
        e.denoise(lat, enc.cuda(), mask.cuda(), sch.timesteps.numpy(), sch.coef_table(), 3.0,
                  prediction_type="v_prediction", rule=sch.rule, noise=noises.cuda() if sched_name == "ddpm" else None,
                  use_graph=use_graph)
        torch.cuda.synchronize()
        outs.append(lat.cpu())
    assert torch.equal(outs[0], outs[1]), "hipGraph replay and eager launches must agree bit for bit"
    err = (outs[0] - ref).abs().max().item()
    print("denoise %s %s max abs err %.3e (|ref| max %.2f)" % (sched_name, dtype, err, ref.abs().max()))
    assert err <= (1e-2 if dtype == "fp32" else 1e-1)
    tot, per = e.last_denoise_ms()
    assert tot > 0 and per > 0


@pytest.mark.parametrize("eta", [0.5, 1.0])
def test_denoise_loop_ddim_eta(eta):
    """DDIM with eta > 0 (scheduling_ddim.py:316-352; audioldm/latent_diffusion/ddim.py:356-370): the stochastic term uses the
    step-noise source like the DDPM rule's; injected noise, fp32 engine vs the oracle's DDIM step"""
    cfg = O.UNET_CONFIG_TINY
    e = unet_engine("tiny", "fp32")
    B, L, N = 2, 9, 4
    enc, mask = text_inputs(2 * B, L, cfg["cross_attention_dim"], 41)
    g = torch.Generator().manual_seed(42)
    lat0 = torch.randn(B, 8, 256, 16, generator=g)
    noises = torch.randn(N, B, 8, 256, 16, generator=g)
    osch = O.DDIMOracle(**dict(O.SD21_SCHEDULER, set_alpha_to_one=False, steps_offset=1, eta=eta))
    keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample", "set_alpha_to_one", "steps_offset")
    sch = DDIMScheduler.from_config(dict({k: SD21_SCHEDULER_CONFIG[k] for k in keys}, eta=eta))
    ref = O.denoise_loop(unet_sd("tiny"), cfg, osch, enc, mask, lat0.clone(), N, 3.0, noises=list(noises), prefix="unet.")
    sch.set_timesteps(N)
    assert sch.timesteps.tolist() == osch.timesteps.tolist() and sch.coef_table()[:, 4].min() > 0
    lat = lat0.clone().cuda()
    e.denoise(lat, enc.cuda(), mask.cuda(), sch.timesteps.numpy(), sch.coef_table(), 3.0, prediction_type="v_prediction", rule="ddim",
              noise=noises.cuda())
    err = (lat.cpu() - ref).abs().max().item()
    det = lat0.clone().cuda()
    d0 = DDIMScheduler.from_config({k: SD21_SCHEDULER_CONFIG[k] for k in keys})
    d0.set_timesteps(N)
    e.denoise(det, enc.cuda(), mask.cuda(), d0.timesteps.numpy(), d0.coef_table(), 3.0, prediction_type="v_prediction", rule="ddim")
    print("DDIM eta=%.1f fp32: max abs err vs oracle %.3e; distance from the eta = 0 trajectory %.2f" % (eta, err, (lat - det).abs().max().item()))
    assert err <= 1e-2 and (lat - det).abs().max().item() > 0.1


@pytest.mark.parametrize("guidance,pred,B", [(1.0, "v_prediction", 3), (0.0, "v_prediction", 1), (3.0, "epsilon", 3), (7.5, "sample", 1)])
def test_denoise_loop_edge_cases(guidance, pred, B):
    """models.py:224-249 off the beaten path: guidance <= 1 runs WITHOUT the unconditional twin (UNet batch B, not 2B;
    models.py:231,236), odd batches, epsilon / sample prediction types (scheduling_ddpm.py:299-309)."""
    cfg = O.UNET_CONFIG_TINY
    e = unet_engine("tiny", "fp32")
    L, N = 7, 3
    nb = 2 * B if guidance > 1.0 else B
    enc, mask = text_inputs(nb, L, cfg["cross_attention_dim"], 57 + B)
    g = torch.Generator().manual_seed(58)
    lat0 = torch.randn(B, 8, 256, 16, generator=g)
    noises = torch.randn(N, B, 8, 256, 16, generator=g)
    conf = dict(O.SD21_SCHEDULER, prediction_type=pred)
    osch = O.DDPMOracle(**conf)
    keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample", "variance_type")
    sch = DDPMScheduler.from_config(dict({k: SD21_SCHEDULER_CONFIG[k] for k in keys}, prediction_type=pred))
    ref = O.denoise_loop(unet_sd("tiny"), cfg, osch, enc, mask, lat0.clone(), N, guidance, noises=list(noises), prefix="unet.")
    sch.set_timesteps(N)
    outs = []
    for use_graph in (True, False):
        lat = lat0.clone().cuda()
        e.denoise(lat, enc.cuda(), mask.cuda(), sch.timesteps.numpy(), sch.coef_table(), guidance,
                  prediction_type=pred, rule=sch.rule, noise=noises.cuda(), use_graph=use_graph)
        torch.cuda.synchronize()
        outs.append(lat.cpu())
    assert torch.equal(outs[0], outs[1])
    err = (outs[0] - ref).abs().max().item()
    print("denoise edge g=%s %s B=%d max abs err %.3e (|ref| max %.2f)" % (guidance, pred, B, err, ref.abs().max()))
    assert err <= 1e-2 * max(1.0, ref.abs().max().item() / 4)


def test_denoise_rejects_bad_arguments():
    """error behaviour at the boundary: wrong embedding batch for the CFG mode, more than 1000 steps (scheduling_ddpm.py:193-198)"""
    cfg = O.UNET_CONFIG_TINY
    e = unet_engine("tiny", "fp32")
    keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample", "variance_type")
    sch = DDPMScheduler.from_config({k: SD21_SCHEDULER_CONFIG[k] for k in keys})
    with pytest.raises(ValueError):
        sch.set_timesteps(1001)
    sch.set_timesteps(2)
    enc, mask = text_inputs(2, 5, cfg["cross_attention_dim"], 3)
    lat = torch.randn(2, 8, 256, 16).cuda()
    with pytest.raises((ValueError, RuntimeError)):      # guidance 3 needs [uncond; cond] = 4 rows for 2 latents
        e.denoise(lat, enc.cuda(), mask.cuda(), sch.timesteps.numpy(), sch.coef_table(), 3.0, rule=sch.rule, seed=1)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_vae_and_vocoder(dtype):
    """decode_first_stage + decode_to_waveform: full-size mel-VAE decoder and HiFi-GAN (B = 2)."""
    shapes = W.vae_decoder_param_shapes(O.VAE_CONFIG)
    shapes.update(W.hifigan_param_shapes(O.HIFIGAN_CONFIG))
    sd = W.synth_state_dict(shapes, 1234)
    e = Engine(vae=O.VAE_CONFIG, hifigan=O.HIFIGAN_CONFIG, dtype=dtype)
    e.load_synthetic(1234)
    g = torch.Generator().manual_seed(41)
    z = torch.randn(2, 8, 256, 16, generator=g)
    mel_ref = O.vae_decode_first_stage(sd, O.VAE_CONFIG, z)
    mel = e.vae_decode(z.cuda())
    assert mel.shape == (2, 1, 1024, 64)
    err = relerr(mel.cpu(), mel_ref)
    print("vae %s rel err %.3e" % (dtype, err))
    assert err <= (1e-3 if dtype == "fp32" else 3e-2)
    # vocoder on the ORACLE mel so the comparison isolates HiFi-GAN + int16 cast
    wav_ref = O.decode_to_waveform(sd, O.HIFIGAN_CONFIG, mel_ref)
    wav = e.vocode(mel_ref.cuda()).cpu().numpy()
    assert wav.dtype == np.int16 and wav.shape == (2, 163872) == wav_ref.shape
    d = np.abs(wav.astype(np.int32) - wav_ref.astype(np.int32))
    frac1 = float((d <= 1).mean())
    print("vocoder %s: |diff| max %d, <=1 LSB on %.4f, ref std %.0f" % (dtype, d.max(), frac1, wav_ref.std()))
    if dtype == "fp32":
        assert frac1 >= 0.999
    else:
        snr = 10 * np.log10((wav_ref.astype(np.float64) ** 2).mean() / ((d.astype(np.float64) ** 2).mean() + 1e-9))
        print("vocoder fp16 SNR %.1f dB" % snr)
        assert snr >= 30.0


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_vae_encoder(dtype):
    """AutoencoderKL.encode_first_stage / get_first_stage_encoding: full-size mel-VAE encoder (B = 2) on the
    engine vs the oracle, and (fp32) vs the committed outputs of the reference itself (tests/golden/vae_enc_ref.npz)"""
    import os
    from tango_amd.autoencoder import AutoencoderKL, DiagonalGaussianDistribution
    sd = W.synth_state_dict(W.vae_encoder_param_shapes(O.VAE_CONFIG), 4321)
    vae = AutoencoderKL(ddconfig=dict(O.VAE_CONFIG, resolution=256, in_channels=1, double_z=True, attn_resolutions=[], dropout=0.0),
                        embed_dim=8, scale_factor=O.VAE_CONFIG["scale_factor"], dtype=dtype, with_encoder=True)
    vae.engine.load_synthetic(4321)
    g = torch.Generator().manual_seed(43)
    mel = torch.randn(2, 1, 1024, 64, generator=g) * 2.0 - 4.0
    ref = O.vae_encode_moments(sd, O.VAE_CONFIG, mel)
    post = vae.encode_first_stage(mel.cuda())
    assert isinstance(post, DiagonalGaussianDistribution) and post.parameters.shape == (2, 16, 256, 16)
    err = relerr(post.parameters.cpu(), ref)
    print("vae encoder %s rel err %.3e" % (dtype, err))
    assert err <= (1e-3 if dtype == "fp32" else 3e-2)
    torch.manual_seed(5)
    z = vae.get_first_stage_encoding(post)                 # draws randn(mean.shape) from the global CPU generator, like the reference
    torch.manual_seed(5)
    z_ref = O.vae_get_first_stage_encoding(ref, O.VAE_CONFIG, torch.randn(2, 8, 256, 16))
    assert z.shape == (2, 8, 256, 16) and relerr(z.cpu(), z_ref) <= (1e-3 if dtype == "fp32" else 3e-2)
    if dtype == "fp32":
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vae_enc_ref.npz"))
        m = gold["mom_slice"]
        assert np.abs(post.parameters[:, :, ::17, ::3].cpu().numpy() - m).max() / np.abs(m).max() <= 1e-4
        zz = gold["z_slice"]
        assert np.abs(z[:, :, ::17, ::3].cpu().numpy() - zz).max() / np.abs(zz).max() <= 1e-4


def test_generate_api_shapes():
    """Tango.generate_from_embeddings: tiny UNet + real VAE/vocoder, 2 prompts, 3 steps, device Philox noise."""
    from tango_amd.autoencoder import AutoencoderKL
    from tango_amd.models import AudioDiffusion
    from tango_amd.tango import Tango
    model = AudioDiffusion(unet_config=O.UNET_CONFIG_TINY, dtype="fp16")
    model.engine.load_synthetic(1234)
    vae = AutoencoderKL(ddconfig=dict(O.VAE_CONFIG, resolution=256, in_channels=1, double_z=True, attn_resolutions=[], dropout=0.0),
                        embed_dim=8, scale_factor=O.VAE_CONFIG["scale_factor"], dtype="fp16")
    vae.engine.load_synthetic(1234)
    t = Tango.from_components(model, vae)
    enc, mask = text_inputs(4, 20, O.UNET_CONFIG_TINY["cross_attention_dim"], 51)
    torch.manual_seed(0)   # prepare_latents draws from the global generator like the reference (models.py:261)
    w1 = t.generate_from_embeddings(enc.cuda(), mask.cuda(), steps=3, guidance=3, seed=7)
    torch.manual_seed(0)
    w2 = t.generate_from_embeddings(enc.cuda(), mask.cuda(), steps=3, guidance=3, seed=7)
    assert w1.dtype == np.int16 and w1.shape == (2, 163872)
    assert np.array_equal(w1, w2), "same seed -> same audio"
    assert vae.device().type == "cuda" and model.unet.config.in_channels == 8
    with pytest.raises(ValueError):
        t.scheduler.set_timesteps(1001)


def test_plan_cache_lru_budget():
    """Serving hygiene (VERDICT r3 missing #7): UNet plans are cached per (batch, text length, ...) and share a byte budget; when a
    new shape does not fit, least recently used plans are freed -- a caller with ragged last batches and mixed prompt lengths
    (tango.py:51-64) neither accumulates plans without bound nor sees different results after an eviction."""
    cfg = O.UNET_CONFIG_TINY
    e = Engine(unet=cfg, dtype="fp32")
    e.load_synthetic(1234)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 256, 16, generator=g)

    def run(L):
        enc, mask = text_inputs(2, L, cfg["cross_attention_dim"], 100 + L)
        return e.unet_forward(x.cuda(), 321, enc.cuda(), mask.cuda()).cpu()

    first = {L: run(L) for L in (8, 16)}
    used2, n2 = e.plan_stats()
    assert n2 == 2 and used2 > 0
    e.set_plan_budget(int(used2 * 1.2))            # room for two plans of this size, not three
    out24 = run(24)
    used, n = e.plan_stats()
    assert n == 2 and used <= int(used2 * 1.2) + (1 << 20), (used, used2, n)     # L = 8 (least recently used) was dropped
    assert torch.equal(run(16), first[16])         # still cached
    assert torch.equal(run(8), first[8])           # rebuilt after its eviction: same bits
    assert torch.equal(run(24), out24)
    used, n = e.plan_stats()
    assert n == 2
    e.set_plan_budget(1)                           # a budget nothing fits into: the current call still gets its plan
    assert torch.equal(run(16), first[16])
    assert e.plan_stats()[1] == 1
