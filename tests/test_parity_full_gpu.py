"""Parity at BASELINE.json's own sizes (VERDICT r1 "next" item 1), through the C ABI.

  * config 1: Tango-full UNet (866 M), B = 1, 10 DDPM steps, guidance 3, injected noise -- fp32 engine vs the CPU oracle
    (`O.denoise_loop`): latents max-abs <= 1e-2, then mel-VAE + HiFi-GAN: int16 <= 1 LSB on >= 99.9 %.
  * the same run on the fp16 / bf16 engines: mel PSNR and waveform SNR against the fp32 ORACLE (SURVEY.md 8d asks for
    these instead of elementwise bounds for the reduced-precision ladder), with asserted floors.
  * XL (FLAN-T5-XL, cross_attention_dim 2048: configs/diffusion_model_xl_config.json) UNet forward, fp32 + fp16.
  * device Philox noise: moments of 1e6 draws, and sharded (sample_offset) runs == unsharded run, bitwise.
  * reloading weights on a live engine invalidates the time-embedding tables.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tango_oracle as O  # noqa: E402  (checker only)
from tango_amd import weights as W  # noqa: E402
from tango_amd.engine import Engine  # noqa: E402
from tango_amd.scheduler import SD21_SCHEDULER_CONFIG, DDPMScheduler  # noqa: E402

_KEYS = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample", "variance_type")
_ref = {}


def _sched():
    return DDPMScheduler.from_config({k: SD21_SCHEDULER_CONFIG[k] for k in _KEYS})


def _vv_sd():
    shapes = W.vae_decoder_param_shapes(O.VAE_CONFIG)
    shapes.update(W.hifigan_param_shapes(O.HIFIGAN_CONFIG))
    return W.synth_state_dict(shapes, 1234)


def config1_reference():
    """fp32 CPU oracle for BASELINE config 1 (computed once per session: ~10 full-size CFG UNet steps on the host)."""
    if "c1" not in _ref:
        cfg = O.UNET_CONFIG_LARGE
        sd = W.synth_state_dict(W.unet_param_shapes(cfg, "unet."), 1234)
        B, L, N = 1, 64, 10
        g = torch.Generator().manual_seed(101)
        cond = torch.randn(B, L, 1024, generator=g)
        unc = torch.randn(B, L, 1024, generator=g)
        enc = torch.cat([unc, cond])
        mask = torch.ones(2 * B, L, dtype=torch.bool)
        mask[:B, 1:] = False                     # T5("") rows attend to token 0 only (models.py:282-289)
        lat0 = torch.randn(B, 8, 256, 16, generator=g)
        noises = torch.randn(N, B, 8, 256, 16, generator=g)
        with torch.no_grad():
            lat = O.denoise_loop(sd, cfg, O.DDPMOracle(**O.SD21_SCHEDULER), enc, mask, lat0.clone(), N, 3.0, noises=list(noises),
                                 prefix="unet.")
            vsd = _vv_sd()
            mel = O.vae_decode_first_stage(vsd, O.VAE_CONFIG, lat)
            wav = O.decode_to_waveform(vsd, O.HIFIGAN_CONFIG, mel)
        _ref["c1"] = dict(enc=enc, mask=mask, lat0=lat0, noises=noises, lat=lat, mel=mel, wav=wav, N=N)
    return _ref["c1"]


def run_config1(dtype, attn_fp8=False):
    r = config1_reference()
    e = Engine(unet=O.UNET_CONFIG_LARGE, dtype=dtype, attn_fp8=attn_fp8)
    e.load_synthetic(1234)
    sch = _sched()
    sch.set_timesteps(r["N"])
    assert sch.timesteps.tolist() == [900, 800, 700, 600, 500, 400, 300, 200, 100, 0]
    lat = r["lat0"].clone().cuda()
    e.denoise(lat, r["enc"].cuda(), r["mask"].cuda(), sch.timesteps.numpy(), sch.coef_table(), 3.0, noise=r["noises"].cuda())
    torch.cuda.synchronize()
    del e
    ev = Engine(vae=O.VAE_CONFIG, hifigan=O.HIFIGAN_CONFIG, dtype=dtype)
    ev.load_synthetic(1234)
    mel = ev.vae_decode(lat)
    wav = ev.vocode(mel).cpu().numpy()
    return lat.cpu(), mel.cpu(), wav, ev


def psnr(x, ref):
    mse = ((x.double() - ref.double()) ** 2).mean().item()
    peak = (ref.max() - ref.min()).item()
    return 10 * np.log10(peak * peak / (mse + 1e-30))


def snr_db(x, ref):
    x, ref = x.astype(np.float64), ref.astype(np.float64)
    return 10 * np.log10((ref ** 2).mean() / (((x - ref) ** 2).mean() + 1e-30))


def test_config1_full_size_fp32():
    r = config1_reference()
    lat, mel, wav, ev = run_config1("fp32")
    err = (lat - r["lat"]).abs().max().item()
    print("config 1 (866M UNet, B=1, 10 DDPM steps, g=3) fp32 engine vs oracle: latents max abs err %.3e (|ref| max %.2f)"
          % (err, r["lat"].abs().max()))
    assert err <= 2e-5           # measured 5.7e-6: floors are <= 3x the measured value
    merr = ((mel - r["mel"]).abs().max() / r["mel"].abs().max()).item()
    print("config 1 fp32: mel rel err %.3e, mel PSNR %.1f dB" % (merr, psnr(mel, r["mel"])))
    assert merr <= 1e-3
    # vocoder on the ORACLE mel isolates HiFi-GAN + int16 cast (<= 1 LSB on >= 99.9 %); end to end is reported too
    w2 = ev.vocode(r["mel"].cuda()).cpu().numpy()
    d = np.abs(w2.astype(np.int32) - r["wav"].astype(np.int32))
    frac = float((d <= 1).mean())
    print("config 1 fp32: int16 from the oracle mel: <=1 LSB on %.5f (max %d); end-to-end waveform SNR %.1f dB"
          % (frac, d.max(), snr_db(wav, r["wav"])))
    assert w2.shape == (1, 163872) and frac >= 0.999
    assert snr_db(wav, r["wav"]) >= 84.0          # measured 90.5 dB


# measured on MI355X (rounds 2-3): fp16 4.6e-3 / 70.5 dB / 47.7 dB, bf16 3.6e-2 / 52.3 dB / 29.7 dB; the floors
# are 3x on the latents and 6 dB (2x in amplitude) on the mel / waveform measures (no looser than 3x)
@pytest.mark.parametrize("dtype,lat_tol,mel_floor,wav_floor", [("fp16", 1.4e-2, 64.5, 41.7), ("bf16", 1.1e-1, 46.3, 23.7)])
def test_config1_reduced_precision_ladder(dtype, lat_tol, mel_floor, wav_floor):
    r = config1_reference()
    lat, mel, wav, _ = run_config1(dtype)
    err = (lat - r["lat"]).abs().max().item()
    p, s = psnr(mel, r["mel"]), snr_db(wav, r["wav"])
    print("config 1 %s engine vs fp32 oracle: latents max abs err %.3e, mel PSNR %.1f dB, waveform SNR %.1f dB" % (dtype, err, p, s))
    assert err <= lat_tol and p >= mel_floor and s >= wav_floor


def test_config5_precision_bf16_with_fp8_attention():
    """BASELINE config 5's precision ("bf16 + fp8 MFMA attention") on the full-size config-1 run: bf16 engine
    with the self-attention P.V products on the fp8 MFMA, against the fp32 oracle -- the same ladder as the bf16 row, so the
    two lines in the log read as what the fp8 switch costs.  Measured 3.7e-2 / 52.4 dB / 29.8 dB; floors 3x / 6 dB / 6 dB."""
    r = config1_reference()
    lat, mel, wav, _ = run_config1("bf16", attn_fp8=True)
    err = (lat - r["lat"]).abs().max().item()
    p, s = psnr(mel, r["mel"]), snr_db(wav, r["wav"])
    print("config 1 bf16 + fp8 P.V attention vs fp32 oracle: latents max abs err %.3e, mel PSNR %.1f dB, waveform SNR %.1f dB" % (err, p, s))
    assert err <= 1.1e-1 and p >= 46.4 and s >= 23.8


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_unet_forward_xl(dtype):
    """configs/diffusion_model_xl_config.json: FLAN-T5-XL text width (cross_attention_dim 2048), everything else as large."""
    cfg = O.UNET_CONFIG_XL
    sd = W.synth_state_dict(W.unet_param_shapes(cfg, "unet."), 1234)
    e = Engine(unet=cfg, dtype=dtype)
    e.load_synthetic(1234)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(2, 8, 256, 16, generator=g)
    enc = torch.randn(2, 64, 2048, generator=g)
    mask = torch.ones(2, 64, dtype=torch.bool)
    mask[0, 1:] = False
    with torch.no_grad():
        ref = O.unet_forward(sd, cfg, x, 500, enc, mask, prefix="unet.")
    out = e.unet_forward(x.cuda(), 500, enc.cuda(), mask.cuda()).cpu()
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print("XL UNet (d_text 2048) %s rel err %.3e" % (dtype, err))
    assert err <= (5e-6 if dtype == "fp32" else 4.3e-3)     # measured 1.6e-6 / 1.4e-3


def _philox(lib, B, C_, HW, step, seed, offset):
    out = torch.empty(B, C_, HW, device="cuda")
    rc = lib.tango_op_philox_normal(C.c_void_p(out.data_ptr()), B, C_, HW, step, seed, offset, None)
    assert rc == 0, lib.tango_last_error().decode()
    return out


def test_philox_noise_statistics(lib):
    """Box-Muller over Philox4x32-10 (elementwise.hip): N(0,1) moments of 1.05e6 draws, independence across steps / seeds /
    samples, and offset invariance of the counter layout."""
    x = _philox(lib, 4, 8, 32768, 3, 1234, 0).double().flatten()      # 1 048 576 draws
    n = x.numel()
    mean, var = x.mean().item(), x.var().item()
    skew = ((x - mean) ** 3).mean().item() / var ** 1.5
    kurt = ((x - mean) ** 4).mean().item() / var ** 2
    tail = (x.abs() > 3).double().mean().item()
    print("philox: n=%d mean %.4f var %.4f skew %.4f kurtosis %.4f P(|x|>3) %.5f max %.2f" % (n, mean, var, skew, kurt, tail, x.abs().max()))
    assert abs(mean) < 4e-3 and abs(var - 1) < 6e-3 and abs(skew) < 1e-2 and abs(kurt - 3) < 3e-2
    assert abs(tail - 0.0026998) < 4e-4 and 4.0 < x.abs().max().item() < 7.0
    a = _philox(lib, 2, 8, 4096, 3, 1234, 0).flatten()
    for other in (_philox(lib, 2, 8, 4096, 4, 1234, 0), _philox(lib, 2, 8, 4096, 3, 1235, 0), _philox(lib, 2, 8, 4096, 3, 1234, 2)):
        c = torch.corrcoef(torch.stack([a, other.flatten()]))[0, 1].item()
        assert abs(c) < 2e-2, c
    # neighbouring channels / positions of one draw are uncorrelated too
    y = _philox(lib, 1, 8, 32768, 0, 9, 0)[0]
    cc = torch.corrcoef(y)                                             # 8 x 8 over channels
    assert (cc - torch.eye(8, device=cc.device)).abs().max().item() < 2e-2
    # samples [2, 4) of a 4-sample draw == a 2-sample draw at offset 2 (what a second DP rank generates)
    full = _philox(lib, 4, 8, 4096, 5, 77, 0)
    part = _philox(lib, 2, 8, 4096, 5, 77, 2)
    assert torch.equal(full[2:], part)


def test_denoise_shard_invariance_on_device_noise():
    """Two shards (sample_offset 0 and 2) of a 4-prompt batch reproduce the unsharded run with the device Philox noise:
    results do not depend on how many GPUs the prompts are split over.  The NOISE is bit-identical
    (test_philox_noise_statistics); the latents agree to fp32 rounding only, because the GEMM tiling / split-K policy --
    hence the summation order -- depends on the UNet batch."""
    cfg = O.UNET_CONFIG_TINY
    e = Engine(unet=cfg, dtype="fp32")
    e.load_synthetic(1234)
    B, L, N = 4, 8, 3
    g = torch.Generator().manual_seed(5)
    cond = torch.randn(B, L, cfg["cross_attention_dim"], generator=g)
    unc = torch.randn(B, L, cfg["cross_attention_dim"], generator=g)
    mask_c = torch.ones(B, L, dtype=torch.bool)
    mask_u = torch.zeros(B, L, dtype=torch.bool)
    mask_u[:, 0] = True
    lat0 = torch.randn(B, 8, 256, 16, generator=g)
    sch = _sched()
    sch.set_timesteps(N)

    def run(lo, hi):
        lat = lat0[lo:hi].clone().cuda()
        enc = torch.cat([unc[lo:hi], cond[lo:hi]]).cuda()
        mask = torch.cat([mask_u[lo:hi], mask_c[lo:hi]]).cuda()
        e.denoise(lat, enc, mask, sch.timesteps.numpy(), sch.coef_table(), 3.0, noise=None, seed=4242, sample_offset=lo)
        torch.cuda.synchronize()
        return lat.cpu()

    full = run(0, 4)
    parts = torch.cat([run(0, 2), run(2, 4)])
    d = (full - parts).abs().max().item()
    print("shard invariance: max abs difference between the 4-prompt run and the 2+2 sharded run %.3e" % d)
    assert d <= 2e-5 * full.abs().max().item()
    assert not torch.equal(full[0], full[1])
    other = lat0.clone().cuda()
    e.denoise(other, torch.cat([unc, cond]).cuda(), torch.cat([mask_u, mask_c]).cuda(), sch.timesteps.numpy(), sch.coef_table(), 3.0,
              noise=None, seed=4243, sample_offset=0)
    assert (other.cpu() - full).abs().max().item() > 1e-2, "a different seed must give different step noise"


def test_weight_reload_invalidates_time_embedding_cache():
    """ensure_temb() caches the per-ResBlock time-embedding tables by timestep list; reloading weights on a
    live engine must drop them."""
    cfg = O.UNET_CONFIG_TINY
    e = Engine(unet=cfg, dtype="fp32")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 8, 256, 16, generator=g)
    enc = torch.randn(2, 6, cfg["cross_attention_dim"], generator=g)
    for seed in (1234, 4321):
        e.load_synthetic(seed)
        sd = W.synth_state_dict(W.unet_param_shapes(cfg, "unet."), seed)
        with torch.no_grad():
            ref = O.unet_forward(sd, cfg, x, 300, enc, None, prefix="unet.")
        out = e.unet_forward(x.cuda(), 300, enc.cuda(), None).cpu()     # same timestep both times
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        assert err <= 1e-3, (seed, err)
