"""Long-horizon precision ladder: BASELINE config 2's length -- 100 DDPM steps, B = 1, guidance 3, full-size
866M UNet -- fp16 and fp32 engines against the fp32 CPU oracle, plus fp16 vs the fp32 engine at config 3's 200 steps.

The 100-step oracle costs minutes of host time, so the test only runs with TANGO_LONG_TESTS=1; the recorded outputs are
profiles/r3_long_horizon_ladder.log, (final round-4 tree) profiles/r4_c20_long_horizon_ladder.log and (final round-6 tree) profiles/r6_final7_long_horizon_ladder.log.  Floors are the 10-step floors of test_parity_full_gpu.py: drift must stay bounded.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tango_oracle as O  # noqa: E402  (checker only)
from tango_amd import weights as W  # noqa: E402
from tango_amd.engine import Engine  # noqa: E402
from tango_amd.scheduler import SD21_SCHEDULER_CONFIG, DDPMScheduler  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("TANGO_LONG_TESTS") != "1", reason="minutes of host time: set TANGO_LONG_TESTS=1")]

_KEYS = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample", "variance_type")


def _inputs(N, seed=202):
    B, L = 1, 64
    g = torch.Generator().manual_seed(seed)
    cond = torch.randn(B, L, 1024, generator=g)
    unc = torch.randn(B, L, 1024, generator=g)
    enc = torch.cat([unc, cond])
    mask = torch.ones(2 * B, L, dtype=torch.bool)
    mask[:B, 1:] = False
    lat0 = torch.randn(B, 8, 256, 16, generator=g)
    noises = torch.randn(N, B, 8, 256, 16, generator=g)
    return enc, mask, lat0, noises


def _engine_run(dtype, N, enc, mask, lat0, noises):
    e = Engine(unet=O.UNET_CONFIG_LARGE, dtype=dtype)
    e.load_synthetic(1234)
    sch = DDPMScheduler.from_config({k: SD21_SCHEDULER_CONFIG[k] for k in _KEYS})
    sch.set_timesteps(N)
    lat = lat0.clone().cuda()
    e.denoise(lat, enc.cuda(), mask.cuda(), sch.timesteps.numpy(), sch.coef_table(), 3.0, noise=noises.cuda())
    torch.cuda.synchronize()
    del e
    ev = Engine(vae=O.VAE_CONFIG, hifigan=O.HIFIGAN_CONFIG, dtype=dtype)
    ev.load_synthetic(1234)
    mel = ev.vae_decode(lat)
    wav = ev.vocode(mel).cpu().numpy()
    return lat.cpu(), mel.cpu(), wav


def _psnr(x, ref):
    mse = ((x.double() - ref.double()) ** 2).mean().item()
    peak = (ref.max() - ref.min()).item()
    return 10 * np.log10(peak * peak / (mse + 1e-30))


def _snr(x, ref):
    x, ref = x.astype(np.float64), ref.astype(np.float64)
    return 10 * np.log10((ref ** 2).mean() / (((x - ref) ** 2).mean() + 1e-30))


def test_100_step_ladder_against_the_oracle():
    N = 100
    enc, mask, lat0, noises = _inputs(N)
    sd = W.synth_state_dict(W.unet_param_shapes(O.UNET_CONFIG_LARGE, "unet."), 1234)
    shapes = W.vae_decoder_param_shapes(O.VAE_CONFIG)
    shapes.update(W.hifigan_param_shapes(O.HIFIGAN_CONFIG))
    vsd = W.synth_state_dict(shapes, 1234)
    torch.set_num_threads(min(16, os.cpu_count() or 16))         # bench.py's sweep: 16 host threads are the oracle's optimum on the GPU box
    with torch.no_grad():
        ref = O.denoise_loop(sd, O.UNET_CONFIG_LARGE, O.DDPMOracle(**O.SD21_SCHEDULER), enc, mask, lat0.clone(), N, 3.0, noises=list(noises),
                             prefix="unet.")
        rmel = O.vae_decode_first_stage(vsd, O.VAE_CONFIG, ref)
        rwav = O.decode_to_waveform(vsd, O.HIFIGAN_CONFIG, rmel)
    out = {}
    for dtype in ("fp32", "fp16"):
        lat, mel, wav = _engine_run(dtype, N, enc, mask, lat0, noises)
        out[dtype] = (lat, mel, wav)
        err = (lat - ref).abs().max().item()
        print("100 DDPM steps (config 2 length, 866M UNet, B=1, g=3) %s engine vs fp32 oracle: latents max abs err %.3e (|ref| max %.2f), "
              "mel PSNR %.1f dB, waveform SNR %.1f dB" % (dtype, err, ref.abs().max(), _psnr(mel, rmel), _snr(wav, rwav)))
    lat, mel, wav = out["fp32"]
    assert (lat - ref).abs().max().item() <= 1e-2 and _snr(wav, rwav) >= 40.0
    lat, mel, wav = out["fp16"]
    assert (lat - ref).abs().max().item() <= 4e-2 and _psnr(mel, rmel) >= 56.0 and _snr(wav, rwav) >= 33.0


def test_200_step_fp16_against_the_fp32_engine():
    N = 200
    enc, mask, lat0, noises = _inputs(N, seed=303)
    l32, m32, w32 = _engine_run("fp32", N, enc, mask, lat0, noises)
    l16, m16, w16 = _engine_run("fp16", N, enc, mask, lat0, noises)
    err = (l16 - l32).abs().max().item()
    print("200 DDPM steps (config 3 length, B=1) fp16 engine vs fp32 engine: latents max abs err %.3e (|ref| max %.2f), mel PSNR %.1f dB, "
          "waveform SNR %.1f dB" % (err, l32.abs().max(), _psnr(m16, m32), _snr(w16, w32)))
    assert err <= 4e-2 and _psnr(m16, m32) >= 56.0 and _snr(w16, w32) >= 33.0
