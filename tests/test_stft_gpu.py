"""Wave -> log-mel front-end on the engine (`tango_engine_mel_spectrogram`) vs the CPU oracle and vs the
committed outputs of the imported reference, through the drop-in `tango_amd.stft.TacotronSTFT`.

Tolerances (fp32 both sides; the engine's f32-MFMA DFT sums in a different order than the CPU conv1d): magnitudes are compared
in the LINEAR domain relative to the loudest bin of the clip -- a spectrogram spans 6+ decades, and the logarithm turns fp32
rounding of a bin that is 1e-6 of the peak into an O(1) difference of no meaning --, plus the log-mel directly wherever the
reference value is above the noise floor of fp32 (linear value >= 1e-4 of the clip's peak)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import stft_oracle as S  # noqa: E402  (checker only)
from oracle.make_golden import stft_wave  # noqa: E402
from tango_amd.stft import TacotronSTFT, wav_to_fbank  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stft_ref.npz")


def _compare(got, ref, what):
    """log-domain tensors [B, C, T]: linear-domain error relative to the clip's peak, and log error above the fp32 noise floor"""
    g, r = got.double().exp(), ref.double().exp()
    peak = r.amax(dim=(1, 2), keepdim=True)
    lin = ((g - r).abs() / peak).max().item()
    loud = r >= 1e-4 * peak
    logerr = (got.double() - ref.double()).abs()[loud].max().item()
    print("%s: linear err %.3e of the peak, log err above the floor %.3e" % (what, lin, logerr))
    assert lin <= 2e-6 and logerr <= 5e-3, (what, lin, logerr)


def test_mel_spectrogram_matches_oracle_and_reference_fixture():
    stft = TacotronSTFT(**S.AUDIOLDM_STFT_CONFIG)
    y = stft_wave()
    mel, logmag, energy = stft.mel_spectrogram(y.cuda())
    assert mel.shape == (2, 64, 126) and logmag.shape == (2, 513, 126) and energy.shape == (2, 126)
    mb = torch.from_numpy(S.slaney_mel_filterbank(16000, 1024, 64, 0, 8000))
    fb = S.stft_forward_basis(1024, 1024)
    m0, l0, e0 = S.mel_spectrogram(y, mb, fb)
    _compare(mel.cpu(), m0, "log-mel vs oracle")
    _compare(logmag.cpu(), l0, "log-magnitudes vs oracle")
    assert ((energy.cpu() - e0).abs() / e0.max()).max().item() <= 2e-6
    gold = np.load(GOLD)                                        # outputs of the IMPORTED reference (oracle/make_golden.py stft)
    _compare(mel.cpu(), torch.from_numpy(gold["mel"]), "log-mel vs reference fixture")
    assert abs(float(mel.min()) - np.log(1e-5)) < 1e-5          # the quiet stretch sits on the clamp, exactly


@pytest.mark.parametrize("B,N", [(1, 163840), (3, 16000 * 2 + 123), (5, 4000)])
def test_mel_spectrogram_shapes_and_batches(B, N):
    """full-length clip (1024 * 160 samples -> 1025 frames -> cut to 1024 by _pad_spec), ragged length, several short items:
    batch items must not leak into each other (the frame GEMM reads overlapping rows of one padded buffer per item)"""
    stft = TacotronSTFT(**S.AUDIOLDM_STFT_CONFIG)
    y = stft_wave(B=B, N=N, seed=B + N)
    mel, logmag, energy = stft.mel_spectrogram(y.cuda())
    T = 1 + N // 160
    assert mel.shape == (B, 64, T) and logmag.shape == (B, 513, T) and energy.shape == (B, T)
    m0, l0, e0 = S.mel_spectrogram(y, stft.mel_basis, stft.stft_fn.forward_basis)
    _compare(mel.cpu(), m0, "B=%d N=%d log-mel" % (B, N))
    _compare(logmag.cpu(), l0, "B=%d N=%d log-magnitudes" % (B, N))
    one = stft.mel_spectrogram(y[B - 1:].cuda())[0]
    assert torch.equal(one[0], mel[B - 1]), "an item's mel must not depend on its batch neighbours"
    fbank, lm, wav = wav_to_fbank(y.cuda(), target_length=1024, fn_STFT=stft)
    assert fbank.shape == (B, 1024, 64) and lm.shape == (B, 1024, 512) and wav.shape == (B, N)
    k = min(T, 1024)
    assert torch.equal(fbank[:, :k], mel.transpose(1, 2)[:, :k]) and (T >= 1024 or float(fbank[:, T:].abs().max()) == 0.0)


def test_state_dict_and_errors():
    stft = TacotronSTFT(**S.AUDIOLDM_STFT_CONFIG)
    y = stft_wave(B=1, N=8000, seed=3)
    base = stft.mel_spectrogram(y.cuda())[0]
    # pytorch_model_stft.bin (tango.py:23-27): buffers replace what the constructor computed
    sd = {"mel_basis": stft.mel_basis * 2.0, "stft_fn.forward_basis": stft.stft_fn.forward_basis.clone(),
          "stft_fn.inverse_basis": torch.zeros(1026, 1, 1024)}
    stft.load_state_dict(sd)
    twice = stft.mel_spectrogram(y.cuda())[0]
    loud = base > np.log(1e-5) + 1.0
    assert (twice - base)[loud].sub(np.log(2.0)).abs().max().item() < 1e-4, "the loaded mel_basis must be the one in use"
    with pytest.raises(RuntimeError, match="Unexpected"):
        stft.load_state_dict({**sd, "bogus": torch.zeros(1)})
    with pytest.raises(RuntimeError, match="size mismatch"):
        stft.load_state_dict({"mel_basis": torch.zeros(80, 513), "stft_fn.forward_basis": sd["stft_fn.forward_basis"]})
    with pytest.raises(AssertionError):
        stft.mel_spectrogram(torch.full((1, 4000), 1.5).cuda())      # stft.py:176-177
    with pytest.raises(RuntimeError, match="reflect"):
        stft.mel_spectrogram(torch.zeros(1, 400).cuda())             # F.pad(mode="reflect") needs N > n_fft / 2
    other = TacotronSTFT(filter_length=512, hop_length=128, win_length=400, n_mel_channels=40, sampling_rate=16000, mel_fmin=20, mel_fmax=7600)
    m, lm, en = other.mel_spectrogram(y.cuda())
    m0, l0, e0 = S.mel_spectrogram(y, other.mel_basis, other.stft_fn.forward_basis, 512, 128)
    assert m.shape == m0.shape == (1, 40, 63)
    _compare(m.cpu(), m0, "n_fft 512 / hop 128 / win 400 / 40 mels")


def test_plan_cache_is_bounded_over_arbitrary_clip_lengths():
    """a data-preparation loop over clips of many different lengths: the engine's plan cache stays inside its byte budget (least
    recently used workspaces are freed inside the engine, round 4) and results do not depend on where in that cycle a clip falls"""
    stft = TacotronSTFT(**S.AUDIOLDM_STFT_CONFIG)
    stft.mel_spectrogram(stft_wave(B=1, N=4000, seed=1).cuda())
    one, n = stft.engine.plan_stats()
    assert n == 1 and one > 0
    budget = int(2.5 * one)                      # room for two clips of about this length
    stft.engine.set_plan_budget(budget)
    for i, N in enumerate([4000, 4160, 5000, 4000, 7777, 4160]):
        y = stft_wave(B=1, N=N, seed=N)
        mel, _, _ = stft.mel_spectrogram(y.cuda())
        m0, _, _ = S.mel_spectrogram(y, stft.mel_basis, stft.stft_fn.forward_basis)
        _compare(mel.cpu(), m0, "clip %d (N=%d)" % (i, N))
        used, n = stft.engine.plan_stats()
        assert n >= 1 and (used <= budget or n == 1), (used, budget, n)
    assert n <= 2
