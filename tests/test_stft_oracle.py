"""CPU checks of the wave -> log-mel front-end's oracle (oracle/stft_oracle.py) and of the product's host-side basis code.

Pinning chain (see the oracle's header): Slaney mel filterbank == transformers.audio_utils.mel_filter_bank (an independent
restatement of librosa's, which is not installable here); DFT basis == numpy.fft; the whole TacotronSTFT.mel_spectrogram ==
outputs of the IMPORTED reference committed in tests/golden/stft_ref.npz (and re-run live under `-m reference`)."""
import os

import numpy as np
import pytest
import torch

from oracle import stft_oracle as S
from oracle.make_golden import stft_wave

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stft_ref.npz")


def _checksum(t):
    a = t.detach().double()
    return np.asarray([float(a.sum()), float(a.abs().sum()), float((a * a).sum())])


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", [(16000, 1024, 64, 0, 8000), (22050, 2048, 80, 55, 7600), (16000, 512, 40, 20, None)])
def test_slaney_filterbank_matches_independent_restatement(sr, n_fft, n_mels, fmin, fmax):
    from transformers.audio_utils import mel_filter_bank
    ours = S.slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    theirs = mel_filter_bank(1 + n_fft // 2, n_mels, float(fmin), float(fmax if fmax is not None else sr / 2), sr, norm="slaney",
                             mel_scale="slaney").T
    assert ours.shape == (n_mels, 1 + n_fft // 2) and ours.dtype == np.float32
    assert np.abs(ours - theirs).max() <= 1e-6 * np.abs(theirs).max()
    # structure of librosa's filters: non-negative, every filter non-empty, Slaney area normalisation, band-limited
    freqs = np.linspace(0, sr / 2, 1 + n_fft // 2)
    assert (ours >= 0).all() and (ours.max(1) > 0).all()
    hi = fmax if fmax is not None else sr / 2
    assert ours[:, freqs < fmin - 1e-9].sum() == 0 and ours[:, freqs > hi + 1e-9].sum() == 0
    # the product's own host-side copy (tango_amd/stft.py) is an independent function: same numbers
    from tango_amd.stft import slaney_mel_filterbank
    assert np.array_equal(slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax), ours)


def test_forward_basis_is_the_windowed_dft():
    from scipy.signal import get_window
    n = 1024
    fb = S.stft_forward_basis(n, n)[:, 0, :].double().numpy()
    win = get_window("hann", n, fftbins=True)
    g = np.random.default_rng(0)
    x = g.standard_normal(n)
    spec = np.fft.rfft(x * win)
    re, im = fb[:513] @ x, fb[513:] @ x
    # np.fft.fft(eye)[k] = exp(-2 pi i k n / N): Re matches, Im is the NEGATIVE sine transform, exactly like rfft
    assert np.abs(re - spec.real).max() < 1e-4 and np.abs(im - spec.imag).max() < 1e-4
    from tango_amd.stft import stft_forward_basis
    assert torch.equal(stft_forward_basis(n, n), S.stft_forward_basis(n, n))
    assert torch.equal(stft_forward_basis(1024, 800), S.stft_forward_basis(1024, 800))      # win_length < filter_length: centred window


def test_mel_spectrogram_matches_reference_fixture():
    """oracle == the imported reference's outputs (committed by oracle/make_golden.py stft)"""
    gold = np.load(GOLD)
    y = stft_wave()
    mb = torch.from_numpy(S.slaney_mel_filterbank(16000, 1024, 64, 0, 8000))
    fb = S.stft_forward_basis(1024, 1024)
    assert np.allclose(_checksum(mb), gold["mel_basis_checksum"], rtol=1e-6) and np.allclose(_checksum(fb), gold["basis_checksum"], rtol=1e-6)
    mel, logmag, energy = S.mel_spectrogram(y, mb, fb)
    assert mel.shape == (2, 64, 126) and logmag.shape == (2, 513, 126) and energy.shape == (2, 126)     # T = 1 + 20000 // 160
    assert np.abs(mel.numpy() - gold["mel"]).max() <= 1e-4
    assert np.abs(logmag[:, ::9, ::5].numpy() - gold["logmag_slice"]).max() <= 1e-3
    assert np.allclose(_checksum(logmag), gold["logmag_checksum"], rtol=1e-5)
    assert np.abs(energy.numpy() - gold["energy"]).max() <= 1e-4 * gold["energy"].max()
    assert float(mel.min()) == pytest.approx(np.log(1e-5), abs=1e-6), "the quiet stretch must hit the 1e-5 clamp"
    # tools/torch_tools.py:31-78: transpose, pad / cut to target_length, drop the odd 513th frequency bin
    fbank, lm, wav = S.wav_to_fbank(y, mb, fb, target_length=128)
    assert fbank.shape == (2, 128, 64) and lm.shape == (2, 128, 512) and float(fbank[:, 126:].abs().max()) == 0.0
    assert S.wav_to_fbank(y, mb, fb, target_length=100)[0].shape == (2, 100, 64)
    with pytest.raises(AssertionError):
        S.mel_spectrogram(y * 3, mb, fb)            # stft.py:176-177: input must be in [-1, 1]


@pytest.mark.reference
def test_mel_spectrogram_matches_imported_reference_live():
    from oracle import ref_import as R
    if not R.available():
        pytest.skip("reference tree not present")
    ref = R.tacotron_stft_cls()(**S.AUDIOLDM_STFT_CONFIG).eval()
    y = stft_wave(B=3, N=16000 * 2 + 123, seed=5)
    m0, l0, e0 = ref.mel_spectrogram(y)
    m1, l1, e1 = S.mel_spectrogram(y, ref.mel_basis, ref.stft_fn.forward_basis)
    assert torch.equal(S.stft_forward_basis(1024, 1024), ref.stft_fn.forward_basis)
    assert (m0 - m1).abs().max().item() <= 1e-5 and (l0 - l1).abs().max().item() <= 1e-5 and (e0 - e1).abs().max().item() <= 1e-4
