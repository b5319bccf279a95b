"""CPU oracle for the wave -> log-mel front-end (VERDICT r2 missing #1).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and never by tango_amd/, which fails loudly without the HIP library).

Restates, in numpy / torch-CPU fp32-fp64, what the reference computes in

  audioldm/audio/stft.py:15-50    STFT.__init__   forward_basis = [Re; Im] of fft(eye(n_fft))[:n_fft/2+1] * hann(win_length) (periodic,
                                                  scipy get_window(..., fftbins=True), centre-padded to n_fft by librosa.util.pad_center)
  audioldm/audio/stft.py:52-85    STFT.transform  reflect-pad n_fft/2 each side, conv1d(stride = hop) with forward_basis,
                                                  magnitude = sqrt(re^2 + im^2)  (phase is computed there too; nothing on this path reads it)
  audioldm/audio/stft.py:136-186  TacotronSTFT    mel = mel_basis @ magnitude;  log(clamp(., 1e-5)) (audio_processing.py:84-92: C = 1,
                                                  clip_val = 1e-5);  energy = ||magnitude||_2 over frequency;  log-magnitudes likewise
  tools/torch_tools.py:57-78      get_mel_from_wav / wav_to_fbank / _pad_spec   clip to [-1, 1], transpose to [B, T, n_mel], pad / cut to
                                                  target_length frames

Third-party arithmetic that is NOT under /root/reference: `librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` (requirements.txt:
librosa==0.9.2; call site stft.py:151-153 with positional arguments) builds the mel filterbank.  librosa is not installed here
and cannot be fetched, so `slaney_mel_filterbank` restates its published algorithm (librosa 0.9.2 filters.py `mel`: Slaney-style
mel scale -- linear below 1 kHz, log above, `htk=False` --, triangular filters on the FFT bin centres, `norm="slaney"` area
normalisation 2 / (f[i+2] - f[i])).  PINNING: (a) against `transformers.audio_utils.mel_filter_bank(norm="slaney",
mel_scale="slaney")`, an independent implementation that exists precisely to reproduce librosa's filters for Whisper
(tests/test_stft_oracle.py, <= 1e-6); (b) everything downstream of the filterbank against the IMPORTED reference
`audioldm.audio.stft.TacotronSTFT` with `librosa` stubbed by these restatements (tests/test_reference_diff.py, marker
`reference`) and against committed outputs of that run (tests/golden/stft_golden.pt, oracle/make_golden.py).  With a released
checkpoint the question does not arise: tango.py:19-27 loads `mel_basis` and `forward_basis` as BUFFERS from
pytorch_model_stft.bin, so the filterbank is data there.  Status: the filterbank itself is "parity unpinned" by the reference
(no librosa to run), pinned to a third-party restatement; STFT / magnitude / log / energy are pinned to the reference."""
import numpy as np
import torch
import torch.nn.functional as F

#: audioldm/utils.py:104-118 default_audioldm_config()["preprocessing"]  (models.py:40-47 passes exactly these)
AUDIOLDM_STFT_CONFIG = dict(filter_length=1024, hop_length=160, win_length=1024, n_mel_channels=64, sampling_rate=16000,
                            mel_fmin=0, mel_fmax=8000)


def _hz_to_mel(f):
    """librosa.core.convert.hz_to_mel, htk=False (Slaney's Auditory Toolbox scale)"""
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    out = np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)
    return out


def _mel_to_hz(m):
    """librosa.core.convert.mel_to_hz, htk=False"""
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with the 0.9.2 defaults htk=False, norm='slaney', dtype=float32:
    [n_mels, 1 + n_fft // 2]"""
    if fmax is None:
        fmax = float(sr) / 2
    n_freq = 1 + n_fft // 2
    fftfreqs = np.linspace(0, float(sr) / 2, n_freq, endpoint=True)                     # librosa.fft_frequencies
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))     # librosa.mel_frequencies
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_freq), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])                                # norm == "slaney"
    weights *= enorm[:, np.newaxis]
    return weights.astype(np.float32)


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True)"""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def pad_center(w, size):
    """librosa.util.pad_center on a 1-D array"""
    lpad = (size - len(w)) // 2
    return np.pad(w, (lpad, size - len(w) - lpad))


def stft_forward_basis(filter_length, win_length):
    """stft.py:25-46: float32 [2 * (n_fft/2 + 1), 1, n_fft]"""
    fourier = np.fft.fft(np.eye(filter_length))
    cutoff = filter_length // 2 + 1
    basis = np.vstack([np.real(fourier[:cutoff]), np.imag(fourier[:cutoff])])
    fb = torch.FloatTensor(basis[:, None, :])
    win = torch.from_numpy(pad_center(hann_periodic(win_length), filter_length)).float()
    return (fb * win).float()


def stft_magnitude(y, forward_basis, filter_length, hop_length):
    """STFT.transform (stft.py:52-85) -> magnitude [B, n_fft/2 + 1, T], T = 1 + N // hop"""
    B, N = y.shape
    x = F.pad(y.view(B, 1, 1, N), (filter_length // 2, filter_length // 2, 0, 0), mode="reflect").squeeze(1)
    ft = F.conv1d(x, forward_basis, stride=hop_length, padding=0)
    cutoff = filter_length // 2 + 1
    re, im = ft[:, :cutoff], ft[:, cutoff:]
    return torch.sqrt(re ** 2 + im ** 2)


def mel_spectrogram(y, mel_basis, forward_basis, filter_length=1024, hop_length=160, clip_val=1e-5):
    """TacotronSTFT.mel_spectrogram (stft.py:164-186) -> (log-mel [B, n_mel, T], log-magnitudes [B, n_fft/2+1, T], energy [B, T])"""
    assert float(y.min()) >= -1 and float(y.max()) <= 1
    mag = stft_magnitude(y.float(), forward_basis, filter_length, hop_length)
    mel = torch.matmul(mel_basis, mag)
    return torch.log(torch.clamp(mel, min=clip_val)), torch.log(torch.clamp(mag, min=clip_val)), torch.norm(mag, dim=1)


def pad_spec(fbank, target_length=1024):
    """tools/torch_tools.py:31-42 (_pad_spec) on [B, T, C]"""
    B, n, C = fbank.shape
    p = target_length - n
    if p > 0:
        fbank = torch.cat([fbank, torch.zeros(B, p, C)], 1)
    elif p < 0:
        fbank = fbank[:, :target_length, :]
    if C % 2 != 0:
        fbank = fbank[:, :, :-1]
    return fbank


def wav_to_fbank(waveform, mel_basis, forward_basis, target_length=1024, filter_length=1024, hop_length=160):
    """tools/torch_tools.py:57-78 minus the file reading: waveform [B, N] in [-1, 1] -> (fbank [B, target_length, n_mel],
    log-magnitudes [B, target_length, n_fft/2 (the odd 513th bin is dropped by _pad_spec)], waveform)"""
    audio = torch.nan_to_num(torch.clip(waveform, -1, 1))
    mel, logmag, _ = mel_spectrogram(audio, mel_basis, forward_basis, filter_length, hop_length)
    return pad_spec(mel.transpose(1, 2), target_length), pad_spec(logmag.transpose(1, 2), target_length), waveform
