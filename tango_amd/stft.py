"""`TacotronSTFT` -- drop-in for audioldm/audio/stft.py:136-186 of the reference (the wave -> log-mel front-end, SURVEY.md 8f
rank 4): same constructor (`filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax` -- the
keys of the snapshot's stft_config.json, tango.py:17-19, and of models.py:40-47), `mel_spectrogram(y) -> (log-mel, log-magnitudes,
energy)`, `load_state_dict` of pytorch_model_stft.bin (buffers `mel_basis`, `stft_fn.forward_basis`, `stft_fn.inverse_basis`),
plus `get_mel_from_wav` / `wav_to_fbank` / `_pad_spec` of tools/torch_tools.py:31-78 on tensors (reading and resampling wav
files -- torchaudio in the reference -- is the caller's business).

The transform runs on the HIP engine (`tango_engine_mel_spectrogram`, fp32).  What is host code here is what the reference
also computes once on the host at construction: the windowed DFT basis (stft.py:25-46) and the mel filterbank, which the
reference takes from `librosa.filters.mel` (stft.py:151-153; librosa 0.9.2: Slaney mel scale, `norm="slaney"`) -- restated in
`slaney_mel_filterbank`.  With a released checkpoint both are overwritten by the buffers of pytorch_model_stft.bin.
Not built: `STFT.inverse` / Griffin-Lim (stft.py:87-128, audio_processing.py:67-81) -- nothing on Tango's paths calls them
(waveforms come from HiFi-GAN)."""
from typing import Optional

import numpy as np
import torch

from .engine import Engine

#: audioldm/utils.py:104-118 default_audioldm_config()["preprocessing"] as models.py:40-47 passes it
AUDIOLDM_STFT_CONFIG = dict(filter_length=1024, hop_length=160, win_length=1024, n_mel_channels=64, sampling_rate=16000,
                            mel_fmin=0, mel_fmax=8000)


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) (0.9.2 defaults htk=False, norm="slaney") -> float32 [n_mels, n_fft//2+1]:
    mel scale linear below 1 kHz (200/3 Hz per mel) and logarithmic above (27 mels per factor 6.4), triangular filters between
    consecutive mel-spaced centre frequencies evaluated at the FFT bin frequencies, each scaled by 2 / (f[i+2] - f[i])."""
    fmax = float(sr) / 2 if fmax is None else float(fmax)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def to_mel(f):
        f = np.asarray(f, np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def to_hz(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    n_freq = 1 + n_fft // 2
    bins = np.linspace(0.0, float(sr) / 2, n_freq)
    edges = to_hz(np.linspace(to_mel(float(fmin)), to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - bins[None, :]
    fb = np.maximum(0.0, np.minimum(-ramps[:-2] / width[:-1, None], ramps[2:] / width[1:, None]))
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return fb.astype(np.float32)


def stft_forward_basis(filter_length, win_length) -> torch.Tensor:
    """stft.py:25-46: rows [Re; Im] of the first n_fft/2+1 DFT vectors times the periodic Hann window (zero-centred to n_fft)"""
    assert filter_length >= win_length
    fourier = np.fft.fft(np.eye(filter_length))
    cutoff = filter_length // 2 + 1
    basis = np.vstack([np.real(fourier[:cutoff]), np.imag(fourier[:cutoff])])
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)      # scipy get_window("hann", n, fftbins=True)
    lpad = (filter_length - win_length) // 2
    win = np.pad(win, (lpad, filter_length - win_length - lpad))                     # librosa.util.pad_center
    return (torch.FloatTensor(basis[:, None, :]) * torch.from_numpy(win).float()).float()


class _StftFn:
    """the attribute callers reach through `stft.stft_fn` (filter / hop / window lengths and the basis buffer)"""

    def __init__(self, filter_length, hop_length, win_length, forward_basis):
        self.filter_length, self.hop_length, self.win_length = filter_length, hop_length, win_length
        self.forward_basis = forward_basis


class TacotronSTFT:
    def __init__(self, filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax, *, device="cuda:0"):
        self.n_mel_channels = n_mel_channels
        self.sampling_rate = sampling_rate
        self.config = dict(filter_length=filter_length, hop_length=hop_length, win_length=win_length, n_mel_channels=n_mel_channels,
                           sampling_rate=sampling_rate, mel_fmin=mel_fmin, mel_fmax=mel_fmax)
        self.device = torch.device(device)
        self.mel_basis = torch.from_numpy(slaney_mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax))
        self.stft_fn = _StftFn(filter_length, hop_length, win_length, stft_forward_basis(filter_length, win_length))
        self.engine = Engine(stft=self.config, dtype="fp32", device=device)      # raises without the HIP library / a GPU
        # one small plan per (batch, n_samples): a data-preparation loop over clips of arbitrary lengths must not grow this cache to the
        # engine-wide 64-GiB default -- 1 GiB holds dozens of 10-s batches, older shapes are evicted first
        self.engine.set_plan_budget(1 << 30)
        self._push()

    def _push(self):
        self.engine.load_state_dict({"mel_basis": self.mel_basis, "stft_fn.forward_basis": self.stft_fn.forward_basis})
        self.engine.finalize()

    # ---- nn.Module surface the callers use (tango.py:19-33, predict.py:71-89, models.py:51) ----
    def state_dict(self):
        return {"mel_basis": self.mel_basis, "stft_fn.forward_basis": self.stft_fn.forward_basis}

    def load_state_dict(self, sd, strict=True):
        """pytorch_model_stft.bin: `mel_basis`, `stft_fn.forward_basis` (and `stft_fn.inverse_basis`, accepted and unused)"""
        known = {"mel_basis", "stft_fn.forward_basis", "stft_fn.inverse_basis"}
        unexpected = sorted(k for k in sd if k not in known)
        missing = sorted(k for k in ("mel_basis", "stft_fn.forward_basis") if k not in sd)
        if strict and (unexpected or missing):
            raise RuntimeError("Error(s) in loading state_dict for TacotronSTFT: Missing key(s) %s, Unexpected key(s) %s" % (missing, unexpected))
        for k, cur in (("mel_basis", self.mel_basis), ("stft_fn.forward_basis", self.stft_fn.forward_basis)):
            if k in sd and tuple(sd[k].shape) != tuple(cur.shape):
                raise RuntimeError("size mismatch for %s: copying a param with shape %s, the shape in current model is %s"
                                   % (k, tuple(sd[k].shape), tuple(cur.shape)))
        if "mel_basis" in sd:
            self.mel_basis = sd["mel_basis"].detach().float().cpu().clone()
        if "stft_fn.forward_basis" in sd:
            self.stft_fn.forward_basis = sd["stft_fn.forward_basis"].detach().float().cpu().clone()
        self._push()
        return missing

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def cuda(self, *_a, **_k):
        return self

    # ---- the transform ------------------------------------------------------------------------
    @torch.no_grad()
    def mel_spectrogram(self, y, normalize_fun=torch.log):
        """stft.py:164-186: y [B, T] in [-1, 1] -> (mel_output [B, n_mel, T'], log_magnitudes [B, n_fft/2+1, T'], energy [B, T'])"""
        if normalize_fun is not torch.log:
            raise NotImplementedError("the engine implements the reference's default normalize_fun=torch.log")
        assert torch.min(y.data) >= -1, torch.min(y.data)           # the reference's own input checks (stft.py:176-177)
        assert torch.max(y.data) <= 1, torch.max(y.data)
        # (one plan -- workspace slab -- per (batch, n_samples); the engine's LRU byte budget frees old ones: Engine.set_plan_budget)
        return self.engine.mel_spectrogram(y)


# ---- tools/torch_tools.py:31-78 on tensors ---------------------------------------------------------
def _pad_spec(fbank, target_length=1024):
    batch, n_frames, channels = fbank.shape
    p = target_length - n_frames
    if p > 0:
        fbank = torch.cat([fbank, torch.zeros(batch, p, channels, device=fbank.device, dtype=fbank.dtype)], 1)
    elif p < 0:
        fbank = fbank[:, :target_length, :]
    if channels % 2 != 0:
        fbank = fbank[:, :, :-1]
    return fbank


def get_mel_from_wav(audio, _stft):
    audio = torch.nan_to_num(torch.clip(audio, -1, 1))
    return _stft.mel_spectrogram(audio)


def wav_to_fbank(waveform, target_length=1024, fn_STFT: Optional[TacotronSTFT] = None):
    """tools/torch_tools.py:66-78 with the waveforms already loaded: waveform [B, target_length * hop] -> (fbank [B, target_length,
    n_mel], log_magnitudes_stft [B, target_length, n_fft/2], waveform)"""
    assert fn_STFT is not None
    fbank, log_magnitudes_stft, _ = get_mel_from_wav(waveform, fn_STFT)
    fbank = fbank.transpose(1, 2)
    log_magnitudes_stft = log_magnitudes_stft.transpose(1, 2)
    return _pad_spec(fbank, target_length), _pad_spec(log_magnitudes_stft, target_length), waveform
