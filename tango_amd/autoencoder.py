"""`AutoencoderKL` -- drop-in for audioldm/variational_autoencoder/autoencoder.py:9-135 of the reference:
`decode_first_stage`, `decode`, `decode_to_waveform`, `device()` (a METHOD there, autoencoder.py:108) and, with
`with_encoder=True`, `encode` / `encode_first_stage` / `get_first_stage_encoding` with the
`DiagonalGaussianDistribution` posterior of distributions.py:24-41.  The mel front-end (STFT) is not part of this class.
"""
import numpy as np
import torch

from .engine import HIFIGAN_CONFIG, Engine


class DiagonalGaussianDistribution:
    """distributions.py:24-41 of the reference: `parameters` = [mean | logvar] along dim 1, logvar clamped to [-30, 20];
    `sample()` draws from torch's global generator on the parameters' device, like the reference."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device)

    def mode(self):
        return self.mean


class AutoencoderKL:
    def __init__(self, ddconfig=None, lossconfig=None, image_key="fbank", embed_dim=None, time_shuffle=1, subband=1,
                 ckpt_path=None, reload_from_ckpt=None, ignore_keys=[], colorize_nlabels=None, monitor=None, base_learning_rate=1e-5,
                 scale_factor=1, *, hifigan_config=None, dtype: str = "fp16", device="cuda:0", with_encoder: bool = False, **_):
        assert subband == 1, "freq_merge_subband is the identity only for subband == 1 (autoencoder.py:126-135)"
        dd = dict(ddconfig)
        self.vae_cfg = dict(ch=dd["ch"], ch_mult=list(dd["ch_mult"]), num_res_blocks=dd["num_res_blocks"],
                            z_channels=dd["z_channels"], out_ch=dd["out_ch"], embed_dim=embed_dim or dd["z_channels"],
                            scale_factor=scale_factor, in_channels=dd.get("in_channels", 1))
        assert not dd.get("attn_resolutions"), "attn_resolutions must be empty (released Tango VAE)"
        self.scale_factor = scale_factor
        self.embed_dim = self.vae_cfg["embed_dim"]
        self._device = torch.device(device)
        self.with_encoder = bool(with_encoder)
        self.engine = Engine(vae=self.vae_cfg, hifigan=hifigan_config or HIFIGAN_CONFIG, dtype=dtype, device=device,
                             vae_encoder=self.with_encoder)

    def load_state_dict(self, sd, strict=True):
        missing = self.engine.load_state_dict(sd, strict=strict)
        self.engine.finalize()
        return missing

    def eval(self):
        return self

    def to(self, device):
        return self

    def device(self):
        return self._device

    @torch.no_grad()
    def encode(self, x):
        """autoencoder.py:52-58: Encoder + quant_conv on the engine -> posterior (subband == 1: no frequency split)."""
        return DiagonalGaussianDistribution(self.engine.vae_encode(x))

    def encode_first_stage(self, x):
        """autoencoder.py:112-113"""
        return self.encode(x)

    def get_first_stage_encoding(self, encoder_posterior):
        """autoencoder.py:126-135: scale_factor * posterior.sample() (or the tensor itself)"""
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            z = encoder_posterior.sample()
        elif isinstance(encoder_posterior, torch.Tensor):
            z = encoder_posterior
        else:
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return self.scale_factor * z

    @torch.no_grad()
    def decode(self, z):
        """autoencoder.py:60-64 (post_quant_conv + Decoder); note: no 1/scale_factor here."""
        return self.engine.vae_decode(z * self.scale_factor)

    @torch.no_grad()
    def decode_first_stage(self, z):
        """autoencoder.py:116-124: z / scale_factor -> decode -> mel [B,1,1024,64]."""
        return self.engine.vae_decode(z)

    @torch.no_grad()
    def decode_to_waveform(self, dec) -> np.ndarray:
        """autoencoder.py:66-69 -> vocoder_infer (hifigan/utilities.py:76-86): np.int16 [B, 163872].
        The int16 cast (C truncation of wav*32768) happens on the device; only int16 crosses PCIe."""
        wav = self.engine.vocode(dec)
        return wav.cpu().numpy()
