"""`AutoencoderKL` -- drop-in for the decode-side surface of
audioldm/variational_autoencoder/autoencoder.py:9-135 of the reference: `decode_first_stage`,
`decode`, `decode_to_waveform`, `device()` (a METHOD there, autoencoder.py:108).  The encoder half
(`encode`, `encode_first_stage`, posterior) is training-only and out of scope.
"""
import numpy as np
import torch

from .engine import HIFIGAN_CONFIG, Engine


class AutoencoderKL:
    def __init__(self, ddconfig=None, lossconfig=None, image_key="fbank", embed_dim=None, time_shuffle=1, subband=1,
                 ckpt_path=None, reload_from_ckpt=None, ignore_keys=[], colorize_nlabels=None, monitor=None, base_learning_rate=1e-5,
                 scale_factor=1, *, hifigan_config=None, dtype: str = "fp16", device="cuda:0", **_):
        assert subband == 1, "freq_merge_subband is the identity only for subband == 1 (autoencoder.py:126-135)"
        dd = dict(ddconfig)
        self.vae_cfg = dict(ch=dd["ch"], ch_mult=list(dd["ch_mult"]), num_res_blocks=dd["num_res_blocks"],
                            z_channels=dd["z_channels"], out_ch=dd["out_ch"], embed_dim=embed_dim or dd["z_channels"],
                            scale_factor=scale_factor)
        assert not dd.get("attn_resolutions"), "attn_resolutions must be empty (released Tango VAE)"
        self.scale_factor = scale_factor
        self.embed_dim = self.vae_cfg["embed_dim"]
        self._device = torch.device(device)
        self.engine = Engine(vae=self.vae_cfg, hifigan=hifigan_config or HIFIGAN_CONFIG, dtype=dtype, device=device)

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
