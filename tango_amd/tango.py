"""`Tango` -- drop-in for tango.py:9-64 of the reference: same constructor, `generate`,
`generate_for_batch`; the three hot calls (inference -> decode_first_stage -> decode_to_waveform) run
on the HIP engine.  `name` may be a local directory holding the HF snapshot files
(vae_config.json, main_config.json, pytorch_model_{vae,main}.bin); hub download is attempted otherwise.
"""
import json
import os

import torch

from .autoencoder import AutoencoderKL
from .models import AudioDiffusion
from .scheduler import SD21_SCHEDULER_CONFIG, DDPMScheduler

_CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


class Tango:
    def __init__(self, name="declare-lab/tango", device="cuda:0", dtype="fp16", scheduler_config=None, text_encoder=None,
                 tokenizer=None):
        """`text_encoder` / `tokenizer` (optional): already-built T5EncoderModel / tokenizer, e.g. from a local directory
        on a box without hub access; the checkpoint's `text_encoder.*` tensors are loaded into it like the reference does."""
        if os.path.isdir(name):
            path = name
        else:
            from huggingface_hub import snapshot_download   # tango.py:12
            path = snapshot_download(repo_id=name)
        vae_config = json.load(open("{}/vae_config.json".format(path)))
        main_config = json.load(open("{}/main_config.json".format(path)))
        self.vae = AutoencoderKL(**vae_config, dtype=dtype, device=device)
        # tango.py:17-27: the wave -> mel front-end and its buffers (generation never calls it; training / evaluation callers
        # reach it as `tango.stft`, inference.py:81).  Snapshots without the two files simply have no `stft`.
        self.stft = None
        if os.path.exists("{}/stft_config.json".format(path)) and os.path.exists("{}/pytorch_model_stft.bin".format(path)):
            from .stft import TacotronSTFT
            self.stft = TacotronSTFT(**json.load(open("{}/stft_config.json".format(path))), device=device)
            self.stft.load_state_dict(torch.load("{}/pytorch_model_stft.bin".format(path), map_location="cpu"))
            self.stft.eval()
        cfg_path = main_config.get("unet_model_config_path")
        if cfg_path is None or not os.path.exists(cfg_path):
            # the reference resolves configs/diffusion_model_config.json relative to the cwd (models.py:83-85)
            base = os.path.basename(cfg_path) if cfg_path else "diffusion_model_config.json"
            cfg_path = os.path.join(_CONFIG_DIR, base)
        main_config = dict(main_config, unet_model_config_path=cfg_path)
        self.model = AudioDiffusion(**main_config, dtype=dtype, device=device, text_encoder=text_encoder, tokenizer=tokenizer)
        vae_weights = torch.load("{}/pytorch_model_vae.bin".format(path), map_location="cpu")
        main_weights = torch.load("{}/pytorch_model_main.bin".format(path), map_location="cpu")
        self.vae.load_state_dict(vae_weights)
        self.model.load_state_dict(main_weights)
        print("Successfully loaded checkpoint from:", name)
        self.vae.eval()
        self.model.eval()
        # tango.py:36 pulls stabilityai/stable-diffusion-2-1's scheduler JSON from the hub; config is data here
        self.scheduler = DDPMScheduler.from_config(_ddpm_keys(scheduler_config or SD21_SCHEDULER_CONFIG))

    @classmethod
    def from_components(cls, model: AudioDiffusion, vae: AutoencoderKL, scheduler=None):
        """Assemble from already-built components (synthetic-weight benchmarks, tests)."""
        self = cls.__new__(cls)
        self.model, self.vae, self.stft = model, vae, None
        self.scheduler = scheduler or DDPMScheduler.from_config(_ddpm_keys(SD21_SCHEDULER_CONFIG))
        return self

    def chunks(self, lst, n):
        """ Yield successive n-sized chunks from a list. """
        for i in range(0, len(lst), n):
            yield lst[i:i + n]

    def generate(self, prompt, steps=100, guidance=3, samples=1, disable_progress=True):
        """ Generate audio for a single prompt string. (tango.py:43-49) """
        with torch.no_grad():
            latents = self.model.inference([prompt], self.scheduler, steps, guidance, samples, disable_progress=disable_progress)
            mel = self.vae.decode_first_stage(latents)
            wave = self.vae.decode_to_waveform(mel)
        return wave[0]

    def generate_for_batch(self, prompts, steps=100, guidance=3, samples=1, batch_size=8, disable_progress=True):
        """ Generate audio for a list of prompt strings. (tango.py:51-64) """
        outputs = []
        for k in range(0, len(prompts), batch_size):
            batch = prompts[k: k + batch_size]
            with torch.no_grad():
                latents = self.model.inference(batch, self.scheduler, steps, guidance, samples, disable_progress=disable_progress)
                mel = self.vae.decode_first_stage(latents)
                wave = self.vae.decode_to_waveform(mel)
                outputs += [item for item in wave]
        if samples == 1:
            return outputs
        return list(self.chunks(outputs, samples))

    def generate_for_batch_dp(self, prompts, steps=100, guidance=3, samples=1, batch_size=8, group=None):
        """Data-parallel `generate_for_batch` over the ranks of the default (or given) torch.distributed group: one
        process per GPU, each holding a full `Tango`.  Every rank makes this call; rank 0's `prompts` are used, the text
        encoder runs on rank 0 only, embeddings travel by one RCCL broadcast per pass of `batch_size * world` prompts,
        int16 waveforms come back by one gather.  Returns the `generate_for_batch` result on rank 0, None elsewhere."""
        from .parallel import generate_for_batch_dp

        def encode(batch, n_per_prompt, g):
            if g > 1.0:
                pe, pm = self.model.encode_text_classifier_free(batch, n_per_prompt)
            else:
                pe, pm = self.model.encode_text(batch)
                pe, pm = pe.repeat_interleave(n_per_prompt, 0), pm.repeat_interleave(n_per_prompt, 0)
            return pe.float(), pm

        def compute(pe, pm, offset, seed):
            # initial latents keyed by (broadcast seed, GLOBAL sample index), like the step noise: ranks that share a
            # torch.manual_seed must not start their samples from identical noise, and the result must not depend on the
            # number of ranks (parallel.py docstring)
            b = pe.shape[0] // 2 if guidance > 1.0 else pe.shape[0]
            ecfg = self.model.engine._cfg
            lat = dp_initial_latents(seed, offset, b, self.model.unet.config.in_channels, ecfg.latent_h, ecfg.latent_w) * self.scheduler.init_noise_sigma
            latents = self.model.inference_from_embeddings(pe, pm, self.scheduler, steps, guidance, latents=lat, seed=seed,
                                                           sample_offset=offset)
            return self.vae.engine.vocode(self.vae.decode_first_stage(latents))      # int16 stays on the device

        with torch.no_grad():
            return generate_for_batch_dp(prompts, encode, compute, self.vae.engine.vocoder_samples(1024), self.model.device,
                                         guidance=guidance, samples=samples, batch_size=batch_size, group=group)

    def generate_from_embeddings(self, prompt_embeds, boolean_prompt_mask, steps=100, guidance=3, **kw):
        """Same three calls given the text-encoder outputs (benchmarks / data-parallel workers)."""
        with torch.no_grad():
            latents = self.model.inference_from_embeddings(prompt_embeds, boolean_prompt_mask, self.scheduler, steps, guidance, **kw)
            mel = self.vae.decode_first_stage(latents)
            return self.vae.decode_to_waveform(mel)


def dp_initial_latents(seed, offset, count, channels=8, height=256, width=16):
    """Initial latents [count, channels, height, width] (fp32, CPU; the engine's latent size) of the samples with GLOBAL indices
    offset .. offset+count-1 under the pass seed: one torch generator per sample, so a shard of a batch draws exactly what the
    unsharded batch would.  NOT the draw of `prepare_latents` (models.py:259-264: one randn of the whole batch from torch's global
    generator): `generate_for_batch_dp` at world size 1 and `generate_for_batch` under the same torch.manual_seed give different
    (equally distributed) samples -- the price of results that do not depend on the number of ranks."""
    out = []
    for i in range(count):
        g = torch.Generator(device="cpu").manual_seed((int(seed) * 1000003 + offset + i) % (2 ** 63 - 1))
        out.append(torch.randn(channels, height, width, generator=g))
    return torch.stack(out) if out else torch.zeros(0, channels, height, width)


def _ddpm_keys(cfg):
    keep = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample",
            "variance_type", "clip_sample_range")
    return {k: v for k, v in cfg.items() if k in keep}
