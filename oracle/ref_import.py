"""TEST INFRASTRUCTURE ONLY -- imports the *real* reference modules from /root/reference.

This file never ships with the product and is never imported by `tango_amd/`.  It exists so that
(1) `oracle/tango_oracle.py` (the CPU restatement) can be differentially pinned against the
reference's own code in this container, and (2) `oracle/make_golden.py` can emit the fixtures
committed under `tests/golden/`.  `/root/reference` does not exist on the GPU box, so everything
here is gated on `available()`.

Import recipe: the in-tree diffusers fork
(`mustango/diffusers/src/diffusers`, v0.15.0.dev0) cannot be imported as a package because its
`__init__` pulls in pipelines that need pip packages missing here; we register path-only package
shells so only the model/scheduler sub-modules are executed.
"""
import os
import sys
import types

import importlib.machinery


def _stub(name):
    """empty module with a real ModuleSpec: other packages probe optional dependencies with importlib.util.find_spec(), which
    raises on a sys.modules entry whose __spec__ is None (transformers.audio_utils does, for soundfile and librosa)"""
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    return m


REF = os.environ.get("TANGO_REFERENCE", "/root/reference")
_FORK = os.path.join(REF, "mustango", "diffusers", "src", "diffusers")


def available() -> bool:
    return os.path.isdir(_FORK) and os.path.isdir(os.path.join(REF, "audioldm"))


_done = False


def _setup():
    global _done
    if _done:
        return
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    import huggingface_hub
    import huggingface_hub.constants as hc

    if not hasattr(hc, "hf_cache_home"):
        hc.hf_cache_home = os.path.expanduser("~/.cache/huggingface")
    for name in ("HfFolder", "cached_download"):
        if not hasattr(huggingface_hub, name):
            setattr(huggingface_hub, name, type(name, (), {}))
    if "diffusers" not in sys.modules:
        pkg = types.ModuleType("diffusers")
        pkg.__path__ = [_FORK]
        pkg.__version__ = "0.15.0.dev0"
        sys.modules["diffusers"] = pkg
    for stub in ("soundfile", "progressbar"):
        if stub not in sys.modules:
            try:
                __import__(stub)
            except Exception:
                sys.modules[stub] = _stub(stub)
    if "audioldm" not in sys.modules:
        pkg = types.ModuleType("audioldm")
        pkg.__path__ = [os.path.join(REF, "audioldm")]
        sys.modules["audioldm"] = pkg
    _done = True


def unet_cls():
    _setup()
    from diffusers.models.unet_2d_condition import UNet2DConditionModel

    return UNet2DConditionModel


def unet_music_cls():
    """mustango/diffusers/src/diffusers/models/unet_2d_condition_music.py UNet2DConditionModelMusic (Mustango)"""
    _setup()
    from diffusers.models.unet_2d_condition_music import UNet2DConditionModelMusic

    return UNet2DConditionModelMusic


def ddpm_cls():
    _setup()
    from diffusers.schedulers.scheduling_ddpm import DDPMScheduler

    return DDPMScheduler


def ddim_cls():
    _setup()
    from diffusers.schedulers.scheduling_ddim import DDIMScheduler

    return DDIMScheduler


def unet_blocks():
    _setup()
    import diffusers.models.unet_2d_blocks as m

    return m


def autoencoder_cls():
    _setup()
    from audioldm.variational_autoencoder.autoencoder import AutoencoderKL

    return AutoencoderKL


def tacotron_stft_cls():
    """audioldm/audio/stft.py:136 `TacotronSTFT`.  stft.py:5-6 and audio_processing.py:3 import librosa, which is neither in
    /root/reference nor installed (requirements.txt pins librosa==0.9.2): a stub module provides the three functions they
    use -- `librosa.util.pad_center`, `librosa.util.tiny`, `librosa.util.normalize` (only window_sumsquare's inverse path) and
    `librosa.filters.mel` -- from oracle/stft_oracle.py's restatements.  So a run of the imported class pins everything
    DOWNSTREAM of the filterbank (window, DFT basis, reflect padding, conv1d framing, magnitude, matmul, log-clamp, energy);
    the filterbank itself is pinned to transformers.audio_utils.mel_filter_bank instead (oracle/stft_oracle.py header)."""
    _setup()
    import numpy as np

    from oracle import stft_oracle as S
    if "librosa" not in sys.modules:
        lib, util, filters = _stub("librosa"), _stub("librosa.util"), _stub("librosa.filters")
        util.pad_center = lambda data, size, axis=-1, **kw: S.pad_center(np.asarray(data), size)
        util.tiny = lambda x: np.finfo(np.asarray(x).dtype if np.issubdtype(np.asarray(x).dtype, np.floating) else np.float32).tiny
        util.normalize = lambda x, norm=None, **kw: x
        filters.mel = lambda sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **kw: S.slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
        lib.util, lib.filters = util, filters
        sys.modules["librosa"], sys.modules["librosa.util"], sys.modules["librosa.filters"] = lib, util, filters
    # audioldm/audio/__init__.py imports tools.py -> torchaudio (absent; only the wav-file reader uses it): register the
    # sub-package path-only, like `audioldm` itself, so that only stft.py / audio_processing.py execute
    if "audioldm.audio" not in sys.modules:
        pkg = types.ModuleType("audioldm.audio")
        pkg.__path__ = [os.path.join(REF, "audioldm", "audio")]
        sys.modules["audioldm.audio"] = pkg
    from audioldm.audio.stft import TacotronSTFT

    return TacotronSTFT


def unet_config(name="diffusion_model_config.json"):
    import json

    cfg = json.load(open(os.path.join(REF, "configs", name)))
    return {k: v for k, v in cfg.items() if not k.startswith("_")}


def vae_config():
    import json

    return json.load(open(os.path.join(REF, "mustango", "configs", "vae_config.json")))
