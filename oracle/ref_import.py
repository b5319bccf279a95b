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
                sys.modules[stub] = types.ModuleType(stub)
    if "audioldm" not in sys.modules:
        pkg = types.ModuleType("audioldm")
        pkg.__path__ = [os.path.join(REF, "audioldm")]
        sys.modules["audioldm"] = pkg
    _done = True


def unet_cls():
    _setup()
    from diffusers.models.unet_2d_condition import UNet2DConditionModel

    return UNet2DConditionModel


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


def unet_config(name="diffusion_model_config.json"):
    import json

    cfg = json.load(open(os.path.join(REF, "configs", name)))
    return {k: v for k, v in cfg.items() if not k.startswith("_")}


def vae_config():
    import json

    return json.load(open(os.path.join(REF, "mustango", "configs", "vae_config.json")))
