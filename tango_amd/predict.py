"""Serving entry point on the engine: the cog `Predictor` of the reference (predict.py:29-67) -- `setup()` loads the
models once, `predict()` turns one prompt into a 16 kHz wav file -- on top of `tango_amd.Tango`.

cog itself is a deployment wrapper, not a dependency: when it is importable its `BasePredictor` / `Input` / `Path` are
used, otherwise small stand-ins with the same call surface, so the class can be driven (and tested) without it.  Weights
are expected under `<model_cache>/<name>/` (vae_config.json, main_config.json, pytorch_model_{vae,main}.bin -- the layout
predict.py:73-86 reads); nothing is downloaded here.  The reference also loads an STFT module there; it is not on the
generation path and is not built.  WAV files are written with the stdlib (batch_inference.write_wav)."""
import copy
import os
from typing import Dict, Iterable, Optional

from .batch_inference import SAMPLE_RATE, write_wav

try:                                                   # pragma: no cover - depends on the deployment image
    from cog import BasePredictor, Input, Path
except ImportError:
    class BasePredictor:                               # noqa: D401 - minimal stand-ins
        def setup(self):
            raise NotImplementedError

    def Input(description: str = "", default=None, choices: Optional[Iterable] = None, **_):
        return default

    Path = str

MODEL_CACHE = "tango_weights"                           # predict.py:18


class Predictor(BasePredictor):
    #: predict.py:35 -- the models the reference's cog image serves
    model_names = ("tango2", "tango2-full")

    def setup(self, model_cache: str = MODEL_CACHE, names: Optional[Iterable[str]] = None, device: str = "cuda:0", dtype: str = "fp16",
              text_encoder=None, tokenizer=None, tango_cls=None):
        """Load the models into memory once (predict.py:30-35).  `tango_cls` defaults to tango_amd.Tango."""
        if tango_cls is None:
            from .tango import Tango as tango_cls       # needs the HIP library + a GPU: fails loudly otherwise
        names = tuple(names) if names is not None else self.model_names
        self.models: Dict[str, object] = {}
        for k in names:
            path = os.path.join(model_cache, k)
            if not os.path.isdir(path):
                raise FileNotFoundError("model directory %s not found (the reference downloads %s here; there is no network)"
                                        % (path, "https://weights.replicate.delivery/default/declare-lab/tango.tar"))
            # every model loads ITS checkpoint's text_encoder.* tensors into the encoder it is given: a shared module
            # would end up with the last model's weights for all of them, so each model gets its own copy
            # (the string "engine" builds a separate on-engine encoder per model anyway)
            te = copy.deepcopy(text_encoder) if text_encoder is not None and not isinstance(text_encoder, str) else text_encoder
            self.models[k] = tango_cls(path, device=device, dtype=dtype, text_encoder=te, tokenizer=tokenizer)

    def predict(self,
                prompt: str = Input(description="Input prompt", default="Quiet speech and then and airplane flying away"),
                model: str = Input(description="choose a model", choices=["tango2", "tango2-full"], default="tango2"),
                steps: int = Input(description="inference steps", default=100),
                guidance: float = Input(description="guidance scale", default=3),
                out: str = "/tmp/output.wav") -> Path:
        """Run a single prediction (predict.py:37-67): generate -> write 16 kHz wav -> return its path."""
        if model not in self.models:
            raise KeyError("unknown model %r; loaded: %s" % (model, sorted(self.models)))
        audio = self.models[model].generate(prompt, steps, guidance)
        write_wav(out, audio, SAMPLE_RATE)
        return Path(out)
