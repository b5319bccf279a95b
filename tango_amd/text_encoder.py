"""`T5EncoderOnEngine` -- the FLAN-T5 text encoder on the HIP engine.

Drop-in for the `transformers.T5EncoderModel` instance the reference keeps in `AudioDiffusion.text_encoder`
(models.py:98-100): `enc(input_ids=ids, attention_mask=am)[0]` -> last_hidden_state [B, L, d_model] (models.py:139-141,
279-281, 291-293).  Weights are the `text_encoder.*` tensors of pytorch_model_main.bin (or a T5EncoderModel state_dict).
Arithmetic is fp32 end to end (f32 MFMA): FLAN-T5 hidden states exceed the fp16 range, and the encoder is <1 % of a
200-step generation, so there is nothing to win from reduced precision here.
"""
from typing import Dict, Optional

import torch

from . import weights as W
from .engine import Engine


class _Output(tuple):
    """`outputs[0]` and `outputs.last_hidden_state`, like transformers' BaseModelOutput."""

    @property
    def last_hidden_state(self):
        return self[0]


class T5EncoderOnEngine:
    def __init__(self, config: dict, device="cuda:0"):
        self.config = dict(config)
        self.device = torch.device(device)
        self.engine = Engine(t5=self.config, dtype="fp32", device=device)
        self.engine.set_plan_budget(4 << 30)     # (batch, length) plans of the text encoder are tens of MB each: keep the cache small

    @classmethod
    def from_state_dict(cls, sd: Dict[str, torch.Tensor], prefix: str = "text_encoder.", device="cuda:0"):
        """Build from checkpoint tensors; the hyper-parameters are recovered from the tensor shapes."""
        enc = cls(W.t5_config_from_state_dict(sd, prefix), device=device)
        enc.load_state_dict({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)})
        return enc

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """Keys as in T5EncoderModel.state_dict() (no prefix).  `shared.weight` may come as `encoder.embed_tokens.weight`."""
        sd = dict(sd)
        if "shared.weight" not in sd and "encoder.embed_tokens.weight" in sd:
            sd["shared.weight"] = sd["encoder.embed_tokens.weight"]
        known = set(k[len("text_encoder."):] for k in self.engine.weight_names()) | {"encoder.embed_tokens.weight"}
        unexpected = sorted(k for k in sd if k not in known)
        if strict and unexpected:
            raise RuntimeError("text_encoder state_dict mismatch: unexpected %s" % unexpected[:4])
        missing = self.engine.load_state_dict({"text_encoder." + k: v for k, v in sd.items()}, strict=strict)
        self.engine.finalize()
        return missing

    def eval(self):
        return self

    def to(self, *_args, **_kw):
        return self

    @torch.no_grad()
    def __call__(self, input_ids=None, attention_mask: Optional[torch.Tensor] = None, **_):
        return _Output((self.engine.encode_text(input_ids, attention_mask),))
