"""Batch generation driver: the caller side of the hot path.

Mirrors the generation + save half of the reference's inference.py:127-176: prompts come from a JSON-lines file
(`--test_file`, key `--text_key`), are generated in chunks of `--batch_size` through the three hot calls
(inference -> decode_first_stage -> decode_to_waveform) and written as 16 kHz int16 `output_{j}.wav` under
`outputs/<exp_id>_steps_<N>_guidance_<g>/`; with `--num_samples > 1` the k-th candidate of every prompt goes to
`rank_{k+1}/`.  A line is appended to `outputs/summary.jsonl`.

Deliberately NOT here (out of the hot-path scope): the audioldm_eval FD/KL/FAD metrics, wandb logging and the CLAP
re-ranking of multi-sample outputs (inference.py:96-125,178-190) -- candidates are written in generation order.

The generator is anything exposing `generate_for_batch(prompts, steps, guidance, samples, batch_size)` -- a
`tango_amd.Tango`, or a test double.  WAV files are written with the stdlib `wave` module (soundfile is not a dependency).
"""
import argparse
import json
import os
import time
import wave
from typing import Iterable, List, Optional, Sequence

import numpy as np

SAMPLE_RATE = 16000            # hifigan/utilities.py:9-39 (HIFIGAN_16K_64), inference.py:150


def read_prompts(test_file: str, text_key: str = "captions", prefix: str = "") -> List[str]:
    """inference.py:127-134: one JSON object per line; `prefix` is the training-time prompt prefix."""
    prompts = []
    with open(test_file) as f:
        for line in f:
            line = line.strip()
            if line:
                prompts.append(prefix + json.loads(line)[text_key])
    return prompts


def write_wav(path: str, samples: np.ndarray, sample_rate: int = SAMPLE_RATE) -> None:
    a = np.asarray(samples)
    if a.dtype != np.int16 or a.ndim != 1:
        raise ValueError("write_wav expects a 1-D int16 waveform, got %s %s" % (a.dtype, a.shape))
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(a.astype("<i2").tobytes())


def generate_and_save(generator, prompts: Sequence[str], num_steps: int = 200, guidance: float = 3, batch_size: int = 8,
                      num_samples: int = 1, out_root: str = "outputs", tag: str = "", exp_id: Optional[str] = None) -> dict:
    """Returns the summary record (also appended to <out_root>/summary.jsonl)."""
    if num_samples < 1 or batch_size < 1:
        raise ValueError("num_samples and batch_size must be >= 1")
    exp_id = exp_id or str(int(time.time()))
    name = "{}_{}steps_{}_guidance_{}".format(exp_id, (tag + "_") if tag else "", num_steps, guidance)
    out_dir = os.path.join(out_root, name)
    t0 = time.perf_counter()
    outs = generator.generate_for_batch(list(prompts), steps=num_steps, guidance=guidance, samples=num_samples,
                                        batch_size=batch_size) if prompts else []
    dt = time.perf_counter() - t0
    if len(outs) != len(prompts):
        raise RuntimeError("generator returned %d items for %d prompts" % (len(outs), len(prompts)))
    n_audio = 0
    if num_samples == 1:
        os.makedirs(out_dir, exist_ok=True)
        for j, wav in enumerate(outs):
            write_wav(os.path.join(out_dir, "output_{}.wav".format(j)), wav)
            n_audio += len(wav)
    else:
        for i in range(num_samples):
            os.makedirs(os.path.join(out_dir, "rank_{}".format(i + 1)), exist_ok=True)
        for j, group in enumerate(outs):
            if len(group) != num_samples:
                raise RuntimeError("prompt %d: %d candidates, expected %d" % (j, len(group), num_samples))
            for i, wav in enumerate(group):      # generation order (the reference re-ranks with CLAP here)
                write_wav(os.path.join(out_dir, "rank_{}".format(i + 1), "output_{}.wav".format(j)), wav)
                n_audio += len(wav)
    rec = {"Steps": num_steps, "Guidance Scale": guidance, "Test Instances": len(prompts), "Samples Per Prompt": num_samples,
           "output_dir": out_dir, "wall_seconds": dt, "audio_seconds": n_audio / float(SAMPLE_RATE),
           "audio_seconds_per_second": (n_audio / float(SAMPLE_RATE) / dt) if dt > 0 else None}
    os.makedirs(out_root, exist_ok=True)
    with open(os.path.join(out_root, "summary.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n\n")
    return rec


def parse_args(argv: Optional[Iterable[str]] = None):
    ap = argparse.ArgumentParser(description="Text-to-audio batch generation on the MI355X engine (generation + save half of inference.py).")
    ap.add_argument("--model", type=str, required=True, help="directory with the HF snapshot files (or a hub id)")
    ap.add_argument("--test_file", type=str, default="data/test_audiocaps_subset.json")
    ap.add_argument("--text_key", type=str, default="captions")
    ap.add_argument("--prefix", type=str, default="")
    ap.add_argument("--num_steps", type=int, default=200)
    ap.add_argument("--guidance", type=float, default=3)
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--num_samples", type=int, default=1)
    ap.add_argument("--device", type=str, default="cuda:0")
    ap.add_argument("--dtype", type=str, default="fp16", choices=["fp32", "fp16", "bf16"])
    ap.add_argument("--out_root", type=str, default="outputs")
    ap.add_argument("--text_encoder", type=str, default="torch", choices=["torch", "engine"],
                    help="torch: transformers T5EncoderModel under PyTorch-ROCm (the reference's module); engine: the checkpoint's "
                         "FLAN-T5 tensors on the HIP engine (tango_engine_encode_text)")
    return ap.parse_args(argv)


def main(argv: Optional[Iterable[str]] = None) -> dict:
    args = parse_args(argv)
    from .tango import Tango          # needs the HIP library and a GPU: fails loudly otherwise
    tango = Tango(args.model, device=args.device, dtype=args.dtype, text_encoder="engine" if args.text_encoder == "engine" else None)
    prompts = read_prompts(args.test_file, args.text_key, args.prefix)
    rec = generate_and_save(tango, prompts, args.num_steps, args.guidance, args.batch_size, args.num_samples, args.out_root,
                            tag="_".join(p for p in args.model.strip("/").split("/")[-2:] if p))
    print(json.dumps(rec))
    return rec


if __name__ == "__main__":
    main()
