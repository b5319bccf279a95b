"""Data-parallel generation across the GPUs of one node.

The path shards embarrassingly: every prompt's trajectory is independent (GroupNorm / LayerNorm /
attention are per-sample; a prompt's CFG twin stays on its GPU), so there is NO per-step collective.
One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI); per batch: one small header
broadcast, a broadcast of the text-encoder outputs from rank 0 (north-star: "RCCL broadcast of text
embeddings over xGMI only") and a gather of the int16 waveforms back to rank 0.  The reference has no
inference-side parallelism (tango.py:54-60 is a Python loop over chunks) -- this replaces that loop.

Step noise is keyed by (seed, GLOBAL sample index): rank 0 draws ONE seed per batch and broadcasts it in the
header, so results do not depend on the number of ranks and no per-rank call counter can drift.
"""
import os
from typing import Callable, List, Optional, Sequence, Tuple

# dmabuf IPC: the host driver of this platform supports no legacy IPC handles, and without this RCCL's intra-node transports fail with
# `hipIpcGetMemHandle: invalid argument`.  It has to be in the environment before the HIP runtime initialises, i.e. it is set where the
# multi-process path is IMPORTED (every rank imports this module before it touches its GPU), not only by bench.py's self-launcher.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of n items over `world` ranks (first n % world ranks get one extra)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_cfg_embeddings(prompt_embeds: torch.Tensor, mask: torch.Tensor, world: int, rank: int, cfg_on: bool):
    """Slice the global [uncond; cond] batch (models.py:301 order) so each rank keeps a prompt and
    its unconditional twin together.  Returns (embeds_local, mask_local, first_global_sample)."""
    if cfg_on:
        B = prompt_embeds.shape[0] // 2
        lo, hi = shard_bounds(B, world, rank)
        pe = torch.cat([prompt_embeds[lo:hi], prompt_embeds[B + lo:B + hi]])
        pm = torch.cat([mask[lo:hi], mask[B + lo:B + hi]])
    else:
        B = prompt_embeds.shape[0]
        lo, hi = shard_bounds(B, world, rank)
        pe, pm = prompt_embeds[lo:hi], mask[lo:hi]
    return pe, pm, lo


class DataParallelGenerator:
    """compute(embeds_local, mask_local, sample_offset, seed) -> int16 [b_local, n_samples]
    (a torch tensor on `device` -- it is gathered without leaving the device -- or a numpy array)."""

    def __init__(self, compute: Callable, device: torch.device, group=None):
        self.compute = compute
        self.device = torch.device(device)
        self.group = group
        # `collective`: a process group exists, so every exchange below goes through it -- also at world size 1, which is how the RCCL
        # calls of this file (int64 / fp32 / uint8 broadcast, byte gather) are exercised on a one-GPU box (tests/test_parallel_nccl_gpu.py)
        self.collective = dist.is_initialized()
        self.world = dist.get_world_size(group) if self.collective else 1
        self.rank = dist.get_rank(group) if self.collective else 0

    def _bcast(self, t: Optional[torch.Tensor], shape, dtype, keep_host: bool = False):
        if not self.collective and keep_host and t is not None and not t.is_cuda:
            return t.to(dtype).contiguous()          # single process: a host-resident mask stays on the host (no read-back later)
        if self.rank != 0:
            t = torch.empty(shape, dtype=dtype, device=self.device)
        else:
            t = t.to(self.device, dtype).contiguous()
        if self.collective:
            dist.broadcast(t, src=0, group=self.group)
        return t

    def generate(self, prompt_embeds: Optional[torch.Tensor], mask: Optional[torch.Tensor], guidance: float,
                 n_samples: int, seed: Optional[int] = None, after_compute: Optional[Callable] = None) -> Optional[np.ndarray]:
        """Rank 0 passes the global embeddings ([2B, L, d] when guidance > 1); other ranks pass None.
        `seed`: rank 0's value is used (None: drawn from torch's default generator on rank 0).
        `after_compute`: called on every rank once this pass's device work is enqueued and before the gather waits for it -- the
        slot where rank 0 tokenises / encodes the NEXT pass (generate_for_batch_dp), hidden behind this pass's denoise.
        Returns the global int16 waveforms [B, n_samples] on rank 0, None elsewhere."""
        cfg_on = guidance > 1.0
        hdr = torch.zeros(4, dtype=torch.int64, device=self.device)
        if self.rank == 0:
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            hdr = torch.tensor(list(prompt_embeds.shape) + [int(seed)], dtype=torch.int64, device=self.device)
        if self.collective:
            dist.broadcast(hdr, src=0, group=self.group)
        n2, L, d, seed = [int(v) for v in hdr.tolist()]
        pe = self._bcast(prompt_embeds, (n2, L, d), torch.float32)
        pm = self._bcast(mask.to(torch.uint8) if mask is not None else None, (n2, L), torch.uint8, keep_host=True)
        B = n2 // 2 if cfg_on else n2
        pe_l, pm_l, lo = shard_cfg_embeddings(pe, pm.bool(), self.world, self.rank, cfg_on)
        b_local = pe_l.shape[0] // 2 if cfg_on else pe_l.shape[0]
        wav = None
        if b_local > 0:
            wav = self.compute(pe_l, pm_l, lo, seed)
            if isinstance(wav, np.ndarray):
                wav = torch.from_numpy(wav)
            assert wav.dtype == torch.int16 and tuple(wav.shape) == (b_local, n_samples), (wav.dtype, tuple(wav.shape))
        if after_compute is not None:
            after_compute()
        if not self.collective:
            return wav.cpu().numpy() if wav is not None else np.zeros((0, n_samples), np.int16)
        bmax = (B + self.world - 1) // self.world
        buf = torch.zeros((bmax, n_samples), dtype=torch.int16, device=self.device)
        if wav is not None:
            buf[:b_local] = wav.to(self.device)
        # neither RCCL nor gloo has an int16 datatype: ship the waveform bytes
        raw = buf.view(torch.uint8)
        outs8 = [torch.empty_like(raw) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(raw, outs8, dst=0, group=self.group)
        if self.rank != 0:
            return None
        parts: List[np.ndarray] = []
        for r in range(self.world):
            a, bnd = shard_bounds(B, self.world, r)
            parts.append(outs8[r].view(torch.int16)[: bnd - a].cpu().numpy())
        return np.concatenate(parts, 0)


def generate_for_batch_dp(prompts: Optional[Sequence[str]], encode: Callable, compute: Callable, n_samples: int, device,
                          guidance: float = 3, samples: int = 1, batch_size: int = 8, group=None):
    """Prompt-level data-parallel `generate_for_batch` (tango.py:51-64 is the loop this shards).

    Every rank calls it; only rank 0's `prompts` are read (other ranks may pass None).  Per pass rank 0 runs the text
    encoder on `batch_size * world` prompts (`encode(batch, samples, guidance) -> (embeds [n2, L, d], bool mask)`, the
    serial stage SURVEY.md 8e names), the embeddings are broadcast, each rank generates its contiguous shard
    (`compute(embeds_local, mask_local, sample_offset, seed) -> int16 [b_local, n_samples]`) and rank 0 gathers.
    Returns on rank 0 what `Tango.generate_for_batch` returns (list of waveforms, grouped per prompt when samples > 1);
    None on the other ranks."""
    dp = DataParallelGenerator(compute, torch.device(device), group)
    n = torch.tensor([len(prompts) if dp.rank == 0 else 0], dtype=torch.int64, device=dp.device)
    if dp.collective:
        dist.broadcast(n, src=0, group=group)
    n = int(n.item())
    per_pass = batch_size * dp.world
    outputs: List[np.ndarray] = []
    # the text encoder is the serial stage (rank 0 encodes for every rank): pass k+1 is tokenised / encoded while pass k's
    # denoise is still running on the device (the engine's calls do not wait for the stream), before the gather waits for it
    nxt = {}

    def encode_pass(k):
        if dp.rank == 0 and k < n:
            nxt[k] = encode(list(prompts[k:k + per_pass]), samples, guidance)

    encode_pass(0)
    for k in range(0, n, per_pass):
        pe, pm = nxt.pop(k) if dp.rank == 0 else (None, None)
        wav = dp.generate(pe, pm, guidance, n_samples, after_compute=lambda k=k: encode_pass(k + per_pass))
        if dp.rank == 0:
            outputs += [w for w in wav]
    if dp.rank != 0:
        return None
    if samples == 1:
        return outputs
    return [outputs[i:i + samples] for i in range(0, len(outputs), samples)]
