"""Data-parallel generation across the GPUs of one node.

The path shards embarrassingly: every prompt's trajectory is independent (GroupNorm / LayerNorm /
attention are per-sample; a prompt's CFG twin stays on its GPU), so there is NO per-step collective.
One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI); exactly two collectives
per batch: a broadcast of the text-encoder outputs from rank 0 (north-star: "RCCL broadcast of text
embeddings over xGMI only") and a gather of the int16 waveforms back to rank 0.  The reference has no
inference-side parallelism (tango.py:54-60 is a Python loop over chunks) -- this replaces that loop.
"""
from typing import Callable, List, Optional, Tuple

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
    """compute(embeds_local, mask_local, sample_offset) -> np.int16 [b_local, n_samples]"""

    def __init__(self, compute: Callable, device: torch.device, group=None):
        self.compute = compute
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def _bcast(self, t: Optional[torch.Tensor], shape, dtype):
        if self.rank != 0:
            t = torch.empty(shape, dtype=dtype, device=self.device)
        else:
            t = t.to(self.device, dtype).contiguous()
        if self.world > 1:
            dist.broadcast(t, src=0, group=self.group)
        return t

    def generate(self, prompt_embeds: Optional[torch.Tensor], mask: Optional[torch.Tensor], guidance: float,
                 n_samples: int) -> Optional[np.ndarray]:
        """Rank 0 passes the global embeddings ([2B, L, d] when guidance > 1); other ranks pass None.
        Returns the global int16 waveforms [B, n_samples] on rank 0, None elsewhere."""
        cfg_on = guidance > 1.0
        hdr = torch.zeros(3, dtype=torch.int64, device=self.device)
        if self.rank == 0:
            hdr = torch.tensor(list(prompt_embeds.shape), dtype=torch.int64, device=self.device)
        if self.world > 1:
            dist.broadcast(hdr, src=0, group=self.group)
        n2, L, d = [int(v) for v in hdr.tolist()]
        pe = self._bcast(prompt_embeds, (n2, L, d), torch.float32)
        pm = self._bcast(mask.to(torch.uint8) if mask is not None else None, (n2, L), torch.uint8)
        B = n2 // 2 if cfg_on else n2
        pe_l, pm_l, lo = shard_cfg_embeddings(pe, pm.bool(), self.world, self.rank, cfg_on)
        b_local = pe_l.shape[0] // 2 if cfg_on else pe_l.shape[0]
        if b_local > 0:
            wav = np.asarray(self.compute(pe_l, pm_l, lo))
            assert wav.dtype == np.int16 and wav.shape == (b_local, n_samples), (wav.dtype, wav.shape)
        else:
            wav = np.zeros((0, n_samples), np.int16)
        if self.world == 1:
            return wav
        bmax = (B + self.world - 1) // self.world
        buf = torch.zeros((bmax, n_samples), dtype=torch.int16, device=self.device)
        buf[:b_local] = torch.from_numpy(wav).to(self.device)
        # neither RCCL nor gloo has an int16 datatype: ship the waveform bytes
        raw = buf.view(torch.uint8)
        outs8 = [torch.empty_like(raw) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(raw, outs8, dst=0, group=self.group)
        outs = [o.view(torch.int16) for o in outs8] if self.rank == 0 else None
        if self.rank != 0:
            return None
        parts: List[np.ndarray] = []
        for r in range(self.world):
            a, bnd = shard_bounds(B, self.world, r)
            parts.append(outs[r][: bnd - a].cpu().numpy())
        return np.concatenate(parts, 0)
