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
import time
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

    def __init__(self, compute: Callable, device: torch.device, group=None, timing: bool = False):
        self.compute = compute
        self.device = torch.device(device)
        self.group = group
        # `timing` (bench.py): every pass records the wall time of its three stages in `self.last_timing` -- header + embedding
        # broadcast, this rank's compute, waveform gather -- with a device synchronisation at each stage boundary, so that a
        # scaling run shows a straggler rank or a slow collective instead of one number.  Off on the product path (no extra syncs).
        self.timing = timing
        self.last_timing = None
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
        t_start = self._stamp()
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
        t_bcast = self._stamp()
        if b_local > 0:
            wav = self.compute(pe_l, pm_l, lo, seed)
            if isinstance(wav, np.ndarray):
                wav = torch.from_numpy(wav)
            assert wav.dtype == torch.int16 and tuple(wav.shape) == (b_local, n_samples), (wav.dtype, tuple(wav.shape))
        if after_compute is not None:
            after_compute()
        t_comp = self._stamp()
        if not self.collective:
            out = wav.cpu().numpy() if wav is not None else np.zeros((0, n_samples), np.int16)
            self._record(t_start, t_bcast, t_comp, self._stamp(), b_local)
            return out
        bmax = (B + self.world - 1) // self.world
        buf = torch.zeros((bmax, n_samples), dtype=torch.int16, device=self.device)
        if wav is not None:
            buf[:b_local] = wav.to(self.device)
        # neither RCCL nor gloo has an int16 datatype: ship the waveform bytes
        raw = buf.view(torch.uint8)
        outs8 = [torch.empty_like(raw) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(raw, outs8, dst=0, group=self.group)
        if self.rank != 0:
            self._record(t_start, t_bcast, t_comp, self._stamp(), b_local)
            return None
        parts: List[np.ndarray] = []
        for r in range(self.world):
            a, bnd = shard_bounds(B, self.world, r)
            parts.append(outs8[r].view(torch.int16)[: bnd - a].cpu().numpy())
        self._record(t_start, t_bcast, t_comp, self._stamp(), b_local)
        return np.concatenate(parts, 0)

    def _stamp(self):
        if not self.timing:
            return 0.0
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return time.perf_counter()

    def _record(self, t0, t1, t2, t3, b_local):
        if self.timing:
            self.last_timing = {"bcast_ms": 1e3 * (t1 - t0), "compute_ms": 1e3 * (t2 - t1), "gather_ms": 1e3 * (t3 - t2), "b_local": b_local}

    def timing_summary(self):
        """Every rank calls it after a timed pass: rank 0 gets {"per_rank_ms": {"min", "max", "all"}, "bcast_ms", "gather_ms"} of the
        LAST pass -- per-rank compute time (the straggler is max - min), the broadcast as the slowest rank saw it, the gather as rank 0
        saw it -- the other ranks None."""
        t = self.last_timing or {"bcast_ms": 0.0, "compute_ms": 0.0, "gather_ms": 0.0}
        mine = torch.tensor([t["compute_ms"], t["bcast_ms"], t["gather_ms"]], dtype=torch.float64, device=self.device)
        if self.collective:
            alls = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(alls, mine, group=self.group)
        else:
            alls = [mine]
        if self.rank != 0:
            return None
        rows = [a.cpu().tolist() for a in alls]
        comp = [r[0] for r in rows]
        return {"per_rank_ms": {"min": min(comp), "max": max(comp), "all": [round(c, 3) for c in comp]},
                "bcast_ms": max(r[1] for r in rows), "gather_ms": rows[0][2]}


def generate_for_batch_dp(prompts: Optional[Sequence[str]], encode: Callable, compute: Callable, n_samples: int, device,
                          guidance: float = 3, samples: int = 1, batch_size: int = 8, group=None, timings: Optional[dict] = None):
    """Prompt-level data-parallel `generate_for_batch` (tango.py:51-64 is the loop this shards).

    Every rank calls it; only rank 0's `prompts` are read (other ranks may pass None).  Per pass rank 0 runs the text
    encoder on `batch_size * world` prompts (`encode(batch, samples, guidance) -> (embeds [n2, L, d], bool mask)`, the
    serial stage SURVEY.md 8e names), the embeddings are broadcast, each rank generates its contiguous shard
    (`compute(embeds_local, mask_local, sample_offset, seed) -> int16 [b_local, n_samples]`) and rank 0 gathers.
    Returns on rank 0 what `Tango.generate_for_batch` returns (list of waveforms, grouped per prompt when samples > 1);
    None on the other ranks.  `timings` (a dict, optional): rank 0 appends the host wall time of every `encode` call to
    `timings["encode_ms"]` and of every pass to `timings["pass_ms"]` -- the text encoder is the only serial stage, and pass k+1's
    encode is issued while pass k's denoise runs, so encode_ms << pass_ms is the measurement that it is hidden."""
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
            t0 = time.perf_counter()
            nxt[k] = encode(list(prompts[k:k + per_pass]), samples, guidance)
            if timings is not None:
                timings.setdefault("encode_ms", []).append(1e3 * (time.perf_counter() - t0))

    encode_pass(0)
    for k in range(0, n, per_pass):
        pe, pm = nxt.pop(k) if dp.rank == 0 else (None, None)
        t0 = time.perf_counter()
        wav = dp.generate(pe, pm, guidance, n_samples, after_compute=lambda k=k: encode_pass(k + per_pass))
        if dp.rank == 0:
            outputs += [w for w in wav]
            if timings is not None:
                timings.setdefault("pass_ms", []).append(1e3 * (time.perf_counter() - t0))
    if dp.rank != 0:
        return None
    if samples == 1:
        return outputs
    return [outputs[i:i + samples] for i in range(0, len(outputs), samples)]
