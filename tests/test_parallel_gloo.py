"""N > 1 path on CPU: world_size-2 gloo processes run the data-parallel generator (broadcast of the text
embeddings, per-rank shard, gather of int16 waveforms) with a stand-in compute function."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tango_amd.parallel import DataParallelGenerator, generate_for_batch_dp

N_SAMPLES = 37


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_compute(pe, pm, offset, seed=0):
    """deterministic per-prompt 'audio': depends on the cond AND uncond rows, the global sample index and the batch seed"""
    b = pe.shape[0] // 2
    out = np.zeros((b, N_SAMPLES), np.int16)
    for i in range(b):
        assert bool(pm[i, 0]) and not bool(pm[i, 1:].any()), "uncond twin must travel with its prompt"
        v = int(pe[b + i].sum().item()) * 3 + int(pe[i].sum().item()) + (offset + i) * 100 + seed % 1000
        out[i] = (np.arange(N_SAMPLES) + v) % 30000
    return out


def _global_inputs(B, L, d):
    g = torch.Generator().manual_seed(3)
    cond = torch.randint(0, 5, (B, L, d), generator=g).float()
    unc = torch.randint(0, 5, (B, L, d), generator=g).float()
    mc = torch.ones(B, L, dtype=torch.bool)
    mu = torch.zeros(B, L, dtype=torch.bool)
    mu[:, 0] = True
    return torch.cat([unc, cond]), torch.cat([mu, mc])


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dp = DataParallelGenerator(_fake_compute, torch.device("cpu"))
        pe, pm = _global_inputs(B, 4, 6) if rank == 0 else (None, None)
        out = dp.generate(pe, pm, 3.0, N_SAMPLES, seed=4711 if rank == 0 else None)   # only rank 0's seed counts
        if rank == 0:
            q.put(out)
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 8, 1])
def test_dp_two_ranks_equals_single(B):
    pe, pm = _global_inputs(B, 4, 6)
    single = DataParallelGenerator(_fake_compute, torch.device("cpu")).generate(pe, pm, 3.0, N_SAMPLES, seed=4711)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.dtype == np.int16 and out.shape == (B, N_SAMPLES)
    assert np.array_equal(out, single), "sharded result must equal the single-process result, in prompt order"


# ---- prompt-level entry point (Tango.generate_for_batch_dp binds these two callables to the engine) ----
def _fake_encode(batch, samples, guidance):
    """stand-in for encode_text_classifier_free: [uncond; cond] rows derived from the prompt strings"""
    L, d = 4, 6
    cond = torch.stack([torch.full((L, d), float(sum(map(ord, p)) % 7)) for p in batch]).repeat_interleave(samples, 0)
    unc = torch.zeros_like(cond)
    mc = torch.ones(cond.shape[0], L, dtype=torch.bool)
    mu = torch.zeros(cond.shape[0], L, dtype=torch.bool)
    mu[:, 0] = True
    return torch.cat([unc, cond]), torch.cat([mu, mc])


def _fake_compute_prompt(pe, pm, offset, seed):
    """audio that depends on the prompt only (not on the shard layout): results must not change with the world size"""
    b = pe.shape[0] // 2
    out = torch.zeros((b, N_SAMPLES), dtype=torch.int16)
    for i in range(b):
        assert bool(pm[i, 0]) and not bool(pm[i, 1:].any())
        out[i] = ((torch.arange(N_SAMPLES) + int(pe[b + i].sum().item())) % 30000).to(torch.int16)
    return out     # a torch tensor: the device-side gather path


PROMPTS = ["a dog barks", "rain on a roof", "wind", "a car passes by", "birds"]


def _worker_prompts(rank, world, port, samples, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)   # per-rank generators differ: the batch seed must still come from rank 0 only
        timings = {}
        out = generate_for_batch_dp(PROMPTS if rank == 0 else None, _fake_encode, _fake_compute_prompt, N_SAMPLES, "cpu",
                                    guidance=3.0, samples=samples, batch_size=2, timings=timings)
        if rank == 0:
            # the serial stage is timed per pass on rank 0 (5 prompts, 2 per rank and pass -> 2 passes); other ranks record nothing
            assert len(timings["encode_ms"]) == 2 and len(timings["pass_ms"]) == 2 and min(timings["encode_ms"]) >= 0.0
            q.put(out)
        else:
            assert out is None and timings == {}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("samples", [1, 2])
def test_generate_for_batch_dp_two_ranks(samples):
    single = generate_for_batch_dp(PROMPTS, _fake_encode, _fake_compute_prompt, N_SAMPLES, "cpu", guidance=3.0, samples=samples,
                                   batch_size=2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_prompts, args=(r, 2, port, samples, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert len(out) == len(PROMPTS) == len(single)
    if samples == 1:
        assert all(np.array_equal(a, b) for a, b in zip(out, single)) and out[0].shape == (N_SAMPLES,)
    else:
        assert all(len(g) == samples for g in out)
        assert all(np.array_equal(a, b) for g, h in zip(out, single) for a, b in zip(g, h))


def test_single_process_keeps_a_host_mask_on_the_host():
    """without a process group the generator must hand the tokenizer's HOST mask through untouched -- the engine then picks
    its plan from it instead of reading a device copy back (Engine.denoise `prompt_mask_host`); with a group the mask is broadcast."""
    seen = {}

    def compute(pe, pm, offset, seed=0):
        seen["mask_device"] = pm.device.type
        seen["mask_dtype"] = pm.dtype
        return _fake_compute(pe, pm, offset, seed)

    pe, pm = _global_inputs(3, 4, 6)
    out = DataParallelGenerator(compute, torch.device("cpu")).generate(pe, pm, 3.0, N_SAMPLES, seed=1)
    assert out.shape == (3, N_SAMPLES) and seen["mask_device"] == "cpu" and seen["mask_dtype"] == torch.bool
    dp = DataParallelGenerator(compute, torch.device("cpu"))
    assert dp.collective is False and dp.world == 1 and dp.rank == 0
