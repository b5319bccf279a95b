"""The multi-GPU path on REAL RCCL (VERDICT r4 weak #9 / next #1c): tango_amd/parallel.py's collectives -- int64 header broadcast, fp32
embedding broadcast, uint8 mask broadcast, the byte-view gather of the int16 waveforms -- and bench.py's `dist.barrier(device_ids=...)`
/ float64 all_reduce, driven through a `nccl` process group with the real (tiny-config) engine as the compute function.

World size = min(2, visible GPUs).  On the driver's 8-GPU node (or any box with >= 2 GPUs) that is two ranks and the result must equal
the single-process result bit for bit (noise is keyed by the global sample index, so sharding must not change anything).  On a one-GPU
box the SAME lines run through a world-size-1 RCCL group (DataParallelGenerator takes the collective path whenever a process group
exists), which still executes every RCCL call with the dtypes and views the 8-GPU run will use.  The world-size-2 sharding logic itself
is covered on CPU by tests/test_parallel_gloo.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_SAMPLES = 4096
B, L, NSTEP = 4, 8, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    from oracle import tango_oracle as O
    g = torch.Generator().manual_seed(77)
    d = O.UNET_CONFIG_TINY["cross_attention_dim"]
    cond, unc = torch.randn(B, L, d, generator=g), torch.randn(B, L, d, generator=g)
    mc = torch.ones(B, L, dtype=torch.bool)
    mc[1, 5:] = False
    mu = torch.zeros(B, L, dtype=torch.bool)
    mu[:, 0] = True
    return torch.cat([unc, cond]), torch.cat([mu, mc])


def _make_compute(device):
    """the real engine (tiny UNet config, seeded synthetic weights) as DataParallelGenerator's compute function: a 2-step CFG denoise with
    device Philox noise keyed by the global sample index; the 'waveform' is the int16 image of the first N_SAMPLES latent values"""
    from oracle import tango_oracle as O
    from tango_amd.engine import Engine
    from tango_amd.scheduler import SD21_SCHEDULER_CONFIG, DDPMScheduler
    keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample", "variance_type")
    e = Engine(unet=O.UNET_CONFIG_TINY, dtype="fp32", device=device)
    e.load_synthetic(1234)
    sch = DDPMScheduler.from_config({k: SD21_SCHEDULER_CONFIG[k] for k in keys})
    sch.set_timesteps(NSTEP)

    def compute(pe, pm, offset, seed):
        b = pe.shape[0] // 2
        lat = torch.stack([torch.randn(8, 256, 16, generator=torch.Generator().manual_seed(500 + offset + i)) for i in range(b)]).to(device)
        e.denoise(lat, pe, pm, sch.timesteps.numpy(), sch.coef_table(), 3.0, seed=seed, sample_offset=offset, use_graph=True)
        return (lat.flatten(1)[:, :N_SAMPLES] * 2000.0).clamp(-32000, 32000).to(torch.int16)

    return compute, e


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from tango_amd.parallel import DataParallelGenerator
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        compute, e = _make_compute(device)
        dp = DataParallelGenerator(compute, device)
        assert dp.collective and dp.world == world
        pe, pm = _inputs() if rank == 0 else (None, None)
        out = dp.generate(pe, pm, 3.0, N_SAMPLES, seed=4711 if rank == 0 else None)
        # bench.py's own RCCL calls (bench.py main(): barrier with device_ids, MAX / SUM all_reduce of a float64 pair)
        dist.barrier(device_ids=[rank])
        t = torch.tensor([1.5 + rank, 1.0], device=device, dtype=torch.float64)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        assert abs(float(t[0]) - (1.5 + world - 1)) < 1e-12 and int(round(float(t[1]))) == world
        if rank == 0:
            q.put(out)
        else:
            assert out is None
        del e
    finally:
        dist.destroy_process_group()


def test_dp_generator_over_rccl_equals_single_process():
    from tango_amd.parallel import DataParallelGenerator
    world = min(2, torch.cuda.device_count())
    assert world >= 1
    # the single-process result (no process group): the reference for "sharding changes nothing"
    compute, e = _make_compute(torch.device("cuda", 0))
    pe, pm = _inputs()
    single = DataParallelGenerator(compute, torch.device("cuda", 0)).generate(pe, pm, 3.0, N_SAMPLES, seed=4711)
    del e, compute
    torch.cuda.synchronize()
    assert single.shape == (B, N_SAMPLES) and single.dtype == np.int16 and np.abs(single.astype(np.int32)).max() > 100
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=600)
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    print("DataParallelGenerator over RCCL, world size %d: %s int16, max |diff| vs single process %d"
          % (world, out.shape, np.abs(out.astype(np.int32) - single.astype(np.int32)).max()))
    assert out.dtype == np.int16 and out.shape == (B, N_SAMPLES)
    assert np.array_equal(out, single), "the RCCL path must reproduce the single-process result bit for bit"
