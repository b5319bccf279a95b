"""Cooperative single-launch GroupNorm (norm.hip gn_coop_kernel): the statistics grid keeps its rows in registers across a
per-sample rendezvous (agent-scope atomics) and normalises them itself.  Checked against torch's fp32 group_norm, against the
two-launch path (TANGO_NO_GN_COOP=1), for run-to-run bit identity, and for the self-resetting barrier words under back-to-back
launches of different geometries."""
import contextlib
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"fp32": 0, "fp16": 1, "bf16": 2}
TOL = {"fp32": 2e-5, "fp16": 4e-3, "bf16": 3e-2}


def q(t, dtype):
    return t.half().float() if dtype == "fp16" else t.bfloat16().float() if dtype == "bf16" else t


def p(t):
    return C.c_void_p(t.data_ptr())


@contextlib.contextmanager
def tuning(lib, **env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    lib.tango_tuning_reload()
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib.tango_tuning_reload()


def gn(lib, dtype, x, ga, be, eps, act):
    B, Cc, HW = x.shape
    out = torch.zeros(B, Cc, HW, device="cuda")
    rc = lib.tango_op_groupnorm(DT[dtype], p(x), p(ga), p(be), p(out), B, Cc, HW, 32, C.c_float(eps), act, None)
    assert rc == 0, lib.tango_last_error().decode()
    return out


# (samples, channels, rows): the UNet's GroupNorm inputs at B = 1 (2 samples) and B = 8 (16 samples), ragged row counts, one row group per
# workgroup (C = 1280 / 1920: 160 / 240 vectors per row), several (C = 320: 6 rows per pass)
SHAPES = [(2, 320, 4096), (2, 960, 4096), (16, 640, 1024), (16, 1280, 256), (2, 1280, 64), (16, 1920, 256), (3, 320, 100), (5, 64, 777), (2, 2048, 40)]


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("B,Cc,HW", SHAPES)
def test_gn_coop_matches_torch_and_two_launch_path(lib, dtype, B, Cc, HW):
    if dtype == "fp32" and Cc > 1024:
        pytest.skip("fp32 rows of more than 256 16-byte vectors take the two-launch path")
    g = torch.Generator().manual_seed(B + Cc + HW)
    x = q(torch.randn(B, Cc, HW, generator=g) * 1.7 + 0.4, dtype).cuda()
    ga, be = torch.randn(Cc, generator=g).cuda(), torch.randn(Cc, generator=g).cuda()
    for eps, act in ((1e-5, 1), (1e-6, 0)):
        ref = F.group_norm(x, 32, ga, be, eps)
        if act:
            ref = F.silu(ref)
        with tuning(lib, TANGO_GN_SLAB=0, TANGO_GN_COOP_ALL=1):
            first = gn(lib, dtype, x, ga, be, eps, act)
            err = ((first - ref).abs().max() / (ref.abs().max() + 1e-9)).item()
            assert err <= TOL[dtype], "gn_coop %s B=%d C=%d rows=%d: rel err %.3e" % (dtype, B, Cc, HW, err)
            for rep in range(10):
                assert torch.equal(gn(lib, dtype, x, ga, be, eps, act), first), "repetition %d differs" % rep
        with tuning(lib, TANGO_GN_SLAB=0, TANGO_NO_GN_COOP=1):
            two = gn(lib, dtype, x, ga, be, eps, act)
        # same arithmetic, different chunking of the fp32 partial sums: at most an output ulp apart
        d = ((first - two).abs().max() / (ref.abs().max() + 1e-9)).item()
        assert d <= TOL[dtype] / 2, "gn_coop vs two-launch: %.3e" % d


def test_gn_coop_barrier_words_survive_mixed_geometries(lib):
    """200 back-to-back launches alternating between geometries with different workgroup counts per sample: the arrival counters must be
    back at rest after every launch (a stale count would deadlock -> the kernel's timeout flag -> an error from the op)"""
    g = torch.Generator().manual_seed(7)
    cases = []
    with tuning(lib, TANGO_GN_SLAB=0, TANGO_GN_COOP_ALL=1):
        for B, Cc, HW in [(2, 320, 4096), (16, 1280, 256), (3, 320, 100), (2, 1280, 64)]:
            x = q(torch.randn(B, Cc, HW, generator=g), "fp16").cuda()
            ga, be = torch.randn(Cc, generator=g).cuda(), torch.randn(Cc, generator=g).cuda()
            cases.append((x, ga, be, gn(lib, "fp16", x, ga, be, 1e-5, 1)))
        for it in range(50):
            for x, ga, be, first in cases:
                assert torch.equal(gn(lib, "fp16", x, ga, be, 1e-5, 1), first), it


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
@pytest.mark.parametrize("B,Cc,HW", [(2, 320, 4096), (16, 640, 1024), (3, 320, 100), (2, 1280, 64)])
def test_gn_coop_fallback_without_rendezvous_is_bit_identical(lib, dtype, B, Cc, HW):
    """the kernel's co-residency is an estimate, so a workgroup whose partners do not arrive must not normalise with
    incomplete sums.  It re-reduces every chunk of its sample itself, in phase 1's order.  TANGO_GN_COOP_FORCE_FALLBACK=1 sends EVERY
    workgroup down that path without waiting for anyone: the output must equal the rendezvous path's bit for bit, and the barrier
    words must be back at rest (the following normal launches neither hang nor differ)."""
    g = torch.Generator().manual_seed(11 * B + Cc + HW)
    x = q(torch.randn(B, Cc, HW, generator=g) * 1.3 - 0.2, dtype).cuda()
    ga, be = torch.randn(Cc, generator=g).cuda(), torch.randn(Cc, generator=g).cuda()
    with tuning(lib, TANGO_GN_SLAB=0, TANGO_GN_COOP_ALL=1):
        normal = gn(lib, dtype, x, ga, be, 1e-5, 1)
    with tuning(lib, TANGO_GN_SLAB=0, TANGO_GN_COOP_ALL=1, TANGO_GN_COOP_FORCE_FALLBACK=1):
        forced = gn(lib, dtype, x, ga, be, 1e-5, 1)
        forced2 = gn(lib, dtype, x, ga, be, 1e-5, 1)
    with tuning(lib, TANGO_GN_SLAB=0, TANGO_GN_COOP_ALL=1):
        again = gn(lib, dtype, x, ga, be, 1e-5, 1)
    assert torch.equal(forced, normal) and torch.equal(forced2, normal) and torch.equal(again, normal)
