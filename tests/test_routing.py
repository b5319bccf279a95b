"""The GEMM dispatcher's measured routing rules (csrc/gemm.hip gemm_route and the *_ok() predicates), pinned on the UNet's linear shapes
through the host-side query tango_debug_linear_route -- no GPU needed.  Every expectation below is a same-process A/B on an MI355X
(profiles/r3_c13_*, r4_c7_duo_*, r4_c8_duo_default_*, r4_c13_ln_xstats_*, r4_c16_wide_pers_*): a change of a threshold that flips one
of them should come with a new measurement."""
import os

import pytest

from tango_amd import _lib

SWITCHES = ["TANGO_DUO_MAXK", "TANGO_DUO_MIN_TILES", "TANGO_DUO_MASK", "TANGO_WIDE_PERS", "TANGO_NO_LN_XSTATS", "TANGO_NO_WIDE_GEMM",
            "TANGO_NO_STREAM", "TANGO_NO_DMA_GEMM", "TANGO_FORCE_DMA_GEMM", "TANGO_NO_SMALL_TILE", "TANGO_NO_STREAM_LN_GEGLU"]


@pytest.fixture()
def route():
    lib = _lib.load()
    saved = {k: os.environ.pop(k) for k in SWITCHES if k in os.environ}
    lib.tango_tuning_reload()
    yield lambda *a, dtype=1: lib.tango_debug_linear_route(dtype, *a).decode()
    os.environ.update(saved)
    lib.tango_tuning_reload()


# (M, N, K, geglu, ln_fold, residual, vt) -> kernel family
B32 = [
    ((262144, 320, 320, 0, 0, 1, 0), "wide+pers"),          # level-0 proj_in / to_out / proj_out: HBM-bound, four tiles per CU
    ((262144, 960, 320, 0, 1, 0, 1), "wide+pers"),          # level-0 q | k | v^T with folded LayerNorm
    ((262144, 2560, 320, 1, 1, 0, 0), "stream"),            # level-0 GEGLU projection: the streaming kernel (wide 4.06, duo 3.55 vs 3.38 ms)
    ((65536, 5120, 640, 1, 1, 0, 0), "wide+xstats+pers"),   # level-1 GEGLU: statistics pass + folded weights
    ((16384, 10240, 1280, 1, 1, 0, 0), "wide+xstats+pers"),
    ((65536, 640, 640, 0, 0, 1, 0), "wide+pers"),           # two tiles per CU
    ((16384, 1280, 1280, 0, 0, 1, 0), "wide"),              # one tile per CU: nothing to prefetch
    ((32768, 640, 640, 0, 0, 1, 0), "duo"),                 # the conditional half of a CFG batch: 256 tiles of 256 x 320 -> 512 of 256 x 160
    ((8192, 1280, 1280, 0, 0, 1, 0), "duo"),
    ((16384, 1280, 5120, 0, 0, 1, 0), "wide"),
]
B8 = [
    ((65536, 320, 320, 0, 0, 1, 0), "duo"),                 # K <= 640 with < 2 tiles of 256 x 320 per CU
    ((65536, 320, 1280, 0, 0, 1, 0), "wide"),               # ... but K = 1280 stays (0.340 vs 0.391 ms)
    ((65536, 960, 320, 0, 1, 0, 1), "stream"),
    ((16384, 640, 640, 0, 0, 1, 0), "duo"),                 # 0.546 -> 0.41 ms
    ((16384, 1920, 640, 0, 1, 0, 1), "duo"),
    ((16384, 640, 2560, 0, 0, 1, 0), "duo"),
    ((4096, 1280, 1280, 0, 0, 1, 0), "duo"),                # split-K before: 0.638 -> 0.546 ms
    ((4096, 1280, 5120, 0, 0, 1, 0), "tile+splitk"),        # 0.435 vs 0.552 ms on duo
    ((16384, 5120, 640, 1, 1, 0, 0), "wide+xstats+pers"),
]
B1 = [
    ((8192, 2560, 320, 1, 1, 0, 0), "stream"),
    ((2048, 5120, 640, 1, 1, 0, 0), "layernorm+duo"),       # too few 256 x 320 tiles for the statistics-pass form
    ((512, 1280, 1280, 0, 0, 1, 0), "tile+splitk"),
    ((2048, 640, 640, 0, 0, 1, 0), "tile"),                 # the 64 x 64 small tiles live under this family
]


@pytest.mark.parametrize("args,want", B32 + B8 + B1)
def test_measured_route(route, args, want):
    assert route(*args) == want, (args, route(*args))
    assert route(*args, dtype=2) == want                       # bf16 takes the same kernels


def test_switches_take_kernels_out_of_the_dispatch(route):
    lib = _lib.load()
    try:
        os.environ["TANGO_DUO_MAXK"] = "0"
        lib.tango_tuning_reload()
        assert route(32768, 640, 640, 0, 0, 1, 0) == "wide"
        assert route(4096, 1280, 1280, 0, 0, 1, 0) == "tile+splitk"
        os.environ["TANGO_WIDE_PERS"] = "0"
        os.environ["TANGO_NO_LN_XSTATS"] = "1"
        lib.tango_tuning_reload()
        assert route(262144, 320, 320, 0, 0, 1, 0) == "wide"
        assert route(65536, 5120, 640, 1, 1, 0, 0) == "layernorm+wide"
    finally:
        for k in ("TANGO_DUO_MAXK", "TANGO_WIDE_PERS", "TANGO_NO_LN_XSTATS"):
            os.environ.pop(k, None)
        lib.tango_tuning_reload()
