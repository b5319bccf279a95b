"""A/B arms and timing probes of ff_fused.hip at the benchmarked size (M = 262144, fp16): one process per TANGO_FF_VAR value (the switch is
read once; bit layout in csrc/ff_fused.hip).  Results of the ablation variants (bits 1-3) are wrong by construction."""
import ctypes as C
import os
import subprocess
import sys

NAMES = {
    -1: "shipped default", 0x000: "8-byte epilogue (round-6 first form)", 0x010: "8-byte epilogue, wn = 1 waves out of step",
    0x100: "packed-f32 GELU (common.h gelu_erf_poly2)", 0x320: "s_setprio 1 over the MFMA part", 0x520: "ring of four fragment pairs", 0x720: "ring of four + s_setprio", 0x140: "LDS-DMAs in one bunch behind the barrier", 0x180: "LDS-DMAs in front of the GEGLU block",
    0x1a0: "LDS-DMAs in front of the GEGLU block + scalar GELU",
    0x122: "ablation: no GEGLU arithmetic", 0x124: "ablation: no LDS-DMA", 0x126: "ablation: no GEGLU arithmetic, no LDS-DMA",
    0x128: "ablation: no GEMM 2 MFMAs", 0x12e: "ablation: GEMM 1 skeleton only",
}
if len(sys.argv) > 1 and sys.argv[1] == "qkv":
    # one timed batch of the activation-stationary QKV projection at the benchmarked size (PMC driver: tools/r6_pmc_stat_kernels.sh)
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from tango_amd import _lib
    lib = _lib.load()
    B, S, Ch = 64, 4096, 320
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(B * S, Ch, generator=g) * 1.2 + 0.3).cuda()
    w = (torch.randn(3 * Ch, Ch, generator=g) / Ch ** 0.5).cuda()
    ga, be = (1 + 0.2 * torch.randn(Ch, generator=g)).cuda(), (0.3 * torch.randn(Ch, generator=g)).cuda()
    qk = torch.zeros(B * S, 2 * Ch, device="cuda")
    vt = torch.zeros(B, Ch, S, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    ms = C.c_float(0.0)
    rc = lib.tango_op_qkv_stat(1, p(x), p(w), p(ga), p(be), p(qk), p(vt), B, S, Ch, C.c_float(1e-5), int(sys.argv[2]) if len(sys.argv) > 2 else 0, 10, C.byref(ms), None)
    assert rc == 0, lib.tango_last_error().decode()
    print("qkv M=%d mode %s: %.3f ms" % (B * S, sys.argv[2] if len(sys.argv) > 2 else "0", ms.value))
elif len(sys.argv) > 1 and sys.argv[1] == "one":
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from tango_amd import _lib
    lib = _lib.load()
    M, Cc, H = 262144, 320, 1280
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(M, Cc, generator=g) * 1.3 + 0.4).half().float().cuda()
    w1 = (torch.randn(2 * H, Cc, generator=g) / Cc ** 0.5).cuda()
    b1 = (0.3 * torch.randn(2 * H, generator=g)).cuda()
    w2 = (torch.randn(Cc, H, generator=g) / H ** 0.5).cuda()
    b2 = (0.3 * torch.randn(Cc, generator=g)).cuda()
    ga, be = (1 + 0.2 * torch.randn(Cc, generator=g)).cuda(), (0.3 * torch.randn(Cc, generator=g)).cuda()
    out = torch.zeros(M, Cc, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    mode = int(sys.argv[2])
    best = []
    for rnd in range(4):
        ms = C.c_float(0.0)
        rc = lib.tango_op_ff_fused(1, p(x), p(w1), p(b1), p(ga), p(be), p(w2), p(b2), p(out), M, Cc, H, C.c_float(1e-5), mode, 20, C.byref(ms), None)
        assert rc == 0, lib.tango_last_error().decode()
        best.append(ms.value)
    var = int(os.environ.get("TANGO_FF_VAR", "-1"), 0)
    what = "two GEMMs (TANGO_FF_FUSED=0 route)" if mode == 1 else "fused, %s" % ("compiler-scheduled loop" if os.environ.get("TANGO_FF_FUSED") == "2" else NAMES[var])
    print("%-75s %.3f ms  checksum %.6e" % (what, sorted(best)[1], out.double().abs().sum().item()))
else:
    runs = [(1, -1, 1), (0, -1, 1), (0, -1, 2), (0, 0x000, 1), (0, 0x100, 1), (0, 0x320, 1), (0, 0x520, 1), (0, 0x720, 1), (0, 0x140, 1), (0, 0x180, 1), (0, 0x1a0, 1),
            (0, 0x122, 1), (0, 0x124, 1), (0, 0x126, 1), (0, 0x128, 1), (0, 0x12e, 1), (0, -1, 1), (1, -1, 1)]
    for mode, var, fused in runs:
        env = dict(os.environ, TANGO_FF_FUSED=str(fused))
        env.pop("TANGO_FF_VAR", None)
        if var >= 0:
            env["TANGO_FF_VAR"] = hex(var)
        subprocess.run([sys.executable, __file__, "one", str(mode)], env=env, check=True)
