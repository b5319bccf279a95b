"""In-wave software pipeline of the self-attention kernel (round 6: attn_pipe_kernel -- S^T of tile t + 1 is multiplied in front of the
softmax of tile t; three K stages, two waves per SIMD).  Same arithmetic in the same order per accumulator as the one-tile-at-a-time
kernel with matrix-pipe row sums: the two must agree to the last bit, on every repetition, for even and odd tile counts."""
import ctypes as C

import pytest
import torch

from test_duo_gpu import tuning

pytestmark = pytest.mark.gpu

DT = {"fp16": 1, "bf16": 2}


def _run(lib, dtype, q, k, v, B, heads, S, pipe):
    out = torch.empty(B, S, heads * 64, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    with tuning(lib, TANGO_ATTN_PIPE=pipe, TANGO_ATTN_DEFER=0, TANGO_ATTN_MSUM=1):
        rc = lib.tango_op_attention(DT[dtype], p(q), p(k), p(v), None, p(out), B, heads, S, S, C.c_float(0.125), None)
    assert rc == 0, lib.tango_last_error().decode()
    return out.cpu()


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("B,heads,S", [(2, 5, 4096), (3, 10, 1024), (2, 20, 256), (5, 3, 128), (2, 2, 192 * 2), (1, 1, 64 * 7 * 2)])
def test_attention_pipe_bit_equal(lib, dtype, B, heads, S):
    g = torch.Generator().manual_seed(S + heads)
    C_ = heads * 64
    q, k, v = (torch.randn(B, S, C_, generator=g).cuda() for _ in range(3))
    ref = _run(lib, dtype, q, k, v, B, heads, S, 0)
    for rep in range(6):
        out = _run(lib, dtype, q, k, v, B, heads, S, 1)
        assert torch.equal(out, ref), "repetition %d: %d elements differ from the unpipelined kernel" % (rep, (out != ref).sum().item())
    qh, kh, vh = (t.cpu().view(B, S, heads, 64).transpose(1, 2).double() for t in (q, k, v))
    exact = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).transpose(1, 2).reshape(B, S, C_).float()
    err = ((out - exact).abs().max() / exact.abs().max()).item()
    assert err <= (4e-3 if dtype == "fp16" else 3e-2), err
