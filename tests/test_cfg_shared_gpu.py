"""Round 5: how `denoise` organises a guidance step, through the C ABI on the tiny UNet config.

  * CFG-shared prefix (engine.hip build_unet / Builder::transformer `shared`): the reference feeds `torch.cat([latents] * 2)` to the UNet
    (models.py:233) and nothing before the first cross-attention looks at the text, so conv_in, the first ResnetBlock2D and the first
    transformer up to attn2 (unet_2d_condition.py:640-660, unet_2d_blocks.py:843-867, attention.py:296-323) run ONCE for both halves.
    Exact: checked against the oracle, against the unshared plan (TANGO_NO_CFG_SHARED=1) and for the work it removes.
  * k denoise steps per captured hipGraph (TANGO_GRAPH_STEPS) and the two-chain variant (TANGO_UNET_CHAINS=2): measured and not the
    default (profiles/r5_c1_graph_steps_ab.txt, r5_c2_dual_chain_ab.txt); these cases keep both paths correct."""
import contextlib
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tango_oracle as O  # noqa: E402  (checker only)
from tango_amd import weights as W  # noqa: E402
from tango_amd.engine import Engine  # noqa: E402
from tango_amd.scheduler import SD21_SCHEDULER_CONFIG, DDPMScheduler  # noqa: E402

_KEYS = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample", "variance_type")
B, L, N = 3, 9, 5


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


def _setup():
    cfg = O.UNET_CONFIG_TINY
    g = torch.Generator().manual_seed(515)
    d = cfg["cross_attention_dim"]
    enc = torch.randn(2 * B, L, d, generator=g)
    mask = torch.ones(2 * B, L, dtype=torch.bool)
    mask[:B, 1:] = False                 # every unconditional row is T5(""): one valid token (models.py:282-289)
    mask[B + 1, 5:] = False              # a ragged conditional prompt
    lat0 = torch.randn(B, 8, 256, 16, generator=g)
    noises = torch.randn(N, B, 8, 256, 16, generator=g)
    sch = DDPMScheduler.from_config({k: SD21_SCHEDULER_CONFIG[k] for k in _KEYS})
    sch.set_timesteps(N)
    return cfg, enc, mask, lat0, noises, sch


def _run(e, enc, mask, lat0, noises, sch, use_graph=True):
    lat = lat0.clone().cuda()
    e.denoise(lat, enc.cuda(), mask, sch.timesteps.numpy(), sch.coef_table(), 3.0, noise=noises.cuda(), use_graph=use_graph)
    torch.cuda.synchronize()
    return lat.cpu(), e.last_step_gflop()


@pytest.mark.parametrize("dtype,tol_oracle,tol_ab", [("fp32", 1e-4, 2e-5), ("fp16", 5e-2, 2e-2)])
def test_cfg_shared_prefix_is_exact_and_removes_work(lib, dtype, tol_oracle, tol_ab):
    cfg, enc, mask, lat0, noises, sch = _setup()
    sd = W.synth_state_dict(W.unet_param_shapes(cfg, "unet."), 1234)
    with torch.no_grad():
        ref = O.denoise_loop(sd, cfg, O.DDPMOracle(**O.SD21_SCHEDULER), enc, mask, lat0.clone(), N, 3.0, noises=list(noises), prefix="unet.")
    e = Engine(unet=cfg, dtype=dtype)
    e.load_synthetic(1234)
    shared, gf_shared = _run(e, enc, mask, lat0, noises, sch)
    eager, _ = _run(e, enc, mask, lat0, noises, sch, use_graph=False)
    with tuning(lib, TANGO_NO_CFG_SHARED=1):
        plain, gf_plain = _run(e, enc, mask, lat0, noises, sch)
    err = (shared - ref).abs().max().item()
    dab = (shared - plain).abs().max().item()
    print("CFG-shared prefix, tiny UNet %s, B=%d, %d steps: vs oracle %.3e, vs the unshared plan %.3e; executed GFLOP per step %.3f vs %.3f"
          % (dtype, B, N, err, dab, gf_shared, gf_plain))
    assert torch.equal(shared, eager), "hipGraph replay and eager launches must agree bit for bit"
    assert err <= tol_oracle and dab <= tol_ab
    assert gf_shared < gf_plain * 0.995, "the shared plan must execute less work"
    # a batch whose first half is NOT single-key keeps the plain plan (same executed work as with the switch off)
    mask2 = mask.clone()
    mask2[0, 1:3] = True
    _, gf2 = _run(e, enc, mask2, lat0, noises, sch)
    with tuning(lib, TANGO_NO_CFG_SHARED=1):
        _, gf2_plain = _run(e, enc, mask2, lat0, noises, sch)
    assert abs(gf2 - gf2_plain) < 1e-9
    del e


def test_k_steps_per_graph_and_two_chains_match_the_default(lib):
    cfg, enc, mask, lat0, noises, sch = _setup()
    e = Engine(unet=cfg, dtype="fp32")
    e.load_synthetic(1234)
    base, _ = _run(e, enc, mask, lat0, noises, sch)
    with tuning(lib, TANGO_GRAPH_STEPS=2):          # 5 steps = two 2-step replays + one 1-step replay
        k2, _ = _run(e, enc, mask, lat0, noises, sch)
    assert torch.equal(k2, base), "k steps per graph: same kernels in the same order"
    enc4, mask4 = torch.cat([enc[:2], enc[B:B + 2]]), torch.cat([mask[:2], mask[B:B + 2]])      # an even batch for the two chains
    lat4, n4 = lat0[:2].contiguous(), noises[:, :2].contiguous()
    base4, _ = _run(e, enc4, mask4, lat4, n4, sch)
    with tuning(lib, TANGO_UNET_CHAINS=2):
        two, _ = _run(e, enc4, mask4, lat4, n4, sch)
        two_eager, _ = _run(e, enc4, mask4, lat4, n4, sch, use_graph=False)
    d = (two - base4).abs().max().item()
    print("two chains vs one (tiny fp32, 2 prompts, %d steps): max abs diff %.3e" % (N, d))
    assert torch.equal(two, two_eager) and d <= 2e-5
    del e
