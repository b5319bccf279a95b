"""Host-side logic that needs no GPU: parameter inventories, synthetic weights, sharding, C-ABI exports,
error behaviour mirrored from the reference."""
import os
import re

import numpy as np
import pytest
import torch

from tango_amd import _lib, weights as W
from tango_amd.engine import HIFIGAN_CONFIG, UNET_CONFIG_LARGE, UNET_CONFIG_XL, VAE_CONFIG
from tango_amd.parallel import shard_bounds, shard_cfg_embeddings
from tango_amd.scheduler import SD21_SCHEDULER_CONFIG, DDIMScheduler, DDPMScheduler

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_param_inventory_counts():
    """686 UNet tensors, 865 933 768 UNet parameters; 308 decode-side VAE+vocoder tensors"""
    u = W.unet_param_shapes(UNET_CONFIG_LARGE)
    assert len(u) == 686
    assert sum(int(np.prod(s)) for s in u.values()) == 865933768
    v = W.vae_decoder_param_shapes(VAE_CONFIG)
    h = W.hifigan_param_shapes(HIFIGAN_CONFIG)
    assert len(v) + len(h) == 308
    assert sum(int(np.prod(s)) for s in h.values()) == 55264897
    assert u["up_blocks.1.resnets.2.conv1.weight"] == (1280, 1920, 3, 3)
    assert W.unet_param_shapes(UNET_CONFIG_XL)["mid_block.attentions.0.transformer_blocks.0.attn2.to_k.weight"] == (1280, 2048)
    assert h["vocoder.ups.0.weight"] == (1024, 512, 16)


def test_synth_weights_deterministic_and_independent():
    a = W.synth_tensor("unet.conv_in.weight", (320, 8, 3, 3), 1234)
    b = W.synth_tensor("unet.conv_in.weight", (320, 8, 3, 3), 1234)
    c = W.synth_tensor("unet.conv_in.weight", (320, 8, 3, 3), 1235)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(a.abs().max().item() - 1 / 72 ** 0.5) < 1e-3
    g = W.synth_tensor("decoder.norm_out.weight", (128,), 1)
    assert abs(g.mean().item() - 1.0) < 0.1


def test_library_exports_every_declared_symbol():
    """the C ABI library loads without a GPU and exports every function include/tango_engine.h declares"""
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "tango_engine.h")).read()
    declared = set(re.findall(r"\b(tango_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"tango_engine_t", "tango_config_t", "tango_denoise_args_t"}
    assert declared, "header parse"
    for s in declared:
        assert hasattr(lib, s), "libtango_hip.so does not export %s" % s
    assert set(_lib.SYMBOLS) == declared
    assert lib.tango_version().decode().startswith("tango-mi355x")


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_engine_fails_loudly_without_gpu():
    """no CPU fallback: creating an engine without a HIP device raises"""
    from tango_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine(unet=UNET_CONFIG_LARGE)
    import ctypes as C
    lib = _lib.load()
    cfg = _lib.TangoConfig()
    h = C.c_void_p()
    assert lib.tango_engine_create(C.byref(cfg), C.byref(h)) != 0
    assert b"no HIP device" in lib.tango_last_error() or b"fail" in lib.tango_last_error().lower()


def test_product_tree_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tango_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'"""[\s\S]*?"""', "", src), "%s references oracle/" % f


def test_shard_bounds_cover_and_order():
    for n in (1, 7, 32, 33, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_cfg_keeps_uncond_twin():
    B, L, d = 5, 3, 4
    pe = torch.arange(2 * B).float()[:, None, None].expand(2 * B, L, d).contiguous()
    pm = torch.ones(2 * B, L, dtype=torch.bool)
    seen = []
    for r in range(2):
        e, m, lo = shard_cfg_embeddings(pe, pm, 2, r, True)
        b = e.shape[0] // 2
        assert e[:b, 0, 0].tolist() == [float(lo + i) for i in range(b)]            # uncond rows
        assert e[b:, 0, 0].tolist() == [float(B + lo + i) for i in range(b)]        # matching cond rows
        seen += list(range(lo, lo + b))
    assert seen == list(range(B))


def test_scheduler_surface_and_errors():
    keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample", "variance_type")
    s = DDPMScheduler.from_config({k: SD21_SCHEDULER_CONFIG[k] for k in keys})
    assert s.order == 1 and s.init_noise_sigma == 1.0 and s.config.prediction_type == "v_prediction"
    x = torch.ones(2)
    assert s.scale_model_input(x, 5) is x
    s.set_timesteps(200)
    assert s.timesteps.tolist() == list(range(995, -1, -5))
    with pytest.raises(ValueError):              # scheduling_ddpm.py:193-198
        s.set_timesteps(1001)
    with pytest.raises(ValueError):              # scheduling_ddpm.py:305-309
        DDPMScheduler(prediction_type="bogus")
    t = s.coef_table()
    assert t.shape == (200, 8) and t.dtype == np.float32
    assert t[-1, 2] == 1.0 and t[-1, 3] == 0.0 and t[-1, 4] == 0.0      # last step: abar_prev = 1, no noise
    d = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                      prediction_type="v_prediction", clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    d.set_timesteps(200)
    assert d.timesteps.tolist() == list(range(996, 0, -5)) and d.rule == "ddim"
    t0 = d.coef_table()
    assert (t0[:, 4] == 0).all()                                        # eta = 0: no noise term
    # eta > 0 (scheduling_ddim.py:316-352): sigma = eta * sqrt(var), direction^2 + sigma^2 = 1 - abar_prev
    from oracle import tango_oracle as O
    d1 = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                       prediction_type="v_prediction", clip_sample=False, set_alpha_to_one=False, steps_offset=1, eta=0.7)
    d1.set_timesteps(200)
    t1 = d1.coef_table()
    o = O.DDIMOracle(**dict(O.SD21_SCHEDULER, set_alpha_to_one=False, steps_offset=1))
    o.set_timesteps(200)
    for i in (0, 57, 199):
        t = int(o.timesteps[i]); pt = t - 5
        var = float(o._variance(t, pt))
        assert abs(t1[i, 4] - 0.7 * var ** 0.5) < 1e-6
        a_prev = float(o.alphas_cumprod[pt] if pt >= 0 else o.final_alpha_cumprod)
        assert abs(t1[i, 6] ** 2 + t1[i, 4] ** 2 - (1 - a_prev)) < 1e-6 and t1[i, 5] == t0[i, 5]
    with pytest.raises(ValueError):
        DDIMScheduler(eta=-1.0)


def test_batch_inference_driver_writes_reference_layout(tmp_path):
    """generation + save half of inference.py:127-176 with a test double for the generator (no GPU)."""
    import json
    import wave

    import numpy as np

    from tango_amd import batch_inference as BI

    tf = tmp_path / "prompts.json"
    tf.write_text("\n".join(json.dumps({"captions": c, "id": i}) for i, c in enumerate(["a dog barks", "rain", "a car passes"])) + "\n")
    prompts = BI.read_prompts(str(tf), "captions", prefix="gen: ")
    assert prompts == ["gen: a dog barks", "gen: rain", "gen: a car passes"]

    class Fake:
        def __init__(self):
            self.calls = []

        def generate_for_batch(self, prompts, steps, guidance, samples, batch_size):
            self.calls.append((len(prompts), steps, guidance, samples, batch_size))
            waves = [np.full(1600, 100 * j + k, np.int16) for j in range(len(prompts)) for k in range(samples)]
            return waves if samples == 1 else [waves[i:i + samples] for i in range(0, len(waves), samples)]

    g = Fake()
    rec = BI.generate_and_save(g, prompts, num_steps=10, guidance=3, batch_size=2, num_samples=1, out_root=str(tmp_path / "outputs"), exp_id="7")
    assert g.calls == [(3, 10, 3, 1, 2)] and rec["Test Instances"] == 3 and rec["Steps"] == 10
    for j in range(3):
        with wave.open(str(tmp_path / "outputs" / "7_steps_10_guidance_3" / ("output_%d.wav" % j)), "rb") as w:
            assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 16000, 1600)
            assert np.frombuffer(w.readframes(1600), "<i2")[0] == 100 * j
    rec2 = BI.generate_and_save(g, prompts, num_steps=5, guidance=2.5, batch_size=8, num_samples=2, out_root=str(tmp_path / "outputs"), exp_id="8")
    for i in (1, 2):
        with wave.open(str(tmp_path / "outputs" / "8_steps_5_guidance_2.5" / ("rank_%d" % i) / "output_2.wav"), "rb") as w:
            assert np.frombuffer(w.readframes(4), "<i2")[0] == 200 + (i - 1)
    lines = [l for l in (tmp_path / "outputs" / "summary.jsonl").read_text().split("\n") if l.strip()]
    assert len(lines) == 2 and json.loads(lines[1])["Samples Per Prompt"] == 2 and abs(rec2["audio_seconds"] - 0.6) < 1e-9
    import pytest
    with pytest.raises(ValueError):
        BI.write_wav(str(tmp_path / "x.wav"), np.zeros(4, np.float32))
    assert BI.generate_and_save(g, [], out_root=str(tmp_path / "o2"), exp_id="9")["Test Instances"] == 0


def test_predictor_serving_surface(tmp_path):
    """cog Predictor (predict.py:29-67): setup() builds one generator per model directory, predict() writes a 16 kHz wav"""
    import wave

    import numpy as np

    from tango_amd.predict import Predictor

    calls = []

    class FakeTango:
        def __init__(self, path, device="cuda:0", dtype="fp16", text_encoder=None, tokenizer=None):
            self.path = path

        def generate(self, prompt, steps=100, guidance=3):
            calls.append((self.path, prompt, steps, guidance))
            return (np.arange(1600) % 100).astype(np.int16)

    (tmp_path / "tango2").mkdir()
    p = Predictor()
    with pytest.raises(FileNotFoundError):
        p.setup(model_cache=str(tmp_path), tango_cls=FakeTango)           # tango2-full missing: no silent download
    p.setup(model_cache=str(tmp_path), names=["tango2"], tango_cls=FakeTango)
    out = p.predict("a dog barks", "tango2", 7, 2.5, out=str(tmp_path / "o.wav"))
    assert calls == [(str(tmp_path / "tango2"), "a dog barks", 7, 2.5)]
    with wave.open(str(out)) as w:
        assert w.getframerate() == 16000 and w.getnframes() == 1600 and w.getsampwidth() == 2
    with pytest.raises(KeyError):
        p.predict("x", "tango2-full", 1, 1.0)


def test_build_pretrained_models_from_audioldm_ckpt():
    """models.py:27-52: `first_stage_model.*` keys and `scale_factor` of an AudioLDM checkpoint feed the VAE"""
    import torch

    from tango_amd.models import build_pretrained_models

    seen = {}

    class FakeVAE:
        def __init__(self, **kw):
            seen["kw"] = kw

        def load_state_dict(self, sd):
            seen["sd"] = sd

        def eval(self):
            return self

    ck = {"state_dict": {"scale_factor": torch.tensor(0.9227914214134216), "first_stage_model.decoder.conv_in.weight": torch.zeros(2),
                         "first_stage_model.vocoder.conv_pre.bias": torch.ones(3), "model.diffusion_model.x": torch.zeros(1)}}
    class FakeSTFT:
        def __init__(self, *a, **kw):
            seen["stft"] = (a, kw)

        def eval(self):
            return self

    vae, stft = build_pretrained_models(ck, autoencoder_cls=FakeVAE, stft_cls=FakeSTFT)
    assert isinstance(stft, FakeSTFT) and isinstance(vae, FakeVAE)
    assert seen["stft"][0] == (1024, 160, 1024, 64, 16000, 0, 8000)       # audioldm/utils.py:104-118 via models.py:40-47
    assert abs(seen["kw"]["scale_factor"] - 0.9227914214134216) < 1e-7 and seen["kw"]["ddconfig"]["ch_mult"] == [1, 2, 4]
    assert sorted(seen["sd"]) == ["decoder.conv_in.weight", "vocoder.conv_pre.bias"]


def test_swizzle64_conflict_free():
    """the 64-byte-row LDS swizzles of gemm_wide.hip / conv_wide.hip are conflict-free for ds_read_b128's lane groups
    (tools/check_swizzle64.py asserts it exhaustively)"""
    import runpy
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_swizzle64.py"))


def test_dp_initial_latents_are_shard_invariant():
    """the DP entry point draws each sample's initial latents from (pass seed, GLOBAL sample index) -- a
    shard reproduces its slice of the unsharded draw, samples differ from each other and from another seed's."""
    from tango_amd.tango import dp_initial_latents
    full = dp_initial_latents(4242, 0, 5)
    parts = torch.cat([dp_initial_latents(4242, 0, 2), dp_initial_latents(4242, 2, 3)])
    assert full.shape == (5, 8, 256, 16) and torch.equal(full, parts)
    assert not torch.equal(full[0], full[1]) and not torch.equal(full, dp_initial_latents(4243, 0, 5))
    assert abs(full.std().item() - 1.0) < 0.02 and dp_initial_latents(1, 0, 0).shape == (0, 8, 256, 16)


def test_ensure_text_loads_only_the_missing_component(monkeypatch):
    """with an encoder already present (user-supplied, or built from the checkpoint by text_encoder='engine')
    and no tokenizer, _ensure_text() must fetch ONLY the tokenizer -- never replace the encoder with stock hub weights."""
    from tango_amd import models as M
    from tango_amd.models import AudioDiffusion

    loaded = []

    class Tok:
        @staticmethod
        def from_pretrained(name):
            loaded.append(("tok", name))
            return "tokenizer"

    class Enc:
        @staticmethod
        def from_pretrained(name):
            loaded.append(("enc", name))
            raise AssertionError("the encoder must not be reloaded")

    monkeypatch.setattr(M, "_hf_classes", lambda: (Tok, Enc))
    m = AudioDiffusion.__new__(AudioDiffusion)          # no engine: only the text plumbing is exercised
    mine = object()
    m.text_encoder, m.tokenizer, m.text_encoder_name, m._text_sd, m.device = mine, None, "google/flan-t5-large", None, "cpu"
    m._ensure_text()
    assert m.text_encoder is mine and m.tokenizer == "tokenizer" and loaded == [("tok", "google/flan-t5-large")]
    m.text_encoder = "engine"                            # not built yet: a clear error, no hub access
    with pytest.raises(RuntimeError):
        m._ensure_text()


def test_predictor_gives_each_model_its_own_text_encoder(tmp_path):
    """every Tango loads its checkpoint's text_encoder.* into the module it receives -- a shared module would
    keep only the last model's weights."""
    import torch.nn as nn

    from tango_amd.predict import Predictor

    got = []

    class FakeTango:
        def __init__(self, path, device="cuda:0", dtype="fp16", text_encoder=None, tokenizer=None):
            got.append(text_encoder)

    for k in ("tango2", "tango2-full"):
        (tmp_path / k).mkdir()
    enc = nn.Linear(2, 2)
    Predictor().setup(model_cache=str(tmp_path), tango_cls=FakeTango, text_encoder=enc)
    assert len(got) == 2 and got[0] is not got[1] and got[0] is not enc and torch.equal(got[0].weight, enc.weight)
    got.clear()
    Predictor().setup(model_cache=str(tmp_path), tango_cls=FakeTango, text_encoder="engine")
    assert got == ["engine", "engine"]


def test_isa_scan_finds_no_streaming_miscompare_pattern():
    """VERDICT r2 next #6 / ADVICE r2: no shipped kernel contains the `ds_read -> v_xor 0x80000000 -> v_pk_fma_f32` data flow
    of the round-1/2 streaming-kernel miscompare (tools/isa_scan.py disassembles every gfx950 code object of the library; the
    folded-LayerNorm epilogues are the strict set), and the experiment variants are no longer compiled in."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tango_amd", "lib", "libtango_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_scan.py"), "--lib", lib, "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    tot = [l for l in r.stdout.splitlines() if l.startswith("# totals")][0]
    assert "xor_fed 0," in tot and "both 0" in tot and "mfma_raw 0" in tot, tot
    ln = [l for l in r.stdout.splitlines() if "[LN]" in l]
    assert len(ln) >= 10, "folded-LayerNorm kernels must be recognised by name (%d found)" % len(ln)
    names = r.stdout
    assert "gemm_pers_kernel" not in names and "Lb1ELb0EEEvNS_10GemmParamsE" not in "".join(l for l in names.splitlines() if "lin_stream" in l)
    # ablation / lock-step variants of the 256 x 160 kernels are not instantiated any more
    assert not [l for l in names.splitlines() if "conv3x3_halo_kernel" in l and ("Lb1ELb0E" in l or "Lb0ELb1E" in l)]


def test_isa_scan_mfma_read_hazard_detector():
    """round 3: hipcc does not pad MFMA -> VALU read hazards for inline-asm operands (the first fused cross-attention kernel read fresh
    accumulators one slot after the MFMA and was not repeatable).  The scanner counts wait states between a v_mfma and the first
    non-MFMA reader of its destination; the shipped form (s_nop padding) must pass, the failing form must be flagged."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import isa_scan as I

    def prog(*lines):
        return [I.parse_inst("\t" + l) for l in lines]

    mfma = "v_mfma_f32_16x16x32_f16 v[122:125], v[186:189], v[42:45], v[122:125]"
    bad = prog(mfma, "v_add_u32_e32 v1, v2, v3", "v_fma_f32 v218, -v140, v218, v122")
    assert I.scan_mfma_raw(bad)[0] == 1
    padded = prog(mfma, "s_nop 15", "s_nop 7", "v_fma_f32 v218, -v140, v218, v122")
    assert I.scan_mfma_raw(padded)[0] == 0
    chained = prog(mfma, "v_mfma_f32_16x16x32_f16 v[122:125], v[190:193], v[46:49], v[122:125]", "s_nop 6", "v_cvt_pk_f16_f32 v5, v122, v123")
    assert I.scan_mfma_raw(chained)[0] == 0           # the accumulate chain is the matrix pipe's own dependency; the last MFMA is 7 away
    f32 = prog("v_mfma_f32_16x16x4_f32 v[116:119], v154, v102, v[116:119]", "s_nop 6", "v_mul_f32_e32 v9, v116, v8")
    assert I.scan_mfma_raw(f32)[0] == 1               # the 8-pass fp32 MFMA needs more than the 4-pass 16-bit one
    stored = prog(mfma, "global_store_dwordx4 v[10:11], v[122:125], off")
    assert I.scan_mfma_raw(stored)[0] == 1            # memory instructions read registers too


def test_bench_labels_and_traffic_gate(tmp_path, monkeypatch):
    """bench.py must say what RAN: the workload label follows the flags, and `roofline.traffic` only comes from
    a PMC record taken on the kernel sources being benchmarked"""
    import argparse
    import json

    import bench

    def ns(**kw):
        d = dict(guidance=3.0, text_len=64, xl=False, fp8_attn=False, batch=32, denoise_steps=200, dtype="fp16")
        d.update(kw)
        return argparse.Namespace(**d)

    assert bench.workload_name(ns()).startswith("BASELINE config 3")
    assert bench.workload_name(ns(batch=1, denoise_steps=100)).startswith("BASELINE config 2")
    assert "config 5" in bench.workload_name(ns(xl=True, fp8_attn=True, dtype="bf16", batch=8))
    for other in (ns(batch=8), ns(denoise_steps=20), ns(guidance=2.0), ns(text_len=128), ns(fp8_attn=True)):
        assert bench.workload_name(other).startswith("custom"), bench.workload_name(other)
    # traffic: a record is used only when batch / dtype / flags AND the source hash match
    sha = bench.kernel_source_sha16()
    assert len(sha) == 16
    prof = tmp_path / "profiles"
    prof.mkdir()
    rec = {"records": [{"batch": 32, "dtype": "fp16", "xl": False, "fp8_attn": False, "src_sha16": "0" * 16, "bytes_per_step": 1.0, "source": "stale"},
                       {"batch": 32, "dtype": "fp16", "xl": False, "fp8_attn": False, "src_sha16": sha, "bytes_per_step": 2.0, "source": "fresh"}]}
    (prof / "hbm_traffic.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_sha16", lambda: sha)
    assert bench.hbm_traffic(32, "fp16", False, False) == (2.0, "fresh")
    assert bench.hbm_traffic(8, "fp16", False, False) == (None, None)
    assert bench.hbm_traffic(32, "bf16", False, False) == (None, None)
    assert bench.hbm_traffic(32, "fp16", False, True) == (None, None)
    monkeypatch.setattr(bench, "kernel_source_sha16", lambda: "f" * 16)
    assert bench.hbm_traffic(32, "fp16", False, False) == (None, None)


def test_committed_traffic_record_matches_the_committed_kernel_sources():
    """the PMC record bench.py reports must have been taken on the sources in the tree (it is re-measured whenever csrc/ changes)"""
    import bench

    t, src = bench.hbm_traffic(32, "fp16", False, False)
    if t is None:          # mid-round state after a kernel edit: bench.py reports traffic = null until tools/final_profiles.sh is re-run
        pytest.skip("profiles/hbm_traffic.json has no record for kernel sources %s (re-measure before the round ends)" % bench.kernel_source_sha16())
    assert os.path.exists(os.path.join(ROOT, src)) and 5e10 < t < 3e11


def test_hot_kernels_do_not_spill():
    """Round 5: the level-0 GEGLU projection's streaming kernel carried 21 spilled VGPRs (scratch reloads share the in-order vmcnt queue
    with its operand prefetch) until its epilogue was specialised (`SPEC = 1`: 3.445 -> 2.927 ms per step, profiles/r5_c5_*).  Spills are
    invisible in the source: pin the kernels of the config-3 step -- 256 x 320 conv / GEMM (one-shot and persistent forms that run),
    256 x 160 duo, attention, fused cross-attention, the specialised streaming kernel -- at zero scratch (tools/kernel_resources.py
    reads the code objects' metadata; no GPU needed)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tango_amd", "lib", "libtango_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    sys.path.insert(0, os.path.join(root, "tools"))
    import kernel_resources as KR
    # register allocation is a property of the TOOLCHAIN as much as of the source: the zero-scratch pin holds for the ROCm release the
    # kernels were tuned on; elsewhere the test reports instead of failing the CPU suite
    try:
        rocm = open("/opt/rocm/.info/version").read().strip()
    except OSError:
        rocm = "unknown"
    if not rocm.startswith("7.2"):
        pytest.skip("zero-scratch pin is for ROCm 7.2's register allocator; this is ROCm %s" % rocm)
    res = KR.kernel_resources(lib)
    assert len(res) > 200
    # kernels are matched by their DEMANGLED name and template arguments (KR.template_args), not by mangled-suffix substrings
    parsed = {k: KR.template_args(k) for k in res}
    HOT = ("conv3x3_wide_kernel", "conv3x3_wide_pipe_kernel", "gemm_wide_pers_kernel", "attn_kernel", "xattn_block_kernel", "conv3x3_halo_kernel", "gemm_dma_kernel",
           "gn_apply_kernel", "gn_stats_kernel")
    hot = [k for k, (name, args) in parsed.items()
           if name in HOT or (name == "lin_stream_kernel" and args and len(args) == 6 and args[5] == 1)]     # <T, KS, TN, LN, FIX, SPEC = 1>
    if len(hot) <= 40 or not any(parsed[k][0] == "lin_stream_kernel" for k in hot):
        pytest.skip("kernel template signatures changed: only %d hot kernels recognised -- update the matcher" % len(hot))
    bad = {k: v for k, v in res.items() if k in hot and (v[1] > 0 or v[2] > 0)}
    assert not bad, "kernels of the hot path with scratch: %s" % bad
    # the fused level-0 feed-forward (round 6): the shipped variant (asm LDS stream: VAR bit 0 clear, no ablation bits) sits at 244 of 256 registers
    ff = [k for k, (name, args) in parsed.items() if name == "ff_fused_kernel" and args and isinstance(args[1], int) and (args[1] & 0xf) == 0]
    assert len(ff) >= 2 and all(res[k][1] == 0 and res[k][2] == 0 for k in ff), {k: res[k] for k in ff}
    # and the one-shot 256 x 320 / 256 x 160 GEMM variants that run in the step (everything but GEGLU + residual + folded LayerNorm, which no plan uses)
    for k, (vg, sp, sc) in res.items():
        name, args = parsed[k]
        if name in ("gemm_wide_kernel", "gemm_duo_kernel") and args and args[1:4] != [True, True, True]:
            assert sc == 0, (k, vg, sp, sc)
