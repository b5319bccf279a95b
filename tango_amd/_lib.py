"""ctypes binding of libtango_hip.so (the C ABI declared in include/tango_engine.h).

There is no CPU fallback: if the shared library is missing or no HIP device is visible, every
entry point raises.  `python -c "import __graft_entry__ as g; g.build()"` (or
`make -C tango_amd/csrc`) builds the library in-tree at tango_amd/lib/libtango_hip.so.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtango_hip.so")
MAX_LEVELS = 8

DTYPES = {"fp32": 0, "float32": 0, "f32": 0, "fp16": 1, "float16": 1, "f16": 1, "bf16": 2, "bfloat16": 2}
PRED = {"epsilon": 0, "sample": 1, "v_prediction": 2}
RULE = {"ddpm": 0, "ddim": 1}


class TangoConfig(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("unet_levels", C.c_int32),
        ("unet_channels", C.c_int32 * MAX_LEVELS),
        ("unet_heads", C.c_int32 * MAX_LEVELS),
        ("unet_cross_attn", C.c_int32 * MAX_LEVELS),
        ("unet_layers_per_block", C.c_int32),
        ("unet_in_channels", C.c_int32),
        ("unet_out_channels", C.c_int32),
        ("unet_cross_dim", C.c_int32),
        ("unet_groups", C.c_int32),
        ("unet_eps", C.c_float),
        ("unet_flip_sin_to_cos", C.c_int32),
        ("unet_freq_shift", C.c_float),
        ("latent_h", C.c_int32),
        ("latent_w", C.c_int32),
        ("vae_levels", C.c_int32),
        ("vae_ch", C.c_int32),
        ("vae_ch_mult", C.c_int32 * MAX_LEVELS),
        ("vae_num_res_blocks", C.c_int32),
        ("vae_z_channels", C.c_int32),
        ("vae_embed_dim", C.c_int32),
        ("vae_out_ch", C.c_int32),
        ("vae_scale_factor", C.c_float),
        ("voc_n_ups", C.c_int32),
        ("voc_rates", C.c_int32 * MAX_LEVELS),
        ("voc_kernels", C.c_int32 * MAX_LEVELS),
        ("voc_initial_channel", C.c_int32),
        ("voc_num_mels", C.c_int32),
        ("voc_n_resblocks", C.c_int32),
        ("voc_res_kernels", C.c_int32 * 4),
        ("voc_res_dilations", (C.c_int32 * 4) * 4),
        ("t5_layers", C.c_int32),
        ("t5_d_model", C.c_int32),
        ("t5_d_kv", C.c_int32),
        ("t5_heads", C.c_int32),
        ("t5_d_ff", C.c_int32),
        ("t5_vocab", C.c_int32),
        ("t5_rel_buckets", C.c_int32),
        ("t5_rel_max_distance", C.c_int32),
        ("t5_eps", C.c_float),
        ("vae_encoder", C.c_int32),
        ("vae_in_channels", C.c_int32),
        ("stft_filter_length", C.c_int32),
        ("stft_hop_length", C.c_int32),
        ("stft_n_mel", C.c_int32),
        ("unet_music", C.c_int32),
        ("unet_attn_fp8", C.c_int32),
    ]


class DenoiseArgs(C.Structure):
    _fields_ = [
        ("latents", C.c_void_p),
        ("prompt_embeds", C.c_void_p),
        ("prompt_mask", C.c_void_p),
        ("batch", C.c_int32),
        ("text_len", C.c_int32),
        ("num_steps", C.c_int32),
        ("timesteps", C.c_void_p),
        ("coef", C.c_void_p),
        ("guidance_scale", C.c_float),
        ("prediction_type", C.c_int32),
        ("rule", C.c_int32),
        ("clip_sample", C.c_int32),
        ("clip_sample_range", C.c_float),
        ("noise", C.c_void_p),
        ("seed", C.c_uint64),
        ("sample_offset", C.c_int32),
        ("use_graph", C.c_int32),
        ("beat_embeds", C.c_void_p),
        ("beat_mask", C.c_void_p),
        ("beat_len", C.c_int32),
        ("chord_embeds", C.c_void_p),
        ("chord_mask", C.c_void_p),
        ("chord_len", C.c_int32),
        ("prompt_mask_host", C.c_void_p),
    ]


#: every symbol include/tango_engine.h declares (tests check the library exports all of them)
SYMBOLS = [
    "tango_last_error", "tango_version", "tango_tuning_reload", "tango_debug_linear_route", "tango_engine_create", "tango_engine_destroy",
    "tango_engine_num_weights", "tango_engine_weight_name", "tango_engine_set_weight",
    "tango_engine_finalize_weights", "tango_engine_denoise", "tango_engine_unet_forward", "tango_engine_unet_forward_music",
    "tango_engine_vae_decode", "tango_engine_vae_encode", "tango_engine_vocode", "tango_engine_vocoder_samples", "tango_engine_encode_text",
    "tango_engine_mel_frames", "tango_engine_mel_spectrogram",
    "tango_engine_last_denoise_ms", "tango_engine_last_step_gflop", "tango_engine_profile_unet", "tango_engine_profile_vae", "tango_engine_profile_vocoder", "tango_engine_set_plan_budget", "tango_engine_plan_stats", "tango_engine_drop_plans", "tango_op_conv2d", "tango_op_linear", "tango_op_linear_ln", "tango_op_ff_fused", "tango_op_qkv_stat", "tango_op_linear_qkv", "tango_op_linear_qkv_perm", "tango_op_conv1d",
    "tango_op_conv_transpose1d", "tango_op_groupnorm", "tango_op_layernorm", "tango_op_attention", "tango_op_attention_ex", "tango_op_xattn_block",
    "tango_op_sched_step", "tango_op_philox_normal",
]

_lib = None


def load():
    """Load the HIP library (raises if it has not been built: no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "tango_amd: %s not found. Build it with `make -C tango_amd/csrc` "
            "(or __graft_entry__.build()); the engine has no CPU/PyTorch fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, ci, cf, i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
    lib.tango_last_error.restype = C.c_char_p
    lib.tango_version.restype = C.c_char_p
    lib.tango_tuning_reload.restype = None
    lib.tango_engine_create.argtypes = [C.POINTER(TangoConfig), C.POINTER(vp)]
    lib.tango_engine_destroy.argtypes = [vp]
    lib.tango_engine_destroy.restype = None
    lib.tango_engine_num_weights.argtypes = [vp]
    lib.tango_engine_weight_name.argtypes = [vp, ci]
    lib.tango_engine_weight_name.restype = C.c_char_p
    lib.tango_engine_set_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(i64), ci]
    lib.tango_engine_finalize_weights.argtypes = [vp]
    lib.tango_engine_denoise.argtypes = [vp, C.POINTER(DenoiseArgs), vp]
    lib.tango_engine_unet_forward.argtypes = [vp, vp, i64, vp, vp, vp, ci, ci, vp]
    lib.tango_engine_unet_forward_music.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.tango_engine_vae_decode.argtypes = [vp, vp, vp, ci, vp]
    lib.tango_engine_vae_encode.argtypes = [vp, vp, vp, ci, vp]
    lib.tango_engine_vocode.argtypes = [vp, vp, vp, ci, ci, C.POINTER(ci), vp]
    lib.tango_engine_vocoder_samples.argtypes = [vp, ci]
    lib.tango_engine_encode_text.argtypes = [vp, vp, vp, vp, ci, ci, vp]
    lib.tango_engine_mel_frames.argtypes = [vp, ci]
    lib.tango_engine_mel_spectrogram.argtypes = [vp, vp, vp, vp, vp, ci, ci, C.POINTER(ci), vp]
    lib.tango_engine_last_denoise_ms.argtypes = [vp, C.POINTER(cf), C.POINTER(cf)]
    lib.tango_engine_last_step_gflop.argtypes = [vp, C.POINTER(C.c_double)]
    lib.tango_engine_profile_unet.argtypes = [vp, ci, ci, C.c_char_p, ci, vp]
    lib.tango_engine_profile_vae.argtypes = [vp, ci, C.c_char_p, ci, vp]
    lib.tango_engine_profile_vocoder.argtypes = [vp, ci, ci, C.c_char_p, ci, vp]
    lib.tango_engine_set_plan_budget.argtypes = [vp, C.c_uint64]
    lib.tango_engine_drop_plans.argtypes = [vp]
    lib.tango_debug_linear_route.argtypes = [ci] * 8
    lib.tango_debug_linear_route.restype = C.c_char_p
    lib.tango_engine_plan_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(ci)]
    lib.tango_op_conv2d.argtypes = [ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]
    lib.tango_op_linear.argtypes = [ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    lib.tango_op_linear_ln.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, vp]
    lib.tango_op_ff_fused.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, ci, ci, vp, vp]
    lib.tango_op_qkv_stat.argtypes = [ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, ci, ci, vp, vp]
    lib.tango_op_linear_qkv.argtypes = [ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, vp]
    lib.tango_op_linear_qkv_perm.argtypes = [ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, ci, vp]
    lib.tango_op_conv1d.argtypes = [ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, cf, ci, cf, vp]
    lib.tango_op_conv_transpose1d.argtypes = [ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, cf, vp]
    lib.tango_op_groupnorm.argtypes = [ci, vp, vp, vp, vp, ci, ci, ci, ci, cf, ci, vp]
    lib.tango_op_layernorm.argtypes = [ci, vp, vp, vp, vp, ci, ci, cf, vp]
    lib.tango_op_attention.argtypes = [ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, vp]
    lib.tango_op_attention_ex.argtypes = [ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, ci, vp]
    lib.tango_op_xattn_block.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, vp]
    lib.tango_op_sched_step.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, cf, ci, ci, ci, cf, vp]
    lib.tango_op_philox_normal.argtypes = [vp, ci, ci, ci, ci, C.c_uint64, ci, vp]
    _lib = lib
    return lib


def check(rc, what="tango engine call"):
    if rc != 0:
        msg = load().tango_last_error()
        msg = msg.decode() if msg else "unknown error"
        if "cannot be larger" in msg or "unknown prediction" in msg:
            raise ValueError(msg)
        raise RuntimeError("%s failed: %s" % (what, msg))
