/* tango_engine.h -- C ABI of the MI355X-native Tango text-to-audio inference engine.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference has no C plugin API:
 * its callers are Python duck-typed call sites.  Every entry point below names the
 * reference interface it replaces (paths relative to the reference tree):
 *
 *   tango_engine_denoise        AudioDiffusion.inference loop body        models.py:224-249
 *                               + DDPMScheduler.step                       mustango/diffusers/src/diffusers/schedulers/scheduling_ddpm.py:254-349
 *                               (+ DDIMScheduler.step                      .../scheduling_ddim.py:238-360)
 *   tango_engine_unet_forward   UNet2DConditionModel.forward               mustango/diffusers/src/diffusers/models/unet_2d_condition.py:520-707
 *   tango_engine_unet_forward_music  UNet2DConditionModelMusic.forward      mustango/diffusers/src/diffusers/models/unet_2d_condition_music.py:536-757
 *   tango_engine_vae_decode     AutoencoderKL.decode_first_stage           audioldm/variational_autoencoder/autoencoder.py:116-124,60-64
 *   tango_engine_vae_encode     AutoencoderKL.encode (encode_first_stage)  audioldm/variational_autoencoder/autoencoder.py:52-58,112-113
 *   tango_engine_vocode         AutoencoderKL.decode_to_waveform           audioldm/variational_autoencoder/autoencoder.py:66-69
 *                               -> vocoder_infer                           audioldm/hifigan/utilities.py:76-86
 *   tango_engine_mel_spectrogram  TacotronSTFT.mel_spectrogram            audioldm/audio/stft.py:164-186 (+ STFT.transform :52-85)
 *   tango_engine_encode_text    T5EncoderModel forward (text_encoder(...)[0])  models.py:139-141,279-281,291-293
 *   tango_engine_set_weight     load_state_dict of pytorch_model_main.bin / pytorch_model_vae.bin   tango.py:22-28
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  Tensor arguments are CALLER-OWNED DEVICE
 *     pointers (contiguous, the reference's own shapes/layouts: NCHW fp32 latents, [B,L,d] fp32
 *     embeddings, bool masks as uint8).  The engine never frees caller memory.
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream); all work is enqueued
 *     on it and is asynchronous unless stated.
 *   - every function returns 0 on success, non-zero on failure; tango_last_error() returns the
 *     message for the calling thread.  The Python shim re-raises ValueError / RuntimeError with the
 *     reference's conditions (e.g. num_inference_steps > num_train_timesteps).
 *   - a handle is not re-entrant; data-parallel use = one handle per GPU / process.
 */
#ifndef TANGO_ENGINE_H
#define TANGO_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TANGO_DTYPE_F32 0
#define TANGO_DTYPE_F16 1
#define TANGO_DTYPE_BF16 2

#define TANGO_PRED_EPSILON 0
#define TANGO_PRED_SAMPLE 1
#define TANGO_PRED_V 2

#define TANGO_RULE_DDPM 0
#define TANGO_RULE_DDIM 1

#define TANGO_MAX_LEVELS 8

typedef struct tango_engine tango_engine_t;

/* Mirrors configs/diffusion_model_config.json + mustango/configs/vae_config.json (ddconfig) +
 * HIFIGAN_16K_64 (audioldm/hifigan/utilities.py:9-39).  A component with n_levels == 0 / n_ups == 0
 * is not built. */
typedef struct tango_config {
  int32_t dtype;                              /* TANGO_DTYPE_*: storage/MFMA operand type; accumulation & statistics are fp32 */
  /* UNet2DConditionModel */
  int32_t unet_levels;                        /* len(block_out_channels) */
  int32_t unet_channels[TANGO_MAX_LEVELS];    /* block_out_channels */
  int32_t unet_heads[TANGO_MAX_LEVELS];       /* "attention_head_dim" (= head COUNT; head_dim is 64) */
  int32_t unet_cross_attn[TANGO_MAX_LEVELS];  /* 1 if down block i is CrossAttnDownBlock2D */
  int32_t unet_layers_per_block;
  int32_t unet_in_channels, unet_out_channels;
  int32_t unet_cross_dim;                     /* cross_attention_dim (1024 FLAN-T5-large, 2048 XL) */
  int32_t unet_groups;                        /* norm_num_groups */
  float unet_eps;                             /* norm_eps */
  int32_t unet_flip_sin_to_cos;
  float unet_freq_shift;
  int32_t latent_h, latent_w;                 /* 256, 16 (models.py:259-260) */
  /* mel-VAE decoder */
  int32_t vae_levels;                         /* len(ch_mult) */
  int32_t vae_ch;
  int32_t vae_ch_mult[TANGO_MAX_LEVELS];
  int32_t vae_num_res_blocks;
  int32_t vae_z_channels, vae_embed_dim, vae_out_ch;
  float vae_scale_factor;
  /* HiFi-GAN */
  int32_t voc_n_ups;
  int32_t voc_rates[TANGO_MAX_LEVELS];
  int32_t voc_kernels[TANGO_MAX_LEVELS];
  int32_t voc_initial_channel;
  int32_t voc_num_mels;
  int32_t voc_n_resblocks;                    /* len(resblock_kernel_sizes) == 3 */
  int32_t voc_res_kernels[4];
  int32_t voc_res_dilations[4][4];            /* [kernel idx][pair idx]; 0-terminated rows */
  /* FLAN-T5 encoder (transformers T5EncoderModel; reference call sites models.py:98-100,129-147,266-305).  Built when
   * t5_layers > 0; weights are the `text_encoder.*` tensors of pytorch_model_main.bin (HF key names). */
  int32_t t5_layers;                          /* num_layers (24 for flan-t5-large / -xl) */
  int32_t t5_d_model, t5_d_kv, t5_heads, t5_d_ff, t5_vocab;   /* d_kv must be 64 */
  int32_t t5_rel_buckets, t5_rel_max_distance;                /* relative_attention_num_buckets (32), _max_distance (128) */
  float t5_eps;                               /* layer_norm_epsilon (1e-6) */
  /* mel-VAE encoder (modules.py:419-543 + quant_conv, autoencoder.py:38,52-58): built when vae_encoder != 0 (needs the
   * vae_* fields above); weights are the `encoder.*` / `quant_conv.*` tensors of pytorch_model_vae.bin */
  int32_t vae_encoder;
  int32_t vae_in_channels;                    /* ddconfig["in_channels"] (1) */
  /* wave -> log-mel front-end (audioldm/audio/stft.py:136-186 TacotronSTFT; config keys of stft_config.json / models.py:40-47):
   * built when stft_filter_length > 0 (fp32 engines only); weights are the module's buffers `mel_basis` [n_mel, n_fft/2+1] and
   * `stft_fn.forward_basis` [2*(n_fft/2+1), 1, n_fft] (pytorch_model_stft.bin, tango.py:23-27) */
  int32_t stft_filter_length, stft_hop_length, stft_n_mel;
  /* Mustango's UNet2DConditionModelMusic (mustango/diffusers/src/diffusers/models/unet_2d_condition_music.py; config
   * mustango/configs/music_diffusion_model_config.json: CrossAttn{Down,Up}Block2DMusic / UNetMidBlock2DCrossAttnMusic): every
   * cross-attention site holds THREE Transformer2DModels applied in sequence -- `attentions` (text), `attentions2` (beat
   * embeddings), `attentions3` (chord embeddings), unet_2d_blocks.py:1199-1260,715-757,2372-2436.  All three conditions have
   * width unet_cross_dim. */
  int32_t unet_music;
  /* BASELINE config 5 ("bf16 + fp8 MFMA attention"): != 0 runs the P.V product of the UNet's self-attention sites on
   * v_mfma_f32_16x16x32_fp8_fp8 (P and V as OCP e4m3, fp32 accumulation, fp32 softmax statistics; Q.K^T stays in `dtype`).
   * 16-bit engines only.  Measured deviation: DESIGN.md section 3.  2 = the same product on the MX instruction
   * v_mfma_scale_f32_16x16x128_f8f6f4 (128 keys per MFMA, unit block scales, twice the matrix-pipe rate) at the sites whose sequence
   * length is a multiple of 128; same roundings. */
  int32_t unet_attn_fp8;
} tango_config_t;

typedef struct tango_denoise_args {
  float* latents;              /* in/out  [B, C_lat, H, W] fp32 NCHW: initial noise * init_noise_sigma in, x_0 out */
  const float* prompt_embeds;  /* [B2, L, d] fp32, B2 = 2B ([uncond; cond], models.py:301) when cfg, else B */
  const uint8_t* prompt_mask;  /* [B2, L] bool (1 = attend), may be NULL (no mask) */
  int32_t batch;               /* B */
  int32_t text_len;            /* L */
  int32_t num_steps;           /* N */
  const int64_t* timesteps;    /* HOST [N] (DDPMScheduler.set_timesteps, scheduling_ddpm.py:184-204) */
  const float* coef;           /* HOST [N][8]: sqrt(abar_t), sqrt(1-abar_t), coef_x0, coef_xt, sigma, sqrt(abar_prev), dir_coef, 0 */
  float guidance_scale;        /* CFG is on when > 1.0 (models.py:213) */
  int32_t prediction_type;     /* TANGO_PRED_* */
  int32_t rule;                /* TANGO_RULE_* */
  int32_t clip_sample;
  float clip_sample_range;
  const float* noise;          /* device [N][B, C_lat, H, W] fp32 injected step noise (parity mode) or NULL -> device Philox */
  uint64_t seed;               /* Philox key when noise == NULL */
  int32_t sample_offset;       /* global index of local sample 0 (keeps Philox noise independent of the DP sharding) */
  int32_t use_graph;           /* 1: replay the captured hipGraph of the UNet step; 0: eager launches */
  /* Music UNet only (cfg.unet_music; mustango/models.py:540-598 MusicAudioDiffusion.inference): beat / chord condition
   * embeddings [B2, beat_len | chord_len, d] fp32 with bool masks [B2, len] (may be NULL), ordered [uncond; cond] like
   * prompt_embeds (mustango/models.py:650-740).  Ignored (may be NULL / 0) for the plain UNet. */
  const float* beat_embeds;
  const uint8_t* beat_mask;
  int32_t beat_len;
  const float* chord_embeds;
  const uint8_t* chord_mask;
  int32_t chord_len;
  /* optional HOST copy of prompt_mask (bool [rows, text_len], the tokenizer's attention mask before it was uploaded).  The engine
   * picks its plan from the mask's structure (the unconditional rows of a CFG batch keep one key, models.py:282-289); with the host
   * copy it does so without reading the device mask back, i.e. without a host sync on `stream`.  NULL: read-back (one sync). */
  const uint8_t* prompt_mask_host;
} tango_denoise_args_t;

const char* tango_last_error(void);
const char* tango_version(void);

int tango_engine_create(const tango_config_t* cfg, tango_engine_t** out);
void tango_engine_destroy(tango_engine_t* h);

/* number of parameter tensors the engine expects / name of the i-th one (reference state_dict keys,
 * "unet." prefix for the UNet as in pytorch_model_main.bin, plain keys for pytorch_model_vae.bin). */
int tango_engine_num_weights(tango_engine_t* h);
const char* tango_engine_weight_name(tango_engine_t* h, int i);
/* fp32 DEVICE tensor in the reference layout (Conv OIHW, Linear [out,in], ConvTranspose1d [in,out,k]);
 * the engine packs its own copy. `shape`/`ndim` are validated. */
int tango_engine_set_weight(tango_engine_t* h, const char* name, const float* dev_ptr, const int64_t* shape, int ndim);
/* fails (listing the first missing key) unless every expected tensor has been set */
int tango_engine_finalize_weights(tango_engine_t* h);

int tango_engine_denoise(tango_engine_t* h, const tango_denoise_args_t* args, void* stream);

/* one UNet call: sample [B2,C,H,W] fp32 NCHW, timestep, embeds [B2,L,d], mask [B2,L] -> out [B2,C,H,W] fp32 */
int tango_engine_unet_forward(tango_engine_t* h, const float* sample, int64_t timestep, const float* prompt_embeds,
                              const uint8_t* prompt_mask, float* out, int batch2, int text_len, void* stream);

/* UNet2DConditionModelMusic.forward (unet_2d_condition_music.py:536-757): as tango_engine_unet_forward plus the beat / chord
 * conditions [B2, beat_len | chord_len, d] fp32 and their bool masks (may be NULL) */
int tango_engine_unet_forward_music(tango_engine_t* h, const float* sample, int64_t timestep, const float* prompt_embeds,
                                    const uint8_t* prompt_mask, const float* beat_embeds, const uint8_t* beat_mask,
                                    const float* chord_embeds, const uint8_t* chord_mask, float* out, int batch2, int text_len,
                                    int beat_len, int chord_len, void* stream);

/* latents [B,8,256,16] fp32 -> mel [B,1,1024,64] fp32 */
int tango_engine_vae_decode(tango_engine_t* h, const float* latents, float* mel, int batch, void* stream);
/* AutoencoderKL.encode (autoencoder.py:52-58: Encoder.forward, modules.py:519-543, then quant_conv): mel [B, in_ch, 4H, 4W] fp32
 * NCHW (H, W = latent_h, latent_w for three levels) -> moments [B, 2*embed_dim, H, W] fp32 NCHW = [mean | logvar]; the posterior
 * (clamp, exp, sample: distributions.py:24-41) and scale_factor (autoencoder.py:126-135) are host-side code of the caller */
int tango_engine_vae_encode(tango_engine_t* h, const float* mel, float* moments, int batch, void* stream);
/* mel [B,1,T,num_mels] fp32 -> int16 [B, samples]; returns samples per item via *n_samples (may be NULL) */
int tango_engine_vocode(tango_engine_t* h, const float* mel, int16_t* wav, int batch, int mel_frames, int* n_samples, void* stream);
int tango_engine_vocoder_samples(tango_engine_t* h, int mel_frames);

/* T5 encoder forward (replaces text_encoder(input_ids, attention_mask)[0], models.py:139-141,279-281,291-293):
 * input_ids int64 [B, L] (device), attention_mask uint8 [B, L] (device, 1 = token, may be NULL), out fp32 [B, L, d_model] */
int tango_engine_encode_text(tango_engine_t* h, const int64_t* input_ids, const uint8_t* attention_mask, float* out, int batch,
                             int text_len, void* stream);

/* TacotronSTFT.mel_spectrogram (audioldm/audio/stft.py:164-186; callers tools/torch_tools.py:57-78): wav fp32 [B, n_samples]
 * in [-1, 1] (device) -> mel fp32 [B, n_mel, T] = log(clamp(mel_basis @ |STFT|, 1e-5)), log_magnitudes fp32 [B, n_fft/2+1, T]
 * (may be NULL), energy fp32 [B, T] (may be NULL); T = 1 + n_samples / hop (tango_engine_mel_frames), also returned via
 * *n_frames (may be NULL).  n_samples must exceed n_fft / 2 (reflect padding). */
int tango_engine_mel_frames(tango_engine_t* h, int n_samples);
int tango_engine_mel_spectrogram(tango_engine_t* h, const float* wav, float* mel, float* log_magnitudes, float* energy, int batch,
                                 int n_samples, int* n_frames, void* stream);

/* timing of the last denoise call's kernels, measured with HIP events on the launch stream */
int tango_engine_last_denoise_ms(tango_engine_t* h, float* total_ms, float* per_step_ms);
/* GFLOP one launch of the last denoise call's UNet step program EXECUTES (sum over its kernel groups; bench.py reports it beside
 * the reference's algorithmic 1606.36 GFLOP per prompt and step, which it undercuts wherever work was designed out: the single-key
 * rows of models.py:282-289, the step-invariant to_k / to_v of attention_processor.py:519-520).  No reference counterpart. */
int tango_engine_last_step_gflop(tango_engine_t* h, double* gflop);

/* diagnostic: eager run of one UNet step with HIP events around every kernel group; writes
 * "label<TAB>ms<TAB>GFLOP" lines into `report` (truncated to report_cap). */
/* Plan cache.  The engine keeps one workspace slab + launch program ("plan") per call shape: UNet (batch, text / beat / chord
 * lengths, single-key prefix), VAE / vocoder (batch, frames), text encoder (batch, length), STFT (batch, samples).  All of them
 * share one byte budget (default 64 GiB, or TANGO_PLAN_BUDGET_MB); when a new plan does not fit, least recently used plans are
 * freed first (hipFree: synchronises the device).  The reference has no counterpart: tango.py:51-64 just re-runs PyTorch eagerly. */
int tango_engine_set_plan_budget(tango_engine_t* h, uint64_t bytes);
int tango_engine_plan_stats(tango_engine_t* h, uint64_t* bytes_in_use, int* plans);
/* frees every cached plan now (the next call of each shape rebuilds its plan); also what a measurement tool calls after
 * tango_tuning_reload() so that build-time routing decisions are taken again */
int tango_engine_drop_plans(tango_engine_t* h);

int tango_engine_profile_unet(tango_engine_t* h, int batch2, int text_len, char* report, int report_cap, void* stream);
/* the same per-op table ("label\tms\tGFLOP" lines, HIP events around every op of the eager plan) for the two stages that follow the
 * denoise loop in every pass: the mel-VAE decoder (audioldm/variational_autoencoder/modules.py:546-683, autoencoder.py:116-124) for
 * `batch` latents, and HiFi-GAN (audioldm/hifigan/models.py:96-165) for `batch` mels of `frames` frames.  Measurement tools only. */
int tango_engine_profile_vae(tango_engine_t* h, int batch, char* report, int report_cap, void* stream);
int tango_engine_profile_vocoder(tango_engine_t* h, int batch, int frames, char* report, int report_cap, void* stream);
/* measurement tools only: re-read the TANGO_* dispatch switches (csrc/tuning.h) from the environment, so that one process can
 * time several arms of an A/B back to back (tools/profile_unet_ops.py --ab).  No reference counterpart. */
void tango_tuning_reload(void);
/* host-side query, no GPU needed: the kernel family the GEMM dispatcher (csrc/gemm.hip gemm_route) picks for the 16-bit linear
 * y[M,N] = epi(x[M,K] W[N,K]^T): "wide", "wide+pers", "duo", "stream", "dma", "tile", "tile+splitk", "wide+xstats" (GEGLU with external
 * LayerNorm statistics), or "layernorm+<route>" when no kernel folds the LayerNorm for that shape.  dtype 1 = fp16, 2 = bf16.
 * tests/test_routing.py pins the measured routing rules of the UNet's shapes with it.  No reference counterpart. */
const char* tango_debug_linear_route(int dtype, int M, int N, int K, int geglu, int ln_fold, int residual, int vt);

/* ---- per-operator entry points (parity tests; fp32 reference-layout tensors on device) ---- */
int tango_op_conv2d(int dtype, const float* x, const float* w, const float* bias, float* out, int B, int Cin, int H, int W,
                    int Cout, int stride, int upsample, void* stream);
int tango_op_linear(int dtype, const float* x, const float* w, const float* bias, const float* residual, float* out, int M,
                    int N, int K, int a_act, int e_act, int geglu, void* stream);
int tango_op_linear_ln(int dtype, const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                       const float* residual, float* out, int M, int N, int K, int geglu, float eps, void* stream);
/* fused self-attention projection as the engine runs it (reference: diffusers attention_processor.py Attention.to_q/to_k/to_v on
   LayerNorm(x), transformer_2d BasicTransformerBlock.norm1): q | k row-major into out_qk [B*S, 2C], v TRANSPOSED into out_vt
   [B][C][S]; gamma == NULL skips the LayerNorm */
int tango_op_linear_qkv(int dtype, const float* x, const float* w, const float* gamma, const float* beta, float* out_qk,
                        float* out_vt, int B, int S, int C, int K, float eps, void* stream);
/* the same with the tokens of every block of 32 of out_vt in the attention kernel's fragment order when vt_perm = 1 (position 8 g + 4 hi + r
   holds token 16 hi + 4 g + r: what tango_op_attention's LDS-DMA form reads; 16-bit engines, S % 32 == 0) */
int tango_op_linear_qkv_perm(int dtype, const float* x, const float* w, const float* gamma, const float* beta, float* out_qk,
                             float* out_vt, int B, int S, int C, int K, float eps, int vt_perm, void* stream);
/* the level-0 feed-forward of BasicTransformerBlock with its LayerNorm and residual (reference: diffusers attention.py:326-335 norm3 -> ff ->
   + hidden_states, :338-387 FeedForward, :412-433 GEGLU): out [M, C] = x + W2 GEGLU(W1 LayerNorm(x) + b1) + b2, w1 [2H, C] in the reference
   row order (value rows, then gate rows), w2 [C, H].  mode 0 = ONE launch (csrc/ff_fused.hip; C = 320, H = 1280, M % 128 == 0, 16-bit),
   mode 1 = the engine's two-GEMM route.  reps > 0 with ms_out != NULL: mean milliseconds of `reps` repeats (HIP events) -- same-process A/B. */
int tango_op_ff_fused(int dtype, const float* x, const float* w1, const float* b1, const float* gamma, const float* beta, const float* w2,
                      const float* b2, float* out, int M, int C, int H, float eps, int mode, int reps, float* ms_out, void* stream);
/* as tango_op_linear_qkv with LayerNorm (K = C): mode 0 = the activation-stationary kernel of the level-0 sites (csrc/ff_fused.hip
   qkv_stat_kernel: C = 320, (B * S) % 256 == 0, S % 256 == 0, 16-bit), mode 1 = the GEMM route; reps / ms_out as tango_op_ff_fused */
int tango_op_qkv_stat(int dtype, const float* x, const float* w, const float* gamma, const float* beta, float* out_qk, float* out_vt, int B,
                      int S, int C, float eps, int mode, int reps, float* ms_out, void* stream);
int tango_op_conv1d(int dtype, const float* x, const float* w, const float* bias, const float* residual, float* out, int B,
                    int Cin, int L, int Cout, int k, int dilation, int a_act, float a_slope, int e_act, float e_slope, void* stream);
int tango_op_conv_transpose1d(int dtype, const float* x, const float* w, const float* bias, float* out, int B, int Cin, int L,
                              int Cout, int k, int stride, int padding, int a_act, float a_slope, void* stream);
int tango_op_groupnorm(int dtype, const float* x, const float* gamma, const float* beta, float* out, int B, int C, int HW,
                       int groups, float eps, int act, void* stream);
int tango_op_layernorm(int dtype, const float* x, const float* gamma, const float* beta, float* out, int rows, int C, float eps,
                       void* stream);
int tango_op_attention(int dtype, const float* q, const float* k, const float* v, const float* bias, float* out, int B, int heads,
                       int Sq, int Skv, float scale, void* stream);
/* as tango_op_attention; flags bit 0: P.V on the fp8 MFMA (16-bit dtypes, no bias, Skv % 64 == 0); bit 1 (with bit 0): on the MX
 * instruction, 128 keys per MFMA (Skv % 128 == 0); bit 2: the rows of v are given in the kernel's fragment order inside every block of
 * 32 keys (row 8 g + 4 hi + r of a block holds key 16 hi + 4 g + r) -- the op's V^T is then the permuted one the engine's producers write
 * and both tiles travel by LDS-DMA (unmasked 16-bit problems with Sq > 512, Skv % 64 == 0; refused otherwise) */
int tango_op_attention_ex(int dtype, const float* q, const float* k, const float* v, const float* bias, float* out, int B, int heads,
                          int Sq, int Skv, float scale, int flags, void* stream);
/* fused cross-attention block of BasicTransformerBlock (diffusers attention.py:312-323: attn2(norm2(x), text, mask) + x) as the engine
   runs it at level 0 (C = 320, 5 heads, 64 text tokens): x [B*HW, 320]; wq / wo [320, 320] Linear weights (to_q has no bias);
   k, v [B*L, 320] = to_k / to_v of the text; bias [B, L] additive or NULL */
int tango_op_xattn_block(int dtype, const float* x, const float* gamma, const float* beta, const float* wq, const float* k, const float* v,
                         const float* bias, const float* wo, const float* bo, float* out, int B, int HW, int L, float eps, void* stream);
int tango_op_sched_step(float* latents, const float* model_out_nchw, const float* noise, const float* coef8, int B, int C, int HW,
                        int cfg, float guidance, int pred_type, int rule, int clip, float clip_range, void* stream);

/* the N(0,1) values the denoise loop's device Philox generator injects at loop index `step` for global samples
 * [sample_offset, sample_offset + B): out fp32 [B, C, HW] (replaces randn_tensor inside DDPMScheduler.step,
 * mustango/diffusers/src/diffusers/schedulers/scheduling_ddpm.py:331-338, for throughput runs) */
int tango_op_philox_normal(float* out, int B, int C, int HW, int step, uint64_t seed, int sample_offset, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TANGO_ENGINE_H */
