// Common types for the Tango MI355X engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace tango {

enum DType : int { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2 };

typedef _Float16 f16;
typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }

template <typename T> struct TypeTag;
template <> struct TypeTag<float> { static constexpr int dt = DT_F32; };
template <> struct TypeTag<f16> { static constexpr int dt = DT_F16; };
template <> struct TypeTag<bf16> { static constexpr int dt = DT_BF16; };

// ---- error plumbing (C ABI returns int, message via tango_last_error) ----
void set_error(const std::string& msg);
#define TANGO_HIP(expr)                                                                        \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      ::tango::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " +   \
                         __FILE__ + ":" + std::to_string(__LINE__));                           \
      (void)hipGetLastError(); /* reset the sticky error: it must not resurface in an unrelated later call */ \
      return -1;                                                                               \
    }                                                                                          \
  } while (0)
#define TANGO_FAIL(msg)                                                                        \
  do {                                                                                         \
    ::tango::set_error(std::string(msg) + " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
    return -1;                                                                                 \
  } while (0)
#define TANGO_TRY(expr)       \
  do {                        \
    int _r = (expr);          \
    if (_r != 0) return _r;   \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: remembered per (function, device) under a
// mutex, so that a second engine on another GPU of the same process opts in as well (gemm.hip)
int ensure_dyn_lds(const void* kfn, int bytes);

// ---- device helpers ----
template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// "gelu_new" of T5 v1.1 / FLAN-T5 (transformers activations.py NewGELUActivation): tanh approximation
__device__ __forceinline__ float gelu_tanh_f(float x) {
  return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
// Exact-erf GELU for the 16-bit engines, two values per call: gelu(x) = h + h * e with h = x / 2 and
// e = erf(x / sqrt 2) ~ xc * Q(xc^2), xc = clamp(x, +-4.25), Q a degree-8 polynomial (weighted minimax fit pinned to e = 1
// at the clamp, tools/fit_gelu_poly.py): |gelu error| <= 6e-5 absolute over all x, below half an fp16 ulp of any output
// >= 0.125 and far below bf16's.  Why: ocml's erff is ~38 VALU instructions including a v_exp, and with 64 lanes both of
// its branches (|z| < 1 polynomial, exp tail) always execute; a GEGLU epilogue evaluates 40 gates per lane per 32 x 160
// output block, which made the fused GEGLU GEMMs VALU-bound (~6000 VALU cycles against 3200 MFMA cycles per block).
// Written on float2 so that hipcc emits v_pk_fma_f32 / v_pk_mul_f32: 15 VALU instructions per PAIR, no transcendentals.
// (The earlier attempt, a branch-free Abramowitz-Stegun form with v_rcp + v_exp per value, measured SLOWER than erff:
//  4.02 vs 3.33 ms on the level-1 GEGLU GEMMs.)  The fp32 engine keeps erff: its parity bar is 1e-5.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf_poly2(const f32x2_t x) {
  f32x2_t xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, -4.25f, 4.25f);
  xc.y = __builtin_amdgcn_fmed3f(x.y, -4.25f, 4.25f);
  const f32x2_t t = xc * xc;
  f32x2_t q = f32x2_t{1.498029197e-11f, 1.498029197e-11f} + t * 1.123676848e-12f;
  q = __builtin_elementwise_fma(q, t, f32x2_t{-7.180728823e-09f, -7.180728823e-09f});
  q = __builtin_elementwise_fma(q, t, f32x2_t{3.825664002e-07f, 3.825664002e-07f});
  q = __builtin_elementwise_fma(q, t, f32x2_t{-1.045047183e-05f, -1.045047183e-05f});
  q = __builtin_elementwise_fma(q, t, f32x2_t{1.811638670e-04f, 1.811638670e-04f});
  q = __builtin_elementwise_fma(q, t, f32x2_t{-2.193588131e-03f, -2.193588131e-03f});
  q = __builtin_elementwise_fma(q, t, f32x2_t{1.957916536e-02f, 1.957916536e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2_t{-1.326353318e-01f, -1.326353318e-01f});
  q = __builtin_elementwise_fma(q, t, f32x2_t{7.977887478e-01f, 7.977887478e-01f});
  const f32x2_t e = q * xc, h = x * 0.5f;
  return __builtin_elementwise_fma(h, e, h);
}
// gate activation of the fused gated-linear-unit epilogues, four gates at a time: exact-erf GELU (diffusers GEGLU) or
// tanh GELU (T5 gated-gelu); v[r] *= gate(g[r])
template <typename T, typename V4> __device__ __forceinline__ void glu_gate4(V4& v, const float g[4], int tanh_form) {
  if (tanh_form) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] * gelu_tanh_f(g[r]);
  } else if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] * gelu_erf_f(g[r]);
  } else {
    const f32x2_t a = gelu_erf_poly2(f32x2_t{g[0], g[1]}), b = gelu_erf_poly2(f32x2_t{g[2], g[3]});
    v[0] = v[0] * a.x; v[1] = v[1] * a.y; v[2] = v[2] * b.x; v[3] = v[3] * b.y;
  }
}

// activation codes shared by prologues / epilogues
enum Act : int { ACT_NONE = 0, ACT_SILU = 1, ACT_LRELU = 2, ACT_GELU = 3, ACT_TANH = 4, ACT_GELU_TANH = 5 };

__device__ __forceinline__ float apply_act(float x, int act, float slope) {
  switch (act) {
    case ACT_SILU: return silu_f(x);
    case ACT_LRELU: return x > 0.f ? x : x * slope;
    case ACT_GELU: return gelu_erf_f(x);
    case ACT_TANH: return tanhf(x);
    case ACT_GELU_TANH: return gelu_tanh_f(x);
    default: return x;
  }
}

// four values at once, the wave-uniform dispatch on `act` outside the element loop: apply_act per element made hipcc inline the
// whole switch (tanh, erf and exp code included) at every one of the 4 x TM x TN unrolled call sites of the tile kernels' epilogues
// (30 000 instructions per kernel).  Same arithmetic per element.
template <typename V4> __device__ __forceinline__ void apply_act4(V4& v, int act, float slope) {
  if (act == ACT_LRELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * slope;
  } else if (act == ACT_SILU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
  } else if (act == ACT_TANH) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
  } else if (act == ACT_GELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
  } else if (act == ACT_GELU_TANH) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(v[r]);
  }
}

// A (rows x C) activation view in HBM: element type is the engine dtype, row stride `ld` elements.
struct View {
  void* p = nullptr;
  int64_t ld = 0;
  int C = 0;
};

// ------------------------------------------------------------------------------------------
// Gather-GEMM: out[orow(m), n] = epi( alpha * sum_k A_gather[m, k] * W[n, k] + bias[n] ... )
// The A operand is gathered on the fly (implicit GEMM): linear rows, 3x3 conv2d taps on NHWC
// (stride 1/2, optional fused nearest-2x upsample of the source), or conv1d / transposed-conv1d
// phase taps on channels-last [B, L, C].  K = taps * Cin, tap-major, Cin % BK == 0.
// ------------------------------------------------------------------------------------------
enum GatherMode : int { GATHER_1D = 0, GATHER_2D = 1 };
enum Epi : int { EPI_NONE = 0, EPI_GEGLU = 1, EPI_I16 = 2, EPI_VT = 3 };

struct GemmParams {
  const void* A = nullptr;   // T
  const void* W = nullptr;   // T  [N][Kp], K contiguous
  const float* bias = nullptr;      // [N] (or [M-rows] when bias_rows)
  const float* bias2 = nullptr;     // [bias2_stride * steps][N]: per-step bias (time embedding)
  const int* step_ptr = nullptr;    // device step counter used to index bias2
  int bias2_stride = 0;
  const void* R = nullptr;   // residual T, indexed like out
  int64_t ldr = 0;
  void* out = nullptr;       // T, or float when out_f32, or int16 when epi == EPI_I16
  int64_t ldo = 0;
  int out_f32 = 0;
  int M = 0, N = 0, K = 0, Cin = 0;
  int64_t Kp = 0;            // W row stride
  int64_t lda = 0;
  int stage_epi = 0;           // set by the launchers: LDS-staged (row-contiguous) output stores
  int mode = GATHER_1D;
  // 2D: output grid H x W per image; source image Hin x Win (conv input is source upsampled if ups)
  int H = 0, Wd = 0, Hin = 0, Win = 0, stride = 1, ups = 0;
  int pad = 1;                 // GATHER_2D: top / left zero padding (1 = symmetric 3x3 "same"; 0 = the VAE Downsample's (0,1,0,1) pad)
  // 1D: rows_pb output rows per batch item; source length Lin; index = q*in_mul + in_off + tap*tap_step
  int rows_pb = 0, Lin = 0, taps = 1, tap_step = 1, in_off = 0, in_mul = 1;
  int Lout = 0, out_mul = 1, out_off = 0;  // out row = b*Lout + q*out_mul + out_off
  int a_act = ACT_NONE;      // prologue on A
  float a_slope = 0.f;
  int epi = EPI_NONE;
  int glu_tanh = 0;          // EPI_GEGLU gate activation: 0 exact-erf GELU (diffusers GEGLU), 1 tanh GELU (T5 gated-gelu)
  int e_act = ACT_NONE;      // epilogue activation (after bias, before residual)
  float e_slope = 0.f;
  float alpha = 1.f;
  float out_scale = 1.f;     // applied last (EPI_I16: 32768)
  int bias_rows = 0;         // bias indexed by output row instead of column
  // EPI_VT: columns n >= vt_n0 are the V projection and are stored TRANSPOSED for the attention kernel:
  //   vt[((m / vt_S) * (N - vt_n0) + (n - vt_n0)) * vt_ld + (m % vt_S)]   ( == [B][heads][64][vt_ld] )
  void* vt = nullptr;
  int vt_n0 = 0, vt_S = 1;
  int64_t vt_ld = 0;
  int vt_perm = 0;           // 1: the tokens of every block of 32 go out in the attention kernel's fragment order (vt_perm_pos; AttnParams::vt_perm); vt_S % 32 == 0
  // Per-sample row vector for a leading range of rows (round 4; gemm_wide_device.h wide_epilogue only -- the 256 x 320 and 256 x 160
  // kernels): rows m < rowvec_rows additionally get rowvec[(m / rowvec_per) * N + n] (fp32) and are written to out_lo (row stride
  // ldo_lo) instead of `out`.  The single-key cross-attention rows of a CFG batch (engine.hip transformer()): attn1's to_out + residual
  // writes x + attn1 + [to_out2(v_key) + b] for the unconditional samples in one pass.  rowvec_rows % 64 == 0, rowvec_per % 64 == 0.
  const float* rowvec = nullptr;
  int64_t rowvec_rows = 0;
  int rowvec_per = 1;
  void* out_lo = nullptr;
  int64_t ldo_lo = 0;
  // Per-sample weights (round 6: GroupNorm folded into proj_in, norm.hip gn_fold_kernel): rows [s * wb_rows, (s + 1) * wb_rows) multiply with
  // W + s * wb_stride elements (wb_rows % 256 == 0: a tile never straddles two samples).  256 x 320 GEMM kernels only.
  int64_t wb_stride = 0;
  int wb_rows = 0;
  // LayerNorm folded into the weights (linear_stream.hip): y = rstd*(W'x - mean*wsum) + b'
  int ln_fold = 0;
  float ln_eps = 1e-5f;
  const float* wsum = nullptr;
  // ln_fold with the row statistics computed by a separate pass (norm.hip ln_stats_kernel): [M][2] fp32 (mean, rstd).  Only the
  // 256 x 320 GEMM's GEGLU epilogue takes this (gemm_wide.hip XS): its main loop then carries no statistics VALU at all
  const float* row_stats = nullptr;
  // split-K (small-M GEMMs that cannot fill 256 CUs): blockIdx.y = split; raw fp32 partials go to
  // ws[split][M][N], a second kernel sums them in fixed order (deterministic) and applies the epilogue
  int splitk = 1;
  float* ws = nullptr;
  // batched GEMM (blockIdx.z)
  int batch = 1;
  int64_t sA = 0, sW = 0, sO = 0, sR = 0, sBias = 0;
};

int launch_gemm(int dtype, const GemmParams& p, hipStream_t s);
enum GemmRoute : int { ROUTE_NONE = 0, ROUTE_WIDE, ROUTE_STREAM, ROUTE_CONV_WIDE, ROUTE_CONV_HALO, ROUTE_DMA, ROUTE_TILE, ROUTE_DUO };
int gemm_route(int dtype, const GemmParams& p);   // which kernel family launch_gemm() picks for this problem
bool gemm_rowvec_ok(int dtype, const GemmParams& p);   // the kernel that takes this problem implements GemmParams::rowvec
bool conv_halo_ok(int dtype, const GemmParams& p);
int launch_conv_halo(int dtype, const GemmParams& p, const unsigned char* zero_page, hipStream_t s);
bool gemm_wide_ok(int dtype, const GemmParams& p);   // 256 x 320 ping-pong LDS-DMA GEMM for the big 16-bit linears (gemm_wide.hip)
int launch_gemm_wide(int dtype, const GemmParams& p, hipStream_t s);
int gemm_wide_pers_cus();                            // workgroups the persistent 256 x 320 GEMM launches (the device's CU count)
bool gemm_duo_ok(int dtype, const GemmParams& p);    // 256 x 160 LDS-DMA GEMM, two workgroups per CU, for the short-K 16-bit linears (gemm_duo.hip)
int launch_gemm_duo(int dtype, const GemmParams& p, hipStream_t s);
// a kernel with the LayerNorm folded into the weights takes this problem (else: LayerNorm kernel + plain GEMM)
inline bool gemm_ln_fold_ok(int dtype, const GemmParams& p);
bool conv_wide_ok(int dtype, const GemmParams& p);   // 3x3 halo-reuse conv on the 256 x 320 tile (conv_wide.hip)
int launch_conv_wide(int dtype, const GemmParams& p, const unsigned char* zero_page, hipStream_t s);
int conv_wide_pick_splitk(int dtype, const GemmParams& p);
int launch_splitk_reduce(int dtype, const GemmParams& p, hipStream_t s);   // gemm.hip: sums ws [splits][M][N] and applies the epilogue
bool gemm_dma_ok(int dtype, const GemmParams& p);
int launch_gemm_dma(int dtype, const GemmParams& p, const unsigned char* zero_page, hipStream_t s);
int gemm_init();   // process-wide one-time setup (zero page); call before any launch and outside stream capture
// split-K factor the launcher wants for this problem (1 = none); caller provides ws = splitk*M*N floats
int gemm_pick_splitk(int dtype, const GemmParams& p);
// weight-stationary streaming linear for K*sizeof(T) in {640, 1280} bytes (linear_stream.hip)
bool linear_stream_ok(int dtype, const GemmParams& p);
int launch_linear_stream(int dtype, const GemmParams& p, hipStream_t s);
inline bool gemm_ln_fold_ok(int dtype, const GemmParams& p) { return linear_stream_ok(dtype, p) || gemm_wide_ok(dtype, p) || gemm_duo_ok(dtype, p); }
int launch_fold_ln(int dtype, const void* W, int64_t Kp, const float* gamma, const float* beta, const float* bias_in, void* Wo,
                   float* bias_out, float* wsum, int N, int K, hipStream_t s);

// ---- fused level-0 feed-forward (ff_fused.hip): out = x + W2 GEGLU(W1' LayerNorm(x) + b1') + b2 ----
struct FFParams {
  const void* x = nullptr; int64_t ldx = 0;      // [M][C] rows (LayerNorm input AND residual)
  const void* w1 = nullptr; int64_t ld1 = 0;     // LayerNorm-folded GEGLU projection W' [2H][ld1], rows interleaved [16 value | 16 gate] (launch_pack perm 1 + launch_fold_ln)
  const float* b1 = nullptr;                     // folded bias b' [2H], same row order
  const void* w2 = nullptr; int64_t ld2 = 0;     // [C][ld2 >= H]
  const float* b2 = nullptr;                     // [C]
  void* out = nullptr; int64_t ldo = 0;          // [M][C]
  int M = 0, C = 0, H = 0;
  float eps = 1e-5f;
  int plain_loop = 0;                            // 1: the compiler-scheduled main loop (TANGO_FF_FUSED=2: the A/B arm of the asm-pipelined LDS stream)
};
bool ff_fused_ok(int dtype, const FFParams& p);
int launch_ff_fused(int dtype, const FFParams& p, hipStream_t s);

// ---- activation-stationary LayerNorm + linear for K = 320 (ff_fused.hip qkv_stat_kernel): y = W' LayerNorm(x) + b', columns [0, n_rm) row-major
// into out, columns [n_rm, N) transposed into vt [M / vt_S][N - n_rm][vt_ld] ----
struct QKVParams {
  const void* x = nullptr; int64_t ldx = 0;      // [M][K]
  const void* w = nullptr; int64_t ldw = 0;      // LayerNorm-folded weights W' [N][ldw] (launch_fold_ln), or plain weights with ln = 0
  const float* b = nullptr;                      // folded bias b' [N] (may be null)
  void* out = nullptr; int64_t ldo = 0;
  void* vt = nullptr; int64_t vt_ld = 0; int vt_S = 0;
  int vt_perm = 0;                             // 1: tokens permuted inside every block of 32 (AttnParams::vt_perm)
  int M = 0, N = 0, K = 0, n_rm = 0;
  int ln = 1; float eps = 1e-5f;               // ln: 0 = no normalisation, 1 = LayerNorm over K inside the kernel, 2 = GroupNorm with given statistics:
  const float* gstats = nullptr;               //   mean / rstd per (sample, group) [M / rows_ps][groups][2] (norm.hip launch_gn_stats_mr), w / b folded with its gamma / beta
  int groups = 0, rows_ps = 0;
};
bool qkv_stat_ok(int dtype, const QKVParams& p);
int launch_qkv_stat(int dtype, const QKVParams& p, hipStream_t s);

// ---- norms ----
struct GroupNormParams {
  const void* x; int64_t ldx;     // [B*rows, C]
  void* y; int64_t ldy;
  const float* gamma; const float* beta;
  int B, rows, C, groups;
  float eps;
  int act;                        // ACT_NONE / ACT_SILU
  float* partial;                 // workspace: [B][chunks][groups][2]
  float* scale_shift;             // workspace: [B][C][2]
  unsigned* sync = nullptr;       // cross-workgroup barrier words (coop_sync_words() of them, zeroed once, owned by the engine): enables the
                                  // single-launch cooperative kernel when the whole grid is co-resident; null = always two launches
};
// Barrier words for kernels whose workgroups rendezvous inside one launch (norm.hip gn_coop_kernel): [0, 1024) arrival counters,
// [1024, 2048) generations; self-resetting, so one buffer serves every such launch of ONE stream-ordered sequence.
constexpr int COOP_SYNC_SLOTS = 1024;
constexpr int coop_sync_words() { return 2 * COOP_SYNC_SLOTS + 16; }   // + [2048]: sticky timeout flag
size_t groupnorm_ws_floats(int B, int rows, int C, int groups);
// GroupNorm folded into the linear that consumes it (round 6): the statistics pass alone (partials as launch_groupnorm's two-launch path
// leaves them), then per sample s: Wf[s][n][k] = round(W[n][k] * gamma[k] * rstd[s][g(k)]), bf[s][n] = b[n] + sum_k W[n][k] * beta[k]
// - sum_k Wf[s][n][k] * mean[s][g(k)] (the mean term against the ROUNDED weights: it cancels exactly what the GEMM adds).
int launch_gn_stats_fold(int dtype, const GroupNormParams& p, const void* W, int64_t Kp, const float* bias, int N, void* Wf, float* bf, hipStream_t s);
bool gn_fold_ok(int dtype, const GroupNormParams& p, int N);
// statistics pass + finalisation: mr [B][groups][2] = mean, rstd (p.partial = workspace of groupnorm_ws_floats)
bool gn_stats_mr_ok(int dtype, const GroupNormParams& p);
int launch_gn_stats_mr(int dtype, const GroupNormParams& p, float* mr, hipStream_t s);
int launch_groupnorm(int dtype, const GroupNormParams& p, hipStream_t s);

int launch_layernorm(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma,
                     const float* beta, int rows, int C, float eps, hipStream_t s);
// LayerNorm statistics only: stats[row] = (mean, 1 / sqrt(var + eps)), two-pass in registers like launch_layernorm
int launch_ln_stats(int dtype, const void* x, int64_t ldx, float* stats, int rows, int C, float eps, hipStream_t s);

// ---- attention (head_dim 64) ----
struct AttnParams {
  const void* q; int64_t ldq;     // [B*Sq, >= heads*64]
  const void* k; int64_t ldk;     // [B*Skv, ...]
  const void* vt; int64_t ldvt;   // V transposed: [B][heads][64][ldvt] (columns >= Skv must be zero / finite)
  void* o; int64_t ldo;
  const float* bias;              // [B][Skv] additive or null
  const float* pos_bias = nullptr; // [heads][Sq][Skv] additive, shared by the batch (T5 relative position bias) or null
  int B, heads, Sq, Skv;
  float scale;
  int vt_perm = 0;                // 1: vt's keys are stored permuted inside every block of 32 (position 8 gg + 4 hi + r holds key 16 hi + 4 gg + r: the order the P^T
                                  // fragments present them), written so by qkv_stat_kernel; the kernel then fetches BOTH tiles by LDS-DMA (attention.hip KDMA + VDMA)
  int fp8_pv = 0;                 // 16-bit engines, unmasked (self-attention) sites: P and V as e4m3 on the fp8 MFMA (attention.hip); 2 = the MX instruction (128 keys per MFMA, unit scales)
};
int launch_attention(int dtype, const AttnParams& p, hipStream_t s);
// position of token s of a sequence in a vt_perm V^T: inside its block of 32, key 16 hi + 4 g + r sits at 8 g + 4 hi + r
__host__ __device__ inline int vt_perm_pos(int s) { return (s & ~31) | ((s & 12) << 1) | ((s & 16) >> 2) | (s & 3); }
bool attention_vt_perm_ok(int dtype, const AttnParams& p);   // this site can read a vt_perm V^T (the producer may then write one)

// ---- fused cross-attention block (xattn.hip): y = x + to_out(softmax(to_q(LayerNorm(x)) K^T / 8 + bias) V) + b_out ----
struct XAttnParams {
  const void* x; int64_t ldx;          // [M][C] T: the residual stream (LayerNorm input AND residual)
  const void* wq;                      // [C][C] T: LayerNorm-folded to_q weight, rows permuted per head (launch_xattn_permute_wq)
  const float* bq; const float* wsum;  // [C] fp32, permuted like wq's rows: folded bias b' = W beta, wsum = sum_k W'[n][k]
  const void* k; int64_t ldk;          // [B*L][C] T: to_k(text)   (step-invariant, computed once per call)
  const void* vt; int64_t ldvt;        // [B][C][ldvt] T: to_v(text) transposed
  const float* bias;                   // [B][L] fp32 additive mask bias or null
  const void* wo; int64_t ldwo;        // [C][C] T: to_out.0 weight, natural layout
  const float* bo;                     // [C] fp32
  void* out; int64_t ldo;              // [M][C] T
  int M, HW, L;                        // rows, rows per sample, text tokens
  float eps, scale;
};
bool xattn_block_ok(int dtype, int C, int heads, int HW, int L, int64_t ldx, int64_t ldo, int64_t ldk, int64_t ldvt);
int launch_xattn_block(int dtype, const XAttnParams& p, hipStream_t s);
int launch_xattn_permute_wq(int dtype, const void* W, const float* b, const float* wsum, void* Wp, float* bp, float* wsp, int C, hipStream_t s);

// single-key cross-attention rows (round 4): cvec[b][n] = bo[n] + sum_j Wo[n][j] * V^T[b][j][key0[b]]  (fp32), b < nb;
// then y[r][c] = x[r][c] + cvec[r / rows_per][c] for the rows of those samples (= attn2(norm2(x)) + x when one key is unmasked)
int launch_xattn_const(int dtype, const void* vt, int64_t ldvt, int C, const int* key0, const void* wo, int64_t ldwo, const float* bo,
                       float* cvec, int nb, hipStream_t s);
int launch_copy_rows(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int C, hipStream_t s);   // strided row copy
int launch_rowbias_add(int dtype, const void* x, int64_t ldx, const float* cvec, void* y, int64_t ldy, int64_t rows, int rows_per, int C,
                       hipStream_t s);

// row softmax (in place) used by the VAE single-head 512-d attention: x[rows][cols] *= scale; softmax
int launch_softmax_rows(int dtype, void* x, int64_t ld, int rows, int cols, float scale, hipStream_t s);

// ---- elementwise / layout ----
struct SchedParams {
  float* lat;            // [B,8,256,16] fp32 NCHW, updated in place
  const float* eps;      // UNet output fp32 NHWC [B2, HW, C] (B2 = 2B when cfg)
  void* xin;             // next UNet input, T, NHWC [B2, HW, Cpad]
  int xin_ld;
  const float* noise;    // [steps][B*C*HW] or null -> philox
  const float* coef;     // [steps][8] device
  const int* step_ptr;
  int B, C, HW;
  int cfg;               // 1: eps holds [uncond; cond]
  float guidance;
  int pred_type;         // 0 epsilon, 1 sample, 2 v_prediction
  int rule;              // 0 DDPM, 1 DDIM
  int clip; float clip_range;
  unsigned long long seed;
  int sample_offset;     // global sample index of local sample 0 (multi-GPU invariant noise)
};
// `dev_params` is a DEVICE copy of SchedParams; the grid covers max_positions >= B*HW (one thread per latent position)
int launch_sched_step(int dtype, const SchedParams* dev_params, int max_positions, hipStream_t s);
int launch_step_inc(int* step_ptr, hipStream_t s);
// test hook: the N(0,1) draws of sched_step at loop index `step` -> out fp32 [B][C][HW]
int launch_philox_normal(float* out, int B, int C, int HW, int step, unsigned long long seed, int sample_offset, hipStream_t s);

// latents fp32 NCHW [B,C,HW] -> T NHWC [rep*B, HW, ld] (replicated `rep` times along batch), zero-pads C..ld? no: writes C channels
int launch_nchw_to_nhwc(int dtype, const float* src, void* dst, int64_t ld, int B, int C, int HW, int rep, float scale, hipStream_t s);
// T NHWC [B,HW,ld] (first C channels) -> fp32 NCHW
int launch_nhwc_to_nchw_f32(int dtype, const void* src, int64_t ld, float* dst, int B, int C, int HW, hipStream_t s);
// fp32 -> T cast with row stride (embeddings): src [rows][C] contiguous
int launch_cast_rows(int dtype, const float* src, void* dst, int64_t ld, int rows, int C, hipStream_t s);
// im2col for tiny-Cin 3x3 convs: src T NHWC [B,H,W,(ld)] first C channels -> dst T [B*H*W][Kp], k = tap*C + c, zero padded to Kp
int launch_im2col3x3(int dtype, const void* src, int64_t ld, void* dst, int64_t Kp, int B, int H, int W, int C, hipStream_t s);
// y = act((a + b + c) * scale)
int launch_avg3_act(int dtype, const void* a, const void* b, const void* c, void* y, int64_t n, float scale, int act, float slope, hipStream_t s);
// mask bool [B][L] -> additive bias fp32 (1-m)*-10000
int launch_mask_bias(const uint8_t* mask, float* bias, int n, hipStream_t s);
// sinusoidal timestep embedding table: out fp32 [n][dim]; [cos|sin] when flip
int launch_timestep_embedding(const int64_t* ts_dev, float* out, int n, int dim, int flip, float freq_shift, hipStream_t s);
// small fp32 linear: y[r][n] = act_out( sum_k act_in(x[r][k]) * W[n][k] + b[n] ), W fp32 [N][K]
int launch_linear_f32(const float* x, const float* W, const float* b, float* y, int rows, int N, int K, int act_in, int act_out, hipStream_t s);

// ---- weight packing (fp32 source in reference layout -> T engine layout) ----
// generic strided permute-cast: dst[o][t][i] = src[o*so + t*st + i*si], dst row stride Kp (zero pad)
// dst_row_off: >= 0 plain row offset; -1 GEGLU interleave of a [2*half][K] matrix (value rows then gate rows);
// -2 / -3: the source holds ONLY the value / gate half ([half][K], O = half) of such a matrix (T5 wi_1 / wi_0)
int launch_pack(int dtype, const float* src, void* dst, int O, int Tn, int I, int64_t so, int64_t st, int64_t si,
                int64_t Kp, int64_t dst_row_off, hipStream_t s);
int launch_fill_zero(void* p, size_t bytes, hipStream_t s);
// lat f32 NCHW [B,Cin,HW] * scale -> 1x1 conv (W f32 [Cout][Cin], b) -> T NHWC [B*HW][ld]
int launch_pointwise_out_nchw(const float* src, int64_t ld, const float* W, const float* b, float* dst, int B, int Cin, int Cout, int HW,
                              hipStream_t s);   // fp32 rows [B*HW][ld] -> 1x1 conv -> fp32 NCHW [B][Cout][HW] (quant_conv)
int launch_pointwise_small(int dtype, const float* src, const float* W, const float* b, void* dst, int64_t ld, int B, int Cin,
                           int Cout, int HW, float scale, hipStream_t s);
int launch_permute_geglu_bias(const float* src, float* dst, int n, hipStream_t s);
// ---- text encoder (T5) helpers ----
// rows out[r][:] = table[ids[r]][:] (fp32 table -> T)
int launch_embed_gather(int dtype, const int64_t* ids, const float* table, void* out, int64_t ld, int rows, int D, int vocab, hipStream_t s);
// T5LayerNorm (RMS norm, no mean subtraction, no bias): y = x * rsqrt(mean(x^2) + eps) * gamma; out_f32: write fp32 instead of T
int launch_rmsnorm(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, int rows, int C, float eps, int out_f32, hipStream_t s);
// pos_bias[h][i][j] = table[bucket[i][j]][h]   (table fp32 [num_buckets][heads], bucket int32 [L][L] from the host)
int launch_t5_pos_bias(const float* table, const int* bucket, float* out, int heads, int L, hipStream_t s);

}  // namespace tango
