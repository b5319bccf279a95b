// Per-operator C entry points used by the parity tests (tests/test_ops_gpu.py).  Each takes fp32
// device tensors in the reference's own layout, converts to the engine dtype/layout, runs the SAME
// kernels the engine plans use, and converts back.  Not on the product hot path.
#include <string>
#include <vector>

#include "../../include/tango_engine.h"
#include "common.h"
#include "tuning.h"

namespace tango {

struct Scratch {
  std::vector<void*> ptrs;
  ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
  void* get(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return p;
  }
};

template <typename T>
__global__ void to_f32_kernel(const T* __restrict__ src, int64_t ld, float* __restrict__ dst, int64_t rows, int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * C) return;
  const int64_t r = i / C;
  dst[i] = to_f(src[r * ld + (i - r * C)]);
}
static int to_f32(int dt, const void* src, int64_t ld, float* dst, int64_t rows, int C, hipStream_t s) {
  const unsigned nb = (unsigned)((rows * C + 255) / 256);
  switch (dt) {
    case DT_F32: hipLaunchKernelGGL((to_f32_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)src, ld, dst, rows, C); break;
    case DT_F16: hipLaunchKernelGGL((to_f32_kernel<f16>), dim3(nb), dim3(256), 0, s, (const f16*)src, ld, dst, rows, C); break;
    case DT_BF16: hipLaunchKernelGGL((to_f32_kernel<bf16>), dim3(nb), dim3(256), 0, s, (const bf16*)src, ld, dst, rows, C); break;
    default: TANGO_FAIL("to_f32: bad dtype");
  }
  TANGO_HIP(hipGetLastError());
  return 0;
}

// v [B][S][C] -> vt [B][C][ldvt] (what the fused projection epilogue produces in the engine)
template <typename T>
__global__ void transpose_v_kernel(const T* __restrict__ v, T* __restrict__ vt, int B, int S, int C, int64_t ldvt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * S * C) return;
  const int c = (int)(i % C);
  const int64_t bs = i / C;
  const int s = (int)(bs % S), b = (int)(bs / S);
  vt[((int64_t)b * C + c) * ldvt + s] = v[i];
}
static int transpose_v(int dt, const void* v, void* vt, int B, int S, int C, int64_t ldvt, hipStream_t s) {
  const unsigned nb = (unsigned)(((int64_t)B * S * C + 255) / 256);
  switch (dt) {
    case DT_F32: hipLaunchKernelGGL((transpose_v_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)v, (float*)vt, B, S, C, ldvt); break;
    case DT_F16: hipLaunchKernelGGL((transpose_v_kernel<f16>), dim3(nb), dim3(256), 0, s, (const f16*)v, (f16*)vt, B, S, C, ldvt); break;
    case DT_BF16: hipLaunchKernelGGL((transpose_v_kernel<bf16>), dim3(nb), dim3(256), 0, s, (const bf16*)v, (bf16*)vt, B, S, C, ldvt); break;
    default: TANGO_FAIL("transpose_v: bad dtype");
  }
  TANGO_HIP(hipGetLastError());
  return 0;
}

// [B, C, L] fp32 <-> channels-last T [B*L, C] are the NCHW<->NHWC kernels with HW = L
}  // namespace tango

using namespace tango;

// test wrappers honour the same split-K policy as the engine plans
static int run_gemm(int dt, GemmParams& p, Scratch& sc, hipStream_t s) {
  TANGO_TRY(gemm_init());
  const int sk = gemm_pick_splitk(dt, p);
  if (sk > 1) {
    p.splitk = sk;
    p.ws = (float*)sc.get((size_t)sk * p.M * p.N * 4);
    if (!p.ws) TANGO_FAIL("run_gemm: alloc");
  }
  return launch_gemm(dt, p, s);
}

extern "C" {

int tango_op_conv2d(int dt, const float* x, const float* w, const float* bias, float* out, int B, int Cin, int H, int W, int Cout,
                    int stride, int upsample, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype_size(dt);
  Scratch sc;
  const int Hc = H << upsample, Wc = W << upsample;
  const int Ho = stride == 2 ? (Hc + 2 - 3) / 2 + 1 : Hc, Wo = stride == 2 ? (Wc + 2 - 3) / 2 + 1 : Wc;
  const bool im2col = ((Cin * esz) % 64) != 0;
  const int cpad = im2col ? ((Cin + 7) / 8) * 8 : Cin;
  void* xt = sc.get((size_t)B * H * W * cpad * esz);
  void* ot = sc.get((size_t)B * Ho * Wo * Cout * esz);
  if (!xt || !ot) TANGO_FAIL("op_conv2d: alloc");
  TANGO_HIP(hipMemsetAsync(xt, 0, (size_t)B * H * W * cpad * esz, s));
  TANGO_TRY(launch_nchw_to_nhwc(dt, x, xt, cpad, B, Cin, H * W, 1, 1.0f, s));
  GemmParams p;
  p.bias = bias; p.N = Cout; p.out = ot; p.ldo = Cout;
  if (im2col) {
    if (stride != 1 || upsample) TANGO_FAIL("op_conv2d: small-Cin path supports stride 1 only");
    const int64_t Kp = ((9 * Cin + 31) / 32) * 32;
    void* wt = sc.get((size_t)Cout * Kp * esz);
    void* col = sc.get((size_t)B * H * W * Kp * esz);
    if (!wt || !col) TANGO_FAIL("op_conv2d: alloc");
    TANGO_TRY(launch_pack(dt, w, wt, Cout, 9, Cin, (int64_t)Cin * 9, 1, 9, Kp, 0, s));
    TANGO_TRY(launch_im2col3x3(dt, xt, cpad, col, Kp, B, H, W, Cin, s));
    p.A = col; p.lda = Kp; p.W = wt; p.Kp = Kp; p.M = B * H * W; p.K = (int)Kp; p.Cin = (int)Kp;
    p.mode = GATHER_1D; p.rows_pb = p.M; p.Lin = p.M; p.Lout = p.M;
  } else {
    const int64_t Kp = 9 * Cin;
    void* wt = sc.get((size_t)Cout * Kp * esz);
    if (!wt) TANGO_FAIL("op_conv2d: alloc");
    TANGO_TRY(launch_pack(dt, w, wt, Cout, 9, Cin, (int64_t)Cin * 9, 1, 9, Kp, 0, s));
    p.A = xt; p.lda = cpad; p.W = wt; p.Kp = Kp; p.M = B * Ho * Wo; p.K = 9 * Cin; p.Cin = Cin;
    p.mode = GATHER_2D; p.H = Ho; p.Wd = Wo; p.Hin = H; p.Win = W; p.stride = stride; p.ups = upsample;
  }
  TANGO_TRY(run_gemm(dt, p, sc, s));
  TANGO_TRY(launch_nhwc_to_nchw_f32(dt, ot, Cout, out, B, Cout, Ho * Wo, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

int tango_op_linear(int dt, const float* x, const float* w, const float* bias, const float* residual, float* out, int M, int N,
                    int K, int a_act, int e_act, int geglu, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype_size(dt);
  Scratch sc;
  const int No = geglu ? N / 2 : N;
  void* xt = sc.get((size_t)M * K * esz);
  void* wt = sc.get((size_t)N * K * esz);
  void* ot = sc.get((size_t)M * No * esz);
  void* rt = residual ? sc.get((size_t)M * No * esz) : nullptr;
  float* bt = bias ? (float*)sc.get((size_t)N * 4) : nullptr;
  if (!xt || !wt || !ot) TANGO_FAIL("op_linear: alloc");
  TANGO_TRY(launch_cast_rows(dt, x, xt, K, M, K, s));
  TANGO_TRY(launch_pack(dt, w, wt, N, 1, K, K, 0, 1, K, geglu ? -1 : 0, s));
  if (residual) TANGO_TRY(launch_cast_rows(dt, residual, rt, No, M, No, s));
  if (bias) {
    if (geglu) TANGO_TRY(launch_permute_geglu_bias(bias, bt, N, s));
    else TANGO_HIP(hipMemcpyAsync(bt, bias, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
  }
  GemmParams p;
  p.A = xt; p.lda = K; p.W = wt; p.Kp = K; p.bias = bt; p.M = M; p.N = N; p.K = K; p.Cin = K;
  p.mode = GATHER_1D; p.rows_pb = M; p.Lin = M; p.Lout = M;
  p.out = ot; p.ldo = No; p.R = rt; p.ldr = No; p.a_act = a_act; p.e_act = e_act; p.epi = geglu ? EPI_GEGLU : EPI_NONE;
  TANGO_TRY(run_gemm(dt, p, sc, s));
  TANGO_TRY(to_f32(dt, ot, No, out, M, No, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

int tango_op_linear_ln(int dt, const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                       const float* residual, float* out, int M, int N, int K, int geglu, float eps, void* stream) {
  // LayerNorm(x) @ W^T (+bias, GEGLU, +residual): streaming kernel with folded LN when eligible, else LN + GEMM
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype_size(dt);
  Scratch sc;
  const int No = geglu ? N / 2 : N;
  void* xt = sc.get((size_t)M * K * esz);
  void* wt = sc.get((size_t)N * K * esz);
  void* wl = sc.get((size_t)N * K * esz);
  void* ot = sc.get((size_t)M * No * esz);
  void* rt = residual ? sc.get((size_t)M * No * esz) : nullptr;
  float* bt = (float*)sc.get((size_t)N * 4);
  float* bl = (float*)sc.get((size_t)N * 4);
  float* ws = (float*)sc.get((size_t)N * 4);
  if (!xt || !wt || !wl || !ot || !bt || !bl || !ws) TANGO_FAIL("op_linear_ln: alloc");
  TANGO_TRY(launch_cast_rows(dt, x, xt, K, M, K, s));
  TANGO_TRY(launch_pack(dt, w, wt, N, 1, K, K, 0, 1, K, geglu ? -1 : 0, s));
  if (residual) TANGO_TRY(launch_cast_rows(dt, residual, rt, No, M, No, s));
  if (bias) {
    if (geglu) TANGO_TRY(launch_permute_geglu_bias(bias, bt, N, s));
    else TANGO_HIP(hipMemcpyAsync(bt, bias, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
  }
  TANGO_TRY(launch_fold_ln(dt, wt, K, gamma, beta, bias ? bt : nullptr, wl, bl, ws, N, K, s));
  GemmParams p;
  p.A = xt; p.lda = K; p.W = wl; p.Kp = K; p.bias = bl; p.M = M; p.N = N; p.K = K; p.Cin = K;
  p.mode = GATHER_1D; p.rows_pb = M; p.Lin = M; p.Lout = M;
  p.out = ot; p.ldo = No; p.R = rt; p.ldr = No; p.epi = geglu ? EPI_GEGLU : EPI_NONE;
  p.ln_fold = 1; p.ln_eps = eps; p.wsum = ws;
  TANGO_TRY(gemm_init());
  GemmParams px = p;
  px.row_stats = (float*)sc.get((size_t)M * 2 * 4);
  if (gemm_ln_fold_ok(dt, p)) {
    TANGO_TRY(launch_gemm(dt, p, s));
  } else if (geglu && !residual && px.row_stats && !tuning().no_ln_xstats && gemm_wide_ok(dt, px) && gemm_route(dt, px) == ROUTE_WIDE) {
    // the engine's route for the GEGLU projections of levels 1-2: read-only statistics pass + folded weights (gemm_wide.hip XS)
    TANGO_TRY(launch_ln_stats(dt, xt, K, (float*)px.row_stats, M, K, eps, s));
    TANGO_TRY(launch_gemm(dt, px, s));
  } else {
    void* nt = sc.get((size_t)M * K * esz);
    if (!nt) TANGO_FAIL("op_linear_ln: alloc");
    TANGO_TRY(launch_layernorm(dt, xt, K, nt, K, gamma, beta, M, K, eps, s));
    p.A = nt; p.W = wt; p.bias = bias ? bt : nullptr; p.ln_fold = 0; p.wsum = nullptr;
    TANGO_TRY(launch_gemm(dt, p, s));
  }
  TANGO_TRY(to_f32(dt, ot, No, out, M, No, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

const char* tango_debug_linear_route(int dt, int M, int N, int K, int geglu, int ln_fold, int residual, int vt) {
  static thread_local std::string out;
  void* const dummy = (void*)(uintptr_t)0x10000;           // aligned, never dereferenced: the *_ok() predicates only look at alignment
  GemmParams p;
  p.A = dummy; p.lda = K; p.W = dummy; p.Kp = K; p.bias = (const float*)dummy; p.M = M; p.N = N; p.K = K; p.Cin = K;
  p.mode = GATHER_1D; p.rows_pb = M; p.Lin = M; p.Lout = M;
  const int No = geglu ? N / 2 : N;
  p.out = dummy; p.ldo = vt ? 2 * (N / 3) : No;
  if (residual) { p.R = dummy; p.ldr = No; }
  p.epi = geglu ? EPI_GEGLU : (vt ? EPI_VT : EPI_NONE);
  if (vt) { p.vt = dummy; p.vt_n0 = 2 * (N / 3); p.vt_S = 256; p.vt_ld = 256; }
  auto name = [&](const GemmParams& g) -> std::string {
    if (gemm_pick_splitk(dt, g) > 1) return "tile+splitk";
    switch (gemm_route(dt, g)) {
      case ROUTE_WIDE: {
        const long tiles = (long)(g.M / 256) * (g.N / 320);
        if (g.row_stats) return tuning().wide_pers > 0 && tiles >= (long)gemm_wide_pers_cus() * tuning().wide_pers ? "wide+xstats+pers" : "wide+xstats";
        const bool pers_ok = !(g.epi == EPI_GEGLU && (g.R || g.ln_fold));
        return tuning().wide_pers > 0 && pers_ok && tiles >= (long)gemm_wide_pers_cus() * tuning().wide_pers ? "wide+pers" : "wide";
      }
      case ROUTE_DUO: return "duo";
      case ROUTE_STREAM: return "stream";
      case ROUTE_DMA: return "dma";
      case ROUTE_TILE: return "tile";
      default: return "none";
    }
  };
  if (!ln_fold) { out = name(p); return out.c_str(); }
  GemmParams q = p;
  q.ln_fold = 1; q.wsum = (const float*)dummy;
  if (gemm_ln_fold_ok(dt, q)) { out = name(q); return out.c_str(); }
  GemmParams q2 = q;
  q2.row_stats = (const float*)dummy;
  if (geglu && !residual && !tuning().no_ln_xstats && gemm_wide_ok(dt, q2) && gemm_route(dt, q2) == ROUTE_WIDE) { out = name(q2); return out.c_str(); }
  out = "layernorm+" + name(p);
  return out.c_str();
}

int tango_op_linear_qkv(int dt, const float* x, const float* w, const float* gamma, const float* beta, float* out_qk, float* out_vt,
                        int B, int S, int C, int K, float eps, void* stream) {
  return tango_op_linear_qkv_perm(dt, x, w, gamma, beta, out_qk, out_vt, B, S, C, K, eps, 0, stream);
}

int tango_op_linear_qkv_perm(int dt, const float* x, const float* w, const float* gamma, const float* beta, float* out_qk, float* out_vt,
                             int B, int S, int C, int K, float eps, int vt_perm, void* stream) {
  // vt_perm = 1: out_vt with the tokens of every block of 32 in the attention kernel's fragment order (GemmParams::vt_perm; 16-bit engines, S % 32 == 0)
  // the self-attention projection of the engine: [LayerNorm](x [B*S, K]) @ Wqkv^T [3C, K] (no bias); q | k -> out_qk [B*S, 2C],
  // v -> out_vt [B][C][S] (EPI_VT).  gamma == nullptr: no LayerNorm.  Folded LN when a kernel takes it, else LN + GEMM.
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype_size(dt);
  Scratch sc;
  const int M = B * S, N = 3 * C;
  void* xt = sc.get((size_t)M * K * esz);
  void* wt = sc.get((size_t)N * K * esz);
  void* wl = sc.get((size_t)N * K * esz);
  void* qk = sc.get((size_t)M * 2 * C * esz);
  void* vt = sc.get((size_t)B * C * S * esz);
  float* bl = (float*)sc.get((size_t)N * 4);
  float* ws = (float*)sc.get((size_t)N * 4);
  if (!xt || !wt || !wl || !qk || !vt || !bl || !ws) TANGO_FAIL("op_linear_qkv: alloc");
  TANGO_TRY(launch_cast_rows(dt, x, xt, K, M, K, s));
  TANGO_TRY(launch_pack(dt, w, wt, N, 1, K, K, 0, 1, K, 0, s));
  GemmParams p;
  p.A = xt; p.lda = K; p.W = wt; p.Kp = K; p.bias = nullptr; p.M = M; p.N = N; p.K = K; p.Cin = K;
  p.mode = GATHER_1D; p.rows_pb = M; p.Lin = M; p.Lout = M;
  p.out = qk; p.ldo = 2 * C; p.epi = EPI_VT; p.vt = vt; p.vt_n0 = 2 * C; p.vt_S = S; p.vt_ld = S; p.vt_perm = vt_perm;
  TANGO_TRY(gemm_init());
  if (gamma) {
    TANGO_TRY(launch_fold_ln(dt, wt, K, gamma, beta, nullptr, wl, bl, ws, N, K, s));
    GemmParams q = p;
    q.W = wl; q.bias = bl; q.ln_fold = 1; q.ln_eps = eps; q.wsum = ws;
    if (gemm_ln_fold_ok(dt, q)) {
      TANGO_TRY(launch_gemm(dt, q, s));
    } else {
      void* nt = sc.get((size_t)M * K * esz);
      if (!nt) TANGO_FAIL("op_linear_qkv: alloc");
      TANGO_TRY(launch_layernorm(dt, xt, K, nt, K, gamma, beta, M, K, eps, s));
      p.A = nt;
      TANGO_TRY(launch_gemm(dt, p, s));
    }
  } else {
    TANGO_TRY(launch_gemm(dt, p, s));
  }
  TANGO_TRY(to_f32(dt, qk, 2 * C, out_qk, M, 2 * C, s));
  TANGO_TRY(to_f32(dt, vt, S, out_vt, B * C, S, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

int tango_op_ff_fused(int dt, const float* x, const float* w1, const float* b1, const float* gamma, const float* beta, const float* w2,
                      const float* b2, float* out, int M, int C, int H, float eps, int mode, int reps, float* ms_out, void* stream) {
  // out = x + ff.net.2(GEGLU(ff.net.0.proj(LayerNorm(x)))) as the engine runs the level-0 feed-forward.  mode 0: ff_fused.hip (one launch);
  // mode 1: the two-GEMM route (LayerNorm-folded GEGLU projection + ff.net.2 with the residual epilogue).  reps > 0 and ms_out: the mean
  // time of `reps` back-to-back repeats of the op's launches (HIP events on `stream`), for same-process A/Bs.
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype_size(dt);
  Scratch sc;
  const int N1 = 2 * H;
  void* xt = sc.get((size_t)M * C * esz);
  void* w1t = sc.get((size_t)N1 * C * esz);
  void* w1l = sc.get((size_t)N1 * C * esz);
  void* w2t = sc.get((size_t)C * H * esz);
  void* ot = sc.get((size_t)M * C * esz);
  void* gg = mode == 1 ? sc.get((size_t)M * H * esz) : nullptr;
  float* b1t = (float*)sc.get((size_t)N1 * 4);
  float* b1l = (float*)sc.get((size_t)N1 * 4);
  float* ws = (float*)sc.get((size_t)N1 * 4);
  float* b2t = (float*)sc.get((size_t)C * 4);
  if (!xt || !w1t || !w1l || !w2t || !ot || !b1t || !b1l || !ws || !b2t || (mode == 1 && !gg)) TANGO_FAIL("op_ff_fused: alloc");
  TANGO_TRY(launch_cast_rows(dt, x, xt, C, M, C, s));
  TANGO_TRY(launch_pack(dt, w1, w1t, N1, 1, C, C, 0, 1, C, -1, s));
  TANGO_TRY(launch_pack(dt, w2, w2t, C, 1, H, H, 0, 1, H, 0, s));
  TANGO_TRY(launch_permute_geglu_bias(b1, b1t, N1, s));
  TANGO_HIP(hipMemcpyAsync(b2t, b2, (size_t)C * 4, hipMemcpyDeviceToDevice, s));
  TANGO_TRY(launch_fold_ln(dt, w1t, C, gamma, beta, b1t, w1l, b1l, ws, N1, C, s));
  TANGO_TRY(gemm_init());
  FFParams f;
  f.x = xt; f.ldx = C; f.w1 = w1l; f.ld1 = C; f.b1 = b1l; f.w2 = w2t; f.ld2 = H; f.b2 = b2t; f.out = ot; f.ldo = C;
  f.M = M; f.C = C; f.H = H; f.eps = eps;
  GemmParams p1, p2;
  p1.A = xt; p1.lda = C; p1.W = w1l; p1.Kp = C; p1.bias = b1l; p1.M = M; p1.N = N1; p1.K = C; p1.Cin = C;
  p1.mode = GATHER_1D; p1.rows_pb = M; p1.Lin = M; p1.Lout = M; p1.out = gg; p1.ldo = H; p1.epi = EPI_GEGLU;
  p1.ln_fold = 1; p1.ln_eps = eps; p1.wsum = ws;
  p2.A = gg; p2.lda = H; p2.W = w2t; p2.Kp = H; p2.bias = b2t; p2.M = M; p2.N = C; p2.K = H; p2.Cin = H;
  p2.mode = GATHER_1D; p2.rows_pb = M; p2.Lin = M; p2.Lout = M; p2.out = ot; p2.ldo = C; p2.R = xt; p2.ldr = C;
  if (mode == 1 && !gemm_ln_fold_ok(dt, p1)) TANGO_FAIL("op_ff_fused: mode 1 needs a LayerNorm-folding GEMM for this shape");
  auto once = [&]() -> int {
    if (mode == 0) return launch_ff_fused(dt, f, s);
    TANGO_TRY(launch_gemm(dt, p1, s));
    return launch_gemm(dt, p2, s);
  };
  TANGO_TRY(once());
  if (reps > 0 && ms_out) {
    hipEvent_t e0, e1;
    TANGO_HIP(hipEventCreate(&e0));
    TANGO_HIP(hipEventCreate(&e1));
    TANGO_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) TANGO_TRY(once());
    TANGO_HIP(hipEventRecord(e1, s));
    TANGO_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    TANGO_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / (float)reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  TANGO_TRY(to_f32(dt, ot, C, out, M, C, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

int tango_op_qkv_stat(int dt, const float* x, const float* w, const float* gamma, const float* beta, float* out_qk, float* out_vt, int B, int S,
                      int C, float eps, int mode, int reps, float* ms_out, void* stream) {
  // LayerNorm(x [B*S, C]) @ Wqkv^T [3C, C] (no bias): q | k -> out_qk [B*S, 2C], v -> out_vt [B][C][S].  mode 0: the activation-stationary kernel
  // (ff_fused.hip qkv_stat_kernel; C = 320), mode 1: the GEMM route the dispatcher picks (folded LayerNorm, EPI_VT).  reps / ms_out as tango_op_ff_fused.
  // mode | 2: out_vt with the tokens of every block of 32 in the attention kernel's fragment order (vt_perm; S % 32 == 0).
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype_size(dt);
  Scratch sc;
  const int M = B * S, N = 3 * C, K = C;
  void* xt = sc.get((size_t)M * K * esz);
  void* wt = sc.get((size_t)N * K * esz);
  void* wl = sc.get((size_t)N * K * esz);
  void* qk = sc.get((size_t)M * 2 * C * esz);
  void* vt = sc.get((size_t)B * C * S * esz);
  float* bl = (float*)sc.get((size_t)N * 4);
  float* ws = (float*)sc.get((size_t)N * 4);
  if (!xt || !wt || !wl || !qk || !vt || !bl || !ws) TANGO_FAIL("op_qkv_stat: alloc");
  TANGO_TRY(launch_cast_rows(dt, x, xt, K, M, K, s));
  TANGO_TRY(launch_pack(dt, w, wt, N, 1, K, K, 0, 1, K, 0, s));
  TANGO_TRY(launch_fold_ln(dt, wt, K, gamma, beta, nullptr, wl, bl, ws, N, K, s));
  TANGO_TRY(gemm_init());
  QKVParams q;
  q.x = xt; q.ldx = K; q.w = wl; q.ldw = K; q.b = bl; q.out = qk; q.ldo = 2 * C; q.vt = vt; q.vt_ld = S; q.vt_S = S;
  q.M = M; q.N = N; q.K = K; q.n_rm = 2 * C; q.ln = 1; q.eps = eps;
  const int perm = (mode & 2) ? 1 : 0;
  mode &= 1;
  q.vt_perm = perm;
  GemmParams g;
  g.A = xt; g.lda = K; g.W = wl; g.Kp = K; g.bias = bl; g.M = M; g.N = N; g.K = K; g.Cin = K;
  g.mode = GATHER_1D; g.rows_pb = M; g.Lin = M; g.Lout = M;
  g.out = qk; g.ldo = 2 * C; g.epi = EPI_VT; g.vt = vt; g.vt_n0 = 2 * C; g.vt_S = S; g.vt_ld = S; g.vt_perm = perm;
  g.ln_fold = 1; g.ln_eps = eps; g.wsum = ws;
  if (mode == 1 && !gemm_ln_fold_ok(dt, g)) TANGO_FAIL("op_qkv_stat: mode 1 needs a LayerNorm-folding GEMM for this shape");
  auto once = [&]() -> int { return mode == 0 ? launch_qkv_stat(dt, q, s) : launch_gemm(dt, g, s); };
  TANGO_TRY(once());
  if (reps > 0 && ms_out) {
    hipEvent_t e0, e1;
    TANGO_HIP(hipEventCreate(&e0));
    TANGO_HIP(hipEventCreate(&e1));
    TANGO_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) TANGO_TRY(once());
    TANGO_HIP(hipEventRecord(e1, s));
    TANGO_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    TANGO_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / (float)reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  TANGO_TRY(to_f32(dt, qk, 2 * C, out_qk, M, 2 * C, s));
  TANGO_TRY(to_f32(dt, vt, S, out_vt, B * C, S, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

int tango_op_conv1d(int dt, const float* x, const float* w, const float* bias, const float* residual, float* out, int B, int Cin,
                    int L, int Cout, int k, int dilation, int a_act, float a_slope, int e_act, float e_slope, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype_size(dt);
  Scratch sc;
  void* xt = sc.get((size_t)B * L * Cin * esz);
  void* wt = sc.get((size_t)Cout * k * Cin * esz);
  void* ot = sc.get((size_t)B * L * Cout * esz);
  void* rt = residual ? sc.get((size_t)B * L * Cout * esz) : nullptr;
  if (!xt || !wt || !ot) TANGO_FAIL("op_conv1d: alloc");
  TANGO_TRY(launch_nchw_to_nhwc(dt, x, xt, Cin, B, Cin, L, 1, 1.0f, s));
  if (residual) TANGO_TRY(launch_nchw_to_nhwc(dt, residual, rt, Cout, B, Cout, L, 1, 1.0f, s));
  TANGO_TRY(launch_pack(dt, w, wt, Cout, k, Cin, (int64_t)Cin * k, 1, k, (int64_t)k * Cin, 0, s));
  GemmParams p;
  p.A = xt; p.lda = Cin; p.W = wt; p.Kp = (int64_t)k * Cin; p.bias = bias; p.M = B * L; p.N = Cout; p.K = k * Cin; p.Cin = Cin;
  p.mode = GATHER_1D; p.rows_pb = L; p.Lin = L; p.taps = k; p.tap_step = dilation; p.in_off = -dilation * (k - 1) / 2;
  p.Lout = L; p.out = ot; p.ldo = Cout; p.R = rt; p.ldr = Cout;
  p.a_act = a_act; p.a_slope = a_slope; p.e_act = e_act; p.e_slope = e_slope;
  TANGO_TRY(run_gemm(dt, p, sc, s));
  TANGO_TRY(launch_nhwc_to_nchw_f32(dt, ot, Cout, out, B, Cout, L, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

int tango_op_conv_transpose1d(int dt, const float* x, const float* w, const float* bias, float* out, int B, int Cin, int L,
                              int Cout, int k, int u, int pd, int a_act, float a_slope, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype_size(dt);
  Scratch sc;
  const int Lo = (L - 1) * u - 2 * pd + k;
  void* xt = sc.get((size_t)B * L * Cin * esz);
  void* ot = sc.get((size_t)B * Lo * Cout * esz);
  if (!xt || !ot) TANGO_FAIL("op_convt1d: alloc");
  TANGO_TRY(launch_nchw_to_nhwc(dt, x, xt, Cin, B, Cin, L, 1, 1.0f, s));
  for (int r = 0; r < u; ++r) {
    const int T = (k - r + u - 1) / u;
    if (T <= 0) continue;
    void* wt = sc.get((size_t)Cout * T * Cin * esz);
    if (!wt) TANGO_FAIL("op_convt1d: alloc");
    TANGO_TRY(launch_pack(dt, w + r, wt, Cout, T, Cin, k, u, (int64_t)Cout * k, (int64_t)T * Cin, 0, s));
    const int qmin = (pd > r) ? (pd - r + u - 1) / u : 0;
    const int qmax = (Lo - 1 + pd - r) / u;
    const int Q = qmax - qmin + 1;
    if (Q <= 0) continue;
    GemmParams p;
    p.A = xt; p.lda = Cin; p.W = wt; p.Kp = (int64_t)T * Cin; p.bias = bias; p.M = B * Q; p.N = Cout; p.K = T * Cin; p.Cin = Cin;
    p.mode = GATHER_1D; p.rows_pb = Q; p.Lin = L; p.taps = T; p.tap_step = -1; p.in_off = qmin;
    p.Lout = Lo; p.out_mul = u; p.out_off = u * qmin + r - pd; p.out = ot; p.ldo = Cout;
    p.a_act = a_act; p.a_slope = a_slope;
    TANGO_TRY(run_gemm(dt, p, sc, s));
  }
  TANGO_TRY(launch_nhwc_to_nchw_f32(dt, ot, Cout, out, B, Cout, Lo, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

int tango_op_groupnorm(int dt, const float* x, const float* gamma, const float* beta, float* out, int B, int C, int HW, int groups,
                       float eps, int act, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype_size(dt);
  Scratch sc;
  void* xt = sc.get((size_t)B * HW * C * esz);
  void* yt = sc.get((size_t)B * HW * C * esz);
  const size_t nf = groupnorm_ws_floats(B, HW, C, groups);
  float* ws = (float*)sc.get(nf * 4);
  if (!xt || !yt || !ws) TANGO_FAIL("op_groupnorm: alloc");
  TANGO_TRY(launch_nchw_to_nhwc(dt, x, xt, C, B, C, HW, 1, 1.0f, s));
  GroupNormParams p;
  p.x = xt; p.ldx = C; p.y = yt; p.ldy = C; p.gamma = gamma; p.beta = beta; p.B = B; p.rows = HW; p.C = C; p.groups = groups;
  p.eps = eps; p.act = act; p.partial = ws; p.scale_shift = ws + (nf - (size_t)B * C * 2);
  // barrier words of the cooperative kernel: one zeroed buffer per process for these single-stream test entry points
  static unsigned* op_sync = nullptr;
  if (!op_sync) {
    TANGO_HIP(hipMalloc((void**)&op_sync, (size_t)coop_sync_words() * 4));
    TANGO_HIP(hipMemset(op_sync, 0, (size_t)coop_sync_words() * 4));
  }
  p.sync = op_sync;
  TANGO_TRY(launch_groupnorm(dt, p, s));
  {
    unsigned flag = 0;
    TANGO_HIP(hipMemcpyAsync(&flag, op_sync + 2 * COOP_SYNC_SLOTS, 4, hipMemcpyDeviceToHost, s));
    TANGO_HIP(hipStreamSynchronize(s));
    if (flag) {
      // workgroups took the no-rendezvous fallback (results stay valid).  On the idle GPU of a test that only happens on request;
      // otherwise the barrier words were left in a bad state by an earlier launch, which is what this entry point exists to catch
      TANGO_HIP(hipMemsetAsync(op_sync + 2 * COOP_SYNC_SLOTS, 0, 4, s));
      if (!tuning().gn_coop_force_fb) TANGO_FAIL("op_groupnorm: the cooperative kernel gave up waiting at its rendezvous on an idle GPU");
    }
  }
  TANGO_TRY(launch_nhwc_to_nchw_f32(dt, yt, C, out, B, C, HW, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

int tango_op_layernorm(int dt, const float* x, const float* gamma, const float* beta, float* out, int rows, int C, float eps,
                       void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype_size(dt);
  Scratch sc;
  void* xt = sc.get((size_t)rows * C * esz);
  void* yt = sc.get((size_t)rows * C * esz);
  if (!xt || !yt) TANGO_FAIL("op_layernorm: alloc");
  TANGO_TRY(launch_cast_rows(dt, x, xt, C, rows, C, s));
  TANGO_TRY(launch_layernorm(dt, xt, C, yt, C, gamma, beta, rows, C, eps, s));
  TANGO_TRY(to_f32(dt, yt, C, out, rows, C, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

int tango_op_attention(int dt, const float* q, const float* k, const float* v, const float* bias, float* out, int B, int heads,
                       int Sq, int Skv, float scale, void* stream) {
  return tango_op_attention_ex(dt, q, k, v, bias, out, B, heads, Sq, Skv, scale, 0, stream);
}

int tango_op_attention_ex(int dt, const float* q, const float* k, const float* v, const float* bias, float* out, int B, int heads,
                          int Sq, int Skv, float scale, int flags, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype_size(dt);
  Scratch sc;
  const int C = heads * 64;
  void* qt = sc.get((size_t)B * Sq * C * esz);
  void* kt = sc.get((size_t)B * Skv * C * esz);
  void* vt = sc.get((size_t)B * Skv * C * esz);
  void* ot = sc.get((size_t)B * Sq * C * esz);
  const int64_t ldvt = (Skv + 7) / 8 * 8;
  void* vtt = sc.get((size_t)B * C * ldvt * esz);
  if (!qt || !kt || !vt || !ot || !vtt) TANGO_FAIL("op_attention: alloc");
  TANGO_HIP(hipMemsetAsync(vtt, 0, (size_t)B * C * ldvt * esz, s));
  TANGO_TRY(launch_cast_rows(dt, q, qt, C, B * Sq, C, s));
  TANGO_TRY(launch_cast_rows(dt, k, kt, C, B * Skv, C, s));
  TANGO_TRY(launch_cast_rows(dt, v, vt, C, B * Skv, C, s));
  TANGO_TRY(transpose_v(dt, vt, vtt, B, Skv, C, ldvt, s));
  AttnParams p;
  p.q = qt; p.ldq = C; p.k = kt; p.ldk = C; p.vt = vtt; p.ldvt = ldvt; p.o = ot; p.ldo = C; p.bias = bias;
  p.B = B; p.heads = heads; p.Sq = Sq; p.Skv = Skv; p.scale = scale;
  if (flags & 1) {
    if (bias || Skv % 64 != 0) TANGO_FAIL("op_attention_ex: the fp8 P.V path takes unmasked problems with Skv % 64 == 0");
    p.fp8_pv = 1;
    if (flags & 2) {
      if (Skv % 128 != 0) TANGO_FAIL("op_attention_ex: the MX fp8 P.V path takes Skv % 128 == 0");
      p.fp8_pv = 2;
    }
  }
  // bit 2: the rows of v arrive in the attention kernel's fragment order inside every block of 32 keys (row vt_perm_pos(s) holds key s): the V^T this
  // op builds is then a vt_perm one and the kernel fetches K and V^T tiles by LDS-DMA (what the engine runs at its Sq > 512 self-attention sites)
  if (flags & 4) p.vt_perm = 1;
  TANGO_TRY(launch_attention(dt, p, s));
  TANGO_TRY(to_f32(dt, ot, C, out, (int64_t)B * Sq, C, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

int tango_op_xattn_block(int dt, const float* x, const float* gamma, const float* beta, const float* wq, const float* k, const float* v,
                         const float* bias, const float* wo, const float* bo, float* out, int B, int HW, int L, float eps, void* stream) {
  // y = x + to_out(softmax(to_q(LayerNorm(x)) K^T / 8 + bias) V) + bo in the fused kernel (xattn.hip); C = 320, 5 heads.
  // x [B*HW, 320]; wq, wo [320, 320] (Linear layout, to_q has no bias); k, v [B*L, 320] = to_k / to_v of the text; bias [B, L] or NULL
  hipStream_t s = (hipStream_t)stream;
  const int C = 320, heads = 5;
  const size_t esz = dtype_size(dt);
  const int64_t M = (int64_t)B * HW, ldvt = (L + 7) / 8 * 8;
  if (!xattn_block_ok(dt, C, heads, HW, L, C, C, C, ldvt)) TANGO_FAIL("op_xattn_block: shape / dtype not supported by the fused kernel");
  Scratch sc;
  void* xt = sc.get(M * C * esz); void* ot = sc.get(M * C * esz);
  void* wqt = sc.get((size_t)C * C * esz); void* wql = sc.get((size_t)C * C * esz); void* wqp = sc.get((size_t)C * C * esz);
  void* wot = sc.get((size_t)C * C * esz);
  void* kt = sc.get((size_t)B * L * C * esz); void* vt = sc.get((size_t)B * L * C * esz); void* vtt = sc.get((size_t)B * C * ldvt * esz);
  float* bl = (float*)sc.get(C * 4); float* ws = (float*)sc.get(C * 4); float* bp = (float*)sc.get(C * 4); float* wsp = (float*)sc.get(C * 4);
  if (!xt || !ot || !wqt || !wql || !wqp || !wot || !kt || !vt || !vtt || !bl || !ws || !bp || !wsp) TANGO_FAIL("op_xattn_block: alloc");
  TANGO_HIP(hipMemsetAsync(vtt, 0, (size_t)B * C * ldvt * esz, s));
  TANGO_TRY(launch_cast_rows(dt, x, xt, C, (int)M, C, s));
  TANGO_TRY(launch_pack(dt, wq, wqt, C, 1, C, C, 0, 1, C, 0, s));
  TANGO_TRY(launch_pack(dt, wo, wot, C, 1, C, C, 0, 1, C, 0, s));
  TANGO_TRY(launch_cast_rows(dt, k, kt, C, B * L, C, s));
  TANGO_TRY(launch_cast_rows(dt, v, vt, C, B * L, C, s));
  TANGO_TRY(transpose_v(dt, vt, vtt, B, L, C, ldvt, s));
  TANGO_TRY(launch_fold_ln(dt, wqt, C, gamma, beta, nullptr, wql, bl, ws, C, C, s));
  TANGO_TRY(launch_xattn_permute_wq(dt, wql, bl, ws, wqp, bp, wsp, C, s));
  XAttnParams p;
  p.x = xt; p.ldx = C; p.wq = wqp; p.bq = bp; p.wsum = wsp; p.k = kt; p.ldk = C; p.vt = vtt; p.ldvt = ldvt; p.bias = bias;
  p.wo = wot; p.ldwo = C; p.bo = bo; p.out = ot; p.ldo = C; p.M = (int)M; p.HW = HW; p.L = L; p.eps = eps; p.scale = 0.125f;
  TANGO_TRY(launch_xattn_block(dt, p, s));
  TANGO_TRY(to_f32(dt, ot, C, out, M, C, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

int tango_op_sched_step(float* latents, const float* model_out_nchw, const float* noise, const float* coef8, int B, int C, int HW,
                        int cfg, float guidance, int pred_type, int rule, int clip, float clip_range, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Scratch sc;
  const int B2 = cfg ? 2 * B : B;
  float* eps = (float*)sc.get((size_t)B2 * HW * C * 4);
  float* xin = (float*)sc.get((size_t)B2 * HW * 8 * 4);
  float* dcoef = (float*)sc.get(8 * 4);
  int* dstep = (int*)sc.get(256);
  if (!eps || !xin || !dcoef || !dstep) TANGO_FAIL("op_sched_step: alloc");
  TANGO_TRY(launch_nchw_to_nhwc(DT_F32, model_out_nchw, eps, C, B2, C, HW, 1, 1.0f, s));
  TANGO_HIP(hipMemcpyAsync(dcoef, coef8, 32, hipMemcpyHostToDevice, s));
  TANGO_HIP(hipMemsetAsync(dstep, 0, 4, s));
  SchedParams p;
  p.lat = latents; p.eps = eps; p.xin = xin; p.xin_ld = 8; p.noise = noise; p.coef = dcoef; p.step_ptr = dstep;
  p.B = B; p.C = C; p.HW = HW; p.cfg = cfg; p.guidance = guidance; p.pred_type = pred_type; p.rule = rule; p.clip = clip;
  p.clip_range = clip_range; p.seed = 0; p.sample_offset = 0;
  SchedParams* dp = (SchedParams*)sc.get(sizeof(SchedParams));
  if (!dp) TANGO_FAIL("op_sched_step: alloc");
  TANGO_HIP(hipMemcpyAsync(dp, &p, sizeof(SchedParams), hipMemcpyHostToDevice, s));
  TANGO_TRY(launch_sched_step(DT_F32, dp, B * HW, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

int tango_op_philox_normal(float* out, int B, int C, int HW, int step, uint64_t seed, int sample_offset, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  TANGO_TRY(launch_philox_normal(out, B, C, HW, step, seed, sample_offset, s));
  TANGO_HIP(hipStreamSynchronize(s));
  return 0;
}

}  // extern "C"
