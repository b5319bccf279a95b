// Host side of the Tango MI355X engine: weight ingestion/packing, static execution plans for the
// UNet / mel-VAE decoder / HiFi-GAN, the denoise loop with a hipGraph-captured UNet step, C ABI.
//
// Data layout in HBM: every activation is channels-last ([B, H, W, C] / [B, L, C]) in the engine
// dtype with an explicit row stride, so tokens == pixels (no NCHW<->token transposes,
// transformer_2d.py:255-300) and skip-connection concats (unet_2d_blocks.py:2235) are free: producers
// write straight into column slices of the pre-allocated concat buffer.
#include "engine.h"
#include "tuning.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace tango {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* last_error() { return g_err.c_str(); }

// ================================================================================================
// Builder: records kernel launches into a Program
// ================================================================================================
struct GOpt {
  const float* bias2 = nullptr;
  int bias2_stride = 0;
  const TView* residual = nullptr;
  int a_act = ACT_NONE;
  float a_slope = 0.f;
  int e_act = ACT_NONE;
  float e_slope = 0.f;
  int epi = EPI_NONE;
  bool out_f32 = false;
  bool use_bias = true;
  // EPI_VT (fused QKV / KV projections): columns >= vt_n0 go transposed into `vt` [B][C][vt_ld]
  void* vt = nullptr;
  int vt_n0 = 0, vt_S = 1;
  int64_t vt_ld = 0;
  int vt_perm = 0;               // the transposed columns in the attention kernel's fragment order (GemmParams::vt_perm)
  const WNorm* ln = nullptr;     // apply LayerNorm(ln) to the input rows first (folded when the streaming kernel applies)
  int glu_tanh = 0;              // EPI_GEGLU gate: tanh GELU (T5 gated-gelu) instead of exact-erf GELU
  int pad = 1;                   // conv3x3: top / left zero padding (0 = the VAE Downsample's asymmetric (0,1,0,1) pad)
  // linear(): rows < rowvec_rows also get rowvec[(row / rowvec_per)][n] and go to out_lo (GemmParams::rowvec); *rowvec_done tells
  // the caller whether the kernel that takes the problem can do it (else the caller keeps its separate pass)
  const float* rowvec = nullptr;
  int64_t rowvec_rows = 0;
  int rowvec_per = 1;
  const TView* out_lo = nullptr;
  bool* rowvec_done = nullptr;
};

struct Builder {
  Engine& E;
  Arena& A;
  Program* prog;
  bool record;
  int dt;
  size_t esz;
  unsigned* sync = nullptr;   // barrier words for cooperative kernels (null: the engine's first set)

  TView alloc(int64_t rows, int C) {
    TView t;
    t.p = A.alloc((size_t)rows * C * esz);
    t.ld = C;
    t.C = C;
    return t;
  }
  float* alloc_f32(size_t n) { return (float*)A.alloc(n * 4); }
  static TView slice(const TView& t, int c0, int C, size_t esz) {
    TView r;
    r.p = (char*)t.p + (size_t)c0 * esz;
    r.ld = t.ld;
    r.C = C;
    return r;
  }
  std::string scope;   // label prefix of the module being built
  void push(Op op, const std::string& label = "op", double fl = 0.0) {
    if (record) {
      prog->ops.push_back(std::move(op));
      prog->labels.push_back(scope + label);
      prog->flops.push_back(fl);
    }
  }
  void gemm(GemmParams p, const char* what = "gemm") {
    const int d = dt;
    const int sk = gemm_pick_splitk(d, p);
    if (sk > 1) {   // the workspace may alias later temporaries: the whole program is stream-ordered
      const size_t m = A.mark();
      p.splitk = sk;
      p.ws = alloc_f32((size_t)sk * p.M * p.N);
      A.release(m);
    }
    char buf[160];
    snprintf(buf, sizeof buf, "%s M=%d N=%d K=%d%s%s", what, p.M, p.N, p.K, p.batch > 1 ? " batched" : "", sk > 1 ? " splitK" : "");
    push([p, d](hipStream_t s) { return launch_gemm(d, p, s); }, buf, 2.0 * p.M * (double)p.N * p.K * p.batch);
  }


  // out[rows, N] = x[rows, K] @ W^T
  void linear(const TView& x, int64_t rows, const WMat& w, const TView& out, const GOpt& o = GOpt()) {
    GemmParams p;
    p.A = x.p; p.lda = x.ld; p.W = w.W; p.Kp = w.Kp;
    p.bias = o.use_bias ? w.b : nullptr;
    p.M = (int)rows; p.N = w.N; p.K = w.K; p.Cin = w.K;
    p.mode = GATHER_1D; p.rows_pb = (int)rows; p.Lin = (int)rows; p.Lout = (int)rows; p.taps = 1;
    p.out = out.p; p.ldo = out.ld; p.out_f32 = o.out_f32;
    if (o.residual) { p.R = o.residual->p; p.ldr = o.residual->ld; }
    p.a_act = o.a_act; p.a_slope = o.a_slope; p.e_act = o.e_act; p.e_slope = o.e_slope; p.epi = o.epi; p.glu_tanh = o.glu_tanh;
    p.bias2 = o.bias2; p.bias2_stride = o.bias2_stride; p.step_ptr = o.bias2 ? E.d_step : nullptr;
    if (o.vt) { p.epi = EPI_VT; p.vt = o.vt; p.vt_n0 = o.vt_n0; p.vt_S = o.vt_S; p.vt_ld = o.vt_ld; p.vt_perm = o.vt_perm; }
    if (o.ln) {
      GemmParams q = p;
      q.W = w.Wln; q.bias = w.bln; q.ln_fold = 1; q.ln_eps = o.ln->eps; q.wsum = w.wsum;
      if (w.Wln && gemm_ln_fold_ok(dt, q)) {
        const int rq = gemm_route(dt, q);
        gemm(q, rq == ROUTE_WIDE ? "linear+ln(wide)" : rq == ROUTE_DUO ? "linear+ln(duo)" : "linear+ln(stream)");
        return;
      }
      // GEGLU projections the in-loop-statistics kernels do not take (levels 1-2): row statistics in a read-only pass, then the
      // folded weights on the 256 x 320 GEMM -- LayerNorm(x) is never written (round 4)
      if (w.Wln && p.epi == EPI_GEGLU && !tuning().no_ln_xstats) {
        const size_t m = A.mark();
        float* st = alloc_f32((size_t)rows * 2);
        GemmParams q2 = q;
        q2.row_stats = st;
        if (gemm_wide_ok(dt, q2) && gemm_route(dt, q2) == ROUTE_WIDE) {
          const int d = dt;
          const void* xp = x.p; const int64_t ldx = x.ld; const int r = (int)rows, c = o.ln->C; const float eps = o.ln->eps;
          push([=](hipStream_t s) { return launch_ln_stats(d, xp, ldx, st, r, c, eps, s); }, "ln_stats C=" + std::to_string(c));
          gemm(q2, "linear+ln(xstats)");
          A.release(m);
          return;
        }
        A.release(m);
      }
      // fallback: materialise LayerNorm(x), then the plain GEMM
      const size_t m = A.mark();
      TView t = alloc(rows, o.ln->C);
      layernorm(x, rows, *o.ln, t);
      p.A = t.p; p.lda = t.ld;
      gemm(p, "linear");
      A.release(m);
      return;
    }
    if (o.rowvec && o.rowvec_rows > 0 && !tuning().no_rowvec_fuse) {
      GemmParams q = p;
      q.rowvec = o.rowvec; q.rowvec_rows = o.rowvec_rows; q.rowvec_per = o.rowvec_per;
      q.out_lo = o.out_lo ? o.out_lo->p : p.out; q.ldo_lo = o.out_lo ? o.out_lo->ld : p.ldo;
      if (gemm_rowvec_ok(dt, q)) {
        p = q;
        if (o.rowvec_done) *o.rowvec_done = true;
      }
    }
    const int route = gemm_pick_splitk(dt, p) > 1 ? ROUTE_TILE : gemm_route(dt, p);
    gemm(p, route == ROUTE_STREAM ? "linear(stream)" : route == ROUTE_WIDE ? "linear(wide)" : route == ROUTE_DUO ? "linear(duo)" : "linear");
  }

  // 3x3 conv (pad 1) on NHWC: output grid B x H x W; source B x Hin x Win (nearest-upsampled x2 when ups)
  void conv3x3(const TView& x, int B, int H, int W, int Hin, int Win, int stride, int ups, const WMat& w, const TView& out,
               const GOpt& o = GOpt()) {
    if (w.im2col) {
      // tiny Cin: materialise im2col rows then plain GEMM
      const size_t m = A.mark();
      TView col = alloc((int64_t)B * H * W, (int)w.Kp);
      const int d = dt;
      const void* src = x.p; const int64_t ld = x.ld; void* dst = col.p; const int64_t Kp = w.Kp; const int Cs = w.Cin;
      push([=](hipStream_t s) { return launch_im2col3x3(d, src, ld, dst, Kp, B, H, W, Cs, s); });
      WMat lw = w; lw.K = (int)w.Kp; lw.im2col = false;
      linear(col, (int64_t)B * H * W, lw, out, o);
      A.release(m);
      return;
    }
    GemmParams p;
    p.A = x.p; p.lda = x.ld; p.W = w.W; p.Kp = w.Kp; p.bias = o.use_bias ? w.b : nullptr;
    p.M = B * H * W; p.N = w.N; p.K = w.K; p.Cin = w.Cin;
    p.mode = GATHER_2D; p.H = H; p.Wd = W; p.Hin = Hin; p.Win = Win; p.stride = stride; p.ups = ups; p.pad = o.pad;
    p.out = out.p; p.ldo = out.ld; p.out_f32 = o.out_f32;
    if (o.residual) { p.R = o.residual->p; p.ldr = o.residual->ld; }
    p.a_act = o.a_act; p.a_slope = o.a_slope; p.e_act = o.e_act; p.e_slope = o.e_slope; p.epi = o.epi;
    p.bias2 = o.bias2; p.bias2_stride = o.bias2_stride; p.step_ptr = o.bias2 ? E.d_step : nullptr;
    gemm(p, stride == 2 ? "conv3x3s2" : (ups ? "conv3x3up" : "conv3x3"));
  }

  // GroupNorm folded into the linear that consumes it (round 6; norm.hip gn_fold_kernel): out = proj(GroupNorm(x)) without the normalised
  // tensor -- statistics pass, per-sample folded weights + bias vector, then the 256 x 320 / 256 x 160 GEMM on the RAW rows (GemmParams::wb_*
  // per-sample weights, ::rowvec per-sample bias).  Returns false (nothing emitted) where the shape / route does not allow it.
  bool linear_gn_fold(const TView& x, int B, int rows, const WNorm& gn, int groups, const WMat& w, const TView& out) {
    GroupNormParams gp;
    gp.x = x.p; gp.ldx = x.ld; gp.y = nullptr; gp.ldy = 0; gp.gamma = gn.g; gp.beta = gn.b;
    gp.B = B; gp.rows = rows; gp.C = gn.C; gp.groups = groups; gp.eps = gn.eps; gp.act = ACT_NONE;
    gp.sync = nullptr;
    if (w.K != gn.C || !gn_fold_ok(dt, gp, w.N)) return false;
    GemmParams p;
    p.A = x.p; p.lda = x.ld; p.Kp = w.Kp; p.bias = nullptr;
    p.M = B * rows; p.N = w.N; p.K = w.K; p.Cin = w.K;
    p.mode = GATHER_1D; p.rows_pb = p.M; p.Lin = p.M; p.Lout = p.M; p.taps = 1;
    p.out = out.p; p.ldo = out.ld;
    p.wb_rows = rows; p.wb_stride = (int64_t)w.N * w.Kp;
    p.rowvec_rows = p.M; p.rowvec_per = rows; p.out_lo = out.p; p.ldo_lo = out.ld;
    const size_t m = A.mark();
    void* Wf = A.alloc((size_t)B * w.N * w.Kp * esz);
    float* bf = alloc_f32((size_t)B * w.N);
    const size_t nf = groupnorm_ws_floats(B, rows, gn.C, groups);
    float* ws = alloc_f32(nf);
    gp.partial = ws; gp.scale_shift = nullptr;
    p.W = Wf; p.rowvec = bf;
    if (gemm_pick_splitk(dt, p) > 1 || !gemm_rowvec_ok(dt, p)) { A.release(m); return false; }
    const int d = dt;
    const void* Wsrc = w.W; const int64_t Kp = w.Kp; const float* bsrc = w.b; const int N = w.N;
    push([=](hipStream_t s) { return launch_gn_stats_fold(d, gp, Wsrc, Kp, bsrc, N, Wf, bf, s); },
         "groupnorm(fold) C=" + std::to_string(gn.C) + " rows=" + std::to_string(rows));
    gemm(p, gemm_route(dt, p) == ROUTE_DUO ? "linear(duo)" : "linear(wide)");
    // (the folded weights are consumed by the GEMM that follows on the same stream: the slab region may be reused behind it)
    A.release(m);
    return true;
  }

  // Transformer2DModel.norm -> proj_in at level 0 without the normalised tensor (round 6): GroupNorm statistics pass + finalisation, then the
  // activation-stationary kernel normalises the rows it loads (ff_fused.hip qkv_stat_kernel, norm mode 2; proj_in folded with gamma / beta)
  bool gn_proj_stat(const TView& x, int B, int rows, const XfW& w, int groups, const TView& out) {
    if (!tuning().gn_proj_stat || !w.proj_in.Wln || (int64_t)B * rows < tuning().qkv_min_rows) return false;
    GroupNormParams gp;
    gp.x = x.p; gp.ldx = x.ld; gp.y = nullptr; gp.ldy = 0; gp.gamma = w.gn.g; gp.beta = w.gn.b;
    gp.B = B; gp.rows = rows; gp.C = w.C; gp.groups = groups; gp.eps = w.gn.eps; gp.act = ACT_NONE; gp.sync = nullptr;
    QKVParams q;
    q.x = x.p; q.ldx = x.ld; q.w = w.proj_in.Wln; q.ldw = w.proj_in.Kp; q.b = w.proj_in.bln; q.out = out.p; q.ldo = out.ld;
    q.M = B * rows; q.N = w.proj_in.N; q.K = w.C; q.n_rm = q.N; q.ln = 2; q.groups = groups; q.rows_ps = rows; q.eps = w.gn.eps;
    q.gstats = (const float*)(uintptr_t)16;          // placeholder for the shape check
    if (!gn_stats_mr_ok(dt, gp) || !qkv_stat_ok(dt, q)) return false;
    const size_t m = A.mark();
    float* ws = alloc_f32(groupnorm_ws_floats(B, rows, w.C, groups));
    float* mr = alloc_f32((size_t)B * groups * 2);
    gp.partial = ws; gp.scale_shift = nullptr;
    q.gstats = mr;
    const int d = dt;
    push([=](hipStream_t s) { return launch_gn_stats_mr(d, gp, mr, s); }, "groupnorm(stats) C=" + std::to_string(w.C) + " rows=" + std::to_string(rows));
    char buf[96];
    snprintf(buf, sizeof buf, "linear+gn(stat) M=%d N=%d K=%d", q.M, q.N, q.K);
    push([q, d](hipStream_t s) { return launch_qkv_stat(d, q, s); }, buf, 2.0 * q.M * (double)q.N * q.K);
    A.release(m);      // (consumed by the launch that follows on the same stream: the region may be reused behind it)
    return true;
  }

  // norm1 -> [to_q | to_k | to_v^T] of the level-0 self-attention on the activation-stationary kernel (round 6; ff_fused.hip qkv_stat_kernel)
  bool qkv_stat(const TView& x, int64_t rows, const XfW& w, const TView& qk, void* vt, int HW, int vt_perm) {
    if (!tuning().qkv_stat || !w.qkv.Wln || rows < tuning().qkv_min_rows) return false;
    QKVParams q;
    q.x = x.p; q.ldx = x.ld; q.w = w.qkv.Wln; q.ldw = w.qkv.Kp; q.b = w.qkv.bln; q.out = qk.p; q.ldo = qk.ld; q.vt = vt; q.vt_ld = HW; q.vt_S = HW;
    q.M = (int)rows; q.N = 3 * w.C; q.K = w.C; q.n_rm = 2 * w.C; q.ln = 1; q.eps = w.ln1.eps; q.vt_perm = vt_perm;
    if (!qkv_stat_ok(dt, q)) return false;
    const int d = dt;
    char buf[96];
    snprintf(buf, sizeof buf, "qkv_stat M=%d N=%d K=%d", q.M, q.N, q.K);
    push([q, d](hipStream_t s) { return launch_qkv_stat(d, q, s); }, buf, 2.0 * rows * (double)q.N * q.K);
    return true;
  }

  // norm3 -> GEGLU projection -> ff.net.2 -> + residual of the level-0 transformer blocks in ONE launch (round 6; ff_fused.hip): the
  // [rows, 4C] GEGLU output is never written.  Returns false (nothing emitted) where the shape does not allow it.
  bool ff_fused(const TView& x, int64_t rows, const XfW& w, const TView& out) {
    if (!tuning().ff_fused || !w.ff1.Wln || rows < tuning().ff_min_rows) return false;
    FFParams f;
    f.x = x.p; f.ldx = x.ld; f.w1 = w.ff1.Wln; f.ld1 = w.ff1.Kp; f.b1 = w.ff1.bln; f.w2 = w.ff2.W; f.ld2 = w.ff2.Kp; f.b2 = w.ff2.b;
    f.out = out.p; f.ldo = out.ld; f.M = (int)rows; f.C = w.C; f.H = 4 * w.C; f.eps = w.ln3.eps;
    if (!ff_fused_ok(dt, f)) return false;
    const int d = dt;
    char buf[96];
    snprintf(buf, sizeof buf, "ff_fused M=%d C=%d", f.M, f.C);
    push([f, d](hipStream_t s) { return launch_ff_fused(d, f, s); }, buf, 2.0 * rows * (double)(8 * w.C) * w.C + 2.0 * rows * (double)(4 * w.C) * w.C);
    return true;
  }

  void groupnorm(const TView& x, int B, int rows, const WNorm& w, int groups, int act, const TView& out) {
    GroupNormParams p;
    p.x = x.p; p.ldx = x.ld; p.y = out.p; p.ldy = out.ld; p.gamma = w.g; p.beta = w.b;
    p.B = B; p.rows = rows; p.C = w.C; p.groups = groups; p.eps = w.eps; p.act = act;
    p.sync = sync ? sync : E.d_sync;
    const size_t m = A.mark();
    const size_t nf = groupnorm_ws_floats(B, rows, w.C, groups);
    float* ws = alloc_f32(nf);
    p.partial = ws;
    p.scale_shift = ws + (nf - (size_t)B * w.C * 2);
    const int d = dt;
    push([p, d](hipStream_t s) { return launch_groupnorm(d, p, s); }, "groupnorm C=" + std::to_string(w.C) + " rows=" + std::to_string(rows));
    // NOTE: the workspace is released immediately; the next allocation may alias it, which is
    // safe because the whole program is serialized on one stream.
    A.release(m);
  }

  void layernorm(const TView& x, int64_t rows, const WNorm& w, const TView& out) {
    const int d = dt;
    const void* xp = x.p; const int64_t ldx = x.ld; void* yp = out.p; const int64_t ldy = out.ld;
    const float* g = w.g; const float* b = w.b; const int C = w.C; const float eps = w.eps; const int r = (int)rows;
    push([=](hipStream_t s) { return launch_layernorm(d, xp, ldx, yp, ldy, g, b, r, C, eps, s); }, "layernorm C=" + std::to_string(C));
  }

  void attention(const TView& q, const TView& k, const void* vt, int64_t ldvt, const TView& o, const float* bias, int B, int heads,
                 int Sq, int Skv, float scale = 0.125f, const float* pos_bias = nullptr, int fp8_pv = 0, int vt_perm = 0) {
    AttnParams p;
    p.vt_perm = vt_perm;
    p.fp8_pv = fp8_pv;                // 0: engine dtype, 1: non-scaled fp8 MFMA, 2: MX fp8 (128 keys per MFMA)
    p.q = q.p; p.ldq = q.ld; p.k = k.p; p.ldk = k.ld; p.vt = vt; p.ldvt = ldvt; p.o = o.p; p.ldo = o.ld;
    p.bias = bias; p.pos_bias = pos_bias; p.B = B; p.heads = heads; p.Sq = Sq; p.Skv = Skv; p.scale = scale;
    const int d = dt;
    push([p, d](hipStream_t s) { return launch_attention(d, p, s); },
         "attention Sq=" + std::to_string(Sq) + " Skv=" + std::to_string(Skv) + " heads=" + std::to_string(heads),
         4.0 * B * heads * (double)Sq * Skv * 64);
  }

  // ResnetBlock2D (resnet.py:549-597) / audioldm ResnetBlock (modules.py:155-175)
  void resblock(const ResW& w, const TView& x, int B, int H, int W, int groups, const TView& out) {
    const int64_t rows = (int64_t)B * H * W;
    const size_t m = A.mark();
    TView t0 = alloc(rows, w.cin);
    groupnorm(x, B, H * W, w.n1, groups, ACT_SILU, t0);
    TView t1 = alloc(rows, w.cout);
    GOpt o1;
    if (w.has_temb) { o1.bias2 = w.temb_table; o1.bias2_stride = w.cout; }
    conv3x3(t0, B, H, W, H, W, 1, 0, w.c1, t1, o1);
    TView t2 = alloc(rows, w.cout);
    groupnorm(t1, B, H * W, w.n2, groups, ACT_SILU, t2);
    TView R = x;
    if (w.has_sc) {
      R = alloc(rows, w.cout);
      linear(x, rows, w.sc, R);
    }
    GOpt o2;
    o2.residual = &R;
    conv3x3(t2, B, H, W, H, W, 1, 0, w.c2, out, o2);
    A.release(m);
  }

  // VAE AttnBlock (modules.py:204-230): single head, d = C, scale C^-0.5; returns x + proj_out(attention(norm(x)))
  TView vae_attn(const VaeAttnW& w, const TView& h, int B, int HW, int C) {
    const int64_t rows = (int64_t)B * HW;
    TView hn = alloc(rows, C);
    groupnorm(h, B, HW, w.gn, 32, ACT_NONE, hn);
    TView qk = alloc(rows, 2 * C);
    linear(hn, rows, w.qk, qk);
    // V^T[b][c][s] = Wv[c,:] . hn[b,s,:] + bv[c]   (weights as the row operand, tokens as columns)
    TView vt = alloc((int64_t)B * C, HW);
    {
      GemmParams p;
      p.A = w.v.W; p.lda = w.v.Kp; p.W = hn.p; p.Kp = hn.ld; p.bias = w.v.b; p.bias_rows = 1;
      p.M = C; p.N = HW; p.K = C; p.Cin = C; p.mode = GATHER_1D; p.rows_pb = C; p.Lin = C; p.Lout = C; p.taps = 1;
      p.out = vt.p; p.ldo = HW; p.batch = B; p.sA = 0; p.sW = (int64_t)HW * hn.ld; p.sO = (int64_t)C * HW;
      gemm(p);
    }
    TView sc = alloc((int64_t)B * HW, HW);
    {
      GemmParams p;
      p.A = qk.p; p.lda = qk.ld; p.W = (char*)qk.p + (size_t)C * esz; p.Kp = qk.ld;
      p.M = HW; p.N = HW; p.K = C; p.Cin = C; p.mode = GATHER_1D; p.rows_pb = HW; p.Lin = HW; p.Lout = HW; p.taps = 1;
      p.out = sc.p; p.ldo = HW; p.alpha = 1.0f / std::sqrt((float)C);
      p.batch = B; p.sA = (int64_t)HW * qk.ld; p.sW = (int64_t)HW * qk.ld; p.sO = (int64_t)HW * HW;
      gemm(p);
    }
    {
      const int d = dt; void* x = sc.p; const int r = B * HW, c = HW;
      push([=](hipStream_t s) { return launch_softmax_rows(d, x, c, r, c, 1.0f, s); });
    }
    TView ao = alloc(rows, C);
    {
      GemmParams p;
      p.A = sc.p; p.lda = HW; p.W = vt.p; p.Kp = HW;
      p.M = HW; p.N = C; p.K = HW; p.Cin = HW; p.mode = GATHER_1D; p.rows_pb = HW; p.Lin = HW; p.Lout = HW; p.taps = 1;
      p.out = ao.p; p.ldo = C; p.batch = B; p.sA = (int64_t)HW * HW; p.sW = (int64_t)C * HW; p.sO = (int64_t)HW * C;
      gemm(p);
    }
    TView o = hn;   // reuse: the normalised copy is dead; the other temporaries (small next to the 1024 x 64 activations) stay allocated
    { GOpt go; go.residual = &h; linear(ao, rows, w.proj, o, go); }
    return o;
  }

  // Transformer2DModel (transformer_2d.py:214-321) + BasicTransformerBlock (attention.py:276-335)
  // `shared` (round 5): the CFG batch [uncond; cond] enters with IDENTICAL rows in both halves (models.py:233: `torch.cat([latents] * 2)`,
  // same timestep) and nothing before the first cross-attention looks at the text, so x holds only B / 2 samples and
  // norm -> proj_in -> norm1 -> q | k | v -> self-attention -> to_out run ONCE; the halves part ways at attn2 (unconditional rows: the
  // single-key constant; conditional rows: the text) and everything from there on works on all B samples.  Exact: the same kernels
  // would have produced the same values twice.  Requires nshort == B / 2 (the CFG structure) -- the caller checks.
  void transformer(const XfW& w, const TView& x, int B, int H, int W, int groups, const TView& kv, const void* kvt,
                   const float* bias, int L, const TView& out, int nshort = 0, const float* cvec = nullptr, bool shared = false) {
    const int C = w.C, HW = H * W;
    const int64_t rows = (int64_t)B * HW;
    const int Bp = shared ? B / 2 : B;                  // samples of the part in front of attn2
    const int64_t rows_p = (int64_t)Bp * HW;
    const size_t m = A.mark();
    auto rows_from = [&](const TView& v, int64_t r0) { TView t = v; t.p = (char*)v.p + (size_t)r0 * v.ld * esz; return t; };
    TView t0 = alloc(rows_p, C);
    TView h = alloc(rows, C);
    if (!gn_proj_stat(x, Bp, HW, w, groups, h) && !linear_gn_fold(x, Bp, HW, w.gn, groups, w.proj_in, h)) {
      groupnorm(x, Bp, HW, w.gn, groups, ACT_NONE, t0);
      linear(t0, rows_p, w.proj_in, h);
    }
    TView qkv = alloc(rows_p, 2 * C);               // [q | k]; v goes transposed into vt [B][C][HW]
    void* vt = A.alloc((size_t)rows_p * C * esz);
    GOpt nb; nb.use_bias = false;
    const int attn_fp8 = (E.cfg.unet_attn_fp8 != 0 && dt != DT_F32 && HW % 64 == 0) ? ((E.cfg.unet_attn_fp8 == 2 && HW % 128 == 0) ? 2 : 1) : 0;
    // round 6: where the projection runs on the activation-stationary kernel AND the attention site can fetch by LDS-DMA, v^T is written with its
    // keys in the attention kernel's fragment order (AttnParams::vt_perm): producer and consumer agree here, in one place
    int vperm = 0;
    {
      AttnParams ap;
      ap.q = nullptr; ap.k = nullptr; ap.vt = nullptr; ap.o = nullptr; ap.ldq = ap.ldk = ap.ldo = 0; ap.ldvt = HW;
      ap.bias = nullptr; ap.B = Bp; ap.heads = w.heads; ap.Sq = HW; ap.Skv = HW; ap.scale = 0.125f; ap.fp8_pv = attn_fp8;
      vperm = tuning().attn_vdma && attention_vt_perm_ok(dt, ap) ? 1 : 0;
    }
    if (!qkv_stat(h, rows_p, w, qkv, vt, HW, vperm)) {
      // GEMM-route producers (levels 1-2; level 0 below the activation-stationary kernel's row threshold): every EPI_VT epilogue can write the
      // permuted order as well (TANGO_ATTN_VDMA=2, the default; 1 = only qkv_stat_kernel does)
      if (tuning().attn_vdma < 2) vperm = 0;
      GOpt o = nb; o.ln = &w.ln1; o.vt = vt; o.vt_n0 = 2 * C; o.vt_S = HW; o.vt_ld = HW; o.vt_perm = vperm; linear(h, rows_p, w.qkv, qkv, o);
    }
    TView a = alloc(rows_p, C);
    // self-attention; `unet_attn_fp8` (BASELINE config 5): P.V on the fp8 MFMA at the sites that dominate the attention time
    // (unmasked, Skv a multiple of 64); cross-attention (64 text tokens, masked) stays in the engine dtype
    // (unet_attn_fp8 == 2, round 6: the MX instruction, 128 keys per MFMA, where the sequence is a multiple of 128)
    attention(slice(qkv, 0, C, esz), slice(qkv, C, C, esz), vt, HW, a, nullptr, Bp, w.heads, HW, HW, 0.125f, nullptr, attn_fp8, vperm);
    TView h1 = alloc(rows, C);
    TView h2 = h;                       // h is dead after h1 was produced
    const int Lp8 = (L + 7) / 8 * 8;
    // Single-key prefix (round 4): the first `nshort` samples keep exactly one text key, so attn2(norm2(x)) + x is x plus a per-sample
    // constant (elementwise.hip: xattn_const / rowbias_add); the cross-attention proper runs on the remaining samples only.
    const int ns = cvec ? nshort : 0;
    const int64_t rs = (int64_t)ns * HW, rc = rows - rs;
    // attn1's to_out + residual; where its kernel can, the single-key rows get their constant in the same epilogue and land in h2
    // directly (in place over the residual h: every element is read and written by the same lane) -- no separate pass over them
    bool fused_const = false;
    if (shared) {
      // one to_out for both halves into the FIRST half of h1 (h1s); the unconditional rows of h2 = h1s + constant (a pass that reads
      // h1s and writes h2's first half -- h's first half, the to_out residual, is dead by then); the conditional rows read h1s as well
      GOpt o; o.residual = &h;
      linear(a, rows_p, w.o1, h1, o);
      const int d = dt, hw = HW, c = C;
      const void* xs = h1.p; void* ys = h2.p; const int64_t lx = h1.ld, ly = h2.ld;
      push([=](hipStream_t s) { return launch_rowbias_add(d, xs, lx, cvec, ys, ly, rs, hw, c, s); },
           "xattn_single_key rows=" + std::to_string(rs) + " C=" + std::to_string(C));
    } else {
      GOpt o; o.residual = &h;
      if (ns > 0) { o.rowvec = cvec; o.rowvec_rows = rs; o.rowvec_per = HW; o.out_lo = &h2; o.rowvec_done = &fused_const; }
      linear(a, rows, w.o1, h1, o);
    }
    if (!shared && ns > 0 && !fused_const) {
      const int d = dt, hw = HW, c = C;
      const void* xs = h1.p; void* ys = h2.p; const int64_t lx = h1.ld, ly = h2.ld;
      push([=](hipStream_t s) { return launch_rowbias_add(d, xs, lx, cvec, ys, ly, rs, hw, c, s); },
           "xattn_single_key rows=" + std::to_string(rs) + " C=" + std::to_string(C));
    }
    if (rc > 0) {
      const TView h1c = shared ? h1 : rows_from(h1, rs), h2c = rows_from(h2, rs), kvc = rows_from(kv, (int64_t)ns * L);
      const void* kvtc = (const char*)kvt + (size_t)ns * C * Lp8 * esz;
      const float* biasc = bias ? bias + (int64_t)ns * L : nullptr;
      const int Bc = B - ns;
      if (w.q2p && xattn_block_ok(dt, C, w.heads, HW, L, h1c.ld, h2c.ld, kvc.ld, Lp8)) {
        // norm2 -> attn2 (to_q, softmax(Q K^T) V over the 64 text tokens, to_out) -> + residual in ONE launch (xattn.hip)
        XAttnParams xp;
        xp.x = h1c.p; xp.ldx = h1c.ld; xp.wq = w.q2p; xp.bq = w.bq2p; xp.wsum = w.wsum2p;
        xp.k = kvc.p; xp.ldk = kvc.ld; xp.vt = kvtc; xp.ldvt = Lp8; xp.bias = biasc;
        xp.wo = w.o2.W; xp.ldwo = w.o2.Kp; xp.bo = w.o2.b; xp.out = h2c.p; xp.ldo = h2c.ld;
        xp.M = (int)rc; xp.HW = HW; xp.L = L; xp.eps = w.ln2.eps; xp.scale = 0.125f;
        const int d = dt;
        push([xp, d](hipStream_t s) { return launch_xattn_block(d, xp, s); },
             "xattn_block M=" + std::to_string(rc) + " C=" + std::to_string(C) + " L=" + std::to_string(L),
             4.0 * rc * (double)C * C + 4.0 * rc * (double)L * C);
      } else {
        TView q = slice(qkv, 0, C, esz);   // reuse the qkv buffer for the cross-attention query
        { GOpt o = nb; o.ln = &w.ln2; linear(h1c, rc, w.q2, q, o); }
        attention(q, kvc, kvtc, Lp8, a, biasc, Bc, w.heads, HW, L);
        { GOpt o; o.residual = &h1c; linear(a, rc, w.o2, h2c, o); }
      }
    }
    TView h3 = h1;                      // h1 is dead after h2 was produced
    if (!ff_fused(h2, rows, w, h3)) {
      TView gg = alloc(rows, 4 * C);
      { GOpt o; o.epi = EPI_GEGLU; o.ln = &w.ln3; linear(h2, rows, w.ff1, gg, o); }
      { GOpt o; o.residual = &h2; linear(gg, rows, w.ff2, h3, o); }
    }
    if (shared) {
      // the residual x exists once (B / 2 samples): one launch per half, both reading it
      GOpt o; o.residual = &x;
      linear(h3, rows_p, w.proj_out, out, o);
      linear(rows_from(h3, rows_p), rows_p, w.proj_out, rows_from(out, rows_p), o);
    } else {
      GOpt o; o.residual = &x; linear(h3, rows, w.proj_out, out, o);
    }
    A.release(m);
  }
};

// ================================================================================================
// Engine: construction and weight registry
// ================================================================================================
Engine::Engine(const tango_config_t& c) : cfg(c) {
  dt = c.dtype;
  esz = dtype_size(dt);
}

Engine::~Engine() {
  (void)hipDeviceSynchronize();     // replays of the graphs destroyed below may still be in flight (denoise does not wait for its stream)
  for (auto& kv : unet_plans) free_unet_plan(*kv.second);
  for (auto& kv : vae_plans) if (kv.second->slab) (void)hipFree(kv.second->slab);
  for (auto& kv : vae_enc_plans) if (kv.second->slab) (void)hipFree(kv.second->slab);
  for (auto& kv : voc_plans) if (kv.second->slab) (void)hipFree(kv.second->slab);
  for (auto& kv : t5_plans) if (kv.second->slab) (void)hipFree(kv.second->slab);
  for (auto& kv : stft_plans) if (kv.second->slab) (void)hipFree(kv.second->slab);
  for (void* p : owned) (void)hipFree(p);
  for (auto& hs : hslots) {
    if (hs.ev) (void)hipEventDestroy(hs.ev);
    if (hs.buf) (void)hipHostFree(hs.buf);
  }
  if (cap_stream) (void)hipStreamDestroy(cap_stream);
  if (cap_stream2) (void)hipStreamDestroy(cap_stream2);
  if (aux_stream) (void)hipStreamDestroy(aux_stream);
  for (hipEvent_t e : {ev_fork, ev_join, ev_fork_e, ev_join_e}) if (e) (void)hipEventDestroy(e);
  if (ev0) (void)hipEventDestroy(ev0);
  if (ev1) (void)hipEventDestroy(ev1);
}

void* Engine::dmalloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0) bytes = 256;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    set_error("hipMalloc of " + std::to_string(bytes) + " bytes failed");
    return nullptr;
  }
  owned.push_back(p);
  return p;
}

// ---- plan caches: one byte budget, least recently used first ------------------------------------------------------------------
namespace {
template <typename Map> void lru_candidate(Map& m, uint64_t& best, int which, int& best_which) {
  for (auto& kv : m)
    if (kv.second->meta.stamp < best) { best = kv.second->meta.stamp; best_which = which; }
}
template <typename Map, typename Free> bool lru_erase(Map& m, uint64_t stamp, Free&& fr) {
  for (auto it = m.begin(); it != m.end(); ++it)
    if (it->second->meta.stamp == stamp) { fr(*it->second); m.erase(it); return true; }
  return false;
}
}  // namespace

int Engine::stage_h2d(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s) {
  if (bytes == 0) return 0;
  if (bytes > kHostSlotBytes) {          // never on the hot path (tables are <= 32 KB): plain synchronous upload
    TANGO_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
    TANGO_HIP(hipStreamSynchronize(s));
    return 0;
  }
  HostSlot& hs = hslots[hslot_next];
  hslot_next = (hslot_next + 1) % kHostSlots;
  if (!hs.buf) {
    TANGO_HIP(hipHostMalloc(&hs.buf, kHostSlotBytes, hipHostMallocDefault));
    TANGO_HIP(hipEventCreateWithFlags(&hs.ev, hipEventDisableTiming));
  }
  if (hs.pending) TANGO_HIP(hipEventSynchronize(hs.ev));     // the upload that last used this slot (eight stagings ago) has run
  memcpy(hs.buf, src_host, bytes);
  TANGO_HIP(hipMemcpyAsync(dst_dev, hs.buf, bytes, hipMemcpyHostToDevice, s));
  TANGO_HIP(hipEventRecord(hs.ev, s));
  hs.pending = true;
  return 0;
}

int Engine::plan_count() const {
  return (int)(unet_plans.size() + vae_plans.size() + vae_enc_plans.size() + voc_plans.size() + t5_plans.size() + stft_plans.size());
}

// frees the least recently used plan of any cache; false when nothing is left to free
bool Engine::evict_lru() {
  uint64_t best = ~0ull;
  int which = -1;
  lru_candidate(unet_plans, best, 0, which);
  lru_candidate(vae_plans, best, 1, which);
  lru_candidate(vae_enc_plans, best, 2, which);
  lru_candidate(voc_plans, best, 3, which);
  lru_candidate(t5_plans, best, 4, which);
  lru_candidate(stft_plans, best, 5, which);
  if (which < 0) return false;
  // a slab may still be read by work queued on a stream: hipFree synchronises the device before it releases memory
  auto drop = [&](auto& P) {
    plan_bytes -= P.meta.bytes < plan_bytes ? P.meta.bytes : plan_bytes;
    if (P.slab) (void)hipFree(P.slab);
  };
  switch (which) {
    case 0:
      return lru_erase(unet_plans, best, [&](UNetPlan& P) {
        //  a replay of this plan may still be in flight -- denoise() no longer waits for its stream, and destroying an
        // executing graph is not safe on HIP: drain the device first (hipFree would do so anyway, but only AFTER the destroy)
        (void)hipDeviceSynchronize();
        free_unet_plan(P);
      });
    case 1: return lru_erase(vae_plans, best, [&](VaePlan& P) { drop(P); });
    case 2: return lru_erase(vae_enc_plans, best, [&](VaePlan& P) { drop(P); });
    case 3: return lru_erase(voc_plans, best, [&](VaePlan& P) { drop(P); });
    case 4: return lru_erase(t5_plans, best, [&](T5Plan& P) { drop(P); });
    default: return lru_erase(stft_plans, best, [&](StftPlan& P) { drop(P); });
  }
}

void Engine::free_unet_plan(UNetPlan& P) {
  if (P.exec) (void)hipGraphExecDestroy(P.exec);
  if (P.graph) (void)hipGraphDestroy(P.graph);
  if (P.exec_k) (void)hipGraphExecDestroy(P.exec_k);
  if (P.graph_k) (void)hipGraphDestroy(P.graph_k);
  P.exec = P.exec_k = nullptr; P.graph = P.graph_k = nullptr;
  for (auto& c : P.child) if (c) { free_unet_plan(*c); c.reset(); }
  release_slab(&P.slab, P.meta);
}

int Engine::make_room(size_t need) {
  while (plan_bytes + need > plan_budget)
    if (!evict_lru()) break;                                // nothing left to free: let the allocation decide
  return 0;
}

// a plan whose program could not be built gives its slab (and its share of the budget) back
void Engine::release_slab(char** slab, PlanMeta& m) {
  if (*slab) (void)hipFree(*slab);
  *slab = nullptr;
  plan_bytes -= m.bytes < plan_bytes ? m.bytes : plan_bytes;
  m.bytes = 0;
}

int Engine::alloc_slab(char** slab, size_t bytes, PlanMeta& m, bool zero) {
  TANGO_TRY(make_room(bytes));
  // the budget is a number, the device's free memory a fact (other engines / torch share it): on out-of-memory below the budget,
  // evict least recently used plans and retry before giving up
  for (;;) {
    const hipError_t e = hipMalloc((void**)slab, bytes);
    if (e == hipSuccess) break;
    (void)hipGetLastError();
    if (e != hipErrorOutOfMemory || !evict_lru())
      TANGO_FAIL(std::string("hipMalloc of a ") + std::to_string(bytes >> 20) + "-MiB plan workspace failed: " + hipGetErrorString(e));
  }
  if (zero && hipMemset(*slab, 0, bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(*slab); *slab = nullptr; TANGO_FAIL("hipMemset of a plan workspace failed"); }
  m.bytes = bytes;
  plan_bytes += bytes;
  touch(m);
  return 0;
}

void Engine::reg_slot(const std::string& name, std::vector<int64_t> shape, std::function<int(const float*, hipStream_t)> pack) {
  Slot s;
  s.name = name;
  s.shape = std::move(shape);
  s.pack = std::move(pack);
  slot_index[name] = (int)slots.size();
  slots.push_back(std::move(s));
}

void Engine::reg_vec(const std::string& name, int n, float** dst) {
  *dst = (float*)dmalloc((size_t)n * 4);
  float* d = *dst;
  reg_slot(name, {n}, [d, n](const float* src, hipStream_t s) {
    TANGO_HIP(hipMemcpyAsync(d, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    return 0;
  });
}

void Engine::reg_norm(const std::string& p, int C, float eps, WNorm& w) {
  w.C = C;
  w.eps = eps;
  reg_vec(p + ".weight", C, &w.g);
  reg_vec(p + ".bias", C, &w.b);
}

// generic [N][K] matrix (Linear / 1x1 conv); `row_off` places it inside a fused matrix
void Engine::reg_mat(const std::string& wname, int N, int K, WMat& w, bool alloc, int row_off, std::vector<int64_t> shape, bool geglu) {
  if (alloc) {
    w.K = K; w.Cin = K; w.taps = 1; w.Kp = K;
    w.W = dmalloc((size_t)w.N * w.Kp * esz);
  }
  void* W = w.W;
  const int64_t Kp = w.Kp;
  const int d = dt;
  const int64_t ro = geglu ? -1 : row_off;
  reg_slot(wname, std::move(shape), [=](const float* src, hipStream_t s) { return launch_pack(d, src, W, N, 1, K, K, 0, 1, Kp, ro, s); });
}

void Engine::reg_linear(const std::string& p, int N, int K, WMat& w, bool bias) {
  w.N = N;
  reg_mat(p + ".weight", N, K, w, true, 0, {N, K});
  if (bias) reg_vec(p + ".bias", N, &w.b);
}

void Engine::reg_conv1x1(const std::string& p, int Cout, int Cin, WMat& w) {
  w.N = Cout;
  reg_mat(p + ".weight", Cout, Cin, w, true, 0, {Cout, Cin, 1, 1});
  reg_vec(p + ".bias", Cout, &w.b);
}

void Engine::reg_conv3x3(const std::string& p, int Cout, int Cin, WMat& w) {
  w.N = Cout;
  w.im2col = ((Cin * esz) % 64) != 0;
  if (w.im2col) {
    w.Kp = ((9 * Cin + 31) / 32) * 32;
    w.K = 9 * Cin; w.Cin = Cin; w.taps = 9;
  } else {
    w.Kp = 9 * Cin; w.K = 9 * Cin; w.Cin = Cin; w.taps = 9;
  }
  w.W = dmalloc((size_t)Cout * w.Kp * esz);
  void* W = w.W;
  const int64_t Kp = w.Kp;
  const int d = dt;
  // OIHW -> [O][tap][I]
  reg_slot(p + ".weight", {Cout, Cin, 3, 3},
           [=](const float* src, hipStream_t s) { return launch_pack(d, src, W, Cout, 9, Cin, (int64_t)Cin * 9, 1, 9, Kp, 0, s); });
  reg_vec(p + ".bias", Cout, &w.b);
}

void Engine::reg_conv1d(const std::string& p, int Cout, int Cin, int k, WMat& w) {
  w.N = Cout; w.K = k * Cin; w.Cin = Cin; w.taps = k; w.Kp = (int64_t)k * Cin;
  w.W = dmalloc((size_t)Cout * w.Kp * esz);
  void* W = w.W;
  const int64_t Kp = w.Kp;
  const int d = dt;
  // [O][I][k] -> [O][k][I]
  reg_slot(p + ".weight", {Cout, Cin, k},
           [=](const float* src, hipStream_t s) { return launch_pack(d, src, W, Cout, k, Cin, (int64_t)Cin * k, 1, k, Kp, 0, s); });
  reg_vec(p + ".bias", Cout, &w.b);
}

void Engine::reg_convt1d(const std::string& p, int Cin, int Cout, int k, int u, ConvTW& w) {
  w.cin = Cin; w.cout = Cout; w.k = k; w.u = u; w.pad = (k - u) / 2;
  w.phase.resize(u);
  std::vector<void*> Ws(u);
  std::vector<int> Ts(u);
  for (int r = 0; r < u; ++r) {
    const int T = (k - r + u - 1) / u;
    WMat& m = w.phase[r];
    m.N = Cout; m.taps = T; m.Cin = Cin; m.K = T * Cin; m.Kp = (int64_t)T * Cin;
    m.W = dmalloc((size_t)Cout * m.Kp * esz);
    Ws[r] = m.W; Ts[r] = T;
  }
  const int d = dt;
  // [I][O][k]: phase r, row o, tap t, channel i  <-  w[i][o][r + u*t]
  reg_slot(p + ".weight", {Cin, Cout, k}, [=](const float* src, hipStream_t s) {
    for (int r = 0; r < u; ++r)
      TANGO_TRY(launch_pack(d, src + r, Ws[r], Cout, Ts[r], Cin, k, u, (int64_t)Cout * k, (int64_t)Ts[r] * Cin, 0, s));
    return 0;
  });
  reg_vec(p + ".bias", Cout, &w.b);
  for (int r = 0; r < u; ++r) w.phase[r].b = w.b;
}

void Engine::reg_linear_f32(const std::string& p, int N, int K, WLinF32& w) {
  w.N = N; w.K = K;
  w.W = (float*)dmalloc((size_t)N * K * 4);
  float* W = w.W;
  reg_slot(p + ".weight", {N, K}, [=](const float* src, hipStream_t s) {
    TANGO_HIP(hipMemcpyAsync(W, src, (size_t)N * K * 4, hipMemcpyDeviceToDevice, s));
    return 0;
  });
  reg_vec(p + ".bias", N, &w.b);
}

void Engine::reg_res(const std::string& p, int cin, int cout, int temb, float eps, ResW& w, bool vae) {
  w.cin = cin; w.cout = cout;
  reg_norm(p + ".norm1", cin, eps, w.n1);
  reg_conv3x3(p + ".conv1", cout, cin, w.c1);
  if (temb > 0) {
    w.has_temb = true;
    reg_linear_f32(p + ".time_emb_proj", cout, temb, w.temb);
    w.temb_table = (float*)dmalloc((size_t)max_steps * cout * 4);
  }
  reg_norm(p + ".norm2", cout, eps, w.n2);
  reg_conv3x3(p + ".conv2", cout, cout, w.c2);
  if (cin != cout) {
    w.has_sc = true;
    reg_conv1x1(p + (vae ? ".nin_shortcut" : ".conv_shortcut"), cout, cin, w.sc);
  }
}

void Engine::reg_xf(const std::string& p, int C, int heads, int cross, XfW& w) {
  w.C = C; w.heads = heads;
  reg_norm(p + ".norm", C, 1e-6f, w.gn);   // transformer_2d.py:145
  reg_linear(p + ".proj_in", C, C, w.proj_in, true);
  const std::string b = p + ".transformer_blocks.0";
  // fused [to_q; to_k; to_v] (no bias, attention_processor.py:34-111)
  w.qkv.N = 3 * C; w.qkv.K = C; w.qkv.Cin = C; w.qkv.Kp = C; w.qkv.taps = 1;
  w.qkv.W = dmalloc((size_t)3 * C * C * esz);
  reg_mat(b + ".attn1.to_q.weight", C, C, w.qkv, false, 0, {C, C});
  reg_mat(b + ".attn1.to_k.weight", C, C, w.qkv, false, C, {C, C});
  reg_mat(b + ".attn1.to_v.weight", C, C, w.qkv, false, 2 * C, {C, C});
  reg_linear(b + ".attn1.to_out.0", C, C, w.o1, true);
  // GEGLU proj: rows interleaved [16 value | 16 gate] so the pair lands in one lane (gemm.hip)
  w.ff1.N = 8 * C;
  reg_mat(b + ".ff.net.0.proj.weight", 8 * C, C, w.ff1, true, 0, {8 * C, C}, true);
  {
    w.ff1.b = (float*)dmalloc((size_t)8 * C * 4);
    float* d = w.ff1.b;
    const int n = 8 * C;
    reg_slot(b + ".ff.net.0.proj.bias", {n}, [d, n](const float* src, hipStream_t s) { return launch_permute_geglu_bias(src, d, n, s); });
  }
  reg_linear(b + ".ff.net.2", C, 4 * C, w.ff2, true);
  reg_linear(b + ".attn2.to_q", C, C, w.q2, false);
  w.kv2.N = 2 * C; w.kv2.K = cross; w.kv2.Cin = cross; w.kv2.Kp = cross; w.kv2.taps = 1;
  w.kv2.W = dmalloc((size_t)2 * C * cross * esz);
  reg_mat(b + ".attn2.to_k.weight", C, cross, w.kv2, false, 0, {C, cross});
  reg_mat(b + ".attn2.to_v.weight", C, cross, w.kv2, false, C, {C, cross});
  reg_linear(b + ".attn2.to_out.0", C, C, w.o2, true);
  reg_norm(b + ".norm1", C, 1e-5f, w.ln1);
  reg_norm(b + ".norm2", C, 1e-5f, w.ln2);
  reg_norm(b + ".norm3", C, 1e-5f, w.ln3);
  reg_linear(p + ".proj_out", C, C, w.proj_out, true);
}

void Engine::build_unet_weights() {
  const int nl = cfg.unet_levels;
  const int* ch = cfg.unet_channels;
  const int temb = ch[0] * 4;
  const int cross = cfg.unet_cross_dim;
  const int lpb = cfg.unet_layers_per_block;
  const float eps = cfg.unet_eps;
  const std::string P = "unet.";
  reg_conv3x3(P + "conv_in", ch[0], cfg.unet_in_channels, conv_in);
  reg_linear_f32(P + "time_embedding.linear_1", temb, ch[0], time1);
  reg_linear_f32(P + "time_embedding.linear_2", temb, temb, time2);
  down.resize(nl);
  int cprev = ch[0];
  for (int i = 0; i < nl; ++i) {
    DownBlock& d = down[i];
    d.res.resize(lpb);
    if (cfg.unet_cross_attn[i]) {
      d.xf.resize(lpb);
      if (cfg.unet_music) { d.xf2.resize(lpb); d.xf3.resize(lpb); }
    }
    for (int j = 0; j < lpb; ++j) {
      const std::string bp = P + "down_blocks." + std::to_string(i);
      reg_res(bp + ".resnets." + std::to_string(j), j == 0 ? cprev : ch[i], ch[i], temb, eps, d.res[j], false);
      if (cfg.unet_cross_attn[i]) {
        reg_xf(bp + ".attentions." + std::to_string(j), ch[i], cfg.unet_heads[i], cross, d.xf[j]);
        if (cfg.unet_music) {   // CrossAttnDownBlock2DMusic (unet_2d_blocks.py:1079-1275)
          reg_xf(bp + ".attentions2." + std::to_string(j), ch[i], cfg.unet_heads[i], cross, d.xf2[j]); d.xf2[j].cond = 1;
          reg_xf(bp + ".attentions3." + std::to_string(j), ch[i], cfg.unet_heads[i], cross, d.xf3[j]); d.xf3[j].cond = 2;
        }
      }
    }
    if (i != nl - 1) {
      d.has_ds = true;
      reg_conv3x3(P + "down_blocks." + std::to_string(i) + ".downsamplers.0.conv", ch[i], ch[i], d.ds);
    }
    cprev = ch[i];
  }
  const int cm = ch[nl - 1];
  reg_res(P + "mid_block.resnets.0", cm, cm, temb, eps, mid_res0, false);
  reg_xf(P + "mid_block.attentions.0", cm, cfg.unet_heads[nl - 1], cross, mid_xf);
  if (cfg.unet_music) {           // UNetMidBlock2DCrossAttnMusic (unet_2d_blocks.py:603-757)
    reg_xf(P + "mid_block.attentions2.0", cm, cfg.unet_heads[nl - 1], cross, mid_xf2); mid_xf2.cond = 1;
    reg_xf(P + "mid_block.attentions3.0", cm, cfg.unet_heads[nl - 1], cross, mid_xf3); mid_xf3.cond = 2;
  }
  reg_res(P + "mid_block.resnets.1", cm, cm, temb, eps, mid_res1, false);
  up.resize(nl);
  int prev_out = ch[nl - 1];
  for (int i = 0; i < nl; ++i) {
    const int lvl = nl - 1 - i;
    const int outc = ch[lvl];
    const int inc = ch[std::max(lvl - 1, 0)];
    UpBlock& u = up[i];
    const bool xa = cfg.unet_cross_attn[lvl] != 0;   // up block i mirrors down block (nl-1-i)
    u.res.resize(lpb + 1);
    if (xa) {
      u.xf.resize(lpb + 1);
      if (cfg.unet_music) { u.xf2.resize(lpb + 1); u.xf3.resize(lpb + 1); }
    }
    const std::string bp = P + "up_blocks." + std::to_string(i);
    for (int j = 0; j <= lpb; ++j) {
      const int skip = (j == lpb) ? inc : outc;
      const int rin = (j == 0) ? prev_out : outc;
      reg_res(bp + ".resnets." + std::to_string(j), rin + skip, outc, temb, eps, u.res[j], false);
      if (xa) {
        reg_xf(bp + ".attentions." + std::to_string(j), outc, cfg.unet_heads[lvl], cross, u.xf[j]);
        if (cfg.unet_music) {   // CrossAttnUpBlock2DMusic (unet_2d_blocks.py:2251-2436)
          reg_xf(bp + ".attentions2." + std::to_string(j), outc, cfg.unet_heads[lvl], cross, u.xf2[j]); u.xf2[j].cond = 1;
          reg_xf(bp + ".attentions3." + std::to_string(j), outc, cfg.unet_heads[lvl], cross, u.xf3[j]); u.xf3[j].cond = 2;
        }
      }
    }
    if (i != nl - 1) {
      u.has_us = true;
      reg_conv3x3(bp + ".upsamplers.0.conv", outc, outc, u.us);
    }
    prev_out = outc;
  }
  reg_norm(P + "conv_norm_out", ch[0], eps, norm_out);
  reg_conv3x3(P + "conv_out", cfg.unet_out_channels, ch[0], conv_out);
  // pointer lists (vectors above are final now)
  auto push_xfs = [&](std::vector<XfW>& a, std::vector<XfW>& b, std::vector<XfW>& c) {
    for (auto& x : a) all_xf.push_back(&x);
    for (auto& x : b) all_xf.push_back(&x);
    for (auto& x : c) all_xf.push_back(&x);
  };
  for (auto& d : down) { for (auto& r : d.res) all_res.push_back(&r); push_xfs(d.xf, d.xf2, d.xf3); }
  all_res.push_back(&mid_res0); all_res.push_back(&mid_res1); all_xf.push_back(&mid_xf);
  if (cfg.unet_music) { all_xf.push_back(&mid_xf2); all_xf.push_back(&mid_xf3); }
  for (auto& u : up) { for (auto& r : u.res) all_res.push_back(&r); push_xfs(u.xf, u.xf2, u.xf3); }
}

void Engine::build_vae_weights() {
  const int nl = cfg.vae_levels;
  const int ch = cfg.vae_ch;
  const int zc = cfg.vae_z_channels, ed = cfg.vae_embed_dim;
  pqc_w = (float*)dmalloc((size_t)zc * ed * 4);
  {
    float* d = pqc_w;
    const size_t n = (size_t)zc * ed;
    reg_slot("post_quant_conv.weight", {zc, ed, 1, 1}, [d, n](const float* src, hipStream_t s) {
      TANGO_HIP(hipMemcpyAsync(d, src, n * 4, hipMemcpyDeviceToDevice, s));
      return 0;
    });
  }
  reg_vec("post_quant_conv.bias", zc, &pqc_b);
  const std::string D = "decoder.";
  int bi = ch * cfg.vae_ch_mult[nl - 1];
  reg_conv3x3(D + "conv_in", bi, zc, vae_conv_in);
  reg_res(D + "mid.block_1", bi, bi, 0, 1e-6f, vae_mid1, true);
  reg_vae_attn(D + "mid.attn_1", bi, vae_attn);
  reg_res(D + "mid.block_2", bi, bi, 0, 1e-6f, vae_mid2, true);
  vae_up.resize(nl);
  for (int lvl = nl - 1; lvl >= 0; --lvl) {
    const int bo = ch * cfg.vae_ch_mult[lvl];
    VaeUp& u = vae_up[lvl];
    u.res.resize(cfg.vae_num_res_blocks + 1);
    for (int b = 0; b <= cfg.vae_num_res_blocks; ++b) {
      reg_res(D + "up." + std::to_string(lvl) + ".block." + std::to_string(b), bi, bo, 0, 1e-6f, u.res[b], true);
      bi = bo;
    }
    if (lvl != 0) {
      u.has_up = true;
      reg_conv3x3(D + "up." + std::to_string(lvl) + ".upsample.conv", bi, bi, u.up);
    }
  }
  reg_norm(D + "norm_out", bi, 1e-6f, vae_norm_out);
  reg_conv3x3(D + "conv_out", cfg.vae_out_ch, bi, vae_conv_out);
}

void Engine::reg_vae_attn(const std::string& a, int bi, VaeAttnW& w) {
  reg_norm(a + ".norm", bi, 1e-6f, w.gn);
  // fused [q; k] 1x1 convs; v and proj_out separate
  w.qk.N = 2 * bi; w.qk.K = bi; w.qk.Cin = bi; w.qk.Kp = bi; w.qk.taps = 1;
  w.qk.W = dmalloc((size_t)2 * bi * bi * esz);
  w.qk.b = (float*)dmalloc((size_t)2 * bi * 4);
  reg_mat(a + ".q.weight", bi, bi, w.qk, false, 0, {bi, bi, 1, 1});
  reg_mat(a + ".k.weight", bi, bi, w.qk, false, bi, {bi, bi, 1, 1});
  {
    float* b0 = w.qk.b;
    float* b1 = w.qk.b + bi;
    const size_t n = (size_t)bi * 4;
    reg_slot(a + ".q.bias", {bi}, [b0, n](const float* src, hipStream_t s) { TANGO_HIP(hipMemcpyAsync(b0, src, n, hipMemcpyDeviceToDevice, s)); return 0; });
    reg_slot(a + ".k.bias", {bi}, [b1, n](const float* src, hipStream_t s) { TANGO_HIP(hipMemcpyAsync(b1, src, n, hipMemcpyDeviceToDevice, s)); return 0; });
  }
  reg_conv1x1(a + ".v", bi, bi, w.v);
  reg_conv1x1(a + ".proj_out", bi, bi, w.proj);
}

// mel-VAE encoder weights: encoder.* + quant_conv (modules.py:419-517, autoencoder.py:38)
void Engine::build_vae_enc_weights() {
  const int nl = cfg.vae_levels, ch = cfg.vae_ch, zc = cfg.vae_z_channels, ed = cfg.vae_embed_dim;
  const int cin = cfg.vae_in_channels > 0 ? cfg.vae_in_channels : 1;
  const std::string E = "encoder.";
  reg_conv3x3(E + "conv_in", ch, cin, vae_enc_conv_in);
  vae_down.resize(nl);
  int bi = ch;
  for (int lvl = 0; lvl < nl; ++lvl) {
    const int bo = ch * cfg.vae_ch_mult[lvl];
    VaeDown& d = vae_down[lvl];
    d.res.resize(cfg.vae_num_res_blocks);
    for (int b = 0; b < cfg.vae_num_res_blocks; ++b) {
      reg_res(E + "down." + std::to_string(lvl) + ".block." + std::to_string(b), bi, bo, 0, 1e-6f, d.res[b], true);
      bi = bo;
    }
    if (lvl != nl - 1) {
      d.has_down = true;
      reg_conv3x3(E + "down." + std::to_string(lvl) + ".downsample.conv", bi, bi, d.down);
    }
  }
  reg_res(E + "mid.block_1", bi, bi, 0, 1e-6f, vae_enc_mid1, true);
  reg_vae_attn(E + "mid.attn_1", bi, vae_enc_attn);
  reg_res(E + "mid.block_2", bi, bi, 0, 1e-6f, vae_enc_mid2, true);
  reg_norm(E + "norm_out", bi, 1e-6f, vae_enc_norm_out);
  reg_conv3x3(E + "conv_out", 2 * zc, bi, vae_enc_conv_out);
  qc_w = (float*)dmalloc((size_t)2 * ed * 2 * zc * 4);
  {
    float* d = qc_w;
    const size_t n = (size_t)2 * ed * 2 * zc;
    reg_slot("quant_conv.weight", {2 * ed, 2 * zc, 1, 1}, [d, n](const float* src, hipStream_t s) {
      TANGO_HIP(hipMemcpyAsync(d, src, n * 4, hipMemcpyDeviceToDevice, s));
      return 0;
    });
  }
  reg_vec("quant_conv.bias", 2 * ed, &qc_b);
}

void Engine::build_voc_weights() {
  const std::string P = "vocoder.";
  const int c0 = cfg.voc_initial_channel;
  reg_conv1d(P + "conv_pre", c0, cfg.voc_num_mels, 7, voc_pre);
  voc_ups.resize(cfg.voc_n_ups);
  for (int i = 0; i < cfg.voc_n_ups; ++i)
    reg_convt1d(P + "ups." + std::to_string(i), c0 >> i, c0 >> (i + 1), cfg.voc_kernels[i], cfg.voc_rates[i], voc_ups[i]);
  const int nk = cfg.voc_n_resblocks;
  voc_res.resize((size_t)cfg.voc_n_ups * nk);
  int ch = c0;
  for (int i = 0; i < cfg.voc_n_ups; ++i) {
    ch = c0 >> (i + 1);
    for (int j = 0; j < nk; ++j) {
      VocResW& r = voc_res[(size_t)i * nk + j];
      r.ch = ch; r.k = cfg.voc_res_kernels[j];
      for (int m = 0; m < 4 && cfg.voc_res_dilations[j][m] > 0; ++m) r.dil.push_back(cfg.voc_res_dilations[j][m]);
      r.c1.resize(r.dil.size()); r.c2.resize(r.dil.size());
      const std::string rp = P + "resblocks." + std::to_string(i * nk + j);
      for (size_t m = 0; m < r.dil.size(); ++m) reg_conv1d(rp + ".convs1." + std::to_string(m), ch, ch, r.k, r.c1[m]);
      for (size_t m = 0; m < r.dil.size(); ++m) reg_conv1d(rp + ".convs2." + std::to_string(m), ch, ch, r.k, r.c2[m]);
    }
  }
  reg_conv1d(P + "conv_post", 1, ch, 7, voc_post);
}

int Engine::init() {
  if (dt != DT_F32 && dt != DT_F16 && dt != DT_BF16) TANGO_FAIL("engine: bad dtype");
  TANGO_TRY(gemm_init());
  if (const char* mb = getenv("TANGO_PLAN_BUDGET_MB")) {
    const long long v = atoll(mb);
    if (v > 0) plan_budget = (size_t)v << 20;
  }
  if (cfg.unet_attn_fp8 && dt == DT_F32) TANGO_FAIL("engine: unet_attn_fp8 needs a 16-bit engine dtype (bf16 / fp16)");
  d_sync = (unsigned*)dmalloc((size_t)coop_sync_words() * 4);
  if (!d_sync) return -1;
  TANGO_HIP(hipMemset(d_sync, 0, (size_t)coop_sync_words() * 4));
  d_sync2 = (unsigned*)dmalloc((size_t)coop_sync_words() * 4);
  if (!d_sync2) return -1;
  TANGO_HIP(hipMemset(d_sync2, 0, (size_t)coop_sync_words() * 4));
  if (cfg.unet_levels > 0) {
    for (int i = 0; i < cfg.unet_levels; ++i)
      if (cfg.unet_channels[i] != cfg.unet_heads[i] * 64) TANGO_FAIL("engine: UNet channels must equal heads * 64 (head_dim 64)");
    build_unet_weights();
    const int temb = cfg.unet_channels[0] * 4;
    d_step = (int*)dmalloc(256);
    d_sched = (SchedParams*)dmalloc(sizeof(SchedParams));
    d_ts = (int64_t*)dmalloc((size_t)max_steps * 8);
    d_coef = (float*)dmalloc((size_t)max_steps * 8 * 4);
    d_sin = (float*)dmalloc((size_t)max_steps * cfg.unet_channels[0] * 4);
    d_t1 = (float*)dmalloc((size_t)max_steps * temb * 4);
    d_temb = (float*)dmalloc((size_t)max_steps * temb * 4);
    if (!d_step || !d_sched || !d_ts || !d_coef || !d_sin || !d_t1 || !d_temb) return -1;
  }
  if (cfg.vae_levels > 0) build_vae_weights();
  if (cfg.vae_levels > 0 && cfg.vae_encoder) {
    if (2 * cfg.vae_z_channels > 32) TANGO_FAIL("engine: VAE encoder with 2 * z_channels > 32 is not supported (quant_conv kernel)");
    build_vae_enc_weights();
  }
  if (cfg.voc_n_ups > 0) build_voc_weights();
  if (cfg.t5_layers > 0) {
    // FLAN-T5 hidden states exceed the fp16 range, and the 16-bit tile epilogues carry the exact-erf gate only (the T5 feed-forward
    // is gated tanh-GELU): refuse here, by name, instead of at the first encode_text
    if (dt != DT_F32) TANGO_FAIL("engine: the FLAN-T5 text encoder runs on an fp32 engine only (dtype must be fp32 when a t5 config is given)");
    if (cfg.t5_d_kv != 64) TANGO_FAIL("engine: T5 d_kv must be 64 (attention head_dim)");
    if (cfg.t5_d_model % 16 || cfg.t5_d_ff % 16) TANGO_FAIL("engine: T5 d_model / d_ff must be multiples of 16");
    build_t5_weights();
  }
  if (cfg.stft_filter_length > 0) {
    if (dt != DT_F32) TANGO_FAIL("engine: the STFT front-end runs on an fp32 engine only (frontend.hip)");
    if (cfg.stft_filter_length % 16 || cfg.stft_hop_length <= 0 || cfg.stft_hop_length % 4 || cfg.stft_n_mel <= 0)
      TANGO_FAIL("engine: STFT front-end needs filter_length % 16 == 0, hop_length % 4 == 0, n_mel > 0");
    build_stft_weights();
  }
  for (void* p : owned) if (!p) return -1;
  TANGO_HIP(hipEventCreate(&ev0));
  TANGO_HIP(hipEventCreate(&ev1));
  return 0;
}

int Engine::set_weight(const char* name, const float* dev, const int64_t* shape, int ndim) {
  auto it = slot_index.find(name);
  if (it == slot_index.end()) TANGO_FAIL(std::string("set_weight: unexpected key '") + name + "'");
  Slot& s = slots[it->second];
  bool ok = (int)s.shape.size() == ndim;
  for (int i = 0; ok && i < ndim; ++i) ok = s.shape[i] == shape[i];
  if (!ok) {
    std::string e = "set_weight: size mismatch for " + s.name + ": expected (";
    for (auto v : s.shape) e += std::to_string(v) + ",";
    e += ") got (";
    for (int i = 0; i < ndim; ++i) e += std::to_string(shape[i]) + ",";
    TANGO_FAIL(e + ")");
  }
  TANGO_TRY(s.pack(dev, 0));
  temb_ts.clear();   // time-embedding tables are derived from weights: never reuse them across a reload
  TANGO_HIP(hipStreamSynchronize(0));   // the caller may free/reuse `dev` right after this returns
  s.set = true;
  return 0;
}

int Engine::fold_ln(WMat& w, const WNorm& ln) {
  if (!w.Wln) {
    w.Wln = dmalloc((size_t)w.N * w.Kp * esz);
    w.bln = (float*)dmalloc((size_t)w.N * 4);
    w.wsum = (float*)dmalloc((size_t)w.N * 4);
    if (!w.Wln || !w.bln || !w.wsum) return -1;
  }
  return launch_fold_ln(dt, w.W, w.Kp, ln.g, ln.b, w.b, w.Wln, w.bln, w.wsum, w.N, w.K, 0);
}

int Engine::finalize_weights() {
  for (auto& s : slots)
    if (!s.set) TANGO_FAIL("finalize_weights: missing key '" + s.name + "'");
  // fold the three LayerNorms of every BasicTransformerBlock into the projections that consume them
  for (XfW* x : all_xf) {
    TANGO_TRY(fold_ln(x->qkv, x->ln1));
    TANGO_TRY(fold_ln(x->q2, x->ln2));
    TANGO_TRY(fold_ln(x->ff1, x->ln3));
    // Transformer2DModel.norm (GroupNorm) into proj_in for the activation-stationary kernel of level 0 (Builder::gn_proj_stat)
    if (dt != DT_F32 && x->C == 320) TANGO_TRY(fold_ln(x->proj_in, x->gn));
    if (dt != DT_F32 && x->C == 320 && x->heads == 5) {     // operands of the fused cross-attention block (xattn.hip)
      if (!x->q2p) {
        x->q2p = dmalloc((size_t)x->C * x->C * esz);
        x->bq2p = (float*)dmalloc((size_t)x->C * 4);
        x->wsum2p = (float*)dmalloc((size_t)x->C * 4);
        if (!x->q2p || !x->bq2p || !x->wsum2p) return -1;
      }
      TANGO_TRY(launch_xattn_permute_wq(dt, x->q2.Wln, x->q2.bln, x->q2.wsum, x->q2p, x->bq2p, x->wsum2p, x->C, 0));
    }
  }
  TANGO_HIP(hipDeviceSynchronize());
  temb_ts.clear();
  finalized = true;
  return 0;
}

// ================================================================================================
// time-embedding tables: sinusoid -> Linear/SiLU/Linear -> per-ResBlock Linear(SiLU(emb))
// (embeddings.py:22-62,200-212; resnet.py:573-577) for all N steps at once, fp32 on the f32 MFMA path
// ================================================================================================
static GemmParams f32_linear(const float* x, int64_t lda, const WLinF32& w, float* out, int M, int a_act, int e_act) {
  GemmParams p;
  p.A = x; p.lda = lda; p.W = w.W; p.Kp = w.K; p.bias = w.b;
  p.M = M; p.N = w.N; p.K = w.K; p.Cin = w.K;
  p.mode = GATHER_1D; p.rows_pb = M; p.Lin = M; p.Lout = M; p.taps = 1;
  p.out = out; p.ldo = w.N;
  p.a_act = a_act; p.e_act = e_act;
  return p;
}

int Engine::ensure_temb(const int64_t* ts_host, int n, hipStream_t s) {
  if (n > max_steps) TANGO_FAIL("denoise: num_steps exceeds 1000");
  if ((int)temb_ts.size() == n && std::equal(temb_ts.begin(), temb_ts.end(), ts_host)) return 0;
  TANGO_TRY(stage_h2d(d_ts, ts_host, (size_t)n * 8, s));   // ts_host may be pageable / transient: pinned staging, no host sync
  const int c0 = cfg.unet_channels[0];
  TANGO_TRY(launch_timestep_embedding(d_ts, d_sin, n, c0, cfg.unet_flip_sin_to_cos, cfg.unet_freq_shift, s));
  TANGO_TRY(launch_gemm(DT_F32, f32_linear(d_sin, c0, time1, d_t1, n, ACT_NONE, ACT_SILU), s));
  // d_temb = silu(linear_2(.)): every consumer applies the nonlinearity first (resnet.py:574)
  TANGO_TRY(launch_gemm(DT_F32, f32_linear(d_t1, time1.N, time2, d_temb, n, ACT_NONE, ACT_SILU), s));
  for (ResW* r : all_res) TANGO_TRY(launch_gemm(DT_F32, f32_linear(d_temb, time2.N, r->temb, r->temb_table, n, ACT_NONE, ACT_NONE), s));
  temb_ts.assign(ts_host, ts_host + n);
  return 0;
}

// ================================================================================================
// UNet plan
// ================================================================================================
int Engine::build_unet(UNetPlan& P, Arena& A, bool record) {
  Builder b{*this, A, &P.step, record, dt, esz, P.sync};
  const int nl = cfg.unet_levels;
  const int* ch = cfg.unet_channels;
  const int B2 = P.B2, G = cfg.unet_groups;
  const int lpb = cfg.unet_layers_per_block;
  auto HH = [&](int lvl) { return cfg.latent_h >> lvl; };
  auto WW = [&](int lvl) { return cfg.latent_w >> lvl; };
  auto rows_at = [&](int lvl) { return (int64_t)B2 * HH(lvl) * WW(lvl); };
  const int cin_pad = 8 > cfg.unet_in_channels ? 8 : ((cfg.unet_in_channels + 7) / 8) * 8;

  // ---- persistent buffers ----
  TView xin;
  if (P.ext_xin) {                       // a chain of a dual plan: the parent owns the whole batch's input / output buffers
    xin.p = P.ext_xin; xin.ld = cin_pad; xin.C = cin_pad;
    P.eps = P.ext_eps;
  } else {
    xin = b.alloc(rows_at(0), cin_pad);
    P.eps = (float*)A.alloc((size_t)rows_at(0) * cfg.unet_out_channels * 4);
  }
  P.xin = xin.p;
  const int ncond = cfg.unet_music ? 3 : 1;
  TView encv[3];
  for (int c = 0; c < ncond; ++c) {
    encv[c] = b.alloc((int64_t)B2 * P.Lc[c], cfg.unet_cross_dim);
    P.encs[c] = encv[c].p;
    P.biases[c] = (float*)A.alloc((size_t)B2 * P.Lc[c] * 4);
  }
  P.enc = P.encs[0];
  P.bias = P.biases[0];

  // cross-attention K/V for every transformer: step-invariant, computed by P.pre once per call
  std::vector<TView> kvs(all_xf.size());       // K  [B2*L][C]
  std::vector<void*> kvts(all_xf.size());      // V^T [B2][C][Lp], Lp = L rounded up to 8 (pad columns stay zero)
  {
    Builder pb{*this, A, &P.pre, record, dt, esz};
    for (size_t i = 0; i < all_xf.size(); ++i) {
      const int C = all_xf[i]->C, Li = P.Lc[all_xf[i]->cond], Lpi = (Li + 7) / 8 * 8;
      kvs[i] = pb.alloc((int64_t)B2 * Li, C);
      kvts[i] = A.alloc((size_t)B2 * C * Lpi * esz);
      GOpt o; o.use_bias = false; o.vt = kvts[i]; o.vt_n0 = C; o.vt_S = Li; o.vt_ld = Lpi;
      pb.linear(encv[all_xf[i]->cond], (int64_t)B2 * Li, all_xf[i]->kv2, kvs[i], o);
    }
  }
  // single-key prefix: per text site the constant to_out(v_key) + b of the first n_short samples, computed by P.pre once per call
  std::vector<float*> cvecs(all_xf.size(), nullptr);
  P.key0 = (int*)A.alloc((size_t)B2 * 4);
  if (P.n_short > 0) {
    Builder pb{*this, A, &P.pre, record, dt, esz};
    for (size_t i = 0; i < all_xf.size(); ++i) {
      if (all_xf[i]->cond != 0) continue;
      const int C = all_xf[i]->C, Lpi = (P.Lc[0] + 7) / 8 * 8, d = dt, nsh = P.n_short;
      cvecs[i] = (float*)A.alloc((size_t)nsh * C * 4);
      const void* vt = kvts[i]; const int* k0 = P.key0; const void* wo = all_xf[i]->o2.W; const int64_t ldwo = all_xf[i]->o2.Kp;
      const float* bo = all_xf[i]->o2.b; float* cv = cvecs[i];
      pb.push([=](hipStream_t s) { return launch_xattn_const(d, vt, Lpi, C, k0, wo, ldwo, bo, cv, nsh, s); }, "xattn_const");
    }
  }
  auto kv_idx = [&](const XfW* w) -> size_t {
    for (size_t i = 0; i < all_xf.size(); ++i) if (all_xf[i] == w) return i;
    return 0;
  };

  // concat buffers of the up path: up resnet u = i*(lpb+1)+j consumes skip number (nskip-1-u)
  const int nup = nl * (lpb + 1);
  struct Cat { TView buf; int c1, c2, lvl; };
  std::vector<Cat> cats(nup);
  {
    int prev_out = ch[nl - 1];
    for (int i = 0; i < nl; ++i) {
      const int lvl = nl - 1 - i;
      const int outc = ch[lvl], inc = ch[std::max(lvl - 1, 0)];
      for (int j = 0; j <= lpb; ++j) {
        Cat& c = cats[i * (lpb + 1) + j];
        c.c2 = (j == lpb) ? inc : outc;
        c.c1 = (j == 0) ? prev_out : outc;
        c.lvl = lvl;
        c.buf = b.alloc(rows_at(lvl), c.c1 + c.c2);
      }
      prev_out = outc;
    }
  }
  // one cross-attention site: `attentions[j]`, then (Music UNet) `attentions2[j]` on the beats and `attentions3[j]` on the chords
  auto xf_site = [&](const XfW& w1, const XfW* w2, const XfW* w3, const TView& x, int lvl, const TView& out, bool shared = false) {
    auto one = [&](const XfW& w, const TView& in, const TView& o) {
      const size_t k = kv_idx(&w);
      b.transformer(w, in, B2, HH(lvl), WW(lvl), G, kvs[k], kvts[k], P.biases[w.cond], P.Lc[w.cond], o, w.cond == 0 ? P.n_short : 0, cvecs[k], shared);
    };
    if (!w2) { one(w1, x, out); return; }
    const size_t m = A.mark();
    TView t1 = b.alloc(rows_at(lvl), w1.C), t2 = b.alloc(rows_at(lvl), w1.C);
    one(w1, x, t1);
    one(*w2, t1, t2);
    one(*w3, t2, out);
    A.release(m);
  };
  const bool music = cfg.unet_music != 0;
  int skip_no = 0;
  auto skip_dst = [&]() -> TView {   // destination view of the next skip tensor
    Cat& c = cats[nup - 1 - skip_no];
    ++skip_no;
    return Builder::slice(c.buf, c.c1, c.c2, esz);
  };

  // ---- down path ----
  // CFG-shared prefix (round 5; Builder::transformer `shared`): the batch is [uncond; cond] with identical latents in both halves, so
  // conv_in, the first ResBlock and the first transformer up to its cross-attention are computed for B2 / 2 samples only.  conv_in's
  // output is also skip 0 -- the last up-block ResBlock reads it for all B2 samples -- so its second half is filled by a row copy.
  const bool shared = P.cfg_shared;
  const int Bh = B2 / 2;
  TView h = skip_dst();                                    // conv_in output = skip 0
  b.conv3x3(xin, shared ? Bh : B2, HH(0), WW(0), HH(0), WW(0), 1, 0, conv_in, h);
  if (shared) {
    const int d = dt, c = ch[0];
    const int64_t rh = rows_at(0) / 2, ld = h.ld;
    const void* src = h.p; void* dst = (char*)h.p + (size_t)rh * ld * esz;
    b.push([=](hipStream_t s) { return launch_copy_rows(d, src, ld, dst, ld, rh, c, s); }, "copy skip0 rows=" + std::to_string(rh));
  }
  for (int i = 0; i < nl; ++i) {
    for (int j = 0; j < lpb; ++j) {
      const bool xa = !down[i].xf.empty();
      if (xa && shared && i == 0 && j == 0) {
        TView r = b.alloc(rows_at(0) / 2, ch[0]);
        b.resblock(down[0].res[0], h, Bh, HH(0), WW(0), G, r);
        TView o = skip_dst();
        xf_site(down[0].xf[0], nullptr, nullptr, r, 0, o, true);
        h = o;
      } else if (xa) {
        TView r = b.alloc(rows_at(i), ch[i]);
        b.resblock(down[i].res[j], h, B2, HH(i), WW(i), G, r);
        TView o = skip_dst();
        xf_site(down[i].xf[j], music ? &down[i].xf2[j] : nullptr, music ? &down[i].xf3[j] : nullptr, r, i, o);
        h = o;
      } else {
        TView o = skip_dst();
        b.resblock(down[i].res[j], h, B2, HH(i), WW(i), G, o);
        h = o;
      }
    }
    if (down[i].has_ds) {
      TView o = skip_dst();
      b.conv3x3(h, B2, HH(i + 1), WW(i + 1), HH(i), WW(i), 2, 0, down[i].ds, o);
      h = o;
    }
  }
  // ---- mid ----
  {
    const int lvl = nl - 1;
    TView r0 = b.alloc(rows_at(lvl), ch[lvl]);
    b.resblock(mid_res0, h, B2, HH(lvl), WW(lvl), G, r0);
    TView x1 = b.alloc(rows_at(lvl), ch[lvl]);
    xf_site(mid_xf, music ? &mid_xf2 : nullptr, music ? &mid_xf3 : nullptr, r0, lvl, x1);
    TView dst = Builder::slice(cats[0].buf, 0, cats[0].c1, esz);
    b.resblock(mid_res1, x1, B2, HH(lvl), WW(lvl), G, dst);
  }
  // ---- up path ----
  TView last;
  for (int i = 0; i < nl; ++i) {
    const int lvl = nl - 1 - i;
    const int outc = ch[lvl];
    const bool xa = !up[i].xf.empty();
    for (int j = 0; j <= lpb; ++j) {
      const int u = i * (lpb + 1) + j;
      Cat& c = cats[u];
      // where does this layer's output go?
      TView dst;
      const bool last_in_block = (j == lpb);
      if (!last_in_block) dst = Builder::slice(cats[u + 1].buf, 0, cats[u + 1].c1, esz);
      else dst = b.alloc(rows_at(lvl), outc);              // feeds the upsampler conv / conv_norm_out
      if (xa) {
        TView r = b.alloc(rows_at(lvl), outc);
        b.resblock(up[i].res[j], c.buf, B2, HH(lvl), WW(lvl), G, r);
        xf_site(up[i].xf[j], music ? &up[i].xf2[j] : nullptr, music ? &up[i].xf3[j] : nullptr, r, lvl, dst);
      } else {
        b.resblock(up[i].res[j], c.buf, B2, HH(lvl), WW(lvl), G, dst);
      }
      last = dst;
    }
    if (up[i].has_us) {
      // Upsample2D (resnet.py:95-161): nearest x2 folded into the conv gather
      TView dst = Builder::slice(cats[(i + 1) * (lpb + 1)].buf, 0, cats[(i + 1) * (lpb + 1)].c1, esz);
      b.conv3x3(last, B2, HH(lvl - 1), WW(lvl - 1), HH(lvl), WW(lvl), 1, 1, up[i].us, dst);
    }
  }
  // ---- out ----
  {
    TView t = b.alloc(rows_at(0), ch[0]);
    b.groupnorm(last, B2, HH(0) * WW(0), norm_out, G, ACT_SILU, t);
    TView o; o.p = P.eps; o.ld = cfg.unet_out_channels; o.C = cfg.unet_out_channels;
    GOpt go; go.out_f32 = true;
    b.conv3x3(t, B2, HH(0), WW(0), HH(0), WW(0), 1, 0, conv_out, o, go);
  }
  return 0;
}

int Engine::make_unet_plan(UNetPlan& P) {
  Arena m;
  TANGO_TRY(build_unet(P, m, false));
  TANGO_TRY(alloc_slab(&P.slab, m.peak + 256, P.meta, true));   // zero-filled: V^T pad columns (L not a multiple of 8) must stay zero
  Arena a; a.base = P.slab;
  P.pre.ops.clear(); P.step.ops.clear(); P.pre.labels.clear(); P.step.labels.clear(); P.pre.flops.clear(); P.step.flops.clear();
  if (int rc = build_unet(P, a, true)) { release_slab(&P.slab, P.meta); return rc; }
  return 0;
}

int Engine::get_unet_plan(int B2, int L, int Lbeat, int Lchord, int n_short, UNetPlan** out, int chains) {
  if (cfg.unet_music && (Lbeat <= 0 || Lchord <= 0)) TANGO_FAIL("engine: the Music UNet needs beat and chord conditions (beat_len, chord_len > 0)");
  if (!cfg.unet_music) { Lbeat = 0; Lchord = 0; }
  if (n_short < 0 || n_short > B2) TANGO_FAIL("engine: bad single-key prefix");
  // `chains` doubles as the plan's mode: 1 one program for the whole batch, 2 two half-batch chains, 3 one program with the CFG-shared prefix
  if (chains != 1 && ((chains != 2 && chains != 3) || B2 % 2 != 0)) TANGO_FAIL("engine: dual / CFG-shared plans need an even batch");
  if (chains == 3 && !cfg_shared_ok(B2, n_short)) TANGO_FAIL("engine: the CFG-shared prefix needs a [single-key; text] batch and a plain UNet");
  const std::array<int, 6> key = {B2, L, Lbeat, Lchord, n_short, chains};
  auto it = unet_plans.find(key);
  if (it != unet_plans.end()) { touch(it->second->meta); *out = it->second.get(); return 0; }
  if (!finalized) TANGO_FAIL("engine: weights not finalized");
  std::unique_ptr<UNetPlan> P(new UNetPlan());
  P->B2 = B2; P->L = L; P->n_short = n_short;
  P->Lc[0] = L; P->Lc[1] = Lbeat; P->Lc[2] = Lchord;
  if (chains == 1 || chains == 3) {
    P->cfg_shared = chains == 3;
    TANGO_TRY(make_unet_plan(*P));
  } else {
    // the parent: input and output of the whole batch (what the scheduler kernel reads and writes), nothing else
    const int HW = cfg.latent_h * cfg.latent_w, hb = B2 / 2;
    const int cin_pad = 8 > cfg.unet_in_channels ? 8 : ((cfg.unet_in_channels + 7) / 8) * 8;
    const size_t xin_bytes = ((size_t)B2 * HW * cin_pad * esz + 255) & ~(size_t)255, eps_bytes = (size_t)B2 * HW * cfg.unet_out_channels * 4;
    TANGO_TRY(alloc_slab(&P->slab, xin_bytes + eps_bytes + 256, P->meta, true));
    P->xin = P->slab;
    P->eps = (float*)(P->slab + xin_bytes);
    for (int k = 0; k < 2; ++k) {
      std::unique_ptr<UNetPlan> C(new UNetPlan());
      C->B2 = hb; C->L = L;
      C->Lc[0] = L; C->Lc[1] = Lbeat; C->Lc[2] = Lchord;
      C->n_short = k == 0 ? (n_short < hb ? n_short : hb) : (n_short > hb ? n_short - hb : 0);
      C->ext_xin = (char*)P->xin + (size_t)k * hb * HW * cin_pad * esz;
      C->ext_eps = P->eps + (size_t)k * hb * HW * cfg.unet_out_channels;
      C->sync = k == 0 ? d_sync : d_sync2;
      if (int rc = make_unet_plan(*C)) { free_unet_plan(*P); return rc; }
      P->child[k] = std::move(C);
    }
  }
  *out = P.get();
  unet_plans[key] = std::move(P);
  return 0;
}

int Engine::bind_text(UNetPlan& P, const Cond (&c)[3], const std::vector<int>& key0, hipStream_t s) {
  const int ncond = cfg.unet_music ? 3 : 1;
  if (P.n_short > 0) {
    if ((int)key0.size() < P.n_short) TANGO_FAIL("engine: single-key prefix without key indices");
    TANGO_TRY(stage_h2d(P.key0, key0.data(), (size_t)P.n_short * 4, s));   // key0 is transient host memory
  }
  for (int i = 0; i < ncond; ++i) {
    if (!c[i].emb) TANGO_FAIL("engine: missing condition embeddings");
    const int n = P.B2 * P.Lc[i];
    TANGO_TRY(launch_cast_rows(dt, c[i].emb, P.encs[i], cfg.unet_cross_dim, n, cfg.unet_cross_dim, s));
    if (c[i].mask) TANGO_TRY(launch_mask_bias(c[i].mask, P.biases[i], n, s));
    else TANGO_TRY(launch_fill_zero(P.biases[i], (size_t)n * 4, s));
  }
  return P.pre.run(s);
}

// How many leading samples keep exactly ONE text key (and which): the unconditional half of a CFG batch is T5("") = one valid token
// (models.py:282-289).  One small device -> host copy per call.  Only the two shapes a CFG caller produces get their own plan --
// the whole batch, or exactly its first half -- so that odd masks cannot multiply the (large) plans.
int Engine::single_key_prefix(const uint8_t* mask_dev, const uint8_t* mask_host, int B2, int L, std::vector<int>& key0, hipStream_t s) {
  key0.clear();
  if ((!mask_dev && !mask_host) || tuning().no_single_key || B2 <= 0 || L <= 0) return 0;
  std::vector<uint8_t> mbuf;
  const uint8_t* m = mask_host;          // the caller's host copy of the mask (tokenizer output): no read-back, no host sync
  if (!m) {
    mbuf.resize((size_t)B2 * L);
    if (hipMemcpyAsync(mbuf.data(), mask_dev, mbuf.size(), hipMemcpyDeviceToHost, s) != hipSuccess) return 0;
    if (hipStreamSynchronize(s) != hipSuccess) return 0;
    m = mbuf.data();
  }
  for (int b = 0; b < B2; ++b) {
    int cnt = 0, idx = 0;
    for (int j = 0; j < L; ++j) if (m[(size_t)b * L + j]) { ++cnt; idx = j; }
    if (cnt != 1) break;
    key0.push_back(idx);
  }
  const int n = (int)key0.size();
  if (n >= B2) return B2;
  if (B2 % 2 == 0 && n >= B2 / 2) return B2 / 2;
  return 0;
}

int Engine::unet_forward(const float* sample, int64_t t, const Cond (&c)[3], float* out, int B2, hipStream_t s) {
  UNetPlan* P;
  std::vector<int> key0;
  const int ns = single_key_prefix(c[0].mask, c[0].mask_host, B2, c[0].len, key0, s);
  TANGO_TRY(get_unet_plan(B2, c[0].len, c[1].len, c[2].len, ns, &P));
  TANGO_TRY(ensure_temb(&t, 1, s));
  TANGO_HIP(hipMemsetAsync(d_step, 0, 4, s));
  TANGO_TRY(bind_text(*P, c, key0, s));
  const int HW = cfg.latent_h * cfg.latent_w;
  TANGO_TRY(launch_fill_zero(P->xin, (size_t)B2 * HW * 8 * esz, s));
  TANGO_TRY(launch_nchw_to_nhwc(dt, sample, P->xin, 8, B2, cfg.unet_in_channels, HW, 1, 1.0f, s));
  TANGO_TRY(P->step.run(s));
  TANGO_TRY(launch_nhwc_to_nchw_f32(DT_F32, P->eps, cfg.unet_out_channels, out, B2, cfg.unet_out_channels, HW, s));
  return 0;
}

// steps per captured graph when TANGO_GRAPH_STEPS is unset (measured, round 5: see DESIGN.md section 5)
static int graph_steps_default(int B2) { (void)B2; return 1; }

// One chain or two (round 5)?  Two: the first and the second half of the UNet batch (with guidance: the unconditional and the
// conditional rows) run as two independent kernel sequences -- two branches of the captured graph -- whose dispatch gaps and kernel
// tails fill each other.  TANGO_UNET_CHAINS = 1 / 2 forces; unset = the measured rule.
// May a guidance batch of B2 UNet rows whose first n_short rows are single-key run with the CFG-shared prefix (build_unet)?  The rows
// b and B2 / 2 + b must carry identical latents -- true inside denoise() with guidance, which fills both halves from one tensor --, the
// first half must be the single-key (unconditional) half, and the first down block must be a plain cross-attention block.
bool Engine::cfg_shared_ok(int B2, int n_short) const {
  return !tuning().no_cfg_shared && B2 >= 2 && B2 % 2 == 0 && n_short == B2 / 2 && !cfg.unet_music && cfg.unet_levels > 0 &&
         cfg.unet_cross_attn[0] != 0 && cfg.unet_layers_per_block >= 1;
}

int Engine::unet_chains_for(int B2) const {
  if (B2 < 2 || B2 % 2 != 0) return 1;
  const int t = tuning().unet_chains;
  if (t == 1 || t == 2) return t;
  return 1;
}

int Engine::denoise(const tango_denoise_args_t& a, hipStream_t s) {
  if (a.num_steps <= 0) TANGO_FAIL("denoise: num_steps must be positive");
  const bool cfg_on = a.guidance_scale > 1.0f;
  const int B = a.batch, B2 = cfg_on ? 2 * B : B;
  UNetPlan* P;
  std::vector<int> key0;
  const int ns = single_key_prefix(a.prompt_mask, a.prompt_mask_host, B2, a.text_len, key0, s);
  int chains = unet_chains_for(B2);
  if (chains == 1 && cfg_on && cfg_shared_ok(B2, ns)) chains = 3;      // one program, CFG-shared prefix (mode 3 of get_unet_plan)
  TANGO_TRY(get_unet_plan(B2, a.text_len, a.beat_len, a.chord_len, ns, &P, chains));
  UNetPlan* const CA = chains == 2 ? P->child[0].get() : nullptr;
  UNetPlan* const CB = chains == 2 ? P->child[1].get() : nullptr;
  TANGO_TRY(ensure_temb(a.timesteps, a.num_steps, s));
  const int HW = cfg.latent_h * cfg.latent_w;
  const int C = cfg.unet_in_channels;
  SchedParams sp;
  sp.lat = a.latents; sp.eps = P->eps; sp.xin = P->xin; sp.xin_ld = 8;
  sp.noise = a.noise; sp.coef = d_coef; sp.step_ptr = d_step;
  sp.B = B; sp.C = C; sp.HW = HW; sp.cfg = cfg_on ? 1 : 0; sp.guidance = a.guidance_scale;
  sp.pred_type = a.prediction_type; sp.rule = a.rule; sp.clip = a.clip_sample; sp.clip_range = a.clip_sample_range;
  sp.seed = a.seed; sp.sample_offset = a.sample_offset;
  // per-call tables and the scheduler parameter block live in device memory, so the captured graph of one denoise
  // step (UNet + CFG/scheduler update + step counter) is independent of the call's pointers and scalars
  // (both sources are pageable / transient host memory: pinned staging slots, no host sync -- stage_h2d)
  TANGO_TRY(stage_h2d(d_coef, a.coef, (size_t)a.num_steps * 8 * 4, s));
  TANGO_TRY(stage_h2d(d_sched, &sp, sizeof(SchedParams), s));
  TANGO_HIP(hipMemsetAsync(d_step, 0, 4, s));
  {
    Cond c[3];
    c[0].emb = a.prompt_embeds; c[0].mask = a.prompt_mask; c[0].len = a.text_len;
    c[1].emb = a.beat_embeds; c[1].mask = a.beat_mask; c[1].len = a.beat_len;
    c[2].emb = a.chord_embeds; c[2].mask = a.chord_mask; c[2].len = a.chord_len;
    if (chains != 2) {
      TANGO_TRY(bind_text(*P, c, key0, s));
    } else {
      // each chain binds its own half of the conditions (rows [0, B2/2) and [B2/2, B2) of every [B2, L, d] tensor)
      const int hb = B2 / 2;
      for (int k = 0; k < 2; ++k) {
        UNetPlan& Ck = *P->child[k];
        Cond ck[3];
        for (int i = 0; i < 3; ++i) {
          ck[i] = c[i];
          if (c[i].emb) ck[i].emb = c[i].emb + (size_t)k * hb * c[i].len * cfg.unet_cross_dim;
          if (c[i].mask) ck[i].mask = c[i].mask + (size_t)k * hb * c[i].len;
          ck[i].mask_host = nullptr;
        }
        std::vector<int> k0;
        if (Ck.n_short > 0) k0.assign(key0.begin() + (size_t)k * hb, key0.begin() + (size_t)k * hb + Ck.n_short);
        TANGO_TRY(bind_text(Ck, ck, k0, s));
      }
    }
  }
  TANGO_TRY(launch_fill_zero(P->xin, (size_t)B2 * HW * 8 * esz, s));
  TANGO_TRY(launch_nchw_to_nhwc(dt, a.latents, P->xin, 8, B, C, HW, cfg_on ? 2 : 1, 1.0f, s));

  if (chains == 2 && !ev_fork) {
    TANGO_HIP(hipStreamCreateWithFlags(&cap_stream2, hipStreamNonBlocking));
    TANGO_HIP(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
    for (hipEvent_t* e : {&ev_fork, &ev_join, &ev_fork_e, &ev_join_e}) TANGO_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  // one denoise step = UNet forward, fused CFG combine + scheduler update (writes the next UNet input), step counter++
  // (two chains: fork after the previous step's update, join before this step's)
  auto run_step = [&](hipStream_t st, hipStream_t st2, hipEvent_t ef, hipEvent_t ej) -> int {
    if (chains != 2) {
      TANGO_TRY(P->step.run(st));
    } else {
      TANGO_HIP(hipEventRecord(ef, st));
      TANGO_HIP(hipStreamWaitEvent(st2, ef, 0));
      TANGO_TRY(CA->step.run(st));
      TANGO_TRY(CB->step.run(st2));
      TANGO_HIP(hipEventRecord(ej, st2));
      TANGO_HIP(hipStreamWaitEvent(st, ej, 0));
    }
    TANGO_TRY(launch_sched_step(dt, d_sched, B2 * HW, st));
    return launch_step_inc(d_step, st);
  };
  // capture `n` steps once per plan; every per-step quantity is read through d_step / d_sched
  // (captured on an engine-owned stream: the caller's stream may be the legacy null stream,
  // which cannot be captured; the instantiated graph is then launched on the caller's stream)
  auto capture = [&](int n, hipGraph_t* g_out, hipGraphExec_t* x_out) -> int {
    if (!cap_stream) TANGO_HIP(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
    TANGO_HIP(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeRelaxed));
    int rc = 0;
    for (int i = 0; i < n && rc == 0; ++i) rc = run_step(cap_stream, cap_stream2, ev_fork, ev_join);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(cap_stream, &g);
    if (rc != 0) return rc;
    if (e != hipSuccess) TANGO_FAIL(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    *g_out = g;
    TANGO_HIP(hipGraphInstantiate(x_out, g, nullptr, nullptr, 0));
    return 0;
  };
  if (a.use_graph && !P->exec) TANGO_TRY(capture(1, &P->graph, &P->exec));
  // k steps per replay (round 5; north_star: "the 100-200 denoise steps captured as a hipGraph"): the same kernel sequence captured k
  // times back to back -- every per-step quantity is read through the device-side step counter, so a k-step graph is just k copies.
  // What it removes is the host-side launch of every replay and the gap between two replays, which only shows at small batches
  // (B = 1: a 7-ms step); the remainder of num_steps / k runs on the one-step graph.
  int kk = tuning().graph_steps > 0 ? tuning().graph_steps : graph_steps_default(B2);
  if (kk > a.num_steps) kk = a.num_steps;
  if (a.use_graph && kk > 1 && (!P->exec_k || P->k_steps != kk)) {
    if (P->exec_k) { TANGO_HIP(hipStreamSynchronize(s)); (void)hipGraphExecDestroy(P->exec_k); (void)hipGraphDestroy(P->graph_k); P->exec_k = nullptr; P->graph_k = nullptr; }
    TANGO_TRY(capture(kk, &P->graph_k, &P->exec_k));
    P->k_steps = kk;
  }
  TANGO_HIP(hipEventRecord(ev0, s));
  {
    int i = 0;
    if (a.use_graph && kk > 1)
      for (; i + kk <= a.num_steps; i += kk) TANGO_HIP(hipGraphLaunch(P->exec_k, s));
    for (; i < a.num_steps; ++i) {
      if (a.use_graph) TANGO_HIP(hipGraphLaunch(P->exec, s));
      else TANGO_TRY(run_step(s, aux_stream, ev_fork_e, ev_join_e));
    }
  }
  TANGO_HIP(hipEventRecord(ev1, s));
  last_steps = a.num_steps;
  last_step_gflop = 0.0;
  for (const UNetPlan* q : {chains != 2 ? (const UNetPlan*)P : (const UNetPlan*)CA, (const UNetPlan*)CB})
    if (q) for (double f : q->step.flops) last_step_gflop += f / 1e9;
  return 0;
}

int Program::run_profiled(hipStream_t s, std::string& report) const {
  hipEvent_t a, b;
  TANGO_HIP(hipEventCreate(&a));
  TANGO_HIP(hipEventCreate(&b));
  for (size_t i = 0; i < ops.size(); ++i) {
    TANGO_HIP(hipEventRecord(a, s));
    TANGO_TRY(ops[i](s));
    TANGO_HIP(hipEventRecord(b, s));
    TANGO_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    TANGO_HIP(hipEventElapsedTime(&ms, a, b));
    char buf[64];
    snprintf(buf, sizeof buf, "\t%.4f\t%.3f\n", ms, i < flops.size() ? flops[i] / 1e9 : 0.0);
    report += (i < labels.size() ? labels[i] : std::string("op")) + buf;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return 0;
}

int Engine::profile_unet(int B2, int L, std::string& report, hipStream_t s) {
  UNetPlan* P;
  // the product's CFG structure: the first half of the batch is the unconditional (single-key) half
  const int ns = (!tuning().no_single_key && B2 % 2 == 0) ? B2 / 2 : 0;
  // (and the plan denoise() would run for it: with the CFG-shared prefix where that applies)
  TANGO_TRY(get_unet_plan(B2, L, cfg.unet_music ? 50 : 0, cfg.unet_music ? 20 : 0, ns, &P, cfg_shared_ok(B2, ns) ? 3 : 1));   // mustango/models.py:336,340: beat_len 50, chord_len 20
  int64_t t = 500;
  TANGO_TRY(ensure_temb(&t, 1, s));
  TANGO_HIP(hipMemsetAsync(d_step, 0, 4, s));
  TANGO_TRY(P->step.run(s));                 // warm-up (inputs are whatever the buffers hold)
  TANGO_HIP(hipStreamSynchronize(s));
  return P->step.run_profiled(s, report);
}

// per-op tables of the two plans that follow the denoise loop in every pass (they had no per-op profile); inputs
// are whatever the plan buffers hold -- only the times matter
int Engine::profile_vae(int B, std::string& report, hipStream_t s) {
  if (cfg.vae_levels <= 0) TANGO_FAIL("engine: VAE not configured");
  VaePlan* P;
  TANGO_TRY(get_vae_plan(B, &P));
  TANGO_TRY(P->prog.run(s));
  TANGO_HIP(hipStreamSynchronize(s));
  return P->prog.run_profiled(s, report);
}
int Engine::profile_vocoder(int B, int frames, std::string& report, hipStream_t s) {
  if (cfg.voc_n_ups <= 0) TANGO_FAIL("engine: vocoder not configured");
  VaePlan* P;
  TANGO_TRY(get_voc_plan(B, frames, &P));
  TANGO_TRY(P->prog.run(s));
  TANGO_HIP(hipStreamSynchronize(s));
  return P->prog.run_profiled(s, report);
}

int Engine::last_denoise_ms(float* total_ms, float* per_step_ms) {
  if (last_steps <= 0) TANGO_FAIL("last_denoise_ms: no denoise call recorded");
  TANGO_HIP(hipEventSynchronize(ev1));
  float ms = 0.f;
  TANGO_HIP(hipEventElapsedTime(&ms, ev0, ev1));
  if (total_ms) *total_ms = ms;
  if (per_step_ms) *per_step_ms = ms / (float)last_steps;
  // the work is complete here.  A cooperative kernel whose partners did not all show up in time (norm.hip gn_coop_kernel: the GPU was
  // shared) recomputed the missing partial sums itself -- same bits, only slower -- and counted it: report once per occurrence, reset
  // (both chains' counter words: TANGO_UNET_CHAINS=2 runs the second half-batch chain on d_sync2)
  unsigned n1 = 0, n2 = 0;
  TANGO_HIP(hipMemcpy(&n1, d_sync + 2 * COOP_SYNC_SLOTS, 4, hipMemcpyDeviceToHost));
  TANGO_HIP(hipMemcpy(&n2, d_sync2 + 2 * COOP_SYNC_SLOTS, 4, hipMemcpyDeviceToHost));
  if (n1) TANGO_HIP(hipMemset(d_sync + 2 * COOP_SYNC_SLOTS, 0, 4));
  if (n2) TANGO_HIP(hipMemset(d_sync2 + 2 * COOP_SYNC_SLOTS, 0, 4));
  const unsigned n = n1 + n2;
  if (n) {
    coop_fallbacks += n;
    if (!tuning().gn_coop_force_fb)
      fprintf(stderr, "tango: %u workgroup(s) of the cooperative GroupNorm took the no-rendezvous fallback (GPU shared with other work?); "
                      "results are unaffected, TANGO_NO_GN_COOP=1 avoids the slow path\n", n);
  }
  return 0;
}

// ================================================================================================
// FLAN-T5 encoder (transformers models/t5/modeling_t5.py T5Stack encoder; reference call sites models.py:98-100,129-147):
// embedding -> 24 x [RMSNorm -> self-attention with shared relative position bias (no 1/sqrt(d) scaling) -> +residual,
// RMSNorm -> gated-GELU feed-forward -> +residual] -> final RMSNorm.  No biases, no dropout at inference.
// ================================================================================================
void Engine::build_t5_weights() {
  const int d = cfg.t5_d_model, inner = cfg.t5_heads * cfg.t5_d_kv, dff = cfg.t5_d_ff;
  const std::string P = "text_encoder.";
  t5_embed = (float*)dmalloc((size_t)cfg.t5_vocab * d * 4);
  {
    float* dst = t5_embed;
    const size_t n = (size_t)cfg.t5_vocab * d * 4;
    reg_slot(P + "shared.weight", {cfg.t5_vocab, d}, [dst, n](const float* src, hipStream_t s) {
      TANGO_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s));
      return 0;
    });
  }
  t5_rel_table = (float*)dmalloc((size_t)cfg.t5_rel_buckets * cfg.t5_heads * 4);
  {
    float* dst = t5_rel_table;
    const size_t n = (size_t)cfg.t5_rel_buckets * cfg.t5_heads * 4;
    reg_slot(P + "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", {cfg.t5_rel_buckets, cfg.t5_heads},
             [dst, n](const float* src, hipStream_t s) {
               TANGO_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s));
               return 0;
             });
  }
  t5_layers.resize(cfg.t5_layers);
  for (int i = 0; i < cfg.t5_layers; ++i) {
    T5LayerW& w = t5_layers[i];
    const std::string b = P + "encoder.block." + std::to_string(i);
    w.ln1.C = d; w.ln1.eps = cfg.t5_eps;
    reg_vec(b + ".layer.0.layer_norm.weight", d, &w.ln1.g);
    const std::string a = b + ".layer.0.SelfAttention";
    w.qkv.N = 3 * inner; w.qkv.K = d; w.qkv.Cin = d; w.qkv.Kp = d; w.qkv.taps = 1;
    w.qkv.W = dmalloc((size_t)3 * inner * d * esz);
    reg_mat(a + ".q.weight", inner, d, w.qkv, false, 0, {inner, d});
    reg_mat(a + ".k.weight", inner, d, w.qkv, false, inner, {inner, d});
    reg_mat(a + ".v.weight", inner, d, w.qkv, false, 2 * inner, {inner, d});
    reg_linear(a + ".o", d, inner, w.o, false);
    w.ln2.C = d; w.ln2.eps = cfg.t5_eps;
    reg_vec(b + ".layer.1.layer_norm.weight", d, &w.ln2.g);
    // gated feed-forward: hidden = gelu_new(wi_0 x) * (wi_1 x) -> GLU interleave with wi_1 as the value, wi_0 as the gate
    const std::string f = b + ".layer.1.DenseReluDense";
    w.wi.N = 2 * dff; w.wi.K = d; w.wi.Cin = d; w.wi.Kp = d; w.wi.taps = 1;
    w.wi.W = dmalloc((size_t)2 * dff * d * esz);
    {
      void* W = w.wi.W;
      const int dt_ = dt;
      reg_slot(f + ".wi_1.weight", {dff, d}, [=](const float* src, hipStream_t s) { return launch_pack(dt_, src, W, dff, 1, d, d, 0, 1, d, -2, s); });
      reg_slot(f + ".wi_0.weight", {dff, d}, [=](const float* src, hipStream_t s) { return launch_pack(dt_, src, W, dff, 1, d, d, 0, 1, d, -3, s); });
    }
    reg_linear(f + ".wo", d, dff, w.wo, false);
  }
  t5_final_ln.C = d; t5_final_ln.eps = cfg.t5_eps;
  reg_vec(P + "encoder.final_layer_norm.weight", d, &t5_final_ln.g);
}

// T5Attention._relative_position_bucket (bidirectional), float32 arithmetic in the reference's operation order
static int t5_bucket(int rel /* memory - query */, int num_buckets, int max_distance) {
  int nb = num_buckets / 2;
  int ret = rel > 0 ? nb : 0;
  int rp = rel < 0 ? -rel : rel;
  const int max_exact = nb / 2;
  if (rp < max_exact) return ret + rp;
  const float v = logf((float)rp / (float)max_exact) / (float)std::log((double)max_distance / (double)max_exact) * (float)(nb - max_exact);
  int large = max_exact + (int)v;
  if (large > nb - 1) large = nb - 1;
  return ret + large;
}

int Engine::build_t5(T5Plan& P, Arena& A, bool record) {
  Builder b{*this, A, &P.prog, record, dt, esz};
  const int B = P.B, L = P.L, d = cfg.t5_d_model, H = cfg.t5_heads, inner = H * cfg.t5_d_kv, dff = cfg.t5_d_ff;
  const int64_t rows = (int64_t)B * L;
  const int Lp = (L + 7) / 8 * 8;
  P.ids = (int64_t*)A.alloc((size_t)rows * 8);
  P.bias = (float*)A.alloc((size_t)rows * 4);
  P.bucket = (int*)A.alloc((size_t)L * L * 4);
  P.pos_bias = (float*)A.alloc((size_t)H * L * L * 4);
  P.out = (float*)A.alloc((size_t)rows * d * 4);
  TView h = b.alloc(rows, d), n = b.alloc(rows, d), h2 = b.alloc(rows, d);
  TView qk = b.alloc(rows, 2 * inner);
  void* vt = A.alloc((size_t)B * inner * Lp * esz);     // V^T [B][inner][Lp]; pad columns stay zero (slab is zero-filled)
  TView att = b.alloc(rows, inner);
  TView gg = b.alloc(rows, dff);
  const int d_ = dt;
  {
    const int64_t* ids = P.ids; const float* tab = t5_embed; void* hp = h.p; const int V = cfg.t5_vocab; const int r = (int)rows;
    b.push([=](hipStream_t s) { return launch_embed_gather(d_, ids, tab, hp, d, r, d, V, s); }, "t5 embedding");
  }
  auto rms = [&](const TView& x, const WNorm& w, void* y, int64_t ldy, int out_f32) {
    const void* xp = x.p; const int64_t ldx = x.ld; const float* g = w.g; const float eps = w.eps; const int r = (int)rows;
    b.push([=](hipStream_t s) { return launch_rmsnorm(d_, xp, ldx, y, ldy, g, r, d, eps, out_f32, s); }, "t5 rmsnorm");
  };
  for (int i = 0; i < cfg.t5_layers; ++i) {
    const T5LayerW& w = t5_layers[i];
    rms(h, w.ln1, n.p, n.ld, 0);
    { GOpt o; o.use_bias = false; o.vt = vt; o.vt_n0 = 2 * inner; o.vt_S = L; o.vt_ld = Lp; b.linear(n, rows, w.qkv, qk, o); }
    b.attention(Builder::slice(qk, 0, inner, esz), Builder::slice(qk, inner, inner, esz), vt, Lp, att, P.bias, B, H, L, L, 1.0f, P.pos_bias);
    { GOpt o; o.use_bias = false; o.residual = &h; b.linear(att, rows, w.o, h2, o); }
    rms(h2, w.ln2, n.p, n.ld, 0);
    { GOpt o; o.use_bias = false; o.epi = EPI_GEGLU; o.glu_tanh = 1; b.linear(n, rows, w.wi, gg, o); }
    { GOpt o; o.use_bias = false; o.residual = &h2; b.linear(gg, rows, w.wo, h, o); }
  }
  rms(h, t5_final_ln, P.out, d, 1);
  return 0;
}

int Engine::get_t5_plan(int B, int L, T5Plan** out) {
  auto key = std::make_pair(B, L);
  auto it = t5_plans.find(key);
  if (it != t5_plans.end()) { touch(it->second->meta); *out = it->second.get(); return 0; }
  if (!finalized) TANGO_FAIL("engine: weights not finalized");
  std::unique_ptr<T5Plan> P(new T5Plan());
  P->B = B; P->L = L;
  Arena m;
  TANGO_TRY(build_t5(*P, m, false));
  TANGO_TRY(alloc_slab(&P->slab, m.peak + 256, P->meta, true));
  Arena a; a.base = P->slab;
  P->prog.ops.clear(); P->prog.labels.clear(); P->prog.flops.clear();
  if (int rc = build_t5(*P, a, true)) { release_slab(&P->slab, P->meta); return rc; }
  std::vector<int> bk((size_t)L * L);
  for (int i = 0; i < L; ++i)
    for (int j = 0; j < L; ++j) bk[(size_t)i * L + j] = t5_bucket(j - i, cfg.t5_rel_buckets, cfg.t5_rel_max_distance);
  TANGO_HIP(hipMemcpy(P->bucket, bk.data(), bk.size() * 4, hipMemcpyHostToDevice));
  *out = P.get();
  t5_plans[key] = std::move(P);
  return 0;
}

int Engine::encode_text(const int64_t* ids, const uint8_t* mask, float* out, int B, int L, hipStream_t s) {
  if (cfg.t5_layers <= 0) TANGO_FAIL("engine: text encoder not configured");
  if (B <= 0 || L <= 0) TANGO_FAIL("encode_text: empty batch");
  T5Plan* P;
  TANGO_TRY(get_t5_plan(B, L, &P));
  TANGO_HIP(hipMemcpyAsync(P->ids, ids, (size_t)B * L * 8, hipMemcpyDeviceToDevice, s));
  if (mask) TANGO_TRY(launch_mask_bias(mask, P->bias, B * L, s));
  else TANGO_TRY(launch_fill_zero(P->bias, (size_t)B * L * 4, s));
  TANGO_TRY(launch_t5_pos_bias(t5_rel_table, P->bucket, P->pos_bias, cfg.t5_heads, L, s));   // table may have been reloaded
  TANGO_TRY(P->prog.run(s));
  TANGO_HIP(hipMemcpyAsync(out, P->out, (size_t)B * L * cfg.t5_d_model * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}

// ================================================================================================
// mel-VAE decoder plan (autoencoder.py:116-124,60-64; modules.py:650-683)
// ================================================================================================
int Engine::build_vae(VaePlan& P, Arena& A, bool record) {
  Builder b{*this, A, &P.prog, record, dt, esz};
  const int B = P.B, nl = cfg.vae_levels;
  const int H0 = cfg.latent_h, W0 = cfg.latent_w, HW0 = H0 * W0;
  const int zc = cfg.vae_z_channels;
  P.in_bytes = (size_t)B * cfg.vae_embed_dim * HW0 * 4;
  P.in = A.alloc(P.in_bytes);
  const int Hf = H0 << (nl - 1), Wf = W0 << (nl - 1);
  P.out_bytes = (size_t)B * Hf * Wf * cfg.vae_out_ch * 4;
  P.out = A.alloc(P.out_bytes);

  TView z = b.alloc((int64_t)B * HW0, 8);
  {
    const int d = dt; const float* src = (const float*)P.in; const float* W = pqc_w; const float* bb = pqc_b; void* dst = z.p;
    const int ed = cfg.vae_embed_dim; const float sc = 1.0f / cfg.vae_scale_factor;
    b.push([=](hipStream_t s) { return launch_pointwise_small(d, src, W, bb, dst, 8, B, ed, zc, HW0, sc, s); });
  }
  int C = vae_conv_in.N;
  TView h = b.alloc((int64_t)B * HW0, C);
  b.conv3x3(z, B, H0, W0, H0, W0, 1, 0, vae_conv_in, h);
  {
    TView o = b.alloc((int64_t)B * HW0, C);
    b.resblock(vae_mid1, h, B, H0, W0, 32, o);
    h = o;
  }
  h = b.vae_attn(vae_attn, h, B, HW0, C);
  {
    TView o = b.alloc((int64_t)B * HW0, C);
    b.resblock(vae_mid2, h, B, H0, W0, 32, o);
    h = o;
  }
  int Hc = H0, Wc = W0;
  for (int lvl = nl - 1; lvl >= 0; --lvl) {
    for (size_t k = 0; k < vae_up[lvl].res.size(); ++k) {
      const ResW& r = vae_up[lvl].res[k];
      TView o = b.alloc((int64_t)B * Hc * Wc, r.cout);
      b.resblock(r, h, B, Hc, Wc, 32, o);
      h = o; C = r.cout;
    }
    if (vae_up[lvl].has_up) {
      TView o = b.alloc((int64_t)B * Hc * Wc * 4, C);
      b.conv3x3(h, B, Hc * 2, Wc * 2, Hc, Wc, 1, 1, vae_up[lvl].up, o);
      h = o; Hc *= 2; Wc *= 2;
    }
  }
  {
    TView t = b.alloc((int64_t)B * Hc * Wc, C);
    b.groupnorm(h, B, Hc * Wc, vae_norm_out, 32, ACT_SILU, t);
    TView o; o.p = P.out; o.ld = cfg.vae_out_ch; o.C = cfg.vae_out_ch;
    GOpt go; go.out_f32 = true;
    b.conv3x3(t, B, Hc, Wc, Hc, Wc, 1, 0, vae_conv_out, o, go);
  }
  return 0;
}

int Engine::get_vae_plan(int B, VaePlan** out) {
  auto it = vae_plans.find(B);
  if (it != vae_plans.end()) { touch(it->second->meta); *out = it->second.get(); return 0; }
  if (!finalized) TANGO_FAIL("engine: weights not finalized");
  std::unique_ptr<VaePlan> P(new VaePlan());
  P->B = B;
  Arena m;
  TANGO_TRY(build_vae(*P, m, false));
  TANGO_TRY(alloc_slab(&P->slab, m.peak + 256, P->meta, false));
  Arena a; a.base = P->slab;
  P->prog.ops.clear();
  if (int rc = build_vae(*P, a, true)) { release_slab(&P->slab, P->meta); return rc; }
  *out = P.get();
  vae_plans[B] = std::move(P);
  return 0;
}

int Engine::vae_decode(const float* lat, float* mel, int B, hipStream_t s) {
  if (cfg.vae_levels <= 0) TANGO_FAIL("engine: VAE not configured");
  if (cfg.vae_out_ch != 1) TANGO_FAIL("engine: VAE out_ch must be 1 (NHWC == NCHW at the boundary)");
  VaePlan* P;
  TANGO_TRY(get_vae_plan(B, &P));
  TANGO_HIP(hipMemcpyAsync(P->in, lat, P->in_bytes, hipMemcpyDeviceToDevice, s));
  TANGO_TRY(P->prog.run(s));
  TANGO_HIP(hipMemcpyAsync(mel, P->out, P->out_bytes, hipMemcpyDeviceToDevice, s));
  return 0;
}

// ================================================================================================
// mel-VAE encoder plan (autoencoder.py:52-58 -> modules.py:519-543 Encoder.forward -> quant_conv): mel [B, 1, 4H, 4W] fp32 ->
// moments [B, 2 * embed_dim, H, W] fp32 NCHW.  Same operators as the decoder; the Downsample convs are stride-2 gathers with the
// reference's asymmetric (0,1,0,1) zero padding (GemmParams::pad = 0).
// ================================================================================================
int Engine::build_vae_enc(VaePlan& P, Arena& A, bool record) {
  Builder b{*this, A, &P.prog, record, dt, esz};
  const int B = P.B, nl = cfg.vae_levels;
  const int cin = cfg.vae_in_channels > 0 ? cfg.vae_in_channels : 1;
  int Hc = cfg.latent_h << (nl - 1), Wc = cfg.latent_w << (nl - 1);
  const int HW0 = cfg.latent_h * cfg.latent_w;
  P.in_bytes = (size_t)B * cin * Hc * Wc * 4;
  P.in = A.alloc(P.in_bytes);
  P.out_bytes = (size_t)B * 2 * cfg.vae_embed_dim * HW0 * 4;
  P.out = A.alloc(P.out_bytes);
  if (cin != 1) TANGO_FAIL("engine: VAE encoder in_channels must be 1 (NCHW == NHWC at the boundary)");

  TView x = b.alloc((int64_t)B * Hc * Wc, cin);
  {
    const int d = dt; const float* src = (const float*)P.in; void* dst = x.p; const int rows = B * Hc * Wc;
    b.push([=](hipStream_t s) { return launch_cast_rows(d, src, dst, 1, rows, 1, s); });
  }
  int C = vae_enc_conv_in.N;
  TView h = b.alloc((int64_t)B * Hc * Wc, C);
  b.conv3x3(x, B, Hc, Wc, Hc, Wc, 1, 0, vae_enc_conv_in, h);
  for (int lvl = 0; lvl < nl; ++lvl) {
    for (size_t k = 0; k < vae_down[lvl].res.size(); ++k) {
      const ResW& r = vae_down[lvl].res[k];
      TView o = b.alloc((int64_t)B * Hc * Wc, r.cout);
      b.resblock(r, h, B, Hc, Wc, 32, o);
      h = o; C = r.cout;
    }
    if (vae_down[lvl].has_down) {
      TView o = b.alloc((int64_t)B * (Hc / 2) * (Wc / 2), C);
      GOpt go; go.pad = 0;                                  // F.pad(x, (0,1,0,1)) + conv(k3, s2, p0), modules.py:87-91
      b.conv3x3(h, B, Hc / 2, Wc / 2, Hc, Wc, 2, 0, vae_down[lvl].down, o, go);
      h = o; Hc /= 2; Wc /= 2;
    }
  }
  {
    TView o = b.alloc((int64_t)B * HW0, C);
    b.resblock(vae_enc_mid1, h, B, Hc, Wc, 32, o);
    h = o;
  }
  h = b.vae_attn(vae_enc_attn, h, B, HW0, C);
  {
    TView o = b.alloc((int64_t)B * HW0, C);
    b.resblock(vae_enc_mid2, h, B, Hc, Wc, 32, o);
    h = o;
  }
  {
    TView t = b.alloc((int64_t)B * HW0, C);
    b.groupnorm(h, B, HW0, vae_enc_norm_out, 32, ACT_SILU, t);
    const int mc = 2 * cfg.vae_z_channels;
    float* mom = b.alloc_f32((size_t)B * HW0 * mc);
    TView o; o.p = mom; o.ld = mc; o.C = mc;
    GOpt go; go.out_f32 = true;
    b.conv3x3(t, B, Hc, Wc, Hc, Wc, 1, 0, vae_enc_conv_out, o, go);
    const float* W = qc_w; const float* bb = qc_b; float* dst = (float*)P.out; const int co = 2 * cfg.vae_embed_dim;
    b.push([=](hipStream_t s) { return launch_pointwise_out_nchw(mom, mc, W, bb, dst, B, mc, co, HW0, s); });
  }
  return 0;
}

int Engine::get_vae_enc_plan(int B, VaePlan** out) {
  auto it = vae_enc_plans.find(B);
  if (it != vae_enc_plans.end()) { touch(it->second->meta); *out = it->second.get(); return 0; }
  if (!finalized) TANGO_FAIL("engine: weights not finalized");
  std::unique_ptr<VaePlan> P(new VaePlan());
  P->B = B;
  Arena m;
  TANGO_TRY(build_vae_enc(*P, m, false));
  TANGO_TRY(alloc_slab(&P->slab, m.peak + 256, P->meta, false));
  Arena a; a.base = P->slab;
  P->prog.ops.clear();
  if (int rc = build_vae_enc(*P, a, true)) { release_slab(&P->slab, P->meta); return rc; }
  *out = P.get();
  vae_enc_plans[B] = std::move(P);
  return 0;
}

int Engine::vae_encode(const float* mel, float* moments, int B, hipStream_t s) {
  if (cfg.vae_levels <= 0 || !cfg.vae_encoder) TANGO_FAIL("engine: VAE encoder not configured (tango_config.vae_encoder)");
  VaePlan* P;
  TANGO_TRY(get_vae_enc_plan(B, &P));
  TANGO_HIP(hipMemcpyAsync(P->in, mel, P->in_bytes, hipMemcpyDeviceToDevice, s));
  TANGO_TRY(P->prog.run(s));
  TANGO_HIP(hipMemcpyAsync(moments, P->out, P->out_bytes, hipMemcpyDeviceToDevice, s));
  return 0;
}

// ================================================================================================
// HiFi-GAN plan (hifigan/models.py:149-165, ResBlock :96-103) on channels-last [B, L, C]
// ================================================================================================
int Engine::vocoder_samples(int frames) const {
  int L = frames;
  for (int i = 0; i < cfg.voc_n_ups; ++i) {
    const int u = cfg.voc_rates[i], k = cfg.voc_kernels[i], p = (k - u) / 2;
    L = (L - 1) * u - 2 * p + k;
  }
  return L;
}

int Engine::build_voc(VaePlan& P, Arena& A, bool record, int frames) {
  Builder b{*this, A, &P.prog, record, dt, esz};
  const int B = P.B;
  const int nm = cfg.voc_num_mels;
  P.in_bytes = (size_t)B * frames * nm * 4;
  P.in = A.alloc(P.in_bytes);
  P.n_out = vocoder_samples(frames);
  P.out_bytes = (size_t)B * P.n_out * 2;
  P.out = A.alloc(P.out_bytes);
  const int d = dt;

  auto conv1d = [&](const TView& x, int L, const WMat& w, int dil, const TView& out, const GOpt& o, bool i16 = false) {
    GemmParams p;
    p.A = x.p; p.lda = x.ld; p.W = w.W; p.Kp = w.Kp; p.bias = w.b;
    p.M = B * L; p.N = w.N; p.K = w.K; p.Cin = w.Cin;
    p.mode = GATHER_1D; p.rows_pb = L; p.Lin = L; p.taps = w.taps; p.tap_step = dil; p.in_mul = 1;
    p.in_off = -dil * (w.taps - 1) / 2;
    p.Lout = L; p.out_mul = 1; p.out_off = 0;
    p.out = out.p; p.ldo = out.ld;
    if (o.residual) { p.R = o.residual->p; p.ldr = o.residual->ld; }
    p.a_act = o.a_act; p.a_slope = o.a_slope; p.e_act = o.e_act; p.e_slope = o.e_slope;
    if (i16) { p.epi = EPI_I16; p.out_scale = 32768.0f; }
    b.gemm(p);
  };

  // mel [B,1,T,nm] fp32 is already channels-last [B,T,nm]
  TView x = b.alloc((int64_t)B * frames, nm);
  {
    const float* src = (const float*)P.in; void* dst = x.p; const int r = B * frames;
    b.push([=](hipStream_t s) { return launch_cast_rows(d, src, dst, nm, r, nm, s); });
  }
  int L = frames, C = voc_pre.N;
  TView h = b.alloc((int64_t)B * L, C);
  { GOpt o; o.e_act = ACT_LRELU; o.e_slope = 0.1f; conv1d(x, L, voc_pre, 1, h, o); }   // lrelu of stage 0 folded in
  const int nk = cfg.voc_n_resblocks;
  for (int i = 0; i < cfg.voc_n_ups; ++i) {
    const ConvTW& ct = voc_ups[i];
    const int u = ct.u, k = ct.k, pd = ct.pad;
    const int Lo = (L - 1) * u - 2 * pd + k;
    TView y = b.alloc((int64_t)B * Lo, ct.cout);
    for (int r = 0; r < u; ++r) {
      // outputs t = u*q + r - pd; input index q - tap
      const int qmin = (pd > r) ? (pd - r + u - 1) / u : 0;
      const int qmax = (Lo - 1 + pd - r) / u;
      const int Q = qmax - qmin + 1;
      if (Q <= 0) continue;
      const WMat& w = ct.phase[r];
      GemmParams p;
      p.A = h.p; p.lda = h.ld; p.W = w.W; p.Kp = w.Kp; p.bias = w.b;
      p.M = B * Q; p.N = w.N; p.K = w.K; p.Cin = w.Cin;
      p.mode = GATHER_1D; p.rows_pb = Q; p.Lin = L; p.taps = w.taps; p.tap_step = -1; p.in_mul = 1; p.in_off = qmin;
      p.Lout = Lo; p.out_mul = u; p.out_off = u * qmin + r - pd;
      p.out = y.p; p.ldo = y.ld;
      b.gemm(p);
    }
    L = Lo; C = ct.cout;
    TView rs[4];
    for (int j = 0; j < nk; ++j) {
      const VocResW& rw = voc_res[(size_t)i * nk + j];
      TView cur = y;
      for (size_t m = 0; m < rw.dil.size(); ++m) {
        TView t1 = b.alloc((int64_t)B * L, C);
        { GOpt o; o.a_act = ACT_LRELU; o.a_slope = 0.1f; o.e_act = ACT_LRELU; o.e_slope = 0.1f; conv1d(cur, L, rw.c1[m], rw.dil[m], t1, o); }
        TView t2 = b.alloc((int64_t)B * L, C);
        { GOpt o; o.residual = &cur; conv1d(t1, L, rw.c2[m], 1, t2, o); }
        cur = t2;
      }
      rs[j] = cur;
    }
    if (nk != 3) TANGO_FAIL("vocoder: exactly 3 resblock kernels supported");
    TView nx = b.alloc((int64_t)B * L, C);
    {
      const void* a0 = rs[0].p; const void* a1 = rs[1].p; const void* a2 = rs[2].p; void* yo = nx.p;
      const int64_t n = (int64_t)B * L * C;
      const float slope = (i == cfg.voc_n_ups - 1) ? 0.01f : 0.1f;   // models.py:161 default slope on the last one
      b.push([=](hipStream_t s) { return launch_avg3_act(d, a0, a1, a2, yo, n, 1.0f / 3.0f, ACT_LRELU, slope, s); });
    }
    h = nx;
  }
  {
    TView o; o.p = P.out; o.ld = 1; o.C = 1;
    GOpt go; go.e_act = ACT_TANH;
    conv1d(h, L, voc_post, 1, o, go, true);
  }
  return 0;
}

int Engine::get_voc_plan(int B, int frames, VaePlan** out) {
  auto key = std::make_pair(B, frames);
  auto it = voc_plans.find(key);
  if (it != voc_plans.end()) { touch(it->second->meta); *out = it->second.get(); return 0; }
  if (!finalized) TANGO_FAIL("engine: weights not finalized");
  std::unique_ptr<VaePlan> P(new VaePlan());
  P->B = B;
  Arena m;
  TANGO_TRY(build_voc(*P, m, false, frames));
  TANGO_TRY(alloc_slab(&P->slab, m.peak + 256, P->meta, false));
  Arena a; a.base = P->slab;
  P->prog.ops.clear();
  if (int rc = build_voc(*P, a, true, frames)) { release_slab(&P->slab, P->meta); return rc; }
  *out = P.get();
  voc_plans[key] = std::move(P);
  return 0;
}

int Engine::vocode(const float* mel, int16_t* wav, int B, int frames, int* n_samples, hipStream_t s) {
  if (cfg.voc_n_ups <= 0) TANGO_FAIL("engine: vocoder not configured");
  VaePlan* P;
  TANGO_TRY(get_voc_plan(B, frames, &P));
  TANGO_HIP(hipMemcpyAsync(P->in, mel, P->in_bytes, hipMemcpyDeviceToDevice, s));
  TANGO_TRY(P->prog.run(s));
  TANGO_HIP(hipMemcpyAsync(wav, P->out, P->out_bytes, hipMemcpyDeviceToDevice, s));
  if (n_samples) *n_samples = P->n_out;
  return 0;
}

}  // namespace tango

// ================================================================================================
// C ABI
// ================================================================================================
using tango::Engine;

extern "C" {

const char* tango_last_error(void) { return tango::last_error(); }
const char* tango_version(void) { return "tango-mi355x 0.1 (gfx950)"; }
void tango_tuning_reload(void) { tango::tuning_reload(); }

int tango_engine_set_plan_budget(tango_engine_t* h, uint64_t bytes) {
  if (!h) { tango::set_error("tango_engine_set_plan_budget: null engine"); return -1; }
  h->e->set_plan_budget((size_t)bytes);
  return 0;
}
int tango_engine_drop_plans(tango_engine_t* h) {
  if (!h) { tango::set_error("tango_engine_drop_plans: null engine"); return -1; }
  h->e->drop_plans();
  return 0;
}
int tango_engine_plan_stats(tango_engine_t* h, uint64_t* bytes_in_use, int* plans) {
  if (!h) { tango::set_error("tango_engine_plan_stats: null engine"); return -1; }
  if (bytes_in_use) *bytes_in_use = (uint64_t)h->e->plan_bytes_in_use();
  if (plans) *plans = h->e->plan_count();
  return 0;
}

int tango_engine_create(const tango_config_t* cfg, tango_engine_t** out) {
  if (!cfg || !out) { tango::set_error("tango_engine_create: null argument"); return -1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    tango::set_error("tango_engine_create: no HIP device visible (the engine has no CPU fallback)");
    return -2;
  }
  Engine* e = new Engine(*cfg);
  if (e->init() != 0) { delete e; return -1; }
  *out = new tango_engine{e};
  return 0;
}

void tango_engine_destroy(tango_engine_t* h) {
  if (!h) return;
  delete h->e;
  delete h;
}

int tango_engine_num_weights(tango_engine_t* h) { return (int)h->e->slots.size(); }
const char* tango_engine_weight_name(tango_engine_t* h, int i) {
  if (i < 0 || i >= (int)h->e->slots.size()) return nullptr;
  return h->e->slots[i].name.c_str();
}
int tango_engine_set_weight(tango_engine_t* h, const char* name, const float* dev_ptr, const int64_t* shape, int ndim) {
  return h->e->set_weight(name, dev_ptr, shape, ndim);
}
int tango_engine_finalize_weights(tango_engine_t* h) { return h->e->finalize_weights(); }

int tango_engine_denoise(tango_engine_t* h, const tango_denoise_args_t* a, void* stream) {
  if (!a) { tango::set_error("denoise: null args"); return -1; }
  return h->e->denoise(*a, (hipStream_t)stream);
}
int tango_engine_unet_forward(tango_engine_t* h, const float* sample, int64_t timestep, const float* prompt_embeds,
                              const uint8_t* prompt_mask, float* out, int batch2, int text_len, void* stream) {
  Engine::Cond c[3];
  c[0].emb = prompt_embeds; c[0].mask = prompt_mask; c[0].len = text_len;
  return h->e->unet_forward(sample, timestep, c, out, batch2, (hipStream_t)stream);
}
int tango_engine_unet_forward_music(tango_engine_t* h, const float* sample, int64_t timestep, const float* prompt_embeds,
                                    const uint8_t* prompt_mask, const float* beat_embeds, const uint8_t* beat_mask,
                                    const float* chord_embeds, const uint8_t* chord_mask, float* out, int batch2, int text_len,
                                    int beat_len, int chord_len, void* stream) {
  Engine::Cond c[3];
  c[0].emb = prompt_embeds; c[0].mask = prompt_mask; c[0].len = text_len;
  c[1].emb = beat_embeds; c[1].mask = beat_mask; c[1].len = beat_len;
  c[2].emb = chord_embeds; c[2].mask = chord_mask; c[2].len = chord_len;
  return h->e->unet_forward(sample, timestep, c, out, batch2, (hipStream_t)stream);
}
int tango_engine_vae_encode(tango_engine_t* h, const float* mel, float* moments, int batch, void* stream) {
  return h->e->vae_encode(mel, moments, batch, (hipStream_t)stream);
}
int tango_engine_vae_decode(tango_engine_t* h, const float* latents, float* mel, int batch, void* stream) {
  return h->e->vae_decode(latents, mel, batch, (hipStream_t)stream);
}
int tango_engine_vocode(tango_engine_t* h, const float* mel, int16_t* wav, int batch, int mel_frames, int* n_samples, void* stream) {
  return h->e->vocode(mel, wav, batch, mel_frames, n_samples, (hipStream_t)stream);
}
int tango_engine_vocoder_samples(tango_engine_t* h, int mel_frames) { return h->e->vocoder_samples(mel_frames); }
int tango_engine_encode_text(tango_engine_t* h, const int64_t* input_ids, const uint8_t* attention_mask, float* out, int batch,
                             int text_len, void* stream) {
  return h->e->encode_text(input_ids, attention_mask, out, batch, text_len, (hipStream_t)stream);
}

int tango_engine_profile_unet(tango_engine_t* h, int batch2, int text_len, char* report, int report_cap, void* stream) {
  std::string r;
  int rc = h->e->profile_unet(batch2, text_len, r, (hipStream_t)stream);
  if (report && report_cap > 0) {
    const size_t n = std::min((size_t)report_cap - 1, r.size());
    memcpy(report, r.data(), n);
    report[n] = 0;
  }
  return rc;
}
int tango_engine_profile_vae(tango_engine_t* h, int batch, char* report, int report_cap, void* stream) {
  if (!h || !report || report_cap <= 0) { tango::set_error("tango_engine_profile_vae: null argument"); return -1; }
  std::string r;
  int rc = h->e->profile_vae(batch, r, (hipStream_t)stream);
  if (rc) return rc;
  snprintf(report, (size_t)report_cap, "%s", r.c_str());
  return 0;
}
int tango_engine_profile_vocoder(tango_engine_t* h, int batch, int frames, char* report, int report_cap, void* stream) {
  if (!h || !report || report_cap <= 0) { tango::set_error("tango_engine_profile_vocoder: null argument"); return -1; }
  std::string r;
  int rc = h->e->profile_vocoder(batch, frames, r, (hipStream_t)stream);
  if (rc) return rc;
  snprintf(report, (size_t)report_cap, "%s", r.c_str());
  return 0;
}

int tango_engine_last_denoise_ms(tango_engine_t* h, float* total_ms, float* per_step_ms) {
  return h->e->last_denoise_ms(total_ms, per_step_ms);
}
int tango_engine_last_step_gflop(tango_engine_t* h, double* gflop) {
  if (!h || !gflop) { tango::set_error("tango_engine_last_step_gflop: null argument"); return -1; }
  if (h->e->last_step_gflop <= 0.0) { tango::set_error("last_step_gflop: no denoise call recorded"); return -1; }
  *gflop = h->e->last_step_gflop;
  return 0;
}

}  // extern "C"
