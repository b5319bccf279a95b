// Weight-stationary, activation-streaming linear for the small-K transformer GEMMs (K*sizeof(T) = 640 or
// 1280 bytes: C = 320 / 640 in 16-bit).  At these K the tiled gather-GEMM re-loads a whole weight panel per
// 128-row tile and is L1/latency-bound (180-300 TFLOP/s); here
//   * one 512-thread workgroup per CU keeps a [BN x K] weight panel resident in LDS (100 KB, XOR-swizzled),
//   * each of the 8 waves streams its own 32-row groups of the activation matrix STRAIGHT from HBM into MFMA
//     B-operand fragments (lane (m, g) owns 16 contiguous bytes of row m), through a 5/10-deep register ring
//     (loads are issued 5-10 k-steps ahead of their use), with NO barrier in the main loop,
//   * LayerNorm is folded algebraically: with W' = W*gamma, b' = b + W.beta, wsum[n] = sum_k W'[n][k],
//       y[m][n] = rstd[m] * (sum_k W'[n][k] x[m][k] - mean[m] * wsum[n]) + b'[n],
//     so the raw activations feed the MFMA and the row statistics (accumulated from the same fragments)
//     enter only in the epilogue: the LayerNorm kernel and its write+read round trip disappear
//     (BasicTransformerBlock norm1/2/3, mustango/diffusers/src/diffusers/models/attention.py:276-335).
#include <cstdlib>

#include <cstdint>

#include "common.h"
#include "tuning.h"

namespace tango {

template <typename T> struct SMma;
template <> struct SMma<float> {
  __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
  }
};
template <> struct SMma<f16> {
  __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  }
};
template <> struct SMma<bf16> {
  __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};

// LayerNorm statistics of a row.  fp32 rows: sums of (x - shift) and (x - shift)^2 with shift = the row's first element, so
// that the one-pass variance E[d^2] - E[d]^2 cancels only to the extent the row's mean differs from one of its own
// elements (a few std), not to the extent |mean| >> std (fp32 rows with mean / std = 100 lost 3 digits without it).
// 16-bit rows: v_dot2c_f32_{f16,bf16} on the packed pairs (products exact, fp32 accumulate), 8 VALU instructions per
// 8-element fragment instead of 24; no shift - the storage rounding of x (2^-11 / 2^-8 of |mean|) already exceeds what
// the cancellation loses in fp32 (tests/test_determinism_gpu.py::test_linear_ln_large_mean_and_outliers).
template <typename T> __device__ __forceinline__ void frag_stats(const u32x4& v, float shift, float& s, float& q) {
  if constexpr (sizeof(T) == 4) {
    float e[4];
    __builtin_memcpy(e, &v, 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float f = e[i] - shift; s += f; q += f * f; }
  } else if constexpr (__is_same(T, f16)) {
    // (the pairs are taken with shufflevector: indexing the dwords of the u32x4 reference in an unrolled loop made hipcc
    //  7.2 emit the FIRST dword four times - caught by the LN parity tests)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const f16x8 h = __builtin_bit_cast(f16x8, v);
    const h2 one = h2{(_Float16)1.f, (_Float16)1.f};
    const h2 p0 = __builtin_shufflevector(h, h, 0, 1), p1 = __builtin_shufflevector(h, h, 2, 3);
    const h2 p2 = __builtin_shufflevector(h, h, 4, 5), p3 = __builtin_shufflevector(h, h, 6, 7);
    s = __builtin_amdgcn_fdot2(p0, one, s, false); q = __builtin_amdgcn_fdot2(p0, p0, q, false);
    s = __builtin_amdgcn_fdot2(p1, one, s, false); q = __builtin_amdgcn_fdot2(p1, p1, q, false);
    s = __builtin_amdgcn_fdot2(p2, one, s, false); q = __builtin_amdgcn_fdot2(p2, p2, q, false);
    s = __builtin_amdgcn_fdot2(p3, one, s, false); q = __builtin_amdgcn_fdot2(p3, p3, q, false);
  } else {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const bf16x8 h = __builtin_bit_cast(bf16x8, v);
    const b2 one = b2{(__bf16)1.f, (__bf16)1.f};
    const b2 p0 = __builtin_shufflevector(h, h, 0, 1), p1 = __builtin_shufflevector(h, h, 2, 3);
    const b2 p2 = __builtin_shufflevector(h, h, 4, 5), p3 = __builtin_shufflevector(h, h, 6, 7);
    s = __builtin_amdgcn_fdot2_f32_bf16(p0, one, s, false); q = __builtin_amdgcn_fdot2_f32_bf16(p0, p0, q, false);
    s = __builtin_amdgcn_fdot2_f32_bf16(p1, one, s, false); q = __builtin_amdgcn_fdot2_f32_bf16(p1, p1, q, false);
    s = __builtin_amdgcn_fdot2_f32_bf16(p2, one, s, false); q = __builtin_amdgcn_fdot2_f32_bf16(p2, p2, q, false);
    s = __builtin_amdgcn_fdot2_f32_bf16(p3, one, s, false); q = __builtin_amdgcn_fdot2_f32_bf16(p3, p3, q, false);
  }
}
template <typename T> __device__ __forceinline__ float frag_first(const u32x4& v) {
  T e0;
  __builtin_memcpy(&e0, &v, sizeof(T));
  return to_f(e0);
}

// KS = K*sizeof(T)/64 k-steps per row, TN = 16-column tiles per panel (BN = 16*TN); ring of R = 10 k-steps
// LN (folded LayerNorm) is a template parameter: a wave-uniform runtime test in the micro-step loop is not free
// (the halo conv gained 7-9 % when its ablation tests were compiled out).
// FIX = false rebuilds the epilogue's "acc - mean * wsum" as plain C++ (hipcc SLP-vectorises it into v_pk_fma_f32 with
// op_sel operands behind a v_xor): that form reproduces the round-1 miscompare at K = 640 16-bit (39 of 300 repeats, same
// box, same call as 0 of 1000 for the FIX form - profiles/r2_race_hunt.txt).  Only instantiated for that one configuration
// (TANGO_STREAM_NOFIX=1, tools/diag_stream_race.py) so the failing form stays reproducible.
// SPEC (round 5): 1 = the epilogue specialised at compile time for the one shape that dominates this kernel's time at config 3, the
// level-0 GEGLU projection with the folded LayerNorm (M = 262144, N = 2560, K = 320: 3.3 ms per step, and PMC shows it VALU-bound
// by its epilogue -- 6.1 VALU per MFMA, profiles/r5_final_pmc_gemm_pers_and_stream_geglu_summary.txt): EPI_GEGLU, no residual, no
// transposed-V columns, staged stores, M a multiple of 32 (no row clamps / masks).  Same arithmetic in the same order as SPEC = 0 --
// only the wave-uniform run-time tests on p.epi / p.R / p.stage_epi / m < M inside the 20 unrolled epilogue iterations and the
// run-time divisor of the store loop are gone.  TANGO_STREAM_SPEC=0 is the A/B switch.
template <typename T, int KS, int TN, bool LN, bool FIX = true, int SPEC = 0>
__global__ __launch_bounds__(512, 2) void lin_stream_kernel(const GemmParams p) {
  constexpr int R = (TN > 5) ? 5 : 10;         // ring depth in k-steps (register budget: acc 8*TN + ring 8*R)
  constexpr int TM = 2;
  constexpr int ROWB = KS * 64;                // bytes per weight row
  constexpr int BN = TN * 16;
  static_assert(KS % R == 0, "ring");
  // Rows staged per output-store pass (round 2: 32-row staging of the 80-column panels, which removes the half-masked
  // last store iteration, was tried against the race and changed nothing: 154 of 1000)
  constexpr int SROWS = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];   // [BN][ROWB], swizzled per 128-byte segment

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  // XCD-aware placement: block b runs on XCD b % 8 (observed round-robin dispatch; speed only).  All NP panels of
  // one row range sit on the SAME XCD and advance in lockstep, so the activation rows are pulled from HBM/MALL into
  // that XCD's L2 once and the other NP-1 panels hit L2.
  const int NP = p.N / BN;                     // panels
  const int rpx = gridDim.x / (8 * NP);        // row ranges per XCD
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int panel = j % NP, mb = xcd * rpx + j / NP, MB = 8 * rpx;
  const int n0 = panel * BN;

  // ---- weight panel -> LDS (once per workgroup) ----
  {
    const unsigned char* Wp = (const unsigned char*)p.W + (int64_t)n0 * p.Kp * (int64_t)sizeof(T);
    constexpr int PPRW = ROWB / 16;
    for (int id = tid; id < BN * PPRW; id += 512) {
      const int row = id / PPRW, pc = id % PPRW;
      const u32x4 v = *(const u32x4*)(Wp + (int64_t)row * p.Kp * (int64_t)sizeof(T) + pc * 16);
      const int seg = pc >> 3, pp = pc & 7;
      *(u32x4*)(wlds + row * ROWB + seg * 128 + ((pp ^ (row & 7)) * 16)) = v;
    }
  }
  {   // per-column epilogue constants of this panel: [bias | wsum] (fp32), read back through lgkmcnt, not vmcnt
    float* cstw = (float*)(wlds + BN * ROWB + 8 * SROWS * (BN * (int)sizeof(T) + 16));
    for (int i = tid; i < BN; i += 512) {
      cstw[i] = p.bias ? p.bias[n0 + i] : 0.f;
      cstw[BN + i] = LN ? p.wsum[n0 + i] : 0.f;
    }
  }
  __syncthreads();

  // ---- this workgroup's rows: groups of 32 rows, dealt round-robin to the 8 waves ----
  const int ngroups = (p.M + 31) / 32;
  const int gper = (ngroups + MB - 1) / MB;
  const int g0 = mb * gper, g1 = min(ngroups, g0 + gper);
  const unsigned char* Ab = (const unsigned char*)p.A;
  const int64_t ldab = p.lda * (int64_t)sizeof(T);

  // Rows beyond M are clamped to a valid row (their results are dropped in the epilogue), and a wave without a
  // next group re-reads its current one, so the main loop has NO branches: every load is unconditional.
  auto row_ptr_c = [&](int grp_, int tm) -> const unsigned char* {
    int m = grp_ * 32 + tm * 16 + l15;
    if (SPEC == 0) m = m < p.M ? m : p.M - 1;
    return Ab + (int64_t)m * ldab + g * 16;
  };
  u32x4 xf[R][TM];
  int grp = g0 + wave;
  const unsigned char* cur[TM];
  const unsigned char* nxt[TM];
  // (a wave -- or a whole workgroup, when ngroups is not a multiple of MB: g0 >= ngroups -- without a row group still issues this
  //  prefetch; SPEC = 1 has no per-row clamp, so the GROUP is clamped into the tensor: ADVICE r5)
  const int g_safe = g0 < ngroups ? g0 : ngroups - 1;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) cur[tm] = row_ptr_c(grp < g1 ? grp : g_safe, tm);
#pragma unroll
  for (int s = 0; s < R; ++s)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) xf[s][tm] = *(const u32x4*)(cur[tm] + s * 64);

  // weight-fragment offsets: row n_local = tn*16 + l15 -> (row & 7) == (lane & 7)
  const unsigned char* wbase = wlds + l15 * ROWB;
  const int sw = lane & 7;
  const int g4 = g * 4;
  auto koff_of = [&](int ks) -> int { return (ks >> 1) * 128 + ((((ks & 1) * 4 + g) ^ sw) * 16); };

  // explicit software pipeline on the LDS side, in micro-steps of H = 5 tiles (10 MFMAs = 160 cycles >= LDS
  // latency): the 5 weight fragments of micro-step u+1 are read while micro-step u multiplies
  constexpr int H = 5, NH = TN / H, NU = KS * NH;
  static_assert(TN % H == 0 && (NU % 2) == 0, "micro-steps");
  u32x4 wf[2][H];
#pragma unroll
  for (int a = 0; a < H; ++a) wf[0][a] = *(const u32x4*)(wbase + a * 16 * ROWB + koff_of(0));

  for (; grp < g1; grp += 8) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) nxt[tm] = row_ptr_c(grp + 8 < g1 ? grp + 8 : grp, tm);
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ssum[TM] = {0.f, 0.f}, ssq[TM] = {0.f, 0.f}, shift[TM] = {0.f, 0.f};
    if (LN && sizeof(T) == 4) {
      // ring slot 0 holds k-step 0 of this group's rows; lane l15 (g == 0) holds the row's first element
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) shift[tm] = __shfl(frag_first<T>(xf[0][tm]), l15);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int ks = u / NH, hh = u % NH;
      const int slot = ks % R;
      const int cb_ = u & 1, nb_ = cb_ ^ 1;
      const int un = (u + 1) % NU;                        // next micro-step (wraps to the next group: same weights)
      const int ksn = un / NH, hn = un % NH;
#pragma unroll
      for (int a = 0; a < H; ++a) wf[nb_][a] = *(const u32x4*)(wbase + (hn * H + a) * 16 * ROWB + koff_of(ksn));
      __builtin_amdgcn_sched_barrier(0);
      if (LN && hh == 0) { frag_stats<T>(xf[slot][0], shift[0], ssum[0], ssq[0]); frag_stats<T>(xf[slot][1], shift[1], ssum[1], ssq[1]); }
#pragma unroll
      for (int a = 0; a < H; ++a) {
        SMma<T>::run(acc[hh * H + a][0], wf[cb_][a], xf[slot][0]);
        SMma<T>::run(acc[hh * H + a][1], wf[cb_][a], xf[slot][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // keep the weight fragments of this micro-step allocated until its VALU work is done (see DESIGN.md section 5:
      // tools/mfma_war_repro.hip shows that a VALU write right behind an MFMA's source operands is safe on gfx950, so
      // this is belt and braces, kept because the templated kernel was validated in this form)
#pragma unroll
      for (int a = 0; a < H; ++a) asm volatile("" ::"v"(wf[cb_][a]));
      if (hh == NH - 1) {
        // refill the ring slot with the k-step that is R ahead in this wave's stream
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const unsigned char* src = (ks + R < KS) ? cur[tm] + (ks + R) * 64 : nxt[tm] + (ks + R - KS) * 64;
          xf[slot][tm] = *(const u32x4*)src;
        }
      }
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) cur[tm] = nxt[tm];

    // ---- epilogue for rows grp*32 .. +31, columns n0 .. n0+BN ----
    float mean[TM] = {0.f, 0.f}, rstd[TM] = {1.f, 1.f};
    if (LN) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        float s = ssum[tm], q = ssq[tm];
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
        const float dmu = s / (float)p.K;
        float var = q / (float)p.K - dmu * dmu;
        var = var < 0.f ? 0.f : var;
        mean[tm] = shift[tm] + dmu; rstd[tm] = rsqrtf(var + p.ln_eps);
      }
    }
    // The epilogue must not contain dependent global-load chains: vmcnt is in-order, so every load -> use here also
    // waits for the next group's prefetched activation rows, and each extra round trip is exposed (2 waves/SIMD).
    // Per-column constants therefore come from LDS (lgkmcnt), the residual quads are fetched in ONE batch up front,
    // and rows beyond M are clamped (loads) / masked (stores) instead of branched around.
    // (n0e / g4e are made opaque per group: otherwise every per-column address below is hoisted out of the group loop
    //  as a loop invariant -- ~30 64-bit values -- and spilled, and scratch reloads are vmcnt traffic too)
    int n0e = n0, g4e = g4;
    asm volatile("" : "+s"(n0e));
    asm volatile("" : "+v"(g4e));
    // (SPEC == 1: every switch below is a compile-time constant)
    const int epi = SPEC == 1 ? (int)EPI_GEGLU : p.epi;
    const bool has_res = SPEC == 1 ? false : p.R != nullptr;
    int64_t orow[TM], vtrow[TM];
    bool rok[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = grp * 32 + tm * 16 + l15;
      rok[tm] = SPEC == 1 ? true : m < p.M;
      orow[tm] = rok[tm] ? m : p.M - 1; vtrow[tm] = 0;
      if (epi == EPI_VT) {
        const int mm = (int)orow[tm];
        const int bb = mm / p.vt_S;
        const int sq = mm - bb * p.vt_S;
        vtrow[tm] = (int64_t)bb * (p.N - p.vt_n0) * p.vt_ld + (p.vt_perm ? vt_perm_pos(sq) : sq);
      }
    }
    const bool panel_vt = (epi == EPI_VT) && n0e >= p.vt_n0;
    const bool stage = SPEC == 1 ? true : (p.stage_epi && !panel_vt && !(epi == EPI_VT && n0e + BN > p.vt_n0));
    constexpr int SPITCH = BN * (int)sizeof(T) + 16;
    unsigned char* const stg = wlds + BN * ROWB + wave * (SROWS * SPITCH);
    const float* const cst = (const float*)(wlds + BN * ROWB + 8 * SROWS * SPITCH);   // [bias BN | wsum BN]

    T rv[TM][TN][4];
    if (has_res) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int a = 0; a < TN; ++a)
          __builtin_memcpy(rv[tm][a], (const T*)p.R + orow[tm] * p.ldr + n0e + a * 16 + g4e, 4 * sizeof(T));
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        if (epi == EPI_GEGLU && (a & 1)) continue;
        const int nt = n0e + a * 16;
        const int n = nt + g4e;
        const f32x4 cb = *(const f32x4*)(cst + a * 16 + g4e);
        const f32x4 cw = *(const f32x4*)(cst + BN + a * 16 + g4e);
        int oc = n, ocl = a * 16 + g4e;                  // output column (global / within the staged slice)
        if (epi == EPI_GEGLU) { oc = (nt >> 1) + g4e; ocl = (a >> 1) * 16 + g4e; }
        const bool to_vt = (epi == EPI_VT) && n >= p.vt_n0;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (FIX) {
            // single-instruction form of acc - mean * wsum, opaque to the SLP vectoriser (see the race notes in DESIGN.md
            // section 5: every miscompare sat in the v_xor + v_pk_fma_f32 op_sel sequence hipcc builds for elements 2, 3)
            float t;
            asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(t) : "v"(mean[tm]), "v"(cw[r]), "v"(acc[a][tm][r]));
            v[r] = rstd[tm] * t + cb[r];
          } else {
            v[r] = rstd[tm] * (acc[a][tm][r] - mean[tm] * cw[r]) + cb[r];
          }
        }
        if (epi == EPI_GEGLU) {
          const int a1 = a + 1 < TN ? a + 1 : a;
          const f32x4 gb = *(const f32x4*)(cst + a1 * 16 + g4e);
          const f32x4 gw = *(const f32x4*)(cst + BN + a1 * 16 + g4e);
          float gt[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (FIX) {
              float t;
              asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(t) : "v"(mean[tm]), "v"(gw[r]), "v"(acc[a1][tm][r]));
              gt[r] = rstd[tm] * t + gb[r];
            } else {
              gt[r] = rstd[tm] * (acc[a1][tm][r] - mean[tm] * gw[r]) + gb[r];
            }
          }
          glu_gate4<T>(v, gt, 0);
        }
        if (has_res) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += to_f(rv[tm][a][r]);
        }
        T tv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) tv[r] = from_f<T>(v[r]);
        if (stage) {
          __builtin_memcpy(stg + ((SROWS == 32 ? tm * 16 : 0) + l15) * SPITCH + ocl * (int)sizeof(T), tv, 4 * sizeof(T));
        } else if (to_vt) {
          if (rok[tm]) {
            T* vp = (T*)p.vt + vtrow[tm] + (int64_t)(n - p.vt_n0) * p.vt_ld;
#pragma unroll
            for (int r = 0; r < 4; ++r) vp[(int64_t)r * p.vt_ld] = tv[r];
          }
        } else {
          if (rok[tm]) __builtin_memcpy((T*)p.out + orow[tm] * p.ldo + oc, tv, 4 * sizeof(T));
        }
      }
      if (stage && (SROWS == 16 || tm == TM - 1)) {
        __builtin_amdgcn_wave_barrier();
        constexpr int EPV = 16 / (int)sizeof(T);
        const int ppr = (epi == EPI_GEGLU ? BN / 2 : BN) / EPV;           // 16-byte pieces per output row
        const int ocol0 = epi == EPI_GEGLU ? (n0e >> 1) : n0e;
        for (int idx = lane; idx < SROWS * ppr; idx += 64) {
          const int row_l = idx / ppr, pcs = idx - row_l * ppr;
          const int m = grp * 32 + (SROWS == 32 ? 0 : tm * 16) + row_l;
          if (SPEC == 1 || m < p.M) {
            const u32x4 t = *(const u32x4*)(stg + row_l * SPITCH + pcs * 16);
            *(u32x4*)((T*)p.out + (int64_t)m * p.ldo + ocol0 + pcs * EPV) = t;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

template <typename T, int KS, int TN, bool LN, bool FIX = true, int SPEC = 0>
static int stream_launch(const GemmParams& p, hipStream_t s) {
  constexpr int BN = TN * 16;
  constexpr int SROWS = 16;
  constexpr int LDS = BN * KS * 64 + 8 * SROWS * (BN * (int)sizeof(T) + 16) + 2 * BN * 4;   // weight panel + per-wave output staging + constants
  auto kfn = lin_stream_kernel<T, KS, TN, LN, FIX, SPEC>;
  TANGO_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(kfn), LDS));
  const int NP = p.N / BN;
  int rpx = 32 / NP;               // one workgroup per CU: 32 per XCD = NP panels x rpx row ranges
  if (rpx < 1) rpx = 1;
  const int ngroups = (p.M + 31) / 32;
  while (rpx > 1 && 8 * rpx * 8 > ngroups) --rpx;   // keep >= 8 row groups (one per wave) per workgroup
  GemmParams q = p;
  constexpr int EPVH = 16 / (int)sizeof(T);
  q.stage_epi = (p.ldo % EPVH == 0 && ((uintptr_t)p.out & 15) == 0 && (p.epi != EPI_GEGLU || (BN / 2) % EPVH == 0)) ? 1 : 0;
  hipLaunchKernelGGL(kfn, dim3((unsigned)(8 * NP * rpx)), dim3(512), LDS, s, q);
  TANGO_HIP(hipGetLastError());
  return 0;
}

// can this GEMM run on the streaming kernel?
bool linear_stream_ok(int dtype, const GemmParams& p) {
  if (tuning().no_stream && !p.ln_fold) return false;
  if (tuning().no_stream_ln_geglu && p.ln_fold && p.epi == EPI_GEGLU) return false;
  if (p.row_stats) return false;
  const int esz = dtype == DT_F32 ? 4 : 2;
  const int rowb = p.K * esz;
  if (p.mode != GATHER_1D || p.taps != 1 || p.rows_pb != p.M || p.in_mul != 1 || p.in_off != 0 || p.out_mul != 1 || p.out_off != 0)
    return false;
  if (p.batch != 1 || p.bias2 || p.bias_rows || p.out_f32 || p.a_act != ACT_NONE || p.e_act != ACT_NONE || p.alpha != 1.f ||
      p.out_scale != 1.f)
    return false;
  if (p.epi != EPI_NONE && p.epi != EPI_GEGLU && p.epi != EPI_VT) return false;
  if (p.glu_tanh) return false;                     // the streaming epilogue implements the exact-erf gate only
  if (p.M < 4096) return false;                     // too few row groups to feed 256 CUs
  // plain linears below 16384 rows (B = 1's level 0) run faster as one launch of 64 x 64 tiles (gemm.hip small_tile_linear:
  // x20 per step 0.514 -> 0.301 ms); at M = 16384 (B = 8's level 1) this kernel beats the 128 x 160 tiles (0.845 vs 1.014 ms) but
  // not the 256 x 160 LDS-DMA kernel at one tile per CU (0.83 vs 0.69 ms, profiles/r3_c13_b8_dispatch_thresholds.txt)
  if (!p.ln_fold && p.epi == EPI_NONE && p.M < 16384 && !tuning().no_small_tile) return false;
  if (!p.ln_fold && p.epi == EPI_NONE && p.M < 32768 && rowb == 1280 && gemm_dma_ok(dtype, p)) return false;
  int tn;
  if (rowb == 640) tn = 10;
  else if (rowb == 1280) tn = 5;
  else return false;
  if (p.N % (16 * tn) != 0) return false;
  // weight panel + per-wave output staging + constants must fit the 160-KB LDS (fp32 rows of 640 bytes do not)
  const int srows = 16;
  if ((long)16 * tn * rowb + 8L * srows * (16 * tn * esz + 16) + 2L * 16 * tn * 4 > 160 * 1024) return false;
  if (p.epi == EPI_GEGLU && (tn & 1)) return false;
  if (p.epi == EPI_VT && (p.vt_n0 % 16) != 0) return false;
  if ((p.ldo % 4) || (p.R && (p.ldr % 4))) return false;
  return true;
}

template <typename T>
static int stream_t(const GemmParams& p, hipStream_t s) {
  const int rowb = p.K * (int)sizeof(T);
  if (rowb == 640) {
    if constexpr (sizeof(T) == 2) {
      // the level-0 GEGLU projection: compile-time epilogue (SPEC = 1) when every assumption it makes holds
      constexpr int EPVH = 8;
      if (p.ln_fold && p.epi == EPI_GEGLU && !p.R && p.M % 32 == 0 && p.ldo % EPVH == 0 && ((uintptr_t)p.out & 15) == 0 && tuning().stream_spec)
        return stream_launch<T, 10, 10, true, true, 1>(p, s);
    }
    return p.ln_fold ? stream_launch<T, 10, 10, true>(p, s) : stream_launch<T, 10, 10, false>(p, s);
  }
  if (rowb == 1280) {
    // (the FIX = false instantiation -- the epilogue form that miscompared 39 / 300 on the bf16 K = 640 LN build,
    //  profiles/r2_race_hunt.txt -- is no longer compiled into the library: tools/experiments/README.md says how to rebuild it;
    //  tools/isa_scan.py checks every shipped kernel for the instruction pattern it had)
    return p.ln_fold ? stream_launch<T, 20, 5, true>(p, s) : stream_launch<T, 20, 5, false>(p, s);
  }
  TANGO_FAIL("linear_stream: unsupported K");
}

int launch_linear_stream(int dtype, const GemmParams& p, hipStream_t s) {
  switch (dtype) {
    case DT_F32: return stream_t<float>(p, s);
    case DT_F16: return stream_t<f16>(p, s);
    case DT_BF16: return stream_t<bf16>(p, s);
  }
  TANGO_FAIL("linear_stream: bad dtype");
}

// ------------------------------------------------------------------------------------------
// LayerNorm folding at weight-finalisation time: W'[n][k] = W[n][k] * gamma[k] (rounded to T),
// b'[n] = b[n] + sum_k W[n][k] beta[k], wsum[n] = sum_k float(W'[n][k]).  One wave per output row.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void fold_ln_kernel(const T* __restrict__ W, int64_t Kp, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ bias_in,
                                                      T* __restrict__ Wo, float* __restrict__ bias_out, float* __restrict__ wsum,
                                                      int N, int K) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float sb = 0.f, sw = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = to_f(W[(int64_t)n * Kp + k]);
    const T wg = from_f<T>(w * gamma[k]);
    Wo[(int64_t)n * Kp + k] = wg;
    sb += w * beta[k];
    sw += to_f(wg);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sb += __shfl_xor(sb, o); sw += __shfl_xor(sw, o); }
  if (lane == 0) {
    bias_out[n] = (bias_in ? bias_in[n] : 0.f) + sb;
    wsum[n] = sw;
  }
}

int launch_fold_ln(int dtype, const void* W, int64_t Kp, const float* gamma, const float* beta, const float* bias_in, void* Wo,
                   float* bias_out, float* wsum, int N, int K, hipStream_t s) {
  const unsigned nb = (unsigned)((N + 3) / 4);
  switch (dtype) {
    case DT_F32: hipLaunchKernelGGL((fold_ln_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)W, Kp, gamma, beta, bias_in, (float*)Wo, bias_out, wsum, N, K); break;
    case DT_F16: hipLaunchKernelGGL((fold_ln_kernel<f16>), dim3(nb), dim3(256), 0, s, (const f16*)W, Kp, gamma, beta, bias_in, (f16*)Wo, bias_out, wsum, N, K); break;
    case DT_BF16: hipLaunchKernelGGL((fold_ln_kernel<bf16>), dim3(nb), dim3(256), 0, s, (const bf16*)W, Kp, gamma, beta, bias_in, (bf16*)Wo, bias_out, wsum, N, K); break;
    default: TANGO_FAIL("fold_ln: bad dtype");
  }
  TANGO_HIP(hipGetLastError());
  return 0;
}

}  // namespace tango
