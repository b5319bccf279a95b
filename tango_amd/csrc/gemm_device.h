// Device-side pieces shared by the MFMA GEMM kernels (gemm.hip, conv_halo.hip): MFMA wrappers, the activation helper
// and the common epilogue (bias / per-step bias / activation / residual / GEGLU / V^T / int16 / split-K partials).
#pragma once
#include <cstdint>

#include "common.h"

namespace tango {

template <typename T> struct Mma;
template <> struct Mma<float> {
  __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
  }
};
template <> struct Mma<f16> {
  __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<bf16> {
  __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};

// Prologue activation on one 16-byte operand piece (the vocoder's `conv(leaky_relu(x))`, hifigan/models.py:96-103).  The wave-uniform
// dispatch on `act` sits OUTSIDE the element loop (round 5): written per element, hipcc inlined the whole apply_act switch -- tanh, erf
// and exp code included -- eight times per piece, inside the k-loop of the tile kernels (12 500 instructions in front of the first MFMA
// of gemm_kernel<f16, 128, 128, 128, 2, 2, CONV1D>, a scalar compare-and-branch chain per ELEMENT), which made the vocoder's `a_act`
// convolutions VALU / branch-bound at ~200 TFLOP/s.  Same arithmetic per element, so results are bit-identical.
template <typename T> __device__ __forceinline__ u32x4 act_vec(u32x4 v, int act, float slope) {
  constexpr int EPV = 16 / sizeof(T);
  T e[EPV];
  __builtin_memcpy(e, &v, 16);
  if (act == ACT_LRELU) {
#pragma unroll
    for (int i = 0; i < EPV; ++i) { const float x = to_f(e[i]); e[i] = from_f<T>(x > 0.f ? x : x * slope); }
  } else {      // ACT_SILU: the only other prologue the launcher admits (gemm.hip launch_t)
#pragma unroll
    for (int i = 0; i < EPV; ++i) e[i] = from_f<T>(silu_f(to_f(e[i])));
  }
  __builtin_memcpy(&v, e, 16);
  return v;
}

enum : int { MODE_LINEAR = 0, MODE_CONV2D = 1, MODE_CONV1D = 2 };

// Shared epilogue of the GEMM kernels: acc[a][b] is the 16x16 tile at rows m_base + b*16.., cols n_base + a*16..
// (A paired-tile variant with 16-byte residual loads/stores was measured in round 1: correct, not faster -- the
// cross-lane exchange costs what the wider accesses save.  Note for future work: cross-lane intrinsics are
// `convergent`; LLVM refuses to fully unroll loops containing them and the accumulator array then lands in scratch.)
// EPIK / ACTK / F32K >= 0: that epilogue kind / activation / output kind fixed at compile time; -1: read from GemmParams at run time
// (round 5: see gemm_epilogue below -- the dispatch happens ONCE, outside the TM x TN unrolled loop)
template <typename T, int TM, int TN, int MODE, int EPIK, int ACTK, int F32K>
__device__ __forceinline__ void gemm_epilogue_body(const GemmParams& p, f32x4 (&acc)[TN][TM], const int m_base, const int n_base,
                                                   const int lane, const int zb, const int split) {
  // ---------------- epilogue ----------------
  const int g4 = (lane >> 4) * 4;
  const int epi = EPIK >= 0 ? EPIK : p.epi;
  const int e_act = ACTK >= 0 ? ACTK : p.e_act;
  const bool out_f32 = F32K >= 0 ? (F32K != 0) : (p.out_f32 != 0);
  if (p.splitk > 1) {   // raw fp32 partial tile -> workspace; the reduce kernel finishes the job
    float* wsb = p.ws + (int64_t)split * p.M * p.N;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int m = m_base + b * 16 + (lane & 15);
      if (m >= p.M) continue;
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        const int n = n_base + a * 16 + g4;
        if (n + 3 < p.N) *(f32x4*)(wsb + (int64_t)m * p.N + n) = acc[a][b];
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < p.N) wsb[(int64_t)m * p.N + n + r] = acc[a][b][r];
        }
      }
    }
    return;
  }
  const float* bias = p.bias ? p.bias + (int64_t)zb * p.sBias : nullptr;
  const float* bias2 = nullptr;
  if (p.bias2) bias2 = p.bias2 + (int64_t)(p.step_ptr ? *p.step_ptr : 0) * p.bias2_stride;
  unsigned char* Ob = (unsigned char*)p.out;
  const unsigned char* Rb = (const unsigned char*)p.R;
  const int osz = epi == EPI_I16 ? 2 : (out_f32 ? 4 : (int)sizeof(T));
  Ob += (int64_t)zb * p.sO * osz;
  if (Rb) Rb += (int64_t)zb * p.sR * (int64_t)sizeof(T);

  int64_t orow[TM], vtrow[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int m = m_base + b * 16 + (lane & 15);
    orow[b] = -1; vtrow[b] = 0;
    if (m < p.M) {
      if (epi == EPI_VT) {
        const int bb = m / p.vt_S;
        const int sq = m - bb * p.vt_S;
        vtrow[b] = (int64_t)bb * (p.N - p.vt_n0) * p.vt_ld + (p.vt_perm ? vt_perm_pos(sq) : sq);
      }
      if (MODE == MODE_CONV1D) {
        const int bb = m / p.rows_pb, q = m - bb * p.rows_pb;
        orow[b] = (int64_t)bb * p.Lout + (int64_t)q * p.out_mul + p.out_off;
      } else {
        orow[b] = m;
      }
    }
  }

#pragma unroll
  for (int a = 0; a < TN; ++a) {
    if (epi == EPI_GEGLU && (a & 1)) continue;
    const int nt = n_base + a * 16;   // tile base column (packed order)
    const int n = nt + g4;
    if (n >= p.N) continue;
    // per-column constants of this lane's 4 output channels
    float cb[4] = {0.f, 0.f, 0.f, 0.f}, cg[4] = {0.f, 0.f, 0.f, 0.f};
    if (!p.bias_rows) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r < p.N) {
          if (bias) cb[r] = bias[n + r];
          if (bias2) cb[r] += bias2[n + r];
          if (epi == EPI_GEGLU && bias) cg[r] = bias[n + 16 + r];
        }
      }
    }
    int oc = n, ncols = p.N;
    if (epi == EPI_GEGLU) { oc = (nt >> 1) + g4; ncols = p.N >> 1; }
    const bool to_vt = (epi == EPI_VT) && n >= p.vt_n0;
    if (epi == EPI_VT) ncols = p.vt_n0;
    const bool full = (oc + 3 < ncols);
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      if (orow[b] < 0) continue;
      float v[4];
      const float rb = (bias && p.bias_rows) ? bias[orow[b]] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r] * p.alpha + cb[r] + rb;
      if (epi == EPI_GEGLU) {
        // packed rows: [16 value | 16 gate] blocks -> out col = nt/2 + g4 + r
        float gt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gt[r] = acc[a + 1 < TN ? a + 1 : a][b][r] * p.alpha + cg[r];
        if constexpr (sizeof(T) == 2) glu_gate4<T>(v, gt, 0);      // 16-bit: exact-erf polynomial gate only (launch_gemm refuses glu_tanh here): the tanh
        else glu_gate4<T>(v, gt, p.glu_tanh);                       // form's ocml code at every unrolled call site was most of these epilogues' 30 000 instructions
      } else if (e_act != ACT_NONE) {
        apply_act4(v, e_act, p.e_slope);
      }
      if (Rb) {
        const T* rp = (const T*)Rb + orow[b] * p.ldr + oc;
        if (full && ((p.ldr | oc) & 3) == 0) {
          T rv[4];
          __builtin_memcpy(rv, rp, 4 * sizeof(T));
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += to_f(rv[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (oc + r < ncols) v[r] += to_f(rp[r]);
        }
      }
      if (p.out_scale != 1.f) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= p.out_scale;
      }
      if (to_vt) {
        T* vp = (T*)p.vt + vtrow[b] + (int64_t)(n - p.vt_n0) * p.vt_ld;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < p.N) vp[(int64_t)r * p.vt_ld] = from_f<T>(v[r]);
      } else if (epi == EPI_I16) {
        int16_t* op = (int16_t*)Ob + orow[b] * p.ldo + oc;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (oc + r < ncols) op[r] = (int16_t)(int)v[r];   // C truncation, int16 wrap (hifigan/utilities.py:81)
      } else if (out_f32) {
        float* op = (float*)Ob + orow[b] * p.ldo + oc;
        if (full && ((p.ldo | oc) & 3) == 0) {
          *(f32x4*)op = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (oc + r < ncols) op[r] = v[r];
        }
      } else {
        T* op = (T*)Ob + orow[b] * p.ldo + oc;
        T tv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) tv[r] = from_f<T>(v[r]);
        if (full && ((p.ldo | oc) & 3) == 0) {
          __builtin_memcpy(op, tv, 4 * sizeof(T));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (oc + r < ncols) op[r] = tv[r];
        }
      }
    }
  }
}


// Shared epilogue of the tile / LDS-DMA gather kernels.  Round 5: the TM x TN unrolled loop used to test p.epi / p.e_act / p.out_f32
// in every iteration, with the code of EVERY epilogue kind and every activation (tanh, erf, exp) inlined at every call site: 30 000
// instructions per kernel, of which a launch executes a sparse few -- a chain of jumps over kilobytes of dead code per iteration
// (the kernels of the B = 1 path, the vocoder and the VAE).  Now the kind is dispatched ONCE and the common ones run straight-line,
// specialised bodies; anything else takes the fully run-time body (identical arithmetic in every body).
template <typename T, int TM, int TN, int MODE>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[TN][TM], const int m_base, const int n_base,
                                              const int lane, const int zb, const int split) {
  if (p.epi == EPI_NONE && !p.out_f32) {
    if (p.e_act == ACT_NONE) return gemm_epilogue_body<T, TM, TN, MODE, EPI_NONE, ACT_NONE, 0>(p, acc, m_base, n_base, lane, zb, split);
    if (p.e_act == ACT_LRELU) return gemm_epilogue_body<T, TM, TN, MODE, EPI_NONE, ACT_LRELU, 0>(p, acc, m_base, n_base, lane, zb, split);
  }
  if (p.epi == EPI_GEGLU && !p.out_f32) return gemm_epilogue_body<T, TM, TN, MODE, EPI_GEGLU, ACT_NONE, 0>(p, acc, m_base, n_base, lane, zb, split);
  if (p.epi == EPI_VT && !p.out_f32 && p.e_act == ACT_NONE) return gemm_epilogue_body<T, TM, TN, MODE, EPI_VT, ACT_NONE, 0>(p, acc, m_base, n_base, lane, zb, split);
  gemm_epilogue_body<T, TM, TN, MODE, -1, -1, -1>(p, acc, m_base, n_base, lane, zb, split);
}

// LDS-staged epilogue for the 8-wave kernels (plain outputs: EPI_NONE, storage dtype T).
// The direct epilogue above stores 8 bytes per lane = 32-byte runs per output row; measured on the level-0 conv
// (ablation in profiles/r1_v16_conv_halo_ablation.txt) those partial-line writes cost 175 us of a 660 us kernel.
// Here each wave parks 32 rows x (TN*16) fp32 results in its private LDS region (bias / per-step bias / activation
// already applied, same fp32 arithmetic and order as the direct path -> bit-identical results), then walks the rows
// with 16-byte pieces: residual read, add, scale, convert, 16-byte store -> TN*16*sizeof(T)-byte contiguous runs.
// Caller guarantees: all waves are past their last LDS read (barrier), p.epi is EPI_NONE or EPI_GEGLU, !p.out_f32, n_base + TN*16 <= N,
// ldo / ldr multiples of 16/sizeof(T), 16-byte aligned out / R.
// KIND (round 5, as gemm_epilogue above): 0 = plain (EPI_NONE, no activation), 1 = GEGLU, -1 = read p.epi / p.e_act at run time
template <typename T, int TM, int TN, int KIND>
__device__ __forceinline__ void gemm_epilogue_staged_body(const GemmParams& p, f32x4 (&acc)[TN][TM], const int m_base, const int n_base,
                                                          const int lane, unsigned char* stage) {
  constexpr int EPV = 16 / (int)sizeof(T);
  constexpr int WN = TN * 16;
  constexpr int PITCH = WN * 4 + 16;       // bytes per staged row (+16: spreads rows over the banks)
  constexpr int PPR = WN / EPV;            // 16-byte output pieces per row
  static_assert((32 * PPR) % 128 == 0 || (TN & 1), "staged epilogue: 32 rows must split into whole wave passes (also at half width for GEGLU)");
  const int g4 = (lane >> 4) * 4;
  const bool geglu = KIND >= 0 ? (KIND == 1) : (p.epi == EPI_GEGLU);
  const int e_act = KIND >= 0 ? (int)ACT_NONE : p.e_act;
  const float* bias = p.bias;
  const float* bias2 = p.bias2 ? p.bias2 + (int64_t)(p.step_ptr ? *p.step_ptr : 0) * p.bias2_stride : nullptr;
  float cb[TN][4];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n_base + a * 16 + g4 + r;
      float c = 0.f;
      if (!p.bias_rows && bias) c = bias[n];
      if (!p.bias_rows && bias2) c += bias2[n];
      cb[a][r] = c;
    }
  const T* Rb = (const T*)p.R;
  T* Ob = (T*)p.out;
#pragma unroll
  for (int half = 0; half < TM / 2; ++half) {
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const int b = half * 2 + bb;
      const int row_l = bb * 16 + (lane & 15);
      const int m = m_base + b * 16 + (lane & 15);
      const float rb = (bias && p.bias_rows && m < p.M) ? bias[m] : 0.f;
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        if (geglu && (a & 1)) continue;
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r] * p.alpha + cb[a][r] + rb;
        if (geglu) {
          // packed rows: [16 value | 16 gate] column blocks -> output column n/2 (same arithmetic as the direct path)
          const int ag = a + 1 < TN ? a + 1 : a;
          float gt[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) gt[r] = acc[ag][b][r] * p.alpha + cb[ag][r];
          if constexpr (sizeof(T) == 2) glu_gate4<T>(v, gt, 0);      // 16-bit: exact-erf polynomial gate only (launch_gemm refuses glu_tanh here): the tanh
        else glu_gate4<T>(v, gt, p.glu_tanh);                       // form's ocml code at every unrolled call site was most of these epilogues' 30 000 instructions
        } else if (e_act != ACT_NONE) {
          apply_act4(v, e_act, p.e_slope);
        }
        *(f32x4*)(stage + row_l * PITCH + ((geglu ? (a >> 1) : a) * 16 + g4) * 4) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int ppr = geglu ? PPR / 2 : PPR;
    for (int idx = lane; idx < 32 * ppr; idx += 64) {
      const int row_l = idx / ppr, pcs = idx - row_l * ppr;
      const int m = m_base + half * 32 + row_l;
      if (m < p.M) {
        float f[EPV];
#pragma unroll
        for (int q = 0; q < EPV / 4; ++q) {
          const f32x4 t = *(const f32x4*)(stage + row_l * PITCH + (pcs * EPV + q * 4) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) f[q * 4 + r] = t[r];
        }
        const int n = (geglu ? (n_base >> 1) : n_base) + pcs * EPV;
        if (Rb) {
          T rv[EPV];
          __builtin_memcpy(rv, Rb + (int64_t)m * p.ldr + n, 16);
#pragma unroll
          for (int e = 0; e < EPV; ++e) f[e] += to_f(rv[e]);
        }
        if (p.out_scale != 1.f) {
#pragma unroll
          for (int e = 0; e < EPV; ++e) f[e] *= p.out_scale;
        }
        T tv[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) tv[e] = from_f<T>(f[e]);
        __builtin_memcpy(Ob + (int64_t)m * p.ldo + n, tv, 16);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmParams& p, f32x4 (&acc)[TN][TM], const int m_base, const int n_base,
                                                     const int lane, unsigned char* stage) {
  if (p.epi == EPI_NONE && p.e_act == ACT_NONE) return gemm_epilogue_staged_body<T, TM, TN, 0>(p, acc, m_base, n_base, lane, stage);
  if (p.epi == EPI_GEGLU) return gemm_epilogue_staged_body<T, TM, TN, 1>(p, acc, m_base, n_base, lane, stage);
  gemm_epilogue_staged_body<T, TM, TN, -1>(p, acc, m_base, n_base, lane, stage);
}

// host-side predicate for the staged epilogue
template <typename T> static inline bool epilogue_can_stage(const GemmParams& p) {
  constexpr int EPV = 16 / (int)sizeof(T);
  if ((p.epi != EPI_NONE && p.epi != EPI_GEGLU) || p.out_f32 || p.splitk > 1 || p.batch != 1) return false;
  if (p.epi == EPI_GEGLU && (p.e_act != ACT_NONE || p.bias_rows)) return false;
  if (p.ldo % EPV != 0 || ((uintptr_t)p.out & 15)) return false;
  if (p.R && (p.ldr % EPV != 0 || ((uintptr_t)p.R & 15))) return false;
  return true;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// counted vmcnt wait with a wave-uniform runtime count (the immediate must be a literal); counts above 8 wait for 8
__device__ __forceinline__ void wait_vmcnt_upto8(const int n) {
  if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
// Phase assignment for the ping-pong loops: 0 for the first wave of this workgroup on its SIMD, 1 for the second (the two
// must run in opposite phases; which waves share a SIMD is the dispatcher's choice, so it is read from HW_REG_HW_ID at run
// time -- bits 5:4 = SIMD id -- and exchanged through 8 words of LDS scratch).  Any assignment is CORRECT (both halves
// execute the same number of barriers); a wrong one only loses the overlap.  mode 0: static wave >> 2 (experiment switch).
__device__ __forceinline__ int pp_phase_half(const int wave, const int lane, unsigned* scratch8, const int mode) {
  if (mode == 0) return wave >> 2;
  const unsigned simd = (__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) >> 4) & 3u;
  if (lane == 0) scratch8[wave] = simd;
  __syncthreads();
  int rank = 0;
  for (int w = 0; w < wave; ++w) rank += scratch8[w] == simd ? 1 : 0;
  return __builtin_amdgcn_readfirstlane(rank & 1);
}

// wave-uniform values forced into SGPRs (v_readfirstlane folds away when the value already lives there): inputs of the
// `asm volatile("" : "+s"(x))` pins that keep a computation in the phase it was written in
__device__ __forceinline__ int sgpr_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned sgpr_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ const unsigned char* sgpr_ptr(const unsigned char* q) {
  const uint64_t v = (uint64_t)(uintptr_t)q;
  const unsigned lo = sgpr_u32((unsigned)v), hi = sgpr_u32((unsigned)(v >> 32));
  return (const unsigned char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

template <int N> __device__ __forceinline__ void wait_vmcnt_lit() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// raw workgroup barrier (no vmcnt drain: LDS-DMAs stay in flight across it) that the scheduler may not move code across
__device__ __forceinline__ void pp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace tango
