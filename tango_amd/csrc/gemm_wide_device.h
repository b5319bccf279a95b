// Device-side epilogues of the 256 x 320 kernels (gemm_wide.hip, conv_wide.hip): one wave's 64 x 160 accumulator block.
#pragma once
#include "common.h"
#include "gemm_device.h"

namespace tango {

constexpr int WIDE_STAGE_BYTES = 11520;   // per-wave staging: max(16 rows x 656 B, 32 rows x 336 B, 80 columns x 144 B); 1280 B of constants follow

// one accumulator quad -> pre-activation values: acc * alpha + bias, or the folded-LayerNorm form rstd * (acc - mean * wsum) + bias
// (cst points at this quad's bias; its wsum sits 160 floats further)
template <bool LN>
__device__ __forceinline__ f32x4 wide_col_value(const f32x4 av, const float mean, const float rstd, const float alpha, const float* cst) {
  const f32x4 cbv = *(const f32x4*)cst;
  f32x4 v;
  if (LN) {
    const f32x4 cwv = *(const f32x4*)(cst + 160);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t;
      asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(t) : "v"(mean), "v"(cwv[r]), "v"(av[r]));
      v[r] = rstd * t + cbv[r];
    }
  } else {
    v = av * alpha + cbv;
  }
  return v;
}

// Epilogue of one wave's 64 x 160 accumulator block, written for this tile (gemm_epilogue_staged16's generic residual /
// predication logic compiled into ~3000 instructions of branches and took as long as the main loop: tools trace, 21-40 us per
// tile).  GEGLU and RES are compile-time; the arithmetic and its order are those of gemm_epilogue_staged (acc * alpha + bias
// [+ bias2] -> activation / gate -> fp32 staging -> + residual -> * out_scale -> one rounding), so results are bit-identical.
// Passes of 16 rows (32 for GEGLU, whose output rows are half as wide) = 320 16-byte output pieces = exactly five wave
// iterations, no partial one.  vmcnt retires in order and counts stores: the residual pieces of pass p+1 are requested BEFORE
// the stores of pass p are issued, so waiting for them (vmcnt <= 2 * NIT) never waits for a store.
// LN: folded LayerNorm (see linear_stream.hip): y = rstd[m] * (acc - mean[m] * wsum[n]) + b'[n] with the row statistics the
// main loop accumulated; acc - mean * wsum is the single-instruction form that linear_stream.hip's race notes call for.
// The per-column constants (bias [+ bias2], wsum) sit in LDS behind the staging rows, not in 80 VGPRs.
template <typename T, bool GEGLU, bool RES, bool LN>
__device__ __forceinline__ void wide_epilogue(const GemmParams& p, f32x4 (&acc)[10][4], const float (&mean)[4], const float (&rstd)[4],
                                              const int m_base, const int n_base, const int lane, unsigned char* stage) {
  constexpr int TN = 10, NIT = 5;
  constexpr int OC = GEGLU ? 80 : 160;                 // output columns of this wave
  constexpr int PITCH = OC * 4 + 16;                   // fp32 staging row
  constexpr int PPR = OC / 8;                          // 16-byte output pieces per row
  constexpr int RPP = GEGLU ? 32 : 16;                 // rows per pass
  constexpr int NPASS = 64 / RPP;
  const int l15 = lane & 15, g4 = (lane >> 4) * 4;
  const float* bias2 = p.bias2 ? p.bias2 + (int64_t)(p.step_ptr ? *p.step_ptr : 0) * p.bias2_stride : nullptr;
  float* const cst = (float*)(stage + WIDE_STAGE_BYTES);       // [bias 160 | wsum 160]
  // rows of the leading range (GemmParams::rowvec): a per-sample vector joins the bias, the output goes to out_lo (wave-uniform: the
  // wave's 64 rows lie in one sample and on one side of the range boundary)
  const bool lo = !GEGLU && p.rowvec != nullptr && (int64_t)m_base < p.rowvec_rows;
  if (lane < 40) {
    f32x4 bv = p.bias ? *(const f32x4*)(p.bias + n_base + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias2) bv += *(const f32x4*)(bias2 + n_base + lane * 4);
    if (lo) bv += *(const f32x4*)(p.rowvec + (int64_t)(m_base / p.rowvec_per) * p.N + n_base + lane * 4);
    *(f32x4*)(cst + lane * 4) = bv;
    if (LN) *(f32x4*)(cst + 160 + lane * 4) = *(const f32x4*)(p.wsum + n_base + lane * 4);
  }
  __builtin_amdgcn_wave_barrier();
  const int ocol0 = GEGLU ? (n_base >> 1) : n_base;
  const T* Rb = (const T*)p.R;
  T* Ob = lo ? (T*)p.out_lo : (T*)p.out;
  const int64_t ldo = lo ? p.ldo_lo : p.ldo;
  int prow[NIT], pcol[NIT];                            // (row within the pass, first output column) of this lane's piece per iteration
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = lane + it * 64;
    prow[it] = idx / PPR;
    pcol[it] = (idx - prow[it] * PPR) * 8;
  }
  u32x4 rv[2][NIT];
  auto fetch_res = [&](const int ps, u32x4 (&dst)[NIT]) {
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      dst[it] = *(const u32x4*)(Rb + (int64_t)(m_base + ps * RPP + prow[it]) * p.ldr + ocol0 + pcol[it]);
  };
  if (RES) fetch_res(0, rv[0]);
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
    for (int bb = 0; bb < RPP / 16; ++bb) {
      const int b = ps * (RPP / 16) + bb;
#pragma unroll
      for (int a = 0; a < TN; a += GEGLU ? 2 : 1) {
        f32x4 v = wide_col_value<LN>(acc[a][b], mean[b], rstd[b], p.alpha, cst + a * 16 + g4);
        if (GEGLU) {
          const f32x4 gv = wide_col_value<LN>(acc[a + 1][b], mean[b], rstd[b], p.alpha, cst + (a + 1) * 16 + g4);
          const float gt[4] = {gv[0], gv[1], gv[2], gv[3]};
          glu_gate4<T>(v, gt, p.glu_tanh);
        }
        *(f32x4*)(stage + (bb * 16 + l15) * PITCH + ((GEGLU ? (a >> 1) : a) * 16 + g4) * 4) = v;
      }
    }
    if (RES && ps + 1 < NPASS) fetch_res(ps + 1, rv[(ps + 1) & 1]);
    __builtin_amdgcn_wave_barrier();
    if (RES) {
      // the NIT residual loads of this pass are older than the previous pass's NIT stores and the NIT loads just issued
      if (ps == 0) { if (NPASS > 1) wait_vmcnt_lit<NIT>(); else wait_vmcnt_lit<0>(); }
      else if (ps + 1 < NPASS) wait_vmcnt_lit<2 * NIT>();
      else wait_vmcnt_lit<NIT>();
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const f32x4 lo = *(const f32x4*)(stage + prow[it] * PITCH + pcol[it] * 4);
      const f32x4 hi = *(const f32x4*)(stage + prow[it] * PITCH + pcol[it] * 4 + 16);
      float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (RES) {
        T r8[8];
        __builtin_memcpy(r8, &rv[ps & 1][it], 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += to_f(r8[e]);
      }
      if (p.out_scale != 1.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= p.out_scale;
      }
      T tv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) tv[e] = from_f<T>(f[e]);
      u32x4 o;
      __builtin_memcpy(&o, tv, 16);
      *(u32x4*)(Ob + (int64_t)(m_base + ps * RPP + prow[it]) * ldo + ocol0 + pcol[it]) = o;
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// EPI_VT tiles that lie in the V projection (n0 >= vt_n0): the wave's 64 rows x 160 columns go out TRANSPOSED,
//   vt[((m / vt_S) * (N - vt_n0) + (n - vt_n0)) * vt_ld + (m % vt_S)],
// through an LDS transpose in T: two passes of 80 columns x 64 rows (row pitch 144 B), then 16-byte pieces along the sequence
// axis -> 128-byte contiguous runs per column per wave (the streaming kernel stores these 2 bytes at a time).
template <typename T, bool LN>
__device__ __forceinline__ void wide_epilogue_vt(const GemmParams& p, f32x4 (&acc)[10][4], const float (&mean)[4], const float (&rstd)[4],
                                                 const int m_base, const int n_base, const int lane, unsigned char* stage) {
  constexpr int TN = 10, TM = 4, CPITCH = 64 * 2 + 16;
  const int l15 = lane & 15, g4 = (lane >> 4) * 4;
  float* const cst = (float*)(stage + WIDE_STAGE_BYTES);
  if (lane < 40) {
    *(f32x4*)(cst + lane * 4) = p.bias ? *(const f32x4*)(p.bias + n_base + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (LN) *(f32x4*)(cst + 160 + lane * 4) = *(const f32x4*)(p.wsum + n_base + lane * 4);
  }
  __builtin_amdgcn_wave_barrier();
  const int bb = m_base / p.vt_S, s0 = m_base - bb * p.vt_S;
  T* const vbase = (T*)p.vt + ((int64_t)bb * (p.N - p.vt_n0) + (n_base - p.vt_n0)) * p.vt_ld + s0;
  // token b * 16 + l15 of the wave's 64 rows -> its position in the staged column (vt_perm: the attention kernel's fragment order inside every
  // block of 32 -- s0 is a multiple of 64, so the blocks of the tile are the blocks of the sequence)
  int tpos[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) tpos[b] = p.vt_perm ? vt_perm_pos(b * 16 + l15) : b * 16 + l15;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
    for (int al = 0; al < TN / 2; ++al) {
      const int a = ps * (TN / 2) + al;
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const f32x4 v = wide_col_value<LN>(acc[a][b], mean[b], rstd[b], p.alpha, cst + a * 16 + g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) *(T*)(stage + (al * 16 + g4 + r) * CPITCH + tpos[b] * 2) = from_f<T>(v[r]);
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 10; ++it) {
      const int idx = lane + it * 64, col = idx >> 3, pc = idx & 7;
      const u32x4 o = *(const u32x4*)(stage + col * CPITCH + pc * 16);
      *(u32x4*)(vbase + (int64_t)(ps * 80 + col) * p.vt_ld + pc * 8) = o;
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// Split-K partial tile: the wave's 64 x 160 fp32 accumulators, unmodified, into the workspace slice of this split
// (ws [splits][M][N]; splitk_reduce_kernel sums the splits in order and applies the epilogue).  16 rows per pass through the
// staging slice, 16-byte pieces: 640 per pass = ten full wave iterations.
__device__ __forceinline__ void wide_epilogue_raw(const GemmParams& p, f32x4 (&acc)[10][4], const int split, const int m_base, const int n_base,
                                                  const int lane, unsigned char* stage) {
  constexpr int PITCH = 160 * 4 + 16;
  const int l15 = lane & 15, g4 = (lane >> 4) * 4;
  float* const wsb = p.ws + ((int64_t)split * p.M + m_base) * p.N + n_base;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int a = 0; a < 10; ++a) *(f32x4*)(stage + l15 * PITCH + (a * 16 + g4) * 4) = acc[a][b];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 10; ++it) {
      const int idx = lane + it * 64, row = idx / 40, pc = idx - row * 40;
      *(f32x4*)(wsb + (int64_t)(b * 16 + row) * p.N + pc * 4) = *(const f32x4*)(stage + row * PITCH + pc * 16);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace tango
