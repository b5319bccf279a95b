// Fused feed-forward of the level-0 BasicTransformerBlock (round 6):
//     out = x + ff.net.2( GEGLU( ff.net.0.proj( LayerNorm3(x) ) ) )
// (mustango/diffusers/src/diffusers/models/attention.py:338-387 FeedForward / :412-433 GEGLU, :326-335 norm3 + residual) in ONE launch: the
// [M, 4C] GEGLU output (0.67 GB written + read per site at config 3, 6.7 GB per step) never reaches HBM.  C = 320, hidden H = 1280, 16-bit.
//
// Organisation (activation-stationary, weights streamed through LDS; the "swapped" product of attention.hip so that nothing crosses lanes
// between the two GEMMs):
//   * a 512-thread workgroup owns 128 rows.  Wave w = (rg = w & 3, wn = w >> 2) holds the NORMALISED rows rg*32 .. +31 as MFMA B-operand
//     fragments in registers for the whole kernel (LayerNorm3 is applied once, on the way in: two-pass fp32 statistics, one rounding to T --
//     the folded weights W' = W gamma, b' = b + W beta of the engine's other LayerNorm-folded linears are multiplied with (x - mean) rstd).
//   * the hidden dimension is walked in 40 chunks of 32 units.  GEMM 1 of chunk c, H^T = W1'(c) x^T: each wave of a pair computes 16 of the
//     32 hidden units (value tile + gate tile, 40 MFMAs), applies bias + GEGLU in registers and leaves its 4 packed outputs per row in LDS
//     next to its partner's, in the order the second product's B operand wants them (hidden 32c + 8g + 4wn + r: the k-slot order is
//     arranged by CHOOSING which weight rows the LDS-DMA puts at which tile row -- no permuted weight copy, no cross-lane traffic).
//   * GEMM 2 of chunk c - 1, out^T += W2(c-1) P^T: each wave owns 160 of the 320 output columns for its 32 rows (20 MFMAs, 80 accumulator
//     registers); it runs in the same iteration as GEMM 1 of chunk c, so the GEGLU VALU work has independent MFMAs beside it.
//   * weights arrive by LDS-DMA (global_load_lds, 16 rows x 64 B per instruction, source-side XOR swizzle as in gemm_wide.hip) as bundles
//     {W1'(c): 64 rows x 640 B, W2(c-1): 320 rows x 64 B} = 60 KB, two stages; ONE workgroup barrier per iteration (60 MFMAs per wave).
//   * epilogue: + bias + residual (the raw x rows, re-read) -> one rounding, 8-byte stores in the accumulator layout.
// LDS: 2 x 60 KB stages + 16 KB P exchange (two parities) + 10 KB bias = 146 KB: one workgroup per CU, two waves per SIMD.
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "tuning.h"
#include "gemm_device.h"

namespace tango {

namespace {
constexpr int FF_C = 320, FF_H = 1280, FF_BM = 128;
constexpr int FF_KS = FF_C / 32;                 // k-steps of GEMM 1
constexpr int FF_NCH = FF_H / 32;                // hidden chunks
constexpr int FF_STAGE = 60 * 1024;
constexpr int FF_PBUF = 8 * 1024;                // one parity of the P exchange: [rg 4][rt 2][g 4][l15 16][wn 2] x 8 B
constexpr int FF_LDS = 2 * FF_STAGE + 2 * FF_PBUF + 2 * FF_H * 4;
}  // namespace

// LDS fragment reads the compiler does not see (hipcc waits lgkmcnt(0) at the first use of ANY ds_read result, so reads issued ahead buy
// nothing when written in C++): "=v" destinations, waited for by a counted statement that names the pair about to be consumed "+v"
// (cdna_hip_programming.md, asm loads form (ii)).  LDS operations return in order and nothing else of this loop counts on lgkmcnt
// (no scalar loads inside it: checked in the .s), so lgkmcnt(N) = "all but the newest N reads have landed".
template <int OFF> __device__ __forceinline__ void ff_lds_read(u32x4& v, const unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void ff_lds_wait(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N> __device__ __forceinline__ void ff_lds_wait4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
// v[r] *= gelu(g[r]) with common.h's exact-erf polynomial (same coefficients, same operation order per value as gelu_erf_poly2) on plain
// v_fma_f32: four independent chains side by side.  The file is compiled with -fno-slp-vectorize so that hipcc does not pair them up again.
__device__ __forceinline__ void ff_gelu_gate4_scalar(f32x4& v, const float (&g)[4]) {
  float xc[4], t[4], q[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { xc[r] = __builtin_amdgcn_fmed3f(g[r], -4.25f, 4.25f); t[r] = xc[r] * xc[r]; }
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = 1.498029197e-11f + t[r] * 1.123676848e-12f;
  constexpr float cf[8] = {-7.180728823e-09f, 3.825664002e-07f, -1.045047183e-05f, 1.811638670e-04f, -2.193588131e-03f, 1.957916536e-02f,
                           -1.326353318e-01f, 7.977887478e-01f};
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) q[r] = __builtin_fmaf(q[r], t[r], cf[k]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float e = q[r] * xc[r], h = g[r] * 0.5f;
    v[r] = v[r] * __builtin_fmaf(h, e, h);
  }
}

template <int I, int N, typename F> __device__ __forceinline__ void ff_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); ff_for<I + 1, N>(f); }
}

// VAR: compile-time variant bits (the shipped one is FF_VAR_DEFAULT; the others are A/B arms and timing probes, TANGO_FF_VAR):
//   bit 0      PLAIN  the compiler-scheduled main loop instead of the asm LDS stream (TANGO_FF_FUSED=2)
//   bits 1-3   ABL    timing probes, results WRONG: 1 = no GEGLU arithmetic (P = value + gate), 2 = no LDS-DMA in the steady state (stale
//                     weights), 4 = no GEMM 2 MFMAs
//   bit 4      ORDER  1 = the wn = 1 waves run GEMM 2 -> GEMM 1 -> GEGLU (out of step with their SIMD partners)
//   bit 5      GSC    GELU polynomial on plain v_fma_f32 (four interleaved chains) instead of v_pk_fma_f32 pairs (MI355X_MICROARCH.md: packed
//                     f32 VALU beside MFMAs is an anti-lever)
//   bits 6-7   DPL    where the LDS-DMAs of bundle c + 1 are issued: 0 = one behind each of GEMM 1's k-steps 0..7, 1 = all of them right behind
//                     the barrier, 2 = all of them in front of the GEGLU block
//   bit 9      PRIO   s_setprio 1 from the barrier to the end of the iteration's MFMAs
//   bit 10     RING4  ring of FOUR fragment pairs (three ahead of their use)
//   bit 8      EST    epilogue through per-wave fp32 staging in LDS with 16-byte residual loads / stores (0: 8-byte accesses in the accumulator layout)
constexpr int FF_VAR_DEFAULT = 0x120;
template <typename T, int VAR>
__global__ __launch_bounds__(512) void ff_fused_kernel(const FFParams p) {
  constexpr bool PLAIN = (VAR & 1) != 0;
  constexpr int ABL = (VAR >> 1) & 7;
  constexpr int ORDER = (VAR >> 4) & 1;
  constexpr bool GSC = ((VAR >> 5) & 1) != 0;
  constexpr int DPL = (VAR >> 6) & 3;
  constexpr bool EST = ((VAR >> 8) & 1) != 0;
  constexpr bool PRIO = ((VAR >> 9) & 1) != 0;
  constexpr int RD = ((VAR >> 10) & 1) ? 4 : 3;       // ring depth in pairs
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  unsigned char* const pbuf = dsm + 2 * FF_STAGE;
  float* const b1s = (float*)(dsm + 2 * FF_STAGE + 2 * FF_PBUF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 3, wn = wave >> 2;
  const int l15 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * FF_BM;

  // ---- LDS-DMA sources: 60 groups of 16 rows x 64 B per bundle; wave w issues groups w + 8 i.  i < 5: W1' (group = tile * 10 + k-step,
  // tile = 2 wn' + gate), i >= 5: W2 (group - 40 = 16-row block of output columns).  Tile row m of (wn', gate) is the weight row of hidden
  // unit 32 c + 8 (m >> 2) + 4 wn' + (m & 3); the packed matrix interleaves [16 value | 16 gate] rows per 16 hidden units (engine.hip reg_xf).
  const int lrow = lane >> 2;
  const int pc = (lane & 3) ^ ((4 - (lrow >> 2)) & 3);
  // per lane: ONE offset into W1' (row of the tile's 16 that this lane fetches, for value tile of wn' = 0) and ONE into W2; everything that
  // depends on the group (tile, k-step, 16-row block) is wave-uniform and rides on the scalar base
  const unsigned v_off1 = (unsigned)((int64_t)((lrow >> 3) * 32 + ((lrow >> 2) & 1) * 8 + (lrow & 3)) * p.ld1 * (int64_t)sizeof(T)) + pc * 16;
  const unsigned v_off2 = (unsigned)((int64_t)lrow * p.ld2 * (int64_t)sizeof(T)) + pc * 16;
  unsigned s_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int grp = wave + 8 * i;
    if (i < 5) {
      const int tile = grp / FF_KS, ks = grp - tile * FF_KS;
      const int twn = tile >> 1, gate = tile & 1;
      s_off[i] = sgpr_u32((unsigned)((int64_t)(gate * 16 + twn * 4) * p.ld1 * (int64_t)sizeof(T)) + ks * 64);
    } else {
      s_off[i] = sgpr_u32((unsigned)((int64_t)((grp - 40) * 16) * p.ld2 * (int64_t)sizeof(T)));
    }
  }
  const bool has8 = wave < 4;                     // group w + 56 < 60
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
  const unsigned char* const W1b = (const unsigned char*)p.w1;
  const unsigned char* const W2b = (const unsigned char*)p.w2;
  auto dma = [&](const int i, const unsigned char* src, const unsigned ldst) {
    unsigned o = i < 5 ? v_off1 : v_off2;
    asm volatile("" : "+v"(o));
    __builtin_amdgcn_global_load_lds((gptr_t)(src + s_off[i] + o), (lptr_t)(uintptr_t)(ldst + (unsigned)i * 8192u), 16, 0, 0);
  };
  // bundle cb = {W1'(cb) if cb < NCH, W2(cb - 1) if cb >= 1} into stage cb & 1
  auto issue_bundle = [&](const int cb) {
    const unsigned ldst = lds0 + (unsigned)(cb & 1) * FF_STAGE + (unsigned)wave * 1024u;
    const unsigned char* s1 = W1b + (int64_t)cb * 64 * p.ld1 * (int64_t)sizeof(T);
    const unsigned char* s2 = W2b + (int64_t)(cb - 1) * 64;
    if (cb < FF_NCH) {
#pragma unroll
      for (int i = 0; i < 5; ++i) dma(i, s1, ldst);
    }
    if (cb >= 1) {
      dma(5, s2, ldst);
      dma(6, s2, ldst);
      if (has8) dma(7, s2, ldst);
    }
  };

  issue_bundle(0);
  // folded bias b' (2H floats, packed order) -> LDS
  for (int i = tid; i < 2 * FF_H / 4; i += 512) *(f32x4*)(b1s + i * 4) = *(const f32x4*)(p.b1 + i * 4);

  // ---- this wave's 32 rows: load, LayerNorm (two-pass fp32), keep as B-operand fragments ----
  u32x4 xf[2][FF_KS];
  {
    const T* X = (const T*)p.x;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const T* xr = X + (int64_t)(m0 + rg * 32 + rt * 16 + l15) * p.ldx + g * 8;
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) xf[rt][ks] = *(const u32x4*)(xr + ks * 32);
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) {
        T e[8];
        __builtin_memcpy(e, &xf[rt][ks], 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += to_f(e[j]);
      }
      s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
      const float mean = s * (1.0f / FF_C);
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) {
        T e[8];
        __builtin_memcpy(e, &xf[rt][ks], 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = to_f(e[j]) - mean; q += d * d; }
      }
      q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
      const float rstd = 1.0f / sqrtf(q * (1.0f / FF_C) + p.eps);
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) {
        T e[8];
        __builtin_memcpy(e, &xf[rt][ks], 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = from_f<T>((to_f(e[j]) - mean) * rstd);
        __builtin_memcpy(&xf[rt][ks], e, 16);
      }
    }
  }

  f32x4 oacc[10][2];
#pragma unroll
  for (int t = 0; t < 10; ++t) { oacc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; oacc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int foff = l15 * 64 + ((g ^ ((4 - (l15 >> 2)) & 3)) * 16);
  const int w1off = (wn * 2 * FF_KS) * 1024 + foff;                 // value tile of this wave; gate tile FF_KS groups further
  const int w2off = (40 + wn * 10) * 1024 + foff;
  // slot order [g][l15] (not [l15][g]): the 16 lanes of a ds_write_b64 lane group (one g, l15 = 0..15) then write 16 distinct 16-byte slots of the
  // 256-byte bank row, and ds_read_b128's lane groups read 16 distinct ones ([l15][g] was a 4-way conflict on every P store: 30 % of the kernel's
  // LDS-active cycles were conflict replays, profiles/r6_final2_pmc_*)
  const int poff = ((rg * 2) * 64 + g * 16 + l15) * 16;             // + rt * 1024
  // bias of this lane's 4 hidden units per chunk: packed index (2c + (g >> 1)) * 32 + gate * 16 + (g & 1) * 8 + wn * 4
  const int boff = (g >> 1) * 32 + (g & 1) * 8 + wn * 4;

  // one iteration: GEMM 1 of chunk c (D1), GEMM 2 of chunk c - 1 (D2)
  auto iter = [&](const int c, auto d1, auto d2) {
    constexpr bool D1 = decltype(d1)::value, D2 = decltype(d2)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMAs of bundle c (issued one iteration ago) have landed
    __syncthreads();
    if (c + 1 <= FF_NCH) issue_bundle(c + 1);
    const unsigned char* const St = dsm + (c & 1) * FF_STAGE;
    f32x4 hv[2], hg[2];
    if (D1) {
      hv[0] = hv[1] = hg[0] = hg[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) {
        const u32x4 wv = *(const u32x4*)(St + w1off + ks * 1024);
        const u32x4 wg = *(const u32x4*)(St + w1off + (FF_KS + ks) * 1024);
        Mma<T>::run(hv[0], wv, xf[0][ks]);
        Mma<T>::run(hg[0], wg, xf[0][ks]);
        Mma<T>::run(hv[1], wv, xf[1][ks]);
        Mma<T>::run(hg[1], wg, xf[1][ks]);
      }
    }
    if (D2) {
      const unsigned char* const Pb = pbuf + ((c - 1) & 1) * FF_PBUF + poff;
      const u32x4 pf0 = *(const u32x4*)(Pb), pf1 = *(const u32x4*)(Pb + 1024);
#pragma unroll
      for (int t = 0; t < 10; ++t) {
        const u32x4 w2f = *(const u32x4*)(St + w2off + t * 1024);
        Mma<T>::run(oacc[t][0], w2f, pf0);
        Mma<T>::run(oacc[t][1], w2f, pf1);
      }
    }
    if (D1) {
      const f32x4 bv = *(const f32x4*)(b1s + c * 64 + boff), bg = *(const f32x4*)(b1s + c * 64 + boff + 16);
      unsigned char* const Pw = pbuf + (c & 1) * FF_PBUF + poff + wn * 8;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f32x4 v = hv[rt] + bv;
        const f32x4 gt4 = hg[rt] + bg;
        const float gt[4] = {gt4[0], gt4[1], gt4[2], gt4[3]};
        glu_gate4<T>(v, gt, 0);
        T h4[4] = {from_f<T>(v[0]), from_f<T>(v[1]), from_f<T>(v[2]), from_f<T>(v[3])};
        u32x2 w;
        __builtin_memcpy(&w, h4, 8);
        *(u32x2*)(Pw + rt * 1024) = w;
      }
    }
  };

  // Steady-state iteration (1 <= c < NCH), the LDS stream in inline asm through a ring of three fragment pairs, two pairs ahead of their use:
  //   items 0..9   GEMM 1 pair of k-step j (value tile, gate tile)          -> 4 MFMAs
  //   item  10     this lane's bias quads of chunk c (value, gate)          -> GEGLU
  //   items 11..15 GEMM 2 pair of output tiles 2 (j - 11), 2 (j - 11) + 1   -> 4 MFMAs
  // The pair at stream position P + 3 is requested right behind the consumption of position P; the P fragments of chunk c - 1 are requested
  // first of all.  The LDS-DMAs of bundle c + 1 go out one by one behind the MFMAs of k-steps 0..7 (gemm_wide.hip SCH = 1).
  // DM: 2 = that bundle has both parts, 1 = W2 only.
  // ORD (TANGO_FF_ORDER): the order of the three phases inside the barrier interval.  0: GEMM 1 -> GEGLU -> GEMM 2 for every wave.  1: the wn = 1
  // waves run GEMM 2 -> GEMM 1 -> GEGLU instead (stream 11..15, 0..9, 10), so the two waves of a SIMD (w and w + 4) are out of step: one's GEGLU
  // VALU block lies beside the other's MFMAs instead of beside the other's GEGLU.  Any order is correct: GEMM 2 reads P(c - 1), written before
  // the barrier that opens the interval; P(c) only has to be written before the barrier that closes it.
  auto steady = [&](const int c, auto dmc, auto ordc) {
    constexpr int DM = decltype(dmc)::value;
    constexpr int ORD = decltype(ordc)::value;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own DMAs of bundle c landed, own P(c - 1) written
    pp_barrier();
    const unsigned sb = lds0 + (unsigned)(c & 1) * FF_STAGE;
    const unsigned a1 = sb + (unsigned)w1off, a2 = sb + (unsigned)w2off;
    const unsigned ap = lds0 + 2 * FF_STAGE + (unsigned)((c - 1) & 1) * FF_PBUF + (unsigned)poff;
    const unsigned ab = lds0 + 2 * FF_STAGE + 2 * FF_PBUF + (unsigned)c * 256u + (unsigned)boff * 4u;
    const unsigned apw = lds0 + 2 * FF_STAGE + (unsigned)(c & 1) * FF_PBUF + (unsigned)poff + (unsigned)wn * 8u;
    const unsigned ldst = lds0 + (unsigned)((c + 1) & 1) * FF_STAGE + (unsigned)wave * 1024u;
    const unsigned char* const s1 = W1b + (int64_t)(c + 1) * 64 * p.ld1 * (int64_t)sizeof(T);
    const unsigned char* const s2 = W2b + (int64_t)c * 64;
    auto dma_all = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      if (DM == 2) {
#pragma unroll
        for (int i = 0; i < 5; ++i) dma(i, s1, ldst);
      }
      dma(5, s2, ldst);
      dma(6, s2, ldst);
      if (has8) dma(7, s2, ldst);
      __builtin_amdgcn_sched_barrier(0);
    };
    u32x4 pf0, pf1, ring[RD][2];
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    ff_lds_read<0>(pf0, ap);
    ff_lds_read<1024>(pf1, ap);
    if constexpr (DPL == 1 && (ABL & 2) == 0) dma_all();
    auto issue = [&](auto pc) {
      constexpr int P = decltype(pc)::value, S = P % RD;
      constexpr int J = ORD == 0 ? P : (P < 5 ? 11 + P : P - 5);
      if constexpr (P >= 16) {
      } else if constexpr (J < 10) {
        ff_lds_read<J * 1024>(ring[S][0], a1);
        ff_lds_read<(FF_KS + J) * 1024>(ring[S][1], a1);
      } else if constexpr (J == 10) {
        ff_lds_read<0>(ring[S][0], ab);
        ff_lds_read<64>(ring[S][1], ab);
      } else {
        ff_lds_read<(2 * (J - 11)) * 1024>(ring[S][0], a2);
        ff_lds_read<(2 * (J - 11) + 1) * 1024>(ring[S][1], a2);
      }
    };
    issue(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 1>{});
    issue(std::integral_constant<int, 2>{});
    if constexpr (RD == 4) issue(std::integral_constant<int, 3>{});
    f32x4 hv[2], hg[2];
    hv[0] = hv[1] = hg[0] = hg[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x2 pw0, pw1;
    ff_for<0, 16>([&](auto pc) {
      constexpr int P = decltype(pc)::value, S = P % RD;
      constexpr int J = ORD == 0 ? P : (P < 5 ? 11 + P : P - 5);
      constexpr int NW = 2 * ((15 - P) < (RD - 1) ? (15 - P) : (RD - 1));     // pairs still allowed in flight behind this one
      if constexpr (P == 0) ff_lds_wait4<NW>(ring[S][0], ring[S][1], pf0, pf1);
      else ff_lds_wait<NW>(ring[S][0], ring[S][1]);
      if constexpr (J < 10) {
        Mma<T>::run(hv[0], ring[S][0], xf[0][J]);
        Mma<T>::run(hg[0], ring[S][1], xf[0][J]);
        Mma<T>::run(hv[1], ring[S][0], xf[1][J]);
        Mma<T>::run(hg[1], ring[S][1], xf[1][J]);
        issue(std::integral_constant<int, P + RD>{});
        if constexpr (J < 8 && (ABL & 2) == 0 && DPL == 0) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (J < 5) { if (DM == 2) dma(J, s1, ldst); }
          else if constexpr (J < 7) dma(J, s2, ldst);
          else { if (has8) dma(7, s2, ldst); }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if constexpr (J == 10) {
        // bias, then GEGLU; the packed outputs wait in registers for the store behind the last counted wait (a compiler-placed ds_write
        // inside the stream would be one more, uncounted, entry on lgkmcnt)
        const f32x4 bv = __builtin_bit_cast(f32x4, ring[S][0]), bg = __builtin_bit_cast(f32x4, ring[S][1]);
        f32x4 v0 = hv[0] + bv, v1 = hv[1] + bv;
        const f32x4 g0 = hg[0] + bg, g1 = hg[1] + bg;
        issue(std::integral_constant<int, P + RD>{});
        if constexpr (DPL == 2 && (ABL & 2) == 0) dma_all();
        auto geglu = [&](f32x4& v, const f32x4& gt4) -> u32x2 {
          const float gt[4] = {gt4[0], gt4[1], gt4[2], gt4[3]};
          if constexpr (ABL & 1) v += gt4;
          else if constexpr (GSC) ff_gelu_gate4_scalar(v, gt);
          else glu_gate4<T>(v, gt, 0);
          T h4[4] = {from_f<T>(v[0]), from_f<T>(v[1]), from_f<T>(v[2]), from_f<T>(v[3])};
          u32x2 w;
          __builtin_memcpy(&w, h4, 8);
          return w;
        };
        pw0 = geglu(v0, g0);
        pw1 = geglu(v1, g1);
      } else {
        constexpr int t0 = 2 * (J - 11);
        if constexpr (ABL & 4) {
          oacc[t0][0][0] += __builtin_bit_cast(f32x4, ring[S][0])[0] + __builtin_bit_cast(f32x4, ring[S][1])[0];
        } else {
          Mma<T>::run(oacc[t0][0], ring[S][0], pf0);
          Mma<T>::run(oacc[t0][1], ring[S][0], pf1);
          Mma<T>::run(oacc[t0 + 1][0], ring[S][1], pf0);
          Mma<T>::run(oacc[t0 + 1][1], ring[S][1], pf1);
        }
        issue(std::integral_constant<int, P + RD>{});
      }
    });
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:1024" ::"v"(apw), "v"(pw0), "v"(pw1) : "memory");
  };

  iter(0, std::true_type{}, std::false_type{});
  if constexpr (PLAIN) {
    for (int c = 1; c < FF_NCH; ++c) iter(c, std::true_type{}, std::true_type{});
  } else {
    if (ORDER == 1 && wn == 1) {   // (wave-uniform)
      for (int c = 1; c < FF_NCH - 1; ++c) steady(c, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
      steady(FF_NCH - 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    } else {
      for (int c = 1; c < FF_NCH - 1; ++c) steady(c, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
      steady(FF_NCH - 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    }
  }
  iter(FF_NCH, std::false_type{}, std::true_type{});

  // ---- epilogue: + bias + residual (the raw rows), one rounding ----
  if constexpr (EST) {
    // per-wave fp32 staging (16 rows x 160 columns per pass, two passes) in the operand stages, then 16-byte pieces: 5 residual loads and 5
    // stores per lane and pass instead of 10 + 10 eight-byte ones in the accumulator layout (the store tail of a row-per-lane epilogue is
    // issue-bound, MI355X_MICROARCH.md).  Same arithmetic per element: (acc + bias) + residual -> one rounding.
    constexpr int PITCH = 160 * 4 + 16;
    __syncthreads();                                   // every wave is past its last fragment read
    unsigned char* const stg = dsm + wave * (16 * PITCH);
    const T* X = (const T*)p.x;
    T* O = (T*)p.out;
    int prow[5], pcol[5];
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int idx = lane + it * 64;
      prow[it] = idx / 20;
      pcol[it] = (idx - prow[it] * 20) * 8;
    }
    u32x4 rv[2][5];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int it = 0; it < 5; ++it)
        rv[rt][it] = *(const u32x4*)(X + (int64_t)(m0 + rg * 32 + rt * 16 + prow[it]) * p.ldx + wn * 160 + pcol[it]);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
      for (int t = 0; t < 10; ++t) {
        const f32x4 b = *(const f32x4*)(p.b2 + wn * 160 + t * 16 + g * 4);
        *(f32x4*)(stg + l15 * PITCH + (t * 16 + g * 4) * 4) = oacc[t][rt] + b;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        const f32x4 lo = *(const f32x4*)(stg + prow[it] * PITCH + pcol[it] * 4);
        const f32x4 hi = *(const f32x4*)(stg + prow[it] * PITCH + pcol[it] * 4 + 16);
        T r8[8], o8[8];
        __builtin_memcpy(r8, &rv[rt][it], 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) { o8[j] = from_f<T>(lo[j] + to_f(r8[j])); o8[4 + j] = from_f<T>(hi[j] + to_f(r8[4 + j])); }
        u32x4 ow;
        __builtin_memcpy(&ow, o8, 16);
        *(u32x4*)(O + (int64_t)(m0 + rg * 32 + rt * 16 + prow[it]) * p.ldo + wn * 160 + pcol[it]) = ow;
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    const T* X = (const T*)p.x;
    T* O = (T*)p.out;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int64_t row = m0 + rg * 32 + rt * 16 + l15;
#pragma unroll
      for (int t = 0; t < 10; ++t) {
        const int col = wn * 160 + t * 16 + g * 4;
        const f32x4 b = *(const f32x4*)(p.b2 + col);
        const u32x2 rw = *(const u32x2*)(X + row * p.ldx + col);
        T r4[4];
        __builtin_memcpy(r4, &rw, 8);
        const f32x4 a = oacc[t][rt];
        T o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = from_f<T>((a[r] + b[r]) + to_f(r4[r]));
        u32x2 ow;
        __builtin_memcpy(&ow, o4, 8);
        *(u32x2*)(O + row * p.ldo + col) = ow;
      }
    }
  }
}

// ================================================================================================================================
// Activation-stationary LayerNorm + QKV projection of the level-0 self-attention (round 6): [q | k] row-major and v TRANSPOSED
// ([B][C][S], the layout attention.hip stages from) out of ONE pass over x -- Attention.to_q / to_k / to_v on norm1(x),
// mustango/diffusers/src/diffusers/models/attention.py:276-296, attention_processor.py:495-520.  K = C = 320 is ten k-steps: on the
// 256 x 320 tile kernel the epilogue (fp32 staging, LDS transpose of the V columns) costs as much as the multiply, 573 TFLOP/s at config 3.
// Here a 512-thread workgroup owns 256 rows, each wave keeps its 32 NORMALISED rows as MFMA operand fragments in registers for the
// whole kernel (as ff_fused_kernel), and the [N][320] weights stream through LDS in chunks of 64 output columns (40 KB, three stages,
// LDS-DMA; fragment stream in inline asm, ring of three k-steps).  Per chunk and wave: 80 MFMAs into 32 accumulator registers that START
// at the folded bias, then 8 eight-byte stores straight from the accumulator layout -- inside the loop, under the next chunk's MFMAs:
//   * q / k columns: H^T = W' x^T (A = weights, B = x): a lane holds 4 consecutive columns of one row;
//   * v columns: the SAME two fragments with the operands exchanged, V = x W'^T (A = x, B = weights: the 16 x 16 x 32 operand layouts are
//     mirror images), so a lane holds 4 consecutive TOKENS of one channel -- the transpose costs nothing.
// vmcnt retires in order and counts the stores: bundle c (issued two iterations before it is read) is followed by the 8 stores of each
// of the two chunks in between and the 5 DMAs of bundle c + 1, so `vmcnt(13)` -- the newest 13 may stay in flight -- covers it with room.
// ================================================================================================================================
namespace {
constexpr int QS_BM = 256, QS_CH = 64, QS_STAGE = 40 * 1024, QS_NST = 3;
constexpr int QS_MAXN = 1024;
constexpr int QS_WSTG = 64 * 64;                 // per-wave staging slice: 16 rows x (128 B + 16) for row-major chunks, 64 channels x (32 B + 16) for transposed ones
constexpr int QS_LDS = QS_NST * QS_STAGE + QS_MAXN * 4 + 8 * QS_WSTG;
}  // namespace

template <int OFF> __device__ __forceinline__ void ff_lds_read32(float& v, const unsigned addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void qs_wait_frags(u32x4 (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N));
}

// QV bits: 0 = stores through a per-wave LDS staging slice in 16-byte pieces (0: 8-byte stores straight from the accumulator layout),
// 1 = timing probe without stores (TANGO_QKV_VAR)
template <typename T, int QV>
__global__ __launch_bounds__(512) void qkv_stat_kernel(const QKVParams p) {
  constexpr bool STG = (QV & 1) != 0, NOST = (QV & 2) != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  float* const bs = (float*)(dsm + QS_NST * QS_STAGE);
  unsigned char* const stg = dsm + QS_NST * QS_STAGE + QS_MAXN * 4 + (threadIdx.x >> 6) * QS_WSTG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * QS_BM;
  const int nch = p.N / QS_CH, nch_rm = p.n_rm / QS_CH;

  // LDS-DMA: 40 groups of 16 weight rows x 64 B per chunk (group = tile * 10 + k-step), wave w issues groups w + 8 i
  const int lrow = lane >> 2;
  const int pc = (lane & 3) ^ ((4 - (lrow >> 2)) & 3);
  const unsigned v_off = (unsigned)((int64_t)lrow * p.ldw * (int64_t)sizeof(T)) + pc * 16;
  unsigned s_off[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int grp = wave + 8 * i, tile = grp / FF_KS, ks = grp - tile * FF_KS;
    s_off[i] = sgpr_u32((unsigned)((int64_t)(tile * 16) * p.ldw * (int64_t)sizeof(T)) + ks * 64);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
  const unsigned char* const Wb = (const unsigned char*)p.w;
  auto dma = [&](const int i, const unsigned char* src, const unsigned ldst) {
    unsigned o = v_off;
    asm volatile("" : "+v"(o));
    __builtin_amdgcn_global_load_lds((gptr_t)(src + s_off[i] + o), (lptr_t)(uintptr_t)(ldst + (unsigned)i * 8192u), 16, 0, 0);
  };
  auto bundle_src = [&](const int cb) { return Wb + (int64_t)cb * QS_CH * p.ldw * (int64_t)sizeof(T); };
  auto bundle_dst = [&](const int cb) { return lds0 + (unsigned)(cb % QS_NST) * QS_STAGE + (unsigned)wave * 1024u; };
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
    if (cb < nch) {
#pragma unroll
      for (int i = 0; i < 5; ++i) dma(i, bundle_src(cb), bundle_dst(cb));
    }
  for (int i = tid; i < p.N; i += 512) bs[i] = p.b ? p.b[i] : 0.f;

  // ---- this wave's 32 rows: load, LayerNorm (two-pass fp32), keep as operand fragments ----
  u32x4 xf[2][FF_KS];
  {
    const T* X = (const T*)p.x;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const T* xr = X + (int64_t)(m0 + wave * 32 + rt * 16 + l15) * p.ldx + g * 8;
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) xf[rt][ks] = *(const u32x4*)(xr + ks * 32);
    }
    if (p.ln == 2) {
      // GroupNorm with given statistics: per-channel {mean, rstd} of this workgroup's sample through the (not yet used) last operand stage
      f32x2* const tab = (f32x2*)(dsm + (QS_NST - 1) * QS_STAGE);
      const int smp = m0 / p.rows_ps, cg = FF_C / p.groups;
      if (tid < FF_C) tab[tid] = *(const f32x2*)(p.gstats + ((int64_t)smp * p.groups + tid / cg) * 2);
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) {
        f32x2 mr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) mr[j] = tab[ks * 32 + g * 8 + j];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          T e[8];
          __builtin_memcpy(e, &xf[rt][ks], 16);
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = from_f<T>((to_f(e[j]) - mr[j].x) * mr[j].y);
          __builtin_memcpy(&xf[rt][ks], e, 16);
        }
      }
      __syncthreads();                                  // the table's stage may be DMA'd from the first chunk on
    } else {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) {
        T e[8];
        __builtin_memcpy(e, &xf[rt][ks], 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += to_f(e[j]);
      }
      s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
      const float mean = s * (1.0f / FF_C);
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) {
        T e[8];
        __builtin_memcpy(e, &xf[rt][ks], 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = to_f(e[j]) - mean; q += d * d; }
      }
      q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
      const float rstd = p.ln ? 1.0f / sqrtf(q * (1.0f / FF_C) + p.eps) : 1.0f;
      const float mu = p.ln ? mean : 0.f;
#pragma unroll
      for (int ks = 0; ks < FF_KS; ++ks) {
        T e[8];
        __builtin_memcpy(e, &xf[rt][ks], 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = from_f<T>((to_f(e[j]) - mu) * rstd);
        __builtin_memcpy(&xf[rt][ks], e, 16);
      }
    }
    }
  }
  const int foff = l15 * 64 + ((g ^ ((4 - (l15 >> 2)) & 3)) * 16);
  T* const O = (T*)p.out;
  const int bb = m0 / p.vt_S, s0 = m0 - bb * p.vt_S;
  T* const Vt = (T*)p.vt + (int64_t)bb * (p.N - p.n_rm) * p.vt_ld + s0 + wave * 32 + g * 4;

  // one chunk: 64 output columns.  VT: the transposed range.
  auto chunk = [&](const int c, auto vtc) {
    constexpr bool VT = decltype(vtc)::value;
    if constexpr (NOST) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    else if constexpr (STG) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");     // 4 stores per chunk in the staged form
    else asm volatile("s_waitcnt vmcnt(13) lgkmcnt(0)" ::: "memory");
    pp_barrier();
    const unsigned a = lds0 + (unsigned)(c % QS_NST) * QS_STAGE + (unsigned)foff;
    const unsigned ab = lds0 + QS_NST * QS_STAGE + (unsigned)(c * QS_CH + (VT ? l15 : g * 4)) * 4u;
    const bool more = c + 2 < nch;
    const unsigned char* const src = bundle_src(c + 2);
    const unsigned ldst = bundle_dst(c + 2);
    f32x4 acc[4][2];
    u32x4 ring[3][4];
    // the bias first: row-major chunks want the quad of this lane's 4 columns per tile, transposed chunks the lane's channel per tile
    u32x4 bq[4];
    float bsc[4];
    if constexpr (VT) {
      ff_lds_read32<0>(bsc[0], ab); ff_lds_read32<64>(bsc[1], ab); ff_lds_read32<128>(bsc[2], ab); ff_lds_read32<192>(bsc[3], ab);
    } else {
      ff_lds_read<0>(bq[0], ab); ff_lds_read<64>(bq[1], ab); ff_lds_read<128>(bq[2], ab); ff_lds_read<192>(bq[3], ab);
    }
    auto issue = [&](auto kc) {
      constexpr int K = decltype(kc)::value, S = K % 3;
      if constexpr (K < FF_KS) {
        ff_lds_read<(0 * FF_KS + K) * 1024>(ring[S][0], a);
        ff_lds_read<(1 * FF_KS + K) * 1024>(ring[S][1], a);
        ff_lds_read<(2 * FF_KS + K) * 1024>(ring[S][2], a);
        ff_lds_read<(3 * FF_KS + K) * 1024>(ring[S][3], a);
      }
    };
    issue(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 1>{});
    issue(std::integral_constant<int, 2>{});
    // bias landed (12 fragment reads may stay in flight)
    if constexpr (VT) {
      asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(bsc[0]), "+v"(bsc[1]), "+v"(bsc[2]), "+v"(bsc[3]));
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][0] = acc[t][1] = f32x4{bsc[t], bsc[t], bsc[t], bsc[t]};
    } else {
      qs_wait_frags<12>(bq);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][0] = acc[t][1] = __builtin_bit_cast(f32x4, bq[t]);
    }
    ff_for<0, FF_KS>([&](auto kc) {
      constexpr int K = decltype(kc)::value, S = K % 3;
      constexpr int NW = 4 * ((FF_KS - 1 - K) < 2 ? (FF_KS - 1 - K) : 2);
      qs_wait_frags<NW>(ring[S]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (VT) {
          Mma<T>::run(acc[t][0], xf[0][K], ring[S][t]);
          Mma<T>::run(acc[t][1], xf[1][K], ring[S][t]);
        } else {
          Mma<T>::run(acc[t][0], ring[S][t], xf[0][K]);
          Mma<T>::run(acc[t][1], ring[S][t], xf[1][K]);
        }
      }
      issue(std::integral_constant<int, K + 3>{});
      if constexpr (K < 5) {
        __builtin_amdgcn_sched_barrier(0);
        if (more) dma(K, src, ldst);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    if constexpr (NOST) {
      float sink = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) sink += acc[t][0][0] + acc[t][1][3];
      if (sink == 12345.678f) O[0] = from_f<T>(sink);
    } else if (STG && VT && p.vt_perm) {
      // transposed chunk in the attention kernel's fragment order (AttnParams::vt_perm): inside the wave's block of 32 tokens, position 8 g + 4 rt + r
      // holds token 16 rt + 4 g + r.  Both passes through the staging slice ([channel 64][64 B], 16-byte pieces XOR-swizzled by (channel >> 2) & 3:
      // conflict-free for the 8-byte writes and the 16-byte reads), then 4 sixteen-byte stores per lane: 64-byte runs per channel
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 v = acc[t][rt];
          T o4[4] = {from_f<T>(v[0]), from_f<T>(v[1]), from_f<T>(v[2]), from_f<T>(v[3])};
          u32x2 ow;
          __builtin_memcpy(&ow, o4, 8);
          const int ch = t * 16 + l15;
          *(u32x2*)(stg + ch * 64 + ((g ^ ((ch >> 2) & 3)) * 16) + rt * 8) = ow;
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = lane + it * 64, ch = idx >> 2, pi = idx & 3;
        const u32x4 o = *(const u32x4*)(stg + ch * 64 + ((pi ^ ((ch >> 2) & 3)) * 16));
        *(u32x4*)((T*)p.vt + ((int64_t)bb * (p.N - p.n_rm) + (c * QS_CH - p.n_rm + ch)) * p.vt_ld + s0 + wave * 32 + pi * 8) = o;
      }
      __builtin_amdgcn_wave_barrier();
    } else if constexpr (STG) {
      // two passes of 16 x-rows: the wave's 16 x 64 block (row-major chunks) or 64 x 16 block (transposed chunks: 64 channels x 16 tokens)
      // through its staging slice, then 16-byte pieces -- 2 stores per lane and pass, whole 128-byte rows (32-byte token runs for v^T)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 v = acc[t][rt];
          T o4[4] = {from_f<T>(v[0]), from_f<T>(v[1]), from_f<T>(v[2]), from_f<T>(v[3])};
          u32x2 ow;
          __builtin_memcpy(&ow, o4, 8);
          if constexpr (VT) *(u32x2*)(stg + (t * 16 + l15) * 48 + g * 8) = ow;        // [channel 64][16 tokens = 32 B, pitch 48]
          else *(u32x2*)(stg + l15 * 144 + (t * 16 + g * 4) * 2) = ow;                 // [row 16][64 columns = 128 B, pitch 144]
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int idx = lane + it * 64;
          if constexpr (VT) {
            const int ch = idx >> 1, pcs = idx & 1;                                     // 64 channels x 2 pieces of 8 tokens
            const u32x4 o = *(const u32x4*)(stg + ch * 48 + pcs * 16);
            *(u32x4*)((T*)p.vt + ((int64_t)bb * (p.N - p.n_rm) + (c * QS_CH - p.n_rm + ch)) * p.vt_ld + s0 + wave * 32 + rt * 16 + pcs * 8) = o;
          } else {
            const int row = idx >> 3, pcs = idx & 7;                                    // 16 rows x 8 pieces of 8 columns
            const u32x4 o = *(const u32x4*)(stg + row * 144 + pcs * 16);
            *(u32x4*)(O + (int64_t)(m0 + wave * 32 + rt * 16 + row) * p.ldo + c * QS_CH + pcs * 8) = o;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
    // 8 eight-byte stores
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const f32x4 v = acc[t][rt];
        T o4[4] = {from_f<T>(v[0]), from_f<T>(v[1]), from_f<T>(v[2]), from_f<T>(v[3])};
        u32x2 ow;
        __builtin_memcpy(&ow, o4, 8);
        if constexpr (VT) *(u32x2*)(Vt + (int64_t)(c * QS_CH - p.n_rm + t * 16 + l15) * p.vt_ld + rt * 16) = ow;
        else *(u32x2*)(O + (int64_t)(m0 + wave * 32 + rt * 16 + l15) * p.ldo + c * QS_CH + t * 16 + g * 4) = ow;
      }
    }
    // a chunk without DMAs still has to present 13 younger operations to the next chunk's vmcnt(13): it does not (8 stores only), which
    // makes that wait STRICTER (it then also covers older stores), never laxer
  };
  // (the first chunk's wait: everything of the prologue -- both bundles and the x rows -- has landed; vmcnt(13) would not cover bundle 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int c = 0; c < nch_rm; ++c) chunk(c, std::false_type{});
  for (int c = nch_rm; c < nch; ++c) chunk(c, std::true_type{});
}

bool qkv_stat_ok(int dtype, const QKVParams& p) {
  if (dtype != DT_F16 && dtype != DT_BF16) return false;
  if (p.K != FF_C || p.M <= 0 || p.M % QS_BM != 0 || p.N <= 0 || p.N % QS_CH != 0 || p.N > QS_MAXN || p.n_rm < 0 || p.n_rm > p.N || p.n_rm % QS_CH != 0) return false;
  if (p.ldw != FF_C || p.ldx % 8 != 0 || p.ldo % 4 != 0 || ((uintptr_t)p.x & 15) || ((uintptr_t)p.w & 15) || ((uintptr_t)p.out & 7)) return false;
  if (p.ln == 2 && (!p.gstats || p.groups <= 0 || FF_C % p.groups != 0 || p.rows_ps % QS_BM != 0 || p.M % p.rows_ps != 0 || p.N / QS_CH < QS_NST)) return false;
  if (p.n_rm < p.N && (!p.vt || p.vt_S % QS_BM != 0 || p.vt_ld % 4 != 0 || ((uintptr_t)p.vt & 7) || p.M % p.vt_S != 0)) return false;
  return true;
}

template <typename T, int QV> static int qkv_stat_go(const QKVParams& p, hipStream_t s) {
  TANGO_TRY(ensure_dyn_lds((const void*)qkv_stat_kernel<T, QV>, QS_LDS));
  hipLaunchKernelGGL((qkv_stat_kernel<T, QV>), dim3((unsigned)(p.M / QS_BM)), dim3(512), QS_LDS, s, p);
  TANGO_HIP(hipGetLastError());
  return 0;
}
template <typename T> static int qkv_stat_t(const QKVParams& p, hipStream_t s) {
  if constexpr (__is_same(T, f16)) {
    static const int var = getenv("TANGO_QKV_VAR") ? atoi(getenv("TANGO_QKV_VAR")) : -1;
    if (var == 0) return qkv_stat_go<f16, 0>(p, s);
    if (var == 2) return qkv_stat_go<f16, 2>(p, s);
  }
  return qkv_stat_go<T, 1>(p, s);
}

int launch_qkv_stat(int dtype, const QKVParams& p, hipStream_t s) {
  if (!qkv_stat_ok(dtype, p)) TANGO_FAIL("qkv_stat: unsupported shape (K = 320, M % 256 == 0, N % 64 == 0, 16-bit)");
  if (dtype == DT_F16) return qkv_stat_t<f16>(p, s);
  return qkv_stat_t<bf16>(p, s);
}

bool ff_fused_ok(int dtype, const FFParams& p) {
  if (dtype != DT_F16 && dtype != DT_BF16) return false;
  if (p.C != FF_C || p.H != FF_H || p.M <= 0 || p.M % FF_BM != 0) return false;
  if (p.ld1 != FF_C || p.ld2 != FF_H || p.ldx % 8 != 0 || p.ldo % 4 != 0) return false;
  if (((uintptr_t)p.x | (uintptr_t)p.w1 | (uintptr_t)p.w2 | (uintptr_t)p.b1 | (uintptr_t)p.b2) & 15) return false;
  if ((uintptr_t)p.out & 7) return false;
  return true;
}

template <typename T, int VAR> static int ff_fused_go(const FFParams& p, hipStream_t s) {
  TANGO_TRY(ensure_dyn_lds((const void*)ff_fused_kernel<T, VAR>, FF_LDS));
  hipLaunchKernelGGL((ff_fused_kernel<T, VAR>), dim3((unsigned)(p.M / FF_BM)), dim3(512), FF_LDS, s, p);
  TANGO_HIP(hipGetLastError());
  return 0;
}

template <typename T> static int ff_fused_t(const FFParams& p, hipStream_t s) {
  if constexpr (__is_same(T, f16)) {
    // A/B arms and timing probes (tools/ff_fused_ablation.py): fp16 only
    static const int var = getenv("TANGO_FF_VAR") ? (int)strtol(getenv("TANGO_FF_VAR"), nullptr, 0) : -1;
    switch (var) {
      case 0x000: return ff_fused_go<f16, 0x000>(p, s);
      case 0x010: return ff_fused_go<f16, 0x010>(p, s);
      case 0x100: return ff_fused_go<f16, 0x100>(p, s);
      case 0x320: return ff_fused_go<f16, 0x320>(p, s);
      case 0x520: return ff_fused_go<f16, 0x520>(p, s);
      case 0x720: return ff_fused_go<f16, 0x720>(p, s);
      case 0x140: return ff_fused_go<f16, 0x140>(p, s);
      case 0x180: return ff_fused_go<f16, 0x180>(p, s);
      case 0x1a0: return ff_fused_go<f16, 0x1a0>(p, s);
      case 0x122: return ff_fused_go<f16, 0x122>(p, s);
      case 0x124: return ff_fused_go<f16, 0x124>(p, s);
      case 0x126: return ff_fused_go<f16, 0x126>(p, s);
      case 0x128: return ff_fused_go<f16, 0x128>(p, s);
      case 0x12e: return ff_fused_go<f16, 0x12e>(p, s);
      default: break;
    }
  }
  if (p.plain_loop || tuning().ff_fused == 2) return ff_fused_go<T, 1>(p, s);
  return ff_fused_go<T, FF_VAR_DEFAULT>(p, s);
}

int launch_ff_fused(int dtype, const FFParams& p, hipStream_t s) {
  if (!ff_fused_ok(dtype, p)) TANGO_FAIL("ff_fused: unsupported shape (C = 320, H = 1280, M % 128 == 0, 16-bit)");
  if (dtype == DT_F16) return ff_fused_t<f16>(p, s);
  return ff_fused_t<bf16>(p, s);
}

}  // namespace tango
