// 256 x 160 LDS-DMA GEMM, four waves, TWO workgroups per CU (round 4) -- for the 16-bit linears whose grids leave the 8-wave
// kernels short of tiles.
//
// The idea it was built for: the 256 x 320 kernel (gemm_wide.hip) owns the whole CU (144 KiB of LDS), so every tile's prologue
// (first chunk latency, 3-4 us) and epilogue (5-20 us of bias / gate / residual / stores) run with the matrix pipe idle; two
// INDEPENDENT workgroups per CU could hide each other's pro- and epilogues.  What the GPU said (profiles/r4_c7_duo_*.txt,
// r4_c8_duo_*.txt): on the big B = 32 grids it only EQUALS the 8-wave kernel (K <= 640) or loses 3-15 % (K >= 1280) -- the
// partners start together, do the same work at the same speed and stay in step, and forcing a phase offset did not help
// either; its main loop has no explicit ping-pong and moves 45 % more DMA bytes per flop.  What it is good for: twice as many,
// half as large tiles.  At B = 8, on the single-key halves at B = 32 and at level 3 the 8-wave kernels have < 2 tiles per CU (or
// fall back to 4-wave tiles with split-K), and this kernel wins 10-35 % there (B = 8 step 19.7 -> 18.7 ms).  gemm_duo_ok() holds
// the measured routing rule.
//   * tile 256 x 160, waves 4 (M) x 1 (N), wave tile 64 x 160 = the 256 x 320 kernel's, so its epilogues are used unchanged and
//     the results are bit-identical to that kernel's (tests/test_duo_gpu.py);
//   * 64-byte k-chunks, THREE stages of (256 + 160) rows x 64 B = 78 KiB per workgroup (two fit the 160-KiB LDS);
//   * ONE barrier per chunk: [ds_read 14 fragments of chunk kc | 40 MFMAs with the DMAs of chunk kc+2 between them | wait for the
//     own DMAs of chunk kc+1 | barrier].  Chunk kc+2 refills the stage of chunk kc-1, whose reads every wave finished before the
//     previous barrier; a chunk has two iterations to land;
//   * LDS image, source-side swizzle and fragment addressing are gemm_wide.hip's.
// Linear problems only; M % 256 == 0, N % 160 == 0, K a multiple of 32 elements.
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "tuning.h"
#include "gemm_device.h"
#include "gemm_wide_device.h"

namespace tango {

template <typename T> __device__ __forceinline__ void duo_frag_stats(const u32x4& v, float& s, float& q) {
  if constexpr (__is_same(T, f16)) {
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

constexpr int DUO_BM = 256, DUO_BN = 160, DUO_CB = 64, DUO_NST = 3;
constexpr int DUO_LDS = DUO_NST * (DUO_BM + DUO_BN) * DUO_CB;      // 79872 B: two workgroups per CU

template <typename T, bool GEGLU, bool RES, bool LN, bool VT>
__global__ __launch_bounds__(256, 2) void gemm_duo_kernel(const GemmParams p, const int prio) {
  constexpr int BM = DUO_BM, BN = DUO_BN, CB = DUO_CB, NST = DUO_NST;
  constexpr int ROWS = BM + BN, STAGE = ROWS * CB;
  constexpr int RG = ROWS / 16;                  // 16-row groups (1 KiB) per stage: 26
  constexpr int RGW = (RG + 3) / 4;              // DMA instructions per wave per chunk: 7 (waves 0-1) or 6 (waves 2-3)
  constexpr int TM = 4, TN = 10;                 // wave tile 64 x 160
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];   // NST stages

  const int NT = p.N / BN;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // XCD x works through a contiguous range of tiles
  }
  const int m0 = (bid / NT) * BM, n0 = (bid % NT) * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- DMA source rows: row group rg = wave + 4 i; i < 4 are activation rows (rg < 16), i >= 4 weight rows, for every wave ----
  const int lrow = lane >> 2;
  const int pc = (lane & 3) ^ ((4 - (lrow >> 2)) & 3);
  const unsigned char* const At = (const unsigned char*)p.A + (int64_t)m0 * p.lda * (int64_t)sizeof(T);
  const unsigned char* const Wt = (const unsigned char*)p.W + ((int64_t)n0 * p.Kp + (p.wb_rows ? (int64_t)(m0 / p.wb_rows) * p.wb_stride : 0)) * (int64_t)sizeof(T);
  unsigned r_off[RGW];
#pragma unroll
  for (int i = 0; i < RGW; ++i) {
    const int row = (wave + 4 * i) * 16 + lrow;
    r_off[i] = i < 4 ? (unsigned)((int64_t)row * p.lda * (int64_t)sizeof(T)) + pc * 16
                     : (unsigned)((int64_t)(row - BM) * p.Kp * (int64_t)sizeof(T)) + pc * 16;
  }
  const int nk = (p.K * (int)sizeof(T)) / CB;
  const bool has7 = wave < RG - 4 * (RGW - 1);   // wave-uniform: waves 0, 1
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
  auto dma = [&](const int i, const unsigned char* sa, const unsigned char* sw, const unsigned ldst) {
    unsigned o = r_off[i];
    asm volatile("" : "+v"(o));     // zero-extension next to the use: hipcc then picks the [SGPR base + 32-bit VGPR offset] form
    __builtin_amdgcn_global_load_lds((gptr_t)((i < 4 ? sa : sw) + o), (lptr_t)(uintptr_t)(ldst + (unsigned)i * 4096u), 16, 0, 0);
  };
  auto issue_chunk = [&](const int kc, const int st) {
    const unsigned char* sa = At + (int64_t)kc * CB;
    const unsigned char* sw = Wt + (int64_t)kc * CB;
#pragma unroll
    for (int i = 0; i < RGW; ++i)
      if (i < RGW - 1 || has7) dma(i, sa, sw, lds0 + st * STAGE + (unsigned)wave * 1024u);
  };
  // at most one whole chunk of this wave's DMAs may stay in flight (none: drain)
  auto wait_all_but_one_chunk = [&](const bool one) {
    if (!one) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    if (has7) wait_vmcnt_lit<RGW>();
    else wait_vmcnt_lit<RGW - 1>();
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ssum[TM] = {0.f, 0.f, 0.f, 0.f}, ssq[TM] = {0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, g = lane >> 4;
  const int foff = l15 * CB + ((g ^ ((4 - (l15 >> 2)) & 3)) * 16);
  const int xrow = (wave * TM * 16) * CB + foff;
  const int wrow = BM * CB + foff;

  const int npro = nk < NST - 1 ? nk : NST - 1;
  for (int c = 0; c < npro; ++c) issue_chunk(c, c);
  wait_all_but_one_chunk(npro > 1);                     // chunk 0 landed
  pp_barrier();
  int st = 0;
  for (int kc = 0; kc < nk; ++kc) {
    const unsigned char* Xs = dsm + st * STAGE;
    const int st2 = st == 0 ? NST - 1 : st - 1;        // chunk kc+2 refills the stage of chunk kc-1
    u32x4 wf[TN], xf[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) xf[b] = *(const u32x4*)(Xs + xrow + b * 16 * CB);
#pragma unroll
    for (int a = 0; a < TN; ++a) wf[a] = *(const u32x4*)(Xs + wrow + a * 16 * CB);
    const bool more = kc + NST - 1 < nk;               // chunk kc+2 exists
    const bool more7 = more && has7;
    const unsigned char* sa = At + (int64_t)(kc + NST - 1) * CB;
    const unsigned char* sw = Wt + (int64_t)(kc + NST - 1) * CB;
    const unsigned ldst = lds0 + st2 * STAGE + (unsigned)wave * 1024u;
    if (prio == 0) __builtin_amdgcn_s_setprio(1);
    if (LN) {
#pragma unroll
      for (int b = 0; b < TM; ++b) duo_frag_stats<T>(xf[b], ssum[b], ssq[b]);
    }
#pragma unroll
    for (int a = 0; a < TN; ++a) {
#pragma unroll
      for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
      if (a < RGW) {
        // after MFMAs 4, 8, ..., 28: the a-th DMA of chunk kc+2
        __builtin_amdgcn_sched_barrier(0);
        if (a < RGW - 1 ? more : more7) dma(a, sa, sw, ldst);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (prio == 0) __builtin_amdgcn_s_setprio(0);
    // this wave's DMAs of chunk kc+1 must have landed before the barrier that precedes anyone's read of that chunk;
    // chunk kc+2 (if issued) may stay in flight
    if (kc + 1 < nk) wait_all_but_one_chunk(more);
    pp_barrier();
    st = st == NST - 1 ? 0 : st + 1;
  }
  // every wave is past its last fragment read (the loop's last barrier): the operand stages become the staging area
  float mean[TM] = {0.f, 0.f, 0.f, 0.f}, rstd[TM] = {1.f, 1.f, 1.f, 1.f};
  if (LN) {
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      float sm = ssum[b], sq = ssq[b];
      sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
      sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
      const float mu = sm / (float)p.K;
      float var = sq / (float)p.K - mu * mu;
      var = var < 0.f ? 0.f : var;
      mean[b] = mu; rstd[b] = rsqrtf(var + p.ln_eps);
    }
  }
  unsigned char* const slice = dsm + wave * (WIDE_STAGE_BYTES + 1280);
  if (VT && n0 >= p.vt_n0) wide_epilogue_vt<T, LN>(p, acc, mean, rstd, m0 + wave * TM * 16, n0, lane, slice);
  else wide_epilogue<T, GEGLU, RES, LN>(p, acc, mean, rstd, m0 + wave * TM * 16, n0, lane, slice);
}

// which problems: what gemm_wide.hip takes (16-bit linear, plain / GEGLU / transposed-V epilogue into T) in whole 256 x 160 tiles.
// Measured (profiles/r4_c7_duo_*): with >= 512 tiles of 256 x 320 (two or more per CU: every big B = 32 shape) this kernel equals the
// 8-wave kernel within noise up to K = 640 and loses 3-15 % beyond (its main loop has no explicit ping-pong and moves 45 % more
// DMA bytes per flop; the two co-resident workgroups start together and stay in step, so their epilogues do NOT hide behind each
// other's main loops; delaying one of the two first workgroups of every CU by 0.5 / 1 / 2 x the main-loop time -- picked by dispatch
// order or by its hardware wave slot -- changed nothing or cost 2-8 %: profiles/r4_c8_duo_stagger_ab_b32.txt).  It wins where the 8-wave kernels run short of tiles -- B = 8, the single-key halves at B = 32, level 3:
//   M=16384 N=640 K=640 x15 0.546 -> 0.41 ms, M=65536 N=960 K=320 (LN) x5 0.439 -> 0.354, M=8192 N=1280 K=1280 x10 0.49 -> 0.41,
//   M=4096 N=1280 K=1280 x15 (split-K before) 0.638 -> 0.546; but M=4096 N=1280 K=5120 0.435 -> 0.552, M=65536 N=320 K=1280 0.340 -> 0.391.
// TANGO_DUO_MAXK >= 0 replaces the rule by "K <= that, >= TANGO_DUO_MIN_TILES tiles" (A/B runs, tests).
bool gemm_duo_ok(int dtype, const GemmParams& p) {
  if (tuning().duo_maxk == 0 || dtype == DT_F32) return false;
  const bool linear = p.mode == GATHER_1D && p.taps == 1 && p.rows_pb == p.M && p.in_mul == 1 && p.in_off == 0 && p.out_mul == 1 &&
                      p.out_off == 0 && p.Lin >= p.M;
  if (!linear || p.batch != 1 || p.a_act != ACT_NONE || p.bias_rows) return false;
  if (p.splitk > 1 || p.out_f32 || p.e_act != ACT_NONE || p.row_stats) return false;
  if (p.ln_fold && (!p.wsum || ((uintptr_t)p.wsum & 15) || p.alpha != 1.f)) return false;
  if (p.epi != EPI_NONE && p.epi != EPI_GEGLU && p.epi != EPI_VT) return false;
  if (p.epi == EPI_VT && (p.R || p.bias2 || p.vt_n0 % 160 != 0 || p.vt_S % 256 != 0 || p.vt_ld % 8 != 0 || ((uintptr_t)p.vt & 15))) return false;
  if (p.M % 256 != 0 || p.N % 160 != 0 || (p.K * 2) % 64 != 0) return false;
  if (p.ldo % 8 != 0 || ((uintptr_t)p.out & 15) || (p.R && (p.ldr % 8 != 0 || ((uintptr_t)p.R & 15)))) return false;
  if ((p.lda * 2) % 16 != 0 || (p.Kp * 2) % 16 != 0 || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return false;
  if (((uintptr_t)p.bias & 15) || ((uintptr_t)p.bias2 & 15) || (p.bias2 && p.bias2_stride % 4 != 0)) return false;
  // classes: 1 plain, 2 GEGLU, 4 folded LayerNorm (incl. transposed V), 8 folded LayerNorm + GEGLU (32 VALU instructions of row
  // statistics per chunk next to 40 MFMAs: 3.55 vs 3.38 ms on the level-0 projection against the streaming kernel -- off by default)
  const int cls = p.ln_fold ? (p.epi == EPI_GEGLU ? 8 : 4) : (p.epi == EPI_GEGLU ? 2 : (p.epi == EPI_VT ? 4 : 1));
  if (!(tuning().duo_mask & cls)) return false;
  const long tiles = (long)(p.M / 256) * (p.N / 160);
  if (tuning().duo_maxk > 0) return p.K <= tuning().duo_maxk && (tuning().force_big_kernels || tiles >= tuning().duo_min_tiles);
  // the measured rule
  if (tiles < 128) return false;
  const long wide_tiles = p.N % 320 == 0 ? (long)(p.M / 256) * (p.N / 320) : 0;
  if (wide_tiles >= 512) return false;
  if (wide_tiles >= 192) return p.K <= 640;            // the 256 x 320 kernel takes it too, with less than two tiles per CU
  return p.K <= (tiles >= 256 ? 2560 : 1280);
}

template <typename T, bool GEGLU, bool RES, bool LN, bool VT = false>
static int launch_duo_cfg(const GemmParams& p, hipStream_t s) {
  auto kfn = gemm_duo_kernel<T, GEGLU, RES, LN, VT>;
  TANGO_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(kfn), DUO_LDS));
  const unsigned grid = (unsigned)((p.M / DUO_BM) * (p.N / DUO_BN));
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), DUO_LDS, s, p, tuning().duo_prio);
  TANGO_HIP(hipGetLastError());
  return 0;
}

template <typename T>
static int launch_duo_t(const GemmParams& p, hipStream_t s) {
  if (p.epi == EPI_VT) return p.ln_fold ? launch_duo_cfg<T, false, false, true, true>(p, s) : launch_duo_cfg<T, false, false, false, true>(p, s);
  if (p.ln_fold) {
    if (p.epi == EPI_GEGLU) return p.R ? launch_duo_cfg<T, true, true, true>(p, s) : launch_duo_cfg<T, true, false, true>(p, s);
    return p.R ? launch_duo_cfg<T, false, true, true>(p, s) : launch_duo_cfg<T, false, false, true>(p, s);
  }
  if (p.epi == EPI_GEGLU) return p.R ? launch_duo_cfg<T, true, true, false>(p, s) : launch_duo_cfg<T, true, false, false>(p, s);
  return p.R ? launch_duo_cfg<T, false, true, false>(p, s) : launch_duo_cfg<T, false, false, false>(p, s);
}

int launch_gemm_duo(int dtype, const GemmParams& p, hipStream_t s) {
  switch (dtype) {
    case DT_F16: return launch_duo_t<f16>(p, s);
    case DT_BF16: return launch_duo_t<bf16>(p, s);
  }
  TANGO_FAIL("gemm_duo: 16-bit dtypes only");
}

}  // namespace tango
