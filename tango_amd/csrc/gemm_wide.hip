// 256 x 320 LDS-DMA GEMM for the big 16-bit linears (round 2).
//
// Why another tile: tools/loop_probe.hip (profiles/r2_loop_probe_*.txt) takes the main loop of gemm_dma.hip apart on the GPU.
// With its 256 x 160 tile (waves of 64 x 80) the LDS is the co-critical resource: per 128-byte k-chunk the eight waves read
// 144 KiB of fragments and the DMA engine writes another 52 KiB, against 1280 MFMA cycles per SIMD -- the MFMA-only loop runs
// at 1800 TF-equivalent, MFMA + fragment reads at 1690, and switching the DMA on costs another 30 %.  Doubling the tile to
// 256 x 320 (waves of 64 x 160) cuts both per flop: fragment reads 0.45 -> 0.35 per MFMA, DMA bytes 0.33 -> 0.22 per
// MFMA-cycle.  Measured in the probe with an epilogue attached: 9-19 % faster than the 256 x 160 ping-pong loop on the UNet's
// linear shapes.  Layout differences to gemm_dma.hip:
//   * 64-byte k-chunks (one MFMA k-step), FOUR stages of (256 + 320) rows x 64 B = 144 KiB, so three chunks are in flight
//     and the ping-pong runs at chunk granularity: [ds_read 14 fragments | wait own DMAs of chunk kc+1] barrier
//     [issue chunk kc+3 | 40 MFMAs] barrier, the two 4-wave halves one barrier out of phase;
//   * a DMA instruction covers 16 rows x 64 B; the LDS image is lane-linear (row = lane >> 2, slot = lane & 3) and the
//     16-byte piece is XOR-swizzled on the SOURCE side by h(row >> 2), h = {0, 3, 2, 1}, which makes the fragment reads
//     (lane (l15, g) reads row l15, piece g) conflict-free for ds_read_b128's four lane groups (MI355X_MICROARCH.md);
//   * its own epilogue (wide_epilogue below: bias / residual / GEGLU, arithmetic bit-identical to the other GEMM kernels),
//     16 or 32 rows per pass through the first stages of the operand ring.
// Linear problems only (A [M][K] row-major, W [N][Kp]); M % 256 == 0, N % 320 == 0, K a multiple of 32 elements (any number
// of k-chunks >= 1: the prologue issues min(nk, 3) of them).
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"
#include "tuning.h"
#include "gemm_device.h"
#include "gemm_wide_device.h"

namespace tango {

// LayerNorm statistics of 8 stored elements (see linear_stream.hip: frag_stats)
template <typename T> __device__ __forceinline__ void wide_frag_stats(const u32x4& v, float& s, float& q) {
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

// SCH: where the LDS-DMAs of chunk kc+3 are issued (round 4):
//   0  all at the head of the multiply part, before the first MFMA (rounds 2-3)
//   1  inside the MFMA stream, one DMA after every eight MFMAs: the matrix pipe starts at once and covers the DMA issue
//      (profiles/r4_c1_dma_schedule_ab_b32.txt: wide linears -2...-5 %, wide convs -10 %)
// Measured and dropped (profiles/r4_c1*, r4_c2*, r4_c4*): DMAs in the read part (before or after the fragment reads: no gain; all
// of them there: -9 %), paired DMAs (= 1), and a "lean" multiply part whose scalar address preparation is pinned in the read
// part (slower than 1: what is added to the read part of one wave delays the MFMA issue of its SIMD partner).
// PRIO (run-time, TANGO_WIDE_PRIO): 0 = s_setprio 1 around the MFMAs of every multiply part, 1 = no priority changes,
// 2 = static priority for the later-dispatched half (MI355X_MICROARCH.md "Two waves per SIMD" item 4).
// XS (round 4): folded LayerNorm with the row statistics from a separate read-only pass (GemmParams::row_stats) -- no statistics
// VALU beside the MFMAs; built for the GEGLU projections of levels 1-2, where the in-loop statistics made the folded form no faster
// than LayerNorm kernel + plain GEMM.
template <typename T, bool GEGLU, bool RES, bool LN, bool VT, bool SK, int SCH, bool XS = false>
__global__ __launch_bounds__(512) void gemm_wide_kernel(const GemmParams p, const int pp_mode, unsigned long long* __restrict__ trace, const int prio) {
  constexpr int BM = 256, BN = 320, CB = 64, NST = 4;
  constexpr int ROWS = BM + BN, STAGE = ROWS * CB;
  constexpr int RG = ROWS / 16;                  // 16-row groups (1 KiB) per stage: 36
  constexpr int RGW = (RG + 7) / 8;              // DMA instructions per wave per chunk: 5 (waves 0-3) or 4 (waves 4-7)
  constexpr int TM = 4, TN = 10;                 // wave tile 64 x 160; waves 4 (M) x 2 (N)
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];   // NST stages

  // TANGO_WIDE_TRACE=1: wave 0 of every workgroup records 100 MHz timestamps at start / first chunk landed / loop end / end
#ifdef TANGO_WIDE_TRACE_BUILD      // diagnostic build only (profiles/r2_wide_trace*.txt): per-workgroup phase times
  unsigned long long t_start = 0, t_first = 0, t_loop = 0;
  if (trace) t_start = __builtin_amdgcn_s_memrealtime();
#endif
  const int NT = p.N / BN;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // XCD x works through a contiguous range of tiles
  }
  const int m0 = (bid / NT) * BM, n0 = (bid % NT) * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;

  // ---- DMA source rows: row group rg = wave + 8 i; i < 2 are activation rows (rg < 16), i >= 2 weight rows, for every wave ----
  const int lrow = lane >> 2;
  const int pc = (lane & 3) ^ ((4 - (lrow >> 2)) & 3);
  // wave-uniform bases of this tile's first activation / weight row; per-lane 32-bit offsets from them (256 / 320 rows: < 4 GiB)
  const unsigned char* const At = (const unsigned char*)p.A + (int64_t)m0 * p.lda * (int64_t)sizeof(T);
  const unsigned char* const Wt = (const unsigned char*)p.W + ((int64_t)n0 * p.Kp + (p.wb_rows ? (int64_t)(m0 / p.wb_rows) * p.wb_stride : 0)) * (int64_t)sizeof(T);
  unsigned r_off[RGW];
#pragma unroll
  for (int i = 0; i < RGW; ++i) {
    const int row = (wave + 8 * i) * 16 + lrow;
    r_off[i] = i < 2 ? (unsigned)((int64_t)row * p.lda * (int64_t)sizeof(T)) + pc * 16
                     : (unsigned)((int64_t)(row - BM) * p.Kp * (int64_t)sizeof(T)) + pc * 16;
  }
  // split-K (plain epilogue only): blockIdx.y takes a contiguous range of k-chunks, partial tiles go to the workspace
  int kc0 = 0, nk = (p.K * (int)sizeof(T)) / CB;
  if (SK) {
    const int per = (nk + (int)gridDim.y - 1) / (int)gridDim.y;
    kc0 = (int)blockIdx.y * per;
    nk = (nk < kc0 + per ? nk : kc0 + per) - kc0;
  }
  const int my_count = wave < RG - 8 * (RGW - 1) ? RGW : RGW - 1;      // wave-uniform
  const bool has5 = my_count == RGW;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
  // one DMA: rows of group wave + 8 i of chunk (source bases sa / sw already at the chunk's k offset) into the stage at LDS byte ldst
  auto dma = [&](const int i, const unsigned char* sa, const unsigned char* sw, const unsigned ldst) {
    unsigned o = r_off[i];
    asm volatile("" : "+v"(o));     // zero-extension next to the use: hipcc then picks the [SGPR base + 32-bit VGPR offset] form
    __builtin_amdgcn_global_load_lds((gptr_t)((i < 2 ? sa : sw) + o), (lptr_t)(uintptr_t)(ldst + (unsigned)i * 8192u), 16, 0, 0);   // ldst: the wave's first group in the stage
  };
  auto issue_chunk = [&](const int kc, const int st) {
    const unsigned char* sa = At + (int64_t)(kc0 + kc) * CB;
    const unsigned char* sw = Wt + (int64_t)(kc0 + kc) * CB;
#pragma unroll
    for (int i = 0; i < RGW; ++i)
      if (i < RGW - 1 || has5) dma(i, sa, sw, lds0 + st * STAGE + (unsigned)wave * 1024u);
  };
  // at most `chunks` whole chunks of this wave's DMAs may stay in flight
  auto wait_inflight = [&](const int chunks) {
    if (chunks <= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    if (has5) {
      if (chunks == 1) wait_vmcnt_lit<RGW>();
      else wait_vmcnt_lit<2 * RGW>();
    } else {
      if (chunks == 1) wait_vmcnt_lit<RGW - 1>();
      else wait_vmcnt_lit<2 * (RGW - 1)>();
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  float ssum[TM] = {0.f, 0.f, 0.f, 0.f}, ssq[TM] = {0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, g = lane >> 4;
  const int foff = l15 * CB + ((g ^ ((4 - (l15 >> 2)) & 3)) * 16);
  const int xrow = (wm * TM * 16) * CB + foff;
  const int wrow = (BM + wn * TN * 16) * CB + foff;

  const int npro = nk < NST - 1 ? nk : NST - 1;
  for (int c = 0; c < npro; ++c) issue_chunk(c, c);
  const int half = pp_phase_half(wave, lane, (unsigned*)(dsm + (NST - 1) * STAGE), pp_mode);   // scratch: last stage, first DMA'd in the loop
  wait_inflight(npro - 1);                              // chunk 0 landed
  pp_barrier();
#ifdef TANGO_WIDE_TRACE_BUILD
  if (trace) t_first = __builtin_amdgcn_s_memrealtime();
#endif
  if (half) pp_barrier();                               // the stagger
  if (prio == 2 && half) __builtin_amdgcn_s_setprio(1);
  int st = 0;
  for (int kc = 0; kc < nk; ++kc) {
    const unsigned char* Xs = dsm + st * STAGE;
    const int st3 = st == 0 ? NST - 1 : st - 1;        // chunk kc+3 refills the stage of chunk kc-1
    // ---- read part ----
    u32x4 wf[TN], xf[TM];
#pragma unroll
    for (int a = 0; a < TN; ++a) wf[a] = *(const u32x4*)(Xs + wrow + a * 16 * CB);
#pragma unroll
    for (int b = 0; b < TM; ++b) xf[b] = *(const u32x4*)(Xs + xrow + b * 16 * CB);
    const bool more = kc + NST - 1 < nk;               // chunk kc+3 exists
    const bool more5 = more && has5;
    const unsigned char* sa = At + (int64_t)(kc0 + kc + NST - 1) * CB;
    const unsigned char* sw = Wt + (int64_t)(kc0 + kc + NST - 1) * CB;
    const unsigned ldst = lds0 + st3 * STAGE + (unsigned)wave * 1024u;
    // this wave's DMAs of chunk kc+1 must have landed before the barrier that precedes anyone's read of that chunk;
    // chunk kc+2 (if issued) may stay in flight
    if (kc + 1 < nk) wait_inflight(kc + 2 < nk ? 1 : 0);
    pp_barrier();
    // ---- multiply part ----
    if (SCH == 0 && more) issue_chunk(kc + NST - 1, st3);
    if (prio == 0) __builtin_amdgcn_s_setprio(1);
    if (LN && !XS) {
      // row statistics from the activation fragments this wave holds anyway (both column halves compute them: 32 VALU
      // instructions per chunk next to 40 MFMAs)
#pragma unroll
      for (int b = 0; b < TM; ++b) wide_frag_stats<T>(xf[b], ssum[b], ssq[b]);
    }
#pragma unroll
    for (int a = 0; a < TN; ++a) {
#pragma unroll
      for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
      if (SCH != 0 && (a & 1) == 0) {
        // after MFMAs 4, 12, 20, 28, 36: the a / 2-th DMA of chunk kc+3
        const int i = a >> 1;
        __builtin_amdgcn_sched_barrier(0);
        if (i < RGW - 1 ? more : more5) dma(i, sa, sw, ldst);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (prio == 0) __builtin_amdgcn_s_setprio(0);
    pp_barrier();
    st = st == NST - 1 ? 0 : st + 1;
  }
  if (!half) pp_barrier();
  if (prio == 2) __builtin_amdgcn_s_setprio(0);
  __syncthreads();   // every wave is past its last fragment read: the operand stages become the staging area
#ifdef TANGO_WIDE_TRACE_BUILD
  if (trace) t_loop = __builtin_amdgcn_s_memrealtime();
#endif
  float mean[TM] = {0.f, 0.f, 0.f, 0.f}, rstd[TM] = {1.f, 1.f, 1.f, 1.f};
  if (LN && XS) {
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const f32x2 st = *(const f32x2*)(p.row_stats + (int64_t)(m0 + wm * TM * 16 + b * 16 + (lane & 15)) * 2);
      mean[b] = st.x; rstd[b] = st.y;
    }
  } else if (LN) {
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
  if (SK) {
    wide_epilogue_raw(p, acc, (int)blockIdx.y, m0 + wm * TM * 16, n0 + wn * TN * 16, lane, slice);
    return;
  }
  if (VT && n0 >= p.vt_n0) wide_epilogue_vt<T, LN>(p, acc, mean, rstd, m0 + wm * TM * 16, n0 + wn * TN * 16, lane, slice);
  else wide_epilogue<T, GEGLU, RES, LN>(p, acc, mean, rstd, m0 + wm * TM * 16, n0 + wn * TN * 16, lane, slice);
#ifdef TANGO_WIDE_TRACE_BUILD
  if (trace && tid == 0) {
    const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();      // every store of wave 0 issued ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // ... and acknowledged
    unsigned long long* t = trace + (size_t)blockIdx.x * 5;
    t[0] = t_start; t[1] = t_first; t[2] = t_loop; t[3] = t_issued; t[4] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// In-wave software pipeline (round 6, PIPE): the same tile, ring and DMA source layout, but no read part.  The fragments of chunk
// kc+1 are requested WHILE chunk kc is multiplied -- wf[a] is refilled in place right behind the four MFMAs that consumed it, xf[b]
// behind its last use in column group 9 -- from inline asm (hipcc would wait lgkmcnt(0) at every use while an LDS-DMA is pending,
// xattn.hip), and are all waited for ONCE, at the end of the item.  Both waves of a SIMD therefore always have MFMAs to issue: the
// 60-185 cycles an LDS-DMA issue holds its wave (MI355X_MICROARCH.md) are covered by the partner's MFMAs instead of leaving the matrix
// pipe idle, which is what the ping-pong schedule above cannot do (its partner is in its read part by construction).
// An item is two half-items of 20 MFMAs with a raw barrier after each; the 4-wave halves run one barrier apart (half B is in
// H1(kc-1) while half A is in H0(kc)), so a wave parked at a barrier or in the end-of-item lgkmcnt(0) has a partner in mid-stream.
//   slot 2kc:   A H0(kc)   | B H1(kc-1)        H0(kc): groups 0-4, refills wf[0..4] from stage (kc+1)%4; then waits its own DMAs of
//   slot 2kc+1: A H1(kc)   | B H0(kc)                  chunk kc+2 (chunk kc+3 stays in flight)
//                                              H1(kc): groups 5-9, refills wf[5..9] / xf[0..3]; issues chunk kc+4 into stage kc%4;
//                                                      lgkmcnt(0)
// Stage kc%4 (chunk kc) is read by A in slots 2kc-2, 2kc-1 and by B in slots 2kc-1, 2kc, all reads returned by the barrier that ends
// slot 2kc (B's end-of-item lgkmcnt(0)); it is overwritten from slot 2kc+1 on.  Chunk kc+2 is first read in slot 2kc+2 (A's
// H0(kc+1)); every wave has waited for its own pieces of it by the end of its H0(kc) (A: slot 2kc, B: slot 2kc+1), one barrier earlier.
// Same MFMA order per accumulator as gemm_wide_kernel: bit-identical results.
// ------------------------------------------------------------------------------------------------------------------------
template <int OFF> __device__ __forceinline__ void wp_lds_read(u32x4& v, const unsigned base) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF));
}
// every fragment read issued so far has returned; ties the 14 fragments to the wait (no consumer can be scheduled above it)
__device__ __forceinline__ void wp_lds_wait_all(u32x4 (&wf)[10], u32x4 (&xf)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4]), "+v"(wf[5]), "+v"(wf[6]), "+v"(wf[7]), "+v"(wf[8]), "+v"(wf[9]),
                 "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]));
}
template <int I, int N, typename F> __device__ __forceinline__ void wp_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); wp_for<I + 1, N>(f); }
}

// LOCK (TANGO_WIDE_PIPE=2): the same in-wave pipeline with all eight waves in step and ONE barrier per item -- no stagger, no mid-item
// barrier.  Both waves of a SIMD are then in their MFMA streams at the same time (each one's DMA / ds_read issue slots are covered by the
// partner's MFMAs) and meet once per 80 MFMAs.  Stage kc % 4 is read (refills for item kc) during item kc - 1 only, all of those reads have
// returned at the barrier that ends item kc - 1 (end-of-item lgkmcnt(0)), so chunk kc + 4 may be DMA'd into it anywhere in item kc; chunk
// kc + 2 is first read during item kc + 1 and every wave has waited for its own pieces of it in the middle of item kc.
template <typename T, bool GEGLU, bool RES, bool LN, bool VT, bool XS, bool LOCK = false>
__global__ __launch_bounds__(512) void gemm_wide_pipe_kernel(const GemmParams p, const int prio) {
  constexpr int BM = 256, BN = 320, CB = 64, NST = 4;
  constexpr int ROWS = BM + BN, STAGE = ROWS * CB;
  constexpr int RG = ROWS / 16, RGW = (RG + 7) / 8;
  constexpr int TM = 4, TN = 10;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];

  const int NT = p.N / BN;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / NT) * BM, n0 = (bid % NT) * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2, half = wave >> 2;

  const int lrow = lane >> 2;
  const int pc = (lane & 3) ^ ((4 - (lrow >> 2)) & 3);
  const unsigned char* const At = (const unsigned char*)p.A + (int64_t)m0 * p.lda * (int64_t)sizeof(T);
  const unsigned char* const Wt = (const unsigned char*)p.W + ((int64_t)n0 * p.Kp + (p.wb_rows ? (int64_t)(m0 / p.wb_rows) * p.wb_stride : 0)) * (int64_t)sizeof(T);
  unsigned r_off[RGW];
#pragma unroll
  for (int i = 0; i < RGW; ++i) {
    const int row = (wave + 8 * i) * 16 + lrow;
    r_off[i] = i < 2 ? (unsigned)((int64_t)row * p.lda * (int64_t)sizeof(T)) + pc * 16
                     : (unsigned)((int64_t)(row - BM) * p.Kp * (int64_t)sizeof(T)) + pc * 16;
  }
  const int nk = (p.K * (int)sizeof(T)) / CB;            // >= NST (launcher)
  const bool has5 = wave < RG - 8 * (RGW - 1);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
  auto dma = [&](const int i, const unsigned char* sa, const unsigned char* sw, const unsigned ldst) {
    unsigned o = r_off[i];
    asm volatile("" : "+v"(o));
    __builtin_amdgcn_global_load_lds((gptr_t)((i < 2 ? sa : sw) + o), (lptr_t)(uintptr_t)(ldst + (unsigned)i * 8192u), 16, 0, 0);
  };
  auto issue_chunk = [&](const int kc) {
    const unsigned char* sa = At + (int64_t)kc * CB;
    const unsigned char* sw = Wt + (int64_t)kc * CB;
#pragma unroll
    for (int i = 0; i < RGW; ++i)
      if (i < RGW - 1 || has5) dma(i, sa, sw, lds0 + (kc & (NST - 1)) * STAGE + (unsigned)wave * 1024u);
  };
  // at most `chunks` (0..2) whole chunks of this wave's DMAs may stay in flight
  auto wait_inflight = [&](const int chunks) {
    if (chunks <= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    if (has5) {
      if (chunks == 1) wait_vmcnt_lit<RGW>();
      else wait_vmcnt_lit<2 * RGW>();
    } else {
      if (chunks == 1) wait_vmcnt_lit<RGW - 1>();
      else wait_vmcnt_lit<2 * (RGW - 1)>();
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ssum[TM] = {0.f, 0.f, 0.f, 0.f}, ssq[TM] = {0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, g = lane >> 4;
  const int foff = l15 * CB + ((g ^ ((4 - (l15 >> 2)) & 3)) * 16);
  const unsigned xrow = lds0 + (unsigned)((wm * TM * 16) * CB + foff);
  const unsigned wrow = lds0 + (unsigned)((BM + wn * TN * 16) * CB + foff);

  for (int c = 0; c < NST; ++c) issue_chunk(c);
  wait_inflight(2);                                     // chunks 0 and 1 landed
  pp_barrier();
  u32x4 wf[TN], xf[TM];
  wp_for<0, TN>([&](auto a) { wp_lds_read<a * 16 * CB>(wf[a], wrow); });
  wp_for<0, TM>([&](auto b) { wp_lds_read<b * 16 * CB>(xf[b], xrow); });
  wp_lds_wait_all(wf, xf);
  __builtin_amdgcn_sched_barrier(0);
  if (!LOCK && half) pp_barrier();                      // the stagger
  if (prio == 2 && half) __builtin_amdgcn_s_setprio(1);
  // one item; MORE1: chunk kc+1 exists (refill the fragments) -- compile-time, the last item is peeled
  auto item = [&](auto more1_tag, const int kc) __attribute__((always_inline)) {
    constexpr bool MORE1 = decltype(more1_tag)::value;
    const bool more4 = kc + NST < nk;                   // chunk kc+4 exists: DMA into the stage of chunk kc
    const bool more45 = more4 && has5;
    const unsigned st1 = (unsigned)((kc + 1) & (NST - 1)) * STAGE;
    unsigned wsrc = wrow + st1, xsrc = xrow + st1;
    asm volatile("" : "+v"(wsrc), "+v"(xsrc));
    const unsigned char* sa = At + (int64_t)(kc + NST) * CB;
    const unsigned char* sw = Wt + (int64_t)(kc + NST) * CB;
    const unsigned ldst = lds0 + (unsigned)(kc & (NST - 1)) * STAGE + (unsigned)wave * 1024u;
    if (LN && !XS) {
#pragma unroll
      for (int b = 0; b < TM; ++b) wide_frag_stats<T>(xf[b], ssum[b], ssq[b]);
    }
    // ---- H0: column groups 0-4 ----
    wp_for<0, 5>([&](auto a_tag) {
      constexpr int a = decltype(a_tag)::value;
#pragma unroll
      for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
      __builtin_amdgcn_sched_barrier(0);
      if (MORE1) wp_lds_read<a * 16 * CB>(wf[a], wsrc);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (kc + 2 < nk) wait_inflight(kc + 3 < nk ? 1 : 0);
    if (!LOCK) pp_barrier();
    // ---- H1: column groups 5-9, the DMAs of chunk kc+4 ----
    wp_for<5, 9>([&](auto a_tag) {
      constexpr int a = decltype(a_tag)::value;
#pragma unroll
      for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
      __builtin_amdgcn_sched_barrier(0);
      if (MORE1) wp_lds_read<a * 16 * CB>(wf[a], wsrc);
      if (more4) dma(a - 5, sa, sw, ldst);
      __builtin_amdgcn_sched_barrier(0);
    });
    wp_for<0, TM>([&](auto b_tag) {
      constexpr int b = decltype(b_tag)::value;
      Mma<T>::run(acc[9][b], wf[9], xf[b]);
      __builtin_amdgcn_sched_barrier(0);
      if (MORE1) wp_lds_read<b * 16 * CB>(xf[b], xsrc);
      if (b == 1 && more45) dma(4, sa, sw, ldst);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (MORE1) {
      wp_lds_read<9 * 16 * CB>(wf[9], wsrc);
      wp_lds_wait_all(wf, xf);
    }
    pp_barrier();
  };
  for (int kc = 0; kc + 1 < nk; ++kc) item(std::true_type{}, kc);
  item(std::false_type{}, nk - 1);
  if (!LOCK && !half) pp_barrier();
  if (prio == 2) __builtin_amdgcn_s_setprio(0);
  __syncthreads();   // every wave is past its last fragment read: the operand stages become the staging area
  float mean[TM] = {0.f, 0.f, 0.f, 0.f}, rstd[TM] = {1.f, 1.f, 1.f, 1.f};
  if (LN && XS) {
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const f32x2 sv = *(const f32x2*)(p.row_stats + (int64_t)(m0 + wm * TM * 16 + b * 16 + (lane & 15)) * 2);
      mean[b] = sv.x; rstd[b] = sv.y;
    }
  } else if (LN) {
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
  if (VT && n0 >= p.vt_n0) wide_epilogue_vt<T, LN>(p, acc, mean, rstd, m0 + wm * TM * 16, n0 + wn * TN * 16, lane, slice);
  else wide_epilogue<T, GEGLU, RES, LN>(p, acc, mean, rstd, m0 + wm * TM * 16, n0 + wn * TN * 16, lane, slice);
}

// ------------------------------------------------------------------------------------------------------------------------
// Persistent form (round 4): one workgroup per CU walks its tiles (blockIdx.x, + gridDim.x, ...) and requests chunk 0 of the NEXT
// tile into ring stage 3 -- the one stage the epilogue's staging area [0, 100 KiB) leaves alone -- BEFORE the current tile's
// epilogue, so the 3-4 us a fresh workgroup spends waiting for its first chunk (profiles/r2_wide_trace2.txt: prologue 3.3-4.3 us of
// a 31-us K = 640 tile) hide behind the epilogue.  Chunks 1 and 2 follow when the epilogue is done with stages 0-2.
// Round 2's persistent kernel (tools/experiments/gemm_pers.hip: the whole operand stream continuous, epilogue squeezed into ONE
// stage, 16 rows per pass) lost 10 %; this one keeps the epilogue as it is and prefetches one chunk.
// vmcnt: the prefetch pieces are OLDER than every load / store of the epilogue, so "chunk 0 landed" = at most the epilogue's
// NSTORES (unpredicated, counted) stores outstanding; the epilogue's own counted waits only become stricter.
// Every tile uses the stages in the order 3, 0, 1, 2, 3, ...  SCH = 1 issue points, s_setprio around the multiply part, static halves.
// ------------------------------------------------------------------------------------------------------------------------
template <typename T, bool GEGLU, bool RES, bool LN, bool VT, bool XS>
__global__ __launch_bounds__(512) void gemm_wide_pers_kernel(const GemmParams p, const int ntiles) {
  constexpr int BM = 256, BN = 320, CB = 64, NST = 4;
  constexpr int ROWS = BM + BN, STAGE = ROWS * CB;
  constexpr int RG = ROWS / 16, RGW = (RG + 7) / 8;
  constexpr int TM = 4, TN = 10;
  constexpr int NSTORES = GEGLU ? 10 : 20;       // wide_epilogue: NPASS x NIT = 4 x 5 (GEGLU 2 x 5); wide_epilogue_vt: 2 x 10
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];

  const int NT = p.N / BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2, half = wave >> 2;
  auto tile_origin = [&](const int v, int& m0, int& n0) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, idx = v >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // XCD x works through a contiguous range of tiles
    m0 = (bid / NT) * BM; n0 = (bid % NT) * BN;
  };
  const int lrow = lane >> 2;
  const int pc = (lane & 3) ^ ((4 - (lrow >> 2)) & 3);
  unsigned r_off[RGW];
#pragma unroll
  for (int i = 0; i < RGW; ++i) {
    const int row = (wave + 8 * i) * 16 + lrow;
    r_off[i] = i < 2 ? (unsigned)((int64_t)row * p.lda * (int64_t)sizeof(T)) + pc * 16
                     : (unsigned)((int64_t)(row - BM) * p.Kp * (int64_t)sizeof(T)) + pc * 16;
  }
  const int nk = (p.K * (int)sizeof(T)) / CB;
  const bool has5 = wave < RG - 8 * (RGW - 1);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
  auto dma = [&](const int i, const unsigned char* sa, const unsigned char* sw, const unsigned ldst) {
    unsigned o = r_off[i];
    asm volatile("" : "+v"(o));
    __builtin_amdgcn_global_load_lds((gptr_t)((i < 2 ? sa : sw) + o), (lptr_t)(uintptr_t)(ldst + (unsigned)i * 8192u), 16, 0, 0);
  };
  auto issue_chunk = [&](const unsigned char* At, const unsigned char* Wt, const int kc) {
    const unsigned char* sa = At + (int64_t)kc * CB;
    const unsigned char* sw = Wt + (int64_t)kc * CB;
    const int st = (NST - 1 + kc) & (NST - 1);
#pragma unroll
    for (int i = 0; i < RGW; ++i)
      if (i < RGW - 1 || has5) dma(i, sa, sw, lds0 + st * STAGE + (unsigned)wave * 1024u);
  };
  auto wait_inflight = [&](const int chunks) {
    if (chunks <= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    if (has5) {
      if (chunks == 1) wait_vmcnt_lit<RGW>();
      else wait_vmcnt_lit<2 * RGW>();
    } else {
      if (chunks == 1) wait_vmcnt_lit<RGW - 1>();
      else wait_vmcnt_lit<2 * (RGW - 1)>();
    }
  };
  const int l15 = lane & 15, g = lane >> 4;
  const int foff = l15 * CB + ((g ^ ((4 - (l15 >> 2)) & 3)) * 16);
  const int xrow = (wm * TM * 16) * CB + foff;
  const int wrow = (BM + wn * TN * 16) * CB + foff;
  const int npro = nk < NST - 1 ? nk : NST - 1;

  int t = blockIdx.x, m0, n0;
  tile_origin(t, m0, n0);
  const unsigned char* At = (const unsigned char*)p.A + (int64_t)m0 * p.lda * (int64_t)sizeof(T);
  const unsigned char* Wt = (const unsigned char*)p.W + ((int64_t)n0 * p.Kp + (p.wb_rows ? (int64_t)(m0 / p.wb_rows) * p.wb_stride : 0)) * (int64_t)sizeof(T);
  for (int c = 0; c < npro; ++c) issue_chunk(At, Wt, c);
  wait_inflight(npro - 1);                              // chunk 0 landed
  pp_barrier();
  for (;;) {
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ssum[TM] = {0.f, 0.f, 0.f, 0.f}, ssq[TM] = {0.f, 0.f, 0.f, 0.f};
    if (half) pp_barrier();                             // the stagger
    int st = NST - 1;
    for (int kc = 0; kc < nk; ++kc) {
      const unsigned char* Xs = dsm + st * STAGE;
      const int st3 = st == 0 ? NST - 1 : st - 1;
      u32x4 wf[TN], xf[TM];
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[a] = *(const u32x4*)(Xs + wrow + a * 16 * CB);
#pragma unroll
      for (int b = 0; b < TM; ++b) xf[b] = *(const u32x4*)(Xs + xrow + b * 16 * CB);
      const bool more = kc + NST - 1 < nk;
      const bool more5 = more && has5;
      const unsigned char* sa = At + (int64_t)(kc + NST - 1) * CB;
      const unsigned char* sw = Wt + (int64_t)(kc + NST - 1) * CB;
      const unsigned ldst = lds0 + st3 * STAGE + (unsigned)wave * 1024u;
      if (kc + 1 < nk) wait_inflight(kc + 2 < nk ? 1 : 0);
      pp_barrier();
      __builtin_amdgcn_s_setprio(1);
      if (LN && !XS) {
#pragma unroll
        for (int b = 0; b < TM; ++b) wide_frag_stats<T>(xf[b], ssum[b], ssq[b]);
      }
#pragma unroll
      for (int a = 0; a < TN; ++a) {
#pragma unroll
        for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
        if ((a & 1) == 0) {
          const int i = a >> 1;
          __builtin_amdgcn_sched_barrier(0);
          if (i < RGW - 1 ? more : more5) dma(i, sa, sw, ldst);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      pp_barrier();
      st = st == NST - 1 ? 0 : st + 1;
    }
    if (!half) pp_barrier();
    __syncthreads();   // every wave is past its last fragment read: stages 0-2 become the staging area, stage 3 takes the next tile's chunk 0
    const int tn = t + (int)gridDim.x;
    const bool has_next = tn < ntiles;
    int m0n = 0, n0n = 0;
    const unsigned char* Atn = At;
    const unsigned char* Wtn = Wt;
    if (has_next) {
      tile_origin(tn, m0n, n0n);
      Atn = (const unsigned char*)p.A + (int64_t)m0n * p.lda * (int64_t)sizeof(T);
      Wtn = (const unsigned char*)p.W + ((int64_t)n0n * p.Kp + (p.wb_rows ? (int64_t)(m0n / p.wb_rows) * p.wb_stride : 0)) * (int64_t)sizeof(T);
      issue_chunk(Atn, Wtn, 0);
    }
    // the epilogue's per-lane indices must be recomputed per tile: derived from an opaque copy of the lane id, or hipcc hoists them out
    // of the tile loop and keeps ~25 more VGPRs alive across the main loop (256 + spills)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    float mean[TM] = {0.f, 0.f, 0.f, 0.f}, rstd[TM] = {1.f, 1.f, 1.f, 1.f};
    if (LN && XS) {
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const f32x2 sv = *(const f32x2*)(p.row_stats + (int64_t)(m0 + wm * TM * 16 + b * 16 + (lane_e & 15)) * 2);
        mean[b] = sv.x; rstd[b] = sv.y;
      }
    } else if (LN) {
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
    if (VT && n0 >= p.vt_n0) wide_epilogue_vt<T, LN>(p, acc, mean, rstd, m0 + wm * TM * 16, n0 + wn * TN * 16, lane_e, slice);
    else wide_epilogue<T, GEGLU, RES, LN>(p, acc, mean, rstd, m0 + wm * TM * 16, n0 + wn * TN * 16, lane_e, slice);
    if (!has_next) break;
    wait_vmcnt_lit<NSTORES>();          // everything older than this tile's stores has retired: chunk 0 of the next tile is in stage 3
    __syncthreads();                    // ... for every wave, and nobody reads the staging area any more
    for (int c = 1; c < npro; ++c) issue_chunk(Atn, Wtn, c);
    t = tn; m0 = m0n; n0 = n0n; At = Atn; Wt = Wtn;
  }
}

static int wide_pers_grid(long ntiles) {
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    else cus = 256;
  }
  return ntiles < cus ? (int)ntiles : cus;
}
// workgroups of the persistent form on this device (= its CU count; 256 when no device is visible): the route query of ops_abi.hip
// labels "+pers" with the same number the launcher uses
int gemm_wide_pers_cus() { return wide_pers_grid(1L << 30); }

template <typename T, bool GEGLU, bool RES, bool LN, bool VT, bool XS>
static int launch_wide_pers_cfg(const GemmParams& p, hipStream_t s) {
  constexpr int LDS = 4 * (256 + 320) * 64;
  auto kfn = gemm_wide_pers_kernel<T, GEGLU, RES, LN, VT, XS>;
  TANGO_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(kfn), LDS));
  const long ntiles = (long)(p.M / 256) * (p.N / 320);
  hipLaunchKernelGGL(kfn, dim3((unsigned)wide_pers_grid(ntiles)), dim3(512), LDS, s, p, (int)ntiles);
  TANGO_HIP(hipGetLastError());
  return 0;
}

// the persistent form takes a problem when it has at least `wide_pers` tiles per CU-workgroup to walk (TANGO_WIDE_PERS; 0 = never)
template <typename T>
static int launch_wide_pers_t(const GemmParams& p, hipStream_t s, bool& taken) {
  taken = true;
  if (p.row_stats) return launch_wide_pers_cfg<T, true, false, true, false, true>(p, s);
  if (p.epi == EPI_VT) return p.ln_fold ? launch_wide_pers_cfg<T, false, false, true, true, false>(p, s) : launch_wide_pers_cfg<T, false, false, false, true, false>(p, s);
  if (p.epi == EPI_GEGLU) {
    if (!p.R && !p.ln_fold) return launch_wide_pers_cfg<T, true, false, false, false, false>(p, s);
    taken = false;
    return 0;
  }
  if (p.ln_fold) return p.R ? launch_wide_pers_cfg<T, false, true, true, false, false>(p, s) : launch_wide_pers_cfg<T, false, false, true, false, false>(p, s);
  return p.R ? launch_wide_pers_cfg<T, false, true, false, false, false>(p, s) : launch_wide_pers_cfg<T, false, false, false, false, false>(p, s);
}

// which problems: 16-bit linear, plain / GEGLU epilogue into T, whole 256 x 320 tiles, at least NST k-chunks, and enough
// tiles to fill the chip
bool gemm_wide_ok(int dtype, const GemmParams& p) {
  if (tuning().no_wide_gemm || dtype == DT_F32) return false;
  const bool linear = p.mode == GATHER_1D && p.taps == 1 && p.rows_pb == p.M && p.in_mul == 1 && p.in_off == 0 && p.out_mul == 1 &&
                      p.out_off == 0 && p.Lin >= p.M;
  if (!linear || p.batch != 1 || p.a_act != ACT_NONE || p.bias_rows) return false;
  // (split-K on this tile was measured in round 2 -- no gain over the 4-wave tiles' split-K at M = 4096 -- and is not compiled)
  if (p.splitk > 1 || p.out_f32) return false;
  if (p.ln_fold && (!p.wsum || ((uintptr_t)p.wsum & 15) || p.alpha != 1.f)) return false;
  if (p.epi != EPI_NONE && p.epi != EPI_GEGLU && p.epi != EPI_VT) return false;
  if (p.epi == EPI_VT && (p.R || p.bias2 || p.vt_n0 % 320 != 0 || p.vt_S % 256 != 0 || p.vt_ld % 8 != 0 || ((uintptr_t)p.vt & 15))) return false;
  // folded LayerNorm: measured against the alternatives on one box -- K = 320 rows stay on the streaming kernel, and for the
  // GEGLU shapes (N = 8 C) the separate LayerNorm kernel + plain wide GEMM is as fast or faster (row statistics cost 32 VALU
  // instructions per k-chunk inside the MFMA phase); the narrow projections (N <= 3 C) gain 25-30 %
  // folded LayerNorm on this tile, measured (round 2; round 3 profiles/r3_c4_level0_ln_routing_ab.txt, ms per step at B = 32):
  //   GEGLU shapes (N = 8 C): never -- the separate LayerNorm kernel + plain wide GEMM is as fast at levels 1-2, and at level 0
  //   (K = 320) the streaming kernel wins outright (x5: 3.27 vs 4.06);
  //   narrow projections (N <= 3 C): levels 1-2 gain 25-30 %; level 0 (K = 320) q | k | v^T x5 1.63 -> 1.47 and attn2.to_q x5
  //   0.59 -> 0.53 at M = 262144, but at M = 65536 (B = 8) the q | k | v^T shape is ~6 % slower here (0.405 vs 0.431): K < 640
  //   comes here from 131072 rows on, or when the whole problem is one column of tiles
  if (p.row_stats && !(p.ln_fold && p.epi == EPI_GEGLU && !p.R && !p.bias2)) return false;      // XS exists for the GEGLU epilogue only
  if (p.ln_fold && !p.row_stats && (p.epi == EPI_GEGLU || (p.K < 640 && p.M < 131072 && p.N > 320))) return false;
  if (p.e_act != ACT_NONE) return false;            // (an inlined activation switch per element bloated this kernel 10x: not supported here)
  if (p.M % 256 != 0 || p.N % 320 != 0 || (p.K * 2) % 64 != 0) return false;
  if (p.ldo % 8 != 0 || ((uintptr_t)p.out & 15) || (p.R && (p.ldr % 8 != 0 || ((uintptr_t)p.R & 15)))) return false;
  if ((p.lda * 2) % 16 != 0 || (p.Kp * 2) % 16 != 0 || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return false;
  if (((uintptr_t)p.bias & 15) || ((uintptr_t)p.bias2 & 15) || (p.bias2 && p.bias2_stride % 4 != 0)) return false;
  const long tiles = (long)(p.M / 256) * (p.N / 320);
  // 192 tiles = 3/4 of the CUs is where this kernel starts to win (round 3, profiles/r3_c13_b8_dispatch_thresholds.txt: level-2
  // q | k | v^T at B = 8, M=4096 N=3840 K=1280 with folded LN, x5 0.68 -> 0.40 ms); at 128 tiles it is level with the streaming kernel
  // on K = 640 and loses to the 256 x 160 kernel on K = 2560 (0.48 vs 0.37 ms)
  return tuning().force_big_kernels || tiles >= 192;
}

template <typename T, bool GEGLU, bool RES, bool LN, bool VT, bool SK, int SCH, bool XS = false>
static int launch_wide_sch(const GemmParams& p, hipStream_t s) {
  constexpr int LDS = 4 * (256 + 320) * 64;
  auto kfn = gemm_wide_kernel<T, GEGLU, RES, LN, VT, SK, SCH, XS>;
  TANGO_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(kfn), LDS));
  const int pp_mode = 0;   // static half assignment: waves w and w + 4 share a SIMD (tools/simd_probe.hip)
  const unsigned grid = (unsigned)((p.M / 256) * (p.N / 320));
  unsigned long long* trace = nullptr;
#ifdef TANGO_WIDE_TRACE_BUILD
  const bool tracing = getenv("TANGO_WIDE_TRACE") != nullptr && p.splitk <= 1;
  if (tracing) TANGO_HIP(hipMalloc((void**)&trace, (size_t)grid * 40));
#endif
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), LDS, s, p, pp_mode, trace, tuning().wide_prio);
  TANGO_HIP(hipGetLastError());
#ifdef TANGO_WIDE_TRACE_BUILD
  if (tracing) {
    std::vector<unsigned long long> h((size_t)grid * 5);
    TANGO_HIP(hipStreamSynchronize(s));
    TANGO_HIP(hipMemcpy(h.data(), trace, (size_t)grid * 40, hipMemcpyDeviceToHost));
    TANGO_HIP(hipFree(trace));
    unsigned long long t0 = ~0ull, t1 = 0;
    double pro = 0, loop = 0, epi = 0, ack = 0;
    for (unsigned i = 0; i < grid; ++i) {
      const unsigned long long* t = &h[(size_t)i * 5];
      t0 = t[0] < t0 ? t[0] : t0;
      t1 = t[4] > t1 ? t[4] : t1;
      pro += (double)(t[1] - t[0]); loop += (double)(t[2] - t[1]); epi += (double)(t[3] - t[2]); ack += (double)(t[4] - t[3]);
    }
    fprintf(stderr, "gemm_wide trace M=%d N=%d K=%d epi=%d res=%d: %u tiles, span %.1f us; per tile: prologue %.2f us, main loop %.2f us, "
            "epilogue issue %.2f us, store drain %.2f us\n",
            p.M, p.N, p.K, p.epi, p.R ? 1 : 0, grid, (double)(t1 - t0) * 0.01, pro / grid * 0.01, loop / grid * 0.01, epi / grid * 0.01, ack / grid * 0.01);
  }
#endif
  return 0;
}

template <typename T, bool GEGLU, bool RES, bool LN, bool VT = false, bool SK = false>
static int launch_wide_cfg(const GemmParams& p, hipStream_t s) {
  switch (tuning().wide_sched) {
    case 0: return launch_wide_sch<T, GEGLU, RES, LN, VT, SK, 0>(p, s);
    default: return launch_wide_sch<T, GEGLU, RES, LN, VT, SK, 1>(p, s);
  }
}

template <typename T>
static int launch_wide_t(const GemmParams& p, hipStream_t s) {
  if (p.row_stats) return launch_wide_sch<T, true, false, true, false, false, 1, true>(p, s);      // gemm_wide_ok(): GEGLU, no residual
  if (p.epi == EPI_VT) return p.ln_fold ? launch_wide_cfg<T, false, false, true, true>(p, s) : launch_wide_cfg<T, false, false, false, true>(p, s);
  if (p.ln_fold) {
    if (p.epi == EPI_GEGLU) return p.R ? launch_wide_cfg<T, true, true, true>(p, s) : launch_wide_cfg<T, true, false, true>(p, s);
    return p.R ? launch_wide_cfg<T, false, true, true>(p, s) : launch_wide_cfg<T, false, false, true>(p, s);
  }
  if (p.epi == EPI_GEGLU) return p.R ? launch_wide_cfg<T, true, true, false>(p, s) : launch_wide_cfg<T, true, false, false>(p, s);
  return p.R ? launch_wide_cfg<T, false, true, false>(p, s) : launch_wide_cfg<T, false, false, false>(p, s);
}

template <typename T, bool GEGLU, bool RES, bool LN, bool VT, bool XS>
static int launch_wide_pipe_cfg(const GemmParams& p, hipStream_t s) {
  constexpr int LDS = 4 * (256 + 320) * 64;
  auto kfn = tuning().wide_pipe == 2 ? gemm_wide_pipe_kernel<T, GEGLU, RES, LN, VT, XS, true> : gemm_wide_pipe_kernel<T, GEGLU, RES, LN, VT, XS, false>;
  TANGO_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(kfn), LDS));
  const unsigned grid = (unsigned)((p.M / 256) * (p.N / 320));
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), LDS, s, p, tuning().wide_prio);
  TANGO_HIP(hipGetLastError());
  return 0;
}

// the in-wave software pipeline (TANGO_WIDE_PIPE): the epilogue kinds the persistent form has, K >= 4 chunks
template <typename T>
static int launch_wide_pipe_t(const GemmParams& p, hipStream_t s, bool& taken) {
  taken = true;
  if (p.row_stats) return launch_wide_pipe_cfg<T, true, false, true, false, true>(p, s);
  if (p.epi == EPI_VT) return p.ln_fold ? launch_wide_pipe_cfg<T, false, false, true, true, false>(p, s) : launch_wide_pipe_cfg<T, false, false, false, true, false>(p, s);
  if (p.epi == EPI_GEGLU) {
    if (!p.R && !p.ln_fold) return launch_wide_pipe_cfg<T, true, false, false, false, false>(p, s);
    taken = false;
    return 0;
  }
  if (p.ln_fold) return p.R ? launch_wide_pipe_cfg<T, false, true, true, false, false>(p, s) : launch_wide_pipe_cfg<T, false, false, true, false, false>(p, s);
  return p.R ? launch_wide_pipe_cfg<T, false, true, false, false, false>(p, s) : launch_wide_pipe_cfg<T, false, false, false, false, false>(p, s);
}

int launch_gemm_wide(int dtype, const GemmParams& p, hipStream_t s) {
  if ((tuning().wide_pipe == 1 || tuning().wide_pipe == 2) && p.splitk <= 1 && p.K * 2 >= 4 * 64) {
    bool taken = false;
    int rc = 0;
    if (dtype == DT_F16) rc = launch_wide_pipe_t<f16>(p, s, taken);
    else if (dtype == DT_BF16) rc = launch_wide_pipe_t<bf16>(p, s, taken);
    if (taken) return rc;
  }
  const int pers = tuning().wide_pers;
  if (pers > 0 && p.splitk <= 1 && (long)(p.M / 256) * (p.N / 320) >= (long)pers * wide_pers_grid(1L << 30)) {
    bool taken = false;
    int rc = 0;
    if (dtype == DT_F16) rc = launch_wide_pers_t<f16>(p, s, taken);
    else if (dtype == DT_BF16) rc = launch_wide_pers_t<bf16>(p, s, taken);
    if (taken) return rc;
  }
  switch (dtype) {
    case DT_F16: return launch_wide_t<f16>(p, s);
    case DT_BF16: return launch_wide_t<bf16>(p, s);
  }
  TANGO_FAIL("gemm_wide: 16-bit dtypes only");
}

}  // namespace tango
