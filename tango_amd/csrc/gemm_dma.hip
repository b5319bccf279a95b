// Wide 8-wave LDS-DMA gather-GEMM (256 x BN tile) for the large conv / linear problems: see gemm.hip for the 4-wave
// tile kernels and the shared conventions (swapped MFMA orientation, tap-major K, epilogue in gemm_device.h).
#include <cstdlib>

#include "common.h"
#include "tuning.h"
#include "gemm_device.h"

namespace tango {

// ------------------------------------------------------------------------------------------------
// Wide variant for the large conv / linear problems: 256 x BN tile, 8 waves (4 x 2, each 64 x BN/2), operands
// staged by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write), THREE 128-byte-row stages so
// two k-chunks are in flight across the (raw) barrier with counted vmcnt waits.  The LDS image is lane-linear
// per wave instruction (8 rows x 128 B), so the XOR swizzle is applied on the SOURCE address: LDS slot s of
// row r receives global piece s ^ (r & 7); fragment reads use the same involution.  Padding / out-of-range
// rows are sourced from a zero page.
// ------------------------------------------------------------------------------------------------
// PP = true: "ping-pong" main loop (round 2, same scheme as conv_halo.hip): the two 4-wave halves (waves w and w + 4
// share a SIMD) run half a k-chunk out of phase; every chunk is two phases of [ds_read 9 fragments] barrier [MFMAs]
// barrier, half B executes one extra barrier up front.  Chunk kc+2 is DMA'd at the start of the MFMA part of phase
// (kc, 0) (every read of chunk kc-1, whose stage it refills, retired at least one barrier earlier); every wave waits for
// its own chunk-(kc+1) DMAs in the read part of phase (kc, 1), at least one barrier before the first read of that chunk.
template <typename T, int BN, int MODE, bool PP>
__global__ __launch_bounds__(512, 2) void gemm_dma_kernel(const GemmParams p, const unsigned char* zero_page, const int staged, const int pp_mode) {
  constexpr int EPV = 16 / (int)sizeof(T);
  constexpr int BM = 256, BKB = 128;
  constexpr int BK = BKB / (int)sizeof(T);
  constexpr int ROWS = BM + BN;
  constexpr int STAGE = ROWS * BKB;
  constexpr int RG = ROWS / 8;                  // 8-row groups (1 KB) per stage
  constexpr int RGW = (RG + 7) / 8;             // DMA instructions per wave per chunk (waves may own one fewer)
  constexpr int WMR = 64, WNR = BN / 2;
  constexpr int TM = 4, TN = WNR / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];   // 3 stages

  const int MT = (p.M + BM - 1) / BM, NT = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nblk = MT * NT;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / NT) * BM, n0 = (bid % NT) * BN;
  const unsigned char* Ab = (const unsigned char*)p.A;
  const unsigned char* Wb = (const unsigned char*)p.W;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int lrow = lane >> 3, slot = lane & 7;
  const int pc = slot ^ lrow;                    // global 16-byte piece this lane fetches (row & 7 == lrow)

  // per-lane source rows of this wave's row groups rg = wave + 8*i
  int64_t r_base[RGW];       // LINEAR / W rows: byte offset of the row start (+piece); conv: batch base row; -1 invalid
  int r_c0[RGW], r_c1[RGW];
  int my_count = 0;
#pragma unroll
  for (int i = 0; i < RGW; ++i) {
    const int rg = wave + 8 * i;
    r_base[i] = -1; r_c0[i] = 0; r_c1[i] = 0;
    if (rg < RG) {
      ++my_count;
      const int row = rg * 8 + lrow;
      if (row < BM) {
        const int m = m0 + row;
        if (m < p.M) {
          if (MODE == MODE_LINEAR) {
            r_base[i] = ((int64_t)m * p.lda + pc * EPV) * (int64_t)sizeof(T);
          } else if (MODE == MODE_CONV2D) {
            const int hw = p.H * p.Wd;
            const int b = m / hw, rem = m - b * hw;
            const int y = rem / p.Wd, x = rem - y * p.Wd;
            r_base[i] = (int64_t)b * p.Hin * p.Win;
            r_c0[i] = y * p.stride - p.pad;
            r_c1[i] = x * p.stride - p.pad;
          } else {
            const int b = m / p.rows_pb, q = m - b * p.rows_pb;
            r_base[i] = (int64_t)b * p.Lin;
            r_c0[i] = q * p.in_mul + p.in_off;
          }
        }
      } else {
        const int n = n0 + row - BM;
        if (n < p.N) r_base[i] = ((int64_t)n * p.Kp + pc * EPV) * (int64_t)sizeof(T);
      }
    }
  }
  my_count = __builtin_amdgcn_readfirstlane(my_count);

  auto issue_chunk = [&](int kc, int st) {
    const int k0 = kc * BK;
    int tap = 0, cc = k0, dy = 0, dx = 0;
    if (MODE != MODE_LINEAR) {
      tap = k0 / p.Cin;
      cc = k0 - tap * p.Cin;
      if (MODE == MODE_CONV2D) { dy = tap / 3; dx = tap - dy * 3; }
    }
#pragma unroll
    for (int i = 0; i < RGW; ++i) {
      const int rg = wave + 8 * i;
      if (rg < RG) {                                   // wave-uniform
        const unsigned char* src = zero_page;
        if (r_base[i] >= 0) {
          if (rg * 8 >= BM) {                           // weight rows (wave-uniform branch)
            src = Wb + r_base[i] + (int64_t)k0 * (int64_t)sizeof(T);
          } else if (MODE == MODE_LINEAR) {
            src = Ab + r_base[i] + (int64_t)k0 * (int64_t)sizeof(T);
          } else {
            int64_t srow; bool ok;
            if (MODE == MODE_CONV2D) {
              const int iy = r_c0[i] + dy, ix = r_c1[i] + dx;
              ok = (unsigned)iy < (unsigned)(p.Hin << p.ups) && (unsigned)ix < (unsigned)(p.Win << p.ups);
              srow = r_base[i] + (int64_t)(iy >> p.ups) * p.Win + (ix >> p.ups);
            } else {
              const int idx = r_c0[i] + tap * p.tap_step;
              ok = (unsigned)idx < (unsigned)p.Lin;
              srow = r_base[i] + idx;
            }
            if (ok) src = Ab + (srow * p.lda + cc + pc * EPV) * (int64_t)sizeof(T);
          }
        }
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dsm + st * STAGE + rg * 1024), 16, 0, 0);
      }
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  int koff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) koff[ks] = (((ks * 4 + (lane >> 4)) ^ (lane & 7)) * 16);
  const int xrow = (wm * WMR + (lane & 15)) * BKB;
  const int wrow = (BM + wn * WNR + (lane & 15)) * BKB;

  if constexpr (PP) {
    const int half = pp_phase_half(wave, lane, (unsigned*)(dsm + 2 * STAGE), pp_mode);   // scratch = head of stage 2 (first DMA'd two barriers later)
    issue_chunk(0, 0);
    if (nk > 1) issue_chunk(1, 1);
    wait_vmcnt_upto8(nk > 1 ? my_count : 0);          // chunk 0 landed, chunk 1 may stay in flight
    pp_barrier();
    if (half) pp_barrier();                           // the stagger
    int st = 0;
    for (int kc = 0; kc < nk; ++kc) {
      const unsigned char* Xs = dsm + st * STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 wf[TN], xf[TM];
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[a] = *(const u32x4*)(Xs + wrow + a * 16 * BKB + koff[ks]);
#pragma unroll
        for (int b = 0; b < TM; ++b) xf[b] = *(const u32x4*)(Xs + xrow + b * 16 * BKB + koff[ks]);
        if (ks == 1 && kc + 1 < nk) wait_vmcnt_upto8(kc + 2 < nk ? my_count : 0);
        pp_barrier();
        if (ks == 0 && kc + 2 < nk) issue_chunk(kc + 2, st == 0 ? 2 : st - 1);   // stage (kc+2) % 3
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
        __builtin_amdgcn_s_setprio(0);
        pp_barrier();
      }
      st = st == 2 ? 0 : st + 1;
    }
    if (!half) pp_barrier();
  } else {
  issue_chunk(0, 0);
  if (nk > 1) issue_chunk(1, 1);
  int st = 0;
  for (int kc = 0; kc < nk; ++kc) {
    // chunk kc must have landed; the DMAs of chunk kc+1 (issued one iteration ago) may stay in flight
    if (kc + 1 < nk) {
      if (my_count == RGW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RGW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RGW - 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kc + 2 < nk) issue_chunk(kc + 2, st == 0 ? 2 : st - 1);   // stage (kc+2) % 3
    const unsigned char* Xs = dsm + st * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 wf[TN], xf[TM];
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[a] = *(const u32x4*)(Xs + wrow + a * 16 * BKB + koff[ks]);
#pragma unroll
      for (int b = 0; b < TM; ++b) xf[b] = *(const u32x4*)(Xs + xrow + b * 16 * BKB + koff[ks]);
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
    }
    st = st == 2 ? 0 : st + 1;
  }
  }
  if (MODE != MODE_CONV1D && staged) {
    __syncthreads();   // every wave is past its last fragment read: the operand stages become the staging area
    gemm_epilogue_staged<T, TM, TN>(p, acc, m0 + wm * WMR, n0 + wn * WNR, lane, dsm + wave * (32 * (WNR * 4 + 16)));
  } else {
    gemm_epilogue<T, TM, TN, MODE>(p, acc, m0 + wm * WMR, n0 + wn * WNR, lane, 0, 0);
  }
}


static const unsigned char* g_zero_page = nullptr;

template <typename T, int BN, int MODE>
static int launch_dma_cfg(const GemmParams& p, hipStream_t s) {
  constexpr int LDS = 3 * (256 + BN) * 128;
  auto kfn = gemm_dma_kernel<T, BN, MODE, true>;     // ping-pong main loop (the lock-step variant is no longer compiled)
  TANGO_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(kfn), LDS));
  const int MT = (p.M + 255) / 256, NT = (p.N + BN - 1) / BN;
  const int staged = (MODE != MODE_CONV1D && epilogue_can_stage<T>(p)) ? 1 : 0;
  const int pp_mode = 0;   // static half assignment: waves w and w + 4 share a SIMD (tools/simd_probe.hip)
  hipLaunchKernelGGL(kfn, dim3((unsigned)(MT * NT)), dim3(512), LDS, s, p, (const unsigned char*)g_zero_page, staged, pp_mode);
  TANGO_HIP(hipGetLastError());
  return 0;
}

// the wide LDS-DMA kernel takes the big problems: enough 256-row tiles to fill the chip, 128-byte k-chunks
bool gemm_dma_ok(int dtype, const GemmParams& p) {
  if (tuning().no_dma_gemm) return false;
  const int esz = dtype == DT_F32 ? 4 : 2;
  if (p.batch != 1 || p.splitk > 1 || p.a_act != ACT_NONE || (p.Cin * esz) % 128 != 0) return false;
  const int bn = (p.epi == EPI_GEGLU || p.N % 160 != 0) ? 128 : 160;
  if (p.N % bn != 0) return false;
  if (p.K / (128 / esz) < 4) return false;
  const long tiles = (long)((p.M + 255) / 256) * (p.N / bn);
  // linears from one tile per CU on (round 3, B = 8's level 1, profiles/r3_c13_b8_dispatch_thresholds.txt: M=16384 N=640 K=2560 x5
  // 0.75 -> 0.37 ms, K=640 x20 0.83 (streaming kernel) -> 0.69 ms); convs / conv1d keep the round-1 threshold of 1.75 tiles per CU
  const bool lin = p.mode == GATHER_1D && p.taps == 1;
  return tuning().force_big_kernels || tiles >= (lin ? 256 : 448);
}

template <typename T>
static int launch_dma(const GemmParams& p, hipStream_t s) {
  if (!g_zero_page) TANGO_FAIL("gemm: gemm_init() was not called (zero page for the LDS-DMA gather)");
  const bool linear = p.mode == GATHER_1D && p.taps == 1 && p.rows_pb == p.M && p.in_mul == 1 && p.in_off == 0 &&
                      p.out_mul == 1 && p.out_off == 0 && p.Lin >= p.M;
  const bool bn128 = (p.epi == EPI_GEGLU || p.N % 160 != 0);
  if (p.mode == GATHER_2D) return bn128 ? launch_dma_cfg<T, 128, MODE_CONV2D>(p, s) : launch_dma_cfg<T, 160, MODE_CONV2D>(p, s);
  if (linear) return bn128 ? launch_dma_cfg<T, 128, MODE_LINEAR>(p, s) : launch_dma_cfg<T, 160, MODE_LINEAR>(p, s);
  return bn128 ? launch_dma_cfg<T, 128, MODE_CONV1D>(p, s) : launch_dma_cfg<T, 160, MODE_CONV1D>(p, s);
}

int launch_gemm_dma(int dtype, const GemmParams& p, const unsigned char* zero_page, hipStream_t s) {
  g_zero_page = zero_page;
  switch (dtype) {
    case DT_F32: return launch_dma<float>(p, s);
    case DT_F16: return launch_dma<f16>(p, s);
    case DT_BF16: return launch_dma<bf16>(p, s);
  }
  TANGO_FAIL("gemm_dma: bad dtype");
}

}  // namespace tango
