// Ablation probe for the main loop of the 8-wave LDS-DMA GEMM (tango_amd/csrc/gemm_dma.hip): same tile (256 x 160, fp16,
// 128-byte k-chunks, three LDS stages, 4 x 2 waves of 64 x 80, swizzled fragment reads, MFMA 16x16x32), same lock-step and
// ping-pong schedules, no epilogue.  Each component can be compiled out so that its cost shows up as a difference:
//   bit 0: no DMA inside the loop            bit 1: fragments read once, before the loop
//   bit 2: no MFMA (fragments are xor-ed)    bit 3: no barriers / vmcnt waits
// Results are garbage by construction; only the time matters.  Output: us per launch and cycles per k-chunk per workgroup.
// build: hipcc --offload-arch=gfx950 -O3 -o build/loop_probe tools/loop_probe.hip
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
      exit(2);                                                                            \
    }                                                                                     \
  } while (0)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mma(f32x4& c, const u32x4& a, const u32x4& b) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}


// Epilogue emulation (bit 5): the wave's accumulators go to HBM as fp16 through a per-wave LDS staging slice, 16 rows at a
// time, so that global stores are 16-byte pieces of contiguous row segments (what gemm_epilogue_staged16 does, minus bias /
// residual / activation).  `stg` must not alias LDS another wave may still read: callers barrier first.
template <int TM_, int TN_>
__device__ __forceinline__ void epilogue_emul(f32x4 (&acc)[TN_][TM_], unsigned char* stg, _Float16* out, long ldo, int row0, int col0, int lane) {
  constexpr int WN = TN_ * 16, PITCH = WN * 2 + 16;
  const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int b = 0; b < TM_; ++b) {
#pragma unroll
    for (int a = 0; a < TN_; ++a) {
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      const h4 v = h4{(_Float16)acc[a][b][0], (_Float16)acc[a][b][1], (_Float16)acc[a][b][2], (_Float16)acc[a][b][3]};
      *(h4*)(stg + l15 * PITCH + (a * 16 + g * 4) * 2) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int PIECES = 16 * WN * 2 / 16;
#pragma unroll
    for (int i = lane; i < PIECES; i += 64) {
      const int r = i / (WN / 8), c = i % (WN / 8);
      const u32x4 v = *(const u32x4*)(stg + r * PITCH + c * 16);
      *(u32x4*)((unsigned char*)(out + (long)(row0 + b * 16 + r) * ldo + col0) + c * 16) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// Workgroup-cooperative epilogue emulation (bit 6): in two passes, all 8 waves park half of their accumulator row-blocks in one
// LDS tile [BM/2][BN] fp16 (row pitch BN*2 + 16), barrier, then every wave stores WHOLE tile rows (BN*2 contiguous bytes).
template <int BM_, int BN_, int WGM, int TM_, int TN_>
__device__ __forceinline__ void epilogue_coop(f32x4 (&acc)[TN_][TM_], unsigned char* lds, _Float16* out, long ldo, int m0, int n0, int wm,
                                              int wcol0, int tid) {
  constexpr int PITCH = BN_ * 2 + 16, HB = TM_ / 2, WROWS = TM_ * 16;
  static_assert(TM_ % 2 == 0, "two passes");
  const int lane = tid & 63, l15 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    if (ps) __syncthreads();
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
      for (int a = 0; a < TN_; ++a) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const f32x4 c = acc[a][ps * HB + b];
        const h4 v = h4{(_Float16)c[0], (_Float16)c[1], (_Float16)c[2], (_Float16)c[3]};
        *(h4*)(lds + (wm * (WROWS / 2) + b * 16 + l15) * PITCH + (wcol0 + a * 16 + g * 4) * 2) = v;
      }
    __syncthreads();
    constexpr int PPR = BN_ * 2 / 16, PIECES = (BM_ / 2) * PPR;
#pragma unroll 4
    for (int i = tid; i < PIECES; i += 512) {
      const int r = i / PPR, c = i % PPR;
      const int grow = (r / (WROWS / 2)) * WROWS + ps * (WROWS / 2) + r % (WROWS / 2);
      const u32x4 v = *(const u32x4*)(lds + r * PITCH + c * 16);
      *(u32x4*)((unsigned char*)(out + (long)(m0 + grow) * ldo + n0) + c * 16) = v;
    }
  }
}

constexpr int BM = 256, BN = 160, BKB = 128, ROWS = BM + BN, STAGE = ROWS * BKB, RG = ROWS / 8, RGW = (RG + 7) / 8;
constexpr int TM = 4, TN = 5;

template <int ABL, bool PP>
__global__ __launch_bounds__(512, 2) void probe_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ W,
                                                       float* __restrict__ out, int M, int N, int Kbytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  constexpr bool NO_DMA = ABL & 1, NO_RD = ABL & 2, NO_MMA = ABL & 4, NO_BAR = ABL & 8;
  const int NT = N / BN;
  int bid = blockIdx.x;
  if (ABL & 16) {   // production's XCD-aware order: XCD x (= blockIdx & 7) works through a contiguous range of tiles
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / NT) * BM, n0 = (bid % NT) * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int lrow = lane >> 3, slot = lane & 7, pc = slot ^ lrow;
  long r_base[RGW];
  int my_count = 0;
#pragma unroll
  for (int i = 0; i < RGW; ++i) {
    const int rg = wave + 8 * i;
    r_base[i] = 0;
    if (rg < RG) {
      ++my_count;
      const int row = rg * 8 + lrow;
      r_base[i] = row < BM ? (long)(m0 + row) * Kbytes + pc * 16 : (long)(n0 + row - BM) * Kbytes + pc * 16;
    }
  }
  auto issue_chunk = [&](int kc, int st) {
#pragma unroll
    for (int i = 0; i < RGW; ++i) {
      const int rg = wave + 8 * i;
      if (rg < RG) {
        const unsigned char* src = (rg * 8 >= BM ? W : A) + r_base[i] + (long)kc * BKB;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dsm + st * STAGE + rg * 1024), 16, 0, 0);
      }
    }
  };
  auto wait_chunk = [&](bool more) {
    if (more) {
      if (my_count == RGW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RGW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RGW - 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 dummy = u32x4{0u, 0u, 0u, 0u};
  const int nk = Kbytes / BKB;
  int koff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) koff[ks] = (((ks * 4 + (lane >> 4)) ^ (lane & 7)) * 16);
  const int xrow = (wm * 64 + (lane & 15)) * BKB;
  const int wrow = (BM + wn * 80 + (lane & 15)) * BKB;

  issue_chunk(0, 0);
  if (nk > 1) issue_chunk(1, 1);
  u32x4 wf[2][TN], xf[2][TM];
  if (NO_RD) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[ks][a] = *(const u32x4*)(dsm + wrow + a * 16 * BKB + koff[ks]);
#pragma unroll
      for (int b = 0; b < TM; ++b) xf[ks][b] = *(const u32x4*)(dsm + xrow + b * 16 * BKB + koff[ks]);
    }
  }
  const int half = wave >> 2;
  if (PP) {
    wait_chunk(nk > 1);
    __builtin_amdgcn_s_barrier();
    if (half) __builtin_amdgcn_s_barrier();
  }
  int st = 0;
  for (int kc = 0; kc < nk; ++kc) {
    const unsigned char* Xs = dsm + st * STAGE;
    if (!PP) {
      if (!NO_BAR) { wait_chunk(kc + 1 < nk); __builtin_amdgcn_s_barrier(); }
      if (!NO_DMA && kc + 2 < nk) issue_chunk(kc + 2, st == 0 ? 2 : st - 1);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (!NO_RD) {
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[ks][a] = *(const u32x4*)(Xs + wrow + a * 16 * BKB + koff[ks]);
#pragma unroll
        for (int b = 0; b < TM; ++b) xf[ks][b] = *(const u32x4*)(Xs + xrow + b * 16 * BKB + koff[ks]);
      }
      if (PP) {
        if (!NO_BAR) {
          if (ks == 1 && kc + 1 < nk) wait_chunk(kc + 2 < nk);
          __builtin_amdgcn_s_barrier();
        }
        if (!NO_DMA && ks == 0 && kc + 2 < nk) issue_chunk(kc + 2, st == 0 ? 2 : st - 1);
        __builtin_amdgcn_s_setprio(1);
      }
      if (!NO_MMA) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) mma(acc[a][b], wf[ks][a], xf[ks][b]);
      } else {
#pragma unroll
        for (int a = 0; a < TN; ++a) dummy ^= wf[ks][a];
#pragma unroll
        for (int b = 0; b < TM; ++b) dummy ^= xf[ks][b];
      }
      if (PP) {
        __builtin_amdgcn_s_setprio(0);
        if (!NO_BAR) __builtin_amdgcn_s_barrier();
      }
    }
    st = st == 2 ? 0 : st + 1;
  }
  if (PP && !half) __builtin_amdgcn_s_barrier();
  if (ABL & 32) {
    __syncthreads();
    epilogue_emul<TM, TN>(acc, dsm + wave * (16 * (TN * 32 + 16)), (_Float16*)out, N, m0 + wm * 64, n0 + wn * 80, lane);
    return;
  }
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) s += acc[a][b];
  s[0] += __builtin_bit_cast(float, dummy[0] ^ dummy[1] ^ dummy[2] ^ dummy[3]);
  if (s[0] + s[1] + s[2] + s[3] == 1234.5f) out[blockIdx.x * 512 + tid] = s[0];
}


// ---- candidate main loops: 64-byte k-chunks (one MFMA k-step per chunk), NST stages, ping-pong at chunk granularity ----
// DMA instruction = 16 rows x 64 B; LDS image lane-linear (row = lane >> 2, slot = lane & 3) with the piece XOR-swizzled on the
// source side by h(row >> 2), h = {0, 3, 2, 1}: conflict-free for ds_read_b128's lane groups (MI355X_MICROARCH.md, LDS table).
template <int BM_, int BN_, int WGM, int WGN, int NST, int ABL, int MINB>
__global__ __launch_bounds__(512, MINB) void probe2_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ W,
                                                     float* __restrict__ out, int M, int N, int Kbytes, int stagger) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  if (stagger > 0 && blockIdx.x < 256 * MINB) {
    // first-round workgroups start k/8 of a tile period late (k from the block index), so that the chip's workgroups do not
    // all read and all write HBM at the same instants; later rounds inherit the phase of the workgroup they replace
    const unsigned long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long wait = (unsigned long)(((blockIdx.x >> 3) & 7) * stagger) >> 3;     // 10 ns units
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
  constexpr bool NO_DMA = ABL & 1, NO_RD = ABL & 2, NO_MMA = ABL & 4;
  constexpr int CB = 64, ROWS_ = BM_ + BN_, STAGE_ = ROWS_ * CB, RG_ = ROWS_ / 16, RGW_ = (RG_ + 7) / 8;
  constexpr int TM_ = BM_ / WGM / 16, TN_ = BN_ / WGN / 16;
  static_assert(WGM * WGN == 8 && ROWS_ % 16 == 0, "layout");
  const int NT = N / BN_;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / NT) * BM_, n0 = (bid % NT) * BN_;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WGM, wn = wave / WGM;
  const int lrow = lane >> 2, pc = (lane & 3) ^ ((4 - (lrow >> 2)) & 3);
  long r_base[RGW_];
  int my_count = 0;
#pragma unroll
  for (int i = 0; i < RGW_; ++i) {
    const int rg = wave + 8 * i;
    r_base[i] = 0;
    if (rg < RG_) {
      ++my_count;
      const int row = rg * 16 + lrow;
      r_base[i] = row < BM_ ? (long)(m0 + row) * Kbytes + pc * 16 : (long)(n0 + row - BM_) * Kbytes + pc * 16;
    }
  }
  my_count = __builtin_amdgcn_readfirstlane(my_count);
  auto issue_chunk = [&](int kc, int st) {
#pragma unroll
    for (int i = 0; i < RGW_; ++i) {
      const int rg = wave + 8 * i;
      if (rg < RG_) {
        const unsigned char* src = (rg * 16 >= BM_ ? W : A) + r_base[i] + (long)kc * CB;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dsm + st * STAGE_ + rg * 1024), 16, 0, 0);
      }
    }
  };
  auto wait_inflight = [&](int chunks) {   // wave-uniform: at most `chunks` whole chunks of this wave's DMAs stay in flight
    if (chunks <= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    if (my_count == RGW_) {
      if (chunks == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RGW_) : "memory");
      else if (chunks == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RGW_) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * RGW_) : "memory");
    } else {
      if (chunks == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RGW_ - 1) : "memory");
      else if (chunks == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (RGW_ - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (RGW_ - 1)) : "memory");
    }
  };
  f32x4 acc[TN_][TM_];
#pragma unroll
  for (int a = 0; a < TN_; ++a)
#pragma unroll
    for (int b = 0; b < TM_; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 dummy = u32x4{0u, 0u, 0u, 0u};
  const int nk = Kbytes / CB;
  const int l15 = lane & 15, g = lane >> 4;
  const int foff = l15 * CB + ((g ^ ((4 - (l15 >> 2)) & 3)) * 16);
  const int xrow = (wm * TM_ * 16) * CB + foff;
  const int wrow = (BM_ + wn * TN_ * 16) * CB + foff;
  for (int c = 0; c < NST - 1 && c < nk; ++c) issue_chunk(c, c);
  u32x4 wf[TN_], xf[TM_];
  if (NO_RD) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int a = 0; a < TN_; ++a) wf[a] = *(const u32x4*)(dsm + wrow + a * 16 * CB);
#pragma unroll
    for (int b = 0; b < TM_; ++b) xf[b] = *(const u32x4*)(dsm + xrow + b * 16 * CB);
  }
  // which waves share a SIMD depends on the number of waves dispatched before this workgroup; the production kernels read
  // HW_ID (pp_phase_half); here waves w and w + 4 are assumed to share one (8-wave workgroup, one per CU)
  const int half = wave >> 2;
  wait_inflight(NST - 2);
  __builtin_amdgcn_s_barrier();
  if (half) __builtin_amdgcn_s_barrier();
  int st = 0;
  for (int kc = 0; kc < nk; ++kc) {
    const unsigned char* Xs = dsm + st * STAGE_;
    if (!NO_RD) {
#pragma unroll
      for (int a = 0; a < TN_; ++a) wf[a] = *(const u32x4*)(Xs + wrow + a * 16 * CB);
#pragma unroll
      for (int b = 0; b < TM_; ++b) xf[b] = *(const u32x4*)(Xs + xrow + b * 16 * CB);
    }
    // chunk kc+1 of this wave must have landed before the barrier that precedes anyone's read of it
    if (kc + 1 < nk) wait_inflight(kc + NST - 2 < nk ? NST - 3 : 0);
    __builtin_amdgcn_s_barrier();
    if (!NO_DMA && kc + NST - 1 < nk) issue_chunk(kc + NST - 1, st == 0 ? NST - 1 : st - 1);   // refills the stage of chunk kc-1
    __builtin_amdgcn_s_setprio(1);
    if (!NO_MMA) {
#pragma unroll
      for (int a = 0; a < TN_; ++a)
#pragma unroll
        for (int b = 0; b < TM_; ++b) mma(acc[a][b], wf[a], xf[b]);
    } else {
#pragma unroll
      for (int a = 0; a < TN_; ++a) dummy ^= wf[a];
#pragma unroll
      for (int b = 0; b < TM_; ++b) dummy ^= xf[b];
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    st = st == NST - 1 ? 0 : st + 1;
  }
  if (!half) __builtin_amdgcn_s_barrier();
  if (ABL & 32) {
    __syncthreads();
    epilogue_emul<TM_, TN_>(acc, dsm + wave * (16 * (TN_ * 32 + 16)), (_Float16*)out, N, m0 + wm * TM_ * 16, n0 + wn * TN_ * 16, lane);
    return;
  }
  if (ABL & 64) {
    __syncthreads();
    epilogue_coop<BM_, BN_, WGM, TM_, TN_>(acc, dsm, (_Float16*)out, N, m0, n0, wm, wn * TN_ * 16, tid);
    return;
  }
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < TN_; ++a)
#pragma unroll
    for (int b = 0; b < TM_; ++b) s += acc[a][b];
  s[0] += __builtin_bit_cast(float, dummy[0] ^ dummy[1] ^ dummy[2] ^ dummy[3]);
  if (s[0] + s[1] + s[2] + s[3] == 1234.5f) out[blockIdx.x * 512 + tid] = s[0];
}

template <int BM_, int BN_, int WGM, int WGN, int NST, int ABL, int MINB = 1>
static double run2(const char* name, const unsigned char* A, const unsigned char* W, float* out, int M, int N, int Kb, int stagger = 0) {
  constexpr int LDS = NST * (BM_ + BN_) * 64;
  static_assert(LDS <= 160 * 1024 && (!(ABL & 64) || (BM_ / 2) * (BN_ * 2 + 16) <= LDS), "LDS");
  if (M % BM_ || N % BN_) { printf("%-58s (shape does not tile)\n", name); return 0.0; }
  auto k = probe2_kernel<BM_, BN_, WGM, WGN, NST, ABL, MINB>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  const int grid = (M / BM_) * (N / BN_);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS, 0, A, W, out, M, N, Kb, stagger);
  CHECK(hipDeviceSynchronize());
  const int reps = 10;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS, 0, A, W, out, M, N, Kb, stagger);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps;
  const double tf = 2.0 * M * N * (Kb / 2.0) / (us * 1e-6) / 1e12;
  printf("%-58s %9.1f us  (%6.0f TF-equivalent, %d workgroups, stagger %d0 ns)\n", name, us, tf, grid, stagger);
  return us * 256.0 * MINB / grid;   // us per tile period
}

template <int ABL, bool PP>
static void run(const char* name, const unsigned char* A, const unsigned char* W, float* out, int M, int N, int Kb, double clk_ghz) {
  constexpr int LDS = 3 * STAGE;
  auto k = probe_kernel<ABL, PP>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  const int grid = (M / BM) * (N / BN);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS, 0, A, W, out, M, N, Kb);
  CHECK(hipDeviceSynchronize());
  const int reps = 10;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS, 0, A, W, out, M, N, Kb);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps;
  const int nk = Kb / BKB;
  const double rounds = (double)grid / 256.0;                         // workgroups per CU, one resident at a time
  const double cyc_chunk = us * 1e-6 * clk_ghz * 1e9 / (rounds * nk);
  const double tf = 2.0 * M * N * (Kb / 2.0) / (us * 1e-6) / 1e12;
  printf("%-44s %9.1f us  %7.0f cyc/chunk/WG  (%6.0f TF-equivalent; MFMA floor 1280 cyc/chunk)\n", name, us, cyc_chunk, tf);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 262144, N = argc > 2 ? atoi(argv[2]) : 320, Kb = argc > 3 ? atoi(argv[3]) : 2560;
  const double clk = argc > 4 ? atof(argv[4]) : 2.4;
  unsigned char *A, *W;
  float* out;
  CHECK(hipMalloc(&A, (size_t)M * Kb));
  CHECK(hipMalloc(&W, (size_t)N * Kb));
  CHECK(hipMalloc(&out, (size_t)M * N * 2 + (size_t)(M / 128) * (N / 64) * 512 * 4));
  CHECK(hipMemset(A, 0x3c, (size_t)M * Kb));
  CHECK(hipMemset(W, 0x2c, (size_t)N * Kb));
  printf("loop probe: M=%d N=%d K=%d (fp16), tile 256x160, %d workgroups, %d chunks each; clock assumed %.2f GHz\n", M, N, Kb / 2,
         (M / BM) * (N / BN), Kb / BKB, clk);
  // with the epilogue emulation (bit 5): what a whole linear costs apart from bias / residual
  run<16 + 32, false>("128B 256x160 lock-step 1 WG/CU + epilogue", A, W, out, M, N, Kb, clk);
  run<16 + 32, true>("128B 256x160 ping-pong 1 WG/CU + epilogue", A, W, out, M, N, Kb, clk);
  run<16, true>("128B 256x160 ping-pong 1 WG/CU, no epilogue", A, W, out, M, N, Kb, clk);
  run2<256, 320, 2, 4, 4, 32>("pp64 256x320 (2x4 of 128x80) 4 st + epilogue", A, W, out, M, N, Kb);
  run2<256, 320, 2, 4, 4, 0>("pp64 256x320 (2x4 of 128x80) 4 st, no epilogue", A, W, out, M, N, Kb);
  {
    const double per = run2<256, 320, 2, 4, 4, 64>("pp64 256x320 (2x4 of 128x80) 4 st + row epilogue", A, W, out, M, N, Kb);
    run2<256, 320, 2, 4, 4, 64>("  ... staggered by 1 tile period", A, W, out, M, N, Kb, (int)(per * 100));
    run2<256, 320, 2, 4, 4, 64>("  ... staggered by 1/2 tile period", A, W, out, M, N, Kb, (int)(per * 50));
    run2<256, 320, 2, 4, 4, 32>("  ... per-wave epilogue, staggered by 1 period", A, W, out, M, N, Kb, (int)(per * 100));
  }
  run2<256, 320, 4, 2, 4, 64>("pp64 256x320 (4x2 of 64x160) 4 st + row epilogue", A, W, out, M, N, Kb);
  run2<256, 160, 4, 2, 6, 64, 1>("pp64 256x160 (4x2 of 64x80) 6 st, 1 WG/CU + row epilogue", A, W, out, M, N, Kb);
  {
    const double per = run2<256, 160, 4, 2, 3, 64, 2>("pp64 256x160 (4x2 of 64x80) 3 st, 2 WG/CU + row epilogue", A, W, out, M, N, Kb);
    run2<256, 160, 4, 2, 3, 64, 2>("  ... staggered by 1 tile period", A, W, out, M, N, Kb, (int)(per * 100));
  }
  run2<256, 128, 4, 2, 3, 32, 2>("pp64 256x128 (4x2 of 64x64) 3 st, 2 WG/CU + epilogue", A, W, out, M, N, Kb);
  run2<256, 128, 4, 2, 3, 0, 2>("pp64 256x128 (4x2 of 64x64) 3 st, 2 WG/CU, no epilogue", A, W, out, M, N, Kb);
  run2<256, 128, 4, 2, 6, 32, 1>("pp64 256x128 (4x2 of 64x64) 6 st, 1 WG/CU + epilogue", A, W, out, M, N, Kb);
  run2<128, 128, 2, 4, 3, 32, 3>("pp64 128x128 (2x4 of 64x32) 3 st, 3 WG/CU + epilogue", A, W, out, M, N, Kb);
  run2<256, 160, 4, 2, 4, 32, 1>("pp64 256x160 (4x2 of 64x80) 4 st, 1 WG/CU + epilogue", A, W, out, M, N, Kb);
  run2<256, 160, 4, 2, 3, 32, 2>("pp64 256x160 (4x2 of 64x80) 3 st, 2 WG/CU + epilogue", A, W, out, M, N, Kb);
  run2<256, 160, 4, 2, 3, 0, 2>("pp64 256x160 (4x2 of 64x80) 3 st, 2 WG/CU, no epilogue", A, W, out, M, N, Kb);
  return 0;
}
