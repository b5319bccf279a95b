// Persistent wide LDS-DMA GEMM for the large LINEAR problems (round 2): the 256 x BN / 8-wave / 3-stage structure of
// gemm_dma.hip, but ONE workgroup per CU walks many tiles and the operand stream never stops at a tile boundary.
//
// Why: per-op profiles put the long-K convs (360 k-chunks per tile) at 1.25 PFLOP/s and the transformer linears (K = 640 ..
// 5120, i.e. 10 .. 80 chunks per tile) at 0.6 .. 0.9 -- same main loop, the difference is the per-tile fixed cost of a
// one-workgroup-per-CU launch: the DMA latency of the first chunks (1-2 us with nothing to overlap), the epilogue, and the
// tail of the last workgroup wave.  Here
//   * the items (tile, k-chunk) form one flat sequence per workgroup; chunk DMAs are issued two items ahead ACROSS tile
//     boundaries, so the next tile's first two chunks land while the current tile finishes and runs its epilogue;
//   * the epilogue parks 16 rows per wave at a time in the ONE stage that is free at that point (the last chunk's), the
//     other two stages already belong to the next tile;
//   * tiles are dealt so that every XCD walks a contiguous tile range (N-tiles of an M-panel share an L2), 32 workgroups
//     per XCD, G = 256 workgroups in total (fewer for small problems).
// vmcnt bookkeeping: LDS-DMAs, the epilogue's residual loads and its stores retire in order on one counter.  At the first
// two items of a tile the previous tile's S store instructions sit between the chunk being waited for and the younger
// chunk that may stay in flight, so the allowance there is my_count + S.  S must be exact (an allowance that is too big
// would let a chunk be read before it landed): the 16-row epilogue issues its stores unpredicated and the host only
// selects this kernel when M % 256 == 0.
#include <cstdlib>

#include "common.h"
#include "gemm_device.h"

namespace tango {

// the same for counts up to 40 (persistent GEMM: DMA pieces + the previous tile's epilogue stores); larger counts wait for 40
__device__ __forceinline__ void wait_vmcnt_upto40(const int n) {
#define TANGO_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n < 0 ? 0 : (n > 40 ? 40 : n)) {
    TANGO_VMC(0) TANGO_VMC(1) TANGO_VMC(2) TANGO_VMC(3) TANGO_VMC(4) TANGO_VMC(5) TANGO_VMC(6) TANGO_VMC(7) TANGO_VMC(8) TANGO_VMC(9)
    TANGO_VMC(10) TANGO_VMC(11) TANGO_VMC(12) TANGO_VMC(13) TANGO_VMC(14) TANGO_VMC(15) TANGO_VMC(16) TANGO_VMC(17) TANGO_VMC(18) TANGO_VMC(19)
    TANGO_VMC(20) TANGO_VMC(21) TANGO_VMC(22) TANGO_VMC(23) TANGO_VMC(24) TANGO_VMC(25) TANGO_VMC(26) TANGO_VMC(27) TANGO_VMC(28) TANGO_VMC(29)
    TANGO_VMC(30) TANGO_VMC(31) TANGO_VMC(32) TANGO_VMC(33) TANGO_VMC(34) TANGO_VMC(35) TANGO_VMC(36) TANGO_VMC(37) TANGO_VMC(38) TANGO_VMC(39)
    TANGO_VMC(40)
  }
#undef TANGO_VMC
}


// (moved here from gemm_device.h with the kernel: only the persistent GEMM used it)
// 16-rows-per-pass variant for the persistent GEMM (gemm_pers.hip): the staging area is ONE operand stage (the other two
// hold the next tile's first chunks, already in flight), so each wave parks 16 rows x (TN*16) fp32 at a time.  Same fp32
// arithmetic and order as gemm_epilogue_staged -> bit-identical results.  Returns nothing; the number of global store
// instructions a wave issues is TM * ceil(16 * ppr / 64) (ppr = 16-byte pieces per output row) -- the caller's vmcnt
// bookkeeping depends on it, so rows are NOT predicated here (the caller guarantees M % 256 == 0).
template <typename T, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_staged16(const GemmParams& p, f32x4 (&acc)[TN][TM], const int m_base, const int n_base,
                                                       const int lane, unsigned char* stage) {
  constexpr int EPV = 16 / (int)sizeof(T);
  constexpr int WN = TN * 16;
  constexpr int PITCH = WN * 4 + 16;
  constexpr int PPR = WN / EPV;
  const int g4 = (lane >> 4) * 4;
  const bool geglu = p.epi == EPI_GEGLU;
  const float* bias = p.bias;
  const float* bias2 = p.bias2 ? p.bias2 + (int64_t)(p.step_ptr ? *p.step_ptr : 0) * p.bias2_stride : nullptr;
  float cb[TN][4];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n_base + a * 16 + g4 + r;
      float c = 0.f;
      if (bias) c = bias[n];
      if (bias2) c += bias2[n];
      cb[a][r] = c;
    }
  const T* Rb = (const T*)p.R;
  T* Ob = (T*)p.out;
  const int ppr = geglu ? PPR / 2 : PPR;
  // nit wave passes over the 16 x ppr pieces of a 16-row block (wave-uniform count: every executed pass issues exactly one
  // store instruction, partially masked in the last one).  vmcnt retires in order and counts stores, so a residual load
  // issued BEHIND a block's stores would wait for those stores to complete: the residual pieces of block b+1 are fetched
  // before block b's stores are issued (double-buffered in registers).
  constexpr int NIT = (16 * PPR + 63) / 64;
  const int nit = (16 * ppr + 63) >> 6;
  const int ncol0 = geglu ? (n_base >> 1) : n_base;
  T rv[2][NIT][EPV];
  auto fetch_res = [&](const int b, T (&dst)[NIT][EPV]) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = lane + it * 64;
      if (it < nit && idx < 16 * ppr) {
        const int rl = idx / ppr, pcs = idx - rl * ppr;
        __builtin_memcpy(dst[it], Rb + (int64_t)(m_base + b * 16 + rl) * p.ldr + ncol0 + pcs * EPV, 16);
      }
    }
  };
  if (Rb) fetch_res(0, rv[0]);
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int row_l = lane & 15;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      if (geglu && (a & 1)) continue;
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r] * p.alpha + cb[a][r];
      if (geglu) {
        const int ag = a + 1 < TN ? a + 1 : a;
        float gt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gt[r] = acc[ag][b][r] * p.alpha + cb[ag][r];
        glu_gate4<T>(v, gt, p.glu_tanh);
      } else if (p.e_act != ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.e_act, p.e_slope);
      }
      *(f32x4*)(stage + row_l * PITCH + ((geglu ? (a >> 1) : a) * 16 + g4) * 4) = v;
    }
    if (Rb && b + 1 < TM) fetch_res(b + 1, rv[(b + 1) & 1]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it < nit) {                                  // wave-uniform
        const int idx = lane + it * 64;
        if (idx < 16 * ppr) {
          const int rl = idx / ppr, pcs = idx - rl * ppr;
          const int m = m_base + b * 16 + rl;
          float f[EPV];
#pragma unroll
          for (int q = 0; q < EPV / 4; ++q) {
            const f32x4 t = *(const f32x4*)(stage + rl * PITCH + (pcs * EPV + q * 4) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) f[q * 4 + r] = t[r];
          }
          if (Rb) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) f[e] += to_f(rv[b & 1][it][e]);
          }
          if (p.out_scale != 1.f) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) f[e] *= p.out_scale;
          }
          T tv[EPV];
#pragma unroll
          for (int e = 0; e < EPV; ++e) tv[e] = from_f<T>(f[e]);
          __builtin_memcpy(Ob + (int64_t)m * p.ldo + ncol0 + pcs * EPV, tv, 16);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}



template <typename T, int BN>
__global__ __launch_bounds__(512, 2) void gemm_pers_kernel(const GemmParams p, const unsigned char* zero_page, const int total_tiles) {
  constexpr int EPV = 16 / (int)sizeof(T);
  constexpr int BM = 256, BKB = 128;
  constexpr int ROWS = BM + BN;
  constexpr int STAGE = ROWS * BKB;
  constexpr int RG = ROWS / 8;
  constexpr int RGW = (RG + 7) / 8;
  constexpr int WMR = 64, WNR = BN / 2;
  constexpr int TM = 4, TN = WNR / 16;
  constexpr int PPRW = WNR / EPV;                 // 16-byte pieces per staged output row
  static_assert(8 * 16 * (WNR * 4 + 16) <= STAGE, "the 16-row staging area of 8 waves must fit in one operand stage");
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];   // 3 stages

  const int NT = p.N / BN;
  const unsigned char* Ab = (const unsigned char*)p.A;
  const unsigned char* Wb = (const unsigned char*)p.W;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int lrow = lane >> 3, slot = lane & 7;
  const int pc = slot ^ lrow;

  // this workgroup's tiles: XCD x (= blockIdx % 8, observed dispatch; speed only) owns the contiguous range [t0, t1),
  // its gridDim/8 workgroups walk it with that stride
  const int G8 = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int q = total_tiles >> 3, r = total_tiles & 7;
  const int t0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int t1 = t0 + q + (xcd < r ? 1 : 0);
  int tile = t0 + j;
  if (tile >= t1) return;

  int my_count = 0;
#pragma unroll
  for (int i = 0; i < RGW; ++i) my_count += (wave + 8 * i < RG) ? 1 : 0;
  my_count = __builtin_amdgcn_readfirstlane(my_count);

  // per-lane DMA source rows of a tile: byte offset of (row, piece) inside A (rows < 256) or W, -1 = zero page
  auto rows_of = [&](const int t, int64_t (&rb)[RGW]) {
    const int m0 = (t / NT) * BM, n0 = (t % NT) * BN;
#pragma unroll
    for (int i = 0; i < RGW; ++i) {
      const int rg = wave + 8 * i;
      rb[i] = -1;
      if (rg < RG) {
        const int row = rg * 8 + lrow;
        if (row < BM) {
          const int m = m0 + row;
          if (m < p.M) rb[i] = ((int64_t)m * p.lda + pc * EPV) * (int64_t)sizeof(T);
        } else {
          const int n = n0 + row - BM;
          if (n < p.N) rb[i] = ((int64_t)n * p.Kp + pc * EPV) * (int64_t)sizeof(T);
        }
      }
    }
  };
  auto issue = [&](const int64_t (&rb)[RGW], const int kc, const int st) {
#pragma unroll
    for (int i = 0; i < RGW; ++i) {
      const int rg = wave + 8 * i;
      if (rg < RG) {                                   // wave-uniform
        const unsigned char* src = zero_page;
        if (rb[i] >= 0) src = (rg * 8 >= BM ? Wb : Ab) + rb[i] + (int64_t)kc * BKB;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dsm + st * STAGE + rg * 1024), 16, 0, 0);
      }
    }
  };

  const int nk = p.K / (BKB / (int)sizeof(T));          // >= 2 (host)
  int koff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) koff[ks] = (((ks * 4 + (lane >> 4)) ^ (lane & 7)) * 16);
  const int xrow = (wm * WMR + (lane & 15)) * BKB;
  const int wrow = (BM + wn * WNR + (lane & 15)) * BKB;
  // store instructions one wave issues per tile epilogue (see gemm_epilogue_staged16): TM passes x ceil(16 * ppr / 64)
  constexpr int SF = TM * ((16 * PPRW + 63) / 64);          // plain epilogue
  constexpr int SG = TM * ((16 * (PPRW / 2) + 63) / 64);    // GEGLU (half-width rows)
  static_assert(RGW + SF <= 60, "vmcnt immediate");
  const bool geglu = p.epi == EPI_GEGLU;

  int64_t rc[RGW], rn[RGW];
  rows_of(tile, rc);
  issue(rc, 0, 0);
  issue(rc, 1, 1);
  int st = 0;                 // stage of the current item (global item index mod 3)
  bool first_tile = true;
  for (;;) {
    const int tnext = tile + G8;
    const bool has_next = tnext < t1;
    if (has_next) rows_of(tnext, rn);
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < nk; ++kc) {
      // chunk kc of this tile must have landed; the next item's DMAs (issued one item ago) may stay in flight, and so may
      // the previous tile's epilogue stores during the first two items
      const bool next_item = (kc + 1 < nk) || has_next;
      const bool full = my_count == RGW;
      if (!next_item) {
        wait_vmcnt_lit<0>();
      } else if (!first_tile && kc < 2) {
        if (geglu) { if (full) wait_vmcnt_lit<RGW + SG>(); else wait_vmcnt_lit<RGW - 1 + SG>(); }
        else { if (full) wait_vmcnt_lit<RGW + SF>(); else wait_vmcnt_lit<RGW - 1 + SF>(); }
      } else {
        if (full) wait_vmcnt_lit<RGW>(); else wait_vmcnt_lit<RGW - 1>();
      }
      pp_barrier();
      {   // item + 2 goes into the stage of item - 1 (free: every wave passed the barrier above after reading it)
        const int st2 = st == 0 ? 2 : st - 1;
        if (kc + 2 < nk) issue(rc, kc + 2, st2);
        else if (has_next) issue(rn, kc + 2 - nk, st2);
      }
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
    // every wave is past its last fragment read of the last chunk's stage: it becomes the staging area
    pp_barrier();
    {
      const int st_last = st == 0 ? 2 : st - 1;
      const int m0 = (tile / NT) * BM, n0 = (tile % NT) * BN;
      gemm_epilogue_staged16<T, TM, TN>(p, acc, m0 + wm * WMR, n0 + wn * WNR, lane,
                                        dsm + st_last * STAGE + wave * (16 * (WNR * 4 + 16)));
    }
    if (!has_next) break;
    tile = tnext;
    first_tile = false;
#pragma unroll
    for (int i = 0; i < RGW; ++i) rc[i] = rn[i];
  }
}

// which problems: plain / GEGLU epilogue into T, M a multiple of 256, whole N tiles, at least two 128-byte k-chunks, and
// enough tiles that every CU gets several (otherwise the one-shot kernel's dispatch is already balanced).
// Opt-in (TANGO_PERS_GEMM=1): measured on MI355X it is ~10 % slower than the one-shot kernel on every UNet linear shape
// (profiles/r2_unet_ops_pers_vs_oneshot.txt) - the hardware dispatcher already overlaps one workgroup's epilogue with the
// next co-resident workgroup's prologue, and the persistent loop adds VGPR pressure.  Kept for the repeat-run tests.
bool gemm_pers_ok(int dtype, const GemmParams& p) {
  static const bool on = getenv("TANGO_PERS_GEMM") != nullptr;
  if (!on) return false;
  const int esz = dtype == DT_F32 ? 4 : 2;
  const bool linear = p.mode == GATHER_1D && p.taps == 1 && p.rows_pb == p.M && p.in_mul == 1 && p.in_off == 0 && p.out_mul == 1 &&
                      p.out_off == 0 && p.Lin >= p.M;
  if (!linear || p.batch != 1 || p.splitk > 1 || p.a_act != ACT_NONE || p.bias_rows || p.out_f32 || p.ln_fold) return false;
  if (p.epi != EPI_NONE && p.epi != EPI_GEGLU) return false;
  if (p.epi == EPI_GEGLU && p.e_act != ACT_NONE) return false;
  if ((p.K * esz) % 128 != 0 || p.K * esz < 256 || p.M % 256 != 0) return false;
  const int bn = (p.epi == EPI_GEGLU || p.N % 160 != 0) ? 128 : 160;
  if (p.N % bn != 0) return false;
  const int epv = 16 / esz;
  if (p.ldo % epv != 0 || ((uintptr_t)p.out & 15) || (p.R && (p.ldr % epv != 0 || ((uintptr_t)p.R & 15)))) return false;
  if ((p.lda * esz) % 16 != 0 || (p.Kp * esz) % 16 != 0) return false;
  static const bool force = getenv("TANGO_FORCE_DMA_GEMM") != nullptr;   // tests: exercise this kernel on small shapes
  const long tiles = (long)(p.M / 256) * (p.N / bn);
  return force || tiles >= 512;
}

template <typename T, int BN>
static int launch_pers_cfg(const GemmParams& p, const unsigned char* zero_page, hipStream_t s) {
  constexpr int LDS = 3 * (256 + BN) * 128;
  auto kfn = gemm_pers_kernel<T, BN>;
  static bool attr_set = false;
  if (!attr_set) {
    TANGO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const int tiles = (p.M / 256) * (p.N / BN);
  int G = tiles < 256 ? (tiles / 8) * 8 : 256;      // a multiple of 8: the same number of workgroups on every XCD
  if (G < 8) G = 8;
  hipLaunchKernelGGL(kfn, dim3((unsigned)G), dim3(512), LDS, s, p, zero_page, tiles);
  TANGO_HIP(hipGetLastError());
  return 0;
}

template <typename T>
static int launch_pers(const GemmParams& p, const unsigned char* zero_page, hipStream_t s) {
  const bool bn128 = (p.epi == EPI_GEGLU || p.N % 160 != 0);
  return bn128 ? launch_pers_cfg<T, 128>(p, zero_page, s) : launch_pers_cfg<T, 160>(p, zero_page, s);
}

int launch_gemm_pers(int dtype, const GemmParams& p, const unsigned char* zero_page, hipStream_t s) {
  if (!zero_page) TANGO_FAIL("gemm_pers: gemm_init() was not called");
  switch (dtype) {
    case DT_F32: return launch_pers<float>(p, zero_page, s);
    case DT_F16: return launch_pers<f16>(p, zero_page, s);
    case DT_BF16: return launch_pers<bf16>(p, zero_page, s);
  }
  TANGO_FAIL("gemm_pers: bad dtype");
}

}  // namespace tango
