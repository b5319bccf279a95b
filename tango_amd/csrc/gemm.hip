// Gather-GEMM for gfx950: implicit-GEMM conv2d(3x3)/conv1d/transposed-conv1d-phase/linear on MFMA.
//
// Orientation: the MFMA "A" operand is the WEIGHT tile (rows n), the "B" operand is the ACTIVATION
// tile (cols m), so each lane's accumulator quad is 4 consecutive output channels of one output row
// -> vector epilogue (bias/act/residual/GEGLU) and 8/16-byte stores.
// 16x16 MFMA tiles; k is consumed in 64-byte groups per operand row: one ds_read_b128 per fragment
// feeds one v_mfma_f32_16x16x32_{f16,bf16} or four v_mfma_f32_16x16x4_f32 (k order inside a group
// is a fixed permutation shared by both operands, which leaves the dot product unchanged).
//
// v2 structure (round 1): 128-byte tile rows, XOR-swizzled 16-byte pieces (piece ^ (row & 7)) so every
// ds_read_b128 lane group hits 16 distinct slots; two LDS stages, ONE barrier per k-chunk: chunk k+1
// is fetched HBM->VGPR while chunk k is multiplied, then written to the other stage.
//
// Reference ops this kernel replaces (ATen dispatches): convolution (66/UNet step),
// mm/addmm (216), and the VAE / HiFi-GAN convolution + conv_transpose1d calls.
#include <cstdlib>

#include <map>
#include <mutex>
#include <utility>

#include "common.h"
#include "gemm_device.h"
#include "tuning.h"

namespace tango {

int ensure_dyn_lds(const void* kfn, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> opted;
  int dev = 0;
  TANGO_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  int& have = opted[std::make_pair(kfn, dev)];
  if (bytes > have) {
    TANGO_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have = bytes;
  }
  return 0;
}



template <typename T, int BM, int BN, int BKB, int WM, int WN, int MODE>
__global__ __launch_bounds__(256, (BKB == 64 ? 3 : 2)) void gemm_kernel(const GemmParams p) {
  constexpr int EPV = 16 / (int)sizeof(T);     // elements per 16-byte vector
  constexpr int BK = BKB / (int)sizeof(T);     // k elements per chunk
  constexpr int NKG = BKB / 64;                // 64-byte k groups per chunk
  constexpr int PPR = BKB / 16;                // 16-byte pieces per tile row
  constexpr int RPP = 256 / PPR;               // tile rows staged per pass
  constexpr int AP = (BM + RPP - 1) / RPP;
  constexpr int BP = (BN + RPP - 1) / RPP;
  constexpr bool SWZ = (BKB == 128);           // XOR swizzle for 128-byte rows, +16 B padding otherwise
  constexpr int LDSR = SWZ ? 128 : BKB + 16;
  constexpr int STAGE = (BM + BN) * LDSR;
  constexpr int WMR = BM / WM, WNR = BN / WN;
  constexpr int TM = WMR / 16, TN = WNR / 16;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(WMR % 16 == 0 && WNR % 16 == 0, "tile");

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int MT = (p.M + BM - 1) / BM, NT = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {  // bijective XCD-aware remap: each XCD (bid % 8) walks a contiguous range of logical tiles so
     // the N-tiles of one M-panel (and neighbouring M-panels of one weight panel) share an L2.
    const int nblk = MT * NT;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / NT) * BM, n0 = (bid % NT) * BN;

  const unsigned char* Ab = (const unsigned char*)p.A + (int64_t)blockIdx.z * p.sA * (int64_t)sizeof(T);
  const unsigned char* Wb = (const unsigned char*)p.W + (int64_t)blockIdx.z * p.sW * (int64_t)sizeof(T);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int prow = tid / PPR, pcol = tid % PPR;
  // staging position of this thread's 16-byte piece inside a tile row (row & 7 == prow & 7: RPP % 8 == 0)
  const int spos = SWZ ? ((pcol ^ (prow & 7)) * 16) : pcol * 16;

  // ---- per-thread gather rows ----
  int64_t a_base[AP];          // LINEAR: byte offset of the row; conv: batch base ROW index (or -1)
  int a_c0[AP], a_c1[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int rl = i * RPP + prow;
    const int m = m0 + rl;
    a_base[i] = -1; a_c0[i] = 0; a_c1[i] = 0;
    if (rl < BM && m < p.M) {
      if (MODE == MODE_LINEAR) {
        a_base[i] = ((int64_t)m * p.lda + pcol * EPV) * (int64_t)sizeof(T);
      } else if (MODE == MODE_CONV2D) {
        const int hw = p.H * p.Wd;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / p.Wd, x = rem - y * p.Wd;
        a_base[i] = (int64_t)b * p.Hin * p.Win;
        a_c0[i] = y * p.stride - p.pad;
        a_c1[i] = x * p.stride - p.pad;
      } else {
        const int b = m / p.rows_pb, q = m - b * p.rows_pb;
        a_base[i] = (int64_t)b * p.Lin;
        a_c0[i] = q * p.in_mul + p.in_off;
      }
    }
  }
  const unsigned char* w_ptr[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int rl = i * RPP + prow;
    const int n = n0 + rl;
    w_ptr[i] = (rl < BN && n < p.N) ? Wb + ((int64_t)n * p.Kp + pcol * EPV) * (int64_t)sizeof(T) : nullptr;
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ar[AP], br[BP];
  int kc_begin = 0, nk = p.K / BK;
  if (p.splitk > 1) {
    const int per = (nk + p.splitk - 1) / p.splitk;
    kc_begin = blockIdx.y * per;
    nk = min(nk, kc_begin + per);
  }

  auto load_chunk = [&](int kc) {
    const int k0 = kc * BK;
    int tap = 0, cc = k0, dy = 0, dx = 0;
    if (MODE != MODE_LINEAR) {
      tap = k0 / p.Cin;
      cc = k0 - tap * p.Cin;
      if (MODE == MODE_CONV2D) { dy = tap / 3; dx = tap - dy * 3; }
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (a_base[i] >= 0) {
        if (MODE == MODE_LINEAR) {
          v = *(const u32x4*)(Ab + a_base[i] + (int64_t)k0 * (int64_t)sizeof(T));
          if (p.a_act != ACT_NONE) v = act_vec<T>(v, p.a_act, p.a_slope);
        } else {
          int64_t srow; bool ok;
          if (MODE == MODE_CONV2D) {
            const int iy = a_c0[i] + dy, ix = a_c1[i] + dx;
            ok = (unsigned)iy < (unsigned)(p.Hin << p.ups) && (unsigned)ix < (unsigned)(p.Win << p.ups);
            srow = a_base[i] + (int64_t)(iy >> p.ups) * p.Win + (ix >> p.ups);
          } else {
            const int idx = a_c0[i] + tap * p.tap_step;
            ok = (unsigned)idx < (unsigned)p.Lin;
            srow = a_base[i] + idx;
          }
          if (ok) {
            v = *(const u32x4*)(Ab + (srow * p.lda + cc + pcol * EPV) * (int64_t)sizeof(T));
            if (p.a_act != ACT_NONE) v = act_vec<T>(v, p.a_act, p.a_slope);
          }
        }
      }
      ar[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (w_ptr[i]) v = *(const u32x4*)(w_ptr[i] + (int64_t)k0 * (int64_t)sizeof(T));
      br[i] = v;
    }
  };
  auto store_stage = [&](int st) {
    unsigned char* Xs = smem + st * STAGE;
    unsigned char* Ws = Xs + BM * LDSR;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int rl = i * RPP + prow;
      if (rl < BM) *(u32x4*)(Xs + rl * LDSR + spos) = ar[i];
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int rl = i * RPP + prow;
      if (rl < BN) *(u32x4*)(Ws + rl * LDSR + spos) = br[i];
    }
  };

  // fragment read offsets inside a row: (row & 7) == (lane & 7) because tile bases are multiples of 16
  int koff[NKG];
#pragma unroll
  for (int ks = 0; ks < NKG; ++ks)
    koff[ks] = SWZ ? (((ks * 4 + (lane >> 4)) ^ (lane & 7)) * 16) : ks * 64 + (lane >> 4) * 16;
  const int xrow = (wm * WMR + (lane & 15)) * LDSR;
  const int wrow = (wn * WNR + (lane & 15)) * LDSR;

  if (kc_begin < nk) {
    load_chunk(kc_begin);
    store_stage(kc_begin & 1);
  }
  __syncthreads();
  for (int kc = kc_begin; kc < nk; ++kc) {
    const bool more = kc + 1 < nk;
    if (more) load_chunk(kc + 1);
    const unsigned char* Xs = smem + (kc & 1) * STAGE;
    const unsigned char* Ws = Xs + BM * LDSR;
#pragma unroll
    for (int ks = 0; ks < NKG; ++ks) {
      u32x4 wf[TN], xf[TM];
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[a] = *(const u32x4*)(Ws + wrow + a * 16 * LDSR + koff[ks]);
#pragma unroll
      for (int b = 0; b < TM; ++b) xf[b] = *(const u32x4*)(Xs + xrow + b * 16 * LDSR + koff[ks]);
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
    }
    if (more) store_stage((kc + 1) & 1);
    __syncthreads();
  }

  gemm_epilogue<T, TM, TN, MODE>(p, acc, m0 + wm * WMR, n0 + wn * WNR, lane, (int)blockIdx.z, (int)blockIdx.y);
}


// sums the split-K partials in split order and applies the epilogue (bias, per-step bias, activation, residual)
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = p.N / 4;
  if (idx >= (int64_t)p.M * n4) return;
  const int64_t m = idx / n4;
  const int n = (int)(idx - m * n4) * 4;
  f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < p.splitk; ++s) a += *(const f32x4*)(p.ws + ((int64_t)s * p.M + m) * p.N + n);
  const float* bias2 = p.bias2 ? p.bias2 + (int64_t)(p.step_ptr ? *p.step_ptr : 0) * p.bias2_stride : nullptr;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = a[r] * p.alpha;
    if (p.bias) x += p.bias[n + r];
    if (bias2) x += bias2[n + r];
    if (p.e_act != ACT_NONE) x = apply_act(x, p.e_act, p.e_slope);
    v[r] = x;
  }
  if (p.R) {
    T rv[4];
    __builtin_memcpy(rv, (const T*)p.R + m * p.ldr + n, 4 * sizeof(T));
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += to_f(rv[r]);
  }
  if (p.out_f32) {
    *(f32x4*)((float*)p.out + m * p.ldo + n) = f32x4{v[0], v[1], v[2], v[3]};
  } else {
    T tv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tv[r] = from_f<T>(v[r]);
    __builtin_memcpy((T*)p.out + m * p.ldo + n, tv, 4 * sizeof(T));
  }
}

int launch_splitk_reduce(int dtype, const GemmParams& p, hipStream_t s) {
  const int64_t work = (int64_t)p.M * (p.N / 4);
  const dim3 grid((unsigned)((work + 255) / 256));
  switch (dtype) {
    case DT_F32: hipLaunchKernelGGL((splitk_reduce_kernel<float>), grid, dim3(256), 0, s, p); break;
    case DT_F16: hipLaunchKernelGGL((splitk_reduce_kernel<f16>), grid, dim3(256), 0, s, p); break;
    case DT_BF16: hipLaunchKernelGGL((splitk_reduce_kernel<bf16>), grid, dim3(256), 0, s, p); break;
    default: TANGO_FAIL("splitk_reduce: bad dtype");
  }
  TANGO_HIP(hipGetLastError());
  return 0;
}

template <typename T, int BM, int BN, int BKB, int WM, int WN, int MODE>
static int launch_cfg(const GemmParams& p, hipStream_t s) {
  const int MT = (p.M + BM - 1) / BM, NT = (p.N + BN - 1) / BN;
  if (p.splitk > 1) {
    if (!p.ws) TANGO_FAIL("gemm: split-K needs a workspace");
    dim3 grid((unsigned)(MT * NT), (unsigned)p.splitk, 1);
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, BKB, WM, WN, MODE>), grid, dim3(256), 0, s, p);
    const int64_t work = (int64_t)p.M * (p.N / 4);
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, p);
    TANGO_HIP(hipGetLastError());
    return 0;
  }
  dim3 grid((unsigned)(MT * NT), 1, (unsigned)p.batch);
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, BKB, WM, WN, MODE>), grid, dim3(256), 0, s, p);
  TANGO_HIP(hipGetLastError());
  return 0;
}

// Small-M linears (B = 1: M = 8192 / 2048 / 512 rows at levels 0 / 1 / 2; B = 8's level 2): the 128-row tilings leave most of the
// 256 CUs idle and the round-1 answer -- split-K into fp32 partials + a reduce launch -- costs a second kernel and a round trip
// per op (~22 us for a 1.7-GFLOP GEMM, profiles/r3_c1_unet_step_per_op_fp16_b1.txt).  64 x 64 tiles fill the chip with ONE
// launch when K is short enough that a workgroup's serial k-loop stays a few microseconds.
static bool small_tile_linear(const GemmParams& p, int esz) {
  if (tuning().no_small_tile || p.mode != GATHER_1D || p.epi != EPI_NONE || p.batch != 1 || p.splitk > 1 || p.bias_rows) return false;
  if (!(p.taps == 1 && p.rows_pb == p.M && p.in_mul == 1 && p.in_off == 0 && p.out_mul == 1 && p.out_off == 0 && p.Lin >= p.M)) return false;
  if (p.N % 64 != 0 || (p.Cin * esz) % 128 != 0 || p.K > 1280) return false;
  const int bn = (p.N % 160 == 0) ? 160 : 128;
  const long big = (long)((p.M + 127) / 128) * ((p.N + bn - 1) / bn);
  const long small = (long)((p.M + 63) / 64) * (p.N / 64);
  // measured at B = 1 (profiles/r3_c2_small_batch_dispatch_ab.txt; x25 / x20 / x5 per step, ms): M=2048 N=640 K=640 0.549 -> 0.430,
  // M=8192 N=320 K=320 0.514 (streaming kernel) -> 0.301, M=8192 N=320 K=1280 0.173 -> 0.146; but M=512 N=1280 K=1280 0.540 -> 0.640
  // and M=2048 N=640 K=2560 0.156 -> 0.212: a long serial k-loop on few workgroups loses to split-K
  return big < 192 && (p.K <= 640 ? small >= 96 : small >= 512);
}

template <typename T, int BKB, int MODE>
static int launch_tile(const GemmParams& p, hipStream_t s) {
  if constexpr (MODE == MODE_LINEAR && BKB == 128) {
    if (small_tile_linear(p, (int)sizeof(T))) return launch_cfg<T, 64, 64, BKB, 2, 2, MODE>(p, s);
  }
  if (p.epi == EPI_GEGLU) {
    if (p.N % 32 != 0) TANGO_FAIL("GEGLU gemm needs N % 32 == 0");
    return launch_cfg<T, 128, 128, BKB, 2, 2, MODE>(p, s);
  }
  if (p.N % 160 == 0) return launch_cfg<T, 128, 160, BKB, 2, 2, MODE>(p, s);
  if (p.N >= 96) return launch_cfg<T, 128, 128, BKB, 2, 2, MODE>(p, s);
  if (p.N > 32) return launch_cfg<T, 128, 64, BKB, 2, 2, MODE>(p, s);
  if (p.N > 16) return launch_cfg<T, 256, 32, BKB, 4, 1, MODE>(p, s);
  return launch_cfg<T, 256, 16, BKB, 4, 1, MODE>(p, s);
}

template <typename T, int BKB>
static int launch_mode(const GemmParams& p, hipStream_t s) {
  if (p.mode == GATHER_2D) return launch_tile<T, BKB, MODE_CONV2D>(p, s);
  const bool linear = p.taps == 1 && p.rows_pb == p.M && p.in_mul == 1 && p.in_off == 0 && p.out_mul == 1 && p.out_off == 0 &&
                      p.Lin >= p.M;
  if (linear) return launch_tile<T, BKB, MODE_LINEAR>(p, s);
  return launch_tile<T, BKB, MODE_CONV1D>(p, s);
}

template <typename T>
static int launch_t(const GemmParams& p, hipStream_t s) {
  const int cb = p.Cin * (int)sizeof(T);
  if (p.M <= 0 || p.N <= 0) return 0;
  if (p.K % p.Cin != 0) TANGO_FAIL("gemm: K must be taps*Cin");
  if (p.a_act != ACT_NONE && p.a_act != ACT_LRELU && p.a_act != ACT_SILU) TANGO_FAIL("gemm: the operand prologue implements leaky-ReLU and SiLU only");
  if ((p.lda * (int64_t)sizeof(T)) % 16 != 0 || (p.Kp * (int64_t)sizeof(T)) % 16 != 0) TANGO_FAIL("gemm: lda/Kp must be 16-byte multiples");
  if (cb % 128 == 0) return launch_mode<T, 128>(p, s);
  if (cb % 64 == 0) return launch_mode<T, 64>(p, s);
  TANGO_FAIL("gemm: Cin*sizeof(T) must be a multiple of 64 bytes");
}

static unsigned char* g_zero_page = nullptr;

// one-time process-wide setup; must run OUTSIDE stream capture (hipMalloc / hipMemset are illegal while capturing)
int gemm_init() {
  if (!g_zero_page) {
    TANGO_HIP(hipMalloc((void**)&g_zero_page, 4096));
    TANGO_HIP(hipMemset(g_zero_page, 0, 4096));
  }
  return 0;
}

// Split-K policy: only plain-epilogue linear / conv2d problems whose 128x160 tiling leaves most CUs idle.
int gemm_pick_splitk(int dtype, const GemmParams& p) {
  if (p.batch != 1 || p.epi != EPI_NONE || p.bias_rows || p.out_scale != 1.f || (p.N & 3) || (p.ldo & 3) || (p.R && (p.ldr & 3)))
    return 1;
  if (p.mode == GATHER_1D && !(p.taps == 1 && p.rows_pb == p.M && p.in_mul == 1 && p.in_off == 0 && p.out_mul == 1 && p.out_off == 0))
    return 1;
  if (gemm_duo_ok(dtype, p) || linear_stream_ok(dtype, p)) return 1;
  if (small_tile_linear(p, dtype == DT_F32 ? 4 : 2)) return 1;      // one launch of 64 x 64 tiles instead (launch_tile)
  if (p.N <= 32 && conv_halo_ok(dtype, p)) return 1;                // narrow-output convs (conv_out) on the halo kernel's 256 x 32 tile from 256 tiles on (round 6)
  {
    const int sw = conv_wide_pick_splitk(dtype, p);    // e.g. 64 tiles of 256 x 320 -> 4 splits = one workgroup per CU
    // (linears: measured at M = 4096 -- 64 tiles x 4 splits -- the wide kernel is no faster than the 4-wave tiles' split-K,
    //  0.250 vs 0.212 ms for N = K = 1280 x5, so gemm_wide_pick_splitk() is not consulted here)
    if (sw > 1) return sw;
  }
  const int esz = dtype == DT_F32 ? 4 : 2;
  const int bk = ((p.Cin * esz) % 128 == 0) ? 128 / esz : 64 / esz;
  const int nk = p.K / bk;
  const int bn = (p.N % 160 == 0) ? 160 : (p.N >= 96 ? 128 : (p.N > 32 ? 64 : 32));
  const int bm = p.N > 32 ? 128 : 256;
  const int tiles = ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
  if (tiles >= 400) return 1;              // (round 3: no split from 256 or 128 tiles on was measured at B = 8 -- M=4096 N=1280 K=1280 x25 1.05 -> 1.44 ms)
  int s = (512 + tiles / 2) / tiles;   // aim for ~2 workgroups per CU
  if (s > nk / 3) s = nk / 3;             // keep >= 3 k-chunks per split (measured sweep, round 1 v15)
  if (s > 32) s = 32;
  return s < 2 ? 1 : s;
}

// Which kernel family takes this problem (one decision, used by the launcher AND by the per-op profile labels).
int gemm_route(int dtype, const GemmParams& p) {
  if (gemm_duo_ok(dtype, p)) return ROUTE_DUO;      // short-K linears: two co-resident workgroups hide each other's pro- and epilogues
  if (!p.ln_fold && gemm_wide_ok(dtype, p)) {
    // the 256 x 320 kernel beats the streaming kernel on plain (no folded LayerNorm) shapes once a row is >= 1280 bytes (round 2:
    // M=65536 N=640 K=640 x20 2.38 -> 1.87 ms), takes the transposed-V and short-row (conv_in im2col, K = 96) shapes the streaming
    // kernel has no instantiation for, and -- round 3, profiles/r3_c3_level0_plain_linear_kernel_ab.txt -- also wins the level-0
    // K = 320 rows: M=262144 N=320 K=320 x20 2.81 -> 2.33 ms, and at B = 8 (M=65536, 256 tiles) 0.94 -> 0.74 ms
    // (profiles/r3_c4_level0_ln_routing_ab.txt).  gemm_wide_ok() already demands >= 224 tiles.
    return ROUTE_WIDE;
  }
  if (p.ln_fold && gemm_wide_ok(dtype, p)) return ROUTE_WIDE;
  if (linear_stream_ok(dtype, p)) return ROUTE_STREAM;
  if (p.ln_fold) return ROUTE_NONE;
  if (conv_wide_ok(dtype, p)) return ROUTE_CONV_WIDE;
  if (conv_halo_ok(dtype, p)) return ROUTE_CONV_HALO;
  if (gemm_wide_ok(dtype, p)) return ROUTE_WIDE;
  if (gemm_dma_ok(dtype, p)) return ROUTE_DMA;
  return ROUTE_TILE;
}

// can the kernel that will take this problem apply GemmParams::rowvec?  (the engine asks at plan-build time; launch_gemm re-checks)
bool gemm_rowvec_ok(int dtype, const GemmParams& p) {
  if (p.epi != EPI_NONE || p.splitk > 1 || gemm_pick_splitk(dtype, p) > 1) return false;
  if (p.rowvec_rows % 64 != 0 || p.rowvec_per % 64 != 0 || p.rowvec_rows > p.M) return false;
  if (p.out_lo && (p.ldo_lo % 8 != 0 || ((uintptr_t)p.out_lo & 15))) return false;
  if ((uintptr_t)p.rowvec & 15) return false;
  const int r = gemm_route(dtype, p);
  return r == ROUTE_WIDE || r == ROUTE_DUO;
}

int launch_gemm(int dtype, const GemmParams& p, hipStream_t s) {
  if (p.rowvec && !gemm_rowvec_ok(dtype, p)) TANGO_FAIL("gemm: rowvec is only implemented by the 256 x 320 / 256 x 160 GEMM epilogues");
  const int route = gemm_route(dtype, p);
  if (p.epi == EPI_VT && p.vt_perm && (p.vt_S % 32 != 0 || dtype == DT_F32)) TANGO_FAIL("gemm: vt_perm permutes whole blocks of 32 tokens of a 16-bit V^T (vt_S % 32 == 0)");
  if (p.wb_rows && route != ROUTE_WIDE && route != ROUTE_DUO) TANGO_FAIL("gemm: per-sample weights are implemented by the 256 x 320 / 256 x 160 GEMMs only");
  if (p.glu_tanh && dtype != DT_F32 && route != ROUTE_WIDE && route != ROUTE_DUO)
    TANGO_FAIL("gemm: the tanh-GELU gate (T5 gated-gelu) is implemented by the fp32 kernels and the 256 x 320 / 256 x 160 GEMMs only");
  switch (route) {
    case ROUTE_WIDE: return launch_gemm_wide(dtype, p, s);
    case ROUTE_DUO: return launch_gemm_duo(dtype, p, s);
    case ROUTE_STREAM: return launch_linear_stream(dtype, p, s);
    case ROUTE_CONV_WIDE: return launch_conv_wide(dtype, p, g_zero_page, s);
    case ROUTE_CONV_HALO: return launch_conv_halo(dtype, p, g_zero_page, s);
    case ROUTE_DMA: return launch_gemm_dma(dtype, p, g_zero_page, s);
    case ROUTE_NONE: TANGO_FAIL("gemm: ln_fold is only implemented by the streaming linear and the 256 x 320 GEMM kernels");
    default: break;
  }
  switch (dtype) {
    case DT_F32: return launch_t<float>(p, s);
    case DT_F16: return launch_t<f16>(p, s);
    case DT_BF16: return launch_t<bf16>(p, s);
  }
  TANGO_FAIL("gemm: bad dtype");
}

}  // namespace tango
