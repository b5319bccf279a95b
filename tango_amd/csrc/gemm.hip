// Gather-GEMM for gfx950: implicit-GEMM conv2d(3x3)/conv1d/transposed-conv1d-phase/linear on MFMA.
//
// Orientation: the MFMA "A" operand is the WEIGHT tile (rows n), the "B" operand is the ACTIVATION
// tile (cols m), so each lane's accumulator quad is 4 consecutive output channels of one output row
// -> vector epilogue (bias/act/residual/GEGLU) and 8/16-byte stores.
// 16x16 MFMA tiles; k is consumed in 64-byte groups per operand row: one ds_read_b128 per fragment
// feeds one v_mfma_f32_16x16x32_{f16,bf16} or four v_mfma_f32_16x16x4_f32 (k order inside a group
// is a fixed permutation shared by both operands, which leaves the dot product unchanged).
//
// Reference ops this kernel replaces (ATen dispatches): convolution (66/UNet step),
// mm/addmm (216), and the VAE / HiFi-GAN convolution + conv_transpose1d calls.
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

template <typename T> __device__ __forceinline__ u32x4 act_vec(u32x4 v, int act, float slope) {
  constexpr int EPV = 16 / sizeof(T);
  T e[EPV];
  __builtin_memcpy(e, &v, 16);
#pragma unroll
  for (int i = 0; i < EPV; ++i) e[i] = from_f<T>(apply_act(to_f(e[i]), act, slope));
  __builtin_memcpy(&v, e, 16);
  return v;
}

template <typename T, int BM, int BN, int BKB, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  constexpr int EPV = 16 / (int)sizeof(T);     // elements per 16-byte vector
  constexpr int BK = BKB / (int)sizeof(T);     // k elements per chunk
  constexpr int NKG = BKB / 64;                // 64-byte k groups per chunk
  constexpr int PPR = BKB / 16;                // 16-byte pieces per tile row
  constexpr int RPP = 256 / PPR;               // tile rows staged per pass
  constexpr int AP = (BM + RPP - 1) / RPP;
  constexpr int BP = (BN + RPP - 1) / RPP;
  constexpr int LDSR = BKB + 16;               // padded LDS row (bytes)
  constexpr int WMR = BM / WM, WNR = BN / WN;
  constexpr int TM = WMR / 16, TN = WNR / 16;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(WMR % 16 == 0 && WNR % 16 == 0, "tile");

  __shared__ __attribute__((aligned(16))) unsigned char smem[(BM + BN) * LDSR];
  unsigned char* Xs = smem;
  unsigned char* Ws = smem + BM * LDSR;

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

  // per-thread gather rows
  int a_rowb[AP], a_c0[AP], a_c1[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int rl = i * RPP + prow;
    const int m = m0 + rl;
    a_rowb[i] = -1; a_c0[i] = 0; a_c1[i] = 0;
    if (rl < BM && m < p.M) {
      if (p.mode == GATHER_2D) {
        const int hw = p.H * p.Wd;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / p.Wd, x = rem - y * p.Wd;
        a_rowb[i] = b * p.Hin * p.Win;
        a_c0[i] = y * p.stride - 1;
        a_c1[i] = x * p.stride - 1;
      } else {
        const int b = m / p.rows_pb, q = m - b * p.rows_pb;
        a_rowb[i] = b * p.Lin;
        a_c0[i] = q * p.in_mul + p.in_off;
      }
    }
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ar[AP], br[BP];
  const int nk = p.K / BK;

  auto load_chunk = [&](int kc) {
    const int k0 = kc * BK;
    const int tap = k0 / p.Cin;
    const int cc = k0 - tap * p.Cin;
    int dy = 0, dx = 0;
    if (p.mode == GATHER_2D) { dy = tap / 3; dx = tap - dy * 3; }
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (a_rowb[i] >= 0) {
        int64_t srow; bool ok;
        if (p.mode == GATHER_2D) {
          const int iy = a_c0[i] + dy, ix = a_c1[i] + dx;
          ok = (unsigned)iy < (unsigned)(p.Hin << p.ups) && (unsigned)ix < (unsigned)(p.Win << p.ups);
          srow = (int64_t)a_rowb[i] + (int64_t)(iy >> p.ups) * p.Win + (ix >> p.ups);
        } else {
          const int idx = a_c0[i] + tap * p.tap_step;
          ok = (unsigned)idx < (unsigned)p.Lin;
          srow = (int64_t)a_rowb[i] + idx;
        }
        if (ok) {
          v = *(const u32x4*)(Ab + (srow * p.lda + cc + pcol * EPV) * (int64_t)sizeof(T));
          if (p.a_act != ACT_NONE) v = act_vec<T>(v, p.a_act, p.a_slope);
        }
      }
      ar[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int rl = i * RPP + prow;
      const int n = n0 + rl;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (rl < BN && n < p.N) v = *(const u32x4*)(Wb + ((int64_t)n * p.Kp + k0 + pcol * EPV) * (int64_t)sizeof(T));
      br[i] = v;
    }
  };

  load_chunk(0);
  for (int kc = 0; kc < nk; ++kc) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int rl = i * RPP + prow;
      if (rl < BM) *(u32x4*)(Xs + rl * LDSR + pcol * 16) = ar[i];
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int rl = i * RPP + prow;
      if (rl < BN) *(u32x4*)(Ws + rl * LDSR + pcol * 16) = br[i];
    }
    __syncthreads();
    if (kc + 1 < nk) load_chunk(kc + 1);
#pragma unroll
    for (int ks = 0; ks < NKG; ++ks) {
      u32x4 wf[TN], xf[TM];
      const int koff = ks * 64 + (lane >> 4) * 16;
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[a] = *(const u32x4*)(Ws + (wn * WNR + a * 16 + (lane & 15)) * LDSR + koff);
#pragma unroll
      for (int b = 0; b < TM; ++b) xf[b] = *(const u32x4*)(Xs + (wm * WMR + b * 16 + (lane & 15)) * LDSR + koff);
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
    }
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const int g4 = (lane >> 4) * 4;
  const float* bias = p.bias ? p.bias + (int64_t)blockIdx.z * p.sBias : nullptr;
  const float* bias2 = nullptr;
  if (p.bias2) bias2 = p.bias2 + (int64_t)(p.step_ptr ? *p.step_ptr : 0) * p.bias2_stride;
  unsigned char* Ob = (unsigned char*)p.out;
  const unsigned char* Rb = (const unsigned char*)p.R;
  const int osz = p.epi == EPI_I16 ? 2 : (p.out_f32 ? 4 : (int)sizeof(T));
  Ob += (int64_t)blockIdx.z * p.sO * osz;
  if (Rb) Rb += (int64_t)blockIdx.z * p.sR * (int64_t)sizeof(T);

#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int m = m0 + wm * WMR + b * 16 + (lane & 15);
    if (m >= p.M) continue;
    int64_t orow = m;
    if (p.mode == GATHER_1D) {
      const int bb = m / p.rows_pb, q = m - bb * p.rows_pb;
      orow = (int64_t)bb * p.Lout + (int64_t)q * p.out_mul + p.out_off;
    }
    const float rbias = (bias && p.bias_rows) ? bias[orow] : 0.f;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      if (p.epi == EPI_GEGLU && (a & 1)) continue;
      const int nt = n0 + wn * WNR + a * 16;   // tile base column (packed order)
      const int n = nt + g4;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[a][b][r] * p.alpha;
        if (n + r < p.N) {
          if (bias) x += p.bias_rows ? rbias : bias[n + r];
          if (bias2) x += bias2[n + r];
        }
        v[r] = x;
      }
      int oc = n;
      int ncols = p.N;
      if (p.epi == EPI_GEGLU) {
        // packed rows: [16 value | 16 gate] blocks -> out col = nt/2 + g4 + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float gt = acc[a + 1 < TN ? a + 1 : a][b][r] * p.alpha;
          if (bias) gt += bias[n + 16 + r];
          v[r] = v[r] * gelu_erf_f(gt);
        }
        oc = (nt >> 1) + g4;
        ncols = p.N >> 1;
      } else if (p.e_act != ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.e_act, p.e_slope);
      }
      const bool full = (oc + 3 < ncols);
      if (Rb) {
        const T* rp = (const T*)Rb + orow * p.ldr + oc;
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
      if (p.epi == EPI_I16) {
        int16_t* op = (int16_t*)Ob + orow * p.ldo + oc;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (oc + r < ncols) op[r] = (int16_t)(int)v[r];   // C truncation, int16 wrap (hifigan/utilities.py:81)
      } else if (p.out_f32) {
        float* op = (float*)Ob + orow * p.ldo + oc;
        if (full && ((p.ldo | oc) & 3) == 0) {
          *(f32x4*)op = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (oc + r < ncols) op[r] = v[r];
        }
      } else {
        T* op = (T*)Ob + orow * p.ldo + oc;
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

template <typename T, int BM, int BN, int BKB, int WM, int WN>
static int launch_cfg(const GemmParams& p, hipStream_t s) {
  const int MT = (p.M + BM - 1) / BM, NT = (p.N + BN - 1) / BN;
  dim3 grid((unsigned)(MT * NT), 1, (unsigned)p.batch);
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, BKB, WM, WN>), grid, dim3(256), 0, s, p);
  TANGO_HIP(hipGetLastError());
  return 0;
}

template <typename T, int BKB>
static int launch_tile(const GemmParams& p, hipStream_t s) {
  if (p.epi == EPI_GEGLU) {
    if (p.N % 32 != 0) TANGO_FAIL("GEGLU gemm needs N % 32 == 0");
    return launch_cfg<T, 128, 128, BKB, 2, 2>(p, s);
  }
  if (p.N % 160 == 0) return launch_cfg<T, 128, 160, BKB, 2, 2>(p, s);
  if (p.N >= 96) return launch_cfg<T, 128, 128, BKB, 2, 2>(p, s);
  if (p.N > 32) return launch_cfg<T, 128, 64, BKB, 2, 2>(p, s);
  if (p.N > 16) return launch_cfg<T, 256, 32, BKB, 4, 1>(p, s);
  return launch_cfg<T, 256, 16, BKB, 4, 1>(p, s);
}

template <typename T>
static int launch_t(const GemmParams& p, hipStream_t s) {
  const int cb = p.Cin * (int)sizeof(T);
  if (p.M <= 0 || p.N <= 0) return 0;
  if (p.K % p.Cin != 0) TANGO_FAIL("gemm: K must be taps*Cin");
  if ((p.lda * (int64_t)sizeof(T)) % 16 != 0 || (p.Kp * (int64_t)sizeof(T)) % 16 != 0) TANGO_FAIL("gemm: lda/Kp must be 16-byte multiples");
  if (cb % 128 == 0) return launch_tile<T, 128>(p, s);
  if (cb % 64 == 0) return launch_tile<T, 64>(p, s);
  TANGO_FAIL("gemm: Cin*sizeof(T) must be a multiple of 64 bytes");
}

int launch_gemm(int dtype, const GemmParams& p, hipStream_t s) {
  switch (dtype) {
    case DT_F32: return launch_t<float>(p, s);
    case DT_F16: return launch_t<f16>(p, s);
    case DT_BF16: return launch_t<bf16>(p, s);
  }
  TANGO_FAIL("gemm: bad dtype");
}

}  // namespace tango
