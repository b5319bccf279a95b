// GroupNorm / LayerNorm / row-softmax kernels for channels-last activations (HBM-bound).
// Reference ops replaced: native_group_norm (61/UNet step, 24/VAE), native_layer_norm (48), softmax
// (VAE AttnBlock, audioldm/variational_autoencoder/modules.py:204-230).
// Statistics are always fp32; 16-byte vector loads; wave64 shuffle reductions.
#include "common.h"

namespace tango {

__device__ __forceinline__ float wave_sum(float v);
__device__ __forceinline__ float wave_max(float v);

static constexpr int GN_NV = 3;      // max 16B vectors per thread per row
static constexpr int GN_MAXC = 4096;

template <typename T> __device__ __forceinline__ void unpack16(const u32x4& v, float* f) {
  constexpr int EPV = 16 / (int)sizeof(T);
  T e[EPV];
  __builtin_memcpy(e, &v, 16);
#pragma unroll
  for (int i = 0; i < EPV; ++i) f[i] = to_f(e[i]);
}
template <typename T> __device__ __forceinline__ u32x4 pack16(const float* f) {
  constexpr int EPV = 16 / (int)sizeof(T);
  T e[EPV];
#pragma unroll
  for (int i = 0; i < EPV; ++i) e[i] = from_f<T>(f[i]);
  u32x4 v;
  __builtin_memcpy(&v, e, 16);
  return v;
}

struct GnGeom {
  int VPR, TPR, RPB, RC, chunks;
};

template <typename T> static GnGeom gn_geom(int B, int rows, int C) {
  constexpr int EPV = 16 / (int)sizeof(T);
  GnGeom g;
  g.VPR = C / EPV;
  g.TPR = g.VPR < 256 ? g.VPR : 256;
  g.RPB = 256 / g.TPR;
  if (g.RPB < 1) g.RPB = 1;
  long total = (long)rows * B;
  int rc = (int)((total + 2047) / 2048);
  if (rc < 16) rc = 16;
  rc = ((rc + g.RPB - 1) / g.RPB) * g.RPB;
  if (rc > rows) rc = ((rows + g.RPB - 1) / g.RPB) * g.RPB;
  g.RC = rc;
  g.chunks = (rows + rc - 1) / rc;
  return g;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int64_t ldx, float* __restrict__ partial,
                                                       int rows, int C, int groups, int VPR, int TPR, int RPB, int RC) {
  constexpr int EPV = 16 / (int)sizeof(T);
  __shared__ float sm[2][GN_MAXC];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int rl = tid / TPR, tv = tid % TPR;
  float s[GN_NV][EPV], ss[GN_NV][EPV];
#pragma unroll
  for (int j = 0; j < GN_NV; ++j)
#pragma unroll
    for (int e = 0; e < EPV; ++e) { s[j][e] = 0.f; ss[j][e] = 0.f; }
  const int r0 = chunk * RC, r1 = min(rows, r0 + RC);
  if (rl < RPB) {
    if (VPR <= TPR) {
      // one 16-byte piece per thread and row (every UNet width): four rows per iteration with all four loads issued before the
      // first use -- the plain loop kept one load per thread in flight, ~30 KB per CU, short of what HBM latency x bandwidth
      // needs; the accumulation order per thread (row order) is unchanged, so the statistics are bit-identical
      if (tv < VPR) {
        int r = r0 + rl;
        for (; r + 3 * RPB < r1; r += 4 * RPB) {
          u32x4 v4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v4[u] = *(const u32x4*)(x + ((int64_t)b * rows + r + u * RPB) * ldx + tv * EPV);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float f[EPV];
            unpack16<T>(v4[u], f);
#pragma unroll
            for (int e = 0; e < EPV; ++e) { s[0][e] += f[e]; ss[0][e] += f[e] * f[e]; }
          }
        }
        for (; r < r1; r += RPB) {
          float f[EPV];
          unpack16<T>(*(const u32x4*)(x + ((int64_t)b * rows + r) * ldx + tv * EPV), f);
#pragma unroll
          for (int e = 0; e < EPV; ++e) { s[0][e] += f[e]; ss[0][e] += f[e] * f[e]; }
        }
      }
    } else {
    for (int r = r0 + rl; r < r1; r += RPB) {
      const T* row = x + ((int64_t)b * rows + r) * ldx;
#pragma unroll
      for (int j = 0; j < GN_NV; ++j) {
        const int v = tv + j * TPR;
        if (v < VPR) {
          float f[EPV];
          unpack16<T>(*(const u32x4*)(row + v * EPV), f);
#pragma unroll
          for (int e = 0; e < EPV; ++e) { s[j][e] += f[e]; ss[j][e] += f[e] * f[e]; }
        }
      }
    }
    }
  }
  // reduce over rl through LDS (RPB*C <= 2048 when RPB > 1)
  for (int pass = 0; pass < RPB; ++pass) {
    if (rl == pass) {
#pragma unroll
      for (int j = 0; j < GN_NV; ++j) {
        const int v = tv + j * TPR;
        if (v < VPR) {
#pragma unroll
          for (int e = 0; e < EPV; ++e) {
            const int c = v * EPV + e;
            if (pass == 0) { sm[0][c] = s[j][e]; sm[1][c] = ss[j][e]; }
            else { sm[0][c] += s[j][e]; sm[1][c] += ss[j][e]; }
          }
        }
      }
    }
    __syncthreads();
  }
  const int cg = C / groups;
  if (tid < groups) {
    float a = 0.f, q = 0.f;
    for (int c = tid * cg; c < (tid + 1) * cg; ++c) { a += sm[0][c]; q += sm[1][c]; }
    float* o = partial + (((int64_t)b * gridDim.x + chunk) * groups + tid) * 2;
    o[0] = a; o[1] = q;
  }
}

// Normalise + affine (+ SiLU).  The former gn_finalize launch is folded in: every workgroup reduces the per-chunk partial
// sums of ITS sample (chunks x groups pairs, double accumulation in a fixed order -- deterministic) and derives scale / shift
// for its own channels; 61 launches per UNet step fewer.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                                       const float* __restrict__ partial, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int chunks, int groups, float eps,
                                                       int rows, int C, int act, int VPR, int TPR, int RPB, int RC) {
  constexpr int EPV = 16 / (int)sizeof(T);
  __shared__ float mean_s[256], rstd_s[256];
  __shared__ double part_s[2][256], tot_s[2][256];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cg = C / groups;
  // Per-sample reduction of the chunk partials, spread over the whole workgroup: `lanes` threads per group each sum every
  // lanes-th chunk (double, fixed order), the group's first thread then adds the `lanes` sub-sums in fixed order.  (Round 2
  // had ONE thread per group walk all chunks: at B = 8 that is a chain of 128 loads at the head of every workgroup -- the
  // level-0 GroupNorm ran at 1.4 TB/s there, profiles/r2_unet_ops_small_batch_splitk.txt.)
  int lanes = 256 / groups;                         // groups <= 256 (checked by the launcher)
  if (lanes > 8) lanes = 8;
  {
    const int g = tid / lanes, sub = tid - g * lanes;
    if (g < groups) {
      double a = 0.0, q = 0.0;
      for (int ch = sub; ch < chunks; ch += lanes) {
        const f32x2 o = *(const f32x2*)(partial + (((int64_t)b * chunks + ch) * groups + g) * 2);
        a += (double)o.x; q += (double)o.y;
      }
      part_s[0][tid] = a; part_s[1][tid] = q;
    }
    __syncthreads();
    if (g < groups && sub == 0) {
      double a = 0.0, q = 0.0;
      for (int k = 0; k < lanes; ++k) { a += part_s[0][tid + k]; q += part_s[1][tid + k]; }
      tot_s[0][g] = a; tot_s[1][g] = q;
    }
  }
  __syncthreads();
  if (tid < groups) {
    const double n = (double)rows * cg;
    const double a = tot_s[0][tid], q = tot_s[1][tid];
    const double mean = a / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_s[tid] = (float)mean;
    rstd_s[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int rl = tid / TPR, tv = tid % TPR;
  if (rl >= RPB) return;
  float sc[GN_NV][EPV], sh[GN_NV][EPV];
#pragma unroll
  for (int j = 0; j < GN_NV; ++j) {
    const int v = tv + j * TPR;
    if (v < VPR) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        const int c = v * EPV + e;
        const int g = c / cg;
        const float s1 = rstd_s[g] * gamma[c];
        sc[j][e] = s1; sh[j][e] = beta[c] - mean_s[g] * s1;
      }
    }
  }
  const int r0 = chunk * RC, r1 = min(rows, r0 + RC);
  int rstart = r0 + rl;
  if (VPR <= TPR && tv < VPR) {
    // four rows per iteration, loads first (see gn_stats_kernel)
    for (; rstart + 3 * RPB < r1; rstart += 4 * RPB) {
      u32x4 v4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v4[u] = *(const u32x4*)(x + ((int64_t)b * rows + rstart + u * RPB) * ldx + tv * EPV);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[EPV];
        unpack16<T>(v4[u], f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          float t = f[e] * sc[0][e] + sh[0][e];
          if (act == ACT_SILU) t = silu_f(t);
          f[e] = t;
        }
        *(u32x4*)(y + ((int64_t)b * rows + rstart + u * RPB) * ldy + tv * EPV) = pack16<T>(f);
      }
    }
  }
  for (int r = rstart; r < r1; r += RPB) {
    const T* row = x + ((int64_t)b * rows + r) * ldx;
    T* orow = y + ((int64_t)b * rows + r) * ldy;
#pragma unroll
    for (int j = 0; j < GN_NV; ++j) {
      const int v = tv + j * TPR;
      if (v < VPR) {
        float f[EPV];
        unpack16<T>(*(const u32x4*)(row + v * EPV), f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          float t = f[e] * sc[j][e] + sh[j][e];
          if (act == ACT_SILU) t = silu_f(t);
          f[e] = t;
        }
        *(u32x4*)(orow + v * EPV) = pack16<T>(f);
      }
    }
  }
}

// Small tensors (L2-resident; B = 1 latency path): ONE kernel, one 1024-thread workgroup per (sample, group).  The group's
// rows x cg slice is read ONCE as element pairs into registers (<= GNS_NP pairs per thread, all loads in flight together),
// reduced through LDS, then normalised + affine + SiLU straight from registers.  Replaces three launches that are pure
// launch latency at this size.
static constexpr int GNS_NP = 24;
template <typename T> struct Pair;
template <> struct Pair<float> { using type = f32x2; };
template <> struct Pair<_Float16> { using type = uint32_t; };
template <> struct Pair<__bf16> { using type = uint32_t; };

template <typename T>
__global__ __launch_bounds__(1024) void gn_fused_small_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int rows, int C, int groups, float eps, int act) {
  using P = typename Pair<T>::type;
  __shared__ float red[2][16];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cg = C / groups, hp = cg >> 1;  // pairs per row of this group
  const int n = rows * hp;
  const T* xb = x + (int64_t)b * rows * ldx + g * cg;
  float v[GNS_NP][2];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < GNS_NP; ++i) {
    const int e = tid + i * 1024;
    v[i][0] = 0.f; v[i][1] = 0.f;
    if (e < n) {
      const int r = e / hp, c = (e - r * hp) * 2;
      const P pv = *(const P*)(xb + (int64_t)r * ldx + c);
      T t2[2];
      __builtin_memcpy(t2, &pv, sizeof(P));
      v[i][0] = to_f(t2[0]); v[i][1] = to_f(t2[1]);
    }
  }
#pragma unroll
  for (int i = 0; i < GNS_NP; ++i) { s += v[i][0] + v[i][1]; q += v[i][0] * v[i][0] + v[i][1] * v[i][1]; }
  s = wave_sum(s); q = wave_sum(q);
  if (lane == 0) { red[0][w] = s; red[1][w] = q; }
  __syncthreads();
  double S = 0.0, Q = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { S += (double)red[0][i]; Q += (double)red[1][i]; }
  const double nn = (double)rows * cg;
  const double mean = S / nn;
  double var = Q / nn - mean * mean;
  if (var < 0.0) var = 0.0;
  const float mu = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
  T* yb = y + (int64_t)b * rows * ldy + g * cg;
#pragma unroll
  for (int i = 0; i < GNS_NP; ++i) {
    const int e = tid + i * 1024;
    if (e < n) {
      const int r = e / hp, c = (e - r * hp) * 2;
      T t2[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float sc = rstd * gamma[g * cg + c + k];
        float t = v[i][k] * sc + (beta[g * cg + c + k] - mu * sc);
        if (act == ACT_SILU) t = silu_f(t);
        t2[k] = from_f<T>(t);
      }
      P pv;
      __builtin_memcpy(&pv, t2, sizeof(P));
      *(P*)(yb + (int64_t)r * ldy + c) = pv;
    }
  }
}

size_t groupnorm_ws_floats(int B, int rows, int C, int groups) {
  // upper bound on B*chunks*groups*2 (chunks <= rows/16 + 1) + B*C*2
  size_t chunks = (size_t)rows / 16 + 2;
  if (chunks > 2048 + 2) chunks = 2048 + 2;
  return (size_t)B * chunks * groups * 2 + (size_t)B * C * 2;
}

template <typename T>
static int gn_launch(const GroupNormParams& p, hipStream_t s) {
  constexpr int EPV = 16 / (int)sizeof(T);
  if (p.C % EPV != 0 || p.C > GN_MAXC || p.C % p.groups != 0 || p.groups > 256) TANGO_FAIL("groupnorm: unsupported C/groups");
  if (p.C / EPV > 256 * GN_NV) TANGO_FAIL("groupnorm: C too large");
  if ((p.ldx * (int64_t)sizeof(T)) % 16 || (p.ldy * (int64_t)sizeof(T)) % 16) TANGO_FAIL("groupnorm: ld alignment");
  const int cg = p.C / p.groups;
  if ((cg & 1) == 0 && (int64_t)p.rows * (cg / 2) <= (int64_t)1024 * GNS_NP && (p.ldx & 1) == 0 && (p.ldy & 1) == 0 &&
      (size_t)p.B * p.rows * p.C * sizeof(T) <= ((size_t)8 << 20)) {
    hipLaunchKernelGGL((gn_fused_small_kernel<T>), dim3((unsigned)p.groups, (unsigned)p.B), dim3(1024), 0, s, (const T*)p.x, p.ldx,
                       (T*)p.y, p.ldy, p.gamma, p.beta, p.rows, p.C, p.groups, p.eps, p.act);
    TANGO_HIP(hipGetLastError());
    return 0;
  }
  const GnGeom g = gn_geom<T>(p.B, p.rows, p.C);
  dim3 grid((unsigned)g.chunks, (unsigned)p.B);
  hipLaunchKernelGGL((gn_stats_kernel<T>), grid, dim3(256), 0, s, (const T*)p.x, p.ldx, p.partial, p.rows, p.C, p.groups,
                     g.VPR, g.TPR, g.RPB, g.RC);
  hipLaunchKernelGGL((gn_apply_kernel<T>), grid, dim3(256), 0, s, (const T*)p.x, p.ldx, (T*)p.y, p.ldy, p.partial, p.gamma, p.beta,
                     g.chunks, p.groups, p.eps, p.rows, p.C, p.act, g.VPR, g.TPR, g.RPB, g.RC);
  TANGO_HIP(hipGetLastError());
  return 0;
}

int launch_groupnorm(int dtype, const GroupNormParams& p, hipStream_t s) {
  switch (dtype) {
    case DT_F32: return gn_launch<float>(p, s);
    case DT_F16: return gn_launch<f16>(p, s);
    case DT_BF16: return gn_launch<bf16>(p, s);
  }
  TANGO_FAIL("groupnorm: bad dtype");
}

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers, two-pass (mean, then centered variance)
// ------------------------------------------------------------------------------------------
static constexpr int LN_NV = 6;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int rows, int C, float eps) {
  constexpr int EPV = 16 / (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int VPR = C / EPV;
  float f[LN_NV][EPV];
  float sum = 0.f;
  const T* xr = x + (int64_t)row * ldx;
#pragma unroll
  for (int j = 0; j < LN_NV; ++j) {
    const int v = lane + j * 64;
    if (v < VPR) {
      unpack16<T>(*(const u32x4*)(xr + v * EPV), f[j]);
#pragma unroll
      for (int e = 0; e < EPV; ++e) sum += f[j][e];
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < LN_NV; ++j) {
    const int v = lane + j * 64;
    if (v < VPR) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) { const float d = f[j][e] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
  T* yr = y + (int64_t)row * ldy;
#pragma unroll
  for (int j = 0; j < LN_NV; ++j) {
    const int v = lane + j * 64;
    if (v < VPR) {
      float o[EPV];
#pragma unroll
      for (int e = 0; e < EPV; ++e) o[e] = (f[j][e] - mean) * rstd * gamma[v * EPV + e] + beta[v * EPV + e];
      *(u32x4*)(yr + v * EPV) = pack16<T>(o);
    }
  }
}

template <typename T>
static int ln_launch(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta, int rows, int C,
                     float eps, hipStream_t s) {
  constexpr int EPV = 16 / (int)sizeof(T);
  if (C % EPV != 0 || C / EPV > 64 * LN_NV) TANGO_FAIL("layernorm: unsupported C");
  hipLaunchKernelGGL((layernorm_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const T*)x, ldx, (T*)y, ldy,
                     gamma, beta, rows, C, eps);
  TANGO_HIP(hipGetLastError());
  return 0;
}

int launch_layernorm(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta,
                     int rows, int C, float eps, hipStream_t s) {
  switch (dtype) {
    case DT_F32: return ln_launch<float>(x, ldx, y, ldy, gamma, beta, rows, C, eps, s);
    case DT_F16: return ln_launch<f16>(x, ldx, y, ldy, gamma, beta, rows, C, eps, s);
    case DT_BF16: return ln_launch<bf16>(x, ldx, y, ldy, gamma, beta, rows, C, eps, s);
  }
  TANGO_FAIL("layernorm: bad dtype");
}

// ------------------------------------------------------------------------------------------
// in-place row softmax(x * scale), one 256-thread block per row, row held in registers
// ------------------------------------------------------------------------------------------
static constexpr int SM_NV = 4;

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* __restrict__ x, int64_t ld, int cols, float scale) {
  constexpr int EPV = 16 / (int)sizeof(T);
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  T* xr = x + (int64_t)blockIdx.x * ld;
  const int VPR = cols / EPV;
  float f[SM_NV][EPV];
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < SM_NV; ++j) {
    const int v = tid + j * 256;
    if (v < VPR) {
      unpack16<T>(*(const u32x4*)(xr + v * EPV), f[j]);
#pragma unroll
      for (int e = 0; e < EPV; ++e) { f[j][e] *= scale; mx = fmaxf(mx, f[j][e]); }
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < SM_NV; ++j) {
    const int v = tid + j * 256;
    if (v < VPR) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) { f[j][e] = __expf(f[j][e] - mx); sum += f[j][e]; }
    }
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + w] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
  for (int j = 0; j < SM_NV; ++j) {
    const int v = tid + j * 256;
    if (v < VPR) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) f[j][e] *= inv;
      *(u32x4*)(xr + v * EPV) = pack16<T>(f[j]);
    }
  }
}

template <typename T>
static int sm_launch(void* x, int64_t ld, int rows, int cols, float scale, hipStream_t s) {
  constexpr int EPV = 16 / (int)sizeof(T);
  if (cols % EPV != 0 || cols / EPV > 256 * SM_NV) TANGO_FAIL("softmax_rows: unsupported cols");
  hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3((unsigned)rows), dim3(256), 0, s, (T*)x, ld, cols, scale);
  TANGO_HIP(hipGetLastError());
  return 0;
}

int launch_softmax_rows(int dtype, void* x, int64_t ld, int rows, int cols, float scale, hipStream_t s) {
  switch (dtype) {
    case DT_F32: return sm_launch<float>(x, ld, rows, cols, scale, s);
    case DT_F16: return sm_launch<f16>(x, ld, rows, cols, scale, s);
    case DT_BF16: return sm_launch<bf16>(x, ld, rows, cols, scale, s);
  }
  TANGO_FAIL("softmax_rows: bad dtype");
}

}  // namespace tango
