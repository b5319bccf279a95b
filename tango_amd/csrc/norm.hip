// GroupNorm / LayerNorm / row-softmax kernels for channels-last activations (HBM-bound).
// Reference ops replaced: native_group_norm (61/UNet step, 24/VAE), native_layer_norm (48), softmax
// (VAE AttnBlock, audioldm/variational_autoencoder/modules.py:204-230).
// Statistics are always fp32; 16-byte vector loads; wave64 shuffle reductions.
#include <map>
#include <mutex>
#include <utility>

#include "common.h"
#include "tuning.h"

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
  // at most 256 chunks per sample: gn_apply folds the finalize step in, i.e. EVERY workgroup re-reduces all `chunks` partials of its
  // sample -- O(chunks^2) bytes per sample.  The VAE's 65536-row tensors at B = 1 had 2048 chunks of 32 rows: 241 us per GroupNorm
  // (1.45 of the 3.6-ms B = 1 decode, profiles/r6_c4_vae_vocoder_per_op_b1.txt); UNet shapes and B >= 8 are unaffected (<= 256 already)
  if (rc < (rows + 255) / 256) rc = (rows + 255) / 256;
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

// ------------------------------------------------------------------------------------------
// Slab form (round 6): the GroupNorms of UNet levels 2-3 (<= 256 rows per sample; C = 1280 / 2560, i.e. groups of 40 / 80 channels = whole 16-byte
// vectors).  One 320-thread workgroup owns G adjacent groups of one sample -- a rows x (G cg) slab, 320 contiguous bytes (20 vectors) per row: it is read
// ONCE as 16-byte vectors into registers (thread = (row of a 16-row pass, vector): rows / 16 vectors per thread, every load in flight before the first
// use; its group, gamma and beta are fixed), the per-group (sum, sum of squares) go through LDS in a fixed order to a double-precision mean / variance, and the slab is normalised (+ affine, SiLU) from the registers and written once.
// One launch, one read, one write: the two-launch path reads these tensors twice (they are 10-84 MB at B = 32: three passes through the memory side),
// and the (sample, group)-per-workgroup kernel above fetches them as 4-byte pairs with 1024-thread workgroups (19 us per launch for a 10-MB tensor).
// The choice depends on (rows, C, groups, dtype) only -- never on the batch -- so a sample's result does not depend on what it is batched with.
// Reference op: torch.nn.GroupNorm of ResnetBlock2D.norm1 / norm2 and Transformer2DModel.norm (diffusers resnet.py:549-597, transformer_2d.py:255-262).
// ------------------------------------------------------------------------------------------
static constexpr int GSL_VPS = 20, GSL_RPP = 16, GSL_TH = GSL_VPS * GSL_RPP, GSL_MAXG = 4;   // 20 vectors per slab row x 16 rows per pass = 320 threads
template <typename T, int NV>
__global__ __launch_bounds__(GSL_TH, 3) void gn_slab_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int rows, int C, int groups, float eps, int act, int G, int VPG) {
  constexpr int EPV = 16 / (int)sizeof(T);
  __shared__ f32x2 part[GSL_TH];
  __shared__ f32x2 part2[GSL_MAXG][GSL_RPP];
  __shared__ float mean_s[GSL_MAXG], rstd_s[GSL_MAXG];
  // thread -> (row of the pass r0, vector of the slab row vv): its group, gamma and beta are fixed; vector i is row r0 + 16 i
  const int tid = threadIdx.x, r0 = tid / GSL_VPS, vv = tid - r0 * GSL_VPS, grp = vv / VPG;
  const int b = blockIdx.y, cg = C / groups, c0 = blockIdx.x * G * cg + vv * EPV;
  const T* xp = x + ((int64_t)b * rows + r0) * ldx + c0;
  T* yp = y + ((int64_t)b * rows + r0) * ldy + c0;
  const int64_t xs = (int64_t)GSL_RPP * ldx, ys = (int64_t)GSL_RPP * ldy;
  u32x4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *(const u32x4*)(xp + i * xs);         // rows == 16 NV (launcher): no tail
  float ga[EPV], be[EPV];                                // (16-byte aligned: the launcher checks gamma / beta, c0 is a multiple of 4)
#pragma unroll
  for (int u = 0; u < EPV; u += 4) {
    *(f32x4*)(ga + u) = *(const f32x4*)(gamma + c0 + u);
    *(f32x4*)(be + u) = *(const f32x4*)(beta + c0 + u);
  }
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float f[EPV];
    unpack16<T>(v[i], f);
#pragma unroll
    for (int u = 0; u < EPV; ++u) { s += f[u]; q += f[u] * f[u]; }
    __builtin_amdgcn_sched_barrier(0);                   // one vector at a time (hipcc otherwise unpacks all NV vectors at once: 8 NV registers, spills at NV = 16)
  }
  // per-group sums in a fixed order: the VPG vectors of a (pass row, group), then the 16 pass rows (double)
  part[tid] = f32x2{s, q};
  __syncthreads();
  if (tid < G * GSL_RPP) {
    const int k = tid / GSL_RPP, rr = tid - k * GSL_RPP;
    float a = 0.f, c = 0.f;
    for (int j = 0; j < VPG; ++j) { const f32x2 o = part[rr * GSL_VPS + k * VPG + j]; a += o.x; c += o.y; }
    part2[k][rr] = f32x2{a, c};
  }
  __syncthreads();
  if (tid < G) {
    double S = 0.0, Q = 0.0;
#pragma unroll
    for (int rr = 0; rr < GSL_RPP; ++rr) { S += (double)part2[tid][rr].x; Q += (double)part2[tid][rr].y; }
    const double nn = (double)rows * cg;
    const double mean = S / nn;
    double var = Q / nn - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_s[tid] = (float)mean;
    rstd_s[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  // (the packed vectors are made opaque here: hipcc otherwise keeps the 8 NV CONVERTED values of the statistics loop alive across the barriers
  //  to save the second conversion -- 128 more registers at NV = 16)
#pragma unroll
  for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[i]));
  const float mu = mean_s[grp], rstd = rstd_s[grp];
  float sc[EPV], sh[EPV];
#pragma unroll
  for (int u = 0; u < EPV; ++u) { sc[u] = rstd * ga[u]; sh[u] = be[u] - mu * sc[u]; }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float f[EPV];
    unpack16<T>(v[i], f);
#pragma unroll
    for (int u = 0; u < EPV; ++u) {
      float t = f[u] * sc[u] + sh[u];
      if (act == ACT_SILU) t = silu_f(t);
      f[u] = t;
    }
    *(u32x4*)(yp + i * ys) = pack16<T>(f);
    if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // two vectors' exponentials in flight at a time, not sixteen
  }
}

// geometry of the slab form for this shape: G groups per workgroup (20 vectors per slab row), vectors per group, vectors per thread; false = the shape does not take it
template <typename T> static bool gn_slab_geom(int rows, int C, int groups, int& G, int& VPG, int& nv) {
  constexpr int EPV = 16 / (int)sizeof(T);
  const int cg = C / groups;
  if (cg % EPV != 0 || rows % GSL_RPP != 0) return false;
  VPG = cg / EPV;
  if (VPG <= 0 || GSL_VPS % VPG != 0) return false;
  G = GSL_VPS / VPG;
  if (G > GSL_MAXG || groups % G != 0) return false;
  nv = rows / GSL_RPP;
  return nv == 4 || nv == 8 || nv == 16;
}
template <typename T, int NV> static void gn_slab_go(const GroupNormParams& p, int G, int VPG, hipStream_t s) {
  hipLaunchKernelGGL((gn_slab_kernel<T, NV>), dim3((unsigned)(p.groups / G), (unsigned)p.B), dim3(GSL_TH), 0, s, (const T*)p.x, p.ldx, (T*)p.y, p.ldy,
                     p.gamma, p.beta, p.rows, p.C, p.groups, p.eps, p.act, G, VPG);
}

// ------------------------------------------------------------------------------------------
// Cooperative single-launch GroupNorm (round 4).  rocprofv3 (profiles/r4_c8_kernel_stats_b1.txt) prices a dispatch inside the
// hipGraph at 4-5 us and the (sample, group)-per-workgroup kernel above at 16 us per call (64 workgroups, 4-byte strided loads);
// the two-launch path reads x twice.  Here the grid of the STATISTICS pass keeps its rows in registers across a per-sample
// rendezvous and normalises them in place of a second launch: one read, one write, one dispatch, 16-byte coalesced accesses.
//   phase 1  every workgroup loads NV x RPB rows (all loads in flight), reduces per-group (sum, sum of squares) of its rows through
//            LDS and publishes them to partial[b][chunk][group] with agent-scope stores;
//   barrier  per sample: arrival counter + generation word (agent-scope atomics; the last arriver resets the counter and bumps
//            the generation, so the words are back at rest when the kernel ends and serve the next launch of any geometry);
//   phase 2  every workgroup sums its sample's chunk partials in the fixed order of gn_apply_kernel (double), derives
//            scale / shift and writes y from the registers.
// Deadlock rule: workgroups that spin must not keep out workgroups they wait for -- the launcher takes this path only when the
// WHOLE grid fits the chip at once (occupancy API x CUs, with margin).  That is an estimate for an otherwise idle GPU (another
// process or stream can hold CUs), so correctness does NOT rest on it: the spin is bounded, and a workgroup whose
// partners have not all arrived in time computes the missing information ITSELF -- it re-reduces every chunk of its sample from x
// with phase 1's arithmetic in phase 1's order and (re)publishes those partials (bit-identical to what the owners write, so the
// duplicate stores are benign), then carries on with phase 2.  The result is the same bits as the rendezvous path, only slower;
// the counter word [2 * COOP_SYNC_SLOTS] counts such fallbacks for diagnostics.  `force_fb` (TANGO_GN_COOP_FORCE_FALLBACK=1, tests)
// sends every workgroup down that path without waiting.
// ------------------------------------------------------------------------------------------
template <typename T, int NV>
__global__ __launch_bounds__(256, 4) void gn_coop_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ partial, unsigned* __restrict__ sync, int rows, int C, int groups,
                                                      float eps, int act, int VPR, int RPB, int force_fb) {
  constexpr int EPV = 16 / (int)sizeof(T);
  constexpr int MAXC = 256 * EPV;                     // VPR <= 256 (launcher)
  // LDS: the per-channel sums of phase 1 and the per-group reductions of phase 2 share one 16-KiB area (a barrier lies between)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * MAXC * 4 > 2 * 2 * 256 * 8 ? 2 * MAXC * 4 : 2 * 2 * 256 * 8];
  __shared__ float mean_s[256], rstd_s[256];
  __shared__ int fb_s;
  float (*const sm)[MAXC] = (float (*)[MAXC])lds;
  double (*const part_s)[256] = (double (*)[256])lds;
  double (*const tot_s)[256] = (double (*)[256])(lds + 2 * 256 * 8);
  const int tid = threadIdx.x;
  const int b = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
  const int rl = tid / VPR, tv = tid - rl * VPR;
  const int cg = C / groups;
  const int RC = RPB * NV, r0 = chunk * RC;
  const bool active = rl < RPB;
  // wave-uniform sample base + 32-bit per-lane byte offsets (a sample is < 4 GiB): no 64-bit address per row in VGPRs
  const unsigned char* const xb = (const unsigned char*)(x + (int64_t)b * rows * ldx);
  unsigned char* const yb = (unsigned char*)(y + (int64_t)b * rows * ldy);
  const unsigned xoff = (unsigned)(((int64_t)(r0 + rl) * ldx + tv * EPV) * (int64_t)sizeof(T)), xstep = (unsigned)((int64_t)RPB * ldx * (int64_t)sizeof(T));
  const unsigned yoff = (unsigned)(((int64_t)(r0 + rl) * ldy + tv * EPV) * (int64_t)sizeof(T)), ystep = (unsigned)((int64_t)RPB * ldy * (int64_t)sizeof(T));
  u32x4 v[NV];
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int r = r0 + rl + u * RPB;
    v[u] = u32x4{0u, 0u, 0u, 0u};
    if (active && r < rows) v[u] = *(const u32x4*)(xb + (xoff + (unsigned)u * xstep));
  }
  // phase-1 reduction of one chunk: per-thread sums over its NV rows (ascending), per-channel sums across the RPB row slots through
  // LDS (ascending), per-group sums over the group's channels (ascending); published with agent-scope stores.  `own`: rows come from
  // the registers loaded above; otherwise (fallback) they are re-read from x -- the SAME values in the SAME order, hence the same bits.
  auto reduce_chunk = [&](int ch, bool own) {
    float s[EPV], ss[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) { s[e] = 0.f; ss[e] = 0.f; }
#pragma unroll
    for (int u = 0; u < NV; ++u) {       // rows past the end contribute zeros
      u32x4 w = v[u];
      if (!own) {
        const int r = ch * RC + rl + u * RPB;
        w = u32x4{0u, 0u, 0u, 0u};
        if (active && r < rows) w = *(const u32x4*)(xb + (((int64_t)r * ldx + tv * EPV) * (int64_t)sizeof(T)));
      }
      float f[EPV];
      unpack16<T>(w, f);
#pragma unroll
      for (int e = 0; e < EPV; ++e) { s[e] += f[e]; ss[e] += f[e] * f[e]; }
    }
    for (int pass = 0; pass < RPB; ++pass) {
      if (rl == pass) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          const int c = tv * EPV + e;
          if (pass == 0) { sm[0][c] = s[e]; sm[1][c] = ss[e]; }
          else { sm[0][c] += s[e]; sm[1][c] += ss[e]; }
        }
      }
      __syncthreads();
    }
    if (tid < groups) {
      float a = 0.f, q = 0.f;
      for (int c = tid * cg; c < (tid + 1) * cg; ++c) { a += sm[0][c]; q += sm[1][c]; }
      float* o = partial + (((int64_t)b * chunks + ch) * groups + tid) * 2;
      __hip_atomic_store(o, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(o + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  reduce_chunk(chunk, true);
  // ---- per-sample rendezvous ----
  // The partials travel as agent-scope (sc1) stores / loads, i.e. through the coherence point, so the words only need ORDER, not cache
  // maintenance: every storing thread waits for its stores (vmcnt(0)), the workgroup barrier orders them before thread 0's arrival,
  // and the poll is a relaxed sc1 load.  Measured (profiles/r4_c11_gn_coop_modes_ab_b*.txt): with acquire loads in the poll loop (one
  // `buffer_inv sc1` per poll per waiting workgroup while late workgroups are still loading x) the kernel is 2x SLOWER than two
  // launches at B = 8; with a release arrival + relaxed poll 1.4x; this form 1.2x.  The rendezvous itself costs ~10 us on this
  // 8-XCD chip -- twice a dispatch (4-5 us) -- so it only pays where it replaces something slower than a second launch: see gn_launch.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    int fb = 0;
    unsigned* const cnt = sync + (b & (COOP_SYNC_SLOTS - 1));
    unsigned* const gen = sync + COOP_SYNC_SLOTS + (b & (COOP_SYNC_SLOTS - 1));
    const unsigned g0 = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // cannot change before everyone arrived
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            // g0 is read BEFORE this workgroup arrives
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)chunks - 1u) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          // the reset lands before the others are released
      __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      int spins = 0;
      while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g0) {
        __builtin_amdgcn_s_sleep(4);
        if (force_fb || ++spins > (1 << 17)) { fb = 1; break; }      // ~0.05-0.1 s: four orders of magnitude above a healthy rendezvous
      }
    }
    if (force_fb) fb = 1;
    fb_s = fb;
  }
  __syncthreads();
  if (fb_s) {
    // Not everyone this workgroup waits for has arrived (they may not even be resident).  Do not depend on them: rebuild the partials
    // of EVERY chunk of this sample here.  (The arrival above still counts, so the words return to rest once the stragglers have run.)
    if (tid == 0) __hip_atomic_fetch_add(sync + 2 * COOP_SYNC_SLOTS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int ch = 0; ch < chunks; ++ch) {
      __syncthreads();                       // the previous chunk's group sums have been read out of `sm`
      reduce_chunk(ch, false);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // ---- per-sample totals: `lanes` threads per group each sum every lanes-th chunk (double, fixed order), as gn_apply_kernel ----
  int lanes = 256 / groups;
  if (lanes > 8) lanes = 8;
  {
    const int g = tid / lanes, sub = tid - g * lanes;
    if (g < groups) {
      double a = 0.0, q = 0.0;
      for (int ch = sub; ch < chunks; ch += lanes) {
        const float* o = partial + (((int64_t)b * chunks + ch) * groups + g) * 2;
        a += (double)__hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q += (double)__hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  if (!active) return;
  float sc[EPV], sh[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    const int c = tv * EPV + e;
    const int g = c / cg;
    const float s1 = rstd_s[g] * gamma[c];
    sc[e] = s1; sh[e] = beta[c] - mean_s[g] * s1;
  }
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int r = r0 + rl + u * RPB;
    if (r < rows) {
      float f[EPV];
      unpack16<T>(v[u], f);
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        float t = f[e] * sc[e] + sh[e];
        if (act == ACT_SILU) t = silu_f(t);
        f[e] = t;
      }
      *(u32x4*)(yb + (yoff + (unsigned)u * ystep)) = pack16<T>(f);
    }
  }
}

// how many workgroups of `kfn` (256 threads, static LDS only) the device holds at once; cached per (kernel, device)
static int coop_capacity(const void* kfn) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find({kfn, dev});
  if (it != cache.end()) return it->second;
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, 0) != hipSuccess) per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
  const int cap = per_cu * cus;
  cache[{kfn, dev}] = cap;
  return cap;
}

template <typename T, int NV>
static bool gn_coop_try(const GroupNormParams& p, hipStream_t s) {
  constexpr int EPV = 16 / (int)sizeof(T);
  const int VPR = p.C / EPV;
  if (VPR > 256 || p.groups > 256 || p.B > COOP_SYNC_SLOTS) return false;
  const int RPB = 256 / VPR, RC = RPB * NV;
  const int chunks = (p.rows + RC - 1) / RC;
  if (chunks > p.rows / 8 + 2 || chunks > 2048) return false;     // the workspace bound of groupnorm_ws_floats
  auto kfn = gn_coop_kernel<T, NV>;
  const long cap = coop_capacity(reinterpret_cast<const void*>(kfn));
  if ((long)p.B * chunks > cap * 3 / 4) return false;             // the whole grid must be co-resident, with room to spare
  hipLaunchKernelGGL(kfn, dim3((unsigned)chunks, (unsigned)p.B), dim3(256), 0, s, (const T*)p.x, p.ldx, (T*)p.y, p.ldy, p.gamma, p.beta,
                     p.partial, p.sync, p.rows, p.C, p.groups, p.eps, p.act, VPR, RPB, tuning().gn_coop_force_fb ? 1 : 0);
  return true;
}

size_t groupnorm_ws_floats(int B, int rows, int C, int groups) {
  // upper bound on B*chunks*groups*2 (two-launch path: chunks <= rows/16 + 1; cooperative kernel: <= rows/8 + 2) + B*C*2
  size_t chunks = (size_t)rows / 8 + 2;
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
  // Where the cooperative kernel wins (profiles/r4_c11_gn_coop_modes_ab_b1.txt, B = 1): against the (sample, group)-per-workgroup kernel
  // below on tensors with >= 1024 rows per sample (C=320 rows=4096 x13 0.438 -> 0.227 ms, C=640 rows=1024 x11 0.198 -> 0.162, C=1280
  // rows=1024 0.034 -> 0.023); at 256 / 64 rows that kernel's 9-13 us beat the ~10-us rendezvous, and against the two-launch path
  // (tensors > 8 MB: B >= 8) the cooperative kernel loses 5-20 % everywhere.  TANGO_GN_COOP_ALL=1 (tests) lifts the size rule.
  // round 6: the slab form where the shape takes it (levels 2-3 of the UNet: rows <= 256, groups of whole 16-byte vectors) -- a function of the shape
  // only, never of the batch (TANGO_GN_SLAB=0: A/B)
  if (tuning().gn_slab) {
    int G = 0, VPG = 0, nv = 0;
    if ((((uintptr_t)p.gamma | (uintptr_t)p.beta) & 15) == 0 && gn_slab_geom<T>(p.rows, p.C, p.groups, G, VPG, nv)) {
      if (nv == 4) gn_slab_go<T, 4>(p, G, VPG, s);
      else if (nv == 8) gn_slab_go<T, 8>(p, G, VPG, s);
      else gn_slab_go<T, 16>(p, G, VPG, s);
      TANGO_HIP(hipGetLastError());
      return 0;
    }
  }
  const bool coop_size = (size_t)p.B * p.rows * p.C * sizeof(T) <= ((size_t)8 << 20) && p.rows >= 1024;
  if (p.sync && !tuning().no_gn_coop && (coop_size || tuning().gn_coop_all)) {
    if (gn_coop_try<T, 8>(p, s)) { TANGO_HIP(hipGetLastError()); return 0; }      // (16 rows per thread: 128 VGPRs + spills at 4 waves / SIMD -- same bytes resident, not built)
  }
  if ((cg & 1) == 0 && (int64_t)p.rows * (cg / 2) <= (int64_t)1024 * GNS_NP && (p.ldx & 1) == 0 && (p.ldy & 1) == 0 &&
      (size_t)p.B * p.rows * p.C * sizeof(T) <= ((size_t)tuning().gn_small_mb << 20)) {
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


// ------------------------------------------------------------------------------------------
// GroupNorm folded into the linear that consumes it (round 6: Transformer2DModel.norm -> proj_in, transformer_2d.py:255-262).  GroupNorm is
// per (sample, group) affine, so proj_in(GroupNorm(x)) = Wf_s x + bf_s with per-SAMPLE weights Wf_s[n][k] = W[n][k] gamma[k] rstd[s][g(k)]
// and bf_s[n] = b[n] + sum_k W[n][k] beta[k] - sum_k Wf_s[n][k] mean[s][g(k)]: the normalised tensor is never written or read (one read + one
// write of the [M, C] stream less per site).  The mean term is summed against the ROUNDED weights the GEMM multiplies with, so it cancels
// exactly what the GEMM adds for the group means.  One wave per output row n of one sample; statistics finalised per workgroup from the
// chunk partials exactly as gn_apply_kernel does (double, fixed order).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_fold_kernel(const float* __restrict__ partial, const T* __restrict__ W, int64_t Kp,
                                                      const float* __restrict__ bias, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, T* __restrict__ Wf, float* __restrict__ bf,
                                                      int chunks, int groups, float eps, int rows, int C, int N) {
  __shared__ float mean_s[256], rstd_s[256];
  __shared__ double part_s[2][256], tot_s[2][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int cg = C / groups;
  int lanes = 256 / groups;
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
    const double mean = tot_s[0][tid] / n;
    double var = tot_s[1][tid] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_s[tid] = (float)mean;
    rstd_s[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  const T* wr = W + (int64_t)n * Kp;
  T* wo = Wf + ((int64_t)b * N + n) * Kp;
  float sb = 0.f, sm = 0.f;
  for (int k = lane; k < C; k += 64) {
    const int g = k / cg;
    const float w = to_f(wr[k]);
    const T wf = from_f<T>(w * (gamma[k] * rstd_s[g]));
    wo[k] = wf;
    sb += w * beta[k];
    sm += to_f(wf) * mean_s[g];
  }
  for (int k = C + lane; k < Kp; k += 64) wo[k] = from_f<T>(0.f);
  sb = wave_sum(sb); sm = wave_sum(sm);
  if (lane == 0) bf[(int64_t)b * N + n] = (bias ? bias[n] : 0.f) + sb - sm;
}

// ------------------------------------------------------------------------------------------
// GroupNorm statistics only (round 6): the chunk partials of gn_stats_kernel finalised to mean / rstd per (sample, group), for a consumer that
// normalises on its own way in (ff_fused.hip qkv_stat_kernel, norm mode 2: Transformer2DModel.norm -> proj_in without the normalised tensor).
// Same arithmetic as gn_apply_kernel's per-workgroup finalisation (double, fixed order), done ONCE per sample instead of once per workgroup.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ mr, int chunks, int groups,
                                                          float eps, int rows, int cg) {
  __shared__ double part_s[2][256];
  const int tid = threadIdx.x, b = blockIdx.x;
  int lanes = 256 / groups;
  if (lanes > 8) lanes = 8;
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
    const double n = (double)rows * cg;
    const double mean = a / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[((int64_t)b * groups + g) * 2] = (float)mean;
    mr[((int64_t)b * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

template <typename T> static int gn_stats_mr_t(const GroupNormParams& p, float* mr, hipStream_t s) {
  const GnGeom g = gn_geom<T>(p.B, p.rows, p.C);
  hipLaunchKernelGGL((gn_stats_kernel<T>), dim3((unsigned)g.chunks, (unsigned)p.B), dim3(256), 0, s, (const T*)p.x, p.ldx, p.partial, p.rows,
                     p.C, p.groups, g.VPR, g.TPR, g.RPB, g.RC);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)p.B), dim3(256), 0, s, p.partial, mr, g.chunks, p.groups, p.eps, p.rows, p.C / p.groups);
  TANGO_HIP(hipGetLastError());
  return 0;
}

bool gn_stats_mr_ok(int dtype, const GroupNormParams& p) {
  constexpr int EPV = 8;
  return dtype != DT_F32 && p.C % EPV == 0 && p.C <= GN_MAXC && p.C % p.groups == 0 && p.groups <= 256;
}

int launch_gn_stats_mr(int dtype, const GroupNormParams& p, float* mr, hipStream_t s) {
  if (!gn_stats_mr_ok(dtype, p)) TANGO_FAIL("gn_stats_mr: unsupported shape");
  if (dtype == DT_F16) return gn_stats_mr_t<f16>(p, mr, s);
  return gn_stats_mr_t<bf16>(p, mr, s);
}

bool gn_fold_ok(int dtype, const GroupNormParams& p, int N) {
  if (dtype == DT_F32 || !tuning().gn_fold) return false;
  constexpr int EPV = 8;
  if (p.C % EPV != 0 || p.C > GN_MAXC || p.C % p.groups != 0 || p.groups > 256 || p.act != ACT_NONE) return false;
  if (p.rows % 256 != 0 || p.C > 640) return false;               // tiles of 256 rows inside one sample; per-sample weights stay small (64 x 640^2 x 2 B = 52 MB)
  // only where GroupNorm would take its two-launch path (the one-launch kernels of the small-batch path are not split)
  return (size_t)p.B * p.rows * p.C * 2 > ((size_t)tuning().gn_small_mb << 20);
}

template <typename T>
static int gn_stats_fold_t(const GroupNormParams& p, const void* W, int64_t Kp, const float* bias, int N, void* Wf, float* bf, hipStream_t s) {
  const GnGeom g = gn_geom<T>(p.B, p.rows, p.C);
  hipLaunchKernelGGL((gn_stats_kernel<T>), dim3((unsigned)g.chunks, (unsigned)p.B), dim3(256), 0, s, (const T*)p.x, p.ldx, p.partial, p.rows,
                     p.C, p.groups, g.VPR, g.TPR, g.RPB, g.RC);
  hipLaunchKernelGGL((gn_fold_kernel<T>), dim3((unsigned)((N + 3) / 4), (unsigned)p.B), dim3(256), 0, s, p.partial, (const T*)W, Kp, bias,
                     p.gamma, p.beta, (T*)Wf, bf, g.chunks, p.groups, p.eps, p.rows, p.C, N);
  TANGO_HIP(hipGetLastError());
  return 0;
}

int launch_gn_stats_fold(int dtype, const GroupNormParams& p, const void* W, int64_t Kp, const float* bias, int N, void* Wf, float* bf, hipStream_t s) {
  if (!gn_fold_ok(dtype, p, N)) TANGO_FAIL("gn_stats_fold: unsupported shape");
  if (dtype == DT_F16) return gn_stats_fold_t<f16>(p, W, Kp, bias, N, Wf, bf, s);
  return gn_stats_fold_t<bf16>(p, W, Kp, bias, N, Wf, bf, s);
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

// statistics only (the consumer is a GEMM with the LayerNorm folded into its weights: gemm_wide.hip XS): one wave per row, read once
template <typename T>
__global__ __launch_bounds__(256) void ln_stats_kernel(const T* __restrict__ x, int64_t ldx, float* __restrict__ stats, int rows, int C, float eps) {
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
  if (lane == 0) *(f32x2*)(stats + (int64_t)row * 2) = f32x2{mean, rstd};
}

int launch_ln_stats(int dtype, const void* x, int64_t ldx, float* stats, int rows, int C, float eps, hipStream_t s) {
  const int epv = dtype == DT_F32 ? 4 : 8;
  if (C % epv != 0 || C / epv > 64 * LN_NV) TANGO_FAIL("ln_stats: unsupported C");
  const dim3 grid((unsigned)((rows + 3) / 4));
  switch (dtype) {
    case DT_F32: hipLaunchKernelGGL((ln_stats_kernel<float>), grid, dim3(256), 0, s, (const float*)x, ldx, stats, rows, C, eps); break;
    case DT_F16: hipLaunchKernelGGL((ln_stats_kernel<f16>), grid, dim3(256), 0, s, (const f16*)x, ldx, stats, rows, C, eps); break;
    case DT_BF16: hipLaunchKernelGGL((ln_stats_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)x, ldx, stats, rows, C, eps); break;
    default: TANGO_FAIL("ln_stats: bad dtype");
  }
  TANGO_HIP(hipGetLastError());
  return 0;
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
