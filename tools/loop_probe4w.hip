// Candidate main loop for the 256 x 320 LDS-DMA GEMM: FOUR waves of 128 x 160 (one per SIMD, 320 accumulator registers each)
// instead of eight waves of 64 x 160.  Why: tools/loop_probe.hip showed the LDS port to be the co-critical resource of the
// 8-wave loop -- per 64-byte k-chunk the eight waves read 8 x 14 KiB of fragments and the DMA engine writes 36 KiB, 148 KiB per
// 1280 MFMA cycles = 116 B/clk of the 128 B/clk port.  A 128 x 160 wave tile needs 18 fragments per 80 MFMAs (0.225 per MFMA
// instead of 0.35): 4 x 18 + 36 = 108 KiB per chunk, 84 B/clk.  With one wave per SIMD nothing hides an LDS round trip, so the
// fragment reads are software-pipelined by hand: inline-asm ds_read_b128 with counted lgkmcnt waits (hipcc would wait
// lgkmcnt(0) before every use while LDS DMAs are pending, see tango_amd/csrc/xattn.hip), the B fragments of chunk c+1 and the
// A fragments three MFMA groups ahead are in flight under the MFMAs of chunk c.  One barrier per chunk.
// The probe checks the candidate BIT-FOR-BIT against the 8-wave loop of loop_probe.hip on pseudo-random data (a wrong wait
// count reads a stale stage and shows up as a mismatch), then times both with the same per-wave epilogue emulation.
// build: hipcc --offload-arch=gfx950 -O3 -o build/loop_probe4w tools/loop_probe4w.hip
#include <type_traits>
#define main loop_probe_main
#include "loop_probe.hip"
#undef main

template <int OFF> __device__ __forceinline__ u32x4 lds_read_async(const unsigned base) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void lds_wait(u32x4& frag) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N)); }
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// MFMA from inline asm with the accumulator pinned to its register file.  320 accumulator registers = the 256 AGPRs + 64 VGPRs;
// left to itself hipcc rotates accumulator tuples (dst != srcC) and moves them between the files at the loop edges
// (1700 v_accvgpr_* and 185 scratch instructions in the first version of this probe).  Tiles 0..63 live in AGPRs, 64..79 in VGPRs.
template <bool AG> __device__ __forceinline__ void mma_pinned(f32x4& c, const u32x4& a, const u32x4& b) {
  if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// reads issued after fragment A(c, b) when group b starts (see the schedule in chunk_body)
__host__ __device__ constexpr int wait_steady(int b) { constexpr int w[8] = {2, 4, 6, 8, 8, 8, 6, 4}; return w[b]; }
__host__ __device__ constexpr int wait_last(int b) { return b < 6 ? 2 : 7 - b; }

template <int ABL>
__global__ __launch_bounds__(256) void probe4w_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ W,
                                                      float* __restrict__ out, int M, int N, int Kbytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  constexpr int BM_ = 256, BN_ = 320, CB = 64, NST = 4, ROWS_ = BM_ + BN_, STAGE_ = ROWS_ * CB, RG_ = ROWS_ / 16, RGW_ = RG_ / 4;
  constexpr int TM_ = 8, TN_ = 10;
  constexpr bool NO_DMA = ABL & 1;
  const int NT = N / BN_;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / NT) * BM_, n0 = (bid % NT) * BN_;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int lrow = lane >> 2, pc = (lane & 3) ^ ((4 - (lrow >> 2)) & 3);
  long r_base[RGW_];
#pragma unroll
  for (int i = 0; i < RGW_; ++i) {
    const int row = (wave + 4 * i) * 16 + lrow;
    r_base[i] = row < BM_ ? (long)(m0 + row) * Kbytes + pc * 16 : (long)(n0 + row - BM_) * Kbytes + pc * 16;
  }
  auto issue_chunk = [&](int kc, int st) {
#pragma unroll
    for (int i = 0; i < RGW_; ++i) {
      const int rg = wave + 4 * i;
      const unsigned char* src = (rg * 16 >= BM_ ? W : A) + r_base[i] + (long)kc * CB;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dsm + st * STAGE_ + rg * 1024), 16, 0, 0);
    }
  };
  f32x4 acc[TN_][TM_];
#pragma unroll
  for (int a = 0; a < TN_; ++a)
#pragma unroll
    for (int b = 0; b < TM_; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = Kbytes / CB;
  const int l15 = lane & 15, g = lane >> 4;
  const int foff = l15 * CB + ((g ^ ((4 - (l15 >> 2)) & 3)) * 16);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
  const unsigned xoff = lds0 + (wm * TM_ * 16) * CB + foff;          // + stage * STAGE_ + b * 1024
  const unsigned woff = lds0 + (BM_ + wn * TN_ * 16) * CB + foff;    // + stage * STAGE_ + a * 1024
  for (int c = 0; c < NST - 1 && c < nk; ++c) issue_chunk(c, c);

  u32x4 wf0[TN_], wf1[TN_], xf[4];
  // prime: chunk 0 landed everywhere, then B(0, 0..9), A(0, 0..2) in the order the steady state leaves them
  if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RGW_) : "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RGW_) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  static_for<0, TN_>([&](auto a) { wf0[a] = lds_read_async<a * 1024>(woff); });
  static_for<0, 3>([&](auto b) { xf[b] = lds_read_async<b * 1024>(xoff); });
  __builtin_amdgcn_sched_barrier(0);

  // one chunk: 8 groups of 10 MFMAs (group b = A fragment b against the ten B fragments).  Group b waits for ITS A fragment
  // (counted: wait_steady / wait_last = the reads issued after it), then issues its reads -- the A fragment three groups ahead
  // (of this chunk, or of the next one for b >= 5) and, in groups 0..4, two B fragments of the next chunk -- then its MFMAs.
  auto chunk_body = [&](auto last_tag, const int c, u32x4(&cur)[TN_], u32x4(&nxt)[TN_]) {
    constexpr bool LAST = decltype(last_tag)::value;
    const int st = c & (NST - 1), st1 = (c + 1) & (NST - 1);
    // chunk c+1 (this wave's share) landed; after the barrier every wave's share has, and every wave is done with chunk c-1
    if (!LAST) {
      if (c + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RGW_) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (!NO_DMA && c + NST - 1 < nk) issue_chunk(c + NST - 1, (c + NST - 1) & (NST - 1));
    const unsigned xs = xoff + st * STAGE_, xs1 = xoff + st1 * STAGE_, ws1 = woff + st1 * STAGE_;
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, TM_>([&](auto bb) {
      constexpr int b = bb;
      lds_wait<LAST ? wait_last(b) : wait_steady(b)>(xf[b & 3]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (b + 3 < TM_) xf[(b + 3) & 3] = lds_read_async<(b + 3) * 1024>(xs);
      else if constexpr (!LAST) xf[(b + 3) & 3] = lds_read_async<(b + 3 - TM_) * 1024>(xs1);
      if constexpr (!LAST && b < 5) {
        nxt[2 * b] = lds_read_async<(2 * b) * 1024>(ws1);
        nxt[2 * b + 1] = lds_read_async<(2 * b + 1) * 1024>(ws1);
      }
      static_for<0, TN_>([&](auto aa) { constexpr int a = aa; mma_pinned<(b * TN_ + a) < 64>(acc[a][b], cur[a], xf[b & 3]); });
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  int c = 0;
  for (; c + 2 < nk; c += 2) {
    chunk_body(std::false_type{}, c, wf0, wf1);
    chunk_body(std::false_type{}, c + 1, wf1, wf0);
  }
  if (nk - c == 2) {
    chunk_body(std::false_type{}, c, wf0, wf1);
    chunk_body(std::true_type{}, c + 1, wf1, wf0);
  } else {
    chunk_body(std::true_type{}, c, wf0, wf1);
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");                 // last MFMA -> accumulator reads below (asm MFMAs: no compiler hazard padding)
  __syncthreads();
  epilogue_emul<TM_, TN_>(acc, dsm + wave * (16 * (TN_ * 32 + 16)), (_Float16*)out, N, m0 + wm * TM_ * 16, n0 + wn * TN_ * 16, lane);
}

__global__ void fill_kernel(unsigned short* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const __half v = __float2half((float)((int)(h & 15) - 8) * 0.0625f);     // multiples of 1/16: every partial sum is exact
    p[i] = __half_as_ushort(v);
  }
}
__global__ void diff_kernel(const unsigned* a, const unsigned* b, size_t n, unsigned long long* cnt) {
  unsigned long long local = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) local += a[i] != b[i];
  if (local) atomicAdd(cnt, local);
}

template <typename K> static double time_kernel(K k, int grid, int block, int lds, const unsigned char* A, const unsigned char* W, float* out, int M, int N, int Kb) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, A, W, out, M, N, Kb);
  CHECK(hipDeviceSynchronize());
  const int reps = 10;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, A, W, out, M, N, Kb);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.0 / reps;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 262144, N = argc > 2 ? atoi(argv[2]) : 320, Kb = argc > 3 ? atoi(argv[3]) : 2560;
  if (M % 256 || N % 320 || Kb % 64) { printf("shape does not tile\n"); return 1; }
  unsigned char *A, *W;
  float *o8, *o4;
  unsigned long long* cnt;
  CHECK(hipMalloc(&A, (size_t)M * Kb));
  CHECK(hipMalloc(&W, (size_t)N * Kb));
  CHECK(hipMalloc(&o8, (size_t)M * N * 2));
  CHECK(hipMalloc(&o4, (size_t)M * N * 2));
  CHECK(hipMalloc(&cnt, 8));
  fill_kernel<<<2048, 256>>>((unsigned short*)A, (size_t)M * Kb / 2, 1u);
  fill_kernel<<<256, 256>>>((unsigned short*)W, (size_t)N * Kb / 2, 77u);
  CHECK(hipMemset(o8, 0, (size_t)M * N * 2));
  CHECK(hipMemset(o4, 0xff, (size_t)M * N * 2));
  CHECK(hipMemset(cnt, 0, 8));
  constexpr int LDS = 4 * (256 + 320) * 64;
  const int grid = (M / 256) * (N / 320);
  auto k8 = probe2_kernel<256, 320, 4, 2, 4, 32, 1>;
  auto k4 = probe4w_kernel<0>;
  auto k4n = probe4w_kernel<1>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k4n), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipLaunchKernelGGL(k8, dim3(grid), dim3(512), LDS, 0, A, W, o8, M, N, Kb, 0);
  hipLaunchKernelGGL(k4, dim3(grid), dim3(256), LDS, 0, A, W, o4, M, N, Kb);
  CHECK(hipDeviceSynchronize());
  diff_kernel<<<1024, 256>>>((const unsigned*)o8, (const unsigned*)o4, (size_t)M * N / 2, cnt);
  unsigned long long h = 0;
  CHECK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
  unsigned short probe[4];
  CHECK(hipMemcpy(probe, o8, 8, hipMemcpyDeviceToHost));
  printf("loop probe 4w: M=%d N=%d K=%d fp16, %d workgroups, %d chunks; 4-wave vs 8-wave output: %llu mismatching words of %zu (o8[0..1] = %04x %04x)\n",
         M, N, Kb / 2, grid, Kb / 64, h, (size_t)M * N / 2, probe[0], probe[1]);
  const double flop = 2.0 * M * N * (Kb / 2.0);
  {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k8, dim3(grid), dim3(512), LDS, 0, A, W, o8, M, N, Kb, 0);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("  8 waves of 64x160, ping-pong, 4 stages + epilogue   %9.1f us  %6.0f TF-equivalent\n", ms * 100.0, flop / (ms * 1e-4) / 1e12);
    }
  }
  const double us4 = time_kernel(k4, grid, 256, LDS, A, W, o4, M, N, Kb);
  printf("  4 waves of 128x160, counted LDS pipeline + epilogue  %9.1f us  %6.0f TF-equivalent\n", us4, flop / (us4 * 1e-6) / 1e12);
  const double us4n = time_kernel(k4n, grid, 256, LDS, A, W, o4, M, N, Kb);
  printf("  ... without the DMA inside the loop                  %9.1f us  %6.0f TF-equivalent\n", us4n, flop / (us4n * 1e-6) / 1e12);
  return 0;
}
