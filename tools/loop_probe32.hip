// Round-4 probe: the 256 x 320 ping-pong main loop of gemm_wide.hip (8 waves of 64 x 160, 64-byte k-chunks, four LDS-DMA stages)
// with its 40 v_mfma_f32_16x16x32_f16 per chunk replaced by 20 v_mfma_f32_32x32x16_f16 (same fragment reads: 10 + 4 ds_read_b128,
// same accumulator count), on RANDOM fp16 data (constant data clocks ~15 % higher), DMAs either at the head of the multiply part
// (SCH 0) or inside the MFMA stream (SCH 1).  Results are not checked (the 32x32 variant reads fragments in the 16x16 layout:
// garbage by construction); only the time matters.  MI355X_MICROARCH.md: 16x16x32 issues every ~17 cycles, 32x32x16 every 32.
// build: hipcc --offload-arch=gfx950 -O3 -o build/loop_probe32 tools/loop_probe32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void bar() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

template <bool M32, int SCH>
__global__ __launch_bounds__(512) void probe(const unsigned char* __restrict__ A, const unsigned char* __restrict__ W, float* __restrict__ out,
                                             int M, int N, int Kbytes) {
  constexpr int BM = 256, BN = 320, CB = 64, NST = 4, ROWS = BM + BN, STAGE = ROWS * CB, RG = ROWS / 16, RGW = (RG + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int NT = N / BN;
  int bid = blockIdx.x;
  { const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
  const int m0 = (bid / NT) * BM, n0 = (bid % NT) * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int lrow = lane >> 2, pc = (lane & 3) ^ ((4 - (lrow >> 2)) & 3);
  long r_base[RGW];
#pragma unroll
  for (int i = 0; i < RGW; ++i) {
    const int row = (wave + 8 * i) * 16 + lrow;
    r_base[i] = i < 2 ? (long)(m0 + row) * Kbytes + pc * 16 : (long)(n0 + row - BM) * Kbytes + pc * 16;
  }
  const int my_count = wave < RG - 8 * (RGW - 1) ? RGW : RGW - 1;
  auto issue_one = [&](int i, int kc, int st) {
    const int rg = wave + 8 * i;
    if (rg < RG) {
      const unsigned char* src = (i < 2 ? A : W) + r_base[i] + (long)kc * CB;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dsm + st * STAGE + rg * 1024), 16, 0, 0);
    }
  };
  auto wait_inflight = [&](int chunks) {
    if (chunks <= 0) { wait_vm<0>(); return; }
    if (my_count == RGW) wait_vm<RGW>(); else wait_vm<RGW - 1>();
  };
  f32x4 acc[10][4];
  f32x16 acc32[5][2];
#pragma unroll
  for (int a = 0; a < 10; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc32[a][b][r] = 0.f;
  const int nk = Kbytes / CB;
  const int l15 = lane & 15, g = lane >> 4;
  const int foff = l15 * CB + ((g ^ ((4 - (l15 >> 2)) & 3)) * 16);
  const int xrow = (wm * 64) * CB + foff, wrow = (BM + wn * 160) * CB + foff;
  for (int c = 0; c < NST - 1 && c < nk; ++c)
#pragma unroll
    for (int i = 0; i < RGW; ++i) issue_one(i, c, c);
  const int half = wave >> 2;
  if (my_count == RGW) wait_vm<2 * RGW>(); else wait_vm<2 * (RGW - 1)>();
  bar();
  if (half) bar();
  int st = 0;
  for (int kc = 0; kc < nk; ++kc) {
    const unsigned char* Xs = dsm + st * STAGE;
    const bool more = kc + NST - 1 < nk;
    const int st3 = st == 0 ? NST - 1 : st - 1;
    u32x4 wf[10], xf[4];
#pragma unroll
    for (int a = 0; a < 10; ++a) wf[a] = *(const u32x4*)(Xs + wrow + a * 16 * CB);
#pragma unroll
    for (int b = 0; b < 4; ++b) xf[b] = *(const u32x4*)(Xs + xrow + b * 16 * CB);
    if (kc + 1 < nk) wait_inflight(kc + 2 < nk ? 1 : 0);
    bar();
    if (SCH == 0 && more) {
#pragma unroll
      for (int i = 0; i < RGW; ++i) issue_one(i, kc + NST - 1, st3);
    }
    __builtin_amdgcn_s_setprio(1);
    if (!M32) {
#pragma unroll
      for (int a = 0; a < 10; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, wf[a]), __builtin_bit_cast(h8, xf[b]), acc[a][b], 0, 0, 0);
        if (SCH == 1 && (a & 1) == 0 && (a >> 1) < RGW) {
          __builtin_amdgcn_sched_barrier(0);
          if (more) issue_one(a >> 1, kc + NST - 1, st3);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      // 5 x 2 tiles of 32 x 32, two k-steps of 16: fragment (tile, k-step) = one ds_read_b128 each -> wf[2 ta + ks], xf[2 tb + ks]
#pragma unroll
      for (int ta = 0; ta < 5; ++ta) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int tb = 0; tb < 2; ++tb)
            acc32[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, wf[2 * ta + ks]), __builtin_bit_cast(h8, xf[2 * tb + ks]), acc32[ta][tb], 0, 0, 0);
        if (SCH == 1 && ta < RGW) {
          __builtin_amdgcn_sched_barrier(0);
          if (more) issue_one(ta, kc + NST - 1, st3);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    bar();
    st = st == NST - 1 ? 0 : st + 1;
  }
  if (!half) bar();
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 10; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc32[a][b][r];
  out[(size_t)blockIdx.x * 512 + tid] = s;      // one dword per thread: negligible next to the loop
}

template <bool M32, int SCH>
static void run(const char* name, const unsigned char* A, const unsigned char* W, float* out, int M, int N, int Kb) {
  constexpr int LDS = 4 * (256 + 320) * 64;
  auto k = probe<M32, SCH>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  const int grid = (M / 256) * (N / 320);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  double best = 1e30, sum = 0;
  const int rounds = 5, reps = 5;
  for (int r = 0; r < rounds + 1; ++r) {
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS, 0, A, W, out, M, N, Kb);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r == 0) continue;
    const double us = ms * 1000.0 / reps; sum += us; if (us < best) best = us;
  }
  const double us = sum / rounds;
  printf("  %-44s %9.1f us mean (%7.1f best)  %6.0f TF/s\n", name, us, best, 2.0 * M * N * (Kb / 2.0) / (us * 1e-6) / 1e12);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 640, Kb = argc > 3 ? atoi(argv[3]) : 11520;
  unsigned char *A, *W; float* out;
  CHECK(hipMalloc(&A, (size_t)M * Kb)); CHECK(hipMalloc(&W, (size_t)N * Kb)); CHECK(hipMalloc(&out, (size_t)(M / 256) * (N / 320) * 512 * 4));
  {  // random fp16 in (-1, 1): sign, exponent 0x30..0x3b, random mantissa
    std::vector<unsigned short> h((size_t)(M > N ? M : N) * Kb / 2);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 16) & 0x8000u) | ((0x30u + ((x >> 8) % 12u)) << 10) | ((x >> 20) & 0x3ffu)); }
    CHECK(hipMemcpy(A, h.data(), (size_t)M * Kb, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(W, h.data() + 777, (size_t)N * Kb, hipMemcpyHostToDevice));
  }
  printf("loop probe 32: M=%d N=%d K=%d fp16 random data, 256x320 tile, %d workgroups, %d chunks\n", M, N, Kb / 2, (M / 256) * (N / 320), Kb / 64);
  for (int rep = 0; rep < 2; ++rep) {
    run<false, 0>("16x16x32, DMAs at the head (SCH 0)", A, W, out, M, N, Kb);
    run<false, 1>("16x16x32, DMAs in the MFMA stream (SCH 1)", A, W, out, M, N, Kb);
    run<true, 0>("32x32x16, DMAs at the head", A, W, out, M, N, Kb);
    run<true, 1>("32x32x16, DMAs in the MFMA stream", A, W, out, M, N, Kb);
  }
  return 0;
}
