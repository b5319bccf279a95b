// Micro-benchmark: how many bytes per clock per CU can a workgroup pull from L2 into LDS (global_load_lds_dwordx4) or into
// VGPRs (global_load_dwordx4) on MI355X?  The GEMM-shaped kernels of this repo stage operands with LDS-DMA; round 1
// measured ~13.6 B/clk/CU on the generic implicit-GEMM conv and took it for the ceiling.  This tool measures the ceiling
// itself, per access pattern, so tile shapes can be chosen against a known number.
//
// Patterns (each wave instruction moves 64 lanes x 16 B = 1 KiB):
//   contig : 1 KiB contiguous
//   rows128: 8 rows x 128 B, row stride S bytes (the K-major GEMM operand gather: one 128-byte line per row)
//   rows64 : 16 rows x 64 B
// The source window per workgroup is 256 KiB (L1-missing, L2-resident after the first sweep); all workgroups of an XCD
// share one 2 MiB region, so HBM is out of the picture.
// build: hipcc --offload-arch=gfx950 -O3 -o build/fill_bench tools/fill_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

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

// MODE 0: LDS-DMA, 1: VGPR loads.  PAT 0 contig, 1 rows128, 2 rows64.  DEPTH = instructions in flight per wave.
template <int MODE, int PAT, int DEPTH>
__global__ __launch_bounds__(1024) void fill_kernel(const unsigned char* __restrict__ src, unsigned* __restrict__ sink, int iters,
                                                    int row_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const unsigned char* base = src + (size_t)(blockIdx.x & 7) * (2u << 20);      // the XCD's 2 MiB region
  const unsigned win = (blockIdx.x >> 3) * 65536u;                               // this workgroup's window start inside it
  int lane_off;
  if (PAT == 0) lane_off = lane * 16;
  else if (PAT == 1) lane_off = (lane >> 3) * row_stride + (lane & 7) * 16;
  else lane_off = (lane >> 2) * row_stride + (lane & 3) * 16;
  u32x4 acc = u32x4{0u, 0u, 0u, 0u};
  const unsigned chunk = PAT == 1 ? 128u : 64u;       // bytes of one row consumed per instruction (row patterns)
  const unsigned rows = PAT == 1 ? 8u : 16u;
  unsigned kc = 0, rg = (unsigned)wave, cnt = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      unsigned off;
      if (PAT == 0) {
        off = win + (cnt * (unsigned)nw + (unsigned)wave) * 1024u + (unsigned)lane_off;
        ++cnt;
      } else {
        off = win + rg * rows * (unsigned)row_stride + kc * chunk + (unsigned)lane_off;
        if (++kc * chunk >= (unsigned)row_stride) { kc = 0; rg += (unsigned)nw; }
      }
      off &= ((2u << 20) - 1u) & ~15u;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(lds + (wave * DEPTH + d) * 1024), 16, 0, 0);
      } else {
        const u32x4 v = *(const u32x4*)(base + off);
        acc += v;
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (MODE == 0) {
    __syncthreads();
    acc[0] = *(const unsigned*)(lds + threadIdx.x * 4);
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345u) sink[0] = acc[0];
}

template <int MODE, int PAT, int DEPTH>
static void run(const char* name, const unsigned char* src, unsigned* sink, int threads, int blocks_per_cu, int row_stride, double clk_ghz) {
  const int iters = 2000;
  const int blocks = 256 * blocks_per_cu;
  const size_t lds = (size_t)(threads / 64) * DEPTH * 1024;
  auto k = fill_kernel<MODE, PAT, DEPTH>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, src, sink, 50, row_stride);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a, 0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, src, sink, iters, row_stride);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)blocks * (threads / 64) * DEPTH * 1024.0 * iters;
  printf("%-34s threads %4d x %d WG/CU depth %2d stride %5d: %8.1f GB/s chip = %6.2f B/clk/CU @ %.2f GHz  (%.3f ms)\n", name, threads,
         blocks_per_cu, DEPTH, row_stride, bytes / ms * 1e-6, bytes / (ms * 1e-3) / 256.0 / (clk_ghz * 1e9), clk_ghz, ms);
}

int main() {
  unsigned char* src;
  unsigned* sink;
  CHECK(hipMalloc(&src, 16u << 20));
  CHECK(hipMemset(src, 1, 16u << 20));
  CHECK(hipMalloc(&sink, 256));
  const double clk = 2.4;
  printf("# fill_bench: L2-resident source, per-CU fill rate by pattern (B/clk/CU quoted at the 2.4 GHz peak clock)\n");
  run<0, 0, 4>("lds-dma contiguous 1KiB", src, sink, 512, 1, 0, clk);
  run<0, 0, 8>("lds-dma contiguous 1KiB", src, sink, 512, 1, 0, clk);
  run<0, 0, 8>("lds-dma contiguous 1KiB", src, sink, 256, 2, 0, clk);
  run<0, 0, 8>("lds-dma contiguous 1KiB", src, sink, 1024, 1, 0, clk);
  run<0, 1, 4>("lds-dma 8 rows x 128 B", src, sink, 512, 1, 1024, clk);
  run<0, 1, 8>("lds-dma 8 rows x 128 B", src, sink, 512, 1, 1024, clk);
  run<0, 1, 8>("lds-dma 8 rows x 128 B", src, sink, 512, 1, 2048, clk);
  run<0, 1, 8>("lds-dma 8 rows x 128 B", src, sink, 512, 1, 8192, clk);
  run<0, 1, 8>("lds-dma 8 rows x 128 B", src, sink, 1024, 1, 2048, clk);
  run<0, 1, 8>("lds-dma 8 rows x 128 B", src, sink, 256, 2, 2048, clk);
  run<0, 2, 8>("lds-dma 16 rows x 64 B", src, sink, 512, 1, 2048, clk);
  run<1, 0, 8>("vgpr loads contiguous 1KiB", src, sink, 512, 1, 0, clk);
  run<1, 0, 8>("vgpr loads contiguous 1KiB", src, sink, 1024, 1, 0, clk);
  run<1, 1, 8>("vgpr loads 8 rows x 128 B", src, sink, 512, 1, 2048, clk);
  run<1, 2, 8>("vgpr loads 16 rows x 64 B", src, sink, 512, 1, 2048, clk);
  run<1, 2, 8>("vgpr loads 16 rows x 64 B", src, sink, 1024, 1, 2048, clk);
  return 0;
}
