// Write-bandwidth probe: how fast can the chip absorb the output of a GEMM epilogue?  Every workgroup (512 threads) writes one
// 256 x 160 fp16 tile of an [M][N] matrix with 16-byte stores, by pattern:
//   tile    : per wave 64 rows x 160-byte row segments, 16 rows per pass (the staged epilogue's store pattern)
//   rows    : each wave writes whole 320-byte tile rows (row-contiguous, two waves per row pair)
//   linear  : the workgroup's 80 KiB laid out contiguously (upper bound, not a usable layout)
// each with plain, nontemporal (nt) and sc1 sc0 stores.  N selects how far apart consecutive tile rows are.
// build: hipcc --offload-arch=gfx950 -O3 -o build/store_probe tools/store_probe.hip
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

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE> __device__ __forceinline__ void st16(unsigned char* p, const u32x4 v) {
  if (MODE == 0) *(u32x4*)p = v;
  else if (MODE == 1) __builtin_nontemporal_store(v, (u32x4*)p);
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <int PAT, int MODE>
__global__ __launch_bounds__(512) void store_kernel(unsigned char* out, int M, int N) {
  const int NT = N / 160;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / NT) * 256, n0 = (bid % NT) * 160;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32x4 v = u32x4{(unsigned)tid, (unsigned)bid, 0x3c003c00u, 0x3c003c00u};
  const long ldb = (long)N * 2;
  if (PAT == 0) {
    const int wm = wave & 3, wn = wave >> 2;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = lane; i < 160; i += 64) {
        const int r = i / 10, c = i % 10;
        st16<MODE>(out + (long)(m0 + wm * 64 + b * 16 + r) * ldb + (n0 + wn * 80) * 2 + c * 16, v);
      }
  } else if (PAT == 1) {
    // 20 pieces per 320-byte row; a wave covers 3.2 rows per instruction: 256 rows * 20 = 5120 pieces = 10 per thread
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int i = k * 512 + tid, r = i / 20, c = i % 20;
      st16<MODE>(out + (long)(m0 + r) * ldb + n0 * 2 + c * 16, v);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 10; ++k) st16<MODE>(out + (long)bid * 81920 + (k * 512 + tid) * 16, v);
  }
}

template <int PAT, int MODE> static void run(const char* name, unsigned char* out, int M, int N) {
  const int grid = (M / 256) * (N / 160);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((store_kernel<PAT, MODE>), dim3(grid), dim3(512), 0, 0, out, M, N);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((store_kernel<PAT, MODE>), dim3(grid), dim3(512), 0, 0, out, M, N);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 100.0, gb = (double)M * N * 2 / 1e9;
  printf("%-34s %8.1f us  %7.2f TB/s\n", name, us, gb / us * 1e-3 * 1e3);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 5120;
  unsigned char* out;
  CHECK(hipMalloc(&out, (size_t)M * N * 2));
  printf("store probe: [%d][%d] fp16 = %.0f MB, %d workgroups\n", M, N, (double)M * N * 2 / 1e6, (M / 256) * (N / 160));
  run<0, 0>("tile pattern, plain", out, M, N);
  run<0, 1>("tile pattern, nontemporal", out, M, N);
  run<0, 2>("tile pattern, sc0 sc1", out, M, N);
  run<1, 0>("tile rows, plain", out, M, N);
  run<1, 1>("tile rows, nontemporal", out, M, N);
  run<2, 0>("linear, plain", out, M, N);
  run<2, 1>("linear, nontemporal", out, M, N);
  return 0;
}
