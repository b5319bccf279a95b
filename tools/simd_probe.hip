// Which SIMD does each wave of a 512-thread workgroup land on?  (the ping-pong kernels want the two waves of a SIMD in
// opposite phases)  Prints, for a few workgroups, the SIMD id (HW_REG_HW_ID bits 5:4) and CU id of waves 0..7.
// build: hipcc --offload-arch=gfx950 -O2 -o build/simd_probe tools/simd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void probe(unsigned* out, int lds_bytes) {
  extern __shared__ unsigned char dummy[];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, 32 bits
    out[blockIdx.x * 8 + wave] = hw;
  }
  if (lds_bytes < 0) dummy[threadIdx.x] = 0;
}
int main() {
  unsigned* d;
  const int blocks = 512;
  hipMalloc(&d, blocks * 8 * 4);
  for (int lds : {0, 150 * 1024}) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), lds, 0, d, lds);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    printf("# dynamic LDS %d bytes: wave -> simd (cu) for the first 12 workgroups, then a histogram of the pairing pattern\n", lds);
    int pair_w4 = 0, pair_adj = 0, other = 0;
    for (int b = 0; b < blocks; ++b) {
      int simd[8];
      for (int w = 0; w < 8; ++w) simd[w] = (h[b * 8 + w] >> 4) & 3;
      if (b < 12) {
        printf("wg %3d:", b);
        for (int w = 0; w < 8; ++w) printf(" w%d->s%d(cu%u,slot%u)", w, simd[w], (h[b * 8 + w] >> 8) & 15, h[b * 8 + w] & 15);
        printf("\n");
      }
      bool w4 = true, adj = true;
      for (int w = 0; w < 4; ++w) w4 &= simd[w] == simd[w + 4];
      for (int w = 0; w < 8; w += 2) adj &= simd[w] == simd[w + 1];
      if (w4) ++pair_w4; else if (adj) ++pair_adj; else ++other;
    }
    printf("pattern: (w, w+4) share a SIMD in %d workgroups, (2k, 2k+1) share in %d, other %d of %d\n", pair_w4, pair_adj, other, blocks);
  }
  return 0;
}
