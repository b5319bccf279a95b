// 3x3 / stride 1 / pad 1 convolution (optionally over a nearest-x2 upsampled input) over channels-last activations
// with HALO REUSE (gfx950).
//
// Why: the generic implicit-GEMM kernel (gemm.hip) re-gathers the 256-pixel activation tile once per tap, i.e. 9x per
// 128-byte channel chunk.  PMC (profiles/r1_v16_pmc_conv_l0.txt) shows that kernel waiting on LDS-DMA arrival:
// ~13.6 B/clk/CU of global->LDS fill is the ceiling it runs into (MFMA pipes 28 % busy, 0 LDS bank conflicts, L2 hit
// rate 92 %).
// Here each workgroup stages the (rows+2) x (W+2) halo of its 256 output pixels ONCE per channel chunk and serves
// all nine taps from it by shifting the fragment row index; only the 9 weight chunks stream per channel chunk.
// Global->LDS bytes per MFMA flop drop 2.1x (activation part 5.8x).
//
// Structure: 512 threads = 8 waves (4 along pixels x 2 along channels), wave tile 64 x BN/2, "swapped" MFMA
// orientation as in gemm.hip (weights = A operand).  LDS: two halo buffers (channel chunk cc and cc+1) + a 3-stage
// ring of weight chunks, all 128-byte rows with the XOR swizzle applied on the DMA source side.  One raw s_barrier
// per (chunk, tap) item with counted vmcnt waits; the next chunk's halo is trickled in one DMA piece per wave per
// item.  Out-of-image halo pixels are sourced from a zero page.
//
// Reference op replaced: the ResnetBlock2D / VAE ResnetBlock 3x3 convolutions (diffusers/src/diffusers/models/
// resnet.py:445-552, audioldm/variational_autoencoder/modules.py:117-176), i.e. ATen convolution.
#include <cstdlib>

#include "common.h"
#include "tuning.h"
#include "gemm_device.h"

namespace tango {

static constexpr int HALO_MAX_ROWS = 400;   // halo pixels per tile the LDS budget allows (2 buffers)
static constexpr int HALO_NA = 7;           // halo DMA pieces per wave per channel chunk (8 waves x 7 x 8 rows >= 400)

// ABL = true compiles the ablation hooks in (TANGO_HALO_ABL, tools/halo_ablation.sh); the product instantiation has none.
// PP = true selects the "ping-pong" main loop (round 2): the two 4-wave halves of the workgroup (waves w and w + 4 share
// a SIMD) run HALF AN ITEM OUT OF PHASE.  Every (chunk, tap) item is two phases (64-byte k-groups), every phase is
//   [ds_read the 9 fragments] s_barrier [20 MFMAs] s_barrier
// and half B executes one extra barrier up front, so while one half of a SIMD's waves multiplies, the other half has its
// LDS reads (and the DMA issue) in flight: the lock-step structure measured 45 % MFMA-busy inside the loop with the LDS
// read latency exposed once per barrier (ablation: 117 of 553 us), here it hides behind the partner's MFMAs.
// LDS-DMA protocol (3 weight stages, 2 halo buffers as before): the DMAs of item i+2 are issued at the START of the MFMA
// part of phase (i, 0) -- by then every wave has retired its reads of item i-1 (whose stage is being refilled): the last
// ones are half B's phase-(i-1, 1) reads, consumed by its MFMAs one barrier ago --, and each wave waits for its own
// item-(i+1) DMAs in the read part of phase (i, 1), i.e. at least one barrier before the first read of item i+1.
template <typename T, int BN, bool ABL, bool PP = false>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const GemmParams p, const unsigned char* zero_page, const int SR,
                                                           const int nseg, const int abytes, const int abl_arg, const int staged,
                                                           const int pp_mode) {
  const int abl = ABL ? abl_arg : 0;
  constexpr int EPV = 16 / (int)sizeof(T);
  constexpr int BM = 256, BKB = 128;
  constexpr int BK = BKB / (int)sizeof(T);
  constexpr int WST = BN * BKB;                 // bytes per weight stage
  constexpr int WRG = BN / 8;                   // 8-row DMA groups per weight chunk
  constexpr int WRGW = (WRG + 7) / 8;           // ... per wave (some waves own one fewer)
  constexpr int WNR = BN / 2;
  constexpr int TM = 4, TN = WNR / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];   // [halo 0 | halo 1 | W stage 0..2]
  unsigned char* const As = dsm;
  unsigned char* const Ws = dsm + 2 * abytes;

  const int NT = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / NT) * BM, n0 = (bid % NT) * BN;
  const unsigned char* Ab = (const unsigned char*)p.A;
  const unsigned char* Wb = (const unsigned char*)p.W;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int lrow = lane >> 3, slot = lane & 7;
  const int pc = slot ^ lrow;

  // tile geometry: 256 consecutive pixels = SR image rows of one image (nseg == 1) or nseg whole images
  const int H = p.H, Wd = p.Wd, hw = H * Wd;
  const int HW2 = Wd + 2, SEG = (SR + 2) * HW2;
  const int HALO = nseg * SEG, HALO_RG = (HALO + 7) >> 3;
  const int b0 = m0 / hw;
  const int y0 = nseg == 1 ? (m0 - b0 * hw) / Wd : 0;

  // halo DMA sources of this lane: piece t covers halo rows (t*8 + wave)*8 + lrow
  int64_t a_src[HALO_NA];
#pragma unroll
  for (int t = 0; t < HALO_NA; ++t) {
    a_src[t] = -1;
    const int h = ((t * 8 + wave) << 3) + lrow;
    if (h < HALO) {
      const int seg = h / SEG, rem = h - seg * SEG;
      const int hy = rem / HW2, hx = rem - hy * HW2;
      const int y = y0 + hy - 1, x = hx - 1;
      // fused nearest x2 upsampling (p.ups): the halo lives on the upsampled grid, pixel (y, x) reads source (y>>1, x>>1)
      if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)Wd)
        a_src[t] = ((((int64_t)(b0 + seg) * p.Hin + (y >> p.ups)) * p.Win + (x >> p.ups)) * p.lda + pc * EPV) * (int64_t)sizeof(T);
    }
  }
  // weight DMA sources
  int64_t w_src[WRGW];
  int my_w = 0;
#pragma unroll
  for (int i = 0; i < WRGW; ++i) {
    const int rg = wave + 8 * i;
    w_src[i] = 0;
    if (rg < WRG) {
      ++my_w;
      w_src[i] = ((int64_t)(n0 + rg * 8 + lrow) * p.Kp + pc * EPV) * (int64_t)sizeof(T);
      if constexpr (BN == 32) { if (n0 + rg * 8 + lrow >= p.N) w_src[i] = -1; }    // narrow outputs (conv_out, N = 8): rows beyond N come from the zero page
    }
  }
  my_w = __builtin_amdgcn_readfirstlane(my_w);

  auto issue_a = [&](const int t, const int cc, const int buf) {   // t compile-time after unrolling
    const int ag = t * 8 + wave;
    if (ag < HALO_RG) {
      const unsigned char* src = a_src[t] >= 0 ? Ab + a_src[t] + (int64_t)cc * BKB : zero_page;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + buf * abytes + ag * 1024), 16, 0, 0);
    }
  };
  auto issue_w = [&](const int tap, const int cc, const int st) {
    const int64_t koff = ((int64_t)tap * p.Cin + (int64_t)cc * BK) * (int64_t)sizeof(T);
#pragma unroll
    for (int i = 0; i < WRGW; ++i) {
      const int rg = wave + 8 * i;
      if (rg < WRG) {
        const unsigned char* src = Wb + w_src[i] + koff;
        if constexpr (BN == 32) { if (w_src[i] < 0) src = zero_page; }
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ws + st * WST + rg * 1024), 16, 0, 0);
      }
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addressing
  int h00[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int pm = wm * 64 + b * 16 + (lane & 15);
    const int seg = pm / hw, r = pm - seg * hw;      // nseg == 1: hw >= 256 > pm -> seg = 0
    const int ly = r / Wd, x = r - ly * Wd;
    h00[b] = seg * SEG + ly * HW2 + x;
  }
  const int kg = lane >> 4;
  int wkoff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) wkoff[ks] = (((ks * 4 + kg) ^ (lane & 7)) * 16);
  const int wrow = (wn * WNR + (lane & 15)) * BKB;

  const int NC = p.Cin / BK;

  if constexpr (PP) {
    const int half = pp_phase_half(wave, lane, (unsigned*)(Ws + 2 * WST), pp_mode);   // scratch = head of weight stage 2 (first DMA'd two barriers later);   // 0: leads, 1: trails by one slot
    // prologue: halo of chunk 0, weight items 0 and 1; item 0 (and the halo) must have landed before the first read
#pragma unroll
    for (int t = 0; t < HALO_NA; ++t) issue_a(t, 0, 0);
    issue_w(0, 0, 0);
    issue_w(1, 0, 1);
    wait_vmcnt_upto8(my_w);                           // leaves only item 1 in flight
    pp_barrier();
    if (half) pp_barrier();                           // the stagger: half B starts one slot late
    int st = 0;
    for (int cc = 0; cc < NC; ++cc) {
      const unsigned char* Ah = As + (cc & 1) * abytes;
      const bool more_c = cc + 1 < NC;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const unsigned char* Wst = Ws + st * WST;
        const int toff = (tap / 3) * HW2 + (tap % 3);
        int hb[TM];
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          hb[b] = h00[b];
          asm volatile("" : "+v"(hb[b]));
        }
        const bool has2 = (tap + 2 < 9) || more_c;    // item i+2 exists
        const bool has1 = (tap + 1 < 9) || more_c;    // item i+1 exists
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          // ---- read part ----
          u32x4 wf[TN], xf[TM];
#pragma unroll
          for (int a = 0; a < TN; ++a) wf[a] = *(const u32x4*)(Wst + wrow + a * 16 * BKB + wkoff[ks]);
#pragma unroll
          for (int b = 0; b < TM; ++b) {
            const int h = hb[b] + toff;
            xf[b] = *(const u32x4*)(Ah + h * BKB + (((ks * 4 + kg) ^ (h & 7)) << 4));
          }
          if (ks == 1) {
            // item i+1 (issued one item ago) must have landed before anybody reads it; item i+2's DMAs (issued in phase 0
            // of this item, AFTER an optional halo piece) may stay in flight
            if (has1) wait_vmcnt_upto8(has2 ? my_w : 0);
          }
          pp_barrier();
          // ---- multiply part ----
          if (ks == 0) {
            if (more_c && tap < HALO_NA) issue_a(tap, cc + 1, (cc + 1) & 1);
            const int t2 = tap + 2;
            const int st2 = st == 0 ? 2 : st - 1;     // (st + 2) % 3
            if (t2 < 9) issue_w(t2, cc, st2);
            else if (more_c) issue_w(t2 - 9, cc + 1, st2);
          }
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
          __builtin_amdgcn_s_setprio(0);
          pp_barrier();
        }
        st = st == 2 ? 0 : st + 1;
      }
    }
    if (!half) pp_barrier();
  } else {
  // prologue: halo of chunk 0, weight items 0 and 1
  if (!(abl & 32)) {
#pragma unroll
  for (int t = 0; t < HALO_NA; ++t) issue_a(t, 0, 0);
  issue_w(0, 0, 0);
  issue_w(1, 0, 1);
  }

  int st = 0;   // weight stage of the current item
  for (int cc = 0; cc < NC; ++cc) {
    const unsigned char* Ah = As + (cc & 1) * abytes;
    const bool more_c = cc + 1 < NC;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const bool last_item = !more_c && tap == 8;
      // weight item (cc, tap) must have landed; item +1 (issued one item ago) may stay in flight
      if (!last_item) {
        if (my_w == WRGW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WRGW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WRGW - 1) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      // refill: one halo piece of the next chunk (older in the queue than the weights issued after it), then weight item +2
      if (more_c && tap < HALO_NA && !(abl & 2)) issue_a(tap, cc + 1, (cc + 1) & 1);
      {
        const int t2 = tap + 2;
        const int st2 = st == 0 ? 2 : st - 1;     // (st + 2) % 3
        if (!(abl & 1)) {
          if (t2 < 9) issue_w(t2, cc, st2);
          else if (more_c) issue_w(t2 - 9, cc + 1, st2);
        }
      }
      const unsigned char* Wst = Ws + st * WST;
      const int toff = (tap / 3) * HW2 + (tap % 3);
      // keep the 72 (tap, fragment, k-step) halo addresses out of the loop-invariant set: recomputing one costs 3 VALU,
      // hoisting them all costs 72 VGPRs and spills the accumulators
      int hb[TM];
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        hb[b] = h00[b];
        asm volatile("" : "+v"(hb[b]));
      }
      if (!(abl & 4)) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 wf[TN], xf[TM];
        if (!(abl & 8)) {
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[a] = *(const u32x4*)(Wst + wrow + a * 16 * BKB + wkoff[ks]);
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          const int h = hb[b] + toff;
          xf[b] = *(const u32x4*)(Ah + h * BKB + (((ks * 4 + kg) ^ (h & 7)) << 4));
        }
        } else {
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[a] = u32x4{(unsigned)lane, 0u, (unsigned)a, 0u};
#pragma unroll
        for (int b = 0; b < TM; ++b) xf[b] = u32x4{0u, (unsigned)lane, (unsigned)b, 0u};
        }
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
      }
      }
      st = st == 2 ? 0 : st + 1;
    }
  }
  }
  if ((abl & 16) && acc[0][0][0] == 0.f) return;
  if (staged) {
    __syncthreads();   // every wave is past its last fragment read: the halo / weight LDS becomes the staging area
    gemm_epilogue_staged<T, TM, TN>(p, acc, m0 + wm * 64, n0 + wn * WNR, lane, dsm + wave * (32 * (WNR * 4 + 16)));
  } else {
    gemm_epilogue<T, TM, TN, MODE_CONV2D>(p, acc, m0 + wm * 64, n0 + wn * WNR, lane, 0, 0);
  }
}

struct HaloGeom {
  int SR, nseg, halo;
};

static bool halo_geom(const GemmParams& p, HaloGeom& g) {
  const int hw = p.H * p.Wd;
  if (p.Wd <= 0 || 256 % p.Wd != 0) return false;
  if (hw >= 256) {
    if (hw % 256 != 0) return false;
    g.SR = 256 / p.Wd; g.nseg = 1;
  } else {
    if (256 % hw != 0) return false;
    g.SR = p.H; g.nseg = 256 / hw;
  }
  g.halo = g.nseg * (g.SR + 2) * (p.Wd + 2);
  return g.halo <= HALO_MAX_ROWS;
}

bool conv_halo_ok(int dtype, const GemmParams& p) {
  if (tuning().no_halo_conv) return false;
  const int esz = dtype == DT_F32 ? 4 : 2;
  if (p.mode != GATHER_2D || p.stride != 1 || p.pad != 1 || p.ups < 0 || p.ups > 1 || (p.Hin << p.ups) != p.H || (p.Win << p.ups) != p.Wd) return false;
  if (p.batch != 1 || p.splitk > 1 || p.a_act != ACT_NONE || p.epi == EPI_GEGLU || p.epi == EPI_VT) return false;
  if ((p.Cin * esz) % 128 != 0 || p.K != 9 * p.Cin || p.M % 256 != 0) return false;
  // round 6: narrow outputs (the UNet's conv_out, 320 -> 8 channels) on a 256 x 32 tile of the same kernel: the activation halo is staged once per chunk instead
  // of being gathered nine times through L2 by the 256 x 16 tile kernel (0.214 ms at 10x its byte floor); weight rows beyond N are zero-page rows
  const bool narrow = p.N <= 32 && dtype != DT_F32 && !tuning().no_halo_narrow;
  const int bn = narrow ? 32 : (p.N % 160 == 0 ? 160 : 128);
  if (!narrow && p.N % bn != 0) return false;
  if (narrow && (p.R || p.bias2 || p.e_act != ACT_NONE)) return false;
  HaloGeom g;
  if (!halo_geom(p, g)) return false;
  const long tiles = (long)(p.M / 256) * ((p.N + bn - 1) / bn);
  return tuning().force_big_kernels || tiles >= 256;
}

template <typename T, int BN>
static int launch_halo_cfg(const GemmParams& p, const unsigned char* zero_page, hipStream_t s) {
  HaloGeom g;
  if (!halo_geom(p, g)) TANGO_FAIL("conv_halo: unsupported geometry");
  const int abytes = ((g.halo + 7) / 8) * 1024;
  const int lds = 2 * abytes + 3 * BN * 128;
  // Only the product instantiation is compiled: lock-step main loop, no ablation hooks.  (The ping-pong variant measured
  // equal on the UNet convs -- 3.81 vs 3.71 ms on the level-0 convs, profiles/r2_unet_ops_pp_vs_lockstep.txt -- and the
  // ablation hooks are a tools/ build: tools/halo_ablation.sh documents how the round-1 table was taken.)
  auto kfn = conv3x3_halo_kernel<T, BN, false, false>;
  TANGO_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(kfn), lds));
  const int tiles = (p.M / 256) * ((p.N + BN - 1) / BN);
  const int staged = (BN != 32 && epilogue_can_stage<T>(p)) ? 1 : 0;
  const int abl = 0, pp_mode = 0;
  hipLaunchKernelGGL(kfn, dim3((unsigned)tiles), dim3(512), lds, s, p, zero_page, g.SR, g.nseg, abytes, abl, staged, pp_mode);
  TANGO_HIP(hipGetLastError());
  return 0;
}

int launch_conv_halo(int dtype, const GemmParams& p, const unsigned char* zero_page, hipStream_t s) {
  if (!zero_page) TANGO_FAIL("conv_halo: gemm_init() was not called (zero page for the LDS-DMA gather)");
  const bool bn160 = p.N % 160 == 0;
  if (p.N <= 32 && dtype != DT_F32) return dtype == DT_F16 ? launch_halo_cfg<f16, 32>(p, zero_page, s) : launch_halo_cfg<bf16, 32>(p, zero_page, s);
  switch (dtype) {
    case DT_F32: return bn160 ? launch_halo_cfg<float, 160>(p, zero_page, s) : launch_halo_cfg<float, 128>(p, zero_page, s);
    case DT_F16: return bn160 ? launch_halo_cfg<f16, 160>(p, zero_page, s) : launch_halo_cfg<f16, 128>(p, zero_page, s);
    case DT_BF16: return bn160 ? launch_halo_cfg<bf16, 160>(p, zero_page, s) : launch_halo_cfg<bf16, 128>(p, zero_page, s);
  }
  TANGO_FAIL("conv_halo: bad dtype");
}

}  // namespace tango
