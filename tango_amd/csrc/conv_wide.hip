// 3x3 / stride 1 / pad 1 convolution with halo reuse on the 256-pixel x 320-channel tile (round 2): conv_halo.hip's algorithm
// on gemm_wide.hip's tile, for the 16-bit engines.
//
// Why: tools/loop_probe.hip shows the 256 x 160 / 128-byte-chunk main loop is co-limited by LDS traffic (fragment reads +
// DMA writes) rather than by the MFMA pipe; waves of 64 x 160 need 0.35 fragment reads per MFMA instead of 0.45, and a
// workgroup that covers 320 output channels stages each activation halo once for twice the work.  The level-0 convolutions
// (N = 320) become single-column tilings: no second workgroup re-stages the same halo.
//
// Layout: 64-byte channel chunks (one MFMA k-step).  LDS = two halo buffers (chunk cc, cc+1; up to 544 halo pixels x 64 B
// each) + a FOUR-stage ring of weight items (320 rows x 64 B per (chunk, tap) item).  A DMA instruction covers 16 rows x 64 B;
// the 16-byte piece of row h is XOR-swizzled on the source side with ((h >> 1) & 2): for 16 consecutive rows starting
// ANYWHERE (the halo fragments start at an arbitrary pixel, shifted per tap) the four lane groups of a ds_read_b128 then
// touch 16 distinct 16-byte bank slots.  Ping-pong at item granularity as in gemm_wide.hip:
//   [ds_read 14 fragments | wait own DMAs of item i+1] barrier [issue halo piece (taps 0..4) + weight item i+3 | 40 MFMAs] barrier
// with the two 4-wave halves one barrier out of phase.  Epilogue: wide_epilogue (bias, per-step bias, residual, out_scale).
//
// Reference op replaced: the ResnetBlock2D 3x3 convolutions and the up-/down-sampler convolutions of the UNet
// (diffusers/src/diffusers/models/resnet.py:445-552), i.e. ATen convolution.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "tuning.h"
#include "gemm_device.h"
#include "gemm_wide_device.h"

namespace tango {

static constexpr int CW_HALO_MAX = 544;    // halo pixels per tile: 2 x 34 KiB halo + 80 KiB weight ring + 12 KiB source offsets = the whole 160-KiB LDS
                                           // (544 = four 32 x 2 images with their borders: the UNet's level 3)
static constexpr int CW_NA_WIDE = 5;       // halo DMA pieces (16 rows) per wave per channel chunk: 8 waves x 5 x 16 rows >= 544
static constexpr int CW_NA = CW_NA_WIDE;
static constexpr int CW_HALO_MAX_TALL = 800;   // 512-pixel tile: two 64 x 4 images with their borders (level 2) = 792 halo pixels; 2 x 50 KiB halo + 40 KiB weight ring + 16 KiB offsets
static constexpr int CW_NA_TALL = 7;       // 8 waves x 7 x 16 rows >= 800

// SCH: where the LDS-DMAs of an item (one halo piece of the next chunk during taps 0..4, the weight rows of item i+3) are issued
// (round 4, as in gemm_wide.hip):
//   0  at the head of the multiply part, source offsets fetched from LDS there (rounds 2-3: two LDS round trips + the DMA issue
//      in front of the first MFMA)
//   1  inside the MFMA stream, one DMA after every eight MFMAs; the halo offset is fetched in the read part   (convs -10 %,
//      profiles/r4_c1_dma_schedule_ab_b32.txt)
// Measured and dropped (calls 1, 2, 4 of round 4, profiles/r4_c*): DMAs in the read part (after the fragment reads: = 0; at its
// very start, from registers: = 0; all of them there: 9 % slower than 0), paired DMAs after MFMAs 8 and 24 (= 1), a "lean"
// multiply part with the scalar address preparation pinned in the read part (5 % slower than 1), 32x32x16 MFMAs (probe: -5 %).
// PRIO: see gemm_wide.hip.
//
// TALL (round 6): the same eight 64 x 160 wave tiles stacked as 512 pixels x 160 channels.  The halo makes the activation operand
// nearly free (staged once per chunk for nine items), so what an item streams through the DMA engine is its weight rows: 160
// instead of 320 per item for the same 320 MFMAs per wave pair, and a halo of 512 pixels has fewer border pixels per output pixel --
// 14.3 instead of 22.3 KiB of LDS-DMA per item at level 0.  Same MFMA order per accumulator, same epilogue: bit-identical results.
template <typename T, bool RES, bool SK, int SCH, bool TALL>
__global__ __launch_bounds__(512) void conv3x3_wide_kernel(const GemmParams p, const unsigned char* zero_page, const int SR, const int nseg,
                                                           const int abytes, const int prio) {
  constexpr int BM = TALL ? 512 : 256, BN = TALL ? 160 : 320, CB = 64, NST = 4;
  constexpr int CW_NA = TALL ? CW_NA_TALL : CW_NA_WIDE;
  constexpr int BK = CB / (int)sizeof(T);       // 32 channels per chunk
  constexpr int WST = BN * CB;                  // bytes per weight stage
  constexpr int WRG = BN / 16;                  // 16-row DMA groups per weight item: 20
  constexpr int WRGW = (WRG + 7) / 8;           // per wave: 3 (waves 0-3) or 2; TALL: 2 (waves 0-1) or 1
  constexpr int TM = 4, TN = 10;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];   // [halo 0 | halo 1 | W stage 0..3]
  unsigned char* const As = dsm;
  unsigned char* const Ws = dsm + 2 * abytes;

  const int NT = p.N / BN;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / NT) * BM, n0 = (bid % NT) * BN;
  const unsigned char* Ab = (const unsigned char*)p.A;
  const unsigned char* Wb = (const unsigned char*)p.W;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = TALL ? wave : wave & 3, wn = TALL ? 0 : wave >> 2;
  const int lrow = lane >> 2, slot = lane & 3;

  // tile geometry: BM consecutive pixels = SR image rows of one image (nseg == 1) or nseg whole images
  const int H = p.H, Wd = p.Wd, hw = H * Wd;
  const int HW2 = Wd + 2, SEG = (SR + 2) * HW2;
  const int HALO = nseg * SEG, HALO_RG = (HALO + 15) >> 4;
  const int b0 = m0 / hw;
  const int y0 = nseg == 1 ? (m0 - b0 * hw) / Wd : 0;

  // halo DMA sources: piece t of wave w covers halo rows (t*8 + w)*16 + lrow.  The 32-bit byte offsets of the (swizzled)
  // 16-byte pieces live in LDS behind the weight ring (5 dwords per thread; ~0u = outside the image -> zero page): with 160
  // accumulator + 56 fragment registers there is no room to keep them in VGPRs (hipcc spilled them and reloaded them with
  // s_waitcnt vmcnt(0) in the middle of the DMA pipeline).
  unsigned* const aoff_lds = (unsigned*)(Ws + NST * WST) + tid;   // [CW_NA halo pieces | weight group 0] x 512 threads
#pragma unroll
  for (int t = 0; t < CW_NA; ++t) {
    unsigned off = ~0u;
    const int h = ((t * 8 + wave) << 4) + lrow;
    if (h < HALO) {
      const int seg = h / SEG, rem = h - seg * SEG;
      const int hy = rem / HW2, hx = rem - hy * HW2;
      const int y = y0 + hy - 1, x = hx - 1;
      const int pc = slot ^ ((h >> 1) & 2);
      // fused nearest x2 upsampling (p.ups): the halo lives on the upsampled grid, pixel (y, x) reads source (y>>1, x>>1)
      if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)Wd)
        off = (unsigned)(((((int64_t)(b0 + seg) * p.Hin + (y >> p.ups)) * p.Win + (x >> p.ups)) * p.lda) * (int64_t)sizeof(T) + pc * 16);
    }
    aoff_lds[t * 512] = off;
  }
  // weight DMA sources: row group rg = wave + 8 i, row = rg*16 + lrow: 32-bit offset of group 0 from the tile's first weight
  // row, groups 1, 2 are 128 rows further each (same swizzle: (128 >> 1) & 2 == 0)
  const int wrow_d = wave * 16 + lrow;
  const unsigned w_off0_init = (unsigned)(((int64_t)wrow_d * p.Kp) * (int64_t)sizeof(T) + ((slot ^ ((wrow_d >> 1) & 2)) * 16));
  aoff_lds[CW_NA * 512] = w_off0_init;
  const unsigned w_step = (unsigned)(128 * p.Kp * (int64_t)sizeof(T));
  const unsigned char* const Wt = Wb + (int64_t)n0 * p.Kp * (int64_t)sizeof(T);
  const int my_w = wave < WRG - 8 * (WRGW - 1) ? WRGW : WRGW - 1;      // wave-uniform

  auto issue_a_off = [&](const int t, const int cc, const int buf, const unsigned off) {
    const unsigned char* src = off != ~0u ? Ab + (int64_t)cc * CB + off : zero_page;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + buf * abytes + (t * 8 + wave) * 1024), 16, 0, 0);
  };
  auto issue_a = [&](const int t, const int cc, const int buf) {
    if (t * 8 + wave < HALO_RG) issue_a_off(t, cc, buf, aoff_lds[t * 512]);
  };
  // (the item's k offset is wave-uniform and made opaque per call: otherwise hipcc hoists the 27 (tap, group) source addresses
  //  out of the chunk loop as 64-bit VGPR pairs and spills them)
  auto issue_w_one = [&](const int i, int koff, const int st, const unsigned w_off0) {
    const int rg = wave + 8 * i;
    if (rg < WRG) {
      asm volatile("" : "+s"(koff));
      const unsigned char* base = Wt + koff;
      unsigned o = w_off0 + i * w_step;
      asm volatile("" : "+v"(o));               // keeps the zero-extended 64-bit forms of the three offsets out of the loop-invariant set
      __builtin_amdgcn_global_load_lds((gptr_t)(base + o), (lptr_t)(Ws + st * WST + rg * 1024), 16, 0, 0);
    }
  };
  auto issue_w = [&](const int tap, const int cc, const int st) {
    const int koff = (tap * p.Cin + cc * BK) * (int)sizeof(T);
    const unsigned w_off0 = aoff_lds[CW_NA * 512];
#pragma unroll
    for (int i = 0; i < WRGW; ++i) issue_w_one(i, koff, st, w_off0);
  };
  auto wait_n = [&](const int n) {    // at most n of this wave's DMAs stay in flight (wave-uniform)
    switch (n) {
      case 0: wait_vmcnt_lit<0>(); break;
      case 1: wait_vmcnt_lit<1>(); break;
      case 2: wait_vmcnt_lit<2>(); break;
      case 3: wait_vmcnt_lit<3>(); break;
      case 4: wait_vmcnt_lit<4>(); break;
      case 5: wait_vmcnt_lit<5>(); break;
      case 6: wait_vmcnt_lit<6>(); break;
      case 7: wait_vmcnt_lit<7>(); break;
      default: wait_vmcnt_lit<8>(); break;
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: halo row of this lane's pixel in row-block b = h0 + hd[b]; the host guarantees geometries in which
  // 16-pixel blocks are whole image rows or aligned pieces of one (Wd | 16 or 16 | Wd, 16 | H*Wd), so hd[b] is wave-uniform
  // (three SGPRs instead of eight VGPRs: there is no register to spare next to 160 accumulators + 56 fragment registers)
  int h0 = 0, hd[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int pm = wm * 64 + b * 16 + (lane & 15);
    const int seg = pm / hw, r = pm - seg * hw;      // nseg == 1: hw >= BM > pm -> seg = 0
    const int ly = r / Wd, x = r - ly * Wd;
    const int hb_ = seg * SEG + ly * HW2 + x;
    if (b == 0) h0 = hb_;
    hd[b] = __builtin_amdgcn_readfirstlane(hb_ - h0);
  }
  const int kg = lane >> 4;
  const int wrow0 = wn * (TN * 16) + (lane & 15);
  const int wfoff = wrow0 * CB + ((kg ^ ((wrow0 >> 1) & 2)) << 4);     // + a*16*CB: (a*16 >> 1) & 2 == 0

  // split-K: blockIdx.y takes a contiguous range of channel chunks (all nine taps each); partial tiles go to the workspace
  int cc0 = 0, cc1 = p.Cin / BK;
  if (SK) {
    const int per = (cc1 + (int)gridDim.y - 1) / (int)gridDim.y;
    cc0 = (int)blockIdx.y * per;
    cc1 = cc1 < cc0 + per ? cc1 : cc0 + per;
  }
  const int NI = (cc1 - cc0) * 9;                   // (chunk, tap) items of this workgroup

  const int half = wave >> 2;                       // one workgroup per CU: waves w and w + 4 share a SIMD (tools/simd_probe.hip)
  // prologue: halo of chunk 0, weight items 0..2; item 0 (and the halo) must have landed before the first read
#pragma unroll
  for (int t = 0; t < CW_NA; ++t) issue_a(t, cc0, cc0 & 1);
  issue_w(0, cc0, 0);
  issue_w(1, cc0, 1);
  issue_w(2, cc0, 2);
  wait_n(2 * my_w);
  pp_barrier();
  if (half) pp_barrier();                           // the stagger: half B starts one slot late
  if (prio == 2 && half) __builtin_amdgcn_s_setprio(1);
  int st = 0, item = 0;
  int prev_h = 0;                                   // did the previous item issue a halo piece (it is younger than item i+1's weights)
  unsigned hoff = 0u;                               // source offset of the halo piece this item issues
  unsigned w_off0 = w_off0_init;                    // (one register; the rolled tap loop leaves room for it)
  asm volatile("" : "+v"(w_off0));
  for (int cc = cc0; cc < cc1; ++cc) {
    const unsigned char* Ah = As + (cc & 1) * abytes;
    const bool more_c = cc + 1 < cc1;
#pragma unroll 1                                    // one loop body: unrolled, the nine bodies pushed the allocator over 256 VGPRs
    for (int tap = 0; tap < 9; ++tap, ++item) {
      const unsigned char* Wst = Ws + st * WST;
      const int toff = (tap / 3) * HW2 + (tap % 3);
      // what this item issues: halo piece `tap` of chunk cc+1, weight rows of item i+3 into the stage of item i-1
      const bool iss_h = more_c && tap < CW_NA && tap * 8 + wave < HALO_RG;
      const bool iss_w = item + 3 < NI;
      const int st3 = st == 0 ? 3 : st - 1;         // (st + 3) % 4
      int koff3;
      {
        int t3 = tap + 3, c3 = cc;
        if (t3 >= 9) { t3 -= 9; c3 = cc + 1; }
        koff3 = (t3 * p.Cin + c3 * BK) * (int)sizeof(T);
      }
      // ---- read part ----
      if (SCH != 0) { if (iss_h) hoff = aoff_lds[tap * 512]; }
      u32x4 wf[TN], xf[TM];
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[a] = *(const u32x4*)(Wst + wfoff + a * 16 * CB);
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        int hs = hd[b] + toff;                       // wave-uniform part, opaque per tap: 36 hoisted lane addresses would not fit
        asm volatile("" : "+s"(hs));
        const int h = h0 + hs;
        xf[b] = *(const u32x4*)(Ah + h * CB + ((kg ^ ((h >> 1) & 2)) << 4));
      }
      // item i+1 must have landed before the barrier that precedes anybody's read of it.  Younger than its last DMA, in issue
      // order: [halo piece of item i-1] [weights of item i+2]
      if (item + 1 < NI) wait_n(prev_h + (item + 2 < NI ? my_w : 0));
      prev_h = iss_h ? 1 : 0;
      pp_barrier();
      // ---- multiply part ----
      if (SCH == 0) {
        if (more_c && tap < CW_NA) issue_a(tap, cc + 1, (cc + 1) & 1);
        const int t3 = tap + 3;
        if (t3 < 9) issue_w(t3, cc, st3);
        else if (more_c) issue_w(t3 - 9, cc + 1, st3);
      }
      if (prio == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < TN; ++a) {
#pragma unroll
        for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
        if (SCH != 0 && (a & 1) == 0 && (a >> 1) <= WRGW) {
          // after MFMAs 4, 12, 20, 28: [halo piece, weight groups 0, 1, 2]
          const int sl = a >> 1;
          __builtin_amdgcn_sched_barrier(0);
          if (sl == 0) { if (iss_h) issue_a_off(tap, cc + 1, (cc + 1) & 1, hoff); }
          else if (iss_w) issue_w_one(sl - 1, koff3, st3, w_off0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (prio == 0) __builtin_amdgcn_s_setprio(0);
      pp_barrier();
      st = st == NST - 1 ? 0 : st + 1;
    }
  }
  if (!half) pp_barrier();
  if (prio == 2) __builtin_amdgcn_s_setprio(0);
  __syncthreads();   // every wave is past its last fragment read: the halo / weight LDS becomes the staging area
  unsigned char* const slice = dsm + wave * (WIDE_STAGE_BYTES + 1280);
  if (SK) {
    wide_epilogue_raw(p, acc, (int)blockIdx.y, m0 + wm * 64, n0 + wn * (TN * 16), lane, slice);
    return;
  }
  const float mean[TM] = {0.f, 0.f, 0.f, 0.f}, rstd[TM] = {1.f, 1.f, 1.f, 1.f};
  wide_epilogue<T, false, RES, false>(p, acc, mean, rstd, m0 + wm * 64, n0 + wn * (TN * 16), lane, slice);
}

// ------------------------------------------------------------------------------------------------------------------------
// In-wave software pipeline (round 6; gemm_wide.hip: gemm_wide_pipe_kernel has the schedule and its proof).  Same LDS image (two
// halo buffers, four weight stages), same DMA source layout.  During item i a wave multiplies the fragments it holds and refills them
// in place for item i+1 (wf[a] right behind column group a, xf[b] behind its last use in group 9; fragment reads from inline asm,
// one lgkmcnt(0) per item); the weights of item i+4 go into the stage item i was read from (by everybody, during item i-1), the halo
// piece of the next channel chunk rides in items (cc, 0..4) as before.  Two half-items of 20 MFMAs, a raw barrier behind each, the
// 4-wave halves one barrier apart.  vmcnt: per wave the queue is ... hp(i-2) W(i+2) hp(i-1) W(i+3) at the end of H0(i), where W(i+2)
// (and with it every older halo piece) must have landed one barrier before anybody reads it: at most hp(i-1) + |W(i+3)| stay in flight.
// Same MFMA order per accumulator as conv3x3_wide_kernel: bit-identical results.
// ------------------------------------------------------------------------------------------------------------------------
template <int OFF> __device__ __forceinline__ void cwp_lds_read(u32x4& v, const unsigned base) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF));
}
__device__ __forceinline__ void cwp_lds_wait_all(u32x4 (&wf)[10], u32x4 (&xf)[4], unsigned& hoff) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4]), "+v"(wf[5]), "+v"(wf[6]), "+v"(wf[7]), "+v"(wf[8]), "+v"(wf[9]),
                 "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(hoff));
}
// the halo piece's source offset for the NEXT item, read with the fragments (a compiler-visible LDS load here makes hipcc wait
// lgkmcnt(0) in front of every DMA of the item: the load's destination register is reused for their offsets)
__device__ __forceinline__ void cwp_lds_read_b32(unsigned& v, const unsigned addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr)); }
template <int I, int N, typename F> __device__ __forceinline__ void cwp_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); cwp_for<I + 1, N>(f); }
}

// LOCK: all eight waves in step, one barrier per item (gemm_wide.hip).  The halo piece of chunk cc + 1 goes into the OTHER halo buffer, whose
// last readers were the refills for item (cc, 0) during item (cc - 1, 8): done at the barrier that ended it.
// MODE 2, SKEW (TANGO_WIDE_PIPE=3): ONE barrier per item AND staggered halves -- every wave meets the others once per item, but the 4-wave
// halves do so at different points of their item (half A behind column group 1, half B behind group 6), so waves w and w + 4, which
// share a SIMD, run half an item apart: the end-of-item tail of one (last refills + lgkmcnt(0)) lies under the other's MFMAs, and nobody
// pays a second barrier.  Order per wave and item: [groups 0..P] wait own DMAs of item i + 2, barrier #i, [this item's DMAs one per group
// behind the barrier, groups P+1..9].  Barrier #i is passed only when every wave has finished item i - 1 completely (its end-of-item
// lgkmcnt(0) covers the refill reads, i.e. all reads of item i's operands), so the DMAs of item i -- weights of item i + 4 into the stage
// item i was read from, halo piece of the next chunk into the other halo buffer -- overwrite nothing that is still read; and a wave's
// first read of item i + 3's operands (group 0 of item i + 2) lies behind barrier #(i + 1), in front of which every wave has waited for
// its own pieces of them.
template <typename T, bool RES, bool SK, int MODE = 0>
__global__ __launch_bounds__(512) void conv3x3_wide_pipe_kernel(const GemmParams p, const unsigned char* zero_page, const int SR, const int nseg,
                                                                const int abytes, const int prio) {
  constexpr bool LOCK = MODE == 1, SKEW = MODE == 2;
  constexpr int BM = 256, BN = 320, CB = 64, NST = 4;
  constexpr int BK = CB / (int)sizeof(T);
  constexpr int WST = BN * CB;
  constexpr int WRG = BN / 16;
  constexpr int WRGW = (WRG + 7) / 8;
  constexpr int TM = 4, TN = 10;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];   // [halo 0 | halo 1 | W stage 0..3 | source offsets]
  unsigned char* const As = dsm;
  unsigned char* const Ws = dsm + 2 * abytes;

  const int NT = p.N / BN;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / NT) * BM, n0 = (bid % NT) * BN;
  const unsigned char* Ab = (const unsigned char*)p.A;
  const unsigned char* Wb = (const unsigned char*)p.W;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int lrow = lane >> 2, slot = lane & 3;

  const int H = p.H, Wd = p.Wd, hw = H * Wd;
  const int HW2 = Wd + 2, SEG = (SR + 2) * HW2;
  const int HALO = nseg * SEG, HALO_RG = (HALO + 15) >> 4;
  const int b0 = m0 / hw;
  const int y0 = nseg == 1 ? (m0 - b0 * hw) / Wd : 0;

  unsigned* const aoff_lds = (unsigned*)(Ws + NST * WST) + tid;   // [CW_NA halo pieces | weight group 0] x 512 threads (conv3x3_wide_kernel)
#pragma unroll
  for (int t = 0; t < CW_NA; ++t) {
    unsigned off = ~0u;
    const int h = ((t * 8 + wave) << 4) + lrow;
    if (h < HALO) {
      const int seg = h / SEG, rem = h - seg * SEG;
      const int hy = rem / HW2, hx = rem - hy * HW2;
      const int y = y0 + hy - 1, x = hx - 1;
      const int pc = slot ^ ((h >> 1) & 2);
      if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)Wd)
        off = (unsigned)(((((int64_t)(b0 + seg) * p.Hin + (y >> p.ups)) * p.Win + (x >> p.ups)) * p.lda) * (int64_t)sizeof(T) + pc * 16);
    }
    aoff_lds[t * 512] = off;
  }
  const int wrow_d = wave * 16 + lrow;
  const unsigned w_off0_init = (unsigned)(((int64_t)wrow_d * p.Kp) * (int64_t)sizeof(T) + ((slot ^ ((wrow_d >> 1) & 2)) * 16));
  const unsigned w_step = (unsigned)(128 * p.Kp * (int64_t)sizeof(T));
  const unsigned char* const Wt = Wb + (int64_t)n0 * p.Kp * (int64_t)sizeof(T);
  const int my_w = wave < WRG - 8 * (WRGW - 1) ? WRGW : WRGW - 1;      // wave-uniform: 3 (waves 0-3) or 2

  auto issue_a_off = [&](const int t, const int cc, const int buf, const unsigned off) {
    const unsigned char* src = off != ~0u ? Ab + (int64_t)cc * CB + off : zero_page;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + buf * abytes + (t * 8 + wave) * 1024), 16, 0, 0);
  };
  auto issue_w_one = [&](const int i, int koff, const int st, const unsigned w_off0) {
    const int rg = wave + 8 * i;
    if (rg < WRG) {
      asm volatile("" : "+s"(koff));
      const unsigned char* base = Wt + koff;
      unsigned o = w_off0 + i * w_step;
      asm volatile("" : "+v"(o));
      __builtin_amdgcn_global_load_lds((gptr_t)(base + o), (lptr_t)(Ws + st * WST + rg * 1024), 16, 0, 0);
    }
  };
  auto wait_n = [&](const int n) {
    switch (n) {
      case 0: wait_vmcnt_lit<0>(); break;
      case 1: wait_vmcnt_lit<1>(); break;
      case 2: wait_vmcnt_lit<2>(); break;
      case 3: wait_vmcnt_lit<3>(); break;
      case 4: wait_vmcnt_lit<4>(); break;
      case 5: wait_vmcnt_lit<5>(); break;
      default: wait_vmcnt_lit<6>(); break;
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int h0 = 0, hd[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int pm = wm * 64 + b * 16 + (lane & 15);
    const int seg = pm / hw, r = pm - seg * hw;
    const int ly = r / Wd, x = r - ly * Wd;
    const int hb_ = seg * SEG + ly * HW2 + x;
    if (b == 0) h0 = hb_;
    hd[b] = __builtin_amdgcn_readfirstlane(hb_ - h0);
  }
  const int kg = lane >> 4;
  const int wrow0 = wn * (TN * 16) + (lane & 15);
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
  const unsigned wbase = lds0 + (unsigned)(2 * abytes) + (unsigned)(wrow0 * CB + ((kg ^ ((wrow0 >> 1) & 2)) << 4));

  int cc0 = 0, cc1 = p.Cin / BK;
  if (SK) {
    const int per = (cc1 + (int)gridDim.y - 1) / (int)gridDim.y;
    cc0 = (int)blockIdx.y * per;
    cc1 = cc1 < cc0 + per ? cc1 : cc0 + per;
  }
  const int NI = (cc1 - cc0) * 9;                   // >= 9, or 0: the last split of an uneven split-K division owns no chunk
  if (SK && NI <= 0) {
    // (the ping-pong kernel's loops simply do not run; this one peels its last item, so the empty split leaves here: a zero partial tile.
    //  Found by the Music UNet at B2 = 2: Cin = 960 -> 30 chunks over 7 splits of 5)
    __syncthreads();                                // the source-offset table above shares LDS with the epilogue's staging rows
    wide_epilogue_raw(p, acc, (int)blockIdx.y, m0 + wm * 64, n0 + wn * (TN * 16), lane, dsm + wave * (WIDE_STAGE_BYTES + 1280));
    return;
  }

  const int half = wave >> 2;
  // LDS address of this lane's activation fragment b of (halo buffer `buf`, tap offset `toff`)
  auto xaddr = [&](const int b, const int buf, const int toff) -> unsigned {
    int hs = hd[b] + toff;
    asm volatile("" : "+s"(hs));
    const int h = h0 + hs;
    return lds0 + (unsigned)(buf * abytes + h * CB + ((kg ^ ((h >> 1) & 2)) << 4));
  };
  // prologue: halo of chunk cc0, weight items 0..3; halo and items 0, 1 must have landed before the first reads
  {
#pragma unroll
    for (int t = 0; t < CW_NA; ++t)
      if (t * 8 + wave < HALO_RG) issue_a_off(t, cc0, cc0 & 1, aoff_lds[t * 512]);
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const int koff = (j * p.Cin + cc0 * BK) * (int)sizeof(T);
#pragma unroll
      for (int i = 0; i < WRGW; ++i) issue_w_one(i, koff, j, w_off0_init);
    }
  }
  wait_n(2 * my_w);
  pp_barrier();
  u32x4 wf[TN], xf[TM];
  cwp_for<0, TN>([&](auto a_tag) { constexpr int a = decltype(a_tag)::value; cwp_lds_read<a * 16 * CB>(wf[a], wbase); });
  cwp_for<0, TM>([&](auto b_tag) { constexpr int b = decltype(b_tag)::value; cwp_lds_read<0>(xf[b], xaddr(b, cc0 & 1, 0)); });
  const unsigned aoff_addr = (unsigned)(uintptr_t)(lptr_t)aoff_lds;
  unsigned hoff0;
  cwp_lds_read_b32(hoff0, aoff_addr);               // item 0 issues halo piece 0 of chunk cc0 + 1
  cwp_lds_wait_all(wf, xf, hoff0);
  __builtin_amdgcn_sched_barrier(0);
  if (MODE == 0 && half) pp_barrier();              // the stagger (SKEW: it establishes itself at the first barrier)
  if (prio == 2 && half) __builtin_amdgcn_s_setprio(1);
  unsigned w_off0 = w_off0_init;
  asm volatile("" : "+v"(w_off0));
  // one item; MORE1: item + 1 exists (refill the fragments) -- compile-time, the last item is peeled.  prev_h: did the previous item
  // issue a halo piece; returns the same for this item.  (tap, cc, prev_h by value: captured by reference and modified here they
  // went to scratch, with a vmcnt(0) reload at the head of every item)
  auto do_item = [&](auto more1_tag, const int item, const int tap, const int cc, const int prev_h, unsigned& hoff) __attribute__((always_inline)) -> int {
    constexpr bool MORE1 = decltype(more1_tag)::value;
    const bool more_c = cc + 1 < cc1;
    // item + 1: tap / halo buffer
    const int tap1 = tap == 8 ? 0 : tap + 1;
    const int buf1 = tap == 8 ? ((cc + 1) & 1) : (cc & 1);
    const int toff1 = (tap1 / 3) * HW2 + (tap1 % 3);
    unsigned wsrc = wbase + (unsigned)((item + 1) & (NST - 1)) * WST;
    asm volatile("" : "+v"(wsrc));
    // what this item issues (in H1): halo piece `tap` of chunk cc+1, weight rows of item + 4 into the stage of this item
    const bool iss_h = more_c && tap < CW_NA && tap * 8 + wave < HALO_RG;
    const bool iss_w = item + NST < NI;
    const int st4 = item & (NST - 1);
    int koff4;
    {
      int t4 = tap + NST, c4 = cc;
      if (t4 >= 9) { t4 -= 9; c4 = cc + 1; }
      koff4 = (t4 * p.Cin + c4 * BK) * (int)sizeof(T);
    }
    // ---- H0: column groups 0-4 ----
    cwp_for<0, 5>([&](auto a_tag) {
      constexpr int a = decltype(a_tag)::value;
#pragma unroll
      for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
      __builtin_amdgcn_sched_barrier(0);
      if (MORE1) cwp_lds_read<a * 16 * CB>(wf[a], wsrc);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (item + 2 < NI) wait_n(prev_h + (item + 3 < NI ? my_w : 0));
    if (!LOCK) pp_barrier();
    // ---- H1: column groups 5-9 and this item's DMAs ----
    unsigned xa[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) xa[b] = MORE1 ? xaddr(b, buf1, toff1) : 0u;
    cwp_for<5, 9>([&](auto a_tag) {
      constexpr int a = decltype(a_tag)::value;
#pragma unroll
      for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
      __builtin_amdgcn_sched_barrier(0);
      if (MORE1) cwp_lds_read<a * 16 * CB>(wf[a], wsrc);
      if (a == 5) { if (iss_h) issue_a_off(tap, cc + 1, (cc + 1) & 1, hoff); }
      else if (iss_w) issue_w_one(a - 6, koff4, st4, w_off0);
      __builtin_amdgcn_sched_barrier(0);
    });
    cwp_for<0, TM>([&](auto b_tag) {
      constexpr int b = decltype(b_tag)::value;
      Mma<T>::run(acc[9][b], wf[9], xf[b]);
      __builtin_amdgcn_sched_barrier(0);
      if (MORE1) cwp_lds_read<0>(xf[b], xa[b]);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (MORE1) {
      cwp_lds_read<9 * 16 * CB>(wf[9], wsrc);
      cwp_lds_read_b32(hoff, aoff_addr + (unsigned)(tap1 < CW_NA ? tap1 : 0) * 2048u);      // the next item's halo piece offset
      cwp_lds_wait_all(wf, xf, hoff);
    }
    pp_barrier();
    return iss_h ? 1 : 0;
  };
  // SKEW item: half A's barrier sits behind column group 1, half B's behind group 6.  ONE body with wave-uniform run-time tests on
  // `half` (two specialised loop bodies behind an if / else made the allocator spill 100+ accumulator registers at the join)
  auto do_item_skew = [&](auto more1_tag, const int item, const int tap, const int cc, const int prev_h, unsigned& hoff) __attribute__((always_inline)) -> int {
    constexpr bool MORE1 = decltype(more1_tag)::value;
    constexpr int PA = 1, PB = 6;
    const bool more_c = cc + 1 < cc1;
    const int tap1 = tap == 8 ? 0 : tap + 1;
    const int buf1 = tap == 8 ? ((cc + 1) & 1) : (cc & 1);
    const int toff1 = (tap1 / 3) * HW2 + (tap1 % 3);
    unsigned wsrc = wbase + (unsigned)((item + 1) & (NST - 1)) * WST;
    asm volatile("" : "+v"(wsrc));
    const bool iss_h = more_c && tap < CW_NA && tap * 8 + wave < HALO_RG;
    const bool iss_w = item + NST < NI;
    const int st4 = item & (NST - 1);
    int koff4;
    {
      int t4 = tap + NST, c4 = cc;
      if (t4 >= 9) { t4 -= 9; c4 = cc + 1; }
      koff4 = (t4 * p.Cin + c4 * BK) * (int)sizeof(T);
    }
    // DMA slot sl = 0: the halo piece, 1..WRGW: weight row groups 0..WRGW-1 -- one per column group behind the wave's barrier
    auto dma_slot = [&](auto sl_tag) {
      constexpr int sl = decltype(sl_tag)::value;
      if constexpr (sl == 0) { if (iss_h) issue_a_off(tap, cc + 1, (cc + 1) & 1, hoff); }
      else if constexpr (sl <= WRGW) { if (iss_w) issue_w_one(sl - 1, koff4, st4, w_off0); }
    };
    auto sync_point = [&]() {
      if (item + 2 < NI) wait_n(prev_h + (item + 3 < NI ? my_w : 0));
      pp_barrier();
    };
    cwp_for<0, 9>([&](auto a_tag) {
      constexpr int a = decltype(a_tag)::value;
#pragma unroll
      for (int b = 0; b < TM; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
      __builtin_amdgcn_sched_barrier(0);
      if (MORE1) cwp_lds_read<a * 16 * CB>(wf[a], wsrc);
      if constexpr (a == PA) { if (!half) sync_point(); }
      if constexpr (a == PB) { if (half) sync_point(); }
      if constexpr (a > PA && a - PA - 1 <= WRGW) { if (!half) dma_slot(std::integral_constant<int, a - PA - 1>{}); }
      if constexpr (a > PB) { if (half) dma_slot(std::integral_constant<int, a - PB - 1>{}); }
      __builtin_amdgcn_sched_barrier(0);
    });
    unsigned xa[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) xa[b] = MORE1 ? xaddr(b, buf1, toff1) : 0u;
    cwp_for<0, TM>([&](auto b_tag) {
      constexpr int b = decltype(b_tag)::value;
      Mma<T>::run(acc[9][b], wf[9], xf[b]);
      __builtin_amdgcn_sched_barrier(0);
      if (MORE1) cwp_lds_read<0>(xf[b], xa[b]);
      if constexpr (2 + b <= WRGW) { if (half) dma_slot(std::integral_constant<int, 2 + b>{}); }     // half B: groups 7, 8 took slots 0, 1
      __builtin_amdgcn_sched_barrier(0);
    });
    if (MORE1) {
      cwp_lds_read<9 * 16 * CB>(wf[9], wsrc);
      cwp_lds_read_b32(hoff, aoff_addr + (unsigned)(tap1 < CW_NA ? tap1 : 0) * 2048u);
      cwp_lds_wait_all(wf, xf, hoff);
    }
    return iss_h ? 1 : 0;
  };
  auto run_items = [&](auto item_fn) __attribute__((always_inline)) {
    int tap = 0, cc = cc0, prev_h = 0;
    unsigned hoff = hoff0;
#pragma unroll 1
    for (int item = 0; item + 1 < NI; ++item) {
      prev_h = item_fn(std::true_type{}, item, tap, cc, prev_h, hoff);
      if (tap == 8) { tap = 0; ++cc; } else ++tap;
    }
    item_fn(std::false_type{}, NI - 1, 8, cc1 - 1, prev_h, hoff);
  };
  if constexpr (SKEW) run_items(do_item_skew);
  else run_items(do_item);
  if (MODE == 0 && !half) pp_barrier();
  if (prio == 2) __builtin_amdgcn_s_setprio(0);
  __syncthreads();   // every wave is past its last fragment read: the halo / weight LDS becomes the staging area
  unsigned char* const slice = dsm + wave * (WIDE_STAGE_BYTES + 1280);
  if (SK) {
    wide_epilogue_raw(p, acc, (int)blockIdx.y, m0 + wm * 64, n0 + wn * (TN * 16), lane, slice);
    return;
  }
  const float mean[TM] = {0.f, 0.f, 0.f, 0.f}, rstd[TM] = {1.f, 1.f, 1.f, 1.f};
  wide_epilogue<T, false, RES, false>(p, acc, mean, rstd, m0 + wm * 64, n0 + wn * (TN * 16), lane, slice);
}

struct WideHaloGeom {
  int SR, nseg, halo;
};

static bool wide_halo_geom(const GemmParams& p, WideHaloGeom& g, const int BM = 256) {
  const int hw = p.H * p.Wd;
  if (p.Wd <= 0 || BM % p.Wd != 0) return false;
  if (hw >= BM) {
    if (hw % BM != 0) return false;
    g.SR = BM / p.Wd; g.nseg = 1;
  } else {
    if (BM % hw != 0) return false;
    g.SR = p.H; g.nseg = BM / hw;
  }
  g.halo = g.nseg * (g.SR + 2) * (p.Wd + 2);
  if (!((16 % p.Wd == 0 || p.Wd % 16 == 0) && hw % 16 == 0)) return false;   // wave-uniform row-block offsets (see the kernel)
  return g.halo <= (BM == 512 ? CW_HALO_MAX_TALL : CW_HALO_MAX);
}

// the 512 x 160 form of the tile (TANGO_CONV_TALL): unsplit problems whose geometry fits its halo (levels 0-2 of the UNet)
static bool conv_wide_tall_ok(const GemmParams& p) {
  if (!tuning().conv_tall || p.splitk > 1 || p.M % 512 != 0) return false;
  WideHaloGeom g;
  return wide_halo_geom(p, g, 512);
}

bool conv_wide_ok(int dtype, const GemmParams& p) {
  if (tuning().no_wide_conv || dtype == DT_F32) return false;
  if (p.mode != GATHER_2D || p.stride != 1 || p.pad != 1 || p.ups < 0 || p.ups > 1 || (p.Hin << p.ups) != p.H || (p.Win << p.ups) != p.Wd) return false;
  if (p.batch != 1 || p.a_act != ACT_NONE || p.epi != EPI_NONE || p.bias_rows) return false;
  if (p.splitk > 1 ? (!p.ws || p.splitk > p.Cin / 32 || p.N % 4 != 0) : (p.e_act != ACT_NONE || p.out_f32)) return false;
  if ((p.Cin * 2) % 64 != 0 || p.K != 9 * p.Cin || p.M % 256 != 0 || p.N % 320 != 0) return false;
  if (p.ldo % 8 != 0 || ((uintptr_t)p.out & 15) || (p.R && (p.ldr % 8 != 0 || ((uintptr_t)p.R & 15)))) return false;
  if ((p.lda * 2) % 16 != 0 || (p.Kp * 2) % 16 != 0 || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return false;
  if (((uintptr_t)p.bias & 15) || ((uintptr_t)p.bias2 & 15) || (p.bias2 && p.bias2_stride % 4 != 0)) return false;
  // 32-bit source offsets in the kernel: activation tensor and one tile's 320 weight rows below 4 GiB
  if ((int64_t)(p.M >> (2 * p.ups)) * p.lda * 2 >= (int64_t)0xFFFF0000 || (int64_t)320 * p.Kp * 2 >= (int64_t)0xFFFF0000) return false;
  WideHaloGeom g;
  if (!wide_halo_geom(p, g)) return false;
  const long tiles = (long)(p.M / 256) * (p.N / 320) * (p.splitk > 1 ? p.splitk : 1);
  return tuning().force_big_kernels || tiles >= 224;
}

// split-K factor the wide conv wants for a problem whose 256 x 320 tiling does not fill the chip (0: not a wide-conv problem)
int conv_wide_pick_splitk(int dtype, const GemmParams& p) {
  GemmParams q = p;
  q.splitk = 1;
  if (tuning().force_big_kernels || p.mode != GATHER_2D || dtype == DT_F32 || p.M % 256 != 0 || p.N % 320 != 0 || p.Cin % 32 != 0) return 0;
  const long tiles = (long)(p.M / 256) * (p.N / 320);
  // measured (B = 8 / 16 UNet steps, profiles/r2_unet_ops_small_batch_splitk.txt): 64 tiles x 4 splits beat the 4-wave tiles'
  // split-K by 21-27 %; 128 tiles x 2 splits LOSE to the unsplit 256 x 160 halo kernel by 14-20 %
  if (tiles > 96 || tiles < 16) return 0;
  int s = (int)((256 + tiles / 2) / tiles);          // one workgroup per CU
  const int nc = p.Cin / 32;
  if (s > nc / 4) s = nc / 4;                        // >= 4 channel chunks (36 items) per split
  if (s < 2) return 0;
  float dummy_ws = 0.f;
  q.splitk = s; q.ws = &dummy_ws;
  return conv_wide_ok(dtype, q) ? s : 0;
}

template <typename T, bool RES, bool SK, int SCH, bool TALL = false>
static int launch_conv_wide_sch(const GemmParams& p, const unsigned char* zero_page, hipStream_t s) {
  constexpr int BM = TALL ? 512 : 256, BN = TALL ? 160 : 320, NA = TALL ? CW_NA_TALL : CW_NA_WIDE;
  WideHaloGeom g;
  if (!wide_halo_geom(p, g, BM)) TANGO_FAIL("conv_wide: unsupported geometry");
  const int abytes = ((g.halo + 15) / 16) * 1024;
  int lds = 2 * abytes + 4 * BN * 64 + (NA + 1) * 512 * 4;
  const int epi_lds = 8 * (WIDE_STAGE_BYTES + 1280);
  if (lds < epi_lds) lds = epi_lds;
  auto kfn = conv3x3_wide_kernel<T, RES, SK, SCH, TALL>;
  TANGO_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(kfn), lds));
  const int tiles = (p.M / BM) * (p.N / BN);
  hipLaunchKernelGGL(kfn, dim3((unsigned)tiles, (unsigned)(p.splitk > 1 ? p.splitk : 1)), dim3(512), lds, s, p, zero_page, g.SR, g.nseg, abytes, tuning().wide_prio);
  TANGO_HIP(hipGetLastError());
  if (p.splitk > 1) TANGO_TRY(launch_splitk_reduce(TypeTag<T>::dt, p, s));
  return 0;
}

template <typename T, bool RES, bool SK>
static int launch_conv_wide_pipe(const GemmParams& p, const unsigned char* zero_page, hipStream_t s) {
  WideHaloGeom g;
  if (!wide_halo_geom(p, g)) TANGO_FAIL("conv_wide: unsupported geometry");
  const int abytes = ((g.halo + 15) / 16) * 1024;
  int lds = 2 * abytes + 4 * 320 * 64 + (CW_NA + 1) * 512 * 4;
  const int epi_lds = 8 * (WIDE_STAGE_BYTES + 1280);
  if (lds < epi_lds) lds = epi_lds;
  const int mode = tuning().conv_pipe;
  auto kfn = mode == 3 ? conv3x3_wide_pipe_kernel<T, RES, SK, 2> : mode == 2 ? conv3x3_wide_pipe_kernel<T, RES, SK, 1> : conv3x3_wide_pipe_kernel<T, RES, SK, 0>;
  TANGO_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(kfn), lds));
  const int tiles = (p.M / 256) * (p.N / 320);
  hipLaunchKernelGGL(kfn, dim3((unsigned)tiles, (unsigned)(p.splitk > 1 ? p.splitk : 1)), dim3(512), lds, s, p, zero_page, g.SR, g.nseg, abytes, tuning().wide_prio);
  TANGO_HIP(hipGetLastError());
  if (p.splitk > 1) TANGO_TRY(launch_splitk_reduce(TypeTag<T>::dt, p, s));
  return 0;
}

template <typename T, bool RES, bool SK = false>
static int launch_conv_wide_cfg(const GemmParams& p, const unsigned char* zero_page, hipStream_t s) {
  if (tuning().conv_pipe && !(SK ? false : conv_wide_tall_ok(p))) return launch_conv_wide_pipe<T, RES, SK>(p, zero_page, s);   // (TANGO_CONV_TALL=1 selects the ping-pong kernel's 512 x 160 form)
  if constexpr (!SK) {
    if (conv_wide_tall_ok(p)) return launch_conv_wide_sch<T, RES, false, 1, true>(p, zero_page, s);
  }
  switch (tuning().wide_sched) {
    case 0: return launch_conv_wide_sch<T, RES, SK, 0>(p, zero_page, s);
    default: return launch_conv_wide_sch<T, RES, SK, 1>(p, zero_page, s);
  }
}

int launch_conv_wide(int dtype, const GemmParams& p, const unsigned char* zero_page, hipStream_t s) {
  if (!zero_page) TANGO_FAIL("conv_wide: gemm_init() was not called (zero page for the LDS-DMA gather)");
  switch (dtype) {
    case DT_F16:
      if (p.splitk > 1) return launch_conv_wide_cfg<f16, false, true>(p, zero_page, s);
      return p.R ? launch_conv_wide_cfg<f16, true>(p, zero_page, s) : launch_conv_wide_cfg<f16, false>(p, zero_page, s);
    case DT_BF16:
      if (p.splitk > 1) return launch_conv_wide_cfg<bf16, false, true>(p, zero_page, s);
      return p.R ? launch_conv_wide_cfg<bf16, true>(p, zero_page, s) : launch_conv_wide_cfg<bf16, false>(p, zero_page, s);
  }
  TANGO_FAIL("conv_wide: 16-bit dtypes only");
}

}  // namespace tango
