// Flash attention (head_dim 64) for the 32 UNet attention sites (16 self + 16 cross).
// Reference: Attention + AttnProcessor / AttnProcessor2_0
// (mustango/diffusers/src/diffusers/models/attention_processor.py:302-337,495-540): softmax(q k^T * d^-0.5 + bias) v
// with the additive -10000 mask bias broadcast over heads and queries (prepare_attention_mask :263-299).
//
// "Swapped" formulation so nothing crosses lanes between the two matmuls:
//   S^T = K Q^T  (MFMA A = K rows from LDS, B = Q fragments held in registers)
//   O^T = V^T P^T (MFMA A = V^T rows from LDS, B = P^T straight from the S^T accumulators)
// In the 16x16 accumulator layout every lane owns ONE query column (q = lane & 15) and 4 kv rows per
// 16-kv block, so the online-softmax state (m, l, rescale) is a per-lane scalar and row reductions are
// in-register + two cross-group shuffles.  Softmax statistics and accumulation are fp32 always.
//
// V arrives already transposed ([B][heads][64][ldvt], written by the QKV / KV projection GEMM epilogue),
// so both K and V^T tiles are plain 128-byte-row copies into XOR-swizzled LDS (conflict-free
// ds_read_b128), two stages, ONE barrier per 64-key tile.  For 16-bit types the V^T tile columns are
// stored in the order the P^T accumulator fragments present them ([4g..4g+3 | 16+4g..16+4g+3] adjacent),
// so each PV fragment is a single ds_read_b128.
#include <cstdint>
#include <cstdlib>

#include "common.h"
#include "tuning.h"

namespace tango {

template <typename T> struct AMma;
template <> struct AMma<float> {
  __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
  }
};
template <> struct AMma<f16> {
  __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  }
};
template <> struct AMma<bf16> {
  __device__ static __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};

// max over the four lanes {l, l^16, l^32, l^48} that hold one query's kv slices, WITHOUT the LDS crossbar:
// v_permlane32_swap / v_permlane16_swap exchange 32- / 16-lane halves between two registers (2 VALU + 1 max per step,
// no ds_bpermute round trip in the softmax dependency chain).  The max is taken in inline asm: (a) hipcc (ROCm 7.2)
// folds fmaxf(r[0], r[1]) of the swap builtin's two results to r[0] -- it treats them as equal --, (b) it would put a
// canonicalising v_max x, x in front of each operand.
__device__ __forceinline__ float asm_max(float a, float b) {
  float m;
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
  return m;
}
__device__ __forceinline__ float quad_max(float v) {
  const unsigned a = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  const float m = asm_max(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
  const unsigned c = __builtin_bit_cast(unsigned, m);
  const auto q = __builtin_amdgcn_permlane16_swap(c, c, false, false);
  return asm_max(__builtin_bit_cast(float, (unsigned)q[0]), __builtin_bit_cast(float, (unsigned)q[1]));
}

// four floats -> four OCP e4m3 bytes in one dword (v_cvt_pk_fp8_f32 x2, round to nearest even)
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}

// NW waves per workgroup share one K / V^T tile stream: the L2 -> LDS fill per workgroup is fixed (the whole K and V of
// the (batch, head)), so the fill bytes per MFMA flop scale with 1 / (NW * QB).
// PB: additive per-(head, query, key) bias shared by the batch (T5 relative position bias); implies MASKED.
// F8 (round 3; BASELINE config 5 "bf16 + fp8 MFMA attention"): the P.V product runs on v_mfma_f32_16x16x32_fp8_fp8 -- P and V as
// OCP e4m3, fp32 accumulation and fp32 softmax statistics; Q.K^T stays in T (three mantissa bits on the logits would move the
// softmax weights by 5-20 %).  P is carried as 2^8 P (so the e4m3 subnormal floor sits at 7.6e-6 instead of 2e-3 of the row
// maximum -- with 4096 keys most weights are far below 2e-3); the factor cancels in O = (P V) / sum(P) because the row sum is
// taken of the same scaled values.  V^T is converted while its tile is staged (clamped to the e4m3 range, +-448).
// MSUM (round 4): the softmax row sums come from the matrix pipe -- one more MFMA per P^T fragment against a constant fragment whose
// row 0 is all ones (no LDS: it is a register constant), instead of 32 v_add_f32 per tile and wave in a loop that is VALU-bound
// (profiles/r4_attention_loop_isa_count.txt: 182 VALU-class instructions per 32 MFMAs).  The sum is then taken of the ROUNDED P the
// P.V MFMAs consume (fp32 accumulation), i.e. numerator and denominator see the same weights.
// X8 (round 6; row g1 of the scope table, BASELINE config 5): P.V on the MX instruction v_mfma_scale_f32_16x16x128_f8f6f4 with unit
// block scales -- 128 KEYS per MFMA (the contraction of P.V runs over keys, so the 64-wide heads do not limit it; Q.K^T contracts over
// head_dim = 64 and stays on the 16-bit MFMA), twice the matrix-pipe rate of the non-scaled fp8 MFMA.  The tile is 128 keys; a lane's
// B operand is the 32 P^T values it already holds ([key block kb = 0..7][r = 0..3] -> byte 4 kb + r), the A operand 32 e4m3 bytes of
// one V^T row staged in that key order (two ds_read_b128).  Same roundings as F8 (e4m3 P carried as 2^8 P, e4m3 V, fp32 accumulation,
// row sums of the rounded P on the matrix pipe): only the summation order inside the MFMA differs.
// DEFER (round 6): deferred rescale of the online softmax for the unmasked 16-bit sites.  The running maximum is only moved when some row's
// tile maximum exceeds it by more than 2^ATTN_DEFER_LOG2 (the exact form moved it -- and rescaled the 32 O accumulators of every row -- whenever
// ANY of the wave's rows saw a new maximum: 45 % of the tiles of an S = 4096 site); in between P = exp2(s - m_old) may exceed 1, by at most
// 2^8, which fp16 / bf16 P and the fp32 accumulators hold with the same RELATIVE rounding as before.  Mathematically O = (sum P V) / (sum P)
// for any offset; the decision precedes the tile's exponentials and every P of the tile sees the same offset (the textbook order of
// cdna_hip_programming.md T13).  Not bit-identical to the exact form: TANGO_ATTN_DEFER=0 selects that.
static constexpr float ATTN_DEFER_LOG2 = 8.0f;
// KDMA (round 6, TANGO_ATTN_KDMA): the K tile arrives by LDS-DMA (global_load_lds, 8 rows x 128 B per instruction, the XOR swizzle applied on the SOURCE
// side) instead of through VGPRs and ds_write_b128 -- the staging path was 21 % of the S = 4096 site in the ablation.  V^T still goes through registers
// (its 8-byte column permutation is below the DMA's 16-byte granularity) -- unless the PRODUCER already stored the keys of every block of 32 in that
// order (AttnParams::vt_perm, written by ff_fused.hip qkv_stat_kernel at the level-0 sites): VDMA fetches the V^T tile by LDS-DMA as well, no staging
// registers, no ds_write in the loop.  Unmasked 16-bit sites.
template <typename T, int QB, bool MASKED, int NW, int MINW, bool PB = false, bool F8 = false, bool MSUM = false, bool X8 = false, bool DEFER = false, bool KDMA = false, bool VDMA = false>
__global__ __launch_bounds__(NW * 64, MINW) void attn_kernel(const AttnParams p) {
  static_assert(!VDMA || KDMA, "V^T by LDS-DMA rides on the K-by-DMA form (vt_perm layout: the producer stored the keys in fragment order)");
  static_assert(!KDMA || (!MASKED && !F8 && !X8 && sizeof(T) == 2 && NW == 4), "K by LDS-DMA: unmasked 16-bit sites, four waves");
  static_assert(!DEFER || (!MASKED && !F8 && sizeof(T) == 2), "deferred rescale: unmasked 16-bit sites (the fp8 forms carry P at 2^8 already)");
  static_assert(!X8 || (F8 && MSUM && !MASKED), "MX P.V: the unmasked fp8 variant with matrix-pipe row sums");
  static_assert(!MSUM || sizeof(T) == 2, "matrix-pipe row sums: the P^T fragments of the 16-bit engines only");
  static_assert(!PB || MASKED, "position bias rides on the masked path");
  static_assert(!F8 || (sizeof(T) == 2 && !PB), "fp8 P.V: 16-bit engines, no position bias");
  constexpr int NTH = NW * 64;
  constexpr int EPV = 16 / (int)sizeof(T);
  constexpr int D = 64, KVT = X8 ? 128 : 64;
  constexpr int NKB = KVT / 16;                     // 16-key blocks per tile
  constexpr bool HALF = sizeof(T) == 2;
  constexpr int ROWB = D * (int)sizeof(T);          // bytes per K row == bytes per V^T row (64 kv)
  constexpr int LDSR = HALF ? ROWB : ROWB + 16;     // 128-byte rows are XOR-swizzled, 256-byte rows padded
  constexpr int NKG = ROWB / 64;                    // 64-byte k groups over head_dim
  constexpr int PPR = ROWB / 16;
  constexpr int NPIECE = KVT * PPR;
  constexpr int NPASS = (NPIECE + NTH - 1) / NTH;
  constexpr int VPPR = X8 ? 16 : PPR;               // 16-byte pieces per staged V^T row (X8: 128 keys of a 16-bit row)
  constexpr int STAGE = X8 ? KVT * LDSR + D * 128 : 2 * KVT * LDSR;
  constexpr float LOG2E = 1.4426950408889634f;

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qbase = blockIdx.x * (NW * 16 * QB) + wave * (16 * QB);

  const T* Qp = (const T*)p.q + (int64_t)b * p.Sq * p.ldq + h * D;
  const T* Kp = (const T*)p.k + (int64_t)b * p.Skv * p.ldk + h * D;
  const T* Vp = (const T*)p.vt + ((int64_t)b * p.heads + h) * D * p.ldvt;    // [64][ldvt]
  T* Op = (T*)p.o + (int64_t)b * p.Sq * p.ldo + h * D;
  const float* bias = p.bias ? p.bias + (int64_t)b * p.Skv : nullptr;
  const float sc2 = p.scale * LOG2E;

  u32x4 qf[QB][NKG];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = qbase + qb * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < NKG; ++ks) {
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (q < p.Sq) v = *(const u32x4*)((const unsigned char*)(Qp + (int64_t)q * p.ldq) + ks * 64 + g * 16);
      qf[qb][ks] = v;
    }
  }

  f32x4 oacc[QB][4];
  f32x4 lacc[QB];                       // MSUM: row 0 of this 16 x 16 block = sum over the keys of P^T, per query column
  u32x4 ones = u32x4{0u, 0u, 0u, 0u};
  // (F8: the A fragment of the fp8 MFMA is 8 e4m3 bytes per lane; 1.0 = 0x38)
  if (MSUM && l15 == 0) { const unsigned o2 = F8 ? 0x38383838u : (__is_same(T, f16) ? 0x3C003C00u : 0x3F803F80u); ones = u32x4{o2, o2, o2, o2}; }
  float mrow[QB], lrow[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    lacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    mrow[qb] = -1.0e30f; lrow[qb] = 0.f;
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging: thread -> (row, 16-byte piece) of the K tile and of the V^T tile ----
  // Addresses are a wave-uniform tile base (advanced per tile on the scalar unit) plus a per-thread 32-bit offset that
  // is constant over the loop: no 64-bit VALU address arithmetic per tile (round 1: ~60 VALU of the ~250 per tile in a
  // VALU-co-limited loop).  MASKED == false means Skv % 64 == 0 and ldvt >= Skv: every piece is in range, no tests.
  // X8: the 128-key tile is staged in two parts (its first half travels under Q.K^T, the second under the softmax and P.V): 16 staging
  // registers live at a time instead of 32 (the 16-row form spilled 19 VGPRs at three waves per SIMD otherwise)
  constexpr int NSTG = X8 ? NPASS / 2 : NPASS;
  u32x4 kreg[NSTG], vreg[NSTG];
  unsigned koff[NPASS], voff[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int id = tid + i * NTH;
    const int row = id / PPR, pc = id % PPR;
    koff[i] = (unsigned)row * (unsigned)(p.ldk * (int64_t)sizeof(T)) + pc * 16;
    voff[i] = X8 ? (unsigned)(id / VPPR) * (unsigned)(p.ldvt * (int64_t)sizeof(T)) + (id % VPPR) * 16
                 : (unsigned)row * (unsigned)(p.ldvt * (int64_t)sizeof(T)) + pc * 16;
  }
  // KDMA: wave w fetches K rows (2 w + i) * 8 .. + 7, i = 0, 1; lane -> (row in the group = lane >> 3, LDS slot = lane & 7), source piece = slot ^ (row & 7)
  unsigned kd_off[2] = {0u, 0u}, vd_off[2] = {0u, 0u};
  if constexpr (KDMA) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave * 2 + i) * 8 + (lane >> 3);
      kd_off[i] = (unsigned)row * (unsigned)(p.ldk * (int64_t)sizeof(T)) + (((lane & 7) ^ (row & 7)) * 16);
      vd_off[i] = (unsigned)row * (unsigned)(p.ldvt * (int64_t)sizeof(T)) + (((lane & 7) ^ (row & 7)) * 16);   // row = head-dim index of V^T
    }
  }
  auto dma_k = [&](int kv0, int st) {
    if constexpr (KDMA) {
      typedef const __attribute__((address_space(1))) void* gp_t;
      typedef __attribute__((address_space(3))) void* lp_t;
      const unsigned char* Kt = (const unsigned char*)Kp + (int64_t)kv0 * p.ldk * (int64_t)sizeof(T);
      const unsigned lb = (unsigned)(uintptr_t)(lp_t)smem + (unsigned)st * STAGE + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 2048u;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        unsigned o = kd_off[i];
        asm volatile("" : "+v"(o));
        __builtin_amdgcn_global_load_lds((gp_t)(Kt + o), (lp_t)(uintptr_t)(lb + (unsigned)i * 1024u), 16, 0, 0);
      }
      if constexpr (VDMA) {
        const unsigned char* Vt = (const unsigned char*)Vp + (int64_t)kv0 * (int64_t)sizeof(T);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          unsigned o = vd_off[i];
          asm volatile("" : "+v"(o));
          __builtin_amdgcn_global_load_lds((gp_t)(Vt + o), (lp_t)(uintptr_t)(lb + (unsigned)(KVT * LDSR) + (unsigned)i * 1024u), 16, 0, 0);
        }
      }
    }
  };
  auto load_tile = [&](int kv0) {
    const unsigned char* Kt = (const unsigned char*)Kp + (int64_t)kv0 * p.ldk * (int64_t)sizeof(T);
    const unsigned char* Vt = (const unsigned char*)Vp + (int64_t)kv0 * (int64_t)sizeof(T);
#pragma unroll
    for (int i = 0; i < NSTG; ++i) {
      const int id = tid + i * NTH;
      const int row = id / PPR, pc = id % PPR;
      if (NPIECE % NTH != 0 && id >= NPIECE) continue;
      if (MASKED) {
        u32x4 kk = u32x4{0u, 0u, 0u, 0u}, vv = u32x4{0u, 0u, 0u, 0u};
        if (kv0 + row < p.Skv) kk = *(const u32x4*)(Kt + koff[i]);
        if (kv0 + pc * EPV < p.ldvt) vv = *(const u32x4*)(Vt + voff[i]);
        kreg[i] = kk; vreg[i] = vv;
      } else {
        if constexpr (!KDMA) kreg[i] = *(const u32x4*)(Kt + koff[i]);
        if constexpr (!VDMA) vreg[i] = *(const u32x4*)(Vt + voff[i]);
      }
    }
  };
  auto store_tile = [&](int st) {
    unsigned char* Ks = smem + st * STAGE;
    unsigned char* Vs = Ks + KVT * LDSR;
#pragma unroll
    for (int i = 0; i < NSTG; ++i) {
      const int id = tid + i * NTH;
      const int row = id / PPR, pc = id % PPR;
      if (NPIECE % NTH != 0 && id >= NPIECE) continue;
      if (F8) {
        *(u32x4*)(Ks + row * LDSR + ((pc ^ (row & 7)) * 16)) = kreg[i];
        // V^T as e4m3, 64-byte rows of 8 x 8-byte slots: slot 4j + gg holds [kv 32j + 4gg .. +3 | kv 32j + 16 + 4gg .. +3] (the
        // key order in which a lane's P^T fragment presents them); slot ^= 2 * ((row >> 2) & 3) spreads the 16 rows x 2 lane groups
        // of a ds_read_b64 half-wave over all 64 banks
        T e[8];
        __builtin_memcpy(e, &vreg[i], 16);
        float f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) f[u] = __builtin_amdgcn_fmed3f(to_f(e[u]), -448.f, 448.f);
        const int j = pc >> 2, hi = (pc >> 1) & 1, gg0 = (pc & 1) * 2, sw = 2 * ((row >> 2) & 3);
        unsigned char* vr = Vs + row * 64;
        *(unsigned*)(vr + (((4 * j + gg0) ^ sw) * 8) + hi * 4) = pack_fp8x4(f[0], f[1], f[2], f[3]);
        *(unsigned*)(vr + (((4 * j + gg0 + 1) ^ sw) * 8) + hi * 4) = pack_fp8x4(f[4], f[5], f[6], f[7]);
      } else if (HALF) {
        if constexpr (!KDMA) *(u32x4*)(Ks + row * LDSR + ((pc ^ (row & 7)) * 16)) = kreg[i];
        // V^T: piece pc holds kv = 8pc..8pc+7 of this 64-tile: 32-block j = pc>>2, c = (pc&3)*8 + e.
        // column c = 16*hi + 4*gg + r is stored at c' = 8*gg + 4*hi + r (so [hi=0 | hi=1] of one gg are adjacent)
        if constexpr (!VDMA) {
        const int j = pc >> 2, hi = (pc >> 1) & 1, gg0 = (pc & 1) * 2;
        const u32x2 lo = u32x2{vreg[i][0], vreg[i][1]}, hh = u32x2{vreg[i][2], vreg[i][3]};
        // bytes within the row: 64*j + 2*(8*gg + 4*hi) ; 16-byte piece index = 4*j + gg, half = hi
        unsigned char* vr = Vs + row * LDSR;
        *(u32x2*)(vr + (((4 * j + gg0) ^ (row & 7)) * 16) + hi * 8) = lo;
        *(u32x2*)(vr + (((4 * j + gg0 + 1) ^ (row & 7)) * 16) + hi * 8) = hh;
        }
      } else {
        *(u32x4*)(Ks + row * LDSR + pc * 16) = kreg[i];
        *(u32x4*)(Vs + row * LDSR + pc * 16) = vreg[i];
      }
    }
  };

  // X8 staging, two parts of NSTG passes each (separate lambdas: sharing load_tile / store_tile changed the register allocation of
  // every other instantiation of this template)
  auto load_part = [&](int kv0, const int part) {
    const unsigned char* Kt = (const unsigned char*)Kp + (int64_t)kv0 * p.ldk * (int64_t)sizeof(T);
    const unsigned char* Vt = (const unsigned char*)Vp + (int64_t)kv0 * (int64_t)sizeof(T);
#pragma unroll
    for (int j = 0; j < NSTG; ++j) {
      kreg[j] = *(const u32x4*)(Kt + koff[part * NSTG + j]);
      vreg[j] = *(const u32x4*)(Vt + voff[part * NSTG + j]);
    }
  };
  auto store_part = [&](int st, const int part) {
    unsigned char* Ks = smem + st * STAGE;
#pragma unroll
    for (int j = 0; j < NSTG; ++j) {
      const int i = part * NSTG + j;
      const int id = tid + i * NTH;
      const int row = id / PPR, pc = id % PPR;
        *(u32x4*)(Ks + row * LDSR + ((pc ^ (row & 7)) * 16)) = kreg[j];
        // V^T as e4m3, 128-byte rows of eight 16-byte slots: lane group gg of the MX MFMA reads bytes [32 gg, 32 gg + 32) = slots 2 gg,
        // 2 gg + 1, byte 4 kb + r of them = key 16 kb + 4 gg + r.  Piece vpc of V^T row vrow holds keys 8 vpc .. 8 vpc + 7: kb = vpc >> 1,
        // gg = 2 (vpc & 1) + (e >> 2).  slot ^= A(row) with A = {0,1,4,5,6,7,2,3}[(row >> 1) & 7]: the 16 lanes of every ds_read_b128
        // lane group (MI355X_MICROARCH.md) then touch 16 distinct 16-byte slots of the 256-byte bank row
        const int vrow = id / VPPR, vpc = id % VPPR;
        T e[8];
        __builtin_memcpy(e, &vreg[j], 16);
        float f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) f[u] = __builtin_amdgcn_fmed3f(to_f(e[u]), -448.f, 448.f);
        const int kb = vpc >> 1, gg0 = 2 * (vpc & 1);
        const int ar = (vrow >> 1) & 7, sw = ar < 2 ? ar : (ar < 6 ? ar + 2 : ar - 4);
        unsigned char* vr = Ks + KVT * LDSR + vrow * 128 + 4 * (kb & 3);
        *(unsigned*)(vr + (((2 * gg0 + (kb >> 2)) ^ sw) * 16)) = pack_fp8x4(f[0], f[1], f[2], f[3]);
        *(unsigned*)(vr + (((2 * gg0 + 2 + (kb >> 2)) ^ sw) * 16)) = pack_fp8x4(f[4], f[5], f[6], f[7]);
    }
  };

  const int ntile = (p.Skv + KVT - 1) / KVT;
  if constexpr (X8) {
    load_part(0, 0); store_part(0, 0);
    load_part(0, 1); store_part(0, 1);
  } else {
    dma_k(0, 0);
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

  for (int t = 0; t < ntile; ++t) {
    const int kv0 = t * KVT;
    const bool more = t + 1 < ntile;
    if constexpr (X8) { if (more) load_part(kv0 + KVT, 0); }
    else { if (more) { dma_k(kv0 + KVT, (t + 1) & 1); load_tile(kv0 + KVT); } }
    const unsigned char* Ks = smem + (t & 1) * STAGE;
    const unsigned char* Vs = Ks + KVT * LDSR;

    // ---- S^T = K Q^T ----
    f32x4 sacc[QB][NKB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) sacc[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int ks = 0; ks < NKG; ++ks) {
        const int off = HALF ? (((ks * 4 + g) ^ (lane & 7)) * 16) : ks * 64 + g * 16;
        const u32x4 kf = *(const u32x4*)(Ks + (kb * 16 + l15) * LDSR + off);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) AMma<T>::run(sacc[qb][kb], kf, qf[qb][ks]);
      }
    }

    if constexpr (X8) { if (more) { store_part((t + 1) & 1, 0); load_part(kv0 + KVT, 1); } }

    // ---- online softmax in the exp2 domain (per lane: one q column, 16 kv values per q block) ----
    // MASKED == false: self-attention with Skv a multiple of 64 (no bias, no tail) -> no per-key offsets at all
    float bv[MASKED ? 4 : 1][4];
    if (MASKED) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kv = kv0 + kb * 16 + g * 4 + r;
          bv[MASKED ? kb : 0][r] = (kv < p.Skv) ? (bias ? bias[kv] * LOG2E : 0.f) : -1.0e30f;
        }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mt = -1.0e30f;
      float mnew;
      if (MASKED) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sv = sacc[qb][kb][r] * sc2 + bv[MASKED ? kb : 0][r];
            if (PB) {
              const int q = qbase + qb * 16 + l15, kv = kv0 + kb * 16 + g * 4 + r;
              if (q < p.Sq && kv < p.Skv) sv += p.pos_bias[((int64_t)h * p.Sq + q) * p.Skv + kv] * LOG2E;
            }
            sacc[qb][kb][r] = sv;
            mt = fmaxf(mt, sv);
          }
        mnew = fmaxf(mrow[qb], quad_max(mt));
      } else {
        // unmasked: max(sc2 * s) == sc2 * max(s) (sc2 > 0), so the scale is applied once to the maximum and otherwise
        // rides along in the exponent's FMA
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mt = fmaxf(mt, sacc[qb][kb][r]);
        mnew = fmaxf(mrow[qb], quad_max(mt) * sc2);
      }
      // lazy rescale: after the first few tiles the running maximum rarely moves; when it moved for NO query of this wave
      // the O / l rescale (alpha == 1 exactly) and its exponential are skipped -- same values, 32 multiplies fewer
      // (DEFER: ... moved by more than 2^8 for no query; the offset of this tile's exponentials is then the OLD maximum)
      const bool move = DEFER ? mnew > mrow[qb] + ATTN_DEFER_LOG2 : mnew > mrow[qb];
      if (__builtin_amdgcn_ballot_w64(move) != 0ull) {
        const float alpha = __builtin_amdgcn_exp2f(mrow[qb] - mnew);
        if (MSUM) lacc[qb] *= alpha;
        else lrow[qb] *= alpha;
        mrow[qb] = mnew;
#pragma unroll
        for (int db = 0; db < 4; ++db) oacc[qb][db] *= alpha;
      }
      float rs = 0.f;
      const float moff = DEFER ? mrow[qb] : (F8 ? mnew - 8.f : mnew);   // F8: P is carried as 2^8 P (<= 256 < 448 = e4m3 max); cancels in O / l
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = MASKED ? __builtin_amdgcn_exp2f(sacc[qb][kb][r] - moff)
                                  : __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qb][kb][r], sc2, -moff));
          sacc[qb][kb][r] = pv;
          if (!MSUM) rs += pv;
        }
      if (!MSUM) lrow[qb] += rs;          // per-lane partial of the row sum: the four kv slices of a query are added once, after the loop
    }

    // ---- O^T += V^T P^T ----
    if constexpr (X8) {
      typedef int i32x8 __attribute__((ext_vector_type(8)));
      constexpr int UNIT = 0x7F7F7F7F;                   // E8M0 block scales 2^0 for both operands
      i32x8 pf8[QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) pf8[qb][kb] = (int)pack_fp8x4(sacc[qb][kb][0], sacc[qb][kb][1], sacc[qb][kb][2], sacc[qb][kb][3]);
      const int ar = (l15 >> 1) & 7, sw = ar < 2 ? ar : (ar < 6 ? ar + 2 : ar - 4);     // (row >> 1) & 7 with row = 16 db + l15
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const unsigned char* vr = Vs + (db * 16 + l15) * 128;
        const u32x4 v0 = *(const u32x4*)(vr + (((2 * g) ^ sw) * 16)), v1 = *(const u32x4*)(vr + (((2 * g + 1) ^ sw) * 16));
        const i32x8 vf8 = i32x8{(int)v0[0], (int)v0[1], (int)v0[2], (int)v0[3], (int)v1[0], (int)v1[1], (int)v1[2], (int)v1[3]};
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          oacc[qb][db] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(vf8, pf8[qb], oacc[qb][db], 0, 0, 0, UNIT, 0, UNIT);
      }
      const int o8 = (int)ones[0];
      const i32x8 ones8 = i32x8{o8, o8, o8, o8, o8, o8, o8, o8};
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
        lacc[qb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones8, pf8[qb], lacc[qb], 0, 0, 0, UNIT, 0, UNIT);
    } else if constexpr (F8) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        long pf8[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const unsigned lo = pack_fp8x4(sacc[qb][2 * j][0], sacc[qb][2 * j][1], sacc[qb][2 * j][2], sacc[qb][2 * j][3]);
          const unsigned hh = pack_fp8x4(sacc[qb][2 * j + 1][0], sacc[qb][2 * j + 1][1], sacc[qb][2 * j + 1][2], sacc[qb][2 * j + 1][3]);
          pf8[qb] = (long)(((unsigned long long)hh << 32) | lo);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const int row = db * 16 + l15;
          const long vf8 = *(const long*)(Vs + row * 64 + (((4 * j + g) ^ (2 * ((row >> 2) & 3))) * 8));
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) oacc[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(vf8, pf8[qb], oacc[qb][db], 0, 0, 0);
        }
        if constexpr (MSUM) {
          // round 5: the denominator is the sum of the e4m3-ROUNDED weights the products above consume, not of
          // the unrounded ones -- numerator and denominator see the same P, so the rounding of P no longer biases O = (P V) / sum(P)
          const long ones8 = (long)(((unsigned long long)ones[1] << 32) | ones[0]);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) lacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(ones8, pf8[qb], lacc[qb], 0, 0, 0);
        }
      }
    } else if constexpr (HALF) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        u32x4 pf[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          T e[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { e[r] = from_f<T>(sacc[qb][2 * j][r]); e[4 + r] = from_f<T>(sacc[qb][2 * j + 1][r]); }
          __builtin_memcpy(&pf[qb], e, 16);
        }
        const int off = ((4 * j + g) ^ (lane & 7)) * 16;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const u32x4 vf = *(const u32x4*)(Vs + (db * 16 + l15) * LDSR + off);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) AMma<T>::run(oacc[qb][db], vf, pf[qb]);
        }
        if constexpr (MSUM) {
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) AMma<T>::run(lacc[qb], ones, pf[qb]);
        }
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const u32x4 vf = *(const u32x4*)(Vs + (db * 16 + l15) * LDSR + (kb * 16 + g * 4) * 4);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            const u32x4 pf = __builtin_bit_cast(u32x4, sacc[qb][kb]);
            AMma<T>::run(oacc[qb][db], vf, pf);
          }
        }
      }
    }
    if constexpr (X8) { if (more) store_part((t + 1) & 1, 1); }
    else { if (more) store_tile((t + 1) & 1); }
    __syncthreads();
  }

  // ---- normalise and store: lane holds O[q][db*16 + g*4 + 0..3] ----
  // Direct stores are 8 (16) bytes per lane = 32-byte runs; with 16-byte-aligned output rows the wave parks its
  // QB*16 x 64 tile in the (now idle) K/V LDS and writes whole 128-byte (256-byte) rows instead -- same values.
  constexpr int OPITCH = ROWB + 16;
  // (measured: -15 % on the short-Skv cross-attention sites, which are store-bound; +1.5 % on the long self-attention
  //  sites, where the extra live state costs more than the stores -> staged only in the MASKED instantiation)
  const bool ostage = MASKED && ((p.ldo * (int64_t)sizeof(T)) % 16 == 0) && (((uintptr_t)p.o) % 16 == 0);
  static_assert(NW * QB * 16 * OPITCH <= 2 * STAGE, "output staging must fit in the K/V stages");
  unsigned char* const ost = smem + wave * (QB * 16 * OPITCH);
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = qbase + qb * 16 + l15;
    float lsum;
    if (MSUM) lsum = __shfl(lacc[qb][0], l15);           // accumulator row 0 lives in lanes 0..15 (g = 0), element 0
    else {
      lsum = lrow[qb];
      lsum += __shfl_xor(lsum, 16);
      lsum += __shfl_xor(lsum, 32);
    }
    const float inv = 1.0f / lsum;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      T e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = from_f<T>(oacc[qb][db][r] * inv);
      if (ostage) __builtin_memcpy(ost + (qb * 16 + l15) * OPITCH + (db * 16 + g * 4) * (int)sizeof(T), e, 4 * sizeof(T));
      else if (q < p.Sq) __builtin_memcpy(Op + (int64_t)q * p.ldo + db * 16 + g * 4, e, 4 * sizeof(T));
    }
  }
  if (ostage) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < QB * PPR / 4; ++i) {
      const int idx = i * 64 + lane;
      const int row = idx / PPR, pc = idx % PPR;
      const int q = qbase + row;
      if (q < p.Sq) *(u32x4*)(Op + (int64_t)q * p.ldo + pc * EPV) = *(const u32x4*)(ost + row * OPITCH + pc * 16);
    }
  }
}

bool attention_vt_perm_ok(int dtype, const AttnParams& p) {
  return dtype != DT_F32 && !p.bias && !p.pos_bias && p.Skv % 64 == 0 && p.Sq > 512 && p.fp8_pv == 0 && p.ldvt % 8 == 0;
}

template <typename T>
static int attn_launch(const AttnParams& p, hipStream_t s) {
  if ((p.ldq * (int64_t)sizeof(T)) % 16 || (p.ldk * (int64_t)sizeof(T)) % 16 || (p.ldvt * (int64_t)sizeof(T)) % 16 ||
      (p.ldo * (int64_t)sizeof(T)) % 8)
    TANGO_FAIL("attention: ld alignment");
  const bool masked = p.bias != nullptr || (p.Skv % 64) != 0;
  // a permuted V^T is only understood by the LDS-DMA form of the long unmasked 16-bit sites: never read it with another kernel
  if (p.vt_perm && !attention_vt_perm_ok(TypeTag<T>::dt, p)) TANGO_FAIL("attention: vt_perm needs an unmasked 16-bit site with Sq > 512, Skv % 64 == 0, no fp8 P.V");
  if (p.fp8_pv && sizeof(T) != 2) TANGO_FAIL("attention: fp8 P.V needs a 16-bit engine (Q.K^T stays in the engine dtype)");
  //  never fall back silently: a config-5 run must not measure the 16-bit kernel under the fp8 label
  if (p.fp8_pv && (masked || p.pos_bias)) TANGO_FAIL("attention: fp8 P.V is implemented for unmasked sites with Skv % 64 == 0 only");
  if (p.fp8_pv == 2 && p.Skv % 128 != 0) TANGO_FAIL("attention: MX fp8 P.V (128 keys per MFMA) needs Skv % 128 == 0");
  if constexpr (sizeof(T) == 2) {
    if (p.fp8_pv == 2) {
      // MX P.V (X8): 128-key tiles.  Two forms, picked by TANGO_ATTN_X8_QB (measured: DESIGN.md section 5, round 6): 32 query rows per
      // wave at two waves per SIMD (the 64 S^T accumulators of a 128-key tile do not fit three), or 16 rows per wave at three
      const bool qb2 = tuning().attn_x8_qb == 2 && p.Sq % 128 == 0;
      if (qb2) {
        dim3 grid((unsigned)(p.Sq / 128), (unsigned)p.heads, (unsigned)p.B);
        hipLaunchKernelGGL((attn_kernel<T, 2, false, 4, 2, false, true, true, true>), grid, dim3(256), 0, s, p);
      } else {
        dim3 grid((unsigned)((p.Sq + 63) / 64), (unsigned)p.heads, (unsigned)p.B);
        hipLaunchKernelGGL((attn_kernel<T, 1, false, 4, 3, false, true, true, true>), grid, dim3(256), 0, s, p);
      }
      TANGO_HIP(hipGetLastError());
      return 0;
    }
  }
  if (p.pos_bias) {   // text-encoder self-attention (short sequences): one query block per wave
    dim3 grid((unsigned)((p.Sq + 63) / 64), (unsigned)p.heads, (unsigned)p.B);
    hipLaunchKernelGGL((attn_kernel<T, 1, true, 4, 3, true>), grid, dim3(256), 0, s, p);
    TANGO_HIP(hipGetLastError());
    return 0;
  }
  if (p.Sq > 512) {
    // 4 waves x 32 query rows per workgroup at 3 workgroups/CU.  Wider workgroups (6 or 8 waves sharing one K/V tile
    // stream, i.e. 1.5-2x fewer L2->LDS bytes per flop) were measured in round 1: 8 waves 10.5 ms vs 9.2 ms per step
    // at Sq = Skv = 4096; 48 or 64 query rows per wave at 2 waves/SIMD: 9.2 / 10.1 ms -- the kernel is VALU (softmax)
    // co-limited, not fill-limited.
    constexpr int QB = 2;
    dim3 grid((unsigned)((p.Sq + 64 * QB - 1) / (64 * QB)), (unsigned)p.heads, (unsigned)p.B);
    if (masked) hipLaunchKernelGGL((attn_kernel<T, QB, true, 4, 3>), grid, dim3(256), 0, s, p);
    else if constexpr (sizeof(T) == 2) {
      if (p.vt_perm) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, false, true, false, false, true, true>), grid, dim3(256), 0, s, p);
      else if (p.fp8_pv && tuning().attn_msum) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, true, true>), grid, dim3(256), 0, s, p);
      else if (p.fp8_pv) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, true>), grid, dim3(256), 0, s, p);
      else if (tuning().attn_msum && tuning().attn_defer) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, false, true, false, true>), grid, dim3(256), 0, s, p);
      else if (tuning().attn_msum && tuning().attn_kdma) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, false, true, false, false, true>), grid, dim3(256), 0, s, p);
      else if (tuning().attn_msum) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, false, true>), grid, dim3(256), 0, s, p);
      else if (tuning().attn_defer) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, false, false, false, true>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3>), grid, dim3(256), 0, s, p);
    } else hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3>), grid, dim3(256), 0, s, p);
  } else if (tuning().attn_qb2_min_wgs > 0 &&
             (long)((p.Sq + 127) / 128) * p.heads * p.B >= tuning().attn_qb2_min_wgs && p.Sq % 128 == 0) {
    // short sequences at large batch (levels 1-2 at B >= 8): 32 query rows per wave halve the K / V tile traffic and the workgroup count
    // once there are enough workgroups to fill the chip anyway (round 4)
    constexpr int QB = 2;
    dim3 grid((unsigned)((p.Sq + 64 * QB - 1) / (64 * QB)), (unsigned)p.heads, (unsigned)p.B);
    if (masked) hipLaunchKernelGGL((attn_kernel<T, QB, true, 4, 3>), grid, dim3(256), 0, s, p);
    else if constexpr (sizeof(T) == 2) {
      if (p.fp8_pv && tuning().attn_msum) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, true, true>), grid, dim3(256), 0, s, p);
      else if (p.fp8_pv) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, true>), grid, dim3(256), 0, s, p);
      else if (tuning().attn_msum && tuning().attn_defer) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, false, true, false, true>), grid, dim3(256), 0, s, p);
      else if (tuning().attn_msum) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, false, true>), grid, dim3(256), 0, s, p);
      else if (tuning().attn_defer) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, false, false, false, true>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3>), grid, dim3(256), 0, s, p);
    } else hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3>), grid, dim3(256), 0, s, p);
  } else {
    constexpr int QB = 1;
    dim3 grid((unsigned)((p.Sq + 64 * QB - 1) / (64 * QB)), (unsigned)p.heads, (unsigned)p.B);
    if (masked) hipLaunchKernelGGL((attn_kernel<T, QB, true, 4, 3>), grid, dim3(256), 0, s, p);
    else if constexpr (sizeof(T) == 2) {
      if (p.fp8_pv && tuning().attn_msum) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, true, true>), grid, dim3(256), 0, s, p);
      else if (p.fp8_pv) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, true>), grid, dim3(256), 0, s, p);
      else if (tuning().attn_msum && tuning().attn_defer) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, false, true, false, true>), grid, dim3(256), 0, s, p);
      else if (tuning().attn_msum) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, false, true>), grid, dim3(256), 0, s, p);
      else if (tuning().attn_defer) hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3, false, false, false, false, true>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3>), grid, dim3(256), 0, s, p);
    } else hipLaunchKernelGGL((attn_kernel<T, QB, false, 4, 3>), grid, dim3(256), 0, s, p);
  }
  TANGO_HIP(hipGetLastError());
  return 0;
}

int launch_attention(int dtype, const AttnParams& p, hipStream_t s) {
  switch (dtype) {
    case DT_F32: return attn_launch<float>(p, s);
    case DT_F16: return attn_launch<f16>(p, s);
    case DT_BF16: return attn_launch<bf16>(p, s);
  }
  TANGO_FAIL("attention: bad dtype");
}

}  // namespace tango
