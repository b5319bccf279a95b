// Every run-time switch of the kernel dispatch layer, in ONE place, read once per process.
//
// The product path needs none of them: the defaults below are what the benchmark and the drop-in API run.  They exist for
//   (a) the parity / determinism tests, which must reach the 8-wave kernels with test-sized problems, and
//   (b) A/B measurements inside one process or one gpurun call (tools/profile_unet_ops.py): taking a kernel family out of
//       the dispatch shows what it buys on the real UNet shapes.
// Experiments that were measured and decided in rounds 1-2 (persistent GEMM, ping-pong variants of the 256 x 160 kernels,
// halo-conv ablation hooks, the un-patched streaming-LN epilogue, per-workgroup phase traces, split-K granularity sweeps)
// are no longer compiled into libtango_hip.so; their sources live under tools/experiments/ with the evidence in profiles/.
#pragma once
#include <cstdlib>

namespace tango {

struct Tuning {
  bool force_big_kernels;   // TANGO_FORCE_DMA_GEMM=1   tests: 8-wave LDS-DMA kernels accept grids that do not fill the chip
  bool no_wide_conv;        // TANGO_NO_WIDE_CONV=1     A/B: conv3x3_wide_kernel (256 x 320 halo conv) out of the dispatch
  bool no_wide_gemm;        // TANGO_NO_WIDE_GEMM=1     A/B: gemm_wide_kernel (256 x 320 GEMM) out
  bool no_halo_conv;        // TANGO_NO_HALO_CONV=1     A/B: conv3x3_halo_kernel (256 x 160 halo conv) out
  bool no_dma_gemm;         // TANGO_NO_DMA_GEMM=1      A/B: gemm_dma_kernel (256 x 160 gather GEMM) out
  bool no_stream;           // TANGO_NO_STREAM=1        A/B: lin_stream_kernel out (plain linears only; folded-LN shapes need it)
  bool no_small_tile;       // TANGO_NO_SMALL_TILE=1    A/B: 64 x 64 tiles for small-M linears out (back to split-K + reduce / the streaming kernel) (round 3)
  bool no_xattn_fused;      // TANGO_NO_XATTN_FUSED=1   A/B: fused cross-attention block kernel out (round 3)
  bool no_single_key;       // TANGO_NO_SINGLE_KEY=1    A/B: single-key (unconditional-row) cross-attention shortcut out (round 4)
  int wide_prio;            // TANGO_WIDE_PRIO=0..2     A/B: wave priorities in those kernels' ping-pong loops (gemm_wide.hip: PRIO; default 0) (round 4)
  int wide_sched;           // TANGO_WIDE_SCHED=0..1    A/B: where the 256 x 320 kernels issue their LDS-DMAs (gemm_wide.hip: SCH; default 1) (round 4)
  bool no_stream_ln_geglu;  // TANGO_NO_STREAM_LN_GEGLU=1 A/B: folded-LayerNorm GEGLU projections leave the streaming kernel (LayerNorm kernel + a GEGLU GEMM instead) (round 4)
  bool no_rowvec_fuse;      // TANGO_NO_ROWVEC_FUSE=1   A/B: the single-key rows' constant stays a separate pass instead of riding attn1's to_out epilogue (round 4)
  bool no_ln_xstats;        // TANGO_NO_LN_XSTATS=1     A/B: GEGLU projections of levels 1-2 back to LayerNorm kernel + plain GEMM (instead of ln_stats + folded weights) (round 4)
  bool no_gn_coop;          // TANGO_NO_GN_COOP=1       A/B: cooperative single-launch GroupNorm (norm.hip gn_coop_kernel) out (round 4)
  bool gn_coop_all;         // TANGO_GN_COOP_ALL=1      tests: that kernel for every geometry it can hold, not only where it was measured faster
  bool gn_coop_force_fb;    // TANGO_GN_COOP_FORCE_FALLBACK=1 tests: every workgroup of that kernel takes the no-rendezvous fallback (bit-identical results required) (round 5)
  int graph_steps;          // TANGO_GRAPH_STEPS=k      denoise: k UNet steps per captured hipGraph (default 1 = one replay per step: measured, no inter-replay gap to remove) (round 5)
  int unet_chains;          // TANGO_UNET_CHAINS=1|2    denoise: the UNet batch as one kernel sequence or as two independent halves in two branches of the captured graph (Engine::unet_chains_for; unset = the measured rule) (round 5)
  bool no_cfg_shared;       // TANGO_NO_CFG_SHARED=1    A/B: the CFG-shared prefix of a guidance step (conv_in ... first self-attention once for both halves) off (round 5)
  bool stream_spec;         // TANGO_STREAM_SPEC=0|1     lin_stream_kernel: compile-time GEGLU + folded-LayerNorm epilogue for the level-0 projection (default on) (round 5)
  int attn_qb2_min_wgs;     // TANGO_ATTN_QB2_MIN_WGS=n attention with Sq <= 512: 32 query rows per wave once that still leaves n workgroups (default 512: level-2 self-attention at B=32 0.319 -> 0.247 ms, profiles/r4_c14_attn_qb2_ab_b32.txt); 0 = never (round 4)
  bool no_halo_narrow;      // TANGO_NO_HALO_NARROW=1   A/B: conv_out (N <= 32) back on the 256 x 16 tile kernel instead of the halo kernel's 256 x 32 tile (round 6)
  int gn_slab;              // TANGO_GN_SLAB=0|1        GroupNorm of <= 256-row samples with vector-aligned groups (UNet levels 2-3): one workgroup keeps a rows x 4-group slab in registers -- one launch, one read, one write (round 6)
  int gn_small_mb;          // TANGO_GN_SMALL_MB=n      GroupNorm: the one-launch (sample, group)-per-workgroup kernel up to n MiB of input (default 8)
  int wide_pers;            // TANGO_WIDE_PERS=n        256 x 320 GEMM: the persistent form (next tile's first chunk prefetched behind the epilogue) from n tiles per CU on (default 2: linears of a B = 32 step 20.0 -> 19.7 ms, bit-identical results, profiles/r4_c16_wide_pers_ab_b32.txt); 0 = never (round 4)
  bool attn_msum;           // TANGO_ATTN_MSUM=0|1      unmasked 16-bit attention: softmax row sums on the matrix pipe (attention.hip MSUM; default on: S = 4096 site 8.00 -> 7.79 ms at B = 32, profiles/r4_c17_attn_msum_ab_b32.txt) (round 4)
  int duo_maxk;             // TANGO_DUO_MAXK=k         gemm_duo_kernel (256 x 160, two workgroups per CU) takes linears with K <= k; 0 = out of the dispatch; unset = the measured rule in gemm_duo_ok() (round 4)
  int duo_min_tiles;        // TANGO_DUO_MIN_TILES=n    ... that have at least n tiles of 256 x 160
  int duo_mask;             // TANGO_DUO_MASK=bits      ... of these classes: 1 plain, 2 GEGLU, 4 folded LayerNorm (incl. transposed V), 8 folded LayerNorm + GEGLU
  int duo_prio;             // TANGO_DUO_PRIO=0..1      0 = s_setprio 1 around its MFMAs, 1 = no priority changes
  int ff_fused;             // TANGO_FF_FUSED=0|1|2     level-0 feed-forward (norm3 -> GEGLU projection -> ff.net.2 -> + residual) in one launch (ff_fused.hip, round 6); 0 = two GEMMs, 2 = its compiler-scheduled main loop (A/B arm)
  int ff_min_rows;          // TANGO_FF_MIN_ROWS=n      ... for at least n rows (default 32768 = one 128-row workgroup per CU: at B = 1, 64 workgroups, the two GEMMs are faster: 0.27 vs 0.41 ms per step)
  int qkv_stat;             // TANGO_QKV_STAT=0|1       level-0 norm1 + QKV projection on the activation-stationary kernel (ff_fused.hip qkv_stat_kernel, round 6) ...
  int qkv_min_rows;         // TANGO_QKV_MIN_ROWS=n     ... for at least n rows
  int gn_proj_stat;         // TANGO_GN_PROJ_STAT=0|1   level-0 GroupNorm -> proj_in: statistics pass + the activation-stationary kernel normalising on its way in (round 6); 0 = GroupNorm kernel + GEMM
  bool gn_fold;             // TANGO_GN_FOLD=1          GroupNorm -> proj_in as ONE GEMM with per-sample folded weights at levels 0-1 (round 6: built, within tolerance, measured +-0.0 ms per step: off)
  int attn_kdma;            // TANGO_ATTN_KDMA=0|1      unmasked 16-bit attention with Sq > 512: the K tile by LDS-DMA instead of through VGPRs (round 6)
  int attn_vdma;            // TANGO_ATTN_VDMA=0|1|2    self-attention with Sq > 512: v^T written in fragment order by the producer, K AND V^T tiles by LDS-DMA (round 6);
                            //                          1 = only where qkv_stat_kernel produces it (level 0), 2 = the GEMM routes' transposed epilogues as well (level 1)
  int attn_defer;           // TANGO_ATTN_DEFER=0|1     unmasked 16-bit attention: move the running softmax maximum only when a tile exceeds it by more than 2^8 (round 6); 0 = exact lazy rescale
  int attn_x8_qb;           // TANGO_ATTN_X8_QB=1|2     MX fp8 P.V attention (unet_attn_fp8 = 2): 16 query rows per wave at three waves per SIMD, or 32 at two (round 6)
  int conv_tall;            // TANGO_CONV_TALL=0|1      3x3 wide conv on the 512-pixel x 160-channel form of the tile where the halo fits (round 6: half the weight DMA per MFMA)
  int wide_pipe;            // TANGO_WIDE_PIPE=0|1|2    256 x 320 GEMM: in-wave software pipeline (fragments of item i+1 requested under the MFMAs of item i) instead of the ping-pong read / multiply parts; 1 = staggered halves, two barriers per item, 2 = all waves in step, one barrier per item (round 6: both slower than ping-pong on the GEMMs)
  int conv_pipe;            // TANGO_CONV_PIPE=0..3     256 x 320 conv: 0 = ping-pong kernel, 1 / 2 as above, 3 = one barrier per item with the halves half an item apart.  Default 2 (round 6: convs -1.4 %, bit-identical)
};

inline Tuning read_tuning() {
  auto on = [](const char* k) { const char* v = getenv(k); return v != nullptr && v[0] != '0'; };
  Tuning x;
  x.force_big_kernels = on("TANGO_FORCE_DMA_GEMM");
  x.no_wide_conv = on("TANGO_NO_WIDE_CONV");
  x.no_wide_gemm = on("TANGO_NO_WIDE_GEMM");
  x.no_halo_conv = on("TANGO_NO_HALO_CONV");
  x.no_dma_gemm = on("TANGO_NO_DMA_GEMM");
  x.no_stream = on("TANGO_NO_STREAM");
  x.no_small_tile = on("TANGO_NO_SMALL_TILE");
  x.no_xattn_fused = on("TANGO_NO_XATTN_FUSED");
  x.no_single_key = on("TANGO_NO_SINGLE_KEY");
  x.no_stream_ln_geglu = on("TANGO_NO_STREAM_LN_GEGLU");
  x.no_gn_coop = on("TANGO_NO_GN_COOP");
  x.no_ln_xstats = on("TANGO_NO_LN_XSTATS");
  x.no_rowvec_fuse = on("TANGO_NO_ROWVEC_FUSE");
  const char* ws = getenv("TANGO_WIDE_SCHED");
  x.wide_sched = (ws && ws[0] >= '0' && ws[0] <= '1') ? ws[0] - '0' : 1;
  auto num = [](const char* k, int dflt) { const char* v = getenv(k); return (v && v[0]) ? atoi(v) : dflt; };
  x.gn_coop_all = on("TANGO_GN_COOP_ALL");
  x.gn_coop_force_fb = on("TANGO_GN_COOP_FORCE_FALLBACK");
  x.graph_steps = num("TANGO_GRAPH_STEPS", 0);
  x.unet_chains = num("TANGO_UNET_CHAINS", 0);
  x.no_cfg_shared = on("TANGO_NO_CFG_SHARED");
  x.stream_spec = num("TANGO_STREAM_SPEC", 1) != 0;
  x.attn_qb2_min_wgs = num("TANGO_ATTN_QB2_MIN_WGS", 512);
  x.gn_small_mb = num("TANGO_GN_SMALL_MB", 8);
  x.gn_slab = num("TANGO_GN_SLAB", 1);
  x.no_halo_narrow = on("TANGO_NO_HALO_NARROW");
  x.wide_pers = num("TANGO_WIDE_PERS", 2);
  x.attn_msum = num("TANGO_ATTN_MSUM", 1) != 0;
  x.duo_maxk = num("TANGO_DUO_MAXK", -1);
  x.duo_min_tiles = num("TANGO_DUO_MIN_TILES", 384);
  x.duo_mask = num("TANGO_DUO_MASK", 7);
  x.duo_prio = num("TANGO_DUO_PRIO", 0);
  x.wide_pipe = num("TANGO_WIDE_PIPE", 0);
  x.conv_pipe = num("TANGO_CONV_PIPE", 2);
  x.conv_tall = num("TANGO_CONV_TALL", 0);
  x.attn_x8_qb = num("TANGO_ATTN_X8_QB", 2);
  x.attn_defer = num("TANGO_ATTN_DEFER", 0);
  x.attn_kdma = num("TANGO_ATTN_KDMA", 1);
  x.attn_vdma = num("TANGO_ATTN_VDMA", 2);
  x.gn_fold = on("TANGO_GN_FOLD");
  x.ff_fused = num("TANGO_FF_FUSED", 1);
  x.qkv_stat = num("TANGO_QKV_STAT", 1);
  x.gn_proj_stat = num("TANGO_GN_PROJ_STAT", 1);
  x.qkv_min_rows = num("TANGO_QKV_MIN_ROWS", 65536);
  x.ff_min_rows = num("TANGO_FF_MIN_ROWS", 32768);
  const char* wp = getenv("TANGO_WIDE_PRIO");
  x.wide_prio = (wp && wp[0] >= '0' && wp[0] <= '2') ? wp[0] - '0' : 0;
  return x;
}

inline Tuning& tuning_storage() {
  static Tuning t = read_tuning();
  return t;
}
inline const Tuning& tuning() { return tuning_storage(); }
// measurement tools only (tango_tuning_reload): re-read the environment between A/B arms inside ONE process.  Switches that
// plans bake in at build time (routing) only take effect for plans built afterwards; launch-time switches (wide_sched) at once.
inline void tuning_reload() { tuning_storage() = read_tuning(); }

}  // namespace tango
