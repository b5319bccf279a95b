// Internal engine classes (host side).  The public surface is include/tango_engine.h.
#pragma once
#include <array>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tango_engine.h"
#include "common.h"

namespace tango {

using Op = std::function<int(hipStream_t)>;

struct Program {
  std::vector<Op> ops;
  std::vector<std::string> labels;   // parallel to ops (per-op profiling)
  std::vector<double> flops;
  int run(hipStream_t s) const {
    for (const auto& o : ops) TANGO_TRY(o(s));
    return 0;
  }
  // eager run with HIP events around every op; appends "label<TAB>ms<TAB>gflop" lines to `report`
  int run_profiled(hipStream_t s, std::string& report) const;
};

// bump allocator over one hipMalloc'd slab; build is run twice (measure, then real)
struct Arena {
  char* base = nullptr;
  size_t off = 0, peak = 0;
  void* alloc(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    void* p = base ? (void*)(base + off) : (void*)(uintptr_t)(0x1000 + off);
    off += bytes;
    if (off > peak) peak = off;
    return p;
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
};

struct TView {
  void* p = nullptr;
  int64_t ld = 0;
  int C = 0;
};

struct WMat {            // packed GEMM weight [N][Kp] (engine dtype) + fp32 bias
  void* W = nullptr;
  float* b = nullptr;
  int N = 0, K = 0, Cin = 0, taps = 1;
  int64_t Kp = 0;
  bool im2col = false;   // 3x3 conv with tiny Cin: K = 9*Cin zero-padded to Kp, A comes from im2col
  // LayerNorm-folded copy (linear_stream.hip): W' = W*gamma, b' = b + W.beta, wsum[n] = sum_k W'[n][k]
  void* Wln = nullptr;
  float* bln = nullptr;
  float* wsum = nullptr;
};
struct WNorm {
  float* g = nullptr;
  float* b = nullptr;
  int C = 0;
  float eps = 1e-5f;
};
struct WLinF32 {         // fp32 [N][K] (time-embedding path stays fp32)
  float* W = nullptr;
  float* b = nullptr;
  int N = 0, K = 0;
};

struct ResW {
  int cin = 0, cout = 0;
  WNorm n1, n2;
  WMat c1, c2, sc;
  bool has_sc = false;
  bool has_temb = false;
  WLinF32 temb;
  float* temb_table = nullptr;  // [max_steps][cout] fp32, filled per denoise call
};
struct XfW {             // Transformer2DModel with one BasicTransformerBlock
  int C = 0, heads = 0;
  int cond = 0;          // which condition its cross-attention reads: 0 text, 1 beat, 2 chord (Music UNet: attentions / attentions2 / attentions3)
  WNorm gn, ln1, ln2, ln3;
  WMat proj_in, qkv, o1, q2, kv2, o2, ff1, ff2, proj_out;
  // fused cross-attention block (xattn.hip; C = 320 on 16-bit engines): LayerNorm-folded to_q with rows permuted per head
  void* q2p = nullptr;
  float* bq2p = nullptr;
  float* wsum2p = nullptr;
};
struct VaeAttnW {
  WNorm gn;
  WMat qk, v, proj;
};
struct ConvTW {          // ConvTranspose1d decomposed into `stride` phase matrices
  int cin = 0, cout = 0, k = 0, u = 0, pad = 0;
  std::vector<WMat> phase;
  float* b = nullptr;
};
struct VocResW {
  int ch = 0, k = 0;
  std::vector<int> dil;
  std::vector<WMat> c1, c2;
};

struct Slot {
  std::string name;
  std::vector<int64_t> shape;
  bool set = false;
  std::function<int(const float*, hipStream_t)> pack;
};

// every plan (a workspace slab + its launch program, keyed by the call's shape) carries its size and a last-use stamp: the plan
// caches of an engine share ONE byte budget, least recently used plans are freed first (Engine::make_room; round 4 -- callers of
// tango.py:51-64 with ragged last batches and mixed prompt lengths used to accumulate ~0.44 GB per prompt without bound)
struct PlanMeta {
  size_t bytes = 0;
  uint64_t stamp = 0;
};

struct UNetPlan {
  PlanMeta meta;
  int B2 = 0, L = 0;
  int Lc[3] = {0, 0, 0};    // condition lengths: [0] text (== L), [1] beat, [2] chord (Music UNet only)
  char* slab = nullptr;
  Program pre, step;
  void* xin = nullptr;      // T  [B2*HW][8]
  float* eps = nullptr;     // f32 [B2*HW][out_ch]
  void* enc = nullptr;      // T  [B2*L][cross]       (== encs[0])
  float* bias = nullptr;    // f32 [B2*L]             (== biases[0])
  void* encs[3] = {nullptr, nullptr, nullptr};
  float* biases[3] = {nullptr, nullptr, nullptr};
  // single-key prefix (round 4): the first n_short samples attend to exactly ONE text key (the unconditional rows of a CFG batch are
  // T5("") = one valid token), so their text cross-attention output is the constant to_out(v_key) + b, independent of the query
  int n_short = 0;
  int* key0 = nullptr;      // i32 [B2]: index of that key per sample (entries >= n_short unused)
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  // k denoise steps per replay (round 5): the same step captured k_steps times back to back
  hipGraph_t graph_k = nullptr;
  hipGraphExec_t exec_k = nullptr;
  int k_steps = 0;
  // Two independent chains (round 5; Engine::denoise): a "dual" plan owns only the UNet input / output buffers of the whole batch; its
  // two children run the first and the second half of the samples (no op of the UNet crosses samples) as two branches of the captured
  // graph, so that one branch's dispatch gaps and kernel tails are filled by the other's kernels.  A child takes xin / eps from its
  // parent (ext_*), owns its slab and program, and is never in the plan map on its own (the parent's eviction frees all three slabs).
  std::unique_ptr<UNetPlan> child[2];
  void* ext_xin = nullptr;
  float* ext_eps = nullptr;
  bool cfg_shared = false;     // mode 3: the part of the UNet in front of the first cross-attention runs on B2 / 2 samples (build_unet)
  unsigned* sync = nullptr;    // barrier words its cooperative kernels use (one set per chain: two chains may run such a kernel at once)
};
struct T5LayerW {          // T5Block of the encoder: self-attention + gated-GELU feed-forward (no biases anywhere)
  WNorm ln1, ln2;
  WMat qkv, o, wi, wo;     // qkv = fused [q; k; v] rows; wi = [wi_1 | wi_0] in the GLU interleave (value | gate)
};
struct T5Plan {
  PlanMeta meta;
  int B = 0, L = 0;
  char* slab = nullptr;
  Program prog;
  int64_t* ids = nullptr;   // [B*L]
  float* bias = nullptr;    // [B*L] additive key mask
  int* bucket = nullptr;    // [L*L] relative-position buckets (host-computed)
  float* pos_bias = nullptr;  // [heads][L][L]
  float* out = nullptr;     // [B*L][d_model] fp32
};
struct VaePlan {           // also used for the vocoder: plan-owned in/out staging buffers
  PlanMeta meta;
  int B = 0;
  char* slab = nullptr;
  Program prog;
  void* in = nullptr;
  void* out = nullptr;
  size_t in_bytes = 0, out_bytes = 0;
  int n_out = 0;            // vocoder: samples per item
};

struct StftPlan {          // wave -> log-mel front-end buffers for one (batch, n_samples)
  PlanMeta meta;
  int B = 0, N = 0, T = 0, Np = 0, Kp2 = 0, ldz = 0;
  char* slab = nullptr;
  float *in = nullptr, *xpad = nullptr, *Z = nullptr, *mag = nullptr, *mel_lin = nullptr, *mel = nullptr, *logmag = nullptr, *energy = nullptr;
};

class Engine {
 public:
  explicit Engine(const tango_config_t& c);
  ~Engine();
  int init();

  int set_weight(const char* name, const float* dev, const int64_t* shape, int ndim);
  int finalize_weights();
  int denoise(const tango_denoise_args_t& a, hipStream_t s);
  struct Cond { const float* emb = nullptr; const uint8_t* mask = nullptr; int len = 0; const uint8_t* mask_host = nullptr; };
  int unet_forward(const float* sample, int64_t t, const Cond (&c)[3], float* out, int B2, hipStream_t s);
  int vae_decode(const float* lat, float* mel, int B, hipStream_t s);
  int vae_encode(const float* mel, float* moments, int B, hipStream_t s);
  int vocode(const float* mel, int16_t* wav, int B, int frames, int* n_samples, hipStream_t s);
  int vocoder_samples(int frames) const;
  int encode_text(const int64_t* ids, const uint8_t* mask, float* out, int B, int L, hipStream_t s);
  int mel_spectrogram(const float* wav, float* mel, float* logmag, float* energy, int B, int N, int* n_frames, hipStream_t s);
  int last_denoise_ms(float* total_ms, float* per_step_ms);
  int profile_unet(int B2, int L, std::string& report, hipStream_t s);
  int profile_vae(int B, std::string& report, hipStream_t s);                  // per-op timing of the mel-VAE decoder plan (round 6)
  int profile_vocoder(int B, int frames, std::string& report, hipStream_t s);  // ... of the HiFi-GAN plan
  // plan-cache budget (bytes of workspace slabs kept alive; default 64 GiB or TANGO_PLAN_BUDGET_MB) and its current use
  void set_plan_budget(size_t bytes) { plan_budget = bytes; }
  void drop_plans() { const size_t b = plan_budget; plan_budget = 0; (void)make_room(1); plan_budget = b; }   // frees every cached plan
  size_t plan_bytes_in_use() const { return plan_bytes; }
  int plan_count() const;

  tango_config_t cfg;
  int dt = DT_F32;
  size_t esz = 4;
  std::vector<Slot> slots;
  std::unordered_map<std::string, int> slot_index;
  bool finalized = false;
  float last_total_ms = 0.f, last_step_ms = 0.f;
  double last_step_gflop = 0.0;  // executed GFLOP of one launch of the last denoise call's step program
  uint64_t coop_fallbacks = 0;   // workgroups of gn_coop_kernel that took the no-rendezvous fallback (diagnostic; results are unaffected)

 private:
  friend struct Builder;
  // ---- memory ----
  std::vector<void*> owned;
  void* dmalloc(size_t bytes);
  // ---- weight registration ----
  void reg_slot(const std::string& name, std::vector<int64_t> shape, std::function<int(const float*, hipStream_t)> pack);
  void reg_vec(const std::string& name, int n, float** dst);
  void reg_norm(const std::string& p, int C, float eps, WNorm& w);
  void reg_mat(const std::string& wname, int N, int K, WMat& w, bool alloc, int row_off, std::vector<int64_t> shape, bool geglu = false);
  void reg_linear(const std::string& p, int N, int K, WMat& w, bool bias);
  void reg_conv3x3(const std::string& p, int Cout, int Cin, WMat& w);
  void reg_conv1x1(const std::string& p, int Cout, int Cin, WMat& w);
  void reg_conv1d(const std::string& p, int Cout, int Cin, int k, WMat& w);
  void reg_convt1d(const std::string& p, int Cin, int Cout, int k, int u, ConvTW& w);
  void reg_linear_f32(const std::string& p, int N, int K, WLinF32& w);
  void reg_res(const std::string& p, int cin, int cout, int temb, float eps, ResW& w, bool vae);
  void reg_xf(const std::string& p, int C, int heads, int cross, XfW& w);
  int fold_ln(WMat& w, const WNorm& ln);
  void build_unet_weights();
  void build_vae_weights();
  void build_vae_enc_weights();
  void reg_vae_attn(const std::string& p, int C, VaeAttnW& w);
  void build_voc_weights();
  void build_t5_weights();
  void build_stft_weights();       // frontend.hip

  // ---- model weights ----
  // UNet
  WMat conv_in, conv_out;
  WNorm norm_out;
  WLinF32 time1, time2;
  // xf[j] is the site's text transformer; xf2 / xf3 are `attentions2` / `attentions3` of the Music blocks (empty otherwise)
  struct DownBlock { std::vector<ResW> res; std::vector<XfW> xf, xf2, xf3; bool has_ds = false; WMat ds; };
  struct UpBlock { std::vector<ResW> res; std::vector<XfW> xf, xf2, xf3; bool has_us = false; WMat us; };
  std::vector<DownBlock> down;
  std::vector<UpBlock> up;
  ResW mid_res0, mid_res1;
  XfW mid_xf, mid_xf2, mid_xf3;
  std::vector<ResW*> all_res;   // every UNet resblock (temb tables)
  std::vector<XfW*> all_xf;
  // VAE
  float* pqc_w = nullptr; float* pqc_b = nullptr;
  WMat vae_conv_in, vae_conv_out;
  ResW vae_mid1, vae_mid2;
  VaeAttnW vae_attn;
  struct VaeUp { std::vector<ResW> res; bool has_up = false; WMat up; };
  std::vector<VaeUp> vae_up;    // indexed by level
  WNorm vae_norm_out;
  // encoder (modules.py:419-543) + quant_conv
  struct VaeDown { std::vector<ResW> res; bool has_down = false; WMat down; };
  std::vector<VaeDown> vae_down;
  WMat vae_enc_conv_in, vae_enc_conv_out;
  ResW vae_enc_mid1, vae_enc_mid2;
  VaeAttnW vae_enc_attn;
  WNorm vae_enc_norm_out;
  float* qc_w = nullptr; float* qc_b = nullptr;
  // vocoder
  WMat voc_pre, voc_post;
  std::vector<ConvTW> voc_ups;
  std::vector<VocResW> voc_res;

  // wave -> log-mel front-end (frontend.hip)
  WMat stft_basis, stft_mel;
  std::map<std::pair<int, int>, std::unique_ptr<StftPlan>> stft_plans;
  int get_stft_plan(int B, int N, StftPlan** out);

  // text encoder
  float* t5_embed = nullptr;        // [vocab][d_model] fp32
  float* t5_rel_table = nullptr;    // [rel_buckets][heads] fp32
  std::vector<T5LayerW> t5_layers;
  WNorm t5_final_ln;
  std::map<std::pair<int, int>, std::unique_ptr<T5Plan>> t5_plans;
  int get_t5_plan(int B, int L, T5Plan** out);
  int build_t5(T5Plan& P, Arena& A, bool record);

  // ---- runtime state ----
  int* d_step = nullptr;
  unsigned* d_sync = nullptr;   // barrier words of the cooperative kernels (common.h: coop_sync_words()), zeroed once at init
  SchedParams* d_sched = nullptr;   // device copy of the current call's scheduler parameter block
  int64_t* d_ts = nullptr;       // [max_steps]
  float* d_coef = nullptr;       // [max_steps][8]
  float* d_sin = nullptr;        // [max_steps][ch0]
  float* d_t1 = nullptr;         // [max_steps][temb]
  float* d_temb = nullptr;       // [max_steps][temb]  silu(emb)
  std::vector<int64_t> temb_ts;  // cache key
  int max_steps = 1000;
  std::map<std::array<int, 6>, std::unique_ptr<UNetPlan>> unet_plans;   // key: (B2, L_text, L_beat, L_chord, n_short, chains)
  std::map<int, std::unique_ptr<VaePlan>> vae_plans;
  std::map<int, std::unique_ptr<VaePlan>> vae_enc_plans;
  std::map<std::pair<int, int>, std::unique_ptr<VaePlan>> voc_plans;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t cap_stream = nullptr;
  // second chain of a dual plan: its capture stream, its eager stream, fork / join events (capture and eager pairs), barrier words
  hipStream_t cap_stream2 = nullptr, aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork_e = nullptr, ev_join_e = nullptr;
  unsigned* d_sync2 = nullptr;
  int last_steps = 0;

  // ---- host -> device staging without a host sync (round 4) ----
  // Small per-call tables (timesteps, scheduler coefficients, the SchedParams block, single-key indices) come from pageable /
  // transient host memory.  They are copied into engine-owned PINNED slots and uploaded with hipMemcpyAsync; a slot is reused
  // only after the event recorded behind its upload has fired, so neither the caller's buffer nor the slot can be overwritten
  // under a pending copy and `denoise` returns without waiting for the stream.
  struct HostSlot { void* buf = nullptr; hipEvent_t ev = nullptr; bool pending = false; };
  static constexpr int kHostSlots = 8;
  static constexpr size_t kHostSlotBytes = 64 << 10;
  HostSlot hslots[kHostSlots];
  int hslot_next = 0;
  int stage_h2d(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s);

  // ---- plan caches: shared LRU byte budget ----
  size_t plan_budget = (size_t)64 << 30, plan_bytes = 0;
  uint64_t plan_clock = 0;
  void touch(PlanMeta& m) { m.stamp = ++plan_clock; }
  int make_room(size_t need);             // frees least-recently-used plans until plan_bytes + need <= plan_budget (or nothing is left)
  bool evict_lru();
  int alloc_slab(char** slab, size_t bytes, PlanMeta& m, bool zero);   // on out-of-memory: evicts LRU plans and retries
  void release_slab(char** slab, PlanMeta& m);

  int ensure_temb(const int64_t* ts_host, int n, hipStream_t s);
  int get_unet_plan(int B2, int L, int Lbeat, int Lchord, int n_short, UNetPlan** out, int chains = 1);
  int make_unet_plan(UNetPlan& P);            // measure, allocate the slab, build the programs
  void free_unet_plan(UNetPlan& P);           // graphs, slab(s), children; gives the bytes back to the budget
  bool cfg_shared_ok(int B2, int n_short) const;
  int unet_chains_for(int B2) const;          // 1 or 2: how denoise() runs a batch of B2 UNet rows
  int single_key_prefix(const uint8_t* mask_dev, const uint8_t* mask_host, int B2, int L, std::vector<int>& key0, hipStream_t s);
  int build_unet(UNetPlan& P, Arena& A, bool record);
  int get_vae_plan(int B, VaePlan** out);
  int build_vae(VaePlan& P, Arena& A, bool record);
  int get_vae_enc_plan(int B, VaePlan** out);
  int build_vae_enc(VaePlan& P, Arena& A, bool record);
  int get_voc_plan(int B, int frames, VaePlan** out);
  int build_voc(VaePlan& P, Arena& A, bool record, int frames);
  int bind_text(UNetPlan& P, const Cond (&c)[3], const std::vector<int>& key0, hipStream_t s);
};

}  // namespace tango

// the opaque handle of include/tango_engine.h
struct tango_engine { tango::Engine* e; };
