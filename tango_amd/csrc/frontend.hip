// Wave -> log-mel front-end on the engine: TacotronSTFT.mel_spectrogram of the reference
// (audioldm/audio/stft.py:164-186) = STFT.transform (stft.py:52-85: reflect pad n_fft/2, conv1d with the windowed DFT basis at
// stride hop, magnitude) -> mel_basis @ magnitude -> log(clamp(., 1e-5)) (audio_processing.py:84-92), plus the log-magnitudes
// and the per-frame energy it returns next to the mel.  Callers: tools/torch_tools.py:57-78 (wav_to_fbank) from train.py /
// the mel side of AutoencoderKL.encode_first_stage.
//
// Always fp32 (f32 MFMA, an exact fmaf chain): a spectrogram spans 6+ decades and everything after it is a logarithm, so
// 16-bit operands would put rounding noise of the loud partials into every quiet bin; the whole front-end is ~2 GFLOP per
// 10-s clip.  Both matrix products go through the gather-GEMM:
//   frames:  Z[b][t][:] = W_dft[2*cutoff][n_fft] . xpad[b][t*hop : t*hop + n_fft]      -- a "linear" whose row stride (hop) is
//            SHORTER than its row length (overlapping rows: no im2col copy of the 6.4x-overlapped frames), batched over b
//   mel:     mel_lin[r][:] = mel_basis[n_mel][cutoff (zero-padded to 16)] . mag[r][:]
// and three small HBM-bound kernels do the rest (reflect padding; magnitude + energy; log-clamp + [B*T, C] -> [B, C, T]
// transposes through LDS so that both the reads and the writes are row-contiguous).
//
// Weights are the BUFFERS of the reference module (pytorch_model_stft.bin, tango.py:19-27): `mel_basis` [n_mel, cutoff] and
// `stft_fn.forward_basis` [2*cutoff, 1, n_fft]; `stft_fn.inverse_basis` is not on this path.
#include "engine.h"

namespace tango {

__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ xp, int N, int P, int Np) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= Np) return;
  float v = 0.f;
  if (i < N + 2 * P) {                       // F.pad(..., mode="reflect"): edge sample not repeated
    int j = i - P;
    if (j < 0) j = -j;
    if (j >= N) j = 2 * N - 2 - j;
    v = x[(int64_t)b * N + j];
  }
  xp[(int64_t)b * Np + i] = v;
}

// one workgroup per frame: Z[r] = [re(0..cutoff) | im(0..cutoff)] -> mag[r][0..Kp2) (zero beyond cutoff), energy[r] = ||mag||_2
__global__ __launch_bounds__(256) void stft_mag_kernel(const float* __restrict__ Z, int64_t ldz, float* __restrict__ mag, int Kp2,
                                                       float* __restrict__ energy, int cutoff) {
  __shared__ float red[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* z = Z + (int64_t)r * ldz;
  float e = 0.f;
  for (int f = tid; f < Kp2; f += 256) {
    float v = 0.f;
    if (f < cutoff) {
      const float re = z[f], im = z[cutoff + f];
      const float s = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));      // real**2 + imag**2 as two rounded products (stft.py:81)
      v = __fsqrt_rn(s);
      e = __fadd_rn(e, s);          // ||mag||^2 = sum re^2 + im^2 (torch.norm squares sqrt(s) again: differs in the last ulp only)
    }
    mag[(int64_t)r * Kp2 + f] = v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
  if ((tid & 63) == 0) red[tid >> 6] = e;
  __syncthreads();
  if (tid == 0) energy[r] = __fsqrt_rn(red[0] + red[1] + red[2] + red[3]);
}

// out[b][c][t] = log(max(in[(b*T + t)*ld + c], clip)) for c < C: 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void log_clamp_transpose_kernel(const float* __restrict__ in, int64_t ld, float* __restrict__ out,
                                                                  int T, int C, float clip) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + i * 8, c = c0 + tx;
    tile[ty + i * 8][tx] = (t < T && c < C) ? in[((int64_t)b * T + t) * ld + c] : 1.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + i * 8, t = t0 + tx;
    if (c < C && t < T) out[((int64_t)b * C + c) * T + t] = logf(fmaxf(tile[tx][ty + i * 8], clip));
  }
}

// ------------------------------------------------------------------------------------------------
void Engine::build_stft_weights() {
  const int nfft = cfg.stft_filter_length, cutoff = nfft / 2 + 1, nmel = cfg.stft_n_mel;
  const int Kp2 = (cutoff + 15) / 16 * 16;
  // windowed DFT basis, rows [Re(0..cutoff) ; Im(0..cutoff)]
  stft_basis.N = 2 * cutoff; stft_basis.K = nfft; stft_basis.Cin = nfft; stft_basis.taps = 1; stft_basis.Kp = nfft;
  stft_basis.W = dmalloc((size_t)stft_basis.N * nfft * 4);
  {
    void* W = stft_basis.W; const int N = stft_basis.N;
    reg_slot("stft_fn.forward_basis", {N, 1, nfft},
             [=](const float* src, hipStream_t s) { return launch_pack(DT_F32, src, W, N, 1, nfft, nfft, 0, 1, nfft, 0, s); });
  }
  // mel filterbank, K zero-padded to a multiple of 16 floats (64-byte k-chunks)
  stft_mel.N = nmel; stft_mel.K = Kp2; stft_mel.Cin = Kp2; stft_mel.taps = 1; stft_mel.Kp = Kp2;
  stft_mel.W = dmalloc((size_t)nmel * Kp2 * 4);
  {
    void* W = stft_mel.W;
    reg_slot("mel_basis", {nmel, cutoff},
             [=](const float* src, hipStream_t s) { return launch_pack(DT_F32, src, W, nmel, 1, cutoff, cutoff, 0, 1, Kp2, 0, s); });
  }
}

int Engine::get_stft_plan(int B, int N, StftPlan** out) {
  auto key = std::make_pair(B, N);
  auto it = stft_plans.find(key);
  if (it != stft_plans.end()) { touch(it->second->meta); *out = it->second.get(); return 0; }
  if (!finalized) TANGO_FAIL("engine: weights not finalized");
  const int nfft = cfg.stft_filter_length, hop = cfg.stft_hop_length, cutoff = nfft / 2 + 1, nmel = cfg.stft_n_mel;
  const int P = nfft / 2;
  if (N <= P) TANGO_FAIL("mel_spectrogram: reflect padding needs more than n_fft / 2 samples (F.pad raises the same way)");
  std::unique_ptr<StftPlan> Pn(new StftPlan());
  StftPlan& S = *Pn;
  S.B = B; S.N = N; S.T = 1 + N / hop;
  S.Np = (N + 2 * P + 3) / 4 * 4;
  S.Kp2 = (cutoff + 15) / 16 * 16;
  S.ldz = (2 * cutoff + 15) / 16 * 16;
  Arena a;
  auto carve = [&](Arena& A) {
    S.in = (float*)A.alloc((size_t)B * N * 4);
    S.xpad = (float*)A.alloc((size_t)B * S.Np * 4 + 64);
    S.Z = (float*)A.alloc((size_t)B * S.T * S.ldz * 4);
    S.mag = (float*)A.alloc((size_t)B * S.T * S.Kp2 * 4);
    S.mel_lin = (float*)A.alloc((size_t)B * S.T * nmel * 4);
    S.mel = (float*)A.alloc((size_t)B * nmel * S.T * 4);
    S.logmag = (float*)A.alloc((size_t)B * cutoff * S.T * 4);
    S.energy = (float*)A.alloc((size_t)B * S.T * 4);
  };
  carve(a);
  TANGO_TRY(alloc_slab(&S.slab, a.peak + 256, S.meta, false));
  Arena r; r.base = S.slab;
  carve(r);
  *out = Pn.get();
  stft_plans[key] = std::move(Pn);
  return 0;
}

int Engine::mel_spectrogram(const float* wav, float* mel, float* logmag, float* energy, int B, int N, int* n_frames, hipStream_t s) {
  if (cfg.stft_filter_length <= 0) TANGO_FAIL("engine: STFT front-end not configured (tango_config.stft_filter_length)");
  StftPlan* Pp;
  TANGO_TRY(get_stft_plan(B, N, &Pp));
  StftPlan& S = *Pp;
  const int nfft = cfg.stft_filter_length, hop = cfg.stft_hop_length, cutoff = nfft / 2 + 1, nmel = cfg.stft_n_mel;
  if (n_frames) *n_frames = S.T;
  TANGO_HIP(hipMemcpyAsync(S.in, wav, (size_t)B * N * 4, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)((S.Np + 255) / 256), (unsigned)B), dim3(256), 0, s, S.in, S.xpad, N, nfft / 2, S.Np);
  TANGO_HIP(hipGetLastError());
  {  // frames x DFT basis: rows overlap (lda = hop < K = n_fft), one batch item per waveform
    GemmParams p;
    p.A = S.xpad; p.lda = hop; p.W = stft_basis.W; p.Kp = nfft;
    p.M = S.T; p.N = 2 * cutoff; p.K = nfft; p.Cin = nfft;
    p.mode = GATHER_1D; p.rows_pb = S.T; p.Lin = S.T; p.Lout = S.T; p.taps = 1;
    p.out = S.Z; p.ldo = S.ldz;
    p.batch = B; p.sA = S.Np; p.sW = 0; p.sO = (int64_t)S.T * S.ldz;
    TANGO_TRY(launch_gemm(DT_F32, p, s));
  }
  hipLaunchKernelGGL(stft_mag_kernel, dim3((unsigned)(B * S.T)), dim3(256), 0, s, S.Z, (int64_t)S.ldz, S.mag, S.Kp2, S.energy, cutoff);
  TANGO_HIP(hipGetLastError());
  {
    GemmParams p;
    p.A = S.mag; p.lda = S.Kp2; p.W = stft_mel.W; p.Kp = S.Kp2;
    p.M = B * S.T; p.N = nmel; p.K = S.Kp2; p.Cin = S.Kp2;
    p.mode = GATHER_1D; p.rows_pb = p.M; p.Lin = p.M; p.Lout = p.M; p.taps = 1;
    p.out = S.mel_lin; p.ldo = nmel;
    TANGO_TRY(launch_gemm(DT_F32, p, s));
  }
  const float clip = 1e-5f;          // dynamic_range_compression(clip_val = 1e-5, C = 1), audio_processing.py:84-92
  hipLaunchKernelGGL(log_clamp_transpose_kernel, dim3((unsigned)((S.T + 31) / 32), (unsigned)((nmel + 31) / 32), (unsigned)B), dim3(256), 0, s,
                     S.mel_lin, (int64_t)nmel, S.mel, S.T, nmel, clip);
  TANGO_HIP(hipGetLastError());
  TANGO_HIP(hipMemcpyAsync(mel, S.mel, (size_t)B * nmel * S.T * 4, hipMemcpyDeviceToDevice, s));
  if (logmag) {
    hipLaunchKernelGGL(log_clamp_transpose_kernel, dim3((unsigned)((S.T + 31) / 32), (unsigned)((cutoff + 31) / 32), (unsigned)B), dim3(256), 0, s,
                       S.mag, (int64_t)S.Kp2, S.logmag, S.T, cutoff, clip);
    TANGO_HIP(hipGetLastError());
    TANGO_HIP(hipMemcpyAsync(logmag, S.logmag, (size_t)B * cutoff * S.T * 4, hipMemcpyDeviceToDevice, s));
  }
  if (energy) TANGO_HIP(hipMemcpyAsync(energy, S.energy, (size_t)B * S.T * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}

}  // namespace tango

extern "C" {

int tango_engine_mel_frames(tango_engine_t* h, int n_samples) {
  const int hop = h->e->cfg.stft_hop_length;
  return hop > 0 ? 1 + n_samples / hop : 0;
}

int tango_engine_mel_spectrogram(tango_engine_t* h, const float* wav, float* mel, float* log_magnitudes, float* energy, int batch,
                                 int n_samples, int* n_frames, void* stream) {
  return h->e->mel_spectrogram(wav, mel, log_magnitudes, energy, batch, n_samples, n_frames, (hipStream_t)stream);
}

}  // extern "C"
