// taco_audio.h -- spectrogram -> waveform on the GPU (audio/__init__.py:54-56,76-96,118-122,149-165;
// synthesizer.py:264 `inv_spectrogram(wav.T)`): denormalise, dB -> amplitude, ^power, Griffin-Lim, inverse pre-emphasis.
// librosa's stft/istft are restated as what they are for this window: a hop-strided gather of win_length samples, one
// windowed-DFT matrix product per direction (on the matrix cores through the same implicit-GEMM kernels as the model,
// 3-term split-bf16), overlap-add with the window sum-square normalisation, reflect padding.  Included from taco_lib.hip.
#pragma once

struct taco_gl {
  taco_audio_hparams hp;
  int n_fft = 0, hop = 0, win = 0, lpad = 0, F = 0;
  taco_model* gm = nullptr;      // container for the two DFT weight packs (GemmVar table + arena)
  ConvL fwd, inv;                // [win -> 2F] analysis, [2F -> win] synthesis (window folded into both)
  size_t w2 = 0;                 // squared padded window [n_fft] (arena offset)
};

// ---- kernels ----
// S = (10^((clip(x,0,1) * -min_db + min_db + ref_db) / 20))^power ; rows t >= T of every utterance slot are zero
__global__ void k_gl_magnitude(const float* spec, float* S, int B, int T, int Tr, int F, float min_db, float ref_db, float power) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * Tr * F) return;
  const int f = (int)(i % F); const size_t row = i / F; const int t = (int)(row % Tr), b = (int)(row / Tr);
  float v = 0.f;
  if (t < T) {
    const float x = fminf(fmaxf(spec[((size_t)b * T + t) * F + f], 0.f), 1.f);
    const float db = x * -min_db + min_db + ref_db;
    v = powf(powf(10.f, db * 0.05f), power);
  }
  S[i] = v;
}
__device__ __forceinline__ float gl_hash_uniform(unsigned long long seed, size_t i) {   // counter-based: splitmix64 -> [0,1)
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
// X = S * exp(2 pi i u): u from the caller ([B,T,F], np.random.rand of audio/__init__.py:77) or from the counter hash
__global__ void k_gl_init_phase(const float* S, const float* u, unsigned long long seed, float* X, int B, int T, int Tr, int F) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * Tr * F) return;
  const int f = (int)(i % F); const size_t row = i / F; const int t = (int)(row % Tr), b = (int)(row / Tr);
  float re = 0.f, im = 0.f;
  if (t < T) {
    const float uu = u ? u[((size_t)b * T + t) * F + f] : gl_hash_uniform(seed, i);
    float sn, cs; sincosf(6.283185307179586f * uu, &sn, &cs);
    re = S[i] * cs; im = S[i] * sn;
  }
  X[row * 2 * F + f] = re; X[row * 2 * F + F + f] = im;
}
// angles = exp(i * angle(est)) (np.angle(0) = 0); X = S * angles
__global__ void k_gl_project(const float* est, const float* S, float* X, size_t rows, int F) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * F) return;
  const size_t row = i / F; const int f = (int)(i % F);
  const float re = est[row * 2 * F + f], im = est[row * 2 * F + F + f];
  const float mag = sqrtf(re * re + im * im), s = S[i];
  X[row * 2 * F + f] = mag > 0.f ? s * re / mag : s;
  X[row * 2 * F + F + f] = mag > 0.f ? s * im / mag : 0.f;
}
// 1 / window-sum-square where it exceeds tiny, else 1 (librosa istft), in padded coordinates [hop*(T-1) + n_fft]
__global__ void k_gl_wss(const float* w2, float* inv, int T, int n_fft, int hop) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int Lp = hop * (T - 1) + n_fft;
  if (p >= Lp) return;
  float s = 0.f;
  const int t1 = min(T - 1, p / hop);
  for (int t = t1; t >= 0 && p - t * hop < n_fft; --t) s += w2[p - t * hop];
  inv[p] = s > 1.17549435e-38f ? 1.0f / s : 1.0f;
}
// overlap-add of the windowed frames Y [B, Tr, win] -> centre part of ypad [B, slot]; then reflect padding of n_fft/2 on both sides
__global__ void k_gl_overlap_add(const float* Y, const float* wss_inv, float* ypad, int B, int T, int Tr, int win, int hop, int lpad,
                                 int n_fft, size_t slot) {
  const int L = hop * (T - 1);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * L) return;
  const int b = (int)(i / L), s = (int)(i % L), p = s + n_fft / 2;
  float acc = 0.f;
  const int t1 = min(T - 1, (p - lpad) / hop);
  for (int t = t1; t >= 0; --t) {
    const int off = p - t * hop - lpad;
    if (off >= win) break;
    acc += Y[((size_t)b * Tr + t) * win + off];
  }
  ypad[(size_t)b * slot + p] = acc * wss_inv[p];
}
__global__ void k_gl_reflect(float* ypad, int B, int T, int hop, int n_fft, size_t slot) {
  const int half = n_fft / 2, L = hop * (T - 1);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i % half + 1;            // k = 1 .. n_fft/2
  float* y = ypad + (size_t)b * slot + half;             // y[0 .. L)
  y[-k] = y[k];
  y[L - 1 + k] = y[L - 1 - k];
}
// scipy.signal.lfilter([1], [1, -a], x): out[n] = x[n] + a*out[n-1]; one workgroup per utterance, chunked scan
__global__ __launch_bounds__(1024) void k_inv_preemphasis(const float* ypad, float* wav, int L, int half, size_t slot, float a) {
  __shared__ float ends[1024];
  __shared__ float carry[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* x = ypad + (size_t)b * slot + half;
  float* out = wav + (size_t)b * L;
  const int C = (L + 1023) / 1024, i0 = tid * C, i1 = min(L, i0 + C);
  float acc = 0.f;
  for (int i = i0; i < i1; ++i) { acc = x[i] + a * acc; out[i] = acc; }
  ends[tid] = acc;
  __syncthreads();
  if (tid == 0) {
    const float aC = powf(a, (float)C);
    float c = 0.f;
    for (int k = 0; k < 1024; ++k) { carry[k] = c; c = ends[k] + aC * c; }   // carry[k] = out[k*C - 1]
  }
  __syncthreads();
  const float c = carry[tid];
  float f = a;
  for (int i = i0; i < i1; ++i) { out[i] += f * c; f *= a; }
}

// ---- host ----
static int gl_rows(const taco_gl* g, int T) { return T + cdiv(g->n_fft, g->hop); }                 // frames per utterance slot (tail frames read the slack)
static size_t gl_slot(const taco_gl* g, int T) { return (size_t)gl_rows(g, T) * g->hop; }          // samples per utterance slot >= hop*(T-1) + n_fft
struct GlWs { float *S, *X, *est, *Y, *ypad, *wss; };
static void carve_gl(Carver& cv, const taco_gl* g, int B, int T, GlWs& w) {
  const size_t R = (size_t)B * gl_rows(g, T);
  w.S = cv.f(R * g->F); w.X = cv.f(R * 2 * g->F); w.est = cv.f(R * 2 * g->F); w.Y = cv.f(R * g->win);
  w.ypad = cv.f((size_t)B * gl_slot(g, T) + 2 * g->n_fft + g->win);
  w.wss = cv.f((size_t)g->hop * (T - 1) + g->n_fft);
}
