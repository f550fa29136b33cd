// taco_kernels.h -- gfx950 (MI355X, CDNA4) device code for the Tacotron hot path.
//
// Three kernel families cover every op of SURVEY.md section 2b:
//   k_gemm    feed-forward contractions with M = batch*time rows (K2,K3,K5,K7,K9,K16 hoist,K18,K19):
//             conv1d(SAME)/dense as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32), input
//             tile + halo staged ONCE in LDS and shared by all taps, weights read straight from a
//             fragment-native pack, fused epilogue bias->act->BatchNorm affine->residual, optional
//             fused max_pooling1d on the staged tile, optional embedding gather, optional second
//             weight matrix (highway H and T share the x tile).
//   k_skinny  per-time-step stages with M = batch rows (K8 scan, K10,K11,K12 query,K16,K17):
//             v_mfma_f32_16x16x4_f32 with K split across the 8 waves of a workgroup, LDS
//             reduction, fused GRU-gate / GRU-candidate / residual / length-mask epilogues.
//   k_attention  Bahdanau score + normaliser (softmax or monotonic "parallel" scan) + context for one
//             batch row per workgroup (K12-K15): coalesced key/value rows, wave64 reductions/scans.
//
// Math follows SURVEY.md Appendix A (the restatement of models/modules.py, models/rnn_wrappers.py
// and the TF 1.4 ops they call); each kernel cites the reference lines it implements.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TACO_KC 64            // channels staged per LDS chunk
#define TACO_LDSW (TACO_KC + 4) // LDS row stride in floats: (68/4)=17 odd -> ds_read_b128 conflict-free

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_TANH = 3, ACT_SOFTSIGN = 4 };

__device__ __forceinline__ float taco_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float taco_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.0f);
    case ACT_SIGMOID: return taco_sigmoid(v);
    case ACT_TANH: return tanhf(v);
    case ACT_SOFTSIGN: return v / (1.0f + fabsf(v));
    default: return v;
  }
}

// ------------------------------------------------------------------------------------------------
// k_gemm
// ------------------------------------------------------------------------------------------------
// Packed weight layout ("W32"): for n-tile nt (32 columns), k-quad kq (4 consecutive k), column j,
// element e:  wp[((nt*Kq + kq)*32 + j)*4 + e] = W[4*kq + e][32*nt + j], zero padded.
// k runs over (tap, channel) as k = tap*cin_pad + c, cin_pad = Cin rounded up to 8.
// One "k8 group" = two k-quads = the four v_mfma_f32_32x32x2_f32 issued from one float4 of A
// (lane (i=l&31, h=l>>5) holds A[i][8g+4h+e]) and one float4 of B (lane (j,h) holds W[8g+4h+e][j]).
struct GemmVar {
  const float* wp;        // packed weights
  const float* wp2;       // second matrix (highway T) or null
  const float* bias;      // [N] or null
  const float* bias2;     // [N] or null
  const float* bn_scale;  // [N] or null : gamma / sqrt(var + eps)
  const float* bn_shift;  // [N] or null : beta - mean * scale
  int kw, padl;           // taps, left padding ((kw-1)/2, TF 'same')
  int Kq;                 // k-quads per n-tile = kw*cin_pad/4
  int NT;                 // 32-column tiles in the pack
  int N;                  // real output channels
  int coff;               // column offset in the output row (conv-bank concat)
};

struct GemmArgs {
  const float* x;         // input rows [M, ldx] (or embedding table when gather != null)
  const int* gather;      // optional [M] row indices into x
  const GemmVar* vars;    // device array, one per blockIdx.z
  const float* res;       // optional residual [M, ldres]
  const float* rowvec;    // optional per-batch-row vector [M/T, ldrv] (deepvoice before_highway)
  float* out;             // [M, ldo]
  int ldx, M, T, Cin, cin_pad, mpw, act, ldres, ldrv, ldo, vec_ok;
};

// One float4 of the (optionally max-pooled, optionally gathered) input at flat row m, channel c.
// max_pooling1d(pool=w, stride 1, 'same') pads (w-1)/2 left and never lets padding win (A.3).
__device__ __forceinline__ float4 taco_stage_load(const GemmArgs& a, int m, int c) {
  float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m < 0 || m >= a.M || c >= a.Cin) return z;
  if (a.mpw <= 1) {
    const float* p = a.x + (size_t)(a.gather ? a.gather[m] : m) * a.ldx + c;
    if (a.vec_ok) return *reinterpret_cast<const float4*>(p);
    float4 r = z;
    r.x = p[0];
    if (c + 1 < a.Cin) r.y = p[1];
    if (c + 2 < a.Cin) r.z = p[2];
    if (c + 3 < a.Cin) r.w = p[3];
    return r;
  }
  const int t = m % a.T;
  const int pl = (a.mpw - 1) >> 1;
  float4 r = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  for (int jj = 0; jj < a.mpw; ++jj) {
    const int tt = t - pl + jj;
    if (tt < 0 || tt >= a.T) continue;
    const float* p = a.x + (size_t)(m - pl + jj) * a.ldx + c;
    float4 q;
    if (a.vec_ok) {
      q = *reinterpret_cast<const float4*>(p);
    } else {
      q = make_float4(p[0], -INFINITY, -INFINITY, -INFINITY);
      if (c + 1 < a.Cin) q.y = p[1];
      if (c + 2 < a.Cin) q.z = p[2];
      if (c + 3 < a.Cin) q.w = p[3];
    }
    r.x = fmaxf(r.x, q.x); r.y = fmaxf(r.y, q.y); r.z = fmaxf(r.z, q.z); r.w = fmaxf(r.w, q.w);
  }
  if (c + 1 >= a.Cin) r.y = 0.f;
  if (c + 2 >= a.Cin) r.z = 0.f;
  if (c + 3 >= a.Cin) r.w = 0.f;
  return r;
}

// Workgroup = WM x WN x KS waves; each wave owns TM x TN MFMA tiles of 32x32; KS waves split the
// (tap, k8-group) iteration space of every staged chunk and are reduced through LDS at the end.
template <int WM, int WN, int TM, int TN, int KS, bool DUAL>
__global__ __launch_bounds__(64 * WM * WN * KS) void k_gemm(const GemmArgs a) {
  constexpr int NTHR = 64 * WM * WN * KS;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GemmVar v = a.vars[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn / WN, wn = wmn % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int rows = BM + v.kw - 1;

  int tloc[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) tloc[tm] = (m0 + (wm * TM + tm) * 32 + l31) % a.T;
  int ntile[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) ntile[tn] = (n0 >> 5) + wn * TN + tn;

  f32x16 acc[TM][TN];
  f32x16 acc2[DUAL ? TM : 1][DUAL ? TN : 1];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
      if constexpr (DUAL)
        for (int r = 0; r < 16; ++r) acc2[tm][tn][r] = 0.f;
    }

  for (int c0 = 0; c0 < a.cin_pad; c0 += TACO_KC) {
    __syncthreads();
    for (int idx = tid; idx < rows * (TACO_KC / 4); idx += NTHR) {
      const int s = idx / (TACO_KC / 4), c4 = idx % (TACO_KC / 4);
      const float4 val = taco_stage_load(a, m0 - v.padl + s, c0 + 4 * c4);
      *reinterpret_cast<float4*>(&smem[s * TACO_LDSW + 4 * c4]) = val;
    }
    __syncthreads();
    const int ng = min(TACO_KC, a.cin_pad - c0) >> 3;
    int it = 0;
    for (int j = 0; j < v.kw; ++j) {
      const int kqb = ((j * a.cin_pad + c0) >> 2) + lh;
      for (int g = 0; g < ng; ++g, ++it) {
        if (KS > 1 && (it % KS) != ks) continue;
        const int kq = kqb + 2 * g;
        float4 b[TN], b2[DUAL ? TN : 1];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          if (ntile[tn] < v.NT) {
            const size_t off = ((size_t)(ntile[tn] * v.Kq + kq) * 32 + l31) * 4;
            b[tn] = *reinterpret_cast<const float4*>(v.wp + off);
            if constexpr (DUAL) b2[tn] = *reinterpret_cast<const float4*>(v.wp2 + off);
          } else {
            b[tn] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (DUAL) b2[tn] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
        float4 aa[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int srow = (wm * TM + tm) * 32 + l31 + j;
          const float4 t4 = *reinterpret_cast<const float4*>(&smem[srow * TACO_LDSW + 8 * g + 4 * lh]);
          const int tt = tloc[tm] + j - v.padl;   // SAME zero padding + batch-row boundary (A.2)
          const bool ok = (tt >= 0) && (tt < a.T);
          aa[tm] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tm].x, b[tn].x, acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tm].y, b[tn].y, acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tm].z, b[tn].z, acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tm].w, b[tn].w, acc[tm][tn], 0, 0, 0);
            if constexpr (DUAL) {
              acc2[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tm].x, b2[tn].x, acc2[tm][tn], 0, 0, 0);
              acc2[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tm].y, b2[tn].y, acc2[tm][tn], 0, 0, 0);
              acc2[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tm].z, b2[tn].z, acc2[tm][tn], 0, 0, 0);
              acc2[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[tm].w, b2[tn].w, acc2[tm][tn], 0, 0, 0);
            }
          }
      }
    }
  }

  if (KS > 1) {  // split-K reduction through LDS (the staged tile is dead by now)
    constexpr int PER_WAVE = TM * TN * 16 * 64 * (DUAL ? 2 : 1);
    __syncthreads();
    if (ks > 0) {
      float* dst = smem + ((size_t)(ks - 1) * WM * WN + wmn) * PER_WAVE;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            dst[((tm * TN + tn) * 16 + r) * 64 + lane] = acc[tm][tn][r];
            if constexpr (DUAL) dst[TM * TN * 1024 + ((tm * TN + tn) * 16 + r) * 64 + lane] = acc2[tm][tn][r];
          }
    }
    __syncthreads();
    if (ks > 0) return;
    for (int k2 = 1; k2 < KS; ++k2) {
      const float* src = smem + ((size_t)(k2 - 1) * WM * WN + wmn) * PER_WAVE;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            acc[tm][tn][r] += src[((tm * TN + tn) * 16 + r) * 64 + lane];
            if constexpr (DUAL) acc2[tm][tn][r] += src[TM * TN * 1024 + ((tm * TN + tn) * 16 + r) * 64 + lane];
          }
    }
  }

  // epilogue.  C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + (wn * TN + tn) * 32 + l31;
      if (col >= v.N) continue;
      const float bia = v.bias ? v.bias[col] : 0.f;
      float bia2 = 0.f;
      if constexpr (DUAL) bia2 = v.bias2 ? v.bias2[col] : 0.f;
      const float sc = v.bn_scale ? v.bn_scale[col] : 1.f;
      const float sh = v.bn_shift ? v.bn_shift[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= a.M) continue;
        float val;
        if constexpr (DUAL) {  // highway (modules.py:105-120): H*T + x*(1-T)
          const float H = fmaxf(acc[tm][tn][r] + bia, 0.f);
          const float Tg = taco_sigmoid(acc2[tm][tn][r] + bia2);
          const float xin = a.x[(size_t)row * a.ldx + col];
          val = H * Tg + xin * (1.f - Tg);
        } else {
          val = taco_act(acc[tm][tn][r] + bia, a.act);   // conv/dense + bias -> activation
          val = val * sc + sh;                            // -> BatchNorm (modules.py:131)
          if (a.res) val += a.res[(size_t)row * a.ldres + col];        // modules.py:62-69
          if (a.rowvec) val += a.rowvec[(size_t)(row / a.T) * a.ldrv + col];
        }
        a.out[(size_t)row * a.ldo + v.coff + col] = val;
      }
    }
}

// ------------------------------------------------------------------------------------------------
// k_skinny
// ------------------------------------------------------------------------------------------------
// Packed weight layout ("W16"): wp[((nt*Kq + kq)*16 + j)*4 + e] = W[4*kq + e][16*nt + j]; Kq is a
// multiple of 4 (K padded to 16).  One k16 group = the four v_mfma_f32_16x16x4_f32 issued from one
// float4 of A (lane (i=l&15, q=l>>4) holds X[i][16g+4q+e]) and one float4 of B (W[16g+4q+e][j]).
enum {
  EPI_LINEAR = 0,     // out0[r,n] = act(acc + bias)                      (dense; A.1)
  EPI_GRU_GATES = 1,  // s = sigmoid(acc + bias + xg); n<H: out0 = s*h (r*h), else out1 = s (u)   (A.6)
  EPI_GRU_CAND = 2,   // c = tanh(acc + bias + xc); h' = u*h + (1-u)*c; state/outputs              (A.6, A.7)
};

struct SkJob {
  const float* x0; const float* x1; const int* gather0;
  const float* wp; const float* bias;
  const float* e0;   // GATES/CAND: h state [R, lde0]
  const float* e1;   // GATES: precomputed x-part of gates or null; CAND: x-part of candidate or null
  const float* e2;   // CAND: u [R, lde2]
  const float* e3;   // CAND: residual input (ResidualWrapper, tacotron.py:172) or null
  float* o0;         // LINEAR: out; GATES: r*h; CAND: new state h (may alias e0)
  float* o1;         // GATES: u; CAND: h' + residual (or null)
  float* o2;         // CAND: sequence output [R, T, ldo2] (BiGRU) or null ; LINEAR: per-row nonzero flag (int*) or null
  const int* lengths;  // BiGRU sequence_length or null
  int ldx0, ldx1, K0, K, Kq, N, H, epi, act;
  int lde0, lde1, lde2, lde3, ldo0, ldo1, ldo2;
  int step, T, dir, seq_coff;   // BiGRU scan position: dir 0 forward, 1 backward (reverse_sequence)
  int tile0;                    // first workgroup of this job
};
#define SK_MAXJOBS 4
struct SkArgs { int R; int njobs; SkJob j[SK_MAXJOBS]; };

#define SK_NW 8

__device__ __forceinline__ float4 taco_sk_load(const SkJob& jb, int r, int k) {
  // X[r][k..k+3] of the concatenation [x0 (K0 cols) | x1 (K-K0 cols)]
  const float* p; int rem;
  if (k < jb.K0) { p = jb.x0 + (size_t)(jb.gather0 ? jb.gather0[r] : r) * jb.ldx0 + k; rem = jb.K0 - k; }
  else { p = jb.x1 + (size_t)r * jb.ldx1 + (k - jb.K0); rem = jb.K - k; }
  if (rem >= 4 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) return *reinterpret_cast<const float4*>(p);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rem > 0) v.x = p[0];
  if (rem > 1) v.y = p[1];
  if (rem > 2) v.z = p[2];
  if (rem > 3) v.w = p[3];
  return v;
}

template <int RT>
__global__ __launch_bounds__(64 * SK_NW) void k_skinny(const SkArgs a) {
  __shared__ float red[SK_NW * RT * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int ji = 0;
#pragma unroll
  for (int q = 1; q < SK_MAXJOBS; ++q)
    if (q < a.njobs && (int)blockIdx.x >= a.j[q].tile0) ji = q;
  const SkJob& jb = a.j[ji];
  const int nt = blockIdx.x - jb.tile0;
  const int l15 = lane & 15, lq = lane >> 4;
  const int R = a.R;

  f32x4 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int ngroups = jb.Kq >> 2;
  for (int g = wave; g < ngroups; g += SK_NW) {
    const int kq = 4 * g + lq;
    const int k = 4 * kq;
    const float4 b = *reinterpret_cast<const float4*>(jb.wp + ((size_t)(nt * jb.Kq + kq) * 16 + l15) * 4);
    float4 x[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int r = rt * 16 + l15;
      x[rt] = (r < R && k < jb.K) ? taco_sk_load(jb, r, k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].x, b.x, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].y, b.y, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].z, b.z, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].w, b.w, acc[rt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * RT + rt) * 256 + r * 64 + lane] = acc[rt][r];
  __syncthreads();

  // C/D map of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg.
  for (int idx = tid; idx < RT * 256; idx += 64 * SK_NW) {
    const int rt = idx >> 8, w = idx & 255, reg = w >> 6, ln = w & 63;
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < SK_NW; ++wv) s += red[(wv * RT + rt) * 256 + w];
    const int r = rt * 16 + (ln >> 4) * 4 + reg;
    const int n = nt * 16 + (ln & 15);
    if (r >= R || n >= jb.N) continue;
    if (jb.bias) s += jb.bias[n];
    if (jb.epi == EPI_LINEAR) {
      const float y = taco_act(s, jb.act);
      jb.o0[(size_t)r * jb.ldo0 + n] = y;
      if (jb.o2 && y != 0.f) reinterpret_cast<int*>(jb.o2)[r] = 1;   // stop rule helpers.py:29
      continue;
    }
    // BiGRU time mapping (A.7): row active iff step < L; forward t = step, backward t = L-1-step.
    int t = jb.step; bool active = true;
    if (jb.T > 0) {
      const int L = jb.lengths ? jb.lengths[r] : jb.T;
      active = jb.step < L;
      t = (jb.dir && active) ? (L - 1 - jb.step) : jb.step;
    }
    const size_t xrow = (jb.T > 0) ? ((size_t)r * jb.T + t) : (size_t)r;
    if (jb.epi == EPI_GRU_GATES) {
      if (jb.e1) s += jb.e1[xrow * jb.lde1 + n];
      const float sg = taco_sigmoid(s);
      if (n < jb.H) jb.o0[(size_t)r * jb.ldo0 + n] = sg * jb.e0[(size_t)r * jb.lde0 + n];
      else jb.o1[(size_t)r * jb.ldo1 + (n - jb.H)] = sg;
    } else {  // EPI_GRU_CAND
      if (jb.e1) s += jb.e1[xrow * jb.lde1 + n];
      const float c = tanhf(s);
      const float h = jb.e0[(size_t)r * jb.lde0 + n];
      const float u = jb.e2[(size_t)r * jb.lde2 + n];
      const float hn = u * h + (1.f - u) * c;
      if (active) jb.o0[(size_t)r * jb.ldo0 + n] = hn;
      if (jb.o1) jb.o1[(size_t)r * jb.ldo1 + n] = hn + (jb.e3 ? jb.e3[(size_t)r * jb.lde3 + n] : 0.f);
      if (jb.o2) jb.o2[((size_t)r * jb.T + t) * jb.ldo2 + jb.seq_coff + n] = active ? hn : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_attention : one workgroup per batch row
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
  const float* q;        // [B, A] processed query (h_att . W_q)
  const float* keys;     // [B, T_in, A]
  const float* values;   // [B, T_in, D]
  const float* v;        // [A] attention_v (bah_norm: g*v/|v| folded at finalize)
  const float* battn;    // [A] attention_b (bah_norm) or null
  const float* score_bias;  // [1] (bah_mon) or null
  const float* manual;   // [B, n_steps, T_in] or null (rnn_wrappers.py:313-317)
  float* align;          // [B, T_in] state: previous alignments in, new alignments out
  float* hist;           // [B, T_in, n_steps] or null (tacotron.py:238-239 layout)
  float* ctx;            // [B, D]
  int T_in, A, D, type, step, n_steps;
};

#define ATT_NW 8
#define ATT_MAXT 2048

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
// inclusive wave64 prefix sum
__device__ __forceinline__ float wave_scan(float x, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  return x;
}

__global__ __launch_bounds__(64 * ATT_NW) void k_attention(const AttnArgs a) {
  __shared__ float sc[ATT_MAXT];     // scores -> alignments
  __shared__ float tmp[ATT_MAXT];
  __shared__ float tmp2[ATT_MAXT];
  __shared__ __attribute__((aligned(16))) float cred[ATT_NW * 256];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T_in;
  float* al = a.align + (size_t)b * T;

  if (a.manual) {
    for (int j = tid; j < T; j += 64 * ATT_NW) sc[j] = a.manual[((size_t)b * a.n_steps + a.step) * T + j];
    __syncthreads();
  } else {
    // scores: e[j] = sum_a v[a] * tanh(keys[b,j,a] + q[b,a] (+ b[a]))   (_bahdanau_score, A.9)
    const float* qb = a.q + (size_t)b * a.A;
    for (int j = wave; j < T; j += ATT_NW) {
      const float* kr = a.keys + ((size_t)b * T + j) * a.A;
      float part = 0.f;
      for (int c = lane * 4; c < a.A; c += 256) {
        const float4 k4 = *reinterpret_cast<const float4*>(kr + c);
        const float4 q4 = *reinterpret_cast<const float4*>(qb + c);
        const float4 v4 = *reinterpret_cast<const float4*>(a.v + c);
        float4 s4 = make_float4(k4.x + q4.x, k4.y + q4.y, k4.z + q4.z, k4.w + q4.w);
        if (a.battn) {
          const float4 b4 = *reinterpret_cast<const float4*>(a.battn + c);
          s4.x += b4.x; s4.y += b4.y; s4.z += b4.z; s4.w += b4.w;
        }
        part += v4.x * tanhf(s4.x) + v4.y * tanhf(s4.y) + v4.z * tanhf(s4.z) + v4.w * tanhf(s4.w);
      }
      part = wave_sum(part);
      if (lane == 0) sc[j] = part;
    }
    __syncthreads();
    if (wave == 0) {
      const int C = (T + 63) >> 6;          // contiguous elements per lane
      const int j0 = lane * C, j1 = min(j0 + C, T);
      if (a.type == 2) {
        // monotonic_attention(mode='parallel') (A.10):
        //   p = sigmoid(e + bias); cp = exp(cumsum_excl(log(clip(1-p, tiny, 1))))
        //   alpha = p * cp * cumsum(prev / clip(cp, 1e-10, 1))
        const float sb = a.score_bias ? a.score_bias[0] : 0.f;
        float run = 0.f;
        for (int j = j0; j < j1; ++j) {
          const float p = taco_sigmoid(sc[j] + sb);
          const float lg = logf(fminf(fmaxf(1.f - p, 1.17549435e-38f), 1.f));
          sc[j] = p;
          tmp[j] = run;      // exclusive within the lane's chunk
          run += lg;
        }
        float off = wave_scan(run, lane) - run;   // exclusive offset of this lane's chunk
        float run2 = 0.f;
        for (int j = j0; j < j1; ++j) {
          const float cp = expf(tmp[j] + off);
          tmp[j] = cp;
          run2 += al[j] / fminf(fmaxf(cp, 1e-10f), 1.f);
          tmp2[j] = run2;    // inclusive within the lane's chunk
        }
        const float off2 = wave_scan(run2, lane) - run2;
        for (int j = j0; j < j1; ++j) sc[j] = sc[j] * tmp[j] * (tmp2[j] + off2);
      } else {
        // softmax over T_in (no memory_sequence_length mask, A.8)
        float mx = -INFINITY;
        for (int j = j0; j < j1; ++j) mx = fmaxf(mx, sc[j]);
        mx = wave_max(mx);
        float sm = 0.f;
        for (int j = j0; j < j1; ++j) { const float e = expf(sc[j] - mx); sc[j] = e; sm += e; }
        sm = wave_sum(sm);
        for (int j = j0; j < j1; ++j) sc[j] = sc[j] / sm;
      }
    }
    __syncthreads();
  }
  for (int j = tid; j < T; j += 64 * ATT_NW) {
    const float x = sc[j];
    al[j] = x;
    if (a.hist) a.hist[((size_t)b * T + j) * a.n_steps + a.step] = x;
  }
  // context[d] = sum_j alpha[j] * values[b,j,d]   (rnn_wrappers.py:322-334)
  for (int d0 = 0; d0 < a.D; d0 += 256) {
    const int d = d0 + lane * 4;
    float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d < a.D) {
      for (int j = wave; j < T; j += ATT_NW) {
        const float w = sc[j];
        const float4 v4 = *reinterpret_cast<const float4*>(a.values + ((size_t)b * T + j) * a.D + d);
        c4.x += w * v4.x; c4.y += w * v4.y; c4.z += w * v4.z; c4.w += w * v4.w;
      }
    }
    __syncthreads();
    *reinterpret_cast<float4*>(&cred[wave * 256 + lane * 4]) = c4;
    __syncthreads();
    if (tid < 256 && d0 + tid < a.D) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < ATT_NW; ++w) s += cred[w * 256 + tid];
      a.ctx[(size_t)b * a.D + d0 + tid] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// stop rule of helpers.py:29 + TF dynamic_decode: a row is finished once a step's r*num_mels outputs are
// all exactly 0; the loop ends after the first step at which every row is finished.
// nz [n_steps, B] : 1 if row b emitted any non-zero at step t.
__global__ void k_stop_step(const int* nz, int B, int n_steps, int* stop) {
  __shared__ int first[1024];
  const int tid = threadIdx.x;
  int worst = 0;
  for (int b = tid; b < B; b += blockDim.x) {
    int f = n_steps;   // first all-zero step of row b
    for (int t = 0; t < n_steps; ++t)
      if (nz[(size_t)t * B + b] == 0) { f = t; break; }
    worst = max(worst, f);
  }
  first[tid] = worst;
  __syncthreads();
  if (tid == 0) {
    int w = 0;
    for (int i = 0; i < (int)blockDim.x; ++i) w = max(w, first[i]);
    *stop = min(w + 1, n_steps);
  }
}

// initial alignments (TF-sem: zeros for Bahdanau, one_hot(0) for BahdanauMonotonic)
__global__ void k_init_align(float* al, int B, int T, int mono) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * T) al[i] = (mono && (i % T) == 0) ? 1.f : 0.f;
}

// rows gather: out[r, :] = table[idx[r], :]
__global__ void k_gather_rows(const float* table, const int* idx, int R, int D, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < R * D) { const int r = i / D, d = i % D; out[i] = table[(size_t)(idx ? idx[r] : 0) * D + d]; }
}

// strided 2-D copy (debug state dumps, initial states)
__global__ void k_copy2d(const float* src, int lds, float* dst, int ldd, int R, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < R * C) { const int r = i / C, c = i % C; dst[(size_t)r * ldd + c] = src ? src[(size_t)r * lds + c] : 0.f; }
}
