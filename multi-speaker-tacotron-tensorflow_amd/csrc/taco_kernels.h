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

// Kernel-argument structs are read with scalar loads.  Left alone, hipcc sinks each field's s_load to
// its first use inside a branch and waits for it there: dozens of serialized scalar-memory round trips
// per launch (measured: ~5 us of a 7 us launch).  PIN forces a value to be materialised in an SGPR at
// the top of the kernel, so all fields arrive in one batch of s_load_dwordx8/x16.
// Pointers are pinned as address-space-1 values: left generic (as anything read out of a by-value struct array is, and as the
// empty asm makes every pointer), each access through them becomes a FLAT instruction, which counts on lgkmcnt as well as
// vmcnt -- so every wait for an LDS read also drained all outstanding global prefetches.  Typed as global before the asm and
// cast back after it, the address-space inference turns the accesses into global_load / global_store.
template <class T> __device__ __forceinline__ void taco_pin(T*& p) {
  __attribute__((address_space(1))) T* g = (__attribute__((address_space(1))) T*)p;
  asm volatile("" : "+s"(g));
  p = (T*)g;
}
template <class T> __device__ __forceinline__ void taco_pin(T& v) { asm volatile("" : "+s"(v)); }
// per-lane pointer: forces the value to exist at this point of the program (same address-space round trip)
template <class T> __device__ __forceinline__ void taco_pin_v(T*& p) {
  __attribute__((address_space(1))) T* g = (__attribute__((address_space(1))) T*)p;
  asm volatile("" : "+v"(g));
  p = (T*)g;
}
#define PIN(x) taco_pin(x)

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
  // split-bf16 packs (k_gemm_bf3): hi/lo bf16 halves of the same weights, fragment-native for 32x32x16 MFMA
  const unsigned short* bh; const unsigned short* bl; const unsigned short* bh2; const unsigned short* bl2;
  int K16, cin_pad16;     // k16 groups per n-tile = kw*cin_pad16/16
  // third planes (training shadow model only): l3 = bf16(w - hi - lo), the operand of the six-product (fp32-grade) instantiation X6
  const unsigned short* bl3; const unsigned short* bl3_2;
};

struct GemmArgs {
  const float* x;         // input rows [M, ldx] (or embedding table when gather != null)
  const int* gather;      // optional [M] row indices into x
  const float* res;       // optional residual [M, ldres]
  const float* rowvec;    // optional per-batch-row vector [M/T, ldrv] (deepvoice before_highway)
  const int* rev_len;     // with rev_col0 >= 0: columns >= rev_col0 of row (b,t) are stored at row (b, L_b-1-t) for t < L_b
                          // (tf.reverse_sequence, A.7); null lengths = T
  float* out;             // [M, ldo]
  int ldx, M, T, Cin, cin_pad, mpw, act, ldres, ldrv, ldo, vec_ok, rev_col0;
  int t_begin, t_len, tiles_per_b;   // time-window mode (t_len > 0): rows (b, t_begin + i), i < t_len, for every batch row b
  float* aux0; float* aux1;          // training tape (DUAL only, nullable): highway H = relu(.) and T = sigmoid(.), [M, ldo]
  GemmVar v[16];          // one per blockIdx.z (conv-bank widths); by value so the fields arrive by scalar loads
};

// One float4 of the (optionally max-pooled, optionally gathered) input at flat row m, channel c.
// max_pooling1d(pool=w, stride 1, 'same') pads (w-1)/2 left and never lets padding win (A.3).
template <typename ArgsT>
__device__ __forceinline__ float4 taco_stage_load(const ArgsT& a, int m, int c) {
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
__global__ __launch_bounds__(64 * WM * WN * KS) void k_gemm(const GemmArgs a_in) {
  constexpr int NTHR = 64 * WM * WN * KS;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  struct { const float* x; const int* gather; const float* res; const float* rowvec; const int* rev_len; float* out;
           int ldx, M, T, Cin, cin_pad, mpw, act, ldres, ldrv, ldo, vec_ok, rev_col0, t_begin, t_len, tiles_per_b;
           float* aux0; float* aux1; } a =
      {a_in.x, a_in.gather, a_in.res, a_in.rowvec, a_in.rev_len, a_in.out, a_in.ldx, a_in.M, a_in.T, a_in.Cin, a_in.cin_pad,
       a_in.mpw, a_in.act, a_in.ldres, a_in.ldrv, a_in.ldo, a_in.vec_ok, a_in.rev_col0, a_in.t_begin, a_in.t_len, a_in.tiles_per_b,
       a_in.aux0, a_in.aux1};
  PIN(a.t_begin); PIN(a.t_len); PIN(a.tiles_per_b);
  PIN(a.rev_len); PIN(a.rev_col0); PIN(a.aux0); PIN(a.aux1);
  PIN(a.x); PIN(a.gather); PIN(a.res); PIN(a.rowvec); PIN(a.out);
  PIN(a.ldx); PIN(a.M); PIN(a.T); PIN(a.Cin); PIN(a.cin_pad); PIN(a.mpw); PIN(a.act); PIN(a.ldres); PIN(a.ldrv); PIN(a.ldo); PIN(a.vec_ok);
  GemmVar v = a_in.v[blockIdx.z];
  PIN(v.wp); PIN(v.wp2); PIN(v.bias); PIN(v.bias2); PIN(v.bn_scale); PIN(v.bn_shift);
  PIN(v.kw); PIN(v.padl); PIN(v.Kq); PIN(v.NT); PIN(v.N); PIN(v.coff);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn / WN, wn = wmn % WN;
  // flat mode: M-tiles over rows m = b*T + t; window mode: tiles over [t_begin, t_begin + t_len) of every batch row
  // (the post-net's feed-forward stages run chunk by chunk behind the decoder; halo rows outside the window are
  //  read like any other row -- they were produced by earlier chunks)
  int m0 = blockIdx.x * BM, row_limit = a.M;
  if (a.t_len > 0) {
    const int bb = blockIdx.x / a.tiles_per_b, tile = blockIdx.x - bb * a.tiles_per_b;
    m0 = bb * a.T + a.t_begin + tile * BM;
    row_limit = bb * a.T + a.t_begin + a.t_len;
  }
  const int n0 = blockIdx.y * BN;
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

  // Staging is software-pipelined: the global loads of chunk c+1 are issued into registers right after
  // chunk c landed in LDS, so their latency overlaps chunk c's MFMAs (one LDS buffer, two barriers/chunk).
  constexpr int NPRE = ((BM + 15) * (TACO_KC / 4) + NTHR - 1) / NTHR;
  float4 pre[NPRE];
  const int nstage = rows * (TACO_KC / 4);
#pragma unroll
  for (int u = 0; u < NPRE; ++u) {
    const int idx = tid + u * NTHR;
    if (idx < nstage) pre[u] = taco_stage_load(a, m0 - v.padl + idx / (TACO_KC / 4), 4 * (idx % (TACO_KC / 4)));
  }
  for (int c0 = 0; c0 < a.cin_pad; c0 += TACO_KC) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int idx = tid + u * NTHR;
      if (idx < nstage) *reinterpret_cast<float4*>(&smem[(idx / (TACO_KC / 4)) * TACO_LDSW + 4 * (idx % (TACO_KC / 4))]) = pre[u];
    }
    __syncthreads();
    if (c0 + TACO_KC < a.cin_pad) {
#pragma unroll
      for (int u = 0; u < NPRE; ++u) {
        const int idx = tid + u * NTHR;
        if (idx < nstage) pre[u] = taco_stage_load(a, m0 - v.padl + idx / (TACO_KC / 4), c0 + TACO_KC + 4 * (idx % (TACO_KC / 4)));
      }
    }
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
        if (row >= row_limit) continue;
        float val;
        if constexpr (DUAL) {  // highway (modules.py:105-120): H*T + x*(1-T)
          const float H = fmaxf(acc[tm][tn][r] + bia, 0.f);
          const float Tg = taco_sigmoid(acc2[tm][tn][r] + bia2);
          const float xin = a.x[(size_t)row * a.ldx + col];
          val = H * Tg + xin * (1.f - Tg);
          if (a.aux0) { a.aux0[(size_t)row * a.ldo + col] = H; a.aux1[(size_t)row * a.ldo + col] = Tg; }
        } else {
          val = taco_act(acc[tm][tn][r] + bia, a.act);   // conv/dense + bias -> activation
          val = val * sc + sh;                            // -> BatchNorm (modules.py:131)
          if (a.res) val += a.res[(size_t)row * a.ldres + col];        // modules.py:62-69
          if (a.rowvec) val += a.rowvec[(size_t)(row / a.T) * a.ldrv + col];
        }
        int orow = row;
        if (a.rev_col0 >= 0 && col >= a.rev_col0) {
          const int bb = row / a.T, tt = row - bb * a.T;
          const int L = a.rev_len ? a.rev_len[bb] : a.T;
          if (tt < L) orow = bb * a.T + (L - 1 - tt);
        }
        a.out[(size_t)orow * a.ldo + v.coff + col] = val;
      }
    }
}

// ------------------------------------------------------------------------------------------------
// k_gemm_bf3 : the same implicit GEMM on the bf16 matrix cores with fp32-grade accuracy
// ------------------------------------------------------------------------------------------------
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32-input MFMA.  Every fp32 operand x is split
// into hi = bf16(x) and lo = bf16(x - hi) (16 mantissa bits together) and each k16 group issues three
// MFMAs: hi*hi + hi*lo + lo*hi (the dropped lo*lo term is ~2^-16 relative), all accumulated in fp32.
// Weights are split once at finalize; activations are split when the tile is staged in LDS (two bf16
// tiles, same bytes as one fp32 tile).  Measured error vs the float64 oracle ~1e-5 on O(1) outputs per layer; used for every
// feed-forward GEMM of inference (167 of 180 GFLOP @C2): end to end mel 2.7e-6, linear 3.4e-6, alignment argmax identical.
// Pack layout (k16-major: the TN column tiles a wave loads for one k16 step are adjacent 1 KB blocks, i.e. consecutive cache
// lines; a tile-major order measured the same):
//   b?[(((k16*NT + nt)*2 + h)*32 + j)*8 + e] = W[16*k16 + 8*h + e][32*nt + j]; A fragment of lane
// (i = l&31, h = l>>5) = X[i][16*g + 8*h + e], e < 8 -- A and B use the same (h, e) -> k pairing.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define BF3_LDSW (TACO_KC + 8)    // bf16 elements per LDS row: 144 bytes = 9 x 16 B (odd) -> conflict-free ds_read_b128

// fp32 -> (hi, lo) bf16 halves with hi + lo ~ x to 16 mantissa bits: hi = RNE(x), lo = RNE(x - hi).  gfx950 rounds two floats
// per instruction (v_cvt_pk_bf16_f32): 12 VALU per float4 instead of ~50 with integer rounding -- this conversion sits between
// the two staging barriers of every chunk, where no wave of the workgroup issues MFMAs.
typedef __bf16 taco_bf16x2 __attribute__((ext_vector_type(2)));
typedef float taco_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned taco_pk_bf16(float a, float b) {
  const taco_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, taco_bf16x2));
}
__device__ __forceinline__ void taco_split_bf16x4(const float4 f, uint2& hi, uint2& lo) {
  hi.x = taco_pk_bf16(f.x, f.y); hi.y = taco_pk_bf16(f.z, f.w);
  lo.x = taco_pk_bf16(f.x - __uint_as_float(hi.x << 16), f.y - __uint_as_float(hi.x & 0xffff0000u));
  lo.y = taco_pk_bf16(f.z - __uint_as_float(hi.y << 16), f.w - __uint_as_float(hi.y & 0xffff0000u));
}

#ifdef TACO_TRACE
__device__ long long taco_trace[64];
#define TRC(i) do { if (trc) taco_trace[i] = clock64(); } while (0)
#else
#define TRC(i) do {} while (0)
#endif
// KS > 1 (small-M layers, where even 64x64 tiles leave most CUs without a workgroup): KS groups of WM x WN waves share the
// workgroup.  Every staging round brings in KS consecutive 64-channel sub-chunks (one LDS tile each); group ks runs the same
// tap x k16 loop over sub-chunk ks, so all groups issue MFMAs at once (KS waves per SIMD hide each other's L2 latency), and the
// partial accumulators are summed through LDS before the epilogue.
// X6: every operand split THREE ways (hi, lo, l3 = bf16(x - hi - lo): 24 mantissa bits) and six products per k16 group -- l3*hi, hi*l3,
// lo*lo, lo*hi, hi*lo, hi*hi, as k_wgrad_bf3 -- i.e. fp32-grade products on the bf16 pipe (2^-24 per product instead of 2^-16): the
// forward GEMMs of the TRAINING step, whose ReLU / max-pool decisions must not differ from fp32's (taco_train_set_exact_gemm mode 4).
template <int WM, int WN, int TM, int TN, bool DUAL, int GPI, int KS = 1, bool X6 = false>
__global__ __launch_bounds__(64 * WM * WN * KS, (KS == 1 && GPI == 1 && TM < 4 && !X6 && !(DUAL && TM * TN >= 4)) ? 2 : 1) void k_gemm_bf3(const GemmArgs a_in) {
  constexpr int NTHR = 64 * WM * WN * KS;
  constexpr int NPL = X6 ? 3 : 2;                                  // planes of a staged tile
  constexpr int SUBSZ = NPL * (WM * TM * 32 + 15) * BF3_LDSW;     // bf16 elements of one sub-chunk tile (hi plane, then lo plane[, then l3])
  constexpr int KCS = TACO_KC * KS;                              // channels per staging round
#ifdef TACO_TRACE
  const bool trc = (blockIdx.x == gridDim.x / 2) && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0;
  TRC(0);
#endif
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  struct { const float* x; const int* gather; const float* res; const float* rowvec; const int* rev_len; float* out;
           int ldx, M, T, Cin, cin_pad, mpw, act, ldres, ldrv, ldo, vec_ok, rev_col0, t_begin, t_len, tiles_per_b; float* aux0; float* aux1; } a =
      {a_in.x, a_in.gather, a_in.res, a_in.rowvec, a_in.rev_len, a_in.out, a_in.ldx, a_in.M, a_in.T, a_in.Cin, a_in.cin_pad,
       a_in.mpw, a_in.act, a_in.ldres, a_in.ldrv, a_in.ldo, a_in.vec_ok, a_in.rev_col0, a_in.t_begin, a_in.t_len, a_in.tiles_per_b,
       a_in.aux0, a_in.aux1};
  PIN(a.x); PIN(a.gather); PIN(a.res); PIN(a.rowvec); PIN(a.out); PIN(a.rev_len); PIN(a.aux0); PIN(a.aux1);
  PIN(a.ldx); PIN(a.M); PIN(a.T); PIN(a.Cin); PIN(a.mpw); PIN(a.act); PIN(a.ldres); PIN(a.ldrv); PIN(a.ldo); PIN(a.vec_ok);
  PIN(a.rev_col0); PIN(a.t_begin); PIN(a.t_len); PIN(a.tiles_per_b);
  GemmVar v = a_in.v[blockIdx.z];
  PIN(v.bias); PIN(v.bias2); PIN(v.bn_scale); PIN(v.bn_shift); PIN(v.bh); PIN(v.bl); PIN(v.bh2); PIN(v.bl2); PIN(v.bl3); PIN(v.bl3_2);
  PIN(v.kw); PIN(v.padl); PIN(v.NT); PIN(v.N); PIN(v.coff); PIN(v.K16); PIN(v.cin_pad16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn / WN, wn = wmn % WN;
  int m0 = blockIdx.x * BM, row_limit = a.M;
  if (a.t_len > 0) {
    const int bb = blockIdx.x / a.tiles_per_b, tile = blockIdx.x - bb * a.tiles_per_b;
    m0 = bb * a.T + a.t_begin + tile * BM;
    row_limit = bb * a.T + a.t_begin + a.t_len;
  }
  const int n0 = blockIdx.y * BN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int rows = BM + v.kw - 1;
  unsigned short* tall = reinterpret_cast<unsigned short*>(smem);
  unsigned short* thi = tall + (size_t)ks * SUBSZ;                 // this wave group's tile
  unsigned short* tlo = thi + (size_t)(BM + 15) * BF3_LDSW;
  unsigned short* tl3 = tlo + (size_t)(BM + 15) * BF3_LDSW;       // (X6 only)

  int tloc[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) tloc[tm] = (m0 + (wm * TM + tm) * 32 + l31) % a.T;
  int ntile[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) ntile[tn] = (n0 >> 5) + wn * TN + tn;

  // per-column epilogue operands, requested now: fetched at the end they cost one cold-miss round trip per column tile
  float ebia[TN], ebia2[TN], esc[TN], esh[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + (wn * TN + tn) * 32 + l31, cc = col < v.N ? col : 0;
    ebia[tn] = v.bias ? v.bias[cc] : 0.f;
    ebia2[tn] = (DUAL && v.bias2) ? v.bias2[cc] : 0.f;
    esc[tn] = (!DUAL && v.bn_scale) ? v.bn_scale[cc] : 1.f;
    esh[tn] = (!DUAL && v.bn_shift) ? v.bn_shift[cc] : 0.f;
  }

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

  constexpr int NPRE = ((BM + 15) * (KCS / 4) + NTHR - 1) / NTHR;
  float4 pre[NPRE];
  const int nstage = rows * (KCS / 4);
#pragma unroll
  for (int u = 0; u < NPRE; ++u) {
    const int idx = tid + u * NTHR;
    if (idx < nstage) pre[u] = taco_stage_load(a, m0 - v.padl + idx / (KCS / 4), 4 * (idx % (KCS / 4)));
  }
  // B fragments (packed weights, straight from L2) ping-pong between two register sets in groups of GPI k16 steps: the loads
  // of group i+1 are issued before the GPI x 3 x TM x TN MFMAs of group i and are first waited for a full group later
  // (s_waitcnt vmcnt(#loads of one group)), also across the LDS re-staging at every 64-channel chunk.  For that count to be
  // static every load is unconditional: column tiles past the pack are clamped to the last one (their accumulators are
  // never stored), and past the end of K the last group is simply fetched again.  The group count per chunk is even
  // (kw x {2,4} k16 steps; GPI = 2 is only launched when every chunk has 4), so the two sets swap roles without copies.
  TRC(1);
  int ntc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) ntc[tn] = min(ntile[tn], v.NT - 1);
  auto load_grp = [&](int c0, int pi, uint4 (&uh)[GPI][TN], uint4 (&ul)[GPI][TN], uint4 (&uh2)[GPI][DUAL ? TN : 1],
                      uint4 (&ul2)[GPI][DUAL ? TN : 1], uint4 (&u3)[GPI][X6 ? TN : 1], uint4 (&u32)[GPI][(X6 && DUAL) ? TN : 1]) {
    const int gpc = (min(TACO_KC, v.cin_pad16 - c0) >> 4) / GPI;      // groups per tap in this chunk
    const int j = pi / gpc, g0 = GPI * (pi - j * gpc);
#pragma unroll
    for (int h = 0; h < GPI; ++h) {
      const int k16 = ((j * v.cin_pad16 + c0) >> 4) + g0 + h;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const size_t off = ((((size_t)k16 * v.NT + ntc[tn]) * 2 + lh) * 32 + l31) * 8;
        uh[h][tn] = *reinterpret_cast<const uint4*>(v.bh + off); ul[h][tn] = *reinterpret_cast<const uint4*>(v.bl + off);
        if constexpr (DUAL) { uh2[h][tn] = *reinterpret_cast<const uint4*>(v.bh2 + off); ul2[h][tn] = *reinterpret_cast<const uint4*>(v.bl2 + off); }
        if constexpr (X6) { u3[h][tn] = *reinterpret_cast<const uint4*>(v.bl3 + off); if constexpr (DUAL) u32[h][tn] = *reinterpret_cast<const uint4*>(v.bl3_2 + off); }
      }
    }
  };
  // the group after (c0, pi): next one of this chunk, first one of the next chunk, or (end of K) this one again
  auto load_next = [&](int c0, int pi, int npair, uint4 (&uh)[GPI][TN], uint4 (&ul)[GPI][TN], uint4 (&uh2)[GPI][DUAL ? TN : 1],
                       uint4 (&ul2)[GPI][DUAL ? TN : 1], uint4 (&u3)[GPI][X6 ? TN : 1], uint4 (&u32)[GPI][(X6 && DUAL) ? TN : 1]) {
    int nc0 = c0, npi = pi + 1;
    if (npi == npair) { nc0 = c0 + KCS; npi = 0; }
    if (nc0 >= v.cin_pad16) { nc0 = c0; npi = pi; }
    load_grp(nc0, npi, uh, ul, uh2, ul2, u3, u32);
  };
  auto mma_grp = [&](int c0, int pi, const uint4 (&uh)[GPI][TN], const uint4 (&ul)[GPI][TN], const uint4 (&uh2)[GPI][DUAL ? TN : 1],
                     const uint4 (&ul2)[GPI][DUAL ? TN : 1], const uint4 (&u3)[GPI][X6 ? TN : 1], const uint4 (&u32)[GPI][(X6 && DUAL) ? TN : 1]) {
    const int gpc = (min(TACO_KC, v.cin_pad16 - c0) >> 4) / GPI;
    const int j = pi / gpc, g0 = GPI * (pi - j * gpc);
#pragma unroll
    for (int h = 0; h < GPI; ++h) {
      const int g = g0 + h;
      bf16x8 ah[TM], al[TM], a3[X6 ? TM : 1];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int srow = (wm * TM + tm) * 32 + l31 + j;
        const int off = srow * BF3_LDSW + 16 * g + 8 * lh;
        uint4 xh = *reinterpret_cast<const uint4*>(thi + off), xl = *reinterpret_cast<const uint4*>(tlo + off);
        const int tt = tloc[tm] + j - v.padl;   // SAME zero padding + batch-row boundary (A.2): masked, not branched
        const unsigned keep = ((tt >= 0) && (tt < a.T)) ? 0xffffffffu : 0u;
        xh.x &= keep; xh.y &= keep; xh.z &= keep; xh.w &= keep; xl.x &= keep; xl.y &= keep; xl.z &= keep; xl.w &= keep;
        ah[tm] = __builtin_bit_cast(bf16x8, xh); al[tm] = __builtin_bit_cast(bf16x8, xl);
        if constexpr (X6) {
          uint4 x3 = *reinterpret_cast<const uint4*>(tl3 + off);
          x3.x &= keep; x3.y &= keep; x3.z &= keep; x3.w &= keep;
          a3[tm] = __builtin_bit_cast(bf16x8, x3);
        }
      }
      // term-major order: the TM*TN independent accumulators sit between two MFMAs on the same accumulator
      constexpr int NTERM = X6 ? 6 : 3;
#pragma unroll
      for (int term = 0; term < NTERM; ++term)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, uh[h][tn]), bl = __builtin_bit_cast(bf16x8, ul[h][tn]);
            bf16x8 aa, bb;
            if constexpr (X6) {      // small terms first: a3*bh, ah*b3, al*bl, al*bh, ah*bl, ah*bh
              const bf16x8 b3 = __builtin_bit_cast(bf16x8, u3[h][tn]);
              aa = (term == 0) ? a3[tm] : (term == 2 || term == 3) ? al[tm] : ah[tm];
              bb = (term == 1) ? b3 : (term == 2 || term == 4) ? bl : bh;
            } else {                 // al*bh, ah*bl, ah*bh
              aa = (term == 0) ? al[tm] : ah[tm];
              bb = (term == 1) ? bl : bh;
            }
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa, bb, acc[tm][tn], 0, 0, 0);
            if constexpr (DUAL) {
              const bf16x8 bh2 = __builtin_bit_cast(bf16x8, uh2[h][tn]), bl2 = __builtin_bit_cast(bf16x8, ul2[h][tn]);
              bf16x8 bb2;
              if constexpr (X6) { const bf16x8 b32 = __builtin_bit_cast(bf16x8, u32[h][tn]); bb2 = (term == 1) ? b32 : (term == 2 || term == 4) ? bl2 : bh2; }
              else bb2 = (term == 1) ? bl2 : bh2;
              acc2[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa, bb2, acc2[tm][tn], 0, 0, 0);
            }
          }
    }
  };
  uint4 pbh[GPI][TN], pbl[GPI][TN], pbh2[GPI][DUAL ? TN : 1], pbl2[GPI][DUAL ? TN : 1], pb3[GPI][X6 ? TN : 1], pb32[GPI][(X6 && DUAL) ? TN : 1];     // set P
  uint4 qbh[GPI][TN], qbl[GPI][TN], qbh2[GPI][DUAL ? TN : 1], qbl2[GPI][DUAL ? TN : 1], qb3[GPI][X6 ? TN : 1], qb32[GPI][(X6 && DUAL) ? TN : 1];     // set Q
  load_grp((ks * TACO_KC < v.cin_pad16) ? ks * TACO_KC : 0, 0, pbh, pbl, pbh2, pbl2, pb3, pb32);
  int trci = 3;
  TRC(2);
  for (int cr = 0; cr < v.cin_pad16; cr += KCS) {
    __syncthreads();
    TRC(trci); ++trci;
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int idx = tid + u * NTHR;
      if (idx < nstage) {
        uint2 h4, l4;
        taco_split_bf16x4(pre[u], h4, l4);
        const int cq = idx % (KCS / 4);
        const int off = (cq / (TACO_KC / 4)) * SUBSZ + (idx / (KCS / 4)) * BF3_LDSW + 4 * (cq % (TACO_KC / 4));
        *reinterpret_cast<uint2*>(tall + off) = h4;
        *reinterpret_cast<uint2*>(tall + off + (BM + 15) * BF3_LDSW) = l4;
        if constexpr (X6) {       // third plane: what hi + lo leave of x
          const float4 f = pre[u];
          uint2 t4;
          t4.x = taco_pk_bf16(f.x - __uint_as_float(h4.x << 16) - __uint_as_float(l4.x << 16),
                              f.y - __uint_as_float(h4.x & 0xffff0000u) - __uint_as_float(l4.x & 0xffff0000u));
          t4.y = taco_pk_bf16(f.z - __uint_as_float(h4.y << 16) - __uint_as_float(l4.y << 16),
                              f.w - __uint_as_float(h4.y & 0xffff0000u) - __uint_as_float(l4.y & 0xffff0000u));
          *reinterpret_cast<uint2*>(tall + off + 2 * (BM + 15) * BF3_LDSW) = t4;
        }
      }
    }
    __syncthreads();
    if (cr + KCS < v.cin_pad16) {
#pragma unroll
      for (int u = 0; u < NPRE; ++u) {
        const int idx = tid + u * NTHR;
        if (idx < nstage) pre[u] = taco_stage_load(a, m0 - v.padl + idx / (KCS / 4), cr + KCS + 4 * (idx % (KCS / 4)));
      }
    }
    TRC(trci); ++trci;
    const int c0 = cr + ks * TACO_KC;                                                  // this wave group's sub-chunk
    // (a wave whose column tiles all lie past the matrix -- N = 1025 leaves seven of the eight waves of the last column tile without
    // a column -- takes part in the staging and its barriers only: no weight loads, no MFMAs)
    const int npair = (c0 < v.cin_pad16 && ntile[0] < v.NT) ? v.kw * ((min(TACO_KC, v.cin_pad16 - c0) >> 4) / GPI) : 0;          // even
    for (int pi = 0; pi < npair; pi += 2) {
      // the scheduling barriers keep the loads ahead of the MFMA group they are meant to hide behind (left alone, the
      // scheduler sinks each load to just before its use to save registers)
      load_next(c0, pi, npair, qbh, qbl, qbh2, qbl2, qb3, qb32);
      __builtin_amdgcn_sched_barrier(0);
      mma_grp(c0, pi, pbh, pbl, pbh2, pbl2, pb3, pb32);
      __builtin_amdgcn_sched_barrier(0);
      load_next(c0, pi + 1, npair, pbh, pbl, pbh2, pbl2, pb3, pb32);
      __builtin_amdgcn_sched_barrier(0);
      mma_grp(c0, pi + 1, qbh, qbl, qbh2, qbl2, qb3, qb32);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  TRC(trci); ++trci;
  if constexpr (KS > 1) {     // sum the KS partial tiles through LDS (the staged tiles are dead by now)
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

  // epilogue (C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).  Everything that does not depend on
  // the element is decided once: the row mapping (time reversal of the BiGRU's backward half) per row tile, the activation and
  // the optional operands per launch -- the common case (bias, none/ReLU, BatchNorm affine) is 16 short store sequences per
  // tile instead of 16 copies of a five-way activation switch with its exp/tanh bodies (that version spent a third of a
  // short-K workgroup's life here, mostly fetching instructions).
  const bool relu = a.act == ACT_RELU, simple_act = a.act == ACT_NONE || a.act == ACT_RELU;
  const bool extras = a.res || a.rowvec;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    int rowv[16], orev[16], bidx[16];
    const int rbase = m0 + (wm * TM + tm) * 32 + 4 * lh;
#pragma unroll
    for (int r = 0; r < 16; ++r) { rowv[r] = rbase + (r & 3) + 8 * (r >> 2); orev[r] = rowv[r]; bidx[r] = 0; }
    const bool anyrev = a.rev_col0 >= 0;
    if (a.rev_col0 >= 0 || a.rowvec) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int bb = rowv[r] / a.T, tt = rowv[r] - bb * a.T;
        bidx[r] = bb;
        if (a.rev_col0 >= 0) {
          const int L = a.rev_len ? a.rev_len[rowv[r] < row_limit ? bb : 0] : a.T;
          if (tt < L) orev[r] = bb * a.T + (L - 1 - tt);
        }
      }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + (wn * TN + tn) * 32 + l31;
      if (col >= v.N) continue;
      const float bia = ebia[tn];
      const bool rev = anyrev && col >= a.rev_col0;
      float* outc = a.out + v.coff + col;
      // row r of this tile starts ldo floats after row r-1 of the same quad, 8 rows per quad group: pointer walks instead of
      // a 64-bit multiply per element (only the reversed half of the BiGRU projection needs the general form)
      float* const obase = outc + (size_t)rbase * a.ldo;
      if constexpr (DUAL) {
        const float bia2 = ebia2[tn];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (rowv[r] >= row_limit) continue;
          const float H = fmaxf(acc[tm][tn][r] + bia, 0.f);
          const float Tg = taco_sigmoid(acc2[tm][tn][r] + bia2);
          const float xin = a.x[(size_t)rowv[r] * a.ldx + col];
          outc[(size_t)rowv[r] * a.ldo] = H * Tg + xin * (1.f - Tg);
          if (a.aux0) { a.aux0[(size_t)rowv[r] * a.ldo + col] = H; a.aux1[(size_t)rowv[r] * a.ldo + col] = Tg; }     // training tape
        }
      } else {
        const float sc = esc[tn], sh = esh[tn];
        if (simple_act && !extras) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (rowv[r] >= row_limit) continue;
            float val = acc[tm][tn][r] + bia;             // conv/dense + bias -> activation -> BatchNorm (modules.py:131)
            val = (relu && val < 0.f) ? 0.f : val;
            float* po = anyrev ? outc + (size_t)(rev ? orev[r] : rowv[r]) * a.ldo : obase + (size_t)((r & 3) + 8 * (r >> 2)) * a.ldo;
            *po = val * sc + sh;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (rowv[r] >= row_limit) continue;
            float val = taco_act(acc[tm][tn][r] + bia, a.act) * sc + sh;
            if (a.res) val += a.res[(size_t)rowv[r] * a.ldres + col];          // modules.py:62-69
            if (a.rowvec) val += a.rowvec[(size_t)bidx[r] * a.ldrv + col];
            outc[(size_t)(rev ? orev[r] : rowv[r]) * a.ldo] = val;
          }
        }
      }
    }
  }
  TRC(trci);
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
  const float* e1;   // GATES: precomputed x-part of the gates (BiGRU hoist) or null; CAND: x-part of candidate
  const float* e2;   // CAND: u [R, lde2]
  const float* e3;   // CAND: residual input (ResidualWrapper, tacotron.py:172) or null
  float* o0;         // LINEAR: out; GATES: r*h; CAND: new state h (may alias e0)
  float* o1;         // GATES: u; CAND: h' + residual (or null)
  float* o2;         // GATES: x-part of the candidate (columns >= 2H) or null; CAND: sequence output [R,T,ldo2]
                     // (BiGRU) or null; LINEAR: per-row non-zero flag (int*) or null
  const int* lengths;  // BiGRU sequence_length or null
  float* o3; int ldo3; // training tape (nullable): GATES: r; CAND: c (the tanh candidate)
  int ldx0, ldx1, K0, K, Kq, N, H, act;
  int lde0, lde1, lde2, lde3, ldo0, ldo1, ldo2;
  int step, T, dir, seq_coff;   // BiGRU scan position: dir 0 forward, 1 backward (reverse_sequence)
  int tile0;                    // first workgroup of this job
};
#define SK_MAXJOBS 4
struct SkArgs { int R; int njobs; SkJob j[SK_MAXJOBS]; };

#define SK_NW 8
#define SK_CH 4   // k16 groups per wave whose loads are issued together before any MFMA

// X[r][k..k+3] of the concatenation [x0 (K0 cols) | x1 (K-K0 cols)]; p0/p1 = row base pointers.
template <bool VEC>
__device__ __forceinline__ float4 taco_sk_load(const SkJob& jb, const float* p0, const float* p1, int k) {
  const float* p = (k < jb.K0) ? p0 + k : p1 + (k - jb.K0);
  if (VEC) return *reinterpret_cast<const float4*>(p);
  const int rem = (k < jb.K0) ? jb.K0 - k : jb.K - k;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rem > 0) v.x = p[0];
  if (rem > 1) v.y = p[1];
  if (rem > 2) v.z = p[2];
  if (rem > 3) v.w = p[3];
  return v;
}

// Latency structure (this kernel is a chain of round trips, not bandwidth): the job descriptor arrives
// in one batch of scalar loads (PIN); every epilogue operand (bias, h, u, x-part, residual, length) is
// requested first; then all weight/activation fragments of up to SK_CH k16-groups per wave are
// requested before the first MFMA -- one memory round trip covers everything; then MFMA, LDS reduction
// over the 8 K-slices, epilogue on the prefetched operands.  EPI and VEC are compile-time so the
// prologue is branch-free.  The BiGRU x-part is stored time-reversed for the backward direction by the
// hoisted GEMM, so its address does not depend on `lengths` (only the final store position does).
template <int RT, int EPI, bool VEC, bool MULTI>
__global__ __launch_bounds__(64 * SK_NW) void k_skinny(const SkArgs a) {
  __shared__ float red[SK_NW * RT * 256];
  constexpr int NE = (RT * 256 + 64 * SK_NW - 1) / (64 * SK_NW);   // outputs per thread
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // single-job launches (every decoder stage) read the descriptor at static kernarg offsets: ONE scalar
  // round trip instead of two (job select, then the selected descriptor)
  int ji = 0;
  if (MULTI) {
#pragma unroll
    for (int q = 1; q < SK_MAXJOBS; ++q)
      if (q < a.njobs && (int)blockIdx.x >= a.j[q].tile0) ji = q;
  }
  SkJob jb = MULTI ? a.j[ji] : a.j[0];
  PIN(jb.x0); PIN(jb.x1); PIN(jb.gather0); PIN(jb.wp); PIN(jb.bias); PIN(jb.e0); PIN(jb.e1); PIN(jb.e2); PIN(jb.e3);
  PIN(jb.o0); PIN(jb.o1); PIN(jb.o2); PIN(jb.lengths); PIN(jb.o3); PIN(jb.ldo3);
  PIN(jb.ldx0); PIN(jb.ldx1); PIN(jb.K0); PIN(jb.K); PIN(jb.Kq); PIN(jb.N); PIN(jb.H); PIN(jb.act);
  PIN(jb.lde0); PIN(jb.lde1); PIN(jb.lde2); PIN(jb.lde3); PIN(jb.ldo0); PIN(jb.ldo1); PIN(jb.ldo2);
  PIN(jb.step); PIN(jb.T); PIN(jb.dir); PIN(jb.seq_coff); PIN(jb.tile0);
  const int nt = blockIdx.x - jb.tile0;
  const int l15 = lane & 15, lq = lane >> 4;
  int R = a.R;
  PIN(R);
  const int row0 = blockIdx.y * (RT * 16);          // grid.y splits the batch rows into groups of RT*16

  // ---- (0) row base pointers.  A gathered row index (embedding lookups) is the only load another address depends on: it goes
  // first, so the wait the compiler puts in front of its use (unconditionally -- also when gather0 is null) finds nothing else in
  // flight.  Behind the epilogue operands that wait serialised two memory round trips per launch. ----
  f32x4 acc[RT];
  const float* xp0[RT]; const float* xp1[RT]; bool rok[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int r = row0 + rt * 16 + l15;
    rok[rt] = r < R;
    const int rr = rok[rt] ? r : 0;
    xp0[rt] = jb.x0 + (size_t)(jb.gather0 ? jb.gather0[rr] : rr) * jb.ldx0;
    xp1[rt] = jb.x1 ? jb.x1 + (size_t)rr * jb.ldx1 : jb.x0;
    taco_pin_v(xp0[rt]);                  // materialise the pointer HERE (otherwise the multiply, and the wait, sink to the loop)
  }
  __builtin_amdgcn_sched_barrier(0);     // nothing below may be hoisted above the (possible) gather round trip

  // ---- (1) epilogue operands ----
  float pb[NE], pe0[NE], pe1[NE], pe2[NE], pe3[NE];
  int pL[NE]; bool pvalid[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int idx = tid + e * 64 * SK_NW;
    const int rt = idx >> 8, w = idx & 255, reg = w >> 6, ln = w & 63;
    const int r = row0 + rt * 16 + (ln >> 4) * 4 + reg;   // C/D map of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
    const int n = nt * 16 + (ln & 15);
    const bool valid = (idx < RT * 256) && r < R && n < jb.N;
    pvalid[e] = valid;
    const int rr = valid ? r : 0, nn = valid ? n : 0;    // clamped so every load below is unconditional
    pb[e] = jb.bias ? jb.bias[nn] : 0.f;
    pe0[e] = pe1[e] = pe2[e] = pe3[e] = 0.f; pL[e] = jb.T;
    if (EPI == EPI_LINEAR) { if (jb.e0) pe0[e] = jb.e0[(size_t)rr * jb.lde0 + nn]; }
    if (EPI != EPI_LINEAR) {
      if (jb.lengths) pL[e] = jb.lengths[rr];
      const size_t xrow = (jb.T > 0) ? ((size_t)rr * jb.T + jb.step) : (size_t)rr;
      if (EPI == EPI_GRU_GATES) {
        if (jb.e1) pe1[e] = jb.e1[xrow * jb.lde1 + nn];
        pe0[e] = jb.e0[(size_t)rr * jb.lde0 + (nn < jb.H ? nn : 0)];
      } else {
        pe1[e] = jb.e1[xrow * jb.lde1 + nn];
        pe0[e] = jb.e0[(size_t)rr * jb.lde0 + nn];
        pe2[e] = jb.e2[(size_t)rr * jb.lde2 + nn];
        if (jb.e3) pe3[e] = jb.e3[(size_t)rr * jb.lde3 + nn];
      }
    }
  }

  // ---- (2) fragments, (3) MFMA ----
  const int ngroups = jb.Kq >> 2;
  const float* wbase = jb.wp + ((size_t)nt * jb.Kq * 16 + l15) * 4;
  for (int g0 = wave; g0 < ngroups; g0 += SK_NW * SK_CH) {
    float4 b[SK_CH], x[SK_CH][RT];
#pragma unroll
    for (int u = 0; u < SK_CH; ++u) {
      const int g = g0 + u * SK_NW;
      const bool gok = g < ngroups;
      const int kq = 4 * (gok ? g : g0) + lq;
      const int k = 4 * kq;
      const float4 bl = *reinterpret_cast<const float4*>(wbase + (size_t)kq * 64);
      b[u] = gok ? bl : make_float4(0.f, 0.f, 0.f, 0.f);
      const int kc = (k < jb.K) ? k : 0;                 // clamped: rows beyond K multiply zero weights
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float4 xl = taco_sk_load<VEC>(jb, xp0[rt], xp1[rt], kc);
        x[u][rt] = (rok[rt] && k < jb.K) ? xl : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < SK_CH; ++u)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u][rt].x, b[u].x, acc[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u][rt].y, b[u].y, acc[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u][rt].z, b[u].z, acc[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u][rt].w, b[u].w, acc[rt], 0, 0, 0);
      }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * RT + rt) * 256 + r * 64 + lane] = acc[rt][r];
  __syncthreads();

#pragma unroll
  for (int e = 0; e < NE; ++e) {
    if (!pvalid[e]) continue;
    const int idx = tid + e * 64 * SK_NW;
    const int rt = idx >> 8, w = idx & 255, reg = w >> 6, ln = w & 63;
    float s = pb[e];
#pragma unroll
    for (int wv = 0; wv < SK_NW; ++wv) s += red[(wv * RT + rt) * 256 + w];
    const int r = row0 + rt * 16 + (ln >> 4) * 4 + reg;
    const int n = nt * 16 + (ln & 15);
    if (EPI == EPI_LINEAR) {
      float y = taco_act(s, jb.act);
      if (jb.e0 && !(pe0[e] > 0.f)) y = 0.f;        // backward through a ReLU: masked by the forward activation (training path)
      jb.o0[(size_t)r * jb.ldo0 + n] = y;
      if (jb.o2 && y != 0.f) reinterpret_cast<int*>(jb.o2)[r] = 1;   // stop rule helpers.py:29
    } else if (EPI == EPI_GRU_GATES) {
      if (n < 2 * jb.H) {
        const float sg = taco_sigmoid(s + pe1[e]);
        if (n < jb.H) { jb.o0[(size_t)r * jb.ldo0 + n] = sg * pe0[e];     // r * h
          if (jb.o3) jb.o3[(size_t)r * jb.ldo3 + n] = sg; }
        else jb.o1[(size_t)r * jb.ldo1 + (n - jb.H)] = sg;                // u
      } else {
        jb.o2[(size_t)r * jb.ldo2 + (n - 2 * jb.H)] = s;                  // x . Wc_x (bias added with the h part)
      }
    } else {  // EPI_GRU_CAND
      // BiGRU time mapping (A.7): row active iff step < L; forward t = step, backward t = L-1-step.
      const bool active = (jb.T <= 0) || jb.step < pL[e];
      const int t = (jb.dir && active) ? (pL[e] - 1 - jb.step) : jb.step;
      const float c = tanhf(s + pe1[e]);
      const float hn = pe2[e] * pe0[e] + (1.f - pe2[e]) * c;            // u*h + (1-u)*c
      if (active) jb.o0[(size_t)r * jb.ldo0 + n] = hn;
      if (jb.o3) jb.o3[(size_t)r * jb.ldo3 + n] = c;
      if (jb.o1) jb.o1[(size_t)r * jb.ldo1 + n] = hn + pe3[e];
      if (jb.o2) jb.o2[((size_t)r * jb.T + t) * jb.ldo2 + jb.seq_coff + n] = active ? hn : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// row-parallel persistent kernels: streaming mat-vec primitive
// ------------------------------------------------------------------------------------------------
// The two sequential loops (BiGRU scans, decoder) are chains of small dependent mat-vecs.  Measured on
// MI355X (profiles/r01_ubench_*.txt): a dependent kernel launch costs >= 4.5 us once it has to fetch
// weights, an in-launch all-gather among 8 workgroups ~2.4 us + an LDS-resident slice's MFMA time, while
// ONE workgroup streams an L2-resident matrix at ~150 GB/s with the FMAs hidden.  Batch rows are
// independent, so a workgroup that owns R rows and streams every weight matrix from L2 each step needs NO
// cross-workgroup synchronisation: only __syncthreads.  Weights stay in TF layout [K, N] row-major:
// thread = 4 consecutive columns x one K-slice, loads are 16 bytes and coalesce over the row.
#define RP_NT 1024   // threads per workgroup: 16 waves x 8 sixteen-byte loads in flight = 128 KB outstanding (~150 GB/s)

// partial[(ks*R + r)*N + n] = sum over K-slice ks of x[r][k] * W[k][n].  x in LDS ([R][ldx]); W global.
// Returns KS (number of K-slices) for the reducer.  No barrier inside.  (Variants that prefetch the next
// stage's first rows across the barrier were measured: the 128-VGPR budget of a 1024-thread workgroup
// spills and the scan gets ~1.5x slower; 512-thread workgroups keep too few loads in flight.)
template <int R>
__device__ __forceinline__ int rp_matvec(const float* __restrict__ W, int K, int N, const float* x, int ldx,
                                         float* part, int tid) {
  const int NC = N >> 2;                                  // host guarantees N % 4 == 0 and N <= 4*RP_NT
  int KS = RP_NT / NC; if (KS > K) KS = K; if (KS < 1) KS = 1;
  const int kper = (K + KS - 1) / KS;
  const int cg = tid % NC, ks = tid / NC;
  if (ks < KS) {
    const int k0 = ks * kper, k1 = min(K, k0 + kper);
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* wp = reinterpret_cast<const float4*>(W) + cg;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) {
      const float4 w = wp[(size_t)k * NC];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float xv = x[r * ldx + k];
        acc[r].x = fmaf(xv, w.x, acc[r].x); acc[r].y = fmaf(xv, w.y, acc[r].y);
        acc[r].z = fmaf(xv, w.z, acc[r].z); acc[r].w = fmaf(xv, w.w, acc[r].w);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) *reinterpret_cast<float4*>(part + ((size_t)ks * R + r) * N + 4 * cg) = acc[r];
  }
  return KS;
}
__device__ __forceinline__ float rp_reduce(const float* part, int KS, int RN, int o) {
  float s = 0.f;
  for (int ks = 0; ks < KS; ++ks) s += part[(size_t)ks * RN + o];
  return s;
}

// ------------------------------------------------------------------------------------------------
// k_bigru_rows : the whole BiGRU scan (K8) as ONE launch, row-parallel
// ------------------------------------------------------------------------------------------------
// grid = 2 directions x ceil(B/R) workgroups; workgroup = R batch rows of one direction for all T steps.
// Per step: gates = h.Wg_h + xg (hoisted GEMM) -> r,u ; c = tanh((r(.)h).Wc_h + xc) ; h' = u*h + (1-u)*c  (A.6)
// with TF's sequence_length masking / reverse_sequence time mapping (A.7).  The hoisted projection stores the
// backward direction time-reversed, so step s reads row s for both directions.
struct BigruRArgs {
  const float* xproj;   // [B*T, 6H]
  const float* wg0; const float* wg1;   // h-rows of gates/kernel      [H, 2H] row-major, per direction
  const float* wc0; const float* wc1;   // h-rows of candidate/kernel  [H, H]
  const float* h0;      // [B, 2H] initial states (fw | bw) or null
  const int* lengths;   // [B] or null
  float* out;           // [B*T, 2H]
  float* gsave;         // training tape (nullable) [B*T, 6H]: (r | u | c) per direction at the TRUE time index of the step
  int B, T, H;
};

template <int R, bool TAPE>
__global__ __launch_bounds__(RP_NT) void k_bigru_rows(const BigruRArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  BigruRArgs a = a_in;
  PIN(a.xproj); PIN(a.wg0); PIN(a.wg1); PIN(a.wc0); PIN(a.wc1); PIN(a.h0); PIN(a.lengths); PIN(a.out); PIN(a.gsave);
  PIN(a.B); PIN(a.T); PIN(a.H);
  const int tid = threadIdx.x;
  const int ngrp = (a.B + R - 1) / R;
  const int d = blockIdx.x / ngrp, r0 = (blockIdx.x % ngrp) * R;
  const int B = a.B, T = a.T, H = a.H;
  const float* Wg = d ? a.wg1 : a.wg0;
  const float* Wc = d ? a.wc1 : a.wc0;
  float* hs = smem;                   // [R][H]
  float* rhs = hs + R * H;            // [R][H]
  float* us = rhs + R * H;            // [R][H]
  float* part = us + R * H;           // [KS][R][N] <= RP_NT*R*4 floats
  for (int i = tid; i < R * H; i += RP_NT) {
    const int r = i / H, c = i % H, b = r0 + r;
    hs[i] = (a.h0 && b < B) ? a.h0[(size_t)b * 2 * H + d * H + c] : 0.f;
  }
  __syncthreads();
  constexpr int NE1 = 2, NE2 = 1;     // epilogue outputs per thread (host guarantees R*2H <= 2*RP_NT)
  int Lg[NE1];                        // sequence length of the row of each gate output (tape only)
#pragma unroll
  for (int e = 0; e < NE1; ++e) {
    const int o = tid + e * RP_NT, b = r0 + o / (2 * H);
    Lg[e] = (TAPE && a.lengths && o < R * 2 * H && b < B) ? a.lengths[b] : T;
  }
  for (int s = 0; s < T; ++s) {
    // x-parts of this thread's outputs: requested now, consumed after the weight stream
    float xg[NE1], xc[NE2]; int Lr[NE2];
#pragma unroll
    for (int e = 0; e < NE2; ++e) Lr[e] = T;
#pragma unroll
    for (int e = 0; e < NE1; ++e) {
      const int o = tid + e * RP_NT; xg[e] = 0.f;
      if (o < R * 2 * H) { const int r = o / (2 * H), n = o % (2 * H), b = r0 + r;
        if (b < B) xg[e] = a.xproj[((size_t)b * T + s) * 6 * H + d * 3 * H + n]; }
    }
#pragma unroll
    for (int e = 0; e < NE2; ++e) {
      const int o = tid + e * RP_NT; xc[e] = 0.f;
      if (o < R * H) { const int r = o / H, n = o % H, b = r0 + r;
        if (b < B) { xc[e] = a.xproj[((size_t)b * T + s) * 6 * H + d * 3 * H + 2 * H + n]; if (a.lengths) Lr[e] = a.lengths[b]; } }
    }
    // ---- gates ----
    int KS = rp_matvec<R>(Wg, H, 2 * H, hs, H, part, tid);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE1; ++e) {
      const int o = tid + e * RP_NT;
      if (o < R * 2 * H) {
        const int r = o / (2 * H), n = o % (2 * H);
        const float sg = taco_sigmoid(rp_reduce(part, KS, R * 2 * H, o) + xg[e]);
        if (n < H) rhs[r * H + n] = sg * hs[r * H + n];
        else us[r * H + (n - H)] = sg;
        if (TAPE && s < Lg[e] && r0 + r < B) {
          const int t = d ? (Lg[e] - 1 - s) : s;
          a.gsave[((size_t)(r0 + r) * T + t) * 6 * H + d * 3 * H + n] = sg;
        }
      }
    }
    __syncthreads();
    // ---- candidate + state update ----
    KS = rp_matvec<R>(Wc, H, H, rhs, H, part, tid);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE2; ++e) {
      const int o = tid + e * RP_NT;
      if (o < R * H) {
        const int r = o / H, n = o % H, b = r0 + r;
        const float c = tanhf(rp_reduce(part, KS, R * H, o) + xc[e]);
        const float h = hs[o], u = us[o];
        const float hn = u * h + (1.f - u) * c;
        // BiGRU time mapping (A.7): row active iff s < L; forward t = s, backward t = L-1-s
        const bool active = s < Lr[e];
        const int t = (d && active) ? (Lr[e] - 1 - s) : s;
        if (active) hs[o] = hn;
        if (b < B) a.out[((size_t)b * T + t) * 2 * H + d * H + n] = active ? hn : 0.f;
        if (TAPE && active && b < B) a.gsave[((size_t)b * T + t) * 6 * H + d * 3 * H + 2 * H + n] = c;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// k_bigru_res : BiGRU scan with the recurrent weights RESIDENT on the CU
// ------------------------------------------------------------------------------------------------
// k_bigru_rows re-streams all recurrent weights of a direction (768 KB at H=256) from L2 every step, and one CU can take
// 64 B/clk (~150 GB/s): 5.1 us of the 7.1 us step.  A CU has 512 KB of vector registers and 160 KB of LDS -- together
// almost the whole matrix.  Here a 512-thread workgroup (2 waves/SIMD -> 256 VGPRs per thread) owns R batch rows of one
// direction; thread (j, q) owns hidden unit j and the K-slice q of size KS = H*H/512; of its KS rows of the recurrent
// kernels it keeps KR in registers (3 floats per row: r, u, c columns of unit j), KL in LDS, and streams the remaining
// KG = KS-KR-KL from L2 every step (H=256: 58 / 22 / 48 of 128 -> 288 KB per step instead of 768; H=128: all 32 in
// registers, nothing streamed).  No cross-workgroup synchronisation.  Weights: g2[k][j] = (Wg_h[k][j], Wg_h[k][H+j]),
// c1[k][j] = Wc_h[k][j] (built at finalize).
struct BigruSArgs {
  const float* xproj;                      // [B*T, 6H] hoisted input projection (backward direction time-reversed)
  const float2* g2_0; const float2* g2_1;  // [H, H] per direction
  const float* c1_0; const float* c1_1;    // [H, H]
  const float* h0; const int* lengths; float* out;
  float* gsave;                            // training tape (TAPE only): [B*T, 6H] (r | u | c) per direction at the true time index
  int B, T;
};
template <int H, int KR, int KL, int R, bool TAPE>
__global__ __launch_bounds__(512) void k_bigru_res(const BigruSArgs a_in) {
  constexpr int NT = 512, NQ = NT / H, KS = H / NQ, KG = KS - KR - KL, CH = 8;
  static_assert(KG >= 0 && KG % CH == 0 && KR % 4 == 0, "K-slice split must come in groups of 4");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  BigruSArgs a = a_in;
  PIN(a.xproj); PIN(a.g2_0); PIN(a.g2_1); PIN(a.c1_0); PIN(a.c1_1); PIN(a.h0); PIN(a.lengths); PIN(a.out); PIN(a.gsave); PIN(a.B); PIN(a.T);
  const int tid = threadIdx.x;
  const int ngrp = (a.B + R - 1) / R;
  const int d = blockIdx.x / ngrp, r0 = (blockIdx.x % ngrp) * R;
  const int B = a.B, T = a.T;
  const float2* G2 = d ? a.g2_1 : a.g2_0;
  const float* C1 = d ? a.c1_1 : a.c1_0;
  const int j = tid % H, q = tid / H, k0 = q * KS;
  float* hs = smem;                      // [R][H] state
  float* xs = hs + R * H;                // [R][H] r * h
  float* us = xs + R * H;                // [R][H] u
  float* part = us + R * H;              // [NQ][R][3][H] partial sums (r, u, c)
  float2* wl_g = reinterpret_cast<float2*>(part + NQ * R * 3 * H);   // [KL][NQ][H]
  float* wl_c = reinterpret_cast<float*>(wl_g + KL * NQ * H);        // [KL][NQ][H]
  // ---- resident weights ----
  float2 wg[KR > 0 ? KR : 1]; float wc[KR > 0 ? KR : 1];
#pragma unroll
  for (int i = 0; i < KR; ++i) { wg[i] = G2[(size_t)(k0 + i) * H + j]; wc[i] = C1[(size_t)(k0 + i) * H + j]; }
  for (int i = 0; i < KL; ++i) {
    wl_g[(i * NQ + q) * H + j] = G2[(size_t)(k0 + KR + i) * H + j];
    wl_c[(i * NQ + q) * H + j] = C1[(size_t)(k0 + KR + i) * H + j];
  }
  const float2* Gs = G2 + (size_t)(k0 + KR + KL) * H + j;     // streamed rows of this thread
  const float* Cs = C1 + (size_t)(k0 + KR + KL) * H + j;
  for (int i = tid; i < R * H; i += NT) {
    const int r = i / H, c = i % H, b = r0 + r;
    hs[i] = (a.h0 && b < B) ? a.h0[(size_t)b * 2 * H + d * H + c] : 0.f;
  }
  // epilogue items of this thread: gates: e1 = tid + i*NT < R*2H ; candidate: e2 = tid + i*NT < R*H
  constexpr int NE1 = (R * 2 * H + NT - 1) / NT, NE2 = (R * H + NT - 1) / NT;
  int Lr[NE2];
#pragma unroll
  for (int e = 0; e < NE2; ++e) { const int o = tid + e * NT, b = r0 + o / H; Lr[e] = (a.lengths && o < R * H && b < B) ? a.lengths[b] : T; }
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < T; ++s) {
    // x-parts of this thread's epilogue items: requested now, consumed after the mat-vec
    float xg[NE1], xc[NE2];
#pragma unroll
    for (int e = 0; e < NE1; ++e) {
      const int o = tid + e * NT; xg[e] = 0.f;
      if (o < R * 2 * H) { const int r = o / (2 * H), n = o % (2 * H), b = r0 + r; if (b < B) xg[e] = a.xproj[((size_t)b * T + s) * 6 * H + d * 3 * H + n]; }
    }
#pragma unroll
    for (int e = 0; e < NE2; ++e) {
      const int o = tid + e * NT; xc[e] = 0.f;
      if (o < R * H) { const int r = o / H, n = o % H, b = r0 + r; if (b < B) xc[e] = a.xproj[((size_t)b * T + s) * 6 * H + d * 3 * H + 2 * H + n]; }
    }
    // ---- gates: partial sums over this thread's K-slice ----
    float ar[R], au[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { ar[r] = 0.f; au[r] = 0.f; }
#pragma unroll
    for (int i = 0; i < KR; i += 4) {
      float4 h4[R];
#pragma unroll
      for (int r = 0; r < R; ++r) h4[r] = *reinterpret_cast<const float4*>(&hs[r * H + k0 + i]);
      asm volatile("" ::: "memory");          // keeps the scheduler from hoisting every group's LDS reads to the top (register pressure)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        ar[r] = fmaf(h4[r].x, wg[i].x, ar[r]); au[r] = fmaf(h4[r].x, wg[i].y, au[r]);
        ar[r] = fmaf(h4[r].y, wg[i + 1].x, ar[r]); au[r] = fmaf(h4[r].y, wg[i + 1].y, au[r]);
        ar[r] = fmaf(h4[r].z, wg[i + 2].x, ar[r]); au[r] = fmaf(h4[r].z, wg[i + 2].y, au[r]);
        ar[r] = fmaf(h4[r].w, wg[i + 3].x, ar[r]); au[r] = fmaf(h4[r].w, wg[i + 3].y, au[r]);
      }
    }
#pragma unroll 1
    for (int i = 0; i < KL; i += 2) {
      const float2 w0 = wl_g[(i * NQ + q) * H + j], w1 = wl_g[((i + 1) * NQ + q) * H + j];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float2 hv = *reinterpret_cast<const float2*>(&hs[r * H + k0 + KR + i]);
        ar[r] = fmaf(hv.x, w0.x, ar[r]); au[r] = fmaf(hv.x, w0.y, au[r]);
        ar[r] = fmaf(hv.y, w1.x, ar[r]); au[r] = fmaf(hv.y, w1.y, au[r]);
      }
    }
#pragma unroll 1
    for (int c0 = 0; c0 < KG; c0 += CH) {
      float2 cur[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) cur[u] = Gs[(size_t)(c0 + u) * H];
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int h4 = 0; h4 < CH; h4 += 4) {
        const float4 hv = *reinterpret_cast<const float4*>(&hs[r * H + k0 + KR + KL + c0 + h4]);
        ar[r] = fmaf(hv.x, cur[h4 + 0].x, ar[r]); au[r] = fmaf(hv.x, cur[h4 + 0].y, au[r]);
        ar[r] = fmaf(hv.y, cur[h4 + 1].x, ar[r]); au[r] = fmaf(hv.y, cur[h4 + 1].y, au[r]);
        ar[r] = fmaf(hv.z, cur[h4 + 2].x, ar[r]); au[r] = fmaf(hv.z, cur[h4 + 2].y, au[r]);
        ar[r] = fmaf(hv.w, cur[h4 + 3].x, ar[r]); au[r] = fmaf(hv.w, cur[h4 + 3].y, au[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { part[((q * R + r) * 3 + 0) * H + j] = ar[r]; part[((q * R + r) * 3 + 1) * H + j] = au[r]; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE1; ++e) {
      const int o = tid + e * NT;
      if (o < R * 2 * H) {
        const int r = o / (2 * H), n = o % (2 * H), g = n / H, nn = n % H;
        float sum = xg[e];
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) sum += part[((qq * R + r) * 3 + g) * H + nn];
        const float sgm = taco_sigmoid(sum);
        if (g == 0) xs[r * H + nn] = sgm * hs[r * H + nn]; else us[r * H + nn] = sgm;
        if (TAPE) {
          const int b = r0 + r, L = (a.lengths && b < B) ? a.lengths[b] : T;
          if (b < B && s < L) a.gsave[((size_t)b * T + (d ? L - 1 - s : s)) * 6 * H + d * 3 * H + n] = sgm;
        }
      }
    }
    __syncthreads();
    // ---- candidate ----
    float ac[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ac[r] = 0.f;
#pragma unroll
    for (int i = 0; i < KR; i += 4) {
      float4 x4[R];
#pragma unroll
      for (int r = 0; r < R; ++r) x4[r] = *reinterpret_cast<const float4*>(&xs[r * H + k0 + i]);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < R; ++r) {
        ac[r] = fmaf(x4[r].x, wc[i], ac[r]); ac[r] = fmaf(x4[r].y, wc[i + 1], ac[r]);
        ac[r] = fmaf(x4[r].z, wc[i + 2], ac[r]); ac[r] = fmaf(x4[r].w, wc[i + 3], ac[r]);
      }
    }
#pragma unroll 1
    for (int i = 0; i < KL; i += 2) {
      const float w0 = wl_c[(i * NQ + q) * H + j], w1 = wl_c[((i + 1) * NQ + q) * H + j];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float2 xv = *reinterpret_cast<const float2*>(&xs[r * H + k0 + KR + i]);
        ac[r] = fmaf(xv.x, w0, ac[r]); ac[r] = fmaf(xv.y, w1, ac[r]);
      }
    }
#pragma unroll 1
    for (int c0 = 0; c0 < KG; c0 += CH) {
      float cur[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) cur[u] = Cs[(size_t)(c0 + u) * H];
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int h4 = 0; h4 < CH; h4 += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(&xs[r * H + k0 + KR + KL + c0 + h4]);
        ac[r] = fmaf(xv.x, cur[h4 + 0], ac[r]); ac[r] = fmaf(xv.y, cur[h4 + 1], ac[r]); ac[r] = fmaf(xv.z, cur[h4 + 2], ac[r]); ac[r] = fmaf(xv.w, cur[h4 + 3], ac[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) part[((q * R + r) * 3 + 2) * H + j] = ac[r];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE2; ++e) {
      const int o = tid + e * NT;
      if (o < R * H) {
        const int r = o / H, n = o % H, b = r0 + r;
        float sum = xc[e];
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) sum += part[((qq * R + r) * 3 + 2) * H + n];
        const float c = tanhf(sum);
        const float h = hs[o], u = us[o];
        const float hn = u * h + (1.f - u) * c;
        const bool active = s < Lr[e];                       // A.7: row active iff s < L; forward t = s, backward t = L-1-s
        const int t = (d && active) ? (Lr[e] - 1 - s) : s;
        if (active) hs[o] = hn;
        if (b < B) a.out[((size_t)b * T + t) * 2 * H + d * H + n] = active ? hn : 0.f;
        if (TAPE && active && b < B) a.gsave[((size_t)b * T + t) * 6 * H + d * 3 * H + 2 * H + n] = c;
      }
    }
    __syncthreads();
  }
}

// k_bigru_resu : the same idea with UJ hidden units per thread and a K-slice UJ times shorter, so that one LDS read of the state
// vector feeds UJ times more FMAs (at H=256 the broadcast reads of h / r*h were the busiest pipe of k_bigru_res: 64 ds_read_b128
// per thread per step).  One batch row per workgroup.  Thread (jb, q): units jb + u*HJ (u < UJ, HJ = H/UJ), K-slice q of KS = H*HJ/512
// rows: KR in registers, KL in LDS, the rest streamed in chunks of CH rows.
template <int H, int UJ, int KR, int KL, int CH, bool TAPE>
__global__ __launch_bounds__(512) void k_bigru_resu(const BigruSArgs a_in) {
  constexpr int NT = 512, HJ = H / UJ, NQ = NT / HJ, KS = H / NQ, KG = KS - KR - KL;
  static_assert(KG >= 0 && KG % CH == 0 && KR % 4 == 0 && NQ * HJ == NT && NQ * KS == H, "bad split");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  BigruSArgs a = a_in;
  PIN(a.xproj); PIN(a.g2_0); PIN(a.g2_1); PIN(a.c1_0); PIN(a.c1_1); PIN(a.h0); PIN(a.lengths); PIN(a.out); PIN(a.gsave); PIN(a.B); PIN(a.T);
  const int tid = threadIdx.x;
  const int d = blockIdx.x / a.B, b = blockIdx.x % a.B;
  const int T = a.T;
  const float2* G2 = d ? a.g2_1 : a.g2_0;
  const float* C1 = d ? a.c1_1 : a.c1_0;
  const int jb = tid % HJ, q = tid / HJ, k0 = q * KS;
  float* hs = smem;                      // [H] state
  float* xs = hs + H;                    // [H] r * h
  float* us = xs + H;                    // [H] u
  float* part = us + H;                  // [NQ][3][H] partial sums (r, u, c)
  float2* wl_g = reinterpret_cast<float2*>(part + NQ * 3 * H);   // [KL][NQ][H]
  float* wl_c = reinterpret_cast<float*>(wl_g + KL * NQ * H);    // [KL][NQ][H]
  float2 wg[KR][UJ]; float wc[KR][UJ];
#pragma unroll
  for (int i = 0; i < KR; ++i)
#pragma unroll
    for (int u = 0; u < UJ; ++u) { wg[i][u] = G2[(size_t)(k0 + i) * H + jb + u * HJ]; wc[i][u] = C1[(size_t)(k0 + i) * H + jb + u * HJ]; }
  for (int i = 0; i < KL; ++i)
    for (int u = 0; u < UJ; ++u) {
      wl_g[(i * NQ + q) * H + jb + u * HJ] = G2[(size_t)(k0 + KR + i) * H + jb + u * HJ];
      wl_c[(i * NQ + q) * H + jb + u * HJ] = C1[(size_t)(k0 + KR + i) * H + jb + u * HJ];
    }
  const float2* Gs = G2 + (size_t)(k0 + KR + KL) * H + jb;
  const float* Cs = C1 + (size_t)(k0 + KR + KL) * H + jb;
  for (int i = tid; i < H; i += NT) hs[i] = a.h0 ? a.h0[(size_t)b * 2 * H + d * H + i] : 0.f;
  const int L = a.lengths ? a.lengths[b] : T;
  constexpr int NE1 = (2 * H + NT - 1) / NT;
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < T; ++s) {
    float xg[NE1], xc = 0.f;
#pragma unroll
    for (int e = 0; e < NE1; ++e) { const int o = tid + e * NT; xg[e] = (o < 2 * H) ? a.xproj[((size_t)b * T + s) * 6 * H + d * 3 * H + o] : 0.f; }
    if (tid < H) xc = a.xproj[((size_t)b * T + s) * 6 * H + d * 3 * H + 2 * H + tid];
    // ---- gates ----
    float ar[UJ], au[UJ];
#pragma unroll
    for (int u = 0; u < UJ; ++u) { ar[u] = 0.f; au[u] = 0.f; }
#pragma unroll
    for (int i = 0; i < KR; i += 4) {
      const float4 h4 = *reinterpret_cast<const float4*>(&hs[k0 + i]);
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        ar[u] = fmaf(h4.x, wg[i][u].x, ar[u]); au[u] = fmaf(h4.x, wg[i][u].y, au[u]);
        ar[u] = fmaf(h4.y, wg[i + 1][u].x, ar[u]); au[u] = fmaf(h4.y, wg[i + 1][u].y, au[u]);
        ar[u] = fmaf(h4.z, wg[i + 2][u].x, ar[u]); au[u] = fmaf(h4.z, wg[i + 2][u].y, au[u]);
        ar[u] = fmaf(h4.w, wg[i + 3][u].x, ar[u]); au[u] = fmaf(h4.w, wg[i + 3][u].y, au[u]);
      }
    }
#pragma unroll 1
    for (int i = 0; i < KL; ++i) {
      const float hv = hs[k0 + KR + i];
#pragma unroll
      for (int u = 0; u < UJ; ++u) { const float2 w = wl_g[(i * NQ + q) * H + jb + u * HJ]; ar[u] = fmaf(hv, w.x, ar[u]); au[u] = fmaf(hv, w.y, au[u]); }
    }
#pragma unroll 1
    for (int c0 = 0; c0 < KG; c0 += CH) {
      float2 cur[CH][UJ];
#pragma unroll
      for (int v = 0; v < CH; ++v)
#pragma unroll
        for (int u = 0; u < UJ; ++u) cur[v][u] = Gs[(size_t)(c0 + v) * H + u * HJ];
#pragma unroll
      for (int v = 0; v < CH; ++v) {
        const float hv = hs[k0 + KR + KL + c0 + v];
#pragma unroll
        for (int u = 0; u < UJ; ++u) { ar[u] = fmaf(hv, cur[v][u].x, ar[u]); au[u] = fmaf(hv, cur[v][u].y, au[u]); }
      }
    }
#pragma unroll
    for (int u = 0; u < UJ; ++u) { part[(q * 3 + 0) * H + jb + u * HJ] = ar[u]; part[(q * 3 + 1) * H + jb + u * HJ] = au[u]; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE1; ++e) {
      const int o = tid + e * NT;
      if (o < 2 * H) {
        const int g = o / H, nn = o % H;
        float sum = xg[e];
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) sum += part[(qq * 3 + g) * H + nn];
        const float sgm = taco_sigmoid(sum);
        if (g == 0) xs[nn] = sgm * hs[nn]; else us[nn] = sgm;
        if (TAPE && s < L) a.gsave[((size_t)b * T + (d ? L - 1 - s : s)) * 6 * H + d * 3 * H + o] = sgm;
      }
    }
    __syncthreads();
    // ---- candidate ----
    float ac[UJ];
#pragma unroll
    for (int u = 0; u < UJ; ++u) ac[u] = 0.f;
#pragma unroll
    for (int i = 0; i < KR; i += 4) {
      const float4 x4 = *reinterpret_cast<const float4*>(&xs[k0 + i]);
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        ac[u] = fmaf(x4.x, wc[i][u], ac[u]); ac[u] = fmaf(x4.y, wc[i + 1][u], ac[u]);
        ac[u] = fmaf(x4.z, wc[i + 2][u], ac[u]); ac[u] = fmaf(x4.w, wc[i + 3][u], ac[u]);
      }
    }
#pragma unroll 1
    for (int i = 0; i < KL; ++i) {
      const float xv = xs[k0 + KR + i];
#pragma unroll
      for (int u = 0; u < UJ; ++u) ac[u] = fmaf(xv, wl_c[(i * NQ + q) * H + jb + u * HJ], ac[u]);
    }
#pragma unroll 1
    for (int c0 = 0; c0 < KG; c0 += CH) {
      float cur[CH][UJ];
#pragma unroll
      for (int v = 0; v < CH; ++v)
#pragma unroll
        for (int u = 0; u < UJ; ++u) cur[v][u] = Cs[(size_t)(c0 + v) * H + u * HJ];
#pragma unroll
      for (int v = 0; v < CH; ++v) {
        const float xv = xs[k0 + KR + KL + c0 + v];
#pragma unroll
        for (int u = 0; u < UJ; ++u) ac[u] = fmaf(xv, cur[v][u], ac[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UJ; ++u) part[(q * 3 + 2) * H + jb + u * HJ] = ac[u];
    __syncthreads();
    if (tid < H) {
      float sum = xc;
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) sum += part[(qq * 3 + 2) * H + tid];
      const float c = tanhf(sum);
      const float h = hs[tid], u = us[tid];
      const float hn = u * h + (1.f - u) * c;
      const bool active = s < L;                           // A.7: row active iff s < L; forward t = s, backward t = L-1-s
      const int t = (d && active) ? (L - 1 - s) : s;
      if (active) hs[tid] = hn;
      a.out[((size_t)b * T + t) * 2 * H + d * H + tid] = active ? hn : 0.f;
      if (TAPE && active) a.gsave[((size_t)b * T + t) * 6 * H + d * 3 * H + 2 * H + tid] = c;
    }
    __syncthreads();
  }
}

// k_bigru_resw : k_bigru_resu (H = 256, 4 units per thread) with the per-step exchanges made wave-local.  Thread (jb, q) works on
// K-slice q, and q is the wave index -- so the 32 state values a wave multiplies by are exactly the 32 units whose gates and
// update that wave can finish itself: after the partial sums are out (one barrier) wave q reduces r and u of units
// 32q..32q+31 on its 64 lanes, keeps u in registers, writes r*h where only it will read it; after the candidate partials (second
// barrier) lanes 32..63 finish those units and write the new state where, again, only this wave reads it.  Two barriers per
// step instead of four, same arithmetic in the same order (bit-identical to k_bigru_resu).
template <int KR, int KL, int CH>
__global__ __launch_bounds__(512) void k_bigru_resw(const BigruSArgs a_in) {
  constexpr int H = 256, UJ = 4; constexpr bool TAPE = false;
  constexpr int NT = 512, HJ = H / UJ, NQ = NT / HJ, KS = H / NQ, KG = KS - KR - KL;
  static_assert(HJ == 64 && 2 * KS == 64, "a wave = one K-slice; its 64 lanes = the r and u gates of the slice's 32 units");
  static_assert(KG >= 0 && KG % CH == 0 && KR % 4 == 0 && NQ * HJ == NT && NQ * KS == H, "bad split");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  BigruSArgs a = a_in;
  PIN(a.xproj); PIN(a.g2_0); PIN(a.g2_1); PIN(a.c1_0); PIN(a.c1_1); PIN(a.h0); PIN(a.lengths); PIN(a.out); PIN(a.gsave); PIN(a.B); PIN(a.T);
  const int tid = threadIdx.x;
  const int d = blockIdx.x / a.B, b = blockIdx.x % a.B;
  const int T = a.T;
  const float2* G2 = d ? a.g2_1 : a.g2_0;
  const float* C1 = d ? a.c1_1 : a.c1_0;
  const int jb = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6), k0 = q * KS;   // q is the wave index: uniform, lives in an SGPR
  float* hs = smem;                      // [H] state
  float* xs = hs + H;                    // [H] r * h
  float* us = xs + H;                    // [H] u
  float* part = us + H;                  // [NQ][3][H] partial sums (r, u, c)
  float2* wl_g = reinterpret_cast<float2*>(part + NQ * 3 * H);   // [KL][NQ][H]
  float* wl_c = reinterpret_cast<float*>(wl_g + KL * NQ * H);    // [KL][NQ][H]
  float2 wg[KR][UJ]; float wc[KR][UJ];
#pragma unroll
  for (int i = 0; i < KR; ++i)
#pragma unroll
    for (int u = 0; u < UJ; ++u) { wg[i][u] = G2[(size_t)(k0 + i) * H + jb * UJ + u]; wc[i][u] = C1[(size_t)(k0 + i) * H + jb * UJ + u]; }
  for (int i = 0; i < KL; ++i)
    for (int u = 0; u < UJ; ++u) {
      wl_g[(i * NQ + q) * H + jb + u * HJ] = G2[(size_t)(k0 + KR + i) * H + jb * UJ + u];
      wl_c[(i * NQ + q) * H + jb + u * HJ] = C1[(size_t)(k0 + KR + i) * H + jb * UJ + u];
    }
  const float2* Gs = G2 + (size_t)(k0 + KR + KL) * H + jb * UJ;     // packs with permuted columns: the thread's UJ units are adjacent
  const float* Cs = C1 + (size_t)(k0 + KR + KL) * H + jb * UJ;
  for (int i = tid; i < H; i += NT) hs[i] = a.h0 ? a.h0[(size_t)b * 2 * H + d * H + i] : 0.f;
  const int L = a.lengths ? a.lengths[b] : T;
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < T; ++s) {
    // the wave's own units: lane l finishes gate g = l>>5 (r, u) of unit nn = k0 + (l & 31); lanes 32..63 also the candidate
    const int lane = tid & 63, gsel = lane >> 5, nn = k0 + (lane & 31);
    const float* xrow = a.xproj + ((size_t)b * T + s) * 6 * H + d * 3 * H;
    const float xg = xrow[gsel * H + nn];
    const float xc = gsel ? xrow[2 * H + nn] : 0.f;
    // ---- gates ----
    float ar[UJ], au[UJ];
#pragma unroll
    for (int u = 0; u < UJ; ++u) { ar[u] = 0.f; au[u] = 0.f; }
#pragma unroll
    for (int i = 0; i < KR; i += 4) {
      const float4 h4 = *reinterpret_cast<const float4*>(&hs[k0 + i]);
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        ar[u] = fmaf(h4.x, wg[i][u].x, ar[u]); au[u] = fmaf(h4.x, wg[i][u].y, au[u]);
        ar[u] = fmaf(h4.y, wg[i + 1][u].x, ar[u]); au[u] = fmaf(h4.y, wg[i + 1][u].y, au[u]);
        ar[u] = fmaf(h4.z, wg[i + 2][u].x, ar[u]); au[u] = fmaf(h4.z, wg[i + 2][u].y, au[u]);
        ar[u] = fmaf(h4.w, wg[i + 3][u].x, ar[u]); au[u] = fmaf(h4.w, wg[i + 3][u].y, au[u]);
      }
    }
#pragma unroll 1
    for (int i = 0; i < KL; ++i) {
      const float hv = hs[k0 + KR + i];
#pragma unroll
      for (int u = 0; u < UJ; ++u) { const float2 w = wl_g[(i * NQ + q) * H + jb + u * HJ]; ar[u] = fmaf(hv, w.x, ar[u]); au[u] = fmaf(hv, w.y, au[u]); }
    }
#pragma unroll 1
    for (int c0 = 0; c0 < KG; c0 += CH) {
      float2 cur[CH][UJ];
#pragma unroll
      for (int v = 0; v < CH; ++v)
#pragma unroll
        for (int u = 0; u < UJ; ++u) cur[v][u] = Gs[(size_t)(c0 + v) * H + u];
#pragma unroll
      for (int v = 0; v < CH; ++v) {
        const float hv = hs[k0 + KR + KL + c0 + v];
#pragma unroll
        for (int u = 0; u < UJ; ++u) { ar[u] = fmaf(hv, cur[v][u].x, ar[u]); au[u] = fmaf(hv, cur[v][u].y, au[u]); }
      }
    }
#pragma unroll
    for (int u = 0; u < UJ; ++u) { part[(q * 3 + 0) * H + jb + u * HJ] = ar[u]; part[(q * 3 + 1) * H + jb + u * HJ] = au[u]; }
    __syncthreads();
    // gate epilogue, wave-local: the 8 partial sums of the slice's own 32 units.  r*h goes to this wave's part of xs (read back
    // by this wave only: LDS operations of one wave execute in order), u stays in a register of the lane that will finish the unit
    float ureg;
    {
      float sum = xg;
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) sum += part[(qq * 3 + gsel) * H + nn];
      const float sgm = taco_sigmoid(sum);
      ureg = sgm;
      if (!gsel) xs[nn] = sgm * hs[nn];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- candidate ----
    float ac[UJ];
#pragma unroll
    for (int u = 0; u < UJ; ++u) ac[u] = 0.f;
#pragma unroll
    for (int i = 0; i < KR; i += 4) {
      const float4 x4 = *reinterpret_cast<const float4*>(&xs[k0 + i]);
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        ac[u] = fmaf(x4.x, wc[i][u], ac[u]); ac[u] = fmaf(x4.y, wc[i + 1][u], ac[u]);
        ac[u] = fmaf(x4.z, wc[i + 2][u], ac[u]); ac[u] = fmaf(x4.w, wc[i + 3][u], ac[u]);
      }
    }
#pragma unroll 1
    for (int i = 0; i < KL; ++i) {
      const float xv = xs[k0 + KR + i];
#pragma unroll
      for (int u = 0; u < UJ; ++u) ac[u] = fmaf(xv, wl_c[(i * NQ + q) * H + jb + u * HJ], ac[u]);
    }
#pragma unroll 1
    for (int c0 = 0; c0 < KG; c0 += CH) {
      float cur[CH][UJ];
#pragma unroll
      for (int v = 0; v < CH; ++v)
#pragma unroll
        for (int u = 0; u < UJ; ++u) cur[v][u] = Cs[(size_t)(c0 + v) * H + u];
#pragma unroll
      for (int v = 0; v < CH; ++v) {
        const float xv = xs[k0 + KR + KL + c0 + v];
#pragma unroll
        for (int u = 0; u < UJ; ++u) ac[u] = fmaf(xv, cur[v][u], ac[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UJ; ++u) part[(q * 3 + 2) * H + jb + u * HJ] = ac[u];
    __syncthreads();
    if (gsel) {                                              // lanes 32..63: the unit whose u they hold
      float sum = xc;
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) sum += part[(qq * 3 + 2) * H + nn];
      const float c = tanhf(sum);
      const float h = hs[nn];
      const float hn = ureg * h + (1.f - ureg) * c;
      const bool active = s < L;                           // A.7: row active iff s < L; forward t = s, backward t = L-1-s
      const int t = (d && active) ? (L - 1 - s) : s;
      if (active) hs[nn] = hn;                             // read next step by this wave only (its K-slice = its own units)
      a.out[((size_t)b * T + t) * 2 * H + d * H + nn] = active ? hn : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// k_attention : one workgroup per batch row
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
  const float* q;        // [B, A] processed query (h_att . W_q), or null when hq/wq are given
  const float* hq;       // [B, As] attention-GRU output: the query mat-vec is done here (saves a launch)
  const float* wq;       // [As, A] query_layer/kernel, TF layout
  const float* keys;     // [B, T_in, A]
  const float* values;   // [B, T_in, D]
  const float* v;        // [A] attention_v (bah_norm: g*v/|v| folded at finalize)
  const float* battn;    // [A] attention_b (bah_norm) or null
  const float* score_bias;  // [1] (bah_mon) or null
  const float* manual;   // [B, n_steps, T_in] or null (rnn_wrappers.py:313-317)
  float* align;          // [B, T_in] state: previous alignments in, new alignments out
  float* hist;           // [B, T_in, n_steps] or null (tacotron.py:238-239 layout)
  float* ctx;            // [B, D]
  int T_in, A, D, type, step, n_steps, As, ldctx;   // ldctx: row stride of ctx (D, or D + speaker columns)
  const float* align_prev;  // training tape: previous alignments read from here (align is then write-only); null = align
  int ldalign, ldhq;        // row strides of align/align_prev and hq (0 = T_in / As)
  float* q_out; float* e_out; int ldq_out, lde_out;   // training tape (nullable): processed query [B, A] and raw scores [B, T_in] of this step
};

#define ATT_NW 16    // waves per workgroup (1024 threads: more key/value loads and tanh evaluations in flight)
#define ATT_MAXT 2048
#define ATT_JU 2     // score iterations (of 4*ATT_NW encoder positions each) whose key loads are issued together
#define ATT_VU 8     // value rows per wave whose loads are issued together (before the normaliser)
#define ATT_QU 16    // rows of W_q per thread whose loads are issued together (query mat-vec)

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
// tanh(x) = 1 - 2/(1 + e^{2x}) on v_exp_f32 / v_rcp_f32 (each ~1 ulp): abs error < 4e-7, exact limits +-1.
__device__ __forceinline__ float taco_tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // e^{2x}
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

// k_bigru_quad : the H = 128 scan (encoder BiGRU, modules.py:82-96; A.6, A.7) with the K split kept inside a QUAD of lanes.
// One workgroup = one (direction, batch row) chain, recurrent weights resident in registers as in k_bigru_res -- but thread
// (unit j = 16 wave + lane / 4, K-slice q = lane & 3) keeps the four slices of a unit in four ADJACENT lanes: the partial sums
// meet by two quad_perm DPP adds instead of a round trip through LDS and a barrier, every lane of the quad then holds the unit's
// pre-activation and state (h stays in a register), and a step has two workgroup barriers (r*h visible, h visible) instead of four.
// The transcendentals are the exp2 / rcp forms of the persistent kernels, the step's x-projection values are requested two steps
// ahead (registers), the state vectors are padded so the four K-slices of a broadcast read fall on different banks.
// Measured at C2 (B = 32, T = 128): k_bigru_res 0.93 us per step.
__device__ __forceinline__ float taco_quadsum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ float taco_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// TAPE: the training forward -- the gates r, u and the candidate c of the active steps at their true time (a.gsave [B*T, 6H]), for the backward scan
template <bool TAPE = false>
__global__ __launch_bounds__(512) void k_bigru_quad(const BigruSArgs a_in) {
  constexpr int H = 128, KS = 32, SP = 36;          // K-slice q of the state lives at floats [q * SP, q * SP + 32)
  __shared__ __attribute__((aligned(16))) float hs[4 * SP];
  __shared__ __attribute__((aligned(16))) float xs[4 * SP];
  BigruSArgs a = a_in;
  PIN(a.xproj); PIN(a.g2_0); PIN(a.g2_1); PIN(a.c1_0); PIN(a.c1_1); PIN(a.h0); PIN(a.lengths); PIN(a.out); PIN(a.B); PIN(a.T);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane & 3, j = wave * 16 + (lane >> 2), k0 = q * KS;
  const int d = blockIdx.x / a.B, b = blockIdx.x - d * a.B;
  const int T = a.T;
  const float2* G2 = d ? a.g2_1 : a.g2_0;
  const float* C1 = d ? a.c1_1 : a.c1_0;
  float2 wg[KS]; float wc[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) { wg[i] = G2[(size_t)(k0 + i) * H + j]; wc[i] = C1[(size_t)(k0 + i) * H + j]; }
  const int L = a.lengths ? a.lengths[b] : T;
  float hj = a.h0 ? a.h0[(size_t)b * 2 * H + d * H + j] : 0.f;
  const int pos = (j >> 5) * SP + (j & 31);        // where unit j sits in the padded state vectors
  if (q == 0) hs[pos] = hj;
  const float* xp = a.xproj + (size_t)b * T * 6 * H + d * 3 * H + j;
  float* op = a.out + (size_t)b * T * 2 * H + d * H + j;
  float xa[3], xb[3];
  auto xload = [&](int s, float (&x)[3]) {
    const float* p = xp + (size_t)min(s, T - 1) * 6 * H;       // (past the end: fetched again, never used)
    x[0] = p[0]; x[1] = p[H]; x[2] = p[2 * H];
  };
  xload(0, xa); xload(1, xb);
  __syncthreads();
  auto step = [&](int s, const float (&x)[3]) {
    // ---- gates: r, u of unit j over K-slice q, summed over the quad ----
    taco_f32x2 g2 = {0.f, 0.f}, g3 = {0.f, 0.f};       // (r, u) partial sums: one v_pk_fma_f32 per state element, two independent chains
#pragma unroll
    for (int i = 0; i < KS; i += 4) {
      const float4 h4 = *reinterpret_cast<const float4*>(&hs[q * SP + i]);
      g2 = __builtin_elementwise_fma((taco_f32x2){h4.x, h4.x}, (taco_f32x2){wg[i].x, wg[i].y}, g2);
      g3 = __builtin_elementwise_fma((taco_f32x2){h4.y, h4.y}, (taco_f32x2){wg[i + 1].x, wg[i + 1].y}, g3);
      g2 = __builtin_elementwise_fma((taco_f32x2){h4.z, h4.z}, (taco_f32x2){wg[i + 2].x, wg[i + 2].y}, g2);
      g3 = __builtin_elementwise_fma((taco_f32x2){h4.w, h4.w}, (taco_f32x2){wg[i + 3].x, wg[i + 3].y}, g3);
    }
    const float ar = taco_quadsum(g2.x + g3.x), au = taco_quadsum(g2.y + g3.y);
    const float r = taco_sigmoid_fast(ar + x[0]), u = taco_sigmoid_fast(au + x[1]);
    if (q == 0) xs[pos] = r * hj;
    __syncthreads();
    // ---- candidate ----
    taco_f32x2 c2 = {0.f, 0.f}, c3 = {0.f, 0.f};       // even / odd K elements in two chains, summed at the end
#pragma unroll
    for (int i = 0; i < KS; i += 4) {
      const float4 x4 = *reinterpret_cast<const float4*>(&xs[q * SP + i]);
      c2 = __builtin_elementwise_fma((taco_f32x2){x4.x, x4.y}, (taco_f32x2){wc[i], wc[i + 1]}, c2);
      c3 = __builtin_elementwise_fma((taco_f32x2){x4.z, x4.w}, (taco_f32x2){wc[i + 2], wc[i + 3]}, c3);
    }
    const float ac = taco_quadsum((c2.x + c2.y) + (c3.x + c3.y));
    const float c = taco_tanh_fast(ac + x[2]);
    const float hn = u * hj + (1.f - u) * c;
    const bool active = s < L;                          // A.7: row active iff s < L; forward t = s, backward t = L-1-s
    const int t = (d && active) ? (L - 1 - s) : s;
    if (active) hj = hn;
    if (q == 0) { hs[pos] = hj; op[(size_t)t * 2 * H] = active ? hn : 0.f; }
    if (TAPE && q == 0 && active) { float* gs = a.gsave + ((size_t)b * T + t) * 6 * H + d * 3 * H + j; gs[0] = r; gs[H] = u; gs[2 * H] = c; }
    __syncthreads();
  };
  int s = 0;
#pragma unroll 1
  for (; s + 1 < T; s += 2) {
    step(s, xa); xload(s + 2, xa);
    step(s + 1, xb); xload(s + 3, xb);
  }
  if (s < T) step(s, xa);
}

// One workgroup (8 waves) per batch row.  Everything a step needs from HBM/L2 -- the row's keys
// [T_in, A] and values [T_in, D] -- is requested up front in two bursts (scores burst; values burst
// before the serial normaliser), so a step costs ~one memory round trip plus the tanh/exp math.
// The attention of ONE batch row b by one 16-wave workgroup.  LDS scratch: sc/tmp/tmp2 [T_in], cred [ATT_NW*256].
// hq_row: the row's attention-GRU output (global or LDS) when the query mat-vec is done here; al: the row's
// alignment state [T_in] (in: previous, out: new; global or LDS); ctx_out: [D] (global or LDS).
__device__ __forceinline__ void att_core(const AttnArgs& a, int b, float* sc, float* tmp, float* tmp2, float* cred,
                                         const float* hq_row, float* al, float* ctx_out, const float* al_prev = nullptr) {
  if (!al_prev) al_prev = al;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T_in;
  const float* vrow = a.values + (size_t)b * T * a.D;

  // values burst for the first 256 output channels: wave w owns positions j = w + 8*i.  Requested after the scores are
  // done (the serial normaliser covers its latency); the KEYS burst is what goes out first, before the query mat-vec.
  float4 vpre[ATT_VU];
  auto load_values = [&]() {
    const int d = lane * 4;
#pragma unroll
    for (int i = 0; i < ATT_VU; ++i) {
      const int j = wave + ATT_NW * i;
      vpre[i] = (j < T && d < a.D) ? *reinterpret_cast<const float4*>(vrow + (size_t)j * a.D + d)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  const int l16 = lane & 15, grp = lane >> 4;
  const float* krow = a.keys + (size_t)b * T * a.A;
  float4 k4[ATT_JU][4];
  auto load_keys = [&](int j0, int c0) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int c = c0 + l16 * 4 + 64 * m;
#pragma unroll
      for (int u = 0; u < ATT_JU; ++u) {
        const int j = j0 + 4 * ATT_NW * u + wave * 4 + grp;
        k4[u][m] = (c < a.A && j < T) ? *reinterpret_cast<const float4*>(krow + (size_t)j * a.A + c)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  if (!a.manual) load_keys(0, 0);
  __builtin_amdgcn_sched_barrier(0);

  if (a.manual) {
    load_values();
    for (int j = tid; j < T; j += 64 * ATT_NW) sc[j] = a.manual[((size_t)b * a.n_steps + a.step) * T + j];
    __syncthreads();
  } else {
    // scores: e[j] = sum_a v[a] * tanh(keys[b,j,a] + q[b,a] (+ b[a]))   (_bahdanau_score, A.9)
    // 16 lanes per encoder position; lane covers channels c = (lane&15)*4 + 64*m.
    const float* qb = a.q ? a.q + (size_t)b * a.A : cred;
    if (!a.q) {
      // q[b,:] = h_att[b,:] . W_q : 4 columns per thread, K split over thread groups, reduced through LDS
      const int NC = a.A >> 2;
      int KS = (64 * ATT_NW) / NC; if (KS > a.As) KS = a.As; if (KS * a.A > ATT_MAXT) KS = ATT_MAXT / a.A; if (KS < 1) KS = 1;
      const int kper = (a.As + KS - 1) / KS, cg = tid % NC, ks = tid / NC;
      float* qpart = tmp;                                     // [KS][A] <= ATT_MAXT floats (tmp holds >= ATT_MAXT)
      if (ks < KS) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* hb = hq_row;
        const float4* wp = reinterpret_cast<const float4*>(a.wq) + cg;
        const int k1 = min(a.As, (ks + 1) * kper);
        if ((NC & 15) == 0) {
          // all of a thread's K-slice (16 rows at the reference widths) is requested before the first FMA: one round trip.
          // The 16 lanes of a group share ks, so each loads ONE element of h and the group passes them round by lane shuffles.
          for (int k = ks * kper; k < k1; k += ATT_QU) {
            float4 w[ATT_QU];
            const float hv = hb[min(k + (lane & 15), k1 - 1)];
#pragma unroll
            for (int u = 0; u < ATT_QU; ++u) w[u] = wp[(size_t)min(k + u, k1 - 1) * NC];
#pragma unroll
            for (int u = 0; u < ATT_QU; ++u) {
              float x1 = __shfl(hv, (lane & 48) | u, 64);
              x1 = (k + u < k1) ? x1 : 0.f;
              acc.x = fmaf(x1, w[u].x, acc.x); acc.y = fmaf(x1, w[u].y, acc.y); acc.z = fmaf(x1, w[u].z, acc.z); acc.w = fmaf(x1, w[u].w, acc.w);
            }
          }
        } else {
#pragma unroll 8
          for (int k = ks * kper; k < k1; ++k) {
            const float4 w = wp[(size_t)k * NC]; const float xv = hb[k];
            acc.x = fmaf(xv, w.x, acc.x); acc.y = fmaf(xv, w.y, acc.y); acc.z = fmaf(xv, w.z, acc.z); acc.w = fmaf(xv, w.w, acc.w);
          }
        }
        *reinterpret_cast<float4*>(qpart + (size_t)ks * a.A + 4 * cg) = acc;
      }
      __syncthreads();
      for (int n = tid; n < a.A; n += 64 * ATT_NW) {
        float sum = 0.f;
        for (int k2 = 0; k2 < KS; ++k2) sum += qpart[(size_t)k2 * a.A + n];
        cred[n] = sum;                                        // cred is free until the context phase
        if (a.q_out) a.q_out[(size_t)b * a.ldq_out + n] = sum;
      }
      __syncthreads();
    }
    for (int j0 = 0; j0 < T; j0 += 4 * ATT_NW * ATT_JU) {
      float part[ATT_JU];
#pragma unroll
      for (int u = 0; u < ATT_JU; ++u) part[u] = 0.f;
      for (int c0 = 0; c0 < a.A; c0 += 256) {
        if (j0 | c0) load_keys(j0, c0);
        float4 q4[4], v4[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int c = c0 + l16 * 4 + 64 * m;
          const bool ok = c < a.A;
          q4[m] = ok ? *reinterpret_cast<const float4*>(qb + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          v4[m] = ok ? *reinterpret_cast<const float4*>(a.v + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok && a.battn) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.battn + c);
            q4[m].x += b4.x; q4[m].y += b4.y; q4[m].z += b4.z; q4[m].w += b4.w;
          }
        }
#pragma unroll
        for (int u = 0; u < ATT_JU; ++u)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            part[u] += v4[m].x * taco_tanh_fast(k4[u][m].x + q4[m].x) + v4[m].y * taco_tanh_fast(k4[u][m].y + q4[m].y) +
                       v4[m].z * taco_tanh_fast(k4[u][m].z + q4[m].z) + v4[m].w * taco_tanh_fast(k4[u][m].w + q4[m].w);
          }
      }
#pragma unroll
      for (int u = 0; u < ATT_JU; ++u) {
        float p = part[u];
        p += __shfl_xor(p, 8, 64); p += __shfl_xor(p, 4, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 1, 64);
        const int j = j0 + 4 * ATT_NW * u + wave * 4 + grp;
        if (l16 == 0 && j < T) { sc[j] = p; if (a.e_out) a.e_out[(size_t)b * a.lde_out + j] = p; }
      }
    }
    load_values();
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
          run2 += al_prev[j] / fminf(fmaxf(cp, 1e-10f), 1.f);
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
      int i = 0;
      if (d0 == 0) {   // the prefetched burst
#pragma unroll
        for (; i < ATT_VU; ++i) {
          const int j = wave + ATT_NW * i;
          const float w = (j < T) ? sc[j] : 0.f;
          c4.x += w * vpre[i].x; c4.y += w * vpre[i].y; c4.z += w * vpre[i].z; c4.w += w * vpre[i].w;
        }
      }
      for (int j = wave + ATT_NW * i; j < T; j += ATT_NW) {
        const float w = sc[j];
        const float4 v4 = *reinterpret_cast<const float4*>(vrow + (size_t)j * a.D + d);
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
      ctx_out[d0 + tid] = s;
    }
  }
}

__global__ __launch_bounds__(64 * ATT_NW) void k_attention(const AttnArgs a_in) {
  AttnArgs a = a_in;
  PIN(a.q); PIN(a.hq); PIN(a.wq); PIN(a.As); PIN(a.keys); PIN(a.values); PIN(a.v); PIN(a.battn); PIN(a.score_bias); PIN(a.manual); PIN(a.align);
  PIN(a.hist); PIN(a.ctx); PIN(a.T_in); PIN(a.A); PIN(a.D); PIN(a.type); PIN(a.step); PIN(a.n_steps); PIN(a.ldctx);
  PIN(a.align_prev); PIN(a.ldalign); PIN(a.ldhq); PIN(a.q_out); PIN(a.e_out); PIN(a.ldq_out); PIN(a.lde_out);
  __shared__ float sc[ATT_MAXT];     // scores -> alignments
  __shared__ float tmp[ATT_MAXT];
  __shared__ float tmp2[ATT_MAXT];
  __shared__ __attribute__((aligned(16))) float cred[ATT_NW * 256];
  const int b = blockIdx.x;
  const int lda = a.ldalign ? a.ldalign : a.T_in, ldh = a.ldhq ? a.ldhq : a.As;
  att_core(a, b, sc, tmp, tmp2, cred, a.hq ? a.hq + (size_t)b * ldh : nullptr, a.align + (size_t)b * lda,
           a.ctx + (size_t)b * a.ldctx, a.align_prev ? a.align_prev + (size_t)b * lda : nullptr);
}

// ------------------------------------------------------------------------------------------------
// split attention for small batches / long inputs (C5: B=8, T_in=512)
// ------------------------------------------------------------------------------------------------
// k_attention gives every batch row ONE workgroup; with 8 rows and 1 MB of keys+values per row that is 8 CUs pulling 64 B/clk
// each.  Two launches instead: k_att_scores (grid B x S) -- every workgroup redoes the query mat-vec and scores its slice of
// the encoder positions; k_att_context (grid B x S) -- every workgroup redoes the (cheap) normaliser over the whole row and
// computes its slice of the context channels, slice 0 also writes the alignments.  No cross-workgroup exchange inside a launch.
#define ATS_NW 8
__global__ __launch_bounds__(64 * ATS_NW) void k_att_scores(const AttnArgs a_in, float* escr, int S) {
  AttnArgs a = a_in;
  PIN(a.hq); PIN(a.wq); PIN(a.As); PIN(a.keys); PIN(a.v); PIN(a.battn); PIN(a.T_in); PIN(a.A);
  __shared__ __attribute__((aligned(16))) float qs[1024];
  __shared__ __attribute__((aligned(16))) float part[ATS_NW * 1024 / 2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, sl = blockIdx.y, T = a.T_in, A = a.A, As = a.As;
  const int ldh = a.ldhq ? a.ldhq : As;
  const float* hb = a.hq + (size_t)b * ldh;
  {  // q = h_att . W_q (4 columns per thread, K split over thread groups)
    const int NC = A >> 2;
    int KS = (64 * ATS_NW) / NC; if (KS > As) KS = As; if (KS * A > ATS_NW * 512) KS = (ATS_NW * 512) / A; if (KS < 1) KS = 1;
    const int kper = (As + KS - 1) / KS, cg = tid % NC, ks = tid / NC;
    if (ks < KS) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4* wp = reinterpret_cast<const float4*>(a.wq) + cg;
      const int k1 = min(As, (ks + 1) * kper);
#pragma unroll 8
      for (int k = ks * kper; k < k1; ++k) {
        const float4 w = wp[(size_t)k * NC]; const float xv = hb[k];
        acc.x = fmaf(xv, w.x, acc.x); acc.y = fmaf(xv, w.y, acc.y); acc.z = fmaf(xv, w.z, acc.z); acc.w = fmaf(xv, w.w, acc.w);
      }
      *reinterpret_cast<float4*>(part + (size_t)ks * A + 4 * cg) = acc;
    }
    __syncthreads();
    for (int n = tid; n < A; n += 64 * ATS_NW) {
      float sum = a.battn ? a.battn[n] : 0.f;
      for (int k2 = 0; k2 < KS; ++k2) sum += part[(size_t)k2 * A + n];
      qs[n] = sum;
      if (sl == 0 && a.q_out) a.q_out[(size_t)b * a.ldq_out + n] = sum;
    }
    __syncthreads();
  }
  const int per = (T + S - 1) / S, j0 = sl * per, j1 = min(T, j0 + per);
  const float* krow = a.keys + (size_t)b * T * A;
  const int l16 = lane & 15, grp = lane >> 4;
  for (int jb = j0 + wave * 4; jb < j1; jb += 4 * ATS_NW) {      // 16 lanes per encoder position, 4 positions per wave
    const int j = jb + grp;
    float p = 0.f;
    if (j < j1) {
      for (int c = l16 * 4; c < A; c += 64) {
        const float4 k4 = *reinterpret_cast<const float4*>(krow + (size_t)j * A + c);
        const float4 q4 = *reinterpret_cast<const float4*>(&qs[c]);
        const float4 v4 = *reinterpret_cast<const float4*>(a.v + c);
        p += v4.x * taco_tanh_fast(k4.x + q4.x) + v4.y * taco_tanh_fast(k4.y + q4.y) + v4.z * taco_tanh_fast(k4.z + q4.z) + v4.w * taco_tanh_fast(k4.w + q4.w);
      }
    }
    p += __shfl_xor(p, 8, 64); p += __shfl_xor(p, 4, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 1, 64);
    if (l16 == 0 && j < j1) escr[(size_t)b * T + j] = p;
  }
}

__global__ __launch_bounds__(64 * ATS_NW) void k_att_context(const AttnArgs a_in, const float* escr, int S) {
  AttnArgs a = a_in;
  PIN(a.values); PIN(a.score_bias); PIN(a.manual); PIN(a.align); PIN(a.hist); PIN(a.ctx); PIN(a.T_in); PIN(a.D); PIN(a.type); PIN(a.step);
  PIN(a.n_steps); PIN(a.ldctx);
  __shared__ float sc[ATT_MAXT], tmp[ATT_MAXT], tmp2[ATT_MAXT];
  __shared__ float red[ATS_NW][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, sl = blockIdx.y, T = a.T_in, D = a.D;
  const int lda = a.ldalign ? a.ldalign : T;
  float* al = a.align + (size_t)b * lda;
  const float* alp = a.align_prev ? a.align_prev + (size_t)b * lda : al;
  for (int j = tid; j < T; j += 64 * ATS_NW) {
    sc[j] = a.manual ? a.manual[((size_t)b * a.n_steps + a.step) * T + j] : escr[(size_t)b * T + j];
    tmp2[j] = alp[j];                       // previous alignments, read before slice 0 overwrites them
  }
  __syncthreads();
  if (wave == 0 && !a.manual) {
    const int C = (T + 63) >> 6, j0 = lane * C, j1 = min(j0 + C, T);
    if (a.type == 2) {   // monotonic_attention(mode='parallel'), same arithmetic as k_attention
      const float sb = a.score_bias ? a.score_bias[0] : 0.f;
      float run = 0.f;
      for (int j = j0; j < j1; ++j) {
        const float p = taco_sigmoid(sc[j] + sb);
        const float lg = logf(fminf(fmaxf(1.f - p, 1.17549435e-38f), 1.f));
        sc[j] = p; tmp[j] = run; run += lg;
      }
      const float off = wave_scan(run, lane) - run;
      float run2 = 0.f;
      for (int j = j0; j < j1; ++j) {
        const float cp = expf(tmp[j] + off);
        tmp[j] = cp;
        run2 += tmp2[j] / fminf(fmaxf(cp, 1e-10f), 1.f);
        tmp2[j] = run2;
      }
      const float off2 = wave_scan(run2, lane) - run2;
      for (int j = j0; j < j1; ++j) sc[j] = sc[j] * tmp[j] * (tmp2[j] + off2);
    } else {
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
  if (sl == 0) {
    for (int j = tid; j < T; j += 64 * ATS_NW) {
      const float x = sc[j];
      al[j] = x;
      if (a.hist) a.hist[((size_t)b * T + j) * a.n_steps + a.step] = x;
    }
  }
  // context channels [d0, d1) of this slice: 16 lanes x float4 cover 64 channels, the 4 lane groups x ATS_NW waves split the positions
  const int per = ((D + S - 1) / S + 3) & ~3, d0 = sl * per, d1 = min(D, d0 + per);
  const float* vrow = a.values + (size_t)b * T * D;
  const int l16 = lane & 15, grp = lane >> 4;
  float* red4 = &red[0][0];                                  // [ATS_NW][64]
  for (int dc = d0; dc < d1; dc += 64) {
    const int d = dc + 4 * l16;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d < d1) {
#pragma unroll 4
      for (int j = wave * 4 + grp; j < T; j += 4 * ATS_NW) {
        const float wj = sc[j];
        const float4 v4 = *reinterpret_cast<const float4*>(vrow + (size_t)j * D + d);
        acc.x = fmaf(wj, v4.x, acc.x); acc.y = fmaf(wj, v4.y, acc.y); acc.z = fmaf(wj, v4.z, acc.z); acc.w = fmaf(wj, v4.w, acc.w);
      }
    }
    acc.x += __shfl_xor(acc.x, 16, 64); acc.x += __shfl_xor(acc.x, 32, 64);
    acc.y += __shfl_xor(acc.y, 16, 64); acc.y += __shfl_xor(acc.y, 32, 64);
    acc.z += __shfl_xor(acc.z, 16, 64); acc.z += __shfl_xor(acc.z, 32, 64);
    acc.w += __shfl_xor(acc.w, 16, 64); acc.w += __shfl_xor(acc.w, 32, 64);
    if (grp == 0) { red4[wave * 64 + 4 * l16 + 0] = acc.x; red4[wave * 64 + 4 * l16 + 1] = acc.y; red4[wave * 64 + 4 * l16 + 2] = acc.z; red4[wave * 64 + 4 * l16 + 3] = acc.w; }
    __syncthreads();
    if (wave == 0 && dc + lane < d1) {
      float s2 = 0.f;
#pragma unroll
      for (int w = 0; w < ATS_NW; ++w) s2 += red4[w * 64 + lane];
      a.ctx[(size_t)b * a.ldctx + dc + lane] = s2;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// stop rule of helpers.py:29 + TF dynamic_decode: a row is finished once a step's r*num_mels outputs are
// all exactly 0; the loop ends after the first step at which every row is finished.
// nz [n_steps, B] : 1 if row b emitted any non-zero at step t.
// errw (nullable): the sticky device error word of the persistent kernels; when it is set the stop word becomes its negative (the
// forward's own error latch, see latch_errors() in taco_lib.hip) -- the whole forward ends in this one launch
__global__ __launch_bounds__(1024) void k_stop_step(const int* nz, int B, int n_steps, int* stop, const unsigned* errw = nullptr) {
  // all loads independent (a per-row serial walk with an early exit was a chain of n_steps dependent cache misses: 34 us at C2); 1024 threads,
  // (step, row) advanced without a division: four loads per thread at C2 (round 5: 10.7 -> ~4 us on the tail of every forward)
  __shared__ int first[1024];     // first all-zero step of the rows of one chunk
  __shared__ int worst;
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (tid == 0) worst = 0;
  for (int b0 = 0; b0 < B; b0 += 1024) {
    const int nb = min(1024, B - b0);
    for (int i = tid; i < nb; i += nthr) first[i] = n_steps;
    __syncthreads();
    const int dt = nthr / nb, db = nthr - dt * nb;
    int t = tid / nb, b = tid - t * nb;
    for (; t < n_steps; ) {
      if (nz[(size_t)t * B + b0 + b] == 0) atomicMin(&first[b], t);
      t += dt; b += db;
      if (b >= nb) { b -= nb; ++t; }
    }
    __syncthreads();
    int w = 0;
    for (int i = tid; i < nb; i += nthr) w = max(w, first[i]);
    if (w) atomicMax(&worst, w);
    __syncthreads();
  }
  if (tid == 0) {
    const unsigned e = errw ? errw[0] : 0u;
    *stop = e ? -(int)e : min(worst + 1, n_steps);
  }
}

// see latch_errors() in taco_lib.hip
__global__ void k_latch_errors(const unsigned* errw, int* stop) {
  if (threadIdx.x == 0) { const unsigned e = errw[0]; if (e) *stop = -(int)e; }
}

// The same stop rule evaluated on a finished mel buffer for groups of `rows` consecutive batch rows (requests that were served
// together through one plan): stop[g] = min(max over the group's rows of (first step whose r*num_mels outputs are all 0) + 1, n).
// One workgroup per batch row; stop must be zeroed before the launch.  y [B, n_steps, width].
__global__ __launch_bounds__(256) void k_stop_groups(const float* y, int n_steps, int width, int rows, int* stop) {
  __shared__ int first;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) first = n_steps;
  __syncthreads();
  const float* yb = y + (size_t)b * n_steps * width;
  for (int t = wave; t < n_steps; t += 4) {          // a wave per step: any non-zero among the step's outputs?
    if (t >= *(volatile int*)&first) break;          // later steps cannot lower the minimum
    bool nz = false;
    for (int c = lane; c < width; c += 64) nz |= yb[(size_t)t * width + c] != 0.f;
    if (!__any(nz) && lane == 0) atomicMin(&first, t);
  }
  __syncthreads();
  if (tid == 0) atomicMax(stop + b / rows, min(first + 1, n_steps));
}

// attention-based trimming of the synthesised spectrogram (synthesizer.py:242-262, `attention_trim`): walk the per-step argmax of
// the alignments until the attention has dwelt on the last attended input position; spec_end = r*j + 3.
// One wave per batch row; align [B, T_in, n] (tacotron.py:238-239 layout), seq_len[b] = len(sequence) of the row.
__global__ __launch_bounds__(64) void k_attention_trim(const float* align, const int* seq_len, int T_in, int n, int r, int* spec_end) {
  extern __shared__ int amax[];          // [n] argmax over input positions of every decoder step
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* a = align + (size_t)b * T_in * n;
  for (int t = lane; t < n; t += 64) {
    float best = a[t]; int arg = 0;
    for (int j = 1; j < T_in; ++j) { const float v = a[(size_t)j * n + t]; if (v > best) { best = v; arg = j; } }   // first maximum, like np.argmax
    amax[t] = arg;
  }
  __syncthreads();
  if (lane == 0) {
    int mx = 0;
    for (int t = 0; t < n; ++t) mx = max(mx, amax[t]);
    const int end_idx = min(seq_len[b] - 1, mx);
    int cnt = 0;
    for (int t = 0; t < n; ++t) cnt += (amax[t] == end_idx);
    const int max_counter = min(cnt, 5);
    int counter = 0, jdx = 0;
    for (jdx = 0; jdx < n; ++jdx) {
      if (n > jdx + 1) {
        if (amax[jdx] == end_idx) ++counter;
        if (amax[jdx] == end_idx && amax[jdx + 1] > end_idx) break;
        if (counter >= max_counter) break;
      } else break;
    }
    spec_end[b] = r * jdx + 3;
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
