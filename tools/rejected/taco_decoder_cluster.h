// taco_decoder_cluster.h -- the whole decoder loop (K10-K17) as ONE persistent launch.
//
// Why: each decoder step is a chain of 11 small dependent mat-vec stages.  As separate launches a stage
// costs >= 4.7 us on MI355X (hipGraph node 1.7 us + cold fetch + fp32-MFMA chain); measured alternatives
// (profiles/r01_ubench_*.txt): one workgroup streams weights from L2 at ~150 GB/s; an in-launch exchange
// among P = 4 workgroups of 1 KB slices costs 1.67 us.  So: a "mini-cluster" of P workgroups owns R batch rows;
// for every stage each workgroup streams only ITS 1/P of the weight columns, applies the epilogue to its
// column slice, publishes the slice (write-through sc1 stores + flag), and all P gather the full vector
// (sc1 loads) into their LDS copy of the decoder state.  Attention is row-local: workgroup p runs row p
// (query mat-vec, score, normaliser, context -- att_core) and publishes the context.
// Clusters never talk to each other; every spin is bounded (err word) so the launch cannot hang.
//
// Reference semantics: rnn_wrappers.py:218-341,367-415; helpers.py:9-72; tacotron.py:127-181; A.6, A.9-A.11.
#pragma once
#include "../../multi-speaker-tacotron-tensorflow_amd/csrc/taco_kernels.h"

// ---- in-launch exchange primitives (cdna guide G16, recipe R1 with write-through payload) ----
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
#define XS_SPIN_LIMIT (1u << 21)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xs_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void xs_store16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
  v4u32 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, 16);      // aux 16 = sc1: write-through
}
__device__ __forceinline__ float4 xs_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const v4u32 u = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);   // sc1: bypass this CU's L1
  return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
// every storing wave drains its write-through stores, then ONE lane raises the flag (relaxed, agent scope)
__device__ __forceinline__ void xs_publish(unsigned* flag, unsigned value, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave 0: lane i polls peer i (relaxed, bounded, s_sleep); everyone else parks at the barrier
__device__ __forceinline__ bool xs_wait(const unsigned* flags, int P, unsigned target, unsigned* err, int tid, int* ok_s) {
  if (tid < 64) {
    bool ok = true;
    if (tid < P) {
      unsigned spins = 0;
      while (__hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > XS_SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }
      }
    }
    ok = __all(ok);
    if (tid == 0) {
      if (!ok) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *ok_s = ok ? 1 : 0;
    }
  }
  __syncthreads();
  return *ok_s != 0;
}

#define DC_NT 1024
#define DC_MAXL 4

struct DcStage { const float* W; const float* bias; int K, N; };   // W in TF layout [K, N] row-major

struct DecCArgs {
  DcStage prenet[4], att_g, att_c, proj, g_g[DC_MAXL], g_c[DC_MAXL], out;
  const float* wq; const float* att_v; const float* att_b; const float* score_bias;
  const float* keys; const float* values; const float* teacher; const float* manual;
  const float* h_att0; const float* hd0[DC_MAXL];   // deepvoice initial states [B, .] or null
  float* mel; float* hist; int* nz; float* dbg;
  float* xbuf; unsigned* flags; unsigned* err;
  long long* trace;     // debug: [n][16] wall_clock64 stamps of workgroup 0, or null
  int nprenet, L, att_type;
  int B, T_in, n, M, rM, D, A, As, Hd, P, xstride, lds_tmp;
};

// own column groups [g0, g1) of the stage; partial sums to LDS, reduce, epilogue(r, n, value)
template <int R, typename Epi>
__device__ __forceinline__ void dc_stage(const DcStage& st, const float* x, int ldx, int p, int P, float* part, int tid, Epi epi) {
  const int NC = st.N >> 2;
  const int g0 = (int)((long)p * NC / P), g1 = (int)((long)(p + 1) * NC / P);
  const int NCp = g1 - g0, Np = 4 * NCp;
  int KS = 1;
  if (NCp > 0) {
    KS = DC_NT / NCp; if (KS > st.K) KS = st.K; if (KS < 1) KS = 1;
    const int kper = (st.K + KS - 1) / KS;
    const int cg = tid % NCp, ks = tid / NCp;
    if (ks < KS) {
      const int k0 = ks * kper, k1 = min(st.K, k0 + kper);
      float4 acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4* wp = reinterpret_cast<const float4*>(st.W) + g0 + cg;
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
      for (int r = 0; r < R; ++r) *reinterpret_cast<float4*>(part + (((size_t)ks * R + r) * NCp + cg) * 4) = acc[r];
    }
  }
  __syncthreads();
  for (int o = tid; o < R * Np; o += DC_NT) {
    const int r = o / Np, j = o % Np, n = 4 * g0 + j;
    float s = st.bias ? st.bias[n] : 0.f;
    for (int k2 = 0; k2 < KS; ++k2) s += part[((size_t)k2 * R + r) * Np + j];
    epi(r, n, s);
  }
  __syncthreads();
}

// publish the own column slice of LDS vector vec[r*ld + coff + n] (n < N) to the exchange array [R][N] at xoff
template <int R>
__device__ __forceinline__ void dc_publish_cols(__amdgpu_buffer_rsrc_t xr, unsigned xoff, const float* vec, int ld, int coff,
                                                int N, int p, int P, unsigned* flag, unsigned seq, int tid) {
  const int NC = N >> 2;
  const int g0 = (int)((long)p * NC / P), g1 = (int)((long)(p + 1) * NC / P), NCp = g1 - g0;
  for (int i = tid; i < R * NCp; i += DC_NT) {
    const int r = i / NCp, g = g0 + i % NCp;
    xs_store16(xr, (xoff + (unsigned)r * N + 4 * g) * 4u, *reinterpret_cast<const float4*>(vec + r * ld + coff + 4 * g));
  }
  xs_publish(flag, seq, tid);
}
// wait for all P flags of the cluster, then land the full [R][N] array in LDS
template <int R>
__device__ __forceinline__ bool dc_gather(__amdgpu_buffer_rsrc_t xr, unsigned xoff, float* vec, int ld, int coff, int N,
                                          const unsigned* flags, int P, unsigned seq, unsigned* err, int tid, int* ok_s) {
  if (!xs_wait(flags, P, seq, err, tid, ok_s)) return false;
  const int N4 = N >> 2;
  for (int i = tid; i < R * N4; i += DC_NT) {
    const int r = i / N4, g = i % N4;
    *reinterpret_cast<float4*>(vec + r * ld + coff + 4 * g) = xs_load16(xr, (xoff + (unsigned)r * N + 4 * g) * 4u);
  }
  __syncthreads();
  return true;
}
__device__ __forceinline__ void dc_copy(float* dst, int ldd, int cd, const float* src, int lds_, int cs, int R, int N, int tid) {
  for (int i = tid; i < R * N; i += DC_NT) { const int r = i / N, c = i % N; dst[r * ldd + cd + c] = src[r * lds_ + cs + c]; }
}

template <int R>
__global__ __launch_bounds__(DC_NT) void k_decoder_cluster(const DecCArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int P = a.P, cl = blockIdx.x / P, p = blockIdx.x % P;
  const int row0 = cl * R;                    // first batch row of this cluster
  const int M = a.M, rM = a.rM, D = a.D, A = a.A, As = a.As, Hd = a.Hd, L = a.L, np = a.nprenet, T = a.T_in, n = a.n, B = a.B;
  const int I = a.prenet[np - 1].N;           // attention-GRU input size
  // ---- LDS carve (every vector [R][ld]; all ld multiples of 4) ----
  float* q = smem;
  const int ld_xin = M + D;   float* xin = q;                q += R * ld_xin;      // [frame | ctx]
  float* zbase = q;                                          // prenet outputs back to back: layer i at zoff(i), ld = prenet[i].N
  for (int i = 0; i < np; ++i) q += R * a.prenet[i].N;        // (no local arrays indexed at run time: they would live in scratch)
  auto zoff = [&](int i) { int o = 0; for (int j = 0; j < i; ++j) o += R * a.prenet[j].N; return o; };
  const int ld_xa = I + As;   float* xa = q;                 q += R * ld_xa;       // [z | h_att]
  float* xb = q;                                             q += R * ld_xa;       // [z | r(.)h_att]
  const int ld_gu = 2 * max(As, Hd); float* gu = q;          q += R * ld_gu;       // gathered [r(.)h | u]
  const int ld_hc = As + D;   float* hc = q;                 q += R * ld_hc;       // [h_att | ctx]
  float* ob = q;                                             q += (L + 1) * R * Hd;   // o_0 .. o_L
  float* hb = q;                                             q += L * R * Hd;         // decoder GRU states
  const int ld_g = 2 * Hd;    float* ga = q;                 q += R * ld_g;        // [o_l | h_l]
  float* gb = q;                                             q += R * ld_g;        // [o_l | r(.)h_l]
  float* yb = q;                                             q += R * rM;
  float* sc = q;  q += T;  float* tmp = q;  q += a.lds_tmp;  float* tmp2 = q;  q += T;  float* al = q;  q += T;
  float* part = q;                                           q += (size_t)DC_NT * R * 4;   // also att_core's cred (>= 4096 floats)
  int* ok_s = reinterpret_cast<int*>(q);

  // ---- exchange arrays of this cluster (floats): one [R][N] array per stage output ----
  const __amdgpu_buffer_rsrc_t xr = xs_rsrc(a.xbuf + (size_t)cl * a.xstride, (unsigned)a.xstride * 4u);
  unsigned xo = 0;
  const unsigned X_Z0 = xo; for (int i = 0; i < np; ++i) xo += R * a.prenet[i].N;     // layer i at X_Z0 + zoff(i)
  const unsigned X_GA = xo; xo += R * 2 * As;
  const unsigned X_HA = xo; xo += R * As;
  const unsigned X_CX = xo; xo += R * D;
  const unsigned X_O0 = xo; xo += R * Hd;
  const unsigned X_G0 = xo; xo += (unsigned)L * R * 3 * Hd;   // layer l: gates at X_G0 + l*3*R*Hd, state at + 2*R*Hd
  const unsigned X_Y = xo;
  unsigned* fl = a.flags + cl * P;
  unsigned seq = 0;

  // ---- initial state (rnn_wrappers.py:186-216; tacotron.py:183-197; helpers.py:70-72) ----
  for (int i = tid; i < (int)(q - smem); i += DC_NT) smem[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < R * As; i += DC_NT) {
    const int r = i / As, c = i % As, b = row0 + r;
    const float v = (a.h_att0 && b < B) ? a.h_att0[(size_t)b * As + c] : 0.f;
    xa[r * ld_xa + I + c] = v; hc[r * ld_hc + c] = v;
  }
  for (int l = 0; l < L; ++l)
    for (int i = tid; i < R * Hd; i += DC_NT) {
      const int r = i / Hd, c = i % Hd, b = row0 + r;
      hb[(l * R + r) * Hd + c] = (a.hd0[l] && b < B) ? a.hd0[l][(size_t)b * Hd + c] : 0.f;
    }
  if (a.att_type == 2 && tid == 0) al[0] = 1.f;    // BahdanauMonotonicAttention.initial_alignments = one_hot(0)
  __syncthreads();

  const int brow = row0 + p;                      // the batch row whose attention this workgroup runs
  const bool has_row = (p < R) && (brow < B);
  const int dbgw = As + D + L * Hd;

#define DC_STAMP(i) do { if (a.trace && blockIdx.x == 0 && tid == 0) a.trace[(size_t)t * 16 + (i)] = wall_clock64(); } while (0)
  for (int t = 0; t < n; ++t) {
    DC_STAMP(0);
    // ---- prenet (rnn_wrappers.py:249,367-378): relu(dense) per layer, input concat(frame, previous context) ----
    for (int i = 0; i < np; ++i) {
      const float* x = (i == 0) ? xin : zbase + zoff(i - 1);
      const int ldx = (i == 0) ? ld_xin : a.prenet[i - 1].N;
      float* z = zbase + zoff(i); const int ldz = a.prenet[i].N;
      const unsigned xz = X_Z0 + (unsigned)zoff(i);
      dc_stage<R>(a.prenet[i], x, ldx, p, P, part, tid, [&](int r, int nn, float s) { z[r * ldz + nn] = fmaxf(s, 0.f); });
      dc_publish_cols<R>(xr, xz, z, ldz, 0, ldz, p, P, fl + p, ++seq, tid);
      if (!dc_gather<R>(xr, xz, z, ldz, 0, ldz, fl, P, seq, a.err, tid, ok_s)) return;
    }
    DC_STAMP(1);
    dc_copy(xa, ld_xa, 0, zbase + zoff(np - 1), I, 0, R, I, tid);
    dc_copy(xb, ld_xa, 0, zbase + zoff(np - 1), I, 0, R, I, tid);
    __syncthreads();
    // ---- attention GRUCell (tacotron.py:127-130; A.6) ----
    dc_stage<R>(a.att_g, xa, ld_xa, p, P, part, tid, [&](int r, int nn, float s) {
      const float sg = taco_sigmoid(s);
      gu[r * ld_gu + nn] = (nn < As) ? sg * xa[r * ld_xa + I + nn] : sg;           // r(.)h | u
    });
    DC_STAMP(2);
    dc_publish_cols<R>(xr, X_GA, gu, ld_gu, 0, 2 * As, p, P, fl + p, ++seq, tid);
    if (!dc_gather<R>(xr, X_GA, gu, ld_gu, 0, 2 * As, fl, P, seq, a.err, tid, ok_s)) return;
    DC_STAMP(3);
    dc_copy(xb, ld_xa, I, gu, ld_gu, 0, R, As, tid);
    __syncthreads();
    dc_stage<R>(a.att_c, xb, ld_xa, p, P, part, tid, [&](int r, int nn, float s) {
      const float c = tanhf(s), h = xa[r * ld_xa + I + nn], u = gu[r * ld_gu + As + nn];
      hc[r * ld_hc + nn] = u * h + (1.f - u) * c;                                   // new h_att (own columns)
    });
    dc_publish_cols<R>(xr, X_HA, hc, ld_hc, 0, As, p, P, fl + p, ++seq, tid);
    if (!dc_gather<R>(xr, X_HA, hc, ld_hc, 0, As, fl, P, seq, a.err, tid, ok_s)) return;
    dc_copy(xa, ld_xa, I, hc, ld_hc, 0, R, As, tid);
    __syncthreads();
    DC_STAMP(4);
    // ---- attention of row p by workgroup p: query, score, normaliser, context (rnn_wrappers.py:304-341) ----
    if (has_row) {
      AttnArgs at;
      at.q = nullptr; at.hq = nullptr; at.wq = a.wq; at.keys = a.keys; at.values = a.values; at.v = a.att_v; at.battn = a.att_b;
      at.score_bias = a.score_bias; at.manual = a.manual; at.align = nullptr; at.hist = a.hist; at.ctx = nullptr;
      at.T_in = T; at.A = A; at.D = D; at.type = a.att_type; at.step = t; at.n_steps = n; at.As = As;
      att_core(at, brow, sc, tmp, tmp2, part, hc + p * ld_hc, al, hc + p * ld_hc + As);
      __syncthreads();
    }
    DC_STAMP(5);
    if (p < R) {      // publish row p of the context array (zeros for rows beyond the batch)
      for (int i = tid; i < D / 4; i += DC_NT)
        xs_store16(xr, (X_CX + (unsigned)p * D + 4 * i) * 4u, *reinterpret_cast<const float4*>(hc + p * ld_hc + As + 4 * i));
    }
    xs_publish(fl + p, ++seq, tid);
    if (!dc_gather<R>(xr, X_CX, hc, ld_hc, As, D, fl, P, seq, a.err, tid, ok_s)) return;
    dc_copy(xin, ld_xin, M, hc, ld_hc, As, R, D, tid);
    __syncthreads();
    DC_STAMP(6);
    // ---- concat(h_att, ctx) -> projection (rnn_wrappers.py:405-415; tacotron.py:166-170) ----
    dc_stage<R>(a.proj, hc, ld_hc, p, P, part, tid, [&](int r, int nn, float s) { ob[r * Hd + nn] = s; });
    dc_publish_cols<R>(xr, X_O0, ob, Hd, 0, Hd, p, P, fl + p, ++seq, tid);
    if (!dc_gather<R>(xr, X_O0, ob, Hd, 0, Hd, fl, P, seq, a.err, tid, ok_s)) return;
    DC_STAMP(7);
    // ---- residual GRU stack (tacotron.py:171-172) ----
    for (int l = 0; l < L; ++l) {
      float* ol = ob + (size_t)l * R * Hd; float* on = ob + (size_t)(l + 1) * R * Hd; float* hl = hb + (size_t)l * R * Hd;
      dc_copy(ga, ld_g, 0, ol, Hd, 0, R, Hd, tid);
      dc_copy(ga, ld_g, Hd, hl, Hd, 0, R, Hd, tid);
      dc_copy(gb, ld_g, 0, ol, Hd, 0, R, Hd, tid);
      __syncthreads();
      dc_stage<R>(a.g_g[l], ga, ld_g, p, P, part, tid, [&](int r, int nn, float s) {
        const float sg = taco_sigmoid(s);
        gu[r * ld_gu + nn] = (nn < Hd) ? sg * hl[r * Hd + nn] : sg;
      });
      const unsigned xg = X_G0 + (unsigned)l * 3 * R * Hd, xh = xg + 2 * R * Hd;
      dc_publish_cols<R>(xr, xg, gu, ld_gu, 0, 2 * Hd, p, P, fl + p, ++seq, tid);
      if (!dc_gather<R>(xr, xg, gu, ld_gu, 0, 2 * Hd, fl, P, seq, a.err, tid, ok_s)) return;
      dc_copy(gb, ld_g, Hd, gu, ld_gu, 0, R, Hd, tid);
      __syncthreads();
      dc_stage<R>(a.g_c[l], gb, ld_g, p, P, part, tid, [&](int r, int nn, float s) {
        const float c = tanhf(s), h = hl[r * Hd + nn], u = gu[r * ld_gu + Hd + nn];
        on[r * Hd + nn] = u * h + (1.f - u) * c;                                    // new h_l (own columns), parked in o_{l+1}
      });
      dc_publish_cols<R>(xr, xh, on, Hd, 0, Hd, p, P, fl + p, ++seq, tid);
      if (!dc_gather<R>(xr, xh, on, Hd, 0, Hd, fl, P, seq, a.err, tid, ok_s)) return;
      for (int i = tid; i < R * Hd; i += DC_NT) { const float hn = on[i]; hl[i] = hn; on[i] = hn + ol[i]; }   // ResidualWrapper
      __syncthreads();
    }
    DC_STAMP(8);
    // ---- frame projection to r frames (tacotron.py:178-179); feedback / stop rule (helpers.py:26-32) ----
    {
      const float* oL = ob + (size_t)L * R * Hd;
      dc_stage<R>(a.out, oL, Hd, p, P, part, tid, [&](int r, int nn, float s) {
        yb[r * rM + nn] = s;
        const int b = row0 + r;
        if (b < B) {
          a.mel[((size_t)b * n + t) * rM + nn] = s;
          if (s != 0.f) a.nz[(size_t)t * B + b] = 1;
        }
      });
      dc_publish_cols<R>(xr, X_Y, yb, rM, 0, rM, p, P, fl + p, ++seq, tid);
      if (!dc_gather<R>(xr, X_Y, yb, rM, 0, rM, fl, P, seq, a.err, tid, ok_s)) return;
      for (int i = tid; i < R * M; i += DC_NT) {
        const int r = i / M, c = i % M, b = row0 + r;
        float f = yb[r * rM + (rM - M) + c];                                           // last of the r frames
        if (a.teacher) f = (b < B) ? a.teacher[((size_t)b * n + t) * M + c] : 0.f;      // helpers.py:44,66
        xin[r * ld_xin + c] = f;
      }
      DC_STAMP(9);
      if (a.dbg && p == 0) {
        for (int i = tid; i < R * dbgw; i += DC_NT) {
          const int r = i / dbgw, c = i % dbgw, b = row0 + r;
          if (b < B) {
            float v;
            if (c < As) v = hc[r * ld_hc + c];
            else if (c < As + D) v = hc[r * ld_hc + c];
            else { const int l = (c - As - D) / Hd, cc = (c - As - D) % Hd; v = hb[((size_t)l * R + r) * Hd + cc]; }
            a.dbg[((size_t)t * B + b) * dbgw + c] = v;
          }
        }
      }
      __syncthreads();
    }
  }
}
