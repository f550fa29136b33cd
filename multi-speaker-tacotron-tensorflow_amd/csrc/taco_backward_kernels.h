// taco_backward_kernels.h -- training path: training-mode forward pieces and the backward pass.
// tf.gradients of the graph of tacotron.py:29-239 under is_training (batch-statistics BatchNorm, teacher forcing),
// written out by hand.  Every weight gradient is accumulated into ONE flat fp32 buffer (the RCCL bucket of the
// data-parallel step) with fp32 atomics; the caller zeroes it before a backward pass.
// Needs taco_kernels.h (PIN, rp_matvec, wave_sum/wave_scan, taco_sigmoid, ATT_MAXT) included first.
#pragma once
#include <hip/hip_runtime.h>

// ---- weight packs regenerated from the flat parameter buffer after every optimizer step ----
// map[i] = 1 + flat index of the parameter that pack element i copies (the host packers run once over
// index-valued tensors); anything else = zero padding / unused.
// Indices NP + 1 .. NP + NF address a second source F (computed values: the GRU-1 fold of the persistent decoder, k_dx_fold).
__device__ __forceinline__ float pack_pick(float v, const float* __restrict__ P, unsigned NP, const float* __restrict__ F, unsigned NF) {
  float o = 0.f;
  if (v >= 1.f && v <= (float)(NP + NF)) {
    const unsigned iv = (unsigned)v;
    if ((float)iv == v) o = iv <= NP ? P[iv - 1] : F[iv - NP - 1];
  }
  return o;
}
__global__ __launch_bounds__(256) void k_pack_gather(const float* __restrict__ map, const float* __restrict__ P,
                                                    float* __restrict__ arena, size_t n, unsigned NP, const float* __restrict__ F, unsigned NF) {
  // four pack elements per thread: the map is read and the arena written 16 bytes at a time (both hipMalloc'ed), the gathers are scalar
  const size_t n4 = n >> 2, stride = (size_t)gridDim.x * 256;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += stride) {
    const float4 v = *reinterpret_cast<const float4*>(map + 4 * q);
    *reinterpret_cast<float4*>(arena + 4 * q) = make_float4(pack_pick(v.x, P, NP, F, NF), pack_pick(v.y, P, NP, F, NF), pack_pick(v.z, P, NP, F, NF), pack_pick(v.w, P, NP, F, NF));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const size_t i = 4 * n4 + threadIdx.x; arena[i] = pack_pick(map[i], P, NP, F, NF); }
}
// The split-bf16 packs of the training model (k_gemm_bf3 operands) from the live parameters: element e of the concatenated index
// list belongs to the segment s with segs[s].start <= e < start + count and becomes hi = bf16(w), lo = bf16(w - hi) at position
// e - start of the segment's two arrays (the same split pack_bf3 does on the host for inference).
struct Bf3Seg { unsigned start, count; unsigned long long hi, lo, l3; };   // [start, start + count) of the index list -> bf16 arrays at arena float offsets hi / lo / l3 (third plane: bf16(w - hi - lo))
__device__ __forceinline__ unsigned short bf16_rne_dev(float f) {
  const unsigned u = __float_as_uint(f);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ void bf3_split(float w, unsigned short& hb, unsigned short& lb, unsigned short& tb) {
  hb = bf16_rne_dev(w);
  lb = bf16_rne_dev(w - __uint_as_float((unsigned)hb << 16));
  tb = bf16_rne_dev(w - __uint_as_float((unsigned)hb << 16) - __uint_as_float((unsigned)lb << 16));
}
__global__ __launch_bounds__(256) void k_bf3_gather(const unsigned* __restrict__ idx, const Bf3Seg* __restrict__ segs, int nseg, const float* __restrict__ P,
                                                   float* __restrict__ arena, size_t n, unsigned NP) {
  // eight elements per thread: ONE segment search, two 16-byte index loads and three 16-byte stores where the eight lie in one segment at a
  // multiple of 8 from its start and the segment's arrays are 16-byte aligned (the packs' fragments are 8 bf16 wide); element by element otherwise
  auto seg_of = [&](size_t e) { int lo = 0, hi = nseg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((size_t)segs[mid].start <= e) lo = mid; else hi = mid - 1; }
    return lo; };
  const size_t n8 = (n + 7) >> 3, stride = (size_t)gridDim.x * 256;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n8; q += stride) {
    const size_t e0 = 8 * q;
    const Bf3Seg sg = segs[seg_of(e0)];
    const size_t k = e0 - sg.start;
    if (e0 + 8 <= n && k + 8 <= (size_t)sg.count && (k & 7) == 0 && ((sg.hi | sg.lo | sg.l3) & 3) == 0) {
      const uint4 i0 = *reinterpret_cast<const uint4*>(idx + e0), i1 = *reinterpret_cast<const uint4*>(idx + e0 + 4);
      const unsigned ii[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
      float w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = (ii[u] >= 1u && ii[u] <= NP) ? P[ii[u] - 1] : 0.f;
      unsigned hh[4], ll[4], tt[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        unsigned short h0, l0, t0, h1, l1, t1;
        bf3_split(w[2 * u], h0, l0, t0); bf3_split(w[2 * u + 1], h1, l1, t1);
        hh[u] = (unsigned)h0 | ((unsigned)h1 << 16); ll[u] = (unsigned)l0 | ((unsigned)l1 << 16); tt[u] = (unsigned)t0 | ((unsigned)t1 << 16);
      }
      *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(arena + sg.hi) + k) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
      *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(arena + sg.lo) + k) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
      *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(arena + sg.l3) + k) = make_uint4(tt[0], tt[1], tt[2], tt[3]);
      continue;
    }
    for (size_t e = e0; e < e0 + 8 && e < n; ++e) {
      const Bf3Seg s1 = segs[seg_of(e)];
      const unsigned i = idx[e];
      const float w = (i >= 1u && i <= NP) ? P[i - 1] : 0.f;
      unsigned short hb, lb, tb;
      bf3_split(w, hb, lb, tb);
      const size_t k1 = e - s1.start;
      reinterpret_cast<unsigned short*>(arena + s1.hi)[k1] = hb;
      reinterpret_cast<unsigned short*>(arena + s1.lo)[k1] = lb;
      reinterpret_cast<unsigned short*>(arena + s1.l3)[k1] = tb;
    }
  }
}
// The concat projection folded into decoder GRU 1 (taco_model_finalize does it on the host, in double, for inference):
//   F[z][j] = sum_k C[z][k] * G[k][j],  z = 0 .. Z (row Z: C = the projection's bias), j = 0 .. 3H-1,
//   G = [gates kernel x rows (2H columns) | candidate kernel x rows (H columns)],  row Z, j < 2H additionally + gates bias[j].
// One workgroup per DXF_ROWS rows z (their C rows staged in LDS), thread = column j of each of the three H-wide column blocks; every
// (z, j) is one fmaf chain over k ascending: fp32 with a fixed summation order.  H = blockDim.x = 256 (the persistent decoder's width).
#define DXF_ROWS 8
__global__ __launch_bounds__(256) void k_dx_fold(const float* __restrict__ Wc, const float* __restrict__ bc, const float* __restrict__ gk,
                                                const float* __restrict__ gb, const float* __restrict__ ck, float* __restrict__ F, int Z, int H) {
  __shared__ float cs[DXF_ROWS][256];
  const int z0 = blockIdx.x * DXF_ROWS, j = threadIdx.x;
#pragma unroll
  for (int r = 0; r < DXF_ROWS; ++r) { const int z = z0 + r; cs[r][j] = z < Z ? Wc[(size_t)z * H + j] : (z == Z ? bc[j] : 0.f); }
  __syncthreads();
  float acc[DXF_ROWS][3];
#pragma unroll
  for (int r = 0; r < DXF_ROWS; ++r) { acc[r][0] = 0.f; acc[r][1] = 0.f; acc[r][2] = 0.f; }
  for (int k = 0; k < H; ++k) {
    const float g0 = gk[(size_t)k * 2 * H + j], g1 = gk[(size_t)k * 2 * H + H + j], g2 = ck[(size_t)k * H + j];
#pragma unroll
    for (int r = 0; r < DXF_ROWS; ++r) {
      const float c = cs[r][k];
      acc[r][0] = fmaf(c, g0, acc[r][0]); acc[r][1] = fmaf(c, g1, acc[r][1]); acc[r][2] = fmaf(c, g2, acc[r][2]);
    }
  }
#pragma unroll
  for (int r = 0; r < DXF_ROWS; ++r) {
    const int z = z0 + r;
    if (z > Z) continue;
    float* o = F + (size_t)z * 3 * H;
    o[j] = acc[r][0] + (z == Z ? gb[j] : 0.f); o[H + j] = acc[r][1] + (z == Z ? gb[H + j] : 0.f); o[2 * H + j] = acc[r][2];
  }
}

// ---- per-column reductions over the rows of a [M, C] matrix (bias gradients, BatchNorm statistics) ----
// mode 0: out1[c] += sum_m a                         mode 1: out2[c] += sum_m (a - mu[c])^2
// mode 2: out1[c] += sum_m b ; out2[c] += sum_m b * (a - mu[c]) * rstd[c]     (BN backward: b = dy, a = BN input)
// part != null (deterministic mode): the row chunk's sums go to part[chunk][0 | 1][C] instead of being added atomically, and
// k_colsum_reduce adds them to out1 / out2 chunk by chunk in a fixed order.
struct ColArgs { const float* a; const float* b; const float* mu; const float* rstd; float* out1; float* out2;
                 int lda, ldb, M, C, mode, rpb; float* part; };
__device__ __forceinline__ void colsum_body(const ColArgs& g, int bx, int by) {
  // 64 columns x rpb rows per workgroup.  Where the rows allow 16-byte loads (strides and C multiples of 4, aligned bases) a thread owns
  // FOUR columns and every 16th row (a wave reads four 256-byte row segments per instruction, four rows in flight per thread);
  // otherwise (the linear head's 1025 columns) one column and every 4th row.  Either way a column's rows are added in an order that
  // depends on the shapes only: per-thread partial sums, then the row lanes in ascending order (fixed order: bit-reproducible).
  __shared__ float s1[16][64], s2[16][64];
  const int m0 = by * g.rpb, m1 = min(g.M, m0 + g.rpb), c0 = bx * 64;
  const bool vec = (g.C & 3) == 0 && (g.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(g.a) & 15) == 0 &&
                   (g.mode != 2 || ((g.ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(g.b) & 15) == 0));
  int nl;
  if (vec) {
    nl = 16;
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4, c = c0 + 4 * cq;
    float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
    if (c < g.C) {
      const float* pa = g.a + (size_t)(m0 + rl) * g.lda + c;
      const size_t sa = (size_t)16 * g.lda;
      const int n = m1 - m0 - rl > 0 ? (m1 - m0 - rl + 15) >> 4 : 0;
      if (g.mode == 0) {
#pragma unroll 4
        for (int k = 0; k < n; ++k) { const float4 v = *reinterpret_cast<const float4*>(pa + k * sa); a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w; }
      } else {
        const float4 mu = g.mu ? make_float4(g.mu[c], g.mu[c + 1], g.mu[c + 2], g.mu[c + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);      // (per-column vectors may sit anywhere in the flat parameter buffer)
        if (g.mode == 1) {
#pragma unroll 4
          for (int k = 0; k < n; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(pa + k * sa);
            const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
            a2.x += dx * dx; a2.y += dy * dy; a2.z += dz * dz; a2.w += dw * dw;
          }
        } else {
          const float4 rs = g.rstd ? make_float4(g.rstd[c], g.rstd[c + 1], g.rstd[c + 2], g.rstd[c + 3]) : make_float4(1.f, 1.f, 1.f, 1.f);
          const float* pb = g.b + (size_t)(m0 + rl) * g.ldb + c;
          const size_t sb = (size_t)16 * g.ldb;
#pragma unroll 4
          for (int k = 0; k < n; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(pa + k * sa), w = *reinterpret_cast<const float4*>(pb + k * sb);
            a1.x += w.x; a1.y += w.y; a1.z += w.z; a1.w += w.w;
            a2.x += w.x * (v.x - mu.x) * rs.x; a2.y += w.y * (v.y - mu.y) * rs.y; a2.z += w.z * (v.z - mu.z) * rs.z; a2.w += w.w * (v.w - mu.w) * rs.w;
          }
        }
      }
    }
    *reinterpret_cast<float4*>(&s1[rl][4 * cq]) = a1; *reinterpret_cast<float4*>(&s2[rl][4 * cq]) = a2;
  } else {
    nl = 4;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = c0 + cl;
    float a1 = 0.f, a2 = 0.f;
    if (c < g.C) {
      const float mu = g.mu ? g.mu[c] : 0.f, rs = g.rstd ? g.rstd[c] : 1.f;
#pragma unroll 8
      for (int m = m0 + rl; m < m1; m += 4) {
        const float av = g.a[(size_t)m * g.lda + c];
        if (g.mode == 0) a1 += av;
        else if (g.mode == 1) { const float d = av - mu; a2 += d * d; }
        else { const float bv = g.b[(size_t)m * g.ldb + c]; a1 += bv; a2 += bv * (av - mu) * rs; }
      }
    }
    s1[rl][cl] = a1; s2[rl][cl] = a2;
  }
  __syncthreads();
  const int cl = threadIdx.x, c = c0 + cl;
  if (cl < 64 && c < g.C) {
    float t1 = 0.f, t2 = 0.f;
    for (int r = 0; r < nl; ++r) { t1 += s1[r][cl]; t2 += s2[r][cl]; }
    if (g.part) {
      g.part[((size_t)by * 2 + 0) * g.C + c] = t1;
      g.part[((size_t)by * 2 + 1) * g.C + c] = t2;
    } else {
      if (g.mode != 1 && g.out1) atomicAdd(g.out1 + c, t1);
      if (g.mode != 0 && g.out2) atomicAdd(g.out2 + c, t2);
    }
  }
}
__global__ __launch_bounds__(256) void k_colsum(const ColArgs g) { colsum_body(g, blockIdx.x, blockIdx.y); }
// several column sums whose operands are all ready (bias / BatchNorm-backward sums of one backward region) as ONE launch; see k_wgrad_bf3_group
#define COL_MAXP 40
struct ColGroup { ColArgs p[COL_MAXP]; int start[COL_MAXP + 1]; int n; };
__global__ __launch_bounds__(256) void k_colsum_group(const ColGroup G) {
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < G.n && G.start[p + 1] <= b) ++p;
  p = __builtin_amdgcn_readfirstlane(p);
  const ColArgs g = G.p[p];
  const int local = b - G.start[p], gx = (g.C + 63) / 64;
  colsum_body(g, local % gx, local / gx);
}
__global__ void k_colsum_reduce(const float* part, int nchunks, int C, int mode, float* out1, float* out2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float t1 = 0.f, t2 = 0.f;
#pragma unroll 8
  for (int k = 0; k < nchunks; ++k) { t1 += part[((size_t)k * 2 + 0) * C + c]; t2 += part[((size_t)k * 2 + 1) * C + c]; }
  const int md = mode & 3;             // bit 3 of mode: the sums REPLACE out1 / out2 (no zero fill in front: BatchNorm statistics)
  if (md != 1 && out1) out1[c] = (mode & 8) ? t1 : out1[c] + t1;
  if (md != 0 && out2) out2[c] = (mode & 8) ? t2 : out2[c] + t2;
}

// the ordered sums of several column-sum problems of a group launch (deterministic mode) as ONE launch
struct ColRed { const float* part; float* out1; float* out2; int nchunks, C, mode; };
struct ColRedGroup { ColRed p[COL_MAXP]; int start[COL_MAXP + 1]; int n; };
__global__ __launch_bounds__(256) void k_colsum_reduce_group(const ColRedGroup G) {
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < G.n && G.start[p + 1] <= b) ++p;
  p = __builtin_amdgcn_readfirstlane(p);
  const ColRed g = G.p[p];
  const int c = (b - G.start[p]) * 256 + threadIdx.x;
  if (c >= g.C) return;
  float t1 = 0.f, t2 = 0.f;
#pragma unroll 8
  for (int k = 0; k < g.nchunks; ++k) { t1 += g.part[((size_t)k * 2 + 0) * g.C + c]; t2 += g.part[((size_t)k * 2 + 1) * g.C + c]; }
  const int md = g.mode & 3;
  if (md != 1 && g.out1) g.out1[c] = (g.mode & 8) ? t1 : g.out1[c] + t1;
  if (md != 0 && g.out2) g.out2[c] = (g.mode & 8) ? t2 : g.out2[c] + t2;
}

// BatchNorm (training): mu = S1/M; second pass gives S2 = sum (a-mu)^2; rstd = 1/sqrt(S2/M + eps) (biased variance);
// moving <- moving*momentum + batch*(1-momentum), written straight into the flat parameter buffer (UPDATE_OPS, tacotron.py:334)
__global__ void k_bn_mean(const float* S1, float* mu, int C, float invM) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) mu[c] = S1[c] * invM;
}
// SyncBN, one exchange per layer: every rank contributes (d_r = its mean - ref, its centred sum of squares S_r, d_r^2); after the sum
// over the W ranks (equal row counts M per rank: Trainer checks it)  mu = ref + sum d_r / W  and
// sum (a - mu)^2 over all rows = sum S_r + M (sum d_r^2 - W dbar^2)   (Chan's pairwise update).
// `ref` is the layer's moving mean -- the same number on every rank, and close to the batch means once training is under way -- so
// the cancellation in the bracket is over the small offsets d_r, not over the means themselves (with ref = 0 the
// bracket lost precision when |mean| was much larger than the spread of the rank means).
__global__ void k_bn_sync_pack(const float* mu_local, const float* S_local, const float* ref, float* pack, int C, int c0, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = c0 + i;
  const float d = mu_local[c] - ref[i];
  pack[c] = d; pack[C + c] = S_local[c]; pack[2 * C + c] = d * d;
}
__global__ void k_bn_sync_combine(const float* pack, const float* ref, float* mu, float* S2, int C, int c0, int n, float M, float W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = c0 + i;
  const float dbar = pack[c] / W;
  mu[c] = ref[i] + dbar;
  S2[c] = pack[C + c] + M * fmaxf(pack[2 * C + c] - W * dbar * dbar, 0.f);
}
__global__ void k_bn_finalize(const float* mu, const float* S2, float* rstd, float* mov_mean, float* mov_var, int C,
                              float invM, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float var = S2[c] * invM;
  rstd[c] = 1.0f / sqrtf(var + eps);
  if (mov_mean) mov_mean[c] = mov_mean[c] * momentum + mu[c] * (1.f - momentum);
  if (mov_var) mov_var[c] = mov_var[c] * momentum + var * (1.f - momentum);
}
__global__ void k_bn_apply(const float* a, int lda, const float* mu, const float* rstd, const float* gamma, const float* beta,
                           float* y, int ldy, int M, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const int m = (int)(i / C), c = (int)(i % C);
  y[(size_t)m * ldy + c] = (a[(size_t)m * lda + c] - mu[c]) * rstd[c] * gamma[c] + beta[c];
}
// Four columns per thread (16-byte loads and stores; C, the strides and the bases multiples of 4 floats -- the launchers in taco_train.h check
// that and fall back to the one-element kernels): same expressions per element, no 64-bit index division.
#define V4_INDEX(M, C) const unsigned C4_ = (unsigned)(C) >> 2, i_ = blockIdx.x * 256u + threadIdx.x; if (i_ >= (unsigned)(M) * C4_) return; \
                       const unsigned m = i_ / C4_, c = 4u * (i_ - m * C4_)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ldp4(const float* p) { return make_float4(p[0], p[1], p[2], p[3]); }      // per-column vectors: any alignment (flat parameter buffer, no padding)
__device__ __forceinline__ void st4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
__global__ __launch_bounds__(256) void k_bn_apply_v4(const float* a, int lda, const float* mu, const float* rstd, const float* gamma, const float* beta,
                                                    float* y, int ldy, int M, int C) {
  V4_INDEX(M, C);
  const float4 av = ld4(a + (size_t)m * lda + c), mv = ldp4(mu + c), rv = ldp4(rstd + c), gv = ldp4(gamma + c), bv = ldp4(beta + c);
  st4(y + (size_t)m * ldy + c, make_float4((av.x - mv.x) * rv.x * gv.x + bv.x, (av.y - mv.y) * rv.y * gv.y + bv.y,
                                           (av.z - mv.z) * rv.z * gv.z + bv.z, (av.w - mv.w) * rv.w * gv.w + bv.w));
}
__global__ __launch_bounds__(256) void k_bn_bwd_v4(const float* a, int lda, const float* dy, int ldy, const float* mu, const float* rstd, const float* gamma,
                                                  const float* sdy, const float* sdyxh, int relu, float* dz, int ldz, int M, int C, float invM) {
  V4_INDEX(M, C);
  const float4 av = ld4(a + (size_t)m * lda + c), dv = ld4(dy + (size_t)m * ldy + c), mv = ldp4(mu + c), rv = ldp4(rstd + c), gv = ldp4(gamma + c),
               s1 = ldp4(sdy + c), s2 = ldp4(sdyxh + c);
  auto one = [&](float a_, float d_, float m_, float r_, float g_, float p_, float q_) {
    const float ah = (a_ - m_) * r_;
    const float da = g_ * r_ * (d_ - p_ * invM - ah * q_ * invM);
    return (relu && !(a_ > 0.f)) ? 0.f : da;
  };
  st4(dz + (size_t)m * ldz + c, make_float4(one(av.x, dv.x, mv.x, rv.x, gv.x, s1.x, s2.x), one(av.y, dv.y, mv.y, rv.y, gv.y, s1.y, s2.y),
                                            one(av.z, dv.z, mv.z, rv.z, gv.z, s1.z, s2.z), one(av.w, dv.w, mv.w, rv.w, gv.w, s1.w, s2.w)));
}
// The BatchNorm layers of a conv bank (one per width: column block k of a [M, K*Cw] matrix, modules.py:35-44) as ONE launch: the layers' own vectors
// come through a pointer table (they are separate tensors of the flat parameter / gradient buffers), everything else is indexed by the full column.
#define BNB_MAXK 16
struct BnBank { const float* gamma[BNB_MAXK]; const float* beta[BNB_MAXK]; const float* sdy[BNB_MAXK]; const float* sdyxh[BNB_MAXK];
                float* mov_mean[BNB_MAXK]; float* mov_var[BNB_MAXK]; int Cw; };
// BatchNorm output of ONE element, in one fixed instruction sequence: the fused forward pool below and the pool's backward pass both
// recompute it from the bank's activation tape and must pick the same maximum of a window (the BatchNorm output itself is not stored).
__device__ __forceinline__ float bn_out1(float a, float m, float r, float g, float b) { return __fmaf_rn(__fmul_rn(__fsub_rn(a, m), r), g, b); }
__device__ __forceinline__ float4 bn_out4(const float4 a, const float4 m, const float4 r, const float4 g, const float4 b) {
  return make_float4(bn_out1(a.x, m.x, r.x, g.x, b.x), bn_out1(a.y, m.y, r.y, g.y, b.y), bn_out1(a.z, m.z, r.z, g.z, b.z), bn_out1(a.w, m.w, r.w, g.w, b.w));
}
// BatchNorm of all widths + max_pooling1d(width w, stride 1, 'same') in one pass over the bank's activations (modules.py:35-47): the pooled
// tensor is the only output -- the BatchNorm output [B T, K C] is neither written nor read back (round 6: 134 MB less tape at the C4 shard).
__global__ __launch_bounds__(256) void k_bn_pool_bank_v4(const float* a, int lda, const float* mu, const float* rstd, const BnBank nb, float* pool, int ldp, int M, int T, int C, int w) {
  V4_INDEX(M, C);
  const unsigned k = c / (unsigned)nb.Cw, cl = c - k * nb.Cw;
  const float4 mv = ldp4(mu + c), rv = ldp4(rstd + c), gv = ldp4(nb.gamma[k] + cl), bv = ldp4(nb.beta[k] + cl);
  const int t = (int)(m % (unsigned)T), pl = (w - 1) >> 1;
  float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  for (int j = 0; j < w; ++j) {
    const int tt = t - pl + j;
    if (tt < 0 || tt >= T) continue;
    const float4 v = bn_out4(ld4(a + (size_t)((int)m - pl + j) * lda + c), mv, rv, gv, bv);
    best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
  }
  st4(pool + (size_t)m * ldp + c, best);
}
__global__ __launch_bounds__(256) void k_bn_bwd_bank_v4(const float* a, int lda, const float* dy, int ldy, const float* mu, const float* rstd, const BnBank nb,
                                                       int relu, float* dz, int ldz, int M, int C, float invM) {
  V4_INDEX(M, C);
  const unsigned k = c / (unsigned)nb.Cw, cl = c - k * nb.Cw;
  const float4 av = ld4(a + (size_t)m * lda + c), dv = ld4(dy + (size_t)m * ldy + c), mv = ldp4(mu + c), rv = ldp4(rstd + c), gv = ldp4(nb.gamma[k] + cl),
               s1 = ldp4(nb.sdy[k] + cl), s2 = ldp4(nb.sdyxh[k] + cl);
  auto one = [&](float a_, float d_, float m_, float r_, float g_, float p_, float q_) {
    const float ah = (a_ - m_) * r_;
    const float da = g_ * r_ * (d_ - p_ * invM - ah * q_ * invM);
    return (relu && !(a_ > 0.f)) ? 0.f : da;
  };
  st4(dz + (size_t)m * ldz + c, make_float4(one(av.x, dv.x, mv.x, rv.x, gv.x, s1.x, s2.x), one(av.y, dv.y, mv.y, rv.y, gv.y, s1.y, s2.y),
                                            one(av.z, dv.z, mv.z, rv.z, gv.z, s1.z, s2.z), one(av.w, dv.w, mv.w, rv.w, gv.w, s1.w, s2.w)));
}
__global__ void k_bn_finalize_bank(const float* mu, const float* S2, float* rstd, const BnBank nb, int C, float invM, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int k = c / nb.Cw, cl = c - k * nb.Cw;
  const float var = S2[c] * invM;
  rstd[c] = 1.0f / sqrtf(var + eps);
  if (nb.mov_mean[k]) nb.mov_mean[k][cl] = nb.mov_mean[k][cl] * momentum + mu[c] * (1.f - momentum);
  if (nb.mov_var[k]) nb.mov_var[k][cl] = nb.mov_var[k][cl] * momentum + var * (1.f - momentum);
}
// dz = [a > 0 if relu] * gamma*rstd * (dy - mean(dy) - ahat*mean(dy*ahat));  sdy = sum dy (= d beta), sdyxh = sum dy*ahat (= d gamma)
__global__ void k_bn_bwd(const float* a, int lda, const float* dy, int ldy, const float* mu, const float* rstd, const float* gamma,
                         const float* sdy, const float* sdyxh, int relu, float* dz, int ldz, int M, int C, float invM) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const int m = (int)(i / C), c = (int)(i % C);
  const float av = a[(size_t)m * lda + c];
  const float ah = (av - mu[c]) * rstd[c];
  const float da = gamma[c] * rstd[c] * (dy[(size_t)m * ldy + c] - sdy[c] * invM - ah * sdyxh[c] * invM);
  dz[(size_t)m * ldz + c] = (relu && !(av > 0.f)) ? 0.f : da;
}

// ---- weight gradient: dw[tap][k][n] += sum_m x[m + tap - padl][k] * dy[m][n]  (fp32 MFMA 32x32x2, k-dimension = rows) ----
// Rows m = b*T + t never reach across batch rows (SAME zero padding, A.2); T <= 0: no time structure.  With `gather`
// row m of x is x[gather[m]] (embedding lookup fused into the first encoder prenet layer).  Workgroup = 4 waves =
// 64 (k) x 64 (n) tile over `rpb` rows; partial sums leave through fp32 atomics.
struct WgArgs { const float* x; const int* gather; const float* dy; float* dw; int ldx, ldy, lddw, M, T, K, N, kw, padl, rpb;
                const int* ygather;
                float* part; };   // deterministic mode: the M-slice's tile goes to part[slice][tap][K][N]; k_wgrad_reduce sums the slices in order   // optional: row m of dy is dy[ygather[m]] (first-step terms of recurrent kernels with ragged lengths)
typedef float wg_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_wgrad(const WgArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k0 = blockIdx.x * 64 + (wave >> 1) * 32, n0 = blockIdx.y * 64 + (wave & 1) * 32;
  const int nsplit = (g.M + g.rpb - 1) / g.rpb;
  const int tap = blockIdx.z / nsplit, sp = blockIdx.z - tap * nsplit;
  const int shift = tap - g.padl;
  const int m0 = sp * g.rpb, m1 = min(g.M, m0 + g.rpb);
  if (k0 >= g.K || n0 >= g.N) return;
  const int i = lane & 31, kk = lane >> 5;
  const bool kok = (k0 + i) < g.K, nok = (n0 + i) < g.N;
  wg_f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // WG_U row pairs per trip: all 2*WG_U loads are issued before the first MFMA (one memory round trip per trip, not per row pair)
  constexpr int WG_U = 8;
  for (int mb = m0 + kk; mb < m1; mb += 2 * WG_U) {
    float av[WG_U], bv[WG_U];
#pragma unroll
    for (int u = 0; u < WG_U; ++u) {
      const int m = mb + 2 * u;
      av[u] = 0.f; bv[u] = 0.f;
      if (m < m1) {
        bool rok = true;
        if (g.T > 0) { const int ts = m % g.T + shift; rok = (ts >= 0) && (ts < g.T); }
        if (kok && rok) {
          const size_t xr = g.gather ? (size_t)g.gather[m] : (size_t)(m + shift);
          av[u] = g.x[xr * g.ldx + k0 + i];
        }
        if (nok) bv[u] = g.dy[(size_t)(g.ygather ? g.ygather[m] : m) * g.ldy + n0 + i];
      }
    }
#pragma unroll
    for (int u = 0; u < WG_U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
  }
  if (g.part) {
    float* pw = g.part + ((size_t)sp * g.kw + tap) * g.K * g.N;
    if (nok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kr = k0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (kr < g.K) pw[(size_t)kr * g.N + n0 + i] = acc[r];
      }
    }
    return;
  }
  float* dw = g.dw + (size_t)tap * g.K * g.lddw;
  if (nok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kr = k0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (kr < g.K) atomicAdd(dw + (size_t)kr * g.lddw + n0 + i, acc[r]);
    }
  }
}
// ---- the same weight gradient on the bf16 matrix cores (three-way split operands, six products, fp32 accumulation) ----
// v_mfma_f32_32x32x16_bf16 contracts 16 rows m per instruction at 16x the rate of the fp32-input MFMA.  Its operand layout wants, per
// lane (i = lane & 31, h = lane >> 5), EIGHT consecutive contraction indices at a fixed output index: X[m0 + 8h + e][k0 + i], e < 8 --
// eight 4-byte loads, each coalesced across the 32 lanes of a half-wave (consecutive k of one row), straight from global memory in
// fragment order: no LDS, no transpose.  The fp32 values are split in registers (wgb_split3).  Workgroup = 4 waves = a 128 (k) x 128 (n) tile over `rpb` rows; wave (wk, wn) owns the 64 x 64 block
// (2 x 2 MFMA tiles, 64 accumulator registers), so every loaded fragment feeds two tiles.  The next 16 rows are requested before
// the MFMAs of the current 16.  Masks, gathers and tap shifts as in k_wgrad; partial sums leave through atomics or, in
// deterministic mode, as per-slice partial tiles.
// Gradients are sums of products that largely cancel (a weight gradient is small next to the sum of the magnitudes of its terms), so
// the 2^-17 per-product error of the two-way split that serves the inference GEMMs shows: here every operand is split THREE ways,
// x = hi + mid + lo exactly to 24 bits (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)), and a tile pair issues the six
// products down to 2^-24 of the result -- lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi -- fp32-grade, on the bf16 pipe.  The kernel is
// bound by operand delivery, not by the matrix pipe: the three extra MFMAs ride in the shadow of the loads.
__device__ __forceinline__ void wgb_split3(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    h[p] = taco_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h[p] << 16), rb = b - __uint_as_float(h[p] & 0xffff0000u);
    m[p] = taco_pk_bf16(ra, rb);
    l[p] = taco_pk_bf16(ra - __uint_as_float(m[p] << 16), rb - __uint_as_float(m[p] & 0xffff0000u));
  }
  hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  mid = __builtin_bit_cast(bf16x8, make_uint4(m[0], m[1], m[2], m[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}
// NW = waves per workgroup: 4 (128 x 128 tile) or 1 (64 x 64 tile: small matrices with many rows, where a finer tiling buys the
// workgroups that would otherwise have to come from splitting M -- every M-slice ends in 4096 atomics per wave).
template <int NW>
__device__ __forceinline__ void wgrad_bf3_body(const WgArgs& g, int bx, int by, int bz) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  constexpr int TS = NW == 4 ? 128 : 64;
  const int kb = bx * TS + (NW == 4 ? (wave >> 1) * 64 : 0), nb = by * TS + (NW == 4 ? (wave & 1) * 64 : 0);
  const int nsplit = (g.M + g.rpb - 1) / g.rpb;
  const int tap = bz / nsplit, sp = bz - tap * nsplit;
  const int shift = tap - g.padl;
  const int m0 = sp * g.rpb, m1 = min(g.M, m0 + g.rpb);
  if (kb >= g.K || nb >= g.N) return;
  const bool k2 = kb + 32 < g.K, n2 = nb + 32 < g.N;               // second tile of the pair inside the matrix? (wave-uniform)
  const bool full = (kb + 64 <= g.K) && (nb + 64 <= g.N) && !g.gather && !g.ygather;   // wave-uniform: no column masks, plain row indices
  bool kok[2], nok[2];
  kok[0] = kb + i < g.K; kok[1] = kb + 32 + i < g.K; nok[0] = nb + i < g.N; nok[1] = nb + 32 + i < g.N;
  wg_f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float av[2][8], bv[2][8];
  auto fetch = [&](int mb) {          // this lane's eight rows mb + 8h .. mb + 8h + 7 of both tile pairs
    // interior block (wave-uniform test on the scalar unit): all 16 rows exist, lie in ONE batch row before the tap shift (tb + 15 < T:
    // with a negative shift a block that straddles two batch rows would otherwise pass and read the previous row's tail where the
    // padding's zeros belong) and stay inside it after the shift -> sixteen unconditional loads per operand from 32-bit offsets
    const int tb = (g.T > 0) ? (mb % g.T) : 0;
    const bool inner = full && (mb + 16 <= m1) && (g.T <= 0 || (tb + 15 < g.T && tb + shift >= 0 && tb + 15 + shift < g.T));
    if (inner) {
      const unsigned xo = (unsigned)(mb + 8 * h + shift) * (unsigned)g.ldx + (unsigned)(kb + i);
      const unsigned yo = (unsigned)(mb + 8 * h) * (unsigned)g.ldy + (unsigned)(nb + i);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        av[0][e] = g.x[xo + (unsigned)e * (unsigned)g.ldx]; av[1][e] = g.x[xo + (unsigned)e * (unsigned)g.ldx + 32u];
        bv[0][e] = g.dy[yo + (unsigned)e * (unsigned)g.ldy]; bv[1][e] = g.dy[yo + (unsigned)e * (unsigned)g.ldy + 32u];
      }
      return;
    }
    const int mr = mb + 8 * h;
    int tt = (g.T > 0) ? (mr % g.T) : 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = mr + e;
      bool rok = m < m1;
      if (g.T > 0) { const int ts = tt + shift; rok = rok && ts >= 0 && ts < g.T; if (++tt == g.T) tt = 0; }
      const bool mok = m < m1;
      size_t xr = 0, yr = 0;
      if (mok) { xr = g.gather ? (size_t)g.gather[m] : (size_t)(m + shift); yr = g.ygather ? (size_t)g.ygather[m] : (size_t)m; }
      const float* xp = g.x + xr * g.ldx + kb + i;
      const float* yp = g.dy + yr * g.ldy + nb + i;
      av[0][e] = (rok && kok[0]) ? xp[0] : 0.f;
      av[1][e] = (rok && kok[1]) ? xp[32] : 0.f;
      bv[0][e] = (mok && nok[0]) ? yp[0] : 0.f;
      bv[1][e] = (mok && nok[1]) ? yp[32] : 0.f;
    }
  };
  fetch(m0);
  for (int mb = m0; mb < m1; mb += 16) {
    bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) { wgb_split3(av[a], ah[a], am[a], al[a]); wgb_split3(bv[a], bh[a], bm[a], bl[a]); }
    if (mb + 16 < m1) fetch(mb + 16);                                // in flight across the MFMAs below
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if (a == 1 && !k2) break;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (b == 1 && !n2) break;
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);      // small terms first
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bm[b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bh[b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bm[b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
      }
    }
  }
  float* out = g.part ? g.part + ((size_t)sp * g.kw + tap) * g.K * g.N : g.dw + (size_t)tap * g.K * g.lddw;
  const int ldo = g.part ? g.N : g.lddw;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if ((a == 1 && !k2) || (b == 1 && !n2) || !nok[b]) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kr = kb + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (kr < g.K) {
          float* p = out + (size_t)kr * ldo + nb + 32 * b + i;
          if (g.part) *p = acc[a][b][r]; else atomicAdd(p, acc[a][b][r]);
        }
      }
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_wgrad_bf3(const WgArgs g) { wgrad_bf3_body<NW>(g, blockIdx.x, blockIdx.y, blockIdx.z); }
// Several small weight gradients whose operands are all ready, as ONE launch: a 64 x 64-tile problem with a few thousand rows is a
// handful of one-wave workgroups that mostly wait for their first loads, and a training step has ~100 of them (the decoder's hoisted
// gradients: 52, of which 32 are the per-row d values products; a conv bank: one per width; a BiGRU: 8).  Workgroup b of the flat grid
// belongs to problem p with start[p] <= b < start[p + 1] and is block (bx, by, bz) of that problem's own grid.
#define WG_MAXP 36
struct WgGroup { WgArgs p[WG_MAXP]; int start[WG_MAXP + 1]; int n; };
__global__ __launch_bounds__(64) void k_wgrad_bf3_group(const WgGroup G) {
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < G.n && G.start[p + 1] <= b) ++p;
  p = __builtin_amdgcn_readfirstlane(p);
  const WgArgs g = G.p[p];
  const int local = b - G.start[p], gx = (g.K + 63) / 64, gy = (g.N + 63) / 64;
  wgrad_bf3_body<1>(g, local % gx, (local / gx) % gy, local / (gx * gy));
}

// dw[tap][k][n] += sum over the M-slices, slice 0 first (fixed order: run-to-run reproducible)
__global__ void k_wgrad_reduce(const float* part, int nsplit, int kw, int K, int N, float* dw, int lddw) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, per = (size_t)kw * K * N;
  if (i >= per) return;
  float s = 0.f;
#pragma unroll 8
  for (int sp = 0; sp < nsplit; ++sp) s += part[(size_t)sp * per + i];        // (independent loads, eight in flight; the adds keep their order)
  const unsigned tk = (unsigned)i / (unsigned)N, n = (unsigned)i - tk * (unsigned)N;      // tk = tap * K + k   (per < 2^32)
  dw[(size_t)tk * lddw + n] += s;
}

// the ordered sums of the problems of a group launch (deterministic mode) as ONE launch: block b of the flat grid belongs to problem p
// with start[p] <= b < start[p + 1] and sums 256 elements of its weight gradient over the M-slices, slice 0 first
struct WgRed { const float* part; float* dw; int nsplit, kw, K, N, lddw; };
struct WgRedGroup { WgRed p[WG_MAXP]; int start[WG_MAXP + 1]; int n; };
__global__ __launch_bounds__(256) void k_wgrad_reduce_group(const WgRedGroup G) {
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < G.n && G.start[p + 1] <= b) ++p;
  p = __builtin_amdgcn_readfirstlane(p);
  const WgRed g = G.p[p];
  const size_t i = (size_t)(b - G.start[p]) * 256 + threadIdx.x, per = (size_t)g.kw * g.K * g.N;
  if (i >= per) return;
  float s = 0.f;
#pragma unroll 8
  for (int sp = 0; sp < g.nsplit; ++sp) s += g.part[(size_t)sp * per + i];
  const unsigned tk = (unsigned)i / (unsigned)g.N, n = (unsigned)i - tk * (unsigned)g.N;
  g.dw[(size_t)tk * g.lddw + n] += s;
}

// ---- small element-wise pieces ----
// max_pooling1d(width w, stride 1, 'same') forward (materialised for the tape) and backward (gradient to the first maximum)
__global__ void k_maxpool_fwd(const float* x, float* y, int M, int T, int C, int w) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const int m = (int)(i / C), c = (int)(i % C), t = m % T, pl = (w - 1) >> 1;
  float best = -INFINITY;
  for (int j = 0; j < w; ++j) { const int tt = t - pl + j; if (tt >= 0 && tt < T) best = fmaxf(best, x[(size_t)(m - pl + j) * C + c]); }
  y[i] = best;
}
__global__ void k_maxpool_bwd(const float* x, const float* dp, float* dx, int M, int T, int C, int w) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const int m = (int)(i / C), c = (int)(i % C), t = m % T, pl = (w - 1) >> 1;
  float acc = 0.f;
  for (int tp = t - (w - 1 - pl); tp <= t + pl; ++tp) {      // output positions whose window contains t
    if (tp < 0 || tp >= T) continue;
    int arg = -1; float best = -INFINITY;
    for (int j = 0; j < w; ++j) {
      const int tt = tp - pl + j;
      if (tt < 0 || tt >= T) continue;
      const float v = x[(size_t)(m + tt - t) * C + c];
      if (v > best) { best = v; arg = tt; }
    }
    if (arg == t) acc += dp[(size_t)(m + tp - t) * C + c];
  }
  dx[i] = acc;
}
// max_pooling1d's backward pass, four columns per thread, with the pooled tensor's input recomputed from the bank's activations (k_bn_pool_bank_v4 stores no BatchNorm output)
__global__ __launch_bounds__(256) void k_maxpool_bwd_bn_v4(const float* a, int lda, const float* mu, const float* rstd, const BnBank nb, const float* dp, float* dx, int M, int T, int C, int w) {
  V4_INDEX(M, C);
  const unsigned kk = c / (unsigned)nb.Cw, cl = c - kk * nb.Cw;
  const float4 mv = ldp4(mu + c), rv = ldp4(rstd + c), gv = ldp4(nb.gamma[kk] + cl), bv = ldp4(nb.beta[kk] + cl);
  const int t = (int)(m % (unsigned)T), pl = (w - 1) >> 1;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int tp = t - (w - 1 - pl); tp <= t + pl; ++tp) {      // output positions whose window contains t
    if (tp < 0 || tp >= T) continue;
    int ax = -1, ay = -1, az = -1, aw = -1;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int j = 0; j < w; ++j) {
      const int tt = tp - pl + j;
      if (tt < 0 || tt >= T) continue;
      const float4 v = bn_out4(ld4(a + (size_t)((int)m + tt - t) * lda + c), mv, rv, gv, bv);
      if (v.x > best.x) { best.x = v.x; ax = tt; }
      if (v.y > best.y) { best.y = v.y; ay = tt; }
      if (v.z > best.z) { best.z = v.z; az = tt; }
      if (v.w > best.w) { best.w = v.w; aw = tt; }
    }
    const float4 d = ld4(dp + (size_t)((int)m + tp - t) * C + c);
    if (ax == t) acc.x += d.x;
    if (ay == t) acc.y += d.y;
    if (az == t) acc.z += d.z;
    if (aw == t) acc.w += d.w;
  }
  st4(dx + (size_t)m * C + c, acc);
}
// highway y = H*T + x*(1-T): dcat = [dH_pre | dT_pre] (input of the transposed GEMM), dxd = direct path dy*(1-T)
__global__ __launch_bounds__(256) void k_highway_bwd_v4(const float* dy, const float* x, const float* H, const float* Tg, float* dcat, float* dxd, int M, int D) {
  V4_INDEX(M, D);
  const size_t i = (size_t)m * D + c;
  const float4 g = ld4(dy + i), h = ld4(H + i), tg = ld4(Tg + i), xv = ld4(x + i);
  st4(dcat + (size_t)m * 2 * D + c, make_float4((h.x > 0.f) ? g.x * tg.x : 0.f, (h.y > 0.f) ? g.y * tg.y : 0.f, (h.z > 0.f) ? g.z * tg.z : 0.f, (h.w > 0.f) ? g.w * tg.w : 0.f));
  st4(dcat + (size_t)m * 2 * D + D + c, make_float4(g.x * (h.x - xv.x) * tg.x * (1.f - tg.x), g.y * (h.y - xv.y) * tg.y * (1.f - tg.y),
                                                    g.z * (h.z - xv.z) * tg.z * (1.f - tg.z), g.w * (h.w - xv.w) * tg.w * (1.f - tg.w)));
  st4(dxd + i, make_float4(g.x * (1.f - tg.x), g.y * (1.f - tg.y), g.z * (1.f - tg.z), g.w * (1.f - tg.w)));
}
__global__ void k_highway_bwd(const float* dy, const float* x, const float* H, const float* Tg, float* dcat, float* dxd, int M, int D) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * D) return;
  const int m = (int)(i / D), c = (int)(i % D);
  const float g = dy[i], h = H[i], tg = Tg[i];
  dcat[(size_t)m * 2 * D + c] = (h > 0.f) ? g * tg : 0.f;
  dcat[(size_t)m * 2 * D + D + c] = g * (h - x[i]) * tg * (1.f - tg);
  dxd[i] = g * (1.f - tg);
}
__global__ void k_relu_bwd(const float* dy, int lddy, const float* y, int ldy, float* dz, int ldz, int M, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const int m = (int)(i / C), c = (int)(i % C);
  dz[(size_t)m * ldz + c] = (y[(size_t)m * ldy + c] > 0.f) ? dy[(size_t)m * lddy + c] : 0.f;
}
// x[m, c] += vec[m / T, c]  (a per-utterance vector tiled over time)
__global__ void k_add_rowvec(float* x, const float* vec, int M, int T, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const int m = (int)(i / C), c = (int)(i % C);
  x[i] += vec[(size_t)(m / T) * C + c];
}
__global__ void k_add2d(float* dst, int ldd, const float* src, int lds, int M, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const int m = (int)(i / C), c = (int)(i % C);
  dst[(size_t)m * ldd + c] += src[(size_t)m * lds + c];
}
// d/d out of mean(|tgt - out| * coeff) (+ the priority-band terms of tacotron.py:283-296): g = sign(out - tgt) * coeff[b] * scale(c)
__global__ void k_l1_grad(const float* out, const float* tgt, const float* coeff, int rows, int T, int C, float scale, int c_lo,
                          int c_hi, float scale_band, float* g) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * C) return;
  const int row = (int)(i / C), c = (int)(i % C);
  const float d = out[i] - tgt[i];
  const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
  const float w = coeff ? coeff[row / T] : 1.f;
  g[i] = sg * w * (scale + ((c >= c_lo && c < c_hi) ? scale_band : 0.f));
}
__global__ void k_embed_bwd(const float* dx, const int* ids, float* dE, int M, int E) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * E) return;
  const int m = (int)(i / E), c = (int)(i % E);
  atomicAdd(dE + (size_t)ids[m] * E + c, dx[i]);
}
// deterministic variant: one WAVE per (table row v, 64 columns) walks the rows in order -- the ids come 64 at a time (one coalesced load and a
// ballot), only the matching rows are visited: the same summation order as one thread per element walking all M rows (round 4: 240 us per
// step at the C4 shard), without its M dependent id loads per thread
__global__ void k_embed_bwd_det(const float* dx, const int* ids, float* dE, int M, int E, int V) {
  const int wave = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  const int ncb = (E + 63) / 64, v = wave / ncb, c = (wave % ncb) * 64 + lane;
  if (v >= V) return;
  float s = 0.f;
  for (int m0 = 0; m0 < M; m0 += 64) {
    const int id = (m0 + lane < M) ? ids[m0 + lane] : -1;
    unsigned long long mask = __ballot(id == v);
    while (mask) {
      const int j = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      if (c < E) s += dx[(size_t)(m0 + j) * E + c];
    }
  }
  if (c < E) dE[(size_t)v * E + c] += s;
}
// y = softsign(z) = z / (1 + |z|)  =>  dz = dy * (1 - |y|)^2   (deepvoice speaker layers, tacotron.py:68-79)
__global__ void k_softsign_bwd(const float* dy, const float* y, float* dz, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float t = 1.f - fabsf(y[i]); dz[i] = dy[i] * t * t; }
}
// dst[(b*n + t)*ld + col0 + c] = src[b*C + c]: a per-utterance vector parked behind every step's row of a [B, n, .] tape ('simple' speaker mode)
__global__ void k_tile_rows(const float* src, float* dst, int ld, int col0, int B, int n, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * n * C) return;
  const int c = (int)(i % C); const size_t row = i / C;
  dst[row * ld + col0 + c] = src[(row / n) * C + c];
}
// out[b, c] = sum_t x[b, t, c]  (gradient of a vector broadcast over time: before_highway)
__global__ void k_time_sum(const float* x, float* out, int B, int T, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += x[((size_t)b * T + t) * C + c];
  out[i] = s;
}
// idx[b] = b*T + max(L_b - 1, 0): the row of the first backward-direction step of every sequence
__global__ void k_last_row_index(const int* lengths, int* idx, int B, int T) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { const int L = lengths ? lengths[b] : T; idx[b] = b * T + (L > 0 ? L - 1 : 0); }
}
// bah_norm (A.9): v_hat = g * v / |v|.  Forward fold (pack refresh) and its backward: dg = (dv_hat . v) / |v| ;
// dv = g/|v| * (dv_hat - v * (dv_hat . v) / |v|^2).  One workgroup.
__global__ __launch_bounds__(256) void k_vnorm_fold(const float* v, const float* g, float* vhat, int A) {
  __shared__ float sm[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < A; i += 256) s += v[i] * v[i];
  sm[threadIdx.x] = s; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
  const float sc = g[0] / sqrtf(sm[0]);
  for (int i = threadIdx.x; i < A; i += 256) vhat[i] = v[i] * sc;
}
__global__ __launch_bounds__(256) void k_vnorm_bwd(const float* v, const float* g, const float* dvhat, float* dv, float* dg, int A) {
  __shared__ float s1[256], s2[256];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < A; i += 256) { a += v[i] * v[i]; b += dvhat[i] * v[i]; }
  s1[threadIdx.x] = a; s2[threadIdx.x] = b; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; } __syncthreads(); }
  const float n2 = s1[0], dot = s2[0], nrm = sqrtf(n2);
  for (int i = threadIdx.x; i < A; i += 256) dv[i] += g[0] / nrm * (dvhat[i] - v[i] * dot / n2);
  if (threadIdx.x == 0) dg[0] += dot / nrm;
}
__global__ void k_sum_all(const float* x, int n, float* out) {   // out[0] += sum x  (attention score bias gradient)
  __shared__ float sm[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  sm[threadIdx.x] = s; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) atomicAdd(out, sm[0]);
}

// ---- GRUCell backward (A.6), split around the two transposed mat-vecs ----
//   h' = u*h + (1-u)*c ; c = tanh([x, r*h].Wc + bc) ; [r,u] = sigmoid([x,h].Wg + bg)
// a: dht = dout (+ carry); dcp = dht*(1-u)*(1-c^2); dgp[H..2H) = dht*(h-c)*u*(1-u)
__global__ void k_gru_bwd_a(const float* dout, int lddo, const float* carry, const float* u, int ldu, const float* c, int ldc,
                            const float* hprev, int ldh, float* dht, float* dcp, int lddcp, float* dgp, int lddgp, int B, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, n = i % H;
  const float g = dout[(size_t)b * lddo + n] + (carry ? carry[i] : 0.f);
  const float uu = u[(size_t)b * ldu + n], cc = c[(size_t)b * ldc + n];
  const float hp = hprev ? hprev[(size_t)b * ldh + n] : 0.f;
  dht[i] = g;
  dcp[(size_t)b * lddcp + n] = g * (1.f - uu) * (1.f - cc * cc);
  dgp[(size_t)b * lddgp + H + n] = g * (hp - cc) * uu * (1.f - uu);
}
// b: tmp1 = dcp . Wc^T  ([B, I+H]); d(r*h) = tmp1[I..): dgp[0..H) = d(rh)*h*r*(1-r); dhp = dht*u + d(rh)*r
__global__ void k_gru_bwd_b(const float* tmp1, int ld1, int I, const float* hprev, int ldh, const float* r, int ldr, const float* u,
                            int ldu, const float* dht, float* dgp, int lddgp, float* dhp, int B, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, n = i % H;
  const float drh = tmp1[(size_t)b * ld1 + I + n];
  const float hp = hprev ? hprev[(size_t)b * ldh + n] : 0.f;
  const float rr = r[(size_t)b * ldr + n];
  dgp[(size_t)b * lddgp + n] = drh * hp * rr * (1.f - rr);
  dhp[i] = dht[i] * u[(size_t)b * ldu + n] + drh * rr;
}
// c: tmp2 = dgp . Wg^T ([B, I+H]): dx = tmp1[0..I) + tmp2[0..I) (+ dres), optionally masked by relu_of > 0 (the layer below is a
// ReLU: saves its own mask kernel); carry = dhp + tmp2[I..)
__global__ void k_gru_bwd_c(const float* tmp1, const float* tmp2, int ld, int I, const float* dres, int lddres, const float* dhp,
                            float* dx, int lddx, float* carry, int B, int H, const float* relu_of, int ldrelu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int W = I + H;
  if (i >= B * W) return;
  const int b = i / W, n = i % W;
  const float s = tmp1[(size_t)b * ld + n] + tmp2[(size_t)b * ld + n];
  if (n < I) {
    float v = s + (dres ? dres[(size_t)b * lddres + n] : 0.f);
    if (relu_of && !(relu_of[(size_t)b * ldrelu + n] > 0.f)) v = 0.f;
    dx[(size_t)b * lddx + n] = v;
  } else carry[(size_t)b * H + (n - I)] = dhp[(size_t)b * H + (n - I)] + tmp2[(size_t)b * ld + n];
}
// c of the upper GRU and a of the GRU below it in one launch (the residual stack: the lower cell's output gradient IS the dx just
// computed; I == H there): dx as above, then dht/dcp/dgp_u of the lower cell from it.
__global__ void k_gru_bwd_ca(const float* tmp1, const float* tmp2, int ld, int I, const float* dres, int lddres, const float* dhp,
                             float* dx, int lddx, float* carry, int B, int H,
                             const float* carry2, const float* u2, int ldu2, const float* c2, int ldc2, const float* hprev2, int ldh2,
                             float* dht2, float* dcp2, int lddcp2, float* dgp2, int lddgp2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int W = I + H;
  if (i >= B * W) return;
  const int b = i / W, n = i % W;
  const float s = tmp1[(size_t)b * ld + n] + tmp2[(size_t)b * ld + n];
  if (n < I) {
    const float g = s + (dres ? dres[(size_t)b * lddres + n] : 0.f);
    dx[(size_t)b * lddx + n] = g;
    // lower cell, unit n (I == H)
    const float gt = g + carry2[(size_t)b * H + n];
    const float uu = u2[(size_t)b * ldu2 + n], cc = c2[(size_t)b * ldc2 + n];
    const float hp = hprev2 ? hprev2[(size_t)b * ldh2 + n] : 0.f;
    dht2[(size_t)b * H + n] = gt;
    dcp2[(size_t)b * lddcp2 + n] = gt * (1.f - uu) * (1.f - cc * cc);
    dgp2[(size_t)b * lddgp2 + H + n] = gt * (hp - cc) * uu * (1.f - uu);
  } else carry[(size_t)b * H + (n - I)] = dhp[(size_t)b * H + (n - I)] + tmp2[(size_t)b * ld + n];
}

// ---- BiGRU backward scan: the mirror of k_bigru_rows (row-parallel, transposed h-weights streamed from L2) ----
struct BigruBArgs {
  const float* dout;    // [B*T, 2H] gradient of the BiGRU output
  const float* out;     // [B*T, 2H] forward output (= state sequence)
  const float* gsave;   // [B*T, 6H] (r | u | c) per direction, true time
  const float* wgT0; const float* wgT1;   // (h-rows of gates/kernel)^T     [2H, H]
  const float* wcT0; const float* wcT1;   // (h-rows of candidate/kernel)^T [H, H]
  const int* lengths;
  float* dg;            // [B*T, 6H] out: (d r_pre | d u_pre | d c_pre) per direction at the true time index (host pre-zeroes)
  float* rh;            // [B*T, 2H] out: r * h_prev (input rows of the candidate kernel's h part, for its weight gradient; host pre-zeroes)
  const float* h0;      // [B, 2H] initial states (fw | bw) or null (deepvoice encoder_rnn_init, modules.py:82-86)
  float* dh0;           // [B, 2H] out: gradient of the initial states (nullable)
  int B, T, H;
};
template <int R>
__global__ __launch_bounds__(RP_NT) void k_bigru_rows_bwd(const BigruBArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  BigruBArgs a = a_in;
  PIN(a.dout); PIN(a.out); PIN(a.gsave); PIN(a.wgT0); PIN(a.wgT1); PIN(a.wcT0); PIN(a.wcT1); PIN(a.lengths); PIN(a.dg); PIN(a.rh);
  PIN(a.h0); PIN(a.dh0); PIN(a.B); PIN(a.T); PIN(a.H);
  const int tid = threadIdx.x;
  const int ngrp = (a.B + R - 1) / R;
  const int d = blockIdx.x / ngrp, r0 = (blockIdx.x % ngrp) * R;
  const int B = a.B, T = a.T, H = a.H;
  const float* WgT = d ? a.wgT1 : a.wgT0;
  const float* WcT = d ? a.wcT1 : a.wcT0;
  float* dh = smem;                 // [R][H] carried gradient of the state
  float* v1 = dh + R * H;           // [R][H]  d c_pre
  float* v2 = v1 + R * H;           // [R][2H] d r_pre | d u_pre
  float* part = v2 + R * 2 * H;     // [KS][R][N]
  const int o = tid;                // one (row, unit) per thread; host guarantees R*H <= RP_NT
  const bool mine = o < R * H;
  const int rr_ = mine ? o / H : 0, n = mine ? o % H : 0, b = r0 + rr_;
  const bool brow = mine && b < B;
  const int L = brow ? (a.lengths ? a.lengths[b] : T) : 0;
  int Lmax = 0;
  for (int r = 0; r < R; ++r) { const int bb = r0 + r; if (bb < B) Lmax = max(Lmax, a.lengths ? a.lengths[bb] : T); }
  if (mine) dh[o] = 0.f;
  __syncthreads();
  for (int s = Lmax - 1; s >= 0; --s) {
    const bool active = brow && s < L;
    const int t = d ? (L - 1 - s) : s, tp = d ? t + 1 : t - 1;
    float keep = 0.f, hp = 0.f, rg = 0.f, dgu = 0.f, dcp = 0.f;
    if (mine) {
      keep = dh[o];
      float x1 = 0.f;
      if (active) {
        const size_t row = (size_t)b * T + t;
        const float g = keep + a.dout[row * 2 * H + d * H + n];
        rg = a.gsave[row * 6 * H + d * 3 * H + n];
        const float ug = a.gsave[row * 6 * H + d * 3 * H + H + n];
        const float cg = a.gsave[row * 6 * H + d * 3 * H + 2 * H + n];
        hp = (s > 0) ? a.out[((size_t)b * T + tp) * 2 * H + d * H + n] : (a.h0 ? a.h0[(size_t)b * 2 * H + d * H + n] : 0.f);
        dcp = g * (1.f - ug) * (1.f - cg * cg);
        dgu = g * (hp - cg) * ug * (1.f - ug);
        keep = g * ug;
        x1 = dcp;
        a.rh[row * 2 * H + d * H + n] = rg * hp;
      }
      v1[o] = x1;
      v2[rr_ * 2 * H + H + n] = active ? dgu : 0.f;
    }
    __syncthreads();
    int KS = rp_matvec<R>(WcT, H, H, v1, H, part, tid);
    __syncthreads();
    float dgr = 0.f;
    if (mine) {
      const float drh = rp_reduce(part, KS, R * H, o);
      if (active) { dgr = drh * hp * rg * (1.f - rg); keep += drh * rg; }
      v2[rr_ * 2 * H + n] = dgr;
    }
    __syncthreads();
    KS = rp_matvec<R>(WgT, 2 * H, H, v2, 2 * H, part, tid);
    __syncthreads();
    if (mine) {
      const float dhg = rp_reduce(part, KS, R * H, o);
      dh[o] = active ? keep + dhg : keep;
      if (active) {
        float* q = a.dg + ((size_t)b * T + t) * 6 * H + d * 3 * H;
        q[n] = dgr; q[H + n] = dgu; q[2 * H + n] = dcp;
      }
    }
    __syncthreads();
  }
  if (a.dh0 && brow) a.dh0[(size_t)b * 2 * H + d * H + n] = dh[o];
}

// ---- the same backward scan with the recurrent kernels RESIDENT in registers (round 5; H = 128: the encoder) ----
// k_bigru_rows_bwd streams both transposed kernels from L2 in every step and reads the step's tape values at the moment it needs them:
// 3.5 us per step at the C4 shard (0.45 ms per training step for 128 steps).  Here one workgroup of 4 H threads owns one (direction,
// row): thread (n, q) keeps the quarter q of unit n's ROWS of Wc_h and Wg_h (r half, u half) -- 3 H / 4 = 96 registers, the interleaved
// float4 chunks q, q + 4, ... so that the four lanes of a unit read 64 consecutive bytes of LDS --, a product is H / 4 FMAs and two DPP
// adds inside the quad, the d c_pre / gate-gradient vectors alternate between two LDS copies (two barriers per step), and the tape
// values of step s - 1 are requested while step s is computed.  Same recurrences, outputs and masking as k_bigru_rows_bwd.
struct BigruQArgs {
  const float* dout; const float* out; const float* gsave;      // as BigruBArgs
  const float* gh0; const float* gh1;     // h-rows of gates/kernel     [H, 2H] (TF layout), forward / backward direction
  const float* ch0; const float* ch1;     // h-rows of candidate/kernel [H, H]
  const int* lengths; float* dg; float* rh; const float* h0; float* dh0;
  int B, T;
};
template <int H>
__global__ __launch_bounds__(4 * H) void k_bigru_resb(const BigruQArgs a) {
  static_assert(H % 16 == 0 && 4 * H <= 1024, "four lanes per unit, float4 chunks dealt round the quad");
  constexpr int NC = H / 16;               // float4 chunks per lane and vector
  __shared__ __attribute__((aligned(16))) float v1[2][H], v2[2][2 * H];
  const int tid = threadIdx.x, n = tid >> 2, q = tid & 3;
  const int d = blockIdx.x / a.B, b = blockIdx.x - d * a.B;
  const int T = a.T, L = a.lengths ? a.lengths[b] : T;
  const float* gh = d ? a.gh1 : a.gh0; const float* ch = d ? a.ch1 : a.ch0;
  float wc[H / 4], wr[H / 4], wu[H / 4];
#pragma unroll
  for (int m = 0; m < NC; ++m) {
    const int j = 4 * (q + 4 * m);
    const float4 x = *reinterpret_cast<const float4*>(ch + (size_t)n * H + j);
    const float4 y = *reinterpret_cast<const float4*>(gh + (size_t)n * 2 * H + j);
    const float4 z = *reinterpret_cast<const float4*>(gh + (size_t)n * 2 * H + H + j);
    wc[4 * m] = x.x; wc[4 * m + 1] = x.y; wc[4 * m + 2] = x.z; wc[4 * m + 3] = x.w;
    wr[4 * m] = y.x; wr[4 * m + 1] = y.y; wr[4 * m + 2] = y.z; wr[4 * m + 3] = y.w;
    wu[4 * m] = z.x; wu[4 * m + 1] = z.y; wu[4 * m + 2] = z.z; wu[4 * m + 3] = z.w;
  }
  auto quad_sum = [](float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
    return x;
  };
  struct Tape { float dout, r, u, c, hp; };
  auto fetch = [&](int s) {                 // the tape values of step s (0 <= s < L) at their true time
    Tape p;
    const int t = d ? (L - 1 - s) : s, tp = d ? t + 1 : t - 1;
    const size_t row = (size_t)b * T + t;
    p.dout = a.dout[row * 2 * H + d * H + n];
    const float* gs = a.gsave + row * 6 * H + d * 3 * H + n;
    p.r = gs[0]; p.u = gs[H]; p.c = gs[2 * H];
    p.hp = (s > 0) ? a.out[((size_t)b * T + tp) * 2 * H + d * H + n] : (a.h0 ? a.h0[(size_t)b * 2 * H + d * H + n] : 0.f);
    return p;
  };
  float dh = 0.f;
  Tape nx = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (L > 0) nx = fetch(L - 1);
  for (int s = L - 1; s >= 0; --s) {
    const int p = s & 1;
    const Tape cur = nx;
    if (s > 0) nx = fetch(s - 1);           // in flight across this step
    const float g = dh + cur.dout;
    const float dcp = g * (1.f - cur.u) * (1.f - cur.c * cur.c);
    const float dgu = g * (cur.hp - cur.c) * cur.u * (1.f - cur.u);
    float keep = g * cur.u;
    if (q == 0) { v1[p][n] = dcp; v2[p][H + n] = dgu; }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int m = 0; m < NC; ++m) {
      const float4 x = *reinterpret_cast<const float4*>(&v1[p][4 * (q + 4 * m)]);
      acc = fmaf(x.x, wc[4 * m], acc); acc = fmaf(x.y, wc[4 * m + 1], acc); acc = fmaf(x.z, wc[4 * m + 2], acc); acc = fmaf(x.w, wc[4 * m + 3], acc);
    }
    const float drh = quad_sum(acc);
    const float dgr = drh * cur.hp * cur.r * (1.f - cur.r);
    keep += drh * cur.r;
    if (q == 0) v2[p][n] = dgr;
    __syncthreads();
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int m = 0; m < NC; ++m) {
      const float4 x = *reinterpret_cast<const float4*>(&v2[p][4 * (q + 4 * m)]);
      const float4 y = *reinterpret_cast<const float4*>(&v2[p][H + 4 * (q + 4 * m)]);
      a0 = fmaf(x.x, wr[4 * m], a0); a0 = fmaf(x.y, wr[4 * m + 1], a0); a0 = fmaf(x.z, wr[4 * m + 2], a0); a0 = fmaf(x.w, wr[4 * m + 3], a0);
      a1 = fmaf(y.x, wu[4 * m], a1); a1 = fmaf(y.y, wu[4 * m + 1], a1); a1 = fmaf(y.z, wu[4 * m + 2], a1); a1 = fmaf(y.w, wu[4 * m + 3], a1);
    }
    dh = keep + quad_sum(a0 + a1);
    if (q == 0) {
      const int t = d ? (L - 1 - s) : s;
      const size_t row = (size_t)b * T + t;
      a.rh[row * 2 * H + d * H + n] = cur.r * cur.hp;
      float* o = a.dg + row * 6 * H + d * 3 * H + n;
      o[0] = dgr; o[H] = dgu; o[2 * H] = dcp;
    }
  }
  if (a.dh0 && q == 0) a.dh0[(size_t)b * 2 * H + d * H + n] = dh;
}

// inclusive wave64 SUFFIX sum (lane l gets sum over lanes >= l); never formed as total - prefix, which cancels
// catastrophically when the tail is many orders of magnitude below the head (it is: the tail carries cumprod(1-p))
__device__ __forceinline__ float wave_rscan(float x, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float y = __shfl_down(x, o, 64);
    if (lane + o < 64) x += y;
  }
  return x;
}

// ---- attention backward ----
// In the BPTT loop only what the recurrence needs (k_attention_bwd, one 16-wave workgroup per batch row, the mirror of
// k_attention): d alpha(t) = carry + dctx . V ; normaliser backward -> d e(t), d alpha(t-1) ; d q = sum_j de_j v (1-th^2) ;
// d h_att += d q . Wq^T.  The forward saved q(t) and the raw scores e(t), so nothing is recomputed but tanh.
// Everything that is a plain sum over steps is hoisted out of the loop: d values (a [T_in x n].[n x D] product per batch
// row, k_wgrad), d keys and d attention_v (k_attention_keys_bwd, parallel over (row, position, channel)).
struct AttnBArgs {
  const float* q; const float* e;                          // tape: processed query [B, ldq], raw scores [B, lde] of step t
  const float* wqT;                                        // query kernel transposed [A, As]
  const float* keys; const float* values; const float* v; const float* score_bias;
  const float* battn;   // attention_b (bah_norm) or null: added to the query inside the tanh
  const float* alpha; const float* alpha_prev;             // tape rows [B, ldal]
  float* dctx;          // [B, lddctx] total gradient of context(t) (also copied to the tape slice dctx_out)
  float* dctx_out;      // [B, lddco]
  float* dalpha;        // [B, T_in] in: gradient of alpha(t) arriving from step t+1; out: gradient of alpha(t-1)
  float* de_out;        // [B, ldde] tape: gradient of the raw scores of step t
  float* dsb_acc;       // [B] per-row accumulator of d score_bias
  float* dq; float* dhq;                                   // out [B, lddq]; in/out [B, lddhq]: += dq . Wq^T
  const float* cat;     // nullable [B, ldcat] = d(concat projection input): columns [0,As) are added to dhq, [As,As+D) to dctx first
  int ldq, lde, ldal, lddctx, lddco, ldde, lddq, lddhq, T_in, A, D, As, type, ldcat;
};
#define ATB_NW 16
__global__ __launch_bounds__(64 * ATB_NW) void k_attention_bwd(const AttnBArgs a_in) {
  AttnBArgs a = a_in;
  PIN(a.q); PIN(a.e); PIN(a.wqT); PIN(a.keys); PIN(a.values); PIN(a.v); PIN(a.score_bias); PIN(a.battn); PIN(a.alpha); PIN(a.alpha_prev);
  PIN(a.dctx); PIN(a.dctx_out); PIN(a.dalpha); PIN(a.de_out); PIN(a.dsb_acc); PIN(a.dq); PIN(a.dhq); PIN(a.cat); PIN(a.ldcat);
  // dynamic LDS (host: attn_bwd_lds_bytes): q[A4] dq[A4] dctx[D4] | p cp ss da de [T4 each] | red[ATB_NW*256]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int A4 = (a_in.A + 3) & ~3, D4 = (a_in.D + 3) & ~3, T4 = (a_in.T_in + 3) & ~3;
  float* qs = smem; float* dqv = qs + A4; float* dcx = dqv + A4;
  float* p = dcx + D4; float* cp = p + T4; float* ss = cp + T4; float* da = ss + T4; float* de = da + T4;
  float* w1 = da;                      // d alpha_prev overwrites d alpha position by position (each is read before it is written)
  float* red = de + T4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, T = a.T_in, A = a.A, D = a.D, As = a.As;
  const float* krow = a.keys + (size_t)b * T * A;
  const float* vrow = a.values + (size_t)b * T * D;
  const float* al = a.alpha + (size_t)b * a.ldal;
  const float* alp = a.alpha_prev + (size_t)b * a.ldal;
  for (int i = tid; i < A; i += 64 * ATB_NW) qs[i] = a.q[(size_t)b * a.ldq + i] + (a.battn ? a.battn[i] : 0.f);
  for (int i = tid; i < D; i += 64 * ATB_NW) {
    float x = a.dctx[(size_t)b * a.lddctx + i];
    if (a.cat) x += a.cat[(size_t)b * a.ldcat + As + i];
    dcx[i] = x; a.dctx_out[(size_t)b * a.lddco + i] = x;
  }
  for (int j = tid; j < T; j += 64 * ATB_NW) de[j] = a.e[(size_t)b * a.lde + j];
  __syncthreads();
  // d alpha_j = carry_j + dctx . V_j : one wave per position, lanes over channels (float4)
  for (int j = wave; j < T; j += ATB_NW) {
    float sd = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
      const float4 v4 = *reinterpret_cast<const float4*>(vrow + (size_t)j * D + c);
      const float4 d4 = *reinterpret_cast<const float4*>(&dcx[c]);
      sd += v4.x * d4.x + v4.y * d4.y + v4.z * d4.z + v4.w * d4.w;
    }
    sd = wave_sum(sd);
    if (lane == 0) da[j] = a.dalpha[(size_t)b * T + j] + sd;
  }
  __syncthreads();
  // normaliser backward (single wave, chunked scans like the forward)
  if (wave == 0) {
    const int C = (T + 63) >> 6, j0 = lane * C, j1 = min(j0 + C, T);
    if (a.type == 2) {
      const float sb = a.score_bias ? a.score_bias[0] : 0.f;
      // forward recompute: p, cp (exclusive cumprod through logs), s = cumsum(prev / clip(cp))
      float run = 0.f;
      for (int j = j0; j < j1; ++j) {
        const float pj = taco_sigmoid(de[j] + sb);
        p[j] = pj; cp[j] = run;
        run += logf(fminf(fmaxf(1.f - pj, 1.17549435e-38f), 1.f));
      }
      const float off = wave_scan(run, lane) - run;
      float run2 = 0.f;
      for (int j = j0; j < j1; ++j) {
        const float c1 = expf(cp[j] + off);
        cp[j] = c1;
        run2 += alp[j] / fminf(fmaxf(c1, 1e-10f), 1.f);
        ss[j] = run2;
      }
      const float off2 = wave_scan(run2, lane) - run2;
      for (int j = j0; j < j1; ++j) ss[j] += off2;
      // backward: ds_j = da*p*cp ; rc = reverse inclusive cumsum(ds)
      float tot = 0.f;
      for (int j = j0; j < j1; ++j) tot += da[j] * p[j] * cp[j];
      float suffix = wave_rscan(tot, lane) - tot;   // sum over later lanes' chunks
      float dLsum = 0.f;                            // chunk total of dL (for the second reverse scan)
      for (int j = j1 - 1; j >= j0; --j) {
        suffix += da[j] * p[j] * cp[j];             // rc_j
        const float cj = fminf(fmaxf(cp[j], 1e-10f), 1.f);
        float dcp = da[j] * p[j] * ss[j];
        if (cp[j] >= 1e-10f && cp[j] <= 1.f) dcp += -suffix * alp[j] / (cj * cj);
        const float dL = dcp * cp[j];
        ss[j] = da[j] * cp[j] * ss[j];              // dp_j (direct term); ss no longer needed as s
        cp[j] = dL;                                 // reuse: cp now holds dL
        w1[j] = suffix / cj;                        // d alpha_prev_j (aliases da[j]: last use of da[j] is above)
        dLsum += dL;
      }
      float suf2 = wave_rscan(dLsum, lane) - dLsum; // sum of dL over later chunks
      float dsb = 0.f;
      for (int j = j1 - 1; j >= j0; --j) {
        const float dlg = suf2;                     // reverse EXCLUSIVE cumsum: sum_{k>j} dL_k
        suf2 += cp[j];
        const float xj = 1.f - p[j];
        float dp = ss[j];
        if (xj >= 1.17549435e-38f && xj <= 1.f) dp -= dlg / xj;
        const float dej = dp * p[j] * (1.f - p[j]);
        de[j] = dej; dsb += dej;
      }
      dsb = wave_sum(dsb);
      if (lane == 0 && a.dsb_acc) a.dsb_acc[b] += dsb;
      for (int j = j0; j < j1; ++j) a.dalpha[(size_t)b * T + j] = w1[j];
    } else {
      // softmax: de_j = alpha_j * (da_j - sum_k alpha_k da_k); no dependence on the previous alignments
      float dot = 0.f;
      for (int j = j0; j < j1; ++j) dot += al[j] * da[j];
      dot = wave_sum(dot);
      for (int j = j0; j < j1; ++j) { de[j] = al[j] * (da[j] - dot); a.dalpha[(size_t)b * T + j] = 0.f; }
    }
  }
  __syncthreads();
  for (int j = tid; j < T; j += 64 * ATB_NW) a.de_out[(size_t)b * a.ldde + j] = de[j];
  // d q_c = v_c * sum_j de_j * (1 - tanh^2(K_jc + q_c)) : 16 lanes per position (4 channels each per 64-channel stripe), like the forward
  {
    const int l16 = lane & 15, grp = lane >> 4;
    for (int c0 = 0; c0 < A; c0 += 256) {
      float4 acc[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = wave * 4 + grp; j < T; j += 4 * ATB_NW) {
        const float dj = de[j];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int c = c0 + l16 * 4 + 64 * m;
          if (c < A) {
            const float4 k4 = *reinterpret_cast<const float4*>(krow + (size_t)j * A + c);
            const float4 q4 = *reinterpret_cast<const float4*>(&qs[c]);
            const float t0 = taco_tanh_fast(k4.x + q4.x), t1 = taco_tanh_fast(k4.y + q4.y), t2 = taco_tanh_fast(k4.z + q4.z), t3 = taco_tanh_fast(k4.w + q4.w);
            acc[m].x += dj * (1.f - t0 * t0); acc[m].y += dj * (1.f - t1 * t1); acc[m].z += dj * (1.f - t2 * t2); acc[m].w += dj * (1.f - t3 * t3);
          }
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {      // fold the 4 positions a wave works on at once
        acc[m].x += __shfl_xor(acc[m].x, 16, 64); acc[m].x += __shfl_xor(acc[m].x, 32, 64);
        acc[m].y += __shfl_xor(acc[m].y, 16, 64); acc[m].y += __shfl_xor(acc[m].y, 32, 64);
        acc[m].z += __shfl_xor(acc[m].z, 16, 64); acc[m].z += __shfl_xor(acc[m].z, 32, 64);
        acc[m].w += __shfl_xor(acc[m].w, 16, 64); acc[m].w += __shfl_xor(acc[m].w, 32, 64);
        if (grp == 0) *reinterpret_cast<float4*>(&red[wave * 256 + l16 * 4 + 64 * m]) = acc[m];
      }
      __syncthreads();
      if (tid < 256 && c0 + tid < A) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < ATB_NW; ++w) s += red[w * 256 + tid];
        s *= a.v[c0 + tid];
        dqv[c0 + tid] = s; a.dq[(size_t)b * a.lddq + c0 + tid] = s;
      }
      __syncthreads();
    }
  }
  // dhq += dq . Wq^T : 4 columns per thread, K split over thread groups, reduced through LDS
  {
    const int NC = As >> 2;
    int KS = (64 * ATB_NW) / NC; if (KS > A) KS = A; if (KS * As > ATB_NW * 256) KS = (ATB_NW * 256) / As; if (KS < 1) KS = 1;
    const int kper = (A + KS - 1) / KS, cg = tid % NC, ks = tid / NC;
    if (ks < KS) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4* wp = reinterpret_cast<const float4*>(a.wqT) + cg;
      const int k1 = min(A, (ks + 1) * kper);
      for (int k = ks * kper; k < k1; ++k) {
        const float4 w = wp[(size_t)k * NC]; const float xv = dqv[k];
        acc.x = fmaf(xv, w.x, acc.x); acc.y = fmaf(xv, w.y, acc.y); acc.z = fmaf(xv, w.z, acc.z); acc.w = fmaf(xv, w.w, acc.w);
      }
      *reinterpret_cast<float4*>(&red[(size_t)ks * As + 4 * cg]) = acc;
    }
    __syncthreads();
    for (int k = tid; k < As; k += 64 * ATB_NW) {
      float s = 0.f;
      for (int k2 = 0; k2 < KS; ++k2) s += red[(size_t)k2 * As + k];
      if (a.cat) s += a.cat[(size_t)b * a.ldcat + k];
      a.dhq[(size_t)b * a.lddhq + k] += s;
    }
  }
}

// hoisted out of the BPTT loop: dkeys[b,j,c] = v_c * sum_t de_t[j] * (1 - th^2), dv_c += sum_{b,j,t} de_t[j] * th,
// th = tanh(keys[b,j,c] + q_t[b,c]).  Workgroup = 256 channels x ATK_J positions of one batch row, loop over the steps.
#define ATK_J 8
struct AttnKArgs { const float* keys; const float* q; const float* de; const float* v; const float* battn; float* dkeys; float* dv; int T_in, A, n;
                   float* part; };   // deterministic mode: part[(b * gridDim.y + position chunk)][A] instead of atomics on dv; k_colsum_reduce-style sum afterwards
__global__ __launch_bounds__(256) void k_attention_keys_bwd(const AttnKArgs a) {
  __shared__ float sde[ATK_J];
  const int tid = threadIdx.x;
  const int b = blockIdx.z, j0 = blockIdx.y * ATK_J, c = blockIdx.x * 256 + tid;
  const bool cok = c < a.A;
  float kv[ATK_J], ak[ATK_J], av = 0.f;
#pragma unroll
  for (int u = 0; u < ATK_J; ++u) { kv[u] = (cok && j0 + u < a.T_in) ? a.keys[((size_t)b * a.T_in + j0 + u) * a.A + c] : 0.f; ak[u] = 0.f; }
  for (int t = 0; t < a.n; ++t) {
    __syncthreads();
    if (tid < ATK_J) sde[tid] = (j0 + tid < a.T_in) ? a.de[((size_t)b * a.n + t) * a.T_in + j0 + tid] : 0.f;
    __syncthreads();
    const float qc = cok ? a.q[((size_t)b * a.n + t) * a.A + c] + (a.battn ? a.battn[c] : 0.f) : 0.f;
#pragma unroll
    for (int u = 0; u < ATK_J; ++u) {
      const float th = taco_tanh_fast(kv[u] + qc);
      ak[u] += sde[u] * (1.f - th * th);
      av += sde[u] * th;
    }
  }
  if (cok) {
    const float vc = a.v[c];
#pragma unroll
    for (int u = 0; u < ATK_J; ++u)
      if (j0 + u < a.T_in) a.dkeys[((size_t)b * a.T_in + j0 + u) * a.A + c] = vc * ak[u];
    if (a.part) a.part[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * a.A + c] = av;
    else atomicAdd(a.dv + c, av);
  }
}
__global__ void k_rows_reduce(const float* part, int nrows, int C, float* out) {      // out[c] += sum_r part[r][c], row 0 first
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
#pragma unroll 8
  for (int r = 0; r < nrows; ++r) s += part[(size_t)r * C + c];
  out[c] += s;
}
