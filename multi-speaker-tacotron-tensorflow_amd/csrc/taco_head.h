// taco_head.h -- a wide dense layer over many rows as a ROW SWEEP: the linear head (tacotron.py:226-235; SURVEY 8a row a17:
// [B*T, 512] x [512, num_freq = 1025] at every output frame), split-bf16 arithmetic of k_gemm_bf3 (three products, same packs).
//
// k_gemm_bf3 tiles the output 64 x 256: five column blocks at N = 1025, the fifth for ONE column; every workgroup stages its 64 input
// rows again (eight 64-channel rounds, two barriers each, nothing in flight while the tile is converted) and 1280 workgroups take
// five turns on the chip: 91 us at C2, 27 % of the bf16 pipe.  Here a workgroup owns 64 rows for ALL columns:
//   * the rows are staged ONCE -- fp32 -> (hi, lo) bf16 planes [64][K + 8] in LDS, one barrier in the whole kernel;
//   * wave w sweeps the column-tile pairs w, w + 8, ... ONE 32-column tile at a time (round 5; 64 x 32 outputs, K / 16 steps of 6
//     MFMAs; round 4: the pair at once); its weight fragments come straight from the layer's pack in L2 through a ring of HD_PF
//     register sets that runs on across the tiles, so the stream never restarts; the stores of a tile ride inside the product loop
//     of the next one -- a quarter of a wave's stores is left after its last products;
//   * the workgroups of an XCD start their sweep at different passes (HD_ROT): they do not all ask the L2 for the same tiles at once;
//   * the columns past the last full tile (ONE at N = 1025) are dot products on the vector ALU from the same planes (hi + lo as fp32,
//     fp32 weights): a 33rd MFMA tile would hand one wave a third pass while seven idle.  They come FIRST, right behind the staging
//     barrier: behind the sweep their weight loads queued up behind the last tile's stores.
#pragma once
#include "taco_kernels.h"

#ifndef HD_ROT
#define HD_ROT 1
#endif
#define HD_BM 64
#define HD_KMAX 512      // widest input (the planes of 64 rows x 512 inputs take 133 KB of LDS)
#ifndef HD_PF
#define HD_PF 4          // register sets of weight fragments per wave: the fragments of step g + HD_PF - 1 are requested before step g is multiplied (a step is 6 MFMAs = ~200 clocks of the wave's own matrix work: three steps ahead cover an L2 round trip under load)
#endif
struct HeadArgs {
  const float* x; int ldx;                                     // input rows [M, ldx], K columns used (K % 64 == 0, K <= HD_KMAX)
  const unsigned short* bh; const unsigned short* bl;          // split-bf16 pack of the layer (pack_bf3)
  int NT, K16;                                                 // 32-column tiles in the pack, k16 steps (K / 16)
  const float* bias;                                           // [N] or null
  const float* wtail; int ntail;                               // the last N % 32 columns as fp32 rows [ntail][K] (ntail <= 4), or null / 0
  const float* rowvec; int ldrv, T;                            // optional per-batch-row vector [M / T, ldrv] added to every row of its batch row
  float* out; int ldo;                                         // [M, ldo]
  int M, K, N;
};

#ifdef TACO_TRACE
__device__ long long taco_trace_head[16];       // 0 entry, 1 rows staged, then per pass (products done, stores issued), last: tail columns done
#define HTRC(i) do { if (trc && (i) < 16) taco_trace_head[i] = clock64(); } while (0)
#else
#define HTRC(i) do {} while (0)
#endif
template <int KK>          // the layer's input width K (256 or 512): the product loop is unrolled over its K / 16 steps
__global__ __launch_bounds__(512) void k_head_sweep(const HeadArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) unsigned short hd_smem[];
  const float* gx = a_in.x; const unsigned short* gbh = a_in.bh; const unsigned short* gbl = a_in.bl; const float* gbias = a_in.bias;
  const float* gwt = a_in.wtail; const float* grv = a_in.rowvec; float* gout = a_in.out;
  PIN(gx); PIN(gbh); PIN(gbl); PIN(gbias); PIN(gwt); PIN(grv); PIN(gout);
  constexpr int K = KK, K16 = KK / 16;
  const int NT = a_in.NT, N = a_in.N, M = a_in.M, ldo = a_in.ldo;
  constexpr int LDSW = K + 8;                                      // bf16 per plane row: (K + 8) * 2 bytes = an odd multiple of 16 (K % 16 == 0)
  unsigned short* xhi = hd_smem;
  unsigned short* xlo = hd_smem + HD_BM * LDSW;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * HD_BM;
#ifdef TACO_TRACE
  const bool trc = (blockIdx.x == gridDim.x / 2) && threadIdx.x == 0;
  int trci = 2;
  HTRC(0);
#endif

  // ---- the workgroup's rows: fp32 -> planes, once ----
  {
    constexpr int nq = K / 4;                                      // float4 per row
    constexpr int NS = HD_BM * (K / 4) / 512;                  // float4 per thread
    float4 f[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {                             // all requests first
      const int i = tid + 512 * u, r = i / nq, c = 4 * (i - r * nq);
      f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < HD_BM * nq && m0 + r < M) f[u] = *reinterpret_cast<const float4*>(gx + (size_t)(m0 + r) * a_in.ldx + c);
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int i = tid + 512 * u, r = i / nq, c = 4 * (i - r * nq);
      if (i >= HD_BM * nq) continue;
      uint2 h4, l4;
      taco_split_bf16x4(f[u], h4, l4);
      *reinterpret_cast<uint2*>(xhi + r * LDSW + c) = h4;
      *reinterpret_cast<uint2*>(xlo + r * LDSW + c) = l4;
    }
  }
  // the wave's share of the sweep: column-tile pairs wave, wave + 8, ... of the full tiles, ONE 32-column tile at a time
  const int NTF = N / 32, npair = (NTF + 1) / 2;
  const int mypass = wave < npair ? (npair - 1 - wave) / 8 + 1 : 0;
  const int ntile = 2 * mypass, nsteps = ntile * K16;
  // HD_ROT: the workgroups of an XCD start their sweep at different passes (the L2 is asked for different column tiles at a time:
  // 75 -> 64 us at C2, profiles/r05_*)
  const int prot = HD_ROT ? (int)(blockIdx.x >> 3) % max(mypass, 1) : 0;
  auto tile_of = [&](int q) { int pp = (q >> 1) + prot; pp = pp >= mypass ? pp - mypass : pp; return 2 * (wave + 8 * pp) + (q & 1); };
  auto bofs = [&](int s) {                                     // flat step (tile, k16) -> element offset of the lane's fragment
    const int sc = min(s, nsteps - 1), q = sc / K16, g = sc - q * K16, t = min(tile_of(q), NT - 1);      // (the second tile of the last pair may not exist: its products are never stored)
    return ((((size_t)g * NT + t) * 2 + lh) * 32 + l31) * 8;
  };
  uint4 rh[HD_PF], rl[HD_PF];
  auto loadb = [&](int s, uint4& h, uint4& l) {
    const size_t o = bofs(s);
    h = *reinterpret_cast<const uint4*>(gbh + o); l = *reinterpret_cast<const uint4*>(gbl + o);
  };
  if (mypass > 0) {
#pragma unroll
    for (int i = 0; i < HD_PF - 1; ++i) loadb(i, rh[i], rl[i]);     // in flight across the staging barrier
  }
  __syncthreads();
  HTRC(1);
  // ---- the columns behind the last full tile: eight lanes per row, K / 8 inputs per lane, fp32 ----
  if (a_in.ntail > 0) {
    const int r = wave * 8 + (lane >> 3), kn = K / 8, k0 = (lane & 7) * kn, row = m0 + r;
    for (int t = 0; t < a_in.ntail; ++t) {
      const float* wt = gwt + (size_t)t * K + k0;
      float s = 0.f;
      for (int kk = 0; kk < kn; kk += 8) {
        const uint4 h = *reinterpret_cast<const uint4*>(xhi + r * LDSW + k0 + kk), l = *reinterpret_cast<const uint4*>(xlo + r * LDSW + k0 + kk);
        const float4 w0 = *reinterpret_cast<const float4*>(wt + kk), w1 = *reinterpret_cast<const float4*>(wt + kk + 4);
        s = fmaf(__uint_as_float(h.x << 16) + __uint_as_float(l.x << 16), w0.x, s);
        s = fmaf(__uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u), w0.y, s);
        s = fmaf(__uint_as_float(h.y << 16) + __uint_as_float(l.y << 16), w0.z, s);
        s = fmaf(__uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u), w0.w, s);
        s = fmaf(__uint_as_float(h.z << 16) + __uint_as_float(l.z << 16), w1.x, s);
        s = fmaf(__uint_as_float(h.z & 0xffff0000u) + __uint_as_float(l.z & 0xffff0000u), w1.y, s);
        s = fmaf(__uint_as_float(h.w << 16) + __uint_as_float(l.w << 16), w1.z, s);
        s = fmaf(__uint_as_float(h.w & 0xffff0000u) + __uint_as_float(l.w & 0xffff0000u), w1.w, s);
      }
      s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
      s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
      s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x141, 0xF, 0xF, false));     // row_half_mirror: the other quad of the eight
      const int col = N - a_in.ntail + t;
      if ((lane & 7) == 0 && row < M) {
        float v = s + (gbias ? gbias[col] : 0.f);
        if (grv) v += grv[(size_t)(row / a_in.T) * a_in.ldrv + col];
        gout[(size_t)row * ldo + col] = v;
      }
    }
  }

  f32x16 acc[2];
  const unsigned short* abh = xhi;                             // the lane's fragment rows in the two planes (set per tile from opaque lane coordinates:
  const unsigned short* abl = xlo;                             // everything else of an A-fragment address is an immediate offset)
  auto mma = [&](int g, const uint4& h, const uint4& l) {
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      ah[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(abh + tm * 32 * LDSW + 16 * g));
      al[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(abl + tm * 32 * LDSW + 16 * g));
    }
    const bf16x8 bh0 = __builtin_bit_cast(bf16x8, h), bl0 = __builtin_bit_cast(bf16x8, l);
    // small terms first; the two independent accumulators alternate
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh0, acc[tm], 0, 0, 0);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl0, acc[tm], 0, 0, 0);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh0, acc[tm], 0, 0, 0);
  };
  // The stores of a tile are issued INSIDE the product loop of the next one (32 per lane: one per k16 step at K = 512), from a copy of its
  // accumulators: left behind the loop, every wave of the chip stores at once and the matrix pipe idles (round 4, with 64 x 64 outputs per
  // pass: 21 K clocks of stores behind 35 K clocks of products, twice; with the pair as the unit the LAST pair's 35 K clocks stayed exposed
  // -- a quarter of the kernel -- so the unit is now one 32-column tile: a wave re-reads its A fragments for the second tile of a pair
  // (LDS has the bandwidth) and only a quarter of its stores are left over at the end).  Workgroups with rows past M, and the
  // per-batch-row vector of model type 'simple', take the plain order (guards / an index division per element).
  const bool inter = (m0 + HD_BM <= M) && !grv;
  f32x16 accP[2];
  float pbia = 0.f;
  bool pv = false;
  float* pbase = gout;                                        // previous tile: element (row m0 + 4 * (lane >> 5), column 32 t + (lane & 31))
  constexpr int SPS = 32 / K16;                                // stores of the previous tile per k16 step
  static_assert(K16 <= 32 && 32 % K16 == 0, "the previous tile's 32 stores spread evenly over the k16 steps");
  auto store_prev = [&](int idx) {                             // idx (compile-time after unrolling): tm | register
    const int tm = (idx >> 4) & 1, r = idx & 15;
    if (pv) pbase[(size_t)(tm * 32 + (r & 3) + 8 * (r >> 2)) * ldo] = accP[tm][r] + pbia;
  };
  for (int q = 0; q < ntile; ++q) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
      for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
    const int s0 = q * K16;
    {
      int l31a = l31, lha = lh;
      asm volatile("" : "+v"(l31a), "+v"(lha));
      abh = xhi + l31a * LDSW + 8 * lha; abl = xlo + l31a * LDSW + 8 * lha;
    }
    if (q > 0 && inter) {
#pragma unroll
      for (int g = 0; g < K16; g += HD_PF) {                   // K16 % HD_PF == 0: the ring stays aligned across the tiles
#pragma unroll
        for (int i = 0; i < HD_PF; ++i) {
          loadb(s0 + g + i + HD_PF - 1, rh[(i + HD_PF - 1) % HD_PF], rl[(i + HD_PF - 1) % HD_PF]);
#pragma unroll
          for (int u = 0; u < SPS; ++u) store_prev((g + i) * SPS + u);
          __builtin_amdgcn_sched_barrier(0);
          mma(g + i, rh[i], rl[i]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      for (int g = 0; g < K16; g += HD_PF) {
#pragma unroll
        for (int i = 0; i < HD_PF; ++i) {
          loadb(s0 + g + i + HD_PF - 1, rh[(i + HD_PF - 1) % HD_PF], rl[(i + HD_PF - 1) % HD_PF]);
          __builtin_amdgcn_sched_barrier(0);
          mma(g + i, rh[i], rl[i]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#ifdef TACO_TRACE
    HTRC(trci); ++trci;
#endif
    const int t = tile_of(q);
    int l31e = l31, lhe = lh;                                  // (opaque per tile: the output addresses are formed here, not kept across the sweep)
    asm volatile("" : "+v"(l31e), "+v"(lhe));
    if (inter && q + 1 < ntile) {                              // hand the tile to the next tile's loop
      const int col = t * 32 + l31e;
      pv = t < NTF;
      pbia = (gbias && pv) ? gbias[col] : 0.f;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) accP[tm] = acc[tm];
      pbase = gout + (size_t)(m0 + 4 * lhe) * ldo + t * 32 + l31e;
      continue;
    }
    // + bias (+ the batch row's vector) -> the output rows
    if (t < NTF) {
      const int col = t * 32 + l31e;
      const float bia = gbias ? gbias[col] : 0.f;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhe;
          if (row < M) {
            float v = acc[tm][r] + bia;
            if (grv) v += grv[(size_t)(row / a_in.T) * a_in.ldrv + col];
            gout[(size_t)row * ldo + col] = v;
          }
        }
    }
  }
#ifdef TACO_TRACE
  HTRC(trci);
#endif
}
