// taco_chain.h -- the point-wise tail of a CBHG (modules.py:72-77 + the hoisted input projection of the BiGRU, :82-96) as ONE
// launch: [dense ->] highway x depth -> BiGRU input projection, for a tile of 64 frames at a time.
//
// As separate launches (k_gemm_bf3) every link writes its [B*T, W] activation to HBM and the next one stages it again: 6 launches
// and 5 round trips of 16 MB at the post-net of C2 for layers of only 4 GFLOP each (30 us per highway layer, 150 TF-equivalent).
// Here a workgroup of 8 waves keeps the tile's activations on the CU from the first layer to the last:
//   * the tile [64, W] lives in LDS as two bf16 planes (hi = bf16(x), lo = bf16(x - hi): the split-bf16 operand of k_gemm_bf3,
//     same arithmetic, same weight packs) -- the A operand of every layer;
//   * wave (wm, wn) owns rows 64/WM * wm .. and the 32 columns 32 wn .. of every W-wide layer, streams its weight fragments from
//     the layer's pack (L2) one k16 step ahead, and keeps its own 32 x 32 (x TM) slice of the layer's fp32 output in registers:
//     the highway carry  y = H*T + x*(1-T)  needs x exactly where the lane produced it one layer earlier;
//   * between layers: barrier, the lanes write their outputs into the planes, barrier;
//   * the last link (N = 6H columns) loops over column groups of W and stores straight to the projection buffer, the backward
//     direction's columns time-reversed per row (tf.reverse_sequence, A.7) exactly as k_gemm_bf3's epilogue does.
//
// Round 5: the same kernel also runs the ENCODER PRENET (taco_lib.hip: run_prenet_chain) -- the embedding rows gathered straight into the planes
// (ChainArgs.gather), a 256-wide ReLU layer, and the second layer as the chain's last link with a ReLU on its way out (ChainLayer.act) --, and the
// workgroups of that launch that own no tile clear the words the forward's persistent kernels poll (ChainArgs.ntiles, zp, znw).  The workgroups of
// an XCD walk every layer's K loop from different starting steps (CH_ROT).
#pragma once
#include "taco_kernels.h"

#define CH_MAXL 8
// Register sets of weight fragments per wave (ch_mma_loop), W = 256 / W = 128.  What bounds the product loops, measured at C2 in round 4
// (tools/trace_chain.py on ablated builds): a k16 step of a highway layer takes ~1040 ticks for ~650 of matrix work.  Rings 2, 3, 4, 5 deep,
// the activation fragments requested a step ahead of their MFMAs, four or eight independent accumulator chains per cluster: no change.
// Every step re-reading the FIRST step's weight fragments (L1 hits, same instruction stream): 16.2 K -> 10.5 K ticks per layer.  So the loops
// are bound by the stream of weight fragments out of L2: 32 CUs of an XCD x 32 KB per step = 1 MB per step through an L2 that returns ~1 KB
// per clock -- bandwidth, not latency.  Rows per workgroup set that ratio (every workgroup streams all 3.6 MB of the stage's weights): 128-row
// tiles would halve the stream but leave half the CUs idle at M = 16384 and double the matrix time per CU, which is the larger term.
#ifndef CH_PF256
#define CH_PF256 2
#endif
#ifndef CH_PF128
#define CH_PF128 2
#endif
#ifndef CH_ROT
#define CH_ROT 1
#endif
#define CH_BM 64
enum { CH_DENSE = 0, CH_HIGHWAY = 1, CH_XPROJ = 2 };
struct ChainLayer {
  const unsigned short* bh; const unsigned short* bl; const unsigned short* bh2; const unsigned short* bl2;   // split-bf16 packs (pack_bf3)
  const float* bias; const float* bias2;
  int type, K16, NT, N, act;       // k16 steps of the layer's K (padded input width / 16), 32-column tiles in the pack, columns
};
// Fused entry (the CBHG's second projection in front of the chain, modules.py:55-69): instead of reading its input rows, the workgroup
// forms them -- proj_1's output from the partial sums of k_cbhg_front (sum over the parts in a fixed order, + bias -> ReLU -> BatchNorm:
// what k_front_combine did as a launch of its own), then the width-3 projection conv proj_2 (+ bias -> BatchNorm affine, no activation)
// on the matrix cores, + the residual input (+ the deepvoice per-row vector).  Saves two launches and the round trip of two activations.
struct ChainEntry {
  const float* part; int P; size_t MN; int N1;           // partial sums [P][M][N1]
  const float* b1; const float* s1; const float* h1; int relu1;       // proj_1: bias, BatchNorm scale / shift (nullable), ReLU?
  const unsigned short* bh; const unsigned short* bl; int NT2, K16tap, N2;   // proj_2 pack (pack_bf3), k16 groups per tap, columns
  const float* b2; const float* s2; const float* h2;     // proj_2: bias, BatchNorm scale / shift (nullable)
  const float* res; int ldres;                           // residual rows [M, ldres] (the CBHG's input)
  const float* rowvec; int ldrv;                         // optional per-batch-row vector [M / T, ldrv]
};
struct ChainArgs {
  ChainEntry e;                    // e.part != null: fused entry (x unused)
  const float* x; int ldx, Cin;    // input rows [M, ldx], Cin columns used
  const int* gather;               // optional: row m of the input is x[gather[m]] (the embedding lookup in front of the encoder prenet, tacotron.py:38-39)
  float* out; int ldo;             // projection output [M, ldo]
  float* y; int ldy;               // optional: the last highway output [M, W] (null: not stored)
  const int* rev_len; int rev_col0;
  int M, T, nlayers;
  int krot;                        // 1: workgroups start their K loops at different steps (CH_ROT, the default); 0: every tile walks K from step 0
  // riders: with ntiles > 0 the workgroups ntiles .. gridDim.x - 1 own no tile; they clear up to four regions of 32-bit words (the words the
  // persistent kernels of the same forward poll: a launch of its own otherwise -- the encoder prenet's chain leaves three quarters of the CUs free)
  int ntiles; unsigned* zp[4]; unsigned long long znw[4];
  ChainLayer L[CH_MAXL];
};

// one layer's K loop for the wave's TM row tiles and one 32-column tile (DUAL: two products that share the A fragments -- the H and
// T matrices of a highway layer at the same column tile, or two column tiles nt, nt2 of one matrix): weight fragments (hi, lo) one
// k16 step ahead in two register sets that swap roles (K16 is even: pack_bf3 pads K to a multiple of 32)
template <int TM, int LDSW, bool DUAL, int CH_PF>
__device__ __forceinline__ void ch_mma_loop(int K16, int NT, const unsigned short* bhA, const unsigned short* blA, int nt,
                                            const unsigned short* bhB, const unsigned short* blB, int nt2,
                                            const unsigned short* xhi, const unsigned short* xlo, int row0,
                                            int l31, int lh, f32x16 (&acc)[TM], f32x16 (&acc2)[TM], int rot = 0) {
  // rot: the workgroup's first k16 step (CH_ROT): the CUs of an XCD walk the layer's K in the same order but from different starts, so
  // that at any moment they ask the L2 for DIFFERENT 8 KB slices of the pack instead of all 32 for the same one
  auto kstep = [&](int g) { const int gg = min(g, K16 - 1) + rot; return gg >= K16 ? gg - K16 : gg; };
  auto boff = [&](int g, int t) { return ((((size_t)kstep(g) * NT + t) * 2 + lh) * 32 + l31) * 8; };
  uint4 ph, pl, ph2, pl2, qh, ql, qh2, ql2;
  ph2 = pl2 = qh2 = ql2 = make_uint4(0, 0, 0, 0);
  auto loadb = [&](int g, uint4& h, uint4& l, uint4& h2, uint4& l2) {
    const size_t o = boff(g, nt);
    h = *reinterpret_cast<const uint4*>(bhA + o); l = *reinterpret_cast<const uint4*>(blA + o);
    if constexpr (DUAL) { const size_t o2 = boff(g, nt2); h2 = *reinterpret_cast<const uint4*>(bhB + o2); l2 = *reinterpret_cast<const uint4*>(blB + o2); }
  };
  auto mma = [&](int g, const uint4& h, const uint4& l, const uint4& h2, const uint4& l2) {
    const bf16x8 bh = __builtin_bit_cast(bf16x8, h), bl = __builtin_bit_cast(bf16x8, l);
    bf16x8 ah[TM], al[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int off = (row0 + tm * 32 + l31) * LDSW + 16 * kstep(g) + 8 * lh;
      ah[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xhi + off));
      al[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xlo + off));
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh, acc[tm], 0, 0, 0);      // small terms first
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl, acc[tm], 0, 0, 0);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh, acc[tm], 0, 0, 0);
    if constexpr (DUAL) {
      const bf16x8 bh2 = __builtin_bit_cast(bf16x8, h2), bl2 = __builtin_bit_cast(bf16x8, l2);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) acc2[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh2, acc2[tm], 0, 0, 0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) acc2[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl2, acc2[tm], 0, 0, 0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) acc2[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh2, acc2[tm], 0, 0, 0);
    }
  };
  if constexpr (CH_PF == 2) {
  loadb(0, ph, pl, ph2, pl2);
  for (int g = 0; g < K16; g += 2) {
    loadb(g + 1, qh, ql, qh2, ql2);
    __builtin_amdgcn_sched_barrier(0);
    mma(g, ph, pl, ph2, pl2);
    __builtin_amdgcn_sched_barrier(0);
    loadb(g + 2, ph, pl, ph2, pl2);
    __builtin_amdgcn_sched_barrier(0);
    mma(g + 1, qh, ql, qh2, ql2);
    __builtin_amdgcn_sched_barrier(0);
  }
  } else {
  // a ring of CH_PF register sets: the fragments of step g + CH_PF - 1 are requested before step g is multiplied.  One step ahead
  // (round 3) left every k16 step waiting for L2: a workgroup had 32 KB in flight, and a CU's share of the L2 stream at that depth
  // is ~27 GB/s -- a quarter of what it reaches with 8 loads per lane outstanding (MI355X_MICROARCH.md, hand-off payload row)
  uint4 rh[CH_PF], rl[CH_PF], rh2[CH_PF], rl2[CH_PF];
#pragma unroll
  for (int i = 0; i < CH_PF; ++i) rh2[i] = rl2[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < CH_PF - 1; ++i) loadb(i, rh[i], rl[i], rh2[i], rl2[i]);
  for (int g = 0; g < K16; g += CH_PF) {
#pragma unroll
    for (int i = 0; i < CH_PF; ++i) {
      loadb(g + i + CH_PF - 1, rh[(i + CH_PF - 1) % CH_PF], rl[(i + CH_PF - 1) % CH_PF], rh2[(i + CH_PF - 1) % CH_PF], rl2[(i + CH_PF - 1) % CH_PF]);
      __builtin_amdgcn_sched_barrier(0);
      if (g + i < K16) mma(g + i, rh[i], rl[i], rh2[i], rl2[i]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  }
}

#ifdef TACO_TRACE
__device__ long long taco_trace_chain[64];      // slots: 0 entry, 1 partial sums in planes, 2 proj_2 products done, 3 entry epilogue done, 4 first barrier; then per layer:
#define CTRC(i) do { if (trc && (i) < 64) taco_trace_chain[i] = clock64(); } while (0)      // products done, epilogue done, planes free, planes written
#else
#define CTRC(i) do {} while (0)
#endif
// (Round 5: the ring depth was tried again where only 64 workgroups run -- the encoder prenet's chain and the encoder's own, 4096 rows: six sets
// instead of two made the prenet chain 17.6 -> 22.9 us and the encoder chain 61 -> 71 us (four sets: 63).  A deeper ring issues its last sets'
// requests again past the end of every short K loop; the loops are paced by the bytes a CU can pull from L2 (~26-31 B per clock and CU
// measured, all CUs streaming), not by the round trip, so more requests in flight only add bytes.  profiles/r05_*.)
template <int W>
__global__ __launch_bounds__(512) void k_pointwise_chain(const ChainArgs a_in) {
  constexpr int WN = W / 32, WM = 8 / WN, TM = CH_BM / 32 / WM;      // W = 256: 1 x 8 waves of 64 x 32; W = 128: 2 x 4 waves of 32 x 32
  constexpr int PF = W == 256 ? CH_PF256 : CH_PF128;
  constexpr int LDSW = W + 8;                                          // bf16 elements per plane row: (W + 8) * 2 bytes = odd multiple of 16
  extern __shared__ __attribute__((aligned(16))) unsigned short ch_smem[];
  unsigned short* xhi = ch_smem;
  unsigned short* xlo = ch_smem + CH_BM * LDSW;
  struct { const float* x; float* out; float* y; const int* rev_len; int ldx, Cin, ldo, ldy, rev_col0, M, T, nlayers; } a =
      {a_in.x, a_in.out, a_in.y, a_in.rev_len, a_in.ldx, a_in.Cin, a_in.ldo, a_in.ldy, a_in.rev_col0, a_in.M, a_in.T, a_in.nlayers};
  PIN(a.x); PIN(a.out); PIN(a.y); PIN(a.rev_len); PIN(a.ldx); PIN(a.Cin); PIN(a.ldo); PIN(a.ldy); PIN(a.rev_col0); PIN(a.M); PIN(a.T);
  PIN(a.nlayers);
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  if (a_in.ntiles > 0 && (int)blockIdx.x >= a_in.ntiles) {
    const size_t i0 = (size_t)((int)blockIdx.x - a_in.ntiles) * 512 + tid, stride = (size_t)((int)gridDim.x - a_in.ntiles) * 512;
    for (int rg = 0; rg < 4; ++rg) {
      unsigned* p = a_in.zp[rg];
      const size_t nw = a_in.znw[rg];
      if (!p || !nw) continue;
      const size_t n16 = (reinterpret_cast<uintptr_t>(p) & 15) ? 0 : nw / 4;
      uint4* q = reinterpret_cast<uint4*>(p);
      for (size_t i = i0; i < n16; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
      for (size_t i = n16 * 4 + i0; i < nw; i += stride) p[i] = 0u;
    }
    return;
  }
  const int m0 = blockIdx.x * CH_BM;
  // first k16 step of this workgroup's K loops, in 32nds of a layer's K (CH_ROT; ChainArgs::krot = 0 switches it off at run time: every row's fp32
  // accumulation order is then the same whichever tile the row lands in -- bit-invariance under batch permutation / sharding at any size)
  const int krot = (CH_ROT && a_in.krot) ? ((int)(blockIdx.x >> 3) & 31) : 0;
#ifdef TACO_TRACE
  const bool trc = (blockIdx.x == (a_in.ntiles > 0 ? a_in.ntiles : gridDim.x) / 2) && threadIdx.x == 0;
  int trci = 5;
  CTRC(0);
#endif

  const int l31_outer = l31, lh_outer = lh;
  float xreg[TM][16];                          // the lane's slice of the current activation in fp32 (highway carry)
  bool have_x = false;
  if (a_in.e.part) {
    // ---- fused entry: proj_1's epilogue over the partial sums -> planes [CH_BM + 2][N1]; proj_2 (width 3) -> the chain's input ----
    const ChainEntry E = a_in.e;
    const float* gpart = E.part; const float* gb1 = E.b1; const float* gs1 = E.s1; const float* gh1 = E.h1;
    const unsigned short* gbh = E.bh; const unsigned short* gbl = E.bl;
    const float* gres = E.res; const float* grv = E.rowvec; const float* gb2 = E.b2; const float* gs2 = E.s2; const float* gh2 = E.h2;
    PIN(gpart); PIN(gb1); PIN(gs1); PIN(gh1); PIN(gbh); PIN(gbl); PIN(gres); PIN(grv); PIN(gb2); PIN(gs2); PIN(gh2);
    const int LDE = E.N1 + 8;                                              // bf16 per plane row (N1 = 128 / 256: odd multiple of 16 bytes)
    unsigned short* ehi = ch_smem + 2 * CH_BM * LDSW;                      // behind the chain's own planes
    unsigned short* elo = ehi + (CH_BM + 2) * LDE;
    const int nq = E.N1 / 4;
    constexpr int NSTG = ((CH_BM + 2) * (256 / 4) + 511) / 512;          // float4 per thread at the widest N1
    {
      float4 f[NSTG];
      const float* src[NSTG];
#pragma unroll
      for (int u = 0; u < NSTG; ++u) {
        const int i = tid + 512 * u, r = i / nq, c = 4 * (i - r * nq), row = m0 - 1 + r;
        f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        src[u] = (i < (CH_BM + 2) * nq && row >= 0 && row < a.M) ? gpart + (size_t)row * E.N1 + c : nullptr;
      }
      // parts in a fixed order (part q before part q + 1: bit-reproducible); the loads of TWO parts are in flight together -- a part's 9 loads per
      // thread alone left a full memory round trip exposed per part (2 at the post-net, 8 at the encoder)
      int q = 0;
      for (; q + 1 < E.P; q += 2) {
        float4 g[NSTG], g2[NSTG];
#pragma unroll
        for (int u = 0; u < NSTG; ++u) {
          g[u] = src[u] ? *reinterpret_cast<const float4*>(src[u] + (size_t)q * E.MN) : make_float4(0.f, 0.f, 0.f, 0.f);
          g2[u] = src[u] ? *reinterpret_cast<const float4*>(src[u] + (size_t)(q + 1) * E.MN) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < NSTG; ++u) { f[u].x += g[u].x; f[u].y += g[u].y; f[u].z += g[u].z; f[u].w += g[u].w; }
#pragma unroll
        for (int u = 0; u < NSTG; ++u) { f[u].x += g2[u].x; f[u].y += g2[u].y; f[u].z += g2[u].z; f[u].w += g2[u].w; }
      }
      if (q < E.P) {
        float4 g[NSTG];
#pragma unroll
        for (int u = 0; u < NSTG; ++u) g[u] = src[u] ? *reinterpret_cast<const float4*>(src[u] + (size_t)q * E.MN) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < NSTG; ++u) { f[u].x += g[u].x; f[u].y += g[u].y; f[u].z += g[u].z; f[u].w += g[u].w; }
      }
#pragma unroll
      for (int u = 0; u < NSTG; ++u) {
        const int i = tid + 512 * u, r = i / nq, c = 4 * (i - r * nq);
        if (i >= (CH_BM + 2) * nq) continue;
        float4 v = f[u];
        if (src[u]) {
          const float4 bi = gb1 ? *reinterpret_cast<const float4*>(gb1 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 sc = gs1 ? *reinterpret_cast<const float4*>(gs1 + c) : make_float4(1.f, 1.f, 1.f, 1.f);
          const float4 sh = gh1 ? *reinterpret_cast<const float4*>(gh1 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          v.x += bi.x; v.y += bi.y; v.z += bi.z; v.w += bi.w;
          if (E.relu1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        }
        uint2 h4, l4;
        taco_split_bf16x4(v, h4, l4);
        *reinterpret_cast<uint2*>(ehi + r * LDE + c) = h4;
        *reinterpret_cast<uint2*>(elo + r * LDE + c) = l4;
      }
    }
    CTRC(1);
    __syncthreads();
    // proj_2: wave (wm, wn) owns its TM row tiles x the 32 columns 32 wn .. (waves past the last column tile idle); SAME padding and the
    // batch-row boundary by masking the A fragment per (row, tap) as k_gemm_bf3 does
    const int K0 = a_in.L[0].K16 * 16;                                     // columns of the chain's input planes (>= N2, zero padded)
    const int col = wn * 32 + l31;
    const bool cin = col < E.N2;
    // (requested BEFORE the MFMA loop, whose duration hides them; all residual values of the lane are requested before the first is used: left in one loop with the LDS stores, every element
    // waited for its own load -- 32 memory round trips)
    float rs[TM][16];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, row = min(m0 + rl, a.M - 1);
        float v = 0.f;
        if (cin) {
          v = gres[(size_t)row * E.ldres + col];
          if (grv) v += grv[(size_t)(row / a.T) * E.ldrv + col];
        }
        rs[tm][r] = v;
      }
    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
      for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
    // W = 256 (one row of eight waves): N2 = 80 columns are three column tiles -- alone, three waves would run the whole contraction on
    // three SIMDs, one wave each, while five wait (31 K of the workgroup's 220 K clocks at C2).  The contraction is cut in two K halves
    // instead: waves NT2 .. 2 NT2 - 1 take the second half of the k16 steps of column tile wn - NT2 and hand their partial tile over
    // through LDS (the chain's own planes are not written yet)
    const int KSPL = (WM == 1 && 2 * E.NT2 <= WN) ? 2 : 1;
    const bool mact = wn < E.NT2 * KSPL;
    const int ect = mact ? wn % E.NT2 : 0, kp = mact ? wn / E.NT2 : 0;
    if (mact) {
      int tloc[TM];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) tloc[tm] = (m0 + (wm * TM + tm) * 32 + l31) % a.T;
      const int nk = (3 * E.K16tap) / KSPL, kbeg = kp * nk, kend = kbeg + nk;      // k16 steps (tap, group) of this wave
      auto boff = [&](int k) { return ((((size_t)k * E.NT2 + ect) * 2 + lh) * 32 + l31) * 8; };
      // few waves per SIMD here, so the weight stream is requested a whole block of NB k16 steps ahead (two register
      // sets that swap roles; nk = 3 N1 / 16 [/ 2] is a multiple of 2 NB): one step ahead left every step waiting ~800 clocks for L2
      constexpr int NB = 6;                            // nk = 48, 24 (K halves) or 24: a multiple of 2 NB (the host admits N1 = 256 with W = 256, 128 with 128)
      uint4 pH[NB], pL[NB], qH[NB], qL[NB];
      auto loadblk = [&](int k0, uint4 (&H)[NB], uint4 (&Lo)[NB]) {
#pragma unroll
        for (int i = 0; i < NB; ++i) { const size_t o = boff(min(k0 + i, kend - 1)); H[i] = *reinterpret_cast<const uint4*>(gbh + o); Lo[i] = *reinterpret_cast<const uint4*>(gbl + o); }
      };
      auto mmablk = [&](int k0, const uint4 (&H)[NB], const uint4 (&Lo)[NB]) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int k = k0 + i, j = k / E.K16tap, g = k - j * E.K16tap;
          const bf16x8 bh8 = __builtin_bit_cast(bf16x8, H[i]), bl8 = __builtin_bit_cast(bf16x8, Lo[i]);
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            const int off = ((wm * TM + tm) * 32 + l31 + j) * LDE + 16 * g + 8 * lh;
            uint4 xh = *reinterpret_cast<const uint4*>(ehi + off), xl = *reinterpret_cast<const uint4*>(elo + off);
            const int tt = tloc[tm] + j - 1;
            const unsigned keep = ((tt >= 0) && (tt < a.T)) ? 0xffffffffu : 0u;
            xh.x &= keep; xh.y &= keep; xh.z &= keep; xh.w &= keep; xl.x &= keep; xl.y &= keep; xl.z &= keep; xl.w &= keep;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, xh), al = __builtin_bit_cast(bf16x8, xl);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh8, acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl8, acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh8, acc[tm], 0, 0, 0);
          }
        }
      };
      loadblk(kbeg, pH, pL);
      for (int k0 = kbeg; k0 < kend; k0 += 2 * NB) {
        loadblk(k0 + NB, qH, qL);
        __builtin_amdgcn_sched_barrier(0);
        mmablk(k0, pH, pL);
        __builtin_amdgcn_sched_barrier(0);
        loadblk(k0 + 2 * NB, pH, pL);
        __builtin_amdgcn_sched_barrier(0);
        mmablk(k0 + NB, qH, qL);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (KSPL == 2) {                                   // the second K half's partial tile -> the first half's wave (fixed order: bit-reproducible)
      float* red = reinterpret_cast<float*>(ch_smem);
      if (mact && kp == 1) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((ect * TM + tm) * 16 + r) * 64 + lane] = acc[tm][r];
      }
      __syncthreads();
      if (mact && kp == 0) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tm][r] += red[((ect * TM + tm) * 16 + r) * 64 + lane];
      } else {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
      }
      __syncthreads();                                 // ... before the epilogue below writes the planes over the partial tiles
    }
    CTRC(2);
    // epilogue: + bias -> BatchNorm affine -> + residual (+ per-row vector); the chain's input planes (zero beyond N2) and the carry registers
    {
      const float bi = (cin && gb2) ? gb2[col] : 0.f, sc = (cin && gs2) ? gs2[col] : 1.f, sh = (cin && gh2) ? gh2[col] : 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(rs[tm][r]));
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const float v = cin ? (acc[tm][r] + bi) * sc + sh + rs[tm][r] : 0.f;
          xreg[tm][r] = v;
          if (col < K0) {
            const unsigned hp = taco_pk_bf16(v, 0.f) & 0xffffu;
            const unsigned lp = taco_pk_bf16(v - __uint_as_float(hp << 16), 0.f) & 0xffffu;
            xhi[rl * LDSW + col] = (unsigned short)hp;
            xlo[rl * LDSW + col] = (unsigned short)lp;
          }
        }
      have_x = true;
    }
    CTRC(3);
  } else
  // ---- stage the input rows: fp32 -> (hi, lo) planes, zero padded to the first layer's K ----
  {
    const int K0 = a_in.L[0].K16 * 16;
    for (int i = tid; i < CH_BM * (K0 / 4); i += 512) {
      const int r = i / (K0 / 4), c = 4 * (i % (K0 / 4));
      float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
      const int row = m0 + r;
      if (row < a.M && c < a.Cin) {
        const float* p = a.x + (size_t)(a_in.gather ? a_in.gather[row] : row) * a.ldx + c;
        if (c + 3 < a.Cin && (a.ldx & 3) == 0) f = *reinterpret_cast<const float4*>(p);
        else { f.x = p[0]; if (c + 1 < a.Cin) f.y = p[1]; if (c + 2 < a.Cin) f.z = p[2]; if (c + 3 < a.Cin) f.w = p[3]; }
      }
      uint2 h4, l4;
      taco_split_bf16x4(f, h4, l4);
      *reinterpret_cast<uint2*>(xhi + r * LDSW + c) = h4;
      *reinterpret_cast<uint2*>(xlo + r * LDSW + c) = l4;
    }
  }
  const bool full = m0 + CH_BM <= a.M;         // every row of the tile exists: the stores need no guards (the common case)
  __syncthreads();
  CTRC(4);

  for (int li = 0; li < a.nlayers; ++li) {
    const ChainLayer L = a_in.L[li];
    // opaque per-layer copies of the lane coordinates: otherwise every address below (32 carry loads, 32 stores, 64 LDS writes per
    // lane) is computed once before the loop and kept -- and spilled -- across all layers (same effect as in taco_decoder_xcd.h)
    int l31 = l31_outer, lh = lh_outer;
    asm volatile("" : "+v"(l31), "+v"(lh));
    // row of the lane's C element r of row tile tm (C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
    auto rloc = [&](int tm, int r) { return (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh; };
    const int col = wn * 32 + l31;
    if (L.type == CH_XPROJ) {
      // ---- last link: N = 6H columns in groups of W, stored straight to the projection buffer ----
      const int ngroups = (L.N + W - 1) / W;
      const int bb0 = m0 / a.T;                // a tile of 64 rows spans at most two batch rows when T >= 64: their lengths once, not once per stored element
      const int nbrow = (a.M - 1) / a.T;
      const int Lb0 = a.rev_len ? a.rev_len[min(bb0, nbrow)] : a.T, Lb1 = a.rev_len ? a.rev_len[min(bb0 + 1, nbrow)] : a.T;
      auto store_group = [&](int ng, const f32x16 (&acc)[TM]) {
        const int ocol = ng * W + col;
        if (ocol >= L.N) return;
        const float bia = L.bias ? L.bias[ocol] : 0.f;
        const bool rev = a.rev_col0 >= 0 && ocol >= a.rev_col0;
        const bool relu_out = L.act == ACT_RELU;                       // (a chain that ends in a ReLU layer: the encoder prenet; otherwise the value passes as it is, a NaN included)
        if (full && !rev) {
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            float* po = a.out + (size_t)(m0 + rloc(tm, 0)) * a.ldo + ocol;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float v = acc[tm][r] + bia; po[(size_t)((r & 3) + 8 * (r >> 2)) * a.ldo] = relu_out ? fmaxf(v, 0.f) : v; }
          }
        } else {                               // the backward direction's columns go to the time-reversed row of the batch row
          int lhs = lh;                        // (opaque per call: the 32 reversed row indices are formed here, not kept -- and spilled -- across the column groups)
          asm volatile("" : "+v"(lhs));
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhs;
              int orow = row;
              if (rev) {
                const int bb = a.T >= CH_BM ? bb0 + (row >= (bb0 + 1) * a.T ? 1 : 0) : row / a.T;
                const int tt = row - bb * a.T, Lb = a.T >= CH_BM ? (bb == bb0 ? Lb0 : Lb1) : (a.rev_len ? a.rev_len[min(bb, (a.M - 1) / a.T)] : a.T);
                orow = tt < Lb ? bb * a.T + (Lb - 1 - tt) : row;
              }
              if (row < a.M) { const float v = acc[tm][r] + bia; a.out[(size_t)orow * a.ldo + ocol] = relu_out ? fmaxf(v, 0.f) : v; }
            }
        }
      };
      // two column groups per pass share the A fragments; an odd group left over (6H = 3 W at W = 256: the backward direction's last
      // columns, whose time-reversed stores are the slow ones) goes FIRST, so that its stores drain behind the products of the pass after it
      const int npass = (ngroups + 1) / 2, odd = ngroups & 1;
      for (int ps = 0; ps < npass; ++ps) {
        const int ng = (odd && ps == 0) ? ngroups - 1 : 2 * (ps - odd);
        const int nt = min(ng * WN + wn, L.NT - 1), nt2 = min((ng + 1) * WN + wn, L.NT - 1);   // tiles past the pack are clamped, never stored
        f32x16 acc[TM], acc2[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          for (int r = 0; r < 16; ++r) { acc[tm][r] = 0.f; acc2[tm][r] = 0.f; }
        if (ng + 1 < ngroups) ch_mma_loop<TM, LDSW, true, PF>(L.K16, L.NT, L.bh, L.bl, nt, L.bh, L.bl, nt2, xhi, xlo, wm * TM * 32, l31, lh, acc, acc2, krot * L.K16 >> 5);
        else ch_mma_loop<TM, LDSW, false, PF>(L.K16, L.NT, L.bh, L.bl, nt, L.bh, L.bl, nt, xhi, xlo, wm * TM * 32, l31, lh, acc, acc2, krot * L.K16 >> 5);     // the odd group: one matrix
#ifdef TACO_TRACE
        CTRC(trci); ++trci;
#endif
        store_group(ng, acc);
        if (ng + 1 < ngroups) store_group(ng + 1, acc2);
#ifdef TACO_TRACE
        CTRC(trci); ++trci;
#endif
      }
      continue;
    }
    // ---- a W-wide hidden layer: dense or highway ----
    const bool dual = L.type == CH_HIGHWAY;
    if (dual && !have_x) {                     // a chain that starts with a highway layer: the carry comes from the input itself
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) xreg[tm][r] = a.x[(size_t)min(m0 + rloc(tm, r), a.M - 1) * a.ldx + col];
    }
    have_x = true;
    {
      f32x16 acc[TM], acc2[TM];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        for (int r = 0; r < 16; ++r) { acc[tm][r] = 0.f; acc2[tm][r] = 0.f; }
      const float bia = L.bias ? L.bias[col] : 0.f;
      if (dual) {
        const float bia2 = L.bias2 ? L.bias2[col] : 0.f;
        ch_mma_loop<TM, LDSW, true, PF>(L.K16, L.NT, L.bh, L.bl, wn, L.bh2, L.bl2, wn, xhi, xlo, wm * TM * 32, l31, lh, acc, acc2, krot * L.K16 >> 5);
#ifdef TACO_TRACE
        CTRC(trci); ++trci;
#endif
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) {       // modules.py:105-120: H = relu(.), T = sigmoid(.), y = H*T + x*(1-T)
            const float Hh = fmaxf(acc[tm][r] + bia, 0.f);
            const float Tg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (acc2[tm][r] + bia2)));
            xreg[tm][r] = Hh * Tg + xreg[tm][r] * (1.f - Tg);
          }
      } else {
        ch_mma_loop<TM, LDSW, false, PF>(L.K16, L.NT, L.bh, L.bl, wn, L.bh, L.bl, wn, xhi, xlo, wm * TM * 32, l31, lh, acc, acc2, krot * L.K16 >> 5);
#ifdef TACO_TRACE
        CTRC(trci); ++trci;
#endif
        const bool relu = L.act == ACT_RELU;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float v = acc[tm][r] + bia; xreg[tm][r] = relu ? fmaxf(v, 0.f) : v; }
      }
    }
    const bool last_hidden = (li + 1 == a.nlayers) || a_in.L[li + 1].type == CH_XPROJ;
    if (last_hidden && a.y) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + rloc(tm, r);
          if (row < a.M) a.y[(size_t)row * a.ldy + col] = xreg[tm][r];
        }
    }
#ifdef TACO_TRACE
    CTRC(trci); ++trci;
#endif
    __syncthreads();                            // every wave has read the planes of this layer
#ifdef TACO_TRACE
    CTRC(trci); ++trci;
#endif
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float f = xreg[tm][r];
        const unsigned hp = taco_pk_bf16(f, 0.f) & 0xffffu;
        const unsigned lp = taco_pk_bf16(f - __uint_as_float(hp << 16), 0.f) & 0xffffu;
        xhi[rloc(tm, r) * LDSW + col] = (unsigned short)hp;
        xlo[rloc(tm, r) * LDSW + col] = (unsigned short)lp;
      }
    __syncthreads();
#ifdef TACO_TRACE
    CTRC(trci); ++trci;
#endif
  }
}
