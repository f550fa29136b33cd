// taco_bigru_xcd.h -- the BiGRU scan of a CBHG (modules.py:82-96: bidirectional_dynamic_rnn over two GRUCell(H))
// at H = 256 (the post-net) as ONE persistent launch on the exchange machinery of taco_decoder_xcd.h.
//
// The input projection is hoisted (one GEMM over all frames, taco_lib.hip); what is sequential is, per direction and batch row,
// T steps of  r,u = sigmoid(xg + h Wg_h);  c = tanh(xc + (r*h) Wc_h);  h' = u*h + (1-u)*c  -- 196 K MACs and two dependent
// mat-vecs per step.  k_bigru_resw gives every (direction, row) chain one CU and keeps 2/3 of the recurrent weights on it:
// 3.74 us per step on 64 of the 256 CUs (post-net, C2: 1.9 ms).  Here the 64 chains are spread over the whole chip:
//   * 16 groups of 16 CUs (two groups per XCD); a group owns ONE direction and RG batch rows (C2: 4) for all T steps;
//   * member m of a group owns hidden units 16m..16m+15, wave w of it the pair 16m+2w, 16m+2w+1, and every lane keeps its 4-input
//     slice of the pair's six weight columns (r, u, candidate x 2 units) in 24 VGPRs for the whole scan: nothing is re-read;
//   * per step two exchanges through the XCD's L2 (8-byte {value, tag} granules, see taco_decoder_xcd.h): r*h after the gates,
//     h' after the candidate.  The x-parts come from HBM / the Infinity Cache (100 MB at C2): the member's slice of them is fetched
//     16 steps at a time, one block ahead, into an LDS ring -- a far load inside the step loop would be waited for at the loop's
//     back edge (and, loads returning in order, by the very next exchange poll) on every step.
// Length masking and the reverse_sequence time mapping of the backward direction follow A.7 exactly as k_bigru_resw does.
#pragma once
#include "taco_decoder_xcd.h"

#define GX_MEMBERS 16
#define GX_NGROUP 16
#define GX_NREG 24
#define GX_H 256
#define GX_BLK 16            // steps per block of prefetched x-parts
__host__ __device__ inline size_t gx_lds_floats(int RG) { return (size_t)2 * RG * GX_H + (size_t)2 * GX_BLK * RG * 48 + 64; }

struct GxArgs {
  const float* wpack0; const float* wpack1;   // [16 members][GX_NREG][DX_NT] per direction
  const float* xproj;                         // [B*T, 6H] hoisted input projection, biases folded, backward direction time-reversed
  const float* h0;                            // [B, 2H] initial states (fw | bw) or null
  const int* lengths;                         // [B] or null (= T)
  float* out;                                 // [B, T, 2H]
  unsigned long long* xbuf; unsigned* ctl; unsigned* err; long long* trace;
  int B, T, force_wt;
};
#define GX_STAMP(slot)                                                                                            \
  do {                                                                                                            \
    if (tracer && s >= 8 && s < 8 + DX_TRACE_STEPS) a.trace[(s - 8) * DX_TRACE_SLOTS + (slot)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
__host__ __device__ inline size_t gx_xbuf_granules(int RG) { return (size_t)GX_NGROUP * 2 * RG * GX_H; }

template <int RG>
__global__ __launch_bounds__(DX_NT) void k_bigru_xcd(const GxArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float gx_smem[];
  GxArgs a = a_in;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int H = GX_H, RL = DxRL<RG>::value;
  float* hs = gx_smem;                 // [RG][H] state
  float* xs = hs + RG * H;             // [RG][H] r * h
  float* xq = xs + RG * H;             // [2][GX_BLK][RG][3][16]: x-parts of the member's 16 units, two blocks of GX_BLK steps
  int* ictl = reinterpret_cast<int*>(xq + 2 * GX_BLK * RG * 48);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, a.force_wt, ictl, tid, 24);
  const int place = __builtin_amdgcn_readfirstlane(ictl[0]), slot = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  // XCD-local: the two halves of an XCD's 32 workgroups are two groups; otherwise (slot = blockIdx / 8, place = blockIdx % 8) the
  // same arithmetic gives 16 groups of 16 by block index
  const int group = place * 2 + (slot >> 4), member = slot & 15;
  const int dir = group & 1, row0 = (group >> 1) * RG;
  if (row0 >= a.B) return;
  const int T = a.T;
  const bool tracer = a.trace && group == 0 && member == 0 && tid == 0;

  float W[GX_NREG];
  {
    const float* wp = (dir ? a.wpack1 : a.wpack0) + ((size_t)member * GX_NREG) * DX_NT + tid;
#pragma unroll
    for (int j = 0; j < GX_NREG; ++j) W[j] = wp[(size_t)j * DX_NT];
  }
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)group * 2 * RG * H;       // [r*h : RG x H][h' : RG x H]
  for (int i = tid; i < RG * H; i += DX_NT) {
    const int r = i / H, n = i % H, b = row0 + r;
    hs[i] = (a.h0 && b < a.B) ? a.h0[(size_t)b * 2 * H + dir * H + n] : 0.f;
    xs[i] = 0.f;
  }
  // epilogue role: quad 0 of every wave owns (rows dx_row(lane, q), units ua, ua+1)
  const bool epl = lane < (RG >= 4 ? 4 : RG);
  const int ua = member * 16 + wave * 2;
  int erow[RL], eL[RL];
  bool evalid[RL];
#pragma unroll
  for (int q = 0; q < RL; ++q) {
    erow[q] = dx_row<RG>(lane & 3, q);
    evalid[q] = epl && (row0 + erow[q] < a.B);
    eL[q] = evalid[q] ? (a.lengths ? a.lengths[row0 + erow[q]] : T) : 0;
  }
  // x-part blocks: item i = (step j, row r, gate g, quarter c4) of a block -> one float4 of the member's 16 units
  constexpr int NIT = GX_BLK * RG * 3 * 4, NLD = (NIT + DX_NT - 1) / DX_NT;
  float4 xld[NLD];
  auto blk_load = [&](int s0) {          // unconditional loads from clamped addresses (nothing is waited for here)
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = min(u * DX_NT + tid, NIT - 1);
      const int c4 = i & 3, g = (i >> 2) % 3, r = (i / 12) % RG, j = i / (12 * RG);
      const int b = min(row0 + r, a.B - 1), sx = min(s0 + j, T - 1);
      xld[u] = *reinterpret_cast<const float4*>(a.xproj + ((size_t)b * T + sx) * 6 * H + dir * 3 * H + g * H + member * 16 + 4 * c4);
    }
  };
  auto blk_store = [&](int ring) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = u * DX_NT + tid;
      if (i < NIT) *reinterpret_cast<float4*>(xq + (size_t)ring * GX_BLK * RG * 48 + 4 * i) = xld[u];
    }
  };
  blk_load(0); blk_store(0);
  blk_load(GX_BLK); blk_store(1);
  __syncthreads();

  const int tid_outer = tid, lane_outer = lane;
  for (int s = 0; s < T; ++s) {
    const unsigned tag = (unsigned)s + 1u;
    int tid = tid_outer, lane = lane_outer;                 // opaque per-iteration copies: see taco_decoder_xcd.h
    asm volatile("" : "+v"(tid), "+v"(lane));
    GX_STAMP(0);
    const int sb = s & (GX_BLK - 1), ring = (s / GX_BLK) & 1;
    if (sb == 0 && s > 0) blk_load(s + GX_BLK);            // the block after next ... (its ring slot was last read one step ago)
    float2 x0[RL][3];
#pragma unroll
    for (int q = 0; q < RL; ++q)
#pragma unroll
      for (int g = 0; g < 3; ++g)
        x0[q][g] = *reinterpret_cast<const float2*>(xq + ((size_t)(ring * GX_BLK + sb) * RG + erow[q]) * 48 + g * 16 + wave * 2);
    float ha[RL], hb[RL], ga[RL], gb[RL];
#pragma unroll
    for (int q = 0; q < RL; ++q) { ha[q] = hs[erow[q] * H + ua]; hb[q] = hs[erow[q] * H + ua + 1]; }
    // ---- gates: r, u of both units ----
    {
      float acc[4][RG], sm[4][RL];
      dx_zero<4, RG>(acc);
      dx_pass<0, 4, RG, GX_NREG, GX_H>(W, hs, lane, acc);
      dx_reduce<4, RG>(acc, sm, lane);
      GX_STAMP(1);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const float ra = dx_sigmoid_fast(sm[0][q] + x0[q][0].x), rb = dx_sigmoid_fast(sm[2][q] + x0[q][0].y);
        ga[q] = dx_sigmoid_fast(sm[1][q] + x0[q][1].x);
        gb[q] = dx_sigmoid_fast(sm[3][q] + x0[q][1].y);
        if (epl) {
          dx_publish(X + erow[q] * H + ua, ra * ha[q], tag, rt);
          dx_publish(X + erow[q] * H + ua + 1, rb * hb[q], tag, rt);
        }
      }
    }
    GX_STAMP(2);
    dx_gather<RG, GX_H, false, GX_H>(X, tag, xs, 0, 0, 0, tid, rt);
    GX_STAMP(3);
    __syncthreads();
    GX_STAMP(4);
    // ---- candidate and the new state ----
    {
      float acc[2][RG], sm[2][RL];
      dx_zero<2, RG>(acc);
      dx_pass<16, 2, RG, GX_NREG, GX_H>(W, xs, lane, acc);
      dx_reduce<2, RG>(acc, sm, lane);
      GX_STAMP(5);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const float ca = taco_tanh_fast(sm[0][q] + x0[q][2].x), cb = taco_tanh_fast(sm[1][q] + x0[q][2].y);
        const bool active = s < eL[q];                       // A.7: row active iff s < L; forward t = s, backward t = L-1-s
        const float na = active ? ga[q] * ha[q] + (1.f - ga[q]) * ca : ha[q];
        const float nb = active ? gb[q] * hb[q] + (1.f - gb[q]) * cb : hb[q];
        if (epl) {
          dx_publish(X + RG * H + erow[q] * H + ua, na, tag, rt);
          dx_publish(X + RG * H + erow[q] * H + ua + 1, nb, tag, rt);
          if (evalid[q]) {
            const int t = (dir && active) ? (eL[q] - 1 - s) : s;
            *reinterpret_cast<float2*>(a.out + ((size_t)(row0 + erow[q]) * T + t) * 2 * H + dir * H + ua) =
                active ? make_float2(na, nb) : make_float2(0.f, 0.f);
          }
        }
      }
    }
    GX_STAMP(6);
    dx_gather<RG, GX_H, false, GX_H>(X + RG * H, tag, hs, 0, 0, 0, tid, rt);
    GX_STAMP(7);
    if (sb == GX_BLK / 2 && s > GX_BLK) blk_store(ring ^ 1);     // ... lands half a block later in the slot the previous block vacated
    GX_STAMP(8);
    __syncthreads();
    GX_STAMP(9);
  }
}
