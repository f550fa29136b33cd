// taco_bigru_xcd.h -- the BiGRU scan of a CBHG (modules.py:82-96: bidirectional_dynamic_rnn over two GRUCell(H))
// at H = 256 (the post-net) as ONE persistent launch on the exchange machinery of taco_decoder_xcd.h.
//
// The input projection is hoisted (one GEMM over all frames, taco_lib.hip); what is sequential is, per direction and batch row,
// T steps of  r,u = sigmoid(xg + h Wg_h);  c = tanh(xc + (r*h) Wc_h);  h' = u*h + (1-u)*c  -- 196 K MACs and two dependent
// mat-vecs per step.  k_bigru_resw gives every (direction, row) chain one CU and keeps 2/3 of the recurrent weights on it:
// 3.74 us per step on 64 of the 256 CUs (post-net, C2: 1.9 ms).  Here the 64 chains are spread over the whole chip:
//   * groups of 16 members; a group owns ONE direction and RG batch rows for all T steps; member m owns hidden units
//     16m..16m+15, wave w of it UPW = 16 / NWV consecutive units, and every lane keeps its 4-input slice of the wave's 3 UPW
//     weight columns (r, u per unit, then the candidates) in VGPRs for the whole scan: nothing is re-read;
//   * per step two exchanges through the XCD's L2 (8-byte {value, tag} granules, see taco_decoder_xcd.h): r*h after the gates,
//     h' after the candidate.  The x-parts come from HBM / the Infinity Cache (100 MB at C2): the member's slice of them is fetched
//     16 steps at a time, one block ahead, STRAIGHT INTO an LDS ring (global_load_lds_dwordx4: no VGPR is involved, so nothing can
//     be demoted to scratch and nothing is waited for where the load is issued).  Loads return in order, so the first exchange
//     poll after the issue also waits for it -- once per 16 steps, behind the gates pass of that step.  (Round 2 staged the block
//     through a float4 array that the compiler kept in scratch: the far load was waited for at issue.)
// Two geometries (NWV = waves per workgroup):
//   NWV = 8 (default): 256 workgroups of 512 threads, one per CU, 16 groups (two per XCD), 2 units per wave.
//   NWV = 4 (taco_debug_set_persistent(m, 9)): 512 workgroups of 256 threads, TWO per CU, 32 groups (four per XCD) of half as
//            many rows, 4 units per wave; the two workgroups of a CU belong to different groups, i.e. to independent chains, so
//            one could compute alone on its SIMDs while the other waits for an exchange.  Measured at C2: 6088 clocks per step
//            against 5224 -- the compute phases do not shrink (they are chains of dependent instructions, not issue-bound) and
//            the epilogue of four units per lane is longer.  Kept as the record of that experiment and as a second test geometry.
// Length masking and the reverse_sequence time mapping of the backward direction follow A.7 exactly as k_bigru_resw does.
#pragma once
#include "taco_decoder_xcd.h"

#define GX_MEMBERS 16
#define GX_H 256
#define GX_BLK 16            // steps per block of prefetched x-parts
__host__ __device__ inline int gx_nreg(int NWV) { return 12 * (16 / NWV); }              // weight registers per thread
__host__ __device__ inline int gx_ngroups(int NWV) { return 8 * (16 / NWV); }            // 16 (NWV 8: two per XCD) or 32 (NWV 4: four)
__host__ __device__ inline size_t gx_lds_floats(int RG) { return (size_t)2 * RG * GX_H + (size_t)2 * GX_BLK * RG * 48 + 64; }
typedef __attribute__((address_space(3))) float gx_lds_float;
// one 16-byte-per-lane load from global memory straight into LDS: the wave's 64 x 16 bytes land contiguously at LDS byte address
// `lds_dst` (wave-uniform, handed over in M0) + 16 * lane.  hipcc does not count this load: the caller waits (s_waitcnt vmcnt) before the
// data is read, and a workgroup barrier lies between that wait and readers in other waves.
__device__ __forceinline__ void gx_load_lds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct GxArgs {
  const float* wpack0; const float* wpack1;   // [16 members][gx_nreg][64 NWV] per direction
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
// exchange granules of a launch: groups x [r*h | h'] x RG rows x H (the same for both geometries at their largest RG: 16 x 8 = 32 x 4)
__host__ __device__ inline size_t gx_xbuf_granules(int ngroups, int RG) { return (size_t)ngroups * 2 * RG * GX_H; }

template <int RG, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_bigru_xcd(const GxArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float gx_smem[];
  GxArgs a = a_in;
  constexpr int NT = 64 * NWV, UPW = 16 / NWV, NREG = 12 * UPW;
  constexpr int GPX = 1024 / NT;              // groups per XCD: 2 or 4
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int H = GX_H, RL = DxRL<RG>::value;
  float* xq = gx_smem;                 // [2][GX_BLK][RG][3][16]: x-parts of the member's 16 units, two blocks of GX_BLK steps (first:
                                       // its LDS addresses go through M0 and stay below 48 KB)
  float* hs = xq + 2 * GX_BLK * RG * 48;   // [RG][H] state
  float* xs = hs + RG * H;             // [RG][H] r * h
  int* ictl = reinterpret_cast<int*>(xs + RG * H);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, a.force_wt, ictl, tid, 24);
  const int place = __builtin_amdgcn_readfirstlane(ictl[0]), slot = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  // XCD-local: the workgroups of an XCD are dealt round-robin to its GPX groups (two workgroups that arrive together -- the pair
  // of one CU, most likely -- end up in different groups); otherwise (slot = blockIdx / 8, place = blockIdx % 8) the same
  // arithmetic gives 8 GPX groups of 16 by block index
  const int group = place * GPX + (slot % GPX), member = slot / GPX;
  const int dir = group & 1, row0 = (group >> 1) * RG;
  if (row0 >= a.B) return;
  const int T = a.T;
  const bool tracer = a.trace && group == 0 && member == 0 && tid == 0;

  float W[NREG];
  {
    const float* wp = (dir ? a.wpack1 : a.wpack0) + ((size_t)member * NREG) * NT + tid;
#pragma unroll
    for (int j = 0; j < NREG; ++j) W[j] = wp[(size_t)j * NT];
  }
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)group * 2 * RG * H;       // [r*h : RG x H][h' : RG x H]
  for (int i = tid; i < RG * H; i += NT) {
    const int r = i / H, n = i % H, b = row0 + r;
    hs[i] = (a.h0 && b < a.B) ? a.h0[(size_t)b * 2 * H + dir * H + n] : 0.f;
    xs[i] = 0.f;
  }
  // epilogue role: quad 0 of every wave owns (rows dx_row(lane, q), units u0 .. u0+UPW-1)
  const bool epl = lane < (RG >= 4 ? 4 : RG);
  const int u0 = member * 16 + wave * UPW;
  int erow[RL], eL[RL];
  bool evalid[RL];
#pragma unroll
  for (int q = 0; q < RL; ++q) {
    erow[q] = dx_row<RG>(lane & 3, q);
    evalid[q] = epl && (row0 + erow[q] < a.B);
    eL[q] = evalid[q] ? (a.lengths ? a.lengths[row0 + erow[q]] : T) : 0;
  }
  // x-part blocks: item i = (step j, row r, gate g, quarter c4) of a block -> one float4 of the member's 16 units; item i of a block
  // lives at float offset 4 * i of its ring slot, so the 64 items of one wave-load are contiguous in LDS as the LDS-direct load needs
  constexpr int NIT = GX_BLK * RG * 3 * 4, NLD = (NIT + NT - 1) / NT;
  static_assert(NIT % 64 == 0, "a wave's 64 items are all inside the block or all outside");
  const unsigned xq_lds = (unsigned)(size_t)(gx_lds_float*)xq;
  auto blk_fetch = [&](int s0, int ring) {       // unconditional addresses (clamped); nothing is waited for here
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      if (u * NT + wave * 64 < NIT) {            // wave-uniform
        const int i = u * NT + tid;
        const int c4 = i & 3, g = (i >> 2) % 3, r = (i / 12) % RG, j = i / (12 * RG);
        const int b = min(row0 + r, a.B - 1), sx = min(s0 + j, T - 1);
        const float* src = a.xproj + ((size_t)b * T + sx) * 6 * H + dir * 3 * H + g * H + member * 16 + 4 * c4;
        const unsigned dst = __builtin_amdgcn_readfirstlane(xq_lds + (unsigned)(ring * GX_BLK * RG * 48 + 4 * (u * NT + wave * 64)) * 4u);
        gx_load_lds16(src, dst);
      }
    }
  };
  blk_fetch(0, 0);
  blk_fetch(GX_BLK, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int tid_outer = tid, lane_outer = lane;
  for (int s = 0; s < T; ++s) {
    const unsigned tag = (unsigned)s + 1u;
    int tid = tid_outer, lane = lane_outer;                 // opaque per-iteration copies: see taco_decoder_xcd.h
    asm volatile("" : "+v"(tid), "+v"(lane));
    GX_STAMP(0);
    const int sb = s & (GX_BLK - 1), ring = (s / GX_BLK) & 1;
    // the block after next goes into the slot whose last reader finished before the barrier that ended the previous step; it is
    // first read GX_BLK steps from now
    if (sb == 0 && s > 0) blk_fetch(s + GX_BLK, ring ^ 1);
    float x0[RL][3][UPW];
#pragma unroll
    for (int q = 0; q < RL; ++q)
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const float* xp = xq + ((size_t)(ring * GX_BLK + sb) * RG + erow[q]) * 48 + g * 16 + wave * UPW;
        if constexpr (UPW == 4) { const float4 v = *reinterpret_cast<const float4*>(xp); x0[q][g][0] = v.x; x0[q][g][1] = v.y; x0[q][g][2] = v.z; x0[q][g][3] = v.w; }
        else { const float2 v = *reinterpret_cast<const float2*>(xp); x0[q][g][0] = v.x; x0[q][g][1] = v.y; }
      }
    float hv[RL][UPW], gv[RL][UPW];
#pragma unroll
    for (int q = 0; q < RL; ++q)
#pragma unroll
      for (int i = 0; i < UPW; ++i) hv[q][i] = hs[erow[q] * H + u0 + i];
    // ---- gates: r, u of the wave's units (columns r_0, u_0, r_1, u_1, ...) ----
    {
      float acc[2 * UPW][RG], sm[2 * UPW][RL];
      dx_zero<2 * UPW, RG>(acc);
      dx_pass<0, 2 * UPW, RG, NREG, GX_H>(W, hs, lane, acc);
      dx_reduce<2 * UPW, RG>(acc, sm, lane);
      GX_STAMP(1);
      // straight-line epilogue: all sigmoids first (independent chains interleave), then all stores behind one branch
      float rh[RL][UPW];
#pragma unroll
      for (int q = 0; q < RL; ++q)
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
          const float rr = dx_sigmoid_fast(sm[2 * i][q] + x0[q][0][i]);
          gv[q][i] = dx_sigmoid_fast(sm[2 * i + 1][q] + x0[q][1][i]);
          rh[q][i] = rr * hv[q][i];
        }
      if (epl) {
#pragma unroll
        for (int q = 0; q < RL; ++q) dx_publish_n<UPW>(X + erow[q] * H + u0, 1, rh[q], tag, rt);
      }
    }
    GX_STAMP(2);
    dx_gather<RG, GX_H, false, GX_H, NT>(X, tag, xs, 0, 0, 0, tid, rt);
    GX_STAMP(3);
    __syncthreads();
    GX_STAMP(4);
    // ---- candidate and the new state ----
    {
      float acc[UPW][RG], sm[UPW][RL];
      dx_zero<UPW, RG>(acc);
      dx_pass<8 * UPW, UPW, RG, NREG, GX_H>(W, xs, lane, acc);
      dx_reduce<UPW, RG>(acc, sm, lane);
      GX_STAMP(5);
      float nv[RL][UPW];
      bool active[RL];
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        active[q] = s < eL[q];                               // A.7: row active iff s < L; forward t = s, backward t = L-1-s
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
          const float cc = taco_tanh_fast(sm[i][q] + x0[q][2][i]);
          float blend = gv[q][i] * hv[q][i] + (1.f - gv[q][i]) * cc;
          DX_PIN(blend);
          nv[q][i] = active[q] ? blend : hv[q][i];
        }
      }
      if (epl) {
#pragma unroll
        for (int q = 0; q < RL; ++q) dx_publish_n<UPW>(X + RG * H + erow[q] * H + u0, 1, nv[q], tag, rt);
      }
#pragma unroll
      for (int q = 0; q < RL; ++q)
        if (evalid[q]) {
          const int t = (dir && active[q]) ? (eL[q] - 1 - s) : s;
          float* po = a.out + ((size_t)(row0 + erow[q]) * T + t) * 2 * H + dir * H + u0;
          if constexpr (UPW == 4) *reinterpret_cast<float4*>(po) = active[q] ? make_float4(nv[q][0], nv[q][1], nv[q][2], nv[q][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
          else *reinterpret_cast<float2*>(po) = active[q] ? make_float2(nv[q][0], nv[q][1]) : make_float2(0.f, 0.f);
        }
    }
    GX_STAMP(6);
    dx_gather<RG, GX_H, false, GX_H, NT>(X + RG * H, tag, hs, 0, 0, 0, tid, rt);
    GX_STAMP(7);
    // the slot fetched at the start of this block is read from the next step on: every wave makes sure ITS part has landed (it
    // has, long ago: waves that poll have waited for it at their first poll already) ahead of the barrier that publishes it
    if (sb == GX_BLK - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GX_STAMP(8);
    __syncthreads();
    GX_STAMP(9);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_bigru_duo<RG>: the same scan with BOTH directions of RG batch rows owned by one group of 32 CUs (one XCD), the two directions
// software-pipelined against each other inside every wave.
//
// k_bigru_xcd spends 42 % of a step waiting for its two exchanges: publish -> L2 -> poll is ~0.5 us and nothing else can run,
// because the step is one dependent chain.  The forward and the backward direction of a BiGRU are two INDEPENDENT chains.  Here a
// member owns hidden units 8m .. 8m+7 of both directions (wave w: unit 8m + w, three weight columns per direction = 24 VGPRs, the
// same register budget as before), and a step interleaves them:
//
//     gates F -> publish r*h(F) | gather h'(B, previous step) | gates B -> publish r*h(B) | gather r*h(F) | candidate F -> publish
//     h'(F) | gather r*h(B) | candidate B -> publish h'(B) | gather h'(F)
//
// so that every exchange has one compute phase of the other direction between its publish and its gather.  The polls are issued
// EARLY as well: the loads of a gather are requested before the compute phase in front of it and examined after it (an L2 read
// round trip is ~300 clocks even when the granule is already there), and only re-polled if a producer was late.  The per-wave
// instruction count of a step is the same as k_bigru_xcd's (each direction still does RG rows per pass: the DPP reduction cost is
// per column, not per row, so splitting the ROWS in two sets would have doubled it -- which is what the NWV = 4 geometry measured).
// Rows per group: B <= 8: 1, <= 16: 2, <= 32: 4, <= 64: 8.
// ------------------------------------------------------------------------------------------------------------------------------
#define GD_MEMBERS 32
#define GD_NREG 24           // weight registers per thread: [dir][r | u | c][4 inputs]
__host__ __device__ inline size_t gd_ring_floats(int RG) { return (size_t)2 * 2 * GX_BLK * RG * 24; }      // [slot][dir][step][row][gate][8 units]
__host__ __device__ inline size_t gd_lds_floats(int RG) { return gd_ring_floats(RG) + (size_t)4 * RG * GX_H + 64; }
__host__ __device__ inline size_t gd_xbuf_granules(int RG) { return (size_t)DX_NGROUP * 4 * RG * GX_H; }  // groups x [dir][r*h | h'] x RG x H

// the loads of a gather, requested now and examined later (dx_gather split in two)
template <int RG, int NT>
struct GdPoll {
  static constexpr int NI = (RG * GX_H + NT - 1) / NT;
  unsigned long long g[NI];
};
template <int RG, int NT>
__device__ __forceinline__ void gd_request(const dx_gu64* X, int tid, GdPoll<RG, NT>& p) {
  constexpr int NI = GdPoll<RG, NT>::NI;
  if ((RG * GX_H >= NT) || tid < RG * GX_H) {
#pragma unroll
    for (int u = 0; u < NI; ++u) p.g[u] = __hip_atomic_load(X + tid + u * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// Makes the requested granules "used" at this point of the program: hipcc places the wait for the loads HERE.  Called right before a
// phase publishes, so that the wait sees only the loads (requested a whole compute phase ago: landed) and not the publish stores
// issued after them -- vmcnt counts in issue order, and a wait behind the stores would also wait for their acknowledgement.
// `after` ties the statement to a value of the phase's epilogue, so that the scheduler cannot hoist it (and the wait) above the
// phase's arithmetic.
template <int RG, int NT>
__device__ __forceinline__ void gd_landed(GdPoll<RG, NT>& p, float& after) {
#pragma unroll
  for (int u = 0; u < GdPoll<RG, NT>::NI; ++u) asm volatile("" : "+v"(p.g[u]), "+v"(after));
}
template <int RG, int NT>
__device__ __forceinline__ void gd_collect(const dx_gu64* X, unsigned tag, float* st, int tid, GdPoll<RG, NT>& p, DxRt& rt) {
  constexpr int NI = GdPoll<RG, NT>::NI;
  if ((RG * GX_H >= NT) || tid < RG * GX_H) {
    bool ok = true;
#pragma unroll
    for (int u = 0; u < NI; ++u) ok = ok && ((unsigned)(p.g[u] >> 32) == tag);
    float v[NI];
    if (ok) {
#pragma unroll
      for (int u = 0; u < NI; ++u) v[u] = __uint_as_float((unsigned)p.g[u]);
    } else {
      dx_poll<NI>(X + tid, NT, tag, v, rt);       // a producer was late: the ordinary bounded poll
    }
#pragma unroll
    for (int u = 0; u < NI; ++u) st[u * NT + tid] = v[u];      // [RG][H] row-major == granule order
  }
}

struct GdArgs {
  const float* wpack;                         // [2 dirs][32 members][GD_NREG / 2][512]
  const float* xproj;                         // [B*T, 6H] hoisted input projection, biases folded, backward direction time-reversed
  const float* h0;                            // [B, 2H] initial states (fw | bw) or null
  const int* lengths;                         // [B] or null (= T)
  float* out;                                 // [B, T, 2H]
  float* gsave;                               // TAPE instantiation: [B*T, 6H] gates (r | u | c per direction) at the TRUE time index (zeroed by the caller)
  unsigned long long* xbuf; unsigned* ctl; unsigned* err; long long* trace;
  int B, T, force_wt;
};
#define GDT_STAMP(slot)                                                                                           \
  do {                                                                                                            \
    if constexpr (TRACE) { if (tracer && s >= 8 && s < 8 + DX_TRACE_STEPS) a.trace[(s - 8) * DX_TRACE_SLOTS + (slot)] = (long long)__builtin_readcyclecounter(); } \
  } while (0)

// (WT / TRACE: the protocol and the stamps as template parameters of the body, as in k_bigru_oct: no protocol branch per publish, no stamp
// branches in the production instantiations)
template <int RG, bool TAPE, bool WT, bool TRACE>
__device__ __forceinline__ void gd_body(const GdArgs& a, float* gx_smem, int group, int member, DxRt rt) {
  constexpr int WTC = WT ? 1 : 0;
  constexpr int NT = 512, H = GX_H, RL = DxRL<RG>::value;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int SLOT = 2 * GX_BLK * RG * 24;      // floats of one ring slot (both directions)
  float* xq = gx_smem;                            // ring first: its LDS addresses go through M0
  float* hs = xq + 2 * SLOT;                      // [2 dirs][RG][H] states
  float* xs = hs + 2 * RG * H;                    // [2 dirs][RG][H] r * h
  const int row0 = group * RG;
  if (row0 >= a.B || member >= GD_MEMBERS) return;
  const int T = a.T;
  const bool tracer = TRACE && a.trace && group == 0 && member == 0 && tid == 0;

  float W[GD_NREG];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const float* wp = a.wpack + (((size_t)d * GD_MEMBERS + member) * 12) * NT + tid;
#pragma unroll
    for (int j = 0; j < 12; ++j) W[12 * d + j] = wp[(size_t)j * NT];
  }
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)group * 4 * RG * H;       // [dir][r*h : RG x H | h' : RG x H]
  for (int i = tid; i < 2 * RG * H; i += NT) {
    const int d = i / (RG * H), r = (i / H) % RG, n = i % H, b = row0 + r;
    hs[i] = (a.h0 && b < a.B) ? a.h0[(size_t)b * 2 * H + d * H + n] : 0.f;
    xs[i] = 0.f;
  }
  // epilogue role: quad 0 of every wave owns (rows dx_row(lane, q), unit 8 member + wave) of both directions
  const bool epl = lane < (RG >= 4 ? 4 : RG);
  const int u0 = member * 8 + wave;
  int erow[RL], eL[RL];
  bool evalid[RL];
#pragma unroll
  for (int q = 0; q < RL; ++q) {
    erow[q] = dx_row<RG>(lane & 3, q);
    evalid[q] = epl && (row0 + erow[q] < a.B);
    eL[q] = evalid[q] ? (a.lengths ? a.lengths[row0 + erow[q]] : T) : 0;
  }
  // x-part blocks of both directions: item i = (dir, step j, row r, gate g, half c2) -> one float4 of the member's 8 units
  constexpr int NIT = 2 * GX_BLK * RG * 3 * 2, NLD = (NIT + NT - 1) / NT;
  static_assert(NIT % 64 == 0, "a wave's 64 items are all inside the block or all outside");
  const unsigned xq_lds = (unsigned)(size_t)(gx_lds_float*)xq;
  auto blk_fetch = [&](int s0, int ring) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      if (u * NT + wave * 64 < NIT) {            // wave-uniform
        const int i = u * NT + tid;
        const int c2 = i & 1, g = (i >> 1) % 3, r = (i / 6) % RG, j = (i / (6 * RG)) % GX_BLK, d = i / (6 * RG * GX_BLK);
        const int b = min(row0 + r, a.B - 1), sx = min(s0 + j, T - 1);
        const float* src = a.xproj + ((size_t)b * T + sx) * 6 * H + d * 3 * H + g * H + member * 8 + 4 * c2;
        const unsigned dst = __builtin_amdgcn_readfirstlane(xq_lds + (unsigned)(ring * SLOT + 4 * (u * NT + wave * 64)) * 4u);
        gx_load_lds16(src, dst);
      }
    }
  };
  blk_fetch(0, 0);
  blk_fetch(GX_BLK, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // per-direction registers that live from the gates phase to the candidate phase of a step
  float x0[2][RL][3], hv[2][RL], gv[2][RL];
  GdPoll<RG, NT> pre;       // the gather whose loads are in flight across the current compute phase
#pragma unroll
  for (int u = 0; u < GdPoll<RG, NT>::NI; ++u) pre.g[u] = 0ull;

  const int tid_outer = tid, lane_outer = lane;
  for (int s = 0; s < T; ++s) {
    const unsigned tag = (unsigned)s + 1u;
    int tid = tid_outer, lane = lane_outer;                 // opaque per-iteration copies: see taco_decoder_xcd.h
    asm volatile("" : "+v"(tid), "+v"(lane));
    GDT_STAMP(0);
    const int sb = s & (GX_BLK - 1), ring = (s / GX_BLK) & 1;
    if (sb == 0 && s > 0) blk_fetch(s + GX_BLK, ring ^ 1);
    // one lambda per phase kind; D is a compile-time direction
    auto gates = [&](auto Dc) {
      constexpr int D = decltype(Dc)::value;
      const float* hsd = hs + D * RG * H;
#pragma unroll
      for (int q = 0; q < RL; ++q) {
#pragma unroll
        for (int g = 0; g < 3; ++g) x0[D][q][g] = xq[(size_t)ring * SLOT + ((size_t)(D * GX_BLK + sb) * RG + erow[q]) * 24 + g * 8 + wave];
        hv[D][q] = hsd[erow[q] * H + u0];
      }
      float acc[2][RG], sm[2][RL];
      dx_zero<2, RG>(acc);
      dx_pass<12 * D, 2, RG, GD_NREG, GX_H>(W, hsd, lane, acc);
      dx_reduce<2, RG>(acc, sm, lane);
      float rh[RL][1];
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const float rr = dx_sigmoid_fast(sm[0][q] + x0[D][q][0]);
        gv[D][q] = dx_sigmoid_fast(sm[1][q] + x0[D][q][1]);
        rh[q][0] = rr * hv[D][q];
        if (TAPE && evalid[q] && s < eL[q]) {      // gates of the active steps at their true time (modules.py:82-96 / A.7), for the backward scan
          float* gs = a.gsave + ((size_t)(row0 + erow[q]) * T + (D ? eL[q] - 1 - s : s)) * 6 * H + D * 3 * H + u0;
          gs[0] = rr; gs[H] = gv[D][q];
        }
      }
      gd_landed<RG, NT>(pre, rh[RL - 1][0]);
      if (epl) {
#pragma unroll
        for (int q = 0; q < RL; ++q) dx_publish_n<1, WTC>(X + (size_t)D * 2 * RG * H + erow[q] * H + u0, 1, rh[q], tag, rt);
      }
    };
    auto cand = [&](auto Dc) {
      constexpr int D = decltype(Dc)::value;
      float acc[1][RG], sm[1][RL];
      dx_zero<1, RG>(acc);
      dx_pass<12 * D + 8, 1, RG, GD_NREG, GX_H>(W, xs + D * RG * H, lane, acc);
      dx_reduce<1, RG>(acc, sm, lane);
      float nv[RL][1];
      bool active[RL];
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        active[q] = s < eL[q];                               // A.7: row active iff s < L; forward t = s, backward t = L-1-s
        const float cc = taco_tanh_fast(sm[0][q] + x0[D][q][2]);
        float blend = gv[D][q] * hv[D][q] + (1.f - gv[D][q]) * cc;
        DX_PIN(blend);
        nv[q][0] = active[q] ? blend : hv[D][q];
        if (TAPE && evalid[q] && active[q]) a.gsave[((size_t)(row0 + erow[q]) * T + (D ? eL[q] - 1 - s : s)) * 6 * H + D * 3 * H + 2 * H + u0] = cc;
      }
      gd_landed<RG, NT>(pre, nv[RL - 1][0]);
      if (epl) {
#pragma unroll
        for (int q = 0; q < RL; ++q) dx_publish_n<1, WTC>(X + (size_t)(D * 2 + 1) * RG * H + erow[q] * H + u0, 1, nv[q], tag, rt);
      }
#pragma unroll
      for (int q = 0; q < RL; ++q)
        if (evalid[q]) {
          const int t = (D && active[q]) ? (eL[q] - 1 - s) : s;
          a.out[((size_t)(row0 + erow[q]) * T + t) * 2 * H + D * H + u0] = active[q] ? nv[q][0] : 0.f;
        }
    };
    using F = std::integral_constant<int, 0>;
    using Bk = std::integral_constant<int, 1>;
    const dx_gu64* X_rhF = X;                       const dx_gu64* X_hF = X + (size_t)RG * H;
    const dx_gu64* X_rhB = X + (size_t)2 * RG * H;  const dx_gu64* X_hB = X + (size_t)3 * RG * H;
    // (the loads for h'(B) of the previous step were requested at the end of that step)
    gates(F{});
    GDT_STAMP(1);
    if (s > 0) gd_collect<RG, NT>(X_hB, tag - 1u, hs + RG * H, tid, pre, rt);
    __syncthreads();
    GDT_STAMP(2);
    gd_request<RG, NT>(X_rhF, tid, pre);
    gates(Bk{});
    GDT_STAMP(3);
    gd_collect<RG, NT>(X_rhF, tag, xs, tid, pre, rt);
    __syncthreads();
    GDT_STAMP(4);
    gd_request<RG, NT>(X_rhB, tid, pre);
    cand(F{});
    GDT_STAMP(5);
    gd_collect<RG, NT>(X_rhB, tag, xs + RG * H, tid, pre, rt);
    __syncthreads();
    GDT_STAMP(6);
    gd_request<RG, NT>(X_hF, tid, pre);
    cand(Bk{});
    GDT_STAMP(7);
    gd_collect<RG, NT>(X_hF, tag, hs, tid, pre, rt);
    if (sb == GX_BLK - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's part of the next ring slot has landed
    __syncthreads();
    GDT_STAMP(8);
    gd_request<RG, NT>(X_hB, tid, pre);
  }
}

template <int RG, bool TAPE = false, bool TRACE = false>
__global__ __launch_bounds__(512) void k_bigru_duo(const GdArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float gx_smem[];
  GdArgs a = a_in;
  int* ictl = reinterpret_cast<int*>(gx_smem + gd_lds_floats(RG) - 64);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, a.force_wt, ictl, threadIdx.x, 24);
  const int group = __builtin_amdgcn_readfirstlane(ictl[0]), member = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  if (__builtin_amdgcn_readfirstlane((int)rt.wt)) gd_body<RG, TAPE, true, TRACE>(a, gx_smem, group, member, rt);
  else gd_body<RG, TAPE, false, TRACE>(a, gx_smem, group, member, rt);
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_bigru_oct<UPW, TAPE> (round 5): ONE batch row per cluster of 32 / UPW CUs -- UPW = 4: four clusters of 8 CUs per XCD, 32 rows (C2);
// UPW = 2: two clusters of 16, 16 rows; UPW = 1: one cluster of 32 (k_bigru_duo<1>'s geometry).
//
// k_bigru_duo gives a group of 32 CUs RG rows: a member owns 8 units x RG rows, a gather collects RG x 256 granules from 31 peers, a
// pass reads RG rows from LDS and the epilogue lanes run RG / 4 ... sigmoid chains one behind the other.  Its one-row geometry (C5)
// steps in 3456 clocks against 4672 at four rows (profiles/r04_v5_scan_timeline.txt).  Here every cluster has that geometry: a member
// owns 8 UPW units of both directions of ONE row -- the same 24 UPW / 96 weight registers per thread and the same FMAs per lane as RG =
// UPW rows of 8 units --, a gather is 256 granules from 32 / UPW - 1 peers (one per thread of HALF the workgroup: waves 0-3 collect the
// forward direction's vectors, waves 4-7 the backward direction's), a pass is one ds_read_b128 per lane, and length masking is a scalar
// test.  Wave w owns units 8 UPW m + UPW w + i: the wave reduction deals them to the lanes -- unit i's totals end up on lanes
// i * 64 / UPW ... -- so its first log2(UPW) levels are HALVING levels on permlane swaps (the partner gets the half it keeps; no
// selects, no DPP) and the epilogue is one sigmoid / tanh chain per lane whatever UPW is; every lane of a unit keeps the unit's state
// in a register, the first one publishes.  The step is k_bigru_duo's: the two directions pipelined against each other, polls requested a
// compute phase ahead.
// ------------------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t go_ring_floats(int UPW) { return (size_t)2 * 2 * GX_BLK * 3 * 8 * UPW; }      // [slot][dir][step][gate][8 UPW units]
__host__ __device__ inline size_t go_lds_floats(int UPW) { return go_ring_floats(UPW) + (size_t)4 * GX_H + 64; }
__host__ __device__ inline size_t go_xbuf_granules() { return (size_t)DX_NGROUP * 4 * 4 * GX_H; }            // <= 32 clusters x [dir][r*h | h'] x H

template <int UPW, int NG>
__device__ __forceinline__ void go_reduce(const float (&v)[UPW * NG], float (&out)[NG]) {
  float t[NG];
  if constexpr (UPW == 4) {
    float h[2 * NG];       // lanes 0-31: units 0, 1; lanes 32-63: units 2, 3
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j * NG + g]), __float_as_uint(v[(j + 2) * NG + g]), false, false);
        h[j * NG + g] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
#pragma unroll
    for (int g = 0; g < NG; ++g) {      // even rows of 16 lanes: units 0 | 2, odd rows: units 1 | 3 -> row i = unit i
      auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[g]), __float_as_uint(h[NG + g]), false, false);
      t[g] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
  } else if constexpr (UPW == 2) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {      // lanes 0-31: unit 0; lanes 32-63: unit 1
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[g]), __float_as_uint(v[NG + g]), false, false);
      t[g] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) t[g] = dx_xrow16(t[g]);
  } else {
#pragma unroll
    for (int g = 0; g < NG; ++g) t[g] = dx_xrow16(dx_xrow32(v[g]));
  }
  // the 16 lanes of a row: quad, then the two mirrors (every lane of a quad holds the quad's sum by then)
#pragma unroll
  for (int g = 0; g < NG; ++g) t[g] += DX_DPP0(t[g], 0xB1);
#pragma unroll
  for (int g = 0; g < NG; ++g) t[g] += DX_DPP0(t[g], 0x4E);
#pragma unroll
  for (int g = 0; g < NG; ++g) t[g] += DX_DPP0(t[g], 0x141);
#pragma unroll
  for (int g = 0; g < NG; ++g) out[g] = t[g] + DX_DPP0(t[g], 0x140);
}

// WT: the census found the fallback placement -- write-through exchanges (decided once per launch: the loop carries no protocol branch);
// TRACE: shader-clock stamps of row 0 / member 0 (tools/trace_bigru.py); the production instantiation has none of their branches.
#define GO_STAMP(slot)                                                                                            \
  do {                                                                                                            \
    if constexpr (TRACE || GO_DYN) { if (tracer && s >= 8 && s < 8 + DX_TRACE_STEPS) a.trace[(s - 8) * DX_TRACE_SLOTS + (slot)] = (long long)__builtin_readcyclecounter(); } \
  } while (0)
#ifndef GO_REQ2
#define GO_REQ2 1          // A/B (tools/scratch): 0 = no second, mid-phase request
#endif
// (Round 6, measured at C2 on top of the two requests, post-net alone 1.155 ms: without the second request 1.203; a sleep of 2 / 4 units in front of the fallback
// poll 1.171-1.191 / 1.219-1.225 -- so about one collect in four does fall through to the poll; a THIRD request behind the phase's reduction, checked
// only by the lanes whose first two answers were stale, 1.243-1.263: the load in front of the publish store delays the store.  Left as it is.)
#ifndef GO_DYN
#define GO_DYN 0           // A/B: 1 = the protocol test of every publish and the tracer test of every stamp at run time, as in k_bigru_duo / k_decoder_xcd
#endif
// A/B build -DGO_KNOB_RT (tools/sweep_go_knobs.py): per phase (gates F, gates B, cand F, cand B) the place of the second request (0: behind the
// products -- the production code --, 1: behind the reduction, 2: none), a sleep in front of the first request, a sleep in front of the
// fallback poll of the collect behind the phase; from a constant table the host fills from TACO_GO_KNOB before each launch.
// tools/sweep_go_knobs.py on that build (profiles/r06_sweep_go_knobs.txt, op-level call at the C2 post-net shape): every sleep loses (64 clocks in
// front of a request cost 8-16 us per call), the second request belongs behind the products where it is -- except in the gates B phase, where
// NONE is better (937 -> 929 us per call, both passes): the r*h(F) vector it asks for was published a whole phase + barrier earlier and the
// first request already has it.  GO_KNOB_TUNED = 0: the second request in all four phases, as rounds 5-6.
#ifndef GO_KNOB_TUNED
#define GO_KNOB_TUNED 1
#endif
__host__ __device__ constexpr int go_knob_default(int i) { return (GO_KNOB_TUNED && i == 1) ? 2 : 0; }
#ifdef GO_KNOB_RT
__constant__ int g_go_knob[16];
#define GO_KNOB(i) kn[i]
#else
#define GO_KNOB(i) go_knob_default(i)
#endif
// the backward scan's four second requests (phase_ca F / B, phase_b F / B): entries 12..15 of the same table (1 = none).  All sixteen settings,
// interleaved, on the whole C4-shard forward + backward with the gradients checked to the bit (tools/sweep_train_knobs.py,
// profiles/r06_sweep_train_knobs.txt): keeping all four is best (dropping any costs 10-30 us per pass).  (A first sweep had shown a gain of
// 14 us for two of them: with a request dropped, the previous phase's answer -- SAME step tag -- passed for the gather's own and the fallback
// poll never ran; the GPU suite caught it, a dropped request now clears the stale answer, and the tool compares gradients.)
__host__ __device__ constexpr int gob_knob_default(int i) { (void)i; return 0; }
#ifdef GO_KNOB_RT
#define GOB_KNOB(i) kb[i]
#else
#define GOB_KNOB(i) gob_knob_default(i)
#endif
__device__ __forceinline__ void go_knob_sleep(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1); }
template <bool WT>
__device__ __forceinline__ void go_publish(dx_gu64* p, float v, unsigned tag) {
  const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
  if constexpr (WT) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // sc1: write-through, any placement
  else __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                  // sc0: stays in this XCD's L2
}
template <int UPW, bool TAPE, bool WT, bool TRACE>
__device__ __forceinline__ void go_body(const GdArgs& a, float* gx_smem, int place, int slot, DxRt rt) {
  constexpr int NT = 512, H = GX_H, MB = DX_GROUP / UPW, UPM = 8 * UPW, NREG = 24 * UPW, LPU = 64 / UPW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int SLOT = 2 * GX_BLK * 3 * UPM;      // floats of one ring slot (both directions)
  float* xq = gx_smem;                            // ring first: its LDS addresses go through M0
  float* hs = xq + 2 * SLOT;                      // [2 dirs][H] states
  float* xs = hs + 2 * H;                         // [2 dirs][H] r * h
  // the workgroups of an XCD are dealt round-robin to its UPW clusters
  const int row = place * UPW + (slot % UPW), member = slot / UPW;
  if (row >= a.B || member >= MB) return;
  const int T = a.T;
#ifdef GO_KNOB_RT
  int kn[12];
  for (int q = 0; q < 12; ++q) kn[q] = __builtin_amdgcn_readfirstlane(g_go_knob[q]);
#endif
  const int L = __builtin_amdgcn_readfirstlane(a.lengths ? a.lengths[row] : T);
  const bool tracer = (TRACE || GO_DYN) && a.trace && row == 0 && member == 0 && tid == 0;

  float W[NREG];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const float* wp = a.wpack + (((size_t)d * MB + member) * (NREG / 2)) * NT + tid;
#pragma unroll
    for (int j = 0; j < NREG / 2; ++j) W[(NREG / 2) * d + j] = wp[(size_t)j * NT];
  }
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)row * 4 * H;       // [dir][r*h : H | h' : H]
  for (int i = tid; i < 2 * H; i += NT) {
    hs[i] = a.h0 ? a.h0[(size_t)row * 2 * H + i] : 0.f;
    xs[i] = 0.f;
  }
  // x-part blocks of both directions: item i = (dir, step j, gate g, quarter c) -> one float4 of the member's 8 UPW units
  constexpr int QPG = UPM / 4, NIT = 2 * GX_BLK * 3 * QPG, NLD = (NIT + NT - 1) / NT;
  static_assert(NIT % 64 == 0, "a wave's 64 items are all inside the block or all outside");
  const unsigned xq_lds = (unsigned)(size_t)(gx_lds_float*)xq;
  auto blk_fetch = [&](int s0, int ring) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      if (u * NT + wave * 64 < NIT) {            // wave-uniform
        const int i = u * NT + tid;
        const int c = i % QPG, g = (i / QPG) % 3, j = (i / (3 * QPG)) % GX_BLK, d = i / (3 * QPG * GX_BLK);
        const int sx = min(s0 + j, T - 1);
        const float* src = a.xproj + ((size_t)row * T + sx) * 6 * H + d * 3 * H + g * H + member * UPM + 4 * c;
        const unsigned dst = __builtin_amdgcn_readfirstlane(xq_lds + (unsigned)(ring * SLOT + 4 * (u * NT + wave * 64)) * 4u);
        gx_load_lds16(src, dst);
      }
    }
  };
  blk_fetch(0, 0);
  blk_fetch(GX_BLK, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // the lane's unit: i = lane / LPU of the wave's UPW; the first lane of the unit publishes and stores
  const int ui_outer = lane / LPU;
  float x0[2][3], hv[2], gv[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) { hv[d] = hs[d * H + member * UPM + wave * UPW + ui_outer]; gv[d] = 0.f; }
  // the granule of a gather in flight across the current compute phase, asked for twice: at the start of the phase (`pre`: a fast producer's value
  // is back long before it is needed) and again behind the phase's products (`pre2`: a request that reaches the L2 ~250 clocks later finds the value of
  // a producer that published late; it is back by the time the phase's reduction and epilogue are done).  256 threads x 8 bytes each: nothing.
  unsigned long long pre = 0ull, pre2 = 0ull;

  // (Round 6, measured: s_setprio 1 for waves 4-7 costs 38 us per C2 scan, for waves 0-3 it changes nothing; profiles/r06_ab_priority.txt.)
  const int tid_outer = tid, lane_outer = lane;
  for (int s = 0; s < T; ++s) {
    const unsigned tag = (unsigned)s + 1u;
    int tid = tid_outer, lane = lane_outer;                 // opaque per-iteration copies: see taco_decoder_xcd.h
    asm volatile("" : "+v"(tid), "+v"(lane));
    const int ui = lane / LPU, unit = member * UPM + wave * UPW + ui;
    const bool pub = (lane & (LPU - 1)) == 0;
    const bool active = s < L;                              // A.7: row active iff s < L; forward t = s, backward t = L-1-s
    GO_STAMP(0);
    const int sb = s & (GX_BLK - 1), ring = (s / GX_BLK) & 1;
    if (sb == 0 && s > 0) blk_fetch(s + GX_BLK, ring ^ 1);
    // gathers: direction D's vectors are collected by waves 4D .. 4D+3, one granule per thread
    auto request = [&](const dx_gu64* Xv, int D) {
      if ((wave >> 2) == D) pre = __hip_atomic_load(Xv + (tid & 255), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto request2 = [&](const dx_gu64* Xv, int D) {
      if ((wave >> 2) == D) pre2 = __hip_atomic_load(Xv + (tid & 255), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    int csite = 0;
    auto collect = [&](const dx_gu64* Xv, unsigned tg, float* dst, int D) {
      if ((wave >> 2) == D) {
        float v[1];
        if ((unsigned)(pre2 >> 32) == tg) v[0] = __uint_as_float((unsigned)pre2);
        else if ((unsigned)(pre >> 32) == tg) v[0] = __uint_as_float((unsigned)pre);
        else { go_knob_sleep(GO_KNOB(8 + csite)); dx_poll<1>(Xv + (tid & 255), 0, tg, v, rt); }       // a producer was late: the ordinary bounded poll
        dst[tid & 255] = v[0];
      }
    };
    // a phase: D = its direction; (Xn, Dn) = the gather in flight across it (null: none)
    auto gates = [&](auto Dc, const dx_gu64* Xn, int Dn) {
      constexpr int D = decltype(Dc)::value;
      constexpr int R0 = (NREG / 2) * D;
#pragma unroll
      for (int g = 0; g < 3; ++g) x0[D][g] = xq[(size_t)ring * SLOT + ((size_t)(D * GX_BLK + sb) * 3 + g) * UPM + wave * UPW + ui];
      // (r_i, u_i) of unit i in one v_pk_fma_f32 per input: registers R0 + 8i + e (r) and R0 + 8i + 4 + e (u), the state value broadcast
      const float4 hx = *reinterpret_cast<const float4*>(hs + D * H + 4 * lane);
      taco_f32x2 acc[UPW];
#pragma unroll
      for (int i = 0; i < UPW; ++i) acc[i] = (taco_f32x2){W[R0 + 8 * i] * hx.x, W[R0 + 8 * i + 4] * hx.x};
#pragma unroll
      for (int i = 0; i < UPW; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx.y, hx.y}, (taco_f32x2){W[R0 + 8 * i + 1], W[R0 + 8 * i + 5]}, acc[i]);
#pragma unroll
      for (int i = 0; i < UPW; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx.z, hx.z}, (taco_f32x2){W[R0 + 8 * i + 2], W[R0 + 8 * i + 6]}, acc[i]);
#pragma unroll
      for (int i = 0; i < UPW; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx.w, hx.w}, (taco_f32x2){W[R0 + 8 * i + 3], W[R0 + 8 * i + 7]}, acc[i]);
      if (GO_REQ2 && Xn && GO_KNOB(D) == 0) request2(Xn, Dn);
      else if (GO_KNOB(D) == 2) pre2 = 0ull;      // no second request in this phase: the previous phase's answer must not pass for this one (its tag can be the same)
      float v[2 * UPW], sm[2];
#pragma unroll
      for (int i = 0; i < UPW; ++i) { v[2 * i] = acc[i].x; v[2 * i + 1] = acc[i].y; }
      go_reduce<UPW, 2>(v, sm);
      if (GO_REQ2 && Xn && GO_KNOB(D) == 1) request2(Xn, Dn);
      const float rr = dx_sigmoid_fast(sm[0] + x0[D][0]);
      gv[D] = dx_sigmoid_fast(sm[1] + x0[D][1]);
      float rh = rr * hv[D];
      asm volatile("" : "+v"(pre), "+v"(pre2), "+v"(rh));      // the gather in flight has landed: wait for it HERE, ahead of the publish store (see gd_landed)
      if (pub) { if (GO_DYN) dx_publish(X + (size_t)D * 2 * H + unit, rh, tag, rt); else go_publish<WT>(X + (size_t)D * 2 * H + unit, rh, tag); }
      if (TAPE && pub && active) {      // gates of the active steps at their true time (modules.py:82-96 / A.7), for the backward scan -- BEHIND the publish: the exchange is the critical path
        float* gs = a.gsave + ((size_t)row * T + (D ? L - 1 - s : s)) * 6 * H + D * 3 * H + unit;
        gs[0] = rr; gs[H] = gv[D];
      }
    };
    auto cand = [&](auto Dc, const dx_gu64* Xn, int Dn) {
      constexpr int D = decltype(Dc)::value;
      constexpr int R0 = (NREG / 2) * D + 8 * UPW;
      const float4 hx = *reinterpret_cast<const float4*>(xs + D * H + 4 * lane);
      float v[UPW], sm[1];
      if constexpr (UPW >= 2) {        // candidates of units (i, i + 1) in one v_pk_fma_f32 per input
        taco_f32x2 acc[UPW / 2];
#pragma unroll
        for (int i = 0; i < UPW / 2; ++i) acc[i] = (taco_f32x2){W[R0 + 8 * i] * hx.x, W[R0 + 8 * i + 4] * hx.x};
#pragma unroll
        for (int i = 0; i < UPW / 2; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx.y, hx.y}, (taco_f32x2){W[R0 + 8 * i + 1], W[R0 + 8 * i + 5]}, acc[i]);
#pragma unroll
        for (int i = 0; i < UPW / 2; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx.z, hx.z}, (taco_f32x2){W[R0 + 8 * i + 2], W[R0 + 8 * i + 6]}, acc[i]);
#pragma unroll
        for (int i = 0; i < UPW / 2; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx.w, hx.w}, (taco_f32x2){W[R0 + 8 * i + 3], W[R0 + 8 * i + 7]}, acc[i]);
#pragma unroll
        for (int i = 0; i < UPW / 2; ++i) { v[2 * i] = acc[i].x; v[2 * i + 1] = acc[i].y; }
      } else {                         // one unit: even / odd inputs in the two halves
        taco_f32x2 acc = (taco_f32x2){W[R0] * hx.x, W[R0 + 1] * hx.y};
        acc = __builtin_elementwise_fma((taco_f32x2){hx.z, hx.w}, (taco_f32x2){W[R0 + 2], W[R0 + 3]}, acc);
        v[0] = acc.x + acc.y;
      }
      if (GO_REQ2 && Xn && GO_KNOB(2 + D) == 0) request2(Xn, Dn);
      else if (GO_KNOB(2 + D) == 2) pre2 = 0ull;
      go_reduce<UPW, 1>(v, sm);
      if (GO_REQ2 && Xn && GO_KNOB(2 + D) == 1) request2(Xn, Dn);
      const float cc = taco_tanh_fast(sm[0] + x0[D][2]);
      float blend = gv[D] * hv[D] + (1.f - gv[D]) * cc;
      DX_PIN(blend);
      float nv = active ? blend : hv[D];
      asm volatile("" : "+v"(pre), "+v"(pre2), "+v"(nv));
      if (pub) {
        if (GO_DYN) dx_publish(X + (size_t)(D * 2 + 1) * H + unit, nv, tag, rt); else go_publish<WT>(X + (size_t)(D * 2 + 1) * H + unit, nv, tag);
        const int t = (D && active) ? (L - 1 - s) : s;
        a.out[((size_t)row * T + t) * 2 * H + D * H + unit] = active ? nv : 0.f;
        if (TAPE && active) a.gsave[((size_t)row * T + t) * 6 * H + D * 3 * H + 2 * H + unit] = cc;
      }
      hv[D] = nv;
    };
    using F = std::integral_constant<int, 0>;
    using Bk = std::integral_constant<int, 1>;
    const dx_gu64* X_rhF = X;                  const dx_gu64* X_hF = X + (size_t)H;
    const dx_gu64* X_rhB = X + (size_t)2 * H;  const dx_gu64* X_hB = X + (size_t)3 * H;
    // (the load for h'(B) of the previous step was requested at the end of that step)
    gates(F{}, s > 0 ? X_hB : nullptr, 1);
    GO_STAMP(1);
    csite = 0;
    if (s > 0) collect(X_hB, tag - 1u, hs + H, 1);
    __syncthreads();
    GO_STAMP(2);
    go_knob_sleep(GO_KNOB(4 + 1));
    request(X_rhF, 0);
    gates(Bk{}, X_rhF, 0);
    GO_STAMP(3);
    csite = 1;
    collect(X_rhF, tag, xs, 0);
    __syncthreads();
    GO_STAMP(4);
    go_knob_sleep(GO_KNOB(4 + 2));
    request(X_rhB, 1);
    cand(F{}, X_rhB, 1);
    GO_STAMP(5);
    csite = 2;
    collect(X_rhB, tag, xs + H, 1);
    __syncthreads();
    GO_STAMP(6);
    go_knob_sleep(GO_KNOB(4 + 3));
    request(X_hF, 0);
    cand(Bk{}, X_hF, 0);
    GO_STAMP(7);
    csite = 3;
    collect(X_hF, tag, hs, 0);
    if (sb == GX_BLK - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's part of the next ring slot has landed
    __syncthreads();
    GO_STAMP(8);
    go_knob_sleep(GO_KNOB(4 + 0));
    request(X_hB, 1);
  }
}

template <int UPW, bool TAPE = false, bool TRACE = false>
__global__ __launch_bounds__(512) void k_bigru_oct(const GdArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float gx_smem[];
  GdArgs a = a_in;
  const int tid = threadIdx.x;
  int* ictl = reinterpret_cast<int*>(gx_smem + go_ring_floats(UPW) + 4 * GX_H);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, a.force_wt, ictl, tid, 24);
  const int place = __builtin_amdgcn_readfirstlane(ictl[0]), slot = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  if (__builtin_amdgcn_readfirstlane((int)rt.wt)) go_body<UPW, TAPE, true, TRACE>(a, gx_smem, place, slot, rt);
  else go_body<UPW, TAPE, false, TRACE>(a, gx_smem, place, slot, rt);
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_bigru_duo_bwd<RG>: the backward scan of the same BiGRU (BPTT through modules.py:82-96 / TF GRUCell, A.6/A.7) on k_bigru_duo's
// machinery: both directions of RG rows on one group of 32 CUs, the two directions software-pipelined against each other, polls
// issued early.  It replaces k_bigru_rows_bwd (one workgroup per (direction, row pair), the transposed recurrent kernels streamed
// from L2 every step: 4 us per step) for H = 256.
//
// Per direction, row and step s (descending; t = true time of the step, hp = the state the step started from):
//   g = dh + dout_t;  d c_pre = g (1-u)(1-c^2);  d u_pre = g (hp - c) u (1-u);  keep = g u            [owner of the unit, local]
//   d(r*h) = d c_pre . Wc_h^T            <- exchange 1: d c_pre of all units
//   d r_pre = d(r*h) hp r (1-r);  keep += d(r*h) r
//   dh(previous step) = keep + [d r_pre | d u_pre] . Wg_h^T     <- exchange 2: the 2H gate pre-activation gradients
// dh never leaves the lane that owns the unit.  Member m owns units 8m .. 8m+7 of both directions, wave w unit 8m + w; a lane keeps
// its 4-input slice of that unit's ROW of Wc_h (4 registers) and of Wg_h (8 registers: r half, u half) per direction.  Everything a
// step reads from global memory (dout, the gate tape r, u, c and the previous state) comes through an LDS ring fetched 16 steps at a
// time straight into LDS; it writes d r_pre, d u_pre, d c_pre (the gradient of the hoisted input projection) and r*hp (the input of
// the candidate kernel's recurrent rows, for its weight gradient) at the true time index.
// ------------------------------------------------------------------------------------------------------------------------------
#define GB_ARR 5             // ring arrays per (direction, step, row): dout, r, u, c, hp -- 8 units (two float4) each
__host__ __device__ inline size_t gb_ring_floats(int RG) { return (size_t)2 * 2 * GX_BLK * RG * GB_ARR * 8; }      // [slot][dir][step][row][array][8]
__host__ __device__ inline size_t gb_lds_floats(int RG) { return gb_ring_floats(RG) + (size_t)2 * RG * GX_H + (size_t)2 * RG * 2 * GX_H + 64; }
__host__ __device__ inline size_t gb_xbuf_granules(int RG) { return (size_t)DX_NGROUP * 2 * RG * 3 * GX_H; }     // groups x dir x [d c_pre : H | d gates : 2H] x RG

template <int RG, int W, int NT>
struct GbPoll {
  static constexpr int NI = (RG * W + NT - 1) / NT;
  unsigned long long g[NI];
};
template <int RG, int W, int NT>
__device__ __forceinline__ void gb_request(const dx_gu64* X, int tid, GbPoll<RG, W, NT>& p) {
  if ((RG * W >= NT) || tid < RG * W) {
#pragma unroll
    for (int u = 0; u < GbPoll<RG, W, NT>::NI; ++u) p.g[u] = __hip_atomic_load(X + tid + u * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <int RG, int W, int NT>
__device__ __forceinline__ void gb_landed(GbPoll<RG, W, NT>& p, float& after) {
#pragma unroll
  for (int u = 0; u < GbPoll<RG, W, NT>::NI; ++u) asm volatile("" : "+v"(p.g[u]), "+v"(after));
}
template <int RG, int W, int NT>
__device__ __forceinline__ void gb_collect(const dx_gu64* X, unsigned tag, float* st, int tid, GbPoll<RG, W, NT>& p, DxRt& rt) {
  constexpr int NI = GbPoll<RG, W, NT>::NI;
  if ((RG * W >= NT) || tid < RG * W) {
    bool ok = true;
#pragma unroll
    for (int u = 0; u < NI; ++u) ok = ok && ((unsigned)(p.g[u] >> 32) == tag);
    float v[NI];
    if (ok) {
#pragma unroll
      for (int u = 0; u < NI; ++u) v[u] = __uint_as_float((unsigned)p.g[u]);
    } else {
      dx_poll<NI>(X + tid, NT, tag, v, rt);
    }
#pragma unroll
    for (int u = 0; u < NI; ++u) st[u * NT + tid] = v[u];
  }
}

struct GbArgs {
  const float* wpack;                         // [2 dirs][32 members][12][512]: rows of Wc_h (4) and of Wg_h (4 r-half + 4 u-half) of the wave's unit
  const float* dout; const float* out; const float* gsave;     // [B*T, 2H], [B*T, 2H], [B*T, 6H]
  const float* h0;                            // [B, 2H] or null
  const int* lengths;                         // [B] or null
  float* dg; float* rh; float* dh0;           // [B*T, 6H], [B*T, 2H] (both pre-zeroed by the caller), [B, 2H] or null
  unsigned long long* xbuf; unsigned* ctl; unsigned* err;
  int B, T, force_wt;
};

template <int RG, bool WT>
__device__ __forceinline__ void gb_body(const GbArgs& a, float* gx_smem, int group, int member, DxRt rt) {
  constexpr int WTC = WT ? 1 : 0;
  constexpr int NT = 512, H = GX_H, RL = DxRL<RG>::value;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int SLOT = 2 * GX_BLK * RG * GB_ARR * 8;      // floats of one ring slot
  float* xq = gx_smem;                                    // ring first (M0-addressed)
  float* v1 = xq + 2 * SLOT;                              // [2 dirs][RG][H]   d c_pre
  float* v2 = v1 + 2 * RG * H;                            // [2 dirs][RG][2H]  d r_pre | d u_pre
  const int row0 = group * RG;
  if (row0 >= a.B || member >= GD_MEMBERS) return;
  const int T = a.T;

  float W[GD_NREG];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const float* wp = a.wpack + (((size_t)d * GD_MEMBERS + member) * 12) * NT + tid;
#pragma unroll
    for (int j = 0; j < 12; ++j) W[12 * d + j] = wp[(size_t)j * NT];
  }
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)group * 2 * RG * 3 * H;      // [dir][d c_pre : RG x H | d gates : RG x 2H]
  for (int i = tid; i < 2 * RG * H; i += NT) v1[i] = 0.f;
  for (int i = tid; i < 2 * RG * 2 * H; i += NT) v2[i] = 0.f;
  const bool epl = lane < (RG >= 4 ? 4 : RG);
  const int u0 = member * 8 + wave;
  int erow[RL], eL[RL];
  bool evalid[RL];
#pragma unroll
  for (int q = 0; q < RL; ++q) {
    erow[q] = dx_row<RG>(lane & 3, q);
    evalid[q] = epl && (row0 + erow[q] < a.B);
    eL[q] = evalid[q] ? (a.lengths ? a.lengths[row0 + erow[q]] : T) : 0;
  }
  // ring blocks: block k covers steps s = T-1-16k .. T-16-16k; item i = (dir, j, row, array, half) -> one float4 of the member's 8 units
  constexpr int NIT = 2 * GX_BLK * RG * GB_ARR * 2, NLD = (NIT + NT - 1) / NT;
  static_assert(NIT % 64 == 0, "a wave's 64 items are all inside the block or all outside");
  const unsigned xq_lds = (unsigned)(size_t)(gx_lds_float*)xq;
  auto blk_fetch = [&](int blk, int ring) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      if (u * NT + wave * 64 < NIT) {            // wave-uniform
        const int i = u * NT + tid;
        const int c2 = i & 1, ar = (i >> 1) % GB_ARR, r = (i / (2 * GB_ARR)) % RG, j = (i / (2 * GB_ARR * RG)) % GX_BLK, d = i / (2 * GB_ARR * RG * GX_BLK);
        const int b = min(row0 + r, a.B - 1);
        const int L = a.lengths ? a.lengths[b] : T;
        const int s = max(T - 1 - GX_BLK * blk - j, 0);
        const int sc = min(s, max(L - 1, 0));                                   // inactive steps: any valid row (the values are not used)
        const int t = d ? (L - 1 - sc) : sc, tp = min(max(d ? t + 1 : t - 1, 0), T - 1);
        const float* src;
        if (ar == 0) src = a.dout + ((size_t)b * T + t) * 2 * H + d * H;
        else if (ar == 4) src = a.out + ((size_t)b * T + tp) * 2 * H + d * H;
        else src = a.gsave + ((size_t)b * T + t) * 6 * H + d * 3 * H + (ar - 1) * H;
        const unsigned dst = __builtin_amdgcn_readfirstlane(xq_lds + (unsigned)(ring * SLOT + 4 * (u * NT + wave * 64)) * 4u);
        gx_load_lds16(src + member * 8 + 4 * c2, dst);
      }
    }
  };
  blk_fetch(0, 0);
  blk_fetch(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float keep[2][RL], hp[2][RL], rg[2][RL], dgu[2][RL];
  bool act[2][RL];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int q = 0; q < RL; ++q) { keep[d][q] = 0.f; hp[d][q] = 0.f; rg[d][q] = 0.f; dgu[d][q] = 0.f; act[d][q] = false; }
  GbPoll<RG, H, NT> pre1;            // a d c_pre gather in flight
  GbPoll<RG, 2 * H, NT> pre2;        // a d gates gather in flight
#pragma unroll
  for (int u = 0; u < GbPoll<RG, H, NT>::NI; ++u) pre1.g[u] = 0ull;
#pragma unroll
  for (int u = 0; u < GbPoll<RG, 2 * H, NT>::NI; ++u) pre2.g[u] = 0ull;

  const int tid_outer = tid, lane_outer = lane;
  // step index k = T-1-s counts up; tag of the exchanges of step k = k + 1
  for (int k = 0; k <= T; ++k) {                 // k == T: only the trailing dh of step s = 0 (the initial-state gradient)
    const int s = T - 1 - k;
    const unsigned tag = (unsigned)k + 1u;
    int tid = tid_outer, lane = lane_outer;
    asm volatile("" : "+v"(tid), "+v"(lane));
    const int sb = k & (GX_BLK - 1), ring = (k / GX_BLK) & 1;
    if (sb == 0 && k > 0 && k < T) blk_fetch(k / GX_BLK + 1, ring ^ 1);
    // phase CA(D): finish step s+1 (dh of the previous step from the gathered gate gradients), start step s
    auto phase_ca = [&](auto Dc) {
      constexpr int D = decltype(Dc)::value;
      float dhg[RL];
#pragma unroll
      for (int q = 0; q < RL; ++q) dhg[q] = 0.f;
      if (k > 0) {
        float acc[1][RG], sm[1][RL];
        dx_zero<1, RG>(acc);
        dx_pass<12 * D + 4, 1, RG, GD_NREG, 2 * GX_H>(W, v2 + D * RG * 2 * H, lane, acc);               // r half of the gate gradients
        dx_pass<12 * D + 8, 1, RG, GD_NREG, 2 * GX_H>(W, v2 + D * RG * 2 * H + H, lane, acc);           // u half
        dx_reduce<1, RG>(acc, sm, lane);
#pragma unroll
        for (int q = 0; q < RL; ++q) dhg[q] = sm[0][q];
      }
      float dcp[RL][1];
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const float dh = act[D][q] ? keep[D][q] + dhg[q] : keep[D][q];          // act = the previous step's
        dcp[q][0] = 0.f; dgu[D][q] = 0.f;
        keep[D][q] = dh;
        if (k < T) {
          const bool active = evalid[q] && s < eL[q];
          act[D][q] = active;
          const float* rq = xq + (size_t)ring * SLOT + (((size_t)(D * GX_BLK + sb) * RG + erow[q]) * GB_ARR) * 8 + wave;
          const float dout = rq[0], r = rq[8], u = rq[16], c = rq[24];
          float hprev = rq[32];
          const int L = eL[q], t = D ? (L - 1 - s) : s;
          if (active && s == 0) hprev = a.h0 ? a.h0[(size_t)(row0 + erow[q]) * 2 * H + D * H + u0] : 0.f;   // the step that started from the initial state
          if (active) {
            const float g = dh + dout;
            dcp[q][0] = g * (1.f - u) * (1.f - c * c);
            dgu[D][q] = g * (hprev - c) * u * (1.f - u);
            keep[D][q] = g * u;
            hp[D][q] = hprev; rg[D][q] = r;
            const size_t row = (size_t)(row0 + erow[q]) * T + t;
            a.rh[row * 2 * H + D * H + u0] = r * hprev;
            float* dq = a.dg + row * 6 * H + D * 3 * H + u0;
            dq[H] = dgu[D][q]; dq[2 * H] = dcp[q][0];
          }
        } else {
          act[D][q] = false;
          if (a.dh0 && evalid[q]) a.dh0[(size_t)(row0 + erow[q]) * 2 * H + D * H + u0] = dh;
        }
      }
      if (k < T) {
        gb_landed<RG, H, NT>(pre1, dcp[RL - 1][0]);        // whichever gather is in flight across this phase has landed by now:
        gb_landed<RG, 2 * H, NT>(pre2, dcp[RL - 1][0]);    // wait for it HERE, ahead of the publish stores (see gd_landed)
        if (epl) {
#pragma unroll
          for (int q = 0; q < RL; ++q) dx_publish_n<1, WTC>(X + (size_t)D * RG * 3 * H + erow[q] * H + u0, 1, dcp[q], tag, rt);
        }
      }
    };
    // phase B(D): d(r*h) from the gathered d c_pre -> d r_pre, publish both gate gradients
    auto phase_b = [&](auto Dc) {
      constexpr int D = decltype(Dc)::value;
      float acc[1][RG], sm[1][RL];
      dx_zero<1, RG>(acc);
      dx_pass<12 * D, 1, RG, GD_NREG, GX_H>(W, v1 + D * RG * H, lane, acc);
      dx_reduce<1, RG>(acc, sm, lane);
      float gr[RL][1], gu[RL][1];
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        float dgr = 0.f;
        if (act[D][q]) {
          const float drh = sm[0][q];
          dgr = drh * hp[D][q] * rg[D][q] * (1.f - rg[D][q]);
          keep[D][q] += drh * rg[D][q];
          const int L = eL[q], t = D ? (L - 1 - s) : s;
          a.dg[((size_t)(row0 + erow[q]) * T + t) * 6 * H + D * 3 * H + u0] = dgr;
        }
        gr[q][0] = dgr; gu[q][0] = dgu[D][q];
      }
      gb_landed<RG, H, NT>(pre1, gr[RL - 1][0]);
      gb_landed<RG, 2 * H, NT>(pre2, gr[RL - 1][0]);
      if (epl) {
#pragma unroll
        for (int q = 0; q < RL; ++q) {
          dx_publish_n<1, WTC>(X + (size_t)D * RG * 3 * H + RG * H + erow[q] * 2 * H + u0, 1, gr[q], tag, rt);
          dx_publish_n<1, WTC>(X + (size_t)D * RG * 3 * H + RG * H + erow[q] * 2 * H + H + u0, 1, gu[q], tag, rt);
        }
      }
    };
    using F = std::integral_constant<int, 0>;
    using Bk = std::integral_constant<int, 1>;
    const dx_gu64* Xc0 = X;                               const dx_gu64* Xg0 = X + (size_t)RG * H;
    const dx_gu64* Xc1 = X + (size_t)RG * 3 * H;          const dx_gu64* Xg1 = X + (size_t)RG * 3 * H + RG * H;
    // (the gate gradients of F of the previous step were collected at the end of the previous iteration; B's are in flight)
    phase_ca(F{});
    if (k > 0) gb_collect<RG, 2 * H, NT>(Xg1, tag - 1u, v2 + RG * 2 * H, tid, pre2, rt);
    __syncthreads();
    if (k == T) { phase_ca(Bk{}); break; }
    gb_request<RG, H, NT>(Xc0, tid, pre1);
    phase_ca(Bk{});
    gb_collect<RG, H, NT>(Xc0, tag, v1, tid, pre1, rt);
    __syncthreads();
    gb_request<RG, H, NT>(Xc1, tid, pre1);
    phase_b(F{});
    gb_collect<RG, H, NT>(Xc1, tag, v1 + RG * H, tid, pre1, rt);
    __syncthreads();
    gb_request<RG, 2 * H, NT>(Xg0, tid, pre2);
    phase_b(Bk{});
    gb_collect<RG, 2 * H, NT>(Xg0, tag, v2, tid, pre2, rt);
    if (sb == GX_BLK - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gb_request<RG, 2 * H, NT>(Xg1, tid, pre2);
  }
}

template <int RG>
__global__ __launch_bounds__(512) void k_bigru_duo_bwd(const GbArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float gx_smem[];
  GbArgs a = a_in;
  int* ictl = reinterpret_cast<int*>(gx_smem + gb_lds_floats(RG) - 64);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, a.force_wt, ictl, threadIdx.x, 24);
  const int group = __builtin_amdgcn_readfirstlane(ictl[0]), member = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  if (__builtin_amdgcn_readfirstlane((int)rt.wt)) gb_body<RG, true>(a, gx_smem, group, member, rt);
  else gb_body<RG, false>(a, gx_smem, group, member, rt);
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_bigru_oct_bwd: the backward scan on k_bigru_oct's geometry (round 5) -- ONE row per cluster of 8 CUs (four clusters per XCD),
// a member owns 32 units of both directions, wave w the units 32 m + 4 w + i.  The recurrences, the exchanges (d c_pre, then the 2H
// gate gradients, per direction and step), the tape ring and the outputs are k_bigru_duo_bwd's (above); what changes is what made
// the forward scan faster: a gather is 256 / 512 granules from 7 peers with one granule per thread, a pass is one or two
// ds_read_b128 per lane, the four units of a wave are dealt to the lanes by permlane swaps (one epilogue chain per lane), the
// products are packed (v_pk_fma_f32: two units of one input vector in phase B; the r and u halves of one unit in phase CA), every
// gather is requested twice (at the start of the phase it is in flight across and again behind the phase's products), and length
// masking is a scalar test.  Weight pack: [2 dirs][8 members][48][512] -- registers 8 p + 4 j + e: row of Wc_h of unit 2 p + j
// (p = 0, 1), 16 + 8 i + e / 16 + 8 i + 4 + e: rows of Wg_h (r half / u half) of unit i, inputs 4 lane + e.
// ------------------------------------------------------------------------------------------------------------------------------
#define GOB_UPW 4
#define GOB_UPM 32
__host__ __device__ inline size_t gob_ring_floats() { return (size_t)2 * 2 * GX_BLK * GB_ARR * GOB_UPM; }      // [slot][dir][step][array][32 units]
__host__ __device__ inline size_t gob_lds_floats() { return gob_ring_floats() + (size_t)2 * GX_H + (size_t)2 * 2 * GX_H + 64; }

template <bool WT>
__device__ __forceinline__ void gob_body(const GbArgs& a, float* gx_smem, int place, int slot, DxRt rt) {
  constexpr int NT = 512, H = GX_H, UPW = GOB_UPW, MB = DX_GROUP / UPW, UPM = GOB_UPM, NREG = 96, LPU = 64 / UPW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int SLOT = 2 * GX_BLK * GB_ARR * UPM;         // floats of one ring slot (both directions)
  float* xq = gx_smem;                                    // ring first (M0-addressed)
  float* v1 = xq + 2 * SLOT;                              // [2 dirs][H]   d c_pre
  float* v2 = v1 + 2 * H;                                 // [2 dirs][r half : H | u half : H]  gate gradients
  const int row = place * UPW + (slot % UPW), member = slot / UPW;
  if (row >= a.B || member >= MB) return;
  const int T = a.T;
#ifdef GO_KNOB_RT
  int kb[4];
  for (int q = 0; q < 4; ++q) kb[q] = __builtin_amdgcn_readfirstlane(g_go_knob[12 + q]);
#endif
  const int L = __builtin_amdgcn_readfirstlane(a.lengths ? a.lengths[row] : T);

  float W[NREG];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const float* wp = a.wpack + (((size_t)d * MB + member) * (NREG / 2)) * NT + tid;
#pragma unroll
    for (int j = 0; j < NREG / 2; ++j) W[(NREG / 2) * d + j] = wp[(size_t)j * NT];
  }
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)row * 2 * 3 * H;      // [dir][d c_pre : H | d r_pre : H | d u_pre : H]
  for (int i = tid; i < 2 * H; i += NT) v1[i] = 0.f;
  for (int i = tid; i < 4 * H; i += NT) v2[i] = 0.f;
  // ring blocks: block b covers steps s = T-1-16b .. T-16-16b; item i = (dir, j, array, c) -> one float4 of the member's 32 units
  constexpr int QPA = UPM / 4, NIT = 2 * GX_BLK * GB_ARR * QPA, NLD = (NIT + NT - 1) / NT;
  static_assert(NIT % 64 == 0, "a wave's 64 items are all inside the block or all outside");
  const unsigned xq_lds = (unsigned)(size_t)(gx_lds_float*)xq;
  auto blk_fetch = [&](int blk, int ring) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      if (u * NT + wave * 64 < NIT) {            // wave-uniform
        const int i = u * NT + tid;
        const int c = i % QPA, ar = (i / QPA) % GB_ARR, j = (i / (QPA * GB_ARR)) % GX_BLK, d = i / (QPA * GB_ARR * GX_BLK);
        const int s = max(T - 1 - GX_BLK * blk - j, 0);
        const int sc = min(s, max(L - 1, 0));                                   // inactive steps: any valid row (the values are not used)
        const int t = d ? (L - 1 - sc) : sc, tp = min(max(d ? t + 1 : t - 1, 0), T - 1);
        const float* src;
        if (ar == 0) src = a.dout + ((size_t)row * T + t) * 2 * H + d * H;
        else if (ar == 4) src = a.out + ((size_t)row * T + tp) * 2 * H + d * H;
        else src = a.gsave + ((size_t)row * T + t) * 6 * H + d * 3 * H + (ar - 1) * H;
        const unsigned dst = __builtin_amdgcn_readfirstlane(xq_lds + (unsigned)(ring * SLOT + 4 * (u * NT + wave * 64)) * 4u);
        gx_load_lds16(src + member * UPM + 4 * c, dst);
      }
    }
  };
  blk_fetch(0, 0);
  blk_fetch(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float keep[2] = {0.f, 0.f}, hpv[2] = {0.f, 0.f}, rgv[2] = {0.f, 0.f}, dguv[2] = {0.f, 0.f};
  bool actd[2] = {false, false};
  unsigned long long pre = 0ull, pre2 = 0ull;      // the granule of the gather in flight across the current phase, asked for twice (see go_body)

  const int tid_outer = tid, lane_outer = lane;
  for (int k = 0; k <= T; ++k) {                 // k == T: only the trailing dh of step s = 0 (the initial-state gradient)
    const int s = T - 1 - k;
    const unsigned tag = (unsigned)k + 1u;
    int tid = tid_outer, lane = lane_outer;
    asm volatile("" : "+v"(tid), "+v"(lane));
    const int ui = lane / LPU, unit = member * UPM + wave * UPW + ui;
    const bool pub = (lane & (LPU - 1)) == 0;
    const bool active = k < T && s < L;
    const int sb = k & (GX_BLK - 1), ring = (k / GX_BLK) & 1;
    if (sb == 0 && k > 0 && k < T) blk_fetch(k / GX_BLK + 1, ring ^ 1);
    // gathers: n granules (256: by the four waves 4 D .. 4 D + 3; 512: by all), one per thread
    auto mine = [&](int n, int D) { return n == 512 || (wave >> 2) == D; };
    auto request = [&](const dx_gu64* Xv, int n, int D) {
      if (mine(n, D)) pre = __hip_atomic_load(Xv + (tid & (n - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto request2 = [&](const dx_gu64* Xv, int n, int D) {
      if (mine(n, D)) pre2 = __hip_atomic_load(Xv + (tid & (n - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto collect = [&](const dx_gu64* Xv, unsigned tg, float* dst, int n, int D) {
      if (mine(n, D)) {
        float v[1];
        if ((unsigned)(pre2 >> 32) == tg) v[0] = __uint_as_float((unsigned)pre2);
        else if ((unsigned)(pre >> 32) == tg) v[0] = __uint_as_float((unsigned)pre);
        else dx_poll<1>(Xv + (tid & (n - 1)), 0, tg, v, rt);       // a producer was late: the ordinary bounded poll
        dst[tid & (n - 1)] = v[0];
      }
    };
    // phase CA(D): finish step s+1 (dh of the previous step from the gathered gate gradients), start step s; (Xn, nn, Dn): the gather in flight
    auto phase_ca = [&](auto Dc, const dx_gu64* Xn, int nn, int Dn) {
      constexpr int D = decltype(Dc)::value;
      constexpr int R0 = (NREG / 2) * D + 16;
      float dhg = 0.f;
      if (k > 0) {
        const float4 gr = *reinterpret_cast<const float4*>(v2 + D * 2 * H + 4 * lane);
        const float4 gu = *reinterpret_cast<const float4*>(v2 + D * 2 * H + H + 4 * lane);
        taco_f32x2 acc[UPW];
#pragma unroll
        for (int i = 0; i < UPW; ++i) acc[i] = (taco_f32x2){W[R0 + 8 * i] * gr.x, W[R0 + 8 * i + 4] * gu.x};
#pragma unroll
        for (int i = 0; i < UPW; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){gr.y, gu.y}, (taco_f32x2){W[R0 + 8 * i + 1], W[R0 + 8 * i + 5]}, acc[i]);
#pragma unroll
        for (int i = 0; i < UPW; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){gr.z, gu.z}, (taco_f32x2){W[R0 + 8 * i + 2], W[R0 + 8 * i + 6]}, acc[i]);
#pragma unroll
        for (int i = 0; i < UPW; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){gr.w, gu.w}, (taco_f32x2){W[R0 + 8 * i + 3], W[R0 + 8 * i + 7]}, acc[i]);
        if (Xn && GOB_KNOB(D) == 0) request2(Xn, nn, Dn);
        else pre2 = 0ull;      // no second request: a stale answer of the previous phase (the same step tag) must not pass for this gather
        float v[UPW], sm[1];
#pragma unroll
        for (int i = 0; i < UPW; ++i) v[i] = acc[i].x + acc[i].y;
        go_reduce<UPW, 1>(v, sm);
        dhg = sm[0];
      }
      const float dh = actd[D] ? keep[D] + dhg : keep[D];          // actd = the previous step's
      keep[D] = dh; dguv[D] = 0.f;
      float dcp = 0.f;
      if (k < T) {
        actd[D] = active;
        const float* rq = xq + (size_t)ring * SLOT + ((size_t)(D * GX_BLK + sb) * GB_ARR) * UPM + wave * UPW + ui;
        const float dout = rq[0], r = rq[UPM], u = rq[2 * UPM], c = rq[3 * UPM];
        float hprev = rq[4 * UPM];
        if (active && s == 0) hprev = a.h0 ? a.h0[(size_t)row * 2 * H + D * H + unit] : 0.f;   // the step that started from the initial state
        float rhp = 0.f;
        if (active) {
          const float g = dh + dout;
          dcp = g * (1.f - u) * (1.f - c * c);
          dguv[D] = g * (hprev - c) * u * (1.f - u);
          keep[D] = g * u;
          hpv[D] = hprev; rgv[D] = r; rhp = r * hprev;
        }
        asm volatile("" : "+v"(pre), "+v"(pre2), "+v"(dcp));      // the gather in flight has landed: wait for it HERE, ahead of the publish store (see gd_landed)
        if (pub) {
          go_publish<WT>(X + (size_t)D * 3 * H + unit, dcp, tag);
          if (active) {                                            // (behind the publish: the exchange is the critical path)
            const size_t rowt = (size_t)row * T + (D ? L - 1 - s : s);
            a.rh[rowt * 2 * H + D * H + unit] = rhp;
            float* dq = a.dg + rowt * 6 * H + D * 3 * H + unit;
            dq[H] = dguv[D]; dq[2 * H] = dcp;
          }
        }
      } else {
        actd[D] = false;
        if (a.dh0 && pub) a.dh0[(size_t)row * 2 * H + D * H + unit] = dh;
      }
    };
    // phase B(D): d(r*h) from the gathered d c_pre -> d r_pre, publish both gate gradients
    auto phase_b = [&](auto Dc, const dx_gu64* Xn, int nn, int Dn) {
      constexpr int D = decltype(Dc)::value;
      constexpr int R0 = (NREG / 2) * D;
      const float4 hx = *reinterpret_cast<const float4*>(v1 + D * H + 4 * lane);
      taco_f32x2 acc[UPW / 2];
#pragma unroll
      for (int i = 0; i < UPW / 2; ++i) acc[i] = (taco_f32x2){W[R0 + 8 * i] * hx.x, W[R0 + 8 * i + 4] * hx.x};
#pragma unroll
      for (int i = 0; i < UPW / 2; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx.y, hx.y}, (taco_f32x2){W[R0 + 8 * i + 1], W[R0 + 8 * i + 5]}, acc[i]);
#pragma unroll
      for (int i = 0; i < UPW / 2; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx.z, hx.z}, (taco_f32x2){W[R0 + 8 * i + 2], W[R0 + 8 * i + 6]}, acc[i]);
#pragma unroll
      for (int i = 0; i < UPW / 2; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx.w, hx.w}, (taco_f32x2){W[R0 + 8 * i + 3], W[R0 + 8 * i + 7]}, acc[i]);
      if (Xn && GOB_KNOB(2 + D) == 0) request2(Xn, nn, Dn);
      else pre2 = 0ull;
      float v[UPW], sm[1];
#pragma unroll
      for (int i = 0; i < UPW / 2; ++i) { v[2 * i] = acc[i].x; v[2 * i + 1] = acc[i].y; }
      go_reduce<UPW, 1>(v, sm);
      float dgr = 0.f;
      if (actd[D]) {
        const float drh = sm[0];
        dgr = drh * hpv[D] * rgv[D] * (1.f - rgv[D]);
        keep[D] += drh * rgv[D];
      }
      float gu = dguv[D];
      asm volatile("" : "+v"(pre), "+v"(pre2), "+v"(dgr), "+v"(gu));
      if (pub) {
        go_publish<WT>(X + (size_t)D * 3 * H + H + unit, dgr, tag);
        go_publish<WT>(X + (size_t)D * 3 * H + 2 * H + unit, gu, tag);
        if (actd[D]) a.dg[((size_t)row * T + (D ? L - 1 - s : s)) * 6 * H + D * 3 * H + unit] = dgr;
      }
    };
    using F = std::integral_constant<int, 0>;
    using Bk = std::integral_constant<int, 1>;
    const dx_gu64* Xc0 = X;                          const dx_gu64* Xg0 = X + (size_t)H;
    const dx_gu64* Xc1 = X + (size_t)3 * H;          const dx_gu64* Xg1 = X + (size_t)3 * H + H;
    // (the gate gradients of F of the previous step were collected at the end of the previous iteration; B's are in flight)
    phase_ca(F{}, k > 0 ? Xg1 : nullptr, 512, 1);
    if (k > 0) collect(Xg1, tag - 1u, v2 + 2 * H, 512, 1);
    __syncthreads();
    if (k == T) { phase_ca(Bk{}, nullptr, 512, 1); break; }
    request(Xc0, 256, 0);
    phase_ca(Bk{}, Xc0, 256, 0);
    collect(Xc0, tag, v1, 256, 0);
    __syncthreads();
    request(Xc1, 256, 1);
    phase_b(F{}, Xc1, 256, 1);
    collect(Xc1, tag, v1 + H, 256, 1);
    __syncthreads();
    request(Xg0, 512, 0);
    phase_b(Bk{}, Xg0, 512, 0);
    collect(Xg0, tag, v2, 512, 0);
    if (sb == GX_BLK - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    request(Xg1, 512, 1);
  }
}

__global__ __launch_bounds__(512) void k_bigru_oct_bwd(const GbArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float gx_smem[];
  GbArgs a = a_in;
  int* ictl = reinterpret_cast<int*>(gx_smem + gob_lds_floats() - 64);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, a.force_wt, ictl, threadIdx.x, 24);
  const int place = __builtin_amdgcn_readfirstlane(ictl[0]), slot = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  if (__builtin_amdgcn_readfirstlane((int)rt.wt)) gob_body<true>(a, gx_smem, place, slot, rt);
  else gob_body<false>(a, gx_smem, place, slot, rt);
}
