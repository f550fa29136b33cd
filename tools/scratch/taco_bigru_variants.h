// tools/scratch/taco_bigru_variants.h -- NOT part of the library.  Two variants of the post-net scan that were built in round 5 on top of
// k_bigru_oct (csrc/taco_bigru_xcd.h), passed tests/test_gpu_decoder_xcd.py::test_post_net_scan_spread_over_the_chip, were measured
// (profiles/r05_scan_variants.txt) and lost to k_bigru_oct; kept as the record of what was tried:
//   k_bigru_dir  directions on different waves, no workgroup barrier, operands polled straight into registers: 852 us per C2 scan (oct: 833)
//   k_bigru_ks   candidate product split by ROWS, one exchange per direction and step: 1012 us (the 2304-granule collects are bound by the
//                CU's read port to the L2)
// To rebuild either: paste it back behind k_bigru_oct, add the host pack / launch of the commit that removed it (git log -- this file).
// ------------------------------------------------------------------------------------------------------------------------------
// k_bigru_dir<CPX, TAPE, TRACE> (round 5): k_bigru_oct's clusters (ONE row per 32 / CPX CUs, CPX = 4 / 2 / 1 clusters per XCD) with the two
// directions on DIFFERENT waves and no workgroup barrier, no LDS state at all.
//
// k_bigru_oct's step is issue bound: all eight waves run the same phase (two per SIMD, ~55 VALU instructions each), then all of them sit in a
// collect (LDS write, s_barrier, LDS read: ~250 clocks with the SIMDs idle), four times per step.  Here waves 0-3 own the FORWARD direction of
// the member's 8 CPX units (2 CPX units per wave, both gates and the candidate: the same 24 CPX weight registers per thread) and waves 4-7 the
// BACKWARD direction; every SIMD hosts one wave of each, so one direction's phase issues while the other waits for its exchange.  A wave needs
// of an exchanged vector exactly the four values its lanes multiply -- granules 4 lane .. 4 lane + 3 -- so it polls THOSE straight into
// registers (two 16-byte requests per lane): the poll is the operand fetch; nothing is staged through LDS, nothing is waited for on behalf of
// another wave.  Traffic: 4 waves x 2 KB per vector and CU instead of 2 KB -- 32 KB per CU and step, a sixth of what the L2 delivers.
// The x-parts ring is wave-private (global_load_lds, two slots of 16 steps: program order is all the synchronisation it needs).
// ------------------------------------------------------------------------------------------------------------------------------
#ifndef GV_DELAY
#define GV_DELAY 4           // s_sleep units (64 clocks) between a wave's publish and its first poll of the vector that publish belongs to
#endif
__host__ __device__ inline size_t gv_lds_floats() { return (size_t)8 * 2 * 512 + 64; }      // [wave][slot][128 items x 4 floats] + census words

// UW units per wave, NG values per unit: unit i's totals end up on lanes i * 64 / UW ... (UW = 8: halving on the two swap levels and on lane bit 3)
template <int UW, int NG>
__device__ __forceinline__ void gv_reduce(const float (&v)[UW * NG], float (&out)[NG], int lane) {
  if constexpr (UW <= 4) { go_reduce<UW, NG>(v, out); return; }
  else {
    float h[4 * NG], t[2 * NG], o[NG];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < NG; ++g) {      // lanes 0-31: units 0-3; lanes 32-63: units 4-7
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j * NG + g]), __float_as_uint(v[(j + 4) * NG + g]), false, false);
        h[j * NG + g] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < NG; ++g) {      // even rows of 16 lanes: units j | j + 4, odd rows: units j + 2 | j + 6 -> row r holds units 2r, 2r + 1
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[j * NG + g]), __float_as_uint(h[(j + 2) * NG + g]), false, false);
        t[j * NG + g] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
    const bool hi8 = (lane & 8) != 0;      // lanes 0-7 of a row keep unit 2r, lanes 8-15 unit 2r + 1
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float keep = hi8 ? t[NG + g] : t[g], send = hi8 ? t[g] : t[NG + g];
      o[g] = keep + DX_DPP0(send, 0x128);      // row_ror:8
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) o[g] += DX_DPP0(o[g], 0xB1);
#pragma unroll
    for (int g = 0; g < NG; ++g) o[g] += DX_DPP0(o[g], 0x4E);
#pragma unroll
    for (int g = 0; g < NG; ++g) out[g] = o[g] + DX_DPP0(o[g], 0x141);      // row_half_mirror: the other quad of the eight
  }
}
// the four granules 4 lane .. 4 lane + 3 of a 256-granule vector, polled until all carry `tag` (two 16-byte L1-bypassing requests; bounded)
__device__ __forceinline__ void gv_poll4(const dx_gu64* X, int lane, unsigned tag, float (&v)[4], DxRt& rt) {
  dx_u64x2 g[2];
  unsigned spins = 0;
  for (;;) {
    const dx_gu64* p = X + 4 * lane;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]) : "v"(p) : "memory");
    const bool ok = ((unsigned)(g[0][0] >> 32) == tag) && ((unsigned)(g[0][1] >> 32) == tag) && ((unsigned)(g[1][0] >> 32) == tag) && ((unsigned)(g[1][1] >> 32) == tag);
    if (ok || rt.dead) break;
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0) {
      if (spins >= DX_SPIN_LIMIT || __hip_atomic_load(rt.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(rt.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rt.dead = true;
      }
    }
  }
  v[0] = __uint_as_float((unsigned)g[0][0]); v[1] = __uint_as_float((unsigned)g[0][1]); v[2] = __uint_as_float((unsigned)g[1][0]); v[3] = __uint_as_float((unsigned)g[1][1]);
}
#define GV_STAMP(slot)                                                                                            \
  do {                                                                                                            \
    if constexpr (TRACE) { if (tracer && s >= 8 && s < 8 + DX_TRACE_STEPS) a.trace[(s - 8) * DX_TRACE_SLOTS + (slot)] = (long long)__builtin_readcyclecounter(); } \
  } while (0)

template <int CPX, bool TAPE, bool WT, bool TRACE>
__device__ __forceinline__ void gv_body(const GdArgs& a, float* gx_smem, int place, int slot, DxRt rt) {
  constexpr int H = GX_H, MB = DX_GROUP / CPX, UPM = 8 * CPX, UW = 2 * CPX, NREG = 12 * UW, LPU = 64 / UW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = wave >> 2, wq = wave & 3;                                    // the wave's direction, its quarter of the member's units
  const int row = place * CPX + (slot % CPX), member = slot / CPX;
  if (row >= a.B || member >= MB) return;
  const int T = a.T;
  const int L = __builtin_amdgcn_readfirstlane(a.lengths ? a.lengths[row] : T);
  const bool tracer = TRACE && a.trace && row == 0 && member == 0 && tid == 0;
  // weights: [dir][member][wq][NREG][64 lanes]: registers 8i + e (r_i), 8i + 4 + e (u_i), then 8 UW + 4i + e (c_i) of units wq UW + i
  float W[NREG];
  {
    const float* wp = a.wpack + ((((size_t)D * MB + member) * 4 + wq) * NREG) * 64 + lane;
#pragma unroll
    for (int j = 0; j < NREG; ++j) W[j] = wp[(size_t)j * 64];
  }
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)row * 4 * H + (size_t)D * 2 * H;      // this direction's [r*h : H | h' : H]
  const int ui = lane / LPU, unit = member * UPM + wq * UW + ui;
  const bool pub = (lane & (LPU - 1)) == 0;
  // wave-private ring of x-parts: item i = (step j, gate g, quarter c) -> one float4 of the wave's units; slot = 128 items
  constexpr int QW = UW >= 4 ? UW / 4 : 1, NIT = GX_BLK * 3 * QW, NLD = (NIT + 63) / 64;
  constexpr int U4 = UW >= 4 ? 0 : 1;                                            // UW = 2: the float4 that holds the wave's two units starts at an even pair
  float* xq = gx_smem + (size_t)wave * 2 * 512;
  const unsigned xq_lds = (unsigned)(size_t)(gx_lds_float*)xq;
  const int ubase = member * UPM + (U4 ? ((wq * UW) & ~3) : wq * UW), uoff = U4 ? ((wq * UW) & 3) : 0;
  auto blk_fetch = [&](int s0, int ring) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = min(u * 64 + lane, NIT - 1);
      const int c = i % QW, g = (i / QW) % 3, j = i / (3 * QW);
      const int sx = min(s0 + j, T - 1);
      const float* src = a.xproj + ((size_t)row * T + sx) * 6 * H + D * 3 * H + g * H + ubase + 4 * c;
      gx_load_lds16(src, __builtin_amdgcn_readfirstlane(xq_lds + (unsigned)(ring * 512 + u * 256) * 4u));
    }
  };
  blk_fetch(0, 0);
  blk_fetch(GX_BLK, 1);
  float hx[4], hv;
  {
    const float* h0 = a.h0 ? a.h0 + (size_t)row * 2 * H + D * H : nullptr;
#pragma unroll
    for (int e = 0; e < 4; ++e) hx[e] = h0 ? h0[4 * lane + e] : 0.f;
    hv = h0 ? h0[unit] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int lane_outer = lane;
  for (int s = 0; s < T; ++s) {
    const unsigned tag = (unsigned)s + 1u;
    int lane = lane_outer;                                     // opaque per-iteration copy: see taco_decoder_xcd.h
    asm volatile("" : "+v"(lane));
    const bool active = s < L;                                 // A.7: row active iff s < L; forward t = s, backward t = L-1-s
    GV_STAMP(0);
    const int sb = s & (GX_BLK - 1), ring = (s / GX_BLK) & 1;
    if (sb == 0 && s > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the slot entered now was requested 16 steps ago
      blk_fetch(s + GX_BLK, ring ^ 1);                         // ... and the one just left is free (program order: this wave was its only reader)
    }
    float x0[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) x0[g] = xq[ring * 512 + ((sb * 3 + g) * QW) * 4 + uoff + (lane_outer / LPU)];
    // ---- gates: (r_i, u_i) of unit i in one v_pk_fma_f32 per input ----
    float rr, uu;
    {
      taco_f32x2 acc[UW];
#pragma unroll
      for (int i = 0; i < UW; ++i) acc[i] = (taco_f32x2){W[8 * i] * hx[0], W[8 * i + 4] * hx[0]};
#pragma unroll
      for (int e = 1; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < UW; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx[e], hx[e]}, (taco_f32x2){W[8 * i + e], W[8 * i + 4 + e]}, acc[i]);
      float v[2 * UW], sm[2];
#pragma unroll
      for (int i = 0; i < UW; ++i) { v[2 * i] = acc[i].x; v[2 * i + 1] = acc[i].y; }
      gv_reduce<UW, 2>(v, sm, lane);
      rr = dx_sigmoid_fast(sm[0] + x0[0]);
      uu = dx_sigmoid_fast(sm[1] + x0[1]);
    }
    const float rh = rr * hv;
    if (pub) dx_publish<WT ? 1 : 0>(X + unit, rh, tag, rt);
    if (TAPE && pub && active) {      // gates of the active steps at their true time (modules.py:82-96 / A.7), for the backward scan
      float* gs = a.gsave + ((size_t)row * T + (D ? L - 1 - s : s)) * 6 * H + D * 3 * H + unit;
      gs[0] = rr; gs[H] = uu;
    }
    GV_STAMP(1);
    __builtin_amdgcn_s_sleep(GV_DELAY);
    float xr[4];
    gv_poll4(X, lane, tag, xr, rt);
    GV_STAMP(2);
    // ---- candidate and the new state: units (i, i + 1) in one v_pk_fma_f32 per input ----
    float nv;
    {
      taco_f32x2 acc[UW / 2];
#pragma unroll
      for (int i = 0; i < UW / 2; ++i) acc[i] = (taco_f32x2){W[8 * UW + 8 * i] * xr[0], W[8 * UW + 8 * i + 4] * xr[0]};
#pragma unroll
      for (int e = 1; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < UW / 2; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){xr[e], xr[e]}, (taco_f32x2){W[8 * UW + 8 * i + e], W[8 * UW + 8 * i + 4 + e]}, acc[i]);
      float v[UW], sm[1];
#pragma unroll
      for (int i = 0; i < UW / 2; ++i) { v[2 * i] = acc[i].x; v[2 * i + 1] = acc[i].y; }
      gv_reduce<UW, 1>(v, sm, lane);
      const float cc = taco_tanh_fast(sm[0] + x0[2]);
      float blend = uu * hv + (1.f - uu) * cc;
      DX_PIN(blend);
      nv = active ? blend : hv;
      if (pub) {
        dx_publish<WT ? 1 : 0>(X + H + unit, nv, tag, rt);
        const int t = (D && active) ? (L - 1 - s) : s;
        a.out[((size_t)row * T + t) * 2 * H + D * H + unit] = active ? nv : 0.f;
        if (TAPE && active) a.gsave[((size_t)row * T + t) * 6 * H + D * 3 * H + 2 * H + unit] = cc;
      }
    }
    hv = nv;
    GV_STAMP(3);
    if (s + 1 < T) {
      __builtin_amdgcn_s_sleep(GV_DELAY);
      gv_poll4(X + H, lane, tag, hx, rt);
    }
    GV_STAMP(4);
  }
}

template <int CPX, bool TAPE = false, bool TRACE = false>
__global__ __launch_bounds__(512) void k_bigru_dir(const GdArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float gx_smem[];
  GdArgs a = a_in;
  int* ictl = reinterpret_cast<int*>(gx_smem + 8 * 2 * 512);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, a.force_wt, ictl, threadIdx.x, 24);
  const int place = __builtin_amdgcn_readfirstlane(ictl[0]), slot = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  if (__builtin_amdgcn_readfirstlane((int)rt.wt)) gv_body<CPX, TAPE, true, TRACE>(a, gx_smem, place, slot, rt);
  else gv_body<CPX, TAPE, false, TRACE>(a, gx_smem, place, slot, rt);
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_bigru_ks<TAPE, TRACE> (round 5): ONE exchange per direction and step instead of two.
//
// k_bigru_oct sits on the bound of its decomposition: per direction a step is gates -> exchange r*h -> candidate -> exchange h', and an
// exchange (publish -> the L2 -> every consumer has seen the last producer's granule) is ~900 clocks however little is computed around it.
// The second exchange exists because the candidate's product (r*h) . Wc_h is split by COLUMNS: every member needs the whole r*h.  Split by
// ROWS it needs none: a member knows r_k h_k of its own 32 units k the moment its gates are done, so it forms, for ALL 256 candidate columns
// j, the partial sum over its own k -- the rows of Wc_h of its units: 32 x 256 weights, as many as the 256 x 32 column block they replace --
// and publishes the 256 partials next to the update gates u of its units.  Every member then collects the 8 partial vectors and the u vector
// of its cluster (2304 granules: nine per thread of one half of the workgroup) and finishes ALL 256 units itself, redundantly and bit-
// identically (c_j = tanh(xc_j + the partials in member order), h'_j = u_j h_j + (1 - u_j) c_j): it holds the whole new state without a
// second exchange.  A step of a direction is
//     G  gates of the own units (as k_bigru_oct)          -> r*h of the own units to LDS, u published
//     P  partial candidate sums for all columns            -> published          [exchange, hidden behind the other direction's E + G + P]
//     E  every unit's candidate and new state              -> the state vector in LDS; the owner of a unit stores the output
// with the two directions half a step apart.  The granules of a step go to buffer (step & 1): a member can overwrite a buffer only after
// every member has published into the other one, i.e. after every member has read this one.  One row per cluster of 8 CUs (17 to 32 rows).
// ------------------------------------------------------------------------------------------------------------------------------
#define GK_MB 8                       // members per cluster
#define GK_UPM 32                     // units per member
#define GK_BLK 8                      // steps per block of prefetched x-parts (both rings together stay below 48 KB: LDS-direct loads address through M0)
__host__ __device__ inline size_t gk_xbuf_granules() { return (size_t)32 * 2 * 2 * 9 * GX_H; }      // rows x dir x buffer x (8 partial vectors + u) x H
__host__ __device__ inline size_t gk_lds_floats() {
  return (size_t)2 * 2 * GK_BLK * 2 * GK_UPM      // x-parts of the own units' gates [slot][dir][step][r | u][32]
       + (size_t)2 * 2 * GK_BLK * GX_H            // x-parts of every unit's candidate [slot][dir][step][256]
       + (size_t)2 * GX_H + 2 * GK_UPM + 64;      // states [dir][256], r*h of the own units [dir][32], census words
}
#define GK_STAMP(slot)                                                                                            \
  do {                                                                                                            \
    if constexpr (TRACE) { if (tracer && s >= 8 && s < 8 + DX_TRACE_STEPS) a.trace[(s - 8) * DX_TRACE_SLOTS + (slot)] = (long long)__builtin_readcyclecounter(); } \
  } while (0)

template <bool TAPE, bool WT, bool TRACE>
__device__ __forceinline__ void gk_body(const GdArgs& a, float* gx_smem, int place, int slot, DxRt rt) {
  constexpr int NT = 512, H = GX_H, UPW = 4, WTC = WT ? 1 : 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int SG = 2 * GK_BLK * 2 * GK_UPM, SC = 2 * GK_BLK * H;      // floats of one ring slot: gates part, candidate part
  float* xg = gx_smem;                            // rings first: their LDS addresses go through M0
  float* xc = xg + 2 * SG;
  float* hs = xc + 2 * SC;                        // [2 dirs][H]
  float* rhs = hs + 2 * H;                        // [2 dirs][32]
  const int row = place * 4 + (slot & 3), member = slot >> 2;
  if (row >= a.B || member >= GK_MB) return;
  const int T = a.T;
  const int L = __builtin_amdgcn_readfirstlane(a.lengths ? a.lengths[row] : T);
  const bool tracer = TRACE && a.trace && row == 0 && member == 0 && tid == 0;
  // gates: wave w owns units 32 member + 4w + i of both directions (registers 8i + e: r_i, 8i + 4 + e: u_i, inputs 4 lane + e) -- as pairs (r_i, u_i);
  // candidate ROWS: thread (wave, lane) owns column j = 32 wave + (lane & 31) for the units 32 member + 16 (lane >> 5) + kk, kk < 16 -- pairs (kk, kk + 1)
  taco_f32x2 WG[2][16], WC[2][8];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const float* wp = a.wpack + (((size_t)member * 2 + d) * 48) * NT + tid;
#pragma unroll
    for (int i = 0; i < UPW; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) WG[d][4 * i + e] = (taco_f32x2){wp[(size_t)(8 * i + e) * NT], wp[(size_t)(8 * i + 4 + e) * NT]};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) WC[d][kk] = (taco_f32x2){wp[(size_t)(32 + 2 * kk) * NT], wp[(size_t)(33 + 2 * kk) * NT]};
  }
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)row * 2 * 2 * 9 * H;       // [dir][buffer][8 partial vectors | u][H]
  for (int i = tid; i < 2 * H; i += NT) hs[i] = a.h0 ? a.h0[(size_t)row * 2 * H + i] : 0.f;
  // x-part blocks: gates of the own units: item i = (dir, step j, gate g, quarter c of 8); candidates of all units: item i = (dir, step j, quarter c of 64)
  constexpr int NIG = 2 * GK_BLK * 2 * 8, NIC = 2 * GK_BLK * 64;      // 256, 1024 float4 items per block
  static_assert(NIG % 64 == 0 && NIC % NT == 0, "whole waves");
  const unsigned xg_lds = (unsigned)(size_t)(gx_lds_float*)xg, xc_lds = (unsigned)(size_t)(gx_lds_float*)xc;
  auto blk_fetch = [&](int s0, int ring) {
    if (wave * 64 < NIG) {                                                        // wave-uniform
      const int i = tid;
      const int c = i & 7, g = (i >> 3) & 1, j = (i >> 4) % GK_BLK, d = i / (16 * GK_BLK);
      const int sx = min(s0 + j, T - 1);
      const float* src = a.xproj + ((size_t)row * T + sx) * 6 * H + d * 3 * H + g * H + member * GK_UPM + 4 * c;
      gx_load_lds16(src, __builtin_amdgcn_readfirstlane(xg_lds + (unsigned)(ring * SG + 4 * (wave * 64)) * 4u));
    }
#pragma unroll
    for (int u = 0; u < NIC / NT; ++u) {
      const int i = u * NT + tid;
      const int c = i & 63, j = (i >> 6) % GK_BLK, d = i / (64 * GK_BLK);
      const int sx = min(s0 + j, T - 1);
      const float* src = a.xproj + ((size_t)row * T + sx) * 6 * H + d * 3 * H + 2 * H + 4 * c;
      gx_load_lds16(src, __builtin_amdgcn_readfirstlane(xc_lds + (unsigned)(ring * SC + 4 * (u * NT + wave * 64)) * 4u));
    }
  };
  blk_fetch(0, 0);
  blk_fetch(GK_BLK, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int tid_outer = tid, lane_outer = lane;
  float uu[2] = {0.f, 0.f}, rr_[2] = {0.f, 0.f};
  // the nine granules of a collect in flight: asked for ONCE, behind the products of the other direction's gates phase (a member reads
  // 18 KB per collect through its 64-byte-per-clock port to the L2: a second request, as k_bigru_oct's 256 granules afford, made the step
  // L2-bound -- 6150 clocks, measured)
  unsigned long long pq2[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) pq2[i] = 0ull;

  // G: the gates of the own units of direction D at step s
  auto gates = [&](auto Dc, int s, int lane, const dx_gu64* Xreq, bool req) {
    constexpr int D = decltype(Dc)::value;
    const int sb = s & (GK_BLK - 1), ring = (s / GK_BLK) & 1, ui = lane >> 4, ul = wave * UPW + ui;
    const float x0r = xg[ring * SG + ((D * GK_BLK + sb) * 2 + 0) * GK_UPM + ul], x0u = xg[ring * SG + ((D * GK_BLK + sb) * 2 + 1) * GK_UPM + ul];
    const float hk = hs[D * H + member * GK_UPM + ul];
    const float4 hx = *reinterpret_cast<const float4*>(hs + D * H + 4 * lane);
    taco_f32x2 acc[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i) acc[i] = WG[D][4 * i] * (taco_f32x2){hx.x, hx.x};
#pragma unroll
    for (int i = 0; i < UPW; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx.y, hx.y}, WG[D][4 * i + 1], acc[i]);
#pragma unroll
    for (int i = 0; i < UPW; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx.z, hx.z}, WG[D][4 * i + 2], acc[i]);
#pragma unroll
    for (int i = 0; i < UPW; ++i) acc[i] = __builtin_elementwise_fma((taco_f32x2){hx.w, hx.w}, WG[D][4 * i + 3], acc[i]);
    if (req && (wave >> 2) != D) {            // second request of the OTHER direction's collect (its waves: the ones that do not own this phase's)
#pragma unroll
      for (int i = 0; i < 9; ++i) pq2[i] = __hip_atomic_load(Xreq + (size_t)i * H + (tid & 255), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float v[2 * UPW], sm[2];
#pragma unroll
    for (int i = 0; i < UPW; ++i) { v[2 * i] = acc[i].x; v[2 * i + 1] = acc[i].y; }
    go_reduce<UPW, 2>(v, sm);
    rr_[D] = dx_sigmoid_fast(sm[0] + x0r);
    uu[D] = dx_sigmoid_fast(sm[1] + x0u);
    if ((lane & 15) == 0) rhs[D * GK_UPM + ul] = rr_[D] * hk;
    // (u is published with the partial sums, behind the wait for the collect in flight: a store here would sit in front of that wait)
  };
  // P: partial candidate sums of direction D over the own units, for all 256 columns
  auto partial = [&](auto Dc, int s, int lane) {
    constexpr int D = decltype(Dc)::value;
    const float* rp = rhs + D * GK_UPM + 16 * (lane >> 5);
    taco_f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 r4 = *reinterpret_cast<const float4*>(rp + 4 * q);
      acc = __builtin_elementwise_fma((taco_f32x2){r4.x, r4.y}, WC[D][2 * q], acc);
      acc = __builtin_elementwise_fma((taco_f32x2){r4.z, r4.w}, WC[D][2 * q + 1], acc);
    }
    const float h2 = acc.x + acc.y;
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(h2), __float_as_uint(h2), false, false);
    float tot = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    // the collect in flight across this phase has landed: wait for it HERE, ahead of the publish stores (see gd_landed)
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(pq2[i]), "+v"(tot));
    if (lane < 32) dx_publish<WTC>(X + (size_t)((D * 2 + (s & 1)) * 9 + member) * H + 32 * wave + lane, tot, (unsigned)s + 1u, rt);
    if ((lane & 15) == 0) {                         // the update gates of the wave's four units (unit = lane >> 4)
      const int ul = wave * UPW + (lane >> 4);
      dx_publish<WTC>(X + (size_t)((D * 2 + (s & 1)) * 9 + 8) * H + member * GK_UPM + ul, uu[D], (unsigned)s + 1u, rt);
      if (TAPE && s < L) {
        float* gs = a.gsave + ((size_t)row * T + (D ? L - 1 - s : s)) * 6 * H + D * 3 * H + member * GK_UPM + ul;
        gs[0] = rr_[D]; gs[H] = uu[D];
      }
    }
  };
  // E: every unit's candidate and new state, by the waves 4D .. 4D+3 (one unit per thread)
  auto finish = [&](auto Dc, int s, int tid) {
    constexpr int D = decltype(Dc)::value;
    if ((wave >> 2) == D) {
      const int j = tid & 255;
      const unsigned tg = (unsigned)s + 1u;
      const dx_gu64* Xv = X + (size_t)((D * 2 + (s & 1)) * 9) * H + j;
      float pv[9];
      bool ok2 = true;
#pragma unroll
      for (int i = 0; i < 9; ++i) ok2 = ok2 && ((unsigned)(pq2[i] >> 32) == tg);
      if (ok2) {
#pragma unroll
        for (int i = 0; i < 9; ++i) pv[i] = __uint_as_float((unsigned)pq2[i]);
      } else {
        dx_poll<9>(Xv, (size_t)H, tg, pv, rt);       // a producer was late: the ordinary bounded poll
      }
      const int sb = s & (GK_BLK - 1), ring = (s / GK_BLK) & 1;
      float cpre = xc[ring * SC + (D * GK_BLK + sb) * H + j];
#pragma unroll
      for (int m = 0; m < 8; ++m) cpre += pv[m];      // member order: every member forms the same sum
      const float cc = taco_tanh_fast(cpre), u = pv[8], hj = hs[D * H + j];
      float blend = u * hj + (1.f - u) * cc;
      DX_PIN(blend);
      const bool active = s < L;
      const float nv = active ? blend : hj;
      hs[D * H + j] = nv;
      if ((j >> 5) == member) {                        // the unit's owner
        const int t = (D && active) ? (L - 1 - s) : s;
        a.out[((size_t)row * T + t) * 2 * H + D * H + j] = active ? nv : 0.f;
        if (TAPE && active) a.gsave[((size_t)row * T + t) * 6 * H + D * 3 * H + 2 * H + j] = cc;
      }
    }
  };
  using F = std::integral_constant<int, 0>;
  using Bk = std::integral_constant<int, 1>;
  for (int s = 0; s < T; ++s) {
    int tid = tid_outer, lane = lane_outer;                 // opaque per-iteration copies: see taco_decoder_xcd.h
    asm volatile("" : "+v"(tid), "+v"(lane));
    GK_STAMP(0);
    const int sb = s & (GK_BLK - 1), ring = (s / GK_BLK) & 1;
    // (B's granules of step s - 1 were requested at the end of the previous iteration)
    gates(F{}, s, lane, X + (size_t)((2 + ((s - 1) & 1)) * 9) * H, s > 0);
    __syncthreads();                                        // r*h of the own units (F) visible
    GK_STAMP(1);
    partial(F{}, s, lane);
    GK_STAMP(2);
    if (s > 0) finish(Bk{}, s - 1, tid);
    __syncthreads();                                        // the new state (B) visible
    GK_STAMP(3);
    // the block after next goes into the slot whose last reader -- finish(B, s - 1) just now: the candidate x-parts of step s - 1 -- is done
    if (sb == 0 && s > 0) blk_fetch(s + GK_BLK, ring ^ 1);
    gates(Bk{}, s, lane, X + (size_t)((0 + (s & 1)) * 9) * H, true);
    __syncthreads();
    GK_STAMP(4);
    partial(Bk{}, s, lane);
    GK_STAMP(5);
    finish(F{}, s, tid);
    if (sb == GK_BLK - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's part of the next ring slot has landed
    __syncthreads();                                        // the new state (F) visible
    GK_STAMP(6);
  }
  {   // the backward direction's last step
    int tid = tid_outer;
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int i = 0; i < 9; ++i) pq2[i] = 0ull;
    finish(Bk{}, T - 1, tid);
  }
}

template <bool TAPE = false, bool TRACE = false>
__global__ __launch_bounds__(512) void k_bigru_ks(const GdArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float gx_smem[];
  GdArgs a = a_in;
  int* ictl = reinterpret_cast<int*>(gx_smem + gk_lds_floats() - 64);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, a.force_wt, ictl, threadIdx.x, 24);
  const int place = __builtin_amdgcn_readfirstlane(ictl[0]), slot = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  if (__builtin_amdgcn_readfirstlane((int)rt.wt)) gk_body<TAPE, true, TRACE>(a, gx_smem, place, slot, rt);
  else gk_body<TAPE, false, TRACE>(a, gx_smem, place, slot, rt);
}

