// taco_front.h -- the front of a CBHG as ONE launch: conv bank (k = 1..K, ReLU, BatchNorm) -> concat -> max_pooling1d(2, 1, 'same')
// -> first projection conv (width 3), modules.py:35-59.  The bank tensor [B*T, K*C] never exists: as separate launches it was
// written (134 MB at the post-net of C2) and staged again, max-pooled, by proj_1 -- 268 MB of the forward's traffic -- and proj_1
// ran on 64-row tiles whose weight stream (6.3 MB per workgroup from L2) bounds it at ~30 % of the matrix pipe.
//
// Work decomposition.  A workgroup (8 waves, one per CU) owns FR_BM = 128 consecutive frames of ONE batch row and one PART of the
// bank's channels (the contraction of proj_1 is split over P parts so that B * ceil(T / 128) * P workgroups fill the chip; parts
// are balanced by their tap counts on the host).  It walks its chunks of FR_CH = 128 bank channels of one width:
//   produce   Z[ch, frame] = sum_{tap, cin} Wk[tap, cin, ch] * x[frame + tap - padl, cin] for 144 frames (131 are needed: 128 + 1
//             for the pool + 2 for the width-3 conv) on v_mfma_f32_16x16x32_bf16, weights as the A operand (so that a lane ends up
//             with four CONSECUTIVE channels of one frame), the input tile -- staged once per workgroup as two bf16 planes (hi, lo)
//             with SAME zero rows outside [0, T) -- as the B operand: wave w owns channel tile w (16 channels) and all nine frame
//             tiles.  Three products per k32 step (lo*hi, hi*lo, hi*hi: the split-bf16 arithmetic of k_gemm_bf3).
//   epilogue  + bias -> ReLU -> BatchNorm affine (modules.py:123-131: activation BEFORE the normalisation) -> max with the next
//             frame (one ds_bpermute per value; the pool never looks past the row's last frame, A.3) -> zero for frames outside
//             [0, T) (proj_1's SAME padding) -> split into bf16 planes -> LDS tile A[130 frames][128 channels].
//   consume   acc[128 frames, N1] += sum_{tap < 3, ch} A[frame + tap, ch] * W1[tap, ch0 + ch, :] on v_mfma_f32_32x32x16_bf16 from the
//             ordinary pack_bf3 pack of proj_1: wave (ks, wn) owns all 128 frames x 32 TN columns (TM = 4 row tiles: every weight
//             fragment meets four row tiles -> half the weight bytes per flop of the 64-row tiles) for half of the k16 steps.
// After the last chunk the two K halves meet through LDS and the tile leaves as a PARTIAL sum [part][B*T][N1]; k_front_combine
// adds the parts in a fixed order and applies proj_1's own bias -> ReLU -> BatchNorm.  No atomics: bit-reproducible.
// Neither phase needs per-fragment padding masks: a tile never crosses a batch row, and what lies outside the row is zero in LDS.
#pragma once
#ifndef FR_ROT
#define FR_ROT 0
#endif
#include "taco_kernels.h"

#define FR_BM 128          // frames of proj_1 output per workgroup
#define FR_NRT 9           // 16-frame tiles of bank output computed per chunk (144 >= 131)
#define FR_PR 130          // pooled frames kept: t0 - 1 .. t0 + 128
#define FR_ALD 136         // bf16 elements per row of a pooled plane: 272 bytes = odd multiple of 16 -> conflict-free ds_read_b128
#define FR_CH 128          // bank channels per chunk
#define FR_MAXW 16
#define FR_MAXCH 32
#define FR_MAXP 16
#define FR_SMEM_CNT (159 * 1024)   // byte offset of the group-barrier counters in LDS (the kernel always asks for 160 KB)

struct FrWidth {
  const unsigned short* wh; const unsigned short* wl;      // produce pack (pack_front): [k32 step][16-channel tile][lane][8]
  const float* bias; const float* scale; const float* shift;
  int kw, xoff, ns, nct;                                   // taps, PADLMAX - padl, k32 steps, 16-channel tiles of this width
};
struct FrChunk { int wi, ct0, cg0; };                      // width index, first channel tile inside the width, first channel in the concatenation
struct FrArgs {
  const float* x; int ldx, B, T, Cin, tiles_per_b, P;
  const unsigned short* ph; const unsigned short* pl;      // proj_1 pack (pack_bf3 layout), NT column tiles, k16 groups per tap
  int pNT, pK16tap;
  float* part; int N1;
  int prio; int delay;                                               // shader clocks the second K half of every workgroup starts late (0: in phase)
  FrWidth w[FR_MAXW];
  FrChunk ch[FR_MAXCH];
  int pstart[FR_MAXP + 1];
};

#ifdef TACO_TRACE
__device__ long long taco_trace_front[64];      // its own slots: the GEMM launches that follow in a stage stamp taco_trace
#define FTRC(i) do { if (trc) taco_trace_front[i] = clock64(); } while (0)
#else
#define FTRC(i) do {} while (0)
#endif
// (every pointer these see has been PINned -- address space 1 -- so they compile to global_load, not flat_load)
__device__ __forceinline__ uint4 fr_ld16(const unsigned short* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ float4 fr_ldf4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Barrier among the four waves of one K half (one LDS counter per half, never reset: the k-th barrier waits for 4 k arrivals).  The
// two halves of a workgroup are independent until the final reduction -- produce wave w writes channels 16 w .. 16 w + 15 of the
// pooled planes and consume waves 4 ks .. 4 ks + 3 read exactly channels 64 ks .. 64 ks + 63 -- and each SIMD hosts one wave of
// each half, so with the halves half a chunk out of phase the VALU epilogue of one runs under the MFMAs of the other.  A
// workgroup-wide s_barrier would lock the phases together (measured: 21 % of a chunk with the matrix pipe idle).
__device__ __forceinline__ void fr_gbar(unsigned* cnt, unsigned target, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(2);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// XS = bf16 elements per row of an input plane, CINP = input channels padded to 16.  XS == CINP ("flat", CINP an odd multiple of
// 16): row r, tap j, channel c is element (r + j) XS + c = r XS + (j CINP + c) -- the (tap, channel) contraction index IS the
// offset, k32 steps may straddle taps and K is padded to 32 only once per width.  Otherwise CINP is a multiple of 32 and XS = CINP
// + 16 keeps the 16 rows of a fragment on distinct bank groups (XS / 8 odd in both cases).
template <int TN, int XS, int CINP, int KWMAX>
__global__ __launch_bounds__(512) void k_cbhg_front(const FrArgs a_in) {
  constexpr bool FLAT = XS == CINP;
  constexpr int PADLMAX = (KWMAX - 1) / 2;
  constexpr int XROWS = FLAT ? 16 * FR_NRT + KWMAX : FR_PR + KWMAX;        // rows staged (see the host's LDS size); reads past them stay inside LDS
  extern __shared__ __attribute__((aligned(16))) unsigned short fr_smem[];
  unsigned short* xhi = fr_smem;
  unsigned short* xlo = xhi + XROWS * XS;
  unsigned short* ahi = xlo + XROWS * XS;
  unsigned short* alo = ahi + FR_PR * FR_ALD;
  unsigned* gcnt = reinterpret_cast<unsigned*>(fr_smem + FR_SMEM_CNT / 2);        // two barrier counters, beyond everything else (fixed offset)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int trci = 3;      // TACO_TRACE slots: 0 entry, 1 staged, from 3 per chunk (produce loop done, other waves done reading, planes written, consume done),
#ifdef TACO_TRACE    // then K halves met, stored
  const bool trc = (blockIdx.x == gridDim.x / 2) && threadIdx.x == 0;
  FTRC(0);
#endif
  const int T = a_in.T, P = a_in.P;
  const float* gx = a_in.x; const unsigned short* gph = a_in.ph; const unsigned short* gpl = a_in.pl; float* gpart = a_in.part;
  PIN(gx); PIN(gph); PIN(gpl); PIN(gpart);
  // workgroup -> (tile, part): consecutive workgroup ids go round the XCDs, so give every XCD ONE part where that divides evenly
  // (its L2 then streams one part's weights)
  int part, tile;
  { const int id = blockIdx.x;
    if (P <= 8 && (8 % P) == 0 && (gridDim.x % 8) == 0) { part = (id & 7) % P; tile = (id >> 3) * (8 / P) + (id & 7) / P; }
    else { part = id % P; tile = id / P; } }
  const int b = tile / a_in.tiles_per_b, t0 = (tile - b * a_in.tiles_per_b) * FR_BM;
  const size_t mrow0 = (size_t)b * T;

  // ---- stage the input frames t0 - 1 - PADLMAX ..: fp32 -> (hi, lo) planes, zero outside [0, T); all loads of a thread in flight at once ----
  {
    const float* xb = gx + mrow0 * a_in.ldx;
    const bool vec = (a_in.ldx & 3) == 0 && (a_in.Cin & 3) == 0 && ((reinterpret_cast<uintptr_t>(a_in.x) & 15) == 0);
    constexpr int NST = (XROWS * (CINP / 4) + 511) / 512;
    float4 f[NST];
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int i = tid + 512 * u;
      const int r = i / (CINP / 4), c = 4 * (i - r * (CINP / 4));
      const int t = t0 - 1 - PADLMAX + r;
      f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < XROWS && t >= 0 && t < T && c < a_in.Cin) {
        const float* p = xb + (size_t)t * a_in.ldx + c;
        if (vec) f[u] = fr_ldf4(p);
        else { f[u].x = p[0]; if (c + 1 < a_in.Cin) f[u].y = p[1]; if (c + 2 < a_in.Cin) f[u].z = p[2]; if (c + 3 < a_in.Cin) f[u].w = p[3]; }
      }
    }
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int i = tid + 512 * u;
      const int r = i / (CINP / 4), c = 4 * (i - r * (CINP / 4));
      if (r < XROWS) {
        uint2 h4, l4;
        taco_split_bf16x4(f[u], h4, l4);
        *reinterpret_cast<uint2*>(xhi + r * XS + c) = h4;
        *reinterpret_cast<uint2*>(xlo + r * XS + c) = l4;
      }
    }
  }
  FTRC(1);
  // produce coordinates: frame lane & 15 of a 16-frame tile, k-quarter lane >> 4 (A/B operands of the 16x16x32 MFMA hold k = 8 q .. 8 q + 7)
  const int pc = lane & 15, pq = lane >> 4;
  unsigned vmask = 0, nmask = 0;     // bit rt: frame t0 - 1 + 16 rt + pc lies in [0, T) / has a successor inside the row
#pragma unroll
  for (int rt = 0; rt < FR_NRT; ++rt) {
    const int t = t0 - 1 + 16 * rt + pc;
    if (t >= 0 && t < T) vmask |= 1u << rt;
    if (t >= 0 && t + 1 < T) nmask |= 1u << rt;
  }
  const int nxt_lane4 = 4 * ((lane & 48) | ((pc + 1) & 15));      // ds_bpermute address of the lane that holds the next frame
  // consume coordinates
  const int l31 = lane & 31, lh = lane >> 5;
  const int ks = wave >> 2, wn = wave & 3;
  int ntc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) ntc[tn] = min(wn * TN + tn, a_in.pNT - 1);
  f32x16 acc[4][TN];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const int c_begin = a_in.pstart[part], c_end = a_in.pstart[part + 1];
  if (tid < 2) gcnt[tid] = 0u;
  __syncthreads();
  unsigned nbar = 0;
  // the second half starts a few microseconds late (a_in.delay shader clocks): from then on one half's VALU epilogue lies under the
  // other half's MFMAs instead of next to its epilogue
  if (ks == 1 && a_in.prio == 1) __builtin_amdgcn_s_setprio(1);        // the younger waves of a SIMD lose the arbitration otherwise: measured,
  if (ks == 1 && a_in.prio == 2) __builtin_amdgcn_s_setprio(2);        // the second half trails the first by 15 % of the kernel
  if (ks == 1 && a_in.prio == 3) __builtin_amdgcn_s_setprio(3);
  if (ks == 1 && a_in.delay > 0) {
    const long long c0 = clock64();
    while (clock64() - c0 < a_in.delay) __builtin_amdgcn_s_sleep(8);
  }

  // FR_ROT: the workgroups of an XCD (one part: the same chunks) start at different chunks, so that they do not all ask the L2 for
  // the same weight slices at the same moment; a workgroup's summation order over its chunks is fixed by its id (bit-reproducible)
  const int nchunks = c_end - c_begin, crot = FR_ROT ? (int)(blockIdx.x >> 3) % max(nchunks, 1) : 0;
  for (int cj = 0; cj < nchunks; ++cj) {
    const int ci = c_begin + (cj + crot >= nchunks ? cj + crot - nchunks : cj + crot);
    const FrChunk ch = a_in.ch[ci];
    FrWidth W = a_in.w[ch.wi];
    PIN(W.wh); PIN(W.wl); PIN(W.bias); PIN(W.scale); PIN(W.shift);
    // opaque per-chunk copies of the lane coordinates (otherwise every LDS address of both phases is hoisted out of the chunk loop
    // and kept -- spilled -- across it: the same effect as in taco_chain.h)
    int pcv = pc, pqv = pq, l31v = l31, lhv = lh;
    asm volatile("" : "+v"(pcv), "+v"(pqv), "+v"(l31v), "+v"(lhv));
    // ================= produce: Z[16 ch of tile ct][144 frames] =================
    f32x4 z[FR_NRT];
#pragma unroll
    for (int rt = 0; rt < FR_NRT; ++rt) z[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      const int ct = ch.ct0 + wave;
      const size_t sstride = (size_t)W.nct * 512;                                    // elements per k32 step
      const unsigned short* wh = W.wh + ((size_t)ct * 64 + lane) * 8;
      const unsigned short* wl = W.wl + ((size_t)ct * 64 + lane) * 8;
      const unsigned short* xbh = xhi + (pcv + W.xoff) * XS + 8 * pqv;
      const unsigned short* xbl = xlo + (pcv + W.xoff) * XS + 8 * pqv;
      // Software pipeline over BATCHES of three frame tiles (6 ds_read_b128 -> 9 MFMAs, the three products of a tile three MFMAs
      // apart): the fragments of batch i + 1 are requested before the MFMAs of batch i are issued, in two register sets that swap
      // roles, so a wave never waits for the LDS it has just asked (with two waves per SIMD nothing else would fill that gap).
      // Weight fragments: two sets, one k32 step ahead (the step count is even: pack_front pads it).
      auto xoff_of = [&](int s) {
        if constexpr (FLAT) return 32 * s;
        else { const int kk = 32 * s; const int tap = kk / CINP; return tap * XS + (kk - tap * CINP); }
      };
      auto xload = [&](int s, int b, bf16x8 (&xh)[3], bf16x8 (&xl)[3]) {
        const int xo = xoff_of(s);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          xh[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xbh + (16 * (3 * b + i)) * XS + xo));
          xl[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xbl + (16 * (3 * b + i)) * XS + xo));
        }
      };
      auto pmma = [&](int b, const uint4& wh4, const uint4& wl4, const bf16x8 (&xh)[3], const bf16x8 (&xl)[3]) {
        const bf16x8 whv = __builtin_bit_cast(bf16x8, wh4), wlv = __builtin_bit_cast(bf16x8, wl4);
#pragma unroll
        for (int i = 0; i < 3; ++i) z[3 * b + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlv, xh[i], z[3 * b + i], 0, 0, 0);   // small terms first
#pragma unroll
        for (int i = 0; i < 3; ++i) z[3 * b + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whv, xl[i], z[3 * b + i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) z[3 * b + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whv, xh[i], z[3 * b + i], 0, 0, 0);
      };
      uint4 p_h = fr_ld16(wh), p_l = fr_ld16(wl), q_h, q_l;
      bf16x8 xPh[3], xPl[3], xQh[3], xQl[3];
      xload(0, 0, xPh, xPl);
#pragma unroll 1
      for (int s = 0; s < W.ns; s += 2) {
        const int sn = min(s + 2, W.ns - 1);                       // past the end: fetched again, never used
        q_h = fr_ld16(wh + (s + 1) * sstride); q_l = fr_ld16(wl + (s + 1) * sstride);
        xload(s, 1, xQh, xQl);
        __builtin_amdgcn_sched_barrier(0);
        pmma(0, p_h, p_l, xPh, xPl);
        __builtin_amdgcn_sched_barrier(0);
        xload(s, 2, xPh, xPl);
        __builtin_amdgcn_sched_barrier(0);
        pmma(1, p_h, p_l, xQh, xQl);
        __builtin_amdgcn_sched_barrier(0);
        xload(s + 1, 0, xQh, xQl);
        __builtin_amdgcn_sched_barrier(0);
        pmma(2, p_h, p_l, xPh, xPl);
        __builtin_amdgcn_sched_barrier(0);
        p_h = fr_ld16(wh + sn * sstride); p_l = fr_ld16(wl + sn * sstride);
        xload(s + 1, 1, xPh, xPl);
        __builtin_amdgcn_sched_barrier(0);
        pmma(0, q_h, q_l, xQh, xQl);
        __builtin_amdgcn_sched_barrier(0);
        xload(s + 1, 2, xQh, xQl);
        __builtin_amdgcn_sched_barrier(0);
        pmma(1, q_h, q_l, xPh, xPl);
        __builtin_amdgcn_sched_barrier(0);
        xload(sn, 0, xPh, xPl);
        __builtin_amdgcn_sched_barrier(0);
        pmma(2, q_h, q_l, xQh, xQl);
        __builtin_amdgcn_sched_barrier(0);
      }
      // bias -> ReLU -> BatchNorm affine, per channel 16 ct + 4 q + reg (C/D map of the 16x16 MFMA: column = lane & 15, row = 4 (lane >> 4) + reg)
      const float4 bi = fr_ldf4(W.bias + 16 * ct + 4 * pqv), sc = fr_ldf4(W.scale + 16 * ct + 4 * pqv), sh = fr_ldf4(W.shift + 16 * ct + 4 * pqv);
#pragma unroll
      for (int rt = 0; rt < FR_NRT; ++rt) {
        z[rt][0] = fmaxf(z[rt][0] + bi.x, 0.f) * sc.x + sh.x;
        z[rt][1] = fmaxf(z[rt][1] + bi.y, 0.f) * sc.y + sh.y;
        z[rt][2] = fmaxf(z[rt][2] + bi.z, 0.f) * sc.z + sh.z;
        z[rt][3] = fmaxf(z[rt][3] + bi.w, 0.f) * sc.w + sh.w;
      }
    }
    // max-pool with the next frame, zero outside the row, split into the two planes -- all in registers, before the barrier
    uint2 ph4[FR_NRT], pl4[FR_NRT];
#pragma unroll
    for (int rt = 0; rt < FR_NRT; ++rt) {
      float4 p;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float own = z[rt][e];
        const float give = (rt + 1 < FR_NRT && pcv == 0) ? z[rt + 1 < FR_NRT ? rt + 1 : rt][e] : own;     // lane 0 of a 16-frame tile hands out the NEXT tile's first frame
        const float nx = __int_as_float(__builtin_amdgcn_ds_bpermute(nxt_lane4, __float_as_int(give)));
        float v = ((nmask >> rt) & 1u) ? fmaxf(own, nx) : own;
        v = ((vmask >> rt) & 1u) ? v : 0.f;
        (&p.x)[e] = v;
      }
      taco_split_bf16x4(p, ph4[rt], pl4[rt]);
      asm volatile("" : "+v"(ph4[rt].x), "+v"(ph4[rt].y), "+v"(pl4[rt].x), "+v"(pl4[rt].y));     // computed HERE, not sunk below the barrier into the guarded stores
    }
    // first weight fragments of the consume phase, requested before the barriers
    const int g0 = 4 * ks;
    const size_t k16base = (size_t)(ch.cg0 >> 4) + g0;
    auto boff = [&](int u, int tn) {          // step u = (tap j = u >> 2, k16 group g0 + (u & 3)) of this wave's half
      const size_t k16 = (size_t)(u >> 2) * a_in.pK16tap + k16base + (u & 3);
      return (((k16 * a_in.pNT + ntc[tn]) * 2 + lhv) * 32 + l31v) * 8;
    };
    uint4 pbh[TN], pbl[TN], qbh[TN], qbl[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) { pbh[tn] = fr_ld16(gph + boff(0, tn)); pbl[tn] = fr_ld16(gpl + boff(0, tn)); }
    FTRC(trci); ++trci;
    nbar += 4; fr_gbar(gcnt + ks, nbar, lane);       // the four waves of this half have finished reading the previous chunk's planes
    FTRC(trci); ++trci;
    // 4 consecutive channels of one frame = 8 bytes per plane
#pragma unroll
    for (int rt = 0; rt < FR_NRT; ++rt) {
      const int fr = 16 * rt + pcv;
      if (16 * rt + 15 < FR_PR || fr < FR_PR) {       // only the last tile has frames past the planes
        *reinterpret_cast<uint2*>(ahi + fr * FR_ALD + 16 * wave + 4 * pqv) = ph4[rt];
        *reinterpret_cast<uint2*>(alo + fr * FR_ALD + 16 * wave + 4 * pqv) = pl4[rt];
      }
    }
    nbar += 4; fr_gbar(gcnt + ks, nbar, lane);       // this half's 64 channels of the planes are complete
    FTRC(trci); ++trci;
    // ================= consume: acc[128 frames][32 TN columns] over this wave's 12 (tap, k16) steps =================
    // the same pipeline: the A fragments (pooled planes) and the weight fragments of step u + 1 are requested before the MFMAs of step u
    auto aload = [&](int u, bf16x8 (&ah)[4], bf16x8 (&al)[4]) {
      const int j = u >> 2, g = g0 + (u & 3);
      const unsigned short* ph_ = ahi + (l31v + j) * FR_ALD + 16 * g + 8 * lhv;
      const unsigned short* pl_ = alo + (l31v + j) * FR_ALD + 16 * g + 8 * lhv;
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        ah[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ph_ + tm * 32 * FR_ALD));
        al[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(pl_ + tm * 32 * FR_ALD));
      }
    };
    auto mma = [&](const bf16x8 (&ah)[4], const bf16x8 (&al)[4], const uint4 (&bh)[TN], const uint4 (&bl)[TN]) {
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const bf16x8 bhv = __builtin_bit_cast(bf16x8, bh[tn]), blv = __builtin_bit_cast(bf16x8, bl[tn]);
            const bf16x8 aa = (term == 0) ? al[tm] : ah[tm];          // al*bh, ah*bl, ah*bh
            const bf16x8 bb = (term == 1) ? blv : bhv;
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa, bb, acc[tm][tn], 0, 0, 0);
          }
    };
    bf16x8 aPh[4], aPl[4], aQh[4], aQl[4];
    aload(0, aPh, aPl);
#pragma unroll 1
    for (int u = 0; u < 12; u += 2) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) { qbh[tn] = fr_ld16(gph + boff(u + 1, tn)); qbl[tn] = fr_ld16(gpl + boff(u + 1, tn)); }
      aload(u + 1, aQh, aQl);
      __builtin_amdgcn_sched_barrier(0);
      mma(aPh, aPl, pbh, pbl);
      __builtin_amdgcn_sched_barrier(0);
      const int un = (u + 2 < 12) ? u + 2 : u + 1;                 // past the end: fetched again, never used
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) { pbh[tn] = fr_ld16(gph + boff(un, tn)); pbl[tn] = fr_ld16(gpl + boff(un, tn)); }
      aload(un, aPh, aPl);
      __builtin_amdgcn_sched_barrier(0);
      mma(aQh, aQl, qbh, qbl);
      __builtin_amdgcn_sched_barrier(0);
    }
    FTRC(trci); ++trci;
  }

  // ---- the two K halves meet through LDS (everything staged is dead): each half hands the other two of its four row tiles, adds what
  // it receives, and stores its two row tiles of this part's partial sum -- all eight waves store ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(fr_smem) + (size_t)wave * (2 * TN * 16 * 64);
  const int give0 = ks == 0 ? 2 : 0, keep0 = ks == 0 ? 0 : 2;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((i * TN + tn) * 16 + r) * 64 + lane] = ks == 0 ? acc[2 + i][tn][r] : acc[i][tn][r];
  __syncthreads();
  FTRC(trci); ++trci;
  (void)give0;
  const float* got = reinterpret_cast<const float*>(fr_smem) + (size_t)(wave ^ 4) * (2 * TN * 16 * 64);
  float* outp = gpart + ((size_t)part * a_in.B * T + mrow0) * a_in.N1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = (wn * TN + tn) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = t0 + (keep0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float mine = ks == 0 ? acc[i][tn][r] : acc[2 + i][tn][r];
        const float v = mine + got[((i * TN + tn) * 16 + r) * 64 + lane];
        if (t < T && col < a_in.N1) outp[(size_t)t * a_in.N1 + col] = v;
      }
    }
  FTRC(trci);
#ifdef TACO_TRACE
  if (trc) taco_trace_front[2] = trci;      // index of the last stamp THIS launch wrote (a workgroup of another launch may have had more chunks: stale slots beyond it)
#endif
}

// out[m][n] = act(sum over the parts (fixed order) + bias[n]) * scale[n] + shift[n]   (proj_1's epilogue, modules.py:123-131)
__global__ __launch_bounds__(256) void k_front_combine(const float* part, int P, size_t MN, int N1, const float* bias, const float* scale,
                                                       const float* shift, int relu, float* out) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= MN) return;
  float4 s = fr_ldf4(part + i);
  for (int p = 1; p < P; ++p) { const float4 q = fr_ldf4(part + (size_t)p * MN + i); s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w; }
  const int n = (int)(i % (size_t)N1);
  const float4 bi = bias ? fr_ldf4(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 sc = scale ? fr_ldf4(scale + n) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 sh = shift ? fr_ldf4(shift + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 o;
  o.x = s.x + bi.x; o.y = s.y + bi.y; o.z = s.z + bi.z; o.w = s.w + bi.w;
  if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
  o.x = o.x * sc.x + sh.x; o.y = o.y * sc.y + sh.y; o.z = o.z * sc.z + sh.z; o.w = o.w * sc.w + sh.w;
  *reinterpret_cast<float4*>(out + i) = o;
}
