// taco_decoder_xcd.h -- the whole decoder loop (K10-K17) as ONE persistent launch,
// weight-stationary and XCD-local.
//
// Reference semantics: rnn_wrappers.py:218-341 (AttentionWrapper.call), :367-378 (DecoderPrenetWrapper), :405-415
// (ConcatOutputAndAttentionWrapper); tacotron.py:127-181 (cells), helpers.py:9-32 (TacoTestHelper: feed back the last of the r
// frames, stop flags); SURVEY App. A.6 (TF GRUCell), A.9-A.11 (scores and normalisers).
//
// Why this shape.  A decoder step is a chain of 12 dependent mat-vec / attention stages over [B, <=768] activations and
// 6.1 MB of weights.  As one launch per stage the chain costs ~4.5 us per link on MI355X (profiles/r01_*: kernel boundary +
// cold weight fetch + a short MFMA chain), 42.5 us per step at C2.  Here the step never leaves the chip's registers:
//   * 256 workgroups of 512 threads (two waves per SIMD: 256 VGPRs per thread), one per CU.  The 32 CUs of one XCD form a GROUP that owns RG batch rows for the whole
//     loop (C2: 8 groups x 4 rows; C5: 8 x 1).  Groups never talk to each other.
//   * Weight-stationary: member m of a group owns 1/32 of the output columns of every stage and keeps exactly those weights
//     in VGPRs for the whole launch (96 registers per thread = 192 KB per CU; loaded once, coalesced, from a per-thread pack).
//   * Attention memory stationary: the member also keeps its slice of the keys (a block of encoder positions of one of the
//     group's rows, all channels) and of the values (all positions, a block of channels) in LDS -- the per-step K/V stream of
//     the launch-per-stage path (8.4 MB per step at C2) disappears.
//   * Between stages the members exchange their column slices through the XCD's own L2: 8-byte {value, tag = step+1}
//     granules (the data is the flag), producer store -> consumer poll with L1-bypassing (sc1) loads.  When every XCD runs
//     exactly one 32-member group (checked by an in-kernel census of HW_REG_XCC_ID) the producer stores are PLAIN stores:
//     the line stays in the shared L2 and the hop costs an L2 round trip.  Otherwise (or with force_wt) the stores are
//     write-through (sc1) and the protocol is placement-independent (MI355X_MICROARCH.md, inter-workgroup visibility).
//   * Every spin is bounded; a timeout raises a.err and the launch drains without hanging.
//
// Thread mapping of a mat-vec unit U<K, NCW, NWV> on a member: lane = c*KSL + ks with KSL = 64/NCW lanes per column, wave w <
// NWV takes the w-th K super-slice; thread (w, c, ks) holds W[k][col(c)] for k = 2*(ks + KSL*(w + NWV*j)) + e, j < NCH,
// e < 2 in registers REG0 + 2j + e, reads the matching float2 of every row's input vector from LDS (conflict-free: a wave
// reads KSL consecutive float2), reduces over ks with 2-4 full-rate DPP steps and leaves NWV partial sums per output in LDS
// for the epilogue threads.
#pragma once
#include "taco_kernels.h"

typedef __attribute__((address_space(1))) unsigned long long dx_gu64;
typedef __attribute__((address_space(1))) unsigned dx_gu32;

#define DX_NT 512
#define DX_NW 8
#define DX_GROUP 32          // members (CUs) per group
#define DX_NGROUP 8          // groups (XCDs)
#define DX_W 256             // attention_state_size = dec_rnn_size = attention_size = 2*enc_rnn_size = dec_prenet[0]
#define DX_P2 128            // dec_prenet[1]
#define DX_SPIN_LIMIT (1u << 21)
#define DX_XREGS 32          // registers a mat-vec unit may spend on input values in flight
#define DX_TRACE_STEPS 8
#define DX_TRACE_SLOTS 16

// register map of the per-thread weight pack (host mirror: dx_build_pack in taco_lib.hip)
enum {
  DXR_P1 = 0,    // [out2 | ctx] (512) -> 8 columns: composite of frame projection and prenet layer 1 (taco_lib.hip, prenet1_next)
  DXR_P2 = 8,    // 256 -> 4 columns
  DXR_AG = 10,   // attention GRU gates: [p2 | h] (384) -> r (8 cols) | u (8 cols) of this member
  DXR_AX = 22,   // attention GRU candidate, x rows: p2 (128) -> 8
  DXR_AC = 24,   // attention GRU candidate, h rows: r*h (256) -> 8
  DXR_Q = 28,    // query layer 256 -> 8
  DXR_CP = 32,   // concat projection [h_att | ctx] (512) -> 8
  DXR_G1G = 40, DXR_G1X = 56, DXR_G1C = 60,   // decoder GRU 1: [o0 | h1] (512) -> 16; o0 -> 8; r*h1 -> 8
  DXR_G2G = 64, DXR_G2X = 80, DXR_G2C = 84,   // decoder GRU 2
  DXR_F = 88,    // frame projection 256 -> up to 16 columns of r*num_mels
  DX_NREG = 96
};

template <int K_, int NCW_, int NWV_, int REG0_>
struct DxU {
  static constexpr int K = K_, NCW = NCW_, NWV = NWV_, REG0 = REG0_;
  static constexpr int KSL = 64 / NCW_;
  static constexpr int NCH = K_ / (2 * KSL * NWV_);
  static_assert(NCH * 2 * KSL * NWV_ == K_ && NCH >= 1 && NWV_ <= DX_NW, "unit does not tile");
};
typedef DxU<512, 8, 8, DXR_P1> DxU_P1;
typedef DxU<256, 4, 8, DXR_P2> DxU_P2;
typedef DxU<384, 16, 8, DXR_AG> DxU_AG;
typedef DxU<128, 8, 8, DXR_AX> DxU_AX;
typedef DxU<256, 8, 8, DXR_AC> DxU_AC;
typedef DxU<256, 8, 8, DXR_Q> DxU_Q;
typedef DxU<512, 8, 8, DXR_CP> DxU_CP;
typedef DxU<512, 16, 8, DXR_G1G> DxU_G1G;
typedef DxU<256, 8, 8, DXR_G1X> DxU_G1X;
typedef DxU<256, 8, 8, DXR_G1C> DxU_G1C;
typedef DxU<512, 16, 8, DXR_G2G> DxU_G2G;
typedef DxU<256, 8, 8, DXR_G2X> DxU_G2X;
typedef DxU<256, 8, 8, DXR_G2C> DxU_G2C;
typedef DxU<256, 16, 8, DXR_F> DxU_F;
static_assert(DXR_P2 - DXR_P1 == 2 * DxU_P1::NCH && DXR_AG - DXR_P2 == 2 * DxU_P2::NCH && DXR_AX - DXR_AG == 2 * DxU_AG::NCH &&
              DXR_AC - DXR_AX == 2 * DxU_AX::NCH && DXR_Q - DXR_AC == 2 * DxU_AC::NCH && DXR_CP - DXR_Q == 2 * DxU_Q::NCH &&
              DXR_G1G - DXR_CP == 2 * DxU_CP::NCH && DXR_G1X - DXR_G1G == 2 * DxU_G1G::NCH && DXR_G1C - DXR_G1X == 2 * DxU_G1X::NCH &&
              DXR_G2G - DXR_G1C == 2 * DxU_G1C::NCH && DXR_G2X - DXR_G2G == 2 * DxU_G2G::NCH && DXR_G2C - DXR_G2X == 2 * DxU_G2X::NCH &&
              DXR_F - DXR_G2C == 2 * DxU_G2C::NCH && DX_NREG - DXR_F == 2 * DxU_F::NCH, "register map");

// per-row state vectors in LDS (floats)
enum { DXS_P2 = 0, DXS_HATT = 128, DXS_CTX = 384, DXS_OUT2 = 640, DXS_T = 896, DXS_O0 = 1152, DXS_H1 = 1408, DXS_OUT1 = 1664,
       DXS_H2 = 1920, DXS_LD = 2176 };

// exchange buffers of one group, in granules, for RG rows (host mirror: dx_xbuf_granules)
struct DxX { int p1, p2, rha, ha, q, sc, ctx, o0, rh1, h1, rh2, h2, total; };
__host__ __device__ inline DxX dx_xlayout(int RG, int T_in) {
  DxX x; int o = 0;
  x.p1 = o; o += RG * DX_W;  x.p2 = o; o += RG * DX_P2; x.rha = o; o += RG * DX_W; x.ha = o; o += RG * DX_W;
  x.q = o; o += RG * DX_W;   x.sc = o; o += RG * T_in;  x.ctx = o; o += RG * DX_W; x.o0 = o; o += RG * DX_W;
  x.rh1 = o; o += RG * DX_W; x.h1 = o; o += RG * DX_W;  x.rh2 = o; o += RG * DX_W; x.h2 = o; o += RG * DX_W;
  x.total = o;
  return x;
}
// LDS bytes of a member (host mirror of the carve in the kernel)
__host__ __device__ inline size_t dx_lds_floats(int RG, int T_in) {
  const int Pr = DX_GROUP / RG, TP = (T_in + Pr - 1) / Pr, DC = DX_W / Pr;
  size_t n = 0;
  n += (size_t)RG * DXS_LD;           // state
  n += (size_t)DX_NW * RG * 16;       // red0
  n += (size_t)DX_NW * RG * 16;       // red1
  n += (size_t)TP * DX_W;             // keys slice
  n += (size_t)T_in * DC;             // values slice
  n += 4 * (size_t)((T_in + 3) & ~3); // sc, tmp, tmp2, alp
  n += 3 * DX_W;                      // qv, vv, bias scratch
  n += (size_t)DX_NW * 64;            // context partials [wave][DC] (DC <= 64)
  n += 16 * 16;                       // own-column biases of the 14 epilogues
  n += 64;                            // control words
  return n;
}

struct DxArgs {
  const float* wpack;                                  // [32 members][DX_NREG][DX_NT]
  const float* b_p1_0; const float* b_p1c; const float* b_p2;      // prenet biases: layer 1 raw (step 0), composite (steps >= 1), layer 2
  const float* b_ag; const float* b_ac;                // attention GRU: gates [2H] (r|u), candidate [H]
  const float* b_cp;
  const float* b_g1g; const float* b_g1c; const float* b_g2g; const float* b_g2c;
  const float* b_f;                                    // [rM]
  const float* att_v; const float* att_b; const float* score_bias;
  const float* keys; const float* values;              // [B, T_in, 256] each
  const float* h_att0; const float* h10; const float* h20;   // deepvoice initial states [B, 256] or null (zeros)
  float* mel; float* hist; int* nz; float* dbg;
  unsigned long long* xbuf; unsigned* ctl; unsigned* err; long long* trace;
  int B, T_in, n, rM, att_type, grp0, ngroups, force_wt, dbgw;
};

#define DX_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, false))
// sum over the KSL lanes of a column (the low bits of the lane index); every lane of the column ends with the total
template <int KSL>
__device__ __forceinline__ float dx_reduce(float v) {
  v += DX_DPP(v, 0xB1);                 // quad_perm [1,0,3,2]
  v += DX_DPP(v, 0x4E);                 // quad_perm [2,3,0,1]
  if (KSL >= 8) v += DX_DPP(v, 0x141);  // row_half_mirror
  if (KSL >= 16) v += DX_DPP(v, 0x140); // row_mirror
  return v;
}
__device__ __forceinline__ float dx_wave_sum(float v) {
  v = dx_reduce<16>(v);
  const int a = __builtin_amdgcn_readlane(__float_as_int(v), 0), b = __builtin_amdgcn_readlane(__float_as_int(v), 16);
  const int c = __builtin_amdgcn_readlane(__float_as_int(v), 32), d = __builtin_amdgcn_readlane(__float_as_int(v), 48);
  return (__int_as_float(a) + __int_as_float(b)) + (__int_as_float(c) + __int_as_float(d));
}

// partial sums of unit U for RG rows: x0 holds the first K0 inputs of a row, x1 the rest (LDS, row stride DXS_LD)
template <class U, int RG, int K0>
__device__ __forceinline__ void dx_matvec(const float (&W)[DX_NREG], const float* x0, const float* x1, float* red, int wave, int lane) {
  if (wave >= U::NWV) return;
  const int ks = lane & (U::KSL - 1), c = lane / U::KSL;
  float acc[RG];
#pragma unroll
  for (int r = 0; r < RG; ++r) acc[r] = 0.f;
  // the input reads of at most DX_XREGS/2 float2 are in flight at a time (left alone, the scheduler hoists every read of the
  // unit above the first FMA: 128 registers at RG = 8)
  constexpr int JG = (DX_XREGS / (2 * RG)) > 0 ? (DX_XREGS / (2 * RG)) : 1;
#pragma unroll
  for (int j = 0; j < U::NCH; ++j) {
    const int kb = 2 * U::KSL * (wave + U::NWV * j);            // wave-uniform: the whole wave reads one segment
    const float* xs = (kb < K0) ? x0 + kb : x1 + (kb - K0);
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      const float2 xv = *reinterpret_cast<const float2*>(xs + r * DXS_LD + 2 * ks);
      acc[r] = fmaf(W[U::REG0 + 2 * j], xv.x, acc[r]);
      acc[r] = fmaf(W[U::REG0 + 2 * j + 1], xv.y, acc[r]);
    }
    if ((j + 1) % JG == 0 && j + 1 < U::NCH) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < RG; ++r) acc[r] = dx_reduce<U::KSL>(acc[r]);
  if (ks == 0) {
#pragma unroll
    for (int r = 0; r < RG; ++r) red[(wave * RG + r) * U::NCW + c] = acc[r];
  }
}
// the epilogue thread's sum over the NWV wave partials of output (r, c)
template <class U, int RG>
__device__ __forceinline__ float dx_partials(const float* red, int r, int c) {
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < U::NWV; ++w) s += red[(w * RG + r) * U::NCW + c];
  return s;
}

struct DxRt {     // run-time state of a thread
  dx_gu32* err; bool wt; bool dead;
};
__device__ __forceinline__ void dx_publish(dx_gu64* p, float v, unsigned tag, const DxRt& rt) {
  const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
  if (rt.wt) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // sc1: write-through, any placement
  else __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);           // plain: stays in this XCD's L2
}
// poll one granule until it carries `tag` (L1-bypassing loads); bounded
__device__ __forceinline__ float dx_consume(const dx_gu64* p, unsigned tag, DxRt& rt) {
  unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!rt.dead) {
    unsigned spins = 0;
    while ((unsigned)(g >> 32) != tag) {
      g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((++spins & 1023u) == 0) {
        if (spins >= DX_SPIN_LIMIT || __hip_atomic_load(rt.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store(rt.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          rt.dead = true;
          break;
        }
      }
    }
  }
  return __uint_as_float((unsigned)g);
}
// all-gather of a published [RG][N] vector into the LDS state vector at column offset `off` (N a power of two);
// RES: dst2[r][n] = value + res[r][n] as well (ResidualWrapper output, tacotron.py:172).  All of a thread's granules are
// requested before the first is examined; only stale ones are polled again.
template <int RG, int N, bool RES>
__device__ __forceinline__ void dx_gather(const dx_gu64* X, unsigned tag, float* st, int off, int off_res, int off2, int tid, DxRt& rt) {
  constexpr int NI = (RG * N + DX_NT - 1) / DX_NT;
  const bool act = (RG * N >= DX_NT) || tid < RG * N;
  unsigned long long g[NI];
  if (act) {
#pragma unroll
    for (int u = 0; u < NI; ++u) g[u] = __hip_atomic_load(X + u * DX_NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int i = u * DX_NT + tid;
      float v = __uint_as_float((unsigned)g[u]);
      if ((unsigned)(g[u] >> 32) != tag) v = dx_consume(X + i, tag, rt);
      const int r = i / N, n = i % N;
      st[r * DXS_LD + off + n] = v;
      if (RES) st[r * DXS_LD + off2 + n] = v + st[r * DXS_LD + off_res + n];
    }
  }
}

#define DX_STAMP(slot)                                                                                     \
  do {                                                                                                     \
    if (tracer && t < DX_TRACE_STEPS) a.trace[t * DX_TRACE_SLOTS + (slot)] = (long long)__builtin_readcyclecounter(); \
  } while (0)

template <int RG>
__global__ __launch_bounds__(DX_NT) void k_decoder_xcd(const DxArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float dx_smem[];
  DxArgs a = a_in;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int Pr = DX_GROUP / RG;          // members per row in the attention phases
  constexpr int DC = DX_W / Pr;              // context channels per member
  const int T = a.T_in, TP = (T + Pr - 1) / Pr, Tpad = (T + 3) & ~3;

  // ---- LDS carve (dx_lds_floats mirrors this) ----
  float* st = dx_smem;
  float* red0 = st + RG * DXS_LD;
  float* red1 = red0 + DX_NW * RG * 16;
  float* Kl = red1 + DX_NW * RG * 16;
  float* Vl = Kl + (size_t)TP * DX_W;
  float* sc = Vl + (size_t)T * DC;
  float* tmp = sc + Tpad;
  float* tmp2 = tmp + Tpad;
  float* alp = tmp2 + Tpad;
  float* qv = alp + Tpad;
  float* vv = qv + DX_W;
  float* bq = vv + DX_W;
  float* cpart = bq + DX_W;
  float* bl = cpart + DX_NW * 64;            // [16 epilogues][16]
  int* ictl = reinterpret_cast<int*>(bl + 16 * 16);

  // ---- census: which XCD am I on, is every XCD hosting exactly one group? ----
  dx_gu32* ctl = (dx_gu32*)a.ctl;
  dx_gu32* errw = (dx_gu32*)a.err;
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 7u;      // HW_REG_XCC_ID (id 20), bits [3:0]
    const unsigned slot = __hip_atomic_fetch_add(ctl + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(ctl + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0; bool ok = true;
    while (__hip_atomic_load(ctl + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (DX_SPIN_LIMIT << 2) || __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }
    }
    bool even = ok;
    for (int i = 0; i < DX_NGROUP; ++i)
      even = even && (__hip_atomic_load(ctl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)DX_GROUP);
    const bool fast = even && !a.force_wt;
    if (!ok) __hip_atomic_store(errw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ictl[0] = fast ? (int)xcc : (int)(blockIdx.x & 7);
    ictl[1] = fast ? (int)slot : (int)(blockIdx.x >> 3);
    ictl[2] = fast ? 0 : 1;
    ictl[3] = ok ? 0 : 1;
    if (blockIdx.x == 0) {   // reported to the host (taco_debug_decoder_info): protocol used, workgroups seen per XCD
      errw[8] = fast ? 1u : 2u;
      for (int i = 0; i < DX_NGROUP; ++i) errw[9 + i] = __hip_atomic_load(ctl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  const int group = __builtin_amdgcn_readfirstlane(ictl[0]);
  const int member = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  const int ga = (group - a.grp0) & 7;
  if (ga >= a.ngroups || member >= DX_GROUP) return;
  const int row0 = ga * RG;
  if (row0 >= a.B) return;
  const int arow = member / Pr, asl = member % Pr;       // attention role: local row, slice
  const int brow = row0 + arow;                          // its batch row (may be >= B: padding)
  const bool tracer = a.trace && ga == 0 && member == 0 && tid == 0;

  // ---- weights: DX_NREG registers per thread, resident for the whole loop ----
  float W[DX_NREG];
  {
    const float* wp = a.wpack + ((size_t)member * DX_NREG) * DX_NT + tid;
#pragma unroll
    for (int j = 0; j < DX_NREG; ++j) W[j] = wp[(size_t)j * DX_NT];
  }
  const DxX xl = dx_xlayout(RG, T);
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)ga * xl.total;
  const int NCF = (a.rM + DX_GROUP - 1) / DX_GROUP;      // frame-projection columns per member (<= 16)

  // ---- stationary attention memory: keys slice (positions), values slice (channels) ----
  const int p0 = asl * TP, pn = max(0, min(T - p0, TP));
  for (int i = tid; i < TP * (DX_W / 4); i += DX_NT) {
    const int p = i / (DX_W / 4), c4 = i % (DX_W / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < pn && brow < a.B) v = *reinterpret_cast<const float4*>(a.keys + ((size_t)brow * T + p0 + p) * DX_W + 4 * c4);
    *reinterpret_cast<float4*>(Kl + (size_t)p * DX_W + 4 * c4) = v;
  }
  for (int i = tid; i < T * (DC / 4); i += DX_NT) {
    const int j = i / (DC / 4), d4 = i % (DC / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (brow < a.B) v = *reinterpret_cast<const float4*>(a.values + ((size_t)brow * T + j) * DX_W + asl * DC + 4 * d4);
    *reinterpret_cast<float4*>(Vl + (size_t)j * DC + 4 * d4) = v;
  }
  // ---- initial state (rnn_wrappers.py:186-216, tacotron.py:183-197): zeros or the deepvoice vectors; step-0 prenet layer 1 =
  // relu(b1) because the go frame and the initial context are zero (helpers.py:70-72) ----
  for (int i = tid; i < RG * DXS_LD; i += DX_NT) st[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < RG * DX_W; i += DX_NT) {
    const int r = i / DX_W, nn = i % DX_W, b = row0 + r;
    if (b < a.B) {
      if (a.h_att0) st[r * DXS_LD + DXS_HATT + nn] = a.h_att0[(size_t)b * DX_W + nn];
      if (a.h10) st[r * DXS_LD + DXS_H1 + nn] = a.h10[(size_t)b * DX_W + nn];
      if (a.h20) st[r * DXS_LD + DXS_H2 + nn] = a.h20[(size_t)b * DX_W + nn];
    }
    st[r * DXS_LD + DXS_T + nn] = fmaxf(a.b_p1_0[nn], 0.f);
  }
  for (int j = tid; j < Tpad; j += DX_NT) { alp[j] = (a.att_type == 2 && j == 0) ? 1.f : 0.f; sc[j] = 0.f; tmp[j] = 0.f; tmp2[j] = 0.f; }
  if (tid < DX_W) { vv[tid] = a.att_v[tid]; bq[tid] = a.att_b ? a.att_b[tid] : 0.f; }
  if (tid < 16 * 16) {     // own-column biases: bl[e][c]
    const int e = tid >> 4, c = tid & 15;
    float v = 0.f;
    const int n8 = member * 8 + (c & 7);
    switch (e) {
      case 0: if (c < 8) v = a.b_p1c[n8]; break;
      case 1: if (c < 4) v = a.b_p2[member * 4 + c]; break;
      case 2: v = a.b_ag[(c < 8 ? 0 : DX_W) + n8]; break;
      case 3: if (c < 8) v = a.b_ac[n8]; break;
      case 4: if (c < 8) v = a.b_cp[n8]; break;
      case 5: v = a.b_g1g[(c < 8 ? 0 : DX_W) + n8]; break;
      case 6: if (c < 8) v = a.b_g1c[n8]; break;
      case 7: v = a.b_g2g[(c < 8 ? 0 : DX_W) + n8]; break;
      case 8: if (c < 8) v = a.b_g2c[n8]; break;
      case 9: if (c < NCF && member * NCF + c < a.rM) v = a.b_f[member * NCF + c]; break;
      default: break;
    }
    bl[tid] = v;
  }
  const float sbias = (a.att_type == 2 && a.score_bias) ? a.score_bias[0] : 0.f;
  __syncthreads();

  // epilogue roles: thread o < RG*8 owns output (r = o>>3, c = o&7) of every 8-column stage
  const int er = tid >> 3, ec = tid & 7;
  const bool epi8 = tid < RG * 8;
  const int en = member * 8 + ec;            // its column in a 256-wide vector
  float g_u = 0.f, g_cx = 0.f, g_h = 0.f;    // GRU gate u, x-part of the candidate, previous state: live between the two stages of a cell

  for (int t = 0; t < a.n; ++t) {
    const unsigned tag = (unsigned)t + 1u;
    DX_STAMP(0);
    // ================= prenet layer 2 (modules.py:18-25) =================
    dx_matvec<DxU_P2, RG, 256>(W, st + DXS_T, st + DXS_T, red0, wave, lane);
    __syncthreads();
    if (tid < RG * 4) {
      const int r = tid >> 2, c = tid & 3;
      const float s = dx_partials<DxU_P2, RG>(red0, r, c) + bl[1 * 16 + c];
      dx_publish(X + xl.p2 + r * DX_P2 + member * 4 + c, fmaxf(s, 0.f), tag, rt);
    }
    dx_gather<RG, DX_P2, false>(X + xl.p2, tag, st, DXS_P2, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(1);
    // ================= attention GRUCell (tacotron.py:127-130; A.6): gates, then candidate =================
    dx_matvec<DxU_AG, RG, 128>(W, st + DXS_P2, st + DXS_HATT, red0, wave, lane);
    dx_matvec<DxU_AX, RG, 128>(W, st + DXS_P2, st + DXS_P2, red1, wave, lane);
    if (epi8) g_h = st[er * DXS_LD + DXS_HATT + en];
    __syncthreads();
    if (epi8) {
      const float rg = taco_sigmoid(dx_partials<DxU_AG, RG>(red0, er, ec) + bl[2 * 16 + ec]);
      g_u = taco_sigmoid(dx_partials<DxU_AG, RG>(red0, er, 8 + ec) + bl[2 * 16 + 8 + ec]);
      g_cx = dx_partials<DxU_AX, RG>(red1, er, ec);
      dx_publish(X + xl.rha + er * DX_W + en, rg * g_h, tag, rt);
    }
    dx_gather<RG, DX_W, false>(X + xl.rha, tag, st, DXS_T, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(2);
    dx_matvec<DxU_AC, RG, 256>(W, st + DXS_T, st + DXS_T, red0, wave, lane);
    __syncthreads();
    if (epi8) {
      const float c = tanhf(g_cx + dx_partials<DxU_AC, RG>(red0, er, ec) + bl[3 * 16 + ec]);
      dx_publish(X + xl.ha + er * DX_W + en, g_u * g_h + (1.f - g_u) * c, tag, rt);
    }
    dx_gather<RG, DX_W, false>(X + xl.ha, tag, st, DXS_HATT, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(3);
    // ================= attention (rnn_wrappers.py:304-341): query, scores, normaliser, context =================
    dx_matvec<DxU_Q, RG, 256>(W, st + DXS_HATT, st + DXS_HATT, red0, wave, lane);
    __syncthreads();
    if (epi8) dx_publish(X + xl.q + er * DX_W + en, dx_partials<DxU_Q, RG>(red0, er, ec), tag, rt);
    if (tid < DX_W) qv[tid] = dx_consume(X + xl.q + arow * DX_W + tid, tag, rt) + bq[tid];   // this member's row only
    __syncthreads();
    DX_STAMP(4);
    {  // scores of the member's block of encoder positions: one wave per position, 4 channels per lane (A.9)
      const float4 q4 = *reinterpret_cast<const float4*>(qv + 4 * lane);
      const float4 v4 = *reinterpret_cast<const float4*>(vv + 4 * lane);
      for (int p = wave; p < pn; p += DX_NW) {
        const float4 k4 = *reinterpret_cast<const float4*>(Kl + (size_t)p * DX_W + 4 * lane);
        float e = v4.x * taco_tanh_fast(k4.x + q4.x) + v4.y * taco_tanh_fast(k4.y + q4.y) + v4.z * taco_tanh_fast(k4.z + q4.z) +
                  v4.w * taco_tanh_fast(k4.w + q4.w);
        e = dx_wave_sum(e);
        if (lane == 0) dx_publish(X + xl.sc + arow * T + p0 + p, e, tag, rt);
      }
    }
    for (int j = tid; j < T; j += DX_NT) sc[j] = dx_consume(X + xl.sc + arow * T + j, tag, rt);
    __syncthreads();
    DX_STAMP(5);
    if (wave == 0) {   // normaliser over the whole row, redundantly on each of the row's members (same code path as att_core)
      const int C = (T + 63) >> 6;
      const int j0 = lane * C, j1 = min(j0 + C, T);
      if (a.att_type == 2) {
        float run = 0.f;
        for (int j = j0; j < j1; ++j) {
          const float p = taco_sigmoid(sc[j] + sbias);
          const float lg = logf(fminf(fmaxf(1.f - p, 1.17549435e-38f), 1.f));
          sc[j] = p; tmp[j] = run; run += lg;
        }
        const float off = wave_scan(run, lane) - run;
        float run2 = 0.f;
        for (int j = j0; j < j1; ++j) {
          const float cp = expf(tmp[j] + off);
          tmp[j] = cp;
          run2 += alp[j] / fminf(fmaxf(cp, 1e-10f), 1.f);
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
    {  // alignment state + history (tacotron.py:238-239 layout) for the member's own positions; context slice
      for (int j = tid; j < T; j += DX_NT) alp[j] = sc[j];
      if (tid < pn && brow < a.B) a.hist[((size_t)brow * T + p0 + tid) * a.n + t] = sc[p0 + tid];
      constexpr int JL = 64 / DC;                    // positions handled side by side inside a wave
      const int d = lane % DC, jsub = lane / DC;
      float part = 0.f;
      for (int j = wave * JL + jsub; j < T; j += DX_NW * JL) part = fmaf(sc[j], Vl[(size_t)j * DC + d], part);
      if (DC <= 32) part += __shfl_xor(part, 32, 64);
      if (DC <= 16) part += __shfl_xor(part, 16, 64);
      if (DC <= 8) part += __shfl_xor(part, 8, 64);
      if (lane < DC) cpart[wave * 64 + lane] = part;
    }
    __syncthreads();
    if (tid < DC) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < DX_NW; ++w) s += cpart[w * 64 + tid];
      dx_publish(X + xl.ctx + arow * DX_W + asl * DC + tid, s, tag, rt);
    }
    dx_gather<RG, DX_W, false>(X + xl.ctx, tag, st, DXS_CTX, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(6);
    // ================= concat projection (rnn_wrappers.py:405-415; tacotron.py:166-170) =================
    dx_matvec<DxU_CP, RG, 512>(W, st + DXS_HATT, st + DXS_HATT, red0, wave, lane);
    __syncthreads();
    if (epi8) dx_publish(X + xl.o0 + er * DX_W + en, dx_partials<DxU_CP, RG>(red0, er, ec) + bl[4 * 16 + ec], tag, rt);
    dx_gather<RG, DX_W, false>(X + xl.o0, tag, st, DXS_O0, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(7);
    // ================= residual GRU 1 (tacotron.py:171-172) =================
    dx_matvec<DxU_G1G, RG, 256>(W, st + DXS_O0, st + DXS_H1, red0, wave, lane);
    dx_matvec<DxU_G1X, RG, 256>(W, st + DXS_O0, st + DXS_O0, red1, wave, lane);
    if (epi8) g_h = st[er * DXS_LD + DXS_H1 + en];
    __syncthreads();
    if (epi8) {
      const float rg = taco_sigmoid(dx_partials<DxU_G1G, RG>(red0, er, ec) + bl[5 * 16 + ec]);
      g_u = taco_sigmoid(dx_partials<DxU_G1G, RG>(red0, er, 8 + ec) + bl[5 * 16 + 8 + ec]);
      g_cx = dx_partials<DxU_G1X, RG>(red1, er, ec);
      dx_publish(X + xl.rh1 + er * DX_W + en, rg * g_h, tag, rt);
    }
    dx_gather<RG, DX_W, false>(X + xl.rh1, tag, st, DXS_T, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(8);
    dx_matvec<DxU_G1C, RG, 256>(W, st + DXS_T, st + DXS_T, red0, wave, lane);
    __syncthreads();
    if (epi8) {
      const float c = tanhf(g_cx + dx_partials<DxU_G1C, RG>(red0, er, ec) + bl[6 * 16 + ec]);
      dx_publish(X + xl.h1 + er * DX_W + en, g_u * g_h + (1.f - g_u) * c, tag, rt);
    }
    dx_gather<RG, DX_W, true>(X + xl.h1, tag, st, DXS_H1, DXS_O0, DXS_OUT1, tid, rt);
    __syncthreads();
    DX_STAMP(9);
    // ================= residual GRU 2 =================
    dx_matvec<DxU_G2G, RG, 256>(W, st + DXS_OUT1, st + DXS_H2, red0, wave, lane);
    dx_matvec<DxU_G2X, RG, 256>(W, st + DXS_OUT1, st + DXS_OUT1, red1, wave, lane);
    if (epi8) g_h = st[er * DXS_LD + DXS_H2 + en];
    __syncthreads();
    if (epi8) {
      const float rg = taco_sigmoid(dx_partials<DxU_G2G, RG>(red0, er, ec) + bl[7 * 16 + ec]);
      g_u = taco_sigmoid(dx_partials<DxU_G2G, RG>(red0, er, 8 + ec) + bl[7 * 16 + 8 + ec]);
      g_cx = dx_partials<DxU_G2X, RG>(red1, er, ec);
      dx_publish(X + xl.rh2 + er * DX_W + en, rg * g_h, tag, rt);
    }
    dx_gather<RG, DX_W, false>(X + xl.rh2, tag, st, DXS_T, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(10);
    dx_matvec<DxU_G2C, RG, 256>(W, st + DXS_T, st + DXS_T, red0, wave, lane);
    __syncthreads();
    if (epi8) {
      const float c = tanhf(g_cx + dx_partials<DxU_G2C, RG>(red0, er, ec) + bl[8 * 16 + ec]);
      dx_publish(X + xl.h2 + er * DX_W + en, g_u * g_h + (1.f - g_u) * c, tag, rt);
    }
    dx_gather<RG, DX_W, true>(X + xl.h2, tag, st, DXS_H2, DXS_OUT1, DXS_OUT2, tid, rt);
    __syncthreads();
    DX_STAMP(11);
    // ================= prenet layer 1 of step t+1 (composite: frame projection folded in, helpers.py:31) and the frame
    // projection of step t (tacotron.py:178-179), straight into the mel buffer =================
    if (t + 1 < a.n) dx_matvec<DxU_P1, RG, 256>(W, st + DXS_OUT2, st + DXS_CTX, red0, wave, lane);
    dx_matvec<DxU_F, RG, 256>(W, st + DXS_OUT2, st + DXS_OUT2, red1, wave, lane);
    __syncthreads();
    if (t + 1 < a.n && epi8)
      dx_publish(X + xl.p1 + er * DX_W + en, fmaxf(dx_partials<DxU_P1, RG>(red0, er, ec) + bl[0 * 16 + ec], 0.f), tag, rt);
    if (tid < RG * 16) {
      const int r = tid >> 4, c = tid & 15, nn = member * NCF + c, b = row0 + r;
      if (c < NCF && nn < a.rM && b < a.B) {
        const float y = dx_partials<DxU_F, RG>(red1, r, c) + bl[9 * 16 + c];
        a.mel[(size_t)b * a.n * a.rM + (size_t)t * a.rM + nn] = y;
        if (y != 0.f) a.nz[(size_t)t * a.B + b] = 1;                    // stop rule helpers.py:29
      }
    }
    if (a.dbg && member == 0) {   // per-step state dump for the stage-level parity test: [h_att | ctx | h1 | h2]
      for (int i = tid; i < RG * 4 * DX_W; i += DX_NT) {
        const int r = i / (4 * DX_W), q = (i / DX_W) & 3, nn = i % DX_W, b = row0 + r;
        const int off = q == 0 ? DXS_HATT : q == 1 ? DXS_CTX : q == 2 ? DXS_H1 : DXS_H2;
        if (b < a.B) a.dbg[((size_t)t * a.B + b) * a.dbgw + q * DX_W + nn] = st[r * DXS_LD + off + nn];
      }
    }
    if (t + 1 < a.n) dx_gather<RG, DX_W, false>(X + xl.p1, tag, st, DXS_T, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(12);
  }
}
