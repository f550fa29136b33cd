// taco_decoder_bwd_xcd.h -- back-propagation through the whole teacher-forced decoder loop (train.py:215-219 -> tf.gradients of
// rnn_wrappers.py:218-341,367-415, tacotron.py:166-181, helpers.py:35-67) as ONE persistent launch: the mirror of k_decoder_xcd.
//
// The launch-per-stage backward is a chain of ~19 dependent launches per decoder step (transposed mat-vecs of k_skinny, the GRU
// element-wise kernels, k_attention_bwd): 11.9 ms of a 29.5 ms training step at the C4 shard (this kernel: 1.7 ms).  Here the step never
// leaves the chip:
// same placement (one XCD = one group of 32 members = RG batch rows), same exchange (8-byte {value, tag} granules through the XCD's
// L2), same pass / reduce / epilogue / publish / gather stages as the forward kernel, with the TRANSPOSED products of BPTT:
// member m owns column 8m + w of every 256-wide gradient vector, wave w of it keeps the matching ROW of every kernel (its 4-input
// slice per lane) in VGPRs for the whole launch.  What a step reads from the tape at the owner's (row, column) -- gates,
// candidates, previous states, prenet outputs, d o2 -- and the rows it needs whole (raw scores, alignments, the processed query) arrive
// one step ahead straight in LDS (global_load_lds_dword: no registers in flight, nothing waited for on the chain).  The state
// gradients dh2, dh1, dh_att and d ctx never leave the lane that owns their column; d alpha (the monotonic recurrence's carry) lives
// in LDS on every member of its row.
//
// Per step t = n-1 .. 0 (twelve exchanges):
//   (do2 = dmel_t . Wf^T: hoisted, one GEMM) GRU 2: d c_pre -> X | d(r*h), d x from Wc^T, gate gradients -> X | d x, dh2 from Wg^T; residual; GRU 1 the same
//   (2 X) | d o0 -> X | concat projection^T: d h_att, d ctx -> X | attention: d alpha partials over the value-channel blocks -> X |
//   normaliser backward (softmax, or the monotonic recurrence with both clips) -> d e; d q partials -> X | d h_att += d q . Wq^T;
//   attention GRU (2 X) -> d p2 -> X | prenet layer 2^T, ReLU mask -> d p1 -> X | prenet layer 1 (context rows)^T -> d ctx(t-1).
// Outputs: the pre-activation gradient tapes the hoisted weight-gradient GEMMs read (decoder_backward in taco_train.h), g_de / g_dq /
// g_dctx for the hoisted key / value gradients, and the gradients of the initial states.
#pragma once
#include "taco_decoder_xcd.h"
#include "taco_backward_kernels.h"

// s_sleep units (64 clocks) in front of the first poll of every gather of the backward loop: as in the forward kernel (DX_FIRST_POLL_DELAY), a poll that
// reaches the L2 ahead of the group's stores costs a second round trip
#ifndef DB_POLL_DELAY
#define DB_POLL_DELAY 5      // measured on the C4-shard step: 0 -> 12.36 ms, 3 -> 12.32, 5 -> 12.29, 7 -> 12.32 (profiles/r06_ab_poll_delay.txt)
#endif
// per-site sleeps as in the forward kernel (DX_DLY): the A/B build -DDX_DLY_RT reads the twelve of this kernel from a second constant table (TACO_DB_DLY).
// tools/sweep_db_delays.py (profiles/r06_sweep_db_delays.txt; the measure is the traced step length): 30 388 clocks per step with every site at 5,
// 29 600-29 700 with the three sites whose curves are clear -- the d q partials (falls monotonically to 0), the d alpha partials, d z2 -- moved; the
// other nine are flat within the noise around 4-6 and keep 5.  DB_DLY_TUNED = 0 restores DB_POLL_DELAY everywhere.
#ifndef DB_DLY_TUNED
#define DB_DLY_TUNED 1
#endif
__host__ __device__ constexpr int db_site_delay(int site) {
  //                      dcp2 dgp2 dcp1 dgp1 do0 dctx da dq dcpa dgpa dz2 dz1
  constexpr int tuned[12] = {5, 5, 5, 5, 5, 5, 1, 0, 5, 5, 2, 5};
  return DB_DLY_TUNED ? tuned[site] : DB_POLL_DELAY;
}
#ifdef DX_DLY_RT
__constant__ int g_db_dly[16];
#define DB_DLY(site) (100 + (site))
#else
#define DB_DLY(site) db_site_delay(site)
#endif

// register map (per thread; host mirror: dbx_build_pack in taco_lib.hip).  A 256-input row = 4 registers (inputs 4l..4l+3), a 512-input
// row = 8 (two halves), a 128-input row = 2.  (The frame projection's data gradient d o2 = dmel . Wf^T does not depend on the
// recurrence: the host computes it for all steps with one GEMM, and the kernel reads its own column of it with the tape values.)
enum {
  DBR_C2X = 0,    // GRU 2 candidate kernel, x row en / h row en (K = 256 each)                4 + 4
  DBR_C2H = 4,
  DBR_G2X = 8,    // GRU 2 gates kernel, x row en / h row en (K = 512)                         8 + 8
  DBR_G2H = 16,
  DBR_C1X = 24, DBR_C1H = 28, DBR_G1X = 32, DBR_G1H = 40,                                    // GRU 1, the same
  DBR_CCA = 48,   // concat projection, h_att row en / context row en (K = 256)                4 + 4
  DBR_CCC = 52,
  DBR_Q = 56,     // query layer row en (K = 256)                                              4
  DBR_CAX = 60,   // attention GRU candidate, x row 4m + w (waves 0-3; K = 256) / h row en     4 + 4
  DBR_CAH = 64,
  DBR_GAX = 68,   // attention GRU gates, x row 4m + w (waves 0-3; K = 512) / h row en         8 + 8
  DBR_GAH = 76,
  DBR_P2 = 84,    // prenet layer 2 row en (K = 128)                                           2
  DBR_P1C = 86,   // prenet layer 1, context row en (K = 256)                                  4
  DB_NREG = 90
};
// per-row gradient vectors in LDS (floats): every gathered vector is read by the stage right behind its gather only, and a
// barrier separates that stage from the gather after the next one, so two alternating 512-float buffers per row hold them all
enum { DBS_DCP2 = 0, DBS_DGP2 = 512, DBS_DCP1 = 0, DBS_DGP1 = 512, DBS_DO0 = 0, DBS_DCTX = 512, DBS_DQ = 0, DBS_DCPA = 512, DBS_DGPA = 0,
       DBS_DZ2 = 512, DBS_DZ1 = 0, DBS_LD = 1024 };
struct DbX { int dcp2, dgp2, dcp1, dgp1, do0, dctx, da, dq, dcpa, dgpa, dz2, dz1, total; };
__host__ __device__ inline DbX db_xlayout(int RG, int T_in) {
  const int Pr = DX_GROUP / RG, Pc = Pr < 8 ? Pr : 8, Pp = Pr / Pc;
  DbX x; int o = 0;
  x.dcp2 = o; o += RG * 256; x.dgp2 = o; o += RG * 512; x.dcp1 = o; o += RG * 256; x.dgp1 = o; o += RG * 512;
  x.do0 = o; o += RG * 256; x.dctx = o; o += RG * 256;
  x.da = o; o += DX_GROUP * T_in;                  // [row][member of the row][position]
  x.dq = o; o += RG * Pp * 256;                    // [row][position block][channel]
  x.dcpa = o; o += RG * 256; x.dgpa = o; o += RG * 512; x.dz2 = o; o += RG * 128; x.dz1 = o; o += RG * 256;
  x.total = o;
  return x;
}
__host__ __device__ inline size_t db_lds_floats(int RG, int T_in) {
  const int Pr = DX_GROUP / RG, DC = DX_W / Pr, Tpad = (T_in + 63) & ~63;
  const int Pc = Pr < 8 ? Pr : 8, Pp = Pr / Pc, DS = DX_W / Pc, TS = (T_in + Pp - 1) / Pp;
  size_t n = (size_t)RG * DBS_LD;
  n += 2 * (size_t)DX_NW * 128;                    // own-column tape values of the step, double buffered
  n += (size_t)TS * DS + (size_t)T_in * DC;        // keys block, values block
  n += 2 * 3 * (size_t)Tpad;                       // raw scores, alignments of the step and of the step before, double buffered
  n += 6 * (size_t)Tpad;                           // da, p, cp, ss, de, d alpha carry
  n += 5 * 64 + (size_t)DX_NW * 64 + 64;           // q + b, v, b, the tape's q rows (double buffered); reduction partials; control words
  return n;
}

struct DbArgs {
  const float* wpack;                                  // [32 members][DB_NREG][DX_NT]
  const float* tape; size_t tstride;                   // the forward's 256-wide per-step arrays (DXT_* slots)
  const float* tp_p2; int ld_p2;                       // prenet output [B, n, ld_p2]
  const float* tp_e; const float* tp_alpha;            // raw scores [B, n, T_in]; alignments [B, n + 1, T_in]
  const float* keys; const float* values;              // [B, T_in, 256]
  const float* att_v; const float* att_b; const float* score_bias;
  const float* g_do2;                                  // [B, n, 256] = dmel . Wf^T (one GEMM ahead of the launch)
  const float* h_att0; const float* h10; const float* h20;   // initial states [B, 256] or null
  float* g_dcp2; float* g_dgp2; float* g_dcp1; float* g_dgp1; float* g_do0; float* g_dcpA; float* g_dgpA;   // [R, 256] / [R, 512]
  float* g_dz1; float* g_dz2; float* g_dq; float* g_de; float* g_dctx;                                     // [R, 256], [R, 128], [R, 256], [R, T_in], [R, 256]
  float* d_att_init; float* d_h10; float* d_h20;       // [B, 256] or null
  float* dsb_acc;                                      // [B] or null: d attention_score_bias per row (summed by the host)
  unsigned long long* xbuf; unsigned* ctl; unsigned* err; long long* trace;
  int B, T_in, n, att_type, force_wt;
};

// K = 512 pass: two 256-halves of the same LD-strided row
template <int REG0, int NCOLS, int RG, int NW, int LD>
__device__ __forceinline__ void db_pass512(const float (&W)[NW], const float* x, int lane, float (&acc)[NCOLS][RG]) {
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const float4 a = *reinterpret_cast<const float4*>(x + r * LD + 4 * lane);
    const float4 b = *reinterpret_cast<const float4*>(x + r * LD + 256 + 4 * lane);
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      acc[c][r] = fmaf(W[REG0 + 8 * c + 0], a.x, acc[c][r]); acc[c][r] = fmaf(W[REG0 + 8 * c + 1], a.y, acc[c][r]);
      acc[c][r] = fmaf(W[REG0 + 8 * c + 2], a.z, acc[c][r]); acc[c][r] = fmaf(W[REG0 + 8 * c + 3], a.w, acc[c][r]);
      acc[c][r] = fmaf(W[REG0 + 8 * c + 4], b.x, acc[c][r]); acc[c][r] = fmaf(W[REG0 + 8 * c + 5], b.y, acc[c][r]);
      acc[c][r] = fmaf(W[REG0 + 8 * c + 6], b.z, acc[c][r]); acc[c][r] = fmaf(W[REG0 + 8 * c + 7], b.w, acc[c][r]);
    }
  }
}
// inclusive SUFFIX sum over the wave (lane l: sum over lanes >= l) on DPP: row_shl inside the rows of 16, the rows above added as
// wave-uniform scalars.  Never total - prefix: the tail of these sums is many orders below the head (cumprod(1 - p)).
__device__ __forceinline__ float db_rscan(float v) {
  float t = v + DX_DPPZ(v, 0x101, 0xF, 0xF);
  t += DX_DPPZ(v, 0x102, 0xF, 0xF);
  t += DX_DPPZ(v, 0x103, 0xF, 0xF);
  t += DX_DPPZ(t, 0x104, 0xF, 0xF);
  t += DX_DPPZ(t, 0x108, 0xF, 0xF);
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 16)), r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 32)),
              r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 48));
  const int row = (int)(threadIdx.x & 63) >> 4;
  return t + (row == 0 ? r1 + r2 + r3 : row == 1 ? r2 + r3 : row == 2 ? r3 : 0.f);
}
// Normaliser backward of one row by ONE wave in registers (the mirror of dx_normalise: lane = `cnt` <= CMAX consecutive positions from
// j0): da = gradient of the step's alignments -> de = gradient of the raw scores, dac = gradient of the previous alignments (the
// carry of the recurrence; zero for softmax).  Returns the row's d score_bias.  Arithmetic of k_attention_bwd (through safe_cumprod /
// cumsum and both clips, as tf.gradients does) on the forward kernel's intrinsics.
template <int CMAX>
__device__ __forceinline__ float db_normalise_bwd(const float* er, const float* alp, const float* al, const float* da, float* de, float* dac,
                                                  int j0, int cnt, int att_type, float sbias) {
  float dav[CMAX];
#pragma unroll
  for (int i = 0; i < CMAX; ++i) dav[i] = i < cnt ? da[j0 + i] : 0.f;
  if (att_type != 2) {   // softmax: de_j = alpha_j (da_j - sum_k alpha_k da_k)
    float av[CMAX], dot = 0.f;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) { av[i] = i < cnt ? al[j0 + i] : 0.f; dot = fmaf(av[i], dav[i], dot); }
    dot = dx_allsum(dot);
#pragma unroll
    for (int i = 0; i < CMAX; ++i) if (i < cnt) { de[j0 + i] = av[i] * (dav[i] - dot); dac[j0 + i] = 0.f; }
    return 0.f;
  }
  float p[CMAX], pv[CMAX], ex[CMAX], run = 0.f;
#pragma unroll
  for (int i = 0; i < CMAX; ++i) {
    p[i] = dx_sigmoid_fast((i < cnt ? er[j0 + i] : 0.f) + sbias);
    pv[i] = i < cnt ? alp[j0 + i] : 0.f;
    ex[i] = run;
    if (i < cnt) run += 0.6931471805599453f * __builtin_amdgcn_logf(fminf(fmaxf(1.f - p[i], 1.17549435e-38f), 1.f));
  }
  const float off = dx_scan(run) - run;
  float cp[CMAX], rc[CMAX], ss[CMAX], run2 = 0.f;
#pragma unroll
  for (int i = 0; i < CMAX; ++i) {
    cp[i] = __builtin_amdgcn_exp2f(1.4426950408889634f * (ex[i] + off));
    rc[i] = __builtin_amdgcn_rcpf(fminf(fmaxf(cp[i], 1e-10f), 1.f));
    if (i < cnt) run2 = fmaf(pv[i], rc[i], run2);
    ss[i] = run2;
  }
  const float off2 = dx_scan(run2) - run2;
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < CMAX; ++i) { ss[i] += off2; if (i < cnt) tot = fmaf(dav[i] * p[i], cp[i], tot); }
  float suffix = db_rscan(tot) - tot;             // the later lanes' blocks
  float dL[CMAX], dpd[CMAX], dLsum = 0.f;
#pragma unroll
  for (int i = CMAX - 1; i >= 0; --i) {
    dL[i] = 0.f; dpd[i] = 0.f;
    if (i < cnt) {
      suffix = fmaf(dav[i] * p[i], cp[i], suffix);
      float dcp = dav[i] * p[i] * ss[i];
      if (cp[i] >= 1e-10f && cp[i] <= 1.f) dcp -= suffix * pv[i] * rc[i] * rc[i];
      dL[i] = dcp * cp[i];
      dpd[i] = dav[i] * cp[i] * ss[i];
      dac[j0 + i] = suffix * rc[i];
      dLsum += dL[i];
    }
  }
  float suf2 = db_rscan(dLsum) - dLsum, dsb = 0.f;
#pragma unroll
  for (int i = CMAX - 1; i >= 0; --i) {
    if (i < cnt) {
      const float dlg = suf2;                     // reverse EXCLUSIVE sum of dL
      suf2 += dL[i];
      const float xj = 1.f - p[i];
      float dp = dpd[i];
      if (xj >= 1.17549435e-38f && xj <= 1.f) dp -= dlg * __builtin_amdgcn_rcpf(xj);
      const float dej = dp * p[i] * xj;
      de[j0 + i] = dej; dsb += dej;
    }
  }
  return dx_allsum(dsb);
}

// WT: the census chose the write-through protocol (decided once per launch: the step carries no protocol branch); TRACE: shader-clock stamps
// (tools/trace_bptt.py) -- the production instantiation has none of their branches (round 5: ~30 clocks per publish / stamp on the chain)
template <int RG, bool WT, bool TRACE>
__device__ __forceinline__ void db_body(const DbArgs& a, float* dx_smem, int group, int member, DxRt rt) {
  constexpr int WTC = WT ? 1 : 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int Pr = DX_GROUP / RG, DC = DX_W / Pr;
  constexpr int Pc = Pr < 8 ? Pr : 8, Pp = Pr / Pc, DS = DX_W / Pc;
  const int T = a.T_in, Tpad = (T + 63) & ~63;
  const int TP = (T + Pr - 1) / Pr, TS = (T + Pp - 1) / Pp;
  const int n = a.n;

  // ---- LDS carve (db_lds_floats mirrors this) ----
  float* st = dx_smem;                        // [RG][DBS_LD]
  float* own = st + RG * DBS_LD;              // [2][DX_NW][128] own-column tape values of the step
  float* Kc = own + 2 * DX_NW * 128;          // keys   [TS][DS]
  float* Vc = Kc + (size_t)TS * DS;           // values [T][DC]
  float* rows = Vc + (size_t)T * DC;          // [2][3][Tpad]: e, alpha(t+1 slot = this step's), alpha(t slot = previous)
  float* da = rows + 6 * Tpad;
  float* pp = da + Tpad; float* cp = pp + Tpad; float* ss = cp + Tpad; float* de = ss + Tpad; float* dac = de + Tpad;
  float* qv = dac + Tpad; float* vv = qv + 64; float* bq = vv + 64;
  float* qraw = bq + 64;                      // [2][64] processed query of the member's channels (tape), one step ahead
  float* cpart = qraw + 128;                  // [DX_NW][64]
  if (member >= DX_GROUP) return;
  const int row0 = group * RG;
  if (row0 >= a.B) return;
  const int arow = member / Pr, asl = member % Pr;
  const int cb = asl % Pc, pb = asl / Pc;
  const int ps0 = pb * TS, psn = max(0, min(T - ps0, TS));
  const int brow = row0 + arow;
  const int browc = min(brow, a.B - 1);
  const bool tracer = TRACE && a.trace && group == 0 && member == 0 && tid == 0;
#define DB_STAMP(slot)                                                                                                   \
  do {                                                                                                                   \
    const int ks_ = n - 1 - t - 8;         /* steps 8 .. 15 of the launch: past the start-up transients */                \
    if constexpr (TRACE) { if (tracer && ks_ >= 0 && ks_ < DX_TRACE_STEPS) a.trace[ks_ * DX_TRACE_SLOTS + (slot)] = (long long)__builtin_readcyclecounter(); } \
  } while (0)

  float W[DB_NREG];
  {
    const float* wp = a.wpack + ((size_t)member * DB_NREG) * DX_NT + tid;
#pragma unroll
    for (int j = 0; j < DB_NREG; ++j) W[j] = wp[(size_t)j * DX_NT];
  }
  const DbX xl = db_xlayout(RG, T);
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)group * xl.total;

  // stationary attention memory of the member's row (as in the forward kernel)
  for (int i = tid; i < TS * (DS / 4); i += DX_NT) {
    const int j = i / (DS / 4), d4 = i % (DS / 4);
    float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (brow < a.B && j < psn) k4 = *reinterpret_cast<const float4*>(a.keys + ((size_t)brow * T + ps0 + j) * DX_W + cb * DS + 4 * d4);
    *reinterpret_cast<float4*>(Kc + (size_t)j * DS + 4 * d4) = k4;
  }
  for (int i = tid; i < T * (DC / 4); i += DX_NT) {
    const int j = i / (DC / 4), d4 = i % (DC / 4);
    float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (brow < a.B) v4 = *reinterpret_cast<const float4*>(a.values + ((size_t)brow * T + j) * DX_W + asl * DC + 4 * d4);
    *reinterpret_cast<float4*>(Vc + (size_t)j * DC + 4 * d4) = v4;
  }
  for (int i = tid; i < RG * DBS_LD; i += DX_NT) st[i] = 0.f;
  for (int i = tid; i < 2 * DX_NW * 128; i += DX_NT) own[i] = 0.f;
  for (int j = tid; j < Tpad; j += DX_NT) { da[j] = 0.f; pp[j] = 0.f; cp[j] = 0.f; ss[j] = 0.f; de[j] = 0.f; dac[j] = 0.f; }
  for (int j = tid; j < 6 * Tpad; j += DX_NT) rows[j] = 0.f;
  if (tid < DS) { vv[tid] = a.att_v[cb * DS + tid]; bq[tid] = a.att_b ? a.att_b[cb * DS + tid] : 0.f; }
  if (tid < 128) qraw[tid] = 0.f;
  const float sbias = (a.att_type == 2 && a.score_bias) ? a.score_bias[0] : 0.f;
  __syncthreads();

  constexpr int RL = DxRL<RG>::value;
  const bool epl = lane < (RG >= 4 ? 4 : RG);
  const int en = member * 8 + wave, en2 = member * 4 + wave;     // own column of 256-wide / (waves 0-3) of 128-wide vectors
  int erow[RL];
  bool ev[RL];
  unsigned trow[RL];                                             // (row * n) : element offset of step 0 in a [B, n, W] array is trow * W
#pragma unroll
  for (int q = 0; q < RL; ++q) {
    erow[q] = dx_row<RG>(lane & 3, q);
    ev[q] = epl && (row0 + erow[q] < a.B);
    trow[q] = (unsigned)min(row0 + erow[q], a.B - 1) * (unsigned)n;
  }
  // Own-column tape values of a step -- u, c, r and the previous state of the three cells, both prenet outputs, for the RG rows of
  // the wave's column: OW_N * RG scalars per wave -- are gathered one step ahead by ONE LDS-direct load per 64 of them (each lane
  // supplies the address of its (value, row) pair), so neither the values in flight nor the ones in use occupy registers.
  enum { OW_U2 = 0, OW_C2, OW_R2, OW_H2P, OW_U1, OW_C1, OW_R1, OW_H1P, OW_UA, OW_CA, OW_RA, OW_HAP, OW_P1, OW_P2, OW_DO2, OW_N };
  constexpr int OW_K = (OW_N * RG + 63) / 64;
  const float* gp[OW_K]; const float* galt[OW_K]; int gstr[OW_K]; bool gprev[OW_K], gok[OW_K];
#pragma unroll
  for (int k = 0; k < OW_K; ++k) {
    const int i = lane + 64 * k, v = i / RG, b = min(row0 + i % RG, a.B - 1);
    gok[k] = i < OW_N * RG && (v != OW_P2 || wave < 4);
    gprev[k] = v == OW_H2P || v == OW_H1P || v == OW_HAP;
    int slot = DXT_P1;
    switch (v) {
      case OW_U2: slot = DXT_U2; break; case OW_C2: slot = DXT_C2; break; case OW_R2: slot = DXT_R2; break; case OW_H2P: slot = DXT_H2; break;
      case OW_U1: slot = DXT_U1; break; case OW_C1: slot = DXT_C1; break; case OW_R1: slot = DXT_R1; break; case OW_H1P: slot = DXT_H1; break;
      case OW_UA: slot = DXT_UA; break; case OW_CA: slot = DXT_CA; break; case OW_RA: slot = DXT_RA; break; case OW_HAP: slot = DXT_HA; break;
      default: break;
    }
    gp[k] = a.tape + (size_t)slot * a.tstride + (size_t)b * n * DX_W + en - (gprev[k] ? DX_W : 0);    // + t * 256: step t (previous state: step t - 1)
    gstr[k] = DX_W;
    if (v == OW_P2) { gp[k] = a.tp_p2 + (size_t)b * n * a.ld_p2 + (wave < 4 ? en2 : 0); gstr[k] = a.ld_p2; }
    if (v == OW_DO2) gp[k] = a.g_do2 + (size_t)b * n * DX_W + en;
    const float* i0 = v == OW_H2P ? a.h20 : v == OW_H1P ? a.h10 : a.h_att0;                           // the state before step 0
    galt[k] = (gprev[k] && i0) ? i0 + (size_t)b * DX_W + en : gp[k] + gstr[k];                         // (none: any valid address; the consumer reads 0)
  }
  const unsigned own_lds = (unsigned)(size_t)(dx_lds_float*)own;
  auto fetch_own = [&](int t, int buf) {
#pragma unroll
    for (int k = 0; k < OW_K; ++k)
      if (gok[k]) dx_load_lds4((t == 0 && gprev[k]) ? galt[k] : gp[k] + (size_t)t * gstr[k],
                               __builtin_amdgcn_readfirstlane(own_lds + (unsigned)((buf * DX_NW + wave) * 128 + 64 * k) * 4u));
  };
  // rows needed whole, one step ahead, straight into LDS: e_t / alpha_{t+1} / alpha_t of the member's row, the query of its channels
  const unsigned rows_lds = (unsigned)(size_t)(dx_lds_float*)rows;
  const unsigned qraw_lds = (unsigned)(size_t)(dx_lds_float*)qraw;
  auto fetch_rows = [&](int t, int buf) {
    if (wave == DX_NW - 1 && lane < DS)       // the processed query W_q h_att(t) of the member's score channels
      dx_load_lds4(a.tape + (size_t)DXT_Q * a.tstride + ((size_t)browc * n + t) * DX_W + cb * DS + lane,
                   __builtin_amdgcn_readfirstlane(qraw_lds + (unsigned)(buf * 64) * 4u));
    const float* se = a.tp_e + ((size_t)browc * n + t) * T;
    const float* sa = a.tp_alpha + ((size_t)browc * (n + 1) + t + 1) * T;
    const float* sp = a.tp_alpha + ((size_t)browc * (n + 1) + t) * T;
    for (int j0 = wave * 64; j0 < T; j0 += DX_NT) {
      const int j = min(j0 + lane, T - 1);
      dx_load_lds4(se + j, __builtin_amdgcn_readfirstlane(rows_lds + (unsigned)((buf * 3 + 0) * Tpad + j0) * 4u));
      dx_load_lds4(sa + j, __builtin_amdgcn_readfirstlane(rows_lds + (unsigned)((buf * 3 + 1) * Tpad + j0) * 4u));
      dx_load_lds4(sp + j, __builtin_amdgcn_readfirstlane(rows_lds + (unsigned)((buf * 3 + 2) * Tpad + j0) * 4u));
    }
  };
  fetch_own(n - 1, 0);
  fetch_rows(n - 1, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float dh2[RL], dh1[RL], dhA[RL], dctxc[RL];      // carried gradients of the owner's column
#pragma unroll
  for (int q = 0; q < RL; ++q) { dh2[q] = 0.f; dh1[q] = 0.f; dhA[q] = 0.f; dctxc[q] = 0.f; }
  float dsb_row = 0.f;

  // (Round 6, measured: s_setprio 1 for waves 0-3, which buys k_decoder_xcd 1 %, changes nothing here: 11.64 / 11.65 against 11.65 / 11.62 ms per step.)
  const int tid_outer = tid, lane_outer = lane;
  for (int t = n - 1; t >= 0; --t) {
    const unsigned tag = (unsigned)(n - 1 - t) + 1u;
    const int buf = (n - 1 - t) & 1;
    int tid = tid_outer, lane = lane_outer;
    asm volatile("" : "+v"(tid), "+v"(lane));
    // (the owner's rows, their validity and tape offsets are formed again from the opaque lane: derived before the loop, the 64-bit
    // addresses of the eleven gradient tapes x the lane's rows are loop invariants -- 22 pointers at eight rows per group -- and spill)
    int erow[RL]; bool ev[RL]; unsigned trow[RL];
#pragma unroll
    for (int q = 0; q < RL; ++q) {
      erow[q] = dx_row<RG>(lane & 3, q);
      ev[q] = (lane < (RG >= 4 ? 4 : RG)) && (row0 + erow[q] < a.B);
      trow[q] = (unsigned)min(row0 + erow[q], a.B - 1) * (unsigned)n;
    }
    DB_STAMP(0);
    if (t > 0) { fetch_own(t - 1, buf ^ 1); fetch_rows(t - 1, buf ^ 1); }
    const float* ow = own + (buf * DX_NW + wave) * 128;
#define OWN(v, q) ow[(v) * RG + erow[q]]
    const bool z2 = t == 0 && !a.h20, z1 = t == 0 && !a.h10, zA = t == 0 && !a.h_att0;   // zero initial state (rnn_wrappers.py:186-216)
    const float* er = rows + (size_t)(buf * 3 + 0) * Tpad;     // raw scores of step t
    const float* al = rows + (size_t)(buf * 3 + 1) * Tpad;     // alignments of step t
    const float* alp = rows + (size_t)(buf * 3 + 2) * Tpad;    // alignments of step t - 1 (slot 0: the initial alignments)
#define DB_OUT(ptr, Wd, q, col, val) do { if (ev[q]) (ptr)[(size_t)(trow[q] + (unsigned)t) * (Wd) + (col)] = (val); } while (0)
    // a GRU cell's backward, part 'a' (owner-local): from the gradient of the cell's output and the carried state gradient
    float dht[RL], dgu[RL], tx[RL], do_[RL];
    // ================= GRU 2 'a' (d o2 = dmel . Wf^T: precomputed for all steps) =================
    {
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        do_[q] = OWN(OW_DO2, q);                                 // d o2 (GRU stack output)
        const float g = do_[q] + dh2[q];
        const float dcp = g * (1.f - OWN(OW_U2, q)) * (1.f - OWN(OW_C2, q) * OWN(OW_C2, q));
        dgu[q] = g * ((z2 ? 0.f : OWN(OW_H2P, q)) - OWN(OW_C2, q)) * OWN(OW_U2, q) * (1.f - OWN(OW_U2, q));
        dht[q] = g;
        if (epl) dx_publish<WTC>(X + xl.dcp2 + erow[q] * 256 + en, dcp, tag, rt);
        DB_OUT(a.g_dcp2, 256, q, en, dcp); DB_OUT(a.g_dgp2, 512, q, 256 + en, dgu[q]);
      }
    }
    dx_gather<RG, 256, false, DBS_LD, DX_NT, DB_DLY(0)>(X + xl.dcp2, tag, st, DBS_DCP2, 0, 0, tid, rt);
    __syncthreads();
    DB_STAMP(1);
    // a cell's parts 'b' and 'c' as two stages; CX/CH/GX/GH: register bases, VC/VG: LDS vectors, XG: gate-gradient exchange
#define DB_CELL_B(CX, CH, VC, XG, HP, RR, UU, DHP, GOUT)                                                     \
    {                                                                                                        \
      float acc[2][RG], s[2][RL];                                                                            \
      dx_zero<2, RG>(acc);                                                                                   \
      dx_pass<CX, 1, RG, DB_NREG, DBS_LD>(W, st + VC, lane, reinterpret_cast<float (&)[1][RG]>(acc[0]));    \
      dx_pass<CH, 1, RG, DB_NREG, DBS_LD>(W, st + VC, lane, reinterpret_cast<float (&)[1][RG]>(acc[1]));    \
      dx_reduce<2, RG>(acc, s, lane);                                                                        \
      _Pragma("unroll") for (int q = 0; q < RL; ++q) {                                                       \
        tx[q] = s[0][q];                                                                                     \
        const float drh = s[1][q];                                                                           \
        const float dgr = drh * (HP) * (RR) * (1.f - (RR));                                                  \
        DHP[q] = dht[q] * (UU) + drh * (RR);                                                                 \
        if (epl) { dx_publish<WTC>(X + (XG) + erow[q] * 512 + en, dgr, tag, rt); dx_publish<WTC>(X + (XG) + erow[q] * 512 + 256 + en, dgu[q], tag, rt); } \
        DB_OUT(GOUT, 512, q, en, dgr);                                                                       \
      }                                                                                                      \
    }
    float dhp[RL];
    // ================= GRU 2 'b' =================
    DB_CELL_B(DBR_C2X, DBR_C2H, DBS_DCP2, xl.dgp2, (z2 ? 0.f : OWN(OW_H2P, q)), OWN(OW_R2, q), OWN(OW_U2, q), dhp, a.g_dgp2)
    dx_gather<RG, 512, false, DBS_LD, DX_NT, DB_DLY(1)>(X + xl.dgp2, tag, st, DBS_DGP2, 0, 0, tid, rt);
    __syncthreads();
    DB_STAMP(2);
    // ================= GRU 2 'c' -> residual -> GRU 1 'a' =================
    {
      float acc[2][RG], s[2][RL];
      dx_zero<2, RG>(acc);
      db_pass512<DBR_G2X, 1, RG, DB_NREG, DBS_LD>(W, st + DBS_DGP2, lane, reinterpret_cast<float (&)[1][RG]>(acc[0]));
      db_pass512<DBR_G2H, 1, RG, DB_NREG, DBS_LD>(W, st + DBS_DGP2, lane, reinterpret_cast<float (&)[1][RG]>(acc[1]));
      dx_reduce<2, RG>(acc, s, lane);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        dh2[q] = dhp[q] + s[1][q];
        do_[q] = tx[q] + s[0][q] + do_[q];                       // d o1 = d x of GRU 2 + residual (o2 = h2 + o1)
        const float g = do_[q] + dh1[q];
        const float dcp = g * (1.f - OWN(OW_U1, q)) * (1.f - OWN(OW_C1, q) * OWN(OW_C1, q));
        dgu[q] = g * ((z1 ? 0.f : OWN(OW_H1P, q)) - OWN(OW_C1, q)) * OWN(OW_U1, q) * (1.f - OWN(OW_U1, q));
        dht[q] = g;
        if (epl) dx_publish<WTC>(X + xl.dcp1 + erow[q] * 256 + en, dcp, tag, rt);
        DB_OUT(a.g_dcp1, 256, q, en, dcp); DB_OUT(a.g_dgp1, 512, q, 256 + en, dgu[q]);
      }
    }
    dx_gather<RG, 256, false, DBS_LD, DX_NT, DB_DLY(2)>(X + xl.dcp1, tag, st, DBS_DCP1, 0, 0, tid, rt);
    __syncthreads();
    DB_STAMP(3);
    // ================= GRU 1 'b' =================
    DB_CELL_B(DBR_C1X, DBR_C1H, DBS_DCP1, xl.dgp1, (z1 ? 0.f : OWN(OW_H1P, q)), OWN(OW_R1, q), OWN(OW_U1, q), dhp, a.g_dgp1)
    dx_gather<RG, 512, false, DBS_LD, DX_NT, DB_DLY(3)>(X + xl.dgp1, tag, st, DBS_DGP1, 0, 0, tid, rt);
    __syncthreads();
    DB_STAMP(4);
    // ================= GRU 1 'c' -> d o0 =================
    {
      float acc[2][RG], s[2][RL];
      dx_zero<2, RG>(acc);
      db_pass512<DBR_G1X, 1, RG, DB_NREG, DBS_LD>(W, st + DBS_DGP1, lane, reinterpret_cast<float (&)[1][RG]>(acc[0]));
      db_pass512<DBR_G1H, 1, RG, DB_NREG, DBS_LD>(W, st + DBS_DGP1, lane, reinterpret_cast<float (&)[1][RG]>(acc[1]));
      dx_reduce<2, RG>(acc, s, lane);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        dh1[q] = dhp[q] + s[1][q];
        const float do0 = tx[q] + s[0][q] + do_[q];              // d o0 = d x of GRU 1 + residual (o1 = h1 + o0)
        if (epl) dx_publish<WTC>(X + xl.do0 + erow[q] * 256 + en, do0, tag, rt);
        DB_OUT(a.g_do0, 256, q, en, do0);
      }
    }
    dx_gather<RG, 256, false, DBS_LD, DX_NT, DB_DLY(4)>(X + xl.do0, tag, st, DBS_DO0, 0, 0, tid, rt);
    __syncthreads();
    DB_STAMP(5);
    // ================= concat projection^T: d h_att (kept), d ctx -> exchange =================
    float dIn_hA[RL];
    {
      float acc[2][RG], s[2][RL];
      dx_zero<2, RG>(acc);
      dx_pass<DBR_CCA, 1, RG, DB_NREG, DBS_LD>(W, st + DBS_DO0, lane, reinterpret_cast<float (&)[1][RG]>(acc[0]));
      dx_pass<DBR_CCC, 1, RG, DB_NREG, DBS_LD>(W, st + DBS_DO0, lane, reinterpret_cast<float (&)[1][RG]>(acc[1]));
      dx_reduce<2, RG>(acc, s, lane);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        dIn_hA[q] = s[0][q];
        const float dc = dctxc[q] + s[1][q];                     // total gradient of context(t)
        if (epl) dx_publish<WTC>(X + xl.dctx + erow[q] * 256 + en, dc, tag, rt);
        DB_OUT(a.g_dctx, 256, q, en, dc);
      }
    }
    dx_gather<RG, 256, false, DBS_LD, DX_NT, DB_DLY(5)>(X + xl.dctx, tag, st, DBS_DCTX, 0, 0, tid, rt);
    __syncthreads();
    DB_STAMP(6);
    // ================= attention backward of the member's row =================
    {  // d alpha partial over the member's value-channel block: sum_c dctx[asl*DC + c] * V[j][c], lanes over positions
      const float* dcx = st + arow * DBS_LD + DBS_DCTX + asl * DC;
      for (int j = tid; j < T; j += DX_NT) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DC; c += 4) {
          const float4 v4 = *reinterpret_cast<const float4*>(Vc + (size_t)j * DC + c);
          s += v4.x * dcx[c] + v4.y * dcx[c + 1] + v4.z * dcx[c + 2] + v4.w * dcx[c + 3];
        }
        dx_publish<WTC>(X + xl.da + (size_t)(arow * Pr + asl) * T + j, s, tag, rt);
      }
    }
    {  // gather the row's partials (fixed order) + the carried d alpha
      constexpr int NPQ = Pr >= 4 ? Pr / 4 : 1;                  // partials per lane: a quad covers the Pr members (Pr >= 4), else one lane all
      const int jl = lane >> 2, part = lane & 3;
      for (int j0 = 0; j0 < T; j0 += 16 * DX_NW) {
        const int j = j0 + wave * 16 + jl;
        float s = 0.f;
        if (j < T && (Pr >= 4 || part < Pr)) {
          float v[NPQ];
          dx_poll<NPQ, DB_DLY(6)>(X + xl.da + (size_t)(arow * Pr + (Pr >= 4 ? part * NPQ : part)) * T + j, (size_t)T, tag, v, rt);
#pragma unroll
          for (int u = 0; u < NPQ; ++u) s += v[u];
        }
        s = dx_quadsum(s);
        if (part == 0 && j < T) da[j] = s + dac[j];
      }
    }
    __syncthreads();
    DB_STAMP(7);
    if (wave == 0) {   // normaliser backward, redundantly on each member of the row
      const int C = (T + 63) >> 6, j0 = lane * C, j1 = min(j0 + C, T);
      if (C <= 2) dsb_row += db_normalise_bwd<2>(er, alp, al, da, de, dac, j0, max(j1 - j0, 0), a.att_type, sbias);
      else if (C <= 4) dsb_row += db_normalise_bwd<4>(er, alp, al, da, de, dac, j0, max(j1 - j0, 0), a.att_type, sbias);
      else if (a.att_type == 2) {   // long inputs: through LDS scratch (the arithmetic of k_attention_bwd as it stands)
        float run = 0.f;
        for (int j = j0; j < j1; ++j) {
          const float pj = taco_sigmoid(er[j] + sbias);
          pp[j] = pj; cp[j] = run;
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
        float tot = 0.f;
        for (int j = j0; j < j1; ++j) tot += da[j] * pp[j] * cp[j];
        float suffix = wave_rscan(tot, lane) - tot;
        float dLsum = 0.f;
        for (int j = j1 - 1; j >= j0; --j) {
          suffix += da[j] * pp[j] * cp[j];
          const float cj = fminf(fmaxf(cp[j], 1e-10f), 1.f);
          float dcp_ = da[j] * pp[j] * ss[j];
          if (cp[j] >= 1e-10f && cp[j] <= 1.f) dcp_ += -suffix * alp[j] / (cj * cj);
          const float dL = dcp_ * cp[j];
          ss[j] = da[j] * cp[j] * ss[j];
          cp[j] = dL;
          dac[j] = suffix / cj;                                  // d alpha(t-1): the carry of the next (earlier) step
          dLsum += dL;
        }
        float suf2 = wave_rscan(dLsum, lane) - dLsum;
        float dsb = 0.f;
        for (int j = j1 - 1; j >= j0; --j) {
          const float dlg = suf2;
          suf2 += cp[j];
          const float xj = 1.f - pp[j];
          float dp = ss[j];
          if (xj >= 1.17549435e-38f && xj <= 1.f) dp -= dlg / xj;
          const float dej = dp * pp[j] * (1.f - pp[j]);
          de[j] = dej; dsb += dej;
        }
        dsb = wave_sum(dsb);
        dsb_row += dsb;
      } else {
        float dot = 0.f;
        for (int j = j0; j < j1; ++j) dot += al[j] * da[j];
        dot = wave_sum(dot);
        for (int j = j0; j < j1; ++j) { de[j] = al[j] * (da[j] - dot); dac[j] = 0.f; }
      }
    }
    // the processed query of the member's score channels (+ attention_b): from the tape, fetched one step ahead
    if (tid < DS) qv[tid] = qraw[buf * 64 + tid] + bq[tid];
    __syncthreads();
    DB_STAMP(8);
    {
      const int p0 = asl * TP;
      if (tid < TP && p0 + tid < T && brow < a.B) a.g_de[((size_t)brow * n + t) * T + p0 + tid] = de[p0 + tid];
    }
    {  // d q partial of the member's (channel block, position block): dq_c = v_c sum_j de_j (1 - tanh^2(K_jc + q_c)).  Thread = (channel c,
       // position residue pc): NPC = 512 / DS residues side by side, so a wave reads consecutive channels of two or one key rows; the
       // NPC partials of a channel meet in LDS (no cross-lane reduction on the chain)
      constexpr int NPC = DX_NT / DS;
      const int c = tid % DS, pc = tid / DS;
      const float qc = qv[c];
      float s = 0.f;
      for (int j = pc; j < psn; j += NPC) {
        const float th = taco_tanh_fast(Kc[(size_t)j * DS + c] + qc);
        s = fmaf(de[ps0 + j], 1.f - th * th, s);
      }
      cpart[pc * DS + c] = s;
      __syncthreads();
      if (tid < DS) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < NPC; ++k) tot += cpart[k * DS + tid];
        dx_publish<WTC>(X + xl.dq + (size_t)(arow * Pp + pb) * 256 + cb * DS + tid, tot * vv[tid], tag, rt);
      }
    }
    {  // gather d q of every row: sum over the Pp position blocks
      constexpr int NI = (RG * 256 + DX_NT - 1) / DX_NT;
#pragma unroll
      for (int u = 0; u < NI; ++u) {
        const int i = u * DX_NT + tid;
        if (i < RG * 256) {
          const int r = i / 256, c = i % 256;
          float v[Pp];
          dx_poll<Pp, DB_DLY(7)>(X + xl.dq + (size_t)(r * Pp) * 256 + c, (size_t)256, tag, v, rt);
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < Pp; ++k) s += v[k];
          st[r * DBS_LD + DBS_DQ + c] = s;
        }
      }
    }
    __syncthreads();
    DB_STAMP(9);
    if (asl == 0 && tid < 256 && brow < a.B) a.g_dq[((size_t)brow * n + t) * 256 + tid] = st[arow * DBS_LD + DBS_DQ + tid];
    // ================= d h_att += d q . Wq^T -> attention GRU 'a' =================
    {
      float acc[1][RG], s[1][RL];
      dx_zero<1, RG>(acc);
      dx_pass<DBR_Q, 1, RG, DB_NREG, DBS_LD>(W, st + DBS_DQ, lane, acc);
      dx_reduce<1, RG>(acc, s, lane);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const float g = dhA[q] + dIn_hA[q] + s[0][q];
        const float dcp = g * (1.f - OWN(OW_UA, q)) * (1.f - OWN(OW_CA, q) * OWN(OW_CA, q));
        dgu[q] = g * ((zA ? 0.f : OWN(OW_HAP, q)) - OWN(OW_CA, q)) * OWN(OW_UA, q) * (1.f - OWN(OW_UA, q));
        dht[q] = g;
        if (epl) dx_publish<WTC>(X + xl.dcpa + erow[q] * 256 + en, dcp, tag, rt);
        DB_OUT(a.g_dcpA, 256, q, en, dcp); DB_OUT(a.g_dgpA, 512, q, 256 + en, dgu[q]);
      }
    }
    dx_gather<RG, 256, false, DBS_LD, DX_NT, DB_DLY(8)>(X + xl.dcpa, tag, st, DBS_DCPA, 0, 0, tid, rt);
    __syncthreads();
    DB_STAMP(10);
    // ================= attention GRU 'b' (the x part has 128 inputs: rows 4m + w of waves 0-3) =================
    DB_CELL_B(DBR_CAX, DBR_CAH, DBS_DCPA, xl.dgpa, (zA ? 0.f : OWN(OW_HAP, q)), OWN(OW_RA, q), OWN(OW_UA, q), dhp, a.g_dgpA)
    dx_gather<RG, 512, false, DBS_LD, DX_NT, DB_DLY(9)>(X + xl.dgpa, tag, st, DBS_DGPA, 0, 0, tid, rt);
    __syncthreads();
    DB_STAMP(11);
    // ================= attention GRU 'c' -> d p2 (ReLU mask) =================
    {
      float acc[2][RG], s[2][RL];
      dx_zero<2, RG>(acc);
      db_pass512<DBR_GAX, 1, RG, DB_NREG, DBS_LD>(W, st + DBS_DGPA, lane, reinterpret_cast<float (&)[1][RG]>(acc[0]));
      db_pass512<DBR_GAH, 1, RG, DB_NREG, DBS_LD>(W, st + DBS_DGPA, lane, reinterpret_cast<float (&)[1][RG]>(acc[1]));
      dx_reduce<2, RG>(acc, s, lane);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        dhA[q] = dhp[q] + s[1][q];
        if (wave < 4) {
          const float dz2 = (OWN(OW_P2, q) > 0.f) ? tx[q] + s[0][q] : 0.f;
          if (epl) dx_publish<WTC>(X + xl.dz2 + erow[q] * 128 + en2, dz2, tag, rt);
          DB_OUT(a.g_dz2, 128, q, en2, dz2);
        }
      }
    }
    dx_gather<RG, 128, false, DBS_LD, DX_NT, DB_DLY(10)>(X + xl.dz2, tag, st, DBS_DZ2, 0, 0, tid, rt);
    __syncthreads();
    DB_STAMP(12);
    // ================= prenet layer 2^T, ReLU mask of layer 1 =================
    {
      float acc[1][RG], s[1][RL];
      dx_zero<1, RG>(acc);
#pragma unroll
      for (int r = 0; r < RG; ++r) {
        const float2 xv = *reinterpret_cast<const float2*>(st + r * DBS_LD + DBS_DZ2 + 2 * lane);
        acc[0][r] = fmaf(W[DBR_P2], xv.x, fmaf(W[DBR_P2 + 1], xv.y, 0.f));
      }
      dx_reduce<1, RG>(acc, s, lane);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const float dz1 = (OWN(OW_P1, q) > 0.f) ? s[0][q] : 0.f;
        if (epl) dx_publish<WTC>(X + xl.dz1 + erow[q] * 256 + en, dz1, tag, rt);
        DB_OUT(a.g_dz1, 256, q, en, dz1);
      }
    }
    dx_gather<RG, 256, false, DBS_LD, DX_NT, DB_DLY(11)>(X + xl.dz1, tag, st, DBS_DZ1, 0, 0, tid, rt);
    __syncthreads();
    DB_STAMP(13);
    // ================= prenet layer 1 (context rows)^T: the gradient of context(t - 1) =================
    {
      float acc[1][RG], s[1][RL];
      dx_zero<1, RG>(acc);
      dx_pass<DBR_P1C, 1, RG, DB_NREG, DBS_LD>(W, st + DBS_DZ1, lane, acc);
      dx_reduce<1, RG>(acc, s, lane);
#pragma unroll
      for (int q = 0; q < RL; ++q) dctxc[q] = s[0][q];
    }
    DB_STAMP(14);
    // the rows fetched for step t - 1 have landed in every wave that issued them by now (twelve polls ago); this barrier publishes them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#undef DB_CELL_B
#undef DB_STAMP
#undef OWN
#undef DB_OUT
  // gradients of the initial states = the carries left after step 0 (tacotron.py:183-197: deepvoice feeds them from the speaker layers)
#pragma unroll
  for (int q = 0; q < RL; ++q) {
    if (ev[q]) {
      const size_t o = (size_t)(row0 + erow[q]) * DX_W + en;
      if (a.d_att_init) a.d_att_init[o] = dhA[q];
      if (a.d_h10) a.d_h10[o] = dh1[q];
      if (a.d_h20) a.d_h20[o] = dh2[q];
    }
  }
  if (a.dsb_acc && asl == 0 && tid == 0 && brow < a.B) a.dsb_acc[brow] = dsb_row;
}

template <int RG, bool TRACE = false>
__global__ __launch_bounds__(DX_NT) void k_decoder_bwd_xcd(const DbArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float dx_smem[];
  const DbArgs& a = a_in;
  int* ictl = reinterpret_cast<int*>(dx_smem + db_lds_floats(RG, a.T_in) - 64);      // census words: the last 64 floats of the LDS request
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, a.force_wt, ictl, threadIdx.x, 40);
  const int group = __builtin_amdgcn_readfirstlane(ictl[0]);
  const int member = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
#ifdef DX_DLY_RT
  for (int i = 0; i < 12; ++i) rt.dly[i] = g_db_dly[i];
#endif
  if (__builtin_amdgcn_readfirstlane((int)rt.wt)) db_body<RG, true, TRACE>(a, dx_smem, group, member, rt);
  else db_body<RG, false, TRACE>(a, dx_smem, group, member, rt);
}
