// taco_decoder_xcd.h -- the whole decoder loop (K10-K17) as ONE persistent launch,
// weight-stationary and XCD-local.
//
// Reference semantics: rnn_wrappers.py:218-341 (AttentionWrapper.call), :367-378 (DecoderPrenetWrapper), :405-415
// (ConcatOutputAndAttentionWrapper); tacotron.py:127-181 (cells), helpers.py:9-32 (TacoTestHelper: feed back the last of the r
// frames, stop flags); SURVEY App. A.6 (TF GRUCell), A.9-A.11 (scores and normalisers).
//
// Why this shape.  A decoder step is a chain of dependent mat-vec / attention stages over [B, <=768] activations and 6 MB of
// weights.  As one launch per stage the chain costs ~4.5 us per link on MI355X (profiles/r01_*: kernel boundary + cold weight
// fetch + a short MFMA chain), 42.5 us per step at C2.  Here the step never leaves the chip's registers:
//   * 256 workgroups of 512 threads (two waves per SIMD: 256 VGPRs per thread), one per CU.  The 32 CUs of one XCD form a
//     GROUP that owns RG batch rows for the whole loop (C2: 8 groups x 4 rows; C5: 8 x 1).  Groups never talk to each other.
//   * Weight-stationary: member m of a group owns 8 of the 256 output columns of every stage, wave w of the member owns
//     column 8m + w, and every lane keeps its K-slice (4 consecutive inputs per 256) of that column's weights in VGPRs for
//     the whole launch (106 registers per thread + the member's slice of the query layer; loaded once, coalesced, from a
//     per-thread pack built at finalize).  A GRU unit (reset gate, update gate, candidate) lives entirely in one wave.
//   * Attention memory stationary: the member also keeps, for ONE of the group's rows, a channel block of the keys and of the
//     values (all encoder positions x 256*RG/32 channels) in LDS -- the per-step K/V stream of the launch-per-stage path
//     (8.4 MB per step at C2) disappears.  Scores are summed over channel blocks, so the query is needed only for the
//     member's own channels and is computed locally (no exchange of the query).
//   * Between stages the members exchange their column slices through the XCD's own L2: 8-byte {value, tag = step+1}
//     granules (the data is the flag), producer store -> consumer poll with L1-bypassing (sc1) loads.  When every XCD runs
//     exactly one 32-member group (checked by an in-kernel census of HW_REG_XCC_ID) the producer stores stay in the shared L2
//     (sc0) and a hop costs an L2 round trip.  Otherwise (or with force_wt) the stores are write-through (sc1) and the protocol
//     is placement-independent (MI355X_MICROARCH.md, inter-workgroup visibility).  10 exchanges per step.
//   * A stage is: gather the previous stage's vector into LDS -> one barrier -> every wave: one float4 of each row's input per
//     lane, FMAs against its resident weights, a full-rate DPP wave reduction, the epilogue in lanes 0..RG-1, publish.  The parts
//     of a stage that do not depend on the vector in flight (the h rows of the gates, the context rows of the next prenet, ...)
//     run before the wait for it.
//   * Every spin is bounded; a timeout raises a.err and the launch drains without hanging.
//   * Manual attention (rnn_wrappers.py:313-317: `alignments = manual_alignments[:, time, :]`): the query, score, exchange and
//     normaliser phases of a step are skipped; the row of the step is fetched at the top of the step straight into LDS
//     (global_load_lds_dword, nothing waits for it there) and is the alignment the context / history / state are built from.
//   * Model type 'simple' (rnn_wrappers.py:372-376, 408-413): the speaker embedding is one more input segment of the attention GRU and
//     of the concat projection; it is constant over the loop, so its products with those rows are formed once per launch by a
//     small kernel (k_dx_rowbias) and enter the epilogues as per-(row, column) biases.
//   * Other presets of hparams.py:71-117 (round 4): the attention width (attention_size: keys, query layer, attention_v) is the template
//     parameter AW -- 128 / 256 / 512 --, and a third decoder prenet layer (dec_prenet_sizes [256, 128, 64]) the parameter PD = 3: one more
//     stage (64 columns, two per member) whose output feeds the attention GRU's 64 input rows (one per lane).  Inference only.
//   * Training forward (TAPE instantiation; helpers.py:35-67, train.py:215-219): the frame fed to the next step's prenet comes from
//     the teacher buffer (fetched straight into LDS one step ahead of its use) instead of the step's own output, and every value the
//     backward pass needs (gates, candidates, r*h, states, the concat projection, the processed query, raw scores, alignments,
//     context) is written to the tape by the lane that produces it -- [B, n, .] arrays, the layout of DecTape in taco_train.h.
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
#define DX_P3 64             // dec_prenet[2] (PD = 3: presets with a third prenet layer)
#define DX_SPIN_LIMIT (1u << 21)
#ifndef DX_POLL_SLEEP
#define DX_POLL_SLEEP 0          // s_sleep units (64 clocks) between two polls of a stale granule
#endif
// s_sleep units (64 clocks) in front of the FIRST poll of a decoder gather (round 6).  A poll that reaches the L2 ahead of the stores it waits
// for comes back stale and costs a second round trip (~500 clocks); where nothing runs between a wave's publish and its gather (DX_FIRST_POLL_DELAY:
// r*h of the three cells, h_att, context, the next step's prenet) a short sleep lets the group's stores land first.  Gathers that follow a
// run-ahead pass (DX_POLL_DELAY_B: p2, the partial scores, h1, h2) start later, and gain as much.  Measured at C2 on two boxes, decoder alone
// (profiles/r06_ab_poll_delay.txt): 0 / 0 1.336 ms, 4 / 4 1.322, 5 / 5 1.298, 6 / 6 1.299, 7 / 7 1.328, 8 / 8 1.350 -- a flat optimum around 320 clocks.
#ifndef DX_FIRST_POLL_DELAY
#define DX_FIRST_POLL_DELAY 5
#endif
#ifndef DX_PRIO_OLD
#define DX_PRIO_OLD 1       // s_setprio level of waves 0-3 (the first wave of every SIMD) for the whole loop; 0: none.  See dx_body.
#endif
#ifndef DX_POLL_DELAY_B
#define DX_POLL_DELAY_B 5
#endif
#define DX_TRACE_STEPS 8
#define DX_TRACE_SLOTS 16

// Register map of the per-thread weight pack (host mirror: dx_build_pack in taco_lib.hip).  A PASS multiplies one 256-wide
// (AGX: 128-wide) input vector with NCOLS weight columns of the wave; lane l holds inputs 4l..4l+3 (AGX: 2l, 2l+1) of each:
// register REG0 + 4*col + e  (AGX: REG0 + 2*col + e).
enum {
  DXR_P2 = 0,     // prenet layer 2: p1 -> column 4m + w (waves 0-3)                                   1 col
  DXR_AGH = 4,    // attention GRU gates, h rows: h_att -> r, u                                        2 cols
  DXR_AGX = 12,   // attention GRU, x rows (128): p2 -> r, u, candidate-x                              3 cols x 2 regs
                  // (PD = 3: x rows (64): p3 -> r, u, candidate-x, 3 cols x 1 reg; then prenet layer 3: p2 -> column 2m + w (waves 0-1), 1 col x 2 regs)
  DXR_AC = 18,    // attention GRU candidate, h rows: r*h -> c                                         1 col
  DXR_G1H = 22,   // decoder GRU 1 gates, h rows: h1 -> r, u                                           2 cols
  DXR_G1A = 30,   // GRU 1 with the concat projection folded in, h_att rows: -> r, u, candidate-x, o0  4 cols
  DXR_G1B = 46,   // the same, context rows                                                            4 cols
  DXR_G1C = 62,   // GRU 1 candidate, h rows: r*h1 -> c
  DXR_G2H = 66,   // decoder GRU 2 gates, h rows
  DXR_G2X = 74,   // GRU 2, x rows: out1 -> r, u, candidate-x                                          3 cols
  DXR_G2C = 86,   // GRU 2 candidate, h rows
  DXR_P1C = 90,   // next step's prenet layer 1 (frame projection folded in), context rows
  DXR_P1O = 94,   // the same, GRU-stack-output rows
  DXR_F = 98,     // frame projection: out2 -> columns NCF*m + w and NCF*m + w + 8                     2 cols
  DX_NREG = 106
};

// per-row state vectors in LDS (floats)
enum { DXS_P2 = 0, DXS_HATT = 128, DXS_CTX = 384, DXS_OUT2 = 640, DXS_T = 896, DXS_H1 = 1152, DXS_OUT1 = 1408, DXS_H2 = 1664,
       DXS_LD = 1920 };
// own-column bias table in LDS: bl[slot][wave]
enum { DXB_P1 = 0, DXB_P2, DXB_AR, DXB_AU, DXB_AC, DXB_G1R, DXB_G1U, DXB_G1X, DXB_O0, DXB_G1C, DXB_G2R, DXB_G2U, DXB_G2C, DXB_F0, DXB_F1,
       DXB_P3, DXB_N };
// per-row bias slots (model type 'simple'): the speaker embedding's product with the speaker rows of the attention GRU (r, u,
// candidate-x) and of the folded GRU 1 (r, u, candidate-x, o0); rowbias[b][slot][256]
enum { DXRB_AR = 0, DXRB_AU, DXRB_AX, DXRB_G1R, DXRB_G1U, DXRB_G1X, DXRB_O0, DXRB_N };

// tape slots of the TAPE instantiation: [B, n, 256] arrays carved back to back (carve_dec_tape), slot s at tape + s * tstride
enum { DXT_P1 = 0, DXT_HA, DXT_RA, DXT_UA, DXT_CA, DXT_RHA, DXT_Q, DXT_O0, DXT_R1, DXT_U1, DXT_C1, DXT_RH1, DXT_H1, DXT_O1,
       DXT_R2, DXT_U2, DXT_C2, DXT_RH2, DXT_H2, DXT_O2, DXT_N };

// exchange buffers of one group, in granules, for RG rows (the host sizes the buffer with RG = 8)
struct DxX { int p1, p2, rha, ha, sc, ctx, rh1, h1, o1, rh2, h2, p3, fb, total; };
__host__ __device__ inline DxX dx_xlayout(int RG, int T_in) {
  DxX x; int o = 0;
  x.p1 = o; o += RG * DX_W;  x.p2 = o; o += RG * DX_P2; x.rha = o; o += RG * DX_W; x.ha = o; o += RG * DX_W;
  x.sc = o; o += DX_GROUP * T_in;                         // partial scores: [row][member of the row][position]
  x.ctx = o; o += RG * DX_W; x.rh1 = o; o += RG * DX_W; x.h1 = o; o += RG * DX_W; x.o1 = o; o += RG * DX_W;
  x.rh2 = o; o += RG * DX_W; x.h2 = o; o += RG * DX_W;
  x.p3 = o; o += RG * DX_P3;                              // (PD = 3 only)
  x.fb = o; o += RG * DX_P2;                              // the step's last frame (TAPE with own_fb: rnn_decoder_test_mode), num_mels <= 128
  x.total = o;
  return x;
}
// LDS floats of a member (host mirror of the carve in the kernel)
__host__ __device__ inline size_t dx_lds_floats(int RG, int T_in, bool teacher = false, int AW = DX_W) {
  const int Pr = DX_GROUP / RG, DC = DX_W / Pr, Tpad = (T_in + 3) & ~3;
  const int Pc = Pr < 8 ? Pr : 8, Pp = Pr / Pc, DS = AW / Pc, TS = (T_in + Pp - 1) / Pp;
  size_t n = 0;
  n += (size_t)RG * DXS_LD;           // state
  n += (size_t)TS * DS;               // keys: position block x score-channel block
  n += (size_t)T_in * DC;             // values: all positions x context-channel block
  n += 4 * (size_t)Tpad;              // sc, tmp, tmp2, alp
  n += 3 * 64;                        // qv, vv, bq (own channels, DC <= 64)
  n += (size_t)DX_NW * 64;            // context partials [wave][DC]
  n += (size_t)DXB_N * DX_NW;         // own-column biases
  n += (size_t)DXRB_N * RG * DX_NW;   // per-row biases of the own columns ('simple': speaker term)
  n += (size_t)((T_in + 63) & ~63);   // manual alignment row of the step
  if (teacher) n += (size_t)RG * DX_W;   // teacher frames of the step (TAPE instantiation)
  n += 64;                            // control words
  return n;
}
__host__ __device__ inline int dx_score_blocks(int RG) { const int Pr = DX_GROUP / RG; return Pr < 8 ? Pr : 8; }   // channel blocks per row
__host__ __device__ inline int dx_q_regs(int RG, int AW = DX_W) { return (AW / dx_score_blocks(RG)) / 2; }   // 4 per query column, DS/8 columns per wave

struct DxArgs {
  const float* wpack;                                  // [32 members][DX_NREG][DX_NT]
  const float* qpack;                                  // [32 members][dx_q_regs(RG)][DX_NT]   (layout depends on RG)
  const float* b_p1_0; const float* b_p1c; const float* b_p2;      // prenet biases: layer 1 raw (step 0), composite (steps >= 1), layer 2
  const float* b_p3;                                   // layer 3 (PD = 3)
  const float* b_ag; const float* b_ac;                // attention GRU: gates [2H] (r|u), candidate [H]
  const float* b_g1f;                                  // folded GRU 1: [4H] = gates (r|u) | candidate-x | o0
  const float* b_g1c; const float* b_g2g; const float* b_g2c;
  const float* b_f;                                    // [rM]
  const float* att_v; const float* att_b; const float* score_bias;
  const float* keys; const float* values;              // [B, T_in, 256] each
  const float* h_att0; const float* h10; const float* h20;   // deepvoice initial states [B, 256] or null (zeros)
  const float* manual;                                 // [B, n, T_in] manual alignments (rnn_wrappers.py:313-317) or null
  const float* rowbias;                                // [B, DXRB_N, 256] ('simple') or null
  // TAPE instantiation only (training forward):
  const float* teacher;                                // [B, n, mels]: frame t feeds the prenet of step t + 1 (helpers.py:44,66)
  const float* p1o_raw;                                // [32 members][4][DX_NT] or null: the raw frame rows of prenet layer 1 for the registers DXR_P1O .. + 3 of a pack built in
                                                       // composite form (teacher-forced decoding on an inference model); the layer's bias is then b_p1_0 at every step
  int own_fb;                                          // rnn_decoder_test_mode (helpers.py:63-64; the test model of train.py:158-166): no teacher, the prenet of step t + 1 reads the
                                                       // LAST of the r frames step t emitted -- with the raw prenet rows of the teacher-form pack, so the frame is exchanged first
  float* tape; size_t tstride;                         // 256-wide per-step arrays: slot s, row (b, t) at tape + s*tstride + (b*n + t)*256
  float* tp_p2; float* tp_ctx; int ld_p2, ld_ctx;      // prenet output [B, n, ld_p2], context [B, n, ld_ctx] (wider rows: 'simple' parks the speaker embedding behind them)
  float* tp_e; float* tp_alpha;                        // raw scores [B, n, T_in]; alignments [B, n + 1, T_in] (slot t + 1 = step t)
  int mels;
  float* mel; float* hist; int* nz; float* dbg;
  unsigned long long* xbuf; unsigned* ctl; unsigned* err; long long* trace;
  int B, T_in, n, rM, att_type, grp0, ngroups, force_wt, dbgw;
  int trc_member, trc_tid;                             // TRACE instantiation: the (member, thread) of group 0 that stamps (default 0, 0; TACO_TRACE_MEMBER / TACO_TRACE_TID)
};

#define DX_DPP0(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, false))
#define DX_DPPZ(v, ctrl, rmask, bmask) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), (rmask), (bmask), true))
// sum over the wave; every lane of a row of 16 first gets its row's total, the row totals are chained into lane 63 and read back
// as a wave-uniform value
__device__ __forceinline__ float dx_allsum(float v) {
  v += DX_DPP0(v, 0xB1);                 // quad_perm [1,0,3,2]
  v += DX_DPP0(v, 0x4E);                 // quad_perm [2,3,0,1]
  v += DX_DPP0(v, 0x141);                // row_half_mirror
  v += DX_DPP0(v, 0x140);                // row_mirror
  v += DX_DPPZ(v, 0x142, 0xA, 0xF);      // row_bcast:15 -> rows 1, 3
  v += DX_DPPZ(v, 0x143, 0xC, 0xF);      // row_bcast:31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float dx_allmax(float v) {
  v = fmaxf(v, DX_DPP0(v, 0xB1)); v = fmaxf(v, DX_DPP0(v, 0x4E)); v = fmaxf(v, DX_DPP0(v, 0x141)); v = fmaxf(v, DX_DPP0(v, 0x140));
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float dx_quadsum(float v) { v += DX_DPP0(v, 0xB1); v += DX_DPP0(v, 0x4E); return v; }
// inclusive prefix sum over the wave on full-rate DPP (row_shr 1,2,3 / 4 / 8, row_bcast 15 / 31)
__device__ __forceinline__ float dx_scan(float v) {
  float t = v + DX_DPPZ(v, 0x111, 0xF, 0xF);
  t += DX_DPPZ(v, 0x112, 0xF, 0xF);
  t += DX_DPPZ(v, 0x113, 0xF, 0xF);
  t += DX_DPPZ(t, 0x114, 0xF, 0xE);
  t += DX_DPPZ(t, 0x118, 0xF, 0xC);
  t += DX_DPPZ(t, 0x142, 0xA, 0xF);
  t += DX_DPPZ(t, 0x143, 0xC, 0xF);
  return t;
}
__device__ __forceinline__ float dx_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// one pass: acc[c][r] += sum_e W[REG0 + 4c + e] * x[r][4*lane + e]   (x: LDS, row stride DXS_LD)
template <int REG0, int NCOLS, int RG, int NW = DX_NREG, int LD = DXS_LD>
__device__ __forceinline__ void dx_pass(const float (&W)[NW], const float* x, int lane, float (&acc)[NCOLS][RG]) {
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const float4 xv = *reinterpret_cast<const float4*>(x + r * LD + 4 * lane);
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      acc[c][r] = fmaf(W[REG0 + 4 * c + 0], xv.x, acc[c][r]);
      acc[c][r] = fmaf(W[REG0 + 4 * c + 1], xv.y, acc[c][r]);
      acc[c][r] = fmaf(W[REG0 + 4 * c + 2], xv.z, acc[c][r]);
      acc[c][r] = fmaf(W[REG0 + 4 * c + 3], xv.w, acc[c][r]);
    }
  }
}
// the 128-wide pass (attention GRU x rows): 2 inputs per lane
template <int REG0, int NCOLS, int RG>
__device__ __forceinline__ void dx_pass2(const float (&W)[DX_NREG], const float* x, int lane, float (&acc)[NCOLS][RG]) {
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const float2 xv = *reinterpret_cast<const float2*>(x + r * DXS_LD + 2 * lane);
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) {
      acc[c][r] = fmaf(W[REG0 + 2 * c + 0], xv.x, acc[c][r]);
      acc[c][r] = fmaf(W[REG0 + 2 * c + 1], xv.y, acc[c][r]);
    }
  }
}
// the 64-wide pass (PD = 3: attention GRU x rows): 1 input per lane
template <int REG0, int NCOLS, int RG>
__device__ __forceinline__ void dx_pass1(const float (&W)[DX_NREG], const float* x, int lane, float (&acc)[NCOLS][RG]) {
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const float xv = x[r * DXS_LD + lane];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) acc[c][r] = fmaf(W[REG0 + c], xv, acc[c][r]);
  }
}
template <int NCOLS, int RG>
__device__ __forceinline__ void dx_zero(float (&acc)[NCOLS][RG]) {
#pragma unroll
  for (int c = 0; c < NCOLS; ++c)
#pragma unroll
    for (int r = 0; r < RG; ++r) acc[c][r] = 0.f;
}
// Wave reduction of NC columns x RG rows of per-lane partial sums.  A plain tree costs 6 cross-lane (DPP) adds per value and DPP
// adds are the expensive instruction here (about four times a plain VALU op), so the first two levels are a butterfly that also
// distributes the ROWS over the lanes of a quad: at each level a lane keeps half of its rows, sends the other half to its
// partner and adds what the partner sends -- the number of live values halves.  After two levels lane l holds, per column, the
// RG/4 rows  (l&1)*RG/2 + ((l>>1)&1)*RG/4 + q  summed over its quad; row_ror 4 / 8 finish the row of 16 lanes and two
// permlane swaps the four rows of the wave, all lane-position preserving.  Every lane ends with the wave totals of its rows.
template <int RG> struct DxRL { static constexpr int value = RG >= 4 ? RG / 4 : 1; };
template <int RG>
__device__ __forceinline__ int dx_row(int lane, int q) {
  if (RG >= 4) return (lane & 1) * (RG / 2) + ((lane >> 1) & 1) * (RG / 4) + q;
  if (RG == 2) return lane & 1;
  return 0;
}
__device__ __forceinline__ float dx_xrow16(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float dx_xrow32(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int NC, int RG>
__device__ __forceinline__ void dx_reduce(const float (&a)[NC][RG], float (&out)[NC][DxRL<RG>::value], int lane) {
  constexpr int RL = DxRL<RG>::value;
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
  float d[NC][RL];
  if (RG >= 4) {
    constexpr int H = RG / 2, Q = RG / 4;
    float b[NC][H];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int p = 0; p < H; ++p) {
        const float keep = b0 ? a[c][H + p] : a[c][p], send = b0 ? a[c][p] : a[c][H + p];
        b[c][p] = keep + DX_DPP0(send, 0xB1);
      }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const float keep = b1 ? b[c][Q + q] : b[c][q], send = b1 ? b[c][q] : b[c][Q + q];
        d[c][q] = keep + DX_DPP0(send, 0x4E);
      }
  } else if (RG == 2) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float keep = b0 ? a[c][1] : a[c][0], send = b0 ? a[c][0] : a[c][1];
      const float t = keep + DX_DPP0(send, 0xB1);
      d[c][0] = t + DX_DPP0(t, 0x4E);
    }
  } else {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float t = a[c][0] + DX_DPP0(a[c][0], 0xB1);
      d[c][0] = t + DX_DPP0(t, 0x4E);
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int q = 0; q < RL; ++q) d[c][q] += DX_DPP0(d[c][q], 0x124);     // row_ror:4
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int q = 0; q < RL; ++q) d[c][q] += DX_DPP0(d[c][q], 0x128);     // row_ror:8
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int q = 0; q < RL; ++q) d[c][q] = dx_xrow16(d[c][q]);
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int q = 0; q < RL; ++q) out[c][q] = dx_xrow32(d[c][q]);
}

// ---- round 5: the step's VALU diet (the passes and reductions of k_decoder_xcd are issue bound: two waves per SIMD, ~130 instructions
// each per gates stage) ----
#ifndef DX_DIET
#define DX_DIET 1            // A/B: 0 = the scalar passes and the select-based reduction of rounds 2-4
#endif
// Column PAIRS in one v_pk_fma_f32 per input: the input broadcast to both halves (op_sel), the two columns' weights of that input side by
// side.  The resident weights ARE register pairs (taco_f32x2 WP[DX_NREG / 2], loaded once in the order the passes consume them; an array of
// scalars left the pairing to the allocator, which parked 40 of them in scratch): pair k of a pass that starts at register REG0 is
// WP[REG0 / 2 + ...] -- every DXR_* base is even and every pass owns an even number of registers.  Accumulators of paired columns are pairs
// from the start; single columns stay scalar (pairing their even / odd inputs costs the adds it saves).
//   single column (4 registers r0..r3):             WP[k] = (r0, r1), WP[k + 1] = (r2, r3)
//   two columns (8 registers, c0: +e, c1: +4+e):    WP[k + e] = (c0_e, c1_e)
//   AGX, 3 columns x 2 inputs (PD = 2):             WP[6] = (c0_0, c1_0), WP[7] = (c0_1, c1_1), WP[8] = (c2_0, c2_1)
//   AGX, 3 columns x 1 input + prenet 3 (PD = 3):   WP[6] = (c0, c1), WP[7] = (c2, p3_0), WP[8] = (p3_1, -)
#define DX_NWP (DX_NREG / 2)
__host__ __device__ constexpr int dxw_two(int base, int r, int h) { return base + (r - base) / 2 + 4 * h; }      // pair e of a two-column block: (base + e, base + 4 + e)
__host__ __device__ constexpr int dxw_src(int k, int h, int PD) {      // pack register that half h of pair k holds
  const int r = 2 * k;
  if (r >= DXR_AGH && r < DXR_AGX) return dxw_two(DXR_AGH, r, h);
  if (r >= DXR_AGX && r < DXR_AC) return PD == 3 ? r + h : (r < DXR_AGX + 4 ? DXR_AGX + (r - DXR_AGX) / 2 + 2 * h : r + h);
  if (r >= DXR_G1H && r < DXR_G1A) return dxw_two(DXR_G1H, r, h);
  if (r >= DXR_G1A && r < DXR_G1C) return dxw_two(DXR_G1A + 8 * ((r - DXR_G1A) / 8), r, h);      // G1A, G1B: two two-column blocks each
  if (r >= DXR_G2H && r < DXR_G2X) return dxw_two(DXR_G2H, r, h);
  if (r >= DXR_G2X && r < DXR_G2X + 8) return dxw_two(DXR_G2X, r, h);
  if (r >= DXR_F) return dxw_two(DXR_F, r, h);
  return r + h;                                                         // single columns: P2, AC, G1C, the third column of G2X, G2C, P1C, P1O
}
#define DXQ_FMA(acc, xs, wp) \
  do { if (DX_DIET) acc = __builtin_elementwise_fma((taco_f32x2){xs, xs}, wp, acc); else { acc.x = fmaf((wp).x, xs, acc.x); acc.y = fmaf((wp).y, xs, acc.y); } } while (0)
template <int RG>
__device__ __forceinline__ void dxq_zero(taco_f32x2 (&acc)[RG]) {
#pragma unroll
  for (int r = 0; r < RG; ++r) acc[r] = (taco_f32x2){0.f, 0.f};
}
// one column over a 256-wide input
template <int REG0, int RG, int LD = DXS_LD>
__device__ __forceinline__ void dxw_single(const taco_f32x2 (&WP)[DX_NWP], const float* x, int lane, float (&acc)[1][RG]) {
  constexpr int k = REG0 / 2;
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const float4 xv = *reinterpret_cast<const float4*>(x + r * LD + 4 * lane);
    acc[0][r] = fmaf(WP[k].x, xv.x, acc[0][r]); acc[0][r] = fmaf(WP[k].y, xv.y, acc[0][r]);
    acc[0][r] = fmaf(WP[k + 1].x, xv.z, acc[0][r]); acc[0][r] = fmaf(WP[k + 1].y, xv.w, acc[0][r]);
  }
}
// two columns over a 256-wide input
template <int REG0, int RG, int LD = DXS_LD>
__device__ __forceinline__ void dxw_pair(const taco_f32x2 (&WP)[DX_NWP], const float* x, int lane, taco_f32x2 (&acc)[RG]) {
  constexpr int k = REG0 / 2;
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const float4 xv = *reinterpret_cast<const float4*>(x + r * LD + 4 * lane);
    DXQ_FMA(acc[r], xv.x, WP[k]); DXQ_FMA(acc[r], xv.y, WP[k + 1]); DXQ_FMA(acc[r], xv.z, WP[k + 2]); DXQ_FMA(acc[r], xv.w, WP[k + 3]);
  }
}
// four columns = two pairs, one read of the input
template <int REG0, int RG, int LD = DXS_LD>
__device__ __forceinline__ void dxw_quad(const taco_f32x2 (&WP)[DX_NWP], const float* x, int lane, taco_f32x2 (&p0)[RG], taco_f32x2 (&p1)[RG]) {
  constexpr int k = REG0 / 2;
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const float4 xv = *reinterpret_cast<const float4*>(x + r * LD + 4 * lane);
    DXQ_FMA(p0[r], xv.x, WP[k]); DXQ_FMA(p1[r], xv.x, WP[k + 4]);
    DXQ_FMA(p0[r], xv.y, WP[k + 1]); DXQ_FMA(p1[r], xv.y, WP[k + 5]);
    DXQ_FMA(p0[r], xv.z, WP[k + 2]); DXQ_FMA(p1[r], xv.z, WP[k + 6]);
    DXQ_FMA(p0[r], xv.w, WP[k + 3]); DXQ_FMA(p1[r], xv.w, WP[k + 7]);
  }
}
// a pair (REGP ..) and a single column (REGS ..) over the same 256-wide input
template <int REGP, int REGS, int RG, int LD = DXS_LD>
__device__ __forceinline__ void dxw_pair_single(const taco_f32x2 (&WP)[DX_NWP], const float* x, int lane, taco_f32x2 (&p)[RG], float (&sg)[1][RG]) {
  constexpr int k = REGP / 2, ks = REGS / 2;
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const float4 xv = *reinterpret_cast<const float4*>(x + r * LD + 4 * lane);
    DXQ_FMA(p[r], xv.x, WP[k]); sg[0][r] = fmaf(WP[ks].x, xv.x, sg[0][r]);
    DXQ_FMA(p[r], xv.y, WP[k + 1]); sg[0][r] = fmaf(WP[ks].y, xv.y, sg[0][r]);
    DXQ_FMA(p[r], xv.z, WP[k + 2]); sg[0][r] = fmaf(WP[ks + 1].x, xv.z, sg[0][r]);
    DXQ_FMA(p[r], xv.w, WP[k + 3]); sg[0][r] = fmaf(WP[ks + 1].y, xv.w, sg[0][r]);
  }
}
// attention GRU x rows, PD = 2: the 128-wide input (two inputs per lane), pair (r, u) and the candidate-x column
template <int RG>
__device__ __forceinline__ void dxw_agx2(const taco_f32x2 (&WP)[DX_NWP], const float* x, int lane, taco_f32x2 (&p)[RG], float (&sg)[1][RG]) {
  constexpr int k = DXR_AGX / 2;
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const float2 xv = *reinterpret_cast<const float2*>(x + r * DXS_LD + 2 * lane);
    DXQ_FMA(p[r], xv.x, WP[k]); sg[0][r] = fmaf(WP[k + 2].x, xv.x, sg[0][r]);
    DXQ_FMA(p[r], xv.y, WP[k + 1]); sg[0][r] = fmaf(WP[k + 2].y, xv.y, sg[0][r]);
  }
}
// ... PD = 3: the 64-wide input (one input per lane)
template <int RG>
__device__ __forceinline__ void dxw_agx1(const taco_f32x2 (&WP)[DX_NWP], const float* x, int lane, taco_f32x2 (&p)[RG], float (&sg)[1][RG]) {
  constexpr int k = DXR_AGX / 2;
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const float xv = x[r * DXS_LD + lane];
    DXQ_FMA(p[r], xv, WP[k]); sg[0][r] = fmaf(WP[k + 1].x, xv, sg[0][r]);
  }
}
// prenet layer 3 (PD = 3): one column over the 128-wide prenet-2 output, registers DXR_AGX + 3, + 4
template <int RG>
__device__ __forceinline__ void dxw_p3(const taco_f32x2 (&WP)[DX_NWP], const float* x, int lane, float (&acc)[1][RG]) {
  constexpr int k = DXR_AGX / 2;
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const float2 xv = *reinterpret_cast<const float2*>(x + r * DXS_LD + 2 * lane);
    acc[0][r] = fmaf(WP[k + 1].y, xv.x, acc[0][r]); acc[0][r] = fmaf(WP[k + 2].x, xv.y, acc[0][r]);
  }
}
// Wave reduction of NC columns x RG rows with the ROWS dealt to the lanes by permlane swaps: a swap hands the partner the half of the rows
// it keeps and one add finishes the level -- no selects, no DPP on the two halving levels --, then four DPP adds inside the row of 16 lanes.
// Row(s) of a lane: dxs_row; every lane of a 16-lane row ends with the totals, its first lane (dxs_epl) publishes.  10 instructions per
// column at four rows (dx_reduce: 15).
template <int RG>
__device__ __forceinline__ int dxs_row(int lane, int q) {
  if (!DX_DIET) return dx_row<RG>(lane & 3, q);
  if (RG >= 8) return ((lane >> 4) & 1) * (RG / 4) + (lane >> 5) * (RG / 2) + q;
  if (RG == 4) return lane >> 4;
  if (RG == 2) return lane >> 5;
  return 0;
}
template <int RG>
__device__ __forceinline__ bool dxs_epl(int lane) {
  if (!DX_DIET) return lane < (RG >= 4 ? 4 : RG);
  return RG >= 4 ? (lane & 15) == 0 : RG == 2 ? (lane & 31) == 0 : lane == 0;
}
template <int NC, int RG>
__device__ __forceinline__ void dxs_reduce(const float (&a)[NC][RG], float (&out)[NC][DxRL<RG>::value], int lane) {
  if constexpr (!DX_DIET) { dx_reduce<NC, RG>(a, out, lane); return; }
  else {
    constexpr int RL = DxRL<RG>::value;
    float d[NC][RL];
    if constexpr (RG >= 4) {
      constexpr int Hh = RG / 2, Q = RG / 4;
      float b[NC][Hh];
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int p = 0; p < Hh; ++p) {        // lanes 0-31 keep rows 0 .. RG/2-1, lanes 32-63 rows RG/2 ..
          auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[c][p]), __float_as_uint(a[c][Hh + p]), false, false);
          b[c][p] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < Q; ++q) {         // even rows of 16 lanes keep the first half of those, odd rows the second
          auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(b[c][q]), __float_as_uint(b[c][Q + q]), false, false);
          d[c][q] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
    } else if constexpr (RG == 2) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[c][0]), __float_as_uint(a[c][1]), false, false);
        d[c][0] = dx_xrow16(__uint_as_float(r[0]) + __uint_as_float(r[1]));
      }
    } else {
#pragma unroll
      for (int c = 0; c < NC; ++c) d[c][0] = dx_xrow16(dx_xrow32(a[c][0]));
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int q = 0; q < RL; ++q) d[c][q] += DX_DPP0(d[c][q], 0xB1);      // quad_perm [1,0,3,2]
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int q = 0; q < RL; ++q) d[c][q] += DX_DPP0(d[c][q], 0x4E);      // quad_perm [2,3,0,1]
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int q = 0; q < RL; ++q) d[c][q] += DX_DPP0(d[c][q], 0x141);     // row_half_mirror
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int q = 0; q < RL; ++q) out[c][q] = d[c][q] + DX_DPP0(d[c][q], 0x140);      // row_mirror
  }
}

// A/B build -DDX_DLY_RT (tools/sweep_dx_delays.py): the sleep in front of every gather's first poll comes from a constant table that the host
// fills from TACO_DX_DLY before each launch, so that one build sweeps all ten gathers; the production build has the per-site immediates.
// Round 6, last: ONE build with the eleven sleeps in a constant table (-DDX_DLY_RT) and a coordinate descent over them on the decoder stage alone
// (tools/sweep_dx_delays.py, profiles/r06_sweep_dx_delays.txt: three passes, every site at 0..9 units, 3 x 20 launches per point) moved the
// stage from 1330 us (all sites at 5) to 1275 us per call: the gathers behind a SHORT producer phase want no sleep at all (the partial scores,
// the context, r*h of GRU 2, the next step's prenet 1: their curves fall monotonically to 0), the others stay at 4-6.  DX_DLY_TUNED = 0
// restores the two classes above for every site.
#ifndef DX_DLY_TUNED
#define DX_DLY_TUNED 1
#endif
__host__ __device__ constexpr int dx_site_delay(int site, int dflt, int RG = 4) {
  //                          p2 p3 rha ha sc ctx rh1 h1 rh2 h2 p1
  constexpr int tuned4[11] = {4, 5, 6, 5, 0, 0, 5, 5, 0, 4, 0};      // swept at C2 (four rows per group); also serves one and two rows (C1 / C5 got faster with it)
  constexpr int tuned8[11] = {7, 5, 3, 5, 0, 6, 5, 6, 4, 6, 2};      // swept on a 64-row pass (eight rows per group: longer producer phases): 1972 (all 5) / 1980 (the table above) -> 1940 us per call
  return DX_DLY_TUNED ? (RG == 8 ? tuned8[site] : tuned4[site]) : dflt;
}
#ifdef DX_DLY_RT
__constant__ int g_dx_dly[32];      // [0..10]: waves 0-3, [16..26]: waves 4-7 (sweep of a per-half table)
#define DX_DLY(site, dflt) (100 + (site))
#else
#define DX_DLY(site, dflt) dx_site_delay(site, dflt, RG)
#endif
struct DxRt {     // run-time state of a thread
  dx_gu32* err; bool wt; bool dead;
#ifdef DX_DLY_RT
  int dly[12];       // this wave's table (waves 0-3 / 4-7 may differ in the A/B build)
#endif
};
template <int DLY>
__device__ __forceinline__ void dx_first_poll_sleep(const DxRt& rt) {
#ifdef DX_DLY_RT
  if constexpr (DLY >= 100) { const int n = __builtin_amdgcn_readfirstlane(rt.dly[DLY - 100]); for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1); return; }
#endif
  if constexpr (DLY > 0 && DLY < 100) __builtin_amdgcn_s_sleep(DLY);
  (void)rt;
}
// WTC: the protocol as a compile-time constant (0: XCD-local, 1: write-through) where the kernel body is instantiated per protocol -- the
// run-time test (-1) costs a branch per publish, ~30 clocks each on the chain (round 5: 11 % of a post-net scan step, measured)
// DX_PUB_MODE (A/B, tools/ubench_mfma_stage): how an XCD-local granule leaves the lane.  0 (default): global_store_dwordx2 sc0; 1: a workgroup-scope
// atomic swap whose result is dropped (executed at the L2); 2: a non-temporal store; 3: a plain store
#ifndef DX_PUB_MODE
#define DX_PUB_MODE 0
#endif
__device__ __forceinline__ void dx_store_local(dx_gu64* p, unsigned long long g) {
  if (DX_PUB_MODE == 1) (void)__hip_atomic_exchange(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else if (DX_PUB_MODE == 2) __builtin_nontemporal_store(g, p);
  else if (DX_PUB_MODE == 3) *p = g;
  else __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);           // sc0: stays in this XCD's L2
}
template <int WTC = -1>
__device__ __forceinline__ void dx_publish(dx_gu64* p, float v, unsigned tag, const DxRt& rt) {
  const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
  if (WTC == 1 || (WTC < 0 && rt.wt)) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // sc1: write-through, any placement
  else dx_store_local(p, g);
}
// N granules of one lane at once (p + u*stride): ONE uniform branch on the protocol around all stores, so that the epilogue that
// produced the values stays a single basic block (a branch per value kept the compiler from interleaving the exp/rcp chains of
// independent units)
template <int N, int WTC = -1>
__device__ __forceinline__ void dx_publish_n(dx_gu64* p, int stride, const float (&v)[N], unsigned tag, const DxRt& rt) {
  unsigned long long g[N];
#pragma unroll
  for (int u = 0; u < N; ++u) g[u] = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v[u]);
  if (WTC == 1 || (WTC < 0 && rt.wt)) {
#pragma unroll
    for (int u = 0; u < N; ++u) __hip_atomic_store(p + u * stride, g[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
#pragma unroll
    for (int u = 0; u < N; ++u) dx_store_local(p + u * stride, g[u]);
  }
}
// keeps a value's computation where it is written (LLVM otherwise sinks an expensive operand of a select into a branch)
#define DX_PIN(x) asm("" : "+v"(x))
// Poll N granules (p0 + u*stride) until every one carries `tag` (L1-bypassing loads).  All N are re-requested together on every
// round, so a late producer costs one L2 round trip after its store lands, not one per granule.  Bounded.
template <int N, int DLY = 0>
__device__ __forceinline__ void dx_poll(const dx_gu64* p0, size_t stride, unsigned tag, float (&v)[N], DxRt& rt) {
  // (keeping a second round of requests in flight behind the one being examined was measured: 11.8 -> 14.2 us per decoder step at
  // C2 -- the extra L2 requests of 16 K pollers delay the very stores they are waiting for)
  unsigned long long g[N];
  unsigned spins = 0;
  dx_first_poll_sleep<DLY>(rt);
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int u = 0; u < N; ++u) g[u] = __hip_atomic_load(p0 + u * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int u = 0; u < N; ++u) ok = ok && ((unsigned)(g[u] >> 32) == tag);
    if (ok || rt.dead) break;
    if (DX_POLL_SLEEP) __builtin_amdgcn_s_sleep(DX_POLL_SLEEP);
    if ((++spins & 1023u) == 0) {
      if (spins >= DX_SPIN_LIMIT || __hip_atomic_load(rt.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(rt.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rt.dead = true;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] = __uint_as_float((unsigned)g[u]);
}
// all-gather of a published [RG][N] vector into the LDS state vector at column offset `off` (N a power of two);
// RES: dst2[r][n] = value + res[r][n] as well (ResidualWrapper output, tacotron.py:172)
// Two ADJACENT granules with one 16-byte request (each 8-byte half is one producer's single 8-byte store; a half that has not landed
// fails its own tag test and the pair is asked for again).  Used where a thread collects four or more granules per gather (eight rows
// per group; the 512-wide gate-gradient vectors of the BPTT kernel): measured on the exchange stage in isolation
// (tools/ubench_rowsets, variant T) 13.96 -> 12.63 us per step at eight rows, and nothing at two granules per thread (C2's decoder).
typedef unsigned long long dx_u64x2 __attribute__((ext_vector_type(2)));
template <int NP, int DLY = 0>
__device__ __forceinline__ void dx_poll_pairs(const dx_gu64* p0, size_t stride, unsigned tag, float (&v)[2 * NP], DxRt& rt) {
  dx_u64x2 g[NP];
  unsigned spins = 0;
  dx_first_poll_sleep<DLY>(rt);
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const dx_gu64* p = p0 + u * stride;
      asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(g[u]) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < NP; ++u) ok = ok && ((unsigned)(g[u][0] >> 32) == tag) && ((unsigned)(g[u][1] >> 32) == tag);
    if (ok || rt.dead) break;
    if ((++spins & 1023u) == 0) {
      if (spins >= DX_SPIN_LIMIT || __hip_atomic_load(rt.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(rt.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rt.dead = true;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NP; ++u) { v[2 * u] = __uint_as_float((unsigned)g[u][0]); v[2 * u + 1] = __uint_as_float((unsigned)g[u][1]); }
}
template <int RG, int N, bool RES, int LD = DXS_LD, int NT = DX_NT, int DLY = 0>
__device__ __forceinline__ void dx_gather(const dx_gu64* X, unsigned tag, float* st, int off, int off_res, int off2, int tid, DxRt& rt) {
  constexpr int NI = (RG * N + NT - 1) / NT;
  if constexpr (NI >= 4 && NI % 2 == 0 && (RG * N) % (2 * NT) == 0 && N % 2 == 0) {
    constexpr int NP = NI / 2;
    float v[NI];
    dx_poll_pairs<NP, DLY>(X + 2 * tid, (size_t)2 * NT, tag, v, rt);
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int i = 2 * (u * NT + tid), r = i / N, n = i % N;
      *reinterpret_cast<float2*>(st + r * LD + off + n) = make_float2(v[2 * u], v[2 * u + 1]);
      if (RES) {
        const float2 rs = *reinterpret_cast<const float2*>(st + r * LD + off_res + n);
        *reinterpret_cast<float2*>(st + r * LD + off2 + n) = make_float2(v[2 * u] + rs.x, v[2 * u + 1] + rs.y);
      }
    }
    return;
  }
  const bool act = (RG * N >= NT) || tid < RG * N;
  if (act) {
    float v[NI];
    dx_poll<NI, DLY>(X + tid, NT, tag, v, rt);
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int i = u * NT + tid;
      const int r = i / N, n = i % N;
      st[r * LD + off + n] = v[u];
      if (RES) st[r * LD + off2 + n] = v[u] + st[r * LD + off_res + n];
    }
  }
}

// The alignment normaliser of one row by ONE wave, lane = `cnt` (<= CMAX) consecutive positions starting at j0, values held in
// registers: sc (scores in, alignments out), alp (previous alignments).
//   bah_mon: monotonic_attention(mode='parallel') (A.10): p = sigmoid(e + bias); cp = exp(cumsum_excl(log(clip(1-p, tiny, 1))));
//            alpha = p * cp * cumsum(prev / clip(cp, 1e-10, 1));   else: softmax over T_in (no memory_sequence_length mask, A.8)
template <int CMAX>
__device__ __forceinline__ void dx_normalise(float* sc, const float* alp, int j0, int cnt, int att_type, float sbias) {
  float e[CMAX], pv[CMAX];
#pragma unroll
  for (int i = 0; i < CMAX; ++i) { e[i] = i < cnt ? sc[j0 + i] : 0.f; pv[i] = i < cnt ? alp[j0 + i] : 0.f; }
  if (att_type == 2) {
    float p[CMAX], ex[CMAX], run = 0.f;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
      p[i] = dx_sigmoid_fast(e[i] + sbias);
      ex[i] = run;                                      // exclusive prefix of the logs inside the lane's block
      if (i < cnt) run += 0.6931471805599453f * __builtin_amdgcn_logf(fminf(fmaxf(1.f - p[i], 1.17549435e-38f), 1.f));
    }
    const float off = dx_scan(run) - run;
    float cp[CMAX], in2[CMAX], run2 = 0.f;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
      cp[i] = __builtin_amdgcn_exp2f(1.4426950408889634f * (ex[i] + off));
      if (i < cnt) run2 += pv[i] * __builtin_amdgcn_rcpf(fminf(fmaxf(cp[i], 1e-10f), 1.f));
      in2[i] = run2;                                    // inclusive
    }
    const float off2 = dx_scan(run2) - run2;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) if (i < cnt) sc[j0 + i] = p[i] * cp[i] * (in2[i] + off2);
  } else {
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) if (i < cnt) mx = fmaxf(mx, e[i]);
    mx = dx_allmax(mx);
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) { e[i] = __builtin_amdgcn_exp2f(1.4426950408889634f * (e[i] - mx)); if (i < cnt) sm += e[i]; }
    sm = dx_allsum(sm);
    const float inv = 1.0f / sm;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) if (i < cnt) sc[j0 + i] = e[i] * inv;
  }
}
// the same through LDS scratch for long inputs (more than 8 positions per lane: T_in > 512)
__device__ __forceinline__ void dx_normalise_lds(float* sc, float* tmp, float* tmp2, const float* alp, int j0, int j1, int att_type, float sbias) {
  if (att_type == 2) {
    float run = 0.f;
    for (int j = j0; j < j1; ++j) {
      const float p = dx_sigmoid_fast(sc[j] + sbias);
      const float lg = 0.6931471805599453f * __builtin_amdgcn_logf(fminf(fmaxf(1.f - p, 1.17549435e-38f), 1.f));
      sc[j] = p; tmp[j] = run; run += lg;
    }
    const float off = dx_scan(run) - run;
    float run2 = 0.f;
    for (int j = j0; j < j1; ++j) {
      const float cp = __builtin_amdgcn_exp2f(1.4426950408889634f * (tmp[j] + off));
      tmp[j] = cp;
      run2 += alp[j] * __builtin_amdgcn_rcpf(fminf(fmaxf(cp, 1e-10f), 1.f));
      tmp2[j] = run2;
    }
    const float off2 = dx_scan(run2) - run2;
    for (int j = j0; j < j1; ++j) sc[j] = sc[j] * tmp[j] * (tmp2[j] + off2);
  } else {
    float mx = -INFINITY;
    for (int j = j0; j < j1; ++j) mx = fmaxf(mx, sc[j]);
    mx = dx_allmax(mx);
    float sm = 0.f;
    for (int j = j0; j < j1; ++j) { const float e = __builtin_amdgcn_exp2f(1.4426950408889634f * (sc[j] - mx)); sc[j] = e; sm += e; }
    sm = dx_allsum(sm);
    const float inv = 1.0f / sm;
    for (int j = j0; j < j1; ++j) sc[j] = sc[j] * inv;
  }
}

typedef __attribute__((address_space(3))) float dx_lds_float;
// 4 bytes per lane from global memory straight into LDS: the wave's 64 floats land at LDS byte address lds_dst (wave-uniform) + 4 * lane.
// Not counted by hipcc: the consumer waits (any later s_waitcnt vmcnt(0) of the issuing wave covers it: loads retire in order).
__device__ __forceinline__ void dx_load_lds4(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Census of a persistent launch of 256 (or 512: two per CU) workgroups: every workgroup reports the XCD it runs on (HW_REG_XCC_ID)
// and takes the next slot there; when all have arrived and every XCD hosts exactly gridDim.x / 8 of them, the XCD-local protocol is used (exchange stores stay
// in the XCD's L2) and a workgroup's place is (xcc, slot); otherwise -- or with force_wt -- places follow blockIdx and the stores
// are write-through.  out[0] = xcc or blockIdx % 8, out[1] = slot (0..31) or blockIdx / 8, out[2] = write-through?, out[3] = failed?
// errw[info0..info0+8] report the protocol and the per-XCD counts to the host.  Called by all threads; ends with a barrier.
__device__ __forceinline__ void dx_census(dx_gu32* ctl, dx_gu32* errw, int force_wt, int* out, int tid, int info0) {
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
      even = even && (__hip_atomic_load(ctl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x / DX_NGROUP);
    const bool fast = even && !force_wt;
    if (!ok) __hip_atomic_store(errw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[0] = fast ? (int)xcc : (int)(blockIdx.x & 7);
    out[1] = fast ? (int)slot : (int)(blockIdx.x >> 3);
    out[2] = fast ? 0 : 1;
    out[3] = ok ? 0 : 1;
    if (blockIdx.x == 0) {   // reported to the host (taco_debug_decoder_info): protocol used, workgroups seen per XCD
      errw[info0] = fast ? 1u : 2u;
      for (int i = 0; i < DX_NGROUP; ++i) errw[info0 + 1 + i] = __hip_atomic_load(ctl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
}

// 'simple' speaker term: rowbias[b][slot][n] = sum_k table[speaker_id[b]][k] * spkw[k][slot][n]   (S = speaker_embedding_size rows)
__global__ __launch_bounds__(DX_W) void k_dx_rowbias(const float* table, const int* speaker_id, const float* spkw, int S, float* rowbias) {
  const int b = blockIdx.x, n = threadIdx.x;
  const float* e = table + (size_t)(speaker_id ? speaker_id[b] : b) * S;      // no ids: `table` already holds the batch's rows
  for (int slot = 0; slot < DXRB_N; ++slot) {
    float acc = 0.f;
    for (int k = 0; k < S; ++k) acc = fmaf(e[k], spkw[((size_t)k * DXRB_N + slot) * DX_W + n], acc);
    rowbias[((size_t)b * DXRB_N + slot) * DX_W + n] = acc;
  }
}

#define DX_STAMP(slot)                                                                                     \
  do {                                                                                                     \
    if constexpr (TRACE) { if (tracer && t < DX_TRACE_STEPS) a.trace[t * DX_TRACE_SLOTS + (slot)] = (long long)__builtin_readcyclecounter(); } \
  } while (0)

// MAN: the manual-attention instantiation (a.manual != null); the plain one carries none of its branches, loads or address selects
// AW: attention_size (128 / 256 / 512); PD: decoder prenet layers (2: [256, 128]; 3: [256, 128, 64])
// WT: the census chose the write-through protocol (decided once per launch; the step carries no protocol branch); TRACE: shader-clock stamps
// (tools/time_decoder.py) -- the production instantiations have none of their branches
template <int RG, bool TAPE, bool MAN, int AW, int PD, bool WT, bool TRACE>
__device__ __forceinline__ void dx_body(const DxArgs& a, float* dx_smem, int group, int member, DxRt rt) {
  static_assert((AW == DX_W && PD == 2) || !TAPE, "the training forward exists at the reference widths only");
  constexpr int WTC = WT ? 1 : 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int Pr = DX_GROUP / RG;          // members per row in the attention phases
  constexpr int DC = DX_W / Pr;              // attention channels per member (8, 16, 32, 64)
  // score phase: the Pr members of a row form Pc channel blocks x Pp position blocks (the partial scores a member has to collect
  // grow with Pc * T_in, its own work with T_in / Pp * 256 / Pc)
  constexpr int Pc = Pr < 8 ? Pr : 8, Pp = Pr / Pc;
  constexpr int DS = AW / Pc;                // score channels per member (AW = 256: 32; 64 at RG = 8)
  static_assert(DS >= 8 && DS <= 64, "query columns per wave / the qv, vv, bq slots");
  constexpr int QC = DS / 8;                 // query columns per wave
  constexpr int QR = 4 * QC;                 // query-layer registers per thread
  constexpr int CH = DS / 4;                 // channels per lane in the score phase (a quad of lanes covers a position)
  constexpr int NP = Pc / 4;                 // score partials per lane in the gather (a quad covers the Pc channel blocks)
  const int T = a.T_in, Tpad = (T + 3) & ~3;
  const int TP = (T + Pr - 1) / Pr;          // positions whose history this member writes
  const int TS = (T + Pp - 1) / Pp;          // positions it scores

  // ---- LDS carve (dx_lds_floats mirrors this; its last 64 floats are the census words of the kernel wrapper) ----
  float* st = dx_smem;
  float* Kc = st + RG * DXS_LD;              // keys   [TS][DS]: the member's position block x channel block
  float* Vc = Kc + (size_t)TS * DS;          // values [T][DC]: all positions x its context channels
  float* sc = Vc + (size_t)T * DC;
  float* tmp = sc + Tpad;
  float* tmp2 = tmp + Tpad;
  float* alp = tmp2 + Tpad;
  float* qv = alp + Tpad;                    // query (+ attention_b) of the member's channels
  float* vv = qv + 64;                       // attention_v of the member's channels
  float* bq = vv + 64;
  float* cpart = bq + 64;                    // [DX_NW][64]
  float* bl = cpart + DX_NW * 64;            // [DXB_N][DX_NW]
  float* rbl = bl + DXB_N * DX_NW;           // [DXRB_N][RG][DX_NW]
  float* mrow = rbl + DXRB_N * RG * DX_NW;   // [roundup(T, 64)] manual alignments of the step
  float* tfb = mrow + ((T + 63) & ~63);      // [RG][DX_W] teacher frames (TAPE only)
  const int ga = (group - a.grp0) & 7;
  if (ga >= a.ngroups || member >= DX_GROUP) return;
  const int row0 = ga * RG;
  if (row0 >= a.B) return;
  const int arow = member / Pr, asl = member % Pr;       // attention role: local row, slot in the row
  const int cb = asl % Pc, pb = asl / Pc;                // score role: channel block, position block
  const int ps0 = pb * TS, psn = max(0, min(T - ps0, TS));
  const int brow = row0 + arow;                          // its batch row (may be >= B: padding)
  const bool tracer = TRACE && a.trace && ga == 0 && member == a.trc_member && tid == a.trc_tid;

  // ---- weights, resident for the whole loop ----
  taco_f32x2 WP[DX_NWP];      // register pairs in the order the passes consume them (dxw_src)
  {
    const float* wp = a.wpack + ((size_t)member * DX_NREG) * DX_NT + tid;
#pragma unroll
    for (int k = 0; k < DX_NWP; ++k) WP[k] = (taco_f32x2){wp[(size_t)dxw_src(k, 0, PD) * DX_NT], wp[(size_t)dxw_src(k, 1, PD) * DX_NT]};
  }
  if (TAPE && a.p1o_raw) {
    const float* wp = a.p1o_raw + ((size_t)member * 4) * DX_NT + tid;
    WP[DXR_P1O / 2] = (taco_f32x2){wp[0], wp[(size_t)DX_NT]};
    WP[DXR_P1O / 2 + 1] = (taco_f32x2){wp[(size_t)2 * DX_NT], wp[(size_t)3 * DX_NT]};
  }
  taco_f32x2 WQP[QR / 2];     // query layer: columns cb*DS + wave*QC + i, inputs 4*lane..4*lane+3; pair 4j + e = columns (2j, 2j + 1), input e
  {
    const float* wp = a.qpack + ((size_t)member * QR) * DX_NT + tid;
#pragma unroll
    for (int j = 0; j < QC / 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) WQP[4 * j + e] = (taco_f32x2){wp[(size_t)(8 * j + e) * DX_NT], wp[(size_t)(8 * j + 4 + e) * DX_NT]};
  }
  const DxX xl = dx_xlayout(RG, T);
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)ga * xl.total;
  const int NCF = (a.rM + DX_GROUP - 1) / DX_GROUP;      // frame-projection columns per member (<= 16)

  // ---- stationary attention memory of the member's row: keys (its position block x score channels), values (all positions x
  // context channels) ----
  for (int i = tid; i < TS * (DS / 4); i += DX_NT) {
    const int j = i / (DS / 4), d4 = i % (DS / 4);
    float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (brow < a.B && j < psn) k4 = *reinterpret_cast<const float4*>(a.keys + ((size_t)brow * T + ps0 + j) * AW + cb * DS + 4 * d4);
    *reinterpret_cast<float4*>(Kc + (size_t)j * DS + 4 * d4) = k4;
  }
  for (int i = tid; i < T * (DC / 4); i += DX_NT) {
    const int j = i / (DC / 4), d4 = i % (DC / 4);
    float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (brow < a.B) v4 = *reinterpret_cast<const float4*>(a.values + ((size_t)brow * T + j) * DX_W + asl * DC + 4 * d4);
    *reinterpret_cast<float4*>(Vc + (size_t)j * DC + 4 * d4) = v4;
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
  if (tid < DS) { vv[tid] = a.att_v[cb * DS + tid]; bq[tid] = a.att_b ? a.att_b[cb * DS + tid] : 0.f; qv[tid] = 0.f; }
  if (tid < DXB_N * DX_NW) {     // own-column biases: bl[slot][wave]
    const int e = tid / DX_NW, w = tid % DX_NW, n8 = member * 8 + w;
    float v = 0.f;
    switch (e) {
      case DXB_P1: v = (TAPE && a.p1o_raw) ? a.b_p1_0[n8] : a.b_p1c[n8]; break;
      case DXB_P2: if (w < 4) v = a.b_p2[member * 4 + w]; break;
      case DXB_AR: v = a.b_ag[n8]; break;
      case DXB_AU: v = a.b_ag[DX_W + n8]; break;
      case DXB_AC: v = a.b_ac[n8]; break;
      case DXB_G1R: v = a.b_g1f[n8]; break;
      case DXB_G1U: v = a.b_g1f[DX_W + n8]; break;
      case DXB_G1X: v = a.b_g1f[2 * DX_W + n8]; break;
      case DXB_O0: v = a.b_g1f[3 * DX_W + n8]; break;
      case DXB_G1C: v = a.b_g1c[n8]; break;
      case DXB_G2R: v = a.b_g2g[n8]; break;
      case DXB_G2U: v = a.b_g2g[DX_W + n8]; break;
      case DXB_G2C: v = a.b_g2c[n8]; break;
      case DXB_F0: if (w < NCF && member * NCF + w < a.rM) v = a.b_f[member * NCF + w]; break;
      case DXB_F1: if (w + 8 < NCF && member * NCF + w + 8 < a.rM) v = a.b_f[member * NCF + w + 8]; break;
      case DXB_P3: if (PD == 3 && w < 2) v = a.b_p3[member * 2 + w]; break;
      default: break;
    }
    bl[tid] = v;
  }
  for (int i = tid; i < DXRB_N * RG * DX_NW; i += DX_NT) {      // rbl[slot][row][wave] = the column's bias + the row's speaker term: ONE read per epilogue
    const int e = i / (RG * DX_NW), r = (i / DX_NW) % RG, w = i % DX_NW, b = row0 + r, n8 = member * 8 + w;
    float v = 0.f;
    switch (e) {
      case DXRB_AR: v = a.b_ag[n8]; break;
      case DXRB_AU: v = a.b_ag[DX_W + n8]; break;
      case DXRB_AX: v = 0.f; break;                       // (the candidate's bias is added with its h part)
      case DXRB_G1R: v = a.b_g1f[n8]; break;
      case DXRB_G1U: v = a.b_g1f[DX_W + n8]; break;
      case DXRB_G1X: v = a.b_g1f[2 * DX_W + n8]; break;
      default: v = a.b_g1f[3 * DX_W + n8]; break;         // DXRB_O0
    }
    if (a.rowbias && b < a.B) v += a.rowbias[((size_t)b * DXRB_N + e) * DX_W + n8];
    rbl[i] = v;
  }
  const float sbias = (a.att_type == 2 && a.score_bias) ? a.score_bias[0] : 0.f;
  constexpr bool man = MAN;
  const unsigned mrow_lds = (unsigned)(size_t)(dx_lds_float*)mrow;
  const unsigned tfb_lds = (unsigned)(size_t)(dx_lds_float*)tfb;
  if (TAPE) for (int i = tid; i < RG * DX_W; i += DX_NT) tfb[i] = 0.f;
  __syncthreads();

  // epilogue role: the lanes of quad 0 own the outputs (rows dx_row(lane, q), column 8*member + wave) of every 256-wide stage
  constexpr int RL = DxRL<RG>::value;
  const bool epl = dxs_epl<RG>(lane);
  const int en = member * 8 + wave;
  int erow[RL];
#pragma unroll
  for (int q = 0; q < RL; ++q) erow[q] = dxs_row<RG>(lane, q);
#define DX_RB(slot, q) rbl[((slot) * RG + erow[q]) * DX_NW + wave]
  float g_u[RL], g_cx[RL], g_h[RL], g_o0[RL];            // live between the two stages of a GRU cell
  // tape (TAPE): the lane that owns (row, column) of a stage writes it; trow = float offset of step 0 of the lane's row in a [B, n, 256] array
  // (32-bit element offsets: the host checks DXT_N * tstride < 2^31, so an address is the SGPR base + one VGPR)
  unsigned trow[RL];
  bool tval[RL];
  float g_o1[RL];
  const unsigned tstr = (unsigned)a.tstride;
#pragma unroll
  for (int q = 0; q < RL; ++q) {
    trow[q] = (unsigned)(row0 + erow[q]) * (unsigned)a.n * DX_W + (unsigned)en;
    tval[q] = TAPE && a.tape && epl && (row0 + erow[q] < a.B);
    g_o1[q] = 0.f;
  }
#define DX_TAPE(slot, q, val) do { if (TAPE && tval[q]) a.tape[(unsigned)(slot) * tstr + trow[q] + (unsigned)t * DX_W] = (val); } while (0)
  if (TAPE) {   // step 0's prenet layer 1 is the constant relu(b1)
    const int t = 0;
#pragma unroll
    for (int q = 0; q < RL; ++q) DX_TAPE(DXT_P1, q, fmaxf(a.b_p1_0[en], 0.f));
  }
  // (accumulators of the passes that run ahead of their stage live across exactly one gather)

  // Static priority for the FIRST wave of every SIMD (waves 0-3; their SIMD partners are waves 4-7): the older wave wins the issue arbitration
  // by age anyway, with s_setprio 1 it also wins every tie -- the stage's first four columns are published earlier and the vector is complete
  // sooner.  A/B of variant builds on one box (profiles/r06_ab_priority.txt): decoder alone 1.305 -> 1.289-1.295 ms at levels 1, 2, 3 alike;
  // priority for the YOUNGER half instead 1.303 (no gain; in the post-net scan it costs 38 us, and the older half gains nothing there).
  if (DX_PRIO_OLD && wave < 4) __builtin_amdgcn_s_setprio(DX_PRIO_OLD);
  const int tid_outer = tid, lane_outer = lane;
  for (int t = 0; t < a.n; ++t) {
    const unsigned tag = (unsigned)t + 1u;
    // per-iteration opaque copies of the thread indices: the address arithmetic of the ~40 LDS / exchange accesses of a step is
    // loop invariant, and hoisted out of the loop it occupies (and spills) dozens of registers next to the resident weights
    int tid = tid_outer, lane = lane_outer;
    asm volatile("" : "+v"(tid), "+v"(lane));
    int erow[RL];                  // (recomputed from the opaque lane: addresses derived from it -- mel / stop-flag / tape rows -- stay out of the hoisted set)
#pragma unroll
    for (int q = 0; q < RL; ++q) erow[q] = dxs_row<RG>(lane, q);
    DX_STAMP(0);
    if (man) {   // this step's manual alignments of the member's row -> LDS (consumed three exchanges from now)
      const float* src = a.manual + ((size_t)min(brow, a.B - 1) * a.n + t) * T;
      for (int j0 = wave * 64; j0 < T; j0 += DX_NT)
        dx_load_lds4(src + min(j0 + lane, T - 1), __builtin_amdgcn_readfirstlane(mrow_lds + (unsigned)j0 * 4u));
    }
    if (TAPE && a.teacher && wave < RG && t + 1 < a.n) {   // teacher frame t (the input of step t + 1's prenet) of row `wave` -> LDS
      const float* src = a.teacher + ((size_t)min(row0 + wave, a.B - 1) * a.n + t) * a.mels;
      for (int j0 = 0; j0 < a.mels; j0 += 64)
        dx_load_lds4(src + min(j0 + lane, a.mels - 1), __builtin_amdgcn_readfirstlane(tfb_lds + (unsigned)(wave * DX_W + j0) * 4u));
    }
    // ================= prenet layer 2 (modules.py:18-25); LDS T = prenet layer 1 =================
    if (wave < 4) {
      float acc[1][RG], s[1][RL];
      dx_zero<1, RG>(acc);
      dxw_single<DXR_P2, RG>(WP, st + DXS_T, lane, acc);
      dxs_reduce<1, RG>(acc, s, lane);
      if (epl) {
#pragma unroll
        for (int q = 0; q < RL; ++q) {
          const float p2v = fmaxf(s[0][q] + bl[DXB_P2 * DX_NW + wave], 0.f);
          dx_publish<WTC>(X + xl.p2 + erow[q] * DX_P2 + member * 4 + wave, p2v, tag, rt);
          if (TAPE && tval[q] && a.tp_p2) a.tp_p2[((size_t)(row0 + erow[q]) * a.n + t) * a.ld_p2 + member * 4 + wave] = p2v;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RL; ++q) g_h[q] = st[erow[q] * DXS_LD + DXS_HATT + en];
    taco_f32x2 ag01[RG];       // attention GRU: (r, u) as a pair, candidate-x; the h rows run ahead of the prenet output
    float ag2[1][RG];
    dxq_zero<RG>(ag01);
    dx_zero<1, RG>(ag2);
    dxw_pair<DXR_AGH, RG>(WP, st + DXS_HATT, lane, ag01);
    dx_gather<RG, DX_P2, false, DXS_LD, DX_NT, DX_DLY(0, DX_POLL_DELAY_B)>(X + xl.p2, tag, st, DXS_P2, 0, 0, tid, rt);
    __syncthreads();
    if constexpr (PD == 3) {
      // ================= prenet layer 3 (64 columns, two per member); its output takes the OUT2 slot, dead until the end of the step =================
      if (wave < 2) {
        float acc[1][RG], s[1][RL];
        dx_zero<1, RG>(acc);
        dxw_p3<RG>(WP, st + DXS_P2, lane, acc);
        dxs_reduce<1, RG>(acc, s, lane);
        if (epl) {
#pragma unroll
          for (int q = 0; q < RL; ++q)
            dx_publish<WTC>(X + xl.p3 + erow[q] * DX_P3 + member * 2 + wave, fmaxf(s[0][q] + bl[DXB_P3 * DX_NW + wave], 0.f), tag, rt);
        }
      }
      dx_gather<RG, DX_P3, false, DXS_LD, DX_NT, DX_DLY(1, DX_FIRST_POLL_DELAY)>(X + xl.p3, tag, st, DXS_OUT2, 0, 0, tid, rt);
      __syncthreads();
    }
    DX_STAMP(1);
    // ================= attention GRUCell (tacotron.py:127-130; A.6): gates, then candidate =================
    {
      float s[3][RL];
      if constexpr (PD == 3) dxw_agx1<RG>(WP, st + DXS_OUT2, lane, ag01, ag2);
      else dxw_agx2<RG>(WP, st + DXS_P2, lane, ag01, ag2);
      float aga[3][RG];
#pragma unroll
      for (int r = 0; r < RG; ++r) { aga[0][r] = ag01[r].x; aga[1][r] = ag01[r].y; aga[2][r] = ag2[0][r]; }
      dxs_reduce<3, RG>(aga, s, lane);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const float rg = dx_sigmoid_fast(s[0][q] + DX_RB(DXRB_AR, q));
        g_u[q] = dx_sigmoid_fast(s[1][q] + DX_RB(DXRB_AU, q));
        g_cx[q] = s[2][q] + DX_RB(DXRB_AX, q);
        if (epl) dx_publish<WTC>(X + xl.rha + erow[q] * DX_W + en, rg * g_h[q], tag, rt);
        DX_TAPE(DXT_RA, q, rg); DX_TAPE(DXT_UA, q, g_u[q]); DX_TAPE(DXT_RHA, q, rg * g_h[q]);
      }
    }
    dx_gather<RG, DX_W, false, DXS_LD, DX_NT, DX_DLY(2, DX_FIRST_POLL_DELAY)>(X + xl.rha, tag, st, DXS_T, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(2);
    {
      float acc[1][RG], s[1][RL];
      dx_zero<1, RG>(acc);
      dxw_single<DXR_AC, RG>(WP, st + DXS_T, lane, acc);
      dxs_reduce<1, RG>(acc, s, lane);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const float c = taco_tanh_fast(g_cx[q] + s[0][q] + bl[DXB_AC * DX_NW + wave]);
        const float hn = g_u[q] * g_h[q] + (1.f - g_u[q]) * c;
        if (epl) dx_publish<WTC>(X + xl.ha + erow[q] * DX_W + en, hn, tag, rt);
        DX_TAPE(DXT_CA, q, c); DX_TAPE(DXT_HA, q, hn);
      }
    }
#pragma unroll
    for (int q = 0; q < RL; ++q) g_h[q] = st[erow[q] * DXS_LD + DXS_H1 + en];
    dx_gather<RG, DX_W, false, DXS_LD, DX_NT, DX_DLY(3, DX_FIRST_POLL_DELAY)>(X + xl.ha, tag, st, DXS_HATT, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(3);
    // ================= attention (rnn_wrappers.py:304-341) =================
    if (!man) {
      {  // query for the member's own score channels of its row: columns cb*DS + wave*QC + i
        const float4 xv = *reinterpret_cast<const float4*>(st + arow * DXS_LD + DXS_HATT + 4 * lane);
        float qa[QC][1], qs[QC][1];
        static_assert(QC % 2 == 0, "query columns in pairs");
#pragma unroll
        for (int i = 0; i < QC; i += 2) {        // columns (i, i + 1) in one v_pk_fma_f32 per input
          taco_f32x2 q2 = {0.f, 0.f};
          DXQ_FMA(q2, xv.x, WQP[2 * i + 0]); DXQ_FMA(q2, xv.y, WQP[2 * i + 1]); DXQ_FMA(q2, xv.z, WQP[2 * i + 2]); DXQ_FMA(q2, xv.w, WQP[2 * i + 3]);
          qa[i][0] = q2.x; qa[i + 1][0] = q2.y;
        }
        dxs_reduce<QC, 1>(qa, qs, lane);
        float qme = qs[0][0];
#pragma unroll
        for (int i = 1; i < QC; ++i) qme = (lane >= i) ? qs[i][0] : qme;
        if (lane < QC) qv[wave * QC + lane] = qme + bq[wave * QC + lane];
        if (TAPE && a.tape && lane < QC && pb == 0 && brow < a.B)      // processed query W_q . h (without attention_b), once per channel block
          a.tape[(unsigned)DXT_Q * tstr + ((unsigned)brow * (unsigned)a.n + (unsigned)t) * DX_W + cb * DS + wave * QC + lane] = qme;
      }
      __syncthreads();
      DX_STAMP(4);
      {  // partial scores over the member's channel block (A.9): a quad of lanes per encoder position, CH channels per lane
        const int jl = lane >> 2, cp = lane & 3;
        constexpr bool QV_REG = CH < 16;     // (16 channels per lane = eight rows per group: the 32 registers are not there; read per use)
        float qreg[QV_REG ? CH : 1], vreg[QV_REG ? CH : 1];
        if constexpr (QV_REG) {
#pragma unroll
          for (int c = 0; c < CH; ++c) { qreg[c] = qv[cp * CH + c]; vreg[c] = vv[cp * CH + c]; }
        }
        for (int j0 = 0; j0 < psn; j0 += 16 * DX_NW) {
          const int j = j0 + wave * 16 + jl;
          const int jc = j < psn ? j : psn - 1;
          const float* kp = Kc + (size_t)jc * DS + cp * CH;
          float e = 0.f;
#pragma unroll
          for (int c = 0; c < CH; c += 4) {
            const float4 k4 = *reinterpret_cast<const float4*>(kp + c);
            if constexpr (QV_REG) {
              e += vreg[c] * taco_tanh_fast(k4.x + qreg[c]) + vreg[c + 1] * taco_tanh_fast(k4.y + qreg[c + 1]) +
                   vreg[c + 2] * taco_tanh_fast(k4.z + qreg[c + 2]) + vreg[c + 3] * taco_tanh_fast(k4.w + qreg[c + 3]);
            } else {
              const float4 q4 = *reinterpret_cast<const float4*>(qv + cp * CH + c), v4 = *reinterpret_cast<const float4*>(vv + cp * CH + c);
              e += v4.x * taco_tanh_fast(k4.x + q4.x) + v4.y * taco_tanh_fast(k4.y + q4.y) +
                   v4.z * taco_tanh_fast(k4.z + q4.z) + v4.w * taco_tanh_fast(k4.w + q4.w);
            }
          }
          e = dx_quadsum(e);
          if (cp == 0 && j < psn) dx_publish<WTC>(X + xl.sc + (size_t)(arow * Pc + cb) * T + ps0 + j, e, tag, rt);
        }
      }
    }
    // ahead of its turn: GRU 1 (concat projection folded in): h1 rows of the gates, then the h_att rows of r, u, candidate-x, o0.
    // With eight rows per group the 32 accumulators would have to live across the whole attention phase next to 138 resident weights:
    // there they are formed after the context has arrived instead (two more passes on the critical path, no spilled registers).
    constexpr bool G1_AHEAD = RG < 8;
    taco_f32x2 g1p0[RG], g1p1[RG];      // (r, u) and (candidate-x, o0)
    dxq_zero<RG>(g1p0); dxq_zero<RG>(g1p1);
    if (G1_AHEAD) {
      dxw_pair<DXR_G1H, RG>(WP, st + DXS_H1, lane, g1p0);
      dxw_quad<DXR_G1A, RG>(WP, st + DXS_HATT, lane, g1p0, g1p1);
    }
    int aoff = 0;                  // the step's alignments are sc[aoff + j]: computed (sc itself) or manual (mrow, a region of the same LDS array)
    if (!man) {
      {  // gather the row's partial scores and sum them over the Pc channel blocks (fixed order)
        const int jl = lane >> 2, part = lane & 3;
        for (int j0 = 0; j0 < T; j0 += 16 * DX_NW) {
          const int j = j0 + wave * 16 + jl;
          float s = 0.f;
          if (j < T) {
            float v[NP];
            dx_poll<NP, DX_DLY(4, DX_POLL_DELAY_B)>(X + xl.sc + (size_t)(arow * Pc + part * NP) * T + j, (size_t)T, tag, v, rt);
#pragma unroll
            for (int u = 0; u < NP; ++u) s += v[u];
          }
          s = dx_quadsum(s);
          if (part == 0 && j < T) sc[j] = s;
        }
      }
      __syncthreads();
      DX_STAMP(5);
      if (TAPE && a.tp_e && tid < TP && asl * TP + tid < T && brow < a.B)     // raw scores (before the normaliser) of the member's positions
        a.tp_e[((size_t)brow * a.n + t) * T + asl * TP + tid] = sc[asl * TP + tid];
      if (TAPE) __syncthreads();                                              // ... read before wave 0 normalises in place
      if (wave == 0) {   // normaliser over the whole row, redundantly on each of the row's members (lane = C consecutive positions)
        const int C = (T + 63) >> 6;
        const int j0 = lane * C;
        if (C <= 2) dx_normalise<2>(sc, alp, j0, min(C, max(T - j0, 0)), a.att_type, sbias);
        else if (C <= 8) dx_normalise<8>(sc, alp, j0, min(C, max(T - j0, 0)), a.att_type, sbias);
        else dx_normalise_lds(sc, tmp, tmp2, alp, j0, min(j0 + C, T), a.att_type, sbias);
      }
      __syncthreads();
    } else {
      // the row requested at the top of the step: every wave makes sure its own part has landed, the barrier publishes all parts
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      aoff = (int)(mrow - sc);
    }
    {  // alignment state + history (tacotron.py:238-239 layout) for the member's block of positions; context channel block
      for (int j = tid; j < T; j += DX_NT) alp[j] = sc[aoff + j];
      const int p0 = asl * TP;
      if (tid < TP && p0 + tid < T && brow < a.B) {
        if (a.hist) a.hist[((size_t)brow * T + p0 + tid) * a.n + t] = sc[aoff + p0 + tid];
        if (TAPE && a.tp_alpha) a.tp_alpha[((size_t)brow * (a.n + 1) + t + 1) * T + p0 + tid] = sc[aoff + p0 + tid];
      }
      constexpr int JL = 64 / DC;                    // positions handled side by side inside a wave
      const int d = lane % DC, jsub = lane / DC;
      float part = 0.f;
      for (int j = wave * JL + jsub; j < T; j += DX_NW * JL) part = fmaf(sc[aoff + j], Vc[(size_t)j * DC + d], part);
      if (DC <= 32) part = dx_xrow32(part);
      if (DC <= 16) part = dx_xrow16(part);
      if (DC <= 8) part += DX_DPP0(part, 0x128);
      if (lane < DC) cpart[wave * 64 + lane] = part;
    }
    __syncthreads();
    if (tid < DC) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < DX_NW; ++w) s += cpart[w * 64 + tid];
      dx_publish<WTC>(X + xl.ctx + arow * DX_W + asl * DC + tid, s, tag, rt);
      if (TAPE && a.tp_ctx && brow < a.B) a.tp_ctx[((size_t)brow * a.n + t) * a.ld_ctx + asl * DC + tid] = s;
    }
    dx_gather<RG, DX_W, false, DXS_LD, DX_NT, DX_DLY(5, DX_FIRST_POLL_DELAY)>(X + xl.ctx, tag, st, DXS_CTX, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(6);
    // ================= concat projection folded into residual GRU 1 (rnn_wrappers.py:405-415; tacotron.py:166-172) =================
    {
      float s[4][RL];
      if (!G1_AHEAD) {
        dxq_zero<RG>(g1p0); dxq_zero<RG>(g1p1);
        dxw_pair<DXR_G1H, RG>(WP, st + DXS_H1, lane, g1p0);
        dxw_quad<DXR_G1A, RG>(WP, st + DXS_HATT, lane, g1p0, g1p1);
      }
      dxw_quad<DXR_G1B, RG>(WP, st + DXS_CTX, lane, g1p0, g1p1);
      float g1a[4][RG];
#pragma unroll
      for (int r = 0; r < RG; ++r) { g1a[0][r] = g1p0[r].x; g1a[1][r] = g1p0[r].y; g1a[2][r] = g1p1[r].x; g1a[3][r] = g1p1[r].y; }
      dxs_reduce<4, RG>(g1a, s, lane);
      DX_STAMP(12);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const float rg = dx_sigmoid_fast(s[0][q] + DX_RB(DXRB_G1R, q));
        g_u[q] = dx_sigmoid_fast(s[1][q] + DX_RB(DXRB_G1U, q));
        g_cx[q] = s[2][q] + DX_RB(DXRB_G1X, q);
        g_o0[q] = s[3][q] + DX_RB(DXRB_O0, q);
        if (epl) dx_publish<WTC>(X + xl.rh1 + erow[q] * DX_W + en, rg * g_h[q], tag, rt);
        DX_TAPE(DXT_R1, q, rg); DX_TAPE(DXT_U1, q, g_u[q]); DX_TAPE(DXT_RH1, q, rg * g_h[q]); DX_TAPE(DXT_O0, q, g_o0[q]);
      }
      DX_STAMP(13);
    }
    dx_gather<RG, DX_W, false, DXS_LD, DX_NT, DX_DLY(6, DX_FIRST_POLL_DELAY)>(X + xl.rh1, tag, st, DXS_T, 0, 0, tid, rt);
    DX_STAMP(14);
    __syncthreads();
    DX_STAMP(7);
    {
      float acc[1][RG], s[1][RL];
      dx_zero<1, RG>(acc);
      dxw_single<DXR_G1C, RG>(WP, st + DXS_T, lane, acc);
      dxs_reduce<1, RG>(acc, s, lane);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const float c = taco_tanh_fast(g_cx[q] + s[0][q] + bl[DXB_G1C * DX_NW + wave]);
        const float hn = g_u[q] * g_h[q] + (1.f - g_u[q]) * c;
        g_o1[q] = hn + g_o0[q];
        if (epl) {
          dx_publish<WTC>(X + xl.h1 + erow[q] * DX_W + en, hn, tag, rt);
          dx_publish<WTC>(X + xl.o1 + erow[q] * DX_W + en, hn + g_o0[q], tag, rt);       // ResidualWrapper: cell output + cell input
        }
        DX_TAPE(DXT_C1, q, c); DX_TAPE(DXT_H1, q, hn); DX_TAPE(DXT_O1, q, g_o1[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < RL; ++q) g_h[q] = st[erow[q] * DXS_LD + DXS_H2 + en];
    taco_f32x2 g2p[RG];        // GRU 2: (r, u) as a pair, candidate-x; the h2 rows run ahead of GRU 1's output (not at eight rows per group: registers)
    float g2c[1][RG];
    dxq_zero<RG>(g2p);
    dx_zero<1, RG>(g2c);
    if (G1_AHEAD) dxw_pair<DXR_G2H, RG>(WP, st + DXS_H2, lane, g2p);
    dx_gather<RG, DX_W, false, DXS_LD, DX_NT, DX_DLY(7, DX_POLL_DELAY_B)>(X + xl.h1, tag, st, DXS_H1, 0, 0, tid, rt);
    dx_gather<RG, DX_W, false, DXS_LD, DX_NT, 0>(X + xl.o1, tag, st, DXS_OUT1, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(8);
    // ================= residual GRU 2 =================
    {
      float s[3][RL];
      if (!G1_AHEAD) dxw_pair<DXR_G2H, RG>(WP, st + DXS_H2, lane, g2p);
      dxw_pair_single<DXR_G2X, DXR_G2X + 8, RG>(WP, st + DXS_OUT1, lane, g2p, g2c);
      float g2a[3][RG];
#pragma unroll
      for (int r = 0; r < RG; ++r) { g2a[0][r] = g2p[r].x; g2a[1][r] = g2p[r].y; g2a[2][r] = g2c[0][r]; }
      dxs_reduce<3, RG>(g2a, s, lane);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const float rg = dx_sigmoid_fast(s[0][q] + bl[DXB_G2R * DX_NW + wave]);
        g_u[q] = dx_sigmoid_fast(s[1][q] + bl[DXB_G2U * DX_NW + wave]);
        g_cx[q] = s[2][q];
        if (epl) dx_publish<WTC>(X + xl.rh2 + erow[q] * DX_W + en, rg * g_h[q], tag, rt);
        DX_TAPE(DXT_R2, q, rg); DX_TAPE(DXT_U2, q, g_u[q]); DX_TAPE(DXT_RH2, q, rg * g_h[q]);
      }
    }
    dx_gather<RG, DX_W, false, DXS_LD, DX_NT, DX_DLY(8, DX_FIRST_POLL_DELAY)>(X + xl.rh2, tag, st, DXS_T, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(9);
    {
      float acc[1][RG], s[1][RL];
      dx_zero<1, RG>(acc);
      dxw_single<DXR_G2C, RG>(WP, st + DXS_T, lane, acc);
      dxs_reduce<1, RG>(acc, s, lane);
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const float c = taco_tanh_fast(g_cx[q] + s[0][q] + bl[DXB_G2C * DX_NW + wave]);
        const float hn = g_u[q] * g_h[q] + (1.f - g_u[q]) * c;
        if (epl) dx_publish<WTC>(X + xl.h2 + erow[q] * DX_W + en, hn, tag, rt);
        DX_TAPE(DXT_C2, q, c); DX_TAPE(DXT_H2, q, hn); DX_TAPE(DXT_O2, q, hn + g_o1[q]);
      }
    }
    // ahead of its turn: next step's prenet layer 1, context rows
    float p1a[1][RG];
    dx_zero<1, RG>(p1a);
    if (G1_AHEAD) dxw_single<DXR_P1C, RG>(WP, st + DXS_CTX, lane, p1a);
    dx_gather<RG, DX_W, true, DXS_LD, DX_NT, DX_DLY(9, DX_POLL_DELAY_B)>(X + xl.h2, tag, st, DXS_H2, DXS_OUT1, DXS_OUT2, tid, rt);
    __syncthreads();
    DX_STAMP(10);
    // ================= prenet layer 1 of step t+1 (composite: frame projection folded in, helpers.py:31) and the frame
    // projection of step t (tacotron.py:178-179), straight into the mel buffer =================
    auto store_frame = [&](int q, float y0, float y1, bool share) {      // the lane's two frame-projection columns of row erow[q]: mel buffer, stop rule, (own_fb) the exchange
      const int b = row0 + erow[q];
      const int n0 = member * NCF + wave, n1 = n0 + 8;
      const bool v0 = wave < NCF && n0 < a.rM, v1 = wave + 8 < NCF && n1 < a.rM;
      if (b < a.B) {
        float* mrow = a.mel + (size_t)b * a.n * a.rM + (size_t)t * a.rM;
        bool nzf = false;
        if (v0) { mrow[n0] = y0; nzf = nzf || (y0 != 0.f); }
        if (v1) { mrow[n1] = y1; nzf = nzf || (y1 != 0.f); }
        if (nzf) a.nz[(size_t)t * a.B + b] = 1;                            // stop rule helpers.py:29
      }
      if (share) {                                                         // last of the r frames -> every member's copy of the next prenet input
        const int f0 = a.rM - a.mels;
        if (v0 && n0 >= f0) dx_publish<WTC>(X + xl.fb + erow[q] * DX_P2 + (n0 - f0), y0, tag, rt);
        if (v1 && n1 >= f0) dx_publish<WTC>(X + xl.fb + erow[q] * DX_P2 + (n1 - f0), y1, tag, rt);
      }
    };
    if (TAPE && a.own_fb) {
      // rnn_decoder_test_mode: the frame first (its own stage and exchange), then the prenet layer over it
      {
        taco_f32x2 fp[RG];
        float fa[2][RG], s[2][RL];
        dxq_zero<RG>(fp);
        dxw_pair<DXR_F, RG>(WP, st + DXS_OUT2, lane, fp);
#pragma unroll
        for (int r = 0; r < RG; ++r) { fa[0][r] = fp[r].x; fa[1][r] = fp[r].y; }
        dxs_reduce<2, RG>(fa, s, lane);
        if (epl) {
#pragma unroll
          for (int q = 0; q < RL; ++q) store_frame(q, s[0][q] + bl[DXB_F0 * DX_NW + wave], s[1][q] + bl[DXB_F1 * DX_NW + wave], t + 1 < a.n);
        }
      }
      if (t + 1 < a.n) {
        for (int i = tid; i < RG * a.mels; i += DX_NT) {
          const int r = i / a.mels, j = i - r * a.mels;
          float v[1];
          dx_poll<1>(X + xl.fb + r * DX_P2 + j, 0, tag, v, rt);
          tfb[r * DX_W + j] = v[0];
        }
      }
      __syncthreads();
      {
        float pa[1][RG], s[1][RL];
#pragma unroll
        for (int r = 0; r < RG; ++r) pa[0][r] = p1a[0][r];
        if (!G1_AHEAD) dxw_single<DXR_P1C, RG>(WP, st + DXS_CTX, lane, pa);
        dxw_single<DXR_P1O, RG, DX_W>(WP, tfb, lane, pa);
        dxs_reduce<1, RG>(pa, s, lane);
        if (epl && t + 1 < a.n) {
#pragma unroll
          for (int q = 0; q < RL; ++q) {
            const float p1v = fmaxf(s[0][q] + bl[DXB_P1 * DX_NW + wave], 0.f);
            dx_publish<WTC>(X + xl.p1 + erow[q] * DX_W + en, p1v, tag, rt);
            if (tval[q]) a.tape[(unsigned)DXT_P1 * tstr + trow[q] + (unsigned)(t + 1) * DX_W] = p1v;
          }
        }
      }
    } else {
      taco_f32x2 fp[RG];       // the frame projection's two columns as a pair
      float f2[1][RG], fa[3][RG], s[3][RL];
      dxq_zero<RG>(fp);
#pragma unroll
      for (int r = 0; r < RG; ++r) f2[0][r] = p1a[0][r];
      if (!G1_AHEAD) dxw_single<DXR_P1C, RG>(WP, st + DXS_CTX, lane, f2);
      // prenet layer 1 of the next step: from this step's own output (frame projection folded into the registers), or -- teacher
      // forcing -- from the teacher's frame (raw kernel rows in the same registers, zero beyond num_mels)
      if (TAPE) { dxw_pair<DXR_F, RG>(WP, st + DXS_OUT2, lane, fp); dxw_single<DXR_P1O, RG, DX_W>(WP, tfb, lane, f2); }
      else dxw_pair_single<DXR_F, DXR_P1O, RG>(WP, st + DXS_OUT2, lane, fp, f2);
#pragma unroll
      for (int r = 0; r < RG; ++r) { fa[0][r] = fp[r].x; fa[1][r] = fp[r].y; fa[2][r] = f2[0][r]; }
      dxs_reduce<3, RG>(fa, s, lane);
      if (epl) {
#pragma unroll
        for (int q = 0; q < RL; ++q) {
          if (t + 1 < a.n) {
            const float p1v = fmaxf(s[2][q] + bl[DXB_P1 * DX_NW + wave], 0.f);
            dx_publish<WTC>(X + xl.p1 + erow[q] * DX_W + en, p1v, tag, rt);
            if (TAPE && tval[q]) a.tape[(unsigned)DXT_P1 * tstr + trow[q] + (unsigned)(t + 1) * DX_W] = p1v;
          }
          store_frame(q, s[0][q] + bl[DXB_F0 * DX_NW + wave], s[1][q] + bl[DXB_F1 * DX_NW + wave], false);
        }
      }
    }
    if (a.dbg && member == 0) {   // per-step state dump for the stage-level parity test: [h_att | ctx | h1 | h2]
      for (int i = tid; i < RG * 4 * DX_W; i += DX_NT) {
        const int r = i / (4 * DX_W), q = (i / DX_W) & 3, nn = i % DX_W, b = row0 + r;
        const int off = q == 0 ? DXS_HATT : q == 1 ? DXS_CTX : q == 2 ? DXS_H1 : DXS_H2;
        if (b < a.B) a.dbg[((size_t)t * a.B + b) * a.dbgw + q * DX_W + nn] = st[r * DXS_LD + off + nn];
      }
    }
    if (t + 1 < a.n) dx_gather<RG, DX_W, false, DXS_LD, DX_NT, DX_DLY(10, DX_FIRST_POLL_DELAY)>(X + xl.p1, tag, st, DXS_T, 0, 0, tid, rt);
    __syncthreads();
    DX_STAMP(11);
  }
}

template <int RG, bool TAPE = false, bool MAN = false, int AW = DX_W, int PD = 2, bool TRACE = false>
__global__ __launch_bounds__(DX_NT) void k_decoder_xcd(const DxArgs a_in) {
  extern __shared__ __attribute__((aligned(16))) float dx_smem[];
  DxArgs a = a_in;
  // ---- census: which XCD am I on, is every XCD hosting exactly one group?  (its words: the last 64 floats of the LDS request) ----
  int* ictl = reinterpret_cast<int*>(dx_smem + dx_lds_floats(RG, a.T_in, TAPE, AW) - 64);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, a.force_wt, ictl, threadIdx.x, 8);
  const int group = __builtin_amdgcn_readfirstlane(ictl[0]);
  const int member = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
#ifdef DX_DLY_RT
  { const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) * 16; for (int i = 0; i < 12; ++i) rt.dly[i] = g_dx_dly[half + i]; }
#endif
  if (__builtin_amdgcn_readfirstlane((int)rt.wt)) dx_body<RG, TAPE, MAN, AW, PD, true, TRACE>(a, dx_smem, group, member, rt);
  else dx_body<RG, TAPE, MAN, AW, PD, false, TRACE>(a, dx_smem, group, member, rt);
}
