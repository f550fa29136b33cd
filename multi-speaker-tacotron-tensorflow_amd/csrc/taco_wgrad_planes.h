// taco_wgrad_planes.h -- weight gradients from PRE-SPLIT operands (round 6; VERDICT r05 next 6 i).
//
// k_wgrad_bf3 (taco_backward_kernels.h) converts every fp32 operand element into its three bf16 planes once per 64 x 64 tile that uses
// it, inside the product kernel: 47 % of its time on the vector ALU against 17 % on the matrix pipe (profiles/r05_v2_train_pmc_sq.txt).
// Here the conversion happens ONCE per operand element, in a pass of its own, and the product kernel converts nothing:
//
//   k_wp_split   fp32 operand [M, C] (row stride ld, optional row gather, optional per-copy time shift with the batch-row mask of a conv
//                tap) -> three bf16 planes in MFMA-FRAGMENT-MAJOR order: block (plane, column tile ct = c / 32, row chunk ms = m / 16)
//                is 1 KB = 64 lanes x 16 bytes, lane (i = c % 32, h = (m % 16) / 8) holding rows ms * 16 + 8 h .. + 7 of column c --
//                exactly the A / B operand of v_mfma_f32_32x32x16_bf16 when the contraction runs over rows.  Blocks of one (plane, ct)
//                are consecutive in ms: a product workgroup streams 1 KB blocks.
//   k_wp_gemm    dW[tap][k][n] = sum over m of X[m + tap - padl][k] dY[m][n]: a (64 WK) x (64 WN) (k x n) tile per workgroup of WK x WN waves
//                (64 x 64 per wave: 2 x 2 MFMA tiles, six products per tile pair as in k_wgrad_bf3 -- lo*hi, hi*lo, mid*mid, mid*hi,
//                hi*mid, hi*hi), over an M-slice.  Blocks go from global memory STRAIGHT INTO LDS (global_load_lds_dwordx4: the
//                block order in memory is the lane order of the load) through a ring of stages; a lane's fragment is one
//                conflict-free ds_read_b128.  No masks, no shifts, no conversions in the loop: the tap shift and its batch-row mask
//                were applied when the shifted copy of the (narrower) operand was written.
// The slice's tile leaves as a partial tile (deterministic mode: k_wgrad_reduce adds the slices in order) or through atomics.
#pragma once

struct WpSplitArgs {
  const float* src; const int* gather; uint4* out;
  int ld, M, T, C;          // T > 0: rows are [batch][T] and a shifted row must stay inside its batch row
  int Mp;                   // padded row count (multiple of 32): rows >= M are zero
  int ncopy, sigma0, dsigma;    // copy c holds rows shifted by sigma0 + c * dsigma: out_c[m] = src[m + sigma] (0 <= t + sigma < T) else 0
};
// bytes of one operand's plane set
__host__ __device__ inline size_t wp_plane_uint4(int C, int Mp, int ncopy) { return (size_t)ncopy * 3 * ((C + 31) / 32) * (Mp / 16) * 64; }

// grid (ceil(CT / 4), Mp / 64, ncopy), 256 threads: wave w owns column tile 4 bx + w, the workgroup four row chunks
__global__ __launch_bounds__(256) void k_wp_split(const WpSplitArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int CT = (a.C + 31) / 32, MS = a.Mp / 16;
  const int ct = blockIdx.x * 4 + wave;
  if (ct >= CT) return;
  const int i = lane & 31, h = lane >> 5, c = ct * 32 + i;
  const int copy = blockIdx.z, sigma = a.sigma0 + copy * a.dsigma;
  const bool cok = c < a.C;
#pragma unroll 1
  for (int q = 0; q < 4; ++q) {
    const int ms = blockIdx.y * 4 + q;
    const int mr = ms * 16 + 8 * h;
    float v[8];
    int tt = (a.T > 0) ? (mr % a.T) : 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = mr + e;
      bool ok = cok && m < a.M;
      if (a.T > 0) { const int ts = tt + sigma; ok = ok && ts >= 0 && ts < a.T; if (++tt == a.T) tt = 0; }
      size_t r = 0;
      if (ok) r = a.gather ? (size_t)a.gather[m] : (size_t)(m + sigma);
      v[e] = ok ? a.src[r * a.ld + c] : 0.f;
    }
    bf16x8 p0, p1, p2;
    wgb_split3(v, p0, p1, p2);
    uint4* o = a.out + (((size_t)copy * 3 * CT + ct) * MS + ms) * 64 + lane;
    const size_t ps = (size_t)CT * MS * 64;
    o[0] = __builtin_bit_cast(uint4, p0); o[ps] = __builtin_bit_cast(uint4, p1); o[2 * ps] = __builtin_bit_cast(uint4, p2);
  }
}

#define WP_MAXW 16
struct WpGemmArgs {
  const uint4* a; const uint4* b;       // plane sets of X (columns k) and dY (columns n)
  int K, N, Mp, kw, rpb;                // rpb: rows per M-slice (multiple of 16 * SM)
  int a_per_tap;                        // 1: copy `tap` of the X planes, copy 0 of dY; 0: copy 0 of X, copy `tap` of dY
  float* part; float* dw; int lddw;     // part != null: part[slice][tap][K][N]; else atomics into dw[tap][K][lddw]
  // bank mode (nw > 0): ALL widths 1 .. nw of a conv bank in one launch.  The X planes hold nw shifted copies (copy j: shift j + smin,
  // smin = -((nw - 1) / 2)), the dY planes all nw * N columns (width k: columns (k - 1) * N ...); blockIdx.z = slice * nw (nw + 1) / 2 +
  // (tap q of the bank, width-major); width k writes part + part_off[k - 1] as [slice][tap][K][N] or adds into dwk[k - 1][tap][K][lddw].
  int nw; float* dwk[WP_MAXW]; unsigned part_off[WP_MAXW];
};
typedef __attribute__((address_space(3))) unsigned char wp_lds_byte;

// WK x WN waves, each a 64 x 64 block of the (64 WK) x (64 WN) workgroup tile; SM = row chunks (of 16) per stage, NB = stages in the
// ring.  LDS per stage: 3 planes x 2 (WK + WN) column tiles x SM KB, as 1 KB blocks [operand][plane][column tile][chunk]; block j of a
// stage is fetched by wave j mod (WK WN).
template <int WK, int WN, int SM, int NB>
__global__ __launch_bounds__(64 * WK * WN) void k_wp_gemm(const WpGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wp_smem[];
  static_assert(NB >= 2 && NB <= 4, "the counted waits below know one or two later stages in flight");
  constexpr int NWV = WK * WN, TA = 2 * WK, TB = 2 * WN;
  constexpr int CHA = 3 * TA * SM, CH = 3 * (TA + TB) * SM;      // 1 KB blocks of a stage: X part, whole stage
  constexpr int LPW = (CH + NWV - 1) / NWV;                    // LDS-direct loads per wave and stage: LPW, or LPW - 1 for waves >= CH % NWV
  constexpr int NFULL = CH % NWV == 0 ? NWV : CH % NWV;        // waves that issue LPW loads
  constexpr unsigned STAGE = CH * 1024u;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wk = wave / WN, wn = wave - wk * WN;
  const int i = lane & 31, h = lane >> 5;
  const int nsplit = (g.Mp + g.rpb - 1) / g.rpb;
  int tap, sp, kwid = 0, CTB = (g.N + 31) / 32, ctb0 = blockIdx.y * TB, ctb_end;
  size_t acopy, bcopy;
  if (g.nw > 0) {                           // bank mode (wave-uniform: everything below comes from blockIdx and the arguments)
    const int ntap = g.nw * (g.nw + 1) / 2;
    sp = blockIdx.z / ntap;
    int q = blockIdx.z - sp * ntap;
    kwid = 1;
    while (q >= kwid) { q -= kwid; ++kwid; }
    tap = q;
    acopy = (size_t)(tap - (kwid - 1) / 2 + (g.nw - 1) / 2); bcopy = 0;
    ctb0 += (kwid - 1) * CTB; ctb_end = kwid * CTB; CTB *= g.nw;
  } else {
    tap = blockIdx.z / nsplit; sp = blockIdx.z - tap * nsplit;
    acopy = g.a_per_tap ? (size_t)tap : 0; bcopy = g.a_per_tap ? 0 : (size_t)tap;
    ctb_end = CTB;
  }
  const int CTA = (g.K + 31) / 32, MS = g.Mp / 16;
  const int cta0 = blockIdx.x * TA;
  const int ms0 = sp * (g.rpb / 16), ms1 = min(MS, ms0 + g.rpb / 16);
  const int nst = (ms1 - ms0 + SM - 1) / SM;
  const uint4* abase = g.a + acopy * 3 * CTA * MS * 64 + lane;
  const uint4* bbase = g.b + bcopy * 3 * CTB * MS * 64 + lane;
  const unsigned lds0 = (unsigned)(size_t)(wp_lds_byte*)wp_smem;
  const bool fullw = wave < NFULL;          // (wave-uniform)

  auto issue = [&](int s, int buf) {        // this wave's blocks of stage s into ring slot buf
#pragma unroll
    for (int u = 0; u < LPW; ++u) {
      const int j = wave + u * NWV;                       // block of the stage (wave-uniform)
      if (u == LPW - 1 && !fullw) break;
      const int msq0 = ms0 + s * SM;
      const uint4* src;
      if (j < CHA) {
        const int p = j / (TA * SM), r2 = j - p * TA * SM, ctl = r2 / SM, q = r2 - ctl * SM;
        src = abase + ((size_t)(p * CTA + min(cta0 + ctl, CTA - 1)) * MS + min(msq0 + q, MS - 1)) * 64;      // (a chunk past the slice's end is re-read and never used)
      } else {
        const int r = j - CHA, p = r / (TB * SM), r2 = r - p * TB * SM, ctl = r2 / SM, q = r2 - ctl * SM;
        src = bbase + ((size_t)(p * CTB + min(ctb0 + ctl, ctb_end - 1)) * MS + min(msq0 + q, MS - 1)) * 64;
      }
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf * STAGE + (unsigned)j * 1024u);
      gx_load_lds16(reinterpret_cast<const float*>(src), dst);
    }
  };
  wg_f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#pragma unroll
  for (int s = 0; s < NB - 1; ++s)
    if (s < nst) issue(s, s);
  for (int s = 0; s < nst; ++s) {
    // stage s has landed for this wave when at most the loads of the later stages in flight remain
    const int later = min(nst - 1 - s, NB - 2);
    if (later >= 2) { if (fullw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (LPW - 1)) : "memory"); }
    else if (later == 1) { if (fullw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW - 1) : "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // every wave's blocks of stage s are in LDS; every wave is done with stage s - 1
    if (s + NB - 1 < nst) issue(s + NB - 1, (s + NB - 1) % NB);
    const unsigned char* st = wp_smem + (size_t)(s % NB) * STAGE;
#pragma unroll
    for (int q = 0; q < SM; ++q) {
      if (ms0 + s * SM + q >= ms1) break;
      bf16x8 af[2][3], bf[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          af[t][p] = *reinterpret_cast<const bf16x8*>(st + ((size_t)(p * TA + 2 * wk + t) * SM + q) * 1024 + lane * 16);
          bf[t][p] = *reinterpret_cast<const bf16x8*>(st + (size_t)CHA * 1024 + ((size_t)(p * TB + 2 * wn + t) * SM + q) * 1024 + lane * 16);
        }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[b][0], acc[a][b], 0, 0, 0);      // small terms first
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][2], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][1], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], acc[a][b], 0, 0, 0);
        }
    }
  }
  const int kb = blockIdx.x * 64 * WK + wk * 64, nb = blockIdx.y * 64 * WN + wn * 64;
  float* out;
  if (g.nw > 0) out = g.part ? g.part + g.part_off[kwid - 1] + ((size_t)sp * kwid + tap) * g.K * g.N : g.dwk[kwid - 1] + (size_t)tap * g.K * g.lddw;
  else out = g.part ? g.part + ((size_t)sp * g.kw + tap) * g.K * g.N : g.dw + (size_t)tap * g.K * g.lddw;
  const int ldo = g.part ? g.N : g.lddw;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = nb + 32 * b + i;
      if (n >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kr = kb + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (kr < g.K) {
          float* p = out + (size_t)kr * ldo + n;
          if (g.part) *p = acc[a][b][r]; else atomicAdd(p, acc[a][b][r]);
        }
      }
    }
}
