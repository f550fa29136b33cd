// ubench_mfma_stage.hip -- VERDICT r05, next 1: does the exact-fp32 multi-block MFMA (v_mfma_f32_4x4x1_16b_f32: 16 blocks of 4 x 4 x 1)
// and / or ONE wave per SIMD take the reduction tree out of a stage of the persistent decoder?  The stage of k_decoder_xcd in isolation,
// on the kernel's own census / publish / gather primitives (csrc/taco_decoder_xcd.h), as tools/ubench_rowsets.hip does: 256 workgroups,
// one group of 32 members per XCD, 4 batch rows per group, 10 dependent stages per step; a member owns 8 units; a stage multiplies the
// rows' gathered 256-wide vector with NCOL weight columns per unit (NCOL = 4: a gates stage such as GRU 1's context rows -- r, u,
// candidate-x, o0; NCOL = 1: a candidate stage), reduces, runs the epilogue, publishes one value per (unit, row), gathers, barrier.
//   MODE 0  vector ALU as the decoder today (v_pk_fma_f32 over column pairs, dxs_reduce: 10 cross-lane instructions per column)
//   MODE 1  MFMA 4x4x1 x 16 blocks: block = K-slice of 16 inputs, M = the 4 batch rows (A operand: the lane's row value from LDS),
//           N = the unit's 4 columns (B operand: the resident weight); the K sum inside a block accumulates in the pipe, the 16 blocks
//           meet by three permlane swaps (which deal the ROWS to the four 16-lane rows) and two row_ror adds: 8 instructions per unit
//   NT 512  eight waves per member, wave = unit (two waves per SIMD, as today);   NT 256  four waves, wave = two units (one per SIMD)
// Every variant writes the reduced sums of its first stage to a buffer that the host checks against a double-precision product, so the
// lane maps of the MFMA operands are verified, not assumed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I multi-speaker-tacotron-tensorflow_amd/csrc tools/ubench_mfma_stage.hip -o tools/ubench_mfma_stage
#include "taco_decoder_xcd.h"
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define UB_NST 10          // dependent stages per step
#define UB_LD 516          // LDS floats per row: two alternating 256-wide vectors + 4 (rows land 4 banks apart: the MFMA variant's A reads are conflict free)
#define UB_RG 4

typedef float ub_f32x4 __attribute__((ext_vector_type(4)));
struct UbArgs { const float* wc; unsigned long long* xbuf; unsigned* ctl; unsigned* err; long long* clk; float* sink; float* dbg; int steps; };
// canonical weights: wc[((member * 8 + unit) * 4 + col) * 256 + k]

template <int NT, int MODE, int NCOL, int CHN = 1>
__global__ __launch_bounds__(NT) void k_stage(const UbArgs a) {
  constexpr int NW = NT / 64, UPW = 8 / NW, RG = UB_RG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* st = smem;
  int* ictl = reinterpret_cast<int*>(st + RG * UB_LD);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, 0, ictl, tid, 8);
  const int group = __builtin_amdgcn_readfirstlane(ictl[0]), member = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  if (member >= DX_GROUP) return;
  // ---- resident weights ----
  constexpr int NV = NCOL == 4 ? 8 : 2;
  taco_f32x2 WV[UPW][NV];      // MODE 0
  float WM[UPW][16];           // MODE 1
#pragma unroll
  for (int u = 0; u < UPW; ++u) {
    const float* w = a.wc + ((size_t)(member * 8 + wave * UPW + u) * 4) * 256;
    if (MODE == 0 || MODE == 3) {
      if (NCOL == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          WV[u][e] = (taco_f32x2){w[0 * 256 + 4 * lane + e], w[1 * 256 + 4 * lane + e]};
          WV[u][4 + e] = (taco_f32x2){w[2 * 256 + 4 * lane + e], w[3 * 256 + 4 * lane + e]};
        }
      } else {
        WV[u][0] = (taco_f32x2){w[4 * lane], w[4 * lane + 1]};
        WV[u][1] = (taco_f32x2){w[4 * lane + 2], w[4 * lane + 3]};
      }
    } else {
      const int b = lane >> 2, j = NCOL == 4 ? (lane & 3) : 0;
#pragma unroll
      for (int s = 0; s < 16; ++s) WM[u][s] = w[j * 256 + 16 * b + s];
    }
  }
  for (int i = tid; i < RG * UB_LD; i += NT) st[i] = 0.01f * (float)(i % 97);
  __syncthreads();
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)group * UB_NST * RG * DX_W;
  const bool tracer = group == 0 && member == 0 && tid == 0;
  long long t0 = 0, ph[3] = {0, 0, 0};
  for (int step = 0; step < a.steps; ++step) {
    if (tracer && step == 8) t0 = (long long)__builtin_readcyclecounter();
    const unsigned tag = (unsigned)step + 1u;
#pragma unroll 1
    for (int sgi = 0; sgi < UB_NST; ++sgi) {
      const int rd = (sgi & 1) * 256, wr = 256 - rd;
      dx_gu64* Xs = X + (size_t)sgi * RG * DX_W;
      const float* vec = st + rd;
      const bool dbg = a.dbg && group == 0 && step == 0 && sgi == 0;
      const long long c0 = tracer ? (long long)__builtin_readcyclecounter() : 0;
      if constexpr (MODE == 3) {
        // ---- no LDS, no barrier: every lane polls ITS K-slice (inputs 4 lane .. 4 lane + 3 of the four rows: two 16-byte requests per row) of the
        // vector the previous stage published straight into registers, multiplies, reduces, publishes.  Eight times the poll requests of the gather.
        const int prev = (sgi + UB_NST - 1) % UB_NST;
        const dx_gu64* Xp = X + (size_t)prev * RG * DX_W + 4 * lane;
        const unsigned ptag = sgi == 0 ? tag - 1u : tag;
        float xr[RG][4];
        if (step == 0 && sgi == 0) {
#pragma unroll
          for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[r][e] = 0.01f * (float)((r * UB_LD + 4 * lane + e) % 97);
        } else {
          dx_u64x2 g[RG][2];
          unsigned spins = 0;
          for (;;) {
            bool ok = true;
#pragma unroll
            for (int r = 0; r < RG; ++r)
#pragma unroll
              for (int h = 0; h < 2; ++h) { const dx_gu64* p = Xp + r * DX_W + 2 * h; asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(g[r][h]) : "v"(p) : "memory"); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < RG; ++r)
#pragma unroll
              for (int h = 0; h < 2; ++h) ok = ok && ((unsigned)(g[r][h][0] >> 32) == ptag) && ((unsigned)(g[r][h][1] >> 32) == ptag);
            if (ok || rt.dead) break;
            if ((++spins & 1023u) == 0 && spins >= DX_SPIN_LIMIT) { __hip_atomic_store(rt.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); rt.dead = true; }
          }
#pragma unroll
          for (int r = 0; r < RG; ++r) { xr[r][0] = __uint_as_float((unsigned)g[r][0][0]); xr[r][1] = __uint_as_float((unsigned)g[r][0][1]); xr[r][2] = __uint_as_float((unsigned)g[r][1][0]); xr[r][3] = __uint_as_float((unsigned)g[r][1][1]); }
        }
        static_assert(MODE != 3 || (UPW == 1), "wave = unit");
        float s[NCOL][1];
        if constexpr (NCOL == 4) {
          taco_f32x2 p0[RG], p1[RG];
          dxq_zero<RG>(p0); dxq_zero<RG>(p1);
#pragma unroll
          for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) { DXQ_FMA(p0[r], xr[r][e], WV[0][e]); DXQ_FMA(p1[r], xr[r][e], WV[0][4 + e]); }
          float g4[4][RG];
#pragma unroll
          for (int r = 0; r < RG; ++r) { g4[0][r] = p0[r].x; g4[1][r] = p0[r].y; g4[2][r] = p1[r].x; g4[3][r] = p1[r].y; }
          dxs_reduce<4, RG>(g4, s, lane);
        } else {
          float g1[1][RG];
#pragma unroll
          for (int r = 0; r < RG; ++r) g1[0][r] = fmaf(WV[0][1].y, xr[r][3], fmaf(WV[0][1].x, xr[r][2], fmaf(WV[0][0].y, xr[r][1], WV[0][0].x * xr[r][0])));
          dxs_reduce<1, RG>(g1, s, lane);
        }
        float v;
        if constexpr (NCOL == 4) v = dx_sigmoid_fast(s[0][0]) * taco_tanh_fast(s[2][0]) + 0.01f * dx_sigmoid_fast(s[1][0]) * s[3][0];
        else v = taco_tanh_fast(s[0][0]);
        if ((lane & 15) == 0) {
          dx_publish<0>(Xs + (lane >> 4) * DX_W + member * 8 + wave, v, tag, rt);
          if (dbg) for (int c = 0; c < NCOL; ++c) a.dbg[((member * 8 + wave) * 4 + c) * 4 + (lane >> 4)] = s[c][0];
        }
      } else if constexpr (MODE == 2) {      // no compute at all: the exchange alone
        if ((lane & 15) == 0) dx_publish<0>(Xs + (lane >> 4) * DX_W + member * 8 + wave, vec[lane], tag, rt);
      } else if constexpr (MODE == 0) {
        // ---- the decoder's passes and reduction as they are (dxw_quad / dxw_single, dxs_reduce) ----
        if constexpr (NCOL == 4) {
          taco_f32x2 p0[UPW][RG], p1[UPW][RG];
#pragma unroll
          for (int u = 0; u < UPW; ++u) { dxq_zero<RG>(p0[u]); dxq_zero<RG>(p1[u]); }
#pragma unroll
          for (int r = 0; r < RG; ++r) {
            const float4 xv = *reinterpret_cast<const float4*>(vec + r * UB_LD + 4 * lane);
#pragma unroll
            for (int u = 0; u < UPW; ++u) {
              DXQ_FMA(p0[u][r], xv.x, WV[u][0]); DXQ_FMA(p1[u][r], xv.x, WV[u][4]);
              DXQ_FMA(p0[u][r], xv.y, WV[u][1]); DXQ_FMA(p1[u][r], xv.y, WV[u][5]);
              DXQ_FMA(p0[u][r], xv.z, WV[u][2]); DXQ_FMA(p1[u][r], xv.z, WV[u][6]);
              DXQ_FMA(p0[u][r], xv.w, WV[u][3]); DXQ_FMA(p1[u][r], xv.w, WV[u][7]);
            }
          }
#pragma unroll
          for (int u = 0; u < UPW; ++u) {
            float g[4][RG], s[4][1];
#pragma unroll
            for (int r = 0; r < RG; ++r) { g[0][r] = p0[u][r].x; g[1][r] = p0[u][r].y; g[2][r] = p1[u][r].x; g[3][r] = p1[u][r].y; }
            dxs_reduce<4, RG>(g, s, lane);
            const float v = dx_sigmoid_fast(s[0][0]) * taco_tanh_fast(s[2][0]) + 0.01f * dx_sigmoid_fast(s[1][0]) * s[3][0];
            if ((lane & 15) == 0) {
              dx_publish<0>(Xs + (lane >> 4) * DX_W + member * 8 + wave * UPW + u, v, tag, rt);
              if (dbg) for (int c = 0; c < 4; ++c) a.dbg[((member * 8 + wave * UPW + u) * 4 + c) * 4 + (lane >> 4)] = s[c][0];
            }
          }
        } else {
          float acc[UPW][RG];
#pragma unroll
          for (int u = 0; u < UPW; ++u)
#pragma unroll
            for (int r = 0; r < RG; ++r) acc[u][r] = 0.f;
#pragma unroll
          for (int r = 0; r < RG; ++r) {
            const float4 xv = *reinterpret_cast<const float4*>(vec + r * UB_LD + 4 * lane);
#pragma unroll
            for (int u = 0; u < UPW; ++u) {
              acc[u][r] = fmaf(WV[u][0].x, xv.x, acc[u][r]); acc[u][r] = fmaf(WV[u][0].y, xv.y, acc[u][r]);
              acc[u][r] = fmaf(WV[u][1].x, xv.z, acc[u][r]); acc[u][r] = fmaf(WV[u][1].y, xv.w, acc[u][r]);
            }
          }
#pragma unroll
          for (int u = 0; u < UPW; ++u) {
            float g[1][RG], s[1][1];
#pragma unroll
            for (int r = 0; r < RG; ++r) g[0][r] = acc[u][r];
            dxs_reduce<1, RG>(g, s, lane);
            const float v = taco_tanh_fast(s[0][0]);
            if ((lane & 15) == 0) {
              dx_publish<0>(Xs + (lane >> 4) * DX_W + member * 8 + wave * UPW + u, v, tag, rt);
              if (dbg) a.dbg[((member * 8 + wave * UPW + u) * 4 + 0) * 4 + (lane >> 4)] = s[0][0];
            }
          }
        }
      } else {
        // ---- MFMA: A = the lane's row value (M = row lane & 3 of block lane >> 2), B = the resident weight (N = column lane & 3) ----
        const float* xr = vec + (lane & 3) * UB_LD + 16 * (lane >> 2);
        float xs[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(xr + 4 * q);
          xs[4 * q] = t.x; xs[4 * q + 1] = t.y; xs[4 * q + 2] = t.z; xs[4 * q + 3] = t.w;
        }
        ub_f32x4 acc[UPW], accc[UPW][CHN];      // CHN independent accumulate chains per unit (a dependent 4x4x1 costs more than its issue slot)
#pragma unroll
        for (int u = 0; u < UPW; ++u)
#pragma unroll
          for (int c = 0; c < CHN; ++c) accc[u][c] = (ub_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
          for (int u = 0; u < UPW; ++u) accc[u][s % CHN] = __builtin_amdgcn_mfma_f32_4x4x1f32(xs[s], WM[u][s], accc[u][s % CHN], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
          if constexpr (CHN == 4) acc[u] = (accc[u][0] + accc[u][1]) + (accc[u][2] + accc[u][3]);
          else if constexpr (CHN == 2) acc[u] = accc[u][0] + accc[u][1];
          else acc[u] = accc[u][0];
        }
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
          // D register i = batch row i, lane = (block, column).  Swaps deal the rows to the 16-lane rows; row_ror finishes the four blocks left
          auto r02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[u][0]), __float_as_uint(acc[u][2]), false, false);
          auto r13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[u][1]), __float_as_uint(acc[u][3]), false, false);
          const float e0 = __uint_as_float(r02[0]) + __uint_as_float(r02[1]);      // lanes 0-31: row 0, lanes 32-63: row 2
          const float e1 = __uint_as_float(r13[0]) + __uint_as_float(r13[1]);      // row 1 | row 3
          auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(e0), __float_as_uint(e1), false, false);
          float f = __uint_as_float(q[0]) + __uint_as_float(q[1]);                 // 16-lane row p: batch row p
          f += DX_DPP0(f, 0x124);                                                  // row_ror:4
          f += DX_DPP0(f, 0x128);                                                  // row_ror:8  -> every lane of the row: (row lane >> 4, column lane & 3)
          float v;
          if constexpr (NCOL == 4) {
            const float g = dx_sigmoid_fast(f), th = taco_tanh_fast(f);
            v = DX_DPP0(g, 0x00) * DX_DPP0(th, 0xAA) + 0.01f * DX_DPP0(g, 0x55) * DX_DPP0(f, 0xFF);      // quad broadcasts of columns 0, 2, 1, 3
          } else v = taco_tanh_fast(f);
          if ((lane & 15) == 0) dx_publish<0>(Xs + (lane >> 4) * DX_W + member * 8 + wave * UPW + u, v, tag, rt);
          if (dbg && (lane & 12) == 0 && (NCOL == 4 || (lane & 3) == 0)) a.dbg[((member * 8 + wave * UPW + u) * 4 + (lane & 3)) * 4 + (lane >> 4)] = f;
        }
      }
      const long long c1 = tracer ? (long long)__builtin_readcyclecounter() : 0;
      if constexpr (MODE != 3) dx_gather<RG, DX_W, false, UB_LD, NT>(Xs, tag, st, wr, 0, 0, tid, rt);
      const long long c2 = tracer ? (long long)__builtin_readcyclecounter() : 0;
      if constexpr (MODE != 3) __syncthreads();
      if (tracer && step >= 8) { ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += (long long)__builtin_readcyclecounter() - c2; }
    }
  }
  if (tracer) { a.clk[0] = (long long)__builtin_readcyclecounter() - t0; a.clk[1] = ph[0]; a.clk[2] = ph[1]; a.clk[3] = ph[2]; }
  if (tid == 0) a.sink[blockIdx.x] = st[7];
}

template <int NT, int MODE, int NCOL, int CHN = 1>
static int run(const char* name, int steps) {
  UbArgs a;
  std::vector<float> hw((size_t)DX_GROUP * 8 * 4 * 256);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.02f * (float)((int)((i * 2654435761u) >> 20) % 101 - 50) / 50.f;
  float* dw; CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  const size_t xg = (size_t)DX_NGROUP * UB_NST * UB_RG * DX_W;
  unsigned long long* xb; CK(hipMalloc(&xb, xg * 8));
  unsigned *ctl, *err; CK(hipMalloc(&ctl, 256)); CK(hipMalloc(&err, 256));
  long long* clk; CK(hipMalloc(&clk, 64)); float* sink; CK(hipMalloc(&sink, 256 * 4));
  const size_t ndbg = (size_t)DX_GROUP * 8 * 4 * 4;
  float* dbg; CK(hipMalloc(&dbg, ndbg * 4));
  a.wc = dw; a.xbuf = xb; a.ctl = ctl; a.err = err; a.clk = clk; a.sink = sink; a.dbg = dbg; a.steps = steps;
  const size_t lds = 96 * 1024;      // one workgroup per CU
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stage<NT, MODE, NCOL, CHN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f; long long hph[4] = {0, 0, 0, 0}; unsigned herr[64];
  double maxerr = 0.0;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(xb, 0, xg * 8));
    CK(hipMemset(ctl, 0, 256)); CK(hipMemset(err, 0, 256)); CK(hipMemset(clk, 0, 64)); CK(hipMemset(dbg, 0xFF, ndbg * 4));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_stage<NT, MODE, NCOL, CHN>), dim3(DX_NGROUP * DX_GROUP), dim3(NT), lds, 0, a);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(herr, err, 256, hipMemcpyDeviceToHost));
    if (herr[0]) { printf("%s: device error word %u\n", name, herr[0]); return 1; }
    if (ms < best) { best = ms; CK(hipMemcpy(hph, clk, 32, hipMemcpyDeviceToHost)); }
    if (rep == 0) {      // the first stage's reduced sums against a double-precision product
      std::vector<float> hd(ndbg); CK(hipMemcpy(hd.data(), dbg, ndbg * 4, hipMemcpyDeviceToHost));
      for (int mu = 0; mu < DX_GROUP * 8; ++mu)
        for (int c = 0; c < NCOL; ++c)
          for (int r = 0; r < UB_RG; ++r) {
            double ref = 0.0;
            for (int k = 0; k < 256; ++k) ref += (double)hw[((size_t)mu * 4 + c) * 256 + k] * (double)(0.01f * (float)((r * UB_LD + k) % 97));
            const double d = std::fabs(ref - (double)hd[((size_t)mu * 4 + c) * 4 + r]);
            if (!(d <= maxerr)) maxerr = d;      // (a NaN sticks)
          }
    }
  }
  const double ns = (double)(steps - 8) * UB_NST;
  printf("%-58s %6.2f us per step  %6.0f clocks per stage | compute + publish %5.0f | poll + LDS write %5.0f | barrier %5.0f | first-stage sums vs float64: %.2e   protocol %u\n",
         name, best * 1e3 / steps, (double)hph[0] / ns, hph[1] / ns, hph[2] / ns, hph[3] / ns, maxerr, herr[8]);
  hipFree(dw); hipFree(xb); hipFree(ctl); hipFree(err); hipFree(clk); hipFree(sink); hipFree(dbg);
  return (MODE == 2 || maxerr < 1e-4) ? 0 : 1;
}

int main() {
  const int steps = 136;
  printf("a stage of k_decoder_xcd in isolation: 4 rows per group, 8 units per member, 10 exchanges per step (tracer: wave 0 of member 0)\n");
  int bad = 0;
  bad |= run<512, 0, 4>("gates stage (4 columns per unit), vector ALU, 8 waves", steps);
  bad |= run<512, 1, 4>("gates stage, MFMA 4x4x1 x 16, 8 waves (wave = unit)", steps);
  bad |= run<256, 0, 4>("gates stage, vector ALU, 4 waves (wave = two units)", steps);
  bad |= run<256, 1, 4>("gates stage, MFMA 4x4x1 x 16, 4 waves (wave = two units)", steps);
  bad |= run<512, 1, 4, 2>("gates stage, MFMA, 8 waves, two accumulate chains", steps);
  bad |= run<512, 1, 4, 4>("gates stage, MFMA, 8 waves, four accumulate chains", steps);
  bad |= run<256, 1, 4, 2>("gates stage, MFMA, 4 waves, two chains per unit", steps);
  bad |= run<512, 3, 4>("gates stage, vector ALU, no LDS / no barrier: lanes poll their K-slice", steps);
  bad |= run<512, 3, 1>("candidate stage, vector ALU, no LDS / no barrier", steps);
  bad |= run<512, 2, 1>("no compute: publish, gather, barrier (8 waves)", steps);
  bad |= run<512, 0, 1>("candidate stage (1 column per unit), vector ALU, 8 waves", steps);
  bad |= run<512, 1, 1>("candidate stage, MFMA, 8 waves", steps);
  bad |= run<512, 1, 1, 4>("candidate stage, MFMA, 8 waves, four chains", steps);
  bad |= run<256, 0, 1>("candidate stage, vector ALU, 4 waves", steps);
  bad |= run<256, 1, 1>("candidate stage, MFMA, 4 waves", steps);
  return bad;
}
