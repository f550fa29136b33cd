// ubench_rowsets.hip -- does splitting a group's rows into two software-pipelined sets hide the exchange hops of the persistent decoder?
// ("two independent row sets ... in k_decoder_xcd".)  The stage of k_decoder_xcd in isolation, built from the
// kernel's OWN primitives (csrc/taco_decoder_xcd.h: census, dx_pass, dx_reduce, dx_publish, dx_gather): 256 workgroups, one group of 32
// members per XCD, 10 dependent stages per step, each stage = pass of 2 weight columns over the rows' 256-wide LDS vector -> DPP
// reduction -> epilogue (sigmoid x tanh) -> publish 8-byte {value, tag} granules -> all-gather into LDS -> barrier.
//   variant A  one set of RG rows          : compute, publish, gather (the hop is exposed), barrier          -- the decoder today
//   variant B  two sets of RG/2 rows each  : compute A, publish A | gather B (in flight since the previous half), barrier |
//              compute B, publish B | gather A, barrier                                                      -- the proposal
// Same weights in registers for both sets (shared, as proposed), per-set LDS vectors, exchange regions and tags.
// Prints clocks per stage (shader clock counter of group 0 / member 0) and microseconds per step (HIP events).
//   hipcc --offload-arch=gfx950 -O3 -I multi-speaker-tacotron-tensorflow_amd/csrc tools/ubench_rowsets.hip -o tools/ubench_rowsets
#include "taco_decoder_xcd.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define UB_NST 10          // dependent stages per step (the decoder has 10 exchanges per step)
#define UB_LD 512          // LDS floats per row: two alternating 256-wide vectors

struct UbArgs { const float* wpack; unsigned long long* xbuf; unsigned* ctl; unsigned* err; long long* clk; float* sink; int steps; };

template <int RG>
__device__ __forceinline__ void ub_compute(const float (&W)[16], const float* vec, int lane, int wave, int member, dx_gu64* X, unsigned tag, DxRt& rt) {
  constexpr int RL = DxRL<RG>::value;
  float acc[2][RG], s[2][RL];
  dx_zero<2, RG>(acc);
  dx_pass<0, 2, RG, 16, UB_LD>(W, vec, lane, acc);
  dx_reduce<2, RG>(acc, s, lane);
  const bool epl = lane < (RG >= 4 ? 4 : RG);
#pragma unroll
  for (int q = 0; q < RL; ++q) {
    const float v = dx_sigmoid_fast(s[0][q]) * taco_tanh_fast(s[1][q]);
    if (epl) dx_publish(X + dx_row<RG>(lane & 3, q) * DX_W + member * 8 + wave, v, tag, rt);
  }
}

// ---- variant S: 4-byte granules, no tag: "not written yet" is a sentinel bit pattern (a NaN payload no stage produces); two parity
// copies of every exchange region, the producer of step t re-arms its slot of the other parity while it publishes (every consumer of
// step t - 1's value has read it by then: nobody reaches stage s of step t before everyone has gathered stage s of step t - 1).
// Half the bytes per value, and a thread polls its RG/2 values with ONE 8-byte load per pair instead of one per value.
#define UB_SENT 0xFFFFDEADu
typedef __attribute__((address_space(1))) unsigned ub_gu32;
template <int RG>
__device__ __forceinline__ void ub_compute_s(const float (&W)[16], const float* vec, int lane, int wave, int member, ub_gu32* Xcur, ub_gu32* Xoth, DxRt& rt) {
  constexpr int RL = DxRL<RG>::value;
  float acc[2][RG], s[2][RL];
  dx_zero<2, RG>(acc);
  dx_pass<0, 2, RG, 16, UB_LD>(W, vec, lane, acc);
  dx_reduce<2, RG>(acc, s, lane);
  const bool epl = lane < (RG >= 4 ? 4 : RG);
#pragma unroll
  for (int q = 0; q < RL; ++q) {
    const float v = dx_sigmoid_fast(s[0][q]) * taco_tanh_fast(s[1][q]);
    if (epl) {
      const int o = dx_row<RG>(lane & 3, q) * DX_W + member * 8 + wave;
      if (rt.wt) { __hip_atomic_store(Xcur + o, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(Xoth + o, UB_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      else { __hip_atomic_store(Xcur + o, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_store(Xoth + o, UB_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
  }
}
template <int RG>
__device__ __forceinline__ void ub_gather_s(const ub_gu32* X, float* st, int off, int tid, DxRt& rt) {
  constexpr int NV = RG * DX_W;                 // values per region
  if (NV >= 2 * DX_NT) {                        // pairs: one 8-byte load per two values
    constexpr int NP = NV >= 2 * DX_NT ? NV / (2 * DX_NT) : 1;
    const dx_gu64* P = (const dx_gu64*)X;
    unsigned long long g[NP];
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int u = 0; u < NP; ++u) g[u] = __hip_atomic_load(P + tid + u * DX_NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < NP; ++u) ok = ok && ((unsigned)g[u] != UB_SENT) && ((unsigned)(g[u] >> 32) != UB_SENT);
      if (ok || rt.dead) break;
      if ((++spins & 1023u) == 0 && spins >= DX_SPIN_LIMIT) { __hip_atomic_store(rt.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); rt.dead = true; }
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int i = 2 * (tid + u * DX_NT);
      *reinterpret_cast<float2*>(st + (i / DX_W) * UB_LD + off + (i % DX_W)) = make_float2(__uint_as_float((unsigned)g[u]), __uint_as_float((unsigned)(g[u] >> 32)));
    }
  } else {
    if (tid < NV) {
      unsigned g; unsigned spins = 0;
      for (;;) {
        g = __hip_atomic_load(X + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g != UB_SENT || rt.dead) break;
        if ((++spins & 1023u) == 0 && spins >= DX_SPIN_LIMIT) { __hip_atomic_store(rt.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); rt.dead = true; }
      }
      st[(tid / DX_W) * UB_LD + off + (tid % DX_W)] = __uint_as_float(g);
    }
  }
}
template <int RG>
__global__ __launch_bounds__(DX_NT) void k_rowsets_s(const UbArgs a) {
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
  float W[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) W[j] = a.wpack[((size_t)member * 16 + j) * DX_NT + tid];
  for (int i = tid; i < RG * UB_LD; i += DX_NT) st[i] = 0.01f * (float)(i % 97);
  __syncthreads();
  ub_gu32* X = (ub_gu32*)a.xbuf + (size_t)group * 2 * UB_NST * RG * DX_W;       // [parity][stage][RG * 256]
  const bool tracer = group == 0 && member == 0 && tid == 0;
  long long t0 = 0, ph[3] = {0, 0, 0};
  for (int step = 0; step < a.steps; ++step) {
    if (tracer && step == 8) t0 = (long long)__builtin_readcyclecounter();
    const int par = step & 1;
#pragma unroll 1
    for (int sgi = 0; sgi < UB_NST; ++sgi) {
      const int rd = (sgi & 1) * 256, wr = 256 - rd;
      ub_gu32* Xc = X + ((size_t)par * UB_NST + sgi) * RG * DX_W;
      ub_gu32* Xo = X + ((size_t)(par ^ 1) * UB_NST + sgi) * RG * DX_W;
      const long long c0 = tracer ? (long long)__builtin_readcyclecounter() : 0;
      ub_compute_s<RG>(W, st + rd, lane, wave, member, Xc, Xo, rt);
      const long long c1 = tracer ? (long long)__builtin_readcyclecounter() : 0;
      ub_gather_s<RG>(Xc, st, wr, tid, rt);
      const long long c2 = tracer ? (long long)__builtin_readcyclecounter() : 0;
      __syncthreads();
      if (tracer && step >= 8) { ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += (long long)__builtin_readcyclecounter() - c2; }
    }
  }
  if (tracer) { a.clk[0] = (long long)__builtin_readcyclecounter() - t0; a.clk[1] = ph[0]; a.clk[2] = ph[1]; a.clk[3] = ph[2]; }
  if (tid == 0) a.sink[blockIdx.x] = st[(tid * 7) % (RG * UB_LD)];
}

// ---- variant T: the {value, tag} granules as they are, but a thread polls two ADJACENT granules with one 16-byte load (each 8-byte half
// is written by a single 8-byte store; a half that is not there yet fails its tag test and the pair is requested again)
typedef unsigned long long ub_u64x2 __attribute__((ext_vector_type(2)));
template <int RG>
__device__ __forceinline__ void ub_gather_t(const dx_gu64* X, unsigned tag, float* st, int off, int tid, DxRt& rt) {
  constexpr int NV = RG * DX_W;
  static_assert(NV >= 2 * DX_NT, "pairs");
  constexpr int NP = NV / (2 * DX_NT);
  ub_u64x2 g[NP];
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const dx_gu64* p = X + 2 * (tid + u * DX_NT);
      asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(g[u]) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < NP; ++u) ok = ok && ((unsigned)(g[u][0] >> 32) == tag) && ((unsigned)(g[u][1] >> 32) == tag);
    if (ok || rt.dead) break;
    if ((++spins & 1023u) == 0 && spins >= DX_SPIN_LIMIT) { __hip_atomic_store(rt.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); rt.dead = true; }
  }
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int i = 2 * (tid + u * DX_NT);
    *reinterpret_cast<float2*>(st + (i / DX_W) * UB_LD + off + (i % DX_W)) = make_float2(__uint_as_float((unsigned)g[u][0]), __uint_as_float((unsigned)g[u][1]));
  }
}
template <int RG>
__global__ __launch_bounds__(DX_NT) void k_rowsets_t(const UbArgs a) {
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
  float W[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) W[j] = a.wpack[((size_t)member * 16 + j) * DX_NT + tid];
  for (int i = tid; i < RG * UB_LD; i += DX_NT) st[i] = 0.01f * (float)(i % 97);
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
      const long long c0 = tracer ? (long long)__builtin_readcyclecounter() : 0;
      ub_compute<RG>(W, st + rd, lane, wave, member, Xs, tag, rt);
      const long long c1 = tracer ? (long long)__builtin_readcyclecounter() : 0;
      ub_gather_t<RG>(Xs, tag, st, wr, tid, rt);
      const long long c2 = tracer ? (long long)__builtin_readcyclecounter() : 0;
      __syncthreads();
      if (tracer && step >= 8) { ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += (long long)__builtin_readcyclecounter() - c2; }
    }
  }
  if (tracer) { a.clk[0] = (long long)__builtin_readcyclecounter() - t0; a.clk[1] = ph[0]; a.clk[2] = ph[1]; a.clk[3] = ph[2]; }
  if (tid == 0) a.sink[blockIdx.x] = st[(tid * 7) % (RG * UB_LD)];
}

template <int RG, int SETS>     // RG rows per set
__global__ __launch_bounds__(DX_NT) void k_rowsets(const UbArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* st = smem;                                    // [SETS][RG][UB_LD]
  int* ictl = reinterpret_cast<int*>(st + SETS * RG * UB_LD);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, 0, ictl, tid, 8);
  const int group = __builtin_amdgcn_readfirstlane(ictl[0]), member = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  if (member >= DX_GROUP) return;
  float W[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) W[j] = a.wpack[((size_t)member * 16 + j) * DX_NT + tid];
  for (int i = tid; i < SETS * RG * UB_LD; i += DX_NT) st[i] = 0.01f * (float)(i % 97);
  __syncthreads();
  // exchange regions: [set][stage][RG * 256] granules per group
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)group * SETS * UB_NST * RG * DX_W;
  const bool tracer = group == 0 && member == 0 && tid == 0;
  long long t0 = 0, ph[3] = {0, 0, 0};
  for (int step = 0; step < a.steps; ++step) {
    if (tracer && step == 8) t0 = (long long)__builtin_readcyclecounter();
    const unsigned tag = (unsigned)step + 1u;
    if (SETS == 1) {
#pragma unroll 1
      for (int sgi = 0; sgi < UB_NST; ++sgi) {
        const int rd = (sgi & 1) * 256, wr = 256 - rd;
        dx_gu64* Xs = X + (size_t)sgi * RG * DX_W;
        const long long c0 = tracer ? (long long)__builtin_readcyclecounter() : 0;
        ub_compute<RG>(W, st + rd, lane, wave, member, Xs, tag, rt);
        const long long c1 = tracer ? (long long)__builtin_readcyclecounter() : 0;
        dx_gather<RG, DX_W, false, UB_LD>(Xs, tag, st, wr, 0, 0, tid, rt);
        const long long c2 = tracer ? (long long)__builtin_readcyclecounter() : 0;
        __syncthreads();
        if (tracer && step >= 8) { ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += (long long)__builtin_readcyclecounter() - c2; }
      }
    } else {
      float* stA = st; float* stB = st + RG * UB_LD;
      dx_gu64* XA = X; dx_gu64* XB = X + (size_t)UB_NST * RG * DX_W;
#pragma unroll 1
      for (int sgi = 0; sgi < UB_NST; ++sgi) {
        const int rd = (sgi & 1) * 256, wr = 256 - rd;
        const int prev = (sgi + UB_NST - 1) % UB_NST;
        // set A: stage sgi (its input vector was gathered at the end of the previous iteration)
        ub_compute<RG>(W, stA + rd, lane, wave, member, XA + (size_t)sgi * RG * DX_W, tag, rt);
        // set B's vector of the PREVIOUS stage has been in flight since the middle of the previous iteration
        if (!(step == 0 && sgi == 0))
          dx_gather<RG, DX_W, false, UB_LD>(XB + (size_t)prev * RG * DX_W, sgi == 0 ? tag - 1u : tag, stB, rd, 0, 0, tid, rt);
        __syncthreads();
        ub_compute<RG>(W, stB + rd, lane, wave, member, XB + (size_t)sgi * RG * DX_W, tag, rt);
        dx_gather<RG, DX_W, false, UB_LD>(XA + (size_t)sgi * RG * DX_W, tag, stA, wr, 0, 0, tid, rt);
        __syncthreads();
      }
    }
  }
  if (tracer) { a.clk[0] = (long long)__builtin_readcyclecounter() - t0; a.clk[1] = ph[0]; a.clk[2] = ph[1]; a.clk[3] = ph[2]; }
  if (tid == 0) a.sink[blockIdx.x] = st[(tid * 7) % (SETS * RG * UB_LD)];
}

template <int RG, int SETS, int PROTO = 0>
static int run(const char* name, int steps) {
  UbArgs a;
  std::vector<float> hw((size_t)DX_GROUP * 16 * DX_NT);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.02f * (float)((int)(i * 2654435761u >> 20) % 101 - 50) / 50.f;
  float* dw; CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  const size_t xg = (size_t)DX_NGROUP * (PROTO ? 1 : SETS) * UB_NST * RG * DX_W;      // 8-byte units (variant S: two parities of 4-byte granules)
  unsigned long long* xb; CK(hipMalloc(&xb, xg * 8));
  unsigned *ctl, *err; CK(hipMalloc(&ctl, 256)); CK(hipMalloc(&err, 256));
  long long* clk; CK(hipMalloc(&clk, 64)); float* sink; CK(hipMalloc(&sink, 256 * 4));
  a.wpack = dw; a.xbuf = xb; a.ctl = ctl; a.err = err; a.clk = clk; a.sink = sink; a.steps = steps;
  const size_t lds = std::max((size_t)(SETS * RG * UB_LD + 64) * sizeof(float), (size_t)96 * 1024);      // one workgroup per CU
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowsets<RG, SETS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowsets_s<RG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  if constexpr (RG >= 4) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowsets_t<RG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f; long long hclk = 0, hph[4] = {0, 0, 0, 0}; unsigned herr[64];
  for (int rep = 0; rep < 5; ++rep) {
    if (PROTO == 1) { std::vector<unsigned> fill(xg * 2, UB_SENT); CK(hipMemcpy(xb, fill.data(), xg * 8, hipMemcpyHostToDevice)); }
    else CK(hipMemset(xb, 0, xg * 8));
    CK(hipMemset(ctl, 0, 256)); CK(hipMemset(err, 0, 256)); CK(hipMemset(clk, 0, 64));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    if (PROTO == 2) { if constexpr (RG >= 4) hipLaunchKernelGGL((k_rowsets_t<RG>), dim3(DX_NGROUP * DX_GROUP), dim3(DX_NT), lds, 0, a); }
    else if (PROTO) hipLaunchKernelGGL((k_rowsets_s<RG>), dim3(DX_NGROUP * DX_GROUP), dim3(DX_NT), lds, 0, a);
    else hipLaunchKernelGGL((k_rowsets<RG, SETS>), dim3(DX_NGROUP * DX_GROUP), dim3(DX_NT), lds, 0, a);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(herr, err, 256, hipMemcpyDeviceToHost));
    if (herr[0]) { printf("%s: device error word %u\n", name, herr[0]); return 1; }
    if (ms < best) { best = ms; CK(hipMemcpy(hph, clk, 32, hipMemcpyDeviceToHost)); hclk = hph[0]; }
  }
  const double per_stage = (double)hclk / ((double)(steps - 8) * UB_NST * SETS);
  printf("%-44s %7.2f us per step (%d rows)  %7.0f clocks per step  %6.0f clocks per (set, stage)   protocol %u\n", name, best * 1e3 / steps,
         RG * SETS, (double)hclk / (steps - 8), per_stage, herr[8]);
  if (SETS == 1) {
    const double ns = (double)(steps - 8) * UB_NST;
    printf("%-44s   of a stage (wave 0 of member 0): pass + reduce + epilogue + publish %5.0f | poll until all granules carry the tag + LDS write %5.0f | barrier %5.0f\n",
           "", hph[1] / ns, hph[2] / ns, hph[3] / ns);
  }
  hipFree(dw); hipFree(xb); hipFree(ctl); hipFree(err); hipFree(clk); hipFree(sink);
  return 0;
}

int main() {
  const int steps = 136;
  printf("stage of k_decoder_xcd in isolation: 10 exchanges per step, 2 weight columns per wave and stage, 32 members per XCD\n");
  if (run<4, 1>("one set of 4 rows (C2 today)", steps)) return 1;
  if (run<2, 2>("two sets of 2 rows, pipelined", steps)) return 1;
  if (run<2, 1>("one set of 2 rows", steps)) return 1;
  if (run<8, 1>("one set of 8 rows (64-row pass today)", steps)) return 1;
  if (run<4, 2>("two sets of 4 rows, pipelined", steps)) return 1;
  if (run<1, 1>("one set of 1 row", steps)) return 1;
  printf("variant S: 4-byte granules with a sentinel instead of {value, tag}, two parity copies, pairs of values polled by one 8-byte load\n");
  if (run<4, 1, 1>("S: one set of 4 rows", steps)) return 1;
  if (run<8, 1, 1>("S: one set of 8 rows", steps)) return 1;
  if (run<2, 1, 1>("S: one set of 2 rows", steps)) return 1;
  if (run<1, 1, 1>("S: one set of 1 row", steps)) return 1;
  printf("variant T: {value, tag} granules as they are, two adjacent granules polled by one 16-byte load\n");
  if (run<4, 1, 2>("T: one set of 4 rows", steps)) return 1;
  if (run<8, 1, 2>("T: one set of 8 rows", steps)) return 1;
  return 0;
}
