// ubench_mfma_scan.hip -- DESIGN section 6 item (6), as an experiment: one phase of the post-net scan (k_bigru_duo's candidate phase:
// h'[r][n] = tanh(x[n] + sum_k h[r][k] W[k][n]), 256 units over 32 members x 8, RG rows per group, one exchange per step) in isolation,
// on the kernel's own machinery (census, XCD-local {value, tag} granules, bounded polls), computed two ways:
//   V  as today: wave w of a member owns unit 8m + w, its 256 weights in 4 VGPRs per lane, dx_pass + dx_reduce (VALU FMAs, DPP /
//      permlane butterfly), epilogue on quad 0 -- cost grows with the rows per group
//   M  on the matrix cores: producers publish a value already split into bf16 planes ({hi, mid, lo, tag16} in the same 8-byte
//      granule), the gather writes the planes into LDS in A-fragment order, wave w contracts k-step w (K = 32) of all 16 rows x
//      16 columns (8 used) with v_mfma_f32_16x16x32_bf16 against register-resident weight planes, the eight partial tiles meet in
//      LDS, wave 0 finishes (sum, tanh, split, publish) -- cost independent of the rows per group up to 16
// Both variants run the SAME recurrence from the same state for `steps` steps; the final state is compared with a double-precision
// host recurrence (so the experiment checks the operand layouts and the split arithmetic as well as the clocks).
// Prints microseconds per step (HIP events), clocks per step and per part of a step (shader clock of group 0 / member 0 / thread 0).
//   hipcc --offload-arch=gfx950 -O3 -I multi-speaker-tacotron-tensorflow_amd/csrc tools/ubench_mfma_scan.hip -o tools/ubench_mfma_scan
// Measured (profiles/r03_ubench_mfma_scan.txt): M is row-independent (~900-1000 clocks of compute + publish) but V is 557 at 4 rows and 953 at 8:
// no gain at the row counts the kernels run.  The numerics (also emulated on the CPU in tools/sim_split_recurrence.py): six products 4e-7, as V.
#include "taco_decoder_xcd.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define US_H 256
#define US_NT 512
#define US_ROWB (US_H + 8)          // bf16 elements per LDS row of a plane: +16 bytes, so the 16 rows of an A fragment hit 16 different bank quads

struct UsArgs {
  const float* wpack;               // V: [32 members][4][512]
  const uint4* wplanes;             // M: [32 members][PW planes][512]   (8 bf16: this lane's B fragment of its wave's k-step)
  const float* xin;                 // [256]
  const float* h0;                  // [8 groups][RG][256]
  float* hout;                      // [8 groups][RG][256] final state
  unsigned long long* xbuf; unsigned* ctl; unsigned* err; long long* clk; int steps;
};

__device__ __forceinline__ unsigned us_bf16_rne(float x) {            // bits of the nearest-even bfloat16 (finite inputs)
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float us_bf16_val(unsigned b) { return __uint_as_float(b << 16); }

// ---------------------------------------------------------------- variant V ----------------------------------------------------------------
template <int RG>
__global__ __launch_bounds__(US_NT) void k_scan_v(const UsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RL = DxRL<RG>::value;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* hs = smem;                                    // [RG][256]
  int* ictl = reinterpret_cast<int*>(hs + RG * US_H);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, 0, ictl, tid, 8);
  const int group = __builtin_amdgcn_readfirstlane(ictl[0]), member = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  if (member >= DX_GROUP) return;
  float W[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) W[j] = a.wpack[((size_t)member * 4 + j) * US_NT + tid];
  for (int i = tid; i < RG * US_H; i += US_NT) hs[i] = a.h0[(size_t)group * RG * US_H + i];
  __syncthreads();
  const int n = member * 8 + wave;
  const float xn = a.xin[n];
  const bool epl = lane < (RG >= 4 ? 4 : RG);
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)group * 2 * RG * US_H;     // [parity][RG][256]
  const bool tracer = group == 0 && member == 0 && tid == 0;
  long long t0 = 0, ph[3] = {0, 0, 0};
  for (int s = 0; s < a.steps; ++s) {
    if (tracer && s == 8) t0 = (long long)__builtin_readcyclecounter();
    const unsigned tag = (unsigned)s + 1u;
    dx_gu64* Xs = X + (size_t)(s & 1) * RG * US_H;
    const long long c0 = tracer ? (long long)__builtin_readcyclecounter() : 0;
    float acc[1][RG], sm[1][RL];
    dx_zero<1, RG>(acc);
    dx_pass<0, 1, RG, 4, US_H>(W, hs, lane, acc);
    dx_reduce<1, RG>(acc, sm, lane);
#pragma unroll
    for (int q = 0; q < RL; ++q) {
      const float v = taco_tanh_fast(sm[0][q] + xn);
      if (epl) dx_publish(Xs + dx_row<RG>(lane & 3, q) * US_H + n, v, tag, rt);
    }
    const long long c1 = tracer ? (long long)__builtin_readcyclecounter() : 0;
    dx_gather<RG, US_H, false, US_H, US_NT>(Xs, tag, hs, 0, 0, 0, tid, rt);
    const long long c2 = tracer ? (long long)__builtin_readcyclecounter() : 0;
    __syncthreads();
    if (tracer && s >= 8) { ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += (long long)__builtin_readcyclecounter() - c2; }
  }
  if (tracer) { a.clk[0] = (long long)__builtin_readcyclecounter() - t0; a.clk[1] = ph[0]; a.clk[2] = ph[1]; a.clk[3] = ph[2]; }
  if (member == 0) for (int i = tid; i < RG * US_H; i += US_NT) a.hout[(size_t)group * RG * US_H + i] = hs[i];
}

// ---------------------------------------------------------------- variant M ----------------------------------------------------------------
// PA planes of the state, PW planes of the weights; products kept: every (i, j) with i + j <= PMAX (plane i is ~2^-8i of the value)
template <int RG, int PA, int PW, int PMAX>
__global__ __launch_bounds__(US_NT) void k_scan_m(const UsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(RG <= 16, "one MFMA row tile");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned short* hp = reinterpret_cast<unsigned short*>(smem);                 // [PA][16][US_ROWB] bf16 planes of the state (rows >= RG unused)
  float* part = smem + (PA * 16 * US_ROWB) / 2;                                 // [8 waves][8 columns][16 rows] partial sums
  int* ictl = reinterpret_cast<int*>(part + 8 * 8 * 16);
  dx_gu32* errw = (dx_gu32*)a.err;
  dx_census((dx_gu32*)a.ctl, errw, 0, ictl, tid, 8);
  const int group = __builtin_amdgcn_readfirstlane(ictl[0]), member = __builtin_amdgcn_readfirstlane(ictl[1]);
  DxRt rt; rt.err = errw; rt.wt = ictl[2] != 0; rt.dead = ictl[3] != 0;
  if (member >= DX_GROUP) return;
  bf16x8 Wb[PW];
#pragma unroll
  for (int p = 0; p < PW; ++p) Wb[p] = __builtin_bit_cast(bf16x8, a.wplanes[((size_t)member * PW + p) * US_NT + tid]);
  for (int i = tid; i < RG * US_H; i += US_NT) {
    float rest = a.h0[(size_t)group * RG * US_H + i];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const unsigned b = us_bf16_rne(rest);
      hp[(p * 16 + i / US_H) * US_ROWB + (i % US_H)] = (unsigned short)b;
      rest -= us_bf16_val(b);
    }
  }
  __syncthreads();
  // A fragment of this lane: row lane & 15, k = 32 wave + 8 (lane >> 4) .. + 7
  const int arow = lane & 15, kq = lane >> 4;
  const bool ahas = arow < RG;
  const int aoff = arow * US_ROWB + 32 * wave + 8 * kq;                         // bf16 elements; 16-byte aligned
  // C fragment: column lane & 15, rows 4 (lane >> 4) .. + 3
  const bool cwrites = (lane & 15) < 8 && 4 * kq < RG;
  float* pdst = part + ((size_t)wave * 8 + (lane & 15)) * 16 + 4 * kq;
  // finishing role (wave 0): thread t < 8 RG: column t & 7, row t >> 3
  const bool fin = tid < 8 * RG;
  const int fc = tid & 7, fr = tid >> 3;
  const int fn = member * 8 + fc;
  const float xn = fin ? a.xin[fn] : 0.f;
  dx_gu64* X = (dx_gu64*)a.xbuf + (size_t)group * 2 * RG * US_H;
  constexpr int NI = (RG * US_H + US_NT - 1) / US_NT;
  const bool tracer = group == 0 && member == 0 && tid == 0;
  long long t0 = 0, ph[3] = {0, 0, 0};
  for (int s = 0; s < a.steps; ++s) {
    if (tracer && s == 8) t0 = (long long)__builtin_readcyclecounter();
    const unsigned tag16 = ((unsigned)s + 1u) & 0xFFFFu;
    dx_gu64* Xs = X + (size_t)(s & 1) * RG * US_H;
    const long long c0 = tracer ? (long long)__builtin_readcyclecounter() : 0;
    bf16x8 A[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (ahas) v = *reinterpret_cast<const uint4*>(hp + p * 16 * US_ROWB + aoff);
      A[p] = __builtin_bit_cast(bf16x8, v);
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};            // two dependent chains of equal length, small terms first in each
    int nth = 0;
#pragma unroll
    for (int lev = PMAX; lev >= 0; --lev)
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int j = lev - i;
        if (j < 0 || j >= PW) continue;
        if ((nth++ & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], Wb[j], acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], Wb[j], acc1, 0, 0, 0);
      }
    const f32x4 acc = acc0 + acc1;
    if (cwrites) *reinterpret_cast<float4*>(pdst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (fin) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) sum += part[((size_t)w * 8 + fc) * 16 + fr];
      float rest = taco_tanh_fast(sum + xn);
      unsigned long long g = (unsigned long long)tag16 << 48;
#pragma unroll
      for (int p = 0; p < PA; ++p) {
        const unsigned b = us_bf16_rne(rest);
        g |= (unsigned long long)b << (16 * p);
        rest -= us_bf16_val(b);
      }
      if (rt.wt) __hip_atomic_store(Xs + fr * US_H + fn, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(Xs + fr * US_H + fn, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const long long c1 = tracer ? (long long)__builtin_readcyclecounter() : 0;
    if ((RG * US_H >= US_NT) || tid < RG * US_H) {
      unsigned long long g[NI];
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int u = 0; u < NI; ++u) g[u] = __hip_atomic_load(Xs + tid + u * US_NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < NI; ++u) ok = ok && ((unsigned)(g[u] >> 48) == tag16);
        if (ok || rt.dead) break;
        if ((++spins & 1023u) == 0) {
          if (spins >= DX_SPIN_LIMIT || __hip_atomic_load(rt.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            __hip_atomic_store(rt.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rt.dead = true;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NI; ++u) {
        const int i = u * US_NT + tid, r = i / US_H, k = i % US_H;
#pragma unroll
        for (int p = 0; p < PA; ++p) hp[(p * 16 + r) * US_ROWB + k] = (unsigned short)(g[u] >> (16 * p));
      }
    }
    const long long c2 = tracer ? (long long)__builtin_readcyclecounter() : 0;
    __syncthreads();
    if (tracer && s >= 8) { ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += (long long)__builtin_readcyclecounter() - c2; }
  }
  if (tracer) { a.clk[0] = (long long)__builtin_readcyclecounter() - t0; a.clk[1] = ph[0]; a.clk[2] = ph[1]; a.clk[3] = ph[2]; }
  if (member == 0)
    for (int i = tid; i < RG * US_H; i += US_NT) {
      float v = 0.f;
#pragma unroll
      for (int p = 0; p < PA; ++p) v += us_bf16_val(hp[(p * 16 + i / US_H) * US_ROWB + (i % US_H)]);
      a.hout[(size_t)group * RG * US_H + i] = v;
    }
}

// ---------------------------------------------------------------- host ----------------------------------------------------------------
static unsigned h_bf16_rne(float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16; }
static float h_bf16_val(unsigned b) { unsigned u = b << 16; float f; memcpy(&f, &u, 4); return f; }

template <int RG, int PA, int PW, int PMAX>
static int run(const char* name, int steps) {
  const int H = US_H;
  std::vector<float> W((size_t)H * H), xin(H), h0((size_t)DX_NGROUP * RG * H);
  unsigned seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((seed >> 8) & 0xFFFF) / 65536.f; };
  for (auto& w : W) w = (rnd() * 2.f - 1.f) * 0.108f;                   // Glorot-uniform limit of a [512, 256] kernel
  for (int n = 0; n < H; ++n) xin[n] = (rnd() * 2.f - 1.f) * 0.5f;
  for (auto& v : h0) v = (rnd() * 2.f - 1.f) * 0.8f;
  // reference: double recurrence on the float32 weights
  std::vector<double> ref(h0.begin(), h0.end()), nxt(ref.size());
  for (int s = 0; s < steps; ++s) {
    for (int gr = 0; gr < DX_NGROUP * RG; ++gr)
      for (int n = 0; n < H; ++n) {
        double acc = xin[n];
        for (int k = 0; k < H; ++k) acc += ref[(size_t)gr * H + k] * (double)W[(size_t)k * H + n];
        nxt[(size_t)gr * H + n] = std::tanh(acc);
      }
    ref.swap(nxt);
  }
  // V pack: [member][j][tid]: W[4 lane + j][8 member + wave]
  std::vector<float> wv((size_t)DX_GROUP * 4 * US_NT);
  for (int m = 0; m < DX_GROUP; ++m)
    for (int j = 0; j < 4; ++j)
      for (int t = 0; t < US_NT; ++t) wv[((size_t)m * 4 + j) * US_NT + t] = W[(size_t)(4 * (t & 63) + j) * H + 8 * m + (t >> 6)];
  // M planes: [member][plane][tid][8]: B fragment of lane (col = lane & 15, k = 32 wave + 8 (lane >> 4) + e); columns 8..15 are zero
  std::vector<unsigned short> wm((size_t)DX_GROUP * (PW > 0 ? PW : 1) * US_NT * 8, 0);
  if (PW > 0)
    for (int m = 0; m < DX_GROUP; ++m)
      for (int t = 0; t < US_NT; ++t) {
        const int lane = t & 63, wave = t >> 6, col = lane & 15;
        if (col >= 8) continue;
        for (int e = 0; e < 8; ++e) {
          float rest = W[(size_t)(32 * wave + 8 * (lane >> 4) + e) * H + 8 * m + col];
          for (int p = 0; p < PW; ++p) {
            const unsigned b = h_bf16_rne(rest);
            wm[(((size_t)m * PW + p) * US_NT + t) * 8 + e] = (unsigned short)b;
            rest -= h_bf16_val(b);
          }
        }
      }
  UsArgs a;
  float *dwv, *dx, *dh0, *dho; uint4* dwm;
  CK(hipMalloc(&dwv, wv.size() * 4)); CK(hipMemcpy(dwv, wv.data(), wv.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dwm, wm.size() * 2)); CK(hipMemcpy(dwm, wm.data(), wm.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&dx, H * 4)); CK(hipMemcpy(dx, xin.data(), H * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dh0, h0.size() * 4)); CK(hipMemcpy(dh0, h0.data(), h0.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dho, h0.size() * 4));
  const size_t xg = (size_t)DX_NGROUP * 2 * RG * H;
  unsigned long long* xb; CK(hipMalloc(&xb, xg * 8));
  unsigned *ctl, *err; CK(hipMalloc(&ctl, 256)); CK(hipMalloc(&err, 256));
  long long* clk; CK(hipMalloc(&clk, 64));
  a.wpack = dwv; a.wplanes = dwm; a.xin = dx; a.h0 = dh0; a.hout = dho; a.xbuf = xb; a.ctl = ctl; a.err = err; a.clk = clk; a.steps = steps;
  const size_t lds = 96 * 1024;                                                 // one workgroup per CU
  if constexpr (PW == 0) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan_v<RG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
  else { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan_m<RG, PA, PW, PMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f; long long hph[4] = {0, 0, 0, 0}; unsigned herr[64];
  std::vector<float> got(h0.size());
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(xb, 0, xg * 8)); CK(hipMemset(ctl, 0, 256)); CK(hipMemset(err, 0, 256)); CK(hipMemset(clk, 0, 64));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    if constexpr (PW == 0) hipLaunchKernelGGL((k_scan_v<RG>), dim3(DX_NGROUP * DX_GROUP), dim3(US_NT), lds, 0, a);
    else hipLaunchKernelGGL((k_scan_m<RG, PA, PW, PMAX>), dim3(DX_NGROUP * DX_GROUP), dim3(US_NT), lds, 0, a);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(herr, err, 256, hipMemcpyDeviceToHost));
    if (herr[0]) { printf("%s: device error word %u\n", name, herr[0]); return 1; }
    if (ms < best) { best = ms; CK(hipMemcpy(hph, clk, 32, hipMemcpyDeviceToHost)); }
  }
  CK(hipMemcpy(got.data(), dho, got.size() * 4, hipMemcpyDeviceToHost));
  double emax = 0, amax = 0;
  for (size_t i = 0; i < got.size(); ++i) { emax = std::max(emax, std::fabs((double)got[i] - ref[i])); amax = std::max(amax, std::fabs(ref[i])); }
  const double ns = (double)(steps - 8);
  printf("%-34s %2d rows  %6.2f us per step  %6.0f clocks per step = compute + publish %5.0f | poll + LDS write %5.0f | barrier %5.0f   max |h - h_double| %.2e (|h| <= %.2f)  protocol %u\n",
         name, RG, best * 1e3 / steps, hph[0] / ns, hph[1] / ns, hph[2] / ns, hph[3] / ns, emax, amax, herr[8]);
  for (void* p : {(void*)dwv, (void*)dwm, (void*)dx, (void*)dh0, (void*)dho, (void*)xb, (void*)ctl, (void*)err, (void*)clk}) (void)hipFree(p);
  return emax < 1e-3 ? 0 : 2;
}

int main() {
  const int steps = 264;
  printf("one phase of the post-net scan in isolation (H = 256, 32 members x 8 units per XCD, one exchange per step, %d steps)\n", steps);
  int rc = 0;
  rc |= run<1, 0, 0, 0>("V: VALU passes + DPP reduction", steps);
  rc |= run<4, 0, 0, 0>("V: VALU passes + DPP reduction", steps);
  rc |= run<8, 0, 0, 0>("V: VALU passes + DPP reduction", steps);
  rc |= run<4, 3, 3, 2>("M: 3 x 3 planes, six products", steps);
  rc |= run<8, 3, 3, 2>("M: 3 x 3 planes, six products", steps);
  rc |= run<16, 3, 3, 2>("M: 3 x 3 planes, six products", steps);
  rc |= run<4, 3, 2, 2>("M: 3 x 2 planes, five products", steps);
  rc |= run<8, 3, 2, 2>("M: 3 x 2 planes, five products", steps);
  rc |= run<4, 2, 2, 1>("M: 2 x 2 planes, three products", steps);
  rc |= run<8, 2, 2, 1>("M: 2 x 2 planes, three products", steps);
  return rc;
}
