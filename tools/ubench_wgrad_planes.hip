// ubench_wgrad_planes.hip -- VERDICT r05, next 6 (i): weight gradients from pre-split operands (csrc/taco_wgrad_planes.h: k_wp_split +
// k_wp_gemm) next to the kernel of the training step today (k_wgrad_bf3, which converts inside the product loop), on the weight-gradient
// shapes of the C4-shard step.  Both paths write per-slice partial tiles and add them in slice order (the deterministic default);
// every result is checked on a sample of output elements against a double-precision sum.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I multi-speaker-tacotron-tensorflow_amd/csrc tools/ubench_wgrad_planes.hip -o tools/ubench_wgrad_planes
#include "taco_bigru_xcd.h"
#include "taco_backward_kernels.h"
#include "taco_wgrad_planes.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

struct Shape { const char* name; int M, T, K, N, kw, padl; };

template <int WK, int WN, int SM, int NB>
static int launch_gemm(const WpGemmArgs& g, int kw_nsplit, hipStream_t st) {
  const size_t lds = (size_t)NB * 3 * 2 * (WK + WN) * SM * 1024;
  static bool set = false;
  if (!set) { CK(hipFuncSetAttribute((const void*)k_wp_gemm<WK, WN, SM, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set = true; }
  hipLaunchKernelGGL((k_wp_gemm<WK, WN, SM, NB>), dim3(cdiv(g.K, 64 * WK), cdiv(g.N, 64 * WN), kw_nsplit), dim3(64 * WK * WN), lds, st, g);
  return 0;
}
struct Cfg { const char* name; int tk, tn, sm; };
static const Cfg cfgs[] = {{"128 x 128, 16 rows x 3 (72 KB)", 128, 128, 1}, {"256 x 128, 16 rows x 3 (108 KB)", 256, 128, 1}, {"128 x 256, 16 rows x 3 (108 KB)", 128, 256, 1},
                           {"256 x 256, 16 rows x 3 (144 KB)", 256, 256, 1}, {"256 x 128, 32 rows x 2 (144 KB)", 256, 128, 2}, {"256 x 256, 16 rows x 2 (96 KB)", 256, 256, 1}};
static int launch_cfg(int cfg, const WpGemmArgs& g, int z, hipStream_t st) {
  switch (cfg) {
    case 0: return launch_gemm<2, 2, 1, 3>(g, z, st);
    case 1: return launch_gemm<4, 2, 1, 3>(g, z, st);
    case 2: return launch_gemm<2, 4, 1, 3>(g, z, st);
    case 3: return launch_gemm<4, 4, 1, 3>(g, z, st);
    case 4: return launch_gemm<4, 2, 2, 2>(g, z, st);
    case 5: return launch_gemm<4, 4, 1, 2>(g, z, st);
  }
  return 1;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const Shape shapes[] = {
    {"post proj_1   2048 x 256 x 3 taps", 16384, 512, 2048, 256, 3, 1},
    {"post highway   256 x 256", 16384, 0, 256, 256, 1, 0},
    {"post bank k=8   80 x 256 x 8 taps", 16384, 512, 80, 256, 8, 3},
    {"post gru x     256 x 512", 16384, 512, 256, 512, 1, 0},
    {"post gru h     256 x 512 shift", 16384, 512, 256, 512, 1, 1},
    {"linear head    512 x 1025", 16384, 0, 512, 1025, 1, 0},
    {"enc bank k=16  128 x 128 x 16 taps", 4096, 128, 128, 128, 16, 7},
    {"enc proj_1    2048 x 128 x 3 taps", 4096, 128, 2048, 128, 3, 1},
    {"dec hoisted    256 x 768", 4096, 0, 256, 768, 1, 0},
  };
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t part_cap = (size_t)64 << 20;       // floats
  float* part; CK(hipMalloc(&part, part_cap * 4));
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (const Shape& s : shapes) {
    const int M = s.M, K = s.K, N = s.N, kw = s.kw;
    std::vector<float> hx((size_t)M * K), hy((size_t)M * N);
    for (auto& v : hx) v = nd(rng);
    for (auto& v : hy) v = nd(rng) * 0.1f;
    float *dx, *dyv, *dw_old, *dw_new;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dyv, hy.size() * 4));
    const size_t per = (size_t)kw * K * N;
    CK(hipMalloc(&dw_old, per * 4)); CK(hipMalloc(&dw_new, per * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dyv, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
    // ---- today's kernel, as run_wgrad launches it in deterministic mode ----
    WgArgs g; g.ygather = nullptr; g.x = dx; g.gather = nullptr; g.dy = dyv; g.dw = dw_old; g.ldx = K; g.ldy = N; g.lddw = N; g.M = M; g.T = s.T; g.K = K; g.N = N;
    g.kw = kw; g.padl = s.padl;
    const long t128 = (long)cdiv(K, 128) * cdiv(N, 128) * kw, t64 = (long)cdiv(K, 64) * cdiv(N, 64) * kw;
    const bool big = t128 >= 64;
    { const long tiles = big ? t128 : t64; int rpb = 1024; const long want = big ? 768 : 1024;
      while (rpb > 128 && tiles * cdiv(M, rpb) < want) rpb >>= 1;
      g.rpb = rpb; }
    while ((size_t)cdiv(M, g.rpb) * per > part_cap) g.rpb *= 2;
    g.part = part;
    const int nsplit_old = cdiv(M, g.rpb);
    auto run_old = [&]() {
      hipMemsetAsync(dw_old, 0, per * 4, st);
      if (big) hipLaunchKernelGGL((k_wgrad_bf3<4>), dim3(cdiv(K, 128), cdiv(N, 128), kw * nsplit_old), dim3(256), 0, st, g);
      else hipLaunchKernelGGL((k_wgrad_bf3<1>), dim3(cdiv(K, 64), cdiv(N, 64), kw * nsplit_old), dim3(64), 0, st, g);
      hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, (const float*)g.part, nsplit_old, kw, K, N, dw_old, N);
    };
    run_old(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) run_old();
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms_old; CK(hipEventElapsedTime(&ms_old, e0, e1)); ms_old /= reps;
    // ---- pre-split planes ----
    const int Mp = cdiv(M, 64) * 64;
    const bool a_per_tap = K <= N;           // the narrower operand carries the tap copies
    const int na = a_per_tap ? kw : 1, nbc = a_per_tap ? 1 : kw;
    const bool shifted = kw > 1 || s.padl != 0;
    uint4 *pa, *pb;
    CK(hipMalloc(&pa, wp_plane_uint4(K, Mp, na) * 16)); CK(hipMalloc(&pb, wp_plane_uint4(N, Mp, nbc) * 16));
    WpSplitArgs sa; sa.src = dx; sa.gather = nullptr; sa.out = pa; sa.ld = K; sa.M = M; sa.T = s.T; sa.C = K; sa.Mp = Mp;
    sa.ncopy = na; sa.sigma0 = (shifted && a_per_tap) ? -s.padl : 0; sa.dsigma = a_per_tap ? 1 : 0;
    WpSplitArgs sb; sb.src = dyv; sb.gather = nullptr; sb.out = pb; sb.ld = N; sb.M = M; sb.T = s.T; sb.C = N; sb.Mp = Mp;
    sb.ncopy = nbc; sb.sigma0 = (shifted && !a_per_tap) ? s.padl : 0; sb.dsigma = a_per_tap ? 0 : -1;
    WpGemmArgs gg; memset(&gg, 0, sizeof gg); gg.a = pa; gg.b = pb; gg.K = K; gg.N = N; gg.Mp = Mp; gg.kw = kw; gg.a_per_tap = a_per_tap ? 1 : 0; gg.part = part; gg.dw = dw_new; gg.lddw = N;
    printf("%-38s M %5d  today %8.1f us (%s, %d slices)\n", s.name, M, ms_old * 1e3, big ? "128-tile" : "64-tile", nsplit_old);
    // double-precision sample
    std::vector<int> si(512); std::vector<double> ref(512);
    for (int q = 0; q < 512; ++q) {
      const int tap = rng() % kw, k = rng() % K, n = rng() % N; si[q] = (tap * K + k) * N + n;
      const int sh = tap - s.padl; double acc = 0;
      for (int m = 0; m < M; ++m) {
        if (s.T > 0) { const int t = m % s.T; if (t + sh < 0 || t + sh >= s.T) continue; } else if (sh != 0) continue;
        acc += (double)hx[(size_t)(m + sh) * K + k] * (double)hy[(size_t)m * N + n];
      }
      ref[q] = acc;
    }
    std::vector<float> ho(per), hn(per);
    CK(hipMemcpy(ho.data(), dw_old, per * 4, hipMemcpyDeviceToHost));
    double eo = 0, scale = 0;
    for (int q = 0; q < 512; ++q) { eo = fmax(eo, fabs(ho[si[q]] - ref[q])); scale = fmax(scale, fabs(ref[q])); }
    printf("    today: max |err| %.2e (sample max |dW| %.2e)\n", eo, scale);
    for (int cfg = 0; cfg < 6; ++cfg) {
      const int sm = cfgs[cfg].sm;
      const long tiles = (long)cdiv(K, cfgs[cfg].tk) * cdiv(N, cfgs[cfg].tn) * kw;
      for (int want : {384, 768}) {
        int rpb = Mp;
        while (rpb > 256 && tiles * cdiv(Mp, rpb) < want) rpb >>= 1;
        rpb = cdiv(rpb, 16 * sm) * 16 * sm;
        while ((size_t)cdiv(Mp, rpb) * per > part_cap) rpb *= 2;
        gg.rpb = rpb;
        const int nsplit = cdiv(Mp, rpb);
        auto run_new = [&](bool with_split) {
          hipMemsetAsync(dw_new, 0, per * 4, st);
          if (with_split) {
            hipLaunchKernelGGL(k_wp_split, dim3(cdiv(cdiv(K, 32), 4), Mp / 64, na), dim3(256), 0, st, sa);
            hipLaunchKernelGGL(k_wp_split, dim3(cdiv(cdiv(N, 32), 4), Mp / 64, nbc), dim3(256), 0, st, sb);
          }
          launch_cfg(cfg, gg, kw * nsplit, st);
          hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, (const float*)gg.part, nsplit, kw, K, N, dw_new, N);
        };
        run_new(true); CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        CK(hipMemcpy(hn.data(), dw_new, per * 4, hipMemcpyDeviceToHost));
        double en = 0, dmax = 0;
        for (int q = 0; q < 512; ++q) en = fmax(en, fabs(hn[si[q]] - ref[q]));
        for (size_t q = 0; q < per; ++q) dmax = fmax(dmax, fabs((double)hn[q] - (double)ho[q]));
        float ms_all, ms_gemm;
        CK(hipEventRecord(e0, st)); for (int r = 0; r < reps; ++r) run_new(true); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms_all, e0, e1)); ms_all /= reps;
        CK(hipEventRecord(e0, st)); for (int r = 0; r < reps; ++r) run_new(false); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms_gemm, e0, e1)); ms_gemm /= reps;
        const double tf = 12.0 * M * (double)K * N * kw / (ms_gemm * 1e-3) * 1e-12;
        printf("    planes, %-32s %4d slices: split + product + sum %8.1f us, product + sum %8.1f us (%6.1f TF/s bf16)  max |err| %.2e  max |new - today| %.2e\n",
               cfgs[cfg].name, nsplit, ms_all * 1e3, ms_gemm * 1e3, tf, en, dmax);
      }
    }
    hipFree(dx); hipFree(dyv); hipFree(dw_old); hipFree(dw_new); hipFree(pa); hipFree(pb);
  }
  return 0;
}
