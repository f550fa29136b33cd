// MFMA issue-rate probe for gfx950: v_mfma_f32_32x32x16_bf16 with NACC independent accumulators per wave, W waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_mfma ubench_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(blockIdx.x + e); }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC> void run(int wgs_per_cu, int iters) {
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * 256 * 256 * 8); hipMalloc(&cyc, 8);
  const int grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, cyc); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double nm = (double)iters * 4 * NACC;                       // MFMAs per wave
  const double flop = nm * 32768.0 * 4 * grid;                      // 4 waves per WG
  printf("NACC=%d wgs/CU=%d: %.3f ms, %.1f TFLOP/s, clock64 ticks per MFMA (wave 0) %.1f\n", NACC, wgs_per_cu, ms, flop / ms * 1e-9, (double)c / nm);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1>(1, 4000); run<2>(1, 4000); run<4>(1, 4000); run<4>(2, 4000); run<4>(4, 2000);
  return 0;
}
