// The inner loop of k_pointwise_chain / k_gemm_bf3 in isolation: 256 workgroups of 8 waves (two per SIMD), per k16 step every wave
// takes its weight fragments (hi, lo of two matrices: 4 x 16 bytes per lane) from an L2-resident pack, its activation fragments
// (hi, lo of two 32-row tiles: 4 x ds_read_b128) from LDS planes, and issues 12 v_mfma_f32_32x32x16_bf16 on four accumulators.
// Which of the three (weights, activations, the matrix pipe itself) keeps the loop at ~57 % of its MFMA time?
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_mmaloop ubench_mmaloop.hip ; run without arguments.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int W = 256, LDSW = W + 8, K16 = 16, NT = W / 32;

// MODE bit 0: weights streamed (else one constant fragment); bit 1: activations read from LDS every step (else constant);
// bit 2: sched_barrier fences as in the product kernels; WAVES = waves per workgroup (8: two per SIMD, 16: four per SIMD)
template <int MODE, int WAVES, int NL = 1, int PD = 1>
__global__ __launch_bounds__(64 * WAVES) void k(const unsigned short* __restrict__ bh, const unsigned short* __restrict__ bl,
                                                const unsigned short* __restrict__ bh2, const unsigned short* __restrict__ bl2, float* out,
                                                int layers, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* xhi = smem;
  unsigned short* xlo = smem + 64 * LDSW;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave % NT;
  for (int i = tid; i < 2 * 64 * LDSW; i += 64 * WAVES) smem[i] = (unsigned short)(0x3c00u + (i * 7u & 0xffu));
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // NL distinct layers of weights walked round-robin (NL x 512 KB: past a few MB the weights no longer stay in the XCD's 4 MB L2)
  auto boff = [&](int g) { return (size_t)((g / K16) % NL) * 4 * ((size_t)K16 * NT * 2 * 32 * 8) + ((((size_t)(g & (K16 - 1)) * NT + wn) * 2 + lh) * 32 + l31) * 8; };
  uint4 ring[PD + 1][4];
  auto loadb = [&](int g, uint4 (&f)[4]) {
    const size_t o = (MODE & 1) ? boff(g) : boff(0);
    f[0] = *reinterpret_cast<const uint4*>(bh + o); f[1] = *reinterpret_cast<const uint4*>(bl + o);
    f[2] = *reinterpret_cast<const uint4*>(bh2 + o); f[3] = *reinterpret_cast<const uint4*>(bl2 + o);
  };
  auto mma = [&](int g, const uint4 (&f)[4]) {
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int off = (tm * 32 + l31) * LDSW + 16 * ((MODE & 2) ? (g & (K16 - 1)) : 0) + 8 * lh;
      ah[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xhi + off));
      al[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xlo + off));
    }
    const bf16x8 wh = __builtin_bit_cast(bf16x8, f[0]), wl = __builtin_bit_cast(bf16x8, f[1]);
    const bf16x8 wh2 = __builtin_bit_cast(bf16x8, f[2]), wl2 = __builtin_bit_cast(bf16x8, f[3]);
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const bf16x8 aa = term == 0 ? al[tm] : ah[tm];
        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa, term == 1 ? wl : wh, acc[tm], 0, 0, 0);
        acc[2 + tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa, term == 1 ? wl2 : wh2, acc[2 + tm], 0, 0, 0);
      }
  };
  const long long t0 = clock64();
#pragma unroll
  for (int d = 0; d < PD; ++d) loadb(d, ring[d]);
  for (int g = 0; g < layers * K16; g += PD + 1) {          // PD groups in flight behind every MFMA group
#pragma unroll
    for (int d = 0; d <= PD; ++d) {
      loadb(g + d + PD, ring[(d + PD) % (PD + 1)]);
      if (MODE & 4) __builtin_amdgcn_sched_barrier(0);
      mma(g + d, ring[d]);
      if (MODE & 4) __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int WAVES, int NL = 1, int PD = 1>
static void run(const char* what, const unsigned short* pk, size_t plane, float* out, long long* cyc) {
  const int layers = 60, grid = 256;        // 960 steps: a multiple of PD + 1 for PD = 1, 2, 3
  const size_t lds = 2 * 64 * LDSW * sizeof(unsigned short);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, WAVES, NL, PD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, WAVES, NL, PD>), dim3(grid), dim3(64 * WAVES), lds, 0, pk, pk + plane, pk + 2 * plane, pk + 3 * plane, out, layers, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, WAVES, NL, PD>), dim3(grid), dim3(64 * WAVES), lds, 0, pk, pk + plane, pk + 2 * plane, pk + 3 * plane, out, layers, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double steps = (double)layers * K16, mfma = steps * 12;
  const double flop = mfma * 32768.0 * WAVES * grid;
  printf("%-66s layers %d prefetch %d waves/WG %2d: %7.3f ms  %7.1f TFLOP/s bf16  clocks per k16 step (wave 0) %6.0f  (12 MFMAs = 384 at full rate x %d waves per SIMD)\n",
         what, NL, PD, WAVES, ms, flop / ms * 1e-9, (double)c / steps, WAVES / 4);
}

int main() {
  const size_t plane = (size_t)K16 * NT * 2 * 32 * 8;           // one matrix plane of one layer: 64 K bf16 = 128 KB
  std::vector<unsigned short> h(8 * 4 * plane);           // 8 layers x (hi, lo, hi2, lo2)
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00u + (i * 13u & 0x7fu));
  unsigned short* pk; float* out; long long* cyc;
  hipMalloc(&pk, h.size() * 2); hipMemcpy(pk, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&out, sizeof(float) * 256 * 1024); hipMalloc(&cyc, 8);
  run<0, 8>("constant fragments (matrix pipe only)", pk, plane, out, cyc);
  run<7, 8, 1, 1>("product loop, 512 KB of weights (L2-hot)", pk, plane, out, cyc);
  run<7, 8, 8, 1>("product loop, 4 MB of weights", pk, plane, out, cyc);
  run<7, 8, 8, 2>("product loop, 4 MB of weights", pk, plane, out, cyc);
  run<7, 8, 8, 3>("product loop, 4 MB of weights", pk, plane, out, cyc);
  run<5, 8, 8, 1>("4 MB of weights, constant activations", pk, plane, out, cyc);
  run<7, 16, 8, 1>("product loop, 4 MB of weights", pk, plane, out, cyc);
  run<7, 16, 8, 2>("product loop, 4 MB of weights", pk, plane, out, cyc);
  return 0;
}
