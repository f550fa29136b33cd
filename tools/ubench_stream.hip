// ubench_stream.hip -- per-CU streaming rate of an L2/MALL-resident weight matrix with a fused matvec:
// G workgroups each repeatedly stream the SAME W [K x N] fp32 (row-major), 4 columns per thread, K split
// over thread groups, R rows of x in LDS.  Reports us per pass and GB/s per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int R, int NT>
__global__ __launch_bounds__(NT) void k_stream(const float* __restrict__ W, int K, int N, float* out, int passes) {
  __shared__ float xs[R][1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < R * 1024; i += NT) xs[i / 1024][i % 1024] = 0.001f * (i % 7);
  __syncthreads();
  const int NC = N / 4, KS = NT / NC;          // column groups, K slices
  const int cg = tid % NC, ks = tid / NC;
  float4 acc[R];
  for (int r = 0; r < R; ++r) acc[r] = make_float4(0, 0, 0, 0);
  if (ks < KS) {
    const int kper = (K + KS - 1) / KS, k0 = ks * kper, k1 = min(K, k0 + kper);
    for (int p = 0; p < passes; ++p) {
      const float4* wp = reinterpret_cast<const float4*>(W) + cg;
#pragma unroll 8
      for (int k = k0; k < k1; ++k) {
        const float4 w = wp[(size_t)k * NC];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float x = xs[r][k];
          acc[r].x += x * w.x; acc[r].y += x * w.y; acc[r].z += x * w.z; acc[r].w += x * w.w;
        }
      }
      // keep passes dependent so the compiler cannot merge them
      xs[0][(p + tid) & 1023] += acc[0].x * 1e-20f;
    }
  }
  float s = 0; for (int r = 0; r < R; ++r) s += acc[r].x + acc[r].y + acc[r].z + acc[r].w;
  out[blockIdx.x * NT + tid] = s;
}

int main() {
  const int K = 768, N = 256;      // 768 KB, the post-net BiGRU recurrent weights of one direction
  float *W, *out; CK(hipMalloc(&W, (size_t)8 * K * N * 4)); CK(hipMalloc(&out, 1 << 22)); CK(hipMemset(W, 0, (size_t)8 * K * N * 4));
  hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int passes = 400; float ms;
  for (int G : {1, 8, 16, 32, 64}) {
#define RUN(R, NT, KK, NN, label) { \
      hipLaunchKernelGGL((k_stream<R, NT>), dim3(G), dim3(NT), 0, s, W, KK, NN, out, 10); \
      CK(hipEventRecord(e0, s)); hipLaunchKernelGGL((k_stream<R, NT>), dim3(G), dim3(NT), 0, s, W, KK, NN, out, passes); \
      CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1)); \
      printf("G=%2d WGs  %-34s: %6.2f us/pass  %6.1f GB/s per WG\n", G, label, ms * 1e3 / passes, (double)KK * NN * 4 / (ms * 1e-3 / passes) / 1e9); }
    RUN(1, 1024, 768, 256, "768KB  R=1 rows, 1024 thr");
    RUN(4, 1024, 768, 256, "768KB  R=4 rows, 1024 thr");
    RUN(4, 512, 768, 256, "768KB  R=4 rows,  512 thr");
    RUN(2, 1024, 1024, 1024, "4MB    R=2 rows, 1024 thr");
    RUN(1, 1024, 1536, 1024, "6MB    R=1 rows, 1024 thr");
  }
  return 0;
}
