// ubench_wfetch.hip -- period of a dependent hipGraph node as a function of how many "weight" bytes it
// must fetch: NB workgroups x 512 threads each read `kb` KB (16-byte loads), reduce through LDS, write one
// float that the next node reads.  Weight buffers cycle through a 6 MB set (like the 10 decoder stages).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(512) void k_w(const float4* __restrict__ w, int n4_per_block, const float* dep, float* out) {
  __shared__ float red[512];
  const float4* p = w + (size_t)blockIdx.x * n4_per_block;
  float s = dep[0];
  for (int i0 = threadIdx.x; i0 < n4_per_block; i0 += 4 * 512) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { int i = i0 + u * 512; v[u] = (i < n4_per_block) ? p[i] : make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  red[threadIdx.x] = s; __syncthreads();
  if (threadIdx.x == 0) { float t = 0; for (int i = 0; i < 512; i += 64) t += red[i]; out[blockIdx.x] = t; }
}
int main() {
  float4* w; float *a, *b; CK(hipMalloc(&w, 64 << 20)); CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096));
  CK(hipMemset(w, 0, 64 << 20)); CK(hipMemset(a, 0, 4096)); CK(hipMemset(b, 0, 4096));
  hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 1200; float ms;
  for (int nb : {16, 48}) for (int kb : {0, 4, 16, 32, 64}) for (int cyc : {1, 12}) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
      const size_t off4 = (size_t)(i % cyc) * (512 << 10) / 16;     // 512 KB apart: 12 buffers = 6 MB
      hipLaunchKernelGGL(k_w, dim3(nb), dim3(512), 0, s, w + off4, kb * 1024 / 16, (i & 1) ? a : b, (i & 1) ? b : a);
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    printf("blocks=%2d  %3d KB/block (%5d KB/launch)  %2d weight buffers: %.2f us per node\n", nb, kb, nb * kb, cyc, ms * 1e3 / N);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
