// ubench_sync.hip -- measures the synchronisation primitives the decode loop can be built from, on the
// box the bench runs on: dependent-launch period (eager / hipGraph) and an in-kernel barrier among G
// co-resident workgroups (monotonic counter, agent-scope release/acquire, bounded spin).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_sync ubench_sync.hip ; run: ./ubench_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty(float* p) { if (p && threadIdx.x == 1024) p[0] = 1.f; }
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }

struct BigArgs { int njobs; int pad; struct J { const float* p[13]; int v[26]; } j[4]; };
// mimics k_skinny's argument handling: job select by blockIdx, then dependent field reads
__global__ __launch_bounds__(512) void k_bigargs(const BigArgs a) {
  __shared__ float red[4096];
  int ji = 0;
  for (int q = 1; q < 4; ++q) if (q < a.njobs && (int)blockIdx.x >= a.j[q].v[25]) ji = q;
  const BigArgs::J& jb = a.j[ji];
  const float* x = jb.p[0]; float* o = (float*)jb.p[1];
  float v = x[(blockIdx.x * 512 + threadIdx.x) % jb.v[0]];
  red[threadIdx.x] = v; __syncthreads();
  float s = 0; for (int w = 0; w < 8; ++w) s += red[(threadIdx.x & 63) + 64 * w];
  o[blockIdx.x * 512 + threadIdx.x] = s + jb.v[1];
}
// same work, arguments through ONE pointer to a device-resident descriptor
__global__ __launch_bounds__(512) void k_ptrargs(const BigArgs* ap) {
  __shared__ float red[4096];
  const BigArgs& a = *ap;
  int ji = 0;
  for (int q = 1; q < 4; ++q) if (q < a.njobs && (int)blockIdx.x >= a.j[q].v[25]) ji = q;
  const BigArgs::J& jb = a.j[ji];
  const float* x = jb.p[0]; float* o = (float*)jb.p[1];
  float v = x[(blockIdx.x * 512 + threadIdx.x) % jb.v[0]];
  red[threadIdx.x] = v; __syncthreads();
  float s = 0; for (int w = 0; w < 8; ++w) s += red[(threadIdx.x & 63) + 64 * w];
  o[blockIdx.x * 512 + threadIdx.x] = s + jb.v[1];
}

// barrier among `nb` workgroups; returns false on timeout
__device__ __forceinline__ bool group_barrier(unsigned* ctr, unsigned target, unsigned* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { *err = 1; ok = false; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

__global__ void k_barrier_loop(unsigned* ctr, unsigned* err, float* data, int iters, int payload) {
  const unsigned nb = gridDim.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // each block publishes `payload` floats, then everyone reads everyone's (an all-gather)
    for (int i = threadIdx.x; i < payload; i += blockDim.x) data[(size_t)blockIdx.x * payload + i] = (float)(it + i);
    if (!group_barrier(ctr, (unsigned)(it + 1) * nb, err)) return;
    for (int i = threadIdx.x; i < payload * (int)nb; i += blockDim.x) acc += data[i];
    // second barrier so nobody overwrites before all have read (WAR)
    // (kept out: we alternate two buffers instead)
    data += (it & 1) ? -(ptrdiff_t)((size_t)nb * payload) : (ptrdiff_t)((size_t)nb * payload);
  }
  if (acc == 12345.678f) err[1] = 1;
}

int main() {
  float* d; unsigned* ctr; unsigned* err;
  CK(hipMalloc(&d, 64 << 20)); CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&err, 256));
  CK(hipMemset(d, 0, 64 << 20));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  const int N = 2000;
  for (int grid : {1, 16, 256}) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s, d);
      CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("eager  empty kernel grid=%3d : %.2f us per launch\n", grid, ms * 1e3 / N);
  }
  {  // kernels that really depend on each other through memory
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_touch, dim3(16), dim3(256), 0, s, d, 4096);
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("eager  touch kernel grid= 16 : %.2f us per launch\n", ms * 1e3 / N);
  }
  {  // graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_touch, dim3(16), dim3(256), 0, s, d, 4096);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("graph  touch kernel grid= 16 : %.2f us per node (%d nodes)\n", ms * 1e3 / N, N);
  }
  {  // big by-value kernarg struct vs pointer to device descriptor, in a graph
    BigArgs h; memset(&h, 0, sizeof h); h.njobs = 2;
    for (int q = 0; q < 4; ++q) { h.j[q].p[0] = d; h.j[q].p[1] = d + (1 << 20); h.j[q].v[0] = 4096; h.j[q].v[25] = 8 * q; }
    BigArgs* dargs; CK(hipMalloc(&dargs, sizeof h)); CK(hipMemcpy(dargs, &h, sizeof h, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 2; ++mode) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < N; ++i) {
        if (mode == 0) hipLaunchKernelGGL(k_bigargs, dim3(16), dim3(512), 0, s, h);
        else hipLaunchKernelGGL(k_ptrargs, dim3(16), dim3(512), 0, s, dargs);
      }
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf("graph  %s kernel grid=16x512 : %.2f us per node (sizeof args %zu)\n", mode ? "ptr-args " : "big-kernarg", ms * 1e3 / N, sizeof h);
    }
  }
  for (int nb : {2, 8, 16}) {
    for (int payload : {64, 1024}) {
      const int iters = 2000;
      CK(hipMemsetAsync(ctr, 0, 256, s)); CK(hipMemsetAsync(err, 0, 256, s));
      hipLaunchKernelGGL(k_barrier_loop, dim3(nb), dim3(256), 0, s, ctr, err, d, 10, payload);  // warm
      CK(hipMemsetAsync(ctr, 0, 256, s));
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_barrier_loop, dim3(nb), dim3(256), 0, s, ctr, err, d, iters, payload);
      CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned herr[2]; CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
      printf("in-kernel all-gather+barrier  G=%3d WGs payload=%4d floats/WG : %.2f us per exchange%s\n", nb, payload,
             ms * 1e3 / iters, herr[0] ? "  (TIMEOUT)" : "");
    }
  }
  return 0;
}
