// ubench_xchg.hip -- where does an in-launch all-gather among P workgroups spend its time?
// Replica of k_bigru_persist's exchange: every WG publishes `slice` floats (sc1 16-byte stores), raises a
// flag; waits for the P flags; gathers P*slice floats (sc1 16-byte loads) into LDS.  Block 0 records
// wall_clock64() (100 MHz) at: A after publish-store drain, B after flag store, C after all flags seen,
// D after the gather landed in LDS.  Variants: polling with/without s_sleep, plain vs sc1 gather.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: sc1 gather + sleep poll, 1: sc1 gather + tight poll, 2: plain gather after acquire fence, 3: flags only (no payload)
__global__ __launch_bounds__(512) void k_x(float* xbuf, unsigned* flags, unsigned* err, long long* ts, int P, int slice, int iters) {
  extern __shared__ float lds[];
  __shared__ int ok_s;
  const int tid = threadIdx.x, p = blockIdx.x;
  const unsigned total = (unsigned)P * slice;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(xbuf, 0, 2u * total * 4u, 0x00020000);
  for (int it = 0; it < iters; ++it) {
    const unsigned base = (it & 1) * total;
    long long tA = 0, tB = 0, tC = 0, tD = 0, t0 = wall_clock64();
    if (MODE != 3)
      for (int i = tid; i < slice / 4; i += 512) {
        v4u32 u = {(unsigned)it, (unsigned)i, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(u, r, (base + p * slice + 4 * i) * 4u, 0, 16);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    tA = wall_clock64();
    if (tid == 0) __hip_atomic_store(flags + p, (unsigned)it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tB = wall_clock64();
    if (tid < 64) {
      bool ok = true;
      if (tid < P) {
        unsigned spins = 0;
        while (__hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it + 1) {
          if (MODE != 1) __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 22)) { ok = false; break; }
        }
      }
      ok = __all(ok);
      if (tid == 0) { if (!ok) *err = 1; if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); ok_s = ok; }
    }
    __syncthreads();
    if (!ok_s) return;
    tC = wall_clock64();
    if (MODE != 3) {
      for (int i0 = tid; i0 < (int)total / 4; i0 += 4 * 512) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int i = i0 + u * 512; if (i >= (int)total / 4) i = i0;
          if (MODE == 2) v[u] = reinterpret_cast<const float4*>(xbuf + base)[i];
          else { v4u32 w = __builtin_amdgcn_raw_buffer_load_b128(r, (base + 4 * i) * 4u, 0, 16); v[u] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), 0, 0); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { int i = i0 + u * 512; if (i < (int)total / 4) reinterpret_cast<float4*>(lds)[i] = v[u]; }
      }
    }
    __syncthreads();
    tD = wall_clock64();
    if (p == 0 && tid == 0 && it >= iters - 64) { long long* o = ts + (it - (iters - 64)) * 5; o[0] = t0; o[1] = tA; o[2] = tB; o[3] = tC; o[4] = tD; }
  }
}

int main() {
  float* xbuf; unsigned* flags; unsigned* err; long long* ts;
  CK(hipMalloc(&xbuf, 8 << 20)); CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&err, 256)); CK(hipMalloc(&ts, 64 * 5 * 8));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int P : {2, 4, 8}) for (int slice : {256, 1024}) for (int mode = 0; mode < 4; ++mode) {
    CK(hipMemsetAsync(flags, 0, 4096, s)); CK(hipMemsetAsync(err, 0, 256, s));
    size_t lds = (size_t)P * slice * 4;
    auto launch = [&](int it) {
      switch (mode) {
        case 0: hipFuncSetAttribute((const void*)&k_x<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k_x<0>, dim3(P), dim3(512), lds, s, xbuf, flags, err, ts, P, slice, it); break;
        case 1: hipFuncSetAttribute((const void*)&k_x<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k_x<1>, dim3(P), dim3(512), lds, s, xbuf, flags, err, ts, P, slice, it); break;
        case 2: hipFuncSetAttribute((const void*)&k_x<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k_x<2>, dim3(P), dim3(512), lds, s, xbuf, flags, err, ts, P, slice, it); break;
        default: hipFuncSetAttribute((const void*)&k_x<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k_x<3>, dim3(P), dim3(512), lds, s, xbuf, flags, err, ts, P, slice, it); break;
      }
    };
    if (lds > 160 * 1024) continue;
    CK(hipEventRecord(e0, s)); launch(iters); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h[64 * 5]; CK(hipMemcpy(h, ts, sizeof h, hipMemcpyDeviceToHost));
    unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    double a = 0, b = 0, c = 0, d = 0;
    for (int i = 0; i < 64; ++i) { a += h[i*5+1]-h[i*5]; b += h[i*5+2]-h[i*5+1]; c += h[i*5+3]-h[i*5+2]; d += h[i*5+4]-h[i*5+3]; }
    const char* mn[4] = {"sc1 gather, sleep poll", "sc1 gather, tight poll", "acq fence + plain gather", "flags only"};
    printf("P=%2d slice=%4d floats  %-26s: %.2f us/exchange | store+drain %.2f  flag %.2f  wait %.2f  gather %.2f us%s\n", P, slice, mn[mode],
           ms * 1e3 / iters, a / 64 / 100.0, b / 64 / 100.0, c / 64 / 100.0, d / 64 / 100.0, herr ? " TIMEOUT" : "");
  }
  return 0;
}
