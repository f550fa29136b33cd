// taco_train_kernels.h -- training-side kernels that need no backward pass:
//   k_l1_partial / k_loss_final   add_loss (tacotron.py:274-302): coefficient-weighted L1 means
//   k_sumsq_partial + k_adam      clip_by_global_norm(1.0) + tf.train.AdamOptimizer + LR schedule (tacotron.py:305-336)
// All streaming (HBM-bound): 16-byte loads, grid-stride, one partial per workgroup, deterministic final pass.
#pragma once
#include <hip/hip_runtime.h>

#define TR_NT 256
#define TR_MAXBLK 1024

__device__ __forceinline__ double tr_block_sum(double x, double* sm) {
  const int tid = threadIdx.x;
  sm[tid] = x;
  __syncthreads();
  for (int o = TR_NT / 2; o > 0; o >>= 1) { if (tid < o) sm[tid] += sm[tid + o]; __syncthreads(); }
  return sm[0];
}

// partial[blk*4 + {0,1,2}] = sum |a-b|, sum |a-b|*coeff[row/T], sum over the priority band of |a-b|*coeff; [3] = band plain
// a, b: [B*T, C]; band [c_lo, c_hi) (empty when c_lo >= c_hi)
__global__ __launch_bounds__(TR_NT) void k_l1_partial(const float* a, const float* b, const float* coeff, int rows, int T, int C,
                                                     int c_lo, int c_hi, double* partial) {
  __shared__ double sm[TR_NT];
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  const size_t total = (size_t)rows * C;
  for (size_t i = (size_t)blockIdx.x * TR_NT + threadIdx.x; i < total; i += (size_t)gridDim.x * TR_NT) {
    const int row = (int)(i / C), c = (int)(i % C);
    const float d = fabsf(a[i] - b[i]);
    const float w = coeff ? coeff[row / T] : 1.f;
    s0 += d; s1 += (double)d * w;
    if (c >= c_lo && c < c_hi) { s2 += (double)d * w; s3 += d; }
  }
  const double r0 = tr_block_sum(s0, sm); __syncthreads();
  const double r1 = tr_block_sum(s1, sm); __syncthreads();
  const double r2 = tr_block_sum(s2, sm); __syncthreads();
  const double r3 = tr_block_sum(s3, sm);
  if (threadIdx.x == 0) { double* p = partial + (size_t)blockIdx.x * 4; p[0] = r0; p[1] = r1; p[2] = r2; p[3] = r3; }
}

// losses[0..3] = loss, mel_loss, linear_loss, loss_without_coeff
// (one workgroup of TR_NT threads: strided partial sums, then a fixed tree -- deterministic; a single thread walking the 2 x 1024 partials
// one dependent load at a time was 0.2 ms on the step's critical path)
__global__ __launch_bounds__(TR_NT) void k_loss_final(const double* pm, int nbm, const double* pl, int nbl, double n_mel, double n_lin, double n_band,
                                                     int prioritize, float* losses) {
  __shared__ double sm[TR_NT];
  if (blockIdx.x != 0) return;
  double m0 = 0, m1 = 0, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
  for (int i = threadIdx.x; i < nbm; i += TR_NT) { m0 += pm[i * 4]; m1 += pm[i * 4 + 1]; }
  for (int i = threadIdx.x; i < nbl; i += TR_NT) { l0 += pl[i * 4]; l1 += pl[i * 4 + 1]; l2 += pl[i * 4 + 2]; l3 += pl[i * 4 + 3]; }
  m0 = tr_block_sum(m0, sm); __syncthreads(); m1 = tr_block_sum(m1, sm); __syncthreads();
  l0 = tr_block_sum(l0, sm); __syncthreads(); l1 = tr_block_sum(l1, sm); __syncthreads();
  l2 = tr_block_sum(l2, sm); __syncthreads(); l3 = tr_block_sum(l3, sm);
  if (threadIdx.x != 0) return;
  const double mel_loss = m0 / n_mel;
  double loss, lin_loss;
  if (prioritize) {
    loss = m1 / n_mel + 0.5 * l1 / n_lin + 0.5 * l2 / n_band;
    lin_loss = 0.5 * (l0 / n_lin + l3 / n_band);
  } else {
    loss = m1 / n_mel + l1 / n_lin;
    lin_loss = l0 / n_lin;
  }
  losses[0] = (float)loss; losses[1] = (float)mel_loss; losses[2] = (float)lin_loss; losses[3] = (float)(mel_loss + lin_loss);
}

// see taco_train_forward_backward: the sticky device error word -> NaN losses and one NaN gradient element
__global__ void k_train_latch(const unsigned* err, float* losses, float* grads) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && err && err[0] != 0u) {
    const float nan = __uint_as_float(0x7FC00000u);
    if (losses) { losses[0] = nan; losses[1] = nan; losses[2] = nan; losses[3] = nan; }
    if (grads) grads[0] = nan;
  }
}

__global__ __launch_bounds__(TR_NT) void k_sumsq_partial(const float* g, size_t n, double* partial) {
  __shared__ double sm[TR_NT];
  double s = 0;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * TR_NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * TR_NT) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; s += (double)v * v; }
  const double r = tr_block_sum(s, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// every workgroup re-reduces the (<= TR_MAXBLK) partials -> global norm; then the fused clip + Adam update.
// TF form (A.14): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g^2; p -= lr_t*m/(sqrt(v)+eps)
__global__ __launch_bounds__(TR_NT) void k_adam(float* p, const float* g, float* m, float* v, size_t n, const double* partial,
                                               int nblk, float lr_t, float b1, float b2, float eps, float clip, float* gnorm_out) {
  __shared__ double sm[TR_NT];
  double s = 0;
  for (int i = threadIdx.x; i < nblk; i += TR_NT) s += partial[i];
  const double gn = sqrt(tr_block_sum(s, sm));
  const float scale = (float)((double)clip / fmax(gn, (double)clip));      // tf.clip_by_global_norm
  if (blockIdx.x == 0 && threadIdx.x == 0 && gnorm_out) *gnorm_out = (float)gn;
  // a non-finite global norm (a gradient poisoned by k_train_latch after a device fault, or a genuine overflow) skips the update:
  // parameters and moments stay as they are.  (tf.clip_by_global_norm would turn every parameter into NaN here; nothing can be
  // learned from such a step either way, and a data-parallel job stays consistent because all ranks see the same reduced norm.)
  if (!(gn < 1.7976931348623157e308)) return;
  for (size_t i = (size_t)blockIdx.x * TR_NT + threadIdx.x; i < n; i += (size_t)gridDim.x * TR_NT) {
    const float gi = g[i] * scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}
