// K11/K12 + misc -- fused multi-tensor Adam(amsgrad), the batched weight-pack gather, and trivial elementwise
// helpers.  Reference: torch.optim.Adam(lr=2e-4, weight_decay=1e-5, amsgrad=True) at train_no_amp.py:136,239
// (L2 decay added to the gradient, max of second moments, bias corrections as in torch/optim/adam.py
// _single_tensor_adam).  All HBM-bound.
#include "common.h"

__global__ void gather_batched_kernel(const cwf_gather_desc* __restrict__ table) {
  const cwf_gather_desc d = table[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t m = d.map[i];
    if (m == -2) continue;                       // owned by another source of a fused layer (three convs sharing one input)
    d.dst[i] = m >= 0 ? d.src[m] : 0.f;
  }
}

extern "C" int cwf_gather_batched(const struct cwf_gather_desc* table, int nlayers, int64_t max_n, void* stream) {
  if (!table || nlayers <= 0 || max_n <= 0) return CWF_E_BADARG;
  int64_t gx = cdiv64(max_n, 256); if (gx > 256) gx = 256;
  hipLaunchKernelGGL(gather_batched_kernel, dim3((unsigned)gx, nlayers), dim3(256), 0, cwf_stream(stream), table);
  CWF_LAUNCH_CHECK();
  return 0;
}

__global__ void adam_kernel(const cwf_adam_desc* __restrict__ table, const float* __restrict__ hyper, float step_size, float omb1, float beta2,
                            float omb2, float eps, float wd, float bc2_sqrt, int amsgrad, float gscale) {
  const cwf_adam_desc d = table[blockIdx.y];
  if (hyper) { step_size = hyper[0]; bc2_sqrt = hyper[1]; }        // device-resident: survives hipGraph replay
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * blockDim.x) {
    const float p = d.p[i];
    const float g = fmaf(wd, p, d.g[i] * gscale);                      // (gscale = 1/world: the gradient AVERAGE over ranks) grad.add(param, alpha=weight_decay)
    const float m = d.m[i] + omb1 * (g - d.m[i]);                    // exp_avg.lerp_(grad, 1 - beta1)
    const float v = beta2 * d.v[i] + omb2 * g * g;                   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    d.m[i] = m; d.v[i] = v;
    float vv = v;
    if (amsgrad) { vv = fmaxf(d.vmax[i], v); d.vmax[i] = vv; }
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    d.p[i] = p - step_size * (m / denom);                            // param.addcdiv_(exp_avg, denom, value=-step_size)
  }
}

extern "C" int cwf_adam_amsgrad(const struct cwf_adam_desc* table, int ntensors, int64_t max_n,
                                double lr, double beta1, double beta2, double eps, double weight_decay, int step, int amsgrad,
                                const float* hyper_dev, void* stream) {
  return cwf_adam_amsgrad_scaled(table, ntensors, max_n, lr, beta1, beta2, eps, weight_decay, step, amsgrad, hyper_dev, 1.0f, stream);
}

extern "C" int cwf_adam_amsgrad_scaled(const struct cwf_adam_desc* table, int ntensors, int64_t max_n,
                                       double lr, double beta1, double beta2, double eps, double weight_decay, int step, int amsgrad,
                                       const float* hyper_dev, float grad_scale, void* stream) {
  if (!table || ntensors <= 0 || max_n <= 0 || (step <= 0 && !hyper_dev)) return CWF_E_BADARG;
  const double bc1 = 1.0 - pow(beta1, (double)(step > 0 ? step : 1));
  const double bc2 = 1.0 - pow(beta2, (double)(step > 0 ? step : 1));
  int64_t gx = cdiv64(max_n, 256); if (gx > 64) gx = 64;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)gx, ntensors), dim3(256), 0, cwf_stream(stream), table, hyper_dev, (float)(lr / bc1),
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)sqrt(bc2), amsgrad, grad_scale);
  CWF_LAUNCH_CHECK();
  return 0;
}

__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a[i] * b[i];
}
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a[i] + b[i];
}
__global__ void channel_scale_kernel(const float* __restrict__ x, int x_ldc, const float* __restrict__ s, float* __restrict__ y, int y_ldc,
                                     int64_t V, int C, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int CQ = C >> 2; const int cq = (int)(idx % CQ); const int64_t gv = idx / CQ; const int64_t n = gv / V;
  const float4 sv = *reinterpret_cast<const float4*>(s + n * C + cq * 4);
  float4 xv = *reinterpret_cast<const float4*>(x + gv * x_ldc + cq * 4);
  xv.x *= sv.x; xv.y *= sv.y; xv.z *= sv.z; xv.w *= sv.w;
  *reinterpret_cast<float4*>(y + gv * y_ldc + cq * 4) = xv;
}
__global__ void copy_strided_kernel(const float* __restrict__ x, int x_ldc, float* __restrict__ y, int y_ldc, int C, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int CQ = C >> 2; const int cq = (int)(idx % CQ); const int64_t gv = idx / CQ;
  *reinterpret_cast<float4*>(y + gv * y_ldc + cq * 4) = *reinterpret_cast<const float4*>(x + gv * x_ldc + cq * 4);
}

// K12 -- pre-scaled dropout keep-mask in ONE launch (F.dropout / nn.Dropout sites of the token path, SelfAttention.py:96-100,
// ResidualNorm.py:25-31,40-45): mask[i] = keep_i / (1 - p), keep_i ~ Bernoulli(1 - p) from a counter-based generator
// (splitmix64 of seed and element counter -- reproducible for a given torch seed and call order, no generator state on the
// device).  p2 > 0 multiplies a second, independent mask in (two dropouts acting in sequence on one tensor).
__global__ void dropout_mask_kernel(float* __restrict__ m, int64_t n, float p, float p2, uint64_t seed, uint64_t offset) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = cwf_u01(seed, offset + (uint64_t)i) >= p ? 1.0f / (1.0f - p) : 0.f;
  if (p2 > 0.f) v *= cwf_u01(seed, offset + (uint64_t)n + (uint64_t)i) >= p2 ? 1.0f / (1.0f - p2) : 0.f;
  m[i] = v;
}
extern "C" int cwf_dropout_mask(float* mask, int64_t n, float p, float p2, uint64_t seed, uint64_t offset, void* stream) {
  if (!mask || n <= 0 || p < 0.f || p >= 1.f || p2 < 0.f || p2 >= 1.f) return CWF_E_BADARG;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, cwf_stream(stream), mask, n, p, p2, seed, offset);
  CWF_LAUNCH_CHECK();
  return 0;
}

// device-resident generator state {seed, step}: see common.h.  The advance is a kernel so that a captured training step draws
// fresh masks on every replay; the keep-mask kernel below is what remains of materialised masks (the stem's dropout3d channel
// scale, Unet_skipconnection.py:31).
__global__ void rng_advance_kernel(uint64_t* rng) { if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += 1; }
extern "C" int cwf_rng_advance(uint64_t* rng, void* stream) {
  if (!rng) return CWF_E_BADARG;
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, cwf_stream(stream), rng);
  CWF_LAUNCH_CHECK();
  return 0;
}
__global__ void dropout_mask_rng_kernel(float* __restrict__ m, int64_t n, float p, float p2, const uint64_t* __restrict__ rng, uint64_t off) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) m[i] = cwf_keep(rng, off, (uint64_t)i, (uint64_t)n, p, p2);
}
extern "C" int cwf_dropout_mask_rng(float* mask, int64_t n, float p, float p2, const uint64_t* rng, uint64_t offset, void* stream) {
  if (!mask || !rng || n <= 0 || p < 0.f || p >= 1.f || p2 < 0.f || p2 >= 1.f) return CWF_E_BADARG;
  hipLaunchKernelGGL(dropout_mask_rng_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, cwf_stream(stream), mask, n, p, p2, rng, offset);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_mul(const float* a, const float* b, float* y, int64_t n, void* stream) {
  if (!a || !b || !y || n <= 0) return CWF_E_BADARG;
  hipLaunchKernelGGL(mul_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, cwf_stream(stream), a, b, y, n);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_add(const float* a, const float* b, float* y, int64_t n, void* stream) {
  if (!a || !b || !y || n <= 0) return CWF_E_BADARG;
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, cwf_stream(stream), a, b, y, n);
  CWF_LAUNCH_CHECK();
  return 0;
}
__global__ void add3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (a[i] + b[i]) + c[i];
}
extern "C" int cwf_add3(const float* a, const float* b, const float* c, float* y, int64_t n, void* stream) {
  if (!a || !b || !c || !y || n <= 0) return CWF_E_BADARG;
  hipLaunchKernelGGL(add3_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, cwf_stream(stream), a, b, c, y, n);
  CWF_LAUNCH_CHECK();
  return 0;
}
__global__ void bcast3_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = x[i]; y[i] = v; y[n + i] = v; y[2 * n + i] = v; }
}
extern "C" int cwf_bcast3(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) return CWF_E_BADARG;
  hipLaunchKernelGGL(bcast3_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, cwf_stream(stream), x, y, n);
  CWF_LAUNCH_CHECK();
  return 0;
}
__global__ void stats_channel_sum_kernel(const double* __restrict__ stats, float* __restrict__ out, int N, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int n = 0; n < N; ++n) s += stats[((int64_t)n * C + c) * 2];
  out[c] = (float)s;
}
extern "C" int cwf_stats_channel_sum(const double* stats, float* out, int N, int C, void* stream) {
  if (!stats || !out || N <= 0 || C <= 0) return CWF_E_BADARG;
  hipLaunchKernelGGL(stats_channel_sum_kernel, dim3(cdiv(C, 64)), dim3(64), 0, cwf_stream(stream), stats, out, N, C);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_channel_scale(const float* x, int x_ldc, const float* s, float* y, int y_ldc, int N, int64_t V, int C, void* stream) {
  if (!x || !s || !y || N <= 0 || V <= 0 || (C & 3) || (x_ldc & 3) || (y_ldc & 3)) return CWF_E_BADARG;
  const int64_t total = (int64_t)N * V * (C >> 2);
  hipLaunchKernelGGL(channel_scale_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream), x, x_ldc, s, y, y_ldc, V, C, total);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_copy_strided(const float* x, int x_ldc, float* y, int y_ldc, int64_t nvox, int C, void* stream) {
  if (!x || !y || nvox <= 0 || (C & 3) || (x_ldc & 3) || (y_ldc & 3)) return CWF_E_BADARG;
  const int64_t total = nvox * (C >> 2);
  hipLaunchKernelGGL(copy_strided_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream), x, x_ldc, y, y_ldc, C, total);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_version(void) { return 2; }
extern "C" const char* cwf_arch(void) { return "gfx950"; }
