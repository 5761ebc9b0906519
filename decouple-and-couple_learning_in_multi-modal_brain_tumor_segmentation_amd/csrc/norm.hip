// K3 -- InstanceNorm3d (affine=False) pieces that are not fused into the conv kernels, and the
// elementwise tails of the residual blocks.  All HBM-bound: 16 B per lane, channels-last.
// Reference: nn.InstanceNorm3d + ReLU/LeakyReLU at Unet_skipconnection.py:39-56, cls_wise_former.py:207-223,
// 697-711,737-752 and their autograd.
#include "common.h"

__global__ void in_finalize_kernel(const double* __restrict__ stats, float* __restrict__ scale, float* __restrict__ shift,
                                   int NC, double invV, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NC) return;
  const double mean = stats[2 * i] * invV;
  double var = stats[2 * i + 1] * invV - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  scale[i] = (float)rstd;
  shift[i] = (float)(-mean * rstd);
}

// Generic per-(n,c) double-sum reduction over voxels.  MODE 0: (x, x^2);  MODE 1: g = dy*act'(xhat), (g, g*xhat).
template <int MODE>
__global__ __launch_bounds__(256) void in_reduce_kernel(const float* __restrict__ dy, int dy_ldc, const float* __restrict__ x, int x_ldc,
                                                       const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                                       double* __restrict__ sums, int64_t V, int C, int vox_per_block) {
  __shared__ float red[256 * 8];
  const int CQ = C >> 2;
  const int nvs = 256 / CQ;
  const int t = threadIdx.x;
  const int cq = t % CQ, vs = t / CQ;
  const int n = blockIdx.y;
  const int64_t v0 = (int64_t)blockIdx.x * vox_per_block;
  const int64_t v1 = min(V, v0 + vox_per_block);
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
  if (vs < nvs) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 1) {
      sc = *reinterpret_cast<const float4*>(scale + (int64_t)n * C + cq * 4);
      sh = *reinterpret_cast<const float4*>(shift + (int64_t)n * C + cq * 4);
    }
    for (int64_t v = v0 + vs; v < v1; v += nvs) {
      const float4 xv = *reinterpret_cast<const float4*>(x + ((int64_t)n * V + v) * x_ldc + cq * 4);
      if (MODE == 0) {
        a1[0] += xv.x; a1[1] += xv.y; a1[2] += xv.z; a1[3] += xv.w;
        a2[0] += xv.x * xv.x; a2[1] += xv.y * xv.y; a2[2] += xv.z * xv.z; a2[3] += xv.w * xv.w;
      } else {
        const float4 gv = *reinterpret_cast<const float4*>(dy + ((int64_t)n * V + v) * dy_ldc + cq * 4);
        const float h0 = xv.x * sc.x + sh.x, h1 = xv.y * sc.y + sh.y, h2 = xv.z * sc.z + sh.z, h3 = xv.w * sc.w + sh.w;
        const float g0 = gv.x * cwf_act_grad(h0, slope), g1 = gv.y * cwf_act_grad(h1, slope);
        const float g2 = gv.z * cwf_act_grad(h2, slope), g3 = gv.w * cwf_act_grad(h3, slope);
        a1[0] += g0; a1[1] += g1; a1[2] += g2; a1[3] += g3;
        a2[0] += g0 * h0; a2[1] += g1 * h1; a2[2] += g2 * h2; a2[3] += g3 * h3;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[t * 8 + i] = a1[i]; red[t * 8 + 4 + i] = a2[i]; }
  __syncthreads();
  // thread (cq, which 0..7) folds the nvs voxel sub-lanes
  for (int idx = t; idx < CQ * 8; idx += 256) {
    const int q = idx >> 3, w = idx & 7;
    double s = 0.0;
    for (int k = 0; k < nvs; ++k) s += (double)red[(k * CQ + q) * 8 + w];
    const int c = q * 4 + (w & 3);
    atomic_add_f64(sums + ((int64_t)n * C + c) * 2 + (w >> 2), s);
  }
}

// y = act(x*scale+shift) + residual
__global__ void norm_act_add_kernel(const float* __restrict__ x, int x_ldc, const float* __restrict__ scale, const float* __restrict__ shift,
                                    float slope, const float* __restrict__ residual, int r_ldc, float* __restrict__ y, int y_ldc,
                                    int64_t V, int C, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int CQ = C >> 2;
  const int cq = (int)(idx % CQ);
  const int64_t gv = idx / CQ;         // n*V + v
  const int64_t n = gv / V;
  const float4 sc = *reinterpret_cast<const float4*>(scale + n * C + cq * 4);
  const float4 sh = *reinterpret_cast<const float4*>(shift + n * C + cq * 4);
  const float4 xv = *reinterpret_cast<const float4*>(x + gv * x_ldc + cq * 4);
  float4 o;
  o.x = cwf_act(xv.x * sc.x + sh.x, slope); o.y = cwf_act(xv.y * sc.y + sh.y, slope);
  o.z = cwf_act(xv.z * sc.z + sh.z, slope); o.w = cwf_act(xv.w * sc.w + sh.w, slope);
  if (residual) {
    const float4 rv = *reinterpret_cast<const float4*>(residual + gv * r_ldc + cq * 4);
    o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
  }
  *reinterpret_cast<float4*>(y + gv * y_ldc + cq * 4) = o;
}

// dx = scale*(g - S1/V - xhat*S2/V) (+ dx_add)
__global__ void in_bwd_apply_kernel(const float* __restrict__ dy, int dy_ldc, const float* __restrict__ x, int x_ldc,
                                    const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                    const double* __restrict__ sums, const float* __restrict__ dx_add, int a_ldc,
                                    float* __restrict__ dx, int dx_ldc, int64_t V, int C, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int CQ = C >> 2;
  const int cq = (int)(idx % CQ);
  const int64_t gv = idx / CQ;
  const int64_t n = gv / V;
  const float4 sc = *reinterpret_cast<const float4*>(scale + n * C + cq * 4);
  const float4 sh = *reinterpret_cast<const float4*>(shift + n * C + cq * 4);
  const float4 xv = *reinterpret_cast<const float4*>(x + gv * x_ldc + cq * 4);
  const float4 gv4 = *reinterpret_cast<const float4*>(dy + gv * dy_ldc + cq * 4);
  const double* sp = sums + (n * C + cq * 4) * 2;
  const float invV = 1.0f / (float)V;
  const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
  const float scs[4] = {sc.x, sc.y, sc.z, sc.w}, shs[4] = {sh.x, sh.y, sh.z, sh.w};
  float o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float h = xs[i] * scs[i] + shs[i];
    const float g = gs[i] * cwf_act_grad(h, slope);
    const float m1 = (float)(sp[2 * i] * (double)invV), m2 = (float)(sp[2 * i + 1] * (double)invV);
    o[i] = scs[i] * (g - m1 - h * m2);
  }
  if (dx_add) {
    const float4 av = *reinterpret_cast<const float4*>(dx_add + gv * a_ldc + cq * 4);
    o[0] += av.x; o[1] += av.y; o[2] += av.z; o[3] += av.w;
  }
  *reinterpret_cast<float4*>(dx + gv * dx_ldc + cq * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

static int check_cl(const void* p, int ldc, int C) {
  if (!p) return CWF_E_BADARG;
  if ((C & 3) || (ldc & 3) || ldc < C || ((uintptr_t)p & 15)) return CWF_E_ALIGN;
  if (C > 1024) return CWF_E_TOOLARGE;
  return 0;
}

extern "C" int cwf_in_finalize(const double* stats, float* scale, float* shift, int NC, int64_t V, float eps, void* stream) {
  if (!stats || !scale || !shift || NC <= 0 || V <= 0) return CWF_E_BADARG;
  hipLaunchKernelGGL(in_finalize_kernel, dim3(cdiv(NC, 256)), dim3(256), 0, cwf_stream(stream), stats, scale, shift, NC, 1.0 / (double)V, eps);
  CWF_LAUNCH_CHECK();
  return 0;
}

static int launch_reduce(int mode, const float* dy, int dy_ldc, const float* x, int x_ldc, const float* scale, const float* shift,
                         float slope, double* sums, int N, int64_t V, int C, void* stream) {
  int rc = check_cl(x, x_ldc, C); if (rc) return rc;
  if (!sums || N <= 0 || V <= 0) return CWF_E_BADARG;
  // aim for ~2048 workgroups, >= 256 voxels each
  int64_t vpb = cdiv64(V * N, 2048); if (vpb < 256) vpb = 256; if (vpb > V) vpb = V;
  dim3 grid((unsigned)cdiv64(V, vpb), N);
  if (mode == 0) hipLaunchKernelGGL(in_reduce_kernel<0>, grid, dim3(256), 0, cwf_stream(stream), dy, dy_ldc, x, x_ldc, scale, shift, slope, sums, V, C, (int)vpb);
  else hipLaunchKernelGGL(in_reduce_kernel<1>, grid, dim3(256), 0, cwf_stream(stream), dy, dy_ldc, x, x_ldc, scale, shift, slope, sums, V, C, (int)vpb);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_in_stats(const float* x, int x_ldc, double* stats, int N, int64_t V, int C, void* stream) {
  return launch_reduce(0, nullptr, 0, x, x_ldc, nullptr, nullptr, 1.f, stats, N, V, C, stream);
}

extern "C" int cwf_in_bwd_stats(const float* dy, int dy_ldc, const float* x, int x_ldc, const float* scale, const float* shift,
                                float slope, double* sums, int N, int64_t V, int C, void* stream) {
  int rc = check_cl(dy, dy_ldc, C); if (rc) return rc;
  if (!scale || !shift) return CWF_E_BADARG;
  return launch_reduce(1, dy, dy_ldc, x, x_ldc, scale, shift, slope, sums, N, V, C, stream);
}

extern "C" int cwf_norm_act_add(const float* x, int x_ldc, const float* scale, const float* shift, float slope,
                                const float* residual, int r_ldc, float* y, int y_ldc, int N, int64_t V, int C, void* stream) {
  int rc = check_cl(x, x_ldc, C); if (rc) return rc;
  rc = check_cl(y, y_ldc, C); if (rc) return rc;
  if (residual) { rc = check_cl(residual, r_ldc, C); if (rc) return rc; }
  if (!scale || !shift) return CWF_E_BADARG;
  const int64_t total = (int64_t)N * V * (C >> 2);
  hipLaunchKernelGGL(norm_act_add_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream),
                     x, x_ldc, scale, shift, slope, residual, r_ldc, y, y_ldc, V, C, total);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_in_bwd_apply(const float* dy, int dy_ldc, const float* x, int x_ldc, const float* scale, const float* shift,
                                float slope, const double* sums, const float* dx_add, int a_ldc, float* dx, int dx_ldc,
                                int N, int64_t V, int C, void* stream) {
  int rc = check_cl(x, x_ldc, C); if (rc) return rc;
  rc = check_cl(dy, dy_ldc, C); if (rc) return rc;
  rc = check_cl(dx, dx_ldc, C); if (rc) return rc;
  if (dx_add) { rc = check_cl(dx_add, a_ldc, C); if (rc) return rc; }
  if (!scale || !shift || !sums) return CWF_E_BADARG;
  const int64_t total = (int64_t)N * V * (C >> 2);
  hipLaunchKernelGGL(in_bwd_apply_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream),
                     dy, dy_ldc, x, x_ldc, scale, shift, slope, sums, dx_add, a_ldc, dx, dx_ldc, V, C, total);
  CWF_LAUNCH_CHECK();
  return 0;
}
