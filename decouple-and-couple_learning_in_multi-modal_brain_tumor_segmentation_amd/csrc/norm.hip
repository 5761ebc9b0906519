// K3 -- InstanceNorm3d (affine=False) pieces that are not fused into the conv kernels, and the
// elementwise tails of the residual blocks.  All HBM-bound: 16 B per lane, channels-last.
// Reference: nn.InstanceNorm3d + ReLU/LeakyReLU at Unet_skipconnection.py:39-56, cls_wise_former.py:207-223,
// 697-711,737-752 and their autograd.
#include "common.h"

__device__ __forceinline__ unsigned norm_pk_bf16(float a, float b) {          // one v_cvt_pk_bf16_f32 (round to nearest even)
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2));
}

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
                                    unsigned short* __restrict__ y16, int64_t V, int C, int64_t total) {
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
  if (y16) *reinterpret_cast<uint2*>(y16 + gv * C + cq * 4) = make_uint2(norm_pk_bf16(o.x, o.y), norm_pk_bf16(o.z, o.w));
}

// dx = scale*(g - S1/V - xhat*S2/V) (+ dx_add), g = dy*act'(xhat).
// Optional bf16 side outputs (the operands of the 16-channel weight-gradient / data-gradient kernels that take bf16 tensors,
// cwf_wgrad16_bf16): DX16 -- dx rounded to bf16 (the gradient the producing layer's kernels consume; they round it to bf16
// themselves otherwise); XA16 -- bf16(act(xhat)), the activated input of THIS layer, i.e. the x operand of its weight gradient
// (the weight-gradient kernels recompute it from x otherwise).  F32 = false skips the fp32 dx (no reader left).
template <bool F32, bool DX16, bool XA16>
__global__ void in_bwd_apply_kernel(const float* __restrict__ dy, int dy_ldc, const float* __restrict__ x, int x_ldc,
                                    const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                    const double* __restrict__ sums, const float* __restrict__ dx_add, int a_ldc,
                                    float* __restrict__ dx, int dx_ldc, unsigned short* __restrict__ dx16, unsigned short* __restrict__ xa16,
                                    int64_t V, int C, int64_t total) {
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
  float o[4], xa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float h = fmaf(xs[i], scs[i], shs[i]);
    const float g = gs[i] * cwf_act_grad(h, slope);
    const float m1 = (float)(sp[2 * i] * (double)invV), m2 = (float)(sp[2 * i + 1] * (double)invV);
    o[i] = scs[i] * (g - m1 - h * m2);
    xa[i] = fmaxf(h, h * slope);                       // slope in [0, 1]: the same form the weight-gradient loaders use
  }
  if (dx_add) {
    const float4 av = *reinterpret_cast<const float4*>(dx_add + gv * a_ldc + cq * 4);
    o[0] += av.x; o[1] += av.y; o[2] += av.z; o[3] += av.w;
  }
  if (F32) *reinterpret_cast<float4*>(dx + gv * dx_ldc + cq * 4) = make_float4(o[0], o[1], o[2], o[3]);
  if (DX16) *reinterpret_cast<uint2*>(dx16 + gv * C + cq * 4) = make_uint2(norm_pk_bf16(o[0], o[1]), norm_pk_bf16(o[2], o[3]));
  if (XA16) *reinterpret_cast<uint2*>(xa16 + gv * C + cq * 4) = make_uint2(norm_pk_bf16(xa[0], xa[1]), norm_pk_bf16(xa[2], xa[3]));
}

// y16 = bf16(act(x*scale+shift)) (scale == NULL: bf16(x)): the bf16 operand image of a tensor for the bf16-operand kernels where no
// producer emitted one (a stream: 4 B read + 2 B written per element)
__global__ void to_bf16_kernel(const float* __restrict__ x, int x_ldc, const float* __restrict__ scale, const float* __restrict__ shift,
                               float slope, unsigned short* __restrict__ y16, int64_t V, int C, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int CQ = C >> 2;
  const int cq = (int)(idx % CQ);
  const int64_t gv = idx / CQ;
  const int64_t n = gv / V;
  float4 v = *reinterpret_cast<const float4*>(x + gv * x_ldc + cq * 4);
  if (scale) {
    const float4 sc = *reinterpret_cast<const float4*>(scale + n * C + cq * 4);
    const float4 sh = *reinterpret_cast<const float4*>(shift + n * C + cq * 4);
    v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
    v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope); v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
  }
  *reinterpret_cast<uint2*>(y16 + gv * C + cq * 4) = make_uint2(norm_pk_bf16(v.x, v.y), norm_pk_bf16(v.z, v.w));
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

extern "C" int cwf_norm_act_add_ex(const float* x, int x_ldc, const float* scale, const float* shift, float slope,
                                   const float* residual, int r_ldc, float* y, int y_ldc, void* y16, int N, int64_t V, int C, void* stream);
extern "C" int cwf_norm_act_add(const float* x, int x_ldc, const float* scale, const float* shift, float slope,
                                const float* residual, int r_ldc, float* y, int y_ldc, int N, int64_t V, int C, void* stream) {
  return cwf_norm_act_add_ex(x, x_ldc, scale, shift, slope, residual, r_ldc, y, y_ldc, nullptr, N, V, C, stream);
}
extern "C" int cwf_norm_act_add_ex(const float* x, int x_ldc, const float* scale, const float* shift, float slope,
                                   const float* residual, int r_ldc, float* y, int y_ldc, void* y16, int N, int64_t V, int C, void* stream) {
  int rc = check_cl(x, x_ldc, C); if (rc) return rc;
  if (y16 && ((uintptr_t)y16 & 7)) return CWF_E_ALIGN;
  rc = check_cl(y, y_ldc, C); if (rc) return rc;
  if (residual) { rc = check_cl(residual, r_ldc, C); if (rc) return rc; }
  if (!scale || !shift) return CWF_E_BADARG;
  const int64_t total = (int64_t)N * V * (C >> 2);
  hipLaunchKernelGGL(norm_act_add_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream),
                     x, x_ldc, scale, shift, slope, residual, r_ldc, y, y_ldc, (unsigned short*)y16, V, C, total);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_in_bwd_apply_ex(const float* dy, int dy_ldc, const float* x, int x_ldc, const float* scale, const float* shift,
                                   float slope, const double* sums, const float* dx_add, int a_ldc, float* dx, int dx_ldc,
                                   void* dx16, void* xa16, int N, int64_t V, int C, void* stream) {
  int rc = check_cl(x, x_ldc, C); if (rc) return rc;
  rc = check_cl(dy, dy_ldc, C); if (rc) return rc;
  if (dx) { rc = check_cl(dx, dx_ldc, C); if (rc) return rc; }
  if (!dx && !dx16) return CWF_E_BADARG;
  if (dx_add) { rc = check_cl(dx_add, a_ldc, C); if (rc) return rc; }
  if (!scale || !shift || !sums) return CWF_E_BADARG;
  if (((uintptr_t)dx16 & 7) || ((uintptr_t)xa16 & 7)) return CWF_E_ALIGN;
  const int64_t total = (int64_t)N * V * (C >> 2);
  const dim3 grid((unsigned)cdiv64(total, 256));
#define CWF_IBA(f, d, xa) hipLaunchKernelGGL((in_bwd_apply_kernel<f, d, xa>), grid, dim3(256), 0, cwf_stream(stream), dy, dy_ldc, x, x_ldc, scale, shift, \
                     slope, sums, dx_add, a_ldc, dx, dx_ldc, (unsigned short*)dx16, (unsigned short*)xa16, V, C, total)
  if (dx) { if (dx16) { if (xa16) CWF_IBA(true, true, true); else CWF_IBA(true, true, false); }
            else      { if (xa16) CWF_IBA(true, false, true); else CWF_IBA(true, false, false); } }
  else    { if (xa16) CWF_IBA(false, true, true); else CWF_IBA(false, true, false); }
#undef CWF_IBA
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_to_bf16(const float* x, int x_ldc, const float* scale, const float* shift, float slope, void* y16,
                           int N, int64_t V, int C, void* stream) {
  int rc = check_cl(x, x_ldc, C); if (rc) return rc;
  if (!y16 || ((uintptr_t)y16 & 7) || (scale && !shift)) return CWF_E_BADARG;
  const int64_t total = (int64_t)N * V * (C >> 2);
  hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream),
                     x, x_ldc, scale, shift, slope, (unsigned short*)y16, V, C, total);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_in_bwd_apply(const float* dy, int dy_ldc, const float* x, int x_ldc, const float* scale, const float* shift,
                                float slope, const double* sums, const float* dx_add, int a_ldc, float* dx, int dx_ldc,
                                int N, int64_t V, int C, void* stream) {
  if (!dx) return CWF_E_BADARG;
  return cwf_in_bwd_apply_ex(dy, dy_ldc, x, x_ldc, scale, shift, slope, sums, dx_add, a_ldc, dx, dx_ldc, nullptr, nullptr, N, V, C, stream);
}
