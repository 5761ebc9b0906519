// K8/K10 -- supervision heads: trilinear upsample (align_corners=False) fused with the channel softmax, and the
// 4-class channel softmax of the decoder.  Reference: SuperviseLabel.py:62-64,69-72,77-80 and
// EdgeSuperviseLabel.py:58-60 (F.interpolate(..., mode='trilinear', align_corners=False) -> Softmax(dim=1));
// cls_wise_former.py:662-664 (endconv -> Softmax).  HBM-bound: the 12 [2,128^3] maps are written once.
#include "common.h"

// PyTorch's area_pixel_compute_source_index for align_corners=False with a given scale factor:
//   src = (dst + 0.5) / scale - 0.5, clamped below at 0;  i0 = floor(src), i1 = min(i0 + 1, in - 1), l1 = src - i0
__device__ __forceinline__ void src_index(int dst, float inv_scale, int in, int& i0, int& i1, float& l1) {
  float s = ((float)dst + 0.5f) * inv_scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

template <int C>
__global__ void upsample_softmax_kernel(const float* __restrict__ logit, int l_ldc, float* __restrict__ prob,
                                        int D, int H, int W, int scale, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over output voxels
  if (idx >= total) return;
  const int Wo = W * scale, Ho = H * scale, Do = D * scale;
  int64_t v = idx;
  const int ow = (int)(v % Wo); v /= Wo; const int oh = (int)(v % Ho); v /= Ho; const int od = (int)(v % Do); const int n = (int)(v / Do);
  const float inv = 1.0f / (float)scale;
  int d0, d1, h0, h1, w0, w1; float ld, lh, lw;
  src_index(od, inv, D, d0, d1, ld); src_index(oh, inv, H, h0, h1, lh); src_index(ow, inv, W, w0, w1, lw);
  float val[C];
#pragma unroll
  for (int c = 0; c < C; ++c) val[c] = 0.f;
  const float wd[2] = {1.f - ld, ld}, wh[2] = {1.f - lh, lh}, ww[2] = {1.f - lw, lw};
  const int ds[2] = {d0, d1}, hs[2] = {h0, h1}, ws[2] = {w0, w1};
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const float wgt = wd[a] * wh[b] * ww[c2];
        const float* p = logit + ((((int64_t)n * D + ds[a]) * H + hs[b]) * W + ws[c2]) * l_ldc;
#pragma unroll
        for (int c = 0; c < C; ++c) val[c] += wgt * p[c];
      }
  float mx = val[0];
#pragma unroll
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, val[c]);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { val[c] = expf(val[c] - mx); sum += val[c]; }
  const float r = 1.f / sum;
#pragma unroll
  for (int c = 0; c < C; ++c) prob[idx * C + c] = val[c] * r;
}

// gather form of the adjoint: one block per low-res voxel; candidates dst in [s*j - s/2, s*j + 3s/2 - 1] per dim
template <int C>
__global__ void upsample_softmax_bwd_kernel(const float* __restrict__ dprob, const float* __restrict__ prob, float* __restrict__ dlogit, int dl_ldc,
                                            int D, int H, int W, int scale) {
  __shared__ float red[4][C];
  int v = blockIdx.x;
  const int jw = v % W; v /= W; const int jh = v % H; const int jd = v / H;
  const int n = blockIdx.y;
  const int Wo = W * scale, Ho = H * scale, Do = D * scale;
  const float inv = 1.0f / (float)scale;
  const int span = 2 * scale, ncand = span * span * span;
  const int bd = scale * jd - scale / 2, bh = scale * jh - scale / 2, bw = scale * jw - scale / 2;
  float acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  for (int q = threadIdx.x; q < ncand; q += blockDim.x) {
    const int cw = q % span, ch = (q / span) % span, cd = q / (span * span);
    const int od = bd + cd, oh = bh + ch, ow = bw + cw;
    if (od < 0 || od >= Do || oh < 0 || oh >= Ho || ow < 0 || ow >= Wo) continue;
    int i0, i1; float l1;
    src_index(od, inv, D, i0, i1, l1); const float fd = (i0 == jd ? 1.f - l1 : 0.f) + (i1 == jd ? l1 : 0.f);
    src_index(oh, inv, H, i0, i1, l1); const float fh = (i0 == jh ? 1.f - l1 : 0.f) + (i1 == jh ? l1 : 0.f);
    src_index(ow, inv, W, i0, i1, l1); const float fw = (i0 == jw ? 1.f - l1 : 0.f) + (i1 == jw ? l1 : 0.f);
    const float wgt = fd * fh * fw;
    if (wgt == 0.f) continue;
    const int64_t o = ((((int64_t)n * Do + od) * Ho + oh) * Wo + ow) * C;
    float p[C], g[C]; float dot = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] = prob[o + c]; g[c] = dprob[o + c]; dot += p[c] * g[c]; }
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += wgt * p[c] * (g[c] - dot);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; ++c) { const float s = wave_sum(acc[c]); if (lane == 0) red[w][c] = s; }
  __syncthreads();
  if (threadIdx.x < C) {
    float s = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k][threadIdx.x];
    dlogit[((((int64_t)n * D + jd) * H + jh) * W + jw) * dl_ldc + threadIdx.x] = s;
  }
}

template <int C>
__global__ void channel_softmax_kernel(const float* __restrict__ logit, int l_ldc, float* __restrict__ prob, int64_t nvox) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  float val[C]; float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < C; ++c) { val[c] = logit[v * l_ldc + c]; mx = fmaxf(mx, val[c]); }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { val[c] = expf(val[c] - mx); sum += val[c]; }
  const float r = 1.f / sum;
#pragma unroll
  for (int c = 0; c < C; ++c) prob[v * C + c] = val[c] * r;
}

template <int C>
__global__ void channel_softmax_bwd_kernel(const float* __restrict__ dprob, const float* __restrict__ prob, float* __restrict__ dlogit, int dl_ldc, int64_t nvox) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  float p[C], g[C]; float dot = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { p[c] = prob[v * C + c]; g[c] = dprob[v * C + c]; dot += p[c] * g[c]; }
#pragma unroll
  for (int c = 0; c < C; ++c) dlogit[v * dl_ldc + c] = p[c] * (g[c] - dot);
}

extern "C" int cwf_upsample_softmax(const float* logit, int l_ldc, float* prob, int N, int D, int H, int W, int C, int scale, void* stream) {
  if (!logit || !prob || N <= 0 || scale <= 0 || l_ldc < C) return CWF_E_BADARG;
  const int64_t total = (int64_t)N * D * H * W * scale * scale * scale;
  dim3 grid((unsigned)cdiv64(total, 256));
  if (C == 2) hipLaunchKernelGGL(upsample_softmax_kernel<2>, grid, dim3(256), 0, cwf_stream(stream), logit, l_ldc, prob, D, H, W, scale, total);
  else if (C == 4) hipLaunchKernelGGL(upsample_softmax_kernel<4>, grid, dim3(256), 0, cwf_stream(stream), logit, l_ldc, prob, D, H, W, scale, total);
  else return CWF_E_BADARG;
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_upsample_softmax_bwd(const float* dprob, const float* prob, float* dlogit, int dl_ldc,
                                        int N, int D, int H, int W, int C, int scale, void* stream) {
  if (!dprob || !prob || !dlogit || N <= 0 || scale <= 0 || (scale & 1) || dl_ldc < C) return CWF_E_BADARG;
  dim3 grid(D * H * W, N);
  const int threads = scale >= 8 ? 256 : 64;
  if (C == 2) hipLaunchKernelGGL(upsample_softmax_bwd_kernel<2>, grid, dim3(threads), 0, cwf_stream(stream), dprob, prob, dlogit, dl_ldc, D, H, W, scale);
  else if (C == 4) hipLaunchKernelGGL(upsample_softmax_bwd_kernel<4>, grid, dim3(threads), 0, cwf_stream(stream), dprob, prob, dlogit, dl_ldc, D, H, W, scale);
  else return CWF_E_BADARG;
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_channel_softmax(const float* logit, int l_ldc, float* prob, int64_t nvox, int C, void* stream) {
  if (!logit || !prob || nvox <= 0 || l_ldc < C) return CWF_E_BADARG;
  dim3 grid((unsigned)cdiv64(nvox, 256));
  if (C == 2) hipLaunchKernelGGL(channel_softmax_kernel<2>, grid, dim3(256), 0, cwf_stream(stream), logit, l_ldc, prob, nvox);
  else if (C == 4) hipLaunchKernelGGL(channel_softmax_kernel<4>, grid, dim3(256), 0, cwf_stream(stream), logit, l_ldc, prob, nvox);
  else return CWF_E_BADARG;
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_channel_softmax_bwd(const float* dprob, const float* prob, float* dlogit, int dl_ldc, int64_t nvox, int C, void* stream) {
  if (!dprob || !prob || !dlogit || nvox <= 0 || dl_ldc < C) return CWF_E_BADARG;
  dim3 grid((unsigned)cdiv64(nvox, 256));
  if (C == 2) hipLaunchKernelGGL(channel_softmax_bwd_kernel<2>, grid, dim3(256), 0, cwf_stream(stream), dprob, prob, dlogit, dl_ldc, nvox);
  else if (C == 4) hipLaunchKernelGGL(channel_softmax_bwd_kernel<4>, grid, dim3(256), 0, cwf_stream(stream), dprob, prob, dlogit, dl_ldc, nvox);
  else return CWF_E_BADARG;
  CWF_LAUNCH_CHECK();
  return 0;
}
