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

// softmax over C channels of the trilinear interpolation of `logit` at high-resolution voxel (od, oh, ow) of sample n.
// One definition for the map-writing kernel and the fused head -> loss kernels: identical arithmetic, identical probabilities.
template <int C>
__device__ __forceinline__ void upsample_softmax_at(const float* __restrict__ logit, int l_ldc, int n, int D, int H, int W,
                                                    int od, int oh, int ow, float inv, float* val) {
  int d0, d1, h0, h1, w0, w1; float ld, lh, lw;
  src_index(od, inv, D, d0, d1, ld); src_index(oh, inv, H, h0, h1, lh); src_index(ow, inv, W, w0, w1, lw);
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
  for (int c = 0; c < C; ++c) val[c] *= r;
}

template <int C>
__global__ void upsample_softmax_kernel(const float* __restrict__ logit, int l_ldc, float* __restrict__ prob,
                                        int D, int H, int W, int scale, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over output voxels
  if (idx >= total) return;
  const int Wo = W * scale, Ho = H * scale, Do = D * scale;
  int64_t v = idx;
  const int ow = (int)(v % Wo); v /= Wo; const int oh = (int)(v % Ho); v /= Ho; const int od = (int)(v % Do); const int n = (int)(v / Do);
  float val[C];
  upsample_softmax_at<C>(logit, l_ldc, n, D, H, W, od, oh, ow, 1.0f / (float)scale, val);
#pragma unroll
  for (int c = 0; c < C; ++c) prob[idx * C + c] = val[c];
}

// Adjoint of upsample + softmax, separable and deterministic (no atomics), two launches:
//   t[od,oh,ow,c]   = p_c (g_c - sum_k p_k g_k)                              (softmax', pointwise at high resolution)
//   ws[od,jh,jw,c]  = sum_oh fh(oh,jh) sum_ow fw(ow,jw) t[od,oh,ow,c]        (rows kernel: one block per (n, od, jh))
//   dlogit[jd,jh,jw,c] = sum_od fd(od,jd) ws[od,jh,jw,c]                     (planes kernel)
// f*(o, j) = trilinear weight of low-res index j in high-res index o; non-zero only for o in [s*j - s/2, s*j + 3s/2).
// The rows kernel reads each high-res row (Wo*C contiguous floats) coalesced, twice in total (two adjacent jh);
// a one-block-per-low-res-voxel gather read every row 8 times with scattered accesses (117 us vs ~35 us at 2 x 2 x 128^3).
__device__ __forceinline__ float tri_weight(int o, float inv, int in, int j) {
  int i0, i1; float l1;
  src_index(o, inv, in, i0, i1, l1);
  return (i0 == j ? 1.f - l1 : 0.f) + (i1 == j ? l1 : 0.f);
}

template <int C>
__global__ void upsample_softmax_bwd_rows_kernel(const float* __restrict__ dprob, const float* __restrict__ prob, float* __restrict__ ws,
                                                 int D, int H, int W, int scale) {
  extern __shared__ float col[];                       // [Wo][C]: per high-res column, the fh-weighted sum over this block's rows
  int b = blockIdx.x;
  const int jh = b % H; b /= H;
  const int Do = D * scale, Ho = H * scale, Wo = W * scale;
  const int od = b % Do; const int n = b / Do;
  const float inv = 1.0f / (float)scale;
  const int oh_lo = max(0, scale * jh - scale / 2), oh_hi = min(Ho, scale * jh + scale + scale / 2);
  const float* pb = prob + (((int64_t)n * Do + od) * Ho) * (int64_t)Wo * C;
  const float* gb = dprob + (((int64_t)n * Do + od) * Ho) * (int64_t)Wo * C;
  for (int ow = threadIdx.x; ow < Wo; ow += blockDim.x) {
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    for (int oh = oh_lo; oh < oh_hi; ++oh) {
      const float fh = tri_weight(oh, inv, H, jh);
      const int64_t o = ((int64_t)oh * Wo + ow) * C;
      float p[C], g[C]; float dot = 0.f;
      if (C == 2) {
        const float2 pv = *reinterpret_cast<const float2*>(pb + o), gv = *reinterpret_cast<const float2*>(gb + o);
        p[0] = pv.x; p[1] = pv.y; g[0] = gv.x; g[1] = gv.y;
      } else {
        const float4 pv = *reinterpret_cast<const float4*>(pb + o), gv = *reinterpret_cast<const float4*>(gb + o);
        p[0] = pv.x; p[1] = pv.y; p[C - 2] = pv.z; p[C - 1] = pv.w; g[0] = gv.x; g[1] = gv.y; g[C - 2] = gv.z; g[C - 1] = gv.w;
      }
#pragma unroll
      for (int c = 0; c < C; ++c) dot += p[c] * g[c];
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] += fh * p[c] * (g[c] - dot);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) col[ow * C + c] = acc[c];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < W * C; q += blockDim.x) {
    const int jw = q / C, c = q % C;
    const int ow_lo = max(0, scale * jw - scale / 2), ow_hi = min(Wo, scale * jw + scale + scale / 2);
    float s = 0.f;
    for (int ow = ow_lo; ow < ow_hi; ++ow) s += tri_weight(ow, inv, W, jw) * col[ow * C + c];
    ws[((((int64_t)n * Do + od) * H + jh) * W + jw) * C + c] = s;
  }
}

template <int C>
__global__ void upsample_softmax_bwd_planes_kernel(const float* __restrict__ ws, float* __restrict__ dlogit, int dl_ldc,
                                                   int D, int H, int W, int scale, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over (n, jd, jh, jw, c)
  if (idx >= total) return;
  int64_t v = idx;
  const int c = (int)(v % C); v /= C;
  const int64_t hw = v % ((int64_t)H * W); v /= (int64_t)H * W;             // jh * W + jw
  const int jd = (int)(v % D); const int n = (int)(v / D);
  const int Do = D * scale;
  const float inv = 1.0f / (float)scale;
  const int od_lo = max(0, scale * jd - scale / 2), od_hi = min(Do, scale * jd + scale + scale / 2);
  float s = 0.f;
  for (int od = od_lo; od < od_hi; ++od) s += tri_weight(od, inv, D, jd) * ws[((((int64_t)n * Do + od) * H * W) + hw) * C + c];
  dlogit[(((int64_t)n * D + jd) * H * W + hw) * dl_ldc + c] = s;
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
                                        int N, int D, int H, int W, int C, int scale, float* workspace, void* stream) {
  if (!dprob || !prob || !dlogit || !workspace || N <= 0 || scale <= 0 || (scale & 1) || dl_ldc < C) return CWF_E_BADARG;
  if (C != 2 && C != 4) return CWF_E_BADARG;
  const int Wo = W * scale;
  const int threads = Wo >= 256 ? 256 : ((Wo + 63) / 64) * 64;
  const size_t lds = (size_t)Wo * C * sizeof(float);
  if (lds > 64 * 1024) return CWF_E_BADARG;
  dim3 grid1((unsigned)((int64_t)N * D * scale * H));
  const int64_t total = (int64_t)N * D * H * W * C;
  dim3 grid2((unsigned)cdiv64(total, 256));
  hipStream_t st = cwf_stream(stream);
  if (C == 2) {
    hipLaunchKernelGGL(upsample_softmax_bwd_rows_kernel<2>, grid1, dim3(threads), lds, st, dprob, prob, workspace, D, H, W, scale);
    hipLaunchKernelGGL(upsample_softmax_bwd_planes_kernel<2>, grid2, dim3(256), 0, st, workspace, dlogit, dl_ldc, D, H, W, scale, total);
  } else {
    hipLaunchKernelGGL(upsample_softmax_bwd_rows_kernel<4>, grid1, dim3(threads), lds, st, dprob, prob, workspace, D, H, W, scale);
    hipLaunchKernelGGL(upsample_softmax_bwd_planes_kernel<4>, grid2, dim3(256), 0, st, workspace, dlogit, dl_ldc, D, H, W, scale, total);
  }
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

// =====================================================================================================================
// Fused head -> loss (training): the per-sub-region Dice / weighted-CE sums straight from the LOW-resolution logits.
//   reference: SuperviseLabel.py:58-81 / EdgeSuperviseLabel.py:56-76 (conv -> conv -> trilinear -> softmax) feeding
//   tools.get_separate_loss / get_edge_separate_loss (tools.py:112-231 -> dice_loss :8-18 + softmax_weighted_loss :21-34).
// The unfused path writes each [N,2,D,H,W] probability map (upsample_softmax), re-reads it for the sums, re-reads it for
// dLoss/dprob, writes dprob and re-reads both in the interpolation adjoint: ~6 full-resolution passes per map, 12 maps per step.
// Here the upsample + softmax is evaluated in registers where it is needed (the 16^3 / 32^3 logits stay in L1/L2), the three
// maps of one supervision call share ONE read of the label volume, and nothing is written at full resolution.
//   forward : sums[m][n][c][4] += (sum p t, sum p, sum t, sum t log clamp(p, 0.005, 1))     (f64 atomics per block, zeroed by caller)
//   backward: ws[m][n][od][jh][jw][c] = sum_oh fh sum_ow fw p_c (g_c - sum_k p_k g_k), g = dLoss/dp from (label, coef);
//             then the planes pass finishes along D and writes dlogit (pad channels zeroed)
// =====================================================================================================================
struct HeadLossArgs {
  const float* logit[3]; uint32_t posmask[3]; int l_ldc; const int64_t* label; double* sums; const float* coef; const float* gscale;
  float* ws; int N, D, H, W, scale, nm; int64_t V; int vox_per_block;
};

// Separable evaluation: for a high-resolution row (od, oh) the D- and H-interpolation is the same for every ow, so a block first
// builds P[row][map][w][c] = sum_{a,b} wd[a] wh[b] logit_m[d_a][h_b][w][c] in LDS (W = 16 or 32 low-resolution columns) and a
// voxel then costs two LDS reads per channel, one lerp and the 2-class softmax -- ~6x fewer instructions than the eight-corner
// gather per voxel and map (the fused kernels were VALU-bound on it).  Rounding order differs from upsample_softmax_at by ~1e-7.
#define HL_ROWS 8
__device__ __forceinline__ void hl_build_rows(float* P, const HeadLossArgs& a, int n, int od, int oh0, int nrows, int tid, int nthreads) {
  const float inv = 1.0f / (float)a.scale;
  int d0, d1; float ld;
  src_index(od, inv, a.D, d0, d1, ld);
  const int per_row = a.nm * a.W * 2;
  for (int q = tid; q < nrows * per_row; q += nthreads) {
    const int c = q & 1; int t = q >> 1;
    const int w = t % a.W; t /= a.W;
    const int m = t % a.nm; const int rr = t / a.nm;
    int h0, h1; float lh;
    src_index(oh0 + rr, inv, a.H, h0, h1, lh);
    const float* L = a.logit[m];
    const int64_t nb = (int64_t)n * a.D;
    const float v00 = L[(((nb + d0) * a.H + h0) * a.W + w) * a.l_ldc + c], v01 = L[(((nb + d0) * a.H + h1) * a.W + w) * a.l_ldc + c];
    const float v10 = L[(((nb + d1) * a.H + h0) * a.W + w) * a.l_ldc + c], v11 = L[(((nb + d1) * a.H + h1) * a.W + w) * a.l_ldc + c];
    P[q] = (1.f - ld) * ((1.f - lh) * v00 + lh * v01) + ld * ((1.f - lh) * v10 + lh * v11);
  }
}
__device__ __forceinline__ void hl_prob(const float* Prow, int W, int w0, int w1, float lw, float* p) {
  float v0 = (1.f - lw) * Prow[w0 * 2] + lw * Prow[w1 * 2];
  float v1 = (1.f - lw) * Prow[w0 * 2 + 1] + lw * Prow[w1 * 2 + 1];
  // (hardware exp2 / rcp: the fused head -> loss kernels are VALU-bound on 25 M of these per step; the full-precision expf / divide
  // sequences cost ~4x the instructions for a difference of ~1e-7 relative, far inside the 1e-4 loss tolerance)
  const float mx = fmaxf(v0, v1);
  v0 = __expf(v0 - mx); v1 = __expf(v1 - mx);
  const float r = __frcp_rn(v0 + v1);
  p[0] = v0 * r; p[1] = v1 * r;
}

// grid (Do * ceil(Ho / (HL_ROWS * HL_GROUPS)), N); a block walks HL_GROUPS groups of HL_ROWS consecutive high-resolution rows of one
// plane and issues its 24 f64 atomics once (4,096 blocks of one group each spent most of the launch serialising 98 k atomics on 48
// addresses)
#define HL_GROUPS 4
__global__ __launch_bounds__(256) void head_loss_sums_kernel(const HeadLossArgs a) {
  __shared__ float P[HL_ROWS * 3 * 64 * 2];
  __shared__ float red[4][3 * 8];
  const int n = blockIdx.y;
  const int Wo = a.W * a.scale, Ho = a.H * a.scale;
  const int rgroups = (Ho + HL_ROWS * HL_GROUPS - 1) / (HL_ROWS * HL_GROUPS);
  const int od = blockIdx.x / rgroups;
  const float inv = 1.0f / (float)a.scale;
  float acc[3][2][4];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int c = 0; c < 2; ++c) { acc[m][c][0] = acc[m][c][1] = acc[m][c][2] = acc[m][c][3] = 0.f; }
  for (int grp = 0; grp < HL_GROUPS; ++grp) {
  const int oh0 = ((blockIdx.x % rgroups) * HL_GROUPS + grp) * HL_ROWS;
  if (oh0 >= Ho) break;
  const int nrows = min(HL_ROWS, Ho - oh0);
  __syncthreads();
  hl_build_rows(P, a, n, od, oh0, nrows, threadIdx.x, 256);
  __syncthreads();
  for (int q = threadIdx.x; q < nrows * Wo; q += 256) {
    const int rr = q / Wo, ow = q % Wo;
    int w0, w1; float lw;
    src_index(ow, inv, a.W, w0, w1, lw);
    const int lab = (int)a.label[(((int64_t)n * a.D * a.scale + od) * Ho + oh0 + rr) * Wo + ow];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      if (m >= a.nm) break;
      float p[2];
      hl_prob(P + (rr * a.nm + m) * a.W * 2, a.W, w0, w1, lw, p);
      const int cls = (int)((a.posmask[m] >> (lab & 31)) & 1u);
      const float lp = __logf(fminf(fmaxf(cls ? p[1] : p[0], 0.005f), 1.0f));      // one logarithm: the true class's
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float tt = (cls == c) ? 1.f : 0.f;
        acc[m][c][0] += p[c] * tt; acc[m][c][1] += p[c]; acc[m][c][2] += tt;
        acc[m][c][3] += tt * lp;
      }
    }
  }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float s = wave_sum(acc[m][c][k]); if (lane == 0) red[w][m * 8 + c * 4 + k] = s; }
  __syncthreads();
  if (threadIdx.x < a.nm * 8) {
    const int m = threadIdx.x >> 3, ck = threadIdx.x & 7;
    const double s = (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x];
    atomic_add_f64(a.sums + (((int64_t)m * a.N + n) * 2) * 4 + ck, s);
  }
}

// one block per (n, od, jh), like upsample_softmax_bwd_rows_kernel; LDS: P[window rows][nm][W][2] then col[nm][Wo][2]
__global__ void head_loss_bwd_rows_kernel(const HeadLossArgs a) {
  extern __shared__ float hl_lds[];
  int b = blockIdx.x;
  const int H = a.H, W = a.W, D = a.D, scale = a.scale;
  const int jh = b % H; b /= H;
  const int Do = D * scale, Ho = H * scale, Wo = W * scale;
  const int od = b % Do; const int n = b / Do;
  const float inv = 1.0f / (float)scale;
  const float gs = a.gscale[0];
  const int oh_lo = max(0, scale * jh - scale / 2), oh_hi = min(Ho, scale * jh + scale + scale / 2);
  float* P = hl_lds;
  float* col = hl_lds + 2 * scale * a.nm * W * 2;
  hl_build_rows(P, a, n, od, oh_lo, oh_hi - oh_lo, threadIdx.x, blockDim.x);
  __syncthreads();
  const int64_t* lb = a.label + (((int64_t)n * Do + od) * Ho) * (int64_t)Wo;
  float kc[3][2][3];                                   // the (a, b, k) coefficients of this sample, pre-multiplied by the upstream scalar
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* k = a.coef + (((int64_t)min(m, a.nm - 1) * a.N + n) * 2 + c) * 4;
      kc[m][c][0] = gs * k[0]; kc[m][c][1] = gs * k[1]; kc[m][c][2] = gs * k[2];
    }
  for (int ow = threadIdx.x; ow < Wo; ow += blockDim.x) {
    int w0, w1; float lw;
    src_index(ow, inv, W, w0, w1, lw);
    float acc[3][2];
#pragma unroll
    for (int m = 0; m < 3; ++m) { acc[m][0] = 0.f; acc[m][1] = 0.f; }
    for (int oh = oh_lo; oh < oh_hi; ++oh) {
      const float fh = tri_weight(oh, inv, H, jh);
      const int lab = (int)lb[(int64_t)oh * Wo + ow];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        if (m >= a.nm) break;
        float p[2], g[2];
        hl_prob(P + ((oh - oh_lo) * a.nm + m) * W * 2, W, w0, w1, lw, p);
        const int cls = (int)((a.posmask[m] >> (lab & 31)) & 1u);
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float gg = kc[m][c][1];
          if (cls == c) { gg += kc[m][c][0]; if (p[c] >= 0.005f && p[c] <= 1.0f) gg += kc[m][c][2] / p[c]; }
          g[c] = gg; dot += p[c] * g[c];
        }
        acc[m][0] += fh * p[0] * (g[0] - dot); acc[m][1] += fh * p[1] * (g[1] - dot);
      }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) { if (m < a.nm) { col[(m * Wo + ow) * 2] = acc[m][0]; col[(m * Wo + ow) * 2 + 1] = acc[m][1]; } }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < a.nm * W * 2; q += blockDim.x) {
    const int c = q & 1, jw = (q >> 1) % W, m = (q >> 1) / W;
    const int ow_lo = max(0, scale * jw - scale / 2), ow_hi = min(Wo, scale * jw + scale + scale / 2);
    float s = 0.f;
    for (int ow = ow_lo; ow < ow_hi; ++ow) s += tri_weight(ow, inv, W, jw) * col[(m * Wo + ow) * 2 + c];
    a.ws[(((((int64_t)m * a.N + n) * Do + od) * H + jh) * W + jw) * 2 + c] = s;
  }
}

struct HeadPlanesArgs { float* dlogit[3]; const float* ws; int dl_ldc, dl_ca, N, D, H, W, scale, nm; int64_t per_map; };
// dlogit_m[n][jd][jh][jw][c] = sum_od fd(od, jd) ws[m][n][od][jh][jw][c] for c < 2, ZERO for the pad channels c in [2, dl_ca);
// voxel rows are dl_ldc floats apart (dl_ldc > dl_ca: the maps are channel groups of one gradient buffer, cwf_head_loss_bwd_ex)
__global__ void head_loss_bwd_planes_kernel(const HeadPlanesArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over (n, jd, jh, jw, c < dl_ca)
  if (idx >= a.per_map) return;
  const int m = blockIdx.y;
  int64_t v = idx;
  const int c = (int)(v % a.dl_ca); v /= a.dl_ca;
  const int64_t vox = v;
  const int64_t hw = v % ((int64_t)a.H * a.W); v /= (int64_t)a.H * a.W;
  const int jd = (int)(v % a.D); const int n = (int)(v / a.D);
  float s = 0.f;
  if (c < 2) {
    const int Do = a.D * a.scale;
    const float inv = 1.0f / (float)a.scale;
    const int od_lo = max(0, a.scale * jd - a.scale / 2), od_hi = min(Do, a.scale * jd + a.scale + a.scale / 2);
    for (int od = od_lo; od < od_hi; ++od)
      s += tri_weight(od, inv, a.D, jd) * a.ws[(((((int64_t)m * a.N + n) * Do + od) * a.H * a.W) + hw) * 2 + c];
  }
  a.dlogit[m][vox * a.dl_ldc + c] = s;
}

extern "C" int cwf_head_loss_sums(const float* const* logits, int nmaps, int l_ldc, const uint32_t* posmasks, const int64_t* label,
                                  double* sums, int N, int D, int H, int W, int scale, void* stream) {
  if (!logits || !posmasks || !label || !sums || nmaps <= 0 || nmaps > 3 || N <= 0 || scale <= 0 || l_ldc < 2) return CWF_E_BADARG;
  HeadLossArgs a = {};
  for (int m = 0; m < nmaps; ++m) { if (!logits[m]) return CWF_E_BADARG; a.logit[m] = logits[m]; a.posmask[m] = posmasks[m]; }
  a.l_ldc = l_ldc; a.label = label; a.sums = sums; a.N = N; a.D = D; a.H = H; a.W = W; a.scale = scale; a.nm = nmaps;
  a.V = (int64_t)D * H * W * scale * scale * scale;
  if (W > 64) return CWF_E_TOOLARGE;                       // the LDS row table holds up to 64 low-resolution columns
  const int rgroups = (H * scale + HL_ROWS * HL_GROUPS - 1) / (HL_ROWS * HL_GROUPS);
  hipLaunchKernelGGL(head_loss_sums_kernel, dim3((unsigned)(D * scale * rgroups), N), dim3(256), 0, cwf_stream(stream), a);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_head_loss_bwd_ex(const float* const* logits, int nmaps, int l_ldc, const uint32_t* posmasks, const int64_t* label,
                                    const float* coef, const float* gscale, float* const* dlogits, int dl_ca, int dl_ldc, float* workspace,
                                    int N, int D, int H, int W, int scale, void* stream);
extern "C" int cwf_head_loss_bwd(const float* const* logits, int nmaps, int l_ldc, const uint32_t* posmasks, const int64_t* label,
                                 const float* coef, const float* gscale, float* const* dlogits, int dl_ldc, float* workspace,
                                 int N, int D, int H, int W, int scale, void* stream) {
  return cwf_head_loss_bwd_ex(logits, nmaps, l_ldc, posmasks, label, coef, gscale, dlogits, dl_ldc, dl_ldc, workspace, N, D, H, W, scale, stream);
}
extern "C" int cwf_head_loss_bwd_ex(const float* const* logits, int nmaps, int l_ldc, const uint32_t* posmasks, const int64_t* label,
                                    const float* coef, const float* gscale, float* const* dlogits, int dl_ca, int dl_ldc, float* workspace,
                                    int N, int D, int H, int W, int scale, void* stream) {
  if (!logits || !posmasks || !label || !coef || !gscale || !dlogits || !workspace || nmaps <= 0 || nmaps > 3 || N <= 0) return CWF_E_BADARG;
  if (scale <= 0 || (scale & 1) || l_ldc < 2 || dl_ca < 2 || dl_ldc < dl_ca) return CWF_E_BADARG;
  HeadLossArgs a = {};
  HeadPlanesArgs pa = {};
  for (int m = 0; m < nmaps; ++m) {
    if (!logits[m] || !dlogits[m]) return CWF_E_BADARG;
    a.logit[m] = logits[m]; a.posmask[m] = posmasks[m]; pa.dlogit[m] = dlogits[m];
  }
  a.l_ldc = l_ldc; a.label = label; a.coef = coef; a.gscale = gscale; a.ws = workspace;
  a.N = N; a.D = D; a.H = H; a.W = W; a.scale = scale; a.nm = nmaps;
  const int Wo = W * scale;
  const int threads = Wo >= 256 ? 256 : ((Wo + 63) / 64) * 64;
  const size_t lds = ((size_t)2 * scale * nmaps * W * 2 + (size_t)nmaps * Wo * 2) * sizeof(float);
  if (lds > 64 * 1024) return CWF_E_BADARG;
  hipStream_t st = cwf_stream(stream);
  hipLaunchKernelGGL(head_loss_bwd_rows_kernel, dim3((unsigned)((int64_t)N * D * scale * H)), dim3(threads), lds, st, a);
  pa.ws = workspace; pa.dl_ldc = dl_ldc; pa.dl_ca = dl_ca; pa.N = N; pa.D = D; pa.H = H; pa.W = W; pa.scale = scale; pa.nm = nmaps;
  pa.per_map = (int64_t)N * D * H * W * dl_ca;
  hipLaunchKernelGGL(head_loss_bwd_planes_kernel, dim3((unsigned)cdiv64(pa.per_map, 256), nmaps), dim3(256), 0, st, pa);
  CWF_LAUNCH_CHECK();
  return 0;
}
