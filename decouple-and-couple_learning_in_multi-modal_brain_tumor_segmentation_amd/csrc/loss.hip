// K9 -- fused Dice + class-frequency-weighted cross-entropy (one pass over the probability map, f64 sums,
// analytic backward).  Reference: utils/tools.py:8-18 (dice_loss), :21-34 (softmax_weighted_loss), :112-231
// (get_separate_loss / get_edge_separate_loss label remaps), models/criterions.py:49-62 (softmax_dice).
// The reference materialises int64 one-hots, .float() copies and a repeat()'d weight volume per class;
// here the label is decoded on the fly: C==4 -> class = label; C==2 -> class = (posmask >> label) & 1.
#include "common.h"

template <int C>
__global__ __launch_bounds__(256) void dice_ce_sums_kernel(const float* __restrict__ prob, const int64_t* __restrict__ label, uint32_t posmask,
                                                          double* __restrict__ sums, int64_t V, int vox_per_block) {
  __shared__ float red[4][C * 4];
  const int n = blockIdx.y;
  const int64_t v0 = (int64_t)blockIdx.x * vox_per_block;
  const int64_t v1 = min(V, v0 + vox_per_block);
  float a[C][4];
#pragma unroll
  for (int c = 0; c < C; ++c) { a[c][0] = a[c][1] = a[c][2] = a[c][3] = 0.f; }
  for (int64_t v = v0 + threadIdx.x; v < v1; v += 256) {
    const int64_t gv = (int64_t)n * V + v;
    const int lab = (int)label[gv];
    const int cls = (C == 4) ? lab : (int)((posmask >> (lab & 31)) & 1u);
    float p[C];
    if (C == 4) { const float4 q = *reinterpret_cast<const float4*>(prob + gv * 4); p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w; }
    else { const float2 q = *reinterpret_cast<const float2*>(prob + gv * 2); p[0] = q.x; p[1] = q.y; }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float t = (cls == c) ? 1.f : 0.f;
      a[c][0] += p[c] * t; a[c][1] += p[c]; a[c][2] += t;
      a[c][3] += t * logf(fminf(fmaxf(p[c], 0.005f), 1.0f));
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float s = wave_sum(a[c][k]); if (lane == 0) red[w][c * 4 + k] = s; }
  __syncthreads();
  if (threadIdx.x < C * 4) {
    const double s = (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x];
    atomic_add_f64(sums + ((int64_t)n * C) * 4 + threadIdx.x, s);
  }
}

// single small block: loss scalar + backward coefficients
//   dice = 1 - (1/C) sum_c 2 I_c / (P_c + T_c + 1e-7)          (sums over the whole batch)
//   ce   = (1/(N V)) sum_n sum_c -w_c[n] S_c[n],  w_c[n] = 1 - T_c[n] / sum_c T_c[n]
//   coef[n][c] = ( a_c = -(2/C)/den_c , b_c = (2/C) I_c/den_c^2 , k_c[n] = -w_c[n]/(N V) , 0 )
__global__ void dice_ce_finalize_kernel(const double* __restrict__ sums, float* __restrict__ loss, float* __restrict__ coef, int N, double V, int C,
                                        int nmaps, float* __restrict__ total) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float tot_all = 0.f;                               // fp32, in order: the reference sums the per-region loss tensors (tools.py:160-162)
  for (int m = 0; m < nmaps; ++m, sums += (int64_t)N * C * 4, coef += (int64_t)N * C * 4, ++loss) {
  double dice = 0.0, ce = 0.0;
  for (int c = 0; c < C; ++c) {
    double I = 0, P = 0, T = 0;
    for (int n = 0; n < N; ++n) { const double* s = sums + ((int64_t)n * C + c) * 4; I += s[0]; P += s[1]; T += s[2]; }
    // the reference accumulates these in fp32 tensors; den mirrors `l + r + eps` there
    const double den = P + T + 1e-7;
    dice += 2.0 * I / den;
    for (int n = 0; n < N; ++n) {
      coef[((int64_t)n * C + c) * 4 + 0] = (float)(-(2.0 / C) / den);
      coef[((int64_t)n * C + c) * 4 + 1] = (float)((2.0 / C) * I / (den * den));
    }
  }
  for (int n = 0; n < N; ++n) {
    double tot = 0;
    for (int c = 0; c < C; ++c) tot += sums[((int64_t)n * C + c) * 4 + 2];
    for (int c = 0; c < C; ++c) {
      const double* s = sums + ((int64_t)n * C + c) * 4;
      const double w = 1.0 - s[2] / tot;
      ce += -w * s[3];
      coef[((int64_t)n * C + c) * 4 + 2] = (float)(-w / ((double)N * V));
      coef[((int64_t)n * C + c) * 4 + 3] = 0.f;
    }
  }
  loss[0] = (float)((1.0 - dice / C) + ce / ((double)N * V));
  tot_all += loss[0];
  }
  if (total) total[0] = tot_all;
}

template <int C>
__global__ void dice_ce_bwd_kernel(const float* __restrict__ prob, const int64_t* __restrict__ label, uint32_t posmask,
                                   const float* __restrict__ coef, const float* __restrict__ gscale, float* __restrict__ dprob,
                                   int64_t V, int64_t total) {
  const int64_t gv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gv >= total) return;
  const int64_t n = gv / V;
  const float gs = gscale[0];
  const int lab = (int)label[gv];
  const int cls = (C == 4) ? lab : (int)((posmask >> (lab & 31)) & 1u);
  float o[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float* k = coef + (n * C + c) * 4;
    const float p = prob[gv * C + c];
    float g = k[1];
    if (cls == c) {
      g += k[0];
      if (p >= 0.005f && p <= 1.0f) g += k[2] / p;      // clamp passes gradient on [min, max] (torch.clamp backward)
    }
    o[c] = gs * g;
  }
  if (C == 4) *reinterpret_cast<float4*>(dprob + gv * 4) = make_float4(o[0], o[1], o[2], o[3]);
  else *reinterpret_cast<float2*>(dprob + gv * 2) = make_float2(o[0], o[1]);
}

extern "C" int cwf_dice_ce_sums(const float* prob, const int64_t* label, uint32_t posmask, double* sums, int N, int64_t V, int C, void* stream) {
  if (!prob || !label || !sums || N <= 0 || V <= 0) return CWF_E_BADARG;
  int64_t vpb = cdiv64(V * N, 2048); if (vpb < 1024) vpb = 1024; if (vpb > V) vpb = V;
  dim3 grid((unsigned)cdiv64(V, vpb), N);
  if (C == 4) hipLaunchKernelGGL(dice_ce_sums_kernel<4>, grid, dim3(256), 0, cwf_stream(stream), prob, label, posmask, sums, V, (int)vpb);
  else if (C == 2) hipLaunchKernelGGL(dice_ce_sums_kernel<2>, grid, dim3(256), 0, cwf_stream(stream), prob, label, posmask, sums, V, (int)vpb);
  else return CWF_E_BADARG;
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_dice_ce_finalize(const double* sums, float* loss, float* coef, int N, int64_t V, int C, void* stream) {
  if (!sums || !loss || !coef || N <= 0 || V <= 0 || (C != 2 && C != 4)) return CWF_E_BADARG;
  hipLaunchKernelGGL(dice_ce_finalize_kernel, dim3(1), dim3(64), 0, cwf_stream(stream), sums, loss, coef, N, (double)V, C, 1, (float*)nullptr);
  CWF_LAUNCH_CHECK();
  return 0;
}

// nmaps problems laid out back to back (sums [nmaps][N][C][4], loss [nmaps], coef [nmaps][N][C][4]) + their sum -> total[0]
extern "C" int cwf_dice_ce_finalize_multi(const double* sums, float* loss, float* coef, float* total, int nmaps, int N, int64_t V, int C, void* stream) {
  if (!sums || !loss || !coef || nmaps <= 0 || N <= 0 || V <= 0 || (C != 2 && C != 4)) return CWF_E_BADARG;
  hipLaunchKernelGGL(dice_ce_finalize_kernel, dim3(1), dim3(64), 0, cwf_stream(stream), sums, loss, coef, N, (double)V, C, nmaps, total);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_dice_ce_bwd(const float* prob, const int64_t* label, uint32_t posmask, const float* coef, const float* gscale,
                               float* dprob, int N, int64_t V, int C, void* stream) {
  if (!prob || !label || !coef || !gscale || !dprob || N <= 0 || V <= 0) return CWF_E_BADARG;
  const int64_t total = (int64_t)N * V;
  dim3 grid((unsigned)cdiv64(total, 256));
  if (C == 4) hipLaunchKernelGGL(dice_ce_bwd_kernel<4>, grid, dim3(256), 0, cwf_stream(stream), prob, label, posmask, coef, gscale, dprob, V, total);
  else if (C == 2) hipLaunchKernelGGL(dice_ce_bwd_kernel<2>, grid, dim3(256), 0, cwf_stream(stream), prob, label, posmask, coef, gscale, dprob, V, total);
  else return CWF_E_BADARG;
  CWF_LAUNCH_CHECK();
  return 0;
}
