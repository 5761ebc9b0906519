// N1 -- sliding-window inference glue on the device: the stitch of the eight 128^3 windows of
// predict_overlap.tailor_and_concat (predict_overlap.py:31-58) and argmax + WT / TC / ET Dice counts
// (predict_overlap.py:134-141, utils/tools.py:44-47,89-109) as one launch each.  The reference does `y = x.clone()` plus eight
// slice assignments, `argmax`, and nine boolean-mask reductions on the host side of ATen.
#include "common.h"

// y[b][ch][a][bb][c] (NCDHW, 240 x 240 x 155) from win[(w*B + b)][la][lb][lc][ch] (channels-last, 128^3, 4 classes), w = window index in
// the reference's order (predict_overlap.py:34-41): starts {0,112} along the first two axes, {0,27} along the last, hard overwrite
// with the later window, INCLUDING the reference's last-axis quirk: y[..., 128:155] = window(27:155)[..., 96:123].
__global__ void stitch_windows_kernel(const float* __restrict__ win, float* __restrict__ y, int B, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over (b, a, bb, c)
  if (idx >= total) return;
  int64_t v = idx;
  const int c = (int)(v % 155); v /= 155;
  const int bb = (int)(v % 240); v /= 240;
  const int a = (int)(v % 240); const int b = (int)(v / 240);
  const int ai = a >= 128, bi = bb >= 128, ci = c >= 128;
  const int la = ai ? a - 112 : a, lb = bi ? bb - 112 : bb, lc = ci ? c - 128 + 96 : c;
  const int w = ci * 4 + ai * 2 + bi;
  const float4 p = *reinterpret_cast<const float4*>(win + ((((int64_t)(w * B + b) * 128 + la) * 128 + lb) * 128 + lc) * 4);
  const int64_t plane = (int64_t)240 * 240 * 155;
  float* dst = y + (int64_t)b * 4 * plane + ((int64_t)a * 240 + bb) * 155 + c;
  dst[0] = p.x; dst[plane] = p.y; dst[2 * plane] = p.z; dst[3 * plane] = p.w;
}

extern "C" int cwf_stitch_windows(const float* windows, float* y, int B, void* stream) {
  if (!windows || !y || B <= 0 || ((uintptr_t)windows & 15)) return CWF_E_BADARG;
  const int64_t total = (int64_t)B * 240 * 240 * 155;
  hipLaunchKernelGGL(stitch_windows_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream), windows, y, B, total);
  CWF_LAUNCH_CHECK();
  return 0;
}

// seg[v] = argmax_c prob[b][c][v] (first maximum, as torch.argmax) ; counts[k][0..2] += (|o & t|, |o|, |t|) for the three BraTS regions
// k = WT (label > 0), TC (label 1 or 3), ET (label 3) of tools.softmax_output_dice and, with NC = 18, for the three classes
// k = 3 + (c - 1), c = 1, 2, 3 of tools.softmax_mIOU_score (|o | t| = |o| + |t| - |o & t|).  target may be NULL (no counts).
template <int NC>
__global__ __launch_bounds__(256) void argmax_dice_kernel(const float* __restrict__ prob, int64_t sb, int64_t sc, int64_t sv, const int64_t* __restrict__ target,
                                                         int64_t* __restrict__ seg, unsigned long long* __restrict__ counts, int64_t V, int64_t total) {
  __shared__ unsigned int red[4][NC];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int cnt[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) cnt[k] = 0;
  if (idx < total) {
    const int64_t b = idx / V, v = idx % V;
    const float* p = prob + b * sb + v * sv;
    int best = 0; float bv = p[0];
#pragma unroll
    for (int c = 1; c < 4; ++c) { const float q = p[c * sc]; if (q > bv) { bv = q; best = c; } }
    seg[idx] = best;
    if (target) {
      const int t = (int)target[idx];
      const bool o[6] = {best > 0, best == 1 || best == 3, best == 3, best == 1, best == 2, best == 3};
      const bool g[6] = {t > 0, t == 1 || t == 3, t == 3, t == 1, t == 2, t == 3};
#pragma unroll
      for (int k = 0; k < NC / 3; ++k) { cnt[k * 3] = o[k] && g[k]; cnt[k * 3 + 1] = o[k]; cnt[k * 3 + 2] = g[k]; }
    }
  }
  if (!target) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    unsigned int s = cnt[k];
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) s += __shfl_xor(s, o2, 64);
    if (lane == 0) red[w][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NC) {
    const unsigned int s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (s) atomicAdd(counts + threadIdx.x, (unsigned long long)s);
  }
}

// prob: 4-class probabilities addressed as prob[b*sb + c*sc + v*sv] (NCDHW: sc = V, sv = 1; channels-last: sc = 1, sv = 4)
extern "C" int cwf_argmax_dice(const float* prob, int64_t sb, int64_t sc, int64_t sv, const int64_t* target, int64_t* seg,
                               uint64_t* counts /* [3][3], zeroed by the caller; may be NULL with target */, int B, int64_t V, void* stream) {
  if (!prob || !seg || B <= 0 || V <= 0 || (target && !counts)) return CWF_E_BADARG;
  const int64_t total = (int64_t)B * V;
  hipLaunchKernelGGL(argmax_dice_kernel<9>, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream), prob, sb, sc, sv, target, seg,
                     reinterpret_cast<unsigned long long*>(counts), V, total);
  CWF_LAUNCH_CHECK();
  return 0;
}

// The same with the per-class counts of tools.softmax_mIOU_score (utils/tools.py:50-61; predict_simple.py reports them next to Dice):
// counts [6][3] = WT, TC, ET, class 1, class 2, class 3, each (|o & t|, |o|, |t|).
extern "C" int cwf_argmax_metrics(const float* prob, int64_t sb, int64_t sc, int64_t sv, const int64_t* target, int64_t* seg,
                                  uint64_t* counts /* [6][3], zeroed by the caller */, int B, int64_t V, void* stream) {
  if (!prob || !seg || !target || !counts || B <= 0 || V <= 0) return CWF_E_BADARG;
  const int64_t total = (int64_t)B * V;
  hipLaunchKernelGGL(argmax_dice_kernel<18>, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream), prob, sb, sc, sv, target, seg,
                     reinterpret_cast<unsigned long long*>(counts), V, total);
  CWF_LAUNCH_CHECK();
  return 0;
}
