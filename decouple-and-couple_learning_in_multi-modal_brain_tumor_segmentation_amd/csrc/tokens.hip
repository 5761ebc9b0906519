// K4/K5 -- window <-> token reshapes, token scoring, device-side top-k (no host sync), gather / scatter / gate.
// Reference: cls_wise_former.py:15-39 (convert_dim / split_dim), :345-376 + :552-560 (score, topk, index_select,
// positional constant, class-token concat), :457-543 + :565-579 (row scatter via fix_index.txt, gating).
// The reference does 7 x 128 `.item()` host syncs per forward here; everything below stays on the stream.
#include "common.h"

// tok[b][t][f], t = ((d/p0)*(H/p1) + h/p1)*(W/p2) + w/p2, f = ((c*p0 + d%p0)*p1 + h%p1)*p2 + w%p2
__global__ void window_to_tokens_kernel(const float* __restrict__ x, int x_ldc, float* __restrict__ tok,
                                        int D, int H, int W, int C, int p0, int p1, int p2, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over tok elements
  if (idx >= total) return;
  const int E = C * p0 * p1 * p2;
  const int T = (D / p0) * (H / p1) * (W / p2);
  const int f = (int)(idx % E); const int64_t bt = idx / E;
  const int t = (int)(bt % T); const int b = (int)(bt / T);
  int ff = f; const int k = ff % p2; ff /= p2; const int j = ff % p1; ff /= p1; const int i = ff % p0; const int c = ff / p0;
  int tt = t; const int tw = tt % (W / p2); tt /= (W / p2); const int th = tt % (H / p1); const int td = tt / (H / p1);
  const int d = td * p0 + i, h = th * p1 + j, w = tw * p2 + k;
  tok[idx] = x[((((int64_t)b * D + d) * H + h) * W + w) * x_ldc + c];
}

__global__ void tokens_to_window_kernel(const float* __restrict__ tok, float* __restrict__ x, int x_ldc,
                                        int D, int H, int W, int C, int p0, int p1, int p2, int accumulate, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over x elements (b,d,h,w,c)
  if (idx >= total) return;
  const int c = (int)(idx % C); int64_t v = idx / C;
  const int w = (int)(v % W); v /= W; const int h = (int)(v % H); v /= H; const int d = (int)(v % D); const int b = (int)(v / D);
  const int E = C * p0 * p1 * p2;
  const int T = (D / p0) * (H / p1) * (W / p2);
  const int t = ((d / p0) * (H / p1) + h / p1) * (W / p2) + w / p2;
  const int f = ((c * p0 + d % p0) * p1 + h % p1) * p2 + w % p2;
  const float val = tok[((int64_t)b * T + t) * E + f];
  float* dst = x + ((((int64_t)b * D + d) * H + h) * W + w) * x_ldc + c;
  *dst = accumulate ? *dst + val : val;
}

// one wave per token row
__global__ __launch_bounds__(256) void token_scores_kernel(const float* __restrict__ feats, const float* __restrict__ query, int64_t qbs,
                                                          float* __restrict__ score, int T, int E, int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int b = (int)(row / T);
  const float* f = feats + row * E; const float* q = query + b * qbs;
  float s = 0.f;
  for (int e = lane * 4; e < E; e += 256) {
    const float4 fv = *reinterpret_cast<const float4*>(f + e);
    const float4 qv = *reinterpret_cast<const float4*>(q + e);
    s += fv.x * qv.x + fv.y * qv.y + fv.z * qv.z + fv.w * qv.w;
  }
  s = wave_sum(s);
  if (lane == 0) score[row] = s;
}

// bitonic sort of (score desc, index asc) in LDS; one workgroup per sample; P = padded power of two
__global__ __launch_bounds__(1024) void topk_kernel(const float* __restrict__ score, int32_t* __restrict__ index, int T, int k, int P) {
  extern __shared__ float4 lds4[];
  float* key = reinterpret_cast<float*>(lds4);
  int* val = reinterpret_cast<int*>(key + P);
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    key[i] = i < T ? score[(int64_t)b * T + i] : -INFINITY;
    val[i] = i < T ? i : 0x7fffffff;
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < (P >> 1); i += blockDim.x) {
        const int lo = 2 * i - (i & (stride - 1));      // index with bit `stride` clear
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);           // this block sorts "best first"
        const float ka = key[lo], kb = key[hi]; const int va = val[lo], vb = val[hi];
        const bool a_first = (ka > kb) || (ka == kb && va < vb);   // a is better than b
        if (desc ? !a_first : a_first) { key[lo] = kb; key[hi] = ka; val[lo] = vb; val[hi] = va; }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < k; i += blockDim.x) index[(int64_t)b * k + i] = val[i];
}

// seq[b][0] = head ; seq[b][1+j] = (feats[b][idx[j]] + pe) * keep      one 128-thread block per output row
__global__ __launch_bounds__(128) void gather_tokens_kernel(const float* __restrict__ feats, const int32_t* __restrict__ index,
                                                           const float* __restrict__ head, int64_t hbs, const float* __restrict__ keep,
                                                           float pe_odd, float* __restrict__ seq, int T, int k, int E) {
  const int j = blockIdx.x, b = blockIdx.y;             // j in [0, k]
  float* out = seq + ((int64_t)b * (k + 1) + j) * E;
  if (j == 0) {
    for (int e = threadIdx.x * 4; e < E; e += 512) *reinterpret_cast<float4*>(out + e) = *reinterpret_cast<const float4*>(head + b * hbs + e);
    return;
  }
  const int t = index[(int64_t)b * k + (j - 1)];
  const float* src = feats + ((int64_t)b * T + t) * E;
  for (int e = threadIdx.x * 4; e < E; e += 512) {
    float4 v = *reinterpret_cast<const float4*>(src + e);
    v.y += pe_odd; v.w += pe_odd;                        // odd channels get cos(0) = 1, even sin(0) = 0
    if (keep) {
      const float4 m = *reinterpret_cast<const float4*>(keep + ((int64_t)b * k + (j - 1)) * E + e);
      v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
    }
    *reinterpret_cast<float4*>(out + e) = v;
  }
}

__global__ __launch_bounds__(128) void gather_tokens_bwd_kernel(const float* __restrict__ dseq, const int32_t* __restrict__ index,
                                                               const float* __restrict__ keep, float* __restrict__ dfeats,
                                                               float* __restrict__ dhead, int64_t dhbs, int T, int k, int E) {
  const int j = blockIdx.x, b = blockIdx.y;
  const float* src = dseq + ((int64_t)b * (k + 1) + j) * E;
  if (j == 0) {
    if (dhead) for (int e = threadIdx.x; e < E; e += 128) atomic_add_f32(dhead + b * dhbs + e, src[e]);
    return;
  }
  if (!dfeats) return;
  const int t = index[(int64_t)b * k + (j - 1)];
  float* dst = dfeats + ((int64_t)b * T + t) * E;
  for (int e = threadIdx.x * 4; e < E; e += 512) {
    float4 v = *reinterpret_cast<const float4*>(src + e);
    if (keep) {
      const float4 m = *reinterpret_cast<const float4*>(keep + ((int64_t)b * k + (j - 1)) * E + e);
      v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
    }
    float4 o = *reinterpret_cast<float4*>(dst + e);
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    *reinterpret_cast<float4*>(dst + e) = o;
  }
}

// out[b][t] = (selected(t) ? rows[b][j] : feats[b][t]) * gate[b]
__global__ __launch_bounds__(128) void scatter_rows_kernel(const float* __restrict__ feats, const int32_t* __restrict__ index,
                                                          const float* __restrict__ rows, int64_t rows_ld, int64_t rows_bs,
                                                          const float* __restrict__ gate, int64_t gate_bs, float* __restrict__ out,
                                                          int T, int k, int E) {
  __shared__ int sel;
  const int t = blockIdx.x, b = blockIdx.y;
  if (threadIdx.x == 0) sel = -1;
  __syncthreads();
  for (int j = threadIdx.x; j < k; j += 128) if (index[(int64_t)b * k + j] == t) sel = j;
  __syncthreads();
  const int j = sel;
  const float* src = j >= 0 ? rows + b * rows_bs + (int64_t)j * rows_ld : feats + ((int64_t)b * T + t) * E;
  float* dst = out + ((int64_t)b * T + t) * E;
  for (int e = threadIdx.x * 4; e < E; e += 512) {
    float4 v = *reinterpret_cast<const float4*>(src + e);
    if (gate) {
      const float4 g = *reinterpret_cast<const float4*>(gate + b * gate_bs + e);
      v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
    }
    *reinterpret_cast<float4*>(dst + e) = v;
  }
}

// blockDim 128 (each thread 4 channels); a block walks ROWS_PER_BLOCK rows so that dgate needs few atomics
#define SCB_ROWS 16
__global__ __launch_bounds__(128) void scatter_rows_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ index,
                                                              const float* __restrict__ scat, const float* __restrict__ gate, int64_t gate_bs,
                                                              float* __restrict__ dfeats, int accumulate, float* __restrict__ drows,
                                                              int64_t drows_ld, int64_t drows_bs, float* __restrict__ dgate, int64_t dgate_bs,
                                                              int T, int k, int E) {
  __shared__ int sel[SCB_ROWS];
  const int b = blockIdx.y, t0 = blockIdx.x * SCB_ROWS;
  if (threadIdx.x < SCB_ROWS) sel[threadIdx.x] = -1;
  __syncthreads();
  for (int j = threadIdx.x; j < k; j += 128) {
    const int t = index[(int64_t)b * k + j];
    if (t >= t0 && t < t0 + SCB_ROWS) sel[t - t0] = j;
  }
  __syncthreads();
  for (int e = threadIdx.x * 4; e < E; e += 512) {
    float4 gt = make_float4(1.f, 1.f, 1.f, 1.f);
    if (gate) gt = *reinterpret_cast<const float4*>(gate + b * gate_bs + e);
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int rr = 0; rr < SCB_ROWS && t0 + rr < T; ++rr) {
      const int t = t0 + rr;
      const float4 d = *reinterpret_cast<const float4*>(dout + ((int64_t)b * T + t) * E + e);
      if (dgate) {
        const float4 s = *reinterpret_cast<const float4*>(scat + ((int64_t)b * T + t) * E + e);
        dg.x += d.x * s.x; dg.y += d.y * s.y; dg.z += d.z * s.z; dg.w += d.w * s.w;
      }
      const float4 g = make_float4(d.x * gt.x, d.y * gt.y, d.z * gt.z, d.w * gt.w);
      const int j = sel[rr];
      if (j >= 0) {
        if (drows) *reinterpret_cast<float4*>(drows + b * drows_bs + (int64_t)j * drows_ld + e) = g;
        if (dfeats && !accumulate) *reinterpret_cast<float4*>(dfeats + ((int64_t)b * T + t) * E + e) = make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (dfeats) {
        float4* p = reinterpret_cast<float4*>(dfeats + ((int64_t)b * T + t) * E + e);
        if (accumulate) { float4 o = *p; o.x += g.x; o.y += g.y; o.z += g.z; o.w += g.w; *p = o; } else *p = g;
      }
    }
    if (dgate) {
      atomic_add_f32(dgate + b * dgate_bs + e + 0, dg.x); atomic_add_f32(dgate + b * dgate_bs + e + 1, dg.y);
      atomic_add_f32(dgate + b * dgate_bs + e + 2, dg.z); atomic_add_f32(dgate + b * dgate_bs + e + 3, dg.w);
    }
  }
}

extern "C" int cwf_window_to_tokens(const float* x, int x_ldc, float* tok, int B, int D, int H, int W, int C,
                                    int p0, int p1, int p2, void* stream) {
  if (!x || !tok || B <= 0 || D % p0 || H % p1 || W % p2) return CWF_E_BADARG;
  const int64_t total = (int64_t)B * D * H * W * C;
  hipLaunchKernelGGL(window_to_tokens_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream), x, x_ldc, tok, D, H, W, C, p0, p1, p2, total);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_tokens_to_window(const float* tok, float* x, int x_ldc, int B, int D, int H, int W, int C,
                                    int p0, int p1, int p2, int accumulate, void* stream) {
  if (!x || !tok || B <= 0 || D % p0 || H % p1 || W % p2) return CWF_E_BADARG;
  const int64_t total = (int64_t)B * D * H * W * C;
  hipLaunchKernelGGL(tokens_to_window_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream), tok, x, x_ldc, D, H, W, C, p0, p1, p2, accumulate, total);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_token_scores(const float* feats, const float* query, int64_t query_bstride, float* score, int B, int T, int E, void* stream) {
  if (!feats || !query || !score || B <= 0 || T <= 0 || (E & 3)) return CWF_E_BADARG;
  const int64_t rows = (int64_t)B * T;
  hipLaunchKernelGGL(token_scores_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, cwf_stream(stream), feats, query, query_bstride, score, T, E, rows);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_topk(const float* score, int32_t* index, int B, int T, int k, void* stream) {
  if (!score || !index || B <= 0 || T <= 0 || k <= 0 || k > T) return CWF_E_BADARG;
  int P = 2; while (P < T) P <<= 1;
  if (P > 16384) return CWF_E_TOOLARGE;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(1024), (size_t)P * 8, cwf_stream(stream), score, index, T, k, P);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_gather_tokens(const float* feats, const int32_t* index, const float* head, int64_t head_bstride,
                                 const float* keep, float pe_odd, float* seq, int B, int T, int k, int E, void* stream) {
  if (!feats || !index || !head || !seq || B <= 0 || k <= 0 || (E & 3)) return CWF_E_BADARG;
  hipLaunchKernelGGL(gather_tokens_kernel, dim3(k + 1, B), dim3(128), 0, cwf_stream(stream), feats, index, head, head_bstride, keep, pe_odd, seq, T, k, E);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_gather_tokens_bwd(const float* dseq, const int32_t* index, const float* keep, float* dfeats, float* dhead,
                                     int64_t dhead_bstride, int B, int T, int k, int E, void* stream) {
  if (!dseq || !index || B <= 0 || k <= 0 || (E & 3)) return CWF_E_BADARG;
  hipLaunchKernelGGL(gather_tokens_bwd_kernel, dim3(k + 1, B), dim3(128), 0, cwf_stream(stream), dseq, index, keep, dfeats, dhead, dhead_bstride, T, k, E);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_scatter_rows(const float* feats, const int32_t* index, const float* rows, int64_t rows_ld, int64_t rows_bs,
                                const float* gate, int64_t gate_bs, float* out, int B, int T, int k, int E, void* stream) {
  if (!feats || !index || !rows || !out || B <= 0 || (E & 3) || (rows_ld & 3) || (rows_bs & 3)) return CWF_E_BADARG;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(T, B), dim3(128), 0, cwf_stream(stream), feats, index, rows, rows_ld, rows_bs, gate, gate_bs, out, T, k, E);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_scatter_rows_bwd(const float* dout, const int32_t* index, const float* scat, const float* gate, int64_t gate_bs,
                                    float* dfeats, int accumulate, float* drows, int64_t drows_ld, int64_t drows_bs,
                                    float* dgate, int64_t dgate_bs, int B, int T, int k, int E, void* stream) {
  if (!dout || !index || B <= 0 || (E & 3) || (dgate && !scat)) return CWF_E_BADARG;
  hipLaunchKernelGGL(scatter_rows_bwd_kernel, dim3(cdiv(T, SCB_ROWS), B), dim3(128), 0, cwf_stream(stream), dout, index, scat, gate, gate_bs,
                     dfeats, accumulate, drows, drows_ld, drows_bs, dgate, dgate_bs, T, k, E);
  CWF_LAUNCH_CHECK();
  return 0;
}
