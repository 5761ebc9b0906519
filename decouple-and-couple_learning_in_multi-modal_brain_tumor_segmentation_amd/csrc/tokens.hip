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
    // NaN scores (a diverged step) would make the comparator inconsistent and could sort padding sentinels into the first k:
    // order NaN as the largest value, like torch.topk
    float s = i < T ? score[(int64_t)b * T + i] : -INFINITY;
    if (s != s) s = INFINITY;
    key[i] = s;
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
  const int t = min(max(index[(int64_t)b * k + (j - 1)], 0), T - 1);     // never index outside feats, whatever the index holds
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
  const int t = min(max(index[(int64_t)b * k + (j - 1)], 0), T - 1);
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
  CWF_MAX_LDS_ONCE((&topk_kernel));
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

// =====================================================================================================================
// Round-2 region-coupler kernels: every stage of a sub-region's token selection / scatter is ONE launch, gradients are
// WRITTEN (never accumulated into zero-filled buffers), dropout masks are recomputed from the element index.
//   forward : scores2 (both class tokens against one token matrix) -> topk_inv (two selections, plus the inverse map
//             inv[b][t] = position of token t in the selection or -1) -> gather_multi (the four 129-token sequences of the
//             Edge-supported Intra-region Coupler straight into the paired [B][2][129][E] operands) -> ... -> scatter_inv
//   backward: scatter_bwd (row gradients + gate gradient written into the transformer's output gradient) -> ... ->
//             token_grad (scatter pass-through + the two gather adjoints in one pass over the token matrix)
// Reference: cls_wise_former.py:345-376 (selection), :457-543 (scatter + gate), :552-579 (fusion).
// =====================================================================================================================
__global__ __launch_bounds__(256) void token_scores2_kernel(const float* __restrict__ feats, const float* __restrict__ q1, int64_t q1bs,
                                                           const float* __restrict__ q2, int64_t q2bs, float* __restrict__ s1,
                                                           float* __restrict__ s2, int T, int E, int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int b = (int)(row / T);
  const float* f = feats + row * E; const float* qa = q1 + b * q1bs; const float* qb = q2 ? q2 + b * q2bs : nullptr;
  float a = 0.f, c = 0.f;
  for (int e = lane * 4; e < E; e += 256) {
    const float4 fv = *reinterpret_cast<const float4*>(f + e);
    const float4 u = *reinterpret_cast<const float4*>(qa + e);
    a += fv.x * u.x + fv.y * u.y + fv.z * u.z + fv.w * u.w;
    if (qb) { const float4 w = *reinterpret_cast<const float4*>(qb + e); c += fv.x * w.x + fv.y * w.y + fv.z * w.z + fv.w * w.w; }
  }
  a = wave_sum(a);
  if (qb) c = wave_sum(c);
  if (lane == 0) { s1[row] = a; if (qb) s2[row] = c; }
}

struct TopkJob { const float* score; int32_t* index; int32_t* inv; };

// the bitonic top-k of topk_kernel for up to two score vectors per launch (grid (B, njobs)), also writing the inverse map
__global__ __launch_bounds__(1024) void topk_inv_kernel(const TopkJob j0, const TopkJob j1, int T, int k, int P) {
  extern __shared__ float4 lds4[];
  float* key = reinterpret_cast<float*>(lds4);
  int* val = reinterpret_cast<int*>(key + P);
  const TopkJob job = blockIdx.y ? j1 : j0;
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    float s = i < T ? job.score[(int64_t)b * T + i] : -INFINITY;
    if (s != s) s = INFINITY;                             // NaN orders as the largest value (torch.topk)
    key[i] = s;
    val[i] = i < T ? i : 0x7fffffff;
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < (P >> 1); i += blockDim.x) {
        const int lo = 2 * i - (i & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const float ka = key[lo], kb = key[hi]; const int va = val[lo], vb = val[hi];
        const bool a_first = (ka > kb) || (ka == kb && va < vb);
        if (desc ? !a_first : a_first) { key[lo] = kb; key[hi] = ka; val[lo] = vb; val[hi] = va; }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    const int t = val[i];
    if (i < k) job.index[(int64_t)b * k + i] = t < T ? t : 0;
    if (t < T && job.inv) job.inv[(int64_t)b * T + t] = i < k ? i : -1;
  }
}

// inverse map of a given (teacher-forced) index set
__global__ __launch_bounds__(256) void index_inv_kernel(const int32_t* __restrict__ index, int32_t* __restrict__ inv, int T, int k) {
  const int b = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += blockDim.x) inv[(int64_t)b * T + t] = -1;
  __syncthreads();
  for (int j = threadIdx.x; j < k; j += blockDim.x) {
    const int t = index[(int64_t)b * k + j];
    if (t >= 0 && t < T) inv[(int64_t)b * T + t] = j;
  }
}

struct GatherJob { const float* feats; const int32_t* index; const float* head; float* out; int64_t hbs, obs; int T; uint64_t drop_off;
                   const float* head_g[4]; int group_B; };
struct GatherArgs { GatherJob job[4]; int k, E; float pe_odd, p; const uint64_t* rng; };

// out[b][0] = head ; out[b][1+j] = (feats[b][index[j]] + pe) * keep((b*k + j)*E + e)     grid (k+1, B, njobs), 128 threads
__global__ __launch_bounds__(128) void gather_multi_kernel(const GatherArgs a) {
  const GatherJob& g = a.job[blockIdx.z];
  const int j = blockIdx.x, b = blockIdx.y, E = a.E, k = a.k;
  float* out = g.out + b * g.obs + (int64_t)j * E;
  if (j == 0) {
    const float* hd = g.group_B > 0 ? g.head_g[b / g.group_B] : g.head + b * g.hbs;
    for (int e = threadIdx.x * 4; e < E; e += 512) *reinterpret_cast<float4*>(out + e) = *reinterpret_cast<const float4*>(hd + e);
    return;
  }
  const int t = min(max(g.index[(int64_t)b * k + (j - 1)], 0), g.T - 1);
  const float* src = g.feats + ((int64_t)b * g.T + t) * E;
  const uint64_t n = (uint64_t)gridDim.y * k * E;
  for (int e = threadIdx.x * 4; e < E; e += 512) {
    float4 v = *reinterpret_cast<const float4*>(src + e);
    v.y += a.pe_odd; v.w += a.pe_odd;
    if (a.p > 0.f) {
      const uint64_t i = ((uint64_t)b * k + (j - 1)) * E + e;
      v.x *= cwf_keep(a.rng, g.drop_off, i, n, a.p, 0.f); v.y *= cwf_keep(a.rng, g.drop_off, i + 1, n, a.p, 0.f);
      v.z *= cwf_keep(a.rng, g.drop_off, i + 2, n, a.p, 0.f); v.w *= cwf_keep(a.rng, g.drop_off, i + 3, n, a.p, 0.f);
    }
    *reinterpret_cast<float4*>(out + e) = v;
  }
}

// scat[b][t] = inv[b][t] >= 0 ? rows[b][inv] : feats[b][t] ; gated = scat * gate[b].  Either output may be NULL.  grid (T, B)
__global__ __launch_bounds__(128) void scatter_inv_kernel(const float* __restrict__ feats, const int32_t* __restrict__ inv,
                                                         const float* __restrict__ rows, int64_t rows_ld, int64_t rows_bs,
                                                         const float* __restrict__ gate, int64_t gate_bs, float* __restrict__ gated,
                                                         float* __restrict__ scat, int T, int E) {
  const int t = blockIdx.x, b = blockIdx.y;
  const int j = inv[(int64_t)b * T + t];
  const float* src = j >= 0 ? rows + b * rows_bs + (int64_t)j * rows_ld : feats + ((int64_t)b * T + t) * E;
  const int64_t o = ((int64_t)b * T + t) * E;
  for (int e = threadIdx.x * 4; e < E; e += 512) {
    float4 v = *reinterpret_cast<const float4*>(src + e);
    if (scat) *reinterpret_cast<float4*>(scat + o + e) = v;
    if (gated) {
      const float4 g = *reinterpret_cast<const float4*>(gate + b * gate_bs + e);
      v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
      *reinterpret_cast<float4*>(gated + o + e) = v;
    }
  }
}

struct ScatBwdArgs {
  const float* dgated; const float* dscat; const float* feats; const int32_t* inv; const int32_t* index;
  const float* rows; int64_t rows_ld, rows_bs; const float* gate; int64_t gate_bs; const float* dgate_extra; int64_t extra_bs;
  float* drows; int64_t drows_ld, drows_bs; float* dgate; int64_t dgate_bs; int T, k, E;
};
// grid (k + E/64, B), 1024 threads.  Blocks [0, k): drows[b][j] = dgated[b][index[j]] * gate[b] (+ dscat[b][index[j]]).
// Blocks [k, k + E/64): dgate[b][c] = sum_t dgated[b][t][c] * scat[b][t][c] (+ dgate_extra[b][c]); scat is re-derived from
// (feats, inv, rows); 64 columns x 16 row lanes, four rows in flight, fixed order -- written, no atomics, no zero fill.
__global__ __launch_bounds__(1024) void scatter_bwd_kernel(const ScatBwdArgs a) {
  const int b = blockIdx.y, E = a.E, T = a.T;
  if ((int)blockIdx.x < a.k) {
    if (threadIdx.x >= 128) return;
    const int j = blockIdx.x;
    const int t = min(max(a.index[(int64_t)b * a.k + j], 0), T - 1);
    const int64_t o = ((int64_t)b * T + t) * E;
    for (int e = threadIdx.x * 4; e < E; e += 512) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.dgated) {
        const float4 d = *reinterpret_cast<const float4*>(a.dgated + o + e);
        const float4 g = *reinterpret_cast<const float4*>(a.gate + b * a.gate_bs + e);
        v = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
      }
      if (a.dscat) { const float4 s = *reinterpret_cast<const float4*>(a.dscat + o + e); v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; }
      *reinterpret_cast<float4*>(a.drows + b * a.drows_bs + (int64_t)j * a.drows_ld + e) = v;
    }
    return;
  }
  __shared__ float red[16][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = (blockIdx.x - a.k) * 64 + lane;
  float acc = 0.f;
  if (c < E && a.dgated) {
    for (int t0 = w; t0 < T; t0 += 64) {
      float d[4], s[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + 16 * u;
        d[u] = 0.f; s[u] = 0.f;
        if (t < T) {
          const int j = a.inv[(int64_t)b * T + t];
          d[u] = a.dgated[((int64_t)b * T + t) * E + c];
          s[u] = j >= 0 ? a.rows[b * a.rows_bs + (int64_t)j * a.rows_ld + c] : a.feats[((int64_t)b * T + t) * E + c];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += d[u] * s[u];
    }
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0 && c < E) {
    float tsum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tsum += red[i][lane];
    if (a.dgate_extra) tsum += a.dgate_extra[b * a.extra_bs + c];
    a.dgate[b * a.dgate_bs + c] = tsum;
  }
}

struct TokGradArgs {
  const float* dgated; const float* dscat; const float* gate; int64_t gate_bs;
  const int32_t* inv_p; const int32_t* inv_q;
  const float* dseq_p; int64_t dseq_p_bs; const float* dseq_q; int64_t dseq_q_bs;
  uint64_t drop_off_p, drop_off_q; float p; const uint64_t* rng;
  float* dfeats; int B, T, k, E;
};
// dfeats[b][t] = [t not selected by p] (dgated[b][t] * gate[b] + dscat[b][t])
//              + [p selects t at j] dseq_p[b][1+j] * keep_p + [q selects t at j] dseq_q[b][1+j] * keep_q            grid (T, B)
// (p = the primary selection, which is also the scatter index; q = the supplementary selection of the same token matrix)
__global__ __launch_bounds__(128) void token_grad_kernel(const TokGradArgs a) {
  const int t = blockIdx.x, b = blockIdx.y, E = a.E, T = a.T, k = a.k;
  const int jp = a.inv_p[(int64_t)b * T + t];
  const int jq = a.inv_q ? a.inv_q[(int64_t)b * T + t] : -1;
  const int64_t o = ((int64_t)b * T + t) * E;
  const uint64_t n = (uint64_t)a.B * k * E;
  for (int e = threadIdx.x * 4; e < E; e += 512) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (jp < 0) {
      if (a.dgated) {
        const float4 d = *reinterpret_cast<const float4*>(a.dgated + o + e);
        const float4 g = *reinterpret_cast<const float4*>(a.gate + b * a.gate_bs + e);
        v = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
      }
      if (a.dscat) { const float4 s = *reinterpret_cast<const float4*>(a.dscat + o + e); v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; }
    } else {
      float4 d = *reinterpret_cast<const float4*>(a.dseq_p + b * a.dseq_p_bs + (int64_t)(1 + jp) * E + e);
      if (a.p > 0.f) {
        const uint64_t i = ((uint64_t)b * k + jp) * E + e;
        d.x *= cwf_keep(a.rng, a.drop_off_p, i, n, a.p, 0.f); d.y *= cwf_keep(a.rng, a.drop_off_p, i + 1, n, a.p, 0.f);
        d.z *= cwf_keep(a.rng, a.drop_off_p, i + 2, n, a.p, 0.f); d.w *= cwf_keep(a.rng, a.drop_off_p, i + 3, n, a.p, 0.f);
      }
      v = d;
    }
    if (jq >= 0) {
      float4 d = *reinterpret_cast<const float4*>(a.dseq_q + b * a.dseq_q_bs + (int64_t)(1 + jq) * E + e);
      if (a.p > 0.f) {
        const uint64_t i = ((uint64_t)b * k + jq) * E + e;
        d.x *= cwf_keep(a.rng, a.drop_off_q, i, n, a.p, 0.f); d.y *= cwf_keep(a.rng, a.drop_off_q, i + 1, n, a.p, 0.f);
        d.z *= cwf_keep(a.rng, a.drop_off_q, i + 2, n, a.p, 0.f); d.w *= cwf_keep(a.rng, a.drop_off_q, i + 3, n, a.p, 0.f);
      }
      v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
    }
    *reinterpret_cast<float4*>(a.dfeats + o + e) = v;
  }
}

// out1[e] = sum_b (a1[b][e] + c1[b][e]) ; out2[e] = sum_b (a2[b][e] + c2[b][e])   (class-token gradients: the token heads two of
// the four sequences of a region, cls_wise_former.py:347,354,364,373)     grid (E/256, 2)
__global__ void head_grad_kernel(const float* a1, const float* c1, const float* a2, const float* c2, int64_t bs,
                                 float* out1, float* out2, int B, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float* pa = blockIdx.y ? a2 : a1; const float* pc = blockIdx.y ? c2 : c1;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += pa[b * bs + e] + pc[b * bs + e];
  (blockIdx.y ? out2 : out1)[e] = s;
}

extern "C" int cwf_token_scores2(const float* feats, const float* q1, int64_t q1_bstride, const float* q2, int64_t q2_bstride,
                                 float* s1, float* s2, int B, int T, int E, void* stream) {
  if (!feats || !q1 || !s1 || (q2 && !s2) || B <= 0 || T <= 0 || (E & 3)) return CWF_E_BADARG;
  const int64_t rows = (int64_t)B * T;
  hipLaunchKernelGGL(token_scores2_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, cwf_stream(stream), feats, q1, q1_bstride, q2, q2_bstride, s1, s2, T, E, rows);
  CWF_LAUNCH_CHECK();
  return 0;
}

static void tok_set_lds(const void* fn) {
  static bool done[64] = {};
  int dev = 0; (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !done[dev]) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (dev >= 0 && dev < 64) done[dev] = true;
  }
}

extern "C" int cwf_topk_inv(const float* score0, int32_t* index0, int32_t* inv0, const float* score1, int32_t* index1, int32_t* inv1,
                            int B, int T, int k, void* stream) {
  if (!score0 || !index0 || B <= 0 || T <= 0 || k <= 0 || k > T || (score1 && !index1)) return CWF_E_BADARG;
  int P = 2; while (P < T) P <<= 1;
  if (P > 16384) return CWF_E_TOOLARGE;
  tok_set_lds(reinterpret_cast<const void*>(&topk_inv_kernel));
  TopkJob j0{score0, index0, inv0}, j1{score1, index1, inv1};
  hipLaunchKernelGGL(topk_inv_kernel, dim3(B, score1 ? 2 : 1), dim3(1024), (size_t)P * 8, cwf_stream(stream), j0, j1, T, k, P);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_index_inv(const int32_t* index, int32_t* inv, int B, int T, int k, void* stream) {
  if (!index || !inv || B <= 0 || T <= 0 || k <= 0) return CWF_E_BADARG;
  hipLaunchKernelGGL(index_inv_kernel, dim3(B), dim3(256), 0, cwf_stream(stream), index, inv, T, k);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_gather_multi(const struct cwf_gather_job* jobs, int njobs, int B, int k, int E, float pe_odd,
                                const uint64_t* rng, float p, void* stream) {
  if (!jobs || njobs <= 0 || njobs > 4 || B <= 0 || k <= 0 || (E & 3) || (p > 0.f && !rng) || p < 0.f || p >= 1.f) return CWF_E_BADARG;
  GatherArgs a = {};
  for (int i = 0; i < njobs; ++i) {
    if (!jobs[i].feats || !jobs[i].index || (!jobs[i].head && jobs[i].group_B <= 0) || !jobs[i].out || jobs[i].T <= 0) return CWF_E_BADARG;
    a.job[i] = GatherJob{jobs[i].feats, jobs[i].index, jobs[i].head, jobs[i].out, jobs[i].head_bstride, jobs[i].out_bstride, jobs[i].T, jobs[i].drop_off,
                         {jobs[i].head_g[0], jobs[i].head_g[1], jobs[i].head_g[2], jobs[i].head_g[3]}, jobs[i].group_B};
    if (jobs[i].group_B > 0) {
      if ((B + jobs[i].group_B - 1) / jobs[i].group_B > 4) return CWF_E_BADARG;
      for (int gI = 0; gI < (B + jobs[i].group_B - 1) / jobs[i].group_B; ++gI) if (!jobs[i].head_g[gI]) return CWF_E_BADARG;
    }
  }
  a.k = k; a.E = E; a.pe_odd = pe_odd; a.p = p; a.rng = rng;
  hipLaunchKernelGGL(gather_multi_kernel, dim3(k + 1, B, njobs), dim3(128), 0, cwf_stream(stream), a);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_scatter_inv(const float* feats, const int32_t* inv, const float* rows, int64_t rows_ld, int64_t rows_bs,
                               const float* gate, int64_t gate_bs, float* gated, float* scat, int B, int T, int E, void* stream) {
  if (!feats || !inv || !rows || (!gated && !scat) || (gated && !gate) || B <= 0 || T <= 0 || (E & 3) || (rows_ld & 3) || (rows_bs & 3) || (gate_bs & 3))
    return CWF_E_BADARG;
  hipLaunchKernelGGL(scatter_inv_kernel, dim3(T, B), dim3(128), 0, cwf_stream(stream), feats, inv, rows, rows_ld, rows_bs, gate, gate_bs, gated, scat, T, E);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_scatter_bwd(const float* dgated, const float* dscat, const float* feats, const int32_t* inv, const int32_t* index,
                               const float* rows, int64_t rows_ld, int64_t rows_bs, const float* gate, int64_t gate_bs,
                               const float* dgate_extra, int64_t extra_bs, float* drows, int64_t drows_ld, int64_t drows_bs,
                               float* dgate, int64_t dgate_bs, int B, int T, int k, int E, void* stream) {
  if (!feats || !inv || !index || !rows || !drows || !dgate || B <= 0 || T <= 0 || k <= 0 || (E & 63)) return CWF_E_BADARG;
  if (dgated && !gate) return CWF_E_BADARG;
  if ((rows_ld | rows_bs | gate_bs | drows_ld | drows_bs) & 3) return CWF_E_BADARG;
  ScatBwdArgs a{dgated, dscat, feats, inv, index, rows, rows_ld, rows_bs, gate, gate_bs, dgate_extra, extra_bs, drows, drows_ld, drows_bs,
                dgate, dgate_bs, T, k, E};
  hipLaunchKernelGGL(scatter_bwd_kernel, dim3(k + E / 64, B), dim3(1024), 0, cwf_stream(stream), a);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_token_grad(const float* dgated, const float* dscat, const float* gate, int64_t gate_bs,
                              const int32_t* inv_p, const int32_t* inv_q, const float* dseq_p, int64_t dseq_p_bs,
                              const float* dseq_q, int64_t dseq_q_bs, const uint64_t* rng, uint64_t drop_off_p, uint64_t drop_off_q, float p,
                              float* dfeats, int B, int T, int k, int E, void* stream) {
  if (!inv_p || !dseq_p || !dfeats || B <= 0 || T <= 0 || k <= 0 || (E & 3) || (inv_q && !dseq_q) || (dgated && !gate)) return CWF_E_BADARG;
  if ((p > 0.f && !rng) || p < 0.f || p >= 1.f || ((gate_bs | dseq_p_bs | dseq_q_bs) & 3)) return CWF_E_BADARG;
  TokGradArgs a{dgated, dscat, gate, gate_bs, inv_p, inv_q, dseq_p, dseq_p_bs, dseq_q, dseq_q_bs, drop_off_p, drop_off_q, p, rng, dfeats, B, T, k, E};
  hipLaunchKernelGGL(token_grad_kernel, dim3(T, B), dim3(128), 0, cwf_stream(stream), a);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_head_grad(const float* a1, const float* c1, const float* a2, const float* c2, int64_t bstride,
                             float* out1, float* out2, int B, int E, void* stream) {
  if (!a1 || !c1 || !out1 || B <= 0 || E <= 0 || (a2 && (!c2 || !out2))) return CWF_E_BADARG;
  hipLaunchKernelGGL(head_grad_kernel, dim3(cdiv(E, 256), a2 ? 2 : 1), dim3(256), 0, cwf_stream(stream), a1, c1, a2, c2, bstride, out1, out2, B, E);
  CWF_LAUNCH_CHECK();
  return 0;
}

// ---- grouped forms: the three sub-regions of a batch in one launch (sample b' = g * group_B + b) --------------------------------
struct QTab { const float* q1[4]; const float* q2[4]; };
__global__ __launch_bounds__(256) void token_scores2g_kernel(const float* __restrict__ feats, const QTab q, int group_B, float* __restrict__ s1,
                                                            float* __restrict__ s2, int T, int E, int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int g = (int)(row / T) / group_B;
  const float* f = feats + row * E; const float* qa = q.q1[g]; const float* qb = q.q2[g];
  float a = 0.f, c = 0.f;
  for (int e = lane * 4; e < E; e += 256) {
    const float4 fv = *reinterpret_cast<const float4*>(f + e);
    const float4 u = *reinterpret_cast<const float4*>(qa + e);
    a += fv.x * u.x + fv.y * u.y + fv.z * u.z + fv.w * u.w;
    if (qb) { const float4 w = *reinterpret_cast<const float4*>(qb + e); c += fv.x * w.x + fv.y * w.y + fv.z * w.z + fv.w * w.w; }
  }
  a = wave_sum(a);
  if (qb) c = wave_sum(c);
  if (lane == 0) { s1[row] = a; if (qb) s2[row] = c; }
}

extern "C" int cwf_token_scores2_g(const float* feats, const float* const* h_q1, const float* const* h_q2, int groups, int group_B,
                                   float* s1, float* s2, int B, int T, int E, void* stream) {
  if (!feats || !h_q1 || !s1 || groups <= 0 || groups > 4 || group_B <= 0 || B != groups * group_B || T <= 0 || (E & 3) || (h_q2 && !s2)) return CWF_E_BADARG;
  QTab q = {};
  for (int g = 0; g < groups; ++g) {
    if (!h_q1[g] || (h_q2 && !h_q2[g])) return CWF_E_BADARG;
    q.q1[g] = h_q1[g]; q.q2[g] = h_q2 ? h_q2[g] : nullptr;
  }
  const int64_t rows = (int64_t)B * T;
  hipLaunchKernelGGL(token_scores2g_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, cwf_stream(stream), feats, q, group_B, s1, s2, T, E, rows);
  CWF_LAUNCH_CHECK();
  return 0;
}

struct HeadGradTab { float* o1[4]; float* o2[4]; };
__global__ void head_grad_g_kernel(const float* a1, const float* c1, const float* a2, const float* c2, int64_t bs, const HeadGradTab t, int group_B, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int g = blockIdx.z;
  const float* pa = blockIdx.y ? a2 : a1; const float* pc = blockIdx.y ? c2 : c1;
  float s = 0.f;
  for (int b = g * group_B; b < (g + 1) * group_B; ++b) s += pa[b * bs + e] + pc[b * bs + e];
  (blockIdx.y ? t.o2[g] : t.o1[g])[e] = s;
}
extern "C" int cwf_head_grad_g(const float* a1, const float* c1, const float* a2, const float* c2, int64_t bstride,
                               float* const* h_out1, float* const* h_out2, int groups, int group_B, int E, void* stream) {
  if (!a1 || !c1 || !a2 || !c2 || !h_out1 || !h_out2 || groups <= 0 || groups > 4 || group_B <= 0 || E <= 0) return CWF_E_BADARG;
  HeadGradTab t = {};
  for (int g = 0; g < groups; ++g) { if (!h_out1[g] || !h_out2[g]) return CWF_E_BADARG; t.o1[g] = h_out1[g]; t.o2[g] = h_out2[g]; }
  hipLaunchKernelGGL(head_grad_g_kernel, dim3(cdiv(E, 256), 2, groups), dim3(256), 0, cwf_stream(stream), a1, c1, a2, c2, bstride, t, group_B, E);
  CWF_LAUNCH_CHECK();
  return 0;
}

__global__ void window_to_tokens_g_kernel(const float* __restrict__ x, int x_ldc, float* __restrict__ tok, int B,
                                          int D, int H, int W, int C, int p0, int p1, int p2, int64_t per_group, int64_t total) {
  const int64_t idx0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over tok elements of all groups
  if (idx0 >= total) return;
  const int g = (int)(idx0 / per_group); const int64_t idx = idx0 % per_group;
  const int E = C * p0 * p1 * p2;
  const int T = (D / p0) * (H / p1) * (W / p2);
  const int f = (int)(idx % E); const int64_t bt = idx / E;
  const int t = (int)(bt % T); const int b = (int)(bt / T);
  int ff = f; const int k = ff % p2; ff /= p2; const int j = ff % p1; ff /= p1; const int i = ff % p0; const int c = ff / p0;
  int tt = t; const int tw = tt % (W / p2); tt /= (W / p2); const int th = tt % (H / p1); const int td = tt / (H / p1);
  const int d = td * p0 + i, h = th * p1 + j, w = tw * p2 + k;
  tok[idx0] = x[((((int64_t)b * D + d) * H + h) * W + w) * x_ldc + g * C + c];
}
__global__ void tokens_to_window_g_kernel(const float* __restrict__ tok, float* __restrict__ x, int x_ldc, int G, int B,
                                          int D, int H, int W, int C, int p0, int p1, int p2, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over x elements (b,d,h,w,g,c)
  if (idx >= total) return;
  const int c = (int)(idx % C); int64_t v = idx / C;
  const int g = (int)(v % G); v /= G;
  const int w = (int)(v % W); v /= W; const int h = (int)(v % H); v /= H; const int d = (int)(v % D); const int b = (int)(v / D);
  const int E = C * p0 * p1 * p2;
  const int T = (D / p0) * (H / p1) * (W / p2);
  const int t = ((d / p0) * (H / p1) + h / p1) * (W / p2) + w / p2;
  const int f = ((c * p0 + d % p0) * p1 + h % p1) * p2 + w % p2;
  x[((((int64_t)b * D + d) * H + h) * W + w) * x_ldc + g * C + c] = tok[(((int64_t)g * B + b) * T + t) * E + f];
}
extern "C" int cwf_window_to_tokens_g(const float* x, int x_ldc, float* tok, int groups, int B, int D, int H, int W, int C,
                                      int p0, int p1, int p2, void* stream) {
  if (!x || !tok || groups <= 0 || B <= 0 || D % p0 || H % p1 || W % p2 || x_ldc < groups * C) return CWF_E_BADARG;
  const int64_t per = (int64_t)B * D * H * W * C, total = per * groups;
  hipLaunchKernelGGL(window_to_tokens_g_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream), x, x_ldc, tok, B, D, H, W, C, p0, p1, p2, per, total);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_tokens_to_window_g(const float* tok, float* x, int x_ldc, int groups, int B, int D, int H, int W, int C,
                                      int p0, int p1, int p2, void* stream) {
  if (!x || !tok || groups <= 0 || B <= 0 || D % p0 || H % p1 || W % p2 || x_ldc < groups * C) return CWF_E_BADARG;
  const int64_t total = (int64_t)B * D * H * W * C * groups;
  hipLaunchKernelGGL(tokens_to_window_g_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream), tok, x, x_ldc, groups, B, D, H, W, C, p0, p1, p2, total);
  CWF_LAUNCH_CHECK();
  return 0;
}

__global__ void cat3_channels_kernel(const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ x2, float* __restrict__ y,
                                     int C, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over float4s of y
  if (idx >= total) return;
  const int CQ = C >> 2;
  const int cq = (int)(idx % CQ); int64_t v = idx / CQ;
  const int g = (int)(v % 3); v /= 3;
  const float* src = g == 0 ? x0 : (g == 1 ? x1 : x2);
  float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
  if (src) val = *reinterpret_cast<const float4*>(src + v * C + cq * 4);
  *reinterpret_cast<float4*>(y + (v * 3 + g) * C + cq * 4) = val;
}
extern "C" int cwf_cat3_channels(const float* x0, const float* x1, const float* x2, float* y, int64_t nvox, int C, void* stream) {
  if (!y || nvox <= 0 || C <= 0 || (C & 3)) return CWF_E_BADARG;
  const int64_t total = nvox * 3 * (C >> 2);
  hipLaunchKernelGGL(cat3_channels_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, cwf_stream(stream), x0, x1, x2, y, C, total);
  CWF_LAUNCH_CHECK();
  return 0;
}
