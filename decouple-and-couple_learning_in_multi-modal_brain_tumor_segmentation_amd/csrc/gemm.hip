// K6/K7 -- token-path math: strided batched GEMM on v_mfma_f32_16x16x4_f32 (all Linear layers, QK^T, attn.V
// and their gradients), LayerNorm fwd/bwd, row softmax fwd/bwd, GELU backward, column sums.
// Reference: SelfAttention.py:74-102 (DualSelfAttention), ResidualNorm.py:4-47, ClsWiseTransformer.py:41-55,
// FusionClsWiseTransformer.py:43-54.  Sequence length is 129 / 258 tokens x 512: launch-latency bound, so the
// kernels are kept simple (64x64 or 32x32 workgroup tiles, K steps of 32, one pass).
#include "common.h"

struct GemmArgs {
  const float* A; int64_t sa_m, sa_k, sa_zb, sa_zh;
  const float* B; int64_t sb_k, sb_n, sb_zb, sb_zh;
  float* C; int64_t sc_m, sc_zb, sc_zh;
  const float* bias; const float* residual; int64_t sr_m, sr_zb, sr_zh;
  int M, N, K, ZH; float alpha; int act; int accumulate;
};

#define GK 32

// TM x TN output tile per workgroup (64 x 64: each of the 4 waves owns 2 x 2 MFMA tiles; 32 x 32: one tile per wave), K in
// steps of 32.  The next K-tile is fetched into registers BEFORE the MFMAs of the current one are issued (software
// pipelining).  These GEMMs are tiny (M = 129..516 rows): with 64 x 64 tiles most of them launch 24-48 workgroups on a
// 256-CU chip and each workgroup grinds through its K loop alone at the fp32 MFMA rate, so small problems use 32 x 32 tiles
// (4x the workgroups, a quarter of the per-step MFMA time).
template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(const GemmArgs a) {
  constexpr int LDA_S = GK + 1, LDB_S = TN + 16;
  constexpr int IM = TM / 32, JN = TN / 32;             // MFMA tiles per wave
  constexpr int SA = TM * GK / 256, SB = TN * GK / 256; // load slots per thread
  __shared__ float As[TM * LDA_S];
  __shared__ float Bs[GK * LDB_S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 15, kq = lane >> 4;
  const int zb = blockIdx.z / a.ZH, zh = blockIdx.z % a.ZH;
  const float* A = a.A + zb * a.sa_zb + zh * a.sa_zh;
  const float* B = a.B + zb * a.sb_zb + zh * a.sb_zh;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;

  f32x4 acc[IM][JN];
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool a_kfast = (a.sa_k == 1);
  const bool b_nfast = (a.sb_n == 1);
  // element (m,k) / (k,n) handled by this thread in load slot i
  int am[SA], ak[SA], bk[SB], bn[SB];
#pragma unroll
  for (int i = 0; i < SA; ++i) {
    if (a_kfast) { ak[i] = tid & 31; am[i] = (tid >> 5) + 8 * i; } else { am[i] = tid % TM; ak[i] = tid / TM + (256 / TM) * i; }
  }
#pragma unroll
  for (int i = 0; i < SB; ++i) {
    if (b_nfast) { bn[i] = tid % TN; bk[i] = tid / TN + (256 / TN) * i; } else { bk[i] = tid & 31; bn[i] = (tid >> 5) + 8 * i; }
  }
  float ra[SA], rb[SB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < SA; ++i) {
      const int gm = m0 + am[i], gk = k0 + ak[i];
      ra[i] = (gm < a.M && gk < a.K) ? A[gm * a.sa_m + gk * a.sa_k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int gn = n0 + bn[i], gkb = k0 + bk[i];
      rb[i] = (gn < a.N && gkb < a.K) ? B[gkb * a.sb_k + gn * a.sb_n] : 0.f;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < a.K; k0 += GK) {
    if (k0) __syncthreads();
#pragma unroll
    for (int i = 0; i < SA; ++i) As[am[i] * LDA_S + ak[i]] = ra[i];
#pragma unroll
    for (int i = 0; i < SB; ++i) Bs[bk[i] * LDB_S + bn[i]] = rb[i];
    __syncthreads();
    if (k0 + GK < a.K) fetch(k0 + GK);          // in flight while the MFMAs below run
#pragma unroll
    for (int kk = 0; kk < GK / 4; ++kk) {
      float av[IM], bv[JN];
#pragma unroll
      for (int i = 0; i < IM; ++i) av[i] = As[(wr * (TM / 2) + i * 16 + r) * LDA_S + kk * 4 + kq];
#pragma unroll
      for (int j = 0; j < JN; ++j) bv[j] = Bs[(kk * 4 + kq) * LDB_S + wc * (TN / 2) + j * 16 + r];
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
  float* C = a.C + zb * a.sc_zb + zh * a.sc_zh;
  const float* R = a.residual ? a.residual + zb * a.sr_zb + zh * a.sr_zh : nullptr;
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int gn = n0 + wc * (TN / 2) + j * 16 + r;
      if (gn >= a.N) continue;
      const float bv = a.bias ? a.bias[gn] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gm = m0 + wr * (TM / 2) + i * 16 + kq * 4 + e;
        if (gm >= a.M) continue;
        float v = acc[i][j][e] * a.alpha + bv;
        if (a.act == 1) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        if (R) v += R[gm * a.sr_m + gn];
        float* p = C + gm * a.sc_m + gn;
        if (a.accumulate) v += *p;
        *p = v;
      }
    }
}

extern "C" int cwf_gemm(const float* A, int64_t sa_m, int64_t sa_k, int64_t sa_zb, int64_t sa_zh,
                        const float* B, int64_t sb_k, int64_t sb_n, int64_t sb_zb, int64_t sb_zh,
                        float* C, int64_t sc_m, int64_t sc_zb, int64_t sc_zh,
                        const float* bias, const float* residual, int64_t sr_m, int64_t sr_zb, int64_t sr_zh,
                        int M, int Nn, int K, int ZB, int ZH, float alpha, int act, int accumulate, void* stream) {
  if (!A || !B || !C || M <= 0 || Nn <= 0 || K <= 0 || ZB <= 0 || ZH <= 0) return CWF_E_BADARG;
  if ((int64_t)ZB * ZH > 65535) return CWF_E_TOOLARGE;
  GemmArgs a{A, sa_m, sa_k, sa_zb, sa_zh, B, sb_k, sb_n, sb_zb, sb_zh, C, sc_m, sc_zb, sc_zh,
             bias, residual, sr_m, sr_zb, sr_zh, M, Nn, K, ZH, alpha, act, accumulate};
  const int64_t wg64 = (int64_t)cdiv(Nn, 64) * cdiv(M, 64) * ZB * ZH;
  if (wg64 >= 256) {
    dim3 grid(cdiv(Nn, 64), cdiv(M, 64), ZB * ZH);
    hipLaunchKernelGGL((gemm_mfma_kernel<64, 64>), grid, dim3(256), 0, cwf_stream(stream), a);
  } else {
    dim3 grid(cdiv(Nn, 32), cdiv(M, 32), ZB * ZH);
    hipLaunchKernelGGL((gemm_mfma_kernel<32, 32>), grid, dim3(256), 0, cwf_stream(stream), a);
  }
  CWF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// one wave per row; E <= 1024, E % 64 == 0
template <int PER>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int rows, int E, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * E;
  float v[PER]; float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) { v[i] = xr[lane + 64 * i]; s += v[i]; }
  const float mu = wave_sum(s) / (float)E;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) { const float d = v[i] - mu; q += d * d; }
  const float rs = rsqrtf(wave_sum(q) / (float)E + eps);
#pragma unroll
  for (int i = 0; i < PER; ++i) { const int c = lane + 64 * i; y[(int64_t)row * E + c] = (v[i] - mu) * rs * gamma[c] + beta[c]; }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

template <int PER>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int E, int accumulate) {
  __shared__ float sg[4 * 64 * PER], sb[4 * 64 * PER];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  float pg[PER], pb[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) { pg[i] = 0.f; pb[i] = 0.f; }
  if (row < rows) {
    const float mu = mean[row], rs = rstd[row];
    float g[PER], h[PER]; float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = lane + 64 * i;
      const float d = dy[(int64_t)row * E + c];
      h[i] = (x[(int64_t)row * E + c] - mu) * rs;
      g[i] = d * gamma[c];
      s1 += g[i]; s2 += g[i] * h[i];
      pg[i] = d * h[i]; pb[i] = d;
    }
    s1 = wave_sum(s1) / (float)E; s2 = wave_sum(s2) / (float)E;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = lane + 64 * i;
      float o = rs * (g[i] - s1 - h[i] * s2);
      if (accumulate) o += dx[(int64_t)row * E + c];
      dx[(int64_t)row * E + c] = o;
    }
  }
  if (!dgamma) return;                                 // parameter gradients come from layernorm_bwd_params_kernel (uniform)
#pragma unroll
  for (int i = 0; i < PER; ++i) { sg[(w * PER + i) * 64 + lane] = pg[i]; sb[(w * PER + i) * 64 + lane] = pb[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < E; c += 256) {
    const int i = c >> 6, l = c & 63;
    float ag = 0.f, ab = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { ag += sg[(ww * PER + i) * 64 + l]; ab += sb[(ww * PER + i) * 64 + l]; }
    atomic_add_f32(dgamma + c, ag);
    atomic_add_f32(dbeta + c, ab);
  }
}

extern "C" int cwf_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                 int rows, int E, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0) return CWF_E_BADARG;
  dim3 grid(cdiv(rows, 4));
  if (E == 512) hipLaunchKernelGGL(layernorm_fwd_kernel<8>, grid, dim3(256), 0, cwf_stream(stream), x, gamma, beta, y, mean, rstd, rows, E, eps);
  else if (E == 256) hipLaunchKernelGGL(layernorm_fwd_kernel<4>, grid, dim3(256), 0, cwf_stream(stream), x, gamma, beta, y, mean, rstd, rows, E, eps);
  else if (E == 128) hipLaunchKernelGGL(layernorm_fwd_kernel<2>, grid, dim3(256), 0, cwf_stream(stream), x, gamma, beta, y, mean, rstd, rows, E, eps);
  else if (E == 64) hipLaunchKernelGGL(layernorm_fwd_kernel<1>, grid, dim3(256), 0, cwf_stream(stream), x, gamma, beta, y, mean, rstd, rows, E, eps);
  else return CWF_E_BADARG;
  CWF_LAUNCH_CHECK();
  return 0;
}

// dgamma[c] = sum_rows dy*xhat, dbeta[c] = sum_rows dy: one block per 64 columns, rows split over the 4 waves, fixed summation
// order (deterministic), plain stores -- no zero-initialised buffers and no float atomics (the token path is launch-bound:
// this removes two fill launches per LayerNorm backward).
__global__ __launch_bounds__(256) void layernorm_bwd_params_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int E) {
  __shared__ float sg[4][64], sb[4][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * 64 + lane;
  float ag = 0.f, ab = 0.f;
  if (c < E) {
    for (int row = w; row < rows; row += 4) {
      const float d = dy[(int64_t)row * E + c];
      ag += d * (x[(int64_t)row * E + c] - mean[row]) * rstd[row];
      ab += d;
    }
  }
  sg[w][lane] = ag; sb[w][lane] = ab;
  __syncthreads();
  if (w == 0 && c < E) {
    dgamma[c] = (sg[0][lane] + sg[1][lane]) + (sg[2][lane] + sg[3][lane]);
    dbeta[c] = (sb[0][lane] + sb[1][lane]) + (sb[2][lane] + sb[3][lane]);
  }
}

extern "C" int cwf_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                 float* dx, float* dgamma, float* dbeta, int rows, int E, int accumulate, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows <= 0) return CWF_E_BADARG;
  if (E != 512 && E != 256 && E != 128 && E != 64) return CWF_E_BADARG;
  hipLaunchKernelGGL(layernorm_bwd_params_kernel, dim3(cdiv(E, 64)), dim3(256), 0, cwf_stream(stream), dy, x, mean, rstd, dgamma, dbeta, rows, E);
  dgamma = nullptr; dbeta = nullptr;                   // the row kernel below computes dx only
  dim3 grid(cdiv(rows, 4));
  if (E == 512) hipLaunchKernelGGL(layernorm_bwd_kernel<8>, grid, dim3(256), 0, cwf_stream(stream), dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, E, accumulate);
  else if (E == 256) hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, dim3(256), 0, cwf_stream(stream), dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, E, accumulate);
  else if (E == 128) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, dim3(256), 0, cwf_stream(stream), dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, E, accumulate);
  else if (E == 64) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, dim3(256), 0, cwf_stream(stream), dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, E, accumulate);
  else return CWF_E_BADARG;
  CWF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ row softmax
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int64_t rows, int cols, int ld) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* p = s + row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, p[c]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) { const float e = expf(p[c] - mx); p[c] = e; sum += e; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int c = lane; c < cols; c += 64) p[c] *= inv;
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, int64_t rows, int cols, int ld) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* pr = p + row * ld; float* dr = dp + row * ld;
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot += pr[c] * dr[c];
  dot = wave_sum(dot);
  for (int c = lane; c < cols; c += 64) dr[c] = pr[c] * (dr[c] - dot);
}

extern "C" int cwf_softmax_rows(float* s, int64_t rows, int cols, int ld, void* stream) {
  if (!s || rows <= 0 || cols <= 0 || ld < cols) return CWF_E_BADARG;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, cwf_stream(stream), s, rows, cols, ld);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_softmax_rows_bwd(const float* p, float* dp_inout, int64_t rows, int cols, int ld, void* stream) {
  if (!p || !dp_inout || rows <= 0 || cols <= 0 || ld < cols) return CWF_E_BADARG;
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, cwf_stream(stream), p, dp_inout, rows, cols, ld);
  CWF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ GELU', colsum
__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
  dx[i] = dy[i] * (cdf + v * pdf);
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t rows, int cols, int ld, float* __restrict__ out, int accumulate) {
  // block = 64 columns x 4 row-lanes
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < cols) for (int64_t r = rl; r < rows; r += 4) s += x[r * ld + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < cols) {
    const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    out[c] = accumulate ? out[c] + t : t;
  }
}

extern "C" int cwf_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  if (!x || !dy || !dx || n <= 0) return CWF_E_BADARG;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, cwf_stream(stream), x, dy, dx, n);
  CWF_LAUNCH_CHECK();
  return 0;
}
extern "C" int cwf_colsum(const float* x, int64_t rows, int cols, int ld, float* out, int accumulate, void* stream) {
  if (!x || !out || rows <= 0 || cols <= 0) return CWF_E_BADARG;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 64)), dim3(256), 0, cwf_stream(stream), x, rows, cols, ld, out, accumulate);
  CWF_LAUNCH_CHECK();
  return 0;
}
