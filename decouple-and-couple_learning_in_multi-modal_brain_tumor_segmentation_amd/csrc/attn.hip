// K6 -- the 129-token multi-head attention core as ONE launch forward and ONE launch backward:
//   forward : O = dropout(softmax(Q K^T / sqrt(d))) V                      (SelfAttention.py:94-98)
//   backward: dQ, dK, dV from dO with the probabilities recomputed in LDS (nothing but q|k|v is saved)
// q, k, v live side by side in one [rows, 3E] matrix (the layout of the reference's fused qkv Linear, SelfAttention.py:80-85:
// columns [q | k | v], head-major inside each third), rows = z * T + t for sequence z.  Both contractions run on
// v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulate); the row softmax uses wavefront shuffles.  T <= 144, d = 64.
// Before this kernel the block was QK^T GEMM -> softmax -> mask mul -> PV GEMM (4 launches forward, 9 backward) with the
// [Z, heads, T, T] probabilities and a dropout mask of the same size going through HBM.
#include "common.h"

#define AT_D 64          // head dimension
#define AT_TMAX 144      // keys padded to 9 MFMA tiles
#define AT_QT 32         // query rows per tile
#define AT_LDK 68        // LDS row pitch of the [*, 64] operand tiles (floats): conflict-free for the "row = lane%16" reads
#define AT_LDS 148       // LDS row pitch of the [32, 144] score tiles

struct AttnArgs {
  const float* qkv; int64_t ld;          // [Z*T, ld], q at column h*64, k at E + h*64, v at 2E + h*64
  float* o; int64_t ldo;                 // forward output [Z*T, ldo] (column h*64)
  const float* d_o;                      // backward: dO, same layout as o
  float* dqkv;                           // backward: [Z*T, ld]
  int T, E, heads; float scale;
  const uint64_t* rng; uint64_t drop_off; float drop_p;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// cooperative load of a [nrows x 64] tile (rows >= valid -> 0) from global (row stride ld) into LDS (pitch AT_LDK)
__device__ __forceinline__ void at_load_tile(float* lds, const float* g, int64_t ld, int row0, int nrows, int valid_rows, int tid) {
  for (int i = tid; i < nrows * (AT_D / 4); i += 256) {
    const int r = i >> 4, c4 = (i & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < valid_rows) v = *reinterpret_cast<const float4*>(g + (int64_t)(row0 + r) * ld + c4);
    *reinterpret_cast<float4*>(lds + r * AT_LDK + c4) = v;
  }
}

// S[32 x 144] = A[32 x 64] . B[144 x 64]^T  (both operands row-major with pitch AT_LDK), times alpha, into Ss
__device__ __forceinline__ void at_scores(float* Ss, const float* As, const float* Bs, float alpha, int wave, int r, int kq) {
  for (int idx = wave; idx < 2 * 9; idx += 4) {
    const int mt = idx / 9, nt = idx % 9;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < AT_D / 4; ++kk)
      acc = mfma4(As[(mt * 16 + r) * AT_LDK + kk * 4 + kq], Bs[(nt * 16 + r) * AT_LDK + kk * 4 + kq], acc);
#pragma unroll
    for (int e = 0; e < 4; ++e) Ss[(mt * 16 + kq * 4 + e) * AT_LDS + nt * 16 + r] = acc[e] * alpha;
  }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
  extern __shared__ float4 at_lds4[];
  float* Ks = reinterpret_cast<float*>(at_lds4);
  float* Vs = Ks + AT_TMAX * AT_LDK;
  float* Qs = Vs + AT_TMAX * AT_LDK;
  float* Ss = Qs + AT_QT * AT_LDK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kq = lane >> 4;
  const int zh = blockIdx.y, z = zh / a.heads, h = zh % a.heads;
  const int q0 = blockIdx.x * AT_QT, T = a.T;
  const float* base = a.qkv + (int64_t)z * T * a.ld + h * AT_D;
  at_load_tile(Ks, base + a.E, a.ld, 0, AT_TMAX, T, tid);
  at_load_tile(Vs, base + 2 * a.E, a.ld, 0, AT_TMAX, T, tid);
  at_load_tile(Qs, base, a.ld, q0, AT_QT, T, tid);
  __syncthreads();
  at_scores(Ss, Qs, Ks, a.scale, wave, r, kq);
  __syncthreads();
  const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  for (int rr = 0; rr < 8; ++rr) {
    const int row = wave * 8 + rr;
    float* s = Ss + row * AT_LDS;
    float v[3]; float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int c = lane + 64 * i; v[i] = c < T ? s[c] : -INFINITY; mx = fmaxf(mx, v[i]); }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { v[i] = (lane + 64 * i) < T ? expf(v[i] - mx) : 0.f; sum += v[i]; }
    const float inv = 1.f / wave_sum(sum);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = lane + 64 * i;
      if (c < AT_TMAX) {
        float p = v[i] * inv;
        if (a.drop_p > 0.f && c < T)
          p *= cwf_rng_u01(a.rng, a.drop_off + ((uint64_t)zh * T + (q0 + row)) * T + c) >= a.drop_p ? inv_keep : 0.f;
        s[c] = p;
      }
    }
  }
  __syncthreads();
  // O[32 x 64] = P[32 x 144] . V[144 x 64]
  for (int idx = wave; idx < 2 * 4; idx += 4) {
    const int mt = idx >> 2, nt = idx & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int kk = 0; kk < AT_TMAX / 4; ++kk)
      acc = mfma4(Ss[(mt * 16 + r) * AT_LDS + kk * 4 + kq], Vs[(kk * 4 + kq) * AT_LDK + nt * 16 + r], acc);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = q0 + mt * 16 + kq * 4 + e;
      if (q < T) a.o[((int64_t)z * T + q) * a.ldo + h * AT_D + nt * 16 + r] = acc[e];
    }
  }
}

// one workgroup per (sequence, head): loops over the query tiles, dK / dV accumulate in registers (wave w owns head-dim tile w)
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnArgs a) {
  extern __shared__ float4 at_lds4[];
  float* Ks = reinterpret_cast<float*>(at_lds4);
  float* Vs = Ks + AT_TMAX * AT_LDK;
  float* Qs = Vs + AT_TMAX * AT_LDK;
  float* Gs = Qs + AT_QT * AT_LDK;             // dO tile
  float* Ss = Gs + AT_QT * AT_LDK;             // scores -> dropped probabilities
  float* Ds = Ss + AT_QT * AT_LDS;             // dP -> dS * scale
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kq = lane >> 4;
  const int zh = blockIdx.x, z = zh / a.heads, h = zh % a.heads, T = a.T;
  const float* base = a.qkv + (int64_t)z * T * a.ld + h * AT_D;
  const float* gbase = a.d_o + (int64_t)z * T * a.ldo + h * AT_D;
  float* dbase = a.dqkv + (int64_t)z * T * a.ld + h * AT_D;
  at_load_tile(Ks, base + a.E, a.ld, 0, AT_TMAX, T, tid);
  at_load_tile(Vs, base + 2 * a.E, a.ld, 0, AT_TMAX, T, tid);
  f32x4 dk[9], dv[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  for (int q0 = 0; q0 < T; q0 += AT_QT) {
    __syncthreads();                                           // previous tile fully consumed (also orders the K/V loads)
    at_load_tile(Qs, base, a.ld, q0, AT_QT, T, tid);
    at_load_tile(Gs, gbase, a.ldo, q0, AT_QT, T, tid);
    __syncthreads();
    at_scores(Ss, Qs, Ks, a.scale, wave, r, kq);               // S  = scale Q K^T
    at_scores(Ds, Gs, Vs, 1.f, wave, r, kq);                   // dP = dO V^T
    __syncthreads();
    for (int rr = 0; rr < 8; ++rr) {
      const int row = wave * 8 + rr;
      float* s = Ss + row * AT_LDS; float* d = Ds + row * AT_LDS;
      const bool live = q0 + row < T;
      float v[3], g[3]; float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 3; ++i) { const int c = lane + 64 * i; v[i] = c < T ? s[c] : -INFINITY; mx = fmaxf(mx, v[i]); }
      mx = wave_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) { v[i] = (lane + 64 * i) < T ? expf(v[i] - mx) : 0.f; sum += v[i]; }
      const float inv = 1.f / wave_sum(sum);
      float dot = 0.f; float m[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        v[i] *= inv; m[i] = 1.f;
        if (a.drop_p > 0.f && c < T)
          m[i] = cwf_rng_u01(a.rng, a.drop_off + ((uint64_t)zh * T + (q0 + row)) * T + c) >= a.drop_p ? inv_keep : 0.f;
        g[i] = c < T ? d[c] * m[i] : 0.f;                      // gradient w.r.t. the un-dropped probability
        dot += g[i] * v[i];
      }
      dot = wave_sum(dot);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        if (c < AT_TMAX) {
          s[c] = live ? v[i] * m[i] : 0.f;                     // dropped probabilities (operand of dV)
          d[c] = live ? v[i] * (g[i] - dot) * a.scale : 0.f;   // dS * scale (operand of dQ, dK)
        }
      }
    }
    __syncthreads();
    // dV[key][d] += sum_q Pd[q][key] dO[q][d] ;  dK[key][d] += sum_q dS[q][key] Q[q][d]      (wave owns d-tile `wave`)
#pragma unroll
    for (int mt = 0; mt < 9; ++mt) {
#pragma unroll
      for (int kk = 0; kk < AT_QT / 4; ++kk) {
        const int q = kk * 4 + kq;
        dv[mt] = mfma4(Ss[q * AT_LDS + mt * 16 + r], Gs[q * AT_LDK + wave * 16 + r], dv[mt]);
        dk[mt] = mfma4(Ds[q * AT_LDS + mt * 16 + r], Qs[q * AT_LDK + wave * 16 + r], dk[mt]);
      }
    }
    // dQ[32 x 64] = dS[32 x 144] . K[144 x 64]
    for (int idx = wave; idx < 2 * 4; idx += 4) {
      const int mt = idx >> 2, nt = idx & 3;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int kk = 0; kk < AT_TMAX / 4; ++kk)
        acc = mfma4(Ds[(mt * 16 + r) * AT_LDS + kk * 4 + kq], Ks[(kk * 4 + kq) * AT_LDK + nt * 16 + r], acc);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = q0 + mt * 16 + kq * 4 + e;
        if (q < T) dbase[(int64_t)q * a.ld + nt * 16 + r] = acc[e];
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < 9; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = mt * 16 + kq * 4 + e;
      if (key < T) {
        dbase[(int64_t)key * a.ld + a.E + wave * 16 + r] = dk[mt][e];
        dbase[(int64_t)key * a.ld + 2 * a.E + wave * 16 + r] = dv[mt][e];
      }
    }
}

static int at_check(const float* qkv, int64_t ld, int Z, int T, int E, int heads) {
  if (!qkv || Z <= 0 || T <= 0 || heads <= 0 || E != heads * AT_D || ld < 3 * (int64_t)E || (ld & 3) || ((uintptr_t)qkv & 15)) return CWF_E_BADARG;
  if (T > AT_TMAX || (int64_t)Z * heads > 65535) return CWF_E_TOOLARGE;
  return 0;
}

static void at_set_lds(const void* fn, size_t bytes) {
  // the opt-in for > 64 KB of dynamic LDS is per device: remember which devices have it
  static bool done[2][64] = {};
  static const void* fns[2] = {nullptr, nullptr};
  int dev = 0; (void)hipGetDevice(&dev);
  int slot = (fns[0] == fn || fns[0] == nullptr) ? 0 : 1;
  fns[slot] = fn;
  if (dev < 0 || dev >= 64 || !done[slot][dev]) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (dev >= 0 && dev < 64) done[slot][dev] = true;
  }
}

extern "C" int cwf_attn_fwd(const float* qkv, int64_t ld, float* o, int64_t ldo, int Z, int T, int E, int heads, float scale,
                            const uint64_t* rng, uint64_t drop_off, float drop_p, void* stream) {
  int rc = at_check(qkv, ld, Z, T, E, heads); if (rc) return rc;
  if (!o || ldo < E || (drop_p > 0.f && !rng) || drop_p < 0.f || drop_p >= 1.f) return CWF_E_BADARG;
  AttnArgs a{qkv, ld, o, ldo, nullptr, nullptr, T, E, heads, scale, rng, drop_off, drop_p};
  const size_t lds = sizeof(float) * (2 * AT_TMAX * AT_LDK + AT_QT * AT_LDK + AT_QT * AT_LDS);
  at_set_lds(reinterpret_cast<const void*>(&attn_fwd_kernel), lds);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(T, AT_QT), Z * heads), dim3(256), lds, cwf_stream(stream), a);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_attn_bwd(const float* qkv, int64_t ld, const float* d_o, int64_t ldo, float* dqkv, int Z, int T, int E, int heads,
                            float scale, const uint64_t* rng, uint64_t drop_off, float drop_p, void* stream) {
  int rc = at_check(qkv, ld, Z, T, E, heads); if (rc) return rc;
  if (!d_o || !dqkv || ldo < E || (ldo & 3) || ((uintptr_t)d_o & 15) || (drop_p > 0.f && !rng) || drop_p < 0.f || drop_p >= 1.f) return CWF_E_BADARG;
  AttnArgs a{qkv, ld, nullptr, ldo, d_o, dqkv, T, E, heads, scale, rng, drop_off, drop_p};
  const size_t lds = sizeof(float) * (2 * AT_TMAX * AT_LDK + 2 * AT_QT * AT_LDK + 2 * AT_QT * AT_LDS);
  at_set_lds(reinterpret_cast<const void*>(&attn_bwd_kernel), lds);
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(Z * heads), dim3(256), lds, cwf_stream(stream), a);
  CWF_LAUNCH_CHECK();
  return 0;
}
