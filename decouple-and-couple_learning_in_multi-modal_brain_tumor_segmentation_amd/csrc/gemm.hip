// K6/K7 -- token-path math: strided batched GEMM on v_mfma_f32_16x16x4_f32 (all Linear layers, QK^T, attn.V
// and their gradients), LayerNorm fwd/bwd, row softmax fwd/bwd, GELU backward, column sums.
// Reference: SelfAttention.py:74-102 (DualSelfAttention), ResidualNorm.py:4-47, ClsWiseTransformer.py:41-55,
// FusionClsWiseTransformer.py:43-54.  Sequence length is 129 / 258 tokens x 512: launch-latency bound, so the
// kernels are kept simple (64x64 or 32x32 workgroup tiles, K steps of 32, one pass).
#include "common.h"
#include <cstdlib>

typedef struct cwf_gemm_args GemmArgs;


// TM x TN output tile per workgroup (64 x 64: each of the 4 waves owns 2 x 2 MFMA tiles; 32 x 32: one tile per wave), K in
// steps of 32.  The next K-tile is fetched into registers BEFORE the MFMAs of the current one are issued (software
// pipelining).  These GEMMs are tiny (M = 129..516 rows): with 64 x 64 tiles most of them launch 24-48 workgroups on a
// 256-CU chip and each workgroup grinds through its K loop alone at the fp32 MFMA rate, so small problems use 32 x 32 tiles
// (4x the workgroups, a quarter of the per-step MFMA time).
// Fusions that keep the token path at a handful of launches (see cwf_hip.h: struct cwf_gemm_args):
//   * operand switch at a tile boundary (A2/split_n, B2/split_m): q = LN1(x) Wq^T and kv = LN2(x2) Wkv^T are ONE launch
//     over the reference's [1536, 512] qkv weight; so is its weight gradient [dq^T a ; dkv^T b];
//   * dropout of the A operand recomputed from the element index (the backward of y = x + drop(o Wo^T + b) reads dy once per
//     GEMM and never materialises dy * mask), dropout in the epilogue (forward of the same);
//   * rowsum of the (dropped) A operand from the workgroups of the first column tile = the bias gradient of a Linear,
//     computed by one extra MFMA against a ones vector;
//   * C2: the pre-activation next to the GELU output (saved for backward instead of being recomputed by a second GEMM).
// GK = K elements staged per step (32; measured: 128 for the 32 x 32 tile and 64 for the 64 x 64 tile were 28 % / 86 % SLOWER --
// the larger LDS tiles and register staging cost more occupancy than the fewer barriers save).
template <int TM, int TN, int GK>
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
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int64_t a_zoff = zb * a.sa_zb + zh * a.sa_zh;
  const float* A = ((a.A2 && n0 >= a.split_n) ? a.A2 : a.A) + a_zoff;
  const bool tabB = a.B_tab[0] != nullptr;
  const float* B = tabB ? a.B_tab[blockIdx.z] : a.B + zb * a.sb_zb + zh * a.sb_zh;
  if (a.B2 && m0 >= a.split_m) B = a.B2 + zb * a.sb_zb + zh * a.sb_zh;      // (B2 is an activation operand: strided also in the grouped form)
  const bool a_drop = a.a_drop_p > 0.f;
  float* rowsum = a.rowsum_tab[0] ? a.rowsum_tab[blockIdx.z] : a.rowsum;
  const bool do_rs = rowsum != nullptr && blockIdx.x == 0 && wc == 0;

  f32x4 acc[IM][JN], accr[IM];
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    accr[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const bool a_kfast = (a.sa_k == 1);
  const bool b_nfast = (a.sb_n == 1);
  // element (m,k) / (k,n) handled by this thread in load slot i
  int am[SA], ak[SA], bk[SB], bn[SB];
#pragma unroll
  for (int i = 0; i < SA; ++i) {
    if (a_kfast) { ak[i] = tid % GK; am[i] = tid / GK + (256 / GK) * i; } else { am[i] = tid % TM; ak[i] = tid / TM + (256 / TM) * i; }
  }
#pragma unroll
  for (int i = 0; i < SB; ++i) {
    if (b_nfast) { bn[i] = tid % TN; bk[i] = tid / TN + (256 / TN) * i; } else { bk[i] = tid % GK; bn[i] = tid / GK + (256 / GK) * i; }
  }
  float ra[SA], rb[SB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < SA; ++i) {
      const int gm = m0 + am[i], gk = k0 + ak[i];
      float v = 0.f;
      if (gm < a.M && gk < a.K) {
        const int64_t off = gm * a.sa_m + gk * a.sa_k;
        v = A[off];
        if (a_drop) v *= cwf_keep(a.rng, a.a_drop_off, (uint64_t)(a_zoff + off), a.a_drop_n, a.a_drop_p, a.a_drop_p2);
      }
      ra[i] = v;
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
      if (do_rs) {
#pragma unroll
        for (int i = 0; i < IM; ++i) accr[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], 1.0f, accr[i], 0, 0, 0);
      }
    }
  }
  const int64_t c_zoff = zb * a.sc_zb + zh * a.sc_zh;
  float* C = a.C_tab[0] ? a.C_tab[blockIdx.z] : a.C + c_zoff;
  float* C2 = a.C2 ? a.C2 + c_zoff : nullptr;
  const float* bias = a.bias_tab[0] ? a.bias_tab[blockIdx.z] : a.bias;
  const float* R = a.residual ? a.residual + zb * a.sr_zb + zh * a.sr_zh : nullptr;
  const bool c_drop = a.c_drop_p > 0.f;
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int gn = n0 + wc * (TN / 2) + j * 16 + r;
      if (gn >= a.N) continue;
      const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gm = m0 + wr * (TM / 2) + i * 16 + kq * 4 + e;
        if (gm >= a.M) continue;
        const int64_t off = gm * a.sc_m + gn;
        float v = acc[i][j][e] * a.alpha + bv;
        if (C2) C2[off] = v;
        if (a.act == 1) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        if (c_drop) v *= cwf_keep(a.rng, a.c_drop_off, (uint64_t)(c_zoff + off), a.c_drop_n, a.c_drop_p, a.c_drop_p2);
        if (R) v += R[gm * a.sr_m + gn];
        float* p = C + off;
        if (a.accumulate) v += *p;
        *p = v;
      }
    }
  if (do_rs && r == 0) {                        // every column of accr holds the row sum; column 0 writes it
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gm = m0 + wr * (TM / 2) + i * 16 + kq * 4 + e;
        if (gm < a.M) rowsum[gm] = (a.rowsum_acc ? rowsum[gm] : 0.f) + accr[i][e];
      }
  }
}

// The same kernel with 16-byte global loads and K steps of 64.  The scalar form above fetches ONE float per lane and load
// instruction (8 dword loads per thread and K step of 32): a K step is one L2 round trip long however little it moves, and a
// 512-deep contraction pays 16 of them (rocprofv3, round 3: 42 us per launch for 0.8 GFLOP).  With float4 loads along each
// operand's unit-stride dimension a thread issues the same number of load instructions for TWICE the K extent: half the round
// trips.  Eligibility (alignment of every operand pointer / stride, extents multiples of 4) is checked on the host; anything else
// takes the scalar kernel.  AK / BN: the unit-stride dimension of A is k (row-major activations, weight^T products) / of B is n.
template <int TM, int TN, int GK, bool AK, bool BN>
__global__ __launch_bounds__(256) void gemm_mfma_v_kernel(const GemmArgs a) {
  constexpr int LDA_S = GK + 1, LDB_S = TN + 16;
  constexpr int IM = TM / 32, JN = TN / 32;
  constexpr int SA = TM * GK / 1024, SB = TN * GK / 1024;      // float4 slots per thread
  __shared__ float As[TM * LDA_S];
  __shared__ __attribute__((aligned(16))) float Bs[GK * LDB_S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 15, kq = lane >> 4;
  const int zb = blockIdx.z / a.ZH, zh = blockIdx.z % a.ZH;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int64_t a_zoff = zb * a.sa_zb + zh * a.sa_zh;
  const float* A = ((a.A2 && n0 >= a.split_n) ? a.A2 : a.A) + a_zoff;
  const bool tabB = a.B_tab[0] != nullptr;
  const float* B = tabB ? a.B_tab[blockIdx.z] : a.B + zb * a.sb_zb + zh * a.sb_zh;
  if (a.B2 && m0 >= a.split_m) B = a.B2 + zb * a.sb_zb + zh * a.sb_zh;
  const bool a_drop = a.a_drop_p > 0.f;
  float* rowsum = a.rowsum_tab[0] ? a.rowsum_tab[blockIdx.z] : a.rowsum;
  const bool do_rs = rowsum != nullptr && blockIdx.x == 0 && wc == 0;

  f32x4 acc[IM][JN], accr[IM];
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    accr[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // slot i of this thread: the 4 consecutive elements starting at (am, ak) of A along k (AK) or m; at (bk, bn) of B along n (BN) or k
  int am[SA], ak[SA], bk[SB], bn[SB];
#pragma unroll
  for (int i = 0; i < SA; ++i) {
    if (AK) { ak[i] = (tid % (GK / 4)) * 4; am[i] = tid / (GK / 4) + (1024 / GK) * i; }
    else { am[i] = (tid % (TM / 4)) * 4; ak[i] = tid / (TM / 4) + (1024 / TM) * i; }
  }
#pragma unroll
  for (int i = 0; i < SB; ++i) {
    if (BN) { bn[i] = (tid % (TN / 4)) * 4; bk[i] = tid / (TN / 4) + (1024 / TN) * i; }
    else { bk[i] = (tid % (GK / 4)) * 4; bn[i] = tid / (GK / 4) + (1024 / GK) * i; }
  }
  f32x4 ra[SA], rb[SB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < SA; ++i) {
      const int gm = m0 + am[i], gk = k0 + ak[i];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gm < a.M && gk < a.K) {                          // (extents are multiples of 4 along the vector dimension: all four or none)
        const int64_t off = gm * a.sa_m + gk * a.sa_k;
        v = *reinterpret_cast<const f32x4*>(A + off);
        if (a_drop) {
          const int64_t st = AK ? 1 : a.sa_m;
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] *= cwf_keep(a.rng, a.a_drop_off, (uint64_t)(a_zoff + off + c * st), a.a_drop_n, a.a_drop_p, a.a_drop_p2);
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int gn = n0 + bn[i], gkb = k0 + bk[i];
      rb[i] = (gn < a.N && gkb < a.K) ? *reinterpret_cast<const f32x4*>(B + gkb * a.sb_k + gn * a.sb_n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < a.K; k0 += GK) {
    if (k0) __syncthreads();
#pragma unroll
    for (int i = 0; i < SA; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (AK) As[am[i] * LDA_S + ak[i] + c] = ra[i][c];
        else As[(am[i] + c) * LDA_S + ak[i]] = ra[i][c];
      }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      if (BN) *reinterpret_cast<f32x4*>(&Bs[bk[i] * LDB_S + bn[i]]) = rb[i];
      else {
#pragma unroll
        for (int c = 0; c < 4; ++c) Bs[(bk[i] + c) * LDB_S + bn[i]] = rb[i][c];
      }
    }
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
      if (do_rs) {
#pragma unroll
        for (int i = 0; i < IM; ++i) accr[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], 1.0f, accr[i], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: identical to the scalar kernel
  const int64_t c_zoff = zb * a.sc_zb + zh * a.sc_zh;
  float* C = a.C_tab[0] ? a.C_tab[blockIdx.z] : a.C + c_zoff;
  float* C2 = a.C2 ? a.C2 + c_zoff : nullptr;
  const float* bias = a.bias_tab[0] ? a.bias_tab[blockIdx.z] : a.bias;
  const float* R = a.residual ? a.residual + zb * a.sr_zb + zh * a.sr_zh : nullptr;
  const bool c_drop = a.c_drop_p > 0.f;
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int gn = n0 + wc * (TN / 2) + j * 16 + r;
      if (gn >= a.N) continue;
      const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gm = m0 + wr * (TM / 2) + i * 16 + kq * 4 + e;
        if (gm >= a.M) continue;
        const int64_t off = gm * a.sc_m + gn;
        float v = acc[i][j][e] * a.alpha + bv;
        if (C2) C2[off] = v;
        if (a.act == 1) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        if (c_drop) v *= cwf_keep(a.rng, a.c_drop_off, (uint64_t)(c_zoff + off), a.c_drop_n, a.c_drop_p, a.c_drop_p2);
        if (R) v += R[gm * a.sr_m + gn];
        float* p = C + off;
        if (a.accumulate) v += *p;
        *p = v;
      }
    }
  if (do_rs && r == 0) {
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gm = m0 + wr * (TM / 2) + i * 16 + kq * 4 + e;
        if (gm < a.M) rowsum[gm] = (a.rowsum_acc ? rowsum[gm] : 0.f) + accr[i][e];
      }
  }
}

// every pointer / stride the vector kernel dereferences with 16-byte loads
static bool gemm_vec_ok(const GemmArgs& a, bool* ak, bool* bn) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  auto m4 = [](int64_t v) { return (v & 3) == 0; };
  if (!m4(a.sa_zb) || !m4(a.sa_zh) || !m4(a.sb_zb) || !m4(a.sb_zh)) return false;
  if (!al(a.A) || (a.A2 && !al(a.A2)) || (a.B && !al(a.B)) || (a.B2 && !al(a.B2))) return false;
  for (int z = 0; z < 4; ++z)
    if (a.B_tab[z] && !al(a.B_tab[z])) return false;
  if (a.sa_k == 1 && m4(a.K) && m4(a.sa_m)) *ak = true;
  else if (a.sa_m == 1 && m4(a.M) && m4(a.sa_k)) *ak = false;
  else return false;
  if (a.sb_n == 1 && m4(a.N) && m4(a.sb_k)) *bn = true;
  else if (a.sb_k == 1 && m4(a.K) && m4(a.sb_n)) *bn = false;
  else return false;
  if (a.B2 && (a.split_m & 3)) return false;
  return true;
}

extern "C" int cwf_gemm_ex(const struct cwf_gemm_args* args, void* stream) {
  if (!args) return CWF_E_BADARG;
  const GemmArgs& a = *args;
  if (!a.A || (!a.B && !a.B_tab[0]) || (!a.C && !a.C_tab[0]) || a.M <= 0 || a.N <= 0 || a.K <= 0 || a.ZB <= 0 || a.ZH <= 0) return CWF_E_BADARG;
  if ((a.B_tab[0] || a.bias_tab[0] || a.C_tab[0] || a.rowsum_tab[0]) && (a.ZH != 1 || a.ZB > 4)) return CWF_E_BADARG;
  if ((int64_t)a.ZB * a.ZH > 65535) return CWF_E_TOOLARGE;
  if ((a.A2 && (a.split_n & 63)) || (a.B2 && (a.split_m & 63))) return CWF_E_BADARG;          // operand switch on a tile boundary
  if ((a.a_drop_p > 0.f || a.c_drop_p > 0.f) && !a.rng) return CWF_E_BADARG;
  if (a.rowsum && (int64_t)a.ZB * a.ZH != 1) return CWF_E_BADARG;
  for (int z = 0; z < a.ZB && z < 4; ++z)
    if ((a.B_tab[0] && !a.B_tab[z]) || (a.bias_tab[0] && !a.bias_tab[z]) || (a.C_tab[0] && !a.C_tab[z]) || (a.rowsum_tab[0] && !a.rowsum_tab[z])) return CWF_E_BADARG;
  const int64_t wg64 = (int64_t)cdiv(a.N, 64) * cdiv(a.M, 64) * a.ZB * a.ZH;
  static const bool no_vec = getenv("CWF_GEMM_SCALAR") != nullptr;      // A/B switch: the scalar-load kernel for everything
  bool ak = false, bn = false;
  if (!no_vec && a.K >= 64 && gemm_vec_ok(a, &ak, &bn)) {
    const bool big = wg64 >= 256;
    dim3 grid(cdiv(a.N, big ? 64 : 32), cdiv(a.M, big ? 64 : 32), a.ZB * a.ZH);
#define CWF_GV(AKv, BNv) do { if (big) hipLaunchKernelGGL((gemm_mfma_v_kernel<64, 64, 32, AKv, BNv>), grid, dim3(256), 0, cwf_stream(stream), a); \
                              else hipLaunchKernelGGL((gemm_mfma_v_kernel<32, 32, 64, AKv, BNv>), grid, dim3(256), 0, cwf_stream(stream), a); } while (0)
    if (ak && bn) CWF_GV(true, true); else if (ak) CWF_GV(true, false); else if (bn) CWF_GV(false, true); else CWF_GV(false, false);
#undef CWF_GV
    CWF_LAUNCH_CHECK();
    return 0;
  }
  if (wg64 >= 256) {
    dim3 grid(cdiv(a.N, 64), cdiv(a.M, 64), a.ZB * a.ZH);
    hipLaunchKernelGGL((gemm_mfma_kernel<64, 64, 32>), grid, dim3(256), 0, cwf_stream(stream), a);
  } else {
    dim3 grid(cdiv(a.N, 32), cdiv(a.M, 32), a.ZB * a.ZH);
    hipLaunchKernelGGL((gemm_mfma_kernel<32, 32, 32>), grid, dim3(256), 0, cwf_stream(stream), a);
  }
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_gemm(const float* A, int64_t sa_m, int64_t sa_k, int64_t sa_zb, int64_t sa_zh,
                        const float* B, int64_t sb_k, int64_t sb_n, int64_t sb_zb, int64_t sb_zh,
                        float* C, int64_t sc_m, int64_t sc_zb, int64_t sc_zh,
                        const float* bias, const float* residual, int64_t sr_m, int64_t sr_zb, int64_t sr_zh,
                        int M, int Nn, int K, int ZB, int ZH, float alpha, int act, int accumulate, void* stream) {
  GemmArgs a = {};
  a.A = A; a.sa_m = sa_m; a.sa_k = sa_k; a.sa_zb = sa_zb; a.sa_zh = sa_zh;
  a.B = B; a.sb_k = sb_k; a.sb_n = sb_n; a.sb_zb = sb_zb; a.sb_zh = sb_zh;
  a.C = C; a.sc_m = sc_m; a.sc_zb = sc_zb; a.sc_zh = sc_zh;
  a.bias = bias; a.residual = residual; a.sr_m = sr_m; a.sr_zb = sr_zb; a.sr_zh = sr_zh;
  a.M = M; a.N = Nn; a.K = K; a.ZB = ZB; a.ZH = ZH; a.alpha = alpha; a.act = act; a.accumulate = accumulate;
  return cwf_gemm_ex(&a, stream);
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

// ------------------------------------------------------------------------------------------------ paired LayerNorm
// The couplers' blocks always normalise TWO operands (PreNormDrop: norm(x), norm2(x2), ResidualNorm.py:23-32), and the four
// cross-attentions of a region run as two batches of sequence PAIRS [B][2][T][E] (ClsWiseTransformer.py:44-50: the second
// batch attends a <-> b, i.e. x2 is x with the two halves of every pair swapped).  `perm_T` > 0 expresses that swap: row r
// of the second problem reads x2[perm(r)], perm(r) = r with bit 0 of (r / perm_T) flipped.  One wave per (problem, row).
__device__ __forceinline__ int ln_perm(int r, int perm_T) { return perm_T > 0 ? (((r / perm_T) ^ 1) * perm_T + r % perm_T) : r; }

typedef struct cwf_ln_group_params LnGroups;

template <int PER>
__global__ __launch_bounds__(256) void ln_pair_fwd_kernel(const float* __restrict__ x, const float* __restrict__ x2, int perm_T,
                                                         const LnGroups P, int rpg,
                                                         float* __restrict__ ya, float* __restrict__ yb, float* __restrict__ stats,
                                                         int rows, int E, float eps) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int prob = wid / rows, row = wid % rows;
  if (prob >= (x2 ? 2 : 1)) return;
  const float* xr = prob ? x2 + (int64_t)ln_perm(row, perm_T) * E : x + (int64_t)row * E;
  const int grp = row / rpg;
  const float* gamma = prob ? P.g2[grp] : P.g1[grp]; const float* beta = prob ? P.b2[grp] : P.b1[grp];
  float* y = (prob ? yb : ya) + (int64_t)row * E;
  float v[PER]; float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) { v[i] = xr[lane + 64 * i]; s += v[i]; }
  const float mu = wave_sum(s) / (float)E;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) { const float d = v[i] - mu; q += d * d; }
  const float rs = rsqrtf(wave_sum(q) / (float)E + eps);
#pragma unroll
  for (int i = 0; i < PER; ++i) { const int c = lane + 64 * i; y[c] = (v[i] - mu) * rs * gamma[c] + beta[c]; }
  if (lane == 0) { stats[((int64_t)prob * rows + row) * 2] = mu; stats[((int64_t)prob * rows + row) * 2 + 1] = rs; }
}

// per-lane LayerNorm input gradient of one row: o[i] += rstd * (g - mean(g) - xhat * mean(g * xhat)), g = d * gamma
template <int PER>
__device__ __forceinline__ void ln_bwd_row(float* o, const float* __restrict__ d, const float* v, const float* __restrict__ gamma,
                                           float mu, float rs, int lane, int E) {
  float g[PER], h[PER]; float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    h[i] = (v[i] - mu) * rs; g[i] = d[c] * gamma[c];
    s1 += g[i]; s2 += g[i] * h[i];
  }
  s1 = wave_sum(s1) / (float)E; s2 = wave_sum(s2) / (float)E;
#pragma unroll
  for (int i = 0; i < PER; ++i) o[i] += rs * (g[i] - s1 - h[i] * s2);
}

// dual (dx2 != NULL):  dx[r] = dy[r] + LN1'(da[r]; x[r]) ;  dx2[r] = LN2'(db[r]; x2[r])                (perm_T must be 0)
// self (dx2 == NULL, db != NULL; x2 == x up to perm):  dx[r] = dy[r] + LN1'(da[r]; x[r]) + LN2'(db[perm(r)]; x[r])
// single (db == NULL):  dx[r] = dy[r] + LN1'(da[r]; x[r])
template <int PER>
__global__ __launch_bounds__(256) void ln_pair_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ da, const float* __restrict__ db,
                                                         const float* __restrict__ x, const float* __restrict__ x2, int perm_T,
                                                         const LnGroups P, int rpg, const float* __restrict__ stats,
                                                         float* __restrict__ dx, float* __restrict__ dx2, int rows, int E) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int prob = wid / rows, row = wid % rows;
  if (prob >= (dx2 ? 2 : 1)) return;
  const float* g1 = P.g1[row / rpg]; const float* g2 = P.g2[row / rpg];
  float v[PER], o[PER];
  const float* xr = (prob ? x2 : x) + (int64_t)row * E;
#pragma unroll
  for (int i = 0; i < PER; ++i) { v[i] = xr[lane + 64 * i]; o[i] = 0.f; }
  if (prob == 0) {
    if (dy) {
#pragma unroll
      for (int i = 0; i < PER; ++i) o[i] = dy[(int64_t)row * E + lane + 64 * i];
    }
    ln_bwd_row<PER>(o, da + (int64_t)row * E, v, g1, stats[(int64_t)row * 2], stats[(int64_t)row * 2 + 1], lane, E);
    if (db && !dx2) {
      const int pr = ln_perm(row, perm_T);
      ln_bwd_row<PER>(o, db + (int64_t)pr * E, v, g2, stats[((int64_t)rows + pr) * 2], stats[((int64_t)rows + pr) * 2 + 1], lane, E);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) dx[(int64_t)row * E + lane + 64 * i] = o[i];
  } else {
    ln_bwd_row<PER>(o, db + (int64_t)row * E, v, g2, stats[((int64_t)rows + row) * 2], stats[((int64_t)rows + row) * 2 + 1], lane, E);
#pragma unroll
    for (int i = 0; i < PER; ++i) dx2[(int64_t)row * E + lane + 64 * i] = o[i];
  }
}

// dgamma_p[c] (+)= sum_rows d_p * xhat_p, dbeta_p[c] (+)= sum_rows d_p for p = 1 (da, x) and p = 2 (db, x2[perm]); grid (E/64, 1 or 2),
// 1024 threads = 64 columns x 16 row lanes, four rows in flight per lane (the reduction is a chain of dependent HBM/L2 round
// trips: at 4 row lanes and one row in flight it took 30 us for 129 rows); fixed summation order, plain stores (accumulate = the
// weight-sharing sum over the uses of one LayerNorm, ClsWiseTransformer.py:44-50)
__global__ __launch_bounds__(1024) void ln_pair_params_kernel(const float* __restrict__ da, const float* __restrict__ db,
                                                             const float* __restrict__ x, const float* __restrict__ x2, int perm_T,
                                                             const float* __restrict__ stats, const LnGroups P, int rpg,
                                                             int rows_all, int E, int accumulate) {
  __shared__ float sg[16][64], sb[16][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, prob = blockIdx.y, grp = blockIdx.z;
  const int c = blockIdx.x * 64 + lane;
  const float* d = prob ? db : da; const float* xs = prob ? x2 : x;
  const float* st = stats + (int64_t)prob * rows_all * 2;
  float* dg1 = P.dg1[grp]; float* db1 = P.db1[grp]; float* dg2 = P.dg2[grp]; float* db2 = P.db2[grp];
  const int rbeg = grp * rpg, rows = rbeg + rpg;                 // this group's rows [rbeg, rows)
  float ag = 0.f, ab = 0.f;
  if (c < E) {
    for (int row0 = rbeg + w; row0 < rows; row0 += 64) {
      float dv[4], xv[4], mu[4], rs[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = row0 + 16 * u;
        const bool ok = row < rows;
        const int xr = ok ? (prob ? ln_perm(row, perm_T) : row) : 0;
        dv[u] = ok ? d[(int64_t)row * E + c] : 0.f;
        xv[u] = ok ? xs[(int64_t)xr * E + c] : 0.f;
        mu[u] = ok ? st[row * 2] : 0.f; rs[u] = ok ? st[row * 2 + 1] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { ag += dv[u] * (xv[u] - mu[u]) * rs[u]; ab += dv[u]; }
    }
  }
  sg[w][lane] = ag; sb[w][lane] = ab;
  __syncthreads();
  if (w == 0 && c < E) {
    float tg = 0.f, tb = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { tg += sg[i][lane]; tb += sb[i][lane]; }
    float* og = prob ? dg2 : dg1; float* ob = prob ? db2 : db1;
    og[c] = accumulate ? og[c] + tg : tg;
    ob[c] = accumulate ? ob[c] + tb : tb;
  }
}

#define LN_DISPATCH(KERNEL, grid, ...)                                                                                       \
  do {                                                                                                                       \
    if (E == 512) hipLaunchKernelGGL(KERNEL<8>, grid, dim3(256), 0, cwf_stream(stream), __VA_ARGS__);                        \
    else if (E == 256) hipLaunchKernelGGL(KERNEL<4>, grid, dim3(256), 0, cwf_stream(stream), __VA_ARGS__);                   \
    else if (E == 128) hipLaunchKernelGGL(KERNEL<2>, grid, dim3(256), 0, cwf_stream(stream), __VA_ARGS__);                   \
    else if (E == 64) hipLaunchKernelGGL(KERNEL<1>, grid, dim3(256), 0, cwf_stream(stream), __VA_ARGS__);                    \
    else return CWF_E_BADARG;                                                                                                \
  } while (0)

static int ln_groups_ok(const LnGroups* p, int groups, bool second, bool grads) {
  if (!p || groups <= 0 || groups > 4) return 0;
  for (int g = 0; g < groups; ++g) {
    if (!p->g1[g] || (!p->b1[g] && !grads)) return 0;
    if (second && (!p->g2[g] || (!grads && !p->b2[g]))) return 0;
    if (grads && (!p->dg1[g] || !p->db1[g] || (second && (!p->dg2[g] || !p->db2[g])))) return 0;
  }
  return 1;
}

extern "C" int cwf_ln_pair_fwd_g(const float* x, const float* x2, int perm_T, const struct cwf_ln_group_params* h_params, int groups,
                                 float* ya, float* yb, float* stats, int rows, int E, float eps, void* stream) {
  if (!x || !ya || !stats || rows <= 0 || (x2 && !yb) || !ln_groups_ok(h_params, groups, x2 != nullptr, false) || rows % groups) return CWF_E_BADARG;
  const int rpg = rows / groups;
  if (perm_T > 0 && rpg % (2 * perm_T)) return CWF_E_BADARG;
  const LnGroups P = *h_params;
  dim3 grid(cdiv(rows * (x2 ? 2 : 1), 4));
  LN_DISPATCH(ln_pair_fwd_kernel, grid, x, x2, perm_T, P, rpg, ya, yb, stats, rows, E, eps);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_ln_pair_fwd(const float* x, const float* x2, int perm_T, const float* g1, const float* b1, const float* g2, const float* b2,
                               float* ya, float* yb, float* stats, int rows, int E, float eps, void* stream) {
  LnGroups P = {};
  P.g1[0] = g1; P.b1[0] = b1; P.g2[0] = g2; P.b2[0] = b2;
  return cwf_ln_pair_fwd_g(x, x2, perm_T, &P, 1, ya, yb, stats, rows, E, eps, stream);
}

extern "C" int cwf_ln_pair_bwd_g(const float* dy, const float* da, const float* db, const float* x, const float* x2, int perm_T,
                                 const struct cwf_ln_group_params* h_params, int groups, const float* stats, float* dx, float* dx2,
                                 int rows, int E, int accumulate_params, void* stream) {
  if (!da || !x || !stats || !dx || rows <= 0 || !ln_groups_ok(h_params, groups, db != nullptr, true) || rows % groups) return CWF_E_BADARG;
  if (db && !x2) return CWF_E_BADARG;
  if (dx2 && (!db || perm_T != 0)) return CWF_E_BADARG;
  const int rpg = rows / groups;
  if (perm_T > 0 && rpg % (2 * perm_T)) return CWF_E_BADARG;
  if (E != 512 && E != 256 && E != 128 && E != 64) return CWF_E_BADARG;
  const LnGroups P = *h_params;
  hipLaunchKernelGGL(ln_pair_params_kernel, dim3(cdiv(E, 64), db ? 2 : 1, groups), dim3(1024), 0, cwf_stream(stream), da, db, x, x2, perm_T, stats,
                     P, rpg, rows, E, accumulate_params);
  dim3 grid(cdiv(rows * (dx2 ? 2 : 1), 4));
  LN_DISPATCH(ln_pair_bwd_kernel, grid, dy, da, db, x, x2, perm_T, P, rpg, stats, dx, dx2, rows, E);
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_ln_pair_bwd(const float* dy, const float* da, const float* db, const float* x, const float* x2, int perm_T,
                               const float* g1, const float* g2, const float* stats, float* dx, float* dx2,
                               float* dg1, float* db1, float* dg2, float* db2, int rows, int E, int accumulate_params, void* stream) {
  LnGroups P = {};
  P.g1[0] = g1; P.g2[0] = g2; P.dg1[0] = dg1; P.db1[0] = db1; P.dg2[0] = dg2; P.db2[0] = db2;
  return cwf_ln_pair_bwd_g(dy, da, db, x, x2, perm_T, &P, 1, stats, dx, dx2, rows, E, accumulate_params, stream);
}

// dz = dh * keep(i) * gelu'(z): the backward of Dropout(GELU(z)) (FeedForward, ResidualNorm.py:40-43) with the mask recomputed
__global__ void gelu_bwd_drop_kernel(const float* __restrict__ z, const float* __restrict__ dh, float* __restrict__ dz, int64_t n,
                                     const uint64_t* __restrict__ rng, uint64_t off, float p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = z[i];
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
  float g = dh[i] * (cdf + v * pdf);
  if (p > 0.f) g *= cwf_keep(rng, off, (uint64_t)i, (uint64_t)n, p, 0.f);
  dz[i] = g;
}
extern "C" int cwf_gelu_bwd_drop(const float* z, const float* dh, float* dz, int64_t n, const uint64_t* rng, uint64_t off, float p, void* stream) {
  if (!z || !dh || !dz || n <= 0 || (p > 0.f && !rng) || p < 0.f || p >= 1.f) return CWF_E_BADARG;
  hipLaunchKernelGGL(gelu_bwd_drop_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, cwf_stream(stream), z, dh, dz, n, rng, off, p);
  CWF_LAUNCH_CHECK();
  return 0;
}
