// K1, stem: the 3x3x3 conv 4 -> 16 at full resolution (InitConv, Unet_skipconnection.py:22-33), split-bf16 / single-bf16 MFMA operands.
//
// conv16s runs it as a 16-channel layer: 12 of every 16 K slots are zero padding, 14 K-steps per 16-voxel M-tile for 27 x 4 = 108
// products' worth of work (221 us per launch for a 42 us HBM floor).  Here K = 32 is EIGHT TAPS x FOUR CHANNELS: 27 taps = 4 K-steps
// (the 28th slot has zero weights), 3.5x less MFMA and LDS work, and the layer becomes what its bytes say -- a stream (67 MB in,
// 268 MB out at batch 2):
//   * the halo tile of a 4x4x16 output tile (6x6x18 voxels x 4 channels) is staged once as bf16 hi / lo images [voxel][4] (10 KB);
//   * lane (r, kq) of the A operand holds taps (8s + 2kq, 8s + 2kq + 1) x 4 channels of voxel r: two 8-byte LDS reads at
//     per-lane tap offsets (computed once);
//   * the B fragments (4 K-steps x hi / lo) come straight from the RAW nn.Conv3d weight [16][4][3][3][3] at kernel start and stay in
//     registers; bias starts the accumulator; dropout3d's per-(sample, channel) scale and the InstanceNorm statistics of the output
//     are the epilogue, as in conv16s; a workgroup owns a contiguous run of tiles and its statistics leave as one atomic instruction
//     per sample.
#include "common.h"
#include <cstdlib>

typedef __bf16 bf16x8_s __attribute__((ext_vector_type(8)));
typedef __bf16 bf2_s __attribute__((ext_vector_type(2)));
typedef float f2_s __attribute__((ext_vector_type(2)));

struct StemArgs {
  const float* x; int x_ldc; const float* w; const float* bias; float* y; int y_ldc; const float* out_scale; double* stats;
  int N, D, H, W, tiles_d, tiles_h, tiles_w, total_tiles, tiles_per_wg;
};

__device__ __forceinline__ unsigned stem_pk(float a, float b) {
  const f2_s f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2_s));
}
__device__ __forceinline__ void stem_split(float a, float b, unsigned& hi, unsigned& lo) {
  hi = stem_pk(a, b);
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  lo = stem_pk(a - ha, b - hb);
}

template <bool X3>
__global__ __launch_bounds__(256) void stem_conv_kernel(const StemArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned short xh[648 * 4];
  __shared__ __attribute__((aligned(16))) unsigned short xl[648 * 4];
  __shared__ float red[4][32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kq = lane >> 4;
  const int tiles_sp = a.tiles_d * a.tiles_h * a.tiles_w;
  const int t_begin = blockIdx.x * a.tiles_per_wg;
  const int t_end = min(a.total_tiles, t_begin + a.tiles_per_wg);

  // ---- B fragments: lane (co = r, kq): k = [tap 8s+2kq: ci 0..3 | tap 8s+2kq+1: ci 0..3]
  uint4 bh[4], bl[4];
  int tofs[4][2];                                          // LDS voxel offsets of this lane's two taps per K-step
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float wv[8];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int t = 8 * s + 2 * kq + e;
      const int tt = t < 27 ? t : 0;
      tofs[s][e] = ((tt / 9) * 6 + (tt / 3) % 3) * 18 + tt % 3;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) wv[e * 4 + ci] = t < 27 ? a.w[(r * 4 + ci) * 27 + t] : 0.f;
    }
    unsigned h[4], l[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (X3) stem_split(wv[2 * p], wv[2 * p + 1], h[p], l[p]);
      else { h[p] = stem_pk(wv[2 * p], wv[2 * p + 1]); l[p] = 0u; }
    }
    bh[s] = make_uint4(h[0], h[1], h[2], h[3]); bl[s] = make_uint4(l[0], l[1], l[2], l[3]);
  }
  const float bv = a.bias ? a.bias[r] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  int stat_n = t_begin < t_end ? t_begin / tiles_sp : 0;
  auto flush_stats = [&](int n_) {                        // (all 256 threads; one atomic instruction per workgroup and sample)
    float u1 = s1, u2 = s2;
    u1 += __shfl_xor(u1, 16, 64); u1 += __shfl_xor(u1, 32, 64);
    u2 += __shfl_xor(u2, 16, 64); u2 += __shfl_xor(u2, 32, 64);
    if (kq == 0) { red[wave][r] = u1; red[wave][16 + r] = u2; }
    __syncthreads();
    if (tid < 32) {
      const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
      atomic_add_f64(a.stats + ((int64_t)n_ * 16 + (tid & 15)) * 2 + (tid >> 4), (double)v);
    }
    __syncthreads();
    s1 = 0.f; s2 = 0.f;
  };

  // staging slots of this thread (voxel v = tid + 256 i of the 648): halo coordinates are tile-independent
  int hv[3], hoff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int v = min(tid + 256 * i, 647);
    const int iw = v % 18, t2 = v / 18, ih = t2 % 6, idd = t2 / 6;
    hv[i] = idd | (ih << 4) | (iw << 8);
    hoff[i] = ((idd * a.H + ih) * a.W + iw) * a.x_ldc;
  }
  const bool last_slot = tid + 512 < 648;
  float4 pf[3];
  auto fetch = [&](int tile) {                            // all loads of a tile issued together; consumed one tile later
    const int n = tile / tiles_sp; int rem = tile - n * tiles_sp;
    const int tile_w = rem % a.tiles_w; rem /= a.tiles_w;
    const int tile_h = rem % a.tiles_h; const int tile_d = rem / a.tiles_h;
    const int d0 = tile_d * 4 - 1, h0 = tile_h * 4 - 1, w0 = tile_w * 16 - 1;
    const float* xb = a.x + ((((int64_t)n * a.D + d0) * a.H + h0) * a.W + w0) * a.x_ldc;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int gd = d0 + (hv[i] & 15), gh = h0 + ((hv[i] >> 4) & 15), gw = w0 + (hv[i] >> 8);
      const bool ok = (i < 2 || last_slot) && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
      pf[i] = ok ? *reinterpret_cast<const float4*>(xb + hoff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  unsigned yofs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) yofs[i] = (unsigned)((kq * 4 + i) * a.y_ldc + r);
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int n = tile / tiles_sp; int rem = tile - n * tiles_sp;
    const int tile_w = rem % a.tiles_w; rem /= a.tiles_w;
    const int tile_h = rem % a.tiles_h; const int tile_d = rem / a.tiles_h;
    const int od0 = tile_d * 4, oh0 = tile_h * 4, ow0 = tile_w * 16;
    if (a.stats && n != stat_n) { flush_stats(stat_n); stat_n = n; }
    // ---- the prefetched halo tile -> bf16 hi / lo images (zero padding came with the fetch)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i == 2 && !last_slot) continue;
      const int v = tid + 256 * i;
      const float4 val = pf[i];
      uint2 h, l;
      if (X3) { stem_split(val.x, val.y, h.x, l.x); stem_split(val.z, val.w, h.y, l.y); }
      else { h.x = stem_pk(val.x, val.y); h.y = stem_pk(val.z, val.w); l = make_uint2(0u, 0u); }
      *reinterpret_cast<uint2*>(xh + v * 4) = h;
      if (X3) *reinterpret_cast<uint2*>(xl + v * 4) = l;
    }
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);                // in flight during this tile's MFMAs and stores
    // ---- four M-tiles per wave: rows (td, th) = (wave, j)
    const float osc = a.out_scale ? a.out_scale[(int64_t)n * 16 + r] : 1.f;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){bv, bv, bv, bv};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int vb = (wave * 6 + j) * 18 + r;
        const uint2 h0 = *reinterpret_cast<const uint2*>(xh + (vb + tofs[s][0]) * 4), h1 = *reinterpret_cast<const uint2*>(xh + (vb + tofs[s][1]) * 4);
        const uint4 ah = make_uint4(h0.x, h0.y, h1.x, h1.y);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_s, ah), __builtin_bit_cast(bf16x8_s, bh[s]), acc[j], 0, 0, 0);
        if (X3) {
          const uint2 l0 = *reinterpret_cast<const uint2*>(xl + (vb + tofs[s][0]) * 4), l1 = *reinterpret_cast<const uint2*>(xl + (vb + tofs[s][1]) * 4);
          const uint4 al = make_uint4(l0.x, l0.y, l1.x, l1.y);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_s, ah), __builtin_bit_cast(bf16x8_s, bl[s]), acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_s, al), __builtin_bit_cast(bf16x8_s, bh[s]), acc[j], 0, 0, 0);
        }
      }
    }
    // ---- epilogue: dropout3d scale, store, statistics of the stored values
    const int od = od0 + wave;
    if (od < a.D) {
      float* yt = a.y + ((((int64_t)n * a.D + od) * a.H + oh0) * a.W + ow0) * a.y_ldc;
      if (oh0 + 4 <= a.H && ow0 + 16 <= a.W) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float* yb = yt + (int64_t)j * a.W * a.y_ldc;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = acc[j][i] * osc;
            yb[yofs[i]] = v;
            s1 += v; s2 = fmaf(v, v, s2);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (oh0 + j >= a.H) continue;
          float* yb = yt + (int64_t)j * a.W * a.y_ldc;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (ow0 + kq * 4 + i >= a.W) continue;
            const float v = acc[j][i] * osc;
            yb[yofs[i]] = v;
            s1 += v; s2 = fmaf(v, v, s2);
          }
        }
      }
    }
    __syncthreads();                                       // the next tile's conversion overwrites the images
  }
  if (a.stats && t_begin < t_end) flush_stats(stat_n);
}

// y = (conv3x3x3(x; w) + bias) * out_scale (+ statistics of y): the stem layer.  x [N][D][H][W][4 (ldc x_ldc)] fp32, w the raw
// nn.Conv3d weight [16][4][3][3][3], y [N][D][H][W][16 (ldc y_ldc)]; out_scale [N][16] and stats [N][16][2] nullable.
extern "C" int cwf_conv_stem_bf16(int x3, const float* x, int x_ldc, const float* w, const float* bias, float* y, int y_ldc,
                                  const float* out_scale, double* stats, int N, int D, int H, int W, void* stream) {
  if (!x || !w || !y || N <= 0 || D <= 0 || H <= 0 || W <= 0) return CWF_E_BADARG;
  if ((x_ldc & 3) || x_ldc < 4 || y_ldc < 16 || ((uintptr_t)x & 15)) return CWF_E_ALIGN;
  StemArgs a;
  a.x = x; a.x_ldc = x_ldc; a.w = w; a.bias = bias; a.y = y; a.y_ldc = y_ldc; a.out_scale = out_scale; a.stats = stats;
  a.N = N; a.D = D; a.H = H; a.W = W;
  a.tiles_d = cdiv(D, 4); a.tiles_h = cdiv(H, 4); a.tiles_w = cdiv(W, 16);
  a.total_tiles = N * a.tiles_d * a.tiles_h * a.tiles_w;
  static const int g0 = getenv("CWF_STEM_GRID") ? atoi(getenv("CWF_STEM_GRID")) : 1024;      // (512 / 1024 / 2048 / 4096 workgroups: 108 / 110 / 121 / 131 us)
  int grid = g0; if (grid > a.total_tiles) grid = a.total_tiles;
  a.tiles_per_wg = cdiv(a.total_tiles, grid);
  grid = cdiv(a.total_tiles, a.tiles_per_wg);
  if (x3) hipLaunchKernelGGL(stem_conv_kernel<true>, dim3(grid), dim3(256), 0, cwf_stream(stream), a);
  else hipLaunchKernelGGL(stem_conv_kernel<false>, dim3(grid), dim3(256), 0, cwf_stream(stream), a);
  CWF_LAUNCH_CHECK();
  return 0;
}
