// K1, first down-sampling layer: the 3x3x3 stride-2 conv 16 -> 32 at full resolution (EnDown1, Unet_skipconnection.py:60-68), split-bf16 or
// single-bf16 MFMA operands, no prologue.  The tap-table kernel runs it with one 64-voxel output tile per workgroup, thirteen dependent
// staging slots per thread and stride-2 LDS fragment reads (4-way bank conflicts): 212 us for a 42 us HBM floor.  Here, with the stem
// kernel's recipe (conv_stem.hip):
//   * persistent workgroups walk contiguous runs of 2x2x16 output tiles; the NEXT tile's 5x5x33-voxel halo (13 x 16-byte loads per thread)
//     is in flight during a tile's MFMAs and stores;
//   * the halo rows are stored PARITY-SPLIT along W ([17 even | 17 odd] voxels x 16 channels): output voxel r with tap kw reads input
//     voxel 2r + kw = slot (kw & 1) * 17 + r + (kw >> 1) -- consecutive lanes read consecutive 32-byte slots, no bank conflicts;
//   * K = 32 is a tap pair x 16 channels (14 K-steps), the weights' hi fragments live in registers (built from the RAW weight
//     [32][16][3][3][3] at kernel start), the lo fragments lane-linear in LDS;
//   * wave w owns M-tile (td, th) = (w >> 1, w & 1) and both 16-channel output tiles; bias starts the accumulators; the InstanceNorm
//     statistics of the output leave as one atomic instruction per workgroup and sample.
#include "common.h"
#include <cstdlib>

typedef __bf16 bf16x8_d __attribute__((ext_vector_type(8)));
typedef __bf16 bf2_d __attribute__((ext_vector_type(2)));
typedef float f2_d __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_d __attribute__((ext_vector_type(4)));

struct S2Args {
  const float* x; int x_ldc; const float* w; const float* bias; float* y; int y_ldc; double* stats;
  int N, Di, Hi, Wi, Do, Ho, Wo, tiles_d, tiles_h, tiles_w, total_tiles, tiles_per_wg;
};

__device__ __forceinline__ unsigned s2_pk(float a, float b) {
  const f2_d f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2_d));
}
__device__ __forceinline__ void s2_split(float a, float b, unsigned& hi, unsigned& lo) {
  hi = s2_pk(a, b);
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  lo = s2_pk(a - ha, b - hb);
}

#define S2_ROW 34                                        // voxel slots per halo row: 17 even + 17 odd
#define S2_NVOX (5 * 5 * 33)                             // 825 halo voxels
#define S2_SLOTS 13                                      // ceil(825 * 4 quads / 256 threads)
#define S2_IMG (5 * 5 * S2_ROW * 16)                     // bf16 elements of one image (27,200 B)

template <bool X3>
__global__ __launch_bounds__(512) void conv_s2c16_kernel(const S2Args a) {
  // 8 waves: waves 0-3 read LDS / issue MFMAs / store (wave w owns M-tile (td, th) = (w >> 1, w & 1) and both 16-channel output tiles),
  // waves 4-7 are loaders: they convert the prefetched halo tile t+1 into the OTHER image buffer while tile t is computed, then request
  // tile t+2; one raw s_barrier per tile (the loaders' global loads stay in flight across it).
  extern __shared__ float4 lds4[];
  constexpr int IMGS = X3 ? 2 : 1;                         // hi (+ lo) image per buffer
  unsigned short* img = reinterpret_cast<unsigned short*>(lds4);            // [2 buffers][IMGS][S2_IMG]
  uint4* wlo = reinterpret_cast<uint4*>(img + 2 * IMGS * S2_IMG);           // [14 steps][2 tiles][64 lanes]
  float (*red)[64] = reinterpret_cast<float (*)[64]>(reinterpret_cast<char*>(wlo) + (X3 ? 14 * 2 * 64 * 16 : 0));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_sp = a.tiles_d * a.tiles_h * a.tiles_w;
  const int t_begin = blockIdx.x * a.tiles_per_wg;
  const int t_end = min(a.total_tiles, t_begin + a.tiles_per_wg);
  const int niter = max(t_end - t_begin, 0);
  // both roles see the same tile sequence; the statistics flush (two barriers) happens where the sample index changes and at the end
  auto sample_of = [&](int tile) { return tile / tiles_sp; };

  if (wave < 4) {
    // =============================================================== MFMA waves
    const int r = lane & 15, kq = lane >> 4;
    uint4 bh[14][2];
    int tofs[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const int t = 2 * s + (kq >> 1);
      const int tt = t < 27 ? t : 0;
      const int kd = tt / 9, kh = (tt / 3) % 3, kw = tt % 3;
      tofs[s] = (kd * 5 + kh) * S2_ROW + (kw & 1) * 17 + (kw >> 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float wv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) wv[c] = t < 27 ? a.w[((j * 16 + r) * 16 + (kq & 1) * 8 + c) * 27 + t] : 0.f;
        unsigned h[4], l[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (X3) s2_split(wv[2 * p], wv[2 * p + 1], h[p], l[p]);
          else { h[p] = s2_pk(wv[2 * p], wv[2 * p + 1]); l[p] = 0u; }
        }
        bh[s][j] = make_uint4(h[0], h[1], h[2], h[3]);
        if (X3 && wave == 0) wlo[(s * 2 + j) * 64 + lane] = make_uint4(l[0], l[1], l[2], l[3]);
      }
    }
    const float bv0 = a.bias ? a.bias[r] : 0.f, bv1 = a.bias ? a.bias[16 + r] : 0.f;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    int stat_n = niter ? sample_of(t_begin) : 0;
    auto flush_stats = [&](int n_) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float u1 = s1[j], u2 = s2[j];
        u1 += __shfl_xor(u1, 16, 64); u1 += __shfl_xor(u1, 32, 64);
        u2 += __shfl_xor(u2, 16, 64); u2 += __shfl_xor(u2, 32, 64);
        if (kq == 0) { red[wave][j * 16 + r] = u1; red[wave][32 + j * 16 + r] = u2; }
        s1[j] = 0.f; s2[j] = 0.f;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid < 64) {
        const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        atomic_add_f64(a.stats + ((int64_t)n_ * 32 + (tid & 31)) * 2 + (tid >> 5), (double)v);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    const unsigned lds_x = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)img;
    const int td = wave >> 1, th = wave & 1;
    const unsigned abase = lds_x + (unsigned)((((2 * td) * 5 + 2 * th) * S2_ROW + r) * 32 + (kq & 1) * 16);
    typedef const u32x4_d __attribute__((address_space(3)))* lds_u4p;
    unsigned yofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) yofs[i] = (unsigned)((kq * 4 + i) * a.y_ldc + r);
    for (int it = 0; it < niter; ++it) {
      const int tile = t_begin + it;
      const int n = tile / tiles_sp; int rem = tile - n * tiles_sp;
      const int tile_w = rem % a.tiles_w; rem /= a.tiles_w;
      const int tile_h = rem % a.tiles_h; const int tile_d = rem / a.tiles_h;
      const int od0 = tile_d * 2, oh0 = tile_h * 2, ow0 = tile_w * 16;
      if (a.stats && n != stat_n) { flush_stats(stat_n); stat_n = n; }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // buffer it & 1 is complete (and wave 0's lo-weight writes have landed)
      const unsigned ab = abase + (unsigned)((it & 1) * IMGS * S2_IMG * 2);
      f32x4 acc0 = {bv0, bv0, bv0, bv0}, acc1 = {bv1, bv1, bv1, bv1};
#pragma unroll
      for (int s = 0; s < 14; ++s) {
        const unsigned ad = ab + (unsigned)tofs[s] * 32u;
        const u32x4_d ah = *(lds_u4p)(uintptr_t)ad;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_d, ah), __builtin_bit_cast(bf16x8_d, bh[s][0]), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_d, ah), __builtin_bit_cast(bf16x8_d, bh[s][1]), acc1, 0, 0, 0);
        if (X3) {
          const u32x4_d al = *(lds_u4p)(uintptr_t)(ad + (unsigned)(S2_IMG * 2));
          const uint4 l0 = wlo[(s * 2 + 0) * 64 + lane], l1 = wlo[(s * 2 + 1) * 64 + lane];
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_d, ah), __builtin_bit_cast(bf16x8_d, l0), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_d, ah), __builtin_bit_cast(bf16x8_d, l1), acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_d, al), __builtin_bit_cast(bf16x8_d, bh[s][0]), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_d, al), __builtin_bit_cast(bf16x8_d, bh[s][1]), acc1, 0, 0, 0);
        }
      }
      const int od = od0 + td, oh = oh0 + th;
      if (od < a.Do && oh < a.Ho) {
        float* yb = a.y + ((((int64_t)n * a.Do + od) * a.Ho + oh) * a.Wo + ow0) * a.y_ldc;
        const bool fullw = ow0 + 16 <= a.Wo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (!fullw && ow0 + kq * 4 + i >= a.Wo) continue;
          const float v0 = acc0[i], v1 = acc1[i];
          yb[yofs[i]] = v0; yb[yofs[i] + 16] = v1;
          s1[0] += v0; s2[0] = fmaf(v0, v0, s2[0]); s1[1] += v1; s2[1] = fmaf(v1, v1, s2[1]);
        }
      }
    }
    asm volatile("s_barrier" ::: "memory");                // (pairs with the loaders' last barrier)
    if (a.stats && niter) flush_stats(stat_n);
  } else {
    // =============================================================== loader waves
    const int lt = tid - 256;
    const int q = lt & 3;
    int hv[S2_SLOTS], hoff[S2_SLOTS]; unsigned lofs[S2_SLOTS];
#pragma unroll
    for (int i = 0; i < S2_SLOTS; ++i) {
      const int v = min((lt >> 2) + 64 * i, S2_NVOX - 1);
      const int iw = v % 33, t2 = v / 33, ih = t2 % 5, idd = t2 / 5;
      hv[i] = idd | (ih << 4) | (iw << 8);
      hoff[i] = ((idd * a.Hi + ih) * a.Wi + iw) * a.x_ldc + q * 4;
      lofs[i] = (unsigned)((((idd * 5 + ih) * S2_ROW) + (iw & 1) * 17 + (iw >> 1)) * 16 + q * 4);
    }
    const bool last_slot = (lt >> 2) + 64 * (S2_SLOTS - 1) < S2_NVOX;
    float4 pf[S2_SLOTS];
    auto fetch = [&](int tile) {
      const int n = tile / tiles_sp; int rem = tile - n * tiles_sp;
      const int tile_w = rem % a.tiles_w; rem /= a.tiles_w;
      const int tile_h = rem % a.tiles_h; const int tile_d = rem / a.tiles_h;
      const int d0 = tile_d * 4 - 1, h0 = tile_h * 4 - 1, w0 = tile_w * 32 - 1;
      const float* xb = a.x + ((((int64_t)n * a.Di + d0) * a.Hi + h0) * a.Wi + w0) * a.x_ldc;
#pragma unroll
      for (int i = 0; i < S2_SLOTS; ++i) {
        const int gd = d0 + (hv[i] & 15), gh = h0 + ((hv[i] >> 4) & 15), gw = w0 + (hv[i] >> 8);
        const bool ok = (i < S2_SLOTS - 1 || last_slot) && (unsigned)gd < (unsigned)a.Di && (unsigned)gh < (unsigned)a.Hi && (unsigned)gw < (unsigned)a.Wi;
        pf[i] = ok ? *reinterpret_cast<const float4*>(xb + hoff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto convert = [&](int buf) {
      unsigned short* xh = img + buf * IMGS * S2_IMG;
      unsigned short* xl = xh + S2_IMG;
#pragma unroll
      for (int i = 0; i < S2_SLOTS; ++i) {
        if (i == S2_SLOTS - 1 && !last_slot) continue;
        const float4 val = pf[i];
        uint2 h, l;
        if (X3) { s2_split(val.x, val.y, h.x, l.x); s2_split(val.z, val.w, h.y, l.y); }
        else { h.x = s2_pk(val.x, val.y); h.y = s2_pk(val.z, val.w); l = make_uint2(0u, 0u); }
        *reinterpret_cast<uint2*>(xh + lofs[i]) = h;
        if (X3) *reinterpret_cast<uint2*>(xl + lofs[i]) = l;
      }
    };
    int stat_n = niter ? sample_of(t_begin) : 0;
    if (niter > 0) { fetch(t_begin); convert(0); }
    if (niter > 1) fetch(t_begin + 1);
    for (int it = 0; it < niter; ++it) {
      const int n = sample_of(t_begin + it);
      if (a.stats && n != stat_n) {                        // the MFMA waves flush their statistics: two barriers
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        stat_n = n;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // buffer it & 1 handed over (its conversion's LDS writes are done)
      if (it + 1 < niter) {
        convert((it + 1) & 1);                             // (the MFMA waves finished reading that buffer before this barrier)
        if (it + 2 < niter) fetch(t_begin + it + 2);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (a.stats && niter) {
      asm volatile("s_barrier" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    }
  }
}

// y = conv3x3x3 stride 2 (x; w) + bias (+ statistics of y) for 16 -> 32 channels.  x [N][Di][Hi][Wi][16] fp32 (ldc x_ldc), w the raw
// nn.Conv3d weight [32][16][3][3][3], y [N][Do][Ho][Wo][32] (ldc y_ldc), Do = (Di + 1) / 2 ...; stats [N][32][2] nullable.
extern "C" int cwf_conv_s2c16_bf16(int x3, const float* x, int x_ldc, const float* w, const float* bias, float* y, int y_ldc, double* stats,
                                   int N, int Di, int Hi, int Wi, void* stream) {
  if (!x || !w || !y || N <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0) return CWF_E_BADARG;
  if ((x_ldc & 3) || x_ldc < 16 || y_ldc < 32 || ((uintptr_t)x & 15)) return CWF_E_ALIGN;
  S2Args a;
  a.x = x; a.x_ldc = x_ldc; a.w = w; a.bias = bias; a.y = y; a.y_ldc = y_ldc; a.stats = stats;
  a.N = N; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = (Di + 1) / 2; a.Ho = (Hi + 1) / 2; a.Wo = (Wi + 1) / 2;
  a.tiles_d = cdiv(a.Do, 2); a.tiles_h = cdiv(a.Ho, 2); a.tiles_w = cdiv(a.Wo, 16);
  a.total_tiles = N * a.tiles_d * a.tiles_h * a.tiles_w;
  static const int g0 = getenv("CWF_S2_GRID") ? atoi(getenv("CWF_S2_GRID")) : 256;
  int grid = g0; if (grid > a.total_tiles) grid = a.total_tiles;
  a.tiles_per_wg = cdiv(a.total_tiles, grid);
  grid = cdiv(a.total_tiles, a.tiles_per_wg);
  const size_t lds = (size_t)2 * (x3 ? 2 : 1) * S2_IMG * 2 + (x3 ? 14 * 2 * 64 * 16 : 0) + 4 * 64 * sizeof(float);      // two image buffers
  if (x3) { CWF_MAX_LDS_ONCE((&conv_s2c16_kernel<true>)); hipLaunchKernelGGL(conv_s2c16_kernel<true>, dim3(grid), dim3(512), lds, cwf_stream(stream), a); }
  else { CWF_MAX_LDS_ONCE((&conv_s2c16_kernel<false>)); hipLaunchKernelGGL(conv_s2c16_kernel<false>, dim3(grid), dim3(512), lds, cwf_stream(stream), a); }
  CWF_LAUNCH_CHECK();
  return 0;
}
