// K1w (split-bf16 form) -- weight/bias gradient of the conv family on v_mfma_f32_16x16x32_bf16.
//
//   dW[tap][ci][co] = sum_vox xa[vox + tap][ci] * dy[vox][co]     M = ci (16), N = co (16), K = voxels (32 per MFMA)
// Both operands are K-major for this product (the voxel index is the summation index) but live voxel-major in LDS
// ([voxel][16 ch] bf16, written coalesced from NDHWC HBM), so both fragments are fetched with the CDNA4 transposing LDS
// read ds_read_b64_tr_b16: per 16-lane group it takes a 4-voxel x 16-channel block and hands lane i channel i of the
// 4 voxels -- two reads give the 8 consecutive-k bf16 values of a 16x16x32 operand, no shuffles, no second LDS image.
// (EXEC is all ones at every such read: all branches around them are wave-uniform.)
// X3 = true: x = xh + xl, dy = dh + dl -> xh.dh + xh.dl + xl.dh (fp32 accumulate);  X3 = false: xh.dh only.
// Work split, persistent tile walk, partial-slab layout and the reduce step are those of wgrad_mfma.hip.
#include "common.h"
#include <type_traits>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// Workgroups of the persistent weight-gradient kernels (wgrad16, wgrad_s1, pw_wgrad: 10-16 waves each, most of a CU's registers).
// They run on the side stream BESIDE the main stream's kernels: at one workgroup per CU (256) a main-stream workgroup often finds
// no CU with room and waits for a whole side kernel; fewer workgroups leave CUs to the main stream (an isolated launch is slower,
// the step faster).  Round 2, launches from Python: 192 (99.3 -> 100.1 volumes/s).  Round 3, launch plan (the main stream never
// waits for the host any more): 128 -- 64 / 96 / 128 / 160 / 192 / 256 give 91.3 / 99.9 / 104.8 / 103.7 / 103.1 / 102.6 volumes/s.
// End of round 3 (LDS-DMA weight gradients; only their chain off the main stream): 128 / 160 / 192 / 256 give 110.0 / 110.4 / 110.0 / 108.9: 160.
// CWF_SIDE_WGS overrides (multiple of 8).
extern "C" int cwf_wgrad_nsplit(int op, int N, int Do, int Ho, int Wo, int Cin, int Cout);
static int side_wgs() { static const int v = getenv("CWF_SIDE_WGS") ? atoi(getenv("CWF_SIDE_WGS")) : 160; return v; }

struct WgArgsB {
  ConvGeom g;
  const float* x; const float* in_scale; const float* in_shift; float in_slope;
  const float* dy; int dy_ldc; float* partial;
  int ngroups, tiles_per_split, total_tiles;
  int64_t slab_floats;
  int cls_slab_base[8];
  // grouped launches (cwf_wgrad_mfma_bf16_grouped: the three sub-regions' head layers): blockIdx.z = group, each with its own
  // x / dy / slab buffer; single-class (3x3x3 stride-1) operators only.  groups == 0: an ordinary launch (blockIdx.z = class)
  int groups;
  const float* x_g[3]; const float* dy_g[3]; float* partial_g[3];
  // wgrad16_kernel only: dy is used as dy * dy_scale[n][co] (the per-(sample, channel) scale of a dropout3d behind the conv, InitConv:
  // Unet_skipconnection.py:29-33 -- its backward is a full-resolution elementwise pass otherwise, the last one before the optimizer)
  const float* dy_scale;
};

__device__ __forceinline__ unsigned pack_bf16w(float a, float b) {
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float bf16_roundw(float a) { return (float)(__bf16)a; }

// two transposed reads -> one 8-element K fragment.  p0 / p1: this lane's block-row addresses (bf16 element pointers)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned short* p0, const unsigned short* p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// LDS row pitches (in voxels) that make the transposed fragment reads bank-conflict free.  One ds_read_b64_tr_b16 half-wave
// meets two image rows (lane bit kq & 1) x four block rows (bq) x four 8-byte columns (bp); the two image rows must start
// 128 B apart modulo the 256-B bank period.  x image: 32 B per voxel -> pitch = 4 (mod 8) for stride-1 geometries (left
// unpadded for the three stride-2 layers).  dy image, 2*CGW bytes per voxel, 16 voxels per row: CGW = 16 -> pitch 20; CGW = 32:
// the four block rows already span the bank period at a 64-B stride, so every odd row adds a 32-B skew instead (pitch stays 16:
// the layer set with 32 output channels keeps two workgroups per CU, which a wider pitch loses); CGW = 64: left as is.
__host__ __device__ inline int wg_x_pitch(int IW, int is) { if (is != 1) return IW; int p = IW; while ((p & 7) != 4) ++p; return p; }
__host__ __device__ inline int wg_dy_pitch(int CGW) { return CGW == 16 ? 20 : 16; }
__host__ __device__ inline int wg_dy_skew(int CGW) { return CGW == 32 ? 16 : 0; }            // bf16 elements added to odd rows

template <int TPW, int NTW, bool TAPSPLIT, bool X3>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const WgArgsB a) {
  constexpr int CG = NTW;
  constexpr int CGW = CG * 16;
  extern __shared__ float4 lds4[];
  const ConvGeom& g = a.g;
  const int nvox_in = g.ID * g.IH * g.IW;
  const int XW = wg_x_pitch(g.IW, g.is), DP = wg_dy_pitch(CGW);   // padded LDS row pitches (voxels)
  const int XIMG = g.ID * g.IH * XW * 16;                          // bf16 elements of one x image
  const int MT = g.TD * g.TH;
  unsigned short* xh = reinterpret_cast<unsigned short*>(lds4);
  unsigned short* xl = xh + XIMG;
  unsigned short* dh = xl + (X3 ? XIMG : 0);
  constexpr int DSK = CGW == 32 ? 16 : 0;
  unsigned short* dl = dh + MT * DP * CGW + (MT / 2 + 1) * DSK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-dependent choices stay uniform
  const int kq = lane >> 4;
  const int bq = (lane & 15) >> 2, bp = lane & 3;        // transposed-read block row / column quad supplied by this lane
  const int split = blockIdx.x;
  const int chunk = blockIdx.y / a.ngroups, grp = blockIdx.y % a.ngroups;
  const int cls = a.groups ? 0 : blockIdx.z;
  const float* ax_ = a.groups ? a.x_g[blockIdx.z] : a.x;             // (workgroup-uniform: group operands)
  const float* ady_ = a.groups ? a.dy_g[blockIdx.z] : a.dy;
  float* apart_ = a.groups ? a.partial_g[blockIdx.z] : a.partial;
  const int Dc = g.cls_dims[cls][0], Hc = g.cls_dims[cls][1], Wc = g.cls_dims[cls][2];
  const int ntaps = g.cls_ntaps[cls];
  const int* tapofs = g.tapofs + (g.ncls > 1 ? cls * 8 : 0);
  const int co0 = grp * CGW;
  int tapo[TPW];                                         // this wave's tap offsets in the padded image (bf16 elements)
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = TAPSPLIT ? wave + 4 * i : i;
    const int o = t < ntaps ? tapofs[t] : 0;
    tapo[i] = ((o / g.IW) * XW + o % g.IW) * 16;
  }

  f32x4 acc[TPW][NTW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int tiles_sp = g.tiles_d * g.tiles_h * g.tiles_w;
  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(a.total_tiles, t_begin + a.tiles_per_split);
  const bool vec_dy = (a.dy_ldc & 3) == 0 && (((uintptr_t)ady_) & 15) == 0;
  const int of0 = g.cls_ooff[cls][0], of1 = g.cls_ooff[cls][1], of2 = g.cls_ooff[cls][2];

  // ones operand for the bias row: bf16 1.0 = 0x3F80
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);

  // set-up of the stride-1 fast path (see the K loop)
  const bool geo_s1 = TAPSPLIT && g.is == 1 && g.TD == 4 && g.TH == 4 && g.ID == 6 && g.IH == 6 && g.IW == 18 && ntaps == 27;
  const s16x8 zeros_s = {0, 0, 0, 0, 0, 0, 0, 0};
  const bf16x8 zeros8 = __builtin_bit_cast(bf16x8, zeros_s);
  const bool s1_bias_wave = wave == 3;
  unsigned s1_xa[TPW], s1_da = 0;
  {
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds4;
    const unsigned lane_x = lds_base + (((kq & 1) * 20 + (kq >> 1) * 8 + bq) * 16 + bp * 4) * 2;
    constexpr int DPc = CGW == 16 ? 20 : 16, DSKc = CGW == 32 ? 16 : 0;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int t = wave + 4 * i;
      s1_xa[i] = lane_x + (t < 27 ? (((t / 9) * 6 + (t / 3) % 3) * 20 + t % 3) * 32 : 0);
    }
    s1_da = lds_base + 36 * 20 * 16 * 2 * (X3 ? 2 : 1) + (((kq & 1) * DPc + (kq >> 1) * 8 + bq) * CGW + (kq & 1) * DSKc + bp * 4) * 2;
  }

  for (int tile = t_begin; tile < t_end; ++tile) {
    const int n = tile / tiles_sp; int rem = tile % tiles_sp;
    const int tile_w = rem % g.tiles_w; rem /= g.tiles_w;
    const int tile_h = rem % g.tiles_h; const int tile_d = rem / g.tiles_h;
    const int od0 = tile_d * g.TD, oh0 = tile_h * g.TH, ow0 = tile_w * 16;
    if (od0 >= Dc || oh0 >= Hc || ow0 >= Wc) continue;
    __syncthreads();
    if (!TAPSPLIT && MT == 16 && g.IW == 16 && g.is == 1) {
      // ---- 1-tap operators (1x1 conv, transposed conv classes): no halo, 256 voxels per tile, almost no MFMA work per
      // byte -- pure streaming.  ALL global loads of the tile (4 x quads + CGW/4 dy quads per thread) are issued before the
      // first conversion, so a tile pays one memory latency instead of one per staging slot (the loop form below measured
      // 1.3 TB/s on the 128^3 layers).
      const int q = tid & 3;
      const int c = chunk * 16 + q * 4;
      const bool cval = c < g.Cin;
      const bool has_norm = a.in_scale != nullptr;
      const bool plain = !has_norm && a.in_slope == 1.f;
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_norm && cval) {
        sc = *reinterpret_cast<const float4*>(a.in_scale + (int64_t)n * g.Cin + c);
        sh = *reinterpret_cast<const float4*>(a.in_shift + (int64_t)n * g.Cin + c);
      }
      const int id0 = od0 + g.lo[0], ih0 = oh0 + g.lo[1], iw0 = ow0 + g.lo[2];
      constexpr int DSL = CGW / 4;                       // dy quads per thread (16 M-tile rows x 16 voxels x CGW/4 over 256 threads)
      float4 vx[4], vd[DSL];
      unsigned okx = 0, okd = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int v = (tid >> 2) + 64 * i;               // voxel of the 4 x 4 x 16 tile
        const int iw = v & 15, ih = (v >> 4) & 3, idd = v >> 6;
        const int gd = id0 + idd, gh = ih0 + ih, gw = iw0 + iw;
        const bool ok = cval && gd >= 0 && gd < g.Di && gh >= 0 && gh < g.Hi && gw >= 0 && gw < g.Wi;
        const float* px = ax_ + ((((int64_t)n * g.Di + gd) * g.Hi + gh) * g.Wi + gw) * g.x_ldc + c;
        vx[i] = ok ? *reinterpret_cast<const float4*>(px) : make_float4(0.f, 0.f, 0.f, 0.f);
        okx |= ok ? (1u << i) : 0u;
      }
#pragma unroll
      for (int k = 0; k < DSL; ++k) {
        const int e = tid + 256 * k;
        const int vox = e / DSL, cq = e % DSL;
        const int tw = vox & 15, mt = vox >> 4;
        const int od = od0 + (mt >> 2), oh = oh0 + (mt & 3), ow = ow0 + tw;
        const int co = co0 + cq * 4;
        const bool ok = od < Dc && oh < Hc && ow < Wc && co < g.Cout;
        const int64_t gv = (((int64_t)n * g.Do + (od * g.os + of0)) * g.Ho + (oh * g.os + of1)) * g.Wo + (ow * g.os + of2);
        const float* pd = ady_ + gv * a.dy_ldc + co;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
          if (vec_dy && co + 3 < g.y_ldc) {
            val = *reinterpret_cast<const float4*>(pd);
            if (co + 1 >= g.Cout) val.y = 0.f;
            if (co + 2 >= g.Cout) val.z = 0.f;
            if (co + 3 >= g.Cout) val.w = 0.f;
          } else {
            val.x = pd[0];
            if (co + 1 < g.Cout) val.y = pd[1];
            if (co + 2 < g.Cout) val.z = pd[2];
            if (co + 3 < g.Cout) val.w = pd[3];
          }
        }
        vd[k] = val;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int v = (tid >> 2) + 64 * i;
        float v0 = vx[i].x, v1 = vx[i].y, v2 = vx[i].z, v3 = vx[i].w;
        if (!plain) {
          v0 = fmaf(v0, sc.x, sh.x); v1 = fmaf(v1, sc.y, sh.y); v2 = fmaf(v2, sc.z, sh.z); v3 = fmaf(v3, sc.w, sh.w);
          v0 = fmaxf(v0, v0 * a.in_slope); v1 = fmaxf(v1, v1 * a.in_slope); v2 = fmaxf(v2, v2 * a.in_slope); v3 = fmaxf(v3, v3 * a.in_slope);
          const bool was = (okx >> i) & 1u;              // zero padding applies after the activation
          v0 = was ? v0 : 0.f; v1 = was ? v1 : 0.f; v2 = was ? v2 : 0.f; v3 = was ? v3 : 0.f;
        }
        uint2 h; h.x = pack_bf16w(v0, v1); h.y = pack_bf16w(v2, v3);
        const int vo = ((v >> 4) * XW + (v & 15)) * 16 + q * 4;
        *reinterpret_cast<uint2*>(xh + vo) = h;
        if (X3) {
          uint2 l;
          l.x = pack_bf16w(v0 - bf16_roundw(v0), v1 - bf16_roundw(v1));
          l.y = pack_bf16w(v2 - bf16_roundw(v2), v3 - bf16_roundw(v3));
          *reinterpret_cast<uint2*>(xl + vo) = l;
        }
      }
#pragma unroll
      for (int k = 0; k < DSL; ++k) {
        const int e = tid + 256 * k;
        const int vox = e / DSL, cq = e % DSL;
        const int tw = vox & 15, mt = vox >> 4;
        const float4 val = vd[k];
        uint2 h; h.x = pack_bf16w(val.x, val.y); h.y = pack_bf16w(val.z, val.w);
        const int dofs = (mt * DP + tw) * CGW + ((mt + 1) >> 1) * DSK + cq * 4;
        *reinterpret_cast<uint2*>(dh + dofs) = h;
        if (X3) {
          uint2 l;
          l.x = pack_bf16w(val.x - bf16_roundw(val.x), val.y - bf16_roundw(val.y));
          l.y = pack_bf16w(val.z - bf16_roundw(val.z), val.w - bf16_roundw(val.w));
          *reinterpret_cast<uint2*>(dl + dofs) = l;
        }
      }
      (void)okd;
    } else {
    // ---- x tile (activated) as bf16 hi/lo
    {
      const int q = tid & 3;
      const int c = chunk * 16 + q * 4;
      const bool cval = c < g.Cin;
      const bool has_norm = a.in_scale != nullptr;
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_norm && cval) {
        sc = *reinterpret_cast<const float4*>(a.in_scale + (int64_t)n * g.Cin + c);
        sh = *reinterpret_cast<const float4*>(a.in_shift + (int64_t)n * g.Cin + c);
      }
      const int id0 = od0 * g.is + g.lo[0], ih0 = oh0 * g.is + g.lo[1], iw0 = ow0 * g.is + g.lo[2];
      for (int v = tid >> 2; v < nvox_in; v += 64) {
        const int iw = v % g.IW; const int t2 = v / g.IW;
        const int ih = t2 % g.IH; const int idd = t2 / g.IH;
        const int gd = id0 + idd, gh = ih0 + ih, gw = iw0 + iw;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cval && gd >= 0 && gd < g.Di && gh >= 0 && gh < g.Hi && gw >= 0 && gw < g.Wi) {
          val = *reinterpret_cast<const float4*>(ax_ + ((((int64_t)n * g.Di + gd) * g.Hi + gh) * g.Wi + gw) * g.x_ldc + c);
          if (has_norm || a.in_slope != 1.f) {
            val.x = cwf_act(val.x * sc.x + sh.x, a.in_slope); val.y = cwf_act(val.y * sc.y + sh.y, a.in_slope);
            val.z = cwf_act(val.z * sc.z + sh.z, a.in_slope); val.w = cwf_act(val.w * sc.w + sh.w, a.in_slope);
          }
        }
        uint2 h; h.x = pack_bf16w(val.x, val.y); h.y = pack_bf16w(val.z, val.w);
        const int vo = (t2 * XW + iw) * 16 + q * 4;
        *reinterpret_cast<uint2*>(xh + vo) = h;
        if (X3) {
          uint2 l;
          l.x = pack_bf16w(val.x - bf16_roundw(val.x), val.y - bf16_roundw(val.y));
          l.y = pack_bf16w(val.z - bf16_roundw(val.z), val.w - bf16_roundw(val.w));
          *reinterpret_cast<uint2*>(xl + vo) = l;
        }
      }
    }
    // ---- dy tile [MV][CGW] as bf16 hi/lo
    for (int e = tid; e < MT * 16 * (CGW / 4); e += 256) {
      const int vox = e / (CGW / 4), cq = e % (CGW / 4);
      const int tw = vox & 15, mt = vox >> 4;
      const int od = od0 + mt / g.TH, oh = oh0 + mt % g.TH, ow = ow0 + tw;
      const int co = co0 + cq * 4;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (od < Dc && oh < Hc && ow < Wc && co < g.Cout) {
        const int64_t gv = (((int64_t)n * g.Do + (od * g.os + of0)) * g.Ho + (oh * g.os + of1)) * g.Wo + (ow * g.os + of2);
        const float* p = ady_ + gv * a.dy_ldc + co;
        if (vec_dy && co + 3 < g.y_ldc) {
          val = *reinterpret_cast<const float4*>(p);
          if (co + 1 >= g.Cout) val.y = 0.f;
          if (co + 2 >= g.Cout) val.z = 0.f;
          if (co + 3 >= g.Cout) val.w = 0.f;
        } else {
          val.x = p[0];
          if (co + 1 < g.Cout) val.y = p[1];
          if (co + 2 < g.Cout) val.z = p[2];
          if (co + 3 < g.Cout) val.w = p[3];
        }
      }
      uint2 h; h.x = pack_bf16w(val.x, val.y); h.y = pack_bf16w(val.z, val.w);
      const int dofs = (mt * DP + tw) * CGW + ((mt + 1) >> 1) * DSK + cq * 4;   // cumulative skew: rows never overlap
      *reinterpret_cast<uint2*>(dh + dofs) = h;
      if (X3) {
        uint2 l;
        l.x = pack_bf16w(val.x - bf16_roundw(val.x), val.y - bf16_roundw(val.y));
        l.y = pack_bf16w(val.z - bf16_roundw(val.z), val.w - bf16_roundw(val.w));
        *reinterpret_cast<uint2*>(dl + dofs) = l;
      }
    }
    }
    __syncthreads();

    if (TAPSPLIT && geo_s1) {
      // ---- 3x3x3 stride-1 geometry (6 x 6 x 18 halo, 16 M-tile rows): everything but the per-lane bases is a compile-time
      // immediate, the bias row is folded in by operand selection (ones / zeros), and the transposed reads of tap step f+2
      // are issued before the MFMAs of step f (cf. wgrad16_kernel).  56 tap steps x NTW x 3 MFMAs per tile and wave.
      constexpr int DEPTH = 2;
      constexpr unsigned XLO = 36 * 20 * 16 * 2;                                     // x lo image (bytes past the hi image)
      constexpr unsigned DLO = (16 * (CGW == 16 ? 20 : 16) * CGW + (16 / 2 + 1) * (CGW == 32 ? 16 : 0)) * 2;   // dy lo image
      constexpr int DPc = CGW == 16 ? 20 : 16, DSKc = CGW == 32 ? 16 : 0;
      bf16x8 ah[DEPTH + 1], al[DEPTH + 1], bh[2][NTW], bl[2][NTW];
      auto trf = [&](unsigned addr, unsigned second) {       // two transposed reads -> one 8-element K fragment
        const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)addr);
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(addr + second));
        const s16x8 w = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
        return __builtin_bit_cast(bf16x8, w);
      };
      auto issue = [&](int f) {                              // f = ks * 7 + i, compile-time after unrolling
        const int ks = f / 7, i = f % 7;
        if (i == 0) {
          const unsigned od_ = (2 * ks * DPc * CGW + ks * DSKc) * 2;
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            bh[ks & 1][j] = trf(s1_da + od_ + j * 32, 4 * CGW * 2);
            if (X3) bl[ks & 1][j] = trf(s1_da + od_ + j * 32 + DLO, 4 * CGW * 2);
          }
        }
        const unsigned ox = (((ks >> 1) * 6 + ((2 * ks) & 3)) * 20) * 32;
        ah[f % (DEPTH + 1)] = trf(s1_xa[i] + ox, 4 * 32);
        if (X3) al[f % (DEPTH + 1)] = trf(s1_xa[i] + ox + XLO, 4 * 32);
      };
#pragma unroll
      for (int f = 0; f < DEPTH; ++f) issue(f);
#pragma unroll
      for (int f = 0; f < 56; ++f) {
        if (f + DEPTH < 56) issue(f + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        const int ks = f / 7, i = f % 7;
        bf16x8 ahf = ah[f % (DEPTH + 1)], alf = al[f % (DEPTH + 1)];
        if (i == 6) {                                        // wave 3: tap slot 27 = bias row (ones . dy); wave-uniform select
          ahf = s1_bias_wave ? ones : ahf;
          if (X3) alf = s1_bias_wave ? zeros8 : alf;
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahf, bh[ks & 1][j], acc[i][j], 0, 0, 0);
          if (X3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahf, bl[ks & 1][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alf, bh[ks & 1][j], acc[i][j], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      continue;
    }
    // ---- K loop: one step = 32 voxels = M-tile rows (2ks, 2ks+1); lane group kq -> row 2ks + (kq&1), voxels 8(kq>>1)..+7
    const int nks = (g.TD * g.TH) >> 1;
#pragma unroll 1
    for (int ks = 0; ks < nks; ++ks) {
      if (!TAPSPLIT && (ks & 3) != wave) continue;           // 1-tap ops: waves split the voxels (wave-uniform)
      const int mt = 2 * ks + (kq & 1);
      const int tw0 = (kq >> 1) * 8 + bq;                    // this lane's block row (first of the two 4-voxel blocks)
      const int vin = ((((mt / g.TH) * g.is) * g.IH + (mt % g.TH) * g.is) * XW + tw0 * g.is) * 16 + bp * 4;
      const int vin4 = vin + 4 * g.is * 16;
      const int vout = (mt * DP + tw0) * CGW + ((mt + 1) >> 1) * DSK + bp * 4;
      const int vout4 = vout + 4 * CGW;
      bf16x8 bh[NTW], bl[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        bh[j] = tr_frag(dh + vout + j * 16, dh + vout4 + j * 16);
        if (X3) bl[j] = tr_frag(dl + vout + j * 16, dl + vout4 + j * 16);
      }
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int t = TAPSPLIT ? wave + 4 * i : i;
        if (t > ntaps) continue;                             // wave-uniform
        if (t == ntaps) {                                    // bias row: ones . dy
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bh[j], acc[i][j], 0, 0, 0);
            if (X3) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bl[j], acc[i][j], 0, 0, 0);
          }
        } else {
          const int to = tapo[i];
          const bf16x8 ah = tr_frag(xh + vin + to, xh + vin4 + to);
          bf16x8 al;
          if (X3) al = tr_frag(xl + vin + to, xl + vin4 + to);
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
            if (X3) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  float4* out = reinterpret_cast<float4*>(apart_ + (int64_t)(TAPSPLIT ? split : split * 4 + wave) * a.slab_floats);
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = TAPSPLIT ? wave + 4 * i : i;
    if (t > ntaps) continue;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int64_t blk = (int64_t)a.cls_slab_base[cls] + (((int64_t)chunk * a.ngroups + grp) * (ntaps + 1) + t) * CG + j;
      out[blk * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// wgrad16: weight gradient of the full-resolution 3x3x3 convolutions with <= 16 input and 16 output channels
// (the conv16 layers).  Same producer / consumer structure as conv16_kernel (conv_bf16.hip), for the same measured reasons:
//   * 8-wave persistent workgroups, one per CU, each owning a CONTIGUOUS range of 4x4x16 tiles;
//   * waves 4-7 (loaders) keep two tiles of global loads in flight (x halo 6x6x18 + dy 4x4x16, fp32), apply the recomputed
//     InstanceNorm + activation prologue to x, split both to bf16 hi/lo and fill the other LDS buffer -- straight-line code so
//     that the compiler emits counted vmcnt waits;
//   * waves 0-3 (MFMA) only do transposed LDS reads + MFMAs; wave w owns taps w, w+4, ... (slot 27 = bias row) and keeps its
//     7 accumulators in registers across ALL tiles of the workgroup: no per-tile epilogue at all;
//   * raw s_barrier (lgkmcnt only) hands the buffers over; one slab per workgroup at the end (reduced by cwf_wgrad_reduce).
// ---------------------------------------------------------------------------------------------------
#ifndef CWF_W16_DEPTH
#define CWF_W16_DEPTH 7
#endif
#ifndef CWF_WS1_DEPTH
#define CWF_WS1_DEPTH 5
#endif
#define W16_NVOX 648
#define W16_XW 20                                       // LDS row pitch of the x image in voxels (18 + 2 pad), see below
#define W16_DW 20                                       // LDS row pitch of the dy image in voxels (16 + 4 pad)
#define W16_LW 6                                        // loader waves (MFMA waves: 4) -> 640-thread workgroups
#define W16_VPP (W16_LW * 16)                           // voxels staged per pass (4 threads per voxel)
#define W16_XS ((W16_NVOX + W16_VPP - 1) / W16_VPP)     // 7 x staging slots per loader thread
#define W16_DS ((256 + W16_VPP - 1) / W16_VPP)          // 3 dy staging slots per loader thread

typedef __bf16 bf16x2_w __attribute__((ext_vector_type(2)));
typedef float f32x2_w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const f32x2_w f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_w));
}
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pk_bf16(a, b);
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  lo = pk_bf16(a - ha, b - hb);
}

template <bool X3>
__global__ __launch_bounds__(256 + 64 * W16_LW) void wgrad16_kernel(const WgArgsB a, int total_tiles) {
  extern __shared__ float4 lds4[];
  const ConvGeom& g = a.g;
  // LDS images are voxel-major [row][voxel][16 ch] bf16 with rows PADDED to 20 voxels (640 B): the two M-tile rows that the
  // lanes of one half-wave read in a transposed fragment then start 128 B apart modulo the 256-B bank period, so the
  // ds_read_b64_tr_b16 are conflict-free (unpadded: 2-way conflicts, SQ_LDS_BANK_CONFLICT ~ 2k cycles per tile).
  constexpr int XI = 36 * W16_XW * 16;                 // bf16 elements of one x image  (6 x 6 rows)
  constexpr int DI = 16 * W16_DW * 16;                 // bf16 elements of one dy image (4 x 4 rows)
  constexpr int BUF = (XI + DI) * (X3 ? 2 : 1);        // per buffer: xh [xl] dh [dl]
  unsigned short* lds = reinterpret_cast<unsigned short*>(lds4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the role branch below is provably wave-uniform
  const int tiles_sp = g.tiles_d * g.tiles_h * g.tiles_w;
  // XCD-aware tile map (see conv16_kernel): iteration `it` covers tiles [it*G, (it+1)*G); XCD x takes segment x of it
  const int G = (int)gridDim.x, per = G >> 3;
  const int first = (blockIdx.x & 7) * per + (blockIdx.x >> 3);      // tile(it) = first + it * G
  const int niter = first < total_tiles ? (total_tiles - first + G - 1) / G : 0;   // empty workgroups still write a zero slab

  if (wave < 4) {
    // =============================================================== MFMA waves
    const int kq = lane >> 4;
    const int bq = (lane & 15) >> 2, bp = lane & 3;
    f32x4 acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
    // K = 32 voxels per MFMA = M-tile rows (2ks, 2ks+1) x 16 voxels.  Lane group kq reads row 2ks + (kq & 1), voxels
    // 8 (kq >> 1) + [0, 8): the two rows met inside one half-wave are 640 B apart (see the padding note above).
    // Addresses are integers (LDS byte addresses): per-tap bases in registers, everything else in the ds_read immediate, so
    // the fully unrolled loop has no address arithmetic (same vector-issue argument as conv16_kernel).
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds4;
    const unsigned lane_x = lds_base + (((kq & 1) * W16_XW + (kq >> 1) * 8 + bq) * 16 + bp * 4) * 2;
    unsigned xa[7];                                      // x hi image: lane part + tap offset (+ buffer parity)
    unsigned da = lds_base + XI * 2 * (X3 ? 2 : 1) + (((kq & 1) * W16_DW + (kq >> 1) * 8 + bq) * 16 + bp * 4) * 2;   // dy hi image
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int t = wave + 4 * i;
      xa[i] = lane_x + (t < 27 ? (((t / 9) * 6 + (t / 3) % 3) * W16_XW + t % 3) * 32 : 0);
    }
    auto trf = [&](unsigned addr) {                      // two transposed reads (block rows +0, +4) -> one 8-element K fragment
      const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)addr);
      const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(addr + 4 * 32));
      const s16x8 w = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
      return __builtin_bit_cast(bf16x8, w);
    };
    // Two straight-line copies of the whole tile loop (wave 3's last slot is the bias row).  A tile is 8 K-steps x 7 taps =
    // 56 "tap steps" of 3 MFMAs (48 cycles); the 4 transposed reads of tap step f+2 are issued before the MFMAs of step f
    // (explicit register double-buffering, pinned with sched_barrier: left alone, the compiler issues each read right
    // before its MFMA and every MFMA waits out the LDS latency).
    auto tiles = [&](auto BIAS_) {
      constexpr bool BIAS = decltype(BIAS_)::value;
      // LDS-read lookahead in tap steps.  Split operands: a step is 3 MFMAs = 48 cycles, two steps cover the transposed-read latency.
      // Single-bf16 operands (what the bench runs): ONE MFMA = 16 cycles per step -- at two steps every MFMA waited out most of an
      // LDS round trip (5.6k cycles per tile against 0.9k of MFMAs); seven steps ahead (the most the two-slot dy-fragment ring allows), in registers the lo fragments do not need.
      constexpr int DEPTH = X3 ? 2 : CWF_W16_DEPTH;
      for (int it = 0; it < niter; ++it) {
        asm volatile("s_barrier" ::: "memory");          // buffer it&1 is complete
        bf16x8 ah[DEPTH + 1], al[DEPTH + 1], bh[2], bl[2];
        auto issue = [&](int f) {                        // f = ks * 7 + i, compile-time after unrolling
          const int ks = f / 7, i = f % 7;
          if (i == 0) {
            const unsigned od_ = (2 * ks * W16_DW) * 32;
            bh[ks & 1] = trf(da + od_);
            if (X3) bl[ks & 1] = trf(da + od_ + DI * 2);
          }
          if (!(BIAS && i == 6)) {
            const unsigned ox = (((ks >> 1) * 6 + ((2 * ks) & 3)) * W16_XW) * 32;
            ah[f % (DEPTH + 1)] = trf(xa[i] + ox);
            if (X3) al[f % (DEPTH + 1)] = trf(xa[i] + ox + XI * 2);
          }
        };
#pragma unroll
        for (int f = 0; f < DEPTH; ++f) issue(f);
#pragma unroll
        for (int f = 0; f < 56; ++f) {
          if (f + DEPTH < 56) issue(f + DEPTH);
          __builtin_amdgcn_sched_barrier(0);
          const int ks = f / 7, i = f % 7;
          const bf16x8 bhf = bh[ks & 1];
          if (BIAS && i == 6) {                          // tap slot 27 = bias row (ones . dy)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bhf, acc[i], 0, 0, 0);
            if (X3) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bl[ks & 1], acc[i], 0, 0, 0);
          } else {
            const bf16x8 ahf = ah[f % (DEPTH + 1)];
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahf, bhf, acc[i], 0, 0, 0);
            if (X3) {
              acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahf, bl[ks & 1], acc[i], 0, 0, 0);
              acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[f % (DEPTH + 1)], bhf, acc[i], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        {                                                // the other buffer is read next
          const unsigned dlt = (it & 1) ? (unsigned)(-(BUF * 2)) : (unsigned)(BUF * 2);
#pragma unroll
          for (int i = 0; i < 7; ++i) xa[i] += dlt;
          da += dlt;
        }
      }
    };
    if (wave == 3) tiles(std::true_type{}); else tiles(std::false_type{});
    // one slab per workgroup: [tap slot 0..27][lane][4]   (chunk 0, group 0, CG = 1)
    float4* out = reinterpret_cast<float4*>(a.partial + (int64_t)blockIdx.x * a.slab_floats);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int t = wave + 4 * i;
      if (t > 27) continue;
      out[(int64_t)t * 64 + lane] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
  } else {
    // =============================================================== loader waves
    const int lt = tid - 256;
    const int q = lt & 3, c = q * 4;
    const bool cval = c < g.Cin;
    const bool has_norm = a.in_scale != nullptr;
    const float slope = a.in_slope;
    const float sl = (has_norm || slope != 1.f) ? slope : 1.f;
    const int HW = g.Hi * g.Wi;
    int relx[W16_XS], reld[W16_DS];
#pragma unroll
    for (int i = 0; i < W16_XS; ++i) {
      const int v = (lt >> 2) + W16_VPP * i;
      const int iw = v % 18, t2 = v / 18;
      relx[i] = ((t2 / 6) * HW + (t2 % 6) * g.Wi + iw) * g.x_ldc + c;
    }
#pragma unroll
    for (int i = 0; i < W16_DS; ++i) {
      const int v = (lt >> 2) + W16_VPP * i;                  // output voxel: td = v>>6, th = (v>>4)&3, tw = v&15
      reld[i] = (((v >> 6) * g.Ho + ((v >> 4) & 3)) * g.Wo + (v & 15)) * a.dy_ldc + c;
    }
    const bool last_ok = (lt >> 2) + W16_VPP * (W16_XS - 1) < W16_NVOX;
    const bool dlast_ok = (lt >> 2) + W16_VPP * (W16_DS - 1) < 256;
    int xo[W16_XS], dofs[W16_DS];                        // LDS element offsets of this thread's staging slots (padded rows)
#pragma unroll
    for (int i = 0; i < W16_XS; ++i) { const int v = (lt >> 2) + W16_VPP * i; xo[i] = ((v / 18) * W16_XW + v % 18) * 16 + q * 4; }
#pragma unroll
    for (int i = 0; i < W16_DS; ++i) { const int v = (lt >> 2) + W16_VPP * i; dofs[i] = ((v >> 4) * W16_DW + (v & 15)) * 16 + q * 4; }
    float4 px[1][W16_XS], pd[1][W16_DS];
    unsigned inbx[1] = {0u}, inbd[1] = {0u};
    __builtin_amdgcn_s_setprio(1);
    struct Org { const float* xb; const float* db; bool interior; int id0, ih0, iw0, od0, oh0, ow0, n; };
    auto origin = [&](int tile) {
      Org o;
      o.n = tile / tiles_sp; int rem = tile - o.n * tiles_sp;
      const int tile_w = rem % g.tiles_w; rem /= g.tiles_w;
      const int tile_h = rem % g.tiles_h; const int tile_d = rem / g.tiles_h;
      o.od0 = tile_d * 4; o.oh0 = tile_h * 4; o.ow0 = tile_w * 16;
      o.id0 = o.od0 - 1; o.ih0 = o.oh0 - 1; o.iw0 = o.ow0 - 1;
      o.xb = a.x + ((((int64_t)o.n * g.Di + o.id0) * g.Hi + o.ih0) * g.Wi + o.iw0) * g.x_ldc;
      o.db = a.dy + ((((int64_t)o.n * g.Do + o.od0) * g.Ho + o.oh0) * g.Wo + o.ow0) * a.dy_ldc;
      o.interior = o.id0 >= 0 && o.id0 + 6 <= g.Di && o.ih0 >= 0 && o.ih0 + 6 <= g.Hi && o.iw0 >= 0 && o.iw0 + 18 <= g.Wi &&
                   o.od0 + 4 <= g.Do && o.oh0 + 4 <= g.Ho && o.ow0 + 16 <= g.Wo;
      return o;
    };
    auto issue = [&](const Org& o, auto S) {
      constexpr int SET = decltype(S)::value;
      unsigned mx = 0, md = 0;
      if (o.interior) {
#pragma unroll
        for (int i = 0; i < W16_XS; ++i) {
          const bool ok = cval && (i < W16_XS - 1 || last_ok);
          px[SET][i] = *reinterpret_cast<const float4*>(ok ? o.xb + relx[i] : a.x);
          mx |= ok ? (1u << i) : 0u;
        }
#pragma unroll
        for (int i = 0; i < W16_DS; ++i) {
          const bool ok = i < W16_DS - 1 || dlast_ok;
          pd[SET][i] = *reinterpret_cast<const float4*>(ok ? o.db + reld[i] : a.dy);
          md |= ok ? (1u << i) : 0u;
        }
      } else {
#pragma unroll
        for (int i = 0; i < W16_XS; ++i) {
          const int v = (lt >> 2) + W16_VPP * i;
          const int iw = v % 18, t2 = v / 18;
          const int gd = o.id0 + t2 / 6, gh = o.ih0 + t2 % 6, gw = o.iw0 + iw;
          const bool ok = cval && (i < W16_XS - 1 || last_ok) && (unsigned)gd < (unsigned)g.Di && (unsigned)gh < (unsigned)g.Hi && (unsigned)gw < (unsigned)g.Wi;
          px[SET][i] = *reinterpret_cast<const float4*>(ok ? o.xb + relx[i] : a.x);
          mx |= ok ? (1u << i) : 0u;
        }
#pragma unroll
        for (int i = 0; i < W16_DS; ++i) {
          const int v = (lt >> 2) + W16_VPP * i;
          const bool ok = (i < W16_DS - 1 || dlast_ok) && o.od0 + (v >> 6) < g.Do && o.oh0 + ((v >> 4) & 3) < g.Ho && o.ow0 + (v & 15) < g.Wo;
          pd[SET][i] = *reinterpret_cast<const float4*>(ok ? o.db + reld[i] : a.dy);
          md |= ok ? (1u << i) : 0u;
        }
      }
      inbx[SET] = mx; inbd[SET] = md;
    };
    auto convert = [&](int tc, int buf, auto S) {
      constexpr int SET = decltype(S)::value;
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_norm && cval) {
        const int n = tc / tiles_sp;
        sc = *reinterpret_cast<const float4*>(a.in_scale + (int64_t)n * g.Cin + c);
        sh = *reinterpret_cast<const float4*>(a.in_shift + (int64_t)n * g.Cin + c);
      }
      float4 dsc = make_float4(1.f, 1.f, 1.f, 1.f);
      if (a.dy_scale) dsc = *reinterpret_cast<const float4*>(a.dy_scale + (int64_t)(tc / tiles_sp) * g.Cout + c);
      unsigned short* xh = lds + buf * BUF;
      unsigned short* xl = xh + XI;
      unsigned short* dh = lds + buf * BUF + XI * (X3 ? 2 : 1);
      unsigned short* dl = dh + DI;
      const unsigned mx = inbx[SET], md = inbd[SET];
      const unsigned fullx = last_ok ? ((1u << W16_XS) - 1u) : ((1u << (W16_XS - 1)) - 1u);
      const unsigned fulld = dlast_ok ? ((1u << W16_DS) - 1u) : ((1u << (W16_DS - 1)) - 1u);
      // specialised on two wave-uniform facts (cf. conv16_kernel): PLAIN = no norm/activation, ALLIN = nothing out of bounds
      auto body = [&](auto PL, auto AI) {
        constexpr bool PLAIN = decltype(PL)::value, ALLIN = decltype(AI)::value;
#pragma unroll
        for (int i = 0; i < W16_XS; ++i) {
          if (i == W16_XS - 1 && !last_ok) continue;
          const float4 val = px[SET][i];
          float v0 = val.x, v1 = val.y, v2 = val.z, v3 = val.w;
          if (!PLAIN) {
            v0 = fmaxf(fmaf(v0, sc.x, sh.x), fmaf(v0, sc.x, sh.x) * sl); v1 = fmaxf(fmaf(v1, sc.y, sh.y), fmaf(v1, sc.y, sh.y) * sl);
            v2 = fmaxf(fmaf(v2, sc.z, sh.z), fmaf(v2, sc.z, sh.z) * sl); v3 = fmaxf(fmaf(v3, sc.w, sh.w), fmaf(v3, sc.w, sh.w) * sl);
          }
          uint2 h, l;
          if (X3) { split2(v0, v1, h.x, l.x); split2(v2, v3, h.y, l.y); } else { h.x = pk_bf16(v0, v1); h.y = pk_bf16(v2, v3); }
          if (!ALLIN) {                                  // zero padding applies after the activation
            const bool was = (mx >> i) & 1u;
            h.x = was ? h.x : 0u; h.y = was ? h.y : 0u;
            if (X3) { l.x = was ? l.x : 0u; l.y = was ? l.y : 0u; }
          }
          *reinterpret_cast<uint2*>(xh + xo[i]) = h;
          if (X3) *reinterpret_cast<uint2*>(xl + xo[i]) = l;
        }
#pragma unroll
        for (int i = 0; i < W16_DS; ++i) {
          if (i == W16_DS - 1 && !dlast_ok) continue;
          float4 val = pd[SET][i];
          val.x *= dsc.x; val.y *= dsc.y; val.z *= dsc.z; val.w *= dsc.w;
          uint2 h, l;
          if (X3) { split2(val.x, val.y, h.x, l.x); split2(val.z, val.w, h.y, l.y); } else { h.x = pk_bf16(val.x, val.y); h.y = pk_bf16(val.z, val.w); }
          if (!ALLIN) {
            const bool was = (md >> i) & 1u;
            h.x = was ? h.x : 0u; h.y = was ? h.y : 0u;
            if (X3) { l.x = was ? l.x : 0u; l.y = was ? l.y : 0u; }
          }
          *reinterpret_cast<uint2*>(dh + dofs[i]) = h;
          if (X3) *reinterpret_cast<uint2*>(dl + dofs[i]) = l;
        }
      };
      using T_ = std::true_type; using F_ = std::false_type;
      const bool allin = __ballot(mx != fullx || md != fulld) == 0ull;     // wave-uniform
      const bool plain = !has_norm && sl == 1.f;
      if (plain) { if (allin) body(T_{}, T_{}); else body(T_{}, F_{}); }
      else       { if (allin) body(F_{}, T_{}); else body(F_{}, F_{}); }
    };
    using S0 = std::integral_constant<int, 0>;
    // one prefetch set: tile it+1 is converted right after the barrier, then tile it+2 is requested (a second register set
    // spills at this kernel's 168-VGPR budget; measured slower)
    if (niter > 0) {
      issue(origin(first), S0{});
      convert(first, 0, S0{});
      if (niter > 1) issue(origin(first + G), S0{});
    }
    for (int it = 0; it < niter; ++it) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (it + 1 < niter) {
        convert(first + (it + 1) * G, (it + 1) & 1, S0{});
        if (it + 2 < niter) issue(origin(first + (it + 2) * G), S0{});
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// wgrad16d: wgrad16 on bf16 OPERAND IMAGES filled by LDS-DMA.  Both operands arrive as bf16 tensors [N][D][H][W][16] -- xa16 =
// bf16(act(IN(x))) written as a side output by the layer's own InstanceNorm-backward apply pass (which has x and its statistics in
// registers anyway), dy16 = the bf16 image of the incoming gradient written by the pass that produced it (norm.hip) -- i.e. exactly
// the values wgrad16's loaders compute per tile from the fp32 tensors (single-bf16 operand products; bit-identical operands).
// What that buys (profiles/round3_conv16s_wgrad_pmc.txt: wgrad16 moves 1.9x its algorithmic bytes from L2 and is bound by one
// tile of register-staged loads in flight per CU, 26 GB/s per CU at the step's 128 workgroups):
//   * half the bytes (32 B per voxel and operand instead of 64);
//   * no conversion, no registers: the six loader waves only issue global_load_lds_dwordx4 pieces (1 KiB of LDS each, 33 per tile),
//     so THREE tiles are in flight per CU in a four-buffer LDS ring behind counted vmcnt waits and one raw s_barrier per tile;
//   * zero padding = out-of-range lanes read a 16-byte zero page.
// LDS image = wgrad16's (voxel-major rows padded to 20 voxels = 640 B = 40 granules of 16 B: conflict-free transposed reads); a
// piece is 64 consecutive granules, per-lane source address; pad granules read the zero page.  MFMA waves: wgrad16's.
// ---------------------------------------------------------------------------------------------------
#define W16D_XP 23                                      // DMA pieces of the x image (36 rows x 40 granules = 1440 -> 23 pieces)
#define W16D_DP 10                                      // DMA pieces of the dy image (16 rows x 40 granules = 640)
#define W16D_XIB (W16D_XP * 1024)
#define W16D_BUFB ((W16D_XP + W16D_DP) * 1024)          // 33,792 B per buffer
#define W16D_NBUF 4
#define W16D_LW 6                                       // loader waves
struct W16dArgs {
  const uint4* xa; const uint4* dy; const uint4* zero; float* partial;
  int N, D, H, W, tiles_d, tiles_h, tiles_w, total_tiles;
  int64_t slab_floats;
};

__global__ __launch_bounds__(256 + 64 * W16D_LW) void wgrad16d_kernel(const W16dArgs a) {
  extern __shared__ float4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_sp = a.tiles_d * a.tiles_h * a.tiles_w;
  const int G = (int)gridDim.x, per = G >> 3;
  const int first = (blockIdx.x & 7) * per + (blockIdx.x >> 3);      // tile(it) = first + it * G   (XCD-aware, see wgrad16_kernel)
  const int niter = first < a.total_tiles ? (a.total_tiles - first + G - 1) / G : 0;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds4;

  if (wave < 4) {
    // =============================================================== MFMA waves (wgrad16_kernel<false>, four buffers)
    const int kq = lane >> 4;
    const int bq = (lane & 15) >> 2, bp = lane & 3;
    f32x4 acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
    const unsigned lane_x = lds_base + (((kq & 1) * W16_XW + (kq >> 1) * 8 + bq) * 16 + bp * 4) * 2;
    unsigned xa[7];
    unsigned da = lds_base + W16D_XIB + (((kq & 1) * W16_DW + (kq >> 1) * 8 + bq) * 16 + bp * 4) * 2;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int t = wave + 4 * i;
      xa[i] = lane_x + (t < 27 ? (((t / 9) * 6 + (t / 3) % 3) * W16_XW + t % 3) * 32 : 0);
    }
    auto trf = [&](unsigned addr) {
      const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)addr);
      const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(addr + 4 * 32));
      const s16x8 w = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
      return __builtin_bit_cast(bf16x8, w);
    };
    auto tiles = [&](auto BIAS_) {
      constexpr bool BIAS = decltype(BIAS_)::value;
      constexpr int DEPTH = CWF_W16_DEPTH;
      for (int it = 0; it < niter; ++it) {
        asm volatile("s_barrier" ::: "memory");          // buffer it & 3 has landed
        bf16x8 ah[DEPTH + 1], bh[2];
        auto issue = [&](int f) {
          const int ks = f / 7, i = f % 7;
          if (i == 0) bh[ks & 1] = trf(da + (2 * ks * W16_DW) * 32);
          if (!(BIAS && i == 6)) ah[f % (DEPTH + 1)] = trf(xa[i] + (((ks >> 1) * 6 + ((2 * ks) & 3)) * W16_XW) * 32);
        };
#pragma unroll
        for (int f = 0; f < DEPTH; ++f) issue(f);
#pragma unroll
        for (int f = 0; f < 56; ++f) {
          if (f + DEPTH < 56) issue(f + DEPTH);
          __builtin_amdgcn_sched_barrier(0);
          const int ks = f / 7, i = f % 7;
          const bf16x8 bhf = bh[ks & 1];
          if (BIAS && i == 6) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bhf, acc[i], 0, 0, 0);
          else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f % (DEPTH + 1)], bhf, acc[i], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        {
          const unsigned dlt = ((it & 3) == 3) ? (unsigned)(-(3 * W16D_BUFB)) : (unsigned)W16D_BUFB;
#pragma unroll
          for (int i = 0; i < 7; ++i) xa[i] += dlt;
          da += dlt;
        }
      }
    };
    if (wave == 3) tiles(std::true_type{}); else tiles(std::false_type{});
    float4* out = reinterpret_cast<float4*>(a.partial + (int64_t)blockIdx.x * a.slab_floats);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int t = wave + 4 * i;
      if (t > 27) continue;
      out[(int64_t)t * 64 + lane] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
  } else {
    // =============================================================== loader waves: DMA only
    const int lw = wave - 4;                               // pieces lw, lw + 6, ... of the 33
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* glb_vp;
    auto run = [&](auto NI_) __attribute__((always_inline)) {
      constexpr int NI = decltype(NI_)::value;
      int off[NI]; unsigned crd[NI];                       // granule offset from the tile's operand origin; (c0, c1, w, valid)
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int s = lw + W16D_LW * k;
        const bool isd = s >= W16D_XP;
        const int gi = 64 * (isd ? s - W16D_XP : s) + lane;
        const int row = gi / 40, gr = gi % 40, w = gr >> 1, half = gr & 1;
        const int c0 = isd ? (row >> 2) : row / 6, c1 = isd ? (row & 3) : row % 6;
        const bool valid = isd ? gr < 32 : (row < 36 && gr < 36);
        off[k] = ((c0 * a.H + c1) * a.W + w) * 2 + half;
        crd[k] = (unsigned)c0 | ((unsigned)c1 << 3) | ((unsigned)w << 6) | (valid ? 1u << 11 : 0u);
      }
      auto issue = [&](int it) __attribute__((always_inline)) {
        const int tile = first + it * G;
        const int n = tile / tiles_sp; int rem = tile - n * tiles_sp;
        const int tile_w = rem % a.tiles_w; rem /= a.tiles_w;
        const int tile_h = rem % a.tiles_h; const int tile_d = rem / a.tiles_h;
        const int od0 = tile_d * 4, oh0 = tile_h * 4, ow0 = tile_w * 16;
        const int64_t vd = (((int64_t)n * a.D + od0) * a.H + oh0) * a.W + ow0;       // voxel index of the dy tile origin
        const int64_t vx = vd - ((int64_t)a.H + 1) * a.W - 1;                        // ... of the x halo origin (-1, -1, -1)
        const unsigned lbuf = lds_base + (unsigned)(it & 3) * W16D_BUFB;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
          const int s = lw + W16D_LW * k;                 // (wave-uniform)
          const bool isd = s >= W16D_XP;
          const int c0 = (int)(crd[k] & 7u), c1 = (int)((crd[k] >> 3) & 7u), w = (int)((crd[k] >> 6) & 31u);
          const int gd = (isd ? od0 : od0 - 1) + c0, gh = (isd ? oh0 : oh0 - 1) + c1, gw = (isd ? ow0 : ow0 - 1) + w;
          const bool ok = (crd[k] >> 11) != 0u && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
          const uint4* src = (isd ? a.dy + vd * 2 : a.xa + vx * 2) + off[k];
          src = ok ? src : a.zero;
          __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(uintptr_t)(lbuf + (unsigned)s * 1024u), 16, 0, 0);
        }
      };
      if (niter > 0) issue(0);
      if (niter > 1) issue(1);
      if (niter > 2) issue(2);
      for (int it = 0; it < niter; ++it) {
        // tile `it` has landed once all but the pieces of the (at most two) younger tiles in flight are done
        if (it + 2 < niter) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (it + 3 < niter) issue(it + 3);               // into the buffer the MFMA waves finished before this barrier
      }
    };
    if (lw < (W16D_XP + W16D_DP) % W16D_LW) run(std::integral_constant<int, (W16D_XP + W16D_DP) / W16D_LW + 1>{});
    else run(std::integral_constant<int, (W16D_XP + W16D_DP) / W16D_LW>{});
  }
}

extern "C" int cwf_wgrad16_bf16(const void* xa16, const void* dy16, const void* zero16, float* partial,
                                int N, int D, int H, int W, int* nsplit_used, void* stream) {
  if (!xa16 || !dy16 || !zero16 || !partial || N <= 0 || D <= 0 || H <= 0 || W <= 0) return CWF_E_BADARG;
  if (((uintptr_t)xa16 & 15) || ((uintptr_t)dy16 & 15) || ((uintptr_t)zero16 & 15) || ((uintptr_t)partial & 15)) return CWF_E_ALIGN;
  if ((int64_t)N * D * H * W >= (1ll << 30)) return CWF_E_TOOLARGE;
  W16dArgs a;
  a.xa = (const uint4*)xa16; a.dy = (const uint4*)dy16; a.zero = (const uint4*)zero16; a.partial = partial;
  a.N = N; a.D = D; a.H = H; a.W = W;
  a.tiles_d = cdiv(D, 4); a.tiles_h = cdiv(H, 4); a.tiles_w = cdiv(W, 16);
  a.total_tiles = N * a.tiles_d * a.tiles_h * a.tiles_w;
  a.slab_floats = 28 * 256;                               // = cwf_wgrad_slab_floats(CWF_CONV3_S1, 16, 16)
  int grid = side_wgs(); while (grid > 8 && grid > a.total_tiles) grid -= 8;      // multiple of 8 (XCD-aware tile map)
  if (grid > cwf_wgrad_nsplit(CWF_CONV3_S1, N, D, H, W, 16, 16)) return CWF_E_BADARG;    // (the caller's slab buffer is sized by it)
  const size_t lds = (size_t)W16D_NBUF * W16D_BUFB;
  CWF_MAX_LDS_ONCE((&wgrad16d_kernel));
  hipLaunchKernelGGL(wgrad16d_kernel, dim3(grid), dim3(256 + 64 * W16D_LW), lds, cwf_stream(stream), a);
  CWF_LAUNCH_CHECK();
  if (nsplit_used) *nsplit_used = grid;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// wgrad_s1: the 3x3x3 stride-1 layers the generic kernel above used to take (32 / 64 / 128 channels; 38 launches per step, the
// largest single item of the step's kernel time), single-bf16 operands, restructured like wgrad16_kernel.  The generic kernel
// stages a tile with one DEPENDENT global load per staging slot (11 x slots + CGW/4 dy slots per thread: load, wait, convert,
// store, next) and only then starts its MFMA phase: ~19 exposed round trips per tile, 4.6 us against 0.75 us of MFMAs
// (32 ch @ 64^3: 111 us per launch under rocprofv3, tools/wgrad_micro.py).  Here a workgroup is 12 waves, one per CU:
//   * waves 4-11 (loaders) convert tile t+1 (recomputed InstanceNorm + activation for x) into the OTHER LDS buffer, then issue ALL
//     loads of tile t+3 (straight-line, masked by pointer select) into the register set that frees: two tiles of loads in flight
//     (eight loader waves keep a set at 8-10 float4 per thread; with four, two sets spill and one set is latency-bound: 59 us);
//   * waves 0-3 (MFMA) run the 56-step transposed-read / MFMA phase of tile t (wave w owns taps w, w+4, ...; slot 27 = bias row)
//     with accumulators persistent over the workgroup's whole tile range;
//   * one raw s_barrier per tile hands the buffers over; one slab per workgroup, fewer and longer workgroups than the generic
//     plan (256 in all), so the slab traffic and the reduce halve as well.
// Same (chunk, channel-group) blocking, LDS image layout (padded / skewed rows: conflict-free transposed reads) and slab layout
// as wgrad_bf16_kernel<7, NTW, true, false>.  Measured (slab kernel, rocprofv3): 32 ch @ 64^3 111 -> 50 us, 64 ch @ 32^3 60 -> 28 us,
// 128 ch @ 16^3 37 -> 22 us.  At 50 us the (tile, chunk) units move 300 MB of L2 -> CU traffic (x halo 41 KB + dy 32 KB each;
// 2.2x the 134 MB the layer reads from HBM) = 6.0 TB/s -- the same aggregate ingest rate wgrad16_kernel sits at (934 MB in
// 150 us), i.e. the CU <- L2 fabric, not the MFMA phase (a deeper LDS-read lookahead changed nothing).  What is left is traffic:
// both chunks of a tile against one dy fetch, a sliding halo.
// ---------------------------------------------------------------------------------------------------
#define WS1_LW 8                                        // loader waves (MFMA waves: 4) -> 768-thread workgroups
template <int NTW>
__global__ __launch_bounds__(256 + 64 * WS1_LW) void wgrad_s1_kernel(const WgArgsB a) {
  constexpr int CG = NTW, CGW = CG * 16;
  constexpr int XW = 20, DP = CGW == 16 ? 20 : 16, DSK = CGW == 32 ? 16 : 0;
  constexpr int XIMG = 36 * XW * 16;                                   // bf16 elements of the x image
  constexpr int DIMG = 16 * DP * CGW + (16 / 2 + 1) * DSK;             // bf16 elements of the dy image
  constexpr unsigned BUFB = (unsigned)(XIMG + DIMG) * 2u;              // bytes of one buffer (x | dy)
  static_assert(BUFB % 16 == 0, "buffer pitch");
  static_assert(CWF_WS1_DEPTH < 8 && CWF_W16_DEPTH < 8, "the dy fragment of K-step ks+2 reuses the registers of K-step ks");
  extern __shared__ float4 lds4[];
  const ConvGeom& g = a.g;
  unsigned short* lds = reinterpret_cast<unsigned short*>(lds4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x;
  const int chunk = blockIdx.y / a.ngroups, grp = blockIdx.y % a.ngroups;
  const float* ax_ = a.groups ? a.x_g[blockIdx.z] : a.x;             // (workgroup-uniform: group operands of a grouped launch)
  const float* ady_ = a.groups ? a.dy_g[blockIdx.z] : a.dy;
  float* apart_ = a.groups ? a.partial_g[blockIdx.z] : a.partial;
  const int co0 = grp * CGW;
  const int tiles_sp = g.tiles_d * g.tiles_h * g.tiles_w;
  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(a.total_tiles, t_begin + a.tiles_per_split);
  const int niter = max(t_end - t_begin, 0);

  if (wave < 4) {
    // =============================================================== MFMA waves
    const int kq = lane >> 4, bq = (lane & 15) >> 2, bp = lane & 3;
    f32x4 acc[7][NTW];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
    const bool bias_wave = wave == 3;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds4;
    const unsigned lane_x = lds_base + (((kq & 1) * XW + (kq >> 1) * 8 + bq) * 16 + bp * 4) * 2;
    unsigned xa0[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int t = wave + 4 * i;
      xa0[i] = lane_x + (t < 27 ? (((t / 9) * 6 + (t / 3) % 3) * XW + t % 3) * 32 : 0);
    }
    const unsigned da0 = lds_base + XIMG * 2 + (((kq & 1) * DP + (kq >> 1) * 8 + bq) * CGW + (kq & 1) * DSK + bp * 4) * 2;
    auto trf = [&](unsigned addr, unsigned second) __attribute__((always_inline)) {   // two transposed reads -> one K fragment
      const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)addr);
      const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(addr + second));
      const s16x8 w = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
      return __builtin_bit_cast(bf16x8, w);
    };
    for (int it = 0; it < niter; ++it) {
      asm volatile("s_barrier" ::: "memory");              // buffer it & 1 is complete
      const unsigned bo = (it & 1) ? BUFB : 0u;
      unsigned xa[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) xa[i] = xa0[i] + bo;
      const unsigned da = da0 + bo;
      constexpr int DEPTH = CWF_WS1_DEPTH;               // tap steps of LDS-read lookahead (a step is NTW MFMAs = 16-32 cycles)
      bf16x8 ah[DEPTH + 1], bh[2][NTW];
      auto issue = [&](int f) __attribute__((always_inline)) {   // f = ks * 7 + i, compile-time after unrolling
        const int ks = f / 7, i = f % 7;
        if (i == 0) {
          const unsigned od_ = (2 * ks * DP * CGW + ks * DSK) * 2;
#pragma unroll
          for (int j = 0; j < NTW; ++j) bh[ks & 1][j] = trf(da + od_ + j * 32, 4 * CGW * 2);
        }
        const unsigned ox = (((ks >> 1) * 6 + ((2 * ks) & 3)) * XW) * 32;
        ah[f % (DEPTH + 1)] = trf(xa[i] + ox, 4 * 32);
      };
#pragma unroll
      for (int f = 0; f < DEPTH; ++f) issue(f);
#pragma unroll
      for (int f = 0; f < 56; ++f) {
        if (f + DEPTH < 56) issue(f + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        const int ks = f / 7, i = f % 7;
        bf16x8 ahf = ah[f % (DEPTH + 1)];
        if (i == 6) ahf = bias_wave ? ones : ahf;            // wave 3: tap slot 27 = bias row (ones . dy); wave-uniform select
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahf, bh[ks & 1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- slab (layout of wgrad_bf16_kernel: block = (chunk, group, tap, tile))
    float4* out = reinterpret_cast<float4*>(apart_ + (int64_t)split * a.slab_floats);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int t = wave + 4 * i;
      if (t > 27) continue;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int64_t blk = (((int64_t)chunk * a.ngroups + grp) * 28 + t) * CG + j;
        out[blk * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
  } else {
    // =============================================================== loader waves
    const int lt = tid - 256;
    constexpr int LT = 64 * WS1_LW;                         // loader threads
    constexpr int VPP = LT / 4;                             // x voxels staged per pass (4 threads per voxel)
    constexpr int XS = (648 + VPP - 1) / VPP;               // 6 x staging slots per loader thread
    constexpr int DSL = CGW / 4;                            // channel quads per dy voxel
    constexpr int DS = 256 * DSL / LT;                      // 2 (16 output channels) or 4 (32) dy staging slots per loader thread
    constexpr int DROWS = LT / DSL / 16;                    // M-tile rows covered by one dy slot over all loader threads
    static_assert(256 * DSL % LT == 0 && LT / DSL % 16 == 0, "dy slots are whole M-tile rows");
    const int q = lt & 3;
    const int c = chunk * 16 + q * 4;
    const bool cval = c < g.Cin;
    const bool has_norm = a.in_scale != nullptr;
    const bool plain = !has_norm && a.in_slope == 1.f;
    int relx[XS];
    unsigned xo[XS];                                        // LDS element offset | (idd, ih, iw) << 16 (bounds tests per tile)
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      int v = (lt >> 2) + VPP * i; if (v > 647) v = 647;    // (last slot: 8 real voxels; the rest re-read the last one)
      const int iw = v % 18, t2 = v / 18, ih = t2 % 6, idd = t2 / 6;
      relx[i] = ((idd * g.Hi + ih) * g.Wi + iw) * g.x_ldc + c;
      xo[i] = (unsigned)((t2 * XW + iw) * 16 + q * 4) | ((unsigned)idd << 16) | ((unsigned)ih << 19) | ((unsigned)iw << 22);
    }
    // dy slot k of this thread: element e = lt + LT k -> voxel e / DSL, channel quad e % DSL: slot k is slot 0 moved by DROWS whole
    // M-tile rows -- per-thread base + wave-uniform increments
    const int vox0 = lt / DSL, cq0 = lt % DSL, m0 = vox0 >> 4, tw = vox0 & 15;
    const int reld0 = (((m0 >> 2) * g.Ho + (m0 & 3)) * g.Wo + tw) * a.dy_ldc + co0 + cq0 * 4;
    const bool last_x = (lt >> 2) + VPP * (XS - 1) < 648;   // the last slot holds a real voxel for this thread
    const bool co_ok = co0 + cq0 * 4 < g.Cout;
    // TWO register sets of loads in flight (a tile period is shorter than one memory round trip under load); eight loader waves
    // keep a set at XS + DS = 8..10 float4 per thread
    float4 vx[2][XS], vd[2][DS], sc[2], sh[2];
    unsigned okx[2] = {0u, 0u}, okd[2] = {0u, 0u};
    sc[0] = sc[1] = make_float4(1.f, 1.f, 1.f, 1.f); sh[0] = sh[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto issue = [&](int tile, auto S) __attribute__((always_inline)) {
      constexpr int SET = decltype(S)::value;
      const int n = tile / tiles_sp; int rem = tile % tiles_sp;
      const int tile_w = rem % g.tiles_w; rem /= g.tiles_w;
      const int tile_h = rem % g.tiles_h; const int tile_d = rem / g.tiles_h;
      const int od0 = tile_d * 4, oh0 = tile_h * 4, ow0 = tile_w * 16;
      const int id0 = od0 - 1, ih0 = oh0 - 1, iw0 = ow0 - 1;
      const float* xb = ax_ + ((((int64_t)n * g.Di + id0) * g.Hi + ih0) * g.Wi + iw0) * g.x_ldc;
      const float* db = ady_ + ((((int64_t)n * g.Do + od0) * g.Ho + oh0) * g.Wo + ow0) * a.dy_ldc;
      if (has_norm && cval) {
        sc[SET] = *reinterpret_cast<const float4*>(a.in_scale + (int64_t)n * g.Cin + c);
        sh[SET] = *reinterpret_cast<const float4*>(a.in_shift + (int64_t)n * g.Cin + c);
      }
      unsigned mx = 0, md = 0;
#pragma unroll
      for (int i = 0; i < XS; ++i) {
        const int gd = id0 + (int)((xo[i] >> 16) & 7u), gh = ih0 + (int)((xo[i] >> 19) & 7u), gw = iw0 + (int)(xo[i] >> 22);
        const bool ok = cval && (i < XS - 1 || last_x) && (unsigned)gd < (unsigned)g.Di && (unsigned)gh < (unsigned)g.Hi && (unsigned)gw < (unsigned)g.Wi;
        vx[SET][i] = *reinterpret_cast<const float4*>(ok ? xb + relx[i] : ax_);      // unconditional load, masked at conversion
        mx |= ok ? (1u << i) : 0u;
      }
#pragma unroll
      for (int k = 0; k < DS; ++k) {
        const int m = m0 + k * DROWS;
        const bool ok = co_ok && od0 + (m >> 2) < g.Do && oh0 + (m & 3) < g.Ho && ow0 + tw < g.Wo;
        const int rel = reld0 + ((((m >> 2) - (m0 >> 2)) * g.Ho + ((m & 3) - (m0 & 3))) * g.Wo) * a.dy_ldc;
        vd[SET][k] = *reinterpret_cast<const float4*>(ok ? db + rel : ady_);
        md |= ok ? (1u << k) : 0u;
      }
      okx[SET] = mx; okd[SET] = md;
    };
    auto commit = [&](int buf, auto S) __attribute__((always_inline)) {
      constexpr int SET = decltype(S)::value;
      unsigned short* xh = lds + (buf ? BUFB / 2 : 0);
      unsigned short* dh = xh + XIMG;
#pragma unroll
      for (int i = 0; i < XS; ++i) {
        if (i == XS - 1 && !last_x) continue;
        float4 val = vx[SET][i];
        if (!plain) {
          val.x = cwf_act(val.x * sc[SET].x + sh[SET].x, a.in_slope); val.y = cwf_act(val.y * sc[SET].y + sh[SET].y, a.in_slope);
          val.z = cwf_act(val.z * sc[SET].z + sh[SET].z, a.in_slope); val.w = cwf_act(val.w * sc[SET].w + sh[SET].w, a.in_slope);
        }
        uint2 h; h.x = pk_bf16(val.x, val.y); h.y = pk_bf16(val.z, val.w);
        const bool was = (okx[SET] >> i) & 1u;              // zero padding applies after the activation
        h.x = was ? h.x : 0u; h.y = was ? h.y : 0u;
        *reinterpret_cast<uint2*>(xh + (xo[i] & 0xffffu)) = h;
      }
#pragma unroll
      for (int k = 0; k < DS; ++k) {
        const int m = m0 + k * DROWS;
        const float4 val = vd[SET][k];
        uint2 h; h.x = pk_bf16(val.x, val.y); h.y = pk_bf16(val.z, val.w);
        const bool was = (okd[SET] >> k) & 1u;
        h.x = was ? h.x : 0u; h.y = was ? h.y : 0u;
        *reinterpret_cast<uint2*>(dh + (m * DP + tw) * CGW + ((m + 1) >> 1) * DSK + cq0 * 4) = h;
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // tile u lives in register set u & 1 and LDS buffer u & 1.  Iteration it: tile it+1 is converted right after the barrier, then
    // tile it+3 is requested into the set it frees (tile it+2 is already in flight in the other set)
    if (niter > 0) {
      issue(t_begin, S0{});
      commit(0, S0{});
      if (niter > 1) issue(t_begin + 1, S1{});
      if (niter > 2) issue(t_begin + 2, S0{});
    }
    for (int it = 0; it < niter; it += 2) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (it + 1 < niter) {
        commit(1, S1{});
        if (it + 3 < niter) issue(t_begin + it + 3, S1{});
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (it + 2 < niter) {
          commit(0, S0{});
          if (it + 4 < niter) issue(t_begin + it + 4, S0{});
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// wgrad_s1d: wgrad_s1 (32-channel output groups) on bf16 OPERAND IMAGES filled by LDS-DMA -- the wgrad16d recipe for the 32 / 64 / 128-
// channel 3x3x3 stride-1 layers: xa16 [N][D][H][W][Cin] and dy16 [N][D][H][W][Cout] bf16 (written by the InstanceNorm-backward apply
// passes / block tails, or by cwf_to_bf16), eight loader waves issue nothing but global_load_lds_dwordx4 pieces -- 23 for the x image of
// the workgroup's 16-channel chunk (wgrad16d's), 16 for the dy image (one M-tile row of 16 voxels x 32 channels IS one 1-KiB piece;
// the 32-byte skew of odd rows moves the piece's base) -- into a four-buffer ring, three tiles in flight.  MFMA waves, slab layout
// and (chunk, group) blocking: wgrad_s1_kernel<2>.  (wgrad_s1 moves 3.3-3.5x its algorithmic bytes from L2 as fp32; this form
// moves the same halo-amplified voxels at half the bytes, without conversion work and with three tiles of latency cover.)
// ---------------------------------------------------------------------------------------------------
#define WS1D_XP 23
#define WS1D_DP 16
#define WS1D_XIB (WS1D_XP * 1024)
#define WS1D_DIB (16 * 1024 + 8 * 32)                   // 16 rows + the skew of the odd rows
#define WS1D_BUFB (WS1D_XIB + WS1D_DIB)                 // 40,192 B per buffer
#define WS1D_NBUF 4
struct Ws1dArgs {
  const uint4* xa; const uint4* dy; const uint4* zero; float* partial;
  int N, D, H, W, Cin, Cout, tiles_d, tiles_h, tiles_w, total_tiles, tiles_per_split, ngroups;
  int64_t slab_floats;
};

__global__ __launch_bounds__(256 + 64 * WS1_LW) void wgrad_s1d_kernel(const Ws1dArgs a) {
  constexpr int NTW = 2, CG = 2, CGW = 32, XW = 20, DP = 16, DSK = 16;
  extern __shared__ float4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x;
  const int chunk = blockIdx.y / a.ngroups, grp = blockIdx.y % a.ngroups;
  const int tiles_sp = a.tiles_d * a.tiles_h * a.tiles_w;
  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(a.total_tiles, t_begin + a.tiles_per_split);
  const int niter = max(t_end - t_begin, 0);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds4;

  if (wave < 4) {
    // =============================================================== MFMA waves (wgrad_s1_kernel<2>, four buffers)
    const int kq = lane >> 4, bq = (lane & 15) >> 2, bp = lane & 3;
    f32x4 acc[7][NTW];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
    const bool bias_wave = wave == 3;
    const unsigned lane_x = lds_base + (((kq & 1) * XW + (kq >> 1) * 8 + bq) * 16 + bp * 4) * 2;
    unsigned xa0[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int t = wave + 4 * i;
      xa0[i] = lane_x + (t < 27 ? (((t / 9) * 6 + (t / 3) % 3) * XW + t % 3) * 32 : 0);
    }
    const unsigned da0 = lds_base + WS1D_XIB + (((kq & 1) * DP + (kq >> 1) * 8 + bq) * CGW + (kq & 1) * DSK + bp * 4) * 2;
    auto trf = [&](unsigned addr, unsigned second) __attribute__((always_inline)) {
      const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)addr);
      const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(addr + second));
      const s16x8 w = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
      return __builtin_bit_cast(bf16x8, w);
    };
    for (int it = 0; it < niter; ++it) {
      asm volatile("s_barrier" ::: "memory");              // buffer it & 3 has landed
      const unsigned bo = (unsigned)(it & 3) * WS1D_BUFB;
      unsigned xa[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) xa[i] = xa0[i] + bo;
      const unsigned da = da0 + bo;
      constexpr int DEPTH = CWF_WS1_DEPTH;
      bf16x8 ah[DEPTH + 1], bh[2][NTW];
      auto issue = [&](int f) __attribute__((always_inline)) {
        const int ks = f / 7, i = f % 7;
        if (i == 0) {
          const unsigned od_ = (2 * ks * DP * CGW + ks * DSK) * 2;
#pragma unroll
          for (int j = 0; j < NTW; ++j) bh[ks & 1][j] = trf(da + od_ + j * 32, 4 * CGW * 2);
        }
        const unsigned ox = (((ks >> 1) * 6 + ((2 * ks) & 3)) * XW) * 32;
        ah[f % (DEPTH + 1)] = trf(xa[i] + ox, 4 * 32);
      };
#pragma unroll
      for (int f = 0; f < DEPTH; ++f) issue(f);
#pragma unroll
      for (int f = 0; f < 56; ++f) {
        if (f + DEPTH < 56) issue(f + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        const int ks = f / 7, i = f % 7;
        bf16x8 ahf = ah[f % (DEPTH + 1)];
        if (i == 6) ahf = bias_wave ? ones : ahf;
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahf, bh[ks & 1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float4* out = reinterpret_cast<float4*>(a.partial + (int64_t)split * a.slab_floats);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int t = wave + 4 * i;
      if (t > 27) continue;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int64_t blk = (((int64_t)chunk * a.ngroups + grp) * 28 + t) * CG + j;
        out[blk * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
  } else {
    // =============================================================== loader waves: DMA only
    const int lw = wave - 4;                               // pieces lw, lw + 8, ... of the 39
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* glb_vp;
    const int gx = a.Cin >> 3, gd = a.Cout >> 3;           // 16-byte granules per voxel
    auto run = [&](auto NI_) __attribute__((always_inline)) {
      constexpr int NI = decltype(NI_)::value;
      int off[NI]; unsigned crd[NI];
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int s = lw + WS1_LW * k;
        if (s < WS1D_XP) {
          const int gi = 64 * s + lane;
          const int row = gi / 40, gr = gi % 40, w = gr >> 1, half = gr & 1;
          const int c0 = row / 6, c1 = row % 6;
          off[k] = ((c0 * a.H + c1) * a.W + w) * gx + chunk * 2 + half;
          crd[k] = (unsigned)c0 | ((unsigned)c1 << 3) | ((unsigned)w << 6) | ((row < 36 && gr < 36) ? 1u << 11 : 0u);
        } else {
          const int m = s - WS1D_XP, w = lane >> 2, q = lane & 3;
          off[k] = (((m >> 2) * a.H + (m & 3)) * a.W + w) * gd + grp * 4 + q;
          crd[k] = (unsigned)(m >> 2) | ((unsigned)(m & 3) << 3) | ((unsigned)w << 6) | (1u << 11);
        }
      }
      auto issue = [&](int it) __attribute__((always_inline)) {
        const int tile = t_begin + it;
        const int n = tile / tiles_sp; int rem = tile - n * tiles_sp;
        const int tile_w = rem % a.tiles_w; rem /= a.tiles_w;
        const int tile_h = rem % a.tiles_h; const int tile_d = rem / a.tiles_h;
        const int od0 = tile_d * 4, oh0 = tile_h * 4, ow0 = tile_w * 16;
        const int64_t vd = (((int64_t)n * a.D + od0) * a.H + oh0) * a.W + ow0;
        const int64_t vx = vd - ((int64_t)a.H + 1) * a.W - 1;
        const unsigned lbuf = lds_base + (unsigned)(it & 3) * WS1D_BUFB;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
          const int s = lw + WS1_LW * k;                   // (wave-uniform)
          const bool isd = s >= WS1D_XP;
          const int c0 = (int)(crd[k] & 7u), c1 = (int)((crd[k] >> 3) & 7u), w = (int)((crd[k] >> 6) & 31u);
          const int gdd = (isd ? od0 : od0 - 1) + c0, gh = (isd ? oh0 : oh0 - 1) + c1, gw = (isd ? ow0 : ow0 - 1) + w;
          const bool ok = (crd[k] >> 11) != 0u && (unsigned)gdd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
          const uint4* src = (isd ? a.dy + vd * gd : a.xa + vx * gx) + off[k];
          src = ok ? src : a.zero;
          const int m = s - WS1D_XP;
          const unsigned dst = isd ? (unsigned)(WS1D_XIB + m * 1024 + ((m + 1) >> 1) * 32) : (unsigned)s * 1024u;
          __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(uintptr_t)(lbuf + dst), 16, 0, 0);
        }
      };
      if (niter > 0) issue(0);
      if (niter > 1) issue(1);
      if (niter > 2) issue(2);
      for (int it = 0; it < niter; ++it) {
        if (it + 2 < niter) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (it + 3 < niter) issue(it + 3);
      }
    };
    constexpr int NP = WS1D_XP + WS1D_DP;
    if (lw < NP % WS1_LW) run(std::integral_constant<int, NP / WS1_LW + 1>{});
    else run(std::integral_constant<int, NP / WS1_LW>{});
  }
}

extern "C" int64_t cwf_wgrad_slab_floats(int op, int Cin, int Cout);
extern "C" int cwf_wgrad_s1_bf16(const void* xa16, const void* dy16, const void* zero16, float* partial,
                                 int N, int D, int H, int W, int Cin, int Cout, int* nsplit_used, void* stream) {
  if (!xa16 || !dy16 || !zero16 || !partial || N <= 0 || D <= 0 || H <= 0 || W <= 0) return CWF_E_BADARG;
  if (((uintptr_t)xa16 & 15) || ((uintptr_t)dy16 & 15) || ((uintptr_t)zero16 & 15) || ((uintptr_t)partial & 15)) return CWF_E_ALIGN;
  if (Cin < 16 || (Cin & 15) || Cout < 32 || (Cout & 31)) return CWF_E_BADARG;
  if ((int64_t)N * D * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 33)) return CWF_E_TOOLARGE;
  Ws1dArgs a;
  a.xa = (const uint4*)xa16; a.dy = (const uint4*)dy16; a.zero = (const uint4*)zero16; a.partial = partial;
  a.N = N; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.tiles_d = cdiv(D, 4); a.tiles_h = cdiv(H, 4); a.tiles_w = cdiv(W, 16);
  a.total_tiles = N * a.tiles_d * a.tiles_h * a.tiles_w;
  const int nchunks = Cin / 16;
  a.ngroups = Cout / 32;
  const int nblk = nchunks * a.ngroups;
  a.slab_floats = (int64_t)nblk * 28 * 2 * 256;
  if (a.slab_floats != cwf_wgrad_slab_floats(CWF_CONV3_S1, Cin, Cout)) return CWF_E_BADARG;
  int want = side_wgs() / nblk; if (want < 1) want = 1; if (want > a.total_tiles) want = a.total_tiles;
  a.tiles_per_split = cdiv(a.total_tiles, want);
  const int splits = cdiv(a.total_tiles, a.tiles_per_split);
  if (splits > cwf_wgrad_nsplit(CWF_CONV3_S1, N, D, H, W, Cin, Cout)) return CWF_E_BADARG;     // (the caller's slab buffer is sized by it)
  const size_t lds = (size_t)WS1D_NBUF * WS1D_BUFB;
  CWF_MAX_LDS_ONCE((&wgrad_s1d_kernel));
  hipLaunchKernelGGL(wgrad_s1d_kernel, dim3(splits, nblk), dim3(256 + 64 * WS1_LW), lds, cwf_stream(stream), a);
  CWF_LAUNCH_CHECK();
  if (nsplit_used) *nsplit_used = splits;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// pw_wgrad: weight gradients of the pointwise family (1x1x1 convs; ConvTranspose k = 2, s = 2), the counterpart of pw_conv_kernel
// (conv_bf16.hip).  dW[ci][co] = sum_vox xa[vox][ci] * dy[vox][co] is a skinny product over a pure stream -- x and dy are each read
// once, ~0.5 FLOP per byte -- and the generic 1-tap path above (stage a 256-voxel tile in LDS, barrier, transposed reads, barrier;
// one launch block per 16-channel chunk, so dy is re-read per chunk) ran it at 10-28 % of the HBM rate (1x1 32->16 @128^3: 366 us
// for 805 MB with its reduce; tools/layer_table.py).  Here there is no LDS staging and no barrier in the loop:
//   * v_mfma_f32_16x16x4_f32 contracts FOUR voxels per instruction, one fp32 value per lane: lane (i, k) of the A operand is
//     x[voxel k][channel i], of the B operand dy[voxel k][channel j] -- exactly what a dword load of 16 consecutive channels x 4
//     consecutive voxels hands the wave, so fragments come straight from global memory (exact fp32 products: no bf16 rounding);
//   * a wave owns a contiguous run of 4-voxel groups of ONE sample and ALL (chunk, tile) blocks of the layer: x and dy are read
//     exactly once; the loads of G groups are issued together (dword loads: 256 B per instruction, so many must be in flight);
//   * InstanceNorm + activation of x recomputed in registers; bias row = ones operand, as in the tiled kernels;
//   * the 8 or 16 waves of a workgroup sum their accumulators through LDS: one slab per workgroup (256 in all: the slab reduce
//     walks the slabs serially per element, 1024 of them cost it 84 us), in the generic slab layout.
// ---------------------------------------------------------------------------------------------------
struct PwWgWork { int Vin, gps, wps, gpw; };            // input voxels per sample, 4-voxel groups per sample, workgroups per sample, groups per wave

// waves per workgroup (one workgroup per CU): 16 where the accumulators leave room (<= 128 VGPRs), else 8
template <int NCH, int NTL, int NCLS> struct PwgCfg { static constexpr int WAVES = (NCLS * NTL * (NCH + 1) <= 12) ? 16 : 8; };
template <int NCH, int NTL, int NCLS>
__global__ __launch_bounds__((64 * PwgCfg<NCH, NTL, NCLS>::WAVES)) void pw_wgrad_kernel(const WgArgsB a, const PwWgWork wk, int CG) {
  constexpr int PWG_WAVES = PwgCfg<NCH, NTL, NCLS>::WAVES, RB = 32 / PWG_WAVES;      // RB: blocks per reduction round
  constexpr int G = (NCLS > 1) ? (NCH * NTL > 1 ? 1 : 2) : (NCH + NTL <= 2 ? 16 : (NCH + NTL <= 3 ? 12 : (NCH + NTL <= 6 ? 4 : 2)));   // groups per iteration
  __shared__ float4 red[PWG_WAVES][RB][64];
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = (int)blockIdx.x / wk.wps, wg = (int)blockIdx.x % wk.wps;
  const int gA = (wg * PWG_WAVES + wave) * wk.gpw;
  const int gB = min(gA + wk.gpw, wk.gps);
  const float* xs = a.x + (int64_t)n * wk.Vin * g.x_ldc;
  const float* ds = a.dy + (int64_t)n * g.Do * g.Ho * g.Wo * a.dy_ldc;
  const bool has_norm = a.in_scale != nullptr;
  const bool plain = !has_norm && a.in_slope == 1.f;
  float sc[NCH], sh[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ci = c * 16 + li;
    sc[c] = (has_norm && ci < g.Cin) ? a.in_scale[(int64_t)n * g.Cin + ci] : 1.f;
    sh[c] = (has_norm && ci < g.Cin) ? a.in_shift[(int64_t)n * g.Cin + ci] : 0.f;
  }
  f32x4 acc[NCLS][NCH][NTL], accb[NCLS][NTL];
#pragma unroll
  for (int q = 0; q < NCLS; ++q)
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      accb[q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[q][c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  for (int gi = gA; gi < gB; gi += G) {
    float xv[G][NCH], dv[G][NCLS][NTL];
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const bool ok = gi + k < gB;                          // wave-uniform
      const int vox = (gi + k) * 4 + lk;                    // this lane's voxel of the group
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ci = c * 16 + li;
        xv[k][c] = (ok && ci < g.Cin) ? xs[(int64_t)vox * g.x_ldc + ci] : 0.f;
      }
      int64_t ov[NCLS];
      if (NCLS == 1) ov[0] = vox;
      else {
        const int w_ = vox % g.Wi, t2 = vox / g.Wi, h_ = t2 % g.Hi, d_ = t2 / g.Hi;
#pragma unroll
        for (int q = 0; q < NCLS; ++q) ov[q < NCLS ? q : 0] = ((int64_t)(2 * d_ + (q >> 2)) * g.Ho + 2 * h_ + ((q >> 1) & 1)) * g.Wo + 2 * w_ + (q & 1);
      }
#pragma unroll
      for (int q = 0; q < NCLS; ++q)
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          const int co = t * 16 + li;
          dv[k][q][t] = (ok && co < g.Cout) ? ds[ov[q] * a.dy_ldc + co] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const bool ok = gi + k < gB;
      float xa[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float v = xv[k][c];
        if (!plain) v = cwf_act(v * sc[c] + sh[c], a.in_slope);
        xa[c] = (ok && c * 16 + li < g.Cin) ? v : 0.f;      // (a masked lane would otherwise carry act(shift))
      }
#pragma unroll
      for (int q = 0; q < NCLS; ++q)
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          accb[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, dv[k][q][t], accb[q][t], 0, 0, 0);
#pragma unroll
          for (int c = 0; c < NCH; ++c) acc[q][c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[c], dv[k][q][t], acc[q][c][t], 0, 0, 0);
        }
    }
  }

  // ---- one slab per workgroup: block (class, chunk, group, tap 0 = weights / 1 = bias row, tile) = 64 lanes x float4.
  // The waves' accumulators meet in LDS RB blocks at a time; waves 0..RB-1 each sum one block over all waves and store it.
  float4* out = reinterpret_cast<float4*>(a.partial + (int64_t)blockIdx.x * a.slab_floats);
  constexpr int NB = NCLS * NTL * (NCH + 1);
  f32x4 flat[NB]; int bid[NB];
  {
    int u = 0;
#pragma unroll
    for (int q = 0; q < NCLS; ++q)
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const int grp = t / CG, j = t % CG;
#pragma unroll
        for (int c = 0; c < NCH; ++c) { flat[u] = acc[q][c][t]; bid[u] = a.cls_slab_base[q] + ((c * a.ngroups + grp) * 2) * CG + j; ++u; }
        // tap slot 1: the all-ones row.  Every chunk's slot is written (the reduce map reads chunk 0's), as in the tiled kernel
        flat[u] = accb[q][t]; bid[u] = a.cls_slab_base[q] + ((0 * a.ngroups + grp) * 2 + 1) * CG + j; ++u;
      }
  }
#pragma unroll
  for (int b0 = 0; b0 < NB; b0 += RB) {
    __syncthreads();
#pragma unroll
    for (int r_ = 0; r_ < RB; ++r_)
      if (b0 + r_ < NB) red[wave][r_][lane] = make_float4(flat[b0 + r_][0], flat[b0 + r_][1], flat[b0 + r_][2], flat[b0 + r_][3]);
    __syncthreads();
    if (wave < RB && b0 + wave < NB) {
      float4 sum = red[0][wave][lane];
#pragma unroll
      for (int w = 1; w < PWG_WAVES; ++w) { const float4 p = red[w][wave][lane]; sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w; }
      int id = bid[b0];
#pragma unroll
      for (int r_ = 1; r_ < RB; ++r_) if (b0 + r_ < NB && wave == r_) id = bid[b0 + r_];
      out[(int64_t)id * 64 + lane] = sum;
      if (NCH > 1) {                                        // the bias rows of chunks 1.. (same values)
#pragma unroll
        for (int r_ = 0; r_ < RB; ++r_) {
          if (b0 + r_ < NB && wave == r_ && (b0 + r_) % (NCH + 1) == NCH) {
            for (int c = 1; c < NCH; ++c) out[((int64_t)id + (int64_t)c * a.ngroups * 2 * CG) * 64 + lane] = sum;
          }
        }
      }
    }
  }
}

template <int NCH, int NTL, int NCLS>
static int launch_pw_wgrad(const WgArgsB& a, int CG, int max_slabs, int* nsplit_used, hipStream_t st) {
  const ConvGeom& g = a.g;
  PwWgWork wk;
  wk.Vin = g.Di * g.Hi * g.Wi;
  wk.gps = wk.Vin >> 2;
  int wps = side_wgs() / g.N; if (wps < 1) wps = 1;        // ~one 8/16-wave workgroup per CU in all, each inside one sample
  if (wps * g.N > max_slabs) wps = max_slabs / g.N;
  if (wps < 1) return -1;
  constexpr int PWG_WAVES = PwgCfg<NCH, NTL, NCLS>::WAVES;
  const int maxw = cdiv(wk.gps, PWG_WAVES * 8);
  if (wps > maxw) wps = maxw;
  wk.gpw = cdiv(wk.gps, wps * PWG_WAVES);
  wk.wps = cdiv(wk.gps, wk.gpw * PWG_WAVES);
  hipLaunchKernelGGL((pw_wgrad_kernel<NCH, NTL, NCLS>), dim3((unsigned)(wk.wps * g.N)), dim3(64 * PWG_WAVES), 0, st, a, wk, CG);
  CWF_LAUNCH_CHECK();
  if (nsplit_used) *nsplit_used = wk.wps * g.N;
  return 0;
}

// plan: identical decisions to wgrad_mfma.hip (the Python side sizes the workspace through cwf_wgrad_nsplit / _slab_floats)
extern "C" int cwf_wgrad_nsplit(int op, int N, int Do, int Ho, int Wo, int Cin, int Cout);
extern "C" int64_t cwf_wgrad_slab_floats(int op, int Cin, int Cout);

static int wgrad_bf16_impl(int op, int x3, const float* x, int x_ldc, const float* in_scale, const float* in_shift, float in_slope,
                           const float* dy, int dy_ldc, float* partial,
                           int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int* nsplit_used, void* stream,
                           int groups, const float* const* xg, const float* const* dyg, float* const* pg, const float* dy_scale = nullptr);

// cwf_wgrad_mfma_bf16 with dy taken as dy * dy_scale[n][co] (full-resolution 16-output-channel 3x3x3 stride-1 layers only: the stem)
extern "C" int cwf_wgrad_mfma_bf16_dys(int op, int x3, const float* x, int x_ldc, const float* in_scale, const float* in_shift, float in_slope,
                                       const float* dy, int dy_ldc, const float* dy_scale, float* partial,
                                       int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int* nsplit_used, void* stream) {
  if (!dy_scale) return CWF_E_BADARG;
  return wgrad_bf16_impl(op, x3, x, x_ldc, in_scale, in_shift, in_slope, dy, dy_ldc, partial, N, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, nsplit_used, stream,
                         0, nullptr, nullptr, nullptr, dy_scale);
}

extern "C" int cwf_wgrad_mfma_bf16(int op, int x3, const float* x, int x_ldc, const float* in_scale, const float* in_shift, float in_slope,
                                   const float* dy, int dy_ldc, float* partial,
                                   int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int* nsplit_used, void* stream) {
  return wgrad_bf16_impl(op, x3, x, x_ldc, in_scale, in_shift, in_slope, dy, dy_ldc, partial, N, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, nsplit_used, stream,
                         0, nullptr, nullptr, nullptr);
}

// `groups` (2 or 3) same-shape 3x3x3 stride-1 layers in ONE launch (blockIdx.z = group): group q has its own activation view x[q]
// (row pitch x_ldc), gradient view dy[q] (row pitch dy_ldc) and slab buffer partial[q] (each sized like a single layer's); no
// normalising prologue.  h_x / h_dy / h_partial: HOST arrays of device pointers.  The three sub-regions' supervision-head layers.
extern "C" int cwf_wgrad_mfma_bf16_grouped(int op, int x3, const float* const* h_x, int x_ldc, const float* const* h_dy, int dy_ldc,
                                           float* const* h_partial, int groups,
                                           int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int* nsplit_used, void* stream) {
  if (!h_x || !h_dy || !h_partial || groups < 2 || groups > 3 || op != CWF_CONV3_S1) return CWF_E_BADARG;
  for (int q = 0; q < groups; ++q)
    if (!h_x[q] || !h_dy[q] || !h_partial[q] || ((uintptr_t)h_x[q] & 15) || ((uintptr_t)h_partial[q] & 15)) return CWF_E_ALIGN;
  return wgrad_bf16_impl(op, x3, h_x[0], x_ldc, nullptr, nullptr, 1.f, h_dy[0], dy_ldc, h_partial[0], N, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, nsplit_used,
                         stream, groups, h_x, h_dy, h_partial);
}

static int wgrad_bf16_impl(int op, int x3, const float* x, int x_ldc, const float* in_scale, const float* in_shift, float in_slope,
                           const float* dy, int dy_ldc, float* partial,
                           int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int* nsplit_used, void* stream,
                           int groups, const float* const* xg, const float* const* dyg, float* const* pg, const float* dy_scale) {
  if (!x || !dy || !partial || N <= 0) return CWF_E_BADARG;
  // dy_scale is implemented by wgrad16_kernel alone
  if (dy_scale && !(!groups && op == CWF_CONV3_S1 && Cin <= 16 && Cout == 16 && (dy_ldc & 3) == 0 && (((uintptr_t)dy) & 15) == 0 && (int64_t)Do * Ho * Wo >= 32768))
    return CWF_E_BADARG;
  if ((Cin & 3) || (x_ldc & 3) || ((uintptr_t)x & 15) || ((uintptr_t)partial & 15)) return CWF_E_ALIGN;
  if (in_scale && !in_shift) return CWF_E_BADARG;
  if (!(op == CWF_CONV3_S1 || op == CWF_CONV3_S2 || op == CWF_CONV1 || op == CWF_CONVT2)) return CWF_E_BADARG;
  const bool tapsplit = (op == CWF_CONV3_S1 || op == CWF_CONV3_S2);
  const int MTOT = (op == CWF_CONV3_S2) ? 4 : 16;
  const int nt_all = cdiv(Cout, 16);
  const int CG = tapsplit ? (nt_all == 1 ? 1 : 2) : (nt_all == 1 ? 1 : (nt_all == 2 ? 2 : 4));
  WgArgsB a;
  int rc = cwf_build_geom(a.g, op, N, Di, Hi, Wi, Cin, x_ldc, Do, Ho, Wo, Cout, dy_ldc, MTOT);
  if (rc) return rc;
  const int nchunks = a.g.nchunks, ncls = a.g.ncls, ngroups = cdiv(a.g.ntiles, CG);
  int64_t blocks = 0;
  for (int c = 0; c < 8; ++c) { a.cls_slab_base[c] = (int)blocks; if (c < ncls) blocks += (int64_t)nchunks * ngroups * (a.g.cls_ntaps[c] + 1) * CG; }
  const int total = N * a.g.tiles_d * a.g.tiles_h * a.g.tiles_w;
  const int nblk = nchunks * ngroups * ncls;
  int want = 512 / nblk; if (want < 1) want = 1; if (want > total) want = total;
  const int tps = cdiv(total, want);
  const int wg_splits = cdiv(total, tps);
  // must agree with wgrad_mfma.hip's plan (the workspace was sized from it)
  if ((tapsplit ? wg_splits : wg_splits * 4) != cwf_wgrad_nsplit(op, N, Do, Ho, Wo, Cin, Cout) || blocks * 256 != cwf_wgrad_slab_floats(op, Cin, Cout))
    return CWF_E_BADARG;
  a.x = x; a.in_scale = in_scale; a.in_shift = in_shift; a.in_slope = in_slope; a.dy = dy; a.dy_ldc = dy_ldc; a.partial = partial;
  a.ngroups = ngroups; a.tiles_per_split = tps; a.total_tiles = total; a.slab_floats = blocks * 256;
  a.groups = groups; a.dy_scale = dy_scale;
  bool dy_al = (((uintptr_t)dy) & 15) == 0;
  for (int q = 0; q < 3; ++q) {
    a.x_g[q] = (groups && q < groups) ? xg[q] : nullptr; a.dy_g[q] = (groups && q < groups) ? dyg[q] : nullptr; a.partial_g[q] = (groups && q < groups) ? pg[q] : nullptr;
    if (groups && q < groups && (((uintptr_t)dyg[q]) & 15)) dy_al = false;
  }
  const int gz = groups ? groups : ncls;                 // grid z: group (grouped launches are single-class) or parity class
  if (nsplit_used) *nsplit_used = tapsplit ? wg_splits : wg_splits * 4;
  if (!groups && op == CWF_CONV3_S1 && Cin <= 16 && Cout == 16 && (dy_ldc & 3) == 0 && (((uintptr_t)dy) & 15) == 0 && (int64_t)Do * Ho * Wo >= 32768) {
    // full-resolution 16-channel layers: persistent producer/consumer kernel, one slab per workgroup (<= 256 <= generic nsplit)
    // (the stem layer -- 4 input channels -- is the LAST kernel of backward: nothing runs beside it, it takes every CU)
    int grid = Cin <= 4 ? 256 : side_wgs(); while (grid > 8 && grid > total) grid -= 8;      // multiple of 8 (XCD-aware tile map)
    const size_t lds16 = (size_t)2 * (36 * W16_XW * 16 + 16 * W16_DW * 16) * sizeof(unsigned short) * (x3 ? 2 : 1);
    hipStream_t st16 = cwf_stream(stream);
    if (x3) {
      CWF_MAX_LDS_ONCE((&wgrad16_kernel<true>));
      hipLaunchKernelGGL((wgrad16_kernel<true>), dim3(grid), dim3(256 + 64 * W16_LW), lds16, st16, a, total);
    } else {
      CWF_MAX_LDS_ONCE((&wgrad16_kernel<false>));
      hipLaunchKernelGGL((wgrad16_kernel<false>), dim3(grid), dim3(256 + 64 * W16_LW), lds16, st16, a, total);
    }
    CWF_LAUNCH_CHECK();
    if (nsplit_used) *nsplit_used = grid;
    return 0;
  }
  if (!groups && (op == CWF_CONV1 || op == CWF_CONVT2) && ((int64_t)Di * Hi * Wi & 3) == 0 && (op == CWF_CONV1 || (Wi & 3) == 0) &&
      (int64_t)Di * Hi * Wi * (x_ldc > 8 * dy_ldc ? x_ldc : 8 * dy_ldc) < (1ll << 31)) {
    // pointwise layers: stream kernel (fp32 MFMA straight from global memory, every operand read once), both precision modes
    static const bool off = getenv("CWF_NO_PW_WGRAD") != nullptr;        // A/B switch (diagnostics)
    const int nch = nchunks, ntl = a.g.ntiles;
    const int max_slabs = wg_splits * 4;                  // what the workspace was sized for
    hipStream_t stp = cwf_stream(stream);
    int r = -2;
    if (!off) {
      if (op == CWF_CONV1) {
        if (nch == 1 && ntl == 1) r = launch_pw_wgrad<1, 1, 1>(a, CG, max_slabs, nsplit_used, stp);
        else if (nch == 2 && ntl == 1) r = launch_pw_wgrad<2, 1, 1>(a, CG, max_slabs, nsplit_used, stp);
        else if (nch == 4 && ntl == 2) r = launch_pw_wgrad<4, 2, 1>(a, CG, max_slabs, nsplit_used, stp);
        else if (nch == 8 && ntl == 4) r = launch_pw_wgrad<8, 4, 1>(a, CG, max_slabs, nsplit_used, stp);
      } else {
        if (nch == 1 && ntl == 1) r = launch_pw_wgrad<1, 1, 8>(a, CG, max_slabs, nsplit_used, stp);
        else if (nch == 2 && ntl == 2) r = launch_pw_wgrad<2, 2, 8>(a, CG, max_slabs, nsplit_used, stp);
      }
    }
    if (r >= 0) return r;
  }
  if (!x3 && op == CWF_CONV3_S1 && CG <= 2 && a.g.TD == 4 && a.g.TH == 4 && a.g.ID == 6 && a.g.IH == 6 && a.g.IW == 18 && (Cout & 3) == 0 &&
      (dy_ldc & 3) == 0 && dy_al) {
    static const bool off = getenv("CWF_NO_WGRAD_S1") != nullptr;       // A/B switch (diagnostics)
    if (!off) {
      // producer / consumer kernel: ~256 eight-wave workgroups in all (one per CU), each a contiguous tile range of one (chunk, group)
      int want1 = side_wgs() / nblk; if (want1 < 1) want1 = 1; if (want1 > total) want1 = total;
      const int tps1 = cdiv(total, want1), splits1 = cdiv(total, tps1);
      if (splits1 <= wg_splits) {                          // (the workspace was sized for wg_splits slabs)
        a.tiles_per_split = tps1;
        if (nsplit_used) *nsplit_used = splits1;
        const size_t lds1 = (size_t)2 * (36 * 20 * 16 + 16 * (CG == 1 ? 20 : 16) * CG * 16 + (16 / 2 + 1) * (CG == 2 ? 16 : 0)) * sizeof(unsigned short);
        dim3 grid1(splits1, nchunks * ngroups, groups ? groups : 1);
        hipStream_t st1 = cwf_stream(stream);
        CWF_MAX_LDS_ONCE((&wgrad_s1_kernel<1>));
        CWF_MAX_LDS_ONCE((&wgrad_s1_kernel<2>));
        if (CG == 1) hipLaunchKernelGGL((wgrad_s1_kernel<1>), grid1, dim3(256 + 64 * WS1_LW), lds1, st1, a);
        else hipLaunchKernelGGL((wgrad_s1_kernel<2>), grid1, dim3(256 + 64 * WS1_LW), lds1, st1, a);
        CWF_LAUNCH_CHECK();
        return 0;
      }
    }
  }
  const size_t ximg = (size_t)a.g.ID * a.g.IH * wg_x_pitch(a.g.IW, a.g.is) * 16;
  const size_t dimg = (size_t)a.g.TD * a.g.TH * wg_dy_pitch(CG * 16) * CG * 16 + (size_t)(a.g.TD * a.g.TH / 2 + 1) * wg_dy_skew(CG * 16);
  const size_t lds = (ximg + dimg) * sizeof(unsigned short) * (x3 ? 2 : 1);
  if (lds > 160 * 1024) return CWF_E_TOOLARGE;
  dim3 grid(wg_splits, nchunks * ngroups, gz);
  hipStream_t st = cwf_stream(stream);
#define CWF_WG(tpw, ntw, ts, xx) do { CWF_MAX_LDS_ONCE((&wgrad_bf16_kernel<tpw, ntw, ts, xx>)); \
    hipLaunchKernelGGL((wgrad_bf16_kernel<tpw, ntw, ts, xx>), grid, dim3(256), lds, st, a); } while (0)
#define CWF_WGX(tpw, ntw, ts) do { if (x3) CWF_WG(tpw, ntw, ts, true); else CWF_WG(tpw, ntw, ts, false); } while (0)
  if (tapsplit) { if (CG == 1) CWF_WGX(7, 1, true); else CWF_WGX(7, 2, true); }
  else { if (CG == 1) CWF_WGX(2, 1, false); else if (CG == 2) CWF_WGX(2, 2, false); else CWF_WGX(2, 4, false); }
#undef CWF_WGX
#undef CWF_WG
  CWF_LAUNCH_CHECK();
  return 0;
}
