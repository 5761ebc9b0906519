// K1w -- weight (+bias) gradient of the conv family on v_mfma_f32_16x16x4_f32, gfx950.
//
//   dW[tap][ci][co] = sum_{n,vox} xa[n, vox*is + tap, ci] * dy[n, vox*os + ooff, co],  xa = act(IN(x)) recomputed
//   in the staging prologue exactly as the forward did (nothing but x and the IN scale/shift is saved).
//
// GEMM view: M = 16 input channels of one chunk, N = 16 output channels, K = voxels (4 per MFMA).
// A workgroup owns one (ci-chunk, co-group) output block for ALL taps and walks a contiguous range of spatial
// tiles ("split"), keeping the 16x16 accumulators in registers; it then writes one partial slab in raw MFMA
// accumulator layout.  cwf_wgrad_reduce sums the slabs and scatters into nn.Conv3d.weight layout through a
// host-built index map (no atomics: results are bitwise reproducible).
//   3x3x3 ops : waves split the 27 taps (t = wave + 4 i); tap slot 27 is a virtual all-ones tap whose
//               accumulator is the bias gradient sum_vox dy[co].
//   1-tap ops : waves split the voxels of each tile (M-tile mt -> wave mt & 3) and each wave writes its own
//               slab (slab index = split*4 + wave); slot 1 is the ones tap.
//
// Replaces the weight/bias halves of aten::convolution_backward for every conv of the model
// (38 % of the reference's CPU step).
#include "common.h"

struct WgArgs {
  ConvGeom g;
  const float* x; const float* in_scale; const float* in_shift; float in_slope;
  const float* dy; int dy_ldc; float* partial;
  int ngroups;          // co groups per chunk
  int tiles_per_split;  // spatial tiles (over n, d, h, w) per workgroup
  int total_tiles;
  int64_t slab_floats;
  int cls_slab_base[8]; // offset of a class inside a slab, in 256-float blocks
};

template <int TPW, int NTW, bool TAPSPLIT>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const WgArgs a) {
  constexpr int CG = NTW;                          // N-tiles per workgroup
  constexpr int CGW = CG * 16;
  extern __shared__ float4 lds4[];
  const ConvGeom& g = a.g;
  float* xt = reinterpret_cast<float*>(lds4);
  float* dyt = xt + g.ID * g.IH * g.IW * 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int split = blockIdx.x;
  const int chunk = blockIdx.y / a.ngroups, grp = blockIdx.y % a.ngroups;
  const int cls = blockIdx.z;
  const int Dc = g.cls_dims[cls][0], Hc = g.cls_dims[cls][1], Wc = g.cls_dims[cls][2];
  const int ntaps = g.cls_ntaps[cls];
  const int* tapofs = g.tapofs + (g.ncls > 1 ? cls * 8 : 0);
  const int co0 = grp * CGW;
  const int MV = g.TD * g.TH * 16;

  f32x4 acc[TPW][NTW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int tiles_sp = g.tiles_d * g.tiles_h * g.tiles_w;
  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(a.total_tiles, t_begin + a.tiles_per_split);
  const bool vec_dy = (a.dy_ldc & 3) == 0 && (((uintptr_t)a.dy) & 15) == 0;
  const int of0 = g.cls_ooff[cls][0], of1 = g.cls_ooff[cls][1], of2 = g.cls_ooff[cls][2];

  for (int tile = t_begin; tile < t_end; ++tile) {
    const int n = tile / tiles_sp; int rem = tile % tiles_sp;
    const int tile_w = rem % g.tiles_w; rem /= g.tiles_w;
    const int tile_h = rem % g.tiles_h; const int tile_d = rem / g.tiles_h;
    const int od0 = tile_d * g.TD, oh0 = tile_h * g.TH, ow0 = tile_w * 16;
    if (od0 >= Dc || oh0 >= Hc || ow0 >= Wc) continue;     // uniform
    __syncthreads();
    cwf_stage_input_tile(xt, g, a.x, a.in_scale, a.in_shift, a.in_slope, n, chunk,
                         od0 * g.is + g.lo[0], oh0 * g.is + g.lo[1], ow0 * g.is + g.lo[2], tid);
    // dy tile [MV][CGW]
    for (int e = tid; e < MV * (CGW / 4); e += 256) {
      const int vox = e / (CGW / 4), cq = e % (CGW / 4);
      const int tw = vox & 15, mt = vox >> 4;
      const int od = od0 + mt / g.TH, oh = oh0 + mt % g.TH, ow = ow0 + tw;
      const int co = co0 + cq * 4;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (od < Dc && oh < Hc && ow < Wc && co < g.Cout) {
        const int64_t gv = (((int64_t)n * g.Do + (od * g.os + of0)) * g.Ho + (oh * g.os + of1)) * g.Wo + (ow * g.os + of2);
        const float* p = a.dy + gv * a.dy_ldc + co;
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
      *reinterpret_cast<float4*>(dyt + vox * CGW + cq * 4) = val;
    }
    __syncthreads();

    const int nks = g.TD * g.TH * 4;
#pragma unroll 2
    for (int ks = 0; ks < nks; ++ks) {
      const int mt = ks >> 2;
      if (!TAPSPLIT && (mt & 3) != wave) continue;     // 1-tap ops: waves split the voxels (wave-uniform)
      const int tw = (ks & 3) * 4 + kq;
      const int vin = (((mt / g.TH) * g.is) * g.IH + (mt % g.TH) * g.is) * g.IW + tw * g.is;
      const int vout = mt * 16 + tw;
      float b[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) b[j] = dyt[vout * CGW + j * 16 + r];
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int t = TAPSPLIT ? wave + 4 * i : i;
        if (t > ntaps) continue;                       // wave-uniform
        const float av = (t == ntaps) ? 1.0f : xt[(vin + tapofs[t < ntaps ? t : 0]) * 16 + r];
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- partial slab: [cls][chunk][grp][tap slot 0..ntaps][nt in group][lane][4]
  float4* out = reinterpret_cast<float4*>(a.partial + (int64_t)(TAPSPLIT ? split : split * 4 + wave) * a.slab_floats);
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

// Sum the slabs in SLAB order (every load is a coalesced 16-B read; the nsplit loads of a thread are independent and
// pipeline) and scatter the 4 sums through the inverse map: inv >= 0 -> dW index, inv <= -2 -> bias index (-2 - inv),
// -1 -> padding.  Fixed summation order: bitwise reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, int64_t slab,
                                                          const int32_t* __restrict__ inv, float* __restrict__ dW, float* __restrict__ db) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index inside the slab
  if (q * 4 >= slab) return;
  const float4* p = reinterpret_cast<const float4*>(partial) + q;
  const int64_t stride4 = slab >> 2;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (; k + 4 <= nsplit; k += 4) {
    const float4 a0 = p[(int64_t)k * stride4], a1 = p[(int64_t)(k + 1) * stride4], a2 = p[(int64_t)(k + 2) * stride4], a3 = p[(int64_t)(k + 3) * stride4];
    s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w;
    s.x += a1.x; s.y += a1.y; s.z += a1.z; s.w += a1.w;
    s.x += a2.x; s.y += a2.y; s.z += a2.z; s.w += a2.w;
    s.x += a3.x; s.y += a3.y; s.z += a3.z; s.w += a3.w;
  }
  for (; k < nsplit; ++k) { const float4 a0 = p[(int64_t)k * stride4]; s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w; }
  const int4 m = reinterpret_cast<const int4*>(inv)[q];
  const int mm[4] = {m.x, m.y, m.z, m.w};
  const float ss[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (mm[i] >= 0) dW[mm[i]] = ss[i];
    else if (mm[i] <= -2 && db) db[-2 - mm[i]] = ss[i];
  }
}

// All layers of a backward phase in ONE launch (descriptor table; grid (32, nlayers), grid-stride over the slab): the per-layer
// reduce above launches slab/1024 workgroups -- 7..28 for most layers of this model -- and was latency-bound at ~36 us each,
// 80 launches and 2.9 ms per step.  dW / db point straight into the flat gradient buffer the optimizer and the all-reduce use.
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const cwf_wgrad_reduce_desc* __restrict__ table) {
  // 64 float4 columns x 4 split lanes per workgroup: a column's nsplit slabs are summed by four threads (k = lane, lane+4, ...,
  // four loads in flight each) and combined through LDS in a fixed order -- the chain of dependent loads per column is
  // nsplit/16 long instead of nsplit/4, and small slabs still fill a workgroup.
  __shared__ float4 red[4][64];
  const cwf_wgrad_reduce_desc d = table[blockIdx.y];
  const int64_t stride4 = d.slab >> 2;
  const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
  for (int64_t q0 = (int64_t)blockIdx.x * 64; q0 < stride4; q0 += (int64_t)gridDim.x * 64) {
    const int64_t q = q0 + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < stride4) {
      const float4* p = reinterpret_cast<const float4*>(d.partial) + q;
      int k = sl;
      for (; k + 12 < d.nsplit; k += 16) {
        const float4 a0 = p[(int64_t)k * stride4], a1 = p[(int64_t)(k + 4) * stride4], a2 = p[(int64_t)(k + 8) * stride4], a3 = p[(int64_t)(k + 12) * stride4];
        s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w;
        s.x += a1.x; s.y += a1.y; s.z += a1.z; s.w += a1.w;
        s.x += a2.x; s.y += a2.y; s.z += a2.z; s.w += a2.w;
        s.x += a3.x; s.y += a3.y; s.z += a3.z; s.w += a3.w;
      }
      for (; k < d.nsplit; k += 4) { const float4 a0 = p[(int64_t)k * stride4]; s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w; }
    }
    __syncthreads();
    red[sl][c] = s;
    __syncthreads();
    if (sl == 0 && q < stride4) {
      const float4 r0 = red[0][c], r1 = red[1][c], r2 = red[2][c], r3 = red[3][c];
      const float ss[4] = {(r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y), (r0.z + r1.z) + (r2.z + r3.z), (r0.w + r1.w) + (r2.w + r3.w)};
      const int4 m = reinterpret_cast<const int4*>(d.inv)[q];
      const int mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (mm[i] >= 0) d.dW[mm[i]] = ss[i];
        else if (mm[i] <= -2 && d.db) d.db[-2 - mm[i]] = ss[i];
      }
    }
  }
}

extern "C" int cwf_wgrad_reduce_batched(const struct cwf_wgrad_reduce_desc* table, int nlayers, void* stream) {
  if (!table || nlayers <= 0 || nlayers > 65535) return CWF_E_BADARG;
  hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(128, nlayers), dim3(256), 0, cwf_stream(stream), table);
  CWF_LAUNCH_CHECK();
  return 0;
}

namespace {
struct WgPlan { bool tapsplit; int MTOT; int CG; int ngroups; int nchunks; int ncls; int ntaps_sum; int64_t slab; int nsplit; int wg_splits; int tps; int total; ConvGeom g; };

int make_plan(WgPlan& p, int op, int N, int Di, int Hi, int Wi, int Cin, int x_ldc, int Do, int Ho, int Wo, int Cout, int dy_ldc) {
  if (!(op == CWF_CONV3_S1 || op == CWF_CONV3_S2 || op == CWF_CONV1 || op == CWF_CONVT2)) return CWF_E_BADARG;
  p.tapsplit = (op == CWF_CONV3_S1 || op == CWF_CONV3_S2);
  p.MTOT = (op == CWF_CONV3_S2) ? 4 : 16;
  const int nt_all = cdiv(Cout, 16);
  p.CG = p.tapsplit ? (nt_all == 1 ? 1 : 2) : (nt_all == 1 ? 1 : (nt_all == 2 ? 2 : 4));
  int rc = cwf_build_geom(p.g, op, N, Di, Hi, Wi, Cin, x_ldc, Do, Ho, Wo, Cout, dy_ldc, p.MTOT);
  if (rc) return rc;
  p.nchunks = p.g.nchunks; p.ncls = p.g.ncls;
  p.ngroups = cdiv(p.g.ntiles, p.CG);
  int64_t blocks = 0;
  for (int c = 0; c < p.ncls; ++c) blocks += (int64_t)p.nchunks * p.ngroups * (p.g.cls_ntaps[c] + 1) * p.CG;
  p.slab = blocks * 256;
  p.total = N * p.g.tiles_d * p.g.tiles_h * p.g.tiles_w;
  const int nblk = p.nchunks * p.ngroups * p.ncls;
  int want = 512 / nblk; if (want < 1) want = 1; if (want > p.total) want = p.total;
  p.tps = cdiv(p.total, want);
  p.nsplit = cdiv(p.total, p.tps);
  p.wg_splits = p.nsplit;
  if (!p.tapsplit) p.nsplit *= 4;                 // every wave writes its own slab
  return 0;
}
// shape-only plan (strides irrelevant)
int make_plan_shape(WgPlan& p, int op, int N, int Do, int Ho, int Wo, int Cin, int Cout) {
  int Di = Do, Hi = Ho, Wi = Wo;
  if (op == CWF_CONV3_S2) { Di = 2 * Do; Hi = 2 * Ho; Wi = 2 * Wo; }   // any size with (Di-1)/2+1 == Do gives the same plan
  if (op == CWF_CONVT2) { Di = Do / 2; Hi = Ho / 2; Wi = Wo / 2; }
  return make_plan(p, op, N, Di, Hi, Wi, Cin, (Cin + 3) & ~3, Do, Ho, Wo, Cout, (Cout + 3) & ~3);
}
}  // namespace

extern "C" int cwf_wgrad_nsplit(int op, int N, int Do, int Ho, int Wo, int Cin, int Cout) {
  WgPlan p; int rc = make_plan_shape(p, op, N, Do, Ho, Wo, Cin, Cout); return rc ? rc : p.nsplit;
}
extern "C" int64_t cwf_wgrad_slab_floats(int op, int Cin, int Cout) {
  WgPlan p; int rc = make_plan_shape(p, op, 1, 16, 16, 16, Cin, Cout); return rc ? rc : p.slab;
}
extern "C" int64_t cwf_wgrad_partial_floats(int op, int N, int Do, int Ho, int Wo, int Cin, int Cout) {
  WgPlan p; int rc = make_plan_shape(p, op, N, Do, Ho, Wo, Cin, Cout); return rc ? rc : p.slab * p.nsplit;
}

extern "C" int cwf_wgrad_mfma(int op, const float* x, int x_ldc, const float* in_scale, const float* in_shift, float in_slope,
                              const float* dy, int dy_ldc, float* partial,
                              int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, void* stream) {
  if (!x || !dy || !partial || N <= 0) return CWF_E_BADARG;
  if ((Cin & 3) || (x_ldc & 3) || ((uintptr_t)x & 15) || ((uintptr_t)partial & 15)) return CWF_E_ALIGN;
  if (in_scale && !in_shift) return CWF_E_BADARG;
  WgPlan p;
  int rc = make_plan(p, op, N, Di, Hi, Wi, Cin, x_ldc, Do, Ho, Wo, Cout, dy_ldc);
  if (rc) return rc;
  WgArgs a;
  a.g = p.g; a.x = x; a.in_scale = in_scale; a.in_shift = in_shift; a.in_slope = in_slope;
  a.dy = dy; a.dy_ldc = dy_ldc; a.partial = partial; a.ngroups = p.ngroups;
  a.tiles_per_split = p.tps; a.total_tiles = p.total; a.slab_floats = p.slab;
  int base = 0;
  for (int c = 0; c < 8; ++c) { a.cls_slab_base[c] = base; if (c < p.ncls) base += p.nchunks * p.ngroups * (p.g.cls_ntaps[c] + 1) * p.CG; }
  const size_t lds = ((size_t)p.g.ID * p.g.IH * p.g.IW * 16 + (size_t)p.g.TD * p.g.TH * 16 * p.CG * 16) * sizeof(float);
  if (lds > 160 * 1024) return CWF_E_TOOLARGE;
  dim3 grid(p.wg_splits, p.nchunks * p.ngroups, p.ncls);
  hipStream_t st = cwf_stream(stream);
#define CWF_WG(tpw, ntw, ts) do { CWF_MAX_LDS_ONCE((&wgrad_mfma_kernel<tpw, ntw, ts>)); \
    hipLaunchKernelGGL((wgrad_mfma_kernel<tpw, ntw, ts>), grid, dim3(256), lds, st, a); } while (0)
  if (p.tapsplit) { if (p.CG == 1) CWF_WG(7, 1, true); else CWF_WG(7, 2, true); }
  else { if (p.CG == 1) CWF_WG(2, 1, false); else if (p.CG == 2) CWF_WG(2, 2, false); else CWF_WG(2, 4, false); }
#undef CWF_WG
  CWF_LAUNCH_CHECK();
  return 0;
}

extern "C" int cwf_wgrad_reduce(const float* partial, int nsplit, int64_t slab_floats,
                                const int32_t* inv_map, float* dW, float* db, void* stream) {
  if (!partial || nsplit <= 0 || slab_floats <= 0 || (slab_floats & 3) || !inv_map || !dW) return CWF_E_BADARG;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(slab_floats / 4, 256)), dim3(256), 0, cwf_stream(stream),
                     partial, nsplit, slab_floats, inv_map, dW, db);
  CWF_LAUNCH_CHECK();
  return 0;
}
