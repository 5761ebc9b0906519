// K1 -- implicit-GEMM 3-D convolution family on v_mfma_f32_16x16x4_f32 (exact fp32), gfx950.
//
// One tap-table driven kernel serves every forward conv of the model and every data gradient
// (see include/cwf_hip.h).  GEMM view: M = output voxels (an M-tile = 16 consecutive voxels along W),
// N = output channels (N-tile = 16), K = taps x input channels (chunks of 16 channels).
//
// Per workgroup (256 threads = 4 waves, wave grid WM x WN):
//   - the input halo tile of one 16-channel chunk is staged global -> registers -> LDS as [voxel][16 ch],
//     with the producer's InstanceNorm + (Leaky)ReLU applied on the way (fused prologue); zero padding is
//     materialised in LDS so the inner loop has no bounds checks;
//   - per tap, a lane reads ONE 16-byte vector of A (4 channels of its voxel) and one of B (packed weights,
//     lane-linear 1 KiB blocks streamed from L2) and issues 4 MFMAs: the MFMA k index is a free summation
//     index, so lane group kq supplies channels 4kq..4kq+3 over the 4 instructions;
//   - epilogue: bias, residual, per-(n,c) output scale (dropout3d), store, and the InstanceNorm statistics
//     (sum, sum of squares) of the OUTPUT reduced wave -> workgroup -> one f64 atomic per channel.
//
// Reference ops replaced: nn.Conv3d / nn.ConvTranspose3d + InstanceNorm3d + ReLU/LeakyReLU chains at
// Unet_skipconnection.py:22-78, cls_wise_former.py:157-204,257-273,284-324,614-754, SuperviseLabel.py:10-51.
#include "common.h"

struct ConvArgs {
  ConvGeom g;
  const float* x; const float* wpk; const float* bias; float* y;
  const float* in_scale; const float* in_shift; float in_slope;
  const float* residual; int r_ldc; const float* out_scale; double* stats;
};

template <int MT, int NT, int WM>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
  constexpr int WN = 4 / WM;
  extern __shared__ float4 lds4[];
  float* lds = reinterpret_cast<float*>(lds4);
  const ConvGeom& g = a.g;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 15, kq = lane >> 4;

  int bx = blockIdx.x;
  const int tile_w = bx % g.tiles_w; bx /= g.tiles_w;
  const int tile_h = bx % g.tiles_h;
  const int tile_d = bx / g.tiles_h;
  const int n = blockIdx.z / g.ncls, cls = blockIdx.z % g.ncls;
  const int Dc = g.cls_dims[cls][0], Hc = g.cls_dims[cls][1], Wc = g.cls_dims[cls][2];
  const int od0 = tile_d * g.TD, oh0 = tile_h * g.TH, ow0 = tile_w * 16;
  if (od0 >= Dc || oh0 >= Hc || ow0 >= Wc) return;   // class grids can be smaller than the launch grid (uniform exit)
  const int ntaps = g.cls_ntaps[cls];
  const int* tapofs = g.tapofs + (g.ncls > 1 ? cls * 8 : 0);
  const int nt0 = (blockIdx.y * WN + wn) * NT;         // first N-tile of this wave

  // lane's LDS voxel base per M-tile
  int abase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mt = wm * MT + m;
    const int td = mt / g.TH, th = mt % g.TH;
    abase[m] = ((td * g.is) * g.IH + th * g.is) * g.IW + r * g.is;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int id0 = od0 * g.is + g.lo[0], ih0 = oh0 * g.is + g.lo[1], iw0 = ow0 * g.is + g.lo[2];
  const float slope = a.in_slope;
  const float4* wp4 = reinterpret_cast<const float4*>(a.wpk);

  for (int chunk = 0; chunk < g.nchunks; ++chunk) {
    if (chunk) __syncthreads();
    cwf_stage_input_tile(lds, g, a.x, a.in_scale, a.in_shift, slope, n, chunk, id0, ih0, iw0, tid);
    __syncthreads();

    // ---- taps
    const float4* wchunk = wp4 + ((int64_t)g.cls_wbase[cls] + (int64_t)chunk * ntaps * g.ntiles) * 64 + lane;
#pragma unroll 1
    for (int t = 0; t < ntaps; ++t) {
      float4 b[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        b[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nt0 + j < g.ntiles) b[j] = wchunk[((int64_t)t * g.ntiles + nt0 + j) * 64];
      }
      const int to = tapofs[t];
      float4 av[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) av[m] = *reinterpret_cast<const float4*>(lds + (abase[m] + to) * 16 + kq * 4);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].x, b[j].x, acc[m][j], 0, 0, 0);
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].y, b[j].y, acc[m][j], 0, 0, 0);
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].z, b[j].z, acc[m][j], 0, 0, 0);
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].w, b[j].w, acc[m][j], 0, 0, 0);
        }
    }
  }

  // ---- epilogue.  acc[m][j][i]: voxel row = kq*4+i of M-tile m, channel = (nt0+j)*16 + r
  const int os = g.os;
  const int of0 = g.cls_ooff[cls][0], of1 = g.cls_ooff[cls][1], of2 = g.cls_ooff[cls][2];
  float s1[NT], s2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = (nt0 + j) * 16 + r;
    if (co >= g.Cout) continue;
    const float bv = a.bias ? a.bias[co] : 0.f;
    const float osc = a.out_scale ? a.out_scale[(int64_t)n * g.Cout + co] : 1.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int mt = wm * MT + m;
      const int od = od0 + mt / g.TH, oh = oh0 + mt % g.TH;
      if (od >= Dc || oh >= Hc) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ow = ow0 + kq * 4 + i;
        if (ow >= Wc) continue;
        const int64_t vox = (((int64_t)n * g.Do + (od * os + of0)) * g.Ho + (oh * os + of1)) * g.Wo + (ow * os + of2);
        float v = acc[m][j][i] + bv;
        if (a.residual) v += a.residual[vox * a.r_ldc + co];
        v *= osc;
        a.y[vox * g.y_ldc + co] = v;
        s1[j] += v; s2[j] += v * v;
      }
    }
  }
  if (a.stats) {
    // lanes l, l^16, l^32, l^48 hold the same channel -> fold kq, then fold the WM waves through LDS
    __syncthreads();                       // everyone is done reading the staged tile
    float* red = lds;                      // [WM][WN][NT][16][2]
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float u1 = s1[j], u2 = s2[j];
      u1 += __shfl_xor(u1, 16, 64); u1 += __shfl_xor(u1, 32, 64);
      u2 += __shfl_xor(u2, 16, 64); u2 += __shfl_xor(u2, 32, 64);
      if (kq == 0) {
        red[(((wm * WN + wn) * NT + j) * 16 + r) * 2 + 0] = u1;
        red[(((wm * WN + wn) * NT + j) * 16 + r) * 2 + 1] = u2;
      }
    }
    __syncthreads();
    if (tid < WN * NT * 16 * 2) {
      const int which = tid & 1, rr = (tid >> 1) & 15, jj = (tid >> 5) % NT, ww = (tid >> 5) / NT;
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < WM; ++w) s += (double)red[(((w * WN + ww) * NT + jj) * 16 + rr) * 2 + which];
      const int co = ((blockIdx.y * WN + ww) * NT + jj) * 16 + rr;
      if (co < g.Cout) atomic_add_f64(a.stats + ((int64_t)n * g.Cout + co) * 2 + which, s);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
int cwf_build_geom(ConvGeom& g, int op, int N, int Di, int Hi, int Wi, int Cin, int x_ldc,
                   int Do, int Ho, int Wo, int Cout, int y_ldc, int MTOT) {
  g.N = N; g.Di = Di; g.Hi = Hi; g.Wi = Wi; g.Cin = Cin; g.x_ldc = x_ldc;
  g.Do = Do; g.Ho = Ho; g.Wo = Wo; g.Cout = Cout; g.y_ldc = y_ldc;
  g.nchunks = cdiv(Cin, 16); g.ntiles = cdiv(Cout, 16);
  g.is = 1; g.os = 1; g.ncls = 1;
  switch (MTOT) {
    case 16: g.TD = 4; g.TH = 4; break;
    case 8: g.TD = 2; g.TH = 4; break;
    case 4: g.TD = 2; g.TH = 2; break;
    case 2: g.TD = 1; g.TH = 2; break;
    case 1: g.TD = 1; g.TH = 1; break;
    default: return CWF_E_BADARG;
  }
  int hi[3] = {0, 0, 0};
  g.lo[0] = g.lo[1] = g.lo[2] = 0;
  int tapd[8][27][3];
  for (int c = 0; c < 8; ++c) { g.cls_ntaps[c] = 0; g.cls_wbase[c] = 0; for (int k = 0; k < 3; ++k) { g.cls_ooff[c][k] = 0; g.cls_dims[c][k] = 0; } }
  auto set_dims = [&](int c, int d, int h, int w) { g.cls_dims[c][0] = d; g.cls_dims[c][1] = h; g.cls_dims[c][2] = w; };
  if (op == CWF_CONV3_S1 || op == CWF_CONV3_S2) {
    g.is = (op == CWF_CONV3_S2) ? 2 : 1;
    if (g.is == 1 && (Do != Di || Ho != Hi || Wo != Wi)) return CWF_E_BADARG;
    if (g.is == 2 && (Do != (Di - 1) / 2 + 1 || Ho != (Hi - 1) / 2 + 1 || Wo != (Wi - 1) / 2 + 1)) return CWF_E_BADARG;
    g.cls_ntaps[0] = 27;
    for (int t = 0; t < 27; ++t) { tapd[0][t][0] = t / 9 - 1; tapd[0][t][1] = (t / 3) % 3 - 1; tapd[0][t][2] = t % 3 - 1; }
    g.lo[0] = g.lo[1] = g.lo[2] = -1; hi[0] = hi[1] = hi[2] = 1;
    set_dims(0, Do, Ho, Wo);
  } else if (op == CWF_CONV1) {
    if (Do != Di || Ho != Hi || Wo != Wi) return CWF_E_BADARG;
    g.cls_ntaps[0] = 1; tapd[0][0][0] = tapd[0][0][1] = tapd[0][0][2] = 0;
    set_dims(0, Do, Ho, Wo);
  } else if (op == CWF_CONVT2) {
    if (Do != 2 * Di || Ho != 2 * Hi || Wo != 2 * Wi) return CWF_E_BADARG;
    g.ncls = 8; g.os = 2;
    for (int c = 0; c < 8; ++c) {
      g.cls_ntaps[c] = 1; tapd[c][0][0] = tapd[c][0][1] = tapd[c][0][2] = 0;
      g.cls_ooff[c][0] = (c >> 2) & 1; g.cls_ooff[c][1] = (c >> 1) & 1; g.cls_ooff[c][2] = c & 1;
      set_dims(c, Di, Hi, Wi);
    }
  } else if (op == CWF_CONV3_S2_DGRAD) {
    // x = dy [Di = (Do-1)/2+1 ...], y = dx [Do,Ho,Wo].  dx[2j+p]: p=0 -> (delta 0, k 1); p=1 -> (delta 0, k 2), (delta 1, k 0)
    if (Di != (Do - 1) / 2 + 1 || Hi != (Ho - 1) / 2 + 1 || Wi != (Wo - 1) / 2 + 1) return CWF_E_BADARG;
    g.ncls = 8; g.os = 2; hi[0] = hi[1] = hi[2] = 1;
    for (int c = 0; c < 8; ++c) {
      const int p[3] = {(c >> 2) & 1, (c >> 1) & 1, c & 1};
      int nt = 0;
      for (int a0 = 0; a0 <= p[0]; ++a0) for (int a1 = 0; a1 <= p[1]; ++a1) for (int a2 = 0; a2 <= p[2]; ++a2) {
        tapd[c][nt][0] = a0; tapd[c][nt][1] = a1; tapd[c][nt][2] = a2; ++nt;
      }
      g.cls_ntaps[c] = nt;
      for (int k = 0; k < 3; ++k) g.cls_ooff[c][k] = p[k];
      set_dims(c, (Do - p[0] + 1) / 2, (Ho - p[1] + 1) / 2, (Wo - p[2] + 1) / 2);
    }
  } else if (op == CWF_CONVT2_DGRAD) {
    if (Di != 2 * Do || Hi != 2 * Ho || Wi != 2 * Wo) return CWF_E_BADARG;
    g.is = 2; g.cls_ntaps[0] = 8; hi[0] = hi[1] = hi[2] = 1;
    for (int t = 0; t < 8; ++t) { tapd[0][t][0] = (t >> 2) & 1; tapd[0][t][1] = (t >> 1) & 1; tapd[0][t][2] = t & 1; }
    set_dims(0, Do, Ho, Wo);
  } else {
    return CWF_E_BADARG;
  }
  g.ID = (g.TD - 1) * g.is + (hi[0] - g.lo[0] + 1);
  g.IH = (g.TH - 1) * g.is + (hi[1] - g.lo[1] + 1);
  g.IW = (16 - 1) * g.is + (hi[2] - g.lo[2] + 1);
  int wb = 0, wb16 = 0, md = 0, mh = 0, mw = 0;
  for (int c = 0; c < 8; ++c) g.cls_wbase16[c] = 0;
  for (int c = 0; c < g.ncls; ++c) {
    g.cls_wbase[c] = wb;
    wb += g.nchunks * g.cls_ntaps[c] * g.ntiles;
    g.cls_wbase16[c] = wb16;
    wb16 += g.nchunks * ((g.cls_ntaps[c] + 1) / 2) * g.ntiles;
    for (int t = 0; t < g.cls_ntaps[c]; ++t)
      g.tapofs[(g.ncls > 1 ? c * 8 : 0) + t] =
          ((tapd[c][t][0] - g.lo[0]) * g.IH + (tapd[c][t][1] - g.lo[1])) * g.IW + (tapd[c][t][2] - g.lo[2]);
    md = g.cls_dims[c][0] > md ? g.cls_dims[c][0] : md;
    mh = g.cls_dims[c][1] > mh ? g.cls_dims[c][1] : mh;
    mw = g.cls_dims[c][2] > mw ? g.cls_dims[c][2] : mw;
  }
  g.tiles_d = cdiv(md, g.TD); g.tiles_h = cdiv(mh, g.TH); g.tiles_w = cdiv(mw, 16);
  return 0;
}

namespace {
struct TileCfg { int MT, NT, WM; };

// Tile choice: enough workgroups to fill 256 CUs first, then the most register-level reuse.
TileCfg choose_cfg(int op, const int cdims[3], int ncls, int N, int ntiles) {
  const bool s2 = (op == CWF_CONV3_S2 || op == CWF_CONVT2_DGRAD);   // 2x input tile -> small spatial tiles only
  static const TileCfg all[] = {{4, 4, 1}, {2, 4, 2}, {2, 4, 4}, {4, 2, 4}, {4, 1, 4}, {1, 4, 4}, {1, 2, 4}, {1, 2, 2}, {1, 1, 4}};
  TileCfg best = {1, 1, 4};
  double best_score = -1.0;
  for (const TileCfg& c : all) {
    const int WN = 4 / c.WM, MTOT = c.MT * c.WM, NTOT = c.NT * WN;
    if (s2 && MTOT > 4) continue;
    int TD, TH;
    switch (MTOT) { case 16: TD = 4; TH = 4; break; case 8: TD = 2; TH = 4; break; case 4: TD = 2; TH = 2; break; case 2: TD = 1; TH = 2; break; default: TD = 1; TH = 1; }
    const double sp_tiles = (double)cdiv(cdims[0], TD) * cdiv(cdims[1], TH) * cdiv(cdims[2], 16);
    const double sp_eff = ((double)cdims[0] * cdims[1] * cdims[2]) / (sp_tiles * MTOT * 16);
    const int ngrp = cdiv(ntiles, NTOT);
    const double n_eff = (double)ntiles / (ngrp * NTOT);
    const double nwg = sp_tiles * ngrp * N * ncls;
    const double fill = nwg >= 512 ? 1.0 : nwg / 512.0;
    const double reuse = 1.0 - 0.35 / (c.MT * c.NT) - 0.1 / MTOT;
    const double score = sp_eff * n_eff * fill * reuse;
    if (score > best_score) { best_score = score; best = c; }
  }
  return best;
}

template <int MT, int NT, int WM>
int launch_cfg(const ConvArgs& a, hipStream_t st) {
  constexpr int WN = 4 / WM;
  const ConvGeom& g = a.g;
  const size_t lds_tile = (size_t)g.ID * g.IH * g.IW * 16 * sizeof(float);
  const size_t lds_red = (size_t)4 * NT * 16 * 2 * sizeof(float);
  const size_t lds = lds_tile > lds_red ? lds_tile : lds_red;
  if (lds > 160 * 1024) return CWF_E_TOOLARGE;
  CWF_MAX_LDS_ONCE((&conv_mfma_kernel<MT, NT, WM>));
  dim3 grid(g.tiles_d * g.tiles_h * g.tiles_w, cdiv(g.ntiles, WN * NT), g.N * g.ncls);
  hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, WM>), grid, dim3(256), lds, st, a);
  CWF_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int cwf_conv_mfma(int op, const float* x, int x_ldc, const float* wpk, const float* bias,
                             float* y, int y_ldc, const float* in_scale, const float* in_shift, float in_slope,
                             const float* residual, int r_ldc, const float* out_scale, double* stats,
                             int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, void* stream) {
  if (!x || !wpk || !y || N <= 0 || Cin <= 0 || Cout <= 0) return CWF_E_BADARG;
  if ((Cin & 3) || (x_ldc & 3) || x_ldc < Cin || y_ldc < Cout) return CWF_E_ALIGN;
  if (((uintptr_t)x & 15) || ((uintptr_t)wpk & 15)) return CWF_E_ALIGN;
  ConvArgs a;
  // class grid extents for the heuristic
  int cd[3] = {Do, Ho, Wo}; int ncls = 1;
  if (op == CWF_CONVT2) { cd[0] = Di; cd[1] = Hi; cd[2] = Wi; ncls = 8; }
  if (op == CWF_CONV3_S2_DGRAD) { cd[0] = (Do + 1) / 2; cd[1] = (Ho + 1) / 2; cd[2] = (Wo + 1) / 2; ncls = 8; }
  const TileCfg c = choose_cfg(op, cd, ncls, N, cdiv(Cout, 16));
  int rc = cwf_build_geom(a.g, op, N, Di, Hi, Wi, Cin, x_ldc, Do, Ho, Wo, Cout, y_ldc, c.MT * c.WM);
  if (rc) return rc;
  a.x = x; a.wpk = wpk; a.bias = bias; a.y = y; a.in_scale = in_scale; a.in_shift = in_shift; a.in_slope = in_slope;
  a.residual = residual; a.r_ldc = r_ldc; a.out_scale = out_scale; a.stats = stats;
  if (in_scale && !in_shift) return CWF_E_BADARG;
  hipStream_t st = cwf_stream(stream);
#define CWF_CFG(mt, nt, wm) if (c.MT == mt && c.NT == nt && c.WM == wm) return launch_cfg<mt, nt, wm>(a, st);
  CWF_CFG(4, 4, 1) CWF_CFG(2, 4, 2) CWF_CFG(2, 4, 4) CWF_CFG(4, 2, 4) CWF_CFG(4, 1, 4)
  CWF_CFG(1, 4, 4) CWF_CFG(1, 2, 4) CWF_CFG(1, 2, 2) CWF_CFG(1, 1, 4)
#undef CWF_CFG
  return CWF_E_BADARG;
}
