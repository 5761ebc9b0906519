// K1, weight-stationary form -- the 3x3x3 stride-1 convolutions of the 32 / 64 / 128-channel levels (EnBlock / DeBlock / EnBlock2 at
// 64^3, 32^3, 16^3: Unet_skipconnection.py:36-57, cls_wise_former.py:691-754) and their data gradients.
//
// Why a second kernel (profiles/round3_generic_conv_pmc.txt, rocprofv3 --pmc on conv_bf16_kernel<4,2,4>): the tap-table kernel gives a
// workgroup ONE 256-voxel tile; per 16-channel chunk it stages the halo (global -> registers -> LDS, nothing else running in that
// workgroup meanwhile) and then streams the packed weights of every tap pair from L2 with one step of lookahead.  Its waves sit in
// s_waitcnt / s_barrier 31 % (split-bf16 forward) to 47 % (single-bf16 data gradient) of their cycles, the matrix pipe is 46 % / 18 %
// busy, and the weight stream (110 KB per tile in the split form) is as much L2 -> CU traffic as the activations.
//
// Here the weights do not move: a persistent 8-wave workgroup (one per CU) copies the packed weights of ITS 16- or 32-channel output
// group -- all taps, all input chunks, hi and lo images -- into LDS once (<= 112 KB) and walks a contiguous range of 4x4x16 output
// tiles.  Per (tile, chunk) iteration every wave
//     converts + writes the halo it prefetched (fused InstanceNorm + activation prologue, bf16 hi / lo split)  -> one LDS A image,
//     issues the global loads of the NEXT iteration (they stay in flight across the barrier and the whole MFMA phase),
//     runs its 14 tap-pair steps: A and B fragments by ds_read_b128 at immediate offsets, v_mfma_f32_16x16x32_bf16.
// Wave w owns M-tiles 2w, 2w+1 (rows (w/2, 2(w&1) + m) of the tile) and all NT output-channel tiles of the group.  The epilogue
// (bias, residual, InstanceNorm statistics or norm-backward sums, stores) is the tap-table kernel's interior fast path; statistics
// stay in registers over all tiles of a sample.
#include "conv_args.h"
#include <cstdlib>
#include <type_traits>

#define WS_ID 6
#define WS_IH 6
#define WS_IW 18
#define WS_NVOX (WS_ID * WS_IH * WS_IW)      // 648 halo voxels of a 4x4x16 tile
#define WS_SLOTS 6                           // staging slots per thread: 648 voxels x 4 channel quads / 512 threads

struct WsWork { int ngroups, slots, tiles, xcd_perm, diag; };   // diag: ablation bits of the DIAG build (CWF_WS_DIAG): 1 no loads, 2 no MFMA phase, 4 no epilogue, 8 no convert

__host__ __device__ constexpr int ws_tap_bytes(int t) { return (((t / 9) * WS_IH + (t / 3) % 3) * WS_IW + t % 3) * 32; }

template <bool X3, int NT, bool DIAG>
__global__ __launch_bounds__(512) void convws_kernel(const ConvArgsB a, const WsWork wk) {
  extern __shared__ float4 lds4[];
  const ConvGeom& g = a.g;
  const int nch = g.nchunks;
  char* lds = reinterpret_cast<char*>(lds4);
  const int b_bytes = nch * 14 * NT * 1024;
  char* Bh = lds;
  char* Bl = lds + b_bytes;
  char* Ah = lds + b_bytes * (X3 ? 2 : 1);
  char* Al = Ah + WS_NVOX * 32;
  float* red = reinterpret_cast<float*>(Ah + WS_NVOX * 32 * (X3 ? 2 : 1));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kq = lane >> 4;
  const bool second = (kq >> 1) != 0;

  // ---- work: output-channel group and a contiguous range of spatial tiles (contiguous per XCD: neighbouring tiles share halo rows in L2)
  int grp, slot;
  {
    const int b = blockIdx.x;
    if (wk.xcd_perm) {
      const int per = gridDim.x >> 3, j = b >> 3;
      grp = j % wk.ngroups;
      slot = (b & 7) * (per / wk.ngroups) + j / wk.ngroups;
    } else {
      grp = b % wk.ngroups;
      slot = b / wk.ngroups;
    }
  }
  const int t_begin = (int)(((int64_t)slot * wk.tiles) / wk.slots), t_end = (int)(((int64_t)(slot + 1) * wk.tiles) / wk.slots);
  const int nt0 = grp * NT;

  // ---- packed weights of this group -> LDS (hi image, lo image), lane-linear 1 KB blocks [chunk][step][j]
  {
    const uint4* wsrc = a.wpk + (int64_t)g.cls_wbase16[0] * 128;
    const int nblk = nch * 14 * NT * 64;
    for (int e = tid; e < nblk; e += 512) {
      const int ln = e & 63, blk = e >> 6;
      const int j = blk % NT, cs = blk / NT;
      const uint4* p = wsrc + ((int64_t)cs * g.ntiles + nt0 + j) * 128 + ln * 2;
      *reinterpret_cast<uint4*>(Bh + (int64_t)e * 16) = p[0];
      if (X3) *reinterpret_cast<uint4*>(Bl + (int64_t)e * 16) = p[1];
    }
  }

  // ---- staging slots of this thread: halo voxel v = (tid >> 2) + 128 i, channel quad q = tid & 3
  const int q = tid & 3;
  int loc[WS_SLOTS];                                       // idd | ih << 8 | iw << 16, or -1
#pragma unroll
  for (int i = 0; i < WS_SLOTS; ++i) {
    const int v = (tid >> 2) + 128 * i;
    const int iw = v % WS_IW, t2 = v / WS_IW;
    loc[i] = v < WS_NVOX ? ((t2 / WS_IH) | ((t2 % WS_IH) << 8) | (iw << 16)) : -1;
  }
  const int tiles_per_n = g.tiles_d * g.tiles_h * g.tiles_w;
  const bool plain = a.in_scale == nullptr && a.in_slope == 1.f;

  // The prefetch loads are written in inline asm: hipcc waits for every load IT tracks before the first LDS read of the MFMA phase
  // (s_waitcnt vmcnt(0) right behind the issue block -- the whole point of the prefetch lost: 99.7 us instead of 60 at 32 ch @ 64^3);
  // loads inside asm are invisible to its counters.  Their completion is counted by hand: ONE s_waitcnt vmcnt(0) at the head of the
  // next iteration (pf_wait, naming every destination register).  hipcc's own counted waits stay correct beside them because its
  // loads (epilogue residual / norm-backward operands, prologue parameters) are always YOUNGER than every asm load in flight, and
  // vector-memory operations return in order.
  f32x4 pf[WS_SLOTS];
  unsigned pf_inb = 0u;
  auto issue_loads = [&](int tile, int chunk) {
    const int n = tile / tiles_per_n;
    int bx = tile - n * tiles_per_n;
    const int tw = bx % g.tiles_w; bx /= g.tiles_w;
    const int th = bx % g.tiles_h, td = bx / g.tiles_h;
    const int id0 = td * 4 - 1, ih0 = th * 4 - 1, iw0 = tw * 16 - 1;
    const float* xb = a.x + (int64_t)n * g.Di * g.Hi * g.Wi * g.x_ldc + chunk * 16;      // wave-uniform (SGPR pair)
    const unsigned ldc4 = (unsigned)g.x_ldc * 4u;
    pf_inb = 0u;
#pragma unroll
    for (int i = 0; i < WS_SLOTS; ++i) {
      const int gd = id0 + (loc[i] & 255), gh = ih0 + ((loc[i] >> 8) & 255), gw = iw0 + (loc[i] >> 16);
      const bool ok = (loc[i] >= 0) & (gd >= 0) & (gd < g.Di) & (gh >= 0) & (gh < g.Hi) & (gw >= 0) & (gw < g.Wi);
      const unsigned lin = (unsigned)((gd * g.Hi + gh) * g.Wi + gw) * ldc4 + (unsigned)q * 16u;
      const unsigned boff = ok ? lin : (unsigned)q * 16u;          // (clamped: the load itself is unconditional)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(pf[i]) : "v"(boff), "s"(xb) : "memory");
      pf_inb |= ok ? (1u << i) : 0u;
    }
  };
  auto pf_wait = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]), "+v"(pf[4]), "+v"(pf[5]) :: "memory");
  };
  auto convert_write = [&](int tile, int chunk) {
    const int n = tile / tiles_per_n;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (a.in_scale) {
      sc = *reinterpret_cast<const f32x4*>(a.in_scale + (int64_t)n * g.Cin + chunk * 16 + q * 4);
      sh = *reinterpret_cast<const f32x4*>(a.in_shift + (int64_t)n * g.Cin + chunk * 16 + q * 4);
    }
    // a use on EVERY path: hipcc must consider these two loads retired here.  Left pending in its model on the path that never reads
    // them (no prologue), it protects their destination registers with s_waitcnt vmcnt(0) in front of the MFMA phase's first LDS
    // reads -- which in hardware also waits for the prefetch just issued.
    asm volatile("" : "+v"(sc), "+v"(sh));
    const float slope = a.in_slope;
#pragma unroll
    for (int i = 0; i < WS_SLOTS; ++i) {
      if (loc[i] < 0) continue;
      float v0 = pf[i][0], v1 = pf[i][1], v2 = pf[i][2], v3 = pf[i][3];
      if (!plain) {
        v0 = act01(fmaf(v0, sc[0], sh[0]), slope); v1 = act01(fmaf(v1, sc[1], sh[1]), slope);
        v2 = act01(fmaf(v2, sc[2], sh[2]), slope); v3 = act01(fmaf(v3, sc[3], sh[3]), slope);
      }
      uint2 h, l;
      if (X3) { split_bf16(v0, v1, h.x, l.x); split_bf16(v2, v3, h.y, l.y); }
      else { h.x = pack_bf16(v0, v1); h.y = pack_bf16(v2, v3); l = make_uint2(0u, 0u); }
      const bool was = (pf_inb >> i) & 1u;                // zero padding applies AFTER the activation
      h.x = was ? h.x : 0u; h.y = was ? h.y : 0u; l.x = was ? l.x : 0u; l.y = was ? l.y : 0u;
      const int vo = (((tid >> 2) + 128 * i) * 16 + q * 4) * 2;
      *reinterpret_cast<uint2*>(Ah + vo) = h;
      if (X3) *reinterpret_cast<uint2*>(Al + vo) = l;
    }
  };

  // ---- MFMA operands: lane base addresses (bytes); everything else is an immediate
  const int a_lane = (((wave >> 1) * WS_IH + (wave & 1) * 2) * WS_IW + r) * 32 + (kq & 1) * 16;
  const int b_lane = lane * 16;

  f32x4 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float s1[NT], s2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  int stats_n = -1;

  auto flush_stats = [&]() {                               // (all threads; called at a sample boundary and at the end)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float u1 = s1[j], u2 = s2[j];
      u1 += __shfl_xor(u1, 16, 64); u1 += __shfl_xor(u1, 32, 64);
      u2 += __shfl_xor(u2, 16, 64); u2 += __shfl_xor(u2, 32, 64);
      if (kq == 0) {
        red[((wave * NT + j) * 16 + r) * 2 + 0] = u1;
        red[((wave * NT + j) * 16 + r) * 2 + 1] = u2;
      }
      s1[j] = 0.f; s2[j] = 0.f;
    }
    __syncthreads();
    if (tid < NT * 16 * 2) {
      const int which = tid & 1, rr = (tid >> 1) & 15, jj = tid >> 5;
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += (double)red[((w * NT + jj) * 16 + rr) * 2 + which];
      const int co = (nt0 + jj) * 16 + rr;
      atomic_add_f64(a.stats + ((int64_t)stats_n * g.Cout + co) * 2 + which, s);
    }
  };

  auto epilogue = [&](int tile) {
    const int n = tile / tiles_per_n;
    int bx = tile - n * tiles_per_n;
    const int tw = bx % g.tiles_w; bx /= g.tiles_w;
    const int th = bx % g.tiles_h, td = bx / g.tiles_h;
    if (a.stats && n != stats_n) {
      if (stats_n >= 0) flush_stats();
      stats_n = n;
    }
    float bvj[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bvj[j] = a.bias ? a.bias[(nt0 + j) * 16 + r] : 0.f;
    auto epi = [&](auto HR, auto HT, auto HN) {
      constexpr bool HAS_RES = decltype(HR)::value, HAS_STATS = decltype(HT)::value, HAS_NB = decltype(HN)::value;
      unsigned yo[4], ro[4], xo[4];
      float nsc[NT], nsh[NT];
#pragma unroll
      for (int i = 0; i < 4; ++i) {                        // opaque copies: keep the zero-extension in this block (saddr form)
        yo[i] = (unsigned)((kq * 4 + i) * g.y_ldc + nt0 * 16 + r) * 4u; asm volatile("" : "+v"(yo[i]));
        if (HAS_RES) { ro[i] = (unsigned)((kq * 4 + i) * a.r_ldc + nt0 * 16 + r) * 4u; asm volatile("" : "+v"(ro[i])); }
        if (HAS_NB) { xo[i] = (unsigned)((kq * 4 + i) * a.nb_ldc + nt0 * 16 + r) * 4u; asm volatile("" : "+v"(xo[i])); }
      }
      if (HAS_NB) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          nsc[j] = a.nb_scale[(int64_t)n * g.Cout + (nt0 + j) * 16 + r];
          nsh[j] = a.nb_shift[(int64_t)n * g.Cout + (nt0 + j) * 16 + r];
        }
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int od = td * 4 + (wave >> 1), oh = th * 4 + (wave & 1) * 2 + m;
        const int64_t vox0 = (((int64_t)n * g.Do + od) * g.Ho + oh) * g.Wo + tw * 16;
        char* yb = reinterpret_cast<char*>(a.y + vox0 * g.y_ldc);
        const char* rb = HAS_RES ? reinterpret_cast<const char*>(a.residual + vox0 * a.r_ldc) : nullptr;
        const char* xb = HAS_NB ? reinterpret_cast<const char*>(a.nb_x + vox0 * a.nb_ldc) : nullptr;
        float rv[NT][4], xv[NT][4];
        if (HAS_RES) {
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) rv[j][i] = *reinterpret_cast<const float*>(rb + ro[i] + j * 64);
        }
        if (HAS_NB) {
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[j][i] = *reinterpret_cast<const float*>(xb + xo[i] + j * 64);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = acc[m][j][i] + bvj[j];
            if (HAS_RES) v += rv[j][i];
            *reinterpret_cast<float*>(yb + yo[i] + j * 64) = v;
            if (HAS_NB) {
              const float h = fmaf(xv[j][i], nsc[j], nsh[j]);
              const float gn = v * (h > 0.f ? 1.f : a.nb_slope);
              s1[j] += gn; s2[j] = fmaf(gn, h, s2[j]);
            } else if (HAS_STATS) { s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
            acc[m][j][i] = 0.f;
          }
      }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    if (a.nb_x) { if (a.residual) epi(T_{}, T_{}, T_{}); else epi(F_{}, T_{}, T_{}); }
    else if (a.residual) { if (a.stats) epi(T_{}, T_{}, F_{}); else epi(T_{}, F_{}, F_{}); }
    else            { if (a.stats) epi(F_{}, T_{}, F_{}); else epi(F_{}, F_{}, F_{}); }
  };

  // ---- main loop over (tile, chunk) iterations
  const int n_it = (t_end - t_begin) * nch;
  if (n_it > 0 && !(DIAG && (wk.diag & 1))) issue_loads(t_begin, 0);
  int tile = t_begin, chunk = 0;
  for (int it = 0; it < n_it; ++it) {
    if (!(DIAG && (wk.diag & 1))) pf_wait();
    __syncthreads();                                       // every wave is done reading the A image (and, first time, B is in LDS)
    if (!(DIAG && (wk.diag & 8))) convert_write(tile, chunk);
    __syncthreads();
    {
      int tn = tile, cn = chunk + 1;
      if (cn == nch) { cn = 0; ++tn; }
      if (it + 1 < n_it && !(DIAG && (wk.diag & 1))) issue_loads(tn, cn);   // in flight across the MFMA phase and the next barrier
    }
    const char* bh = Bh + chunk * (14 * NT * 1024) + b_lane;
    const char* bl = Bl + chunk * (14 * NT * 1024) + b_lane;
    const char* ah0 = Ah + a_lane;
    const char* al0 = Al + a_lane;
    if (!(DIAG && (wk.diag & 2)))
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const int c0 = ws_tap_bytes(2 * s), c1 = ws_tap_bytes(2 * s + 1 < 27 ? 2 * s + 1 : 2 * s);   // padded tap: zero weights, valid address
      const int to = second ? c1 : c0;
      uint4 fa[2], fl[2], fb[NT], fbl[NT];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        fa[m] = *reinterpret_cast<const uint4*>(ah0 + to + m * (WS_IW * 32));
        if (X3) fl[m] = *reinterpret_cast<const uint4*>(al0 + to + m * (WS_IW * 32));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        fb[j] = *reinterpret_cast<const uint4*>(bh + (s * NT + j) * 1024);
        if (X3) fbl[j] = *reinterpret_cast<const uint4*>(bl + (s * NT + j) * 1024);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[m]), __builtin_bit_cast(bf16x8, fb[j]), acc[m][j], 0, 0, 0);
          if (X3) {
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[m]), __builtin_bit_cast(bf16x8, fbl[j]), acc[m][j], 0, 0, 0);
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fl[m]), __builtin_bit_cast(bf16x8, fb[j]), acc[m][j], 0, 0, 0);
          }
        }
    }
    if (chunk == nch - 1 && !(DIAG && (wk.diag & 4))) epilogue(tile);
    if (++chunk == nch) { chunk = 0; ++tile; }
  }
  if (a.stats && stats_n >= 0) flush_stats();
}

namespace {
template <bool X3, int NT, bool DIAG>
int launch_ws_impl(const ConvArgsB& a, const WsWork& wk, int grid, hipStream_t st) {
  const size_t lds = (size_t)a.g.nchunks * 14 * NT * 1024 * (X3 ? 2 : 1) + (size_t)WS_NVOX * 32 * (X3 ? 2 : 1) + 8 * NT * 16 * 2 * sizeof(float);
  if (lds > 160 * 1024) return CWF_E_TOOLARGE;
  static bool attr_set = false;                            // (one process per GPU: set once per process)
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convws_kernel<X3, NT, DIAG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((convws_kernel<X3, NT, DIAG>), dim3(grid), dim3(512), lds, st, a, wk);
  CWF_LAUNCH_CHECK();
  return 0;
}
template <bool X3, int NT>
int launch_ws(const ConvArgsB& a, WsWork wk, int grid, hipStream_t st) {
  static const char* diag = getenv("CWF_WS_DIAG");          // ablation build for profiling; the product kernel has no such branches
  if (diag) { wk.diag = atoi(diag); return launch_ws_impl<X3, NT, true>(a, wk, grid, st); }
  wk.diag = 0;
  return launch_ws_impl<X3, NT, false>(a, wk, grid, st);
}
}  // namespace

// Returns 1 and launches if the layer is one this kernel takes (3x3x3 stride 1, Cin a multiple of 16 and >= 32, Cout a multiple of 16,
// extents multiples of the 4x4x16 tile, no per-channel output scale, weights of one output group fit LDS); 0 = not eligible (the caller
// falls through to the tap-table kernel); < 0 / > 1 never (errors are returned through *rc).
int cwf_try_conv_ws(int op, int x3, ConvArgsB& a, hipStream_t st, int* rc) {
  static const bool off = getenv("CWF_NO_CONV_WS") != nullptr;
  const ConvGeom& g = a.g;
  if (off || op != CWF_CONV3_S1 || a.groups || a.out_scale) return 0;
  if (g.Cin < 32 || (g.Cin & 15) || (g.Cout & 15) || g.Cout < 16) return 0;
  if ((g.Do & 3) || (g.Ho & 3) || (g.Wo & 15)) return 0;
  if (g.x_ldc < g.Cin || (g.x_ldc & 3)) return 0;
  const int nch = g.Cin / 16, ntiles = g.Cout / 16;
  const int budget = 160 * 1024 - WS_NVOX * 32 * (x3 ? 2 : 1) - 2048;
  int NT = 0;
  if ((ntiles & 1) == 0 && nch * 14 * 2 * 1024 * (x3 ? 2 : 1) <= budget) NT = 2;
  else if (nch * 14 * 1 * 1024 * (x3 ? 2 : 1) <= budget) NT = 1;
  if (!NT) return 0;
  WsWork wk;
  wk.ngroups = ntiles / NT;
  if (wk.ngroups > 32) return 0;
  // the geometry of a 4x4x16 tile (the caller built it for its own tile choice)
  int e = cwf_build_geom(a.g, op, g.N, g.Di, g.Hi, g.Wi, g.Cin, g.x_ldc, g.Do, g.Ho, g.Wo, g.Cout, g.y_ldc, 16);
  if (e) { *rc = e; return 1; }
  wk.tiles = a.g.N * a.g.tiles_d * a.g.tiles_h * a.g.tiles_w;
  int per = (32 / wk.ngroups) * wk.ngroups;               // workgroups per XCD, a multiple of the group count
  wk.slots = 8 * (per / wk.ngroups);
  wk.xcd_perm = 1;
  if (wk.slots > wk.tiles) { wk.slots = wk.tiles; wk.xcd_perm = 0; }
  const int grid = wk.slots * wk.ngroups;
  if (x3) *rc = NT == 2 ? launch_ws<true, 2>(a, wk, grid, st) : launch_ws<true, 1>(a, wk, grid, st);
  else *rc = NT == 2 ? launch_ws<false, 2>(a, wk, grid, st) : launch_ws<false, 1>(a, wk, grid, st);
  return 1;
}
