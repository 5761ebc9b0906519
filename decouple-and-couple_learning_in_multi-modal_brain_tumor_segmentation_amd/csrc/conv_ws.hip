// K1, weight-stationary form -- the 3x3x3 stride-1 convolutions of the 32 / 64 / 128-channel levels (EnBlock / DeBlock / EnBlock2 at
// 64^3, 32^3, 16^3: Unet_skipconnection.py:36-57, cls_wise_former.py:691-754) and their data gradients.
//
// Why a second kernel (profiles/round3_generic_conv_pmc.txt, rocprofv3 --pmc on conv_bf16_kernel<4,2,4>): the tap-table kernel gives a
// workgroup ONE 256-voxel tile; per 16-channel chunk it stages the halo (global -> registers -> LDS, nothing else running in that
// workgroup meanwhile) and then streams the packed weights of every tap pair from L2 with one step of lookahead.  Its waves sit in
// s_waitcnt / s_barrier 31 % (split-bf16 forward) to 47 % (single-bf16 data gradient) of their cycles, the matrix pipe is 46 % / 18 %
// busy, and the weight stream (110 KB per tile in the split form) is as much L2 -> CU traffic as the activations.
//
// Here a persistent 8-wave workgroup (one per CU) owns a 32-channel output group and a contiguous range of 4x4x16 output tiles, and
// the weights it needs sit in LDS: one 16-input-channel chunk at a time (all 27 taps, hi and lo images: 57 KB), loaded ONCE per
// round of 2T tiles -- the loop runs chunk-outer over the round's tiles, whose accumulators persist in registers.
// The two 4-wave halves of the workgroup ("groups", one wave per SIMD each) alternate roles step by step:
//     step t:   one group runs the MFMA phase of a (tile, chunk) item on ITS LDS image of the halo      (matrix pipe)
//               the other converts + writes the halo of its next item (fused InstanceNorm + activation prologue, bf16 hi / lo split)
//               into its own image and issues the global loads of the item after that                   (VALU / LDS / memory)
// so each SIMD always has one wave issuing MFMAs and one wave doing everything else; one workgroup barrier per step hands over.
// A wave of a group owns M-tiles 4w .. 4w+3 (plane w of the tile) and both output-channel tiles: A and B fragments come from LDS by
// ds_read_b128 at immediate offsets.  Epilogue (bias, residual, InstanceNorm statistics or norm-backward sums, stores) as in the
// tap-table kernel's interior fast path; statistics stay in registers over all tiles of a sample.
#include "conv_args.h"
#include <cstdlib>
#include <type_traits>

#define WS_ID 6
#define WS_IH 6
#define WS_IW 18
#define WS_NVOX (WS_ID * WS_IH * WS_IW)      // 648 halo voxels of a 4x4x16 tile
#define WS_SLOTS 11                          // staging slots per thread of a group: 648 voxels x 4 channel quads / 256 threads
#define WS_NT 2                              // output-channel tiles (of 16) per workgroup

// diag: ablation bits of the DIAG build (CWF_WS_DIAG): 1 no loads, 2 no MFMA phase, 4 no epilogue, 8 no convert
struct WsWork { int ngroups, slots, tiles, xcd_perm, diag; };

__host__ __device__ constexpr int ws_tap_bytes(int t) { return (((t / 9) * WS_IH + (t / 3) % 3) * WS_IW + t % 3) * 32; }

template <bool X3, bool DIAG>
__global__ __launch_bounds__(512) void convws_kernel(const ConvArgsB a, const WsWork wk) {
  constexpr int NT = WS_NT;
  constexpr int T = 1;                                     // tiles per group and round (two per group do not fit the register file beside the prefetch)
  extern __shared__ float4 lds4[];
  const ConvGeom& g = a.g;
  const int nch = g.nchunks;
  char* lds = reinterpret_cast<char*>(lds4);
  constexpr int B_IMG = 14 * NT * 1024;                    // one chunk of one image (hi or lo)
  constexpr int B_ALL = B_IMG * (X3 ? 2 : 1);
  constexpr int A_IMG = WS_NVOX * 32;
  constexpr int A_GRP = A_IMG * (X3 ? 2 : 1);
  constexpr int NB = (B_ALL + 4095) / 4096;                // uint4 per thread of GROUP 0 for one chunk's weights
  char* Bh = lds;
  char* Bl = lds + B_IMG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wl = wave & 3, tg = tid & 255;      // group (0 / 1), wave in group, thread in group
  char* Ah = lds + B_ALL + grp * A_GRP;
  char* Al = Ah + A_IMG;
  float* red = reinterpret_cast<float*>(lds + B_ALL + 2 * A_GRP);
  const int r = lane & 15, kq = lane >> 4;
  const bool second = (kq >> 1) != 0;

  // ---- work: output-channel group and a contiguous range of spatial tiles (contiguous per XCD: neighbouring tiles share halo rows in L2)
  int cgrp, slot;
  {
    const int b = blockIdx.x;
    if (wk.xcd_perm) {
      const int per = gridDim.x >> 3, j = b >> 3;
      cgrp = j % wk.ngroups;
      slot = (b & 7) * (per / wk.ngroups) + j / wk.ngroups;
    } else {
      cgrp = b % wk.ngroups;
      slot = b / wk.ngroups;
    }
  }
  const int t_begin = (int)(((int64_t)slot * wk.tiles) / wk.slots), t_end = (int)(((int64_t)(slot + 1) * wk.tiles) / wk.slots);
  const int nt0 = cgrp * NT;
  const int n_rounds = (t_end - t_begin + 2 * T - 1) / (2 * T);
  const int n_blocks = n_rounds * nch;                     // block = (round, chunk): T items per group
  const int tiles_per_n = g.tiles_d * g.tiles_h * g.tiles_w;
  // the tile of item (round, k) of this group, or -1 (the range does not fill its last round)
  auto item_tile = [&](int round, int k) { const int t = t_begin + round * 2 * T + 2 * k + grp; return t < t_end ? t : -1; };

  // ---- packed weights of one chunk: global -> registers -> LDS, by GROUP 0 alone: it issues the loads at the head of the step in
  // which it converts (nothing else of it is live then: the other group is in its MFMA phase with its registers full of fragments)
  // and writes them at the block's switch point, one step later.
  // (asm loads like the halo prefetch below: as tracked loads the 14 destination vectors were demoted to scratch memory)
  u32x4 bpf[NB];
  const uint4* wsrc = a.wpk + (int64_t)g.cls_wbase16[0] * 128;
  auto b_issue = [&](int chunk) {
    int tgo = tg;
    asm volatile("" : "+v"(tgo));                          // (opaque: addresses are formed one by one, not hoisted as 14 register pairs)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int e = tgo + 256 * i;                               // uint4 index in [image][step][j][lane]
      if (e >= B_ALL / 16) e = 0;                          // (clamped: the load itself is unconditional)
      const int img = e / (14 * NT * 64), e2 = e % (14 * NT * 64);
      const int ln = e2 & 63, blk = e2 >> 6;
      const int j = blk % NT, s = blk / NT;
      const uint4* p = wsrc + ((int64_t)(chunk * 14 + s) * g.ntiles + nt0 + j) * 128 + ln * 2 + img;
      bpf[i] = *reinterpret_cast<const u32x4*>(p);
    }
  };
  auto b_write = [&]() {
    int wb = tg * 16;
    asm volatile("" : "+v"(wb));
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (256 * (i + 1) <= B_ALL / 16 || tg + 256 * i < B_ALL / 16) *reinterpret_cast<u32x4*>(Bh + wb + i * 4096) = bpf[i];     // (the lo image follows the hi image)
  };

  // ---- staging slots of this thread: halo voxel v = (tg >> 2) + 64 i, channel quad q = tg & 3
  const int q = tg & 3;
  const int vox0 = tg >> 2;                                // slot i holds voxel vox0 + 64 i (valid below WS_NVOX: slot 10 only for vox0 < 8)
  const int iw_0 = vox0 % WS_IW, row_0 = vox0 / WS_IW;        // (row = idd * WS_IH + ih; 64 voxels = 3 rows + 10)
  const bool plain = a.in_scale == nullptr && a.in_slope == 1.f;

  // The prefetch loads are written in inline asm: hipcc waits for every load IT tracks before it reuses a register it believes
  // pending, which put an s_waitcnt vmcnt(0) right behind the issue block (the prefetch waited for on the spot); loads inside asm
  // are invisible to its counters.  Their completion is counted by hand: ONE s_waitcnt vmcnt(0) in front of the convert (pf_wait,
  // naming every destination register).  hipcc's own counted waits stay sufficient beside them: vector-memory operations return
  // in order, so "all but my k youngest" can only wait for MORE than it assumes.
  f32x4 pf[WS_SLOTS];
  f32x4 pf_sc, pf_sh;                                      // the item's prologue parameters (in_scale / in_shift of its sample and chunk)
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
    int iw = iw_0, row = row_0;
    asm volatile("" : "+v"(iw), "+v"(row));                // (opaque: per-slot coordinates are recomputed here, not kept in 30+ hoisted registers)
#pragma unroll
    for (int i = 0; i < WS_SLOTS; ++i) {
      const int idd = (row * 43) >> 8, ih = row - idd * WS_IH;                 // row / 6 for row < 48
      const int gd = id0 + idd, gh = ih0 + ih, gw = iw0 + iw;
      const bool ok = (row < WS_ID * WS_IH) & ((unsigned)gd < (unsigned)g.Di) & ((unsigned)gh < (unsigned)g.Hi) & ((unsigned)gw < (unsigned)g.Wi);
      iw += 10; row += 3;
      if (iw >= WS_IW) { iw -= WS_IW; ++row; }
      const unsigned lin = (unsigned)((gd * g.Hi + gh) * g.Wi + gw) * ldc4 + (unsigned)q * 16u;
      const unsigned boff = ok ? lin : (unsigned)q * 16u;          // (clamped: the load itself is unconditional)
      pf[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xb) + boff);
      pf_inb |= ok ? (1u << i) : 0u;
    }
    // (always two loads, so that the hand-counted waits see a fixed number per item: without a prologue they read the tensor base)
    const float* scp = a.in_scale ? a.in_scale + (int64_t)n * g.Cin + chunk * 16 : a.x;
    const float* shp = a.in_scale ? a.in_shift + (int64_t)n * g.Cin + chunk * 16 : a.x;
    const unsigned qo = (unsigned)q * 16u;
    pf_sc = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(scp) + qo);
    pf_sh = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(shp) + qo);
  };
  auto convert_write = [&](int tile, int chunk, const f32x4& sc, const f32x4& sh) {
    const float slope = a.in_slope;
    int vbase = (tg >> 2) * 32 + q * 8;                    // byte offset of slot 0 in the image; slot i adds an immediate
    asm volatile("" : "+v"(vbase));
#pragma unroll
    for (int i = 0; i < WS_SLOTS; ++i) {
      if (i == WS_SLOTS - 1 && vox0 + 64 * i >= WS_NVOX) continue;
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
      *reinterpret_cast<uint2*>(Ah + vbase + i * (64 * 32)) = h;
      if (X3) *reinterpret_cast<uint2*>(Al + vbase + i * (64 * 32)) = l;
    }
  };

  // ---- MFMA operands: lane base addresses (bytes); everything else is an immediate
  const int a_lane = ((wl * WS_IH) * WS_IW + r) * 32 + (kq & 1) * 16;
  const int b_lane = lane * 16;

  f32x4 acc[T][4][NT];
#pragma unroll
  for (int k = 0; k < T; ++k)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[k][m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float s1[NT], s2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  int stats_n = -1;

  // a sample boundary inside a workgroup's tile range (at most one workgroup per output group and boundary): this wave's partial
  // sums go out directly; the common flush at the end of the kernel goes through LDS (one atomic per channel and workgroup)
  auto flush_stats_wave = [&]() {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float u1 = s1[j], u2 = s2[j];
      u1 += __shfl_xor(u1, 16, 64); u1 += __shfl_xor(u1, 32, 64);
      u2 += __shfl_xor(u2, 16, 64); u2 += __shfl_xor(u2, 32, 64);
      if (kq == 0) {
        const int co = (nt0 + j) * 16 + r;
        atomic_add_f64(a.stats + ((int64_t)stats_n * g.Cout + co) * 2 + 0, (double)u1);
        atomic_add_f64(a.stats + ((int64_t)stats_n * g.Cout + co) * 2 + 1, (double)u2);
      }
      s1[j] = 0.f; s2[j] = 0.f;
    }
  };

  auto epilogue = [&](int tile, auto KK) {
    constexpr int k = decltype(KK)::value;
    const int n = tile / tiles_per_n;
    int bx = tile - n * tiles_per_n;
    const int tw = bx % g.tiles_w; bx /= g.tiles_w;
    const int th = bx % g.tiles_h, td = bx / g.tiles_h;
    if (a.stats && n != stats_n) {
      if (stats_n >= 0) flush_stats_wave();
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
      for (int m = 0; m < 4; ++m) {
        const int od = td * 4 + wl, oh = th * 4 + m;
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
            float v = acc[k][m][j][i] + bvj[j];
            if (HAS_RES) v += rv[j][i];
            *reinterpret_cast<float*>(yb + yo[i] + j * 64) = v;
            if (HAS_NB) {
              const float h = fmaf(xv[j][i], nsc[j], nsh[j]);
              const float gn = v * (h > 0.f ? 1.f : a.nb_slope);
              s1[j] += gn; s2[j] = fmaf(gn, h, s2[j]);
            } else if (HAS_STATS) { s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
            acc[k][m][j][i] = 0.f;
          }
      }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    if (a.nb_x) { if (a.residual) epi(T_{}, T_{}, T_{}); else epi(F_{}, T_{}, T_{}); }
    else if (a.residual) { if (a.stats) epi(T_{}, T_{}, F_{}); else epi(T_{}, F_{}, F_{}); }
    else            { if (a.stats) epi(F_{}, T_{}, F_{}); else epi(F_{}, F_{}, F_{}); }
  };

  // MFMA phase of item (block, k) of this group: 14 tap-pair steps on the group's A image and the resident chunk of weights
  auto mfma_item = [&](int block, auto KK) -> bool {
    constexpr int k = decltype(KK)::value;
    const int round = block / nch, chunk = block - round * nch;
    const int tile = item_tile(round, k);
    if (tile < 0) return false;
    const char* bh = Bh + b_lane;
    const char* bl = Bl + b_lane;
    const char* ah0 = Ah + a_lane;
    const char* al0 = Al + a_lane;
    if (!(DIAG && (wk.diag & 2))) {
      // Fragment reads are pipelined by hand; the scheduling barrier per step keeps hipcc from hoisting further steps' reads on top
      // (three steps of split-bf16 fragments in flight spilled registers).  The hi fragments of step s + 1 are requested before the
      // MFMAs of step s (two named sets); the lo fragments of step s are requested at the head of step s and first used by its
      // ninth MFMA (one set): per accumulator the order stays hi.hi, hi.lo, lo.hi.
      uint4 fa[2][4], fb[2][NT], fl[4], fbl[NT];
      auto rd_hi = [&](auto S, auto P) {
        constexpr int s = decltype(S)::value, p = decltype(P)::value;
        constexpr int c0 = ws_tap_bytes(2 * s), c1 = ws_tap_bytes(2 * s + 1 < 27 ? 2 * s + 1 : 2 * s);   // padded tap: zero weights, valid address
        const int to = second ? c1 : c0;
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[p][m] = *reinterpret_cast<const uint4*>(ah0 + to + m * (WS_IW * 32));
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[p][j] = *reinterpret_cast<const uint4*>(bh + (s * NT + j) * 1024);
      };
      auto rd_lo = [&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr int c0 = ws_tap_bytes(2 * s), c1 = ws_tap_bytes(2 * s + 1 < 27 ? 2 * s + 1 : 2 * s);
        const int to = second ? c1 : c0;
#pragma unroll
        for (int m = 0; m < 4; ++m) fl[m] = *reinterpret_cast<const uint4*>(al0 + to + m * (WS_IW * 32));
#pragma unroll
        for (int j = 0; j < NT; ++j) fbl[j] = *reinterpret_cast<const uint4*>(bl + (s * NT + j) * 1024);
      };
      auto mm = [&](auto P) {
        constexpr int p = decltype(P)::value;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[k][m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[p][m]), __builtin_bit_cast(bf16x8, fb[p][j]), acc[k][m][j], 0, 0, 0);
        if (X3) {
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[k][m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[p][m]), __builtin_bit_cast(bf16x8, fbl[j]), acc[k][m][j], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[k][m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fl[m]), __builtin_bit_cast(bf16x8, fb[p][j]), acc[k][m][j], 0, 0, 0);
        }
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
#define WS_IC(v) std::integral_constant<int, (v)>{}
#define WS_STEP2(S0)                                                                                           \
      if (X3) rd_lo(WS_IC(S0)); rd_hi(WS_IC((S0) + 1), I1{}); mm(I0{}); __builtin_amdgcn_sched_barrier(0);       \
      if (X3) rd_lo(WS_IC((S0) + 1)); if ((S0) + 2 < 14) rd_hi(WS_IC((S0) + 2 < 14 ? (S0) + 2 : 13), I0{});      \
      mm(I1{}); __builtin_amdgcn_sched_barrier(0);
      rd_hi(I0{}, I0{});
      WS_STEP2(0) WS_STEP2(2) WS_STEP2(4) WS_STEP2(6) WS_STEP2(8) WS_STEP2(10) WS_STEP2(12)
#undef WS_STEP2
#undef WS_IC
    }
    if (chunk == nch - 1 && !(DIAG && (wk.diag & 4))) { epilogue(tile, KK); return true; }
    return false;
  };
  // convert step of item (block, k): the halo prefetched for it -> the group's A image; then the loads of the group's next item
  // ---- schedule.  Block b = (round, chunk): one item per group.  Group 0 converts its item in step 0 of the block and runs its MFMA
  // phase in step 1; group 1 converts in step 1 and runs its MFMA phase in step 0 of the NEXT block; the block's weights replace
  // the previous block's between the two steps (written by group 0 between two barriers, while nobody reads them).
  //
  // Every loop iteration issues the loads of the group's NEXT item, runs the MFMA phase of the current one, and only then waits for
  // the loads and converts: request, wait and use of the asm loads sit in ONE iteration, so their destination registers are never
  // live across a loop back-edge.  (With the request at the end of one iteration and the wait at the head of the next, hipcc
  // inserted register copies of the in-flight destinations at the loop boundary -- reads of data that had not landed: wrong tiles
  // whenever the input was not already in cache.)  tools/audit_asm_loads.py checks the emitted code for such reads.
  // The two groups run separate loops with equal barrier counts (3 per block + 1): group 0's weight registers then have no live
  // range through group 1's MFMA code.
  using K0 = std::integral_constant<int, 0>;
  const bool no_loads = DIAG && (wk.diag & 1), no_conv = DIAG && (wk.diag & 8);
  auto do_issue = [&](int blk) {                           // the halo + prologue parameters of this group's item of block blk
    const int round = blk / nch, tile = item_tile(round, 0);
    if (tile >= 0 && !no_loads) issue_loads(tile, blk - round * nch);
  };
  auto do_convert = [&](int blk, int younger) {           // younger < 0: the loads have been waited for already
    const int round = blk / nch, tile = item_tile(round, 0);
    if (tile < 0) return;
    (void)younger;
    // (a use of the two parameter vectors on EVERY path: left pending in hipcc's model on the path that never reads them -- no
    // prologue -- it would protect their registers later with a full s_waitcnt vmcnt(0))
    asm volatile("" : "+v"(pf_sc), "+v"(pf_sh));
    if (!no_conv) convert_write(tile, blk - round * nch, pf_sc, pf_sh);
  };
  if (n_blocks > 0) {
    // the first item of either group: requested AND waited for before the code of the two groups parts (hipcc hoists the common
    // request above the branch and copies the destination registers into each side's own: after the wait that is harmless; the
    // latency of this one fetch is exposed once per launch)
    do_issue(0);
    if (grp == 0) {
      b_issue(0);
      do_convert(0, -1);
      __syncthreads();
      b_write();
      __syncthreads();
      for (int blk = 0; blk < n_blocks; ++blk) {
        const bool more = blk + 1 < n_blocks;
        if (more) do_issue(blk + 1);
        const bool epi = mfma_item(blk, K0{});             // step 1 of block blk
        __syncthreads();
        if (more) {                                        // step 0 of block blk + 1
          b_issue((blk + 1) % nch);
          do_convert(blk + 1, NB + (epi ? 32 : 0));
        }
        __syncthreads();                                   // group 1 has finished with the weights of block blk
        if (more) {
          b_write();
          __syncthreads();
        }
      }
    } else {
      __syncthreads();
      __syncthreads();
      do_convert(0, -1);                                   // step 1 of block 0
      for (int blk = 0; blk < n_blocks; ++blk) {
        const bool more = blk + 1 < n_blocks;
        __syncthreads();
        if (more) do_issue(blk + 1);
        const bool epi = mfma_item(blk, K0{});             // step 0 of block blk + 1
        __syncthreads();
        if (more) {
          __syncthreads();
          do_convert(blk + 1, epi ? 32 : 0);               // step 1 of block blk + 1
        }
      }
    }
  }

  // ---- statistics of each group's last sample: one atomic per channel, group and workgroup
  if (a.stats) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float u1 = s1[j], u2 = s2[j];
      u1 += __shfl_xor(u1, 16, 64); u1 += __shfl_xor(u1, 32, 64);
      u2 += __shfl_xor(u2, 16, 64); u2 += __shfl_xor(u2, 32, 64);
      if (kq == 0) {
        red[((wave * NT + j) * 16 + r) * 2 + 0] = u1;
        red[((wave * NT + j) * 16 + r) * 2 + 1] = u2;
      }
    }
    int* red_n = reinterpret_cast<int*>(red + 8 * NT * 16 * 2);
    if (lane == 0) red_n[wave] = stats_n;
    __syncthreads();
    if (tid < 2 * NT * 16 * 2) {
      const int gsel = tid / (NT * 16 * 2), t2 = tid % (NT * 16 * 2);
      const int which = t2 & 1, rr = (t2 >> 1) & 15, jj = t2 >> 5;
      const int sn = red_n[gsel * 4];
      if (sn >= 0) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += (double)red[(((gsel * 4 + w) * NT + jj) * 16 + rr) * 2 + which];
        const int co = (nt0 + jj) * 16 + rr;
        atomic_add_f64(a.stats + ((int64_t)sn * g.Cout + co) * 2 + which, s);
      }
    }
  }
}

namespace {
template <bool X3, bool DIAG>
int launch_ws_impl(const ConvArgsB& a, const WsWork& wk, int grid, hipStream_t st) {
  const size_t lds = (size_t)14 * WS_NT * 1024 * (X3 ? 2 : 1) + (size_t)2 * WS_NVOX * 32 * (X3 ? 2 : 1) + 8 * WS_NT * 16 * 2 * sizeof(float) + 64;
  if (lds > 160 * 1024) return CWF_E_TOOLARGE;
  CWF_MAX_LDS_ONCE((&convws_kernel<X3, DIAG>));
  hipLaunchKernelGGL((convws_kernel<X3, DIAG>), dim3(grid), dim3(512), lds, st, a, wk);
  CWF_LAUNCH_CHECK();
  return 0;
}
template <bool X3>
int launch_ws(const ConvArgsB& a, WsWork wk, int grid, hipStream_t st) {
  static const char* diag = getenv("CWF_WS_DIAG");          // ablation build for profiling; the product kernel has no such branches
  if (diag) { wk.diag = atoi(diag); return launch_ws_impl<X3, true>(a, wk, grid, st); }
  wk.diag = 0;
  return launch_ws_impl<X3, false>(a, wk, grid, st);
}
}  // namespace

static int g_ws_min_units = getenv("CWF_WS_MIN_UNITS") ? atoi(getenv("CWF_WS_MIN_UNITS")) : 256;
// tests / tools: lower the size threshold so that small shapes reach this kernel too (returns the previous value)
extern "C" int cwf_debug_ws_min_units(int v) { const int old = g_ws_min_units; g_ws_min_units = v; return old; }
// The split-bf16 (forward) instantiation is correct and tested but NOT faster than the tap-table kernel (32 ch @ 64^3: 105 us against
// 101; 64 ch @ 32^3: 51 against 50): the product sends only single-bf16 launches (the data gradients: 62 against 74 us, 31 against
// 34) here.  CWF_WS_X3=1 / cwf_debug_ws_x3(1) sends the split form here too (tests, experiments).
static int g_ws_x3 = getenv("CWF_WS_X3") ? atoi(getenv("CWF_WS_X3")) : 0;
extern "C" int cwf_debug_ws_x3(int v) { const int old = g_ws_x3; g_ws_x3 = v; return old; }

// Returns 1 and launches if the layer is one this kernel takes (3x3x3 stride 1, Cin a multiple of 16 and >= 32, Cout a multiple of 32,
// extents multiples of the 4x4x16 tile, no per-channel output scale, enough tiles to occupy the chip); 0 = not eligible (the caller
// falls through to the tap-table kernel).  The launch status is returned through *rc.
int cwf_try_conv_ws(int op, int x3, ConvArgsB& a, hipStream_t st, int* rc) {
  static const bool off = getenv("CWF_NO_CONV_WS") != nullptr;
  const ConvGeom& g = a.g;
  if (off || op != CWF_CONV3_S1 || a.groups || a.out_scale || (x3 && !g_ws_x3)) return 0;
  if (g.Cin < 32 || (g.Cin & 15) || (g.Cout & 31)) return 0;
  if ((g.Do & 3) || (g.Ho & 3) || (g.Wo & 15)) return 0;
  if (g.x_ldc < g.Cin || (g.x_ldc & 3)) return 0;
  WsWork wk;
  wk.ngroups = g.Cout / 32;
  if (wk.ngroups > 32) return 0;
  const int64_t tiles = (int64_t)g.N * (g.Do / 4) * (g.Ho / 4) * (g.Wo / 16);
  // small layers (fewer (tile, output group) units than CUs; measured at 128 ch @ 16^3: 68 us here against 48): the tap-table kernel's
  // many small workgroups fill the chip better
  if (tiles * wk.ngroups < g_ws_min_units) return 0;
  // the geometry of a 4x4x16 tile (the caller built it for its own tile choice)
  int e = cwf_build_geom(a.g, op, g.N, g.Di, g.Hi, g.Wi, g.Cin, g.x_ldc, g.Do, g.Ho, g.Wo, g.Cout, g.y_ldc, 16);
  if (e) { *rc = e; return 1; }
  wk.tiles = (int)tiles;
  int per = (32 / wk.ngroups) * wk.ngroups;               // workgroups per XCD, a multiple of the group count
  wk.slots = 8 * (per / wk.ngroups);
  wk.xcd_perm = 1;
  if (wk.slots * 2 > wk.tiles) { wk.slots = (wk.tiles + 1) / 2; wk.xcd_perm = 0; }      // at least two tiles per workgroup (one per group)
  const int grid = wk.slots * wk.ngroups;
  *rc = x3 ? launch_ws<true>(a, wk, grid, st) : launch_ws<false>(a, wk, grid, st);
  return 1;
}
