// K1 (split-bf16 form) -- the conv family of conv_mfma.hip on v_mfma_f32_16x16x32_bf16.
//
// Activations and weights stay fp32 in HBM; only the MFMA operands are bf16:
//   X3 = true  ("bf16x3"): x = xh + xl, w = wh + wl (bf16 each, lo = bf16(v - hi));  x.w ~= xh.wh + xh.wl + xl.wh with fp32
//                accumulation: ~2^-16 relative per product (the dropped xl.wl term), 3 MFMAs = 5.3x the fp32 MFMA rate.
//   X3 = false ("bf16")  : x ~= xh, w ~= wh: one MFMA, 16x the fp32 MFMA rate, 2^-9 relative per product.
// The input halo tile is converted while it is staged (after the fused InstanceNorm + activation prologue), into two LDS
// images [voxel][16 ch] bf16 (hi, lo): 32-byte voxel rows make the 16-byte A-fragment reads bank-conflict free.
// K ordering: one MFMA (K = 32) = two taps x 16 channels: lane group kq reads channels 8(kq&1).. of tap 2s + (kq>>1).
// Packed weights: [class][ci_chunk16][tap pair][co_tile16][lane64][hi 8 | lo 8] bf16 (cwf_gather_split_bf16).
// Geometry, tiling, epilogue (bias / residual / out_scale / InstanceNorm statistics) are those of conv_mfma.hip.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "conv_args.h"

// Staging of one 16-channel chunk as bf16 hi (and lo) images [voxel][16].  A workgroup owns one spatial tile for all its
// channel chunks, so the halo geometry (which global voxel each staging slot of this thread reads, and whether it is in
// bounds) is computed ONCE (StageSlots) instead of per chunk: the per-chunk loop is load + convert + LDS store only.
#define CB_MAXS 13                                       // staging slots per thread: ceil(nvox_in / 64) <= 13 (5x5x33 halo)
struct StageSlots { int goff[CB_MAXS]; unsigned inb; int nslots; };

__device__ __forceinline__ void stage_slots_init(StageSlots& ss, const ConvGeom& g, int id0, int ih0, int iw0, int tid) {
  const int nvox_in = g.ID * g.IH * g.IW;
  ss.nslots = (nvox_in + 63) >> 6; ss.inb = 0u;
  // (idd, ih, iw) of slot i's voxel v = (tid >> 2) + 64 i, INCREMENTALLY: two integer divisions for slot 0, then adds and carries
  // (64 = dq * IW + dr).  Thirteen slots x two runtime divisions per thread was ~1000 VALU instructions per workgroup -- a third of
  // the stride-2 forward launches, whose workgroups own a single 64-voxel tile (16 -> 32 @128^3: 209 us).
  const int dq = 64 / g.IW, dr = 64 - dq * g.IW;          // wave-uniform
  int iw = (tid >> 2) % g.IW, t2 = (tid >> 2) / g.IW;
  int ih = t2 % g.IH, idd = t2 / g.IH;
#pragma unroll
  for (int i = 0; i < CB_MAXS; ++i) {
    const int v = (tid >> 2) + 64 * i;
    const int gd = id0 + idd, gh = ih0 + ih, gw = iw0 + iw;
    const bool ok = v < nvox_in && gd >= 0 && gd < g.Di && gh >= 0 && gh < g.Hi && gw >= 0 && gw < g.Wi;
    ss.goff[i] = ok ? ((gd * g.Hi + gh) * g.Wi + gw) * g.x_ldc : 0;
    ss.inb |= ok ? (1u << i) : 0u;
    iw += dr; int adv = dq;
    if (iw >= g.IW) { iw -= g.IW; ++adv; }
    ih += adv;
    while (ih >= g.IH) { ih -= g.IH; ++idd; }              // (adv <= 64 / IW + 1: a few rows at most)
  }
}

template <bool X3>
__device__ __forceinline__ void stage_tile_bf16(unsigned short* xh, unsigned short* xl, const ConvGeom& g, const float* x,
                                                const float* in_scale, const float* in_shift, float slope,
                                                int n, int chunk, const StageSlots& ss, int tid) {
  const int q = tid & 3;
  const int c = chunk * 16 + q * 4;
  const bool cval = c < g.Cin;
  const bool has_norm = in_scale != nullptr;
  const bool plain = !has_norm && slope == 1.f;          // data gradients: no prologue
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has_norm && cval) {
    sc = *reinterpret_cast<const float4*>(in_scale + (int64_t)n * g.Cin + c);
    sh = *reinterpret_cast<const float4*>(in_shift + (int64_t)n * g.Cin + c);
  }
  const int nvox_in = g.ID * g.IH * g.IW;
  const float* xb = x + (int64_t)n * g.Di * g.Hi * g.Wi * g.x_ldc + c;
  const unsigned inb = cval ? ss.inb : 0u;
  unsigned short* dh = xh + (tid >> 2) * 16 + q * 4;
  unsigned short* dl = xl + (tid >> 2) * 16 + q * 4;
  float4 val[CB_MAXS];
#pragma unroll
  for (int i = 0; i < CB_MAXS; ++i)                      // all loads first (independent), then convert
    if (i < ss.nslots) val[i] = ((inb >> i) & 1u) ? *reinterpret_cast<const float4*>(xb + ss.goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < CB_MAXS; ++i) {
    if (i >= ss.nslots) continue;                        // workgroup-uniform
    if ((tid >> 2) + 64 * i >= nvox_in) continue;
    float v0 = val[i].x, v1 = val[i].y, v2 = val[i].z, v3 = val[i].w;
    if (!plain) {
      v0 = act01(fmaf(v0, sc.x, sh.x), slope); v1 = act01(fmaf(v1, sc.y, sh.y), slope);
      v2 = act01(fmaf(v2, sc.z, sh.z), slope); v3 = act01(fmaf(v3, sc.w, sh.w), slope);
    }
    uint2 h, l;
    if (X3) { split_bf16(v0, v1, h.x, l.x); split_bf16(v2, v3, h.y, l.y); }
    else { h.x = pack_bf16(v0, v1); h.y = pack_bf16(v2, v3); l = make_uint2(0u, 0u); }
    if (!plain) {                                        // zero padding applies after the activation
      const bool was = (inb >> i) & 1u;
      h.x = was ? h.x : 0u; h.y = was ? h.y : 0u; l.x = was ? l.x : 0u; l.y = was ? l.y : 0u;
    }
    *reinterpret_cast<uint2*>(dh + i * 64 * 16) = h;
    if (X3) *reinterpret_cast<uint2*>(dl + i * 64 * 16) = l;
  }
}

template <int MT, int NT, int WM, bool X3>
__global__ __launch_bounds__(256) void conv_bf16_kernel(const ConvArgsB a) {
  constexpr int WN = 4 / WM;
  extern __shared__ float4 lds4[];
  const ConvGeom& g = a.g;
  const int nvox_in = g.ID * g.IH * g.IW;
  unsigned short* xh = reinterpret_cast<unsigned short*>(lds4);
  unsigned short* xl = xh + nvox_in * 16;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: per-wave quantities below stay in SGPRs
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 15, kq = lane >> 4;

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so give every XCD a
  // CONTIGUOUS range of spatial tiles -- neighbouring tiles share halo rows, which then hit in that XCD's L2.
  int bx = blockIdx.x;
  {
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bx = (bx & 7) * (nb >> 3) + (bx >> 3);
  }
  const int tile_w = bx % g.tiles_w; bx /= g.tiles_w;
  const int tile_h = bx % g.tiles_h;
  const int tile_d = bx / g.tiles_h;
  const int zq = blockIdx.z / g.ncls, cls = blockIdx.z % g.ncls;
  const int grp = a.groups ? zq / g.N : 0, n = a.groups ? zq % g.N : zq;
  const float* a_x = a.x + grp * a.x_goff;               // (workgroup-uniform: group operands)
  float* a_y = a.y + grp * a.y_goff;
  const uint4* a_wpk = a.groups ? a.wpk_g[grp] : a.wpk;
  const float* a_bias = a.groups ? a.bias_g[grp] : a.bias;
  const int Dc = g.cls_dims[cls][0], Hc = g.cls_dims[cls][1], Wc = g.cls_dims[cls][2];
  const int od0 = tile_d * g.TD, oh0 = tile_h * g.TH, ow0 = tile_w * 16;
  if (od0 >= Dc || oh0 >= Hc || ow0 >= Wc) return;
  const int ntaps = g.cls_ntaps[cls];
  const int nsteps = (ntaps + 1) >> 1;
  const int* tapofs = g.tapofs + (g.ncls > 1 ? cls * 8 : 0);
  const int nt0 = (blockIdx.y * WN + wn) * NT;

  int abase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mt = wm * MT + m;
    const int td = mt / g.TH, th = mt % g.TH;
    abase[m] = (((td * g.is) * g.IH + th * g.is) * g.IW + r * g.is) * 16 + (kq & 1) * 8;   // bf16 element offset
  }
  const bool second = (kq >> 1) != 0;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int id0 = od0 * g.is + g.lo[0], ih0 = oh0 * g.is + g.lo[1], iw0 = ow0 * g.is + g.lo[2];

  StageSlots ss;
  stage_slots_init(ss, g, id0, ih0, iw0, tid);
  // Packed weights stream from L2 (~1-2k cycles of latency) while one tap-pair step is only MT*NT*3 MFMAs (96 cycles for the
  // deep 1x2 configuration): a ring of PD steps of B fragments is kept in flight, and the first PD loads of a chunk are
  // issued BEFORE the chunk's activation staging so that their latency hides under it.
  constexpr int PD = MT == 1 ? (NT <= 2 ? 4 : 2) : 1;   // large-MT configurations: one step ahead (deeper rings cost them an occupancy step)
  for (int chunk = 0; chunk < g.nchunks; ++chunk) {
    // packed weights: block = 64 lanes x (hi 16 B | lo 16 B) = 128 uint4
    const uint4* wchunk = a_wpk + ((int64_t)g.cls_wbase16[cls] + (int64_t)chunk * nsteps * g.ntiles) * 128 + lane * 2;
    uint4 bh[PD][NT], bl[PD][NT];
    // unconditional loads (indices clamped into the packed buffer: a partial channel group or a step past the end re-reads
    // valid data that is never used) -- a branch around a global load makes the compiler give up counted s_waitcnt vmcnt(N),
    // and a vmcnt(0) per tap pair costs one full L2 round trip per step (measured: ~1000 cycles per 96-cycle step at 16^3)
    // (MT == 1, the deep small-step configurations; the large-tile configurations keep the guarded form: the straight-line
    // one costs them an occupancy step -- 32 ch @ 64^3: 111 -> 132 us)
    auto load_b = [&](int s_, int slot) {
      if (MT == 1) {
        const int sc_ = min(s_, nsteps - 1);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int jt = min(nt0 + j, g.ntiles - 1);
          const uint4* p = wchunk + ((int64_t)sc_ * g.ntiles + jt) * 128;
          bh[slot][j] = p[0];
          if (X3) bl[slot][j] = p[1];
        }
      } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          bh[slot][j] = make_uint4(0, 0, 0, 0); bl[slot][j] = make_uint4(0, 0, 0, 0);
          if (nt0 + j < g.ntiles && s_ < nsteps) {
            const uint4* p = wchunk + ((int64_t)s_ * g.ntiles + nt0 + j) * 128;
            bh[slot][j] = p[0];
            if (X3) bl[slot][j] = p[1];
          }
        }
      }
    };
#pragma unroll
    for (int d = 0; d < PD; ++d) load_b(d, d);
    if (chunk) __syncthreads();
    stage_tile_bf16<X3>(xh, xl, g, a_x, a.in_scale, a.in_shift, a.in_slope, n, chunk, ss, tid);
    __syncthreads();
    auto step = [&](int s, int d, bool refill) {          // tap pair s with ring slot d
      const int t0 = tapofs[2 * s];
      const int t1 = tapofs[(2 * s + 1 < ntaps) ? 2 * s + 1 : 2 * s];     // padded tap: weights are zero, address stays valid
      const int to = (second ? t1 : t0) * 16;
      uint4 ah[MT], al[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        ah[m] = *reinterpret_cast<const uint4*>(xh + abase[m] + to);
        if (X3) al[m] = *reinterpret_cast<const uint4*>(xl + abase[m] + to);
      }
      uint4 ch[NT], cl[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) { ch[j] = bh[d][j]; cl[j] = bl[d][j]; }
      if (refill) load_b(s + PD, d);                       // refill this ring slot
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[m]), __builtin_bit_cast(bf16x8, ch[j]), acc[m][j], 0, 0, 0);
          if (X3) {
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[m]), __builtin_bit_cast(bf16x8, cl[j]), acc[m][j], 0, 0, 0);
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[m]), __builtin_bit_cast(bf16x8, ch[j]), acc[m][j], 0, 0, 0);
          }
        }
    };
    if (MT == 1 && nsteps == 14) {
      // 27-tap operators (all 3x3x3 forward / stride-1 data-gradient launches): straight-line, refills known at compile time
#pragma unroll
      for (int s = 0; s < 14; ++s) step(s, s % PD, s + PD < 14);
    } else {
#pragma unroll 1
      for (int s0 = 0; s0 < nsteps; s0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d)
          if (s0 + d < nsteps) step(s0 + d, d, true);       // workgroup-uniform
      }
    }
  }

  // ---- epilogue (identical to conv_mfma.hip: C/D layout is shape-determined, row = kq*4+i, col = r)
  const int os = g.os;
  const int of0 = g.cls_ooff[cls][0], of1 = g.cls_ooff[cls][1], of2 = g.cls_ooff[cls][2];
  float s1[NT], s2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  // fast path (workgroup/wave-uniform test): interior tile of a plain (single-class, unit output stride) geometry with all NT
  // channel tiles valid -- every store is SGPR row base + tile-invariant 32-bit lane offset, no per-element bounds or
  // 64-bit address arithmetic (cf. conv16_kernel's epilogue)
  // (output-parity classes -- stride-2 data gradient, ConvTranspose -- take it too: voxel stride os = 2 and the class offset
  // only change the row base and the per-lane voxel step; the scalar path below cost the EnDown data gradient 2x)
  const bool fast = od0 + g.TD <= Dc && oh0 + g.TH <= Hc && ow0 + 16 <= Wc && (nt0 + NT) * 16 <= g.Cout &&
                    !(a.residual && a.out_scale);
  if (fast) {
    float bvj[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = (nt0 + j) * 16 + r;
      bvj[j] = a_bias ? a_bias[co] : 0.f;
      if (a.out_scale) {
        const float osc = a.out_scale[(int64_t)n * g.Cout + co];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][j] = (acc[m][j] + bvj[j]) * osc;
        bvj[j] = 0.f;
      }
    }
    auto epi = [&](auto HR, auto HT, auto HN) {
      constexpr bool HAS_RES = decltype(HR)::value, HAS_STATS = decltype(HT)::value, HAS_NB = decltype(HN)::value;
      unsigned yo[4], ro[4], xo[4];
      float nsc[NT], nsh[NT];
#pragma unroll
      for (int i = 0; i < 4; ++i) {                      // opaque copies: keep the zero-extension in this block (saddr form)
        yo[i] = (unsigned)((kq * 4 + i) * os * g.y_ldc + nt0 * 16 + r) * 4u; asm volatile("" : "+v"(yo[i]));
        if (HAS_RES) { ro[i] = (unsigned)((kq * 4 + i) * os * a.r_ldc + nt0 * 16 + r) * 4u; asm volatile("" : "+v"(ro[i])); }
        if (HAS_NB) { xo[i] = (unsigned)((kq * 4 + i) * os * a.nb_ldc + nt0 * 16 + r) * 4u; asm volatile("" : "+v"(xo[i])); }
      }
      if (HAS_NB) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          nsc[j] = a.nb_scale[(int64_t)n * g.Cout + (nt0 + j) * 16 + r];
          nsh[j] = a.nb_shift[(int64_t)n * g.Cout + (nt0 + j) * 16 + r];
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int mt = wm * MT + m;
        const int od = od0 + mt / g.TH, oh = oh0 + mt % g.TH;
        const int64_t vox0 = (((int64_t)n * g.Do + od * os + of0) * g.Ho + oh * os + of1) * g.Wo + ow0 * os + of2;
        char* yb = reinterpret_cast<char*>(a_y + vox0 * g.y_ldc);
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
          }
      }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    if (a.nb_x) { if (a.residual) epi(T_{}, T_{}, T_{}); else epi(F_{}, T_{}, T_{}); }
    else if (a.residual) { if (a.stats) epi(T_{}, T_{}, F_{}); else epi(T_{}, F_{}, F_{}); }
    else            { if (a.stats) epi(F_{}, T_{}, F_{}); else epi(F_{}, F_{}, F_{}); }
  } else {
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = (nt0 + j) * 16 + r;
    if (co >= g.Cout) continue;
    const float bv = a_bias ? a_bias[co] : 0.f;
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
        a_y[vox * g.y_ldc + co] = v;
        if (a.nb_x) {
          const float h = fmaf(a.nb_x[vox * a.nb_ldc + co], a.nb_scale[(int64_t)n * g.Cout + co], a.nb_shift[(int64_t)n * g.Cout + co]);
          const float gn = v * (h > 0.f ? 1.f : a.nb_slope);
          s1[j] += gn; s2[j] += gn * h;
        } else { s1[j] += v; s2[j] += v * v; }
      }
    }
  }
  }
  if (a.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds4);
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
// conv16: the 3x3x3 stride-1 convolutions with <= 16 input and <= 16 output channels (every conv at full
// resolution: 44 % of the model's FLOPs, the step's dominant kernel).  Same math and packed-weight layout as
// conv_bf16_kernel<4,1,4>, restructured after profiling it (rocprofv3 PMC, profiles/): HBM traffic was 1.1x the
// algorithmic bytes and LDS conflict-free, but the MFMA pipe was 14 % busy -- waves spent their time issuing ~8 VALU
// instructions per MFMA (staging index arithmetic, bf16 splitting, epilogue addressing) and waiting on their own loads.
//   * producer / consumer waves: a workgroup is 8 waves, one per CU.  Waves 0-3 ("MFMA waves", one per SIMD) hold the whole
//     weight set in registers (14 tap pairs x (hi|lo) x 16 B per lane) and do nothing but LDS reads, MFMAs and the epilogue;
//     waves 4-7 ("loader waves", their SIMD partners) fetch the next halo tile from HBM, apply InstanceNorm + activation,
//     split to bf16 hi/lo and write the OTHER LDS buffer.  The matrix pipe and the VALU/memory pipes of a SIMD run side by
//     side; one barrier per tile hands the buffers over.
//   * persistent workgroups walk the 4x4x16 output tiles; a tile's global loads are issued a full tile ahead.
//   * tile geometry is compile-time (6x6x18 halo): LDS offsets are immediates; the loader keeps one precomputed
//     voxel-relative offset per staging slot and adds it to a scalar tile base (no div/mod or 64-bit math per voxel).
// ---------------------------------------------------------------------------------------------------
// Tap order of the <= 16-channel 3x3x3 stride-1 layers (cwf/packing.py: _taps3_order16): position -> natural tap index
// kd*9 + kh*3 + kw.  Pairs (2s, 2s+1): s = 0..8 (kw0, kw1) of row (kd, kh) = (s/3, s%3); s = 9..11 (kd0, kd1) of (kh = s-9,
// kw2); s = 12 (kh0, kh1) of (kd2, kw2); s = 13 the single tap (2,2,2).
__host__ __device__ constexpr int c16_tap(int pos) {
  return pos < 18 ? ((pos >> 1) / 3) * 9 + ((pos >> 1) % 3) * 3 + (pos & 1)
       : pos < 24 ? ((pos - 18) & 1) * 9 + ((pos - 18) >> 1) * 3 + 2
       : pos == 24 ? 18 + 0 + 2 : pos == 25 ? 18 + 3 + 2 : 18 + 6 + 2;
}
#define C16_TD 4
#define C16_TH 4
#define C16_ID 6
#define C16_IH 6
#define C16_IW 18
#define C16_NVOX (C16_ID * C16_IH * C16_IW)          // 648
#define C16_SLOTS ((C16_NVOX + 63) / 64)             // 11 staging slots per loader thread (4 threads per voxel)

// In-kernel phase stamps (diagnostic build only, DIAG = true: cwf_debug_conv16_diag): s_memtime per phase and wave role.
#define CWF_STAMP(v) unsigned long long v = 0; if (DIAG) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); }

template <bool X3, bool DIAG>
__global__ __launch_bounds__(512) void conv16_kernel(const ConvArgsB a, int total_tiles) {
  extern __shared__ float4 lds4[];
  const ConvGeom& g = a.g;
  constexpr int IMG = C16_NVOX * 16;                   // bf16 elements per image
  constexpr int BUF = IMG * (X3 ? 2 : 1);              // per buffer: hi image (+ lo image)
  unsigned short* lds = reinterpret_cast<unsigned short*>(lds4);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the role branch below is provably wave-uniform
  const int tiles_sp = g.tiles_d * g.tiles_h * g.tiles_w;
  // Tile -> workgroup map (speed only, never correctness).  Workgroups are dealt round-robin over the 8 XCDs, each with its own
  // L2: XCD x = b & 7 hosts the `per` workgroups s = b >> 3.  Each XCD owns a contiguous range of "strips" of `per`
  // consecutive tiles (at 128^3: 4 full rows of tiles along H x 8 along W) and its workgroups take one strip per iteration:
  // neighbours in W and H run on the same XCD in the same or the previous iteration, so their shared halo hits in that XCD's
  // L2; only the D halo (8 iterations = 10 MB away) falls through to the Infinity Cache.
  // Measured (2 x 16 x 128^3 bf16x3, PMC FETCH_SIZE x 2 = L2-miss reads; ideal 268 MB):
  //   tile = b + it*256                      320 MB   0.350 ms
  //   per-workgroup contiguous chunks        593 MB   0.333 ms
  //   XCD strips (this map)                  406 MB   0.310 ms   <- fastest
  //   XCD segment of a whole-plane front     308 MB   0.335 ms   (4 MB address jumps per iteration; used by wgrad16)
  const int G = (int)gridDim.x, per = G >> 3;                // grid is a multiple of 8, see launcher
  const int nstrips = (total_tiles + per - 1) / per;
  const int S = (nstrips + 7) >> 3;                          // strips per XCD
  const int xcd = blockIdx.x & 7;
  const int first = xcd * S * per + (blockIdx.x >> 3);       // tile(it) = first + it * per
  int niter = 0;
  {
    const int my_strips = min(S, nstrips - xcd * S);         // may be <= 0 for the last XCDs of a small problem
    if (my_strips > 0) niter = (first + (my_strips - 1) * per < total_tiles) ? my_strips : my_strips - 1;
  }
  if (niter == 0) return;                                    // uniform for the whole workgroup

  if (wave < 4) {
    // =============================================================== MFMA waves
    const int r = lane & 15, kq = lane >> 4;
    const bool second = (kq >> 1) != 0;
    // hi weight images in registers (56 VGPRs); lo images in LDS, lane-linear [14][64] x 16 B (conflict-free), fetched with the
    // A fragments one tap pair ahead.  (hi + lo both in registers spills next to the double-buffered A fragments.)
    uint4 bh[14];
    const uint4* wl = reinterpret_cast<const uint4*>(lds + 2 * BUF) + lane;
    {
      const uint4* wp = a.wpk + lane * 2;
#pragma unroll
      for (int s = 0; s < 14; ++s) bh[s] = wp[s * 128];
      if (X3) {
        uint4* wls = reinterpret_cast<uint4*>(lds + 2 * BUF);
        for (int i = tid; i < 14 * 64; i += 256) wls[i] = a.wpk[(i >> 6) * 128 + (i & 63) * 2 + 1];
      }
    }
    // ---- A-fragment addressing.  LDS byte address = buffer parity + per-lane base[m] + tap offset.  The two taps of
    // one K = 32 step are split over lane halves (lanes 32-63 take the second), whose offset differs from the first's by one of
    // three constants (next kw / next kh row / next kd plane: see c16_tap) or 0 (last, unpaired tap).  Four classes x four M-tiles of per-lane bases
    // stay in registers, so the tap offset is the ds_read immediate and the MFMA loop carries no address arithmetic: the
    // kernel is bound by each SIMD's vector-issue port, shared by its MFMA wave and its loader wave (an MFMA holds it for 8
    // of its 16 cycles, every other vector instruction for >= 4) -- instructions, not bytes, are what is being saved here.
    constexpr int D_KW = 32, D_ROW = C16_IW * 32, D_PLANE = C16_IH * C16_IW * 32;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef const u32x4_t __attribute__((address_space(3)))* lds_u4p;    // 32-bit LDS pointer formed from an integer address
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds4;
    unsigned ab[4][4];                                   // [class: 0 = +kw, 1 = +row (kh), 2 = +plane (kd), 3 = same tap][m]
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const unsigned base = lds_base + (((wave * C16_IH + m) * C16_IW + r) * 16 + (kq & 1) * 8) * 2;
      ab[0][m] = base + (second ? D_KW : 0); ab[1][m] = base + (second ? D_ROW : 0);
      ab[2][m] = base + (second ? D_PLANE : 0); ab[3][m] = base;
    }
    const float bv = (a.bias && r < g.Cout) ? a.bias[r] : 0.f;
    const f32x4 bias4 = {bv, bv, bv, bv};                // accumulators start from the bias (lane = output channel r)
    // epilogue addressing: uniform 64-bit row base (SGPRs) + tile-invariant 32-bit lane offsets -> no address VALU per store
    unsigned yofs[4], rofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { yofs[i] = (unsigned)((kq * 4 + i) * g.y_ldc + r) * 4u; rofs[i] = (unsigned)((kq * 4 + i) * a.r_ldc + r) * 4u; }   // bytes
    // InstanceNorm statistics of the output: kept in registers over this workgroup's tiles, flushed with one f64 atomic
    // pair per wave and channel when the sample index changes and at the end.
    float s1 = 0.f, s2 = 0.f;
    int stat_n = first / tiles_sp;
    float osc = 1.f; int osc_n = -1;                     // per-(n, channel) output scale (stem dropout3d), reloaded when n changes
    auto flush_stats = [&](int n_) {
      float u1 = s1, u2 = s2;
      u1 += __shfl_xor(u1, 16, 64); u1 += __shfl_xor(u1, 32, 64);
      u2 += __shfl_xor(u2, 16, 64); u2 += __shfl_xor(u2, 32, 64);
      if (kq == 0 && r < g.Cout) {
        atomic_add_f64(a.stats + ((int64_t)n_ * g.Cout + r) * 2 + 0, (double)u1);
        atomic_add_f64(a.stats + ((int64_t)n_ * g.Cout + r) * 2 + 1, (double)u2);
      }
      s1 = 0.f; s2 = 0.f;
    };
    // every load issued so far (weights, bias) is complete before the loop: the only VMEM traffic of the loop are the
    // epilogue's stores, which are never waited for (an s_waitcnt vmcnt(0) at the loop head would expose their latency)
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0), expcnt/lgkmcnt untouched

    if (DIAG && (a.diag_mode & 32)) __builtin_amdgcn_s_setprio(1);
    // Deferred epilogue: the common case (interior tile, no residual, no output scale) is not written out after its MFMA
    // phase but DURING the next tile's (one element -- a store and two statistics FMAs -- after each of the first 16 MFMA
    // triples), where those ~50 vector instructions cost issue slots only; done serially they cost ~2k cycles per tile,
    // because the loader wave of the SIMD is converting at the same time.
    f32x4 prev[4];                                       // accumulators of the deferred tile
    float* prev_yb = nullptr;                            // its output base (uniform)
    int pend = 0;                                        // 0 = nothing deferred, 1 = deferred with statistics, 2 = without
    auto drain = [&](auto HT) {                          // non-interleaved form (after the last tile)
      constexpr bool HAS_STATS = decltype(HT)::value;
      unsigned yo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { yo[i] = yofs[i]; asm volatile("" : "+v"(yo[i])); }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        char* ybm = reinterpret_cast<char*>(prev_yb + (int64_t)m * g.Wo * g.y_ldc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = prev[m][i];
          if (!(DIAG && (a.diag_mode & 1))) *reinterpret_cast<float*>(ybm + yo[i]) = v;
          if (HAS_STATS) { s1 += v; s2 = fmaf(v, v, s2); }
        }
      }
    };
    unsigned long long d_bar = 0, d_mfma = 0, d_epi = 0;
    for (int it = 0; it < niter; ++it) {
      const int tile = first + it * per;
      CWF_STAMP(t0);
      // Raw barrier: __syncthreads() would add s_waitcnt vmcnt(0) and make this wave wait for its own output stores.
      asm volatile("s_barrier" ::: "memory");            // buffer it&1 is complete
      CWF_STAMP(t1);
      f32x4 acc[4];
      // A fragments are double-buffered in registers: the 8 LDS reads of tap pair s+1 are issued BEFORE the 12 MFMAs of
      // pair s (one MFMA wave per SIMD: nothing else hides the LDS latency).  sched_barrier pins that order.
      u32x4_t fa[2][4], fl[2][4]; uint4 fb[2];
      auto load_step = [&](int s_, int b_) {
        if (X3) fb[b_] = wl[s_ * 64];
        const int ta = c16_tap(2 * s_);                      // natural index of the step's first tap (second: +kw / +kd / +kh)
        const int oa = (((ta / 9) * C16_IH + (ta / 3) % 3) * C16_IW + ta % 3) * 32;
        const int cls = s_ < 9 ? 0 : s_ < 12 ? 2 : s_ == 12 ? 1 : 3;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          fa[b_][m] = *(lds_u4p)(uintptr_t)(ab[cls][m] + (unsigned)oa);
          if (X3) fl[b_][m] = *(lds_u4p)(uintptr_t)(ab[cls][m] + (unsigned)(oa + IMG * 2));
        }
      };
      auto phase = [&](auto PEND_) {
        constexpr int PEND = decltype(PEND_)::value;
        unsigned yo[4];
        if (PEND) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { yo[i] = yofs[i]; asm volatile("" : "+v"(yo[i])); }     // see the epilogue: keeps the saddr form
        }
        load_step(0, 0);
#pragma unroll
        for (int s = 0; s < 14; ++s) {
          if (s + 1 < 14) load_step(s + 1, (s + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[s & 1][m]), __builtin_bit_cast(bf16x8, bh[s]), s == 0 ? bias4 : acc[m], 0, 0, 0);
            if (X3) {
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[s & 1][m]), __builtin_bit_cast(bf16x8, fb[s & 1]), acc[m], 0, 0, 0);
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fl[s & 1][m]), __builtin_bit_cast(bf16x8, bh[s]), acc[m], 0, 0, 0);
            }
            if (PEND != 0 && s < 4) {                    // deferred element: output row s of the previous tile, register m
              char* ybm = reinterpret_cast<char*>(prev_yb + (int64_t)s * g.Wo * g.y_ldc);
              const float v = prev[s][m];
              if (!(DIAG && (a.diag_mode & 1))) *reinterpret_cast<float*>(ybm + yo[m]) = v;
              if (PEND == 1) { s1 += v; s2 = fmaf(v, v, s2); }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (pend == 0) phase(std::integral_constant<int, 0>{});
      else if (pend == 1) phase(std::integral_constant<int, 1>{});
      else phase(std::integral_constant<int, 2>{});
      pend = 0;
      // the other buffer is read next: toggle the parity of the 16 base addresses
      {
        const unsigned dlt = (it & 1) ? (unsigned)(-(BUF * 2)) : (unsigned)(BUF * 2);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int m = 0; m < 4; ++m) ab[c][m] += dlt;
      }
      CWF_STAMP(t2);
      // ---- epilogue.  (A variant that transposes the accumulators through LDS and stores one contiguous 1 KiB dwordx4 per
      // M-tile was measured SLOWER.)
      const int n = tile / tiles_sp; int rem = tile - n * tiles_sp;
      const int tile_w = rem % g.tiles_w; rem /= g.tiles_w;
      const int tile_h = rem % g.tiles_h; const int tile_d = rem / g.tiles_h;
      const int od = tile_d * C16_TD + wave, oh0 = tile_h * C16_TH, ow0 = tile_w * 16;
      if (a.stats && n != stat_n) { flush_stats(stat_n); stat_n = n; }     // wave-uniform
      if (od < g.Do && !(DIAG && (a.diag_mode & 8))) {                      // wave-uniform
        const int64_t vox0 = (((int64_t)n * g.Do + od) * g.Ho + oh0) * g.Wo + ow0;
        float* yb = a.y + vox0 * g.y_ldc;
        const float* rb = a.residual ? a.residual + vox0 * a.r_ldc : nullptr;
        const bool hs = a.out_scale != nullptr;
        if (hs && n != osc_n) {                                               // rare; its wait must not sit in the tile loop
          osc = r < g.Cout ? a.out_scale[(int64_t)n * g.Cout + r] : 1.f;
          asm volatile("" :: "v"(osc));                                      // consume here -> the s_waitcnt lands here
          osc_n = n;
        }
        const bool full = (oh0 + C16_TH <= g.Ho) && (ow0 + 16 <= g.Wo) && g.Cout == 16 && !(hs && rb);     // wave-uniform
        if (full && !rb && !hs) {
          // common case: deferred into the next tile's MFMA phase
#pragma unroll
          for (int m = 0; m < 4; ++m) prev[m] = acc[m];
          prev_yb = yb; pend = a.stats ? 1 : 2;
        } else if (full) {
          // branch-free fast path (interior tiles, 16 output channels), specialised on the wave-uniform options so that
          // an element costs its store + 2 statistics FMAs: 16 stores at SGPR-base + 32-bit lane-offset addresses.
          // Without a residual the loop issues no loads, so nothing ever waits for the stores (vmcnt is in-order).
          if (hs) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] *= osc;
          }
          auto epi = [&](auto HR, auto HT) {
            constexpr bool HAS_RES = decltype(HR)::value, HAS_STATS = decltype(HT)::value;
            // The 32-bit lane offsets are made opaque HERE so that their zero-extension stays in this basic block: only then
            // does instruction selection see "SGPR base + zext(VGPR)" and emit the saddr form (no 64-bit address VALU).
            unsigned yo[4], ro[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { yo[i] = yofs[i]; asm volatile("" : "+v"(yo[i])); if (HAS_RES) { ro[i] = rofs[i]; asm volatile("" : "+v"(ro[i])); } }
            float rv[4][4];
            if (HAS_RES) {
#pragma unroll
              for (int m = 0; m < 4; ++m) {
                const char* rbm = reinterpret_cast<const char*>(rb + (int64_t)m * g.Wo * a.r_ldc);
#pragma unroll
                for (int i = 0; i < 4; ++i) rv[m][i] = *reinterpret_cast<const float*>(rbm + ro[i]);
              }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              char* ybm = reinterpret_cast<char*>(yb + (int64_t)m * g.Wo * g.y_ldc);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                float v = acc[m][i];
                if (HAS_RES) v += rv[m][i];
                if (!(DIAG && (a.diag_mode & 1))) *reinterpret_cast<float*>(ybm + yo[i]) = v;
                if (HAS_STATS) { s1 += v; s2 = fmaf(v, v, s2); }
              }
            }
          };
          using T_ = std::true_type; using F_ = std::false_type;
          const bool ht = a.stats != nullptr;
          if (rb) { if (ht) epi(T_{}, T_{}); else epi(T_{}, F_{}); }
          else    { if (ht) epi(F_{}, T_{}); else epi(F_{}, F_{}); }
        } else {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool ok = r < g.Cout && oh0 + m < g.Ho && ow0 + kq * 4 + i < g.Wo;
              if (!ok) continue;
              const int eo = m * g.Wo;
              float v = acc[m][i];
              if (rb) v += rb[eo * a.r_ldc + (rofs[i] >> 2)];
              v *= osc;
              yb[eo * g.y_ldc + (yofs[i] >> 2)] = v;
              s1 += v; s2 += v * v;
            }
          }
        }
      }
      CWF_STAMP(t3);
      if (DIAG) { d_bar += t1 - t0; d_mfma += t2 - t1; d_epi += t3 - t2; if (a.diag_mode & 8) { s1 += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]; } }
    }
    if (pend == 1) drain(std::true_type{}); else if (pend == 2) drain(std::false_type{});
    if (a.stats) flush_stats(stat_n);
    if (DIAG && a.diag && lane == 0) {
      unsigned long long* o = a.diag + ((int64_t)blockIdx.x * 8 + wave) * 4;
      o[0] = d_bar; o[1] = d_mfma; o[2] = d_epi; o[3] = 0;
    }
  } else {
    // =============================================================== loader waves
    const int lt = tid - 256;                            // 0..255
    const int q = lt & 3;
    const int c = q * 4;
    const bool cval = c < g.Cin;
    const bool has_norm = a.in_scale != nullptr;
    const float slope = a.in_slope;
    const int HW = g.Hi * g.Wi;
    int rel[C16_SLOTS];
#pragma unroll
    for (int i = 0; i < C16_SLOTS; ++i) {
      const int v = (lt >> 2) + 64 * i;
      const int iw = v % C16_IW, t2 = v / C16_IW;
      const int ih = t2 % C16_IH, idd = t2 / C16_IH;
      rel[i] = (idd * HW + ih * g.Wi + iw) * g.x_ldc + c;
    }
    const bool last_slot_ok = (lt >> 2) + 64 * (C16_SLOTS - 1) < C16_NVOX;
    // Two tiles of loads are kept in flight in two register sets (2 x 11 float4): the ~4 us latency of this halo access
    // pattern exceeds one tile time (measured: with a one-tile distance the tile time equilibrates at the latency).
    // Everything below is STRAIGHT-LINE per tile (out-of-range slots load a dummy address and are zeroed by a select): any
    // per-slot branch makes the compiler fall back from counted s_waitcnt vmcnt(N) to vmcnt(0), which serialises the pipeline.
    float4 pre[2][C16_SLOTS];
    unsigned pre_inb[2] = {0u, 0u};
    if (!(DIAG && (a.diag_mode & 16))) __builtin_amdgcn_s_setprio(1);   // loaders are the critical path: win VALU/VMEM issue arbitration
    struct TileOrg { const float* base; bool interior; int id0, ih0, iw0, n; };
    auto origin = [&](int tile) {
      TileOrg o;
      o.n = tile / tiles_sp; int rem = tile - o.n * tiles_sp;
      const int tile_w = rem % g.tiles_w; rem /= g.tiles_w;
      const int tile_h = rem % g.tiles_h; const int tile_d = rem / g.tiles_h;
      o.id0 = tile_d * C16_TD - 1; o.ih0 = tile_h * C16_TH - 1; o.iw0 = tile_w * 16 - 1;
      o.base = a.x + ((((int64_t)o.n * g.Di + o.id0) * g.Hi + o.ih0) * g.Wi + o.iw0) * g.x_ldc;
      o.interior = o.id0 >= 0 && o.id0 + C16_ID <= g.Di && o.ih0 >= 0 && o.ih0 + C16_IH <= g.Hi && o.iw0 >= 0 && o.iw0 + C16_IW <= g.Wi;
      return o;
    };
    const bool lane_ok = cval && !(DIAG && (a.diag_mode & 2));
    // issue the 11 loads of tile `o` into register set S; returns the in-bounds mask.  No branches inside.
    auto issue = [&](const TileOrg& o, auto S) -> unsigned {
      constexpr int SET = decltype(S)::value;
      unsigned inb = 0;
      if (o.interior) {                                  // wave-uniform; both arms are straight-line
#pragma unroll
        for (int i = 0; i < C16_SLOTS; ++i) {
          if (DIAG && (a.diag_mode & 4) && i >= 7) continue;
          const bool ok = lane_ok && (i < C16_SLOTS - 1 || last_slot_ok);
          const float* p = ok ? o.base + rel[i] : a.x;
          pre[SET][i] = *reinterpret_cast<const float4*>(p);
          inb |= ok ? (1u << i) : 0u;
        }
      } else {
#pragma unroll
        for (int i = 0; i < C16_SLOTS; ++i) {
          const int v = (lt >> 2) + 64 * i;
          const int iw = v % C16_IW, t2 = v / C16_IW;
          const int ih = t2 % C16_IH, idd = t2 / C16_IH;
          const int gd = o.id0 + idd, gh = o.ih0 + ih, gw = o.iw0 + iw;
          const bool ok = lane_ok && (i < C16_SLOTS - 1 || last_slot_ok) &&
                          (unsigned)gd < (unsigned)g.Di && (unsigned)gh < (unsigned)g.Hi && (unsigned)gw < (unsigned)g.Wi;
          const float* p = ok ? o.base + rel[i] : a.x;
          pre[SET][i] = *reinterpret_cast<const float4*>(p);
          inb |= ok ? (1u << i) : 0u;
        }
      }
      return inb;
    };
    // convert register set S (tile tc) into LDS buffer `buf`.  Specialised on two wave-uniform facts so that the common case
    // (interior tile, all 16 channels) carries no per-value selects: PLAIN = no norm/activation prologue (data gradients),
    // ALLIN = every slot of every lane of this wave was in bounds.  ~22 vector instructions per float4 instead of 34.
    const unsigned full_mask = last_slot_ok ? ((1u << C16_SLOTS) - 1u) : ((1u << (C16_SLOTS - 1)) - 1u);
    const bool plain = !has_norm && slope == 1.f;
    auto convert = [&](int tc, int buf, auto S) {
      constexpr int SET = decltype(S)::value;
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_norm && cval) {
        const int n = tc / tiles_sp;
        sc = *reinterpret_cast<const float4*>(a.in_scale + (int64_t)n * g.Cin + c);
        sh = *reinterpret_cast<const float4*>(a.in_shift + (int64_t)n * g.Cin + c);
      }
      const float sl = slope;
      const unsigned inb = pre_inb[SET];
      unsigned short* dh = lds + buf * BUF + (lt >> 2) * 16 + q * 4;
      unsigned short* dl = dh + IMG;
      auto body = [&](auto PL, auto AI) {
        constexpr bool PLAIN = decltype(PL)::value, ALLIN = decltype(AI)::value;
#pragma unroll
        for (int i = 0; i < C16_SLOTS; ++i) {
          if (i == C16_SLOTS - 1 && !last_slot_ok) continue;         // (whole-quad predicate, not per slot: cheap)
          if (DIAG && (a.diag_mode & 4) && i >= 7) continue;
          const float4 val = pre[SET][i];
          float v0 = val.x, v1 = val.y, v2 = val.z, v3 = val.w;
          if (!PLAIN) {
            v0 = act01(fmaf(v0, sc.x, sh.x), sl); v1 = act01(fmaf(v1, sc.y, sh.y), sl);
            v2 = act01(fmaf(v2, sc.z, sh.z), sl); v3 = act01(fmaf(v3, sc.w, sh.w), sl);
          }
          uint2 h, l;
          if (X3) { split_bf16(v0, v1, h.x, l.x); split_bf16(v2, v3, h.y, l.y); }
          else { h.x = pack_bf16(v0, v1); h.y = pack_bf16(v2, v3); }
          if (!ALLIN) {
            // zero padding is applied AFTER the activation: out-of-range voxels are exactly 0 in both images
            const bool was = (inb >> i) & 1u;
            h.x = was ? h.x : 0u; h.y = was ? h.y : 0u;
            if (X3) { l.x = was ? l.x : 0u; l.y = was ? l.y : 0u; }
          }
          *reinterpret_cast<uint2*>(dh + i * 64 * 16) = h;
          if (X3) *reinterpret_cast<uint2*>(dl + i * 64 * 16) = l;
        }
      };
      using T_ = std::true_type; using F_ = std::false_type;
      const bool allin = __ballot(inb != full_mask) == 0ull;          // wave-uniform
      if (plain) { if (allin) body(T_{}, T_{}); else body(T_{}, F_{}); }
      else       { if (allin) body(F_{}, T_{}); else body(F_{}, F_{}); }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    unsigned long long d_bar = 0, d_commit = 0;
    // prologue: tile 0 -> set 0 -> LDS buffer 0; then tiles 1 (set 1) and 2 (set 0) in flight
    pre_inb[0] = issue(origin(first), S0{});
    convert(first, 0, S0{});
    if (niter > 1) pre_inb[1] = issue(origin(first + per), S1{});
    if (niter > 2) pre_inb[0] = issue(origin(first + 2 * per), S0{});
    // iteration it: tile it+1 sits in set (it+1)&1 -> LDS buffer (it+1)&1; that set is then refilled with tile it+3
    for (int it = 0; it < niter; ++it) {
      CWF_STAMP(t0);
      // LDS writes done -> barrier.  Raw form: __syncthreads() would also wait (vmcnt(0)) for the prefetches in flight.
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // buffer it&1 handed over; buffer (it+1)&1 is free
      CWF_STAMP(t1);
      if (it + 1 < niter) {
        if ((it + 1) & 1) {
          convert(first + (it + 1) * per, 1, S1{});
          if (it + 3 < niter) pre_inb[1] = issue(origin(first + (it + 3) * per), S1{});
        } else {
          convert(first + (it + 1) * per, 0, S0{});
          if (it + 3 < niter) pre_inb[0] = issue(origin(first + (it + 3) * per), S0{});
        }
      }
      if (DIAG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      CWF_STAMP(t2);
      if (DIAG) { d_bar += t1 - t0; d_commit += t2 - t1; }
    }
    if (DIAG && a.diag && lane == 0) {
      unsigned long long* o = a.diag + ((int64_t)blockIdx.x * 8 + wave) * 4;
      o[0] = d_bar; o[1] = 0; o[2] = d_commit; o[3] = 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// conv16s: conv16 with a SLIDING WINDOW along H.  A workgroup walks a column of 4x4x16 tiles (fixed n, tile_d, tile_w,
// increasing tile_h); the halo rows two neighbouring tiles share stay in LDS, so a tile stages 4 new rows per plane
// (7 staging slots per loader thread) instead of 6 (11 slots): the loader waves -- the longer of the kernel's two per-tile
// chains -- do 36 % less work (loads, converts, LDS writes).  LDS holds, per plane, a RING of 12 rows:
//     input row j of a column segment (j = 0 at global row 4*th0 - 1) lives in slot (j + 2) % 12,
// tile t reads rows 4t..4t+5 and the loader meanwhile writes rows 4t+6..4t+9 of tile t+1 (slots disjoint: 10 <= 12).
// With t % 3 a compile-time phase (the tile loop is unrolled by three) every LDS offset is an immediate again; the tap
// pairing of c16_tap keeps the second tap of a K = 32 step at a phase-independent offset except for the one (kh0, kh1) pair,
// which gets two base registers (row + 1, or the ring wrap).  A column segment starts with all six rows new: it is preceded
// by a loader-only "pre-tile" (element e = 0) that stages rows -2..1 exactly like any other 4-row block (the MFMA waves only
// pass its barrier), so the ring simply keeps turning across segments and the loader has no special case at all
// (cost: one extra 4-row block per segment, ~3 % at 128^3).  Elements u = 0, 1, 2, ... of a workgroup: new rows -> group u % 3.
// Everything else (weights in registers, deferred epilogue, statistics in registers, XCD-aware work order) is conv16_kernel's.
// ---------------------------------------------------------------------------------------------------
#define C16_RH 12
#define C16S_SLOTS 7                                     // ceil(6 planes * 4 rows * 18 / 64)
#define C16S_HSLOTS 4                                    // ceil(6 planes * 2 rows * 18 / 64)

struct C16sWork { int nseg, seg_len, hsplit, ncols; };
static int g_conv16_diag_mode = 0;                   // diagnostics (cwf_debug_conv16_mode): 1 no stores, 2 no loads, 8 no epilogue

// IN16 = true (single-bf16 data-gradient launches whose input gradient exists as a bf16 image, [N][D][H][W][16]): the loader waves
// convert nothing -- they issue LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB of LDS each; zero padding = a 16-byte zero page) and the
// ring has FOUR row groups per plane (16 rows) so that two elements stay in flight while a tile reads one and a half groups.
template <bool X3, bool DIAG, bool IN16 = false>
__global__ __launch_bounds__(512) void conv16s_kernel(const ConvArgsB a, const C16sWork wk) {
  static_assert(!(IN16 && X3), "the bf16 input image is a single-bf16 operand");
  constexpr int RH = IN16 ? 16 : C16_RH;               // ring rows per plane
  constexpr int NPH = RH / 4;                          // ring groups = compile-time phases of the tile loop
  extern __shared__ float4 lds4[];
  const ConvGeom& g = a.g;
  constexpr int ROWB = C16_IW * 32;                    // bytes of one LDS row (18 voxels x 16 ch bf16)
  constexpr int PLANEB = RH * ROWB;                   // one plane of the ring
  constexpr int IMGB = C16_ID * PLANEB;                // one image (hi or lo): 41,472 B
  unsigned short* lds = reinterpret_cast<unsigned short*>(lds4);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = (int)gridDim.x, per = G >> 3;
  const int first_seg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);      // segments first_seg + k*G
  if (first_seg >= wk.nseg) return;
  const int nk = (wk.nseg - first_seg + G - 1) / G;                      // segments of this workgroup
  // segment sg -> (hseg, n, tile_d, tile_w), rows th0 .. th0+len-1
  struct Seg { int n, tile_d, tile_w, th0, len; };
  auto seg_of = [&](int k) __attribute__((always_inline)) {
    Seg sgm;
    const int sg = first_seg + k * G;
    const int hseg = sg / wk.ncols; int col = sg - hseg * wk.ncols;
    sgm.tile_w = col % g.tiles_w; col /= g.tiles_w;
    sgm.tile_d = col % g.tiles_d; sgm.n = col / g.tiles_d;
    sgm.th0 = hseg * wk.seg_len;
    sgm.len = min(wk.seg_len, g.tiles_h - sgm.th0);
    return sgm;
  };

  if (wave < 4) {
    // =============================================================== MFMA waves
    const int r = lane & 15, kq = lane >> 4;
    const bool second = (kq >> 1) != 0;
    uint4 bh[14];
    const uint4* wl = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lds) + 2 * IMGB) + lane;
    {
      const uint4* wp = a.wpk + lane * 2;
#pragma unroll
      for (int s = 0; s < 14; ++s) bh[s] = wp[s * 128];
      if (X3) {
        uint4* wls = reinterpret_cast<uint4*>(reinterpret_cast<char*>(lds) + 2 * IMGB);
        for (int i = tid; i < 14 * 64; i += 256) wls[i] = a.wpk[(i >> 6) * 128 + (i & 63) * 2 + 1];
      }
    }
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef const u32x4_t __attribute__((address_space(3)))* lds_u4p;
    // per-lane bases: plane `wave`, row slot 0, voxel r, channel half kq&1; second-half lanes carry the pair's offset
    const unsigned bl = lds_base + (unsigned)(wave * PLANEB + r * 32 + (kq & 1) * 16);
    const unsigned b_kw = bl + (second ? 32u : 0u);                      // (kw0, kw1) pairs
    const unsigned b_pl = bl + (second ? (unsigned)PLANEB : 0u);         // (kd0, kd1) pairs
    const unsigned b_up = bl + (second ? (unsigned)ROWB : 0u);           // (kh0, kh1) pair, next slot
    const unsigned b_wr = bl + (second ? 0u : (unsigned)((RH - 1) * ROWB));    // (kh0, kh1) pair across the ring wrap (slot 11 -> 0): the
                                                                         // immediate addresses the SECOND tap's row, first lanes add 11 rows
    const float bv = (a.bias && r < g.Cout) ? a.bias[r] : 0.f;
    const f32x4 bias4 = {bv, bv, bv, bv};
    unsigned yofs[4], rofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { yofs[i] = (unsigned)((kq * 4 + i) * g.y_ldc + r) * 4u; rofs[i] = (unsigned)((kq * 4 + i) * a.r_ldc + r) * 4u; }
    float s1 = 0.f, s2 = 0.f;
    int stat_n = seg_of(0).n;
    float osc = 1.f; int osc_n = -1;
    auto flush_stats = [&](int n_) __attribute__((always_inline)) {
      float u1 = s1, u2 = s2;
      u1 += __shfl_xor(u1, 16, 64); u1 += __shfl_xor(u1, 32, 64);
      u2 += __shfl_xor(u2, 16, 64); u2 += __shfl_xor(u2, 32, 64);
      if (kq == 0 && r < g.Cout) {
        atomic_add_f64(a.stats + ((int64_t)n_ * g.Cout + r) * 2 + 0, (double)u1);
        atomic_add_f64(a.stats + ((int64_t)n_ * g.Cout + r) * 2 + 1, (double)u2);
      }
      s1 = 0.f; s2 = 0.f;
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // prologue loads done: the loop's stores are never waited for

    f32x4 prev[4]; float* prev_yb = nullptr; int pend = 0;               // deferred epilogue (see conv16_kernel)
    auto drain = [&](auto HT) __attribute__((always_inline)) {
      constexpr bool HAS_STATS = decltype(HT)::value;
      unsigned yo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { yo[i] = yofs[i]; asm volatile("" : "+v"(yo[i])); }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        char* ybm = reinterpret_cast<char*>(prev_yb + (int64_t)m * g.Wo * g.y_ldc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = prev[m][i];
          *reinterpret_cast<float*>(ybm + yo[i]) = v;
          if (HAS_STATS) { s1 += v; s2 = fmaf(v, v, s2); }
        }
      }
    };

    // one tile: MFMA phase (ring phase PH = t % 3 compile-time) with the previous tile's deferred epilogue, then this tile's
    auto tile = [&](auto PH_, const Seg& sg, int t) __attribute__((always_inline)) {
      constexpr int PH = decltype(PH_)::value;
      asm volatile("s_barrier" ::: "memory");            // rows of tile t are complete
      f32x4 acc[4];
      // LDS-read lookahead in K-steps.  Split operands: one step (12 MFMAs = 192 cycles per step cover the read latency, and two
      // more fragment sets would not fit the 256-VGPR budget).  Single-bf16 operands (the data-gradient launches): 4 MFMAs = 64
      // cycles per step do NOT cover it -- the wave stalled on every step (PMC: MFMA busy 15 %, LDS busy 16 %) -- so the reads run
      // two steps ahead there, in the registers the lo fragments do not need.
      constexpr int LA = X3 ? 1 : 2, NB = LA + 1;
      u32x4_t fa[NB][4], fl[X3 ? NB : 1][4]; uint4 fb[NB];
      auto load_step = [&](int s_, int b_) __attribute__((always_inline)) {
        if (X3) fb[b_] = wl[s_ * 64];
        const int ta = c16_tap(2 * s_);
        const int kd = ta / 9, kh = (ta / 3) % 3, kw = ta % 3;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int slot = (4 * PH + RH - 2 + m + kh) % RH;             // first tap's row slot (rows 0,1 of the tile: previous group)
          unsigned base; int imm;
          if (s_ < 9) { base = b_kw; imm = (kd * RH + slot) * ROWB + kw * 32; }
          else if (s_ < 12) { base = b_pl; imm = (kd * RH + slot) * ROWB + kw * 32; }
          else if (s_ == 12) {
            const int slot2 = (slot + 1) % RH;
            if (slot2 == slot + 1) { base = b_up; imm = (kd * RH + slot) * ROWB + kw * 32; }
            else { base = b_wr; imm = (kd * RH + slot2) * ROWB + kw * 32; }        // wrap: immediate = second tap's row (slot 0)
          } else { base = bl; imm = (kd * RH + slot) * ROWB + kw * 32; }
          fa[b_][m] = *(lds_u4p)(uintptr_t)(base + (unsigned)imm);
          if (X3) fl[b_][m] = *(lds_u4p)(uintptr_t)(base + (unsigned)(imm + IMGB));
        }
      };
      auto phase = [&](auto PEND_) __attribute__((always_inline)) {
        constexpr int PEND = decltype(PEND_)::value;
        unsigned yo[4];
        if (PEND) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { yo[i] = yofs[i]; asm volatile("" : "+v"(yo[i])); }
        }
#pragma unroll
        for (int s0 = 0; s0 < LA; ++s0) load_step(s0, s0);
#pragma unroll
        for (int s = 0; s < 14; ++s) {
          if (s + LA < 14) load_step(s + LA, (s + LA) % NB);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[s % NB][m]), __builtin_bit_cast(bf16x8, bh[s]), s == 0 ? bias4 : acc[m], 0, 0, 0);
            if (X3) {
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[s % NB][m]), __builtin_bit_cast(bf16x8, fb[s % NB]), acc[m], 0, 0, 0);
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fl[X3 ? s % NB : 0][m]), __builtin_bit_cast(bf16x8, bh[s]), acc[m], 0, 0, 0);
            }
            if (PEND != 0 && s < 4) {
              char* ybm = reinterpret_cast<char*>(prev_yb + (int64_t)s * g.Wo * g.y_ldc);
              const float v = prev[s][m];
              *reinterpret_cast<float*>(ybm + yo[m]) = v;
              if (PEND == 1) { s1 += v; s2 = fmaf(v, v, s2); }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (pend == 0) phase(std::integral_constant<int, 0>{});
      else if (pend == 1) phase(std::integral_constant<int, 1>{});
      else phase(std::integral_constant<int, 2>{});
      pend = 0;
      // ---- epilogue of this tile
      const int n = sg.n;
      const int od = sg.tile_d * C16_TD + wave, oh0 = (sg.th0 + t) * C16_TH, ow0 = sg.tile_w * 16;
      if (a.stats && n != stat_n) { flush_stats(stat_n); stat_n = n; }
      if (od < g.Do && !(DIAG && (a.diag_mode & 8))) {     // diag_mode 8: no epilogue (DIAG build only)
        const int64_t vox0 = (((int64_t)n * g.Do + od) * g.Ho + oh0) * g.Wo + ow0;
        float* yb = a.y + vox0 * g.y_ldc;
        const float* rb = a.residual ? a.residual + vox0 * a.r_ldc : nullptr;
        const bool hs = a.out_scale != nullptr;
        if (hs && n != osc_n) {
          osc = r < g.Cout ? a.out_scale[(int64_t)n * g.Cout + r] : 1.f;
          asm volatile("" :: "v"(osc));
          osc_n = n;
        }
        // norm-backward statistics (nb launches only; they take the immediate epilogue, so nothing of this lives across tiles --
        // the MFMA waves have no register to spare)
        const bool nb = a.nb_x != nullptr;
        const float* xbn = nb ? a.nb_x + vox0 * a.nb_ldc : nullptr;
        float nsc = 0.f, nsh = 0.f;
        if (nb && r < g.Cout) { nsc = a.nb_scale[(int64_t)n * g.Cout + r]; nsh = a.nb_shift[(int64_t)n * g.Cout + r]; }
        const bool full = (oh0 + C16_TH <= g.Ho) && (ow0 + 16 <= g.Wo) && g.Cout == 16 && !(hs && rb);
        if (full && !rb && !hs && !nb) {
#pragma unroll
          for (int m = 0; m < 4; ++m) prev[m] = acc[m];
          prev_yb = yb; pend = a.stats ? 1 : 2;
        } else if (full) {
          if (hs) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] *= osc;
          }
          auto epi = [&](auto HR, auto HT, auto HN) __attribute__((always_inline)) {
            constexpr bool HAS_RES = decltype(HR)::value, HAS_STATS = decltype(HT)::value, HAS_NB = decltype(HN)::value;
            unsigned yo[4], ro[4], xo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              yo[i] = yofs[i]; asm volatile("" : "+v"(yo[i]));
              if (HAS_RES) { ro[i] = rofs[i]; asm volatile("" : "+v"(ro[i])); }
              if (HAS_NB) { xo[i] = (unsigned)((kq * 4 + i) * a.nb_ldc + r) * 4u; asm volatile("" : "+v"(xo[i])); }
            }
            float xv[4][4];
            if (HAS_NB) {
#pragma unroll
              for (int m = 0; m < 4; ++m) {
                const char* xbm = reinterpret_cast<const char*>(xbn + (int64_t)m * g.Wo * a.nb_ldc);
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[m][i] = *reinterpret_cast<const float*>(xbm + xo[i]);
              }
            }
            float rv[4][4];
            if (HAS_RES) {
#pragma unroll
              for (int m = 0; m < 4; ++m) {
                const char* rbm = reinterpret_cast<const char*>(rb + (int64_t)m * g.Wo * a.r_ldc);
#pragma unroll
                for (int i = 0; i < 4; ++i) rv[m][i] = *reinterpret_cast<const float*>(rbm + ro[i]);
              }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              char* ybm = reinterpret_cast<char*>(yb + (int64_t)m * g.Wo * g.y_ldc);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                float v = acc[m][i];
                if (HAS_RES) v += rv[m][i];
                *reinterpret_cast<float*>(ybm + yo[i]) = v;
                if (HAS_NB) {
                  const float h = fmaf(xv[m][i], nsc, nsh);
                  const float gn = v * (h > 0.f ? 1.f : a.nb_slope);
                  s1 += gn; s2 = fmaf(gn, h, s2);
                } else if (HAS_STATS) { s1 += v; s2 = fmaf(v, v, s2); }
              }
            }
          };
          using T_ = std::true_type; using F_ = std::false_type;
          const bool ht = a.stats != nullptr;
          if (nb) { if (rb) epi(T_{}, T_{}, T_{}); else epi(F_{}, T_{}, T_{}); }
          else if (rb) { if (ht) epi(T_{}, T_{}, F_{}); else epi(T_{}, F_{}, F_{}); }
          else    { if (ht) epi(F_{}, T_{}, F_{}); else epi(F_{}, F_{}, F_{}); }
        } else {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool ok = r < g.Cout && oh0 + m < g.Ho && ow0 + kq * 4 + i < g.Wo;
              if (!ok) continue;
              const int eo = m * g.Wo;
              float v = acc[m][i];
              if (rb) v += rb[eo * a.r_ldc + (rofs[i] >> 2)];
              v *= osc;
              yb[eo * g.y_ldc + (yofs[i] >> 2)] = v;
              if (nb) {
                const float h = fmaf(xbn[eo * a.nb_ldc + (kq * 4 + i) * a.nb_ldc + r], nsc, nsh);
                const float gn = v * (h > 0.f ? 1.f : a.nb_slope);
                s1 += gn; s2 += gn * h;
              } else { s1 += v; s2 += v * v; }
            }
          }
        }
      }
    };

    int u = 0;                                           // element counter of this workgroup (pre-tiles included)
    for (int k = 0; k < nk; ++k) {
      const Seg sg = seg_of(k);
      asm volatile("s_barrier" ::: "memory");            // pre-tile: loader only
      ++u;
      for (int t = 0; t < sg.len; ++t, ++u) {
        const int ph = u % NPH;
        if (ph == 0) tile(std::integral_constant<int, 0>{}, sg, t);
        else if (ph == 1) tile(std::integral_constant<int, 1>{}, sg, t);
        else if (NPH == 3 || ph == 2) tile(std::integral_constant<int, 2>{}, sg, t);
        else tile(std::integral_constant<int, NPH - 1>{}, sg, t);
      }
    }
    if (pend == 1) drain(std::true_type{}); else if (pend == 2) drain(std::false_type{});
    if (a.stats) flush_stats(stat_n);
  } else if constexpr (IN16) {
    // =============================================================== loader waves, bf16 input image: LDS-DMA only
    // Element (k, e) (e = 0: the segment's pre-tile) brings 6 planes x 4 new rows x 18 voxels x 32 B: per plane one contiguous
    // 2,304-B block of the ring = 144 granules of 16 B = pieces of 64 + 64 + 16 lanes.  Wave lw takes planes lw, lw + 4.
    // Element u lands in ring group u % 4.  Tile u reads groups u - 1 (its last two rows) and u, elements u + 1 and u + 2 are in
    // flight: element it + 2 is issued right after barrier `it` (group it + 2 was last read by tile it - 1, which every MFMA wave
    // finished before that barrier), and before barrier `it` this wave waits for its pieces of element `it` (counted vmcnt).
    const int lw = wave - 4;
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* glb_vp;
    struct It { int k, e; };
    auto next = [&](It it) __attribute__((always_inline)) { It o = it; if (it.e < seg_of(it.k).len) o.e = it.e + 1; else { o.k = it.k + 1; o.e = 0; } return o; };
    int niter = 0;
    for (int k = 0; k < nk; ++k) niter += seg_of(k).len + 1;
    auto run = [&](auto NPL_) __attribute__((always_inline)) {
      constexpr int NPL = decltype(NPL_)::value, NI = 3 * NPL;
      int off[NI]; unsigned crd[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int pl = lw + 4 * (j / 3), gi = 64 * (j % 3) + lane;
        const int r4 = gi / 36, gr = gi % 36, w = gr >> 1, half = gr & 1;
        off[j] = ((pl * g.Hi + r4) * g.Wi + w) * 2 + half;
        crd[j] = (unsigned)pl | ((unsigned)r4 << 3) | ((unsigned)w << 6) | (gi < 144 ? 1u << 11 : 0u);
      }
      auto issue = [&](It it, int u) __attribute__((always_inline)) {
        const Seg sgm = seg_of(it.k);
        const int id0 = sgm.tile_d * C16_TD - 1, ih0 = (sgm.th0 + it.e - 1) * C16_TH + 1, iw0 = sgm.tile_w * 16 - 1;
        const int64_t v0 = (((int64_t)sgm.n * g.Di + id0) * g.Hi + ih0) * g.Wi + iw0;
        const unsigned lgrp = lds_base + (unsigned)((u & 3) * 4 * ROWB);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int pl = lw + 4 * (j / 3);                 // (wave-uniform)
          if ((crd[j] >> 11) != 0u) {                      // piece 2 of a plane: 16 lanes (the DMA writes LDS for active lanes only)
            const int gd = id0 + (int)(crd[j] & 7u), gh = ih0 + (int)((crd[j] >> 3) & 7u), gw = iw0 + (int)((crd[j] >> 6) & 31u);
            const bool ok = (unsigned)gd < (unsigned)g.Di && (unsigned)gh < (unsigned)g.Hi && (unsigned)gw < (unsigned)g.Wi;
            const uint4* src = ok ? a.x16 + v0 * 2 + off[j] : a.zero16;
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(uintptr_t)(lgrp + (unsigned)(pl * PLANEB + (j % 3) * 1024)), 16, 0, 0);
          }
        }
      };
      It ti{0, 0};
      issue(ti, 0); ti = next(ti);
      if (niter > 1) { issue(ti, 1); ti = next(ti); }
      for (int it = 0; it < niter; ++it) {
        if (it + 1 < niter) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");          // element `it` handed over
        if (it + 2 < niter) { issue(ti, it + 2); ti = next(ti); }
      }
    };
    if (lw < 2) run(std::integral_constant<int, 2>{}); else run(std::integral_constant<int, 1>{});
  } else {
    // =============================================================== loader waves
    const int lt = tid - 256;
    const int q = lt & 3, c = q * 4;
    const bool cval = c < g.Cin && !(DIAG && (a.diag_mode & 2));   // diag_mode 2: no loads (DIAG build only)
    const bool has_norm = a.in_scale != nullptr;
    const float slope = a.in_slope;
    const bool plain = !has_norm && slope == 1.f;
    const int HW = g.Hi * g.Wi;
    // staging slots of the 4 new rows: voxel v -> (plane, row 0..3, w)
    int rel[C16S_SLOTS]; unsigned lrel[C16S_SLOTS];
#pragma unroll
    for (int i = 0; i < C16S_SLOTS; ++i) {
      const int v = (lt >> 2) + 64 * i;
      const int pl = v / 72, r4 = (v % 72) / 18, w = v % 18;
      rel[i] = (pl * HW + r4 * g.Wi + w) * g.x_ldc + c;
      lrel[i] = (unsigned)((pl * RH + r4) * ROWB + w * 32 + q * 8);
    }
    const bool last_ok = (lt >> 2) + 64 * (C16S_SLOTS - 1) < 432;
    const unsigned full_mask = last_ok ? ((1u << C16S_SLOTS) - 1u) : ((1u << (C16S_SLOTS - 1)) - 1u);
    float4 pre[2][C16S_SLOTS];
    unsigned pre_inb[2] = {0u, 0u};
    __builtin_amdgcn_s_setprio(1);
    // flattened element iterator (k, e): e = 0 is the segment's pre-tile (rows -2..1), e >= 1 the tile t = e - 1
    struct It { int k, e; };
    auto next = [&](It it) __attribute__((always_inline)) { It o = it; if (it.e < seg_of(it.k).len) o.e = it.e + 1; else { o.k = it.k + 1; o.e = 0; } return o; };
    struct Org { const float* base; bool interior; int id0, ih0, iw0, n; };
    // origin of the 4 NEW rows of element (k, e): planes od0-1.., rows 4*(th0+e-1)+1.., cols ow0-1..
    auto origin = [&](It it) __attribute__((always_inline)) {
      const Seg sgm = seg_of(it.k);
      Org o;
      o.n = sgm.n;
      o.id0 = sgm.tile_d * C16_TD - 1; o.ih0 = (sgm.th0 + it.e - 1) * C16_TH + 1; o.iw0 = sgm.tile_w * 16 - 1;
      o.base = a.x + ((((int64_t)o.n * g.Di + o.id0) * g.Hi + o.ih0) * g.Wi + o.iw0) * g.x_ldc;
      o.interior = o.id0 >= 0 && o.id0 + C16_ID <= g.Di && o.ih0 >= 0 && o.ih0 + 4 <= g.Hi && o.iw0 >= 0 && o.iw0 + C16_IW <= g.Wi;
      return o;
    };
    float4 pre_sc[2], pre_sh[2];                         // the element's per-sample scale / shift travel with its register set
    auto issue = [&](const Org& o, auto S) __attribute__((always_inline)) -> unsigned {
      constexpr int SET = decltype(S)::value;
      unsigned inb = 0;
      if (o.interior) {
#pragma unroll
        for (int i = 0; i < C16S_SLOTS; ++i) {
          const bool ok = cval && (i < C16S_SLOTS - 1 || last_ok);
          const float* p = ok ? o.base + rel[i] : a.x;
          pre[SET][i] = *reinterpret_cast<const float4*>(p);
          inb |= ok ? (1u << i) : 0u;
        }
      } else {
#pragma unroll
        for (int i = 0; i < C16S_SLOTS; ++i) {
          const int v = (lt >> 2) + 64 * i;
          const int pl = v / 72, r4 = (v % 72) / 18, w = v % 18;
          const int gd = o.id0 + pl, gh = o.ih0 + r4, gw = o.iw0 + w;
          const bool ok = cval && (i < C16S_SLOTS - 1 || last_ok) &&
                          (unsigned)gd < (unsigned)g.Di && (unsigned)gh < (unsigned)g.Hi && (unsigned)gw < (unsigned)g.Wi;
          const float* p = ok ? o.base + rel[i] : a.x;
          pre[SET][i] = *reinterpret_cast<const float4*>(p);
          inb |= ok ? (1u << i) : 0u;
        }
      }
      if (has_norm) {                                    // kernel-uniform; dummy address for lanes beyond Cin
        const int cc = cval ? c : 0;
        pre_sc[SET] = *reinterpret_cast<const float4*>(a.in_scale + (int64_t)o.n * g.Cin + cc);
        pre_sh[SET] = *reinterpret_cast<const float4*>(a.in_shift + (int64_t)o.n * g.Cin + cc);
      }
      return inb;
    };
    // one float4 -> bf16 hi/lo -> LDS; PLAIN (no norm/activation) and SELECT (zero out-of-bounds voxels after the activation)
    // are compile-time so that a staging slot is straight-line code
    auto cvt_store = [&](float4 val, bool was, const float4& sc, const float4& sh, unsigned dst, auto PL, auto SEL) __attribute__((always_inline)) {
      constexpr bool PLAIN = decltype(PL)::value, SELECT = decltype(SEL)::value;
      float v0 = val.x, v1 = val.y, v2 = val.z, v3 = val.w;
      if (!PLAIN) {
        v0 = act01(fmaf(v0, sc.x, sh.x), slope); v1 = act01(fmaf(v1, sc.y, sh.y), slope);
        v2 = act01(fmaf(v2, sc.z, sh.z), slope); v3 = act01(fmaf(v3, sc.w, sh.w), slope);
      }
      uint2 h, l;
      if (X3) { split_bf16(v0, v1, h.x, l.x); split_bf16(v2, v3, h.y, l.y); }
      else { h.x = pack_bf16(v0, v1); h.y = pack_bf16(v2, v3); l = make_uint2(0u, 0u); }
      if (SELECT) { h.x = was ? h.x : 0u; h.y = was ? h.y : 0u; l.x = was ? l.x : 0u; l.y = was ? l.y : 0u; }
      char* db = reinterpret_cast<char*>(lds);
      *reinterpret_cast<uint2*>(db + dst) = h;
      if (X3) *reinterpret_cast<uint2*>(db + dst + IMGB) = l;
    };
    // convert register set S into ring group GRP (slots 4*GRP .. 4*GRP+3)
    auto convert = [&](auto S, auto GRP_) __attribute__((always_inline)) {
      constexpr int SET = decltype(S)::value, GRP = decltype(GRP_)::value;
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_norm) { sc = pre_sc[SET]; sh = pre_sh[SET]; }
      const unsigned inb = pre_inb[SET];
      const bool allin = __ballot(inb != full_mask) == 0ull;
      auto body = [&](auto PL, auto AI) __attribute__((always_inline)) {
        constexpr bool ALLIN = decltype(AI)::value;
#pragma unroll
        for (int i = 0; i < C16S_SLOTS; ++i) {
          if (i == C16S_SLOTS - 1 && !last_ok) continue;
          cvt_store(pre[SET][i], (inb >> i) & 1u, sc, sh, lrel[i] + (unsigned)(GRP * 4 * ROWB), PL, std::integral_constant<bool, !ALLIN>{});
        }
      };
      using T_ = std::true_type; using F_ = std::false_type;
      if (plain) { if (allin) body(T_{}, T_{}); else body(T_{}, F_{}); }
      else       { if (allin) body(F_{}, T_{}); else body(F_{}, F_{}); }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto convert_u = [&](int uu, auto S) __attribute__((always_inline)) {   // element number uu -> ring group uu % 3
      const int grp = uu % 3;
      if (grp == 0) convert(S, std::integral_constant<int, 0>{});
      else if (grp == 1) convert(S, std::integral_constant<int, 1>{});
      else convert(S, std::integral_constant<int, 2>{});
    };
    // Flattened element sequence of this workgroup; element number `it` uses register set it & 1 and two elements of loads are
    // in flight (three sets in flight measured SLOWER: 0.265 vs 0.247 ms).  Same straight-line shape as conv16_kernel's loader
    // (each parity arm converts one set, then refills that same set) so that the compiler keeps counted s_waitcnt vmcnt(N).
    int niter = 0;
    for (int k = 0; k < nk; ++k) niter += seg_of(k).len + 1;
    It ti{0, 0};                                         // element whose loads are issued next
    pre_inb[0] = issue(origin(ti), S0{}); ti = next(ti);
    convert_u(0, S0{});
    if (niter > 1) { pre_inb[1] = issue(origin(ti), S1{}); ti = next(ti); }
    if (niter > 2) { pre_inb[0] = issue(origin(ti), S0{}); ti = next(ti); }
    for (int it = 0; it < niter; ++it) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // element `it` handed over
      if (it + 1 < niter) {
        if ((it + 1) & 1) {
          convert_u(it + 1, S1{});
          if (it + 3 < niter) { pre_inb[1] = issue(origin(ti), S1{}); ti = next(ti); }
        } else {
          convert_u(it + 1, S0{});
          if (it + 3 < niter) { pre_inb[0] = issue(origin(ti), S0{}); ti = next(ti); }
        }
      }
    }
  }
}

template <bool X3, bool IN16 = false>
static int launch_conv16s(const ConvArgsB& a, hipStream_t st) {
  const ConvGeom& g = a.g;
  C16sWork wk;
  wk.ncols = g.N * g.tiles_d * g.tiles_w;
  int hsplit = (256 + wk.ncols - 1) / wk.ncols;          // enough segments for 256 workgroups ...
  if (hsplit > g.tiles_h / 3) hsplit = g.tiles_h / 3;    // ... but at least 3 tiles per segment (a segment start costs 2 extra rows)
  if (hsplit < 1) hsplit = 1;
  wk.seg_len = (g.tiles_h + hsplit - 1) / hsplit;
  wk.hsplit = (g.tiles_h + wk.seg_len - 1) / wk.seg_len;
  wk.nseg = wk.ncols * wk.hsplit;
  const size_t lds = IN16 ? (size_t)C16_ID * 16 * C16_IW * 32                                    // hi ring of 16 rows
                          : (size_t)2 * C16_ID * C16_RH * C16_IW * 32 + (X3 ? 14 * 64 * 16 : 0);   // hi + lo rings (+ lo weights)
  int grid = 256; while (grid > 8 && grid > wk.nseg) grid -= 8;
  ConvArgsB aa = a; aa.diag = nullptr; aa.diag_mode = g_conv16_diag_mode;
  // the product kernel has no diagnostic branches; cwf_debug_conv16_mode (tools/) selects the ablation instantiation
  auto go = [&](auto D) {
    constexpr bool DG = decltype(D)::value;
    CWF_MAX_LDS_ONCE((&conv16s_kernel<X3, DG, IN16>));
    hipLaunchKernelGGL((conv16s_kernel<X3, DG, IN16>), dim3(grid), dim3(512), lds, st, aa, wk);
  };
  if (g_conv16_diag_mode && !IN16) go(std::true_type{}); else go(std::false_type{});
  CWF_LAUNCH_CHECK();
  return 0;
}

static unsigned long long* g_conv16_diag = nullptr;
extern "C" void cwf_debug_conv16_diag(unsigned long long* buf) { g_conv16_diag = buf; }
extern "C" void cwf_debug_conv16_mode(int m) { g_conv16_diag_mode = m; }   // diagnostic builds: 1 = no stores, 2 = no loads   // [256][8][4] u64, or NULL = off

template <bool X3, bool DIAG>
static int launch_conv16_impl(ConvArgsB a, hipStream_t st) {
  const ConvGeom& g = a.g;
  const int total = g.N * g.tiles_d * g.tiles_h * g.tiles_w;
  const size_t lds = (size_t)2 * C16_NVOX * 16 * sizeof(unsigned short) * (X3 ? 2 : 1) + (X3 ? 14 * 64 * 16 : 0);
  CWF_MAX_LDS_ONCE((&conv16_kernel<X3, DIAG>));
  a.diag = DIAG ? g_conv16_diag : nullptr; a.diag_mode = g_conv16_diag_mode;
  int grid = 256; while (grid > 8 && grid > total) grid -= 8;   // one 8-wave workgroup per CU; multiple of 8 (XCD map)
  hipLaunchKernelGGL((conv16_kernel<X3, DIAG>), dim3(grid), dim3(512), lds, st, a, total);
  CWF_LAUNCH_CHECK();
  return 0;
}
template <bool X3>
static int launch_conv16(const ConvArgsB& a, hipStream_t st) {
  return g_conv16_diag ? launch_conv16_impl<X3, true>(a, st) : launch_conv16_impl<X3, false>(a, st);
}

// ---------------------------------------------------------------------------------------------------
// Pointwise family: 1x1x1 convs (forward and data gradient, CWF_CONV1) and the ConvTranspose k = 2, s = 2 forward (CWF_CONVT2:
// eight 1x1x1 convs, one per output parity class, of the SAME input voxel).  An output voxel depends on ONE input voxel: there is
// no halo and no reuse across taps, so staging through LDS (what conv_bf16_kernel does for every op) only costs -- these layers
// are pure streams (0.3-8 FLOP/B) and ran at 17-29 % of the HBM rate in the generic kernel (1x1 32->16 @128^3: 346 us for
// 805 MB; tools/layer_table.py).  Here a wave streams 16-voxel groups straight from global memory INTO MFMA fragments:
//   * the MFMA runs as  D[co][voxel] = W[co][ci] * X[ci][voxel]  (weights are the A operand): lane (r, kq) then needs 8
//     consecutive input channels of voxel r -- two float4 loads, a wave reads 16 voxels x 128 B contiguous -- and holds four
//     consecutive OUTPUT channels of voxel r, one float4 store (the generic kernel's D layout gives 16 dword stores);
//   * bf16 (hi, lo) split, bias / residual / statistics / norm-backward statistics epilogue all in registers; the loads of G = 2..4
//     groups (x, residual, nb_x) are issued together, then the groups are computed and stored;
//   * weights: read from the SAME packed buffer as the generic kernel (block (chunk, tile): lane (kq, r) = channels
//     chunk*16 + (kq&1)*8.., column r), permuted by index into the A layout; in registers for one class, in LDS for eight;
//   * a workgroup stays inside one sample: per-sample parameters are loaded once and the statistics leave through LDS as ONE
//     coalesced fp64 atomic instruction per workgroup; no LDS traffic per voxel, no barrier in the loop; 64-140 VGPRs.
// Semantics identical to conv_bf16_kernel for what it takes (v = acc + bias + residual; statistics of v or the norm-backward
// pair); launches with a normalising prologue or an out_scale stay on the generic kernel (pw_eligible).
// ---------------------------------------------------------------------------------------------------
struct PwWork { int Vin, gps, wps, gpw; };              // input voxels per sample, 16-voxel groups per sample, workgroups per sample, groups per wave

template <bool X3, int KS, int NT, int NCLS, int EPI>
__global__ __launch_bounds__(256) void pw_conv_kernel(const ConvArgsB a, const PwWork wk) {
  constexpr bool HAS_RES = (EPI & 1) != 0, HAS_NB = (EPI & 2) != 0;
  constexpr int G = (KS * NT <= 1) ? 4 : 2;              // groups per iteration: all their loads are in flight together
  extern __shared__ float4 lds4[];
  __shared__ float red[4][NT * 16][2];
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = (int)blockIdx.x / wk.wps, wg = (int)blockIdx.x % wk.wps;      // a workgroup stays inside ONE sample
  const int gA = (wg * 4 + wave) * wk.gpw;
  const int gB = min(gA + wk.gpw, wk.gps);

  // ---- weights
  uint4 wh[NCLS == 1 ? KS : 1][NT], wl[NCLS == 1 ? KS : 1][NT];
  const int wlane = ((kq & 1) * 16 + r) * 2;              // uint4 index of this lane's (hi, lo) pair inside a packed block
  if (NCLS == 1) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int chunk = 2 * s + (kq >> 1);
        wh[s][t] = make_uint4(0, 0, 0, 0); wl[s][t] = make_uint4(0, 0, 0, 0);
        if (chunk < g.nchunks) {
          const uint4* p = a.wpk + ((int64_t)chunk * g.ntiles + t) * 128 + wlane;
          wh[s][t] = p[0];
          if (X3) wl[s][t] = p[1];
        }
      }
  } else {
    // all eight classes' packed blocks, verbatim ([cls][chunk][tile][lane][hi|lo]), into LDS
    const int n4 = NCLS * g.nchunks * g.ntiles * 128;
    for (int i = tid; i < n4; i += 256) lds4[i] = reinterpret_cast<const float4*>(a.wpk)[i];
    __syncthreads();
  }

  // ---- per-sample bases and parameters
  const float* xs = a.x + (int64_t)n * wk.Vin * g.x_ldc;
  const float* rs = HAS_RES ? a.residual + (int64_t)n * wk.Vin * a.r_ldc : nullptr;
  const float* ns = HAS_NB ? a.nb_x + (int64_t)n * wk.Vin * a.nb_ldc : nullptr;
  float* ys = a.y + (int64_t)n * g.Do * g.Ho * g.Wo * g.y_ldc;
  f32x4 nsc4[NT], nsh4[NT], bias4[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int cb = t * 16 + kq * 4;
    bias4[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.bias && cb < g.Cout) bias4[t] = *reinterpret_cast<const f32x4*>(a.bias + cb);
    if (HAS_NB && cb < g.Cout) {
      nsc4[t] = *reinterpret_cast<const f32x4*>(a.nb_scale + (int64_t)n * g.Cout + cb);
      nsh4[t] = *reinterpret_cast<const f32x4*>(a.nb_shift + (int64_t)n * g.Cout + cb);
    }
  }
  f32x4 s1[NT], s2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { s1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; s2[t] = s1[t]; }
  const bool has_stats = a.stats != nullptr;

  for (int gi = gA; gi < gB; gi += G) {
    // ---- loads of G groups: x (8 input channels per K-step), residual / nb_x (4 output channels per tile)
    f32x4 xv[G][KS][2], rv[HAS_RES ? G : 1][NT], nv[HAS_NB ? G : 1][NT];
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const bool ok = gi + k < gB;                        // wave-uniform
      const int vox = (gi + k) * 16 + r;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int c0 = s * 32 + kq * 8;
        xv[k][s][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; xv[k][s][1] = xv[k][s][0];
        if (ok && c0 < g.Cin) xv[k][s][0] = *reinterpret_cast<const f32x4*>(xs + (int64_t)vox * g.x_ldc + c0);
        if (ok && c0 + 4 < g.Cin) xv[k][s][1] = *reinterpret_cast<const f32x4*>(xs + (int64_t)vox * g.x_ldc + c0 + 4);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int cb = t * 16 + kq * 4;
        if (HAS_RES && ok && cb < g.Cout) rv[k][t] = *reinterpret_cast<const f32x4*>(rs + (int64_t)vox * a.r_ldc + cb);
        if (HAS_NB && ok && cb < g.Cout) nv[k][t] = *reinterpret_cast<const f32x4*>(ns + (int64_t)vox * a.nb_ldc + cb);
      }
    }
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const bool ok = gi + k < gB;
      const int v0 = (gi + k) * 16;
      // ---- B fragments: x -> bf16 hi (, lo)   (no layer of this family has a normalising prologue: pw_eligible)
      uint4 bh[KS], bl[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        unsigned h[4], l[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v0_ = xv[k][s][j >> 1][(2 * j) & 3], v1_ = xv[k][s][j >> 1][(2 * j + 1) & 3];
          if (X3) split_bf16(v0_, v1_, h[j], l[j]);
          else h[j] = pack_bf16(v0_, v1_);
        }
        bh[s] = make_uint4(h[0], h[1], h[2], h[3]); bl[s] = make_uint4(l[0], l[1], l[2], l[3]);
      }
      int od = 0, oh = 0, ow = 0;
      if (NCLS > 1) { ow = v0 % g.Wi; const int t2 = v0 / g.Wi; oh = t2 % g.Hi; od = t2 / g.Hi; }
#pragma unroll 1
      for (int cls = 0; cls < NCLS; ++cls) {
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t] = bias4[t];
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            uint4 ah, al;
            if (NCLS == 1) { ah = wh[s][t]; al = wl[s][t]; }
            else {
              const int chunk = 2 * s + (kq >> 1);
              ah = make_uint4(0, 0, 0, 0); al = ah;
              if (chunk < g.nchunks) {
                const uint4* p = reinterpret_cast<const uint4*>(lds4) + ((cls * g.nchunks + chunk) * g.ntiles + t) * 128 + wlane;
                ah = p[0];
                if (X3) al = p[1];
              }
            }
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh[s]), acc[t], 0, 0, 0);
            if (X3) {
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl[s]), acc[t], 0, 0, 0);
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh[s]), acc[t], 0, 0, 0);
            }
          }
        }
        // ---- epilogue: lane (r, kq) holds output channels t*16 + kq*4 .. +3 of voxel r
        int64_t ovox;
        if (NCLS == 1) ovox = v0 + r;
        else ovox = ((int64_t)(2 * od + (cls >> 2)) * g.Ho + 2 * oh + ((cls >> 1) & 1)) * g.Wo + 2 * (ow + r) + (cls & 1);
        float* yp = ys + ovox * g.y_ldc + kq * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int cb = t * 16 + kq * 4;
          if (ok && cb < g.Cout) {
            f32x4 v = acc[t];
            if (HAS_RES) v += rv[HAS_RES ? k : 0][t];
            *reinterpret_cast<f32x4*>(yp + t * 16) = v;
            if (NCLS == 1 && a.y16)
              *reinterpret_cast<uint2*>(a.y16 + ((int64_t)n * g.Do * g.Ho * g.Wo + ovox) * g.Cout + cb) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
            if (HAS_NB) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float h = fmaf(nv[HAS_NB ? k : 0][t][i], nsc4[t][i], nsh4[t][i]);
                const float gn = v[i] * (h > 0.f ? 1.f : a.nb_slope);
                s1[t][i] += gn; s2[t][i] = fmaf(gn, h, s2[t][i]);
              }
            } else if (has_stats) {
#pragma unroll
              for (int i = 0; i < 4; ++i) { s1[t][i] += v[i]; s2[t][i] = fmaf(v[i], v[i], s2[t][i]); }
            }
          }
        }
      }
    }
  }

  // ---- statistics: lanes -> wave (shuffles over r) -> workgroup (LDS) -> ONE coalesced fp64 atomic instruction per workgroup
  // (an atomic costs per instruction, not per lane: thousands of 4-lane atomics on the same two cache lines serialise in L2)
  if (has_stats) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float u1 = s1[t][i], u2 = s2[t][i];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { u1 += __shfl_xor(u1, o, 64); u2 += __shfl_xor(u2, o, 64); }
        if (r == 0) { red[wave][t * 16 + kq * 4 + i][0] = u1; red[wave][t * 16 + kq * 4 + i][1] = u2; }
      }
    __syncthreads();
    if (tid < NT * 16 * 2) {
      const int co = tid >> 1, which = tid & 1;
      const double s = (double)red[0][co][which] + (double)red[1][co][which] + (double)red[2][co][which] + (double)red[3][co][which];
      if (co < g.Cout) atomic_add_f64(a.stats + ((int64_t)n * g.Cout + co) * 2 + which, s);
    }
  }
}

template <bool X3, int KS, int NT, int NCLS, int EPI>
static int launch_pw(const ConvArgsB& a, hipStream_t st) {
  const ConvGeom& g = a.g;
  constexpr int G = (KS * NT <= 1) ? 4 : 2;
  PwWork wk;
  wk.Vin = g.Di * g.Hi * g.Wi;
  wk.gps = wk.Vin >> 4;
  // ~2048 workgroups in all (8 per CU), each inside one sample; a wave takes a contiguous run of groups, a multiple of G
  int wps = 2048 / g.N; if (wps < 1) wps = 1;
  const int maxw = cdiv(wk.gps, 4 * G);
  if (wps > maxw) wps = maxw;
  wk.gpw = cdiv(cdiv(wk.gps, wps * 4), G) * G;
  wk.wps = cdiv(wk.gps, wk.gpw * 4);
  const size_t lds = NCLS == 1 ? 0 : (size_t)NCLS * g.nchunks * g.ntiles * 2048;
  hipLaunchKernelGGL((pw_conv_kernel<X3, KS, NT, NCLS, EPI>), dim3((unsigned)(wk.wps * g.N)), dim3(256), lds, st, a, wk);
  CWF_LAUNCH_CHECK();
  return 0;
}

// Is (op, shape, epilogue) one the pointwise kernel takes?  (anything else: the generic kernel)
static bool pw_eligible(int op, const ConvArgsB& a, int* ks, int* nt) {
  const ConvGeom& g = a.g;
  if (op != CWF_CONV1 && op != CWF_CONVT2) return false;
  static const bool off = getenv("CWF_NO_POINTWISE") != nullptr;          // A/B switch (diagnostics)
  if (off) return false;
  const int64_t Vin = (int64_t)g.Di * g.Hi * g.Wi;
  if ((Vin & 15) || Vin * (g.x_ldc > g.y_ldc * 8 ? g.x_ldc : g.y_ldc * 8) >= (1ll << 31)) return false;
  if ((g.Cout & 3) || (g.y_ldc & 3) || ((uintptr_t)a.y & 15)) return false;
  if (a.bias && ((uintptr_t)a.bias & 15)) return false;
  if (a.in_scale || a.out_scale) return false;             // (no 1x1x1 / transposed layer of the model has either)
  if (a.residual && ((a.r_ldc & 3) || ((uintptr_t)a.residual & 15))) return false;
  if (a.nb_x && ((a.nb_ldc & 3) || ((uintptr_t)a.nb_x & 15) || ((uintptr_t)a.nb_scale & 15) || ((uintptr_t)a.nb_shift & 15))) return false;
  *ks = cdiv(g.Cin, 32); *nt = cdiv(g.Cout, 16);
  if (*nt != g.ntiles) return false;
  const bool shape_ok = (*ks == 1 && (*nt == 1 || *nt == 2 || *nt == 4)) || (*ks == 2 && *nt == 2);
  if (!shape_ok) return false;
  if (op == CWF_CONVT2) {
    if ((g.Wi & 15) || a.residual || a.nb_x || *nt > 2 || *ks > 1) return false;
  }
  return true;
}

template <bool X3>
static int dispatch_pw(int op, const ConvArgsB& a, hipStream_t st, int ks, int nt) {
  if (op == CWF_CONVT2) {
    if (nt == 1) return launch_pw<X3, 1, 1, 8, 0>(a, st);
    return launch_pw<X3, 1, 2, 8, 0>(a, st);
  }
  const int epi = (a.residual ? 1 : 0) | (a.nb_x ? 2 : 0);
#define CWF_PW(k, t) if (ks == k && nt == t) { \
    switch (epi) { case 0: return launch_pw<X3, k, t, 1, 0>(a, st); case 1: return launch_pw<X3, k, t, 1, 1>(a, st); \
                   case 2: return launch_pw<X3, k, t, 1, 2>(a, st); default: return launch_pw<X3, k, t, 1, 3>(a, st); } }
  CWF_PW(1, 1) CWF_PW(1, 2) CWF_PW(1, 4) CWF_PW(2, 2)
#undef CWF_PW
  return CWF_E_BADARG;
}

// ---------------------------------------------------------------------------------------------------
// weight packing: dst[(i/8)*16 + i%8] = hi(src[map[i]]), dst[(i/8)*16 + 8 + i%8] = lo(...)   (bf16, zeros where map < 0)
// ---------------------------------------------------------------------------------------------------
__global__ void gather_split_bf16_kernel(const cwf_gather_desc* __restrict__ table) {
  const cwf_gather_desc d = table[blockIdx.y];
  unsigned short* dst = reinterpret_cast<unsigned short*>(d.dst);
  // a thread takes EIGHT consecutive entries = one [hi 8 | lo 8] group of the packed operand: two 16-byte map loads, eight gathered
  // source reads, two 16-byte stores (one entry per thread wrote two scattered 2-byte stores: 166 us per step for the 77 layers)
  const int64_t ngrp = d.n >> 3;
  const bool vec = ((d.n & 7) == 0) && ((reinterpret_cast<uintptr_t>(d.map) & 15) == 0) && ((reinterpret_cast<uintptr_t>(d.dst) & 15) == 0);
  if (vec) {
    for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < ngrp; gi += (int64_t)gridDim.x * blockDim.x) {
      const int4 m0 = *reinterpret_cast<const int4*>(d.map + gi * 8), m1 = *reinterpret_cast<const int4*>(d.map + gi * 8 + 4);
      const int mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
      bool other = false;
#pragma unroll
      for (int e = 0; e < 8; ++e) other |= mm[e] == -2;
      unsigned short hh[8], ll[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = mm[e] >= 0 ? d.src[mm[e]] : 0.f;
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        hh[e] = __builtin_bit_cast(unsigned short, h); ll[e] = __builtin_bit_cast(unsigned short, l);
      }
      unsigned short* o = dst + gi * 16;
      if (!other) {
        uint4 H, L;
        H.x = hh[0] | ((unsigned)hh[1] << 16); H.y = hh[2] | ((unsigned)hh[3] << 16); H.z = hh[4] | ((unsigned)hh[5] << 16); H.w = hh[6] | ((unsigned)hh[7] << 16);
        L.x = ll[0] | ((unsigned)ll[1] << 16); L.y = ll[2] | ((unsigned)ll[3] << 16); L.z = ll[4] | ((unsigned)ll[5] << 16); L.w = ll[6] | ((unsigned)ll[7] << 16);
        *reinterpret_cast<uint4*>(o) = H;
        *reinterpret_cast<uint4*>(o + 8) = L;
      } else {                                     // a fused layer's group with entries owned by another source: those stay
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (mm[e] != -2) { o[e] = hh[e]; o[e + 8] = ll[e]; }
      }
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t m = d.map[i];
    if (m == -2) continue;                       // owned by another source of a fused layer
    const float v = m >= 0 ? d.src[m] : 0.f;
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    const int64_t o = (i >> 3) * 16 + (i & 7);
    dst[o] = __builtin_bit_cast(unsigned short, h);
    dst[o + 8] = __builtin_bit_cast(unsigned short, l);
  }
}

extern "C" int cwf_gather_split_bf16(const struct cwf_gather_desc* table, int nlayers, int64_t max_n, void* stream) {
  if (!table || nlayers <= 0 || max_n <= 0) return CWF_E_BADARG;
  int64_t gx = cdiv64(max_n, 8 * 256); if (gx > 256) gx = 256; if (gx < 1) gx = 1;      // (a thread packs eight entries)
  hipLaunchKernelGGL(gather_split_bf16_kernel, dim3((unsigned)gx, nlayers), dim3(256), 0, cwf_stream(stream), table);
  CWF_LAUNCH_CHECK();
  return 0;
}

namespace {
struct TileCfg { int MT, NT, WM; };
// same heuristic as conv_mfma.hip (kept local: the two files are independent translation units)
TileCfg choose_cfg(int op, const int cdims[3], int ncls, int N, int ntiles) {
  const bool s2 = (op == CWF_CONV3_S2 || op == CWF_CONVT2_DGRAD);
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

template <int MT, int NT, int WM, bool X3>
int launch_cfg(const ConvArgsB& a, hipStream_t st) {
  constexpr int WN = 4 / WM;
  const ConvGeom& g = a.g;
  const size_t lds_tile = (size_t)g.ID * g.IH * g.IW * 16 * sizeof(unsigned short) * (X3 ? 2 : 1);
  const size_t lds_red = (size_t)4 * NT * 16 * 2 * sizeof(float);
  const size_t lds = lds_tile > lds_red ? lds_tile : lds_red;
  if (lds > 160 * 1024) return CWF_E_TOOLARGE;
  CWF_MAX_LDS_ONCE((&conv_bf16_kernel<MT, NT, WM, X3>));
  dim3 grid(g.tiles_d * g.tiles_h * g.tiles_w, cdiv(g.ntiles, WN * NT), g.N * g.ncls * (a.groups ? a.groups : 1));
  hipLaunchKernelGGL((conv_bf16_kernel<MT, NT, WM, X3>), grid, dim3(256), lds, st, a);
  CWF_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int cwf_conv_mfma_bf16_nb(int op, int x3, const float* x, int x_ldc, const void* wpk16, const float* bias,
                                     float* y, int y_ldc, const float* in_scale, const float* in_shift, float in_slope,
                                     const float* residual, int r_ldc, const float* out_scale, double* stats,
                                     const float* nb_x, int nb_ldc, const float* nb_scale, const float* nb_shift, float nb_slope,
                                     int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, void* stream);

extern "C" int cwf_conv_mfma_bf16(int op, int x3, const float* x, int x_ldc, const void* wpk16, const float* bias,
                                  float* y, int y_ldc, const float* in_scale, const float* in_shift, float in_slope,
                                  const float* residual, int r_ldc, const float* out_scale, double* stats,
                                  int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, void* stream) {
  return cwf_conv_mfma_bf16_nb(op, x3, x, x_ldc, wpk16, bias, y, y_ldc, in_scale, in_shift, in_slope, residual, r_ldc, out_scale, stats,
                               nullptr, 0, nullptr, nullptr, 1.f, N, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, stream);
}

static int conv_bf16_impl(int op, int x3, const float* x, int x_ldc, const void* wpk16, const float* bias,
                          float* y, int y_ldc, const float* in_scale, const float* in_shift, float in_slope,
                          const float* residual, int r_ldc, const float* out_scale, double* stats,
                          const float* nb_x, int nb_ldc, const float* nb_scale, const float* nb_shift, float nb_slope,
                          int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, void* stream,
                          const void* x16, const void* zero16, void* y16 = nullptr);

extern "C" int cwf_conv_mfma_bf16_nb(int op, int x3, const float* x, int x_ldc, const void* wpk16, const float* bias,
                                     float* y, int y_ldc, const float* in_scale, const float* in_shift, float in_slope,
                                     const float* residual, int r_ldc, const float* out_scale, double* stats,
                                     const float* nb_x, int nb_ldc, const float* nb_scale, const float* nb_shift, float nb_slope,
                                     int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, void* stream) {
  return conv_bf16_impl(op, x3, x, x_ldc, wpk16, bias, y, y_ldc, in_scale, in_shift, in_slope, residual, r_ldc, out_scale, stats,
                        nb_x, nb_ldc, nb_scale, nb_shift, nb_slope, N, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, stream, nullptr, nullptr);
}

// cwf_conv_mfma_bf16 for a 1x1x1 conv that also writes its output as a bf16 image y16 [N][Do*Ho*Wo][Cout] (DeUp_Cat.conv3, cls_wise_former.py:
// 716-729: its output is the un-normalised input of the next block's first conv, whose weight gradient reads that image).  CWF_E_BADARG
// if the layer is not one the pointwise stream kernel takes (the caller then converts with cwf_to_bf16).
extern "C" int cwf_conv_mfma_bf16_y16(int op, int x3, const float* x, int x_ldc, const void* wpk16, const float* bias,
                                      float* y, int y_ldc, void* y16, const float* in_scale, const float* in_shift, float in_slope,
                                      const float* residual, int r_ldc, double* stats,
                                      int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, void* stream) {
  if (!y16) return CWF_E_BADARG;
  return conv_bf16_impl(op, x3, x, x_ldc, wpk16, bias, y, y_ldc, in_scale, in_shift, in_slope, residual, r_ldc, nullptr, stats,
                        nullptr, 0, nullptr, nullptr, 1.f, N, Di, Hi, Wi, Cin, Do, Ho, Wo, Cout, stream, nullptr, nullptr, y16);
}

// The same conv with its INPUT given as a bf16 image x16 [N][Di][Hi][Wi][16] (single-bf16 operand launches of the 3x3x3 stride-1
// 16 -> 16 full-resolution layers without prologue, i.e. their data gradients: x16 = the bf16 image of dy that the producer of dy wrote,
// cwf_in_bwd_apply_ex).  x (fp32) is not read and may be NULL.  zero16: 16 zero bytes (padding source of the LDS-DMA loaders).
extern "C" int cwf_conv_mfma_bf16_in16(int op, const void* x16, const void* zero16, const void* wpk16, const float* bias,
                                       float* y, int y_ldc, const float* residual, int r_ldc, double* stats,
                                       const float* nb_x, int nb_ldc, const float* nb_scale, const float* nb_shift, float nb_slope,
                                       int N, int D, int H, int W, void* stream) {
  if (!x16 || !zero16 || ((uintptr_t)x16 & 15) || ((uintptr_t)zero16 & 15)) return CWF_E_BADARG;
  if (op != CWF_CONV3_S1 || (int64_t)D * H * W < 32768 || (int64_t)N * D * H * W >= (1ll << 30)) return CWF_E_BADARG;
  return conv_bf16_impl(op, 0, reinterpret_cast<const float*>(x16), 16, wpk16, bias, y, y_ldc, nullptr, nullptr, 1.f, residual, r_ldc, nullptr, stats,
                        nb_x, nb_ldc, nb_scale, nb_shift, nb_slope, N, D, H, W, 16, D, H, W, 16, stream, x16, zero16);
}

static int conv_bf16_impl(int op, int x3, const float* x, int x_ldc, const void* wpk16, const float* bias,
                          float* y, int y_ldc, const float* in_scale, const float* in_shift, float in_slope,
                          const float* residual, int r_ldc, const float* out_scale, double* stats,
                          const float* nb_x, int nb_ldc, const float* nb_scale, const float* nb_shift, float nb_slope,
                          int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, void* stream,
                          const void* x16, const void* zero16, void* y16) {
  if (!x || !wpk16 || !y || N <= 0 || Cin <= 0 || Cout <= 0) return CWF_E_BADARG;
  if (nb_x && (!stats || !nb_scale || !nb_shift || nb_ldc < Cout)) return CWF_E_BADARG;
  if ((Cin & 3) || (x_ldc & 3) || x_ldc < Cin || y_ldc < Cout) return CWF_E_ALIGN;
  if (((uintptr_t)x & 15) || ((uintptr_t)wpk16 & 15)) return CWF_E_ALIGN;
  if (in_scale && !in_shift) return CWF_E_BADARG;
  ConvArgsB a;
  int cd[3] = {Do, Ho, Wo}; int ncls = 1;
  if (op == CWF_CONVT2) { cd[0] = Di; cd[1] = Hi; cd[2] = Wi; ncls = 8; }
  if (op == CWF_CONV3_S2_DGRAD) { cd[0] = (Do + 1) / 2; cd[1] = (Ho + 1) / 2; cd[2] = (Wo + 1) / 2; ncls = 8; }
  TileCfg c = choose_cfg(op, cd, ncls, N, cdiv(Cout, 16));
  {                                                    // tuning aid: CWF_FORCE_CFG="MT,NT,WM" overrides the tile choice
    static const char* force = getenv("CWF_FORCE_CFG");
    int fm, fn, fw;
    if (force && sscanf(force, "%d,%d,%d", &fm, &fn, &fw) == 3) { c.MT = fm; c.NT = fn; c.WM = fw; }
  }
  int rc = cwf_build_geom(a.g, op, N, Di, Hi, Wi, Cin, x_ldc, Do, Ho, Wo, Cout, y_ldc, c.MT * c.WM);
  if (rc) return rc;
  a.x = x; a.wpk = reinterpret_cast<const uint4*>(wpk16); a.bias = bias; a.y = y; a.in_scale = in_scale; a.in_shift = in_shift;
  a.in_slope = in_slope; a.residual = residual; a.r_ldc = r_ldc; a.out_scale = out_scale; a.stats = stats;
  a.nb_x = nb_x; a.nb_ldc = nb_ldc; a.nb_scale = nb_scale; a.nb_shift = nb_shift; a.nb_slope = nb_slope;
  a.diag = nullptr; a.diag_mode = 0;
  a.groups = 0; a.x_goff = 0; a.y_goff = 0;
  a.x16 = reinterpret_cast<const uint4*>(x16); a.zero16 = reinterpret_cast<const uint4*>(zero16);
  a.y16 = reinterpret_cast<unsigned short*>(y16);
  for (int q = 0; q < 3; ++q) { a.wpk_g[q] = nullptr; a.bias_g[q] = nullptr; }
  hipStream_t st = cwf_stream(stream);
  if (x16) {                                           // (cwf_conv_mfma_bf16_in16 has checked the layer)
    int nat[27];
    for (int t = 0; t < 27; ++t) nat[t] = a.g.tapofs[t];
    for (int t = 0; t < 27; ++t) a.g.tapofs[t] = nat[c16_tap(t)];
    rc = cwf_build_geom(a.g, op, N, Di, Hi, Wi, Cin, x_ldc, Do, Ho, Wo, Cout, y_ldc, 16);
    if (rc) return rc;
    return launch_conv16s<false, true>(a, st);
  }
  {
    int ks, nt;
    if (pw_eligible(op, a, &ks, &nt)) {                   // 1x1x1 / ConvTranspose streams: no LDS staging (pw_conv_kernel)
      if (y16 && (op != CWF_CONV1 || (Cout & 3) || ((uintptr_t)y16 & 7))) return CWF_E_BADARG;
      return x3 ? dispatch_pw<true>(op, a, st, ks, nt) : dispatch_pw<false>(op, a, st, ks, nt);
    }
    if (y16) return CWF_E_BADARG;                        // the bf16 side output is the pointwise stream kernel's
  }
  {
    int rcw = 0;                                           // 32 / 64 / 128-channel 3x3x3 layers: weight-stationary kernel (conv_ws.hip)
    if (cwf_try_conv_ws(op, x3, a, st, &rcw)) return rcw;
  }
  if (op == CWF_CONV3_S1 && Cin <= 16 && Cout <= 16) {      // these layers are packed in conv16's tap order (c16_tap)
    int nat[27];
    for (int t = 0; t < 27; ++t) nat[t] = a.g.tapofs[t];
    for (int t = 0; t < 27; ++t) a.g.tapofs[t] = nat[c16_tap(t)];
  }
  if (op == CWF_CONV3_S1 && Cin <= 16 && Cout <= 16 && (int64_t)Do * Ho * Wo >= 32768) {
    // full-resolution 16-channel convs: the persistent register-resident-weight kernel (tile 4x4x16)
    rc = cwf_build_geom(a.g, op, N, Di, Hi, Wi, Cin, x_ldc, Do, Ho, Wo, Cout, y_ldc, 16);
    if (rc) return rc;
    // sliding-window kernel by default; CWF_CONV16_NOSLIDE=1 selects the double-buffered full-halo kernel (A/B, diagnostics)
    static const bool noslide = getenv("CWF_CONV16_NOSLIDE") != nullptr;
    if ((!noslide && !g_conv16_diag) || nb_x) return x3 ? launch_conv16s<true>(a, st) : launch_conv16s<false>(a, st);   // (conv16 has no nb epilogue)
    return x3 ? launch_conv16<true>(a, st) : launch_conv16<false>(a, st);
  }
#define CWF_CFG(mt, nt, wm) if (c.MT == mt && c.NT == nt && c.WM == wm) return x3 ? launch_cfg<mt, nt, wm, true>(a, st) : launch_cfg<mt, nt, wm, false>(a, st);
  CWF_CFG(4, 4, 1) CWF_CFG(2, 4, 2) CWF_CFG(2, 4, 4) CWF_CFG(4, 2, 4) CWF_CFG(4, 1, 4)
  CWF_CFG(1, 4, 4) CWF_CFG(1, 2, 4) CWF_CFG(1, 2, 2) CWF_CFG(1, 1, 4)
#undef CWF_CFG
  return CWF_E_BADARG;
}

// Channel-grouped 3x3x3 stride-1 conv (forward, or data gradient through transposed packed weights): `groups` (2 or 3) independent
// convs Cin -> Cout on channel groups of one input tensor, written into channel groups of one output tensor, ONE launch (the
// supervision heads of the three sub-regions: SuperviseLabel.py:58-81, EdgeSuperviseLabel.py:56-76 -- 24 tiny layers per step whose
// launches, not their arithmetic, are the cost).  Bias only: no normalising prologue, residual, out_scale or statistics.
extern "C" int cwf_conv_mfma_bf16_grouped(int op, int x3, const float* x, int x_ldc, int x_goff, const void* const* wpk16, const float* const* bias,
                                          float* y, int y_ldc, int y_goff, int groups,
                                          int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, void* stream) {
  if (!x || !wpk16 || !y || N <= 0 || Cin <= 0 || Cout <= 0 || groups < 2 || groups > 3) return CWF_E_BADARG;
  if (op != CWF_CONV3_S1) return CWF_E_BADARG;
  if ((Cin & 3) || (x_ldc & 3) || (x_goff & 3) || x_ldc < (groups - 1) * x_goff + Cin || y_ldc < (groups - 1) * y_goff + Cout) return CWF_E_ALIGN;
  if ((uintptr_t)x & 15) return CWF_E_ALIGN;
  ConvArgsB a;
  int cd[3] = {Do, Ho, Wo};
  TileCfg c = choose_cfg(op, cd, 1, N * groups, cdiv(Cout, 16));
  int rc = cwf_build_geom(a.g, op, N, Di, Hi, Wi, Cin, x_ldc, Do, Ho, Wo, Cout, y_ldc, c.MT * c.WM);
  if (rc) return rc;
  a.x = x; a.wpk = nullptr; a.bias = nullptr; a.y = y; a.in_scale = nullptr; a.in_shift = nullptr; a.in_slope = 1.f;
  a.residual = nullptr; a.r_ldc = 0; a.out_scale = nullptr; a.stats = nullptr;
  a.nb_x = nullptr; a.nb_ldc = 0; a.nb_scale = nullptr; a.nb_shift = nullptr; a.nb_slope = 1.f;
  a.diag = nullptr; a.diag_mode = 0;
  a.groups = groups; a.x_goff = x_goff; a.y_goff = y_goff;
  a.x16 = nullptr; a.zero16 = nullptr; a.y16 = nullptr;
  for (int q = 0; q < 3; ++q) {
    a.wpk_g[q] = q < groups ? reinterpret_cast<const uint4*>(wpk16[q]) : nullptr;
    a.bias_g[q] = (q < groups && bias) ? bias[q] : nullptr;
    if (q < groups && (!wpk16[q] || ((uintptr_t)wpk16[q] & 15))) return CWF_E_ALIGN;
  }
  if (Cin <= 16 && Cout <= 16) {                           // these layers are packed in conv16's tap order (c16_tap)
    int nat[27];
    for (int t = 0; t < 27; ++t) nat[t] = a.g.tapofs[t];
    for (int t = 0; t < 27; ++t) a.g.tapofs[t] = nat[c16_tap(t)];
  }
  hipStream_t st = cwf_stream(stream);
#define CWF_CFG(mt, nt, wm) if (c.MT == mt && c.NT == nt && c.WM == wm) return x3 ? launch_cfg<mt, nt, wm, true>(a, st) : launch_cfg<mt, nt, wm, false>(a, st);
  CWF_CFG(4, 4, 1) CWF_CFG(2, 4, 2) CWF_CFG(2, 4, 4) CWF_CFG(4, 2, 4) CWF_CFG(4, 1, 4)
  CWF_CFG(1, 4, 4) CWF_CFG(1, 2, 4) CWF_CFG(1, 2, 2) CWF_CFG(1, 1, 4)
#undef CWF_CFG
  return CWF_E_BADARG;
}
