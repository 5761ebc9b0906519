// Shared device/host helpers for libcwf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cwf_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CWF_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

// Raise a kernel's dynamic-LDS limit to the full 160 KiB, once per DEVICE (the attribute is per device; the flag used to be per
// process, so a second device used from the same process kept the default 64 KiB limit and its launches failed).
#define CWF_MAX_LDS_ONCE(fn)                                                                                              \
  do {                                                                                                                    \
    static bool done_[64] = {};                                                                                           \
    int dev_ = 0;                                                                                                         \
    (void)hipGetDevice(&dev_);                                                                                            \
    if (dev_ < 0 || dev_ >= 64 || !done_[dev_]) {                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (dev_ >= 0 && dev_ < 64) done_[dev_] = true;                                                                     \
    }                                                                                                                     \
  } while (0)

static inline hipStream_t cwf_stream(void* s) { return (hipStream_t)s; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float cwf_act(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float cwf_act_grad(float v, float slope) { return v > 0.f ? 1.f : slope; }

// 64-lane butterfly sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Counter-based uniform in [0,1): splitmix64 of (seed, element counter).  Dropout masks are never materialised: every kernel
// that applies one recomputes keep(ctr) from the element's index, forward and backward alike.
__device__ __forceinline__ float cwf_u01(uint64_t seed, uint64_t ctr) {
  uint64_t z = (seed ^ 0x9E3779B97F4A7C15ull) + ctr * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
// Device-resident generator state {seed, step}.  `step` is advanced by a KERNEL (cwf_rng_advance) once per training step, so it
// also advances on every replay of a captured hipGraph; per-site counters offsets are static within a step.
__device__ __forceinline__ float cwf_rng_u01(const uint64_t* __restrict__ rng, uint64_t ctr) {
  return cwf_u01(rng[0] + rng[1] * 0xD6E8FEB86659FD93ull, ctr);
}
// pre-scaled keep factor of one (p) or two chained (p, p2) dropouts of the same tensor of n elements
__device__ __forceinline__ float cwf_keep(const uint64_t* __restrict__ rng, uint64_t off, uint64_t i, uint64_t n, float p, float p2) {
  float v = cwf_rng_u01(rng, off + i) >= p ? 1.0f / (1.0f - p) : 0.f;
  if (p2 > 0.f) v *= cwf_rng_u01(rng, off + n + i) >= p2 ? 1.0f / (1.0f - p2) : 0.f;
  return v;
}

// hardware f64 / f32 atomic add (no CAS loop)
__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ---------------------------------------------------------------------------------------------------
// Geometry shared by the forward-like conv kernel (conv_mfma.hip) and the weight-gradient kernel
// (wgrad_mfma.hip).  One "class" = one set of taps writing one parity of the output grid.
// ---------------------------------------------------------------------------------------------------
struct ConvGeom {
  int N, Di, Hi, Wi, Cin, x_ldc;
  int Do, Ho, Wo, Cout, y_ldc;
  int is, os;              // input / output stride
  int ncls;                // 1 or 8
  int TD, TH;              // spatial tile in M-tiles (TW = 16 voxels)
  int ID, IH, IW;          // LDS input tile extents (voxels)
  int lo[3];               // min tap offset per dim
  int nchunks;             // ceil(Cin/16)
  int ntiles;              // ceil(Cout/16)
  int tiles_d, tiles_h, tiles_w;
  int cls_ntaps[8];
  int cls_ooff[8][3];
  int cls_dims[8][3];      // class grid extents (Dc,Hc,Wc)
  int cls_wbase[8];        // offset of the class in the packed weights, in 256-float blocks
  int cls_wbase16[8];      // same for the split-bf16 packing, in blocks of 64 lanes x 32 B (tap PAIRS)
  int tapofs[64];          // LDS voxel offset of tap t of class c at [c*8+t] (ncls==8) or [t] (ncls==1)
};

// Fill a ConvGeom for (op, dims).  MTOT = M-tiles (16 voxels each) per workgroup.  Returns 0 or CWF_E_*.
int cwf_build_geom(ConvGeom& g, int op, int N, int Di, int Hi, int Wi, int Cin, int x_ldc,
                   int Do, int Ho, int Wo, int Cout, int y_ldc, int MTOT);

// Stage one 16-channel chunk of the input halo tile into LDS as [voxel][16], applying the fused
// InstanceNorm + activation prologue; out-of-range voxels / channels are written as zeros (the
// reference pads the ACTIVATED tensor).  All 256 threads of the workgroup take part.
__device__ __forceinline__ void cwf_stage_input_tile(float* lds, const ConvGeom& g, const float* x,
                                                     const float* in_scale, const float* in_shift, float slope,
                                                     int n, int chunk, int id0, int ih0, int iw0, int tid) {
  const int q = tid & 3;
  const int c = chunk * 16 + q * 4;
  const bool cval = c < g.Cin;
  const bool has_norm = in_scale != nullptr;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has_norm && cval) {
    sc = *reinterpret_cast<const float4*>(in_scale + (int64_t)n * g.Cin + c);
    sh = *reinterpret_cast<const float4*>(in_shift + (int64_t)n * g.Cin + c);
  }
  const int nvox_in = g.ID * g.IH * g.IW;
  for (int v = tid >> 2; v < nvox_in; v += 64) {
    const int iw = v % g.IW; const int t2 = v / g.IW;
    const int ih = t2 % g.IH; const int idd = t2 / g.IH;
    const int gd = id0 + idd, gh = ih0 + ih, gw = iw0 + iw;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cval && gd >= 0 && gd < g.Di && gh >= 0 && gh < g.Hi && gw >= 0 && gw < g.Wi) {
      const int64_t off = ((((int64_t)n * g.Di + gd) * g.Hi + gh) * g.Wi + gw) * g.x_ldc + c;
      val = *reinterpret_cast<const float4*>(x + off);
      if (has_norm || slope != 1.f) {
        val.x = cwf_act(val.x * sc.x + sh.x, slope); val.y = cwf_act(val.y * sc.y + sh.y, slope);
        val.z = cwf_act(val.z * sc.z + sh.z, slope); val.w = cwf_act(val.w * sc.w + sh.w, slope);
      }
    }
    *reinterpret_cast<float4*>(lds + v * 16 + q * 4) = val;
  }
}
