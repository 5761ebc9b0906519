// Declarations shared by the split-bf16 conv kernels (conv_bf16.hip: tap-table kernel, conv16 / conv16s, pointwise streams;
// conv_ws.hip: weight-stationary kernel for the 32/64/128-channel 3x3x3 layers).
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ConvArgsB {
  ConvGeom g;
  const float* x; const uint4* wpk; const float* bias; float* y;
  const float* in_scale; const float* in_shift; float in_slope;
  const float* residual; int r_ldc; const float* out_scale; double* stats;
  unsigned long long* diag; int diag_mode;
  // "norm-backward" statistics (data-gradient launches whose output g feeds the backward of y = act(IN(x))): with nb_x set,
  // stats receives per (n, channel)  S1 = sum g*act'(h), S2 = sum g*act'(h)*h,  h = nb_x*nb_scale + nb_shift  -- what
  // cwf_in_bwd_stats would compute in a separate pass over g and x (norm.hip) -- instead of (sum y, sum y^2).
  const float* nb_x; int nb_ldc; const float* nb_scale; const float* nb_shift; float nb_slope;
  // channel-grouped launches (cwf_conv_mfma_bf16_grouped: the three sub-regions' supervision-head convs as one launch): group q reads
  // input channels [q*x_goff, q*x_goff + Cin) and writes output channels [q*y_goff, q*y_goff + Cout) of the same voxel rows, with its
  // own packed weights and bias; blockIdx.z = (group * N + n) * ncls + class.  groups == 0: an ordinary launch.
  int groups, x_goff, y_goff;
  const uint4* wpk_g[3]; const float* bias_g[3];
  // conv16s, IN16 instantiation: the input as a bf16 image (16-byte granules, two per voxel) and the 16-byte zero page of its loaders
  const uint4* x16; const uint4* zero16;
  // pointwise stream kernel (1x1x1 forward): the output also as a bf16 image [N][V][Cout] (the operand image of a consuming layer's weight
  // gradient, see cwf_conv_mfma_bf16_y16): one extra 8-byte store per lane
  unsigned short* y16;
};

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {          // one v_cvt_pk_bf16_f32
  const f32x2_t f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ float bf16_round(float a) { return (float)(__bf16)a; }
// split two floats into packed bf16 hi and lo (lo = bf16(v - hi)):  cvt_pk, shl/and, 2 sub, cvt_pk
__device__ __forceinline__ void split_bf16(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pack_bf16(a, b);
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  lo = pack_bf16(a - ha, b - hb);
}
// branch-free (Leaky)ReLU / identity for slope in [0,1]: max(v, slope*v)
__device__ __forceinline__ float act01(float v, float slope) { return fmaxf(v, v * slope); }


// conv_ws.hip: launches the weight-stationary kernel when the layer is one it takes (returns 1, status in *rc); 0 = not eligible.
int cwf_try_conv_ws(int op, int x3, ConvArgsB& a, hipStream_t st, int* rc);
