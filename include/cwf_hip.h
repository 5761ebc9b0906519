/*
 * cwf_hip.h -- C ABI of libcwf_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * ClsWiseFormer forward/backward hot path.
 *
 * The reference (mathwrx/Decouple-and-Couple_Learning_in_Multi-Modal_Brain_Tumor_Segmentation) is
 * pure Python over ATen; it owns no native code, so there is no reference FFI to mirror.  Each entry
 * point below replaces the ATen op sequence the cited reference lines dispatch, and is what the
 * reference-side ctypes stub in INTEGRATION.md binds.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless named h_*.
 *   - activations are fp32, channels-last: [N][D][H][W][C] ("NDHWC"); a tensor argument is
 *     (ptr, ldc) where ldc = floats between consecutive voxels (>= C, multiple of 4; ptr 16-B aligned).
 *     Channel slices / zero-copy concatenation are expressed by offsetting ptr and keeping ldc.
 *   - token matrices are row-major [B][T][E].
 *   - no allocation, no host synchronisation, no ownership transfer inside; work buffers are passed in.
 *   - every call enqueues on `stream` (a hipStream_t passed as void*) and returns 0 on success,
 *     a negative CWF_E_* for argument errors, or the positive hipError_t of the failed launch.
 *   - arithmetic: fp32 in / fp32 accumulate; contractions use v_mfma_f32_16x16x4_f32 (exact f32).
 */
#ifndef CWF_HIP_H
#define CWF_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CWF_E_BADARG   (-1)
#define CWF_E_TOOLARGE (-2)
#define CWF_E_ALIGN    (-3)

int cwf_version(void);            /* ABI version, bumped on any signature change */
const char* cwf_arch(void);       /* "gfx950" */

/* ------------------------------------------------------------------------------------------------
 * K1  implicit-GEMM 3-D convolution family on MFMA (fwd and data-gradient of every conv in the model)
 *
 * op selects the geometry; all share one tap-table driven kernel:
 *   CWF_CONV3_S1      3x3x3, stride 1, pad 1      nn.Conv3d(k=3,p=1)        Unet_skipconnection.py:26,42,46
 *   CWF_CONV3_S2      3x3x3, stride 2, pad 1      EnDown                     Unet_skipconnection.py:63
 *   CWF_CONV1         1x1x1                       down_channel/DeUp/endconv  cls_wise_former.py:623,719,721,642
 *   CWF_CONVT2        ConvTranspose3d k=2 s=2     DeUp_Cat.conv2             cls_wise_former.py:720
 *   CWF_CONV3_S2_DGRAD  data gradient of CWF_CONV3_S2 (x = dy at half res, y = dx at full res)
 *   CWF_CONVT2_DGRAD    data gradient of CWF_CONVT2   (x = dy at double res, y = dx)
 *   (the data gradients of CONV3_S1 / CONV1 are the same ops with flipped / transposed packed weights)
 *
 * y = out_scale[n,co] * ( sum_taps sum_ci act(x*in_scale[n,ci]+in_shift[n,ci]) * W + bias[co] + residual )
 *   act(v) = v > 0 ? v : in_slope*v   (in_slope 0 = ReLU, 0.01 = LeakyReLU, 1 = identity); zero padding is
 *   applied AFTER the activation (pads the activated tensor, as conv(relu(IN(x))) does).
 *   in_scale/in_shift, bias, residual, out_scale may be NULL.
 * stats (nullable): double [N][Cout][2], accumulates sum(y), sum(y*y) per (n, co)  -- the InstanceNorm
 *   statistics of the output, fused into the epilogue.  Must be zeroed by the caller.
 * wpk: weights packed by cwf_gather_batched with an index map built by the host (layout documented in
 *   cwf/packing.py: [class][ci_chunk16][tap][co_tile16][lane64][4]).
 * Input dims (Di,Hi,Wi,Cin), output dims (Do,Ho,Wo,Cout) are the tensor extents of x and y.
 * ---------------------------------------------------------------------------------------------- */
enum { CWF_CONV3_S1 = 0, CWF_CONV3_S2 = 1, CWF_CONV1 = 2, CWF_CONVT2 = 3, CWF_CONV3_S2_DGRAD = 4, CWF_CONVT2_DGRAD = 5 };

int cwf_conv_mfma(int op,
                  const float* x, int x_ldc, const float* wpk, const float* bias,
                  float* y, int y_ldc,
                  const float* in_scale, const float* in_shift, float in_slope,
                  const float* residual, int r_ldc, const float* out_scale, double* stats,
                  int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                  void* stream);

/* Weight gradient (+ bias gradient) of the same family (op in {CONV3_S1, CONV3_S2, CONV1, CONVT2}):
 *   dW[tap][ci][co] = sum_{n,vox} act(x*in_scale+in_shift)[n, vox*is+tap, ci] * dy[n, vox, co]
 * in two launches: cwf_wgrad_mfma writes `nsplit` partial slabs (MFMA accumulator layout) into `partial`
 * (cwf_wgrad_partial_floats() floats); cwf_wgrad_reduce sums the slabs in slab order and scatters through a host-built
 * INVERSE map (int32 [slab_floats]: >= 0 index into dW ([Cout][Cin][k][k][k] as nn.Conv3d.weight), <= -2 bias index
 * -2-v into db ([Cout], may be NULL), -1 padding).  The split count is chosen by the library: cwf_wgrad_nsplit().      */
int cwf_wgrad_nsplit(int op, int N, int Do, int Ho, int Wo, int Cin, int Cout);
int64_t cwf_wgrad_partial_floats(int op, int N, int Do, int Ho, int Wo, int Cin, int Cout);
int64_t cwf_wgrad_slab_floats(int op, int Cin, int Cout);
int cwf_wgrad_mfma(int op,
                   const float* x, int x_ldc, const float* in_scale, const float* in_shift, float in_slope,
                   const float* dy, int dy_ldc, float* partial,
                   int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                   void* stream);
int cwf_wgrad_reduce(const float* partial, int nsplit, int64_t slab_floats,
                     const int32_t* inv_map, float* dW, float* db, void* stream);
/* every layer of a backward phase in one launch: table = DEVICE array of descriptors (same slab / inverse-map conventions as
 * cwf_wgrad_reduce); dW / db may point into one flat gradient buffer */
struct cwf_wgrad_reduce_desc { const float* partial; const int32_t* inv; float* dW; float* db; int64_t slab; int32_t nsplit; int32_t pad_; };
int cwf_wgrad_reduce_batched(const struct cwf_wgrad_reduce_desc* table, int nlayers, void* stream);

struct cwf_gather_desc { const float* src; float* dst; const int32_t* map; int64_t n; };

/* The stem: y = (conv3x3x3(x; w) + bias) * out_scale for 4 -> 16 channels (InitConv + its always-on dropout3d, Unet_skipconnection.py:22-33)
 * with K = 8 taps x 4 channels per MFMA step; x [N][D][H][W][4] fp32 (ldc x_ldc), w the RAW nn.Conv3d weight [16][4][3][3][3], y ldc y_ldc;
 * out_scale [N][16] and stats [N][16][2] (sum, sum of squares of y) nullable.  x3: split-bf16 (3 MFMAs) / single bf16. */
int cwf_conv_stem_bf16(int x3, const float* x, int x_ldc, const float* w, const float* bias, float* y, int y_ldc,
                       const float* out_scale, double* stats, int N, int D, int H, int W, void* stream);

/* The first down-sampling layer: y = conv3x3x3 stride 2 (x; w) + bias for 16 -> 32 channels (EnDown1, Unet_skipconnection.py:60-68); x
 * [N][Di][Hi][Wi][16] fp32 (ldc x_ldc), w the RAW nn.Conv3d weight [32][16][3][3][3], y [N][(Di+1)/2][(Hi+1)/2][(Wi+1)/2][32] (ldc y_ldc);
 * stats [N][32][2] nullable. */
int cwf_conv_s2c16_bf16(int x3, const float* x, int x_ldc, const float* w, const float* bias, float* y, int y_ldc, double* stats,
                        int N, int Di, int Hi, int Wi, void* stream);

/* Split-bf16 forms of K1 (same geometry, epilogues and argument meaning; activations and weights stay fp32 in HBM):
 * MFMA operands are bf16 on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
 *   x3 != 0 ("bf16x3"): v = hi + lo per operand, products hi.hi + hi.lo + lo.hi  (~2^-16 relative per product, 3 MFMAs)
 *   x3 == 0 ("bf16")  : hi.hi only                                                 (2^-9 relative per product, 1 MFMA)
 * wpk16: weights packed by cwf_gather_split_bf16 ([class][ci_chunk16][tap pair][co_tile16][lane64][hi 8 | lo 8] bf16);
 * its index map has one int32 per bf16 element of the hi image (8 per lane), see cwf/packing.py.                       */
int cwf_conv_mfma_bf16(int op, int x3,
                       const float* x, int x_ldc, const void* wpk16, const float* bias,
                       float* y, int y_ldc,
                       const float* in_scale, const float* in_shift, float in_slope,
                       const float* residual, int r_ldc, const float* out_scale, double* stats,
                       int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                       void* stream);
/* Same, with "norm-backward" statistics for data-gradient launches whose output g is the gradient of act(IN(x)) (the fused
 * prologue of the forward conv, Unet_skipconnection.py:31-77 / cls_wise_former.py:157-204): with nb_x set, stats receives per
 * (n, channel)  S1 = sum g*act'(h), S2 = sum g*act'(h)*h  with h = nb_x*nb_scale + nb_shift -- exactly what cwf_in_bwd_stats
 * computes in a separate pass over g and x -- so that only cwf_in_bwd_apply remains of the InstanceNorm backward. */
int cwf_conv_mfma_bf16_nb(int op, int x3,
                          const float* x, int x_ldc, const void* wpk16, const float* bias,
                          float* y, int y_ldc,
                          const float* in_scale, const float* in_shift, float in_slope,
                          const float* residual, int r_ldc, const float* out_scale, double* stats,
                          const float* nb_x, int nb_ldc, const float* nb_scale, const float* nb_shift, float nb_slope,
                          int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                          void* stream);
/* `groups` (2 or 3) same-shape 3x3x3 stride-1 layers' weight-gradient slabs in ONE launch: group q has its own activation view
 * h_x[q] (row pitch x_ldc), gradient view h_dy[q] (row pitch dy_ldc) and slab buffer h_partial[q] (each sized like a single layer's,
 * cwf_wgrad_partial_floats); no prologue.  h_*: HOST arrays of device pointers.  Reduce each group like a single layer. */
int cwf_wgrad_mfma_bf16_grouped(int op, int x3, const float* const* h_x, int x_ldc, const float* const* h_dy, int dy_ldc,
                                float* const* h_partial, int groups,
                                int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int* nsplit_used, void* stream);
/* Channel-grouped 3x3x3 stride-1 conv (op = CWF_CONV3_S1; forward, or the data gradient through the transposed packed weights):
 * `groups` (2 or 3) independent convs Cin -> Cout, group q reading input channels [q*x_goff, q*x_goff + Cin) and writing output
 * channels [q*y_goff, q*y_goff + Cout) of the same voxel rows, each with its own packed weights / bias (h_wpk16, h_bias: HOST arrays
 * of `groups` device pointers; h_bias or its entries may be NULL) -- one launch for the three sub-regions' supervision-head convs
 * (SuperviseLabel.py:58-81, EdgeSuperviseLabel.py:56-76).  Bias only: no prologue, residual, out_scale or statistics. */
int cwf_conv_mfma_bf16_grouped(int op, int x3, const float* x, int x_ldc, int x_goff, const void* const* h_wpk16, const float* const* h_bias,
                               float* y, int y_ldc, int y_goff, int groups,
                               int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, void* stream);
int cwf_wgrad_mfma_bf16(int op, int x3,
                        const float* x, int x_ldc, const float* in_scale, const float* in_shift, float in_slope,
                        const float* dy, int dy_ldc, float* partial,
                        int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout,
                        int* nsplit_used /* host out, nullable: slabs actually written (<= cwf_wgrad_nsplit()) */, void* stream);
int cwf_gather_split_bf16(const struct cwf_gather_desc* table, int nlayers, int64_t max_n, void* stream);

/* dst[i] = map[i] >= 0 ? src[map[i]] : 0 for a table of `nlayers` descriptors resident in device memory
 * (struct cwf_gather_desc).  Used once per step to pack every layer's weights for K1.               */
int cwf_gather_batched(const struct cwf_gather_desc* table, int nlayers, int64_t max_n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K3  InstanceNorm3d (affine=False, eps)  nn.InstanceNorm3d -- Unet_skipconnection.py:13,39-45;
 *     cls_wise_former.py:207-223,697-700,737-742
 * ---------------------------------------------------------------------------------------------- */
/* stats double[NC][2] (sum, sumsq over V voxels) -> scale = rstd, shift = -mean*rstd  (float[NC] each) */
int cwf_in_finalize(const double* stats, float* scale, float* shift, int NC, int64_t V, float eps, void* stream);
/* sum / sumsq of a tensor that no conv epilogue produced */
int cwf_in_stats(const float* x, int x_ldc, double* stats, int N, int64_t V, int C, void* stream);
/* y = act(x*scale+shift) + residual                     EnBlock2/DeBlock tail, cls_wise_former.py:709-711,751-752 */
int cwf_norm_act_add(const float* x, int x_ldc, const float* scale, const float* shift, float slope,
                     const float* residual, int r_ldc, float* y, int y_ldc, int N, int64_t V, int C, void* stream);
/* backward of y = act(IN(x)): g = dy*act'(xhat);  sums[NC][2] += (sum g, sum g*xhat)  (zeroed by caller) */
int cwf_in_bwd_stats(const float* dy, int dy_ldc, const float* x, int x_ldc, const float* scale, const float* shift,
                     float slope, double* sums, int N, int64_t V, int C, void* stream);
/* dx = scale*(g - S1/V - xhat*S2/V) (+ dx_add if not NULL) */
int cwf_in_bwd_apply(const float* dy, int dy_ldc, const float* x, int x_ldc, const float* scale, const float* shift,
                     float slope, const double* sums, const float* dx_add, int a_ldc, float* dx, int dx_ldc,
                     int N, int64_t V, int C, void* stream);

/* cwf_in_bwd_apply with bf16 side outputs (each nullable; dx or dx16 must be given):
 *   dx16 [N*V][C] bf16 = bf16(dx)                 -- the gradient image the 16-channel weight-gradient kernel below takes
 *   xa16 [N*V][C] bf16 = bf16(act(x*scale+shift)) -- the activated input of the layer whose backward this is, i.e. the x operand of
 *                                                   its weight gradient (convolution_backward's `input`, Unet_skipconnection.py:39-56) */
int cwf_in_bwd_apply_ex(const float* dy, int dy_ldc, const float* x, int x_ldc, const float* scale, const float* shift,
                        float slope, const double* sums, const float* dx_add, int a_ldc, float* dx, int dx_ldc,
                        void* dx16, void* xa16, int N, int64_t V, int C, void* stream);
/* cwf_norm_act_add that also writes y16 [N*V][C] bf16 = bf16(y) (nullable) */
int cwf_norm_act_add_ex(const float* x, int x_ldc, const float* scale, const float* shift, float slope,
                        const float* residual, int r_ldc, float* y, int y_ldc, void* y16, int N, int64_t V, int C, void* stream);
/* y16 [N*V][C] bf16 = bf16(act(x*scale+shift))  (scale == NULL: bf16(x)) */
int cwf_to_bf16(const float* x, int x_ldc, const float* scale, const float* shift, float slope, void* y16,
                int N, int64_t V, int C, void* stream);
/* cwf_conv_mfma_bf16 for a 1x1x1 conv (CWF_CONV1, Cout a multiple of 4) that ALSO writes its output as a bf16 image y16 [N][Do*Ho*Wo][Cout]
 * (DeUp_Cat.conv3, cls_wise_former.py:716-729: the un-normalised input of the next block's first conv, whose weight gradient reads that
 * image).  CWF_E_BADARG if the layer is not one the pointwise stream kernel takes. */
int cwf_conv_mfma_bf16_y16(int op, int x3, const float* x, int x_ldc, const void* wpk16, const float* bias,
                           float* y, int y_ldc, void* y16, const float* in_scale, const float* in_shift, float in_slope,
                           const float* residual, int r_ldc, double* stats,
                           int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, void* stream);
/* cwf_conv_mfma_bf16_nb (single-bf16 operand products) for a 3x3x3 stride-1 16 -> 16 layer of >= 32768 voxels whose INPUT exists as a
 * bf16 image x16 [N][D][H][W][16] -- the data gradient of EnBlock1 / EnBlock1_1 / DeBlock2 / DeBlock2_1 reading the bf16 image of dy that
 * cwf_in_bwd_apply_ex wrote (the data half of aten::convolution_backward, Unet_skipconnection.py:36-57).  zero16: 16 zero bytes. */
int cwf_conv_mfma_bf16_in16(int op, const void* x16, const void* zero16, const void* wpk16, const float* bias,
                            float* y, int y_ldc, const float* residual, int r_ldc, double* stats,
                            const float* nb_x, int nb_ldc, const float* nb_scale, const float* nb_shift, float nb_slope,
                            int N, int D, int H, int W, void* stream);
/* Weight / bias gradient slabs of a 3x3x3 stride-1 16 -> 16 conv (padding 1) from bf16 operand images (single-bf16 products, fp32
 * accumulate): xa16 = bf16(act(IN(x))) [N][D][H][W][16], dy16 [N][D][H][W][16], zero16 = 16 zero bytes (the padding source of the
 * LDS-DMA loaders).  Slab layout and reduction: cwf_wgrad_mfma_bf16(CWF_CONV3_S1, 16, 16) / cwf_wgrad_reduce.
 * Replaces the weight half of aten::convolution_backward for EnBlock1 / EnBlock1_1 / DeBlock2 / DeBlock2_1 (Unet_skipconnection.py:36-57,
 * cls_wise_former.py:732-754). */
int cwf_wgrad16_bf16(const void* xa16, const void* dy16, const void* zero16, float* partial,
                     int N, int D, int H, int W, int* nsplit_used, void* stream);

/* cwf_wgrad_mfma_bf16 with dy taken as dy * dy_scale[n][co] -- the backward of the always-on dropout3d behind InitConv
 * (Unet_skipconnection.py:29-33) folded into the stem's weight gradient; CWF_CONV3_S1 layers with Cin <= 16, Cout = 16 and
 * >= 32768 voxels only (CWF_E_BADARG otherwise). */
int cwf_wgrad_mfma_bf16_dys(int op, int x3, const float* x, int x_ldc, const float* in_scale, const float* in_shift, float in_slope,
                            const float* dy, int dy_ldc, const float* dy_scale, float* partial,
                            int N, int Di, int Hi, int Wi, int Cin, int Do, int Ho, int Wo, int Cout, int* nsplit_used, void* stream);
/* The same for the 3x3x3 stride-1 layers with Cin a multiple of 16 (>= 32) and Cout a multiple of 32 (EnBlock2/3/4, DeBlock3/4, Enblock8,
 * decouplers; Unet_skipconnection.py:36-57, cls_wise_former.py:691-754): xa16 [N][D][H][W][Cin], dy16 [N][D][H][W][Cout] bf16.
 * Slab layout and reduction: cwf_wgrad_mfma_bf16(CWF_CONV3_S1, Cin, Cout) / cwf_wgrad_reduce. */
int cwf_wgrad_s1_bf16(const void* xa16, const void* dy16, const void* zero16, float* partial,
                      int N, int D, int H, int W, int Cin, int Cout, int* nsplit_used, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K6/K7  token path: LayerNorm, Linear (strided batched MFMA GEMM), softmax rows, GELU
 *        ResidualNorm.py:4-47, SelfAttention.py:74-102
 * ---------------------------------------------------------------------------------------------- */
/* C[z][m][n] = act( sum_k A[z](m,k)*B[z](k,n) + bias[n] ) + residual[z][m][n]
 * element strides are given explicitly so that NT / NN / TN products and per-head slices need no copies.
 * z = zb*H + zh with separate strides for the outer (zb) and inner (zh) batch index.
 * act: 0 none, 1 exact GELU (erf).  alpha scales the product (attention 1/sqrt(d)).                     */
int cwf_gemm(const float* A, int64_t sa_m, int64_t sa_k, int64_t sa_zb, int64_t sa_zh,
             const float* B, int64_t sb_k, int64_t sb_n, int64_t sb_zb, int64_t sb_zh,
             float* C, int64_t sc_m, int64_t sc_zb, int64_t sc_zh,
             const float* bias, const float* residual, int64_t sr_m, int64_t sr_zb, int64_t sr_zh,
             int M, int Nn, int K, int ZB, int ZH, float alpha, int act, int accumulate, void* stream);
/* Extended form: the same product with the fusions that keep a coupler block at a handful of launches.  Zero-initialise the
 * struct; fields beyond `accumulate` are optional.
 *   A2/split_n   output columns n >= split_n read A2 (same strides) instead of A: q = LN1(x) Wq^T and k|v = LN2(x2) Wkv^T over the
 *                reference's single [1536,512] qkv weight in ONE launch (SelfAttention.py:80-93); split_n % 64 == 0
 *   B2/split_m   output rows m >= split_m read B2: the weight gradient [dq^T a ; dkv^T b] of the same layer in one launch
 *   C2           receives alpha*AB + bias BEFORE the activation (the GELU input, kept for backward; ResidualNorm.py:40-41)
 *   rowsum       rowsum[m] (+)= sum_k A'(m,k): with A' = dy^T this is the bias gradient of a Linear, from one extra MFMA
 *   a_drop_*     A' = A * keep(offset of the element inside A): dropout of dy recomputed, not stored (backward of
 *                x + Dropout(Linear(.)), ResidualNorm.py:9-10,25-31); *_n = numel of the dropped tensor, p2 = a second chained dropout
 *   c_drop_*     epilogue order: alpha*AB + bias -> C2 -> act -> dropout -> + residual -> (accumulate) -> C
 *   rng          device uint64[2] {seed, step} (cwf_rng_advance)                                                           */
struct cwf_gemm_args {
  const float* A; int64_t sa_m, sa_k, sa_zb, sa_zh;
  const float* B; int64_t sb_k, sb_n, sb_zb, sb_zh;
  float* C; int64_t sc_m, sc_zb, sc_zh;
  const float* bias; const float* residual; int64_t sr_m, sr_zb, sr_zh;
  int M, N, K, ZB, ZH; float alpha; int act; int accumulate;
  const float* A2; int split_n;
  const float* B2; int split_m;
  float* C2;
  float* rowsum; int rowsum_acc;
  const uint64_t* rng;
  uint64_t a_drop_off, a_drop_n; float a_drop_p, a_drop_p2;
  uint64_t c_drop_off, c_drop_n; float c_drop_p, c_drop_p2;
  /* grouped form (the three sub-regions' couplers in one launch, z = group): per-z pointers override B / bias / C / rowsum
   * when the first entry is non-NULL (separate weight tensors per group; A, residual, C2 stay strided) */
  const float* B_tab[4]; const float* bias_tab[4]; float* C_tab[4]; float* rowsum_tab[4];
};
int cwf_gemm_ex(const struct cwf_gemm_args* args /* host */, void* stream);

/* The attention core of one coupler block in one launch each way (SelfAttention.py:94-98; K6):
 *   qkv [Z*T][ld] holds q | k | v side by side (columns [0,E) [E,2E) [2E,3E), head-major), T <= 144, E = heads*64
 *   o[z*T+t][h*64+d] = sum_key dropout(softmax_key(scale * q.k))[t][key] * v[key][d]
 *   backward recomputes the probabilities in LDS: dqkv (same layout as qkv) from d_o.  Attention dropout is keep(drop_off +
 *   ((z*heads+h)*T + t)*T + key) from the device generator state -- no mask tensor, no [Z,heads,T,T] probabilities in HBM. */
int cwf_attn_fwd(const float* qkv, int64_t ld, float* o, int64_t ldo, int Z, int T, int E, int heads, float scale,
                 const uint64_t* rng, uint64_t drop_off, float drop_p, void* stream);
int cwf_attn_bwd(const float* qkv, int64_t ld, const float* d_o, int64_t ldo, float* dqkv, int Z, int T, int E, int heads,
                 float scale, const uint64_t* rng, uint64_t drop_off, float drop_p, void* stream);

/* Paired LayerNorm of a coupler block (PreNormDrop: norm(x), norm2(x2); ResidualNorm.py:23-32):
 *   ya = LN(x; g1,b1) ; yb[r] = LN(x2[perm(r)]; g2,b2) ; stats [2][rows][2] = (mean, rstd) ; x2 may be NULL (PreNorm of the FFN)
 *   perm_T > 0: the second operand is read with the two halves of every sequence pair swapped (rows = pairs * 2 * perm_T),
 *   which is how "a attends b, b attends a" (ClsWiseTransformer.py:47-50) runs as one batch.
 * backward (one launch for the inputs, one for the four parameter gradients, both deterministic, nothing pre-zeroed):
 *   dx2 != NULL:  dx = dy + LN1'(da) ; dx2 = LN2'(db)                    (perm_T == 0)
 *   dx2 == NULL:  dx[r] = dy[r] + LN1'(da[r]) + LN2'(db[perm(r)])        (x2 is x; db may be NULL -> single LayerNorm)
 *   dg*, db* (+)= per-column sums (accumulate_params: the weight-sharing sum over the uses of one block)                  */
int cwf_ln_pair_fwd(const float* x, const float* x2, int perm_T, const float* g1, const float* b1, const float* g2, const float* b2,
                    float* ya, float* yb, float* stats, int rows, int E, float eps, void* stream);
/* grouped forms: rows = groups * rows_per_group, group g uses the g-th LayerNorm parameter set (host arrays of `groups` <= 4 device
 * pointers; the parameter-gradient outputs likewise) -- the three sub-regions' couplers normalised in one launch */
struct cwf_ln_group_params { const float* g1[4]; const float* b1[4]; const float* g2[4]; const float* b2[4];
                             float* dg1[4]; float* db1[4]; float* dg2[4]; float* db2[4]; };
int cwf_ln_pair_fwd_g(const float* x, const float* x2, int perm_T, const struct cwf_ln_group_params* h_params, int groups,
                      float* ya, float* yb, float* stats, int rows, int E, float eps, void* stream);
int cwf_ln_pair_bwd_g(const float* dy, const float* da, const float* db, const float* x, const float* x2, int perm_T,
                      const struct cwf_ln_group_params* h_params, int groups, const float* stats, float* dx, float* dx2,
                      int rows, int E, int accumulate_params, void* stream);
int cwf_ln_pair_bwd(const float* dy, const float* da, const float* db, const float* x, const float* x2, int perm_T,
                    const float* g1, const float* g2, const float* stats, float* dx, float* dx2,
                    float* dg1, float* db1, float* dg2, float* db2, int rows, int E, int accumulate_params, void* stream);
/* dz = dh * keep(off + i) * gelu'(z)      backward of Dropout(GELU(z)), mask recomputed (p may be 0) */
int cwf_gelu_bwd_drop(const float* z, const float* dh, float* dz, int64_t n, const uint64_t* rng, uint64_t off, float p, void* stream);

/* rows x E LayerNorm (eps 1e-5): y = (x-mean)*rstd*gamma+beta; saves mean/rstd [rows] */
int cwf_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                      int rows, int E, float eps, void* stream);
/* dx (+= if accumulate); dgamma/dbeta are WRITTEN (deterministic column reduction, no zero-initialisation needed) */
int cwf_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                      float* dx, float* dgamma, float* dbeta, int rows, int E, int accumulate, void* stream);
/* in-place row softmax over `cols` (rows are contiguous, stride ld) and its backward dS = P*(dP - sum(dP*P)) */
int cwf_softmax_rows(float* s, int64_t rows, int cols, int ld, void* stream);
int cwf_softmax_rows_bwd(const float* p, float* dp_inout, int64_t rows, int cols, int ld, void* stream);
/* dx = dy * gelu'(x) */
int cwf_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* out[c] (+)= sum_rows x[row][c]   (bias gradients of the Linear layers) */
int cwf_colsum(const float* x, int64_t rows, int cols, int ld, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K4/K5  window <-> token reshapes, scoring, top-k, gather / scatter / gate
 *        cls_wise_former.py:15-39 (convert_dim/split_dim), :345-376 (selection), :457-543 (scatter + gate)
 * ---------------------------------------------------------------------------------------------- */
/* tok[b][ (d/p0,h/p1,w/p2) ][ ((c*p0+i)*p1+j)*p2+k ] = x[b][d][h][w][c]   (and the inverse) */
int cwf_window_to_tokens(const float* x, int x_ldc, float* tok, int B, int D, int H, int W, int C,
                         int p0, int p1, int p2, void* stream);
int cwf_tokens_to_window(const float* tok, float* x, int x_ldc, int B, int D, int H, int W, int C,
                         int p0, int p1, int p2, int accumulate, void* stream);
/* score[b][t] = dot(feats[b][t][:], query[b or 0][:])      token @ X^T, fp32 */
int cwf_token_scores(const float* feats, const float* query, int64_t query_bstride, float* score,
                     int B, int T, int E, void* stream);
/* indices of the k largest scores per sample, sorted descending (ties: lower index first)  -> int32 [B][k] */
int cwf_topk(const float* score, int32_t* index, int B, int T, int k, void* stream);
/* seq[b][0] = head[b or 0]; seq[b][1+j] = (feats[b][index[b][j]] + (e odd ? pe1 : 0)) * keep[b][j][e]
 * keep (nullable) is a pre-scaled dropout mask.  (:347-350; PositionalEncoding.py:20-22)   */
int cwf_gather_tokens(const float* feats, const int32_t* index, const float* head, int64_t head_bstride,
                      const float* keep, float pe_odd, float* seq, int B, int T, int k, int E, void* stream);
/* backward: dfeats[b][index[b][j]] += dseq[b][1+j]*keep ; dhead[e] += sum_b dseq[b][0][e] (both pre-zeroed or live) */
int cwf_gather_tokens_bwd(const float* dseq, const int32_t* index, const float* keep, float* dfeats, float* dhead,
                          int64_t dhead_bstride, int B, int T, int k, int E, void* stream);
/* out = feats with rows index[b][j] replaced by rows[b][j] (row stride rows_ld, batch stride rows_bs);
 * optional gate: out[b][t][e] *= gate[b][e]  (gate batch stride gate_bs)                   (:467,481) */
int cwf_scatter_rows(const float* feats, const int32_t* index, const float* rows, int64_t rows_ld, int64_t rows_bs,
                     const float* gate, int64_t gate_bs, float* out, int B, int T, int k, int E, void* stream);
/* backward of scatter (+gate): given dout, pre-gate values `scat` (= scatter result before gating; may be NULL if no gate):
 *   dgate[b][e] = sum_t dout*scat ; g = dout*gate ; dfeats = g with selected rows zeroed (+= if accumulate) ;
 *   drows[b][j] = g[b][index[b][j]]                                                                    */
int cwf_scatter_rows_bwd(const float* dout, const int32_t* index, const float* scat, const float* gate, int64_t gate_bs,
                         float* dfeats, int accumulate, float* drows, int64_t drows_ld, int64_t drows_bs,
                         float* dgate, int64_t dgate_bs, int B, int T, int k, int E, void* stream);

/* ---- round-2 forms: one launch per stage of a sub-region's selection / scatter, gradients written (not accumulated) ------- */
/* s1[b][t] = feats[b][t].q1[b or 0], s2 likewise for q2 (nullable): both class tokens of a region against one token matrix */
int cwf_token_scores2(const float* feats, const float* q1, int64_t q1_bstride, const float* q2, int64_t q2_bstride,
                      float* s1, float* s2, int B, int T, int E, void* stream);
/* grouped: sample b belongs to group b / group_B and is scored against that group's shared queries (host arrays of pointers) */
int cwf_token_scores2_g(const float* feats, const float* const* h_q1, const float* const* h_q2, int groups, int group_B,
                        float* s1, float* s2, int B, int T, int E, void* stream);
/* top-k of one or two score vectors [B][T] (same T, k) in one launch; inv*[b][t] = rank of token t if selected else -1 (nullable);
 * NaN scores order as the largest value (torch.topk) */
int cwf_topk_inv(const float* score0, int32_t* index0, int32_t* inv0, const float* score1, int32_t* index1, int32_t* inv1,
                 int B, int T, int k, void* stream);
/* inverse map of a given index set (teacher-forced selections in tests) */
int cwf_index_inv(const int32_t* index, int32_t* inv, int B, int T, int k, void* stream);
/* up to four gathers in one launch: out[b][0] = head[b or 0] ; out[b][1+j] = (feats[b][index[b][j]] + pe) * keep
 * (the four 129-token sequences of a region written straight into the paired [B][2][129][E] operands; :345-376) */
struct cwf_gather_job { const float* feats; const int32_t* index; const float* head; float* out;
                        int64_t head_bstride, out_bstride; int T; uint64_t drop_off;
                        const float* head_g[4]; int group_B; /* group_B > 0: sample b takes head_g[b / group_B] (shared per group) */ };
int cwf_gather_multi(const struct cwf_gather_job* jobs /* host */, int njobs, int B, int k, int E, float pe_odd,
                     const uint64_t* rng, float p, void* stream);
/* scat = feats with the selected rows replaced (via inv) ; gated = scat * gate ; either output may be NULL      (:463-485) */
int cwf_scatter_inv(const float* feats, const int32_t* inv, const float* rows, int64_t rows_ld, int64_t rows_bs,
                    const float* gate, int64_t gate_bs, float* gated, float* scat, int B, int T, int E, void* stream);
/* backward of the above into the producer of (rows, gate): drows[b][j] = dgated[b][index[j]]*gate + dscat[b][index[j]] ;
 * dgate[b] = sum_t dgated*scat + dgate_extra[b]   (scat re-derived from feats / inv / rows; written, deterministic) */
int cwf_scatter_bwd(const float* dgated, const float* dscat, const float* feats, const int32_t* inv, const int32_t* index,
                    const float* rows, int64_t rows_ld, int64_t rows_bs, const float* gate, int64_t gate_bs,
                    const float* dgate_extra, int64_t extra_bs, float* drows, int64_t drows_ld, int64_t drows_bs,
                    float* dgate, int64_t dgate_bs, int B, int T, int k, int E, void* stream);
/* gradient of a token matrix from its three uses in one pass (written): scatter pass-through of the non-selected rows
 * (dgated*gate + dscat) + adjoint of the primary gather (inv_p, dseq_p) + adjoint of the supplementary gather (inv_q, dseq_q) */
int cwf_token_grad(const float* dgated, const float* dscat, const float* gate, int64_t gate_bs,
                   const int32_t* inv_p, const int32_t* inv_q, const float* dseq_p, int64_t dseq_p_bs,
                   const float* dseq_q, int64_t dseq_q_bs, const uint64_t* rng, uint64_t drop_off_p, uint64_t drop_off_q, float p,
                   float* dfeats, int B, int T, int k, int E, void* stream);
/* class-token gradients: out1 = sum_b (a1[b] + c1[b]), out2 = sum_b (a2[b] + c2[b]) over rows of stride bstride */
int cwf_head_grad(const float* a1, const float* c1, const float* a2, const float* c2, int64_t bstride,
                  float* out1, float* out2, int B, int E, void* stream);
/* grouped: group g sums its samples [g*group_B, (g+1)*group_B) into h_out1[g] / h_out2[g] */
int cwf_head_grad_g(const float* a1, const float* c1, const float* a2, const float* c2, int64_t bstride,
                    float* const* h_out1, float* const* h_out2, int groups, int group_B, int E, void* stream);
/* window <-> token reshapes of `groups` channel groups of one NDHWC tensor in one launch:
 *   tok[g][b][t][f] = x[b][voxel][g*C + c]   and the inverse (x_ldc >= groups*C) */
int cwf_window_to_tokens_g(const float* x, int x_ldc, float* tok, int groups, int B, int D, int H, int W, int C,
                           int p0, int p1, int p2, void* stream);
int cwf_tokens_to_window_g(const float* tok, float* x, int x_ldc, int groups, int B, int D, int H, int W, int C,
                           int p0, int p1, int p2, void* stream);
/* y[v][g*C + c] = xs[g][v][c] (NULL source: zeros), g < 3: the adjoint of slicing one tensor into three channel groups */
int cwf_cat3_channels(const float* x0, const float* x1, const float* x2, float* y, int64_t nvox, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K8/K10  heads: trilinear upsample (align_corners=False) + channel softmax; 4-class channel softmax
 *         SuperviseLabel.py:62-64, EdgeSuperviseLabel.py:58-60, cls_wise_former.py:662-664
 * ---------------------------------------------------------------------------------------------- */
/* prob[n][D*s][H*s][W*s][C] = softmax_c( trilinear_up(logit[n][D][H][W][C (ldc)]) ), C in {2,4} */
int cwf_upsample_softmax(const float* logit, int l_ldc, float* prob, int N, int D, int H, int W, int C, int scale,
                         void* stream);
/* dlogit (low res) from dprob and prob (high res); workspace: N * D*scale * H * W * C floats (separable two-pass adjoint,
 * deterministic); writes only channels [0, C) of dlogit */
int cwf_upsample_softmax_bwd(const float* dprob, const float* prob, float* dlogit, int dl_ldc,
                             int N, int D, int H, int W, int C, int scale, float* workspace, void* stream);
/* prob = softmax over C contiguous channels per voxel; dlogit = p*(dp - sum p*dp) */
int cwf_channel_softmax(const float* logit, int l_ldc, float* prob, int64_t nvox, int C, void* stream);
int cwf_channel_softmax_bwd(const float* dprob, const float* prob, float* dlogit, int dl_ldc, int64_t nvox, int C,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * K9  fused Dice + weighted cross-entropy   utils/tools.py:8-34,112-231; models/criterions.py:49-62
 *   prob  [N][V][C] channels-last, C in {2,4};  label int64 [N][V]
 *   C==4: class = label (0..3).   C==2: class = (posmask >> label) & 1  (label in 0..15)
 *   sums double [N][C][4] += (sum p*t, sum p, sum t, sum t*log(clamp(p,0.005,1)))   (zeroed by caller)
 *   cwf_dice_ce_finalize -> loss[0] = dice + ce ; coef float [N][C][4] for the backward
 *   cwf_dice_ce_bwd: dprob = gscale[0] * dLoss/dprob
 * ---------------------------------------------------------------------------------------------- */
int cwf_dice_ce_sums(const float* prob, const int64_t* label, uint32_t posmask, double* sums,
                     int N, int64_t V, int C, void* stream);
int cwf_dice_ce_finalize(const double* sums, float* loss, float* coef, int N, int64_t V, int C, void* stream);
/* nmaps problems back to back (sums [nmaps][N][C][4], loss [nmaps], coef [nmaps][N][C][4]); total[0] = sum of the losses (nullable) */
int cwf_dice_ce_finalize_multi(const double* sums, float* loss, float* coef, float* total, int nmaps, int N, int64_t V, int C, void* stream);
/* Fused head -> loss (training mode): the sums above for up to three sub-region maps of one supervision call straight from their
 * LOW-resolution 2-channel logits [N][D][H][W][l_ldc] -- trilinear x scale (align_corners=False) + softmax evaluated in registers,
 * one shared read of the label volume, nothing written at full resolution (SuperviseLabel.py:58-81 -> tools.py:112-231).
 * h_logits / h_posmasks / h_dlogits are HOST arrays of nmaps entries.  Backward: dlogits[m] [N][D][H][W][dl_ldc] WRITTEN (channels
 * >= 2 zeroed) from (label, coef, gscale); workspace: nmaps * N * D*scale * H * W * 2 floats.                                      */
int cwf_head_loss_sums(const float* const* h_logits, int nmaps, int l_ldc, const uint32_t* h_posmasks, const int64_t* label,
                       double* sums, int N, int D, int H, int W, int scale, void* stream);
int cwf_head_loss_bwd(const float* const* h_logits, int nmaps, int l_ldc, const uint32_t* h_posmasks, const int64_t* label,
                      const float* coef, const float* gscale, float* const* h_dlogits, int dl_ldc, float* workspace,
                      int N, int D, int H, int W, int scale, void* stream);
/* Same, the maps being channel groups of ONE gradient buffer: dl_ca channels written per voxel (2 gradients + zeroed padding), voxel
 * rows dl_ldc floats apart (what the channel-grouped head convs below consume). */
int cwf_head_loss_bwd_ex(const float* const* h_logits, int nmaps, int l_ldc, const uint32_t* h_posmasks, const int64_t* label,
                         const float* coef, const float* gscale, float* const* h_dlogits, int dl_ca, int dl_ldc, float* workspace,
                         int N, int D, int H, int W, int scale, void* stream);
int cwf_dice_ce_bwd(const float* prob, const int64_t* label, uint32_t posmask, const float* coef, const float* gscale,
                    float* dprob, int N, int64_t V, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N1  sliding-window inference glue   predict_overlap.py:31-58 (tailor_and_concat), :134-141 + utils/tools.py:44-47,89-109
 * ---------------------------------------------------------------------------------------------- */
/* y [B][4][240][240][155] (NCDHW) <- the eight 128^3 window outputs windows[(w*B + b)][128][128][128][4] (channels-last, the model's
 * own output memory), w in the reference's window order; hard overwrite by the later window incl. the reference's last-axis offset
 * quirk (voxels 123..149 of the second depth window land at 128..154).  Replaces x.clone() + eight slice assignments.              */
int cwf_stitch_windows(const float* windows, float* y, int B, void* stream);
/* seg[b][v] = argmax over the 4 classes of prob[b*sb + c*sc + v*sv] (first maximum); with a target (int64 labels 0..3) also the
 * counts[k][3] += (|o & t|, |o|, |t|) of tools.softmax_output_dice's three regions k = WT, TC, ET (uint64, zeroed by the caller)   */
int cwf_argmax_dice(const float* prob, int64_t sb, int64_t sc, int64_t sv, const int64_t* target, int64_t* seg, uint64_t* counts,
                    int B, int64_t V, void* stream);
/* the same plus the per-class counts of tools.softmax_mIOU_score (utils/tools.py:50-61, reported by predict_simple.py): counts[6][3] =
 * WT, TC, ET, class 1, class 2, class 3, each (|o & t|, |o|, |t|); IoU = |o & t| / (|o| + |t| - |o & t|).  target is required.      */
int cwf_argmax_metrics(const float* prob, int64_t sb, int64_t sc, int64_t sv, const int64_t* target, int64_t* seg, uint64_t* counts,
                       int B, int64_t V, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K11 fused Adam (amsgrad, L2 weight decay in the gradient)  torch.optim.Adam as used at train_no_amp.py:136,239
 *   table: device array of cwf_adam_desc; one launch updates every parameter.
 * ---------------------------------------------------------------------------------------------- */
struct cwf_adam_desc { float* p; const float* g; float* m; float* v; float* vmax; int64_t n; };
int cwf_adam_amsgrad(const struct cwf_adam_desc* table, int ntensors, int64_t max_n,
                     double lr, double beta1, double beta2, double eps, double weight_decay, int step, int amsgrad,
                     const float* hyper_dev, void* stream);
/* the same with the gradient read as g * grad_scale (1 / world size: the summed all-reduce result becomes the average here) */
int cwf_adam_amsgrad_scaled(const struct cwf_adam_desc* table, int ntensors, int64_t max_n,
                            double lr, double beta1, double beta2, double eps, double weight_decay, int step, int amsgrad,
                            const float* hyper_dev, float grad_scale, void* stream);
/* hyper-parameters in double: torch derives step_size = lr/(1-beta1^t) and sqrt(1-beta2^t) in double.  hyper_dev
 * (nullable): device float[2] = {step_size, sqrt(1-beta2^t)} overriding the values derived from lr/step -- lets the
 * launch be captured in a hipGraph and replayed while the host advances the schedule.                                  */

/* ------------------------------------------------------------------------------------------------
 * K12 / misc elementwise
 * ---------------------------------------------------------------------------------------------- */
/* y[i] = a[i]*b[i]  (dropout masks) ; y = a + b ; y[n][v][c] = x[n][v][c]*s[n][c] (dropout3d) ; fill */
/* K12: mask[i] = Bernoulli(1-p)/(1-p) [* Bernoulli(1-p2)/(1-p2)], counter-based (seed, offset + i): replaces the
 * rand / compare / cast / scale sequence behind F.dropout (SelfAttention.py:96-100, ResidualNorm.py:25-31,40-45) */
int cwf_dropout_mask(float* mask, int64_t n, float p, float p2, uint64_t seed, uint64_t offset, void* stream);
/* Device-resident generator state rng = uint64[2] {seed, step}.  cwf_rng_advance is a KERNEL (step += 1): captured in a hipGraph
 * it advances on every replay, so replayed training steps draw fresh masks.  Every fused dropout site evaluates
 * keep(i) = u01(seed, step, site_offset + i) >= p ? 1/(1-p) : 0 from the element index, forward and backward alike. */
int cwf_rng_advance(uint64_t* rng, void* stream);
int cwf_dropout_mask_rng(float* mask, int64_t n, float p, float p2, const uint64_t* rng, uint64_t offset, void* stream);
int cwf_mul(const float* a, const float* b, float* y, int64_t n, void* stream);
int cwf_add(const float* a, const float* b, float* y, int64_t n, void* stream);
int cwf_add3(const float* a, const float* b, const float* c, float* y, int64_t n, void* stream);   /* (a + b) + c: the three-region sums of the Mutual Cross-region Coupler, cls_wise_former.py:549-552 */
int cwf_bcast3(const float* x, float* y, int64_t n, void* stream);                 /* y[g][i] = x[i], g < 3: adjoint of the three-region sums */
int cwf_stats_channel_sum(const double* stats, float* out, int N, int C, void* stream);   /* out[c] = sum_n stats[n][c][0]: ConvTranspose bias gradient */
int cwf_channel_scale(const float* x, int x_ldc, const float* s, float* y, int y_ldc, int N, int64_t V, int C, void* stream);
int cwf_copy_strided(const float* x, int x_ldc, float* y, int y_ldc, int64_t nvox, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Launch plans: the captured training step (train_no_amp.py:181-239: forward, five losses, backward, gradient reduces) re-issued
 * as a plain launch list.  The step is static (fixed patch size, device-side top-k, device-resident dropout counters); the host
 * captures it once with HIP stream capture and hands the hipGraph_t over.  cwf_plan_create orders the graph's nodes
 * topologically and assigns them to streams along the capture's chains (main stream; weight-gradient side stream, low priority);
 * cwf_plan_run issues one plain launch per node plus an event pair per cross-stream edge -- hipGraphLaunch is not used (it costs
 * the host ~44 us per node on ROCm 7.2, more than an eager launch).  The graph must outlive the plan (argument blocks are read
 * from its nodes).  Unsupported node kinds (host nodes, child graphs, memcpy nodes) make cwf_plan_create return
 * CWF_E_TOOLARGE; the caller then replays the graph the ordinary way.
 * ---------------------------------------------------------------------------------------------- */
int cwf_plan_create(void* hip_graph, void** plan_out);
const char* cwf_plan_last_error(void);   /* which node made the last cwf_plan_create return CWF_E_TOOLARGE */
/* info8 = {nodes, kernel nodes, markers, streams used, cross-stream events, nodes on stream 0, on stream 1, on streams 2+} */
int cwf_plan_info(void* plan, int* info8);
/* Issues nodes [start, ...) on main_stream (chain 0) and the plan's own side streams until the list ends or a marker has been
 * processed; *next = position to continue from (= node count when finished), *marker_id = the marker's id or -1 when finished.
 * A marker's dependencies are enqueued as waits on comm_stream: work the caller puts on comm_stream after the call (the
 * data-parallel all-reduce of the gradient slice the marker closes) runs behind them.  On finish main_stream waits for the side
 * streams.                                                                                                                       */
int cwf_plan_run(void* plan, void* main_stream, void* comm_stream, int start, int* next, int* marker_id);
int cwf_plan_destroy(void* plan);
/* a no-op kernel carrying `id`: launched (under capture) on the communication stream to mark a cut point of the step */
int cwf_plan_marker(int id, void* stream);

#ifdef __cplusplus
}
#endif
#endif
