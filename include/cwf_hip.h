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

struct cwf_gather_desc { const float* src; float* dst; const int32_t* map; int64_t n; };

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
int cwf_dice_ce_bwd(const float* prob, const int64_t* label, uint32_t posmask, const float* coef, const float* gscale,
                    float* dprob, int N, int64_t V, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K11 fused Adam (amsgrad, L2 weight decay in the gradient)  torch.optim.Adam as used at train_no_amp.py:136,239
 *   table: device array of cwf_adam_desc; one launch updates every parameter.
 * ---------------------------------------------------------------------------------------------- */
struct cwf_adam_desc { float* p; const float* g; float* m; float* v; float* vmax; int64_t n; };
int cwf_adam_amsgrad(const struct cwf_adam_desc* table, int ntensors, int64_t max_n,
                     double lr, double beta1, double beta2, double eps, double weight_decay, int step, int amsgrad,
                     const float* hyper_dev, void* stream);
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
int cwf_mul(const float* a, const float* b, float* y, int64_t n, void* stream);
int cwf_add(const float* a, const float* b, float* y, int64_t n, void* stream);
int cwf_channel_scale(const float* x, int x_ldc, const float* s, float* y, int y_ldc, int N, int64_t V, int C, void* stream);
int cwf_copy_strided(const float* x, int x_ldc, float* y, int y_ldc, int64_t nvox, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif
