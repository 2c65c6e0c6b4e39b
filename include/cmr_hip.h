/* libcmr_hip.so -- C ABI of the MI355X (gfx950) kernels behind CMR-Agent's hot path.
 *
 * The reference (y2w-oc/CMR-Agent) has NO FFI: its hot path is PyTorch op sequences inside
 * models/*.py and environment/environment.py plus the third-party torch_scatter extension.  Each
 * entry point below therefore cites the reference op sequence (file:line) it replaces; the
 * Python host layer (cmr_agent_amd/models, cmr_agent_amd/environment) keeps the reference's
 * nn.Module / function API on top of these calls (see INTEGRATION.md for the binding).
 *
 * Conventions
 *   - every function returns 0 (CMR_OK), -1 (bad argument) or -2 (launch failure); none throws,
 *     allocates or synchronises; all work is enqueued on `stream` (graph-capturable);
 *   - the caller owns every buffer, including workspaces (sizes from the *_workspace_bytes helpers);
 *   - tensors are fp32, row-major "channels-last": images [B,H,W,C], point/token sets [B*L, C];
 *     `ld*` arguments are row strides in floats; float4-accessed buffers must be 16-byte aligned;
 *   - index tensors handed over by the reference API are int64; internal row ids are int32 GLOBAL
 *     row numbers (batch offset already added, cmr_index_to_global_i32).
 */
#ifndef CMR_HIP_H
#define CMR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

#define CMR_OK 0
#define CMR_EINVAL -1
#define CMR_ELAUNCH -2
#define CMR_EUNSUPPORTED -3

enum { CMR_ACT_NONE = 0, CMR_ACT_RELU = 1, CMR_ACT_LRELU = 2, CMR_ACT_GELU = 3, CMR_ACT_ELU1 = 4 };

/* ---- dense contractions (fp32 MFMA) -------------------------------------------------------- */

/* Y[r,:n_out] = act([X1[r,:k1] | X2[map(r),:k2]] W^T + bias + RES[r % res_mod]),  W = [n_out][k1+k2].
 * map(r) = idx2[r] if idx2 else r / div2.  Replaces nn.Linear / Conv1d(k=1) / Conv2d(k=1) (+ folded
 * BatchNorm, + activation, + residual) and the cat/gather feeding them: PointNN.py:96-123,149-156,
 * 209-226,260-282; PointViT.py:66-73; ImageViT.py:81-133; IMGPCEncoder.py:36-102;
 * LinearAttention.py:46-48,63,68; IMGPCEnDecoder.py:77-81; MultiHeadModel.py:34-39,61-67,126-131,
 * 227-233; CMRAgent.py:57-59,70-86,92-101. */
int cmr_linear_f32(const float* x1, int64_t ld1, int k1, const float* x2, int64_t ld2, int k2, const int32_t* idx2,
                   int64_t div2, const float* w, int64_t ldw, const float* bias, const float* res, int64_t ldres,
                   int64_t res_mod, float* y, int64_t ldy, int64_t rows, int n_out, int act, float act_param,
                   hipStream_t stream);
/* Whole ConvBNReLURes1D block in one kernel (PointNN.py:260-282 with BN folded):
 *   hid = lrelu(W1 x + b1);  y = lrelu(W2 hid + b2 + (Wsc x | x)),  x = [x1[:, :k1] | x2[map][:, :kx-k1]].
 * The hidden activations stay in MFMA accumulator registers (transposed GEMM: accumulators of layer 1 are the
 * B-operand fragments of layer 2).  b1 / b2 may be per-batch rows (stride > 0: the broadcast global max-pool
 * concatenation of CMRAgent.py:95-99 folded into the bias); colmax_part receives per-32-row-tile channel maxima
 * of y (CMRAgent.py:95) and is reduced by cmr_colmax_partials_f32.  (kx, ch, co, shortcut) must be one of the
 * instantiated shapes, otherwise CMR_EUNSUPPORTED is returned and the caller composes cmr_linear_f32 calls. */
int cmr_cbr_block_f32(const float* x1, int64_t ld1, int k1, const float* x2, int64_t ld2, const int32_t* idx2,
                      int64_t div2, int kx, int ch, int co, const float* w1, const float* b1, int64_t b1_stride,
                      const float* w2, const float* b2, int64_t b2_stride, const float* wsc, float* y, int64_t ldy,
                      float* colmax_part, int64_t rows, int64_t rows_per_batch, float slope, hipStream_t stream);
/* bf16 matrix-core variant of the same block (same arguments, fp32 rows / weights / biases; operands rounded to bf16 on their
 * way into v_mfma_f32_32x32x16_bf16, fp32 accumulation; the hidden activations go from the accumulator registers of the first
 * GEMM straight into the second as its bf16 operand).  Enabled by cmr_agent_amd.ops.CONV_BF16 (BASELINE configs[2] / [3]). */
int cmr_cbr_block_bf16_f32(const float* x1, int64_t ld1, int k1, const float* x2, int64_t ld2, const int32_t* idx2,
                           int64_t div2, int kx, int ch, int co, const float* w1, const float* b1, int64_t b1_stride,
                           const float* w2, const float* b2, int64_t b2_stride, const float* wsc, float* y, int64_t ldy,
                           float* colmax_part, int64_t rows, int64_t rows_per_batch, float slope, hipStream_t stream);
int cmr_colmax_partials_f32(const float* part, float* out, int B, int tiles_per_batch, int C, hipStream_t stream);
/* The glue between two ConvBNReLURes1D blocks of the agent's 3-D branch in one launch (CMRAgent.py:92-101: cat([feat, max over the points])
 * in front of every block after the first): g [B][64] = column maxima of the block kernel's per-tile maxima (as cmr_colmax_partials_f32; g
 * optional), y1 [B][n1] = g w1^T + b1 and y2 [B][n2] = g w2^T + b2 -- the broadcast half of the next block's input folded into its per-sample
 * biases (w1 [n1][64], w2 [n2][64] contiguous; 64-term fp32 sums in the lane order of the skinny cmr_linear_f32 path: the same maxima, products
 * within a unit or two in the last place of the three launches it replaces).  C = 64, n1 % 4 == 0, n2 % 4 == 0; CMR_EUNSUPPORTED (-3) otherwise. */
int cmr_colmax_bias2_f32(const float* part, int B, int tiles_per_batch, int C, const float* w1, const float* b1, int n1, const float* w2,
                         const float* b2, int n2, float* g, float* y1, float* y2, hipStream_t stream);

/* y = LayerNorm_64(x) * gamma + beta (+ res).  ImageViT.py:139-140, IMGPCEncoder.py:86-87 (eps 1e-6),
 * LinearAttention.py:33-34,64,69,71 (eps 1e-5, residual x + norm2(.)). */
int cmr_layernorm64_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, const float* res,
                        int64_t ldres, float* y, int64_t ldy, int64_t rows, hipStream_t stream);

/* 3x3 convolution, pad 1, stride 1|2, NHWC, w = [9][Cout][Cin] (BN folded), y = lrelu(conv + bias + res) + post.
 * ImageResNet.py:9-14,24-36 (ResidualBlock convs + strided shortcut), IMGPCEnDecoder.py:90-94 (post = 2-D
 * sine table, utils/positional_embedding_2d.py:35-40), MultiHeadModel.py:70-72,236-238, CMRAgent.py:34-56.
 * pool = 2 fuses the following AvgPool2d(2,2) (CMRAgent.py:39,45,51; y is then [B,Ho/2,Wo/2,Cout]); returns
 * CMR_EUNSUPPORTED (-3) for maps too small for the tiled kernel: the caller then pools with cmr_avgpool_nhwc_f32. */
int cmr_conv3x3_nhwc_f32(const float* x, int B, int H, int W, int Cin, const float* w, const float* bias,
                         const float* res, const float* post, float* y, int Cout, int stride, float slope, int pool,
                         hipStream_t stream);

/* Same convolution (stride 1) by Winograd F(2x2,3x3): u = G g G^T of the folded weights, transformed host-side and
 * stored as MFMA A fragments [16 positions][Cout/32][Cin/8][64 lanes][4] with lane 32h+l holding
 * U[pos][32 tile + l][8 kgroup + 4h .. +3] (16*Cout*Cin floats; cmr_agent_amd/models/_pack.py:winograd_u); input / output transforms, the 16 position GEMMs and the epilogue of cmr_conv3x3_nhwc_f32
 * (bias, residual, LeakyReLU, table, optional 2x2 average pool) are fused in one kernel.
 * Launch policy of the PERSISTENT kernel (maps of >= 200 tiles), per call -- the library keeps no state:
 *   cu_budget  CUs it may occupy (0 = all; rounded down to a multiple of 8).  It fills a CU completely, so a branch forked onto another
 *              stream only progresses between its launches unless it is left some CUs (the image tower beside the point tower);
 *   slices     workgroups per CU of that budget, each walking 1 / slices of the tiles (never fewer than 8 tiles per workgroup), so CUs
 *              return to the dispatcher during the launch; <= 1 = one workgroup per CU.  Results do not depend on either. */
int cmr_conv3x3_wino_nhwc_f32(const float* x, int B, int H, int W, int Cin, const float* u, const float* bias,
                              const float* res, const float* post, float* y, int Cout, float slope, int pool,
                              int cu_budget, int slices, hipStream_t stream);
/* Training forward of a 3x3 convolution that feeds a batch-statistics BatchNorm (models/ImageResNet.py:5-40 under Train_Geo.py:166-174):
 * y = conv(x) + bias as the call above with slope 1, no residual / table / pool, AND the sums the BatchNorm needs, accumulated by the helper
 * waves of the wave-specialised kernel while they finish the output tiles (they wait 25 - 50 % of a launch): part [parts][2][64] = per helper
 * wave the sums of (y - bias) and (y - bias)^2 over its pixels -- the separate cmr_bn_stats_f32 pass over y goes away.
 * parts = cmr_conv3x3_wino_stats_parts(...): 0 when this form is not served (Cout = 64, Cin >= 64, maps of >= 200 8x16-pixel tiles are): the
 * caller runs the plain entry point and cmr_bn_stats_f32.  cmr_bn_stats_from_sums_f32(part, parts, B H W, 64, bias, ...) finishes the
 * statistics. */
int64_t cmr_conv3x3_wino_stats_parts(int B, int H, int W, int Cin, int Cout, int cu_budget, int slices);
int cmr_conv3x3_wino_stats_nhwc_f32(const float* x, int B, int H, int W, int Cin, const float* u, const float* bias, float* y, int Cout,
                                    int cu_budget, int slices, float* part, int64_t parts, hipStream_t stream);
/* Data gradient of a stride-1 3x3 convolution whose INPUT was lrelu_{bn_slope}(BatchNorm(bn_x)) with no other consumer (ImageResNet.py:9-14
 * conv -> BatchNorm -> LeakyReLU -> conv under Train_Geo.py:166-174): dx [B][H][W][Cout] = conv(dy; u) (u from the transposed / flipped
 * weights, no bias) AND, from the same helper waves, the two sums of that BatchNorm's backward reduction over the pixels they finish --
 * part [parts][2][64] = sum d, sum d xhat with d = dx * act'(bn_x * stat[2] + stat[3]), xhat = (bn_x - stat[0]) stat[1]: what
 * cmr_bn_bwd_f32's first pass over (dx, bn_x) computes.  bn_x [B][H][W][Cout], bn_stat [4][Cout]; shapes and parts as the statistics
 * launch above (0 <= bn_slope <= 1).  cmr_bn_bwd_from_sums_f32 finishes the BatchNorm backward.  COMPILED OUT by default (answers -3;
 * csrc/conv_wino.hip CMR_WS_BNBWD: its registers cost every launch of the kernel more than the update gains). */
int cmr_conv3x3_wino_bnbwd_nhwc_f32(const float* dy, int B, int H, int W, int Cin, const float* u, float* dx, int Cout, const float* bn_x,
                                    const float* bn_stat, float bn_slope, int cu_budget, int slices, float* part, int64_t parts,
                                    hipStream_t stream);
/* Stride-2 3x3 convolution (the two strided convolutions of a down-sampling ResidualBlock, ImageResNet.py:9-14, :24-27) with the weights
 * as MFMA A fragments [9 taps][Cout/32][Cin/8][64 lanes][4] read straight from L2 (cmr_agent_amd/models/_pack.py:conv_s2_frags): same
 * arithmetic and epilogue as cmr_conv3x3_nhwc_f32 at stride 2 (bias, residual, LeakyReLU), two barriers per 16-channel halo chunk instead
 * of one per tap.  Served: Cin = 64, Cout % 64 == 0, no table operand; otherwise CMR_EUNSUPPORTED (-3) and the caller uses
 * cmr_conv3x3_nhwc_f32. */
int cmr_conv3x3_s2_nhwc_f32(const float* x, int B, int H, int W, int Cin, const float* wfrag, const float* bias, const float* res,
                            const float* post, float* y, int Cout, float slope, hipStream_t stream);
/* MiniResNet block 0 (3 -> 64 channels, 1x1 shortcut), NCHW image in, NHWC features out (fp32, or bf16 when out_bf16 != 0: the
 * bf16-stored image tower of the bf16 mode).  tmp_nchw is scratch of [B][6][H][W] floats (conv-a output | a copy of the image);
 * 0 <= slope <= 1.  ImageResNet.py:50 with :9-23. */
int cmr_stem_block_f32(const float* x_nchw, const float* w_a, const float* b_a, const float* w3, const float* w1,
                       const float* b_b, float* tmp_nchw, void* y_nhwc, int out_bf16, int B, int H, int W, float slope,
                       hipStream_t stream);

/* AvgPool2d((kh,kw), stride=(kh,kw)) on NHWC; (kh,kw)=(H,W) is the global pool.  CMRAgent.py:39,45,51,56. */
int cmr_avgpool_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, int kh, int kw, hipStream_t stream);

/* out = cat([f, nearest_upsample(proxy, scale)], channel).  IMGPCEnDecoder.py:85-89. */
int cmr_upsample_concat_f32(const float* f, const float* proxy, float* out, int B, int H, int W, int C1, int C2,
                            int scale, hipStream_t stream);

/* non-overlapping PxP patches as rows [(ky,kx,c)] for the stride-P patch conv.  ImageViT.py:19-22,51. */
int cmr_patchify_nhwc_f32(const float* x, float* out, int B, int H, int W, int C, int P, hipStream_t stream);
/* The patch embedding itself (ImageViT.py:37-56: Conv2d(C, n_out, kernel = stride = P) + position rows) as ONE GEMM that reads the P x P
 * patches of x [B][H][W][C] in place: y[(b, Y, X)][:] = sum_{py,px,c} w[:][(py P + px) C + c] x[b][P Y + py][P X + px][c] + bias (+ res
 * rows, row % res_mod when res_mod > 0).  = cmr_patchify_nhwc_f32 + cmr_linear_f32 without the patchified copy.  P P C % 64 == 0,
 * (P C) % 8 == 0, at most 65 536 patches. */
int cmr_patch_embed_f32(const float* x_nhwc, int B, int H, int W, int C, int P, const float* w, int64_t ldw, const float* bias,
                        const float* res, int64_t ldres, int64_t res_mod, float* y, int64_t ldy, int n_out, hipStream_t stream);
/* bf16 mode: y = act(x W^T + bias (+ res)) for the big contiguous row maps on the bf16 matrix cores (rows and weights rounded to bf16,
 * fp32 accumulate, fp32 tensors): K in {32, 64, 128}, n_out <= 128 (K = 128: <= 64); other shapes return CMR_EUNSUPPORTED (-3) and the
 * caller uses cmr_linear_f32.  Same argument meaning as cmr_linear_f32 without the second source. */
int cmr_linear_rows_bf16_f32(const float* x, int64_t ldx, int k, const float* w, int64_t ldw, const float* bias, const float* res,
                             int64_t ldres, int64_t res_mod, float* y, int64_t ldy, int64_t rows, int n_out, int act, float act_param,
                             hipStream_t stream);

/* [batch][R][C] -> [batch][C][R]: layout changes at the nn.Module boundary (NCHW <-> NHWC). */
int cmr_transpose_f32(const float* x, float* y, int batch, int R, int Cn, hipStream_t stream);

/* ---- attention ----------------------------------------------------------------------------- */

/* softmax(Q K^T / sqrt(8)) V, 8 heads x 8 dims.  ImageViT.py:93-104, IMGPCEncoder.py:45-53. */
int cmr_mha_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* o,
                int64_t ldo, int B, int Tq, int Tk, hipStream_t stream);
/* The same attention with the exponentials through libm's expf instead of v_exp_f32(x log2 e) (about 25 % slower): the no-dropout forward
 * of the training tape, whose backward kernels (cmr_mha_bwd_f32) recompute the probabilities with expf. */
int cmr_mha_expf_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* o, int64_t ldo, int B,
                     int Tq, int Tk, hipStream_t stream);
/* LayerNorm(64) + the block's query / key / value projections + the attention above in ONE launch (inference; ImageViT.py:81-108,
 * IMGPCEncoder.py:36-58, :93-94 -- x and y pass through the SAME norm): o = softmax(Q K^T / sqrt 8) V with Q = LN(x) Wq^T + bq,
 * K | V = LN(y) Wk|v^T + bk|v (y == x: self-attention).  Every workgroup (64 queries of one (batch, head)) normalises and projects the
 * source rows onto its head's dims while staging them, so q / k / v never exist in memory.  wq_frag / wkv_frag: [8 heads][16][64] floats
 * (cmr_agent_amd/models/_pack.py:mha_ln_frags).  Same value as cmr_ln64_linear_f32 + cmr_mha_f32 up to fp32 summation order. */
int cmr_mha_ln_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gamma, const float* beta, float eps,
                   const float* wq_frag, const float* wkv_frag, const float* bq, const float* bk, const float* bv, float* o, int64_t ldo,
                   int B, int Tq, int Tk, hipStream_t stream);
/* kvsum[b] = { KV[h][d][v] = sum_s K~ V / S (512), Ksum[h][d] (64) }.  LinearAttention.py:55-58. */
int64_t cmr_la_reduce_workspace_bytes(int B, int S);
int cmr_la_reduce_f32(const float* kf, int64_t ldk, const float* v, int64_t ldv, float* kvsum, void* workspace,
                      int64_t workspace_bytes, int B, int S, hipStream_t stream);
/* msg = (Q~ KV) * S / (Q~ Ksum + eps).  LinearAttention.py:59-60. */
int cmr_la_apply_f32(const float* qf, int64_t ldq, const float* kvsum, float* msg, int64_t ldm, int B, int L, int S,
                     float eps, hipStream_t stream);

/* Tail of CMRAgent.forward (CMRAgent.py:52-56 global pool + two 1x1 convs, :101-116 the three MLP heads on
 * cat([embed_2d, embed_3d])) in one launch, one workgroup per sample.  x [B][npix][128] is the activated output of the
 * last 3x3 conv; weights are the PyTorch [out][in] matrices; head i: 256 -> n0 -> n1 -> n2 with LeakyReLU(slope)
 * between, logits written to out[b * ldo + 0..n2).  r_act / t_act (optional, int64 [B][degree]): the deterministic actions of
 * CMRAgent.action_from_logits (CMRAgent.py:118-123), argmax over each group of num_steps logits with the first maximum on ties
 * (= cmr_argmax_rows_f32 on the same rows), emitted by the same launch. */
int cmr_agent_heads_f32(const float* x, int B, int npix, const float* w24, const float* b24, const float* w26,
                        const float* b26, const float* e3d,
                        const float* r_w0, const float* r_b0, const float* r_w1, const float* r_b1, const float* r_w2, const float* r_b2,
                        int r_n0, int r_n1, int r_n2, float* r_out, int r_ldo,
                        const float* t_w0, const float* t_b0, const float* t_w1, const float* t_b1, const float* t_w2, const float* t_b2,
                        int t_n0, int t_n1, int t_n2, float* t_out, int t_ldo,
                        const float* v_w0, const float* v_b0, const float* v_w1, const float* v_b1, const float* v_w2, const float* v_b2,
                        int v_n0, int v_n1, int v_n2, float* v_out, int v_ldo,
                        int num_steps, int degree_r, int degree_t, int64_t* r_act, int64_t* t_act,
                        float slope, hipStream_t stream);
/* Training forward of the same tail (Train_Agent.py:263-305 through CMRAgent.py:52-56, 101-116): the logits and, in `saves`, every
 * intermediate the backward needs -- pooled [B][128] | t1 = lrelu(conv 24) [B][128] | e2d = conv 26 [B][128] | per head (r, t, v):
 * h0 [B][n0], h1 [B][n1] (post-activation) -- from ONE launch instead of 13 (column mean, two 1x1 convs, nine head layers).
 * saves_floats >= B (384 + sum of n0 + n1 over the heads).  Weights [out4][in4] as stored in the flat bucket. */
int cmr_agent_heads_train_f32(const float* x, int B, int npix, const float* w24, const float* b24, const float* w26,
                              const float* b26, const float* e3d,
                              const float* r_w0, const float* r_b0, const float* r_w1, const float* r_b1, const float* r_w2, const float* r_b2,
                              int r_n0, int r_n1, int r_n2, float* r_out, int r_ldo,
                              const float* t_w0, const float* t_b0, const float* t_w1, const float* t_b1, const float* t_w2, const float* t_b2,
                              int t_n0, int t_n1, int t_n2, float* t_out, int t_ldo,
                              const float* v_w0, const float* v_b0, const float* v_w1, const float* v_b1, const float* v_w2, const float* v_b2,
                              int v_n0, int v_n1, int v_n2, float* v_out, int v_ldo,
                              float* saves, int64_t saves_floats, float slope, hipStream_t stream);
/* The same tail with every weight matrix stored TRANSPOSED: W^T [in][out4] (out4 = the output width padded to a multiple of 4 with zero
 * columns; hidden widths n0, n1 multiples of 16): a lane owns four outputs and a wave a slice of the inputs, so that a layer is one memory
 * round trip and no cross-lane reduction (35 -> ~15 us per agent step on the serial chain of a registration).  Same arguments otherwise;
 * results agree with cmr_agent_heads_f32 to fp32 rounding of the sums (other summation order). */
int cmr_agent_heads_t_f32(const float* x, int B, int npix, const float* w24t, const float* b24, const float* w26t, const float* b26,
                          const float* e3d,
                          const float* r_w0t, const float* r_b0, const float* r_w1t, const float* r_b1, const float* r_w2t, const float* r_b2,
                          int r_n0, int r_n1, int r_n2, float* r_out, int r_ldo,
                          const float* t_w0t, const float* t_b0, const float* t_w1t, const float* t_b1, const float* t_w2t, const float* t_b2,
                          int t_n0, int t_n1, int t_n2, float* t_out, int t_ldo,
                          const float* v_w0t, const float* v_b0, const float* v_w1t, const float* v_b1, const float* v_w2t, const float* v_b2,
                          int v_n0, int v_n1, int v_n2, float* v_out, int v_ldo,
                          int num_steps, int degree_r, int degree_t, int64_t* r_act, int64_t* t_act, float slope, hipStream_t stream);

/* Pre-LN transformer block (ImageViT.py:61-158 = PointViT.py:96-183 = IMGPCEncoder.py:14-102) in three launches.
 * Weights marked _f are MFMA A fragments: W [n_out][k] stored as [n_out/32][k/8][64 lanes][4] with lane 32h+l holding
 * W[32 tile + l][8 kgroup + 4h .. +3] (cmr_agent_amd/models/_pack.py:frag_pack).
 *   cmr_ln64_linear_f32: out_x = LN(x) Wx^T + bx and (y != NULL) out_y = LN(y) Wy^T + by with the SAME LayerNorm
 *                        (attention_norm feeds query and key/value, IMGPCEncoder.py:93-94); n_out multiples of 64.
 *   cmr_vit_out_ffn_f32: x1 = ctx Wo^T + bo + x (attn.out + residual); out = x1 + fc2(gelu(fc1(LN(x1))))
 *                        (ffn_norm, Mlp 64 -> 1024 -> 64, exact erf GELU). */
int cmr_ln64_linear_f32(const float* x, int64_t ldx, int64_t rows_x, const float* wf_x, const float* bias_x, int n_out_x,
                        float* out_x, int64_t ldo_x, const float* y, int64_t ldy, int64_t rows_y, const float* wf_y,
                        const float* bias_y, int n_out_y, float* out_y, int64_t ldo_y, const float* gamma, const float* beta,
                        float eps, hipStream_t stream);
int cmr_vit_out_ffn_f32(const float* ctx, int64_t ldc, const float* x, int64_t ldx, const float* wo_f, const float* bo,
                        const float* ln_g, const float* ln_b, float eps, const float* w1_f, const float* b1, const float* w2_f,
                        const float* b2, float* out, int64_t ldo, int64_t rows, hipStream_t stream);
/* The same block tail on 16-row tiles (v_mfma_f32_16x16x4_f32): twice the workgroups for the small proxy sets of this path
 * (3 344 / 2 048 rows -> 209 / 128 workgroups).  Weights as 16x16x4 A fragments [n_out/16][k/16][64 lanes][4], lane = 16 g + m
 * holding W[16 To + m][16 T + 4 g + r], r = 0..3 (cmr_agent_amd/models/_pack.py:frag_pack16); same arguments otherwise. */
int cmr_vit_out_ffn16_f32(const float* ctx, int64_t ldc, const float* x, int64_t ldx, const float* wo_f16, const float* bo,
                          const float* ln_g, const float* ln_b, float eps, const float* w1_f16, const float* b1,
                          const float* w2_f16, const float* b2, float* out, int64_t ldo, int64_t rows, hipStream_t stream);
/* bf16 matrix-core variants of the two kernels above (same arguments; the _f weights are bf16 A fragments
 * [n_out/32][k/16][64 lanes][8], cmr_agent_amd/models/_pack.py:frag_pack_bf16 -- natural k order for wf_x / wf_y / wo_f, accumulator
 * order for w1_f / w2_f; products on v_mfma_f32_32x32x16_bf16, fp32 accumulation; LayerNorm, GELU, biases, residuals fp32). */
int cmr_ln64_linear_bf16_f32(const float* x, int64_t ldx, int64_t rows_x, const void* wf_x, const float* bias_x, int n_out_x,
                             float* out_x, int64_t ldo_x, const float* y, int64_t ldy, int64_t rows_y, const void* wf_y,
                             const float* bias_y, int n_out_y, float* out_y, int64_t ldo_y, const float* gamma, const float* beta,
                             float eps, hipStream_t stream);
int cmr_vit_out_ffn_bf16_f32(const float* ctx, int64_t ldc, const float* x, int64_t ldx, const void* wo_f, const float* bo,
                             const float* ln_g, const float* ln_b, float eps, const void* w1_f, const float* b1, const void* w2_f,
                             const float* b2, float* out, int64_t ldo, int64_t rows, hipStream_t stream);

/* Fused linear-attention layer (LinearAttention.py:38-73) in two kernels; the unfused entry points above stay for
 * callers that need the intermediates.
 *   cmr_la_kv_state_f32:    y [B*S, 64] -> K = elu(Wk y)+1 (:45-46,52), V = (Wv y)/S (:47,57) -> kvsum [B][576] in the
 *                           layout of cmr_la_reduce_f32 (KV[h][d][v] at h*64 + d*8 + v, Ksum[h][d] at 512 + h*8 + d).
 *                           Deterministic (per-wave partials in `workspace`, summed in a fixed order).
 *   cmr_la_query_layer_f32: x [B*L, 64] -> Q = elu(Wq x)+1 -> message (:58-60) -> merge -> LayerNorm (:63-64)
 *                           -> mlp(cat[x, message]): 128 -> 128 ReLU -> 64 (:67) -> LayerNorm (:68) -> x + . (:70).
 *                           Weights are the PyTorch [out][in] matrices (no bias, as in the reference).
 *                           CMR_EUNSUPPORTED when the B states do not fit in LDS beside the weights (B > 11). */
int64_t cmr_la_kv_state_workspace_bytes(int B, int S);
int cmr_la_kv_state_f32(const float* y, int64_t ldy, const float* wk, const float* wv, float* kvsum, void* workspace,
                        int64_t workspace_bytes, int B, int S, hipStream_t stream);
/* The same state with the two projections on the bf16 matrix cores (rows and weights rounded to bf16, fp32 accumulate; elu + 1, 1 / S and
 * the row contraction K^T V stay fp32): the bf16 mode's source side of the layer, next to cmr_la_query_layer_bf16_f32.  Same arguments and
 * workspace. */
int cmr_la_kv_state_bf16_f32(const float* y, int64_t ldy, const float* wk, const float* wv, float* kvsum, void* workspace,
                             int64_t workspace_bytes, int B, int S, hipStream_t stream);
int cmr_la_query_layer_f32(const float* x, int64_t ldx, const float* kvsum, const float* wq, const float* wmerge,
                           const float* ln1_g, const float* ln1_b, const float* w_mlp0, const float* w_mlp3,
                           const float* ln2_g, const float* ln2_b, float* out, int64_t ldo, int B, int L, int S, float eps,
                           float ln_eps, hipStream_t stream);
/* bf16 matrix-core variant of the same layer (same arguments; the four GEMMs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation,
 * operands rounded to bf16; the per-head state product, both LayerNorms and the residual stay fp32).  Enabled by
 * cmr_agent_amd.ops.CONV_BF16. */
int cmr_la_query_layer_bf16_f32(const float* x, int64_t ldx, const float* kvsum, const float* wq, const float* wmerge,
                                const float* ln1_g, const float* ln1_b, const float* w_mlp0, const float* w_mlp3,
                                const float* ln2_g, const float* ln2_b, float* out, int64_t ldo, int B, int L, int S,
                                float eps, float ln_eps, hipStream_t stream);

/* Front half of the vector attention of GroupPointTransformer / KnnPointTransformer (PointNN.py:151-170, 219-226) in one
 * launch, per row r (a (point, owning node) or (node, neighbour) pair):
 *   k, v = [Wk | Wv] (W10 feat[r] + b10)           when feat != NULL   (group transformer)
 *        = kv[ik ? ik[r] : r][0:64], [64:128]      when feat == NULL   (kNN transformer: precomputed table)
 *   pos = Wd2 relu(Wd0 (pa4[ia ? ia[r] : r / diva] - pb4[ib[r]]) + bd0) + bd2        (Wd0: [64][4], 3 columns used)
 *   a_out[r] = Wg2 relu(Wg0 (q[iq ? iq[r] : r / divq] - k + pos) + bg0) + bg2 ;  vp_out[r] = v + pos
 * a_out / vp_out are [rows][64] and feed cmr_segment_softmax_f32.  Replaces cmr_rel_pos_f32 + cmr_vecattn_prep_f32 +
 * six cmr_linear_f32 launches. */
int cmr_vecattn_front_f32(const float* feat, int64_t ldf, const float* w10, const float* b10, const float* wkv, const float* kv,
                          int64_t ldkv, const int32_t* ik, const float* q, int64_t ldq, const int32_t* iq, int64_t divq,
                          const float* pa4, const int32_t* ia, int64_t diva, const float* pb4, const int32_t* ib,
                          const float* wd0, const float* bd0, const float* wd2, const float* bd2, const float* wg0,
                          const float* bg0, const float* wg2, const float* bg2, float* a_out, float* vp_out, int64_t rows,
                          hipStream_t stream);
/* The same front under model.train() (Train_Geo.py:166-174): k and v from their own [*][64] maps (row ikv ? ikv[r] : r of both), q gathered
 * as above, and beside
 * a_out / vp_out the three activations a layer-by-layer backward reads, all [rows][64]: hd = relu(fc_delta[0]), t = q - k + pos,
 * g1 = relu(fc_gamma[0]).  Replaces four cmr_linear_f32 launches, the gather of q and cmr_vecattn_mix_f32 of a training step.  rows a
 * multiple of 32, else CMR_EUNSUPPORTED. */
int cmr_vecattn_front_train_f32(const float* k, int64_t ldk, const float* v, int64_t ldv, const int32_t* ikv, const float* q, int64_t ldq,
                                const int32_t* iq,
                                int64_t divq, const float* pa4, const int32_t* ia, int64_t diva, const float* pb4, const int32_t* ib,
                                const float* wd0, const float* bd0, const float* wd2, const float* bd2, const float* wg0, const float* bg0,
                                const float* wg2, const float* bg2, float* a_out, float* vp_out, float* hd_out, float* t_out, float* g1_out,
                                int64_t rows, hipStream_t stream);
/* ... and with k, v computed inside as well (group transformer: x = W10 feat + b10, k = Wk x, v = Wv x): x_out [rows][64] is stored for the
 * projections' backward, k and v never leave the registers. */
int cmr_vecattn_front_kv_train_f32(const float* feat, int64_t ldf, const float* w10, const float* b10, const float* wk, const float* wv,
                                   const float* q, int64_t ldq, const int32_t* iq, int64_t divq, const float* pa4, const int32_t* ia,
                                   int64_t diva, const float* pb4, const int32_t* ib, const float* wd0, const float* bd0, const float* wd2,
                                   const float* bd2, const float* wg0, const float* bg0, const float* wg2, const float* bg2, float* a_out,
                                   float* vp_out, float* hd_out, float* t_out, float* g1_out, float* x_out, int64_t rows,
                                   hipStream_t stream);

/* ---- loss / metric values of the heads (forward only) ---------------------------------------- */

/* MultiHeadModel.py:68-97 on 2-class logits rows [rows][ld >= 2] with int64 labels: out4 = {focal loss (focal_loss.py:55-110,
 * gamma 2, mean), precision, recall, accuracy}; rows = B * n (accuracy is (correct / B) / n as in the reference). */
int64_t cmr_focal_metrics_workspace_bytes(int64_t rows);
int cmr_focal_metrics_f32(const float* logits, int64_t ld, const int64_t* label, float alpha, int64_t rows, int B, float* out4,
                          void* workspace, int64_t workspace_bytes, hipStream_t stream);
/* MultiHeadModel.py:141-178, 240-268: circle loss over the n sampled (point, pixel) pairs of every sample.  pc_feat rows
 * [B*N][64], img_feat NHWC [B][h][w][64], pc_idx int64 [B][n], xy_int int64 [B][2][n] (x, y), xy_float [B][2][n];
 * out[0] = lambda * mean. */
int64_t cmr_circle_loss_workspace_bytes(int B, int n);
int cmr_circle_loss_f32(const float* pc_feat, const float* img_feat, const int64_t* pc_idx, const int64_t* xy_int,
                        const float* xy_float, int B, int N, int h, int w, int n, float dist_thres, float pos_margin,
                        float neg_margin, float log_scale, float lambda, float* out, void* workspace, int64_t workspace_bytes,
                        hipStream_t stream);

/* ---- rollout ops of the training loop ---------------------------------------- */

/* environment.py:143-176 (expert): residual pose target * source^-1 -> extrinsic-xyz Euler angles (folded back when the
 * x angle exceeds 3 rad) and translation -> nearest entries of the float64 step tables.  act_r int64 [B,1] / act_t [B,2]
 * (six_dof = 0: y rotation, x / z translation) or [B,3] each. */
int cmr_expert_action_f32(const float* pose_source, const float* pose_target, const double* r_steps, const double* t_steps,
                          int num_steps, int six_dof, int64_t* act_r, int64_t* act_t, int B, hipStream_t stream);
/* environment.py:263-302 (reward): distance[b] = mean over mask[b] != 0 of |pc_in_cam - (pc - centroid(pc))|^2 on planar
 * [B,3,N] clouds; reward[b] = +0.5 / -0.5 / 0 against prev_distance[b] (0 when prev_distance is NULL). */
int cmr_reward_f32(const float* pc, const float* pc_in_cam, const int64_t* mask, const float* prev_distance, float* distance,
                   float* reward, int B, int N, hipStream_t stream);
/* buffer.py:24-33 (discounted): out[r][i] = vals[r][i] + gamma * out[r][i+1] over rows of length T. */
int cmr_discounted_f32(const float* vals, float* out, float gamma, int64_t rows, int T, hipStream_t stream);

/* ---- point-cloud ops ----------------------------------------------------------------------- */

/* planar [B,C,N] (the reference's point layout) -> rows [B*N, Cpad], Cpad in {4, 8}, zero padded. */
int cmr_planar_to_rows_f32(const float* x, float* y, int B, int C, int N, int Cpad, hipStream_t stream);
/* out[r] = [x1[r] | x2[map(r)]]: a materialised torch.cat for the one consumer that needs it as a
 * residual (CMRAgent.py:97-99 feeding the identity shortcut of state_3d_embed[3]). */
int cmr_concat_rows_f32(const float* x1, int64_t ld1, int C1, const float* x2, int64_t ld2, int C2, const int32_t* idx2,
                        int64_t div2, float* out, int64_t rows, hipStream_t stream);
int cmr_index_to_global_i32(const int64_t* idx, int32_t* out, int B, int N, int M, hipStream_t stream);

/* CSR of the points owned by each node (replaces the expanded-index torch_scatter calls,
 * PointNN.py:171,175,182; third-party torch_scatter, version unpinned in the reference). */
int cmr_csr_build_i32(const int32_t* key, int32_t* count, int32_t* offsets, int32_t* order, int B, int n_per_batch,
                      int seg_per_batch, hipStream_t stream);

/* 16 nearest nodes of every node, ascending distance.  PointNN.py:215-216 (square_distance + argsort). */
int cmr_knn16_f32(const float* xyz4, int32_t* out, int B, int M, hipStream_t stream);

/* K nearest candidates (rows c4 [B*N,4]) of every query (rows q4 [B*S,4]), 1 <= K <= 64: out int64 [B,S,K] = LOCAL candidate indices in
 * ascending distance, equal distances in ascending index -- square_distance(new_xyz, xyz).argsort()[:, :, :K] of pointnet_util.py:114-116
 * (sample_and_group, knn=True) and :232-234 (PointNetSetAbstractionMsg).  N < K: the missing entries are -1. */
int cmr_knn_f32(const float* q4, const float* c4, int64_t* out, int B, int S, int N, int K, hipStream_t stream);

/* nearest candidate per query.  PointViT.py:85-87 (node -> proxy), dataset/KittiDataset.py:366-367 (point -> node). */
int cmr_nearest_f32(const float* q4, const float* c4, int32_t* out_global, int64_t* out_local, int B, int Nq, int Nc,
                    hipStream_t stream);

/* out[r] = a[map_a(r)] - b[map_b(r)] (xyz rows).  PointNN.py:164-166, :223. */
int cmr_rel_pos_f32(const float* a, const int32_t* ia, int64_t diva, const float* b, const int32_t* ib, int64_t divb,
                    float* out, int64_t rows, hipStream_t stream);

/* t = q[map_q] - k[map_k] + pos ; vp = v[map_k] + pos.  PointNN.py:162,168,179 and :220,225,228. */
int cmr_vecattn_prep_f32(const float* q, int64_t ldq, const int32_t* iq, int64_t divq, const float* k, int64_t ldk,
                         const float* v, int64_t ldv, const int32_t* ik, const float* pos, float* t, float* vp,
                         int64_t rows, hipStream_t stream);

/* per-segment, per-channel softmax(attn*scale) weighted sum of vp.  PointNN.py:170-182 and :226-228. */
int cmr_segment_softmax_f32(const float* attn, const float* vp, const int32_t* order, const int32_t* offsets,
                            int fixed_len, float scale, float* out, int64_t nseg, hipStream_t stream);

/* torch_scatter.scatter_{sum,max,mean} over CSR segments (mode 0 / 1 / 2; empty segments -> 0): the op API of the
 * third-party extension the reference calls at PointNN.py:171-182 and environment.py:79. */
int cmr_segment_reduce_f32(const float* src, int64_t lds, const int32_t* order, const int32_t* offsets, float* out,
                           int64_t ldo, int64_t nseg, int C, int mode, hipStream_t stream);

/* out[r,:C] = src[idx[r],:C].  pointnet_util.py:36-47 (index_points), torch.gather of feature rows. */
int cmr_gather_rows_f32(const float* src, int64_t lds, const int32_t* idx, float* out, int64_t ldo, int64_t rows, int C,
                        hipStream_t stream);

/* pointnet_util.py:50-70 (start index explicit), :73-93, :19-33.  start[b] outside [0, N) is forced into the cloud (the
 * reference raises IndexError at xyz[batch_indices, farthest]; a kernel cannot, and must not read outside the cloud). */
int cmr_fps_f32(const float* xyz4, const int64_t* start, int64_t* out, int B, int N, int npoint, hipStream_t stream);
/* The same sampling with several workgroups per cloud (slices in registers, one 64-bit atomic max + an arrival counter per
 * round; bit-identical indices): for clouds above 16 384 points (BASELINE configs[4]: 65 536), B * groups <= 256.  ws = scratch of
 * cmr_fps_workspace_bytes.  The waits between the workgroups of a cloud are bounded; a cloud whose workgroups could not all
 * become resident in time (CU-filling kernels of another stream) is recomputed by ONE workgroup in the same call (repair
 * launch behind the cooperative one; no host round trip, capturable), so `out` always holds the reference's indices.  After
 * the call the int32 word at ws + B*npoint*8 + 4*(B + b) is 0 (cloud b sampled cooperatively) or 2 (repaired). */
int64_t cmr_fps_workspace_bytes(int B, int N, int npoint);
int cmr_fps_ws_f32(const float* xyz4, const int64_t* start, int64_t* out, int B, int N, int npoint, void* ws, int64_t ws_bytes,
                   hipStream_t stream);
/* The same call with the spin bound as an argument (polls per round before a workgroup gives its cloud up; 0 = give up at once):
 * lets a caller trade waiting for the repair pass, and lets the tests drive the repair path on purpose. */
int cmr_fps_ws_spin_f32(const float* xyz4, const int64_t* start, int64_t* out, int B, int N, int npoint, void* ws, int64_t ws_bytes,
                        int spin_limit, hipStream_t stream);
int cmr_ball_query_f32(const float* xyz4, const float* new4, int64_t* out, int B, int N, int S, int nsample,
                       float radius2, hipStream_t stream);
int cmr_square_distance_f32(const float* a4, const float* b4, float* out, int B, int N, int M, hipStream_t stream);

/* pointnet_util.py:287-296 (PointNetFeaturePropagation): 3 nearest of xyz2 + inverse-distance weights, then
 * the weighted sum of their feature rows. */
int cmr_three_nn_f32(const float* q4, const float* c4, int32_t* idx, float* wgt, int B, int Nq, int Nc,
                     hipStream_t stream);
int cmr_weighted_gather3_f32(const float* src, int64_t lds, const int32_t* idx, const float* wgt, float* out, int64_t ldo,
                             int64_t rows, int C, hipStream_t stream);
/* its backward w.r.t. src (train-mode PointNetFeaturePropagation): out[t] = sum of wgt[e] * dy[e / 3] over the entries e of idx that name
 * row t, in ascending e through the CSR (offsets, order) of idx over the nseg source rows (cmr_csr_build_i32): deterministic, no atomics. */
int cmr_weighted_scatter3_f32(const float* dy, int64_t ldd, const float* wgt, const int32_t* order, const int32_t* offsets, float* out,
                              int64_t ldo, int64_t nseg, int C, hipStream_t stream);

/* per-batch max / mean over rows.  CMRAgent.py:95 (torch.max over points), environment.py:46,88 (pc.mean). */
int64_t cmr_colreduce_workspace_bytes(int B, int N, int C);
int cmr_colmax_f32(const float* x, int64_t ldx, float* out, void* ws, int64_t ws_bytes, int B, int N, int C,
                   hipStream_t stream);
int cmr_colmean_f32(const float* x, int64_t ldx, float* out, void* ws, int64_t ws_bytes, int B, int N, int C,
                    hipStream_t stream);

/* ---- agent iteration ------------------------------------------------------------------------ */

/* environment.py:24-126: projection, in-frustum test, scatter-mean numerator/denominator, state_3d.  acc [B*h*w][64] and
 * cnt [B*h*w] must be zero on entry: zero_first = 1 clears them here (two memsets), zero_first = 0 relies on the caller
 * (cmr_observation_finalize_f32 with clear = 1 re-zeroes exactly the cells that were hit). */
int cmr_project_scatter_f32(const float* pc4, const float* feat, const uint8_t* overlap, const float* pose,
                            const float* Kmat, const float* mean4, float* acc, float* cnt, float* state3d, int B, int N,
                            int h, int w, int zero_first, hipStream_t stream);
/* state2d = [img_feat | acc / max(cnt,1)]; proj (optional) receives the second half alone as [B,h,w,64].  state2d may be null
 * when proj is given (a caller that consumes the two halves separately, as CMRAgent's first convolution does: the 128-channel
 * map is then neither written nor is img_feat read).  clear = 1: cells with cnt > 0 are reset to zero after they have been read. */
int cmr_observation_finalize_f32(const float* img_feat, float* acc, float* cnt, float* state2d, float* proj,
                                 int B, int h, int w, int write_img, int clear, hipStream_t stream);
/* The projected half of the observation alone, maintained in place across the steps of a registration (inference loops that hand the two
 * halves of state_2d to CMRAgent separately; environment.observation_from_a_pose(materialize_state_2d = False)): proj [B,h,w,64], cnt
 * [B*h*w] and cell [B*N] (int32, the cell of every point in the previous call, -1 = none) are STATE owned by the caller -- zero / zero / -1
 * before the first call, untouched between calls.  Each call zeroes the cells of the previous call per point, projects with `pose`, writes
 * state3d [B*N,8] like cmr_project_scatter_f32 and adds feat / count into the cells hit now: proj = scatter_mean (environment.py:39-80) with
 * the mean formed as a sum of pre-divided terms, cells without points 0.  Three launches over the points; the full map is never swept. */
int cmr_observation_proj_f32(const float* pc4, const float* feat, const uint8_t* overlap, const float* pose, const float* Kmat,
                             const float* mean4, float* proj, float* cnt, int32_t* cell, float* state3d, int B, int N, int h, int w,
                             hipStream_t stream);
/* environment.py:179-260 (step + euler_angles_to_matrix 'XYZ'), :14-21 (to_disentangled). */
int cmr_pose_step_f32(float* pose, const int64_t* act_r, const int64_t* act_t, const double* r_steps,
                      const double* t_steps, int B, int six_dof, hipStream_t stream);
int cmr_to_disentangled_f32(float* pose, const float* mean4, int B, hipStream_t stream);
/* CMRAgent.py:118-127 (deterministic action = argmax). */
int cmr_argmax_rows_f32(const float* x, int64_t* out, int outer, int inner, int n, int64_t stride_outer,
                        int64_t stride_inner, hipStream_t stream);
/* MultiHeadModel.py:330-341 (softmax over 2 classes, thresholds .5/.8). */
int cmr_softmax2_f32(const float* logits, int64_t ld, float* prob, uint8_t* pred_lo, uint8_t* pred_hi, float thr_lo,
                     float thr_hi, int64_t rows, hipStream_t stream);
/* MultiHeadModel.py:233,241 (F.normalize over channels). */
int cmr_l2norm64_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, hipStream_t stream);

/* ---- agent update: training-mode forward, backward, optimizer ------------------ */
/* The reference trains CMRAgent with `agent.train(); loss.backward(); optimizer.step()` on minibatches of 10 buffered
 * samples (Train_Agent.py:256-305).  There is no FFI to mirror; these entry points are what the Python host layer
 * (cmr_agent_amd/train/agent_update.py) calls in place of torch autograd.  Data-gradient contractions reuse the forward
 * entry points (cmr_conv3x3_wino_nhwc_f32 / cmr_conv3x3_nhwc_f32 / cmr_linear_f32) with transposed weights. */

/* nn.BatchNorm2d / nn.BatchNorm1d in train() mode (CMRAgent.py:35,41,47,53; PointNN.py:265,268,277 via CMRAgent.py:25-29)
 * over a row map x [rows, C]: batch mean / biased variance per channel; stat [4][C] = (mean, rstd, gamma*rstd,
 * beta - mean*gamma*rstd); running_mean / running_var (optional) are updated with `momentum` and the unbiased variance. */
int64_t cmr_bn_workspace_bytes(int64_t rows, int C);
int cmr_bn_stats_f32(const float* x, int64_t ldx, int64_t rows, int C, float eps, float momentum, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float* stat, void* ws, int64_t ws_bytes,
                     hipStream_t stream);
/* BatchNorm statistics from partial sums a producer left (cmr_conv3x3_wino_stats_nhwc_f32): part [parts][2][C] = sums of (x - pivot[c]) and
 * (x - pivot[c])^2 over disjoint row sets covering all `rows` rows (pivot null = 0); stat [4][C] and the running statistics exactly as
 * cmr_bn_stats_f32 leaves them. */
/* cmr_bn_bwd_f32 (z null: mask from the sign of x * stat[2] + stat[3]; no add / masked output) with the reduction's partial sums given:
 * part [parts][2][C] as cmr_conv3x3_wino_bnbwd_nhwc_f32 leaves them.  ws: 2 C floats. */
int cmr_bn_bwd_from_sums_f32(const float* dz, int64_t lddz, float slope, const float* x, int64_t ldx, const float* stat, const float* part,
                             int64_t parts, float* dx, int64_t lddx, float* dgamma, float* dbeta, int64_t rows, int C, void* ws, int64_t ws_bytes,
                             hipStream_t stream);
int cmr_bn_stats_from_sums_f32(const float* part, int64_t parts, int64_t rows, int C, const float* pivot, float eps, float momentum,
                               const float* gamma, const float* beta, float* running_mean, float* running_var, float* stat, hipStream_t stream);
/* y = LeakyReLU_slope(x * scale + shift + (res * rscale + rshift | res)): BatchNorm application + activation, and the
 * `final_relu(net(x) + shortcut(x))` of ConvBNReLURes1D (PointNN.py:282).  Null scale / rscale = identity; slope 1 = none. */
int cmr_affine_act_f32(const float* x, int64_t ldx, const float* scale, const float* shift, const float* res, int64_t ldres,
                       const float* rscale, const float* rshift, float* y, int64_t ldy, int64_t rows, int C, float slope,
                       hipStream_t stream);
/* Backward of [BatchNorm(train) -> LeakyReLU]: dz = gradient w.r.t. the activation output z, x = the BatchNorm input, stat from
 * cmr_bn_stats_f32.  z null: slope 1 = no activation; slope != 1 = the activation sat directly on the BatchNorm output (no residual in
 * between) and its mask is taken from the sign of x * scale + shift, recomputed with cmr_affine_act_f32's own fused multiply-add (the
 * stored output is then not read: one map pass less in each of the two sweeps; CMR_EINVAL unless 0 <= slope <= 1, the range in which that
 * sign is the forward's mask -- a caller with a residual in front of the activation MUST pass z).  dx = gamma rstd (dy - mean(dy) - xhat mean(dy xhat)) (+ add),
 * dgamma = sum dy xhat, dbeta = sum dy (written when non-null).  dzm (optional): receives dy = dz * act'(z), the gradient at the
 * activation's input -- what a residual branch added in front of the activation gets (`lrelu(BN(x) + res)`, ImageResNet.py:36-40,
 * PointNN.py:282), from the same pass instead of a separate activation-backward sweep. */
int64_t cmr_bn_bwd_workspace_bytes(int64_t rows, int C);
int cmr_bn_bwd_f32(const float* dz, int64_t lddz, const float* z, int64_t ldz, float slope, const float* x, int64_t ldx,
                   const float* stat, const float* add, int64_t ldadd, float* dx, int64_t lddx, float* dzm, int64_t lddzm,
                   float* dgamma, float* dbeta, int64_t rows, int C, void* ws, int64_t ws_bytes, hipStream_t stream);
/* The reduction half of cmr_bn_bwd_f32 alone (same workspace): coef [2][C] = (mean(dy), mean(dy xhat)), dgamma, dbeta. */
int cmr_bn_bwd_coef_f32(const float* dz, int64_t lddz, const float* z, int64_t ldz, float slope, const float* x, int64_t ldx,
                        const float* stat, float* coef, float* dgamma, float* dbeta, int64_t rows, int C, void* ws, int64_t ws_bytes,
                        hipStream_t stream);
/* Backward of a train-mode [1x1 conv -> BatchNorm -> LeakyReLU (+ residual)] layer on a big row map in one pass over the maps
 * (Train_Geo.py:166-174 through PointNN.py:96-123 MiniPointNet / :260-282 ConvBNReLURes1D; Train_Agent.py:296-305 through
 * CMRAgent.py:25-33): with h = x W^T + b the BatchNorm input, z the layer output and (stat, coef) from cmr_bn_stats_f32 /
 * cmr_bn_bwd_coef_f32:  d = dz * act'(z) (-> dzm when non-null),  dh = scale (d - c1 - xhat c2),  dw (+)= dh^T x,  dx = dh W (+ res;
 * dx may alias res; dx null: weight gradient only).  stat = coef = null: no BatchNorm (dh = d).  z null: no activation.  db non-null:
 * (+)= column sums of dh (the bias gradient of a layer without BatchNorm).  seg_db non-null: seg_db [rows / seg_rows][n] = the column sums
 * of dh per seg_rows-row segment (per sample: the gradient of a per-sample vector broadcast to its points, CMRAgent.py:95-99).
 * mask_from_h: z was never stored (the next layer consumed it through cmr_linear_bn_fwd_f32's prologue): act' from the sign of
 * h * stat[2] + stat[3].  xstat non-null: x is the PREVIOUS layer's BatchNorm input, the operand is lrelu_{xslope}(x * xstat[2] +
 * xstat[3]), dx is the gradient at that never-stored activation and the previous layer's BatchNorm-backward reduction comes out of the
 * same pass: xcoef [2][k] (= cmr_bn_bwd_coef_f32's result for it), xdgamma / xdbeta [k] (written when non-null).
 * Serves n, k in {64, 128} (xstat: n = 64) and rows a multiple of 32; anything else -> CMR_EUNSUPPORTED (compose cmr_bn_bwd_f32,
 * cmr_linear_wgrad_f32 and cmr_linear_f32 on the transposed weights). */
int64_t cmr_bn_linear_bwd_workspace_bytes(int64_t rows, int n, int k);
int cmr_bn_linear_bwd_f32(const float* dz, int64_t lddz, const float* z, int64_t ldz, float slope, const float* h, int64_t ldh,
                          const float* stat, const float* coef, int mask_from_h, float* dzm, int64_t lddzm, const float* x, int64_t ldx,
                          const float* xstat, float xslope, float* xcoef, float* xdgamma, float* xdbeta, const float* w, int64_t ldw,
                          const float* res, int64_t ldres, float* dx, int64_t lddx, int64_t rows, int n, int k, float* dw, int64_t lddw,
                          int accumulate, float* db, int accumulate_db, int64_t seg_rows, float* seg_db, void* ws, int64_t ws_bytes,
                          hipStream_t stream);
/* The same layer backward with its two products (dw = dh^T x', dx = dh W) on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32
 * accumulate; operands rounded to bf16, round-to-nearest-even, on the way into LDS) for the bf16 mode of the agent update (BASELINE
 * configs[2]; Train_Agent.py:296-305 through CMRAgent.py:25-33, 92-101).  Maps stay fp32 in HBM; masks, the BatchNorm-backward arithmetic,
 * db / seg_db (sums of the UNROUNDED dh), the lazy operand's sums, partial sums and their reduction exactly as cmr_bn_linear_bwd_f32.
 * Same arguments, shapes served and return codes; its own workspace size. */
int64_t cmr_bn_linear_bwd_bf16_workspace_bytes(int64_t rows, int n, int k);
int cmr_bn_linear_bwd_bf16_f32(const float* dz, int64_t lddz, const float* z, int64_t ldz, float slope, const float* h, int64_t ldh,
                          const float* stat, const float* coef, int mask_from_h, float* dzm, int64_t lddzm, const float* x, int64_t ldx,
                          const float* xstat, float xslope, float* xcoef, float* xdgamma, float* xdbeta, const float* w, int64_t ldw,
                          const float* res, int64_t ldres, float* dx, int64_t lddx, int64_t rows, int n, int k, float* dw, int64_t lddw,
                          int accumulate, float* db, int accumulate_db, int64_t seg_rows, float* seg_db, void* ws, int64_t ws_bytes,
                          hipStream_t stream);
/* Forward of the same layer with its statistics from the same pass: h [rows][n] = x' W^T + bias and stat [4][n] = the batch statistics
 * of h exactly as cmr_bn_stats_f32 defines them (mean, rstd, scale, shift; running statistics updated when given).  pro_stat non-null:
 * x' = lrelu_{pro_slope}(x * pro_stat[2 k ..] + pro_stat[3 k ..]) -- x is then the PREVIOUS layer's BatchNorm input and that layer's
 * activated output is never written (PointNN.py:96-123 layer_1 -> layer_2 -> layer_3, :260-282 net[0] -> net[3]).  bias_seg_rows > 0:
 * bias is [rows / bias_seg_rows][n] with row stride bias_stride, one row per segment (sample) of bias_seg_rows rows: the broadcast half of
 * cat([feat, max]) times its weights (CMRAgent.py:95-99).  n, k in {64, 128}, rows a multiple of 32; else CMR_EUNSUPPORTED (compose
 * cmr_linear_f32 + cmr_bn_stats_f32). */
int64_t cmr_linear_bn_fwd_workspace_bytes(int64_t rows, int n, int k);
int cmr_linear_bn_fwd_f32(const float* x, int64_t ldx, int k, const float* pro_stat, float pro_slope, const float* w, int64_t ldw,
                          const float* bias, int64_t bias_seg_rows, int64_t bias_stride, float* h, int64_t ldh, int64_t rows, int n, float eps,
                          float momentum, const float* gamma, const float* beta, float* running_mean, float* running_var, float* stat,
                          void* ws, int64_t ws_bytes, hipStream_t stream);
/* The same layer forward with its product on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate; x' and W rounded to bf16,
 * round-to-nearest-even) for the bf16 mode of the agent update (BASELINE configs[2]; CMRAgent.py:25-33, 92-101 under Train_Agent.py:296-305):
 * h stays fp32 in HBM, stat = the batch statistics of THAT h (per-wave pivoted sums merged in double), running statistics as
 * cmr_linear_bn_fwd_f32 leaves them.  Same arguments and shapes served (+ w 16-byte aligned, ldw % 4 == 0); its own workspace size. */
int64_t cmr_linear_bn_fwd_bf16_workspace_bytes(int64_t rows, int n, int k);
int cmr_linear_bn_fwd_bf16_f32(const float* x, int64_t ldx, int k, const float* pro_stat, float pro_slope, const float* w, int64_t ldw,
                          const float* bias, int64_t bias_seg_rows, int64_t bias_stride, float* h, int64_t ldh, int64_t rows, int n, float eps,
                          float momentum, const float* gamma, const float* beta, float* running_mean, float* running_var, float* stat,
                          void* ws, int64_t ws_bytes, hipStream_t stream);
/* The elementwise glue of a vector-attention layer in training (PointNN.py:151-170, 219-226: `fc_gamma(q - k + pos)`, `v + pos`) in one pass
 * each way: a_in = q - k + pos, vp = v + pos (contiguous [rows][C] outputs);  backward dk = -da_in, dpos = da_in + dvp (dq and dv are the
 * incoming gradients themselves). */
int cmr_vecattn_mix_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* pos, int64_t ldp,
                        float* a_in, float* vp, int64_t rows, int C, hipStream_t stream);
int cmr_vecattn_mix_bwd_f32(const float* da, int64_t ldda, const float* dvp, int64_t lddv, float* dk, float* dpos, int64_t rows, int C,
                            hipStream_t stream);
/* dy = dz * LeakyReLU'(z) (+ add): activation backward where no BatchNorm sits in front (identity shortcut, PointNN.py:271). */
int cmr_act_bwd_f32(const float* dz, int64_t lddz, const float* z, int64_t ldz, float slope, const float* add, int64_t ldadd,
                    float* dy, int64_t lddy, int64_t rows, int C, hipStream_t stream);
/* Backward of [LeakyReLU -> AvgPool2d(ph, pw)] (CMRAgent.py:38-39,44-45,50-51,56-57): g [B,H/ph,W/pw,C] -> dc [B,H,W,C]
 * with d = the activation output. */
int cmr_pool_act_bwd_f32(const float* g, const float* d, float* dc, int B, int H, int W, int C, int ph, int pw, float slope,
                         hipStream_t stream);
/* per-batch column sum; per-batch column max with the arg-max row (first index on ties, as torch.max(dim) on the CPU);
 * backward of the max: dx[b, arg[b,c], c] += g[b,c]   (CMRAgent.py:95). */
int64_t cmr_colarg_workspace_bytes(int B, int N, int C);
int cmr_colsum_f32(const float* x, int64_t ldx, float* out, void* ws, int64_t ws_bytes, int B, int N, int C, hipStream_t stream);
int cmr_colmax_arg_f32(const float* x, int64_t ldx, float* out, int32_t* arg, void* ws, int64_t ws_bytes, int B, int N, int C,
                       hipStream_t stream);
int cmr_add_at_arg_f32(float* dx, int64_t lddx, const int32_t* arg, const float* g, int64_t ldg, int B, int N, int C,
                       hipStream_t stream);
/* Backward of y = act(x W^T + b) on a small row map (rows <= 4096; n % 32 == 0, n <= 128; k in {32, 64, 128}) in ONE launch -- the
 * autograd of nn.Linear / Conv1d(k=1) (+ ReLU / LeakyReLU) in Train_Geo.py:166-174's backward for the transformer / proxy layers:
 *   dYe = dy * (y > 0 ? 1 : slope) (y null: dYe = dy);  dw [n][k] (+)= dYe^T x;  db [n] (+)= column sums of dYe (optional);
 *   dx [rows][k] = dYe w (+ res) (optional; res may alias dx).  w [n][k] is the forward weight (no transposed copy needed).
 * Deterministic (fixed summation orders).  Returns -3 (unsupported) outside the shapes above: the caller composes cmr_act_bwd_f32,
 * cmr_linear_wgrad_f32 and cmr_linear_f32. */
int cmr_linear_bwd_rows_f32(const float* dy, int64_t lddy, const float* y, int64_t ldy, float slope, const float* x, int64_t ldx,
                            const float* w, int64_t ldw, int64_t rows, int n, int k, float* dw, int64_t lddw, int accumulate_dw,
                            float* db, int accumulate_db, const float* res, int64_t ldres, float* dx, int64_t lddx, hipStream_t stream);
/* nn.Linear / 1x1 conv backward on a handful of rows (<= 1024; the layers after the global pools, CMRAgent.py:57-59,70-86):
 * dYe = dY * LeakyReLU'(Y) (Y null: none); dW = dYe^T [X1|X2]; db = sum dYe; dX1 / dX2 (+)= dYe W. */
int cmr_linear_bwd_small_f32(const float* x1, int64_t ldx1, int k1, const float* x2, int64_t ldx2, int k2, const float* y,
                             int64_t ldy, float slope, const float* dy, int64_t lddy, const float* w, int64_t ldw, float* dw,
                             int64_t lddw, float* db, float* dx1, int64_t lddx1, float* dx2, int64_t lddx2, int acc_dx, int rows,
                             int n, hipStream_t stream);
/* Loss of one minibatch and its gradient w.r.t. logits / value (Train_Agent.py:268-302 with CMRAgent.py:129-144):
 * cross-entropy against the expert + alpha (PPO clip + w_value MSE - w_entropy entropy).  logits rows [B, d*S]; actions
 * int64 [B,d]; old_logprob [B, dr+dt]; returns / adv [B]; out [8] = (loss, clone, policy, value, entropy, ppo, 0, 0);
 * gradients are multiplied by grad_scale. */
int cmr_agent_loss_f32(const float* r_logits, int64_t ldr, const float* t_logits, int64_t ldt, const float* value, int64_t ldv,
                       const int64_t* expert_r, const int64_t* expert_t, const int64_t* act_r, const int64_t* act_t,
                       const float* old_logprob, const float* returns, const float* adv, float* d_r, int64_t lddr, float* d_t,
                       int64_t lddt, float* d_v, int64_t lddv, float* out, int B, int dr, int dt, int S, float alpha,
                       float clip_eps, float w_value, float w_entropy, float grad_scale, hipStream_t stream);
/* torch.optim.Adam step (Train_Agent.py:121-127, :305) over the flat parameter bucket: g <- grad_scale * g + wd * p;
 * m, v moments; p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps).  n % 4 == 0 (the bucket pads every tensor).  grad_clip > 0
 * clamps the scaled gradient to [-clip, clip] first (nn.utils.clip_grad_value_, Train_Geo.py:172). */
int cmr_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, float bias_correction1, float bias_correction2, float grad_scale, float grad_clip,
                 hipStream_t stream);
/* torch.optim.SGD step (the 'SGD' branch of Train_Agent.py:111-117 / Train_Geo.py:65-71: momentum, L2 weight decay, dampening 0,
 * no Nesterov) over the flat parameter bucket: g <- grad_scale * g (clamped when grad_clip > 0) + wd * p; buf = g on the first
 * step, momentum * buf + g afterwards; p -= lr * buf.  n % 4 == 0. */
int cmr_sgd_f32(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float weight_decay, float grad_scale,
                float grad_clip, int first_step, hipStream_t stream);
/* Transposed shadow of the matrix parameters of a flat bucket (operands of the data-gradient GEMMs): table [nslots][5] int64 =
 * (src offset, n, k, dst offset, first tile index), tiles of 32 x 32; dst[k][n] = src[n][k] for every slot, one launch. */
int cmr_transpose_slots_f32(const float* src, float* dst, const int64_t* table, int nslots, int64_t total_tiles,
                            hipStream_t stream);
/* nn.Conv2d(3x3, stride 1, pad 1) weight gradient on the matrix cores: dw [Cout][Cin][3][3] = sum over the minibatch
 * pixels of dy (x) shifted x (NHWC maps, W >= 2, Cin in {32,64,128}, Cout % 32 == 0). */
int64_t cmr_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
/* The same gradient with the products on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16: operands rounded to bf16, fp32 accumulate;
 * the bf16 training mode of BASELINE configs[2]).  x / dy stay fp32 NHWC; Cin in {64, 128}; workspace of
 * cmr_conv3x3_wgrad_workspace_bytes. */
int cmr_conv3x3_wgrad_bf16_f32(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, float* dw, void* ws,
                               int64_t ws_bytes, hipStream_t stream);
/* The same launch + the bias gradient db [Cout] = sum of dy over all pixels (torch.nn.Conv2d's bias.grad; autograd of
 * Train_Agent.py:296-305): every dy element is staged by exactly one lane of the weight-gradient kernel, which keeps fp32 running
 * sums per channel -- replaces a cmr_colsum_f32 pass over dy.  db null: identical to the entry point above. */
int cmr_conv3x3_wgrad_bias_bf16_f32(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, float* dw, float* db, void* ws,
                                    int64_t ws_bytes, hipStream_t stream);
/* The same with x given as the INPUT of a train-mode BatchNorm whose LeakyReLU(xslope) output was the convolution's operand and was never
 * stored (forward: cmr_conv3x3_bf16_pro_nhwc_f32; models/CMRAgent.py:34-56 conv -> BatchNorm -> LeakyReLU -> conv under Train_Agent.py:296-305):
 * the operand lrelu(x * xscale + xshift) (xscale, xshift [Cin] = stat[2], stat[3] of cmr_bn_stats_f32) is formed in the staging pass with the
 * arithmetic of cmr_affine_act_f32 -- bit-identical gradients to the call above on the stored map.  Cin = 128, Cout % 64 == 0, maps of
 * >= 32 768 pixels; CMR_EUNSUPPORTED (-3) otherwise. */
int cmr_conv3x3_wgrad_bias_bf16_pro_f32(const float* x, const float* xscale, const float* xshift, float xslope, const float* dy, int B, int H,
                                        int W, int Cin, int Cout, float* dw, float* db, void* ws, int64_t ws_bytes, hipStream_t stream);
int cmr_conv3x3_wgrad_f32(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, float* dw, void* ws,
                          int64_t ws_bytes, hipStream_t stream);
/* The same gradient in the Winograd domain (round 6; Train_Geo.py:166-174 through models/ImageResNet.py:5-40: the forward and data-gradient
 * convolutions run on F(2x2,3x3), this is the transpose of that identity): dw = sum over the 2x2 output tiles of G^T[(A dy A^T) (.) (B^T x B)]G
 * -- 16 position GEMMs over the tiles (16 of the 36 multiplies of the direct sum) on v_mfma_f32_32x32x2_f32, fp32 accumulate, partial sums
 * reduced in double in a fixed order, the fold with G once per launch.  Served: Cin = Cout = 64, H and W even; CMR_EUNSUPPORTED (-3)
 * otherwise (the caller uses cmr_conv3x3_wgrad_f32).  Error against the float64 sum: 1.5 x the direct fp32 kernel's at K = 856 064 tiles
 * (tools/wino_wgrad_check.py). */
int64_t cmr_conv3x3_wgrad_wino_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int cmr_conv3x3_wgrad_wino_f32(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, float* dw, void* ws,
                               int64_t ws_bytes, hipStream_t stream);
/* nn.Conv1d(k=1) / nn.Linear weight and bias gradient over a row map: dw [n][k] (+)= dy^T x, db [n] (+)= column sums of dy
 * (db optional), any n, k: one launch over (row slices) x (n blocks of 128) x (k blocks of 128) + one deterministic
 * reduction of the slices. */
int64_t cmr_linear_wgrad_workspace_bytes(int64_t rows, int n, int k);
int cmr_linear_wgrad_f32(const float* dy, int64_t lddy, int n, const float* x, int64_t ldx, int k, int64_t rows, float* dw,
                         int64_t lddw, int accumulate, float* db, int accumulate_db, void* ws, int64_t ws_bytes,
                         hipStream_t stream);
/* nn.Conv2d weight [Cout][Cin][3][3] -> operand layouts of the forward kernels (w9 [9][Co'][Ci'] and the Winograd
 * U fragments); transpose = 1 packs the data-gradient convolution W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx].  bf16_frag
 * (optional) receives the bf16 A fragments of cmr_conv3x3_bf16_nhwc_f32 for cout groups of 32 * bf16_nt. */
int cmr_pack_conv3x3_f32(const float* w, int Cout, int Cin, int transpose, float* w9, float* ufrag, void* bf16_frag, int bf16_nt,
                         hipStream_t stream);

/* ---- dataset-side geometry of one frame ------------------------------------------ */
/* dataset/KittiDataset.py:273-276, :284, :312-336 (same code in NuScenesDataset.py): velodyne -> camera transform of the
 * down-sampled cloud (choice = the np.random.choice indices, null = all points in order), projection with the 1/4-scale
 * intrinsics, round-half-even pixel, in-picture test, pc_mask, img_mask (coo_matrix(...).toarray() > 0), the random pose.
 * raw: planar float32 rows [>=3][ld_raw] as stored in the .npy; tr12 / k9 / prand12 are HOST arrays (float64, row-major
 * 3x4 / 3x3 / 3x4) copied by value; everything else is device memory.  Arithmetic in float64 like numpy.
 * Outputs: pc_cam / pc_out planar float32 [3][N], pc_mask int64 [N], xy float64 [2][N] (pc_[0:2]), img_mask int64 [h*w]. */
int cmr_dataset_project_f64(const float* raw, int64_t ld_raw, const int64_t* choice, const double* tr12, const double* k9,
                            const double* prand12, int w, int h, float* pc_cam, float* pc_out, int64_t* pc_mask, double* xy,
                            int64_t* img_mask, int64_t N, hipStream_t stream);
/* KittiDataset.py:338-345: pc_idx_for_circle_loss = np.where(is_in_picture)[0][perm[:nsel]] (ordered compaction by a
 * scan, compact_ws int32 [N]), the float32 pixel coordinates of those points and their np.round as int64.  count receives
 * the number of in-picture points; samples whose perm entry is >= count are marked with index -1. */
int cmr_dataset_circle_select_f64(const int64_t* pc_mask, const double* xy, const int64_t* perm, int nsel, int64_t N,
                                  int32_t* compact_ws, int64_t* count, int64_t* idx_out, float* xy_float, int64_t* xy_int,
                                  hipStream_t stream);

/* ---- bf16 matrix-core variant of the 3x3 convolution (BASELINE configs[2] / [3]) ---- */
/* Same contract as cmr_conv3x3_nhwc_f32 (fp32 NHWC in / out, stride 1|2, folded-BN bias, residual, LeakyReLU, positional
 * table, optional fused 2x2 average pool at stride 1) with the products on v_mfma_f32_32x32x16_bf16 (operands rounded to
 * bf16, fp32 accumulate).  wfrag: bf16 A fragments [Cout/(32 nt)][9][Cin/16][nt][64][8]
 * (cmr_agent_amd/models/_pack.py:conv_bf16_frags).  Served shapes: stride 1: (Cin 64, nt 1|2), (Cin 128, nt 1); stride 2:
 * (Cin 64, nt 1|2); anything else returns CMR_EUNSUPPORTED.  cu_budget: CUs the persistent two-team / matrix-class kernels may occupy
 * (0 = all), as for cmr_conv3x3_wino_nhwc_f32. */
int cmr_conv3x3_bf16_nhwc_f32(const float* x, int B, int H, int W, int Cin, const void* wfrag, int nt, const float* bias,
                              const float* res, const float* post, float* y, int Cout, int stride, float slope, int pool,
                              int cu_budget, hipStream_t stream);

/* ---- geometric-model update: backward pieces (reference Train_Geo.py:166-174) ---------------------- */
/* The training forward of MultiHeadModel is the op-level composition of the inference entry points above with BatchNorm
 * in batch-statistics mode (cmr_agent_amd/train/geo_update.py); these are the backward kernels of its ops.  `accumulate`
 * flags add into the destination (a value that feeds several consumers, a parameter used twice). */
int cmr_axpy_f32(float* y, int64_t ldy, const float* x, int64_t ldx, float alpha, int64_t rows, int C, hipStream_t stream);
/* y = act(x) (ReLU / LeakyReLU / erf-GELU / elu+1) and dx (+)= dy act'(x) with x the PRE-activation. */
int cmr_act_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int C, int act, float act_param, hipStream_t stream);
int cmr_act_bwd_x_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dx, int64_t lddx, int64_t rows, int C, int act,
                      float act_param, int accumulate, hipStream_t stream);
/* nn.LayerNorm(64) backward (ImageViT.py:140-141, IMGPCEncoder.py:86-87, LinearAttention.py:33-34). */
int64_t cmr_layernorm64_bwd_workspace_bytes(int64_t rows);
int cmr_layernorm64_bwd_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, float eps, float* dx,
                            int64_t lddx, int accumulate_dx, float* dgamma, float* dbeta, int accumulate_params, int64_t rows, void* ws,
                            int64_t ws_bytes, hipStream_t stream);
/* F.normalize(dim=1) backward over 64 channels (MultiHeadModel.py:233,241). */
int cmr_l2norm64_bwd_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dx, int64_t lddx, int accumulate, int64_t rows,
                         hipStream_t stream);
/* out[b,2y,2x,:] = g[b,y,x,:], zero elsewhere: a stride-2 convolution's data / weight gradients are the stride-1 ones of its
 * zero-inserted output gradient (ImageResNet.py:13,24-36). */
int cmr_zero_insert2_f32(const float* g, float* out, int B, int Ho, int Wo, int H, int W, int C, hipStream_t stream);
/* adjoints of cmr_patchify_nhwc_f32, of the nearest up-sampling half of cmr_upsample_concat_f32, and the 3-channel im2col pair
 * that turns the stem's convolutions (ImageResNet.py:5-40 with 3 input channels) into row GEMMs in the training path. */
int cmr_patchify_bwd_f32(const float* dpatches, float* dx, int B, int H, int W, int C, int P, int accumulate, hipStream_t stream);
int cmr_upsample_bwd_f32(const float* dcat, int64_t ldc, int coff, float* dproxy, int B, int H, int W, int C2, int scale, int accumulate,
                         hipStream_t stream);
int cmr_im2col3_f32(const float* x4, float* cols, int B, int H, int W, hipStream_t stream);
int cmr_col2im3_f32(const float* dcols, float* dx4, int B, int H, int W, int accumulate, hipStream_t stream);
/* softmax attention backward (forward cmr_mha_f32; ImageViT.py:81-108, IMGPCEncoder.py:36-58): ws = 16 floats per query row. */
int cmr_mha_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o, int64_t ldo,
                    const float* dout, int64_t lddo, float* dq, int64_t lddq, int acc_dq, float* dk, int64_t lddk, int acc_dk, float* dv,
                    int64_t lddv, int acc_dv, float* ws, int64_t ws_bytes, int B, int Tq, int Tk, hipStream_t stream);
/* linear attention core backward (forward cmr_la_reduce_f32 + cmr_la_apply_f32; LinearAttention.py:53-60). */
int64_t cmr_la_bwd_workspace_bytes(int B, int L);
int cmr_la_bwd_f32(const float* qf, int64_t ldq, const float* kf, int64_t ldk, const float* v, int64_t ldv, const float* kvsum,
                   const float* dmsg, int64_t lddm, float* dqf, int64_t lddq, int acc_dq, float* dkf, int64_t lddk, int acc_dk, float* dv,
                   int64_t lddv, int acc_dv, void* ws, int64_t ws_bytes, int B, int L, int S, float eps, hipStream_t stream);
/* group / neighbourhood softmax backward (forward cmr_segment_softmax_f32; PointNN.py:171-182, :227-229). */
int cmr_segment_softmax_bwd_f32(const float* attn, const float* vp, const int32_t* order, const int32_t* offsets, int fixed_len, float scale,
                                const float* dout, float* dattn, float* dvp, int64_t nseg, hipStream_t stream);
/* focal / circle loss backward (forward cmr_focal_metrics_f32 / cmr_circle_loss_f32): gradients w.r.t. the 2-class logits rows; the
 * circle loss ADDS its gradient into the dense maps d_pc_feat [B*N,64] / d_img_feat [B,h,w,64] at the sampled positions. */
int cmr_focal_bwd_f32(const float* logits, int64_t ld, const int64_t* label, float alpha, int64_t rows, float grad_scale, float* dlogits,
                      int64_t ldd, hipStream_t stream);
int64_t cmr_circle_bwd_workspace_bytes(int B, int n);
int cmr_circle_loss_bwd_f32(const float* pc_feat, const float* img_feat, const int64_t* pc_idx, const int64_t* xy_int, const float* xy_float,
                            int B, int N, int h, int w, int n, float dist_thres, float pos_margin, float neg_margin, float log_scale,
                            float grad_scale, float* d_pc_feat, float* d_img_feat, void* ws, int64_t ws_bytes, hipStream_t stream);

/* ---- IterModel: pose cost volume (models/IterModel.py:24-475) ------------------------------------------------
 * The reference warps the predicted-overlap points of ONE pair under nlabel^3 sampled poses, scatter-means their features onto the
 * 1/4-scale image grid, stacks [image features | warped features | occupancy | image overlap] per pose and runs a Conv3d chain with
 * (1, 3, 3) kernels.  Here the volume is a batch of P = nlabel^3 NHWC maps on cmr_conv3x3_*_nhwc_f32; these entry points are the
 * stages around the convolutions.
 * cmr_iter_sample_poses_f32 (IterModel.py:132-173): delta_r / delta_t [nlabel] = 2 amp / (nlabel - 1) * (-(nlabel-1)/2 .. (nlabel-1)/2),
 *   rt [P][3][4] = rows 0..2 of the inverse of [R_y(delta_r[i]) | (delta_t[j], 0, delta_t[k])], p = (i nlabel + j) nlabel + k.
 * cmr_iter_warp_scatter_f32 (IterModel.py:273-345): sel = mask if any(mask) else standby; for selected points in view of pose p,
 *   acc[p][y w + x][0:64] += feat[n], cnt += 1, occ += score[n] (pc planar [3][N], feat rows [N][64]; the accumulators are cleared here).
 * cmr_iter_finalize_f32 (IterModel.py:347-377 and the one-channel input halves of cost_volume_convs[0]): acc <- acc / max(cnt, 1) when
 *   acc is given; res[p][cell][0:64] = base[cell] + sum_taps w1[tap][c] plane[p][cell + tap] (zero padding), plane [P][h][w].
 * cmr_iter_head_f32 (IterModel.py:62-66): global average of channels 0..7 of x [P][cells][ldc], 1x1 conv 8 -> 4, LeakyReLU, 4 -> 1.
 * cmr_iter_decide_f32 (IterModel.py:175-193, 391-473): label_out [P] = label_r (x) (label_tx (x) label_tz); out_f = [cross entropy of
 *   logits vs arg-max(label_out), chosen ry, tx, tz]; out_i = [label, arg-max of the three softmax marginals, arg-max of the joint];
 *   matrix_i [4][4] = inverse of the chosen step.
 * cmr_iter_apply_f32 (IterModel.py:466-472): pc_out = matrix_i[0:3, 0:3] pc + matrix_i[0:3, 3]; acc_out = matrix_i acc_in. */
int cmr_iter_sample_poses_f32(const float* r_amp, const float* t_amp, int nlabel, float* delta_r, float* delta_t, float* rt,
                              hipStream_t stream);
int cmr_iter_warp_scatter_f32(const float* pc, const float* feat, const float* score, const uint8_t* mask, const uint8_t* standby,
                              uint8_t* sel, const float* rt, const float* Kmat, float* acc, float* cnt, float* occ, int N, int P,
                              int h, int w, hipStream_t stream);
/* cmr_iter_warp_bin_f32: cmr_iter_warp_scatter_f32 + the per-pose cmr_iter_finalize_f32 in one launch and without global atomics: a workgroup
 * bins the points of one pose that land in its band of map rows in LDS and writes warped [P][h*w][64] = scatter mean, occ [P][h*w] and
 * res [P][h*w][64] = base + 3x3 stencil of the occupancy plane with w1 [9][64], each once (w <= 384). */
int cmr_iter_warp_bin_f32(const float* pc, const float* feat, const float* score, const uint8_t* mask, const uint8_t* standby,
                          uint8_t* sel, const float* rt, const float* Kmat, const float* w1, const float* base, float* warped,
                          float* res, float* occ, int N, int P, int h, int w, hipStream_t stream);
int cmr_iter_finalize_f32(float* acc, const float* cnt, const float* plane, const float* w1, const float* base, float* res, int P,
                          int h, int w, hipStream_t stream);
int cmr_iter_head_f32(const float* x, int ldc, int cells, const float* w24, const float* b24, const float* w26, const float* b26,
                      float slope, float* logits, int P, hipStream_t stream);
int cmr_iter_decide_f32(const float* logits, int nlabel, const float* label_r, const float* label_tx, const float* label_tz,
                        const float* delta_r, const float* delta_t, float* label_out, float* out_f, int64_t* out_i,
                        float* matrix_i, hipStream_t stream);
int cmr_iter_apply_f32(const float* matrix_i, const float* pc, float* pc_out, int N, const float* acc_in, float* acc_out,
                       hipStream_t stream);

/* cmr_conv3x3_bf16_nhwc_f32 with the activations optionally STORED as bf16 NHWC (x_bf16 / y_bf16 != 0): for chains of bf16 convolutions
 * (conv a -> conv b of a ResidualBlock, ImageResNet.py:9-14; the eight convolutions of the agent's 2-D embedding, CMRAgent.py:34-56).  The
 * consumer rounds its fp32 input to bf16 (RNE) anyway, so a producer that writes those bf16 values gives bit-identical results with
 * half the bytes.  Served: stride 1 with any combination, stride 2 (64 -> 64 k) with fp32 or bf16 input; bias fp32; no table operand
 * with bf16 activations.  res_bf16 != 0: the residual is a bf16 NHWC map too (bf16-STORED towers, round 3: the input of a
 * ResidualBlock is then only ever read as bf16; 64 -> 64 k layers with a bf16 input, stride 1, plain epilogue) -- this one is NOT
 * bit-neutral: the residual enters the fp32 epilogue rounded to bf16.  Otherwise CMR_EUNSUPPORTED (-3). */
int cmr_conv3x3_bf16io_nhwc(const void* x, int x_bf16, int B, int H, int W, int Cin, const void* wfrag, int nt, const float* bias,
                            const void* res, int res_bf16, const float* post, void* y, int y_bf16, int Cout, int stride, float slope,
                            int pool, int cu_budget, hipStream_t stream);
/* Train-mode [BatchNorm -> LeakyReLU(in_slope) -> 3x3 convolution (+ bias, LeakyReLU(slope))] with the BatchNorm applied in the
 * convolution's own staging pass (models/CMRAgent.py:34-56 under Train_Agent.py:296-305; ImageResNet.py:5-40): x [B][H][W][Cin] is the
 * BatchNorm INPUT, in_scale / in_shift [Cin] its folded affine (stat[2], stat[3] of cmr_bn_stats_f32); the operand
 * lrelu(x * in_scale + in_shift) is formed while the halo is converted to bf16 (zero padding applied after it) with the fused
 * multiply-add and the select of cmr_affine_act_f32: bit for bit the result of cmr_affine_act_f32 + cmr_conv3x3_bf16_nhwc_f32, without the
 * activated map's write and read.  Stride 1, Cin = 128, Cout % 128 == 0, no residual / table / pool, maps of >= 128 8x32-pixel tiles
 * (x Cout / 128): the matrix-class kernel; CMR_EUNSUPPORTED (-3) otherwise (the caller materialises the activation). */
int cmr_conv3x3_bf16_pro_nhwc_f32(const float* x, const float* in_scale, const float* in_shift, float in_slope, int B, int H, int W, int Cin,
                                  const void* wfrag, int nt, const float* bias, float* y, int Cout, float slope, int cu_budget,
                                  hipStream_t stream);

/* ---- dropout (train mode; the reference trains MultiHeadModel with p = 0.1 in 141 nn.Dropout modules) --------------------------
 * Counter-based masks: element idx of site `site` is kept iff mix64(seed[0], site, idx) >= p 2^32 (csrc/cmr_common.h:cmr_keep); seed is
 * a DEVICE int64 (advanced once per optimizer step by the host, so that a captured hipGraph draws fresh masks on every replay).
 * cmr_dropout_f32: y = x * keep / (1 - p), idx = row * C + column (ImageViT.py:56, :107, :130-132, LinearAttention.py:27-29, :66,
 *   IMGPCEnDecoder.py:38, :54 ...); the backward pass is the same call on the gradient.
 * cmr_mha_dropout_f32 / cmr_mha_dropout_bwd_f32: cmr_mha_f32 / cmr_mha_bwd_f32 with dropout on the attention probabilities
 *   (ImageViT.py:100, PointViT.py:129, IMGPCEncoder.py:47), idx = ((b 8 + head) Tq + query) Tk + key. */
int cmr_dropout_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int C, float p, const int64_t* seed,
                    int64_t site, hipStream_t stream);
int cmr_mha_dropout_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* o,
                        int64_t ldo, int B, int Tq, int Tk, float p, const int64_t* seed, int64_t site, hipStream_t stream);
int cmr_mha_dropout_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o,
                            int64_t ldo, const float* dout, int64_t lddo, float* dq, int64_t lddq, int acc_dq, float* dk,
                            int64_t lddk, int acc_dk, float* dv, int64_t lddv, int acc_dv, float* ws, int64_t ws_bytes, int B, int Tq,
                            int Tk, float p, const int64_t* seed, int64_t site, hipStream_t stream);

/* Weight gradient of a STRIDE-2 3x3 convolution (pad 1; ImageResNet.py:9-14, :24-27 under loss.backward()): x [B][H][W][Cin] (H, W even), dy
 * [B][H/2][W/2][Cout] -> dw [Cout][Cin][3][3].  Contracts over the OUTPUT pixels (cmr_conv3x3_wgrad_f32 on the zero-inserted gradient contracts
 * over all input pixels, three quarters of which multiply zeros).  Workspace: cmr_conv3x3_wgrad_workspace_bytes(B, H / 2, W / 2, Cin, Cout). */
int cmr_conv3x3_wgrad_s2_f32(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, float* dw, void* ws, int64_t ws_bytes,
                             hipStream_t stream);
/* cmr_pack_conv3x3_f32 for EVERY 3x3 convolution of a training step, both orientations, in one launch: table = device [nslots][8] int64
 * {src offset in the flat parameter buffer, Cout, Cin, transpose, w9 offset in dst, U offset in dst (-1: none), bf16 fragment offset in
 * dst_bf16 in bf16 elements (-1: none), bf16_nt}; max_pairs = the largest Cout * Cin.  (Train_Geo.py:166-174 / Train_Agent.py:263-305: the
 * weights move every optimizer step, so the forward kernels' operand layouts are rebuilt per step -- one launch instead of one per
 * convolution and direction.) */
int cmr_pack_conv3x3_slots_f32(const float* src, float* dst, void* dst_bf16, const int64_t* table, int nslots, int64_t max_pairs,
                               hipStream_t stream);

/* ---- train-mode transformer block in fused launches (round 4; reference models/ImageViT.py:61-158, PointViT.py:96-183,
 * IMGPCEncoder.py:14-102 under model.train(), Train_Geo.py:166-174).  Replaces, per block and step, 16 forward and ~20 backward calls of
 * the op-level entry points above by 3 + 4: cmr_ln64_linear_f32, cmr_mha_dropout_f32, cmr_vit_out_ffn16_train_f32 forward;
 * cmr_vit_ffn_bwd16_f32, cmr_mha_dropout_bwd_f32, cmr_vit_lnqkv_bwd_f32, cmr_wgrad_group_f32 backward.  Dropout masks are those of
 * cmr_dropout_f32 (same (seed, site, row * width + column) -> same mask), p = 0 or seed null: no dropout. ---- */
/* Weights from the flat parameter bucket `src` into MFMA fragment order in `dst`, every slot of a step in ONE launch.  table: device
 * [nslots][10] int64 {src offset, n, k (shape of this slot's block of the packed matrix W'), row stride of the stored matrix, dst offset of
 * W' (multiple of 4), kind, transpose (the block is the transpose of the stored [k][n] matrix), elements (n k; kind 2: n rounded up to 4),
 * ktot (columns of the whole W'), koff (first column of the block inside W'; row blocks go through the dst offset)}; kind 0 = 32x32x2
 * fragments (_pack.frag_pack), 1 = 16x16x4 fragments (_pack.frag_pack16), 2 = plain copy of n floats.  max_elements = the largest
 * `elements`.  A stacked matrix ([Wq; Wk; Wv], or its transpose for the data gradient) is several slots writing one W'. */
int cmr_pack_frags_f32(const float* src, float* dst, const int64_t* table, int nslots, int64_t max_elements, hipStream_t stream);
/* x1 = x + drop_proj(ctx Wo^T + bo);  out = x1 + drop_fc2(W2 drop_act(gelu(W1 LN(x1) + b1)) + b2)   (ImageViT.py:103-108, 128-133, 150-157).
 * cmr_vit_out_ffn16_f32 with the three nn.Dropout sites applied in place and x1 written for the backward.  Weights: frag16 fragments. */
int cmr_vit_out_ffn16_train_f32(const float* ctx, int64_t ldc, const float* x, int64_t ldx, const float* wo_f16, const float* bo,
                                const float* ln_g, const float* ln_b, float eps, const float* w1_f16, const float* b1,
                                const float* w2_f16, const float* b2, float* out, int64_t ldo, float* x1, int64_t ldx1, int64_t rows,
                                float p_proj, float p_mlp, const int64_t* seed, int64_t site_proj, int64_t site_act, int64_t site_fc2,
                                hipStream_t stream);
/* Backward of that launch from d out: recomputes LN(x1), the fc1 pre-activations and the masks from x1;
 *   dm = drop_fc2(d out);  d hidden = dm W2;  du = drop_act(d hidden) gelu'(u);  d LN = du W1;  d x1 = d out + LayerNorm_bwd(d LN);
 *   da = drop_proj(d x1);  d ctx = da Wo.
 * Writes d x1, d ctx and the operands of the weight gradients: gs [rows][1024] (hidden activations as fc2 saw them), du [rows][1024],
 * h = LN(x1) [rows][64], dm [rows][64], da [rows][64]; lnpart [ceil(rows / 16)][128] = per-tile sums of (d gamma | d beta) of ffn_norm
 * (summed by cmr_wgrad_group_f32).  w2t / w1t / wot: frag16 fragments of W2^T, W1^T, Wo^T (cmr_pack_frags_f32, transpose = 1).  The
 * seven row outputs must have room for whole 16-row tiles (ceil(rows / 16) * 16 rows): rows past the end are written, never read. */
int cmr_vit_ffn_bwd16_f32(const float* dout, int64_t lddo, const float* x1, int64_t ldx1, const float* ln_g, const float* ln_b, float eps,
                          const float* w1_f16, const float* b1, const float* w2t_f16, const float* w1t_f16, const float* wot_f16,
                          float* dx1, int64_t lddx1, float* dctx, int64_t lddc, float* gs, float* du, float* h, float* dm, float* da,
                          float* lnpart, int64_t rows, float p_proj, float p_mlp, const int64_t* seed, int64_t site_proj,
                          int64_t site_act, int64_t site_fc2, hipStream_t stream);
/* Backward of attention_norm + the q / k / v projections for up to two row sets in one launch (self block: x with d = [dq | dk | dv],
 * k = 192; cross block: x with dq, k = 64, and y with [dk | dv], k = 128; IMGPCEncoder.py:93-94: both through the SAME LayerNorm):
 *   d LN(x) = d W_cat (W_cat = the projections' weights stacked, passed as frag32 fragments of W_cat^T);  dx = LayerNorm_bwd(d LN(x)) (+ res,
 * the gradient arriving on the residual stream; x set only).  Also writes LN(x) / LN(y) (operands of the projections' weight gradients)
 * and lnpart [tiles_x + tiles_y][128] (32-row tiles; per-tile sums of d gamma | d beta). */
int cmr_vit_lnqkv_bwd_f32(const float* d_x, int64_t ldd_x, int k_x, const float* wt_f_x, const float* x, int64_t ldx, const float* res,
                          int64_t ldres, float* dx, int64_t lddx, float* xn, int64_t ldxn, int64_t rows_x, const float* d_y,
                          int64_t ldd_y, int k_y, const float* wt_f_y, const float* y, int64_t ldy, float* dy, int64_t lddy, float* yn,
                          int64_t ldyn, int64_t rows_y, const float* gamma, const float* beta, float eps, float* lnpart,
                          hipStream_t stream);
/* Weight / bias gradients of up to 8 row-map linears (dw [n][k] (+)= dy^T x, db (+)= column sums of dy: cmr_linear_wgrad_f32's contract)
 * and up to 4 "vector jobs" (out_a | out_b [len] (+)= sums over nparts rows of part [nparts][2 len]: the LayerNorm parameter gradients the
 * row kernels left per tile) in ONE call of two kernels.  desc is a HOST array: nprob x 12 int64 {dy, lddy, n, x, ldx, k, rows, dw, lddw,
 * accumulate, db (0: none), accumulate_db}, then nvec x 6 int64 {part, nparts, len, out_a, out_b, accumulate}; pointers as integers.
 * Deterministic (fixed slices, fixed summation order, no atomics). */
int64_t cmr_wgrad_group_workspace_bytes(const int64_t* desc, int nprob, int nvec);
int cmr_wgrad_group_f32(const int64_t* desc, int nprob, int nvec, void* ws, int64_t ws_bytes, hipStream_t stream);

/* ---- train-mode linear-attention layer in fused launches (round 4; reference models/LinearAttention.py:38-73 under model.train(),
 * Train_Geo.py:166-174): 2 forward + 4 backward calls per layer instead of ~16 + ~20 op-level ones. ---- */
/* cmr_la_kv_state_f32 that also writes kf = elu(Wk y) + 1 and v = Wv y [B S][64] for the backward. */
int cmr_la_kv_state_train_f32(const float* y, int64_t ldy, const float* wk, const float* wv, float* kvsum, float* kf, float* v,
                              void* workspace, int64_t workspace_bytes, int B, int S, hipStream_t stream);
/* cmr_la_query_layer_f32 with the layer's three nn.Dropout(p) sites applied (attention message after LayerNorm 1, MLP hidden activations,
 * MLP output; masks of cmr_dropout_f32, element index = row * width + column) and the activations the backward needs written once per row:
 * qf = elu(Wq x) + 1, msg, mm = Wm msg, d1 = drop(LN1(mm)), o = drop(W3 hid) [B L][64] and hid = drop(relu(W0 [x | d1])) [B L][128]; these six
 * buffers must have room for whole 32-row tiles (ceil(B L / 32) * 32 rows; rows past the end are written, never read). */
int cmr_la_query_layer_train_f32(const float* x, int64_t ldx, const float* kvsum, const float* wq, const float* wmerge,
                                 const float* ln1_g, const float* ln1_b, const float* w_mlp0, const float* w_mlp3,
                                 const float* ln2_g, const float* ln2_b, float* out, int64_t ldo, float* qf, float* msg, float* mm,
                                 float* d1, float* hid, float* o, int B, int L, int S, float eps, float ln_eps, float p,
                                 const int64_t* seed, int64_t site_att, int64_t site_hid, int64_t site_out, hipStream_t stream);
/* Backward of the query side's MLP half from d out (one pass over the rows): LayerNorm-2 backward, output dropout, W3^T, ReLU / hidden
 * dropout, W0^T, attention dropout, LayerNorm-1 backward, Wm^T.  Writes d_o [rows][64] (gradient at W3's output), d_hid [rows][128] (at
 * W0's output), d_mm [rows][64] (at Wm's output) -- the dy operands of the three weight gradients --, d_msg [rows][64] (gradient of the
 * attention message, input of cmr_la_bwd_f32), d_xa [rows][64] (gradient of x through the MLP) and lnpart1 / lnpart2 [min(256, ceil(rows / 256))][128]
 * (per-workgroup sums of d gamma | d beta of norm1 / norm2, summed by cmr_wgrad_group_f32).  The five row outputs must have room for whole
 * 32-row tiles (ceil(rows / 32) * 32 rows): rows past the end are written, so that no store is predicated. */
int cmr_la_mlp_bwd_f32(const float* dout, int64_t lddo, const float* o, const float* hid, const float* mm, const float* wmerge,
                       const float* w_mlp0, const float* w_mlp3, const float* ln1_g, const float* ln2_g, float* d_o, float* d_hid,
                       float* d_mm, float* d_msg, float* d_xa, float* lnpart1, float* lnpart2, int64_t rows, float ln_eps, float p,
                       const int64_t* seed, int64_t site_att, int64_t site_hid, int64_t site_out, hipStream_t stream);
/* Backward of the q / k / v projections for one or two row sets in one launch: per set dx = sum_i (d_i * elu1'(f_i)) W_i + res0 + res1, where
 * f_i is the SAVED activation elu(z) + 1 (derivative 1 where f > 1, else f; 0 = no activation) and W_i^T is passed as frag32 fragments
 * (cmr_pack_frags_f32).  desc: HOST array, 23 int64 per set: rows, nterm (1..3), dx, lddx, res0, ldr0, res1, ldr1, then 3 x {d, ldd, f, wt_f,
 * e_out}; e_out (0: none; may equal d) receives d_i * elu1'(f_i), the dy operand of W_i's weight gradient. */
int cmr_la_proj_bwd_f32(const int64_t* desc, int nprob, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CMR_HIP_H */
