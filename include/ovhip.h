/* ovhip.h — C ABI of libovhip.so: the MI355X (gfx950) encode-and-contrast path of OpenVision.
 *
 * Drop-in boundary.  The reference has no FFI layer: its "operator API" for this path
 * is the Python surface of open_clip.model.CLIP / open_clip.loss.ClipLoss, whose arithmetic is
 * delegated to torch aten kernels.  Each entry point below replaces the aten work behind one piece of
 * that surface (reference paths relative to /root/reference/src/convert_upload/open_clip/):
 *
 *   ov_layernorm        LayerNorm / LayerNormFp32.forward                     transformer.py:15-30
 *   ov_gemm             nn.Linear (c_fc/c_proj/out_proj/in_proj), `@ proj`    transformer.py:225,232-236,645-646
 *   ov_rowstats+ov_gemm_ln  ln_1 -> in_proj and ln_2 -> c_fc fused (LN folded into the GEMM epilogue) transformer.py:263-264
 *   ov_attention        nn.MultiheadAttention core (softmax(qk^T/sqrt(hd)) v) transformer.py:239-252
 *   ov_im2col_patches   conv1 (stride = kernel = P) operand gather            transformer.py:469,610-612
 *   ov_cls_rows         class_embedding concat + pos-emb row 0                transformer.py:615-617
 *   ov_mean_pool        _global_pool 'avg' / 'tok'                            transformer.py:599-603
 *   ov_text_embed       token_embedding(text) + positional_embedding         model.py:272-274
 *   ov_gather_rows      text_global_pool 'last' / 'first'                     transformer.py:655-658
 *   ov_l2norm           F.normalize(x, dim=-1)                                model.py:267,284
 *   ov_logits           CLIP.get_logits (scale * img @ txt^T)                 model.py:286-293
 *   ov_clip_loss        ClipLoss.get_logits + cross_entropy both ways         loss.py:102-131
 *   ov_gemm_fp8         the same nn.Linear on fp8 e4m3 operands (config #5)           transformer.py:225,232-236
 *   ov_preprocess_image transforms.Resize -> ToTensor -> Normalize (Pillow-exact)  ov-zero-shot-test.py:72-77, transform.py:355-392
 *   ov_class_mean_normalize / ov_topk   zero-shot classifier weights, argmax / recall@k ranking   zero_shot_classifier.py:54-57,
 *                       src/evaluators/proj/image_text/{discriminative_classifier.py:305-323, image_text_retrieval.py:24-87}
 *   ov_tower_*          Transformer.forward (the resblock loop)               transformer.py:355-366, 254-265
 *   ov_vision_* / ov_text_*   VisionTransformer.forward / CLIP.encode_text    transformer.py:609-651, model.py:269-284
 *
 * Conventions: plain pointers and sizes only (no torch types).  All pointers are DEVICE pointers owned
 * by the caller unless stated; every call enqueues on `stream` (a hipStream_t passed as void*) and
 * returns without synchronising.  Return value: 0 = OK, negative = error (see ov_error_string); nothing
 * throws across the boundary.  No call allocates device memory: workspaces are sized by
 * ov_*_workspace_bytes and provided by the caller.  Activations are bf16 row-major; LN affine params,
 * biases and final embeddings are fp32 (the reference's bf16 mode keeps LN in fp32: model.py:143).
 */
#ifndef OVHIP_H
#define OVHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OV_ABI_VERSION 2       /* 2: logit scale / upstream loss gradient are device scalars (ov_clip_loss*, ov_logits) */

typedef void* ov_stream_t;            /* hipStream_t */
typedef uint16_t ov_bf16;             /* raw bfloat16 bits */

enum ov_status {
    OV_OK = 0,
    OV_ERR_INVALID = -1,              /* bad argument (null pointer, size, alignment) */
    OV_ERR_UNSUPPORTED = -2,          /* shape outside what the kernels cover */
    OV_ERR_WORKSPACE = -3,            /* workspace too small */
    OV_ERR_NO_DEVICE = -4,            /* no gfx950 device visible */
    OV_ERR_HIP = -1000                /* -(1000 + hipError_t) */
};

enum ov_dtype { OV_F32 = 0, OV_BF16 = 1 };

enum ov_epilogue {
    OV_EPI_BIAS = 0,                  /* C = bf16(acc + bias)                (bias may be NULL) */
    OV_EPI_BIAS_GELU_ERF = 1,         /* C = bf16(gelu_erf(acc + bias))      vision MLP, vit.py:202 */
    OV_EPI_BIAS_GELU_TANH = 2,        /* C = bf16(gelu_tanh(acc + bias))     text MLP, text_transformer.py:117 */
    OV_EPI_BIAS_RESIDUAL = 3,         /* C = bf16(bf16(acc + bias) + R)      x = x + f(x), transformer.py:263-264 */
    OV_EPI_GELU_GRAD_ERF = 4,         /* C = bf16(bf16(acc + bias) * gelu_erf'(R))   backward through the MLP's GELU: R = the c_fc */
    OV_EPI_GELU_GRAD_TANH = 5         /* C = bf16(bf16(acc + bias) * gelu_tanh'(R))  pre-activation (ov_gemm_keep), acc = dy Wproj */
};

int         ov_abi_version(void);
const char* ov_error_string(int status);
/* 0 if a gfx950 device is usable by this process, else OV_ERR_NO_DEVICE / OV_ERR_HIP-x. */
int         ov_device_check(void);

/* ---- granular operators ------------------------------------------------------------------------ */

/* y[r,:] = (x[r,:] - mean) * rsqrt(var + eps) * gamma + beta ; fp32 statistics (biased variance).
 * x_dtype/y_dtype: OV_BF16 or OV_F32.  D % 8 == 0, D <= 8192.  ldx/ldy in elements. */
int ov_layernorm(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta,
                 void* y, int y_dtype, int64_t ldy, int64_t rows, int D, float eps, ov_stream_t stream);

/* C[M,N] = epilogue(A[M,K] * W[N,K]^T + bias[N]).  A, W, C, R bf16 row-major; bias fp32 or NULL.
 * K % 64 == 0; lda, ldw, ldc, ldr % 8 == 0; N % 8 == 0.
 * Row remapping for the patch-embed use (0 = identity):
 *   out_group  > 0 : output row = m + m / out_group + 1      (skip one cls row per image)
 *   resid_mod  > 0 : residual row = (m % resid_mod) + resid_off   (pos-emb table broadcast over batch)
 * C may alias R when the row maps are identities (in-place residual add). */
int ov_gemm(const ov_bf16* A, int64_t lda, const ov_bf16* W, int64_t ldw, const float* bias,
            ov_bf16* C, int64_t ldc, int64_t M, int N, int K, int epilogue,
            const ov_bf16* R, int64_t ldr, int out_group, int resid_mod, int resid_off,
            ov_stream_t stream);

/* `batch` independent products C_z = A_z W_z^T (bf16 out, no bias, no epilogue) in one launch; operand z = base + z * stride
 * (elements).  With A_z / W_z = column ranges of one operand pair this is split-K with bf16 partials (ov_linear_backward's dW).
 * Same shape rules as ov_gemm; strides % 8 == 0; batch <= 65535. */
int ov_gemm_batched(const ov_bf16* A, int64_t lda, int64_t stride_a, const ov_bf16* W, int64_t ldw, int64_t stride_w, ov_bf16* C,
                    int64_t ldc, int64_t stride_c, int64_t M, int N, int K, int batch, ov_stream_t stream);

/* Split-K partials of C = P^T Q with BOTH operands row-major over the contraction rows (P [Mc, NI], Q [Mc, NJ]): partial z contracts
 * rows [z * chunk, min(Mc, (z + 1) * chunk)) into C + z * stride_c (bf16 [NI, NJ], no bias).  The weight-gradient product
 * dW = dY^T X of ov_linear_backward without explicit transposes (transposing LDS reads, ds_read_b64_tr_b16).  Mc % 64 == 0,
 * chunk % 64 == 0, (batch - 1) * chunk < Mc <= batch * chunk, NI / NJ / leading dimensions % 8 == 0.
 * psum (or NULL): fp32 [batch][NI], psum[z][i] = sum of P[m, i] over the rows of partial z -- the bias gradient sum_m dY[m, i] from the
 * fragments the product already holds (one more MFMA against an all-ones operand), instead of a second pass over dY. */
int ov_gemm_tn_batched(const ov_bf16* P, int64_t ldp, const ov_bf16* Q, int64_t ldq, ov_bf16* C, int64_t ldc, int64_t stride_c, int64_t Mc,
                       int NI, int NJ, int64_t chunk, int batch, float* psum, ov_stream_t stream);

/* LayerNorm folded into the following Linear (LN(x) W^T + b without materialising LN(x)):
 *   C = epilogue( rstd[m] * (x W'^T - mean[m] * colsum[n]) + cvec[n] ),
 *   W'[n,k] = bf16(gamma[k] W[n,k]), colsum[n] = sum_k W'[n,k], cvec[n] = sum_k beta[k] W[n,k] + b[n]   (built once at pack time),
 *   rowstats[m] = {mean, rstd} of row m of x from ov_rowstats (fp32 [M,2]).  Epilogues BIAS / GELU only. */
int ov_gemm_ln(const ov_bf16* X, int64_t ldx, const ov_bf16* Wg, int64_t ldw, const float* cvec, const float* colsum,
               const float* rowstats, ov_bf16* C, int64_t ldc, int64_t M, int N, int K, int epilogue, ov_stream_t stream);

/* ov_gemm with a GELU epilogue (OV_EPI_BIAS_GELU_ERF / _TANH) that also keeps the pre-activation:
 *   C = bf16(gelu(A W^T + bias)),  C2 = bf16(A W^T + bias)   [M, N], ld = ldc2
 * -- the c_fc of the training forward, so that the backward does not run that product again (transformer.py:232-236). */
int ov_gemm_keep(const ov_bf16* A, int64_t lda, const ov_bf16* W, int64_t ldw, const float* bias, ov_bf16* C, int64_t ldc,
                 ov_bf16* C2, int64_t ldc2, int64_t M, int N, int K, int epilogue, ov_stream_t stream);

/* rowstats[r] = {mean, rsqrt(var + eps)} of x[r, 0:D] (bf16 rows, fp32 two-pass statistics, biased variance). */
int ov_rowstats(const ov_bf16* x, int64_t ldx, float* rowstats, int64_t rows, int D, float eps, ov_stream_t stream);

/* The same statistics through partial sums (the LayerNorm of transformer.py:15-30 in front of the folded QKV / c_fc GEMMs, without a
 * separate pass over the residual stream): parts[r][D / 32] = {sum, sum of squares} (fp32) of the bf16 values x[r, 32 g .. 32 g + 31], in
 * ONE fixed association order -- ov_gemm_rowparts leaves exactly these numbers for its output rows (from its epilogue where the kernel
 * can, by this pass otherwise: bitwise the same), ov_rowstats_finalize turns them into {mean, rsqrt(var + eps)} with var = E[x^2] -
 * mean^2 (one pass; fp32).  D % 32 == 0. */
int ov_rowparts(const ov_bf16* x, int64_t ldx, float* parts, int64_t rows, int D, ov_stream_t stream);
int ov_rowstats_finalize(const float* parts, float* rowstats, int64_t rows, int D, float eps, ov_stream_t stream);
/* ov_gemm(..., OV_EPI_BIAS_RESIDUAL, R, ldr, no row maps) that also writes parts[M][N / 32] of its OUTPUT rows (N % 32 == 0). */
int ov_gemm_rowparts(const ov_bf16* A, int64_t lda, const ov_bf16* W, int64_t ldw, const float* bias, ov_bf16* C, int64_t ldc,
                     int64_t M, int N, int K, const ov_bf16* R, int64_t ldr, float* parts, ov_stream_t stream);

/* Non-causal, unmasked multi-head self-attention on a packed qkv activation.
 * qkv: [B*L, 3*H*hd] bf16 (q | k | v column blocks, heads contiguous inside each, ld = ld_qkv);
 * out: [B*L, H*hd] bf16 (heads merged, ld = ld_out).  scale multiplies q.k (hd^-0.5).
 * hd == 64 takes the tuned kernels; any other multiple of 8 up to 96 (So400m: 72, H/14: 80) a generic one. */
int ov_attention(const ov_bf16* qkv, int64_t ld_qkv, ov_bf16* out, int64_t ld_out,
                 int B, int L, int H, int hd, float scale, ov_stream_t stream);
/* ov_attention that also keeps lse[B*H][Lp] (Lp = L rounded up to 32, fp32): the log-sum-exp of every query row's scaled scores in
 * log2 units, for ov_attention_backward_saved.  head_dim 64 and L <= 288 only (OV_ERR_UNSUPPORTED otherwise). */
int ov_attention_lse(const ov_bf16* qkv, int64_t ld_qkv, ov_bf16* out, int64_t ld_out, float* lse, int B, int L, int H, int hd, float scale,
                     ov_stream_t stream);

/* Patch gather for conv1: image [B,3,S,S] (img_dtype) -> P[B*g*g, Kpad] bf16,
 * column index c*P*P + i*P + j (the flattening of conv1.weight[D,3,P,P]); columns >= 3*P*P zeroed. */
int ov_im2col_patches(const void* image, int img_dtype, ov_bf16* out, int B, int S, int P, int Kpad,
                      ov_stream_t stream);

/* x[b*L + 0, :] = bf16(bf16(cls[:]) + bf16(pos[0,:]))  for every image b. cls/pos fp32. */
int ov_cls_rows(ov_bf16* x, int64_t ldx, const float* cls, const float* pos0, int B, int L, int D,
                ov_stream_t stream);

/* out[b,:] = mean over tokens first..L-1 of x[b*L + t, :]  (first = 1: skip cls; 'avg' pooling). fp32 out. */
int ov_mean_pool(const ov_bf16* x, int64_t ldx, float* out, int B, int L, int D, int first,
                 ov_stream_t stream);

/* x[b*T + t, :] = bf16(bf16(table[tokens[b,t], :]) + bf16(pos[t, :])).  tokens int64; table/pos bf16.
 * Token ids outside [0, V) set *err_flag (device int, may be NULL) and are clamped. */
int ov_text_embed(const int64_t* tokens, const ov_bf16* table, const ov_bf16* pos, ov_bf16* x,
                  int64_t ldx, int B, int T, int D, int V, int* err_flag, ov_stream_t stream);

/* out[b, :] = x[b*L + t, :] (row gather; t = L-1 for text 'last' pooling). bf16 -> bf16. */
int ov_gather_rows(const ov_bf16* x, int64_t ldx, ov_bf16* out, int64_t ldo, int B, int L, int t, int D,
                   ov_stream_t stream);

/* bf16 <-> fp32 row conversion helpers (strided rows). */
int ov_convert(const void* src, int src_dtype, int64_t lds, void* dst, int dst_dtype, int64_t ldd,
               int64_t rows, int cols, ov_stream_t stream);

/* y[r,:] = x[r,:] / max(||x[r,:]||_2, 1e-12)   (F.normalize).  x bf16 or fp32, y fp32. */
int ov_l2norm(const void* x, int x_dtype, int64_t ldx, float* y, int64_t ldy, int64_t rows, int E,
              ov_stream_t stream);

/* out[i, j] = scale * (*scale_dev) * <X[i,:], Y[j,:]>  fp32 in/out (CLIP.get_logits: model.py:286-293).  E % 8 == 0.
 * scale_dev: optional device scalar (NULL = 1): exp(logit_scale) stays on the device, as in the reference (model.py:288). */
int ov_logits(const float* X, const float* Y, float* out, int64_t ldo, int n1, int n2, int E, float scale, const float* scale_dev,
              ov_stream_t stream);

/* ov_attention with the output written as e4m3 bytes out8[B*L, H*64] (ld_out in bytes) under the static scale 2 * (*out_amax) / 448
 * (device scalar; fp8 path: the out-proj GEMM then reads it through ov_gemm_fp8_static).  head_dim 64; else OV_ERR_UNSUPPORTED. */
int ov_attention_fp8out(const ov_bf16* qkv, int64_t ld_qkv, unsigned char* out8, int64_t ld_out, int B, int L, int H, int hd,
                        float scale, const float* out_amax, float* out_amax_next /* optional running maximum */, ov_stream_t stream);

/* ---- fp8 GEMM (BASELINE.json config #5: fp8 weights/activations on the CDNA4 fp8 MFMA) --------------------------------------
 * C = epilogue(rowscale[m] * colscale[n] * (A . W^T) + bias):  A [M, K], W [N, K] OCP e4m3fn bytes (K contiguous, lda/ldw in
 * bytes, % 16), rowscale [M] / colscale [N] fp32 dequantisation scales (per activation row / per weight row), bias fp32 or NULL,
 * C (and R for OV_EPI_BIAS_RESIDUAL) bf16.  K % 128 == 0, K >= 384, N % 8 == 0.  Epilogues: OV_EPI_BIAS, OV_EPI_BIAS_GELU_ERF,
 * OV_EPI_BIAS_RESIDUAL.  Accumulation in fp32 on v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales). */
int ov_gemm_fp8(const unsigned char* A, int64_t lda, const unsigned char* W, int64_t ldw, const float* rowscale,
                const float* colscale, const float* bias, ov_bf16* C, int64_t ldc, int64_t M, int N, int K, int epilogue,
                const ov_bf16* R, int64_t ldr, ov_stream_t stream);

/* ov_gemm_fp8 with a STATIC activation scale on one side (scale = 2 * amax / 448, amax a calibrated device scalar):
 * out_amax != NULL: C is e4m3 bytes [M, N] (ldc in bytes, % 16; N % 16 == 0) quantised with out_amax's scale, epilogue
 *                   OV_EPI_BIAS_GELU_ERF / _TANH -- the c_fc -> c_proj hand-over without a bf16 round trip and a re-quantisation pass;
 * in_amax  != NULL: every row of A carries in_amax's scale (rowscale ignored), epilogue OV_EPI_BIAS_RESIDUAL, C / R bf16.
 * Exactly one of the two is set. */
int ov_gemm_fp8_static(const unsigned char* A, int64_t lda, const unsigned char* W, int64_t ldw, const float* rowscale,
                       const float* in_amax, const float* colscale, const float* bias, void* C, int64_t ldc, const float* out_amax,
                       float* out_amax_next /* optional: running maximum of |C| for the next call */, int64_t M, int N, int K,
                       int epilogue, const ov_bf16* R, int64_t ldr, ov_stream_t stream);
/* cur[i] = max(cur[i], next[i]), i < n (delayed scaling: roll the recorded maxima into the scales at the top of a forward). */
int ov_amax_roll(float* cur, const float* next, int n, ov_stream_t stream);

/* Activation quantisation for ov_gemm_fp8: q[r,:] = e4m3(y[r,:] / s_r), s_r = max|y[r,:]| / 448, with y = x (ov_quant_rows_fp8) or
 * y = LayerNorm(x) * gamma + beta in fp32 (ov_layernorm_quant_fp8; eps, biased variance as transformer.py:15-30).  x bf16
 * [rows, D] (ldx), q bytes [rows, D] (ldq, % 8), rowscale [rows] fp32.  D % 8 == 0, D <= 8192. */
/* amax_acc (device float, may be NULL): running maximum of |x| over every row quantised so far (calibration of static scales). */
int ov_quant_rows_fp8(const ov_bf16* x, int64_t ldx, unsigned char* q, int64_t ldq, float* rowscale, int64_t rows, int D,
                      float* amax_acc, ov_stream_t stream);
int ov_layernorm_quant_fp8(const ov_bf16* x, int64_t ldx, const float* gamma, const float* beta, unsigned char* q, int64_t ldq,
                           float* rowscale, int64_t rows, int D, float eps, ov_stream_t stream);

/* ---- image front-end: Resize [+ CenterCrop] -> ToTensor -> Normalize on the device ----------------
 * Bit-exact counterpart of PIL.Image.resize (what torchvision's Resize runs on a PIL image: reference ov-zero-shot-test.py:72-77,
 * open_clip/transform.py:355-392) followed by x / 255 and (x - mean) / std in IEEE fp32.
 * src: uint8 RGB [H, W, 3] (device).  bounds_x [Wr][2] = {first source column, tap count}, coef_x [Wr][ksize_x] = Pillow's 22-bit
 * fixed-point taps (device int32), same for y over Hr output rows; row0/nrows = the source rows the vertical pass reads
 * (Pillow's ybox); tmp: uint8 [nrows, Wr, 3] scratch (device).  The [out_h, out_w] window at (crop_x, crop_y) of the resized
 * [Hr, Wr] image is written as CHW [3, out_h, out_w] in out_dtype (OV_F32 / OV_BF16).  mean / stdv: HOST float[3]. */
int ov_preprocess_image(const unsigned char* src, int H, int W, const int* bounds_x, const int* coef_x, int ksize_x, int Wr,
                        const int* bounds_y, const int* coef_y, int ksize_y, int Hr, int row0, int nrows, unsigned char* tmp,
                        int crop_x, int crop_y, int out_h, int out_w, const float* mean, const float* stdv, void* out,
                        int out_dtype, ov_stream_t stream);

/* ---- consumers of encode + logits: zero-shot classifier weights and ranking ------------
 * out[c,:] = normalize(mean_t emb[c*T + t, :])   (open_clip/zero_shot_classifier.py:54-57).  fp32 in/out. */
int ov_class_mean_normalize(const float* emb, float* out, int C, int T, int E, ov_stream_t stream);
/* Per-row top-k of x[rows, cols] (fp32): idx_out[rows, k] int64 (and val_out[rows, k], optional), best first;
 * largest != 0 ranks by value descending (logits), else ascending (distances, as image_text_retrieval.py:44,78 argsort);
 * ties go to the smaller index.  k <= cols. */
int ov_topk(const float* x, int64_t ldx, int rows, int cols, int k, int largest, int64_t* idx_out, float* val_out,
            ov_stream_t stream);

/* Local-strip InfoNCE (ClipLoss with local_loss semantics; world_size 1 = plain ClipLoss).
 *   img, txt      : this rank's L2-normalised embeddings [b, E] fp32
 *   all_img/all_txt: gathered embeddings [N, E] fp32 in rank order (== img/txt when N == b)
 *   logit_scale   : DEVICE scalar, the multiplier exp(CLIP.logit_scale) (model.py:315) -- ABI 2: it never crosses the host, so
 *                   the step enqueues without a stream drain (the reference keeps it on the device too: loss.py:120-131)
 *   labels are i + label_offset (label_offset = b * rank, loss.py:93-94)
 *   loss_out[0] = (CE(scale*img@all_txt^T) + CE(scale*txt@all_img^T)) / 2      (fp32, device)
 *   lse_out (optional, may be NULL): [4, b] fp32 = lse_img, diag_img, lse_txt, diag_txt
 * workspace: ov_clip_loss_workspace_bytes(b, N) bytes. Logits are never materialised. */
size_t ov_clip_loss_workspace_bytes(int b, int N);
int ov_clip_loss(const float* img, const float* txt, const float* all_img, const float* all_txt,
                 int b, int N, int E, const float* logit_scale, int label_offset, float* loss_out,
                 float* lse_out, void* workspace, size_t workspace_bytes, ov_stream_t stream);

/* Backward of ov_clip_loss (loss.py:102-131 differentiated; the reference gets it from autograd): with P = softmax - onehot of
 * each [b, N] strip (recomputed, never materialised; lse_terms = the [4, b] block ov_clip_loss wrote) and c = grad_loss *
 * logit_scale / (2 b) (both DEVICE scalars; grad_loss NULL = 1):   d_img = c P_i all_txt,  d_txt = c P_t all_img            (local rows, [b, E], always written)
 *                        d_all_txt = c P_i^T img, d_all_img = c P_t^T txt          (gathered rows, [N, E]; NULL = skip)
 *                        d_scale = grad_loss / (2 b) * sum P .* (x . y)            (device scalar; NULL = skip)
 * The caller routes the gathered-side terms (gather_features, loss.py:19-63: own chunk only, or reduce-scatter when
 * gather_with_grad).  E % 32 == 0, E <= 1152; OV_ERR_UNSUPPORTED otherwise.  Deterministic (no atomics). */
size_t ov_clip_loss_backward_workspace_bytes(int b, int N);
int ov_clip_loss_backward(const float* img, const float* txt, const float* all_img, const float* all_txt, int b, int N, int E,
                          const float* logit_scale, int label_offset, const float* lse_terms, const float* grad_loss, float* d_img,
                          float* d_txt, float* d_all_img, float* d_all_txt, float* d_scale, void* workspace,
                          size_t workspace_bytes, ov_stream_t stream);

/* ---- operator-level backward of the block (the reference gets these from torch autograd through nn.Linear,
 * nn.LayerNorm and nn.GELU: transformer.py:15-30, 232-236).  bf16 activations and gradients, fp32 arithmetic and parameter-gradient
 * sums, deterministic (two-stage reductions in a fixed order). ---------------------------------------------------------------- */

/* out[c, r] = in[r, c]; columns rows..pad64(rows)-1 of out are written as zeros (the GEMM's K granule).  cols % 8 == 0,
 * ld_out >= pad64(rows). */
int ov_transpose_bf16(const ov_bf16* in, int64_t ld_in, int64_t rows, int cols, ov_bf16* out, int64_t ld_out, ov_stream_t stream);

/* y = x W^T + b  (x [M, K], W [N, K], dY [M, N]):  dX = dY W  [M, K],  dW = dY^T x  [N, K] (bf16),  db = column sums of dY (fp32 [N]).
 * Any of dX / dW / db may be NULL (skipped; x is only read for dW).  N % 64 == 0, K % 64 == 0.  Both products run on ov_gemm, fed by LDS-staged transposes
 * held in the workspace (ov_linear_backward_workspace_bytes). */
size_t ov_linear_backward_workspace_bytes(int64_t M, int N, int K);
int ov_linear_backward(const ov_bf16* dY, int64_t lddy, const ov_bf16* X, int64_t ldx, const ov_bf16* W, int64_t ldw, int64_t M, int N,
                       int K, ov_bf16* dX, int64_t lddx, ov_bf16* dW, int64_t lddw, float* db, void* workspace,
                       size_t workspace_bytes, ov_stream_t stream);

/* Backward of ov_layernorm on bf16 rows: dx = rstd (q - mean(q) - xhat mean(q xhat)) with q = dy * gamma, dgamma = sum_rows dy * xhat,
 * dbeta = sum_rows dy (fp32 [D]).  D % 8 == 0, D <= 4096. */
size_t ov_layernorm_backward_workspace_bytes(int64_t rows, int D);
int ov_layernorm_backward(const ov_bf16* x, int64_t ldx, const float* gamma, const ov_bf16* dy, int64_t lddy,
                          const ov_bf16* dres /* optional: added to dx (the residual branch's gradient) */, int64_t lddres, ov_bf16* dx,
                          int64_t lddx, float* dgamma, float* dbeta, int64_t rows, int D, float eps, void* workspace,
                          size_t workspace_bytes, ov_stream_t stream);

/* Backward of ov_attention (unmasked softmax attention of nn.MultiheadAttention, transformer.py:225,239-252): from the packed
 * qkv [B*L, 3*H*hd], the forward output out [B*L, H*hd] and the upstream gradient dout, writes dqkv [B*L, 3*H*hd] = (dQ | dK | dV).
 * The row log-sum-exp is recomputed (the forward keeps none).  hd % 8 == 0, hd <= 96.  hd == 64 and L <= 288: one kernel with the
 * head resident in LDS, no workspace; otherwise (long sequences; head dims 72 / 80 of So400m / H-14, zero-padded to 96 in LDS) two
 * streaming kernels and a workspace of ov_attention_backward_workspace_bytes (row lse and delta).  Deterministic. */
size_t ov_attention_backward_workspace_bytes(int B, int L, int H, int hd);
int ov_attention_backward(const ov_bf16* qkv, int64_t ld_qkv, const ov_bf16* out, int64_t ld_out, const ov_bf16* dout, int64_t ld_dout,
                          ov_bf16* dqkv, int64_t ld_dqkv, int B, int L, int H, int hd, float scale, void* workspace,
                          size_t workspace_bytes, ov_stream_t stream);
/* The same with the forward's row statistics kept by ov_attention_lse (lse [B*H][L rounded up to 32] fp32, or NULL): the resident
 * kernel (head_dim 64, L <= 288) then skips its own score pass; the streaming kernels ignore it. */
int ov_attention_backward_saved(const ov_bf16* qkv, int64_t ld_qkv, const ov_bf16* out, int64_t ld_out, const ov_bf16* dout, int64_t ld_dout,
                                ov_bf16* dqkv, int64_t ld_dqkv, const float* lse, int B, int L, int H, int hd, float scale, void* workspace,
                                size_t workspace_bytes, ov_stream_t stream);

/* da = dh * gelu'(a) on the pre-activation a [rows, N] (tanh_form = 0: exact erf GELU, vision; 1: tanh form, text).  N % 8 == 0.
 * h_out (optional) receives gelu(a).  da may alias dh and h_out may alias a (element-wise, in place). */
int ov_gelu_backward(const ov_bf16* a, int64_t lda, const ov_bf16* dh, int64_t lddh, ov_bf16* da, int64_t ldda, ov_bf16* h_out,
                     int64_t ldh, int64_t rows, int N, int tanh_form, ov_stream_t stream);

/* ---- in-situ kernel timing (used by bench.py for the roofline object; off by default) ------------------
 * ov_profile_enable(mask, n): bracket every launch of the selected classes inside ov_tower_forward with a pair
 * of HIP events recorded on the launch stream (n = max launches recorded; resets earlier records; mask 0 = off).
 * ov_profile_read(cls, &ms, &count, &rows): synchronises the recorded events and returns the summed device time and
 * the summed row count M of those launches (launches on the internal tail stream are not recorded). */
enum ov_profile_class {
    OV_PROF_LN = 0, OV_PROF_GEMM_QKV = 1, OV_PROF_ATTN = 2, OV_PROF_GEMM_OUT = 3, OV_PROF_GEMM_FC = 4 /* erf-GELU c_fc (vision) */, OV_PROF_GEMM_PROJ = 5,
    OV_PROF_GEMM_FC_TANH = 6 /* tanh-GELU c_fc (text) */
};
int ov_profile_enable(unsigned class_mask, int max_records);
int ov_profile_read(int cls, double* total_ms, int* count, double* total_rows /* sum of launch M, may be NULL */);

/* Diagnostics: device buffer [workgroup][slots][8] of s_memtime stamps written by the persistent GEMM (NULL = off). */
int ov_debug_gemm_stamps(unsigned long long* buf, int slots);
/* Diagnostics: per-wave epilogue timeline [workgroup][slots][8 waves][8] (NULL = off; `slots` as given to ov_debug_gemm_stamps). */
int ov_debug_gemm_wave_stamps(unsigned long long* wbuf);

/* ---- parameter update of the training step.  The reference's trainer chains optax transforms
 * (src/optim/build_optax.py:272-278, applied at src/main_clip.py:480-483): clip_by_global_norm -> scale_by_adam(b1, b2,
 * mu_dtype=bfloat16) -> add_decayed_weights(wd) -> scale(lr) -> scale_by_schedule -> scale(-1).  One fused pass over flat fp32
 * buffers: p -= lr * ( mu_hat / (sqrt(nu_hat) + eps) + wd * p ), mu kept in bf16, nu in fp32, bias correction by `step` (>= 1).
 * grad_scale multiplies every gradient first (1 / world_size after a SUM all-reduce); gnorm_sq (device scalar from ov_sumsq over the
 * same unscaled gradients, NULL = off) enables global-norm clipping at clip_norm without a host round trip.  HBM-bound: 24 B / element. */
int ov_adamw_step(float* p, const float* g, ov_bf16* mu, float* nu, int64_t n, float lr, float b1, float b2, float eps, float wd,
                  int step, float grad_scale, const float* gnorm_sq, float clip_norm, ov_stream_t stream);
/* out[0] = (accumulate ? out[0] : 0) + sum_i g[i]^2   (two-stage fixed-order reduction: deterministic). */
size_t ov_sumsq_workspace_bytes(void);
int ov_sumsq(const float* g, int64_t n, float* out, int accumulate, void* workspace, size_t workspace_bytes, ov_stream_t stream);

/* ---- tower level (the resblock loop and the two encoders) --------------------------------------- */

typedef struct ov_tower ov_tower;     /* opaque host object holding BORROWED device weight pointers */

typedef struct {
    int width;        /* D */
    int layers;
    int heads;
    int mlp;          /* true hidden width (int(D*mlp_ratio)) */
    int mlp_pad;      /* leading dimension / padded hidden width, % 64 == 0, >= mlp */
    int gelu_tanh;    /* 0 = erf (vision), 1 = tanh (text) */
    float ln_eps;     /* 1e-6 */
} ov_tower_cfg;

typedef struct {      /* one ResidualAttentionBlock; weights bf16 [out,in] row-major, LN/bias fp32 */
    const float *ln1_w, *ln1_b;
    const ov_bf16* qkv_w;  const float* qkv_b;     /* [3D, D], [3D]   attn.in_proj_*        */
    const ov_bf16* out_w;  const float* out_b;     /* [D, D],  [D]    attn.out_proj.*       */
    const float *ln2_w, *ln2_b;
    const ov_bf16* fc_w;   const float* fc_b;      /* [mlp_pad, D], [mlp_pad] (pad rows = 0) */
    const ov_bf16* proj_w; const float* proj_b;    /* [D, mlp_pad] (pad cols = 0), [D]      */
    /* Optional LN fold (both NULL = off).  When set, qkv_w / fc_w hold gamma-scaled weights W', qkv_b / fc_b hold cvec, and
     * these are the column sums of W' (see ov_gemm_ln); ln1_* / ln2_* are then unused by the tower. */
    const float* qkv_colsum;                       /* [3D]      */
    const float* fc_colsum;                        /* [mlp_pad] */
} ov_block_weights;

typedef struct {      /* optional fp8 (OCP e4m3) copies of a block's four weight matrices for the fp8 path (config #5): bytes
                       * [out, in] row-major + one fp32 dequantisation scale per output row; qkv_b / fc_b are the module's own
                       * biases (ov_block_weights holds the LN-folded ones), out/proj biases are taken from ov_block_weights */
    const unsigned char* qkv_w8;  const float* qkv_s;  const float* qkv_b;     /* [3D, D], [3D], [3D]           */
    const unsigned char* out_w8;  const float* out_s;                         /* [D, D], [D]                   */
    const unsigned char* fc_w8;   const float* fc_s;   const float* fc_b;      /* [mlp_pad, D], [mlp_pad] x 2   */
    const unsigned char* proj_w8; const float* proj_s;                        /* [D, mlp_pad], [D]             */
} ov_block_fp8;

ov_tower* ov_tower_create(const ov_tower_cfg* cfg);
void      ov_tower_destroy(ov_tower* t);
int       ov_tower_set_block(ov_tower* t, int layer, const ov_block_weights* w);
/* Once EVERY layer has an fp8 copy (and until one is cleared with q == NULL) ov_tower_forward runs the fp8 path: LayerNorm fused
 * with row quantisation, fp8 GEMMs (ov_gemm_fp8), bf16 attention, row re-quantisation in front of out_proj / c_proj.  Needs
 * width and mlp_pad % 128 == 0 and >= 384.  ov_tower_workspace_bytes grows accordingly: query it after setting the copies. */
int       ov_tower_set_block_fp8(ov_tower* t, int layer, const ov_block_fp8* q);
/* Static (delayed) scales of the two re-quantised activations of the fp8 path.  amax: device float[4 * layers] (borrowed):
 * [layer] = MLP hidden, [layers + layer] = attention output, [2 * layers ...) = the same two sets as recorded during the
 * running forward (rolled into the first half at the top of the next one).  mode 0: off (both are written in bf16 and re-quantised row by row); 1: same, and
 * their running maxima are recorded into amax (calibration); 2: the producers (c_fc epilogue, attention epilogue for head_dim 64)
 * write e4m3 directly with the scale 2 * amax / 448 and the consumers (c_proj, out_proj) read it with that scalar scale; 3: as 2
 * but FROZEN: the recorded maxima are never rolled into the scales (bitwise repeatable, batch-composition invariant results). */
int       ov_tower_set_fp8_hidden_scale(ov_tower* t, float* amax, int mode);
/* Mixed precision of the fp8 path: mask[layer] (HOST array of n == layers bytes, copied) says which of the block's four GEMMs take e4m3
 * operands; the others run exactly as in the bf16 path (LN fold included).  NULL restores OV_FP8_ALL for every layer.  The c_fc ->
 * c_proj hand-over stays in e4m3 only where both are in the mask; a lone fp8 c_proj / out_proj re-quantises its bf16 input row by
 * row (or, out_proj with mode >= 2 and head_dim 64, reads the attention kernel's e4m3 output).  Reference: none (the reference has no
 * fp8 mode); shape source scripts/project/openvision/train.sh:18 (BASELINE.json config #5). */
#define OV_FP8_QKV 1
#define OV_FP8_OUT 2
#define OV_FP8_FC 4
#define OV_FP8_PROJ 8
#define OV_FP8_ALL 15
int       ov_tower_set_fp8_mask(ov_tower* t, const unsigned char* mask, int n);
size_t    ov_tower_workspace_bytes(const ov_tower* t, int B, int L);
/* x[B*L, D] bf16 is updated in place through all `layers` blocks. */
int       ov_tower_forward(const ov_tower* t, ov_bf16* x, int B, int L, void* workspace,
                           size_t workspace_bytes, ov_stream_t stream);

/* ---- backward of one ResidualAttentionBlock (transformer.py:254-265 differentiated by autograd in the
 * reference).  Activation recomputation: the caller keeps only the block INPUT x [B*L, D]; ln_1, qkv, attention, x1, ln_2 and the
 * c_fc pre-activation are recomputed with the forward kernels, then ov_gelu_backward / ov_linear_backward / ov_layernorm_backward /
 * ov_attention_backward run the chain rule.  `w` holds the module's own weights (no LN fold: qkv_colsum == fc_colsum == NULL,
 * qkv_b / fc_b the module biases).  Gradients: weights bf16 [out, in] (same layout as the weights), biases and LN parameters fp32;
 * all written (not accumulated; rows / columns of the MLP padding, mlp .. mlp_pad, come out as zeros when the weights' padding is
 * zero).  dx may alias dy.  width % 64 == 0, head_dim % 8 == 0 and <= 96, mlp_pad % 64 == 0; OV_ERR_UNSUPPORTED otherwise. */
typedef struct {
    float *ln1_w, *ln1_b;          /* [D] */
    ov_bf16* qkv_w; float* qkv_b;  /* [3D, D], [3D] */
    ov_bf16* out_w; float* out_b;  /* [D, D], [D] */
    float *ln2_w, *ln2_b;          /* [D] */
    ov_bf16* fc_w;  float* fc_b;   /* [mlp, D], [mlp] */
    ov_bf16* proj_w; float* proj_b;/* [D, mlp], [D] */
} ov_block_grads;
typedef struct {      /* optional forward intermediates kept for the backward (the first three together or none) */
    const ov_bf16* qkv;        /* [B*L, 3D]  packed q | k | v */
    const ov_bf16* attn_out;   /* [B*L, D]   attention output before out_proj */
    const ov_bf16* x1;         /* [B*L, D]   x + attention branch */
    const ov_bf16* fc_pre;     /* [B*L, mlp_pad]  c_fc output before GELU (ov_gemm_keep), or NULL = recomputed */
    const ov_bf16* ln1_out;    /* [B*L, D]   ln_1(x), or NULL = recomputed */
    const ov_bf16* ln2_out;    /* [B*L, D]   ln_2(x1), or NULL = recomputed */
    const ov_bf16* fc_act;     /* [B*L, mlp_pad]  gelu(fc_pre), or NULL = recomputed; with fc_pre set too, the backward folds the GELU
                                * derivative into the epilogue of dy Wproj (OV_EPI_GELU_GRAD_*) and runs no element-wise pass */
    const float* attn_lse;     /* [B*heads][L rounded up to 32]  row log-sum-exp from ov_attention_lse, or NULL = recomputed */
} ov_block_saved;
size_t ov_block_backward_workspace_bytes(const ov_tower_cfg* cfg, int B, int L);
int ov_block_backward(const ov_tower_cfg* cfg, const ov_block_weights* w, const ov_bf16* x, const ov_block_saved* saved /* or NULL */,
                      const ov_bf16* dy, ov_bf16* dx, const ov_block_grads* g, int B, int L, void* workspace, size_t workspace_bytes,
                      ov_stream_t stream);

/* Training-side tower entry points.  ov_tower_forward_saving = the tower forward on the caller's stream that keeps, per layer and
 * token, [x | qkv | attention out | x1 | ln_1 out | ln_2 out | c_fc pre-activation | c_fc activation] (8 D + 2 mlp_pad bf16, plus the
 * attention's fp32 row log-sum-exp per layer; `saved` holds ov_tower_saved_bytes: 52 GB for L/14 at batch 256 — sized for the 288 GB of an MI355X: nothing of the forward is run a second time
 * by the backward).  bf16 path; the blocks must hold
 * the module's own, unfolded weights.  ov_tower_backward runs ov_block_backward over the layers in reverse: dx [B*L, D] holds
 * d loss / d (tower output) on entry and d loss / d (tower input) on return; grads[layer] receives that block's parameter gradients
 * (written, not accumulated). */
size_t ov_tower_saved_bytes(const ov_tower* t, int B, int L);
int    ov_tower_forward_saving(const ov_tower* t, ov_bf16* x, ov_bf16* saved, int B, int L, void* workspace, size_t workspace_bytes,
                               ov_stream_t stream);
size_t ov_tower_backward_workspace_bytes(const ov_tower* t, int B, int L);
int    ov_tower_backward(const ov_tower* t, const ov_bf16* saved, ov_bf16* dx, const ov_block_grads* grads, int B, int L,
                         void* workspace, size_t workspace_bytes, ov_stream_t stream);

typedef struct {      /* VisionTransformer front/back ends (OpenVision: no ln_pre, no conv bias) */
    int image_size, patch_size, kpad;              /* kpad = roundup(3*P*P, 64) */
    int pool_avg;                                  /* 1 = 'avg' (skip cls), 0 = 'tok' */
    int final_ln_after_pool;                       /* 1 for OpenVision */
    int embed_dim, embed_pad;                      /* E, roundup(E, 8) */
    const ov_bf16* conv_w;                         /* [D, kpad] bf16, pad cols = 0 */
    const float* cls;                              /* [D] */
    const ov_bf16* pos;                            /* [L, D] bf16 */
    const float* pos_f32;                          /* [L, D] fp32 (row 0 used for the cls row) */
    const float *ln_post_w, *ln_post_b;            /* [D] */
    const ov_bf16* proj_t;                         /* [E, D] bf16 = visual.proj^T */
} ov_vision_head;

size_t ov_vision_workspace_bytes(const ov_tower* t, const ov_vision_head* h, int B);
/* image [B,3,S,S] -> tokens x[B*L, D] bf16 (patch embed + cls + pos); x caller-provided. */
int ov_vision_embed(const ov_tower* t, const ov_vision_head* h, const void* image, int img_dtype, int B,
                    ov_bf16* x, void* workspace, size_t workspace_bytes, ov_stream_t stream);
/* tokens x[B*L, D] -> features [B, E] fp32 (pool -> ln_post -> proj), optionally L2-normalised. */
int ov_vision_head_forward(const ov_tower* t, const ov_vision_head* h, const ov_bf16* x, int B,
                           float* features, int normalize, void* workspace, size_t workspace_bytes,
                           ov_stream_t stream);
/* VisionTransformer.forward + optional F.normalize: image -> [B, E] fp32. */
int ov_encode_image(const ov_tower* t, const ov_vision_head* h, const void* image, int img_dtype, int B,
                    float* features, int normalize, void* workspace, size_t workspace_bytes,
                    ov_stream_t stream);

typedef struct {      /* CLIP text front/back ends */
    int context_length, vocab_size;
    int pool_last;                                 /* 1 = 'last', 0 = 'first' */
    int embed_dim;
    const ov_bf16* token_embedding;                /* [V, D] bf16 */
    const ov_bf16* pos;                            /* [T, D] bf16 */
    const float *ln_final_w, *ln_final_b;          /* [D] */
    const ov_bf16* proj_t;                         /* [E, D] bf16 = text_projection^T */
} ov_text_head;

size_t ov_text_workspace_bytes(const ov_tower* t, const ov_text_head* h, int B);
/* CLIP.encode_text: tokens int64 [B, T] -> [B, E] fp32.  *err_flag (device int, may be NULL) is set
 * to 1 when a token id is outside [0, V). */
int ov_encode_text(const ov_tower* t, const ov_text_head* h, const int64_t* tokens, int B,
                   float* features, int normalize, int* err_flag, void* workspace,
                   size_t workspace_bytes, ov_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OVHIP_H */
