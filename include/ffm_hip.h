/*
 * ffm_hip.h — C ABI of libffm_hip.so, the MI355X (gfx950) kernels behind the
 * FairLoRA local-training hot path.
 *
 * The reference (Harvard-AI-and-Robotics-Lab/FairFedMed) has no FFI: its hot
 * path is PyTorch calls.  Each entry point below replaces the PyTorch ops named
 * in its comment (reference file:line); INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's
 *     allocator); the library allocates nothing and keeps no global state;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work;
 *   - `dtype` selects the activation / frozen-weight type: FFM_F32 or FFM_BF16.
 *     LoRA parameters, LayerNorm parameters, biases, statistics, logits and
 *     all gradients of trainable tensors are always fp32;
 *   - activations are row-major [rows, features]; a row of the ViT token matrix
 *     is (image b, token l) -> row b*L + l ("image-major"; the reference's
 *     [L, B, d] tensors are the same data permuted);
 *   - return value: 0 ok, FFM_EINVAL (-1) bad shape/alignment/dtype,
 *     FFM_EUNSUP (-2) unsupported size, otherwise a positive hipError_t.
 */
#ifndef FFM_HIP_H
#define FFM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFM_F32 0
#define FFM_BF16 1
/* ffm_gemm_nt only: every operand f32 in memory as under FFM_F32, the products on the bf16 matrix cores with both
 * operands split into bf16 hi + lo pairs (hi*hi + lo*hi + hi*lo, f32 accumulation: ~2^-16 instead of the f32 MFMA's
 * 2^-24, at 5x its rate); products of at most 64 rows (the text tower beside a bf16 vision tower), FFM_EUNSUP otherwise */
#define FFM_F32_X3 2
/* IEEE half storage (the reference's PREC="fp16", federated_main.py:85 / clip/model.py:609-630): activations and
 * frozen weights in fp16, fp32 accumulation and fp32 trainable tensors exactly as under FFM_BF16 */
#define FFM_F16 3
/* ffm_gemm_nt only: FFM_F32_X3 with the weight operand `b` stored as IEEE half [N, K] (ldb in half elements): a frozen
 * weight rounded to 11 significant bits - what the reference's own PREC="fp16" holds (clip/model.py:609-630) - is split
 * exactly into the bf16 hi + lo pair in the kernel, so the products are those of FFM_F32_X3 on the rounded weight at
 * half its bytes (the text tower's weights are what its 40-row products move: DESIGN.md section 4.6) */
#define FFM_F32_X3_W16 4

#define FFM_OK 0
#define FFM_EINVAL (-1)
#define FFM_EUNSUP (-2)

#define FFM_MAX_RANK 32
#define FFM_MAX_GROUPS 8

/* library / build identification: returns FFM_ABI_VERSION */
#define FFM_ABI_VERSION 12  /* 12: FFM_EPI_LNB_STAT / FFM_EPI_LNB_APPLY (LayerNorm backward folded into the dX products of the MLP; ffm_gemm_args.lnb_*), ffm_gemm_args.sk_part + ffm_gemm_splitk_floats (text-tower products split over K), ffm_scale_acc, ffm_loss_scale / ffm_unscale_check / ffm_sgd_momentum_gated (device-resident fp16 gradient scale); 11: FFM_EPI_BNBWD (ffm_gemm_args.bn_x / bn_mask / bn_mean / bn_rstd / bn_gout), ffm_bn_bwd part_rows; 10: ffm_lora_down_blocks is exact again, ffm_lora_down_blocks_max sizes buffers, ffm_slice_wgrad_blocks (wpart rows, no longer ffm_slice_blocks); 9: FFM_EPI_LGRAD (ffm_gemm_args.lg_v / lg_part_c / lg_part_a), ffm_gemm_lgrad_rows; 8: FFM_F16 (IEEE-half twins of every 16-bit kernel behind the same entry points), ffm_scale_check; 7: ffm_text_embed / ffm_text_tail_fwd / ffm_text_tail_bwd / ffm_text_ctx_grad; 6: FFM_F32_X3, ffm_gemm_args.lw_wide / LayerNorm folding fields, ffm_pack_desc.dst_wide, ffm_eval_counts_sorted, ffm_gemm_tiles_n, colstat_part / ffm_bn_fwd part_rows; 5: ffm_sgd_momentum_n; 4: ffm_reduce_partials_multi launch width (max_n), new entry points (conv3x3, eval counts, uint8) */
int ffm_abi_version(void);

/* ---- epilogue flags for ffm_gemm_nt ------------------------------------ */
#define FFM_EPI_BIAS      1   /* + bias[n]                         (fp32 [N])        */
#define FFM_EPI_LORA      2   /* + sum_j ts[m][j] * lw[j][n]       (rank-r update)   */
#define FFM_EPI_LORA_KR   4   /* lw is stored [N, r] (lora_A) instead of [r, N]      */
#define FFM_EPI_RESIDUAL  8   /* + res[m][n]                       (dtype, ld = ldc) */
#define FFM_EPI_GELU      16  /* also write c2 = quick_gelu(c)                        */
#define FFM_EPI_DGELU     32  /* c *= quick_gelu'(aux[m][n])       (aux dtype, ldc)  */
#define FFM_EPI_RANKOP    64  /* with LORA: ts is computed in-kernel from the packed rank operand rk  */
/* LayerNorm folded into the two GEMMs around it (bf16 panel kernel only; clip/model.py:354-357 ln_1 / ln_2 in front
 * of attn.in_proj and mlp.c_fc).  With y = LayerNorm(x) = (x - mu) rstd gamma + beta,
 *     y W^T + b = rstd (x W'^T - mu c) + d,   W' = gamma (.) W,  c_n = sum_k W'[n][k],  d_n = sum_k beta_k W[n][k] + b_n,
 * so the consumer multiplies the RAW rows by the gamma-scaled frozen weight and corrects in its epilogue, and the
 * per-row (mu, rstd) come from partial row sums that the producer of x left behind: no LayerNorm launch, no
 * normalised copy of x in HBM. */
#define FFM_EPI_ROWSTATS  128 /* producer: rowstat_part[tn][m] = {sum, sum of squares} of the stored row over this block's columns */
#define FFM_EPI_LNIN      256 /* consumer: b / b_packed hold W', bias holds d, ln_c holds c; rows are normalised in the epilogue */
#define FFM_EPI_LGRAD     512 /* with DGELU | RANKOP: also lg_part_c / lg_part_a, two rank-r gradient partial products (below) */
#define FFM_EPI_BNBWD     1024 /* colstat_part receives the BatchNorm-BACKWARD column sums of the stored output (bn_* below) */
/* ABI 12: LayerNorm's BACKWARD folded into the two dX products around it (the MLP's ln_2, clip/model.py:304-310, 354-357).
 * With h = LayerNorm(x), pre = h W_eff^T + b and g_h = dpre W_eff, autograd's dL/dx = rstd (gamma g_h - c1/K - xhat c2/K)
 * needs two row sums that are linear in what the dX product of c_proj holds per element:
 *   c1 = sum_k g_h gamma      = sum_n dpre[n] (W gamma)[n]      + sum_j us[j] (A^T gamma)[j]
 *   c2 = sum_k g_h gamma xhat = sum_n dpre[n] (pre[n] - d[n])   - sum_j us[j] (A^T beta)[j],     d = W beta + b
 * (us = scaling s_b (dpre B^T), the rank vector the dX product of c_fc forms anyway; W gamma, d, A^T gamma, A^T beta are the
 * vectors the forward's FFM_EPI_LNIN folding already builds).  FFM_EPI_LNB_STAT (producer, with DGELU | RANKOP | LGRAD): the
 * dX product of c_proj leaves lnb_part[tn][m] = {0, sum_n c[m][n] aux[m][n]} over its columns (packed 16-bit dot products on
 * the rows as stored); the two sums against the fixed vectors W gamma and d ride in the consumer's rank operand as rows 14 and
 * 15 (ffm_pack_desc.row14 / row15: t[14] = sum_n c (W gamma), t[15] = sum_n c d come out of its matrix cores for free).  FFM_EPI_LNB_APPLY (consumer, with LORA | LORA_KR | RANKOP): the dX product of c_fc stores
 * rstd (gamma g_h - c1/K - xhat c2/K) + res instead of g_h - one launch (ffm_layernorm_bwd) and a round trip of g_h fewer. */
#define FFM_EPI_LNB_STAT  2048
#define FFM_EPI_LNB_APPLY 4096

typedef struct ffm_gemm_args {
    const void* a;      /* [M, K] dtype, row stride lda (elements) */
    const void* b;      /* [N, K] dtype, row stride ldb: C = A * B^T */
    void*       c;      /* [M, N] dtype, row stride ldc */
    int32_t M, N, K;
    int32_t lda, ldb, ldc;
    int32_t flags;
    int32_t rank;       /* r, <= FFM_MAX_RANK (LORA only) */
    const float* bias;  /* [N] */
    const float* ts;    /* [M, r] fp32: scaling * (xA) * s_b rows (fwd) or scaling * (gB^T) * s_b (bwd) */
    const float* lw;    /* LoRA matrix, [r, N] or (LORA_KR) [N, r], fp32 */
    const void*  res;   /* residual [M, N] dtype, stride ldc */
    void*        c2;    /* GELU: activated output [M, N] dtype, stride ldc */
    const void*  aux;   /* DGELU: pre-activation [M, N] dtype, stride ldc */
    /* FFM_EPI_RANKOP (rank <= 16): t = A . rk^T is accumulated by the GEMM itself (x A forward, g B^T
     * backward), ts = scaling * t * s_b feeds the rank-r update above, `ts` is ignored. */
    const void*  rk;    /* [16, K] dtype, row stride K: packed lora_A^T or lora_B (rows >= r zero), see ffm_lora_pack_multi */
    const float* S;     /* lora_S [G, r] */
    const int32_t* attr;/* [nsamples] group index or NULL (uniform mix) */
    float*       t_out; /* optional [M, r]: t  */
    float*       ts_out;/* optional [M, r]: ts */
    const float* t_fwd; /* optional [M, r]: with ds_part, dS partials = sum_rows pi_b scaling t_fwd t */
    float*       ds_part;/* [ffm_gemm_tiles_m(M), G, r] */
    int32_t G, rows_per_sample;
    float scaling, lambda_group;
    /* optional (bf16): the same matrix as `b` in MFMA-fragment order, written once by ffm_pack_b for FROZEN
     * weights.  When present and the shape fills the chip in one round of large tiles, the panel kernel
     * streams it straight into registers (csrc/gemm_panel.hip); `b` must still be valid. */
    const void*  b_packed;
    /* optional (bf16, FFM_EPI_RANKOP): the LoRA matrix `lw` as [N, 32] dtype rows, row n = the r entries of output
     * column n followed by zeros (ffm_lora_pack_multi's dst_wide): the panel kernel copies its tile of it straight
     * into LDS instead of converting `lw`; `lw` must still be valid. */
    const void*  lw_wide;
    /* FFM_EPI_ROWSTATS: [ffm_gemm_tiles_n][M][2] fp32, written once per call (no accumulation) */
    float*       rowstat_part;
    /* FFM_EPI_LNIN: partial row sums of A [ln_np][M][2] (a producer's rowstat_part, or ffm_embed_lnpre's), c [N];
     * optional outputs mean / rstd [M] (for the LayerNorm backward), eps = 1e-5 */
    const float* ln_part;
    const float* ln_c;
    float*       ln_mean;
    float*       ln_rstd;
    int32_t      ln_np;
    /* FFM_EPI_GELU / FFM_EPI_DGELU, 0 or 1.  0: `c` (GELU) receives the pre-activation and `aux` (DGELU) holds it -
     * c *= quick_gelu'(aux).  1: the forward stores the DERIVATIVE instead, c = quick_gelu'(x) beside c2 = quick_gelu(x)
     * (the sigmoid is shared), and the backward multiplies by what `aux` holds, c *= aux.  Nothing else reads the
     * pre-activation of QuickGELU on this path (clip/model.py:313-332: the reference's autograd saves it for exactly this
     * product).  Same bytes, one more rounding of the derivative in 16-bit storage, bit-identical results in fp32.
     * Measured on MI355X: no gain - the engines leave it 0. */
    int32_t      gelu_deriv;
    /* FFM_EPI_LNIN with FFM_EPI_RANKOP: rk holds (gamma (.) lora_A)^T and ln_rk [2][16] its corrections
     * {c_j = sum_k rk[j][k], d_j = sum_k beta_k lora_A[k][j]} (ffm_lora_pack_multi writes both):
     * t = rstd (x rk^T - mu c) + d = LayerNorm(x) lora_A */
    const float* ln_rk;
    /* optional (128x128 kernel: RN50's 1x1 convolutions): [ffm_gemm_tiles_m][2][N] fp32, the column sums {sum, sum of
     * squares} of the STORED output over each row tile - the batch statistics of the BatchNorm that follows
     * (ffm_bn_fwd's part / part_rows), so that it does not read the tensor once more to form them.  A launch that
     * asks for them always runs on the 128x128 / 128xN kernels (b_packed is then ignored: the panel kernel has no
     * column-sum epilogue; ask ffm_gemm_tiles_m with packed = 0 for the row tiles); together with
     * FFM_EPI_ROWSTATS / FFM_EPI_LNIN: FFM_EUNSUP */
    float*       colstat_part;
    /* FFM_EPI_LGRAD (16-bit panel kernel, with FFM_EPI_DGELU | FFM_EPI_RANKOP, rank % 4 == 0, gelu_deriv == 0; ask
     * ffm_gemm_lgrad_rows first): the dX product of c_proj holds both [M, N] operands of the two LARGE rank-r gradient
     * reductions of an MLP block in registers - the rows it stores (c = dL/d pre) and quick_gelu(aux) (the forward's
     * activation) - so their per-row-tile partial products leave with its epilogue instead of being formed by two
     * launches of ffm_lora_grad_partial that read 2 x M x N elements once more:
     *   lg_part_c[tile][n][j] = sum over the tile's rows of c[m][n] * lg_v[m][j]              (dB of c_fc: lg_v = its forward ts)
     *   lg_part_a[tile][n][j] = sum over the tile's rows of quick_gelu(aux[m][n]) * ts[m][j]  (dA of c_proj: ts = this launch's)
     * both [ffm_gemm_lgrad_rows][N][rank] fp32, written once per call (sum them with ffm_reduce_partials_multi);
     * trainers/GLP_OT_SVLoRA.py:450-482 (the autograd of lora_B / lora_A). */
    const float* lg_v;      /* [M, rank] fp32 */
    float*       lg_part_c;
    float*       lg_part_a;
    /* FFM_EPI_BNBWD (ABI 11; 128x128 kernel, with colstat_part): the stored output IS dL/dy of a train-mode BatchNorm
     * (+ ReLU) whose input bn_x and output bn_mask [M, N] (dtype, stride ldc; bn_mask may be NULL: no ReLU) the caller
     * holds - RN50: the dX product of conv3 produces the gradient of relu(bn2(.)), clip/model.py:41-60.  colstat_part
     * [ffm_gemm_tiles_m][2][N] then receives, per row tile, {sum g, sum g * xhat} with g = c * (bn_mask > 0) and
     * xhat = (bn_x - bn_mean[n]) * bn_rstd[n]: the two column sums ffm_bn_bwd needs (its part / part_rows), so that it does
     * not read dy, x and the mask once more to form them. */
    const void*  bn_x;
    const void*  bn_mask;
    const float* bn_mean;   /* [N] */
    const float* bn_rstd;   /* [N] */
    void*        bn_gout;   /* optional [M, N] dtype, stride ldc: also receives g = c * (bn_mask > 0), the gradient an identity-skip
                             * Bottleneck passes on beside bn3 (ffm_bn_bwd's g_out, which a call with part_rows cannot write) */
    /* ABI 12, FFM_F32_X3 / FFM_F32_X3_W16 products of at most 48 rows (the text tower: 40 token rows against 512..2048-wide
     * frozen weights): optional scratch of ffm_gemm_splitk_floats(M, N, K) floats.  With it, a product whose N is too narrow
     * to give every CU a column tile is split over K across the grid - partial tiles [slice][M][N] land here and a second
     * small launch sums them IN SLICE ORDER (deterministic) and applies the epilogue - instead of a handful of blocks each
     * walking the whole K.  NULL: the one-launch kernel. */
    float*       sk_part;
    /* FFM_EPI_LNB_STAT: lnb_part [ffm_gemm_tiles_n][M][2] fp32 is written (lnb_wg / lnb_d: reserved, not read).
     * FFM_EPI_LNB_APPLY: lnb_part holds lnb_np such partial rows (<= 8 with FFM_EPI_RANKOP; <= 24 on the plain product, whose
     * producer is ffm_attention_bwd_lnstat and whose rank-r corrections do not exist: ln_rk is not read), lnb_x [M, N] (dtype, stride ldc) is the LayerNorm's
     * input, lnb_gamma [N] its weight, ln_mean / ln_rstd [M] its saved statistics (inputs here), ln_rk [2][16] the
     * corrections {A^T gamma, A^T beta} (as under FFM_EPI_LNIN | FFM_EPI_RANKOP), res the gradient that joins behind the
     * LayerNorm (the residual path).  With FFM_EPI_RANKOP (rank <= 14) rows 14 / 15 of rk hold W gamma and W beta + b. */
    const float* lnb_wg;
    const float* lnb_d;
    float*       lnb_part;
    int32_t      lnb_np;
    int32_t      lnb_pad_;
    const void*  lnb_x;
    const float* lnb_gamma;
} ffm_gemm_args;

/*
 * C = epilogue(A * B^T).  Replaces nn.Linear / F.linear on frozen weights
 * (trainers/GLP_OT_SVLoRA.py:451; clip/model.py:352 in/out projections, :445
 * final proj) and, with B = W^T prepared once at load time, the dX products
 * autograd derives from them.  With FFM_EPI_LORA it is the fused
 * (W + A diag(s_b) B) x forward / LoRA-dx backward of FairLoRALinear
 * (trainers/GLP_OT_SVLoRA.py:450-482): the dense dW is never formed.
 * Requires K*sizeof(dtype) % 128 == 0, N % 8 == 0, 16-byte aligned rows.
 */
int ffm_gemm_nt(const ffm_gemm_args* args, int dtype, void* stream);
/* floats ffm_gemm_args.sk_part needs for this product (0: the product is not split over K) */
int64_t ffm_gemm_splitk_floats(int M, int N, int K, int dtype);
/* row tiles (= dS partial rows written under FFM_EPI_RANKOP) of the kernel ffm_gemm_nt picks for this call */
int ffm_gemm_tiles_m(int M, int N, int K, int flags, int rank, int dtype, int packed);
/* row tiles of lg_part_c / lg_part_a under FFM_EPI_LGRAD for this call (flags with or without the bit), or FFM_EUNSUP when
 * the kernel ffm_gemm_nt picks for it has no such epilogue (then leave the bit out and call ffm_lora_grad_partial) */
int ffm_gemm_lgrad_rows(int M, int N, int K, int flags, int rank, int dtype, int packed);
/* column tiles (= rows of rowstat_part under FFM_EPI_ROWSTATS) of that kernel; FFM_EUNSUP when no kernel serves the
 * flags for this shape (FFM_EPI_ROWSTATS / FFM_EPI_LNIN exist in the bf16 panel kernel only: ask before relying on them) */
int ffm_gemm_tiles_n(int M, int N, int K, int flags, int rank, int dtype, int packed);
/* which kernel / tile that is (diagnostics, tests): returns the panel-kernel configuration index (csrc/gemm_panel.h) or
 * -1 for the 128x128 kernel, and writes {tile rows, tile columns, waves per CU} to shape3 (optional) */
int ffm_gemm_tile_shape(int M, int N, int K, int flags, int rank, int dtype, int packed, int32_t* shape3);
/*
 * dst = src [N, K] bf16 (row stride ld) in MFMA-fragment order: [N/16][K/32][64 lanes][8] with lane l holding
 * row (l & 15), k-group (l >> 4).  N % 16 == 0, K % 32 == 0.  Load-time only (weights are frozen).
 */
int ffm_pack_b(const void* src, void* dst, int N, int K, int ld, void* stream);

/*
 * Pack LoRA matrices into the GEMM's rank-operand form: dst [16, K] dtype with dst[j][k] = lora_A[k][j]
 * (layout_rk = 0, src [K, r]) or lora_B[j][k] (layout_rk = 1, src [r, K]); rows >= r are zero.
 * descs_dev is a DEVICE array; one launch packs every adapter of the model.
 */
typedef struct ffm_pack_desc {
    const float* src;
    void* dst;
    int32_t K, r, layout_rk, pad_;
    void* dst_wide;     /* optional [K, 32] dtype: dst_wide[k][j] = the same element as dst[j][k], columns >= r zero */
    /* optional (layout_rk = 0): a LayerNorm folded into the product this operand rides in (ffm_gemm_args.ln_rk):
     * dst[j][k] = gamma[k] * src[k][j], ln_rk[j] = sum_k dst[j][k] (as rounded), ln_rk[16 + j] = sum_k beta[k] src[k][j];
     * dst_wide keeps the unscaled matrix */
    const float* gamma;
    const float* beta;
    float* ln_rk;
    /* optional (ABI 12, r <= 14): two more rows of the packed operand from caller vectors, dst[14][k] = row14[k],
     * dst[15][k] = row15[k] - the rank slots beyond r are masked by every consumer, so the product's t[14] / t[15] are free
     * row-wise dot products with two fixed vectors (FFM_EPI_LNB_APPLY takes W gamma and W beta + b there) */
    const float* row14;
    const float* row15;
} ffm_pack_desc;
int ffm_lora_pack_multi(const ffm_pack_desc* descs_dev, int ndesc, int max_K, int dtype, void* stream);
/* second pass for the descriptors that carry gamma / beta / ln_rk (others return at once): the ln_rk sums, which read
 * what ffm_lora_pack_multi has just written (call it after, on the same stream) */
int ffm_lora_pack_ln(const ffm_pack_desc* descs_dev, int ndesc, int dtype, void* stream);

/*
 * LayerNorm over the last dimension, fp32 statistics, eps 1e-5
 * (clip/model.py:304-310).  y = (x-mean)*rstd*gamma + beta; mean/rstd saved
 * for the backward pass (either may be NULL at inference).
 */
int ffm_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta,
                      float* mean, float* rstd, int rows, int width, int dtype, void* stream);

/*
 * dx = LayerNorm backward w.r.t. the input only (gamma/beta are frozen),
 * plus an optional residual gradient: out = (res ? res : 0) + dx.
 */
int ffm_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                      const float* rstd, const void* res, void* out, int rows, int width,
                      int dtype, void* stream);

/*
 * Patch gather with the input normalisation fused in:
 *   cols[b*gh*gw + p][c*ps*ps + ky*ps + kx] = ((img/255 - mean[c]) / std[c])
 * (trainers/GLP_OT_SVLoRA.py:680,692-693 + the im2col view of conv1,
 * clip/model.py:431).  img is fp32 [B, 3, H, W] raw 0..255.  If
 * prenormalised != 0 the image is taken as already normalised (3D path).
 */
int ffm_patchify(const float* img, void* cols, int B, int H, int W, int patch,
                 const float* mean3, const float* std3, int prenormalised, int dtype, void* stream);
/* (mean3 / std3 are HOST pointers to 3 floats each; they are passed to the kernel by value) */

/*
 * Token assembly + ln_pre (clip/model.py:432-436): row (b,0) = cls + pos[0],
 * row (b,1+p) = patch[b*P+p] + pos[1+p]; x = LayerNorm(row).
 * cls [width] / pos [L, width] are dtype; gamma/beta fp32.
 */
int ffm_embed_lnpre(const void* patch, const void* cls, const void* pos, const float* gamma,
                    const float* beta, void* x, float* rowstat, int B, int L, int width, int dtype, void* stream);
/* rowstat: optional [B*L][2] = {sum, sum of squares} of every output row as stored (the ln_part of a following
 * FFM_EPI_LNIN product, ln_np = 1) */

/*
 * 3D OCT input path (trainers/GLP_OT_SVLoRA.py:585-595, 681-693; BASELINE.json configs[3]).
 * img: fp32 [N = B*S, D, H, W] raw 0..255 (the reference's reshape(-1, D, h, w) view of [B, S*D, H, W]).
 *   ffm_slice_conv_fwd   conv = conv5x5(img/255; w [3,D,5,5], bias [3], pad 2) fp32 [N,3,H,W]; per-image
 *                        mnmx [N,2] = {min, max} over (3,H,W); cnt [N,2] zeroed (tie counters);
 *                        mm_part: scratch [N * ffm_slice_blocks(H,W) * 2]
 *   ffm_patchify_minmax  cols = patches of ((conv - min)/(max - min + 1e-5) - mean)/std; counts the tied extrema
 *   ffm_embed_lnpre_bwd  backward of ffm_embed_lnpre w.r.t. the patch rows (class token / pos are frozen)
 *   ffm_slice_bwd        dcols [N*P, 3*ps*ps] -> gradient partials of the conv weight and bias:
 *                        wpart [N * ffm_slice_wgrad_blocks(H,W)][3*D*25 + 3] (reduce with ffm_reduce_partials);
 *                        scratch: dconv [N,3,H,W], ab_part [N * ffm_slice_bwd_ab_blocks() * 2], gmm [N,2]
 * mean3 / std3 are HOST pointers.
 */
int ffm_slice_blocks(int H, int W);         /* {min, max} partial pairs per image (forward) */
int ffm_slice_wgrad_blocks(int H, int W);   /* weight-gradient partial rows per image (ffm_slice_bwd) */
int ffm_slice_bwd_ab_blocks(void);
int ffm_slice_conv_fwd(const float* img, const float* w, const float* bias, float* conv, float* mm_part,
                       float* mnmx, int32_t* cnt, int N, int D, int H, int W, void* stream);
int ffm_patchify_minmax(const float* conv, const float* mnmx, int32_t* cnt, void* cols, int N, int H, int W,
                        int patch, const float* mean3, const float* std3, int dtype, void* stream);
int ffm_embed_lnpre_bwd(const void* dx, const void* patch, const void* pos, const float* gamma, void* dpatch,
                        int B, int L, int width, int dtype, void* stream);
int ffm_slice_bwd(const void* dcols, const float* img, const float* conv, const float* mnmx, const int32_t* cnt,
                  float* dconv, float* ab_part, float* gmm, float* wpart, int N, int D, int H, int W, int patch,
                  const float* std3, int dtype, void* stream);

/*
 * RN50 trunk (BASELINE.json configs[4]; clip/model.py:11-118, 227-301) on NHWC rows: row = (b, y, x), C contiguous.
 * 1x1 convolutions (Bottleneck.conv1 / conv3 with FairLoRA, downsample.0) are ffm_gemm_nt on these rows.
 *   ffm_stem_im2col      raw fp32 NCHW image -> normalised 3x3 / pad 1 patches of conv1 (stride 2), k = (ky*3+kx)*3 + c,
 *                        zero padded to Kp columns (trainers/GLP_OT_SVLoRA.py:680,692-693 + clip/model.py:240)
 *   ffm_im2col3x3        x [B*H*W, C] -> cols [B*Ho*Wo, Kp], k = (ky*3+kx)*C + c  (weights as [Cout, 3, 3, Cin])
 *   ffm_col2im3x3        dcols -> dx, gathering the <= 9 taps of every input pixel (autograd of F.conv2d w.r.t. input)
 *   ffm_bn_fwd           nn.BatchNorm2d, eps 1e-5, momentum 0.1: training != 0 uses batch statistics (biased variance)
 *                        and updates run_mean / run_var (unbiased) in place; y = bn(x) (+ res) (ReLU if relu != 0);
 *                        mean / rstd [C] are saved for the backward; part: scratch [ffm_bn_blocks(rows)][2][C]
 *   ffm_bn_bwd           g = dy * (relu_out > 0 if relu_out); dgamma, dbeta, dx (train-mode formula); k12: scratch [2][C]
 *   ffm_avgpool2         nn.AvgPool2d(2) forward (in [B,H,W,C] -> out [B,H/2,W/2,C]) / backward (in = d pooled, out = d x)
 *   ffm_add, ffm_relu_bwd   a + b;  g * (y > 0)
 *   ffm_attnpool_tokens  AttentionPool2d's token assembly (mean token + positional embedding) and its backward
 */
/*
 * F.conv2d(x, w, padding=1) (3x3, stride 1, no bias) on NHWC rows as an IMPLICIT GEMM: y [B*H*W, N] = patches(x) . w^T
 * with the patches never written - the GEMM's A-tile loader fetches chunk (tap, c) of a pixel's patch straight from
 * the neighbouring pixel's channels (zeros outside the image and in the K padding come from `zeros`, >= 16 zero bytes
 * of device memory).  w: [N, Kp] with k = (ky*3 + kx)*C + c, zero padded to Kp (Kp * sizeof(dtype) % 128 == 0).
 * splitk_scratch (optional, scratch_elems floats): with few output tiles and a long K (14 x 14 / 7 x 7 maps) the K
 * tiles are split over up to 8 grid slices writing fp32 partial tiles [S][M][N] there, summed by a second small kernel.
 * The input gradient is the same call on dY with the weight re-ordered once to
 * w'[ci][(ky', kx', co)] = w[co][(2 - ky', 2 - kx', ci)]  (clip/model.py:20-24, 241-244 and their autograd).
 */
int ffm_conv3x3_nhwc(const void* x, const void* w, void* y, int B, int H, int W, int C, int N, int Kp,
                     const void* zeros, float* splitk_scratch, int64_t scratch_elems, float* colstat_part, int dtype,
                     void* stream);
/* colstat_part (optional): as ffm_gemm_args.colstat_part.  ffm_conv3x3_colstat_rows returns the number of partial rows
 * [rows][2][N] the launch will leave there, which is what the caller sizes the buffer by and hands to ffm_bn_fwd /
 * ffm_bn_bwd as part_rows:
 *   - launch not split over K:  ceil(M / 128), one per row tile (M = B*H*W);
 *   - launch split over K (few tiles, long K: the 14 x 14 / 7 x 7 maps): the split-K reduction kernel writes them, one
 *     per 8 rows when M < 4096 and one per 32 rows from there on: ceil(M / 8) or ceil(M / 32) - NOT monotone in M, so
 *     size for the worst batch, not the largest;
 *   - 0 when that count would exceed 4096 or N % 4 != 0: no statistics, the BatchNorm makes its own pass. */
int ffm_conv3x3_colstat_rows(int B, int H, int W, int C, int N, int Kp, int64_t scratch_elems, int dtype);
/* the same product with FFM_EPI_BNBWD's column sums (ABI 11): y is dL/dy of a train-mode BatchNorm (+ ReLU with output
 * bn_mask, or NULL) on bn_x, and colstat_part receives {sum g, sum g xhat} per row tile (ffm_bn_bwd's part / part_rows =
 * ffm_conv3x3_colstat_rows, both the one-launch and the split-K geometry above).  FFM_EUNSUP only when
 * ffm_conv3x3_colstat_rows == 0 for this shape. */
int ffm_conv3x3_nhwc_bnbwd(const void* x, const void* w, void* y, int B, int H, int W, int C, int N, int Kp,
                           const void* zeros, float* splitk_scratch, int64_t scratch_elems, float* colstat_part,
                           const void* bn_x, const void* bn_mask, const float* bn_mean, const float* bn_rstd, int dtype,
                           void* stream);
int ffm_stem_im2col(const float* img, void* cols, int B, int H, int W, int stride, int Kp, const float* mean3,
                    const float* std3, int dtype, void* stream);
int ffm_im2col3x3(const void* x, void* cols, int B, int H, int W, int C, int stride, int Kp, int dtype, void* stream);
int ffm_col2im3x3(const void* dcols, void* dx, int B, int H, int W, int C, int stride, int Kp, int dtype, void* stream);
int ffm_bn_blocks(int rows);
int ffm_bn_fwd(const void* x, const float* gamma, const float* beta, float* run_mean, float* run_var, float* mean,
               float* rstd, float* part, int part_rows, const void* res, void* y, int rows, int C, int training, int relu,
               int dtype, void* stream);
/* part_rows > 0 (training, at most 4096): part already holds that many partial rows [part_rows][2][C] of column sums written by the
 * producer of x (colstat_part): the column-sum pass over x is skipped */
int ffm_bn_bwd(const void* dy, const void* relu_out, const void* x, const float* gamma, const float* mean,
               const float* rstd, float* part, int part_rows, float* k12, float* dgamma, float* dbeta, void* dx, void* g_out,
               int rows, int C, int dtype, void* stream);
/* g_out (optional, [rows, C] dtype): also receives g = dy * (relu_out > 0) - the gradient that an identity-skip
 * Bottleneck passes on beside bn3 (clip/model.py:57-59), which is otherwise one more pass (ffm_relu_bwd).
 * part_rows > 0 (ABI 11, at most 4096; g_out must be NULL): part already holds that many partial rows [part_rows][2][C] of
 * {sum g, sum g xhat} written by the producer of dy (ffm_gemm_args FFM_EPI_BNBWD): the column-sum pass is skipped */
int ffm_avgpool2(const void* in, void* out, int B, int H, int W, int C, int backward, int dtype, void* stream);
int ffm_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);
int ffm_relu_bwd(const void* g, const void* y, void* out, int64_t n, int dtype, void* stream);
int ffm_attnpool_tokens(const void* in, const void* pos, void* out, int B, int HW, int E, int backward, int dtype,
                        void* stream);

/*
 * Multi-head self-attention core, softmax(Q K^T / sqrt(64)) V, head_dim 64,
 * optional causal mask (text tower, clip/model.py:562-568); no dropout.
 * qkv: [B*L, 3*heads*64] rows (b,l) with q|k|v concatenated as nn.MultiheadAttention's
 * packed in-projection produces them (clip/model.py:352).  out: [B*L, heads*64].
 * lse: [B, heads, L] fp32 log-sum-exp of the scaled scores (saved for backward).
 */
int ffm_attention_fwd(const void* qkv, void* out, float* lse, int B, int L, int heads,
                      int causal, int dtype, void* stream);

/*
 * Backward of the above: given dout, recomputes P from qkv and lse and writes
 * dqkv [B*L, 3*heads*64].  delta: [B, heads, L] fp32, unused (the row sums
 * of dO * O are formed inside the kernels since ABI 3); kept so that callers need not change.
 */
int ffm_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                      float* delta, void* dqkv, int B, int L, int heads, int causal,
                      int dtype, void* stream);
/*
 * ABI 12: the same, ALSO leaving the two row sums of the LayerNorm backward that follows the in-projection's dX product
 * (ln_1 of a ResidualAttentionBlock, clip/model.py:354-357; FFM_EPI_LNB_* above, no rank-r terms: the in-projection is
 * frozen): ln_part [2 heads][B L][2] fp32 <- {sum_n dqkv[n] ln_wg[n], sum_n dqkv[n] (qkv[n] - ln_d[n])} over 64-column
 * slices - slot h: the q columns of head h, slot heads + h: its k and v columns; ln_wg = W gamma, ln_d = W beta + b of the
 * LayerNorm-folded in-projection [3 heads 64].  The dX product of qkv then takes them as FFM_EPI_LNB_APPLY's lnb_part with
 * lnb_np = 2 heads (<= 24).  16-bit storage, no mask, 97..256 tokens only: ffm_attention_bwd_lnstat_ok says whether a
 * shape is served (1) or ffm_attention_bwd_lnstat returns FFM_EUNSUP (0).
 */
int ffm_attention_bwd_lnstat_ok(int L, int causal, int dtype);
int ffm_attention_bwd_lnstat(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                             const float* ln_wg, const float* ln_d, float* ln_part, int B, int L, int heads, int causal,
                             int dtype, void* stream);

/*
 * Rank-r down projection with the group-mixed singular values applied
 * (trainers/GLP_OT_SVLoRA.py:453-477):
 *   t[m][j]  = sum_k x[m][k] * P[k][j]        (P = lora_A [K,r], layout_rk = 0)
 *            = sum_k x[m][k] * P[j][k]        (P = lora_B [r,K], layout_rk = 1: u = g B^T)
 *   ts[m][j] = scaling * t[m][j] * s_b[j],  s_b = pi_b . S,  b = m / rows_per_sample
 * attr: int32 [nsamples] group index, or NULL for the uniform 1/G mix (:462).
 * If t_fwd != NULL also accumulates the dS partials
 *   ds_part[blk][g][j] = sum_{m in blk} pi_{b(m)}[g] * scaling * t_fwd[m][j] * t[m][j]
 * The call writes exactly ffm_lora_down_blocks(M, K, r, dtype) partial rows of G * r floats (the count a reduction over
 * ds_part must use); ffm_lora_down_blocks_max(M_max, ...) bounds that count over every M <= M_max and is for sizing a
 * buffer once from the largest batch - never for summing.
 */
int ffm_lora_down(const void* x, int ldx, const float* P, int layout_rk, const float* S,
                  const int32_t* attr, int M, int K, int r, int G, int rows_per_sample,
                  float scaling, float lambda_group, float* t, float* ts,
                  const float* t_fwd, float* ds_part, int dtype, void* stream);
int ffm_lora_down_blocks(int M, int K, int r, int dtype);       /* exact: dS partial rows the call above writes */
int ffm_lora_down_blocks_max(int M_max, int K, int r, int dtype); /* upper bound over M <= M_max (buffer sizing) */

/*
 * Rank-r gradient reduction over the token rows (the dA / dB sums of
 * FairLoRALinear's backward) without forming dW:
 *   part[s][k][j] = sum_{m in split s} x[m][k] * v[m][j]
 * v = ts-style rows (already scaled by scaling * s_b).  part has
 * ffm_lora_grad_splits(M) * K * r floats.
 */
int ffm_lora_grad_partial(const void* x, int ldx, const float* v, int M, int K, int r,
                          float* part, int dtype, void* stream);
/* The same for x = LayerNorm(x_raw) that was never materialised (ln_2 folded into c_fc): x is the RAW rows,
 * mean / rstd [M] and gamma / beta [K] the LayerNorm, and with y = (x_raw - mean) rstd gamma + beta
 *   part[split][k][j] = gamma[k] (sum_m x_raw[m][k] rstd[m] v[m][j] - sum_m mean[m] rstd[m] v[m][j]) + beta[k] sum_m v[m][j].
 * bf16, K % 128 == 0, r <= 16. */
int ffm_lora_grad_partial_ln(const void* x, int ldx, const float* v, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, int M, int K, int r, float* part, int dtype,
                             void* stream);
int ffm_lora_grad_splits(int M);

/*
 * out[i] (+)= sum_s part[s][i] for i < n, optionally transposing a [K, r]
 * partial into an [r, K] destination (lora_B.grad).  accumulate != 0 adds to out.
 */
int ffm_reduce_partials(const float* part, int nsplit, int n, float* out, int transpose_K,
                        int transpose_r, int accumulate, void* stream);

/*
 * The same reduction for many tensors in ONE launch (all LoRA gradients of a
 * step): descs_dev is a DEVICE array of ndesc descriptors, max_n = max over the
 * descriptors of n * (nsplit > 64 ? 8 : 1) (the launch width: 4 lanes per output,
 * 32 for tensors with more than 64 partial rows).
 */
typedef struct ffm_reduce_desc {
    const float* part;
    float* out;
    int32_t nsplit, n, transpose_K, transpose_r;
} ffm_reduce_desc;
int ffm_reduce_partials_multi(const ffm_reduce_desc* descs_dev, int ndesc, int max_n, void* stream);

/*
 * Logits head with OT='None' (trainers/GLP_OT_SVLoRA.py:713-757):
 *   fbar[b] = mean_{l>=1} f[b,l] / max(|f[b,l]|, 1e-12)
 *   logits_img[b][c] = exp(logit_scale) * <fbar[b], tbar[c]>
 * f: [B*L, D] dtype; tbar: [n_cls, D] fp32 = mean_n normalize(text[n,c]).
 * rnorm: [B*L] fp32 (saved), fbar: [B, D] fp32 (saved).
 */
int ffm_head_fwd(const void* f, const float* tbar, const float* logit_scale, float* fbar,
                 float* rnorm, float* logits_img, int B, int L, int D, int n_cls, int dtype,
                 void* stream);

/*
 * Slice-mean + softmax cross-entropy (trainers/GLP_OT_SVLoRA.py:753-754,908):
 *   logits[b][c] = mean_s logits_img[b*S+s][c];  loss = mean_b CE(logits[b], label[b])
 * Writes logits [nb, n_cls], prob [nb, n_cls], loss [1], dlogits_img [nb*S, n_cls]
 * (= dloss/dlogits_img), and sets *finite_flag = 0 if the loss is not finite
 * (Dassl/dassl/engine/trainer.py:260-262 raises on the host from it).
 */
int ffm_ce_loss(const float* logits_img, const int64_t* label, float* logits, float* prob,
                float* loss, float* dlogits_img, int32_t* finite_flag, int nb, int S, int n_cls,
                void* stream);

/*
 * Head backward: df [B*L, D] dtype (row l = 0 gets zeros) and
 * dtbar [n_cls, D] fp32.
 */
int ffm_head_bwd(const void* f, const float* tbar, const float* logit_scale, const float* fbar,
                 const float* rnorm, const float* dlogits_img, void* df, float* dtbar, int B,
                 int L, int D, int n_cls, int dtype, void* stream);

/*
 * The two ends of the text tower (float32; n_text = n_prompts * n_cls rows, text row p = n * n_cls + c).
 *
 * ffm_text_embed: PromptLearner.forward with the class token at the end + the positional embedding
 * (trainers/GLP_OT_SVLoRA.py:131-152, :57): x [n_text * TL, w], row (p, t) = {prefix[p] | ctx[p / n_cls][t - 1] |
 * suffix[p][t - 1 - n_ctx]} + pos[t]; prefix [n_text, 1, w], ctx [n_prompts, n_ctx, w], suffix [n_text, suffix_rows, w]
 * (the first TL - 1 - n_ctx of its rows are read), pos [>= TL, w].
 *
 * ffm_text_tail_fwd: x[eot_row[p]] (eot_row: absolute row of each prompt's EOT token, tokenized_prompts.argmax, :64)
 * -> ln_final (eps 1e-5) -> @ text_projection [w, D] (:62-64) -> F.normalize (:716): tf [n_text, D] (un-normalised),
 * tn [n_text, D], rnorm [n_text] = 1 / max(|tf|, 1e-12), stats [n_text, 2] = {mean, rstd} of ln_final, and - when tbar is
 * not NULL - tbar [n_cls, D] = mean over the prompts of tn (the OT = 'None' head's text operand, :713-717).
 *
 * ffm_text_tail_bwd: exactly one of dtbar [n_cls, D] (gradient of that mean) and dtn [n_text, D] (transport heads) ->
 * g [n_text * TL, w]: the gradient w.r.t. the tower's output, zero on every row that is not an EOT row; dy [n_text, w]
 * is scratch.  ffm_text_ctx_grad: dctx [n_prompts, n_ctx, w] = the tower's INPUT gradient g at the ctx rows, summed
 * over the classes that share a prompt's context.
 */
int ffm_text_embed(const float* prefix, const float* ctx, const float* suffix, int suffix_rows, const float* pos, float* x,
                   int n_prompts, int n_cls, int n_ctx, int TL, int w, void* stream);
int ffm_text_tail_fwd(const float* x, const int32_t* eot_row, const float* lnw, const float* lnb, const float* proj,
                      float* tf, float* tn, float* rnorm, float* stats, float* tbar, int n_prompts, int n_cls, int w,
                      int D, void* stream);
int ffm_text_tail_bwd(const float* x, const int32_t* eot_row, const float* lnw, const float* proj, const float* tn,
                      const float* rnorm, const float* stats, const float* dtbar, const float* dtn, float* dy, float* g,
                      int n_prompts, int n_cls, int TL, int w, int D, void* stream);
int ffm_text_ctx_grad(const float* g, float* dctx, int n_prompts, int n_cls, int n_ctx, int TL, int w, void* stream);

/*
 * uint8 input transport: dst fp32 [B, C1*rep, HW] = (float) src u8 [B, C1, HW] with every source channel repeated
 * `rep` times in place (np.repeat(x, rep, axis=0): utils/data_utils.py:667-679, 771-778).  HW % 4 == 0.  The
 * result is what the reference's loader would have shipped as float32, bit for bit.
 */
int ffm_expand_u8(const uint8_t* src, float* dst, int B, int C1, int HW, int rep, void* stream);

/*
 * Evaluator counts for binary tasks (evaluation/evaluator_oph.py:37-150; evaluation/metrics.py:197-311, 513-552;
 * Dassl/dassl/engine/trainer.py:523-569).  prob: fp32 [N, 2] softmax scores, label: int64 [N] in {0, 1},
 * attr: int64 [N] group ids (values outside [0, G) are "unknown"), or NULL.
 * out: uint64 [(G + 2)][FFM_EVAL_SLOTS], zeroed by the call; row g < G = group g, row G = unknown, row G + 1 = all:
 *   0 n_pos   1 n_neg   2 win1  3 tie1   (label-1 sample i vs label-0 sample j: p1_i > p1_j, p1_i == p1_j)
 *   4 win0  5 tie0   (class-0 column: p0_j > p0_i, p0_j == p0_i)   6 TP  7 FP  8 TN  9 FN  (pred = p1 > p0)
 * one-vs-rest AUC of class c = (win_c + tie_c / 2) / (n_pos * n_neg)  (== sklearn.metrics.roc_auc_score), pairs of a
 * group row are counted among that group's samples only.  All counts are exact integers.
 */
#define FFM_EVAL_SLOTS 10
int ffm_eval_counts(const float* prob, const int64_t* label, const int64_t* attr, int N, int G, uint64_t* out,
                    void* stream);
/*
 * The same table for large test sets in O(N log N) (sort by score, prefix counts of the negatives; csrc/evalsort.hip):
 * bit-identical to ffm_eval_counts, which compares all pairs (N^2 / 2: 100 x slower at 200 k samples).  workspace: a
 * 256-byte aligned device buffer of at least ffm_eval_counts_ws_bytes(N) bytes (the library allocates nothing).
 */
int64_t ffm_eval_counts_ws_bytes(int N);
int ffm_eval_counts_sorted(const float* prob, const int64_t* label, const int64_t* attr, int N, int G, uint64_t* out,
                           void* workspace, int64_t workspace_bytes, void* stream);

/*
 * Logits heads with a transport plan, TRAINER.GLP_OT.OT = 'Sinkhorn' (mode 1) / 'COT' (mode 2)
 * (trainers/GLP_OT_SVLoRA.py:615-675, 713-757).  f: [B*L, D] dtype (token 0 of every image is dropped), tn: fp32
 * [N*n_cls, D] L2-normalised text features, row n*n_cls + c.  Per problem p = b*n_cls + c over M = L-1 tokens:
 *   sim [P][M][N] cosine similarities, K = exp(-(1 - sim)/eps), T [P][M][N] = plan after the iteration at which the
 *   batch-mean change of the iterate first drops below thresh (<= max_iter; ONE stopping index for the whole batch,
 *   as in the reference), logits_img[b][c] = exp(logit_scale) * sum T * sim.
 * Saved for the backward: rnorm [B*L], T.  Scratch: sim, errs [max_iter][P], istop [1], tsum [P].
 * Backward (T is a constant, the reference builds it under no_grad): df [B*L, D] dtype and the per-image partials
 * dtn_part [B][N*n_cls][D] of d tn (reduce over B with ffm_reduce_partials).  L-1 <= 256, N <= 8, n_cls <= 8, D <= 1024.
 */
int ffm_ot_head_fwd(const void* f, const float* tn, const float* logit_scale, float* rnorm, float* sim, float* T,
                    float* errs, int32_t* istop, float* tsum, float* logits_img, int B, int L, int D, int n_cls,
                    int N, int mode, float eps, float thresh, int max_iter, float top_percent, int dtype, void* stream);
int ffm_ot_head_bwd(const void* f, const float* tn, const float* logit_scale, const float* rnorm, const float* T,
                    const float* dlogits_img, void* df, float* dtn_part, int B, int L, int D, int n_cls, int N,
                    int dtype, void* stream);

/*
 * torch.optim.SGD(momentum, weight_decay, dampening=0) over one flat fp32
 * buffer (Dassl/dassl/optim/optimizer.py:105-113): d = g + wd*p;
 * buf = first_step ? d : mu*buf + d; p -= lr*buf.
 */
int ffm_sgd_momentum(float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                     float weight_decay, int first_step, void* stream);

/*
 * `repeats` (1..16) consecutive applications of that update on the SAME gradient, in one pass over memory.  The
 * reference registers 'prompt_learner' and 'image_encoder' with one shared optimizer
 * (trainers/GLP_OT_SVLoRA.py:866-870) and TrainerBase.model_update steps the optimizer of every registered name
 * (Dassl/dassl/engine/trainer.py:333-337): with UNFREEZE_IMAGE_ENCODER, optim.step() runs twice per batch.
 */
int ffm_sgd_momentum_n(float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                       float weight_decay, int first_step, int repeats, void* stream);

/*
 * p[i] *= scale in place; clears *finite_flag (may be NULL) when a product is not finite.  The IEEE-half mode (FFM_F16)
 * scales dloss/dlogits by 2^k before the backward pass so that the 16-bit activation gradients stay clear of half's
 * subnormals (the reference's fp16 mode lets them underflow: no GradScaler outside PREC='amp',
 * trainers/GLP_OT_SVLoRA.py:889-907) and takes the factor out of the fp32 gradient buffer here, in front of the SGD
 * step; the flag is the overflow guard (Dassl/dassl/engine/trainer.py:260-262 raises on the host from it).
 */
int ffm_scale_check(float* p, float scale, int64_t n, int32_t* finite_flag, void* stream);

/*
 * IEEE-half mode (PREC="fp16"): a DYNAMIC gradient scale that lives in device memory, so that recorded launch plans and
 * captured graphs stay valid while it changes.  state = 8 floats {scale, 1/scale, ok, good_run, overflows, max_scale,
 * growth_interval, min_scale}.  One training step is
 *   ffm_loss_scale(dlogits)      dlogits *= scale; ok = 1
 *   ... backward ...
 *   ffm_unscale_check(grad)      grad *= 1/scale; any non-finite entry clears ok
 *   ffm_sgd_momentum_gated(...)  the ffm_sgd_momentum_n update, skipped entirely when ok == 0; then the scale moves:
 *                                overflow -> scale = max(scale / 2, min_scale), overflows += 1, good_run = 0;
 *                                otherwise good_run += 1 and after growth_interval (> 0) good steps scale = min(2 scale, max_scale).
 * The reference's fp16 run has no scaler (model_backward_and_update, Dassl/dassl/engine/trainer.py:323-342): an overflowed
 * backward pass poisons its weights and the next loss raises; here the step is skipped and training goes on, as
 * torch.cuda.amp.GradScaler does for the reference's PREC="amp" (trainers/GLP_OT_SVLoRA.py:889-898).
 */
int ffm_loss_scale(float* p, int64_t n, float* state, void* stream);
int ffm_unscale_check(float* g, int64_t n, float* state, void* stream);
int ffm_sgd_momentum_gated(float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                           float weight_decay, int first_step, int repeats, float* state, void* stream);

/* The same update with {lr, momentum, weight_decay} read from DEVICE memory (hp[3]) and a momentum buffer
 * that starts at zero: safe to capture in a hipGraph while the LR schedule changes lr between replays. */
int ffm_sgd_momentum_dev(float* p, const float* g, float* buf, int64_t n, const float* hp, void* stream);

/*
 * Round boundary helpers (utils/fed_utils.py:42-100) on the flat trainable
 * buffer: out[i] = p[i] * w[i] (per-element FedAvg weights: n_k/N, or
 * n_{k,g}/N_g on lora_S rows) ...
 */
int ffm_scale_by(const float* p, const float* w, float* out, int64_t n, void* stream);
/* ... acc[i] += p[i] * w[i] for the second and later clients a rank holds in one round (the `+=` of
 * utils/fed_utils.py:79-86: product and sum rounded separately, never fused) ... */
int ffm_scale_acc(const float* p, const float* w, float* acc, int64_t n, void* stream);
/* ... and, after the all-reduce: shared_half_s column means on each lora_S
 * block listed in s_offsets (offset of a [G, r] block in the flat buffer),
 * then EMA with the previous global: p = (1-beta)*avg + beta*prev. */
int ffm_fedavg_finish(float* avg, const float* prev, float* out, int64_t n, const int64_t* s_offsets,
                      int n_s, int G, int r, int shared_half_s, float beta, void* stream);

/* dtype conversion helpers (weight preparation at load time). */
int ffm_cast_f32_to(const float* src, void* dst, int64_t n, int dtype, void* stream);
int ffm_cast_to_f32(const void* src, float* dst, int64_t n, int dtype, void* stream);
/* dst[c][r] = src[r][c] as dtype (W^T copies of frozen weights for the dX products). */
int ffm_transpose_cast(const float* src, void* dst, int rows, int cols, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FFM_HIP_H */
