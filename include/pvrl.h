/*
 * pvrl.h -- C ABI of libpvrl_hip.so, the MI355X (gfx950) implementation of the
 * ProcedureVRL video-narration pre-training hot path.
 *
 * The reference (facebookresearch/ProcedureVRL) has no FFI: its hot path is eager PyTorch
 * below `MODEL_REGISTRY.get(name)(cfg)` (lib/models/build.py:36).  Each entry point here
 * replaces the ATen op sequence of the cited reference lines.  The Python modules in
 * procedurevrl_amd/ (same class names / state_dict keys as lib/models/vit.py) bind these
 * through ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers to DEVICE memory + sizes; no torch types.  `stream` is a hipStream_t.
 *   - the caller owns every buffer, including workspaces; nothing is allocated, nothing
 *     synchronises, no global state; every call is asynchronous on `stream`.
 *   - return 0 on success, PVRL_EINVAL (-1) on a bad argument, <= -2 on a HIP launch error.
 *   - "bf16" in entry-point names, PVRL_EPI_* names and comments means THE LIBRARY'S 16-BIT OPERAND TYPE: fp16 in the default
 *     library (libpvrl_hip_f16.so, what procedurevrl_amd loads unless PVRL_OPERAND=bf16 -- the flavour that meets the 1e-3 parity
 *     bar), bf16 in libpvrl_hip.so.  Same layouts, same MFMA rate; pvrl_operand_dtype() tells which.  "ld*" are leading dimensions
 *     in ELEMENTS.
 *   - token layout of the TimeSformer encoder (B*N*T + B rows, C = 768):
 *       rows [0, B*N*T)      patch tokens ordered (b, n, t), t innermost
 *       rows [B*N*T, +B)     the cls token of each clip
 *     (the reference keeps [B, 1 + N*T, C] with token index 1 + n*T + t, vit.py:396-407).  Since round 6 the residual stream and its
 *     gradient are SPLIT: the patch rows live in a 16-bit matrix, the cls rows in an fp32 one (`pvrl_rows`, PVRL_EPI_RESID_16);
 *     every other activation between kernels is 16-bit, statistics / head / losses / parameter gradients are fp32.
 */
#ifndef PVRL_H_
#define PVRL_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* GEMM epilogues (pvrl_gemm_nt_bf16) */
#define PVRL_EPI_BF16 0      /* out0 bf16 = rowscale * (acc + bias)                                   */
#define PVRL_EPI_GELU 1      /* u = acc + bias; out0 bf16 = u; out1 bf16 = GELU_erf(u)  (vit.py:54-60) */
#define PVRL_EPI_QGELU 2     /* same with QuickGELU x*sigmoid(1.702x)          (tfm_model.py:27-29)   */
#define PVRL_EPI_RESID_F32 3 /* out0 f32 = aux_f32[m % rowmod] + rowscale * (acc + bias) + bias2      */
#define PVRL_EPI_F32 4       /* out0 f32 = rowscale * (acc + bias)                                    */
#define PVRL_EPI_DGELU 5     /* out0 bf16 = rowscale * acc * GELU_erf'(aux_bf16)   (MLP backward)     */
#define PVRL_EPI_DQGELU 6    /* out0 bf16 = rowscale * acc * QuickGELU'(aux_bf16)                     */
#define PVRL_EPI_RESID_16 7  /* out0 bf16 = aux + rowscale * (acc + bias) + bias2: the residual add on the 16-bit patch rows of the
                              * split residual stream (pvrl_rows); aux bf16 [M, aux_ld], or -- aux_rowmod != 0 -- the fp32 table
                              * aux_f32[m % rowmod] of the embedding prologue (vit.py:370-407)            */

/* The 16-bit operand type this library was built with: 0 = bf16 (libpvrl_hip.so), 1 = fp16 (libpvrl_hip_f16.so, built
 * with -DPVRL_OPERAND_F16).  Wherever an entry point below says "bf16" (names, PVRL_EPI_BF16, comments) read "the
 * library's 16-bit operand type": same layouts, same MFMA rate; with fp16 the caller scales the gradients it feeds the backward entry points
 * (5 exponent bits; procedurevrl_amd/engine.py GradStore.begin_scaled does this per engine). */
int pvrl_operand_dtype(void);

/* C[M,N] = epilogue(A[M,K] . W[N,K]^T), 16-bit operands (see above), fp32 accumulate on MFMA.
 * Replaces nn.Linear forward (vit.py:54-60,75-92,133; tfm_model.py:35-41) and, with the
 * transposed 16-bit weight copy as W, its data gradient.  N % 128 == 0, K % 64 == 0, any M.  bias2 (fp32 [N] or null,
 * PVRL_EPI_RESID_F32 / PVRL_EPI_RESID_16 only) is added after the row scale: x + rs * (o W_e^T + b_e) + b_fc of the fused temporal branch.
 * M <= 192 with K % 256 == 0 (the order / diffusion stack's 36- and 144-row products, tfm_model.py:129-204) runs a few-row kernel whose
 * workgroups split K over their waves (csrc/gemm_nt_skinny.h): same arithmetic up to the order of the fp32 sums over k. */
int pvrl_gemm_nt_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N, int64_t K,
                      int epilogue, const float* bias, const float* rowscale, const void* aux, int64_t aux_ld,
                      int64_t aux_rowmod, void* out0, int64_t ld0, void* out1, int64_t ld1, const float* bias2,
                      void* stream);

/* Several SMALL problems of pvrl_gemm_nt_bf16 in one launch (128 x 128 tiles; epilogue PVRL_EPI_BF16, PVRL_EPI_F32 or PVRL_EPI_RESID_F32,
 * the same for all): the 768^3 products of the temporal branch's two back-to-back linear maps (vit.py:131-134) -- W_fc W_proj of every
 * block, and in backward dW_e W_proj^T / W_fc^T dW_e -- are 36 tiles each on a 256-CU chip; a dozen per launch fill it.  Per problem:
 * out0 = [aux +] rowscale * (A W^T + bias), N % 128 == 0, K % 64 == 0, leading dimensions multiples of 8.  Any nprob (12 per launch). */
typedef struct pvrl_nt_problem {
  const void* A; int64_t lda;     /* bf16 [M, K] */
  const void* W; int64_t ldw;     /* bf16 [N, K] */
  int64_t M, N, K;
  const float* bias;              /* fp32 [N] or null */
  const float* rowscale;          /* fp32 [M] or null */
  const void* aux; int64_t aux_ld; /* fp32 [M, aux_ld] (PVRL_EPI_RESID_F32) or null */
  void* out0; int64_t ld0;
} pvrl_nt_problem;
int pvrl_gemm_nt_batched_bf16(int nprob, const pvrl_nt_problem* problems, int epilogue, void* stream);

/* C[M,N] = alpha * (A[M,K] . B[N,K]^T) + bias, all fp32 (projection head vit.py:299, step logits
 * `x @ label_emb.t() / temp` vit.py:307,334,340,432).  Long reductions with few output tiles are split over K into fp32
 * partials (workspace >= pvrl_gemm_nt_f32_small_workspace_bytes, may be 0 / null when that returns 0) summed in a fixed
 * order: deterministic. */
int64_t pvrl_gemm_nt_f32_small_workspace_bytes(int64_t M, int64_t N, int64_t K);
int pvrl_gemm_nt_f32_small(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float alpha,
                           float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* Weight gradient dW[N,K] = beta*dW + P[M,N]^T . Q[M,K]; dbias[N] = beta*dbias + colsum(P) (optional).
 * Backward of nn.Linear / the patch-embed conv (loss.backward(), tools/train_net.py:176-181).
 * N % 128 == 0, K % 128 == 0.  M is cut into `splits` slices whose fp32 partial tiles a second kernel sums
 * (deterministic, no atomics): take splits = pvrl_gemm_tn_plan_splits(M, N, K) (the count that fills the 256 CUs
 * for the kernel the shape selects: N * K >= 256 * 256 -> 256x256 register-transposed 8-wave kernel, any splits >= 1, half
 * tiles staged with zero columns; smaller shapes the 128x128 kernel, splits a positive multiple of 8 = one slice per XCD).
 * workspace >= pvrl_gemm_tn_workspace_bytes(N, K, splits). */
int64_t pvrl_gemm_tn_plan_splits(int64_t M, int64_t N, int64_t K);
int64_t pvrl_gemm_tn_workspace_bytes(int64_t N, int64_t K, int64_t splits);
/* Common tail of the entry points that WRITE PARAMETER GRADIENTS (this one, _into, _grouped, pvrl_layernorm_bwd):
 *   gscale    device scalar or null: the new contribution is multiplied by it, dW = beta*dW + gscale * (P^T Q).  The fp16-operand
 *             flavour runs a backward in S-scaled units (5 exponent bits; engine.GradStore.begin_scaled picks the power of two S on
 *             the device) and the kernel that writes a parameter gradient takes 1/S out again -- no separate pass over the buffer;
 *   nonfinite device flag or null: set to 1 when a value written is inf / nan.  It is the `skip` flag of pvrl_adam_step /
 *             pvrl_sgd_step: misc.check_nan_losses (tools/train_net.py:174) raises in front of optimizer.step(); here the bad step is
 *             dropped on the device and reported at the next log point.  Only ever raised; pvrl_flag_roll re-arms it. */
int pvrl_gemm_tn_bf16(const void* P, int64_t ldp, const void* Q, int64_t ldq, int64_t M, int64_t N, int64_t K,
                      int64_t splits, float beta, float* dW, float* dbias, void* workspace, int64_t workspace_bytes,
                      const float* gscale, float* nonfinite, void* stream);
/* The same product reduced straight into an UN-PADDED destination: dW[n][k] (leading dimension ldw) for n < n_valid,
 * k < k_valid only, dbias[n] for n < n_valid (its own beta_bias) -- zero-padded operands (MViT widths 96 / 288 / 441 ...
 * padded to the tile multiples) write their weight gradient directly into parameter.grad. */
int pvrl_gemm_tn_into_bf16(const void* P, int64_t ldp, const void* Q, int64_t ldq, int64_t M, int64_t N, int64_t K,
                           int64_t splits, float beta, float* dW, int64_t ldw, int64_t n_valid, int64_t k_valid,
                           float* dbias, float beta_bias, void* workspace, int64_t workspace_bytes, const float* gscale,
                           float* nonfinite, void* stream);

/* Several weight gradients of the same backward pass in ONE launch (a transformer block's seven nn.Linear dW,
 * loss.backward(), tools/train_net.py:176-181): the (row slice, 256x256 tile) work items of all problems share the 256 CUs,
 * so every problem is cut into the same small number of slices (pvrl_gemm_tn_grouped_plan_splits) instead of the 7-28 a
 * lone dW needs -- longer reduction loops, less fp32 partial traffic.  1 <= nprob <= 8, every N and K a multiple of
 * 128 (half tiles are staged with zero columns), M >= 1 (else PVRL_EINVAL: use pvrl_gemm_tn_bf16 per problem).  Semantics per problem as pvrl_gemm_tn_bf16;
 * deterministic.  workspace >= pvrl_gemm_tn_grouped_workspace_bytes(nprob, problems, splits). */
typedef struct pvrl_tn_problem {
  const void* P; int64_t ldp;     /* bf16 [M, N] */
  const void* Q; int64_t ldq;     /* bf16 [M, K] */
  int64_t M, N, K;
  float beta;
  float* dW;                      /* fp32 [N, K] */
  float* dbias;                   /* fp32 [N] or null */
  const float* gscale;            /* device scalar or null (see pvrl_gemm_tn_bf16) */
  float* nonfinite;               /* device flag or null */
} pvrl_tn_problem;
int64_t pvrl_gemm_tn_grouped_plan_splits(int nprob, const pvrl_tn_problem* problems);
int64_t pvrl_gemm_tn_grouped_workspace_bytes(int nprob, const pvrl_tn_problem* problems, int64_t splits);
int pvrl_gemm_tn_grouped_bf16(int nprob, const pvrl_tn_problem* problems, int64_t splits, void* workspace,
                              int64_t workspace_bytes, void* stream);

/* LayerNorm over fp32 rows, C in {512, 768} (vit.py:104,109,116,228 eps 1e-6; tfm_model.py:18-24 eps 1e-5).
 * fwd: y = (x - mean) * rstd * gamma + beta  -> bf16 (GEMM operand) or fp32.
 * bwd: dx_out = dx_in(optional) + dLN; dgamma/dbeta = beta_acc * old + sums over rows; optionally also writes
 *      dxs_bf16[m] = bf16(dxs_scale[m] * dx_out[m]) for m < dxs_rows (the next stage's bf16 GEMM operand, DropPath-scaled)
 *      and dxsum[c] = beta_acc * old + sum over m < dxs_rows of dx_out[m][c] (the UNscaled column sums = the gradient of a
 *      bias added after the DropPath scale, optional).  gscale / nonfinite: see pvrl_gemm_tn_bf16 (dgamma, dbeta and dxsum are
 *      parameter gradients: beta_acc * old + gscale * sums). */
int pvrl_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* y,
                       int64_t ldy, int out_is_f32, float* mean, float* rstd, int64_t M, int64_t C, void* stream);
int64_t pvrl_layernorm_bwd_workspace_bytes(int64_t M, int64_t C);
int pvrl_layernorm_bwd(const void* dy, int64_t lddy, int dy_is_f32, const float* x, int64_t ldx, const float* mean,
                       const float* rstd, const float* gamma, const float* dx_in, int64_t ldi, float* dx_out,
                       int64_t ldo, float beta_acc, float* dgamma, float* dbeta, void* workspace,
                       int64_t workspace_bytes, int64_t M, int64_t C, void* dxs_bf16, int64_t ldxs, const float* dxs_scale,
                       int64_t dxs_rows, float* dxsum, const float* gscale, float* nonfinite, void* stream);

/* SPLIT residual stream (round 6).  The TimeSformer engine keeps the PATCH rows of the residual stream x (vit.py:129-157: the `x`
 * every Block.forward reads and returns) and of its gradient in the 16-bit operand type and the B cls rows in fp32: a matrix of M rows
 * is then two matrices -- rows [0, rows16) in `lo` (16-bit, row stride ldlo), rows [rows16, M) in `hi` (fp32, row stride ldhi; its row 0
 * is row rows16).  rows16 = 0: the plain fp32 matrix of pvrl_layernorm_fwd / _bwd, which are these entry points with that split.
 * _bwd_split: x, dx_in and dx_out share ONE rows16; a null part of dx_in reads as zeros. */
typedef struct pvrl_rows {
  void* lo; int64_t ldlo;
  void* hi; int64_t ldhi;
  int64_t rows16;
} pvrl_rows;
int pvrl_layernorm_fwd_split(const void* x16, int64_t ldx16, int64_t rows16, const float* x, int64_t ldx, const float* gamma,
                             const float* beta, float eps, void* y, int64_t ldy, int out_is_f32, float* mean, float* rstd,
                             int64_t M, int64_t C, void* stream);
int pvrl_layernorm_bwd_split(const void* dy, int64_t lddy, int dy_is_f32, const pvrl_rows* x, const float* mean, const float* rstd,
                             const float* gamma, const pvrl_rows* dx_in, const pvrl_rows* dx_out, float beta_acc, float* dgamma,
                             float* dbeta, void* workspace, int64_t workspace_bytes, int64_t M, int64_t C, void* dxs_bf16,
                             int64_t ldxs, const float* dxs_scale, int64_t dxs_rows, float* dxsum, const float* gscale,
                             float* nonfinite, void* stream);

/* pvrl_layernorm_bwd with dgamma = dbeta = null leaves its per-workgroup partial sums in `workspace` (the caller keeps that workspace
 * to itself) instead of reducing them; this entry point then reduces the partials of MANY LayerNorms in one launch -- an encoder backward
 * has 37, each otherwise followed by its own 7-us reduce: dgamma = beta * dgamma + gscale * sum, dbeta likewise, dxsum with beta_sum. */
typedef struct pvrl_ln_reduce {
  const float* part;         /* the workspace a deferred pvrl_layernorm_bwd(M, C, dxsum != null ? 1 : 0) wrote */
  int64_t M, C;
  int want_sum;              /* that call had a dxsum target */
  float beta, beta_sum;
  float* dgamma; float* dbeta; float* dxsum;
} pvrl_ln_reduce;
int pvrl_layernorm_bwd_reduce_batched(int n, const pvrl_ln_reduce* items, const float* gscale, float* nonfinite, void* stream);

/* Temporal attention for T = 8 (Block.forward temporal branch, vit.py:129-135 via Attention.forward
 * vit.py:75-92): sequences are 8 consecutive rows of the packed qkv [rows][3*H*64]. */
int pvrl_attn_t8_fwd(const void* qkv, int64_t ld, int64_t nseq, int64_t H, float scale, void* o, int64_t ldo,
                     void* stream);
int pvrl_attn_t8_bwd(const void* qkv, int64_t ld, int64_t nseq, int64_t H, float scale, const void* d_o, int64_t ldo,
                     void* dqkv, int64_t ldd, void* stream);

/* General MFMA attention, head_dim 64 (spatial branch vit.py:137-151; nn.MultiheadAttention in
 * tfm_model.py:43-48 with key_padding_mask; CLIP text causal mask).  S <= 416 without masks (the reference takes any crop
 * through its pos-embed resize, vit.py:374-386: 224^2 -> 197 tokens, 256^2 -> 257, 320^2 -> 401), S <= 208 with a causal /
 * key-padding mask; PVRL_EINVAL beyond.  The backward of 96 < S <= 224 without masks and with a power-of-two `scale` is ONE
 * persistent kernel (csrc/attn_bwd_fused.hip; PVRL_ATTN_BWD_FUSED=0 selects the two-pass kernels), that of 16 < S <= 32 contiguous
 * tokens one wave per (sequence, head) (csrc/attn_bwd_s32.hip; PVRL_ATTN_BWD_S32=0); `dvec` is then unused.
 * mode 0: row(seq, j) = seq*S + j.   mode 1 (TimeSformer spatial, seq = b*T + t): token 0 = cls row
 * cls_base + b, token j>=1 = row b*(S-1)*T + (j-1)*T + t; token-0 outputs go to the *_cls side buffers
 * ([nseq] rows).  lse/dvec: [nseq][H][S] fp32. */
int pvrl_attn_fwd(const void* qkv, int64_t ld, int64_t nseq, int64_t S, int64_t H, int mode, int64_t T,
                  int64_t cls_base, float scale, int causal, const void* key_padding_mask, void* o, void* o_cls,
                  int64_t ldo, float* lse, void* stream);
int pvrl_attn_bwd(const void* qkv, int64_t ld, int64_t nseq, int64_t S, int64_t H, int mode, int64_t T,
                  int64_t cls_base, float scale, int causal, const void* key_padding_mask, const void* o,
                  const void* o_cls, const void* d_o, const void* d_o_cls, int64_t ldo, const float* lse, float* dvec,
                  void* dqkv, void* dqkv_cls, int64_t ldd, void* stream);

/* The same attention (mode 1 addressing) for the cls query of every sequence ONLY -- the spatial attention of the encoder's LAST block,
 * of whose output only x[:, 0] is read (vit.py:418-421; csrc/attn_cls.hip).  Forward: o_cls [nseq][H*64] and lse[(seq*H + h)*S + 0] (the
 * other lse entries are not written).  Backward: dO is taken to be zero for every patch query: dK / dV of all S tokens and zeros for the
 * patch tokens' dQ (only with zero_patch_dq != 0: a caller that never reads that third of those rows saves the writes) go to dqkv
 * (token 0's partial row to dqkv_cls[seq], as pvrl_attn_bwd does), dQ of the cls query to dqkv_cls[seq].
 * S >= 2, nseq % T == 0. */
int pvrl_attn_cls_fwd(const void* qkv, int64_t ld, int64_t nseq, int64_t S, int64_t H, int64_t T, int64_t cls_base,
                      float scale, void* o_cls, int64_t ldo, float* lse, void* stream);
int pvrl_attn_cls_bwd(const void* qkv, int64_t ld, int64_t nseq, int64_t S, int64_t H, int64_t T, int64_t cls_base,
                      float scale, const void* o_cls, const void* d_o_cls, int64_t ldo, const float* lse, void* dqkv,
                      void* dqkv_cls, int64_t ldd, int zero_patch_dq, void* stream);

/* PatchEmbed im2col: frames fp32 [B,3,T,HI,WI] -> bf16 rows (b, n, t) x (c, py, px) (vit.py:174-180,396). */
int pvrl_patchify(const float* frames, int64_t B, int64_t T, int64_t HI, int64_t WI, void* out, int64_t ldo,
                  void* stream);
/* GPU-side input pipeline fused into the im2col: decoded uint8 frames [B,T,H0,W0,3] -> normalise ((v/255 - mean)/std),
 * short-side rescale to (new_h, new_w) (bilinear, align_corners=False), crop at (y_off, x_off), optional horizontal
 * flip -> bf16 patch rows of the crop x crop clip (same layout as pvrl_patchify).  Replaces the CPU-worker chain
 * lib/datasets/howto100m.py:437-452 -> utils.py:110-160,309-326 -> transform.py:8-147; the random draws stay on the
 * host (params int32 [B][5] = {new_h, new_w, y_off, x_off, flip}; mean3/std3 are HOST pointers to 3 floats). */
int pvrl_frames_u8_patchify(const void* frames, const int32_t* params, int64_t B, int64_t T, int64_t H0, int64_t W0,
                            int64_t crop, const float* mean3, const float* std3, void* out, int64_t ldo,
                            void* stream);
/* The same input pipeline producing the fp32 clip tensor [B,3,T,crop,crop] of the reference's loader (MViT stem input). */
int pvrl_frames_u8_to_f32(const void* frames, const int32_t* params, int64_t B, int64_t T, int64_t H0, int64_t W0,
                          int64_t crop, const float* mean3, const float* std3, float* out, void* stream);
/* E[n*T+t] = bias + pos_embed[1+n] + time_embed[t]  (vit.py:370-407), and its batch-summed gradient. */
int pvrl_embed_table(const float* pos, const float* time, const float* bias, float* E, int64_t N, int64_t T, int64_t C,
                     void* stream);
int pvrl_batch_sum(const float* dx, int64_t ld, int64_t B, int64_t rows, int64_t C, float* G, void* stream);
/* ... over the 16-bit patch rows of the split residual gradient stream (pvrl_rows), fp32 sums */
int pvrl_batch_sum_bf16(const void* dx, int64_t ld, int64_t B, int64_t rows, int64_t C, float* G, void* stream);
/* out bf16 = rowscale[m] * in fp32 (DropPath scaling lib/models/vit_utils.py:140-155; bf16 operand casts). */
int pvrl_cast_scale_bf16(const float* in, int64_t ldi, const float* rowscale, void* out, int64_t ldo, int64_t M,
                         int64_t C, void* stream);
int pvrl_cast_transpose_bf16(const float* in, void* out, int64_t R, int64_t C, void* stream);
/* both bf16 operand copies of an fp32 weight [R, C] in one pass: out [R, C] and (optional) out_t [C, R]. */
int pvrl_cast_weight_bf16(const float* in, void* out, void* out_t, int64_t R, int64_t C, void* stream);
/* cls-token bookkeeping of the spatial branch (vit.py:139-141,147-149):
 * out[g] = resid[g] + alpha * sum_t scale[g*G+t] * in[g*G+t];   out[g*G+t] = alpha*scale[g*G+t]*in[g]. */
int pvrl_group_reduce(const void* in, int in_is_f32, int64_t ldi, int64_t groups, int64_t G, int64_t C,
                      const float* scale, float alpha, const float* resid, int64_t ldr, void* out, int out_is_f32,
                      int64_t ldo, void* stream);
int pvrl_group_bcast_bf16(const float* in, int64_t ldi, int64_t groups, int64_t G, int64_t C, const float* scale,
                          float alpha, void* out, int64_t ldo, void* stream);

/* Loss head (vit.py:300-303; tools/train_net.py:152-162). */
int pvrl_l2norm_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, float* inv_norm, int64_t M, int64_t D,
                    void* stream);
int pvrl_l2norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* inv_norm, float* dx,
                    int64_t ldx, int64_t M, int64_t D, void* stream);
int pvrl_kl_topk(const float* pred, int64_t ldp, const float* teacher, int64_t ldt, int64_t rows, int64_t K,
                 int64_t topk, float grad_scale, float* row_loss, float* dpred, int64_t ldd, float* target_out,
                 int64_t ldto, void* stream);
int pvrl_mse(const float* a, const float* b, int64_t n, float grad_scale, float* loss, float* da, float* db,
             void* stream);
/* MIL-NCE over x[n][n][C] = video . text^T (lib/models/losses.py:15-23): nom[i] / den[i] log-sum-exps (loss =
 * mean(den - nom)) and dx = grad_scale * d(sum_i (den_i - nom_i)) / dx. */
int pvrl_milnce(const float* x, int64_t n, int64_t C, float grad_scale, float* nom, float* den, float* dx,
                void* stream);

/* Fused optimiser steps over flat fp32 buffers (torch.optim.AdamW / Adam / SGD-nesterov as built by
 * lib/models/optimizer.py:93-118; gscale = 1/num_iters of the accumulation branch, train_net.py:187-189).
 * decoupled = 1: AdamW, 0: Adam with L2 weight decay.  `step` counts from 1 and is the count of THIS range's parameters
 * (torch.optim keeps it per parameter).  Hyper-parameters are doubles, as torch.optim holds them: the fp32 constants of
 * the update (1 - lr*wd, 1 - beta, lr / bias_correction1 ...) are formed in double and rounded once, like torch's.
 * `skip` (device pointer or null): when *skip != 0 the call leaves p and the state untouched -- the device-side form of
 * `misc.check_nan_losses(loss)` raising in front of optimizer.step() (tools/train_net.py:174): the flag is the non-finiteness of
 * the loss (and, for the fp16 flavour, of the gradients: pvrl_nonfinite_flag_f32), no host sync per iteration. */
int pvrl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, int64_t step, double gscale, int decoupled, const float* skip, void* stream);
int pvrl_sgd_step(float* p, const float* g, float* buf, int64_t n, double lr, double momentum, double dampening,
                  double weight_decay, int nesterov, int first_step, double gscale, const float* skip, void* stream);
/* *flag = 1 when any of x[0, n) is inf / nan; never cleared here (zero it, then chain the buffers to check).  x 16-byte aligned. */
int pvrl_nonfinite_flag_f32(const float* x, int64_t n, float* flag, void* stream);
/* The fp16-operand flavour's per-backward gradient scale (procedurevrl_amd/engine.py GradStore.begin_scaled) in one launch: S = the power of
 * two that brings max|g| to ~target, chosen on the device (no host sync); out[0, n) = g * S, scale[0] = S, inv[0, ninv) = 1 / S (the `gscale`
 * of the gradient-writing entry points; repeated so that it can also serve as a GEMM epilogue's per-row scale).  inf / nan in g leave S finite. */
int pvrl_grad_scale_begin(const float* g, int64_t n, float target, float* out, float* scale, float* inv, int64_t ninv, void* stream);
/* End of an optimiser step: *total += 1 when *flag != 0 (total may be null), then *flag = 0 -- the flag's life cycle belongs to the
 * step that consumed it, whatever loop calls it. */
int pvrl_flag_roll(float* flag, float* total, void* stream);

/* softmax over the rows of an fp32 logit matrix: the eval-mode output `self.softmax(x)` (vit.py:355-356, mvit.py:203-204) */
int pvrl_softmax_rows_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int64_t N, void* stream);
/* exact-erf GELU on a small fp32 tensor (time_mlp, tfm_model.py:89-94): out = gelu(x), or out = dy * gelu'(x) when dy != 0 */
int pvrl_gelu_f32(const float* x, const float* dy, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * MViTv2 encoder path (reference lib/models/slowfast_mvit/).  Token matrices: rows [0, B*L) patch tokens
 * ordered (b, t, h, w), rows [B*L, B*L + B) the cls tokens; channel widths padded with zero columns to multiples of 128.
 * Pooled per-head tensors: [B*H][L' + 1][96] bf16 with the cls token LAST.
 * ------------------------------------------------------------------------------------------------------------------ */

/* im2col of PatchEmbed's Conv3d (stem_helper.py:290-321): frames fp32 [B,Cin,T,H,W] -> bf16 rows (b,to,ho,wo) x ldo columns,
 * column ((c*kt + a)*kh + y)*kw + x = the flatten order of Conv3d.weight; columns >= Cin*kt*kh*kw are zero. */
int pvrl_im2col3d_bf16(const float* frames, int64_t B, int64_t Cin, int64_t T, int64_t H, int64_t W, int64_t kt,
                       int64_t kh, int64_t kw, int64_t st, int64_t sh, int64_t sw, int64_t pt, int64_t ph, int64_t pw,
                       void* out, int64_t ldo, void* stream);

/* nn.LayerNorm of any width C <= 768 over fp32 rows with leading dimension ldx (mvit.py:78, attention.py:502,524);
 * y (bf16, or fp32 when y_is_f32) gets zeros in columns [C, Cpad).  Backward: dx = dres + dLN(dy); dgamma / dbeta are
 * ACCUMULATED into the given buffers from per-workgroup partials summed in a fixed order.  dx16 (optional, 16-bit,
 * [M][lddx16 >= Cpad]) receives rowscale16[row] * dx (rowscale16 optional) in the same pass: the operand copy the next
 * backward GEMM reads (DropPath factor folded in), which otherwise costs a pvrl_cast_scale_bf16 pass over dx. */
int pvrl_layernorm_g_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* y,
                         int64_t ldy, int y_is_f32, int64_t M, int64_t C, int64_t Cpad, float* mean, float* rstd,
                         void* stream);
int64_t pvrl_layernorm_g_bwd_workspace_bytes(int64_t M, int64_t C);
int pvrl_layernorm_g_bwd(const void* dy, int64_t lddy, int dy_is_f32, const float* x, int64_t ldx, const float* mean,
                         const float* rstd, const float* gamma, const float* dres, int64_t ldr, float* dx, int64_t lddx,
                         void* dx16, int64_t lddx16, const float* rowscale16, int64_t M, int64_t C, int64_t Cpad,
                         float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, void* stream);

/* attention_pool (attention.py:14-48) for mode "conv": depthwise Conv3d(96 ch, kernel 3x3x3, padding 1, stride st,sh,sw, no
 * bias; weight fp32 [96][27]) + LayerNorm(96) on one of q / k / v taken in place from the packed qkv activation (bf16
 * [B*T*Hh*Ww + B][ld], columns col0 + h*96 ..); the cls token skips the conv.  y / conv_out: [B*H][To*Ho*Wo + 1][96] bf16.
 * Backward writes this tensor's slice of dqkv and ACCUMULATES dw [96][27], dgamma, dbeta; dc_scratch: bf16, size of y;
 * workspace >= pvrl_mvit_pool_bwd_workspace_bytes() holds per-workgroup fp32 partials of dw that a second kernel sums in a
 * fixed order (same-address fp32 atomics from 8 XCDs cost ~0.5 us each and made every call ~200 us). */
int pvrl_mvit_pool_fwd(const void* qkv, int64_t ld, int64_t col0, int64_t B, int64_t H, int64_t T, int64_t Hh, int64_t Ww,
                       int64_t st, int64_t sh, int64_t sw, const float* w, const float* gamma, const float* beta,
                       float eps, void* y, void* conv_out, void* stream);
int64_t pvrl_mvit_pool_bwd_workspace_bytes(void);
int pvrl_mvit_pool_bwd(const void* dy, const void* conv_out, const void* qkv, void* dqkv, int64_t ld, int64_t col0,
                       int64_t B, int64_t H, int64_t T, int64_t Hh, int64_t Ww, int64_t st, int64_t sh, int64_t sw,
                       const float* w, const float* gamma, float eps, void* dc_scratch, float* dw, float* dgamma,
                       float* dbeta, void* workspace, int64_t workspace_bytes, void* stream);

/* MaxPool3d skip of MultiScaleBlock (attention.py:537-552): kernel (1,s+1,s+1), stride (1,s,s), padding (0,(s+1)/2,..),
 * fp32 token matrix in / out, cls rows copied.  Backward routes to the first maximum (torch semantics).
 * `argmax` (optional, uint8 [B*T*Ho*Wo][C]): the forward records each window's winner (window-local index), the backward
 * then routes by it (4 + 16 bytes per window) instead of re-scanning the windows of x (790 -> ~150 us at 25k tokens x 32 clips). */
int pvrl_mvit_maxpool_fwd(const float* x, int64_t ldi, int64_t B, int64_t T, int64_t H, int64_t W, int64_t s, int64_t C,
                          float* y, int64_t ldo, void* argmax, void* stream);
int pvrl_mvit_maxpool_bwd(const float* x, int64_t ldi, const float* dy, int64_t ldo, int64_t B, int64_t T, int64_t H,
                          int64_t W, int64_t s, int64_t C, float* dx, const void* argmax, void* stream);

/* Decomposed relative-position terms (attention.py:67-159): rel[bh][q][j] = Q[bh][q] . R_j(q), j over kh heights, kw
 * widths, kt times (each <= 16); R_j(q) = rel_pos_h[idx_h[qh(q)][j]] ... with the int32 index tables of
 * attention.py:80-98,130-137.  The forward writes the OPERAND FORM that pvrl_mvit_attn_* consume: relp, 16-bit,
 * [BH][Lq][2 * JP] with JP = pvrl_mvit_rel_width(kt, kh, kw) (32 or 64); columns [0, JP) hold hi and [JP, 2 JP) hold lo of
 * the pair hi + lo = out_scale * rel[bh][q][j] (~16 mantissa bits), zeros for j >= kh + kw + kt.  The attention kernels
 * expect out_scale = 1 / scale (the bias joins the q.k score before the scale is applied).
 * Backward: dQ += drel . R (in place on the 16-bit dQ of the attention backward; drel is the fp32 gradient with respect to
 * the UNSCALED rel, [BH][Lq][kh+kw+kt]), dR* ([nrows_*][96]) ACCUMULATED from per-workgroup partials in the workspace
 * (>= pvrl_mvit_rel_bwd_workspace_bytes) summed in a fixed order. */
int64_t pvrl_mvit_rel_width(int64_t kt, int64_t kh, int64_t kw);
int pvrl_mvit_rel_fwd(const void* Q, int64_t BH, int64_t qt, int64_t qh, int64_t qw, int64_t kt, int64_t kh, int64_t kw,
                      const float* Rh, const float* Rw, const float* Rt, const int32_t* idx_h, const int32_t* idx_w,
                      const int32_t* idx_t, float out_scale, void* relp, void* stream);
int64_t pvrl_mvit_rel_bwd_workspace_bytes(int64_t BH, int64_t qt, int64_t qh, int64_t qw, int64_t kt, int64_t kh,
                                          int64_t kw);
int pvrl_mvit_rel_bwd(const float* drel, const void* Q, void* dQ, int64_t BH, int64_t qt, int64_t qh, int64_t qw,
                      int64_t kt, int64_t kh, int64_t kw, const float* Rh, const float* Rw, const float* Rt,
                      const int32_t* idx_h, const int32_t* idx_w, const int32_t* idx_t, int64_t nrows_h, int64_t nrows_w,
                      int64_t nrows_t, float* dRh, float* dRw, float* dRt, void* workspace, int64_t workspace_bytes,
                      void* stream);

/* Pooling attention (attention.py:404-442): softmax(scale q k^T + rel bias) v (+ q, residual pooling) for head_dim 96;
 * q [B*H][Lq+1][96], k / v [B*H][kt*kh*kw+1][96]; o / d_o token-major [B*Lq + B][ldo] with column h*96 + d.
 * relp = the operand form written by pvrl_mvit_rel_fwd with out_scale = 1 / scale.  keymap = the 0/1 matrix
 * E[key][j] (j = h(key), kh + w(key), kh + kw + t(key)) of the key geometry as MFMA tile images, a function of
 * (kt, kh, kw) only: build it once with pvrl_mvit_attn_keymap into pvrl_mvit_attn_keymap_bytes bytes and reuse it.
 * lse / delta fp32 [B*H][Lq+1]; drel fp32 [B*H][Lq][kh+kw+kt] (gradient w.r.t. the unscaled rel).  The dK / dV kernel
 * shares the query range out over workgroups (fp32 partials in `workspace`, pvrl_mvit_attn_bwd_workspace_bytes) and
 * reduces deterministically. */
int64_t pvrl_mvit_attn_keymap_bytes(int64_t kt, int64_t kh, int64_t kw);
int pvrl_mvit_attn_keymap(int64_t kt, int64_t kh, int64_t kw, void* keymap, void* stream);
int pvrl_mvit_attn_fwd(const void* q, const void* k, const void* v, const void* relp, const void* keymap, int64_t B,
                       int64_t H, int64_t Lq, int64_t kt, int64_t kh, int64_t kw, float scale, void* o, int64_t ldo,
                       float* lse, void* stream);
int64_t pvrl_mvit_attn_bwd_workspace_bytes(int64_t B, int64_t H, int64_t Lq, int64_t kt, int64_t kh, int64_t kw);
int pvrl_mvit_attn_bwd(const void* q, const void* k, const void* v, const void* relp, const void* keymap, int64_t B,
                       int64_t H, int64_t Lq, int64_t kt, int64_t kh, int64_t kw, float scale, const void* o,
                       const void* d_o, int64_t ldo, const float* lse, float* delta, void* dq, void* dk, void* dv,
                       float* drel, void* workspace, int64_t workspace_bytes, void* stream);

/* pvrl_cast_weight_bf16 into caller-zeroed padded buffers: out bf16 [>=R][ldo], out_t bf16 [>=C][ldt] (MViT widths 96,
 * 192, 288, 441, 576 are padded to the GEMM tile multiples with zero rows / columns); `bias` (optional, fp32 [R]) is copied into the
 * caller-zeroed padded `bias_out` by the same launch. */
/* y[r] = beta * y[r] + sum_c W[r][c] * x[c] for a small dense matrix W (fp32, or bf16 when w_is_bf16), fp32 x and y:
 * the bias products of the fused temporal branch, b_e = W_fc b_proj and db_proj = W_fc^T db_e (the latter on the
 * transposed bf16 operand copy), vit.py:131-134.  gscale (device scalar or null) multiplies the product (see pvrl_gemm_tn_bf16). */
int pvrl_gemv_rows_f32(const void* W, int w_is_bf16, int64_t ld, int64_t R, int64_t C, const float* x, float beta, float* y,
                       const float* gscale, void* stream);

/* out[r][c] += a[r] * b[c] (fp32; C and ld multiples of 4): the bias term of the fused temporal branch's chain rule,
 * dW_fc += gscale * db_e b_proj^T -- what autograd adds to temporal_fc.weight.grad through proj's bias, vit.py:131-134
 * (gscale: device scalar or null = 1). */
int pvrl_rank1_add_f32(float* out, int64_t ld, const float* a, const float* b, int64_t R, int64_t C, const float* gscale,
                       void* stream);

/* pvrl_gemv_rows_f32 / pvrl_rank1_add_f32 for many equally-shaped problems in one launch each (HOST arrays of n device pointers; 16 per
 * launch): b_e = W_fc b_proj of every block at the start of a forward, db_proj = W_fc^T db_e and dW_fc += db_e b_proj^T of every block at
 * the end of a backward (vit.py:131-134). */
int pvrl_gemv_rows_batched_f32(int n, const void** W, int w_is_bf16, int64_t ld, int64_t R, int64_t C, const float** x,
                               const float* beta, float** y, const float* gscale, void* stream);
int pvrl_rank1_add_batched_f32(int n, float** out, int64_t ld, const float** a, const float** b, int64_t R, int64_t C,
                               const float* gscale, void* stream);

/* pvrl_cast_weight_bf16 for many weight matrices in one launch (the bf16 operand copies of every nn.Linear of the
 * encoder after an optimiser step): out [R][C] and, when out_t is not null, out_t [C][R], both dense. */
typedef struct pvrl_cast_problem {
  const float* in;   /* fp32 [R][C] */
  void* out;         /* bf16 [R][C] */
  void* out_t;       /* bf16 [C][R] or null */
  int64_t R, C;
} pvrl_cast_problem;
int pvrl_cast_weights_multi_bf16(int n, const pvrl_cast_problem* problems, void* stream);
int pvrl_cast_weight_pad_bf16(const float* in, void* out, int64_t ldo, void* out_t, int64_t ldt, int64_t R, int64_t C,
                              const float* bias, float* bias_out, void* stream);

/* The cls rows' own chain in fp32 (csrc/cls_chain.hip).  One cls token per clip passes every block (vit.py:139-157: mean over
 * the T frames of the spatial attention's cls outputs -> attn.proj -> residual; norm2 -> mlp.fc1 -> GELU -> mlp.fc2 ->
 * residual) and only those B rows reach the head (vit.py:418-421).  Their rounding errors do not average out the way the
 * patch tokens' do inside the attention, so these few rows are computed from the fp32 MASTER weights:
 *   epilogue 0:  out[m][n] = aux[m][n] + rowscale[m] * (X[m] . W[n]) + biasscale[m] * bias[n]      (each of aux / rowscale /
 *                biasscale / bias may be null = 0 / 1 / 1 / 0; rowscale = biasscale = the DropPath factor of vit.py:157, or
 *                biasscale = the mean DropPath factor over the T frames of vit.py:144-149)
 *   epilogue 1:  u = X[m] . W[n] + bias[n];  out = GELU_erf(u) (nn.GELU, vit.py:45); out16_pre / out16_act (optional, the
 *                library's 16-bit operand type, leading dimension ld16) receive u / GELU(u)
 * X [M, K], W [N, K], out fp32; N % 16 == 0, K % 128 == 0; HBM-bound on W, deterministic (fixed summation order).
 * Long reductions with few output columns (fc2: N = 768, K = 3072) are cut into four slices of K over workgroups whose products a
 * second small kernel sums in slice order: workspace >= pvrl_cls_linear_f32_workspace_bytes (0 / null when that returns 0). */
int64_t pvrl_cls_linear_f32_workspace_bytes(int64_t M, int64_t N, int64_t K);
int pvrl_cls_linear_f32(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, int64_t M, int64_t N,
                        int64_t K, int epilogue, const float* rowscale, const float* biasscale, const float* aux,
                        int64_t ld_aux, float* out, int64_t ldo, void* out16_pre, void* out16_act, int64_t ld16,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* out[r][c] = beta*out[r][c] + in[r][c] for an R x C block (unpadding weight gradients into the parameter's grad). */
int pvrl_copy2d_f32(const float* in, int64_t ldi, float* out, int64_t ldo, int64_t R, int64_t C, float beta, void* stream);

/* ---- collectives (RCCL over xGMI), for hosts that bind this library without torch.distributed ------------------------
 * Replaces: the DistributedDataParallel gradient reducer (lib/models/build.py:49-53) = in-place SUM all-reduce of the flat
 * fp32 gradient buffer (averaging is the optimiser's gscale), and the all-gather of lib/utils/distributed.py:13-50
 * (rank r's block lands at recv + r * bytes_per_rank).  One communicator per process / GPU; rank 0 creates the 128-byte id
 * and hands it to the others out of band (file, env, socket).  Asynchronous on `stream` like every other entry point.
 * RCCL is bound at run time: status -3 = librccl.so not found or a RCCL call failed. */
int pvrl_comm_unique_id(void* id128);
int pvrl_comm_init(void** comm, int world, int rank, const void* id128);
int pvrl_comm_allreduce_f32(void* comm, float* buf, int64_t n, void* stream);
int pvrl_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
/* rank r receives sum over ranks of send[r * n_per_rank, (r + 1) * n_per_rank) in recv (recv may be the caller's own shard of
 * send).  reducescatter + allgather of the reduced shards = the gradient all-reduce as two direct exchanges over all 7 xGMI
 * links of a fully-connected 8-GPU node (procedurevrl_amd/distributed.py GradReducer, PVRL_GRAD_COLL=rsag) in place of the
 * bucketed ring all-reduce of lib/models/build.py:49-53. */
int pvrl_comm_reducescatter_f32(void* comm, const float* send, float* recv, int64_t n_per_rank, void* stream);
int pvrl_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* PVRL_H_ */
