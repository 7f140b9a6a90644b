/* libhsenet_hip.so — C ABI of the MI355X (gfx950) kernels behind HSENet's dual 3D vision-encoder path.
 *
 * Every entry point takes raw DEVICE pointers, sizes and a hipStream_t (passed as void*), enqueues work on
 * that stream and returns an int status (HS_OK, or a negative HS_ERR_*).  No entry point allocates, frees
 * or synchronises (graph-capture safe); callers own every buffer, including workspaces whose size comes
 * from hs_vit_workspace_bytes().  No torch types cross this boundary.
 *
 * The reference (YanzhaoShi/HSENet) has no native layer: each function below names the reference Python
 * (or MONAI 1.3.0) operator sequence it replaces, relative to Preprint/LaMed/src/model/.  The Python host
 * mirror of the reference's module API lives in hsenet_amd/{encoder,projector}.py; the binding a reference
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions: bf16 = raw uint16 bits.  "rows" = B*N tokens.  Matrices are row-major.  prec: HS_PREC_*.
 */
#ifndef HSENET_HIP_H
#define HSENET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HS_OK 0
#define HS_ERR_SHAPE (-1)   /* unsupported / inconsistent shape or alignment */
#define HS_ERR_ARCH (-2)    /* no gfx950 device */
#define HS_ERR_LAUNCH (-3)  /* hip launch error */
#define HS_ERR_ARG (-4)     /* null pointer / bad enum */

#define HS_PREC_BF16 0      /* bf16 MFMA inputs, fp32 accumulate, fp32 residual stream (reference autocast policy) */
#define HS_PREC_F32 1       /* exact-f32 MFMA (v_mfma_f32_32x32x2_f32): parity mode */
#define HS_PREC_F16 2       /* IEEE fp16 MFMA inputs, fp32 accumulate, fp32 residual stream.  The reference's eval / generation scripts run the
                             * model under torch.cuda.amp.autocast() (default dtype fp16, Bench/eval/eval_HSENet_CT_Rate_MRG.py:383,423).
                             * The SAME kernels as HS_PREC_BF16 with fp16 operand packs and the fp16 MFMA opcode (same rate); since round 5
                             * forward AND backward: hs_gemm (every layout / epilogue), hs_gemm_group_prec, hs_attn_fwd (running-maximum
                             * kernel), hs_attn_bwd (HS_ATTN_F16), hs_layernorm_bwd (HS_LN_DY_F16 / HS_LN_OUT_F16), hs_colsum (2), hs_patchify,
                             * hs_cast_f16, hs_vit_blocks_fwd/bwd, hs_embed_*, hs_packer_* (descriptor prec).  fp16 has 5 exponent bits: a
                             * training step scales its loss gradient by a power of two (hs_mse_grad_scaled; the engines' loss_scale) and
                             * removes the factor in hs_adamw_masked's grad_scale; forward values need no scaling. */

/* GEMM operand layouts: C[M,N] = sum_k A(m,k) * B(n,k).  "T" = the operand is stored [K][dim] (k is the row). */
#define HS_GEMM_NT 0        /* A[M][K], B[N][K]   : forward nn.Linear  y = x W^T                      */
#define HS_GEMM_NN 1        /* A[M][K], B[K][N]   : dgrad              dx = dy W                     */
#define HS_GEMM_TN 2        /* A[K][M], B[K][N]   : wgrad              dW = dy^T x                   */

/* epilogue flags (applied in this order): v = alpha*acc; +bias[n]; GELU fwd (optionally save pre-act);
 * *gelu'(aux[m,n]); +resid[m,n]; store fp32 (or atomicAdd when split_k>1) and/or bf16. */
#define HS_EPI_BIAS 1
#define HS_EPI_GELU 2
#define HS_EPI_GELU_BWD 4
#define HS_EPI_RESID 8
#define HS_EPI_ATOMIC 16    /* out_f32 += v with float atomics (split-K wgrad; caller zeroes out_f32)          */
#define HS_EPI_BIAS_M 32    /* bias indexed by m (row) instead of n                                         */
#define HS_EPI_SAVE_DGELU 64 /* with HS_EPI_GELU: aux_out receives gelu'(z) instead of z; with HS_EPI_GELU_BWD: aux holds
                               gelu'(z) (the backward epilogue is then a plain multiply)                        */

typedef struct hs_gemm_args {
    const void* A;          /* bf16 (HS_PREC_BF16) or float (HS_PREC_F32) */
    const void* B;
    int M, N, K;
    int64_t lda, ldb;       /* row strides (elements) of A and B as stored */
    int batch;              /* grid.z batches (>=1) */
    int64_t strideA, strideB, strideC; /* batch strides, elements (strideC applies to every [M,N] tensor) */
    float* out_f32;         /* [M][ldc] or NULL */
    void* out_bf16;         /* [M][ldc] bf16 or NULL */
    int64_t ldc;
    const float* bias;      /* [N] (or [M] with HS_EPI_BIAS_M) */
    const float* resid;     /* [M][ldc] fp32 */
    const void* aux;        /* [M][ldc] pre-activation z for HS_EPI_GELU_BWD (bf16; float in F32 mode) */
    void* aux_out;          /* [M][ldc] pre-activation saved by HS_EPI_GELU (bf16), or NULL */
    float* colsum_out;      /* [N] or NULL: += column sums of the stored values (bias gradient fused into a dgrad) */
    float alpha;
    int flags;
    int layout;             /* HS_GEMM_* */
    int split_k;            /* >=1; >1 requires HS_EPI_ATOMIC and batch==1 (HS_PREC_F32: any batch, the grid carries batch x split_k) */
    int scale_cols;         /* 0, or a multiple of 4: output columns n < scale_cols get the extra factor           */
    float scale_cols_alpha; /*   scale_cols_alpha on top of alpha (before bias).  The ViT driver uses it to emit the q  */
                            /*   third of a qkv projection pre-multiplied by dh^-0.5 * log2(e): HS_ATTN_Q_PRESCALED      */
} hs_gemm_args;

/* ---- library ------------------------------------------------------------------------------------------ */
/* ABI version.  2 (round 6): every descriptor struct starts with `struct_size` (= sizeof of the struct the CALLER was compiled
 * against; a library built from a different header refuses the call with HS_ERR_ARG instead of reading past a shorter struct);
 * hs_vit_desc gained w16 / w16t (caller-owned 16-bit weight copies) and qkv_bias_grad, hs_embed_desc / hs_packer_desc carry `prec` since round 5;
 * hs_attn_bwd2, hs_attn_bwd_ws_bytes and hs_attn_bwd_ws_error (round 4) are gone, hs_attn_bwd's flag bits 4 and 8 are retired;
 * new: hs_vit_weights_elems, hs_vit_cast_weights, hs_adamw_fused.  A consumer checks hs_version() == HS_ABI_VERSION at load time
 * (hsenet_amd/_lib.py does). */
#define HS_ABI_VERSION 2
int hs_version(void);
/* 16 hex digits: content hash of every source, header and compiler flag the loaded library was built from
 * (hsenet_amd/build.py:source_id).  __graft_entry__.build() and smoke() compare it with the tree: a stale .so is an error. */
const char* hs_build_id(void);
/* HS_OK when device `dev` is gfx950 */
int hs_device_check(int dev);

/* ---- K3: GEMM family.  Replaces nn.Linear fwd/bwd inside MONAI SABlock/MLPBlock/PatchEmbeddingBlock
 * (call sites multimodal_encoder/vit.py:428-443), regular_attention (vit.py:39-45,57-62),
 * resolution_attention_v3 + proj_mpls (multimodal_projector/spatial_pooling_projector.py:51-57,73-82,127-131). */
int hs_gemm(const hs_gemm_args* args, int prec, void* stream);
/* `count` (1..HS_GEMM_GROUP_MAX) independent weight-gradient GEMMs in ONE launch: every problem bf16, HS_GEMM_TN, flags ==
 * HS_EPI_ATOMIC, batch 1 (the four dW = dY^T X of a transformer layer, vit.py:467's TransformerBlock backward).  Results equal
 * `count` hs_gemm calls; at one volume per GPU a lone wgrad cannot fill the chip without a deep split-K, the group can. */
#define HS_GEMM_GROUP_MAX 4
int hs_gemm_group(const hs_gemm_args* args, int count, void* stream);
/* the same with the operands' 16-bit format given: HS_PREC_BF16 (= hs_gemm_group) or HS_PREC_F16 (fp16 training, round 5) */
int hs_gemm_group_prec(const hs_gemm_args* args, int count, int prec, void* stream);

/* ---- K1: 3D patch embedding.  Replaces x.clone() + einops 'b c (h p1)(w p2)(d p3) -> b (h w d)(p1 p2 p3 c)'
 * (vit.py:437,455) = the A operand of the patch-embed Linear.  out: [B*Np][Pd] bf16 (prec 0) / float (prec 1). */
int hs_patchify(const float* vol, void* out, int B, int C, int D, int H, int W, int p1, int p2, int p3, int prec,
                void* stream);
/* tokens[b,0,:] = cls; tokens[b,1+p,:] = (pe[b,p,:] + pos[p,:]) * (score ? score[b,p] : 1).   (vit.py:345-349,459-461)
 * pe_out (optional) receives pe+pos (the un-weighted embedding, needed by the stage-2 scorer and its backward).
 * cls == NULL (classification=False): no cls row is written and tokens is [B][Np][d]. */
int hs_assemble_tokens(const float* pe, const float* pos, const float* cls, const float* score, float* tokens,
                       float* pe_out, int B, int Np, int d, void* stream);
/* backward of hs_assemble_tokens: dpe[b,p,:] = dtok[b,1+p,:]*score; dcls += sum_b dtok[b,0,:]; dpos += sum_b dpe;
 * dscore[b,p] = <dtok[b,1+p,:], pe_pos[b,p,:]> (when score given). dpos/dcls accumulate (caller zeroes).
 * dcls == NULL <=> the forward had no cls row (dtok is [B][Np][d]). */
int hs_assemble_tokens_bwd(const float* dtok, const float* pe_pos, const float* score, float* dpe, float* dpos,
                           float* dcls, float* dscore, int B, int Np, int d, void* stream);

/* ---- K2: LayerNorm (nn.LayerNorm(d), eps 1e-5; MONAI TransformerBlock.norm1/2, vit.py:467, vit.py:62).
 * x fp32 [rows][d]; optional resid added first (y = LN(x + resid)); writes y as fp32 and/or bf16; mean/rstd optional. */
int hs_layernorm_fwd(const float* x, const float* resid, const float* gamma, const float* beta, float* y_f32,
                     void* y_bf16, float* mean, float* rstd, int rows, int d, float eps, void* stream);
/* dx = LNbwd(dy) (+ dx_in when given); dy fp32, bf16 or fp16 (dy_is_bf16 = HS_LN_DY_*); also emits a 16-bit copy of the result (bf16; fp16
 * with HS_LN_DY_F16, or with HS_LN_DY_F32 | HS_LN_OUT_F16 = fp32 dy at the top of an fp16 backward pass);
 * dgamma/dbeta accumulate atomically (caller zeroes).  x (and resid) are the forward inputs.  dxsum (optional, [d])
 * accumulates the column sums of the produced dx = the bias gradient of the Linear this dx is the output-gradient of. */
#define HS_LN_DY_F32 0
#define HS_LN_DY_BF16 1
#define HS_LN_DY_F16 2
#define HS_LN_OUT_F16 4
int hs_layernorm_bwd(const void* dy, int dy_is_bf16, const float* x, const float* resid, const float* gamma,
                     const float* mean, const float* rstd, const float* dx_in, float* dx, void* dx_bf16, float* dgamma,
                     float* dbeta, float* dxsum, int rows, int d, void* stream);

/* ---- K4: global multi-head self-attention core of MONAI SABlock: softmax(q k^T * dh^-0.5) v, no mask, no
 * bias, non-causal (reached from the hot loops vit.py:352-354,464-466).  qkv: [B][N][3][H][dh] exactly as
 * the qkv Linear emits it ('b n (qkv h dh)'); o: [B][N][H*dh]; lse: [B][H][N] fp32 (natural log-sum-exp of the
 * scaled scores).  dh must be 64.  Flash-style: the N x N matrix is never materialised.
 * flags: HS_ATTN_Q_PRESCALED = the q third of qkv already holds q * dh^-0.5 * log2(e) (hs_gemm's scale_cols emits it
 * that way, one rounding); without it the bf16 kernels apply that factor to their register fragments (a second
 * bf16 rounding of q).  o, lse and dqkv mean the same either way (dqkv is the gradient of the UNSCALED q, k, v). */
#define HS_ATTN_Q_PRESCALED 1
#define HS_ATTN_F32 2          /* hs_attn_bwd only: qkv, o, d_o, dqkv are fp32 (exact-f32 MFMA parity mode; raw q) */
#define HS_ATTN_F16 16         /* hs_attn_bwd only: qkv, o, d_o, dqkv are IEEE fp16 (HS_PREC_F16 training: the two bf16 sweeps with fp16
                                * packs and the fp16 MFMA opcode; P = exp2(S') <= 1 and dS = P dP' sit in fp16's range for a loss-scaled dO) */
int hs_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int dh, int prec, int flags, void* stream);
/* dqkv from (qkv, o, do, lse); delta_ws: fp32 scratch 2 x [B][H][N] (row constants -sum(dO*O) and -lse*log2(e)).
 * bf16 tensors, fp16 ones with HS_ATTN_F16, or fp32 ones with HS_ATTN_F32 (parity mode: the same two sweeps on
 * v_mfma_f32_32x32x2_f32, no rounding). */
int hs_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, float* delta_ws,
                int B, int N, int H, int dh, int flags, void* stream);

/* ---- K5: slice-guided scoring core of regular_attention (vit.py:25-33,57-60): single head, d_k = d,
 * P = softmax(q k^T / sqrt(d)) over nk keys (nk <= 64, nk % 4 == 0), out = P v.  q: [B][nq][d] fp32;
 * k, v: [B][nk][d] fp32.  p: [B][nq][nk] fp32, REQUIRED (the attention map the reference returns; saved for bwd).
 * Runs as exact-f32 MFMA GEMMs + a row softmax (4*nq*nk*d FLOPs: negligible).  drop_p = 0, pd = NULL: eval mode. */
int hs_cross_attn_fwd(const float* q, const float* k, const float* v, float* out, float* p, float* pd, int B, int nq, int nk,
                      int d, float drop_p, uint64_t seed, uint64_t offset, void* stream);
/* ws: fp32 scratch [B][nq][nk].  Train mode (module in train(): the reference's hard-coded Dropout(0.1) on the probabilities,
 * vit.py:46): drop_p > 0 with a Philox mask that is a pure function of (seed, offset, element index); pd receives
 * dropout(P) (what the reference returns) and out = pd V; the backward call takes the same pd / drop_p / seed / offset. */
int hs_cross_attn_bwd(const float* q, const float* k, const float* v, const float* p, const float* pd, const float* dout,
                      float* dq, float* dk, float* dv, float* ws, int B, int nq, int nk, int d, float drop_p, uint64_t seed,
                      uint64_t offset, void* stream);
/* y = x * keep / (1 - p), mask = Philox4x32-10(seed, offset + i/4): elementwise dropout; apply the same call to the gradient. */
int hs_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);
/* score[r] = sigmoid(<x[r,:], w> + b)   (patch_score_proj + Sigmoid, vit.py:338-339).  x fp32 [rows][d]. */
int hs_rowdot_sigmoid_fwd(const float* x, const float* w, const float* b, float* score, int rows, int d, void* stream);
/* dx[r,:] = dscore[r]*s(1-s)*w ; dw += sum_r ..*x[r,:] ; db += sum_r ..   (dw/db accumulate atomically) */
int hs_rowdot_sigmoid_bwd(const float* x, const float* w, const float* score, const float* dscore, float* dx,
                          float* dw, float* db, int rows, int d, void* stream);

/* ---- K6: Spatial Packer window step (spatial_pooling_projector.py:62-76,138-141).
 * avg-pool: lr[b,cell,:] = mean over the cell's kd*kh*kw children of hr[b,child,:]; hr is the token grid
 * [B][gd][gh][gw][d] fp32, cells ordered row-major over (gd/kd, gh/kh, gw/kw). */
int hs_pool3d_fwd(const float* hr, float* lr, int B, int gd, int gh, int gw, int kd, int kh, int kw, int d, void* stream);
int hs_pool3d_bwd(const float* dlr, float* dhr, int B, int gd, int gh, int gw, int kd, int kh, int kw, int d,
                  int accumulate, void* stream);
/* one query per cell vs its children: out[b,cell,:] = sum_j softmax_j(<q,k_j>/sqrt(d)) v_j.
 * q: [B][cells][d]; kv: [B][gd*gh*gw][2*d] fp32 (k | v per token, as one fused Wk|Wv GEMM emits it). */
int hs_window_attn_fwd(const float* q, const float* kv, float* out, int B, int gd, int gh, int gw, int kd, int kh,
                       int kw, int d, float drop_p, uint64_t seed, uint64_t offset, void* stream);
int hs_window_attn_bwd(const float* q, const float* kv, const float* dout, float* dq, float* dkv, int B, int gd,
                       int gh, int gw, int kd, int kh, int kw, int d, float drop_p, uint64_t seed, uint64_t offset, void* stream);

/* ---- K7/K8: small fused element/reduction kernels ------------------------------------------------------- */
/* out[n] (+)= sum_m x[m][n]  (bias gradients).  x bf16 (is_bf16) or fp32.  accumulates atomically (caller zeroes). */
int hs_colsum(const void* x, int is_bf16, float* out, int M, int N, int64_t ldx, void* stream);
/* fp32 -> bf16 cast of n elements (weights before each step; activations at module boundaries) */
int hs_cast_bf16(const float* x, void* y, int64_t n, void* stream);
int hs_cast_f32(const void* x_bf16, float* y, int64_t n, void* stream);
/* fp32 -> IEEE fp16, round to nearest even (= torch's .half()): the operand cast of HS_PREC_F16 */
int hs_cast_f16(const float* x, void* y_f16, int64_t n, void* stream);
/* y = gelu(x) fp32, exact erf (nn.GELU in proj_mpls, spatial_pooling_projector.py:129) and its backward */
int hs_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int hs_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* y[r,:] = x[r,:] / max(||x[r,:]||, eps)   (F.normalize, CLIP_stage1.py:101) and backward */
int hs_l2norm_fwd(const float* x, float* y, int rows, int d, float eps, void* stream);
int hs_l2norm_bwd(const float* x, const float* dy, float* dx, int rows, int d, float eps, void* stream);
/* fused AdamW over a flat fp32 parameter buffer (optim="adamw_torch", train/train_CLIP_stage1.py:85).
 * grad_scale multiplies the gradient first (1/world for DP mean). step >= 1. */
int hs_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
             float weight_decay, float grad_scale, int step, void* stream);

/* ---- N1: CLIP InfoNCE head (CLIP_stage1.py:141-155): symmetric cross-entropy with labels = arange(n) on an n x n
 * fp32 logits matrix.  loss[0] += (CE(logits) + CE(logits^T)) / 2;  dlogits = dL/dlogits;  loss[1] += sum(dlogits*logits)
 * (= logit_scale * dL/dlogit_scale).  loss: 2 floats, caller zeroes. */
int hs_clip_ce(const float* logits, float* dlogits, float* loss, int n, void* stream);

/* y = alpha * x (fp32): the synthetic bench loss gradient dL/dy = y * 2/numel of L = mean(y^2) */
int hs_scale(const float* x, float* y, float alpha, int64_t n, void* stream);

/* ---- N2: CT volume preprocessing in front of the path (Data/data_processing/CT-RATE/
 * CT-RATE_nii_to_3D_volume_npy_file.py:25-124) and the train-time augmentation (LaMed/src/dataset/multi_dataset.py:148-158).
 * hs_prep_hu_window: out[d][h][w] (float64) = clamp(slope*raw[h][w][d] + intercept, lo, hi), raw (H,W,D) float32/float64.
 * hs_prep_resample: trilinear, align_corners=False (torch F.interpolate(size=..)), of the window [c0, c0+c) of a
 *   (iD,iH,iW) volume to (oD,oH,oW) float32; every tap is mapped v -> (v - sub)/den first (min-max fused; 0, 1 = none);
 *   float64 input is interpolated in float64 (resize_array on the f64 tensor), float32 input in float32 (MONAI Resize).
 * hs_prep_minmax: out2 = {min, max}, caller initialises {+inf, -inf}.  hs_prep_foreground_bbox: box6 = {min d,h,w,
 *   max d,h,w} of {x > thresh} (MONAI CropForeground with its default select_fn), caller initialises {D,H,W,-1,-1,-1}.
 * hs_prep_augment: np.rot90(rot_k, axes=(H,W)) -> flips (bit0 D, bit1 H, bit2 W) -> *scale -> +shift on (C,D,H,W). */
int hs_prep_hu_window(const void* raw, int raw_is_f64, double* out, int H, int W, int D, double slope, double intercept,
                      double lo, double hi, void* stream);
int hs_prep_resample(const void* in, int in_is_f64, float* out, int iD, int iH, int iW, int cd0, int ch0, int cw0, int cD,
                     int cH, int cW, int oD, int oH, int oW, float sub, float den, void* stream);
int hs_prep_minmax(const float* x, int64_t n, float* out2, void* stream);
int hs_prep_foreground_bbox(const float* x, int D, int H, int W, float thresh, int* box6, void* stream);
int hs_prep_augment(const float* in, float* out, int C, int D, int H, int W, int rot_k, int flip_mask, float scale, float shift,
                    void* stream);

/* ---- measurement: per-kernel-class timing with HIP events on the launch stream (bench.py roofline leg).
 * hs_prof_enable(1) clears and starts recording, (0) stops.  hs_prof_read synchronises on the recorded events
 * (the only synchronising call in the library) and returns launches, summed duration, and the algorithmic
 * FLOPs / bytes the launches accounted for.  Class ids 0..hs_prof_count()-1, names from hs_prof_name(). */
int hs_prof_enable(int on);
int hs_prof_read(int id, int* count, double* total_ms, double* flops, double* bytes);
const char* hs_prof_name(int id);
int hs_prof_count(void);

/* ---- fused encoder driver: the hot loops vit.py:464-467 / 352-355 (12 x MONAI TransformerBlock + final norm) as
 * one enqueue.  All pointers are device pointers.
 * layer_params: L*11 fp32 tensors in MONAI state-dict order per block:
 *   [mlp.linear1.weight, mlp.linear1.bias, mlp.linear2.weight, mlp.linear2.bias, norm1.weight, norm1.bias,
 *    attn.out_proj.weight, attn.out_proj.bias, attn.qkv.weight, norm2.weight, norm2.bias]
 * norm_params: [norm.weight, norm.bias].  x0: tokens [B][N][d] fp32.  y: [B][N][d] fp32 (post-norm).
 * hidden: [L][B][N][d] fp32 (the 12 hidden states the reference returns).  ws: workspace. */
typedef struct hs_vit_desc {
    uint32_t struct_size; /* sizeof(hs_vit_desc) of the caller's header (ABI 2) */
    int B, N, d, H, mlp, L;
    int prec;            /* HS_PREC_* */
    int save_for_bwd;    /* 1: keep per-layer activations in ws for hs_vit_blocks_bwd; 2 (HS_PREC_BF16): the same WITHOUT the MLP's two
                          * [rows][mlp] activations (46 % of the saved bytes): the backward re-runs fc1 + GELU per layer (+1 GEMM of 12) */
    float ln_eps;        /* LayerNorm epsilon; 0 = 1e-5 (MONAI / nn.LayerNorm default).  timm's ViT (the 2-D branch) uses 1e-6 */
    const float* const* qkv_bias;   /* NULL (MONAI SABlock, qkv_bias=False: every shipped call site), or L device pointers to the [3d] bias
                                       of each layer's qkv Linear (MONAI qkv_bias=True, vit.py:385; timm's ViT in the 2-D branch) */
    /* 16-bit precisions only: CALLER-OWNED 16-bit copies of the four weight matrices of every layer, hs_vit_weights_elems(desc)
     * elements each, layer l at l * (elems / L) in the order [Wqkv | Wo | W1 | W2]; w16 = the matrices as stored ([out][in]), w16t =
     * their transposes ([in][out]: the dgrads' NT operand).  NULL (either): the driver re-makes that copy inside `ws` on every call
     * (forward: w16; backward: w16t of the layers it runs).  Non-NULL: the driver reads it and casts nothing -- the caller keeps it
     * equal to the fp32 parameters (hs_vit_cast_weights after a parameter change; hs_adamw_fused writes w16 as it updates). */
    const void* w16;
    const void* w16t;
    float* const* qkv_bias_grad;    /* hs_vit_blocks_bwd with qkv_bias: NULL (biases frozen), or L device pointers; layer l's [3d] bias
                                       gradient (the column sums of dqkv) ACCUMULATES into entry l (NULL entries are skipped) */
} hs_vit_desc;

size_t hs_vit_workspace_bytes(const hs_vit_desc* desc);
/* elements of ONE 16-bit weight-copy set (L * (4 d^2 + 2 d mlp)); 0 for a bad descriptor or HS_PREC_F32 */
size_t hs_vit_weights_elems(const hs_vit_desc* desc);
/* fill caller-owned weight copies in the descriptor's 16-bit format: w16 (straight) and / or w16t (transposed; d and mlp multiples
 * of 64) from the fp32 layer_params (MONAI order, as hs_vit_blocks_fwd takes them).  layer_params == NULL with both w16 and w16t
 * given: w16t is made from w16 (16-bit source: what an optimizer that already wrote w16 needs).  One or two launches. */
int hs_vit_cast_weights(const hs_vit_desc* desc, const float* const* layer_params, void* w16, void* w16t, void* stream);
int hs_vit_blocks_fwd(const hs_vit_desc* desc, const float* x0, const float* const* layer_params,
                      const float* const* norm_params, float* y, float* hidden, void* ws, size_t ws_bytes, void* stream);
/* the same for layers [l_lo, l_hi) only (x0 is read when l_lo == 0; the final norm runs and y is written when l_hi == L): a caller whose
 * optimizer pass overlaps the forward (hs_adamw_fused, HS_ADAMW_SWEEP) enqueues the stack chunk by chunk, each behind the event of the
 * sweep that updated its layers.  A proper sub-range needs caller-owned weight copies (desc.w16) in the 16-bit precisions: the
 * per-call cast of ALL layers' weights cannot run before the later layers' sweeps have. */
int hs_vit_blocks_fwd_range(const hs_vit_desc* desc, const float* x0, const float* const* layer_params,
                            const float* const* norm_params, float* y, float* hidden, void* ws, size_t ws_bytes, int l_lo, int l_hi,
                            void* stream);
/* backward of the above (either precision; the save_for_bwd=1 forward must have run on the same ws).
 * dy: [B][N][d] fp32.  dx0: [B][N][d] fp32 out.  layer_grads / norm_grads: same order as the params; written
 * (weights) or accumulated atomically (biases, LN): the caller zeroes all gradient buffers before the call.
 * Layers are processed from l_hi-1 down to l_lo (inclusive), so a caller can interleave gradient all-reduce
 * buckets: call with (L, L-3), then (L-3, L-6), ...  The final norm's backward runs when l_hi == L; dx0 is valid
 * when l_lo == 0. */
int hs_vit_blocks_bwd(const hs_vit_desc* desc, const float* x0, const float* const* layer_params,
                      const float* const* norm_params, const float* dy, float* dx0, float* const* layer_grads,
                      float* const* norm_grads, const float* hidden, void* ws, size_t ws_bytes, int l_hi, int l_lo,
                      void* stream);

/* ---- fused embedding driver: P1 (MONAI PatchEmbeddingBlock: Linear over the patchified volume + bias + learned positions,
 * vit.py:428-437,455) and the token assembly of ViT_stage1 (cls concat, vit.py:459-461); with scorer = 1 also P5/P6 of
 * ViT_stage2 (vit.py:325-349): x = embedding + pos; regular_attention(x, sem, sem) (single head, d_k = d, softmax over the ns
 * slice features, Dropout(0.1) on the probabilities and on the pre-residual output in train mode, LN(q + out), vit.py:36-64);
 * score = sigmoid(patch_score_proj(.)); tokens = cat(cls, x * score).
 * patches: [B*Np][Pd] bf16 from hs_patchify (shared by both encoders of the dual tower: the volume is read once).
 * image_2d: [B][ns][d] fp32 (the pre-computed BiomedCLIP slice features; scorer only).  tokens: [B][Np + has_cls][d] fp32.
 * params / grads: 16 device pointers in the reference's state-dict order of these tensors:
 *   [cls_token, patch_embedding.position_embeddings, patch_embedding.patch_embeddings.1.weight, ...1.bias,
 *    patch_score_proj.weight, .bias, slice_guided_attention.Wq.weight, .bias, Wk.weight, .bias, Wv.weight, .bias,
 *    output_linear.weight, .bias, slice_guided_attention.norm.weight, .bias]     (entries 4.. ignored when scorer = 0;
 *    entry 0 ignored when has_cls = 0).  Gradients ACCUMULATE (split-K atomics): the caller zeroes them.
 * seed/offset: Philox streams of the two dropout sites [probabilities, pre-residual output]; the backward call must get the
 * same descriptor as the forward call it differentiates. */
typedef struct hs_embed_desc {
    uint32_t struct_size; /* sizeof(hs_embed_desc) of the caller's header (ABI 2) */
    int B, Np, Pd, d;
    int has_cls;          /* classification=True: row 0 of every volume's tokens is cls_token */
    int scorer;           /* 0 = ViT_stage1, 1 = ViT_stage2 (slice-guided scorer) */
    int ns;               /* slices per volume (32, vit.py:332) */
    int save_for_bwd;
    float drop_p;         /* 0 = eval(); 0.1 = train() (vit.py:46-47) */
    uint64_t seed[2], offset[2];
    int prec;             /* HS_PREC_BF16 (0) or HS_PREC_F16: the 16-bit operand format of the driver's GEMMs (patches must be in it) */
} hs_embed_desc;
size_t hs_embed_workspace_bytes(const hs_embed_desc* desc);
int hs_embed_fwd(const hs_embed_desc* desc, const void* patches, const float* image_2d, const float* const* params,
                 float* tokens, void* ws, size_t ws_bytes, void* stream);
int hs_embed_bwd(const hs_embed_desc* desc, const void* patches, const float* dtokens, const float* const* params,
                 float* const* grads, void* ws, size_t ws_bytes, void* stream);

/* ---- fused Spatial Packer driver: P2 VisualPacker_3d_phi_v3.forward (multimodal_projector/spatial_pooling_projector.py:138-146)
 * = view as token grid (:140) -> avg_pool3d (:141) -> resolution_attention_v3 (:62-83: Wq on the pooled cells, ONE [Wk;Wv] GEMM
 * over every HR token, 1 query x kd*kh*kw children softmax, Dropout(0.1) x2 in train mode, output_linear, LN(q + .)) ->
 * proj_mpls Linear-GELU-Linear (:127-131,144).
 * feat: fp32 token tensor; volume b's HR tokens are rows [in_row_offset, in_row_offset + gd*gh*gw) of the block starting at
 *   feat + b*in_batch_stride (elements): in_row_offset = 1, in_batch_stride = (T+1)*d reads an encoder output [B][T+1][d] in place,
 *   skipping the cls row (select_feature == 'patch', vit.py:934-936).
 * out: fp32 [B][cells][out_dim] with batch stride out_batch_stride (elements): 2*cells*out_dim lets the two packers of the dual
 *   tower fill the two halves of one [B][2*cells][out_dim] tensor (lamed_arch.py:127-132's cat) without a copy.
 * params / grads: 14 pointers in state-dict order [proj_mpls.0.weight, .0.bias, proj_mpls.2.weight, .2.bias,
 *   resolution_attention.Wq.weight, .bias, Wk.weight, .bias, Wv.weight, .bias, output_linear.weight, .bias, norm.weight, .bias];
 *   gradients accumulate (caller zeroes).  dfeat (optional): gradient w.r.t. feat, same layout as feat, rows below
 *   in_row_offset zeroed (dense [B][in_row_offset + T][d]: requires in_batch_stride == (in_row_offset + T)*d). */
typedef struct hs_packer_desc {
    uint32_t struct_size; /* sizeof(hs_packer_desc) of the caller's header (ABI 2) */
    int B, gd, gh, gw, kd, kh, kw, d, out_dim;
    int in_row_offset;
    int64_t in_batch_stride, out_batch_stride;
    int save_for_bwd;
    float drop_p;
    uint64_t seed[2], offset[2];
    int prec;             /* HS_PREC_BF16 (0) or HS_PREC_F16 */
} hs_packer_desc;
size_t hs_packer_workspace_bytes(const hs_packer_desc* desc);
int hs_packer_fwd(const hs_packer_desc* desc, const float* feat, const float* const* params, float* out, void* ws, size_t ws_bytes,
                  void* stream);
int hs_packer_bwd(const hs_packer_desc* desc, const float* dout, const float* const* params, float* const* grads, float* dfeat,
                  void* ws, size_t ws_bytes, void* stream);

/* ---- train-step glue (the measurement harness' stand-in for HF Trainer: optim adamw_torch, weight_decay 0.1, max_grad_norm 1.0)
 * hs_mse_grad: L = mean(y^2): dy = 2 y / n, loss[0] += L (loss optional, caller zeroes).
 * hs_sumsq: out[0] = sum(x^2) -- the squared global gradient norm, kept on the device.  Deterministic (fixed reduction order, no
 *   atomics; scratch: 1024 floats): every data-parallel rank must derive the SAME clip coefficient or the replicas drift.
 * hs_adamw_masked: hs_adamw with (a) decay_mask: one byte per 64-element chunk of the flat buffer, 0 = no weight decay (biases and
 *   LayerNorm weights, as HF Trainer's get_decay_parameter_names), NULL = decay everything; (b) grad_sumsq (device, optional):
 *   squared norm of the SUMMED gradient; the update uses g * grad_scale * min(1, max_grad_norm / (sqrt(sumsq)*grad_scale + 1e-6))
 *   = torch.nn.utils.clip_grad_norm_ on the DP-mean gradient, without a host round trip. */
int hs_mse_grad(const float* y, float* dy, float* loss, int64_t n, void* stream);
/* hs_mse_grad with dy multiplied by grad_scale (> 0): the power-of-two loss scale of an HS_PREC_F16 backward pass -- every gradient of
 * the step then carries the factor, and hs_adamw_masked's grad_scale (1 / (world * loss_scale)) takes it out; `loss` is unscaled.
 * hs_adamw_masked skips the update (parameters and moments untouched) when grad_sumsq is not finite: an overflowed fp16 step. */
int hs_mse_grad_scaled(const float* y, float* dy, float* loss, int64_t n, float grad_scale, void* stream);
int hs_sumsq(const float* x, float* out, float* scratch, int64_t n, void* stream);
int hs_adamw_masked(float* p, const float* g, float* m, float* v, const unsigned char* decay_mask, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, float grad_scale, const float* grad_sumsq, float max_grad_norm,
                    int step, void* stream);

/* hs_adamw_fused (round 6): the optimizer pass of a train step as ONE sweep over the flat buffers -- hs_adamw_masked plus
 *  (a) zero_grad != 0: g is zeroed behind the read (the next backward accumulates into it: no separate memset);
 *  (b) w16_table (device, optional): one destination ADDRESS per 64-element chunk of the flat buffer (0 = none); the updated
 *      parameters of a listed chunk are also written there as 64 bf16 (w16_is_f16 = 0) or fp16 (1) values -- the 16-bit weight
 *      copies the next forward multiplies with (hs_vit_desc.w16), so that no cast pass re-reads the fp32 masters;
 *  (c) state (device, REQUIRED, 8 x 4 bytes, caller zeroes it once): [0] int32 steps applied, [1] int32 steps SKIPPED, [2..7]
 *      scratch.  The step number of the bias corrections is state[0] + 1 -- kept on the device, so that a skipped step (grad_sumsq
 *      not finite: an fp16 backward that overflowed its loss scale) does not advance it (torch.cuda.amp.GradScaler semantics) and
 *      the host can read both counters whenever it likes without a sync inside the step.
 *  grad_sumsq (device): squared norm of the summed gradient; required when max_grad_norm > 0, optional otherwise (given without
 *  clipping it only arms the non-finite skip).  n must be a multiple of 64.
 *  phase: HS_ADAMW_STEP = the whole pass over [p, p + n).  A caller that wants the optimizer pass to OVERLAP the next forward splits
 *  it: one HS_ADAMW_PREPARE call (the one-thread prologue: skip decision, step counter, clip coefficient, bias corrections -> state;
 *  p / g / m / v / n ignored) and then HS_ADAMW_SWEEP calls over disjoint 64-aligned ranges of the buffers (pointers advanced by the
 *  range start, decay_mask and w16_table by start / 64) in the order the next forward needs the parameters, on any stream ordered behind
 *  the prologue -- the engines sweep layer chunk by layer chunk on a side stream and let each forward chunk wait for its event. */
#define HS_ADAMW_STEP 0
#define HS_ADAMW_PREPARE 1
#define HS_ADAMW_SWEEP 2
int hs_adamw_fused(float* p, float* g, float* m, float* v, const unsigned char* decay_mask, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float grad_scale, const float* grad_sumsq, float max_grad_norm,
                   int zero_grad, const uint64_t* w16_table, int w16_is_f16, int32_t* state, int phase, void* stream);

/* dst[b*dst_batch_stride + r*d + :] = src[b*src_batch_stride + r*d + :] for r < R, as fp32 and/or bf16: a block of R rows per batch
 * entry between two strided [B][.][d] tensors (the cls row of every volume: R = 1, src stride N*d = `[:, 0]`, CLIP_stage1.py:117;
 * the patch rows: src + d, R = N-1 = `[:, 1:]`, vit.py:934-936).  hs_rows_scatter: dst [B][Rt][d] gets the dense src rows at
 * [r0, r0+R) and zeros elsewhere (the adjoint of such a row selection). */
int hs_rows_copy(const float* src, int64_t src_batch_stride, float* dst_f32, void* dst_bf16, int64_t dst_batch_stride, int B, int R, int d,
                 void* stream);
int hs_rows_scatter(const float* src, float* dst, int B, int r0, int R, int Rt, int d, void* stream);

/* ---- N1 glue with the logit scale ON THE DEVICE (it is a trained nn.Parameter, CLIP_stage2.py:99; no host read per step)
 * hs_scale_dev: y = x * alpha_dev[0] * alpha (alpha_dev NULL: y = x * alpha).  hs_dot: out[0] += alpha * sum(a*b).
 * hs_relation_mse: image_text_relation_regulation on ONE pair of logit matrices (CLIP_stage2.py:187-190): loss[0] += mean((l-l_ref)^2);
 *   dl = (accumulate ? dl : 0) + weight * 2/n * (l - l_ref)   (weight = the relation_loss_weight schedule, :141-149).
 * hs_ce_rows: F.cross_entropy(logits [rows][cols], labels = label0 + row), the local_loss branch (:173-177):
 *   loss[0] += weight * mean CE;  dlogits = (accumulate ? dlogits : 0) + weight * (softmax - onehot) / rows.
 * hs_rowmix: out[i, :] = sum_j w(i,j) X[j, :], w(i,j) = W[i*ldw + j] (trans 0) or W[j*ldw + i] (trans 1); i < n_out, j < n_in: the
 *   backward products of the logits GEMM (d img = dlogits @ txt, d txt = dlogits^T @ img) for any global batch (n x n x d, tiny). */
int hs_scale_dev(const float* x, float* y, const float* alpha_dev, float alpha, int64_t n, void* stream);
int hs_rowmix(const float* W, const float* X, float* out, int n_out, int n_in, int d, int64_t ldw, int64_t ldx, int64_t ldo, int trans,
              void* stream);
int hs_dot(const float* a, const float* b, float* out, float alpha, int64_t n, void* stream);
int hs_relation_mse(const float* l_ref, const float* l, float* dl, float* loss, int64_t n, float weight, int accumulate, void* stream);
int hs_ce_rows(const float* logits, float* dlogits, float* loss, int rows, int cols, int label0, float weight, int accumulate,
               void* stream);

/* diagnostic: selects a kernel variant per class for tools/kbench A/B runs (every key defaults to 0 = the shipped kernel).
 * The PRODUCT library honours only the selector whose variants compute the same result (key 0: 1 = the exact running-max
 * attention forward alone instead of the optimistic kernel + fix-up pass); every other (key, value != 0) is refused with a
 * one-time warning on stderr: the timing-only / truncating variants and the s_memtime stamps below are compiled only into
 * -DHS_DEBUG_HOOKS builds (tools/build_variant.sh).  Process-global, not thread-safe. */
void hs_debug_set(int key, int value);
/* diagnostic (-DHS_DEBUG_HOOKS builds: hs_debug_set(3, 4 | 7) / (3, 22), tools/kbench only; all zeros in the product library):
 * s_memtime phase sums of the bf16 GEMM's blocks (8 counters;
 * reset != 0 clears them) / the per-block timeline {tile start, main loop end, epilogue end, HW_ID | XCC_ID << 32} of the first
 * min(nblocks, 4096) blocks of the last launch */
void hs_debug_gemm_stamps(unsigned long long* out, int reset);
void hs_debug_gemm_trace(unsigned long long* out, int nblocks);

#ifdef __cplusplus
}
#endif
#endif /* HSENET_HIP_H */
