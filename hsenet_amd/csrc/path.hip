// Fused drivers for the parts of the hot path either side of the block stack (host-side sequencing of the kernels in
// gemm.hip / small.hip / norm.hip, plus a few HBM-bound glue kernels defined here):
//
//   hs_embed_fwd/bwd   P1 MONAI PatchEmbeddingBlock (Linear over the patchified volume + learned positions) and the token
//                      assembly of ViT_stage1 (vit.py:455-461); for ViT_stage2 also P5/P6, the slice-guided scorer
//                      (regular_attention vit.py:36-64 + patch_score_proj + Sigmoid + re-weighting, vit.py:325-349).
//   hs_packer_fwd/bwd  P2 VisualPacker_3d_phi_v3 (spatial_pooling_projector.py:121-153) incl. resolution_attention_v3 (:48-83).
//   hs_mse_grad, hs_sumsq, hs_adamw_masked   the train-step glue around them (loss gradient, global-norm clip, AdamW with a
//                      per-parameter decay mask: HF Trainer's adamw_torch + max_grad_norm, train/train_CLIP_stage1.py:85).
//
// Precision follows the reference's autocast policy exactly as the block driver does: Linear operands bf16 with fp32
// accumulation, everything else (softmax, LayerNorm, sigmoid, pooling, residuals, all parameter gradients) fp32.
#include "driver.h"
#include <cstdlib>
#include <vector>

using namespace hsdrv;

// ---------------------------------------------------------------------------------------------------------------------
// glue kernels
// ---------------------------------------------------------------------------------------------------------------------
// dst[b*dst_bstride + r*d + :] = src[b*src_bstride + r*d + :] for r < R  (fp32 and/or bf16 copies): strided row blocks, e.g. the
// packer reading rows 1.. of [B][N][d]
template <bool F16>
__global__ __launch_bounds__(256) void rows_gather_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          bf16_t* __restrict__ dstb, long src_bstride, long dst_bstride, int R, int d,
                                                          long total) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int d4 = d >> 2;
    const int ch = (int)(g % d4);
    const long row = g / d4;
    const int r = (int)(row % R);
    const long b = row / R;
    const f32x4 v = *(const f32x4*)(src + b * src_bstride + (long)r * d + ch * 4);
    const long o = b * dst_bstride + (long)r * d + ch * 4;
    if (dst) *(f32x4*)(dst + o) = v;
    if (dstb) {
        u32x2 pk;
        pk[0] = pack2h<F16>(v[0], v[1]); pk[1] = pack2h<F16>(v[2], v[3]);
        *(u32x2*)(dstb + o) = pk;
    }
}
// dst is [B][Rt][d]: row r0 + r = src[(b*R + r)][:] for r < R, every other row is zeroed (a packer gives the cls row no
// gradient; the CLIP head gives ONLY the cls row one)
__global__ __launch_bounds__(256) void rows_scatter_kernel(const float* __restrict__ src, float* __restrict__ dst, int r0, int R, int Rt,
                                                           int d, long total) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int d4 = d >> 2;
    const int ch = (int)(g % d4);
    const long row = g / d4;
    const int r = (int)(row % Rt);
    const long b = row / Rt;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r >= r0 && r < r0 + R) v = *(const f32x4*)(src + (b * R + (r - r0)) * d + ch * 4);
    *(f32x4*)(dst + row * d + ch * 4) = v;
}
static int rows_gather(const float* src, float* dst, void* dstb, long src_bstride, int B, int R, int d, void* stream, long dst_bstride = -1,
                       int f16 = 0) {
    const long total = (long)B * R * (d / 4);
    ProfScope prof(HS_PROF_OTHER, stream, 0, (double)B * R * d * (4 + (dst ? 4 : 0) + (dstb ? 2 : 0)));
    if (f16) hipLaunchKernelGGL(rows_gather_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                                (bf16_t*)dstb, src_bstride, dst_bstride < 0 ? (long)R * d : dst_bstride, R, d, total);
    else hipLaunchKernelGGL(rows_gather_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                            (bf16_t*)dstb, src_bstride, dst_bstride < 0 ? (long)R * d : dst_bstride, R, d, total);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
static int rows_scatter(const float* src, float* dst, int B, int r0, int R, int Rt, int d, void* stream) {
    const long total = (long)B * Rt * (d / 4);
    ProfScope prof(HS_PROF_OTHER, stream, 0, (double)B * (Rt + R) * d * 4);
    hipLaunchKernelGGL(rows_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, r0, R, Rt, d,
                       total);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
extern "C" int hs_rows_copy(const float* src, int64_t src_batch_stride, float* dst_f32, void* dst_bf16, int64_t dst_batch_stride, int B,
                            int R, int d, void* stream) {
    if (!src || (!dst_f32 && !dst_bf16)) return HS_ERR_ARG;
    if (B < 1 || R < 1 || d < 4 || (d & 3) || (src_batch_stride & 3) || (dst_batch_stride & 3) || dst_batch_stride < (int64_t)R * d) return HS_ERR_SHAPE;
    if (((uintptr_t)src | (uintptr_t)dst_f32) & 15 || ((uintptr_t)dst_bf16 & 7)) return HS_ERR_SHAPE;
    return rows_gather(src, dst_f32, dst_bf16, (long)src_batch_stride, B, R, d, stream, (long)dst_batch_stride);
}
extern "C" int hs_rows_scatter(const float* src, float* dst, int B, int r0, int R, int Rt, int d, void* stream) {
    if (!src || !dst) return HS_ERR_ARG;
    if (B < 1 || R < 1 || r0 < 0 || r0 + R > Rt || d < 4 || (d & 3) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return HS_ERR_SHAPE;
    return rows_scatter(src, dst, B, r0, R, Rt, d, stream);
}

// fp32 -> the driver's 16-bit operand format (bf16, or IEEE fp16 for a descriptor with prec = HS_PREC_F16)
static int cast16(const float* x, void* y, int64_t n, void* stream, int f16) { return f16 ? hs_cast_f16(x, y, n, stream) : hs_cast_bf16(x, y, n, stream); }
static int prec_check(int prec) { return prec == HS_PREC_BF16 || prec == HS_PREC_F16 ? HS_OK : HS_ERR_ARG; }

// ---------------------------------------------------------------------------------------------------------------------
// embedding driver
// ---------------------------------------------------------------------------------------------------------------------
namespace {

enum { E_CLS = 0, E_POS, E_WPE, E_BPE, E_WS, E_BS, E_WQ, E_BQ, E_WK, E_BK, E_WV, E_BV, E_WO, E_BO, E_NW, E_NB, E_COUNT };

struct EmbedLayout {
    size_t w_pe, w_q, w_k, w_v, w_o;                                       // bf16 weight copies
    size_t pe, xe, xe_b, q, sem_b, k, v, p, pd, ao, ao_b, xo, ln, stats, score;  // forward (kept for backward)
    size_t dpe, dpe_b, dscore, dln, dsum, dsum_b, dxo, dxo_b, dao, dq2, dk, dv, pws, dq_b, dk_b, dv_b, dpe2;   // backward scratch
    size_t total;
};

EmbedLayout embed_layout(const hs_embed_desc& c) {
    EmbedLayout L = {};
    Bump b;
    const size_t M = (size_t)c.B * c.Np, d = c.d, Ms = (size_t)c.B * c.ns;
    const bool sc = c.scorer != 0, bw = c.save_for_bwd != 0;
    L.w_pe = b.take(d * c.Pd * 2);
    L.pe = b.take(M * d * 4);
    if (sc) {
        L.w_q = b.take(d * d * 2); L.w_k = b.take(d * d * 2); L.w_v = b.take(d * d * 2); L.w_o = b.take(d * d * 2);
        L.xe = b.take(M * d * 4); L.xe_b = b.take(M * d * 2); L.q = b.take(M * d * 4);
        L.sem_b = b.take(Ms * d * 2); L.k = b.take(Ms * d * 4); L.v = b.take(Ms * d * 4);
        L.p = b.take(M * c.ns * 4); L.pd = b.take(c.drop_p > 0.f ? M * c.ns * 4 : 0);
        L.ao = b.take(M * d * 4); L.ao_b = b.take(M * d * 2); L.xo = b.take(M * d * 4); L.ln = b.take(M * d * 4);
        L.stats = b.take(2 * M * 4); L.score = b.take(M * 4);
    }
    if (bw) {
        L.dpe = b.take(M * d * 4); L.dpe_b = b.take(M * d * 2);
        if (sc) {
            L.dscore = b.take(M * 4); L.dln = b.take(M * d * 4); L.dsum = b.take(M * d * 4); L.dsum_b = b.take(M * d * 2);
            L.dxo = b.take(c.drop_p > 0.f ? M * d * 4 : 0); L.dxo_b = b.take(c.drop_p > 0.f ? M * d * 2 : 0);
            L.dao = b.take(M * d * 4); L.dq2 = b.take(M * d * 4); L.dk = b.take(Ms * d * 4); L.dv = b.take(Ms * d * 4);
            L.pws = b.take(M * c.ns * 4); L.dq_b = b.take(M * d * 2); L.dk_b = b.take(Ms * d * 2); L.dv_b = b.take(Ms * d * 2);
            L.dpe2 = b.take(M * d * 4);
        }
    }
    L.total = b.off;
    return L;
}

int embed_check(const hs_embed_desc* c) {
    if (!c) return HS_ERR_ARG;
    if (c->struct_size != sizeof(hs_embed_desc)) return HS_ERR_ARG;   // a caller compiled against another header (ABI 2)
    if (c->B < 1 || c->Np < 1 || c->Pd < 8 || (c->Pd & 7) || c->d < 8 || (c->d & 7) || c->d > 1024) return HS_ERR_SHAPE;
    if (!(c->drop_p >= 0.f && c->drop_p < 1.f)) return HS_ERR_ARG;
    RC(prec_check(c->prec));
    if (c->scorer) {
        if (c->ns < 4 || c->ns > 64 || (c->ns & 3) || (c->Np & 3)) return HS_ERR_SHAPE;     // hs_cross_attn_* limits
    }
    return HS_OK;
}

}  // namespace

extern "C" size_t hs_embed_workspace_bytes(const hs_embed_desc* desc) {
    if (embed_check(desc) != HS_OK) return 0;
    return embed_layout(*desc).total;
}

extern "C" int hs_embed_fwd(const hs_embed_desc* desc, const void* patches, const float* image_2d, const float* const* ep,
                            float* tokens, void* ws, size_t ws_bytes, void* stream) {
    RC(embed_check(desc));
    const hs_embed_desc& c = *desc;
    if (!patches || !ep || !tokens || !ws || (c.scorer && !image_2d)) return HS_ERR_ARG;
    if (!ep[E_POS] || !ep[E_WPE] || !ep[E_BPE] || (c.has_cls && !ep[E_CLS])) return HS_ERR_ARG;
    const EmbedLayout L = embed_layout(c);
    if (ws_bytes < L.total || ((uintptr_t)ws & 255)) return HS_ERR_SHAPE;
    char* w = (char*)ws;
    const int M = c.B * c.Np, d = c.d, Ms = c.B * c.ns;
    const size_t Md = (size_t)M * d;
    const int P = c.prec, f16 = P == HS_PREC_F16;       // 16-bit operand format of every GEMM / cast of this call
    {
        std::vector<const float*> src = {ep[E_WPE]};
        std::vector<void*> dst = {w + L.w_pe};
        std::vector<int64_t> cnt = {(int64_t)d * c.Pd};
        if (c.scorer) {
            const int idx[4] = {E_WQ, E_WK, E_WV, E_WO};
            const size_t off[4] = {L.w_q, L.w_k, L.w_v, L.w_o};
            for (int i = 0; i < 4; ++i) {
                if (!ep[idx[i]] || !ep[idx[i] + 1]) return HS_ERR_ARG;
                src.push_back(ep[idx[i]]); dst.push_back(w + off[i]); cnt.push_back((int64_t)d * d);
            }
            if (!ep[E_WS] || !ep[E_BS] || !ep[E_NW] || !ep[E_NB]) return HS_ERR_ARG;
        }
        RC(hs_cast_bf16_multi(src.data(), dst.data(), cnt.data(), (int)src.size(), stream, f16));
    }
    float* pe = (float*)(w + L.pe);
    RC(gemm(P, HS_GEMM_NT, patches, w + L.w_pe, M, d, c.Pd, c.Pd, c.Pd, pe, nullptr, d, ep[E_BPE], nullptr, nullptr, nullptr, HS_EPI_BIAS, 1,
            stream));
    const float* cls = c.has_cls ? ep[E_CLS] : nullptr;
    if (!c.scorer) return hs_assemble_tokens(pe, ep[E_POS], cls, nullptr, tokens, nullptr, c.B, c.Np, d, stream);

    float *xe = (float*)(w + L.xe), *q = (float*)(w + L.q), *k = (float*)(w + L.k), *v = (float*)(w + L.v);
    float *p = (float*)(w + L.p), *pd = c.drop_p > 0.f ? (float*)(w + L.pd) : nullptr;
    float *ao = (float*)(w + L.ao), *xo = (float*)(w + L.xo), *ln = (float*)(w + L.ln), *st = (float*)(w + L.stats);
    float* score = (float*)(w + L.score);
    RC(hs_assemble_tokens(pe, ep[E_POS], nullptr, nullptr, xe, nullptr, c.B, c.Np, d, stream));          // x = embedding + pos (vit.py:325)
    RC(cast16(xe, w + L.xe_b, (int64_t)Md, stream, f16));
    RC(gemm(P, HS_GEMM_NT, w + L.xe_b, w + L.w_q, M, d, d, d, d, q, nullptr, d, ep[E_BQ], nullptr, nullptr, nullptr, HS_EPI_BIAS, 1, stream));
    RC(cast16(image_2d, w + L.sem_b, (int64_t)Ms * d, stream, f16));                                    // image_2d.view(B, 32, -1) (:332)
    RC(gemm(P, HS_GEMM_NT, w + L.sem_b, w + L.w_k, Ms, d, d, d, d, k, nullptr, d, ep[E_BK], nullptr, nullptr, nullptr, HS_EPI_BIAS, 1, stream));
    RC(gemm(P, HS_GEMM_NT, w + L.sem_b, w + L.w_v, Ms, d, d, d, d, v, nullptr, d, ep[E_BV], nullptr, nullptr, nullptr, HS_EPI_BIAS, 1, stream));
    RC(hs_cross_attn_fwd(q, k, v, ao, p, pd, c.B, c.Np, c.ns, d, c.drop_p, c.seed[0], c.offset[0], stream));
    RC(cast16(ao, w + L.ao_b, (int64_t)Md, stream, f16));
    RC(gemm(P, HS_GEMM_NT, w + L.ao_b, w + L.w_o, M, d, d, d, d, xo, nullptr, d, ep[E_BO], nullptr, nullptr, nullptr, HS_EPI_BIAS, 1, stream));
    if (c.drop_p > 0.f) RC(hs_dropout(xo, xo, (int64_t)Md, c.drop_p, c.seed[1], c.offset[1], stream));
    RC(hs_layernorm_fwd(q, xo, ep[E_NW], ep[E_NB], ln, nullptr, st, st + M, M, d, 1e-5f, stream));      // norm(query + dropout(x)) (:62)
    RC(hs_rowdot_sigmoid_fwd(ln, ep[E_WS], ep[E_BS], score, M, d, stream));                              // :338-339
    return hs_assemble_tokens(pe, ep[E_POS], cls, score, tokens, nullptr, c.B, c.Np, d, stream);         // :345-349
}

extern "C" int hs_embed_bwd(const hs_embed_desc* desc, const void* patches, const float* dtokens, const float* const* ep,
                            float* const* eg, void* ws, size_t ws_bytes, void* stream) {
    RC(embed_check(desc));
    const hs_embed_desc& c = *desc;
    if (!c.save_for_bwd) return HS_ERR_ARG;
    if (!patches || !dtokens || !ep || !eg || !ws) return HS_ERR_ARG;
    if (!eg[E_POS] || !eg[E_WPE] || !eg[E_BPE] || (c.has_cls && !eg[E_CLS])) return HS_ERR_ARG;
    const EmbedLayout L = embed_layout(c);
    if (ws_bytes < L.total || ((uintptr_t)ws & 255)) return HS_ERR_SHAPE;
    char* w = (char*)ws;
    const int M = c.B * c.Np, d = c.d, Ms = c.B * c.ns;
    const size_t Md = (size_t)M * d;
    const int P = c.prec, f16 = P == HS_PREC_F16;       // 16-bit operand format of every GEMM / cast of this call
    float* dpe = (float*)(w + L.dpe);
    float* dcls = c.has_cls ? eg[E_CLS] : nullptr;
    const float* dpe_tot = dpe;
    if (!c.scorer) {
        RC(hs_assemble_tokens_bwd(dtokens, nullptr, nullptr, dpe, eg[E_POS], dcls, nullptr, c.B, c.Np, d, stream));
    } else {
        for (int i = E_WS; i < E_COUNT; ++i) if (!eg[i]) return HS_ERR_ARG;
        const float *xe = (const float*)(w + L.xe), *q = (const float*)(w + L.q), *k = (const float*)(w + L.k), *v = (const float*)(w + L.v);
        const float *p = (const float*)(w + L.p), *pd = c.drop_p > 0.f ? (const float*)(w + L.pd) : nullptr;
        const float *xo = (const float*)(w + L.xo), *ln = (const float*)(w + L.ln), *st = (const float*)(w + L.stats);
        const float* score = (const float*)(w + L.score);
        float *dscore = (float*)(w + L.dscore), *dln = (float*)(w + L.dln), *dsum = (float*)(w + L.dsum), *dao = (float*)(w + L.dao);
        float *dq2 = (float*)(w + L.dq2), *dk = (float*)(w + L.dk), *dv = (float*)(w + L.dv), *pws = (float*)(w + L.pws);
        float* dpe2 = (float*)(w + L.dpe2);
        // tokens = cat(cls, x * score): dx (token path), dscore; the position gradient is taken from the TOTAL dx below
        RC(hs_assemble_tokens_bwd(dtokens, xe, score, dpe, nullptr, dcls, dscore, c.B, c.Np, d, stream));
        RC(hs_rowdot_sigmoid_bwd(ln, ep[E_WS], score, dscore, dln, eg[E_WS], eg[E_BS], M, d, stream));
        const bool drop = c.drop_p > 0.f;
        RC(hs_layernorm_bwd(dln, f16 ? HS_LN_OUT_F16 : 0, q, xo, ep[E_NW], st, st + M, nullptr, dsum, w + L.dsum_b, eg[E_NW], eg[E_NB], drop ? nullptr : eg[E_BO],
                            M, d, stream));
        const void* dxo_b = w + L.dsum_b;
        if (drop) {                         // d(dropout_2(x)) : same Philox mask as the forward
            float* dxo = (float*)(w + L.dxo);
            RC(hs_dropout(dsum, dxo, (int64_t)Md, c.drop_p, c.seed[1], c.offset[1], stream));
            RC(hs_addcast_colsum(dxo, nullptr, w + L.dxo_b, eg[E_BO], nullptr, M, d, d, f16, stream));       // 16-bit copy + bias gradient
            dxo_b = w + L.dxo_b;
        }
        RC(wgrad(dxo_b, d, w + L.ao_b, d, eg[E_WO], d, d, M, stream, P));
        RC(gemm(P, HS_GEMM_NN, dxo_b, w + L.w_o, M, d, d, d, d, dao, nullptr, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
        RC(hs_cross_attn_bwd(q, k, v, p, pd, dao, dq2, dk, dv, pws, c.B, c.Np, c.ns, d, c.drop_p, c.seed[0], c.offset[0], stream));
        RC(hs_addcast_colsum(dsum, dq2, w + L.dq_b, eg[E_BQ], nullptr, M, d, d, f16, stream));   // dq = residual branch + attention branch; + its column sums
        RC(wgrad(w + L.dq_b, d, w + L.xe_b, d, eg[E_WQ], d, d, M, stream, P));
        // dx total = token path + Query path
        RC(gemm(P, HS_GEMM_NN, w + L.dq_b, w + L.w_q, M, d, d, d, d, dpe2, nullptr, d, nullptr, dpe, nullptr, nullptr, HS_EPI_RESID, 1, stream));
        dpe_tot = dpe2;
        RC(hs_assemble_tokens_bwd(dpe2, nullptr, nullptr, nullptr, eg[E_POS], nullptr, nullptr, c.B, c.Np, d, stream));   // dpos += sum_b dx
        RC(hs_addcast_colsum(dk, nullptr, w + L.dk_b, eg[E_BK], nullptr, Ms, d, d, f16, stream));
        RC(hs_addcast_colsum(dv, nullptr, w + L.dv_b, eg[E_BV], nullptr, Ms, d, d, f16, stream));
        RC(wgrad(w + L.dk_b, d, w + L.sem_b, d, eg[E_WK], d, d, Ms, stream, P));
        RC(wgrad(w + L.dv_b, d, w + L.sem_b, d, eg[E_WV], d, d, Ms, stream, P));
    }
    RC(hs_addcast_colsum(dpe_tot, nullptr, w + L.dpe_b, eg[E_BPE], nullptr, M, d, d, f16, stream));
    return wgrad(w + L.dpe_b, d, patches, c.Pd, eg[E_WPE], d, c.Pd, M, stream, P);
}

// ---------------------------------------------------------------------------------------------------------------------
// Spatial Packer driver
// ---------------------------------------------------------------------------------------------------------------------
namespace {

enum { K_W1 = 0, K_B1, K_W2, K_B2, K_WQ, K_BQ, K_WK, K_BK, K_WV, K_BV, K_WO, K_BO, K_NW, K_NB, K_COUNT };

struct PackLayout {
    size_t w1, w2, wq, wkv, wo, bkv;
    size_t hr, hr_b, lr, lr_b, q, kv, xo, xo_b, xl, fe_b, stats, a, gz;
    size_t dout_b, dz, dfe, dsum, dsum_b, dxl, dxl_b, dxo, dq2, dkv, dkv_b, dq_b, dlr, dhr;
    size_t total;
};

PackLayout pack_layout(const hs_packer_desc& c) {
    PackLayout L = {};
    Bump b;
    const size_t d = c.d, od = c.out_dim;
    const size_t T = (size_t)c.gd * c.gh * c.gw, cells = (size_t)(c.gd / c.kd) * (c.gh / c.kh) * (c.gw / c.kw);
    const size_t Mt = (size_t)c.B * T, Mc = (size_t)c.B * cells;
    const bool drop = c.drop_p > 0.f;
    L.w1 = b.take(od * d * 2); L.w2 = b.take(od * od * 2); L.wq = b.take(d * d * 2); L.wkv = b.take(2 * d * d * 2); L.wo = b.take(d * d * 2);
    L.bkv = b.take(2 * d * 4);
    L.hr = b.take(Mt * d * 4); L.hr_b = b.take(Mt * d * 2); L.lr = b.take(Mc * d * 4); L.lr_b = b.take(Mc * d * 2);
    L.q = b.take(Mc * d * 4); L.kv = b.take(Mt * 2 * d * 4); L.xo = b.take(Mc * d * 4); L.xo_b = b.take(Mc * d * 2);
    L.xl = b.take(Mc * d * 4); L.fe_b = b.take(Mc * d * 2); L.stats = b.take(2 * Mc * 4);
    L.a = b.take(Mc * od * 2); L.gz = b.take(Mc * od * 2);
    if (c.save_for_bwd) {
        L.dout_b = b.take(Mc * od * 2); L.dz = b.take(Mc * od * 2); L.dfe = b.take(Mc * d * 2);
        L.dsum = b.take(Mc * d * 4); L.dsum_b = b.take(Mc * d * 2);
        L.dxl = b.take(drop ? Mc * d * 4 : 0); L.dxl_b = b.take(drop ? Mc * d * 2 : 0);
        L.dxo = b.take(Mc * d * 4); L.dq2 = b.take(Mc * d * 4); L.dkv = b.take(Mt * 2 * d * 4); L.dkv_b = b.take(Mt * 2 * d * 2);
        L.dq_b = b.take(Mc * d * 2); L.dlr = b.take(Mc * d * 4); L.dhr = b.take(Mt * d * 4);
    }
    L.total = b.off;
    return L;
}

int pack_check(const hs_packer_desc* c) {
    if (!c) return HS_ERR_ARG;
    if (c->struct_size != sizeof(hs_packer_desc)) return HS_ERR_ARG;  // a caller compiled against another header (ABI 2)
    if (c->B < 1 || c->gd < 1 || c->gh < 1 || c->gw < 1 || c->kd < 1 || c->kh < 1 || c->kw < 1) return HS_ERR_SHAPE;
    if (c->kd > c->gd || c->kh > c->gh || c->kw > c->gw || c->kd * c->kh * c->kw > 64) return HS_ERR_SHAPE;
    if (c->d < 8 || (c->d & 7) || c->d > 1024 || c->out_dim < 8 || (c->out_dim & 7)) return HS_ERR_SHAPE;
    if (c->in_row_offset < 0 || c->in_batch_stride < (int64_t)(c->in_row_offset + c->gd * c->gh * c->gw) * c->d) return HS_ERR_SHAPE;
    const int64_t cells = (int64_t)(c->gd / c->kd) * (c->gh / c->kh) * (c->gw / c->kw);
    if (c->out_batch_stride < cells * c->out_dim || (c->out_batch_stride & 7)) return HS_ERR_SHAPE;
    if (!(c->drop_p >= 0.f && c->drop_p < 1.f)) return HS_ERR_ARG;
    RC(prec_check(c->prec));
    return HS_OK;
}

}  // namespace

extern "C" size_t hs_packer_workspace_bytes(const hs_packer_desc* desc) {
    if (pack_check(desc) != HS_OK) return 0;
    return pack_layout(*desc).total;
}

extern "C" int hs_packer_fwd(const hs_packer_desc* desc, const float* feat, const float* const* pp, float* out, void* ws,
                             size_t ws_bytes, void* stream) {
    RC(pack_check(desc));
    const hs_packer_desc& c = *desc;
    if (!feat || !pp || !out || !ws) return HS_ERR_ARG;
    for (int i = 0; i < K_COUNT; ++i) if (!pp[i]) return HS_ERR_ARG;
    const PackLayout L = pack_layout(c);
    if (ws_bytes < L.total || ((uintptr_t)ws & 255)) return HS_ERR_SHAPE;
    char* w = (char*)ws;
    const int d = c.d, od = c.out_dim;
    const int T = c.gd * c.gh * c.gw, cells = (c.gd / c.kd) * (c.gh / c.kh) * (c.gw / c.kw);
    const int Mt = c.B * T, Mc = c.B * cells;
    const int P = c.prec, f16 = P == HS_PREC_F16;       // 16-bit operand format of every GEMM / cast of this call
    {   // bf16 weight copies; [Wk; Wv] adjacent so that K and V of every HR token come out of ONE GEMM (59 % of the packer's FLOPs)
        const float* src[6] = {pp[K_W1], pp[K_W2], pp[K_WQ], pp[K_WK], pp[K_WV], pp[K_WO]};
        void* dst[6] = {w + L.w1, w + L.w2, w + L.wq, w + L.wkv, w + L.wkv + (size_t)d * d * 2, w + L.wo};
        const int64_t cnt[6] = {(int64_t)od * d, (int64_t)od * od, (int64_t)d * d, (int64_t)d * d, (int64_t)d * d, (int64_t)d * d};
        RC(hs_cast_bf16_multi(src, dst, cnt, 6, stream, f16));
        float* bkv = (float*)(w + L.bkv);
        RC(hs_scale(pp[K_BK], bkv, 1.0f, d, stream));
        RC(hs_scale(pp[K_BV], bkv + d, 1.0f, d, stream));
    }
    float *hr = (float*)(w + L.hr), *lr = (float*)(w + L.lr), *q = (float*)(w + L.q), *kv = (float*)(w + L.kv);
    float *xo = (float*)(w + L.xo), *xl = (float*)(w + L.xl), *st = (float*)(w + L.stats);
    // visual_inputs.view(B, gd, gh, gw, 768) (:140); the 'patch' feature selection (vit.py:934-936) = in_row_offset 1
    RC(rows_gather(feat + (size_t)c.in_row_offset * d, hr, w + L.hr_b, (long)c.in_batch_stride, c.B, T, d, stream, -1, f16));
    RC(hs_pool3d_fwd(hr, lr, c.B, c.gd, c.gh, c.gw, c.kd, c.kh, c.kw, d, stream));                          // :141 avg_pool3d
    RC(cast16(lr, w + L.lr_b, (int64_t)Mc * d, stream, f16));
    RC(gemm(P, HS_GEMM_NT, w + L.lr_b, w + L.wq, Mc, d, d, d, d, q, nullptr, d, pp[K_BQ], nullptr, nullptr, nullptr, HS_EPI_BIAS, 1, stream));
    RC(gemm(P, HS_GEMM_NT, w + L.hr_b, w + L.wkv, Mt, 2 * d, d, d, d, kv, nullptr, 2 * d, (const float*)(w + L.bkv), nullptr, nullptr, nullptr,
            HS_EPI_BIAS, 1, stream));
    RC(hs_window_attn_fwd(q, kv, xo, c.B, c.gd, c.gh, c.gw, c.kd, c.kh, c.kw, d, c.drop_p, c.seed[0], c.offset[0], stream));
    RC(cast16(xo, w + L.xo_b, (int64_t)Mc * d, stream, f16));
    RC(gemm(P, HS_GEMM_NT, w + L.xo_b, w + L.wo, Mc, d, d, d, d, xl, nullptr, d, pp[K_BO], nullptr, nullptr, nullptr, HS_EPI_BIAS, 1, stream));
    if (c.drop_p > 0.f) RC(hs_dropout(xl, xl, (int64_t)Mc * d, c.drop_p, c.seed[1], c.offset[1], stream));
    RC(hs_layernorm_fwd16(q, xl, pp[K_NW], pp[K_NB], nullptr, w + L.fe_b, st, st + Mc, Mc, d, 1e-5f, f16, stream));   // :82
    // proj_mpls: Linear -> GELU -> Linear (:127-131,144)
    RC(gemm(P, HS_GEMM_NT, w + L.fe_b, w + L.w1, Mc, od, d, d, d, nullptr, w + L.a, od, pp[K_B1], nullptr, nullptr,
            c.save_for_bwd ? w + L.gz : nullptr, HS_EPI_BIAS | HS_EPI_GELU | HS_EPI_SAVE_DGELU, 1, stream));
    if (c.out_batch_stride == (int64_t)cells * od)
        return gemm(P, HS_GEMM_NT, w + L.a, w + L.w2, Mc, od, od, od, od, out, nullptr, od, pp[K_B2], nullptr, nullptr, nullptr, HS_EPI_BIAS, 1,
                    stream);
    // two packers share one [B][2*cells][out] tensor (lamed_arch.py:127-132's torch.cat): one GEMM per batch entry via grid.z
    return gemm(P, HS_GEMM_NT, w + L.a, w + L.w2, cells, od, od, od, od, out, nullptr, od, pp[K_B2], nullptr, nullptr, nullptr, HS_EPI_BIAS, 1,
                stream, nullptr, 0, 1.0f, c.B, (long)cells * od, 0, (long)c.out_batch_stride);
}

extern "C" int hs_packer_bwd(const hs_packer_desc* desc, const float* dout, const float* const* pp, float* const* pg, float* dfeat,
                             void* ws, size_t ws_bytes, void* stream) {
    RC(pack_check(desc));
    const hs_packer_desc& c = *desc;
    if (!c.save_for_bwd) return HS_ERR_ARG;
    if (!dout || !pp || !pg || !ws) return HS_ERR_ARG;
    for (int i = 0; i < K_COUNT; ++i) if (!pp[i] || !pg[i]) return HS_ERR_ARG;
    const PackLayout L = pack_layout(c);
    if (ws_bytes < L.total || ((uintptr_t)ws & 255)) return HS_ERR_SHAPE;
    char* w = (char*)ws;
    const int d = c.d, od = c.out_dim;
    const int T = c.gd * c.gh * c.gw, cells = (c.gd / c.kd) * (c.gh / c.kh) * (c.gw / c.kw);
    const int Mt = c.B * T, Mc = c.B * cells;
    const int P = c.prec, f16 = P == HS_PREC_F16;       // 16-bit operand format of every GEMM / cast of this call
    const bool drop = c.drop_p > 0.f;
    // dfeat is written as a DENSE [B][in_row_offset + T][d] tensor: a padded input stride would put every batch entry after the
    // first at the wrong offset, so it is refused (before anything is launched) rather than scattered silently
    if (dfeat && c.in_batch_stride != (int64_t)(c.in_row_offset + T) * d) return HS_ERR_SHAPE;
    const float *q = (const float*)(w + L.q), *kv = (const float*)(w + L.kv), *xl = (const float*)(w + L.xl), *st = (const float*)(w + L.stats);
    float *dsum = (float*)(w + L.dsum), *dxo = (float*)(w + L.dxo), *dq2 = (float*)(w + L.dq2), *dkv = (float*)(w + L.dkv);
    float *dlr = (float*)(w + L.dlr), *dhr = (float*)(w + L.dhr);
    // ---- proj_mpls
    RC(rows_gather(dout, nullptr, w + L.dout_b, (long)c.out_batch_stride, c.B, cells, od, stream, -1, f16));
    RC(wgrad(w + L.dout_b, od, w + L.a, od, pg[K_W2], od, od, Mc, stream, P));
    RC(hs_colsum(w + L.dout_b, 1 + f16, pg[K_B2], Mc, od, od, stream));
    RC(gemm(P, HS_GEMM_NN, w + L.dout_b, w + L.w2, Mc, od, od, od, od, nullptr, w + L.dz, od, nullptr, nullptr, w + L.gz, nullptr,
            HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU, 1, stream, pg[K_B1]));
    RC(wgrad(w + L.dz, od, w + L.fe_b, d, pg[K_W1], od, d, Mc, stream, P));
    RC(gemm(P, HS_GEMM_NN, w + L.dz, w + L.w1, Mc, d, od, od, d, nullptr, w + L.dfe, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
    // ---- LN(q + dropout(output_linear(attn)))
    RC(hs_layernorm_bwd(w + L.dfe, f16 ? HS_LN_DY_F16 : HS_LN_DY_BF16, q, xl, pp[K_NW], st, st + Mc, nullptr, dsum, w + L.dsum_b, pg[K_NW], pg[K_NB], drop ? nullptr : pg[K_BO],
                        Mc, d, stream));
    const void* dxl_b = w + L.dsum_b;
    if (drop) {
        float* dxl = (float*)(w + L.dxl);
        RC(hs_dropout(dsum, dxl, (int64_t)Mc * d, c.drop_p, c.seed[1], c.offset[1], stream));
        RC(hs_addcast_colsum(dxl, nullptr, w + L.dxl_b, pg[K_BO], nullptr, Mc, d, d, f16, stream));
        dxl_b = w + L.dxl_b;
    }
    RC(wgrad(dxl_b, d, w + L.xo_b, d, pg[K_WO], d, d, Mc, stream, P));
    RC(gemm(P, HS_GEMM_NN, dxl_b, w + L.wo, Mc, d, d, d, d, dxo, nullptr, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
    // ---- window attention: tokens outside every window (grid not divisible by the kernel) get no gradient
    if ((c.gd % c.kd) || (c.gh % c.kh) || (c.gw % c.kw)) {
        if (hipMemsetAsync(dkv, 0, (size_t)Mt * 2 * d * 4, (hipStream_t)stream) != hipSuccess) return HS_ERR_LAUNCH;
    }
    RC(hs_window_attn_bwd(q, kv, dxo, dq2, dkv, c.B, c.gd, c.gh, c.gw, c.kd, c.kh, c.kw, d, c.drop_p, c.seed[0], c.offset[0], stream));
    RC(hs_addcast_colsum(dsum, dq2, w + L.dq_b, pg[K_BQ], nullptr, Mc, d, d, f16, stream));
    RC(wgrad(w + L.dq_b, d, w + L.lr_b, d, pg[K_WQ], d, d, Mc, stream, P));
    RC(gemm(P, HS_GEMM_NN, w + L.dq_b, w + L.wq, Mc, d, d, d, d, dlr, nullptr, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
    RC(hs_addcast_colsum(dkv, nullptr, w + L.dkv_b, pg[K_BK], pg[K_BV], Mt, 2 * d, d, f16, stream));   // [dk | dv] 16-bit + both bias gradients
    const char* dkv_b = w + L.dkv_b;
    RC(wgrad(dkv_b, 2 * d, w + L.hr_b, d, pg[K_WK], d, d, Mt, stream, P));
    RC(wgrad(dkv_b + (size_t)d * 2, 2 * d, w + L.hr_b, d, pg[K_WV], d, d, Mt, stream, P));
    if (!dfeat) return HS_OK;                 // input features not differentiated (frozen towers: lamed_arch.py:66)
    RC(gemm(P, HS_GEMM_NN, dkv_b, w + L.wkv, Mt, d, 2 * d, 2 * d, d, dhr, nullptr, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
    RC(hs_pool3d_bwd(dlr, dhr, c.B, c.gd, c.gh, c.gw, c.kd, c.kh, c.kw, d, 1, stream));
    return rows_scatter(dhr, dfeat, c.B, c.in_row_offset, T, c.in_row_offset + T, d, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// train-step glue
// ---------------------------------------------------------------------------------------------------------------------
// L = mean(y^2):  dy = y * 2/n ;  loss[0] += sum(y^2)/n
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ y, float* __restrict__ dy, float* __restrict__ loss, long n,
                                                       float inv_n, float gscale) {
    float acc = 0.f;
    const float k = 2.0f * inv_n * gscale;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const f32x4 v = *(const f32x4*)(y + i);
            *(f32x4*)(dy + i) = v * k;
            acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        } else {
            for (long j = i; j < n; ++j) { dy[j] = y[j] * k; acc += y[j] * y[j]; }
        }
    }
    // one atomic per BLOCK (a wave-level atomic per 256 elements put 3000 serialised atomics on one address for a (1, 256, 3072) output:
    // 40 us for a 3-MB pass)
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (loss && threadIdx.x == 0) atomicAdd(loss, ((red[0] + red[1]) + (red[2] + red[3])) * inv_n);
}
extern "C" int hs_mse_grad(const float* y, float* dy, float* loss, int64_t n, void* stream) { return hs_mse_grad_scaled(y, dy, loss, n, 1.0f, stream); }
// the same with dy multiplied by `grad_scale` (the loss scale of an fp16 backward pass; `loss` stays the unscaled mean square)
extern "C" int hs_mse_grad_scaled(const float* y, float* dy, float* loss, int64_t n, float grad_scale, void* stream) {
    if (!y || !dy) return HS_ERR_ARG;
    if (!(grad_scale > 0.f)) return HS_ERR_ARG;
    if (n < 1 || (((uintptr_t)y | (uintptr_t)dy) & 15)) return HS_ERR_SHAPE;
    long blocks = (n + 1023) / 1024;
    if (blocks > 512) blocks = 512;
    ProfScope prof(HS_PROF_OTHER, stream, 0, (double)n * 8);
    hipLaunchKernelGGL(mse_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, dy, loss, (long)n, 1.0f / (float)n, grad_scale);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// out[0] = sum(x^2), DETERMINISTIC (fixed grid, fixed reduction order, no atomics): the clip coefficient derived from it must be
// bit-identical on every data-parallel rank, or the replicas drift apart by an ulp per step.  scratch: SUMSQ_BLOCKS floats.
#define SUMSQ_BLOCKS 1024
__device__ __forceinline__ float block_sum_256(float acc, float* red) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, float* __restrict__ part, long n) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)SUMSQ_BLOCKS * 1024) {
        if (i + 4 <= n) {
            const f32x4 v = *(const f32x4*)(x + i);
            acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        } else {
            for (long j = i; j < n; ++j) acc += x[j] * x[j];
        }
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < SUMSQ_BLOCKS / 256; ++k) acc += part[threadIdx.x + 256 * k];
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[0] = acc;
}
extern "C" int hs_sumsq(const float* x, float* out, float* scratch, int64_t n, void* stream) {
    if (!x || !out || !scratch) return HS_ERR_ARG;
    if (n < 1 || ((uintptr_t)x & 15)) return HS_ERR_SHAPE;
    ProfScope prof(HS_PROF_OTHER, stream, 0, (double)n * 4);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, x, scratch, (long)n);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, out);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// torch.optim.AdamW over a flat buffer whose 64-element chunks carry a decay flag (HF Trainer decays every parameter except
// biases and LayerNorm weights), with the global-norm clip folded into the gradient scale (max_grad_norm, no host sync: the
// squared norm of the SUMMED gradient is read from device memory).
__global__ __launch_bounds__(256) void adamw_masked_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, const unsigned char* __restrict__ decay, long n, float lr,
                                                           float b1, float b2, float eps, float wd, float gs,
                                                           const float* __restrict__ gsq, float max_norm, float bc1, float bc2s) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float scale = gs;
    if (gsq) {     // clip_grad_norm_: coef = max_norm / (||g|| + 1e-6), applied when < 1
        // a non-finite gradient norm (an fp16 backward pass that overflowed its loss scale) skips the update, moments included -- what
        // torch.cuda.amp.GradScaler.step does; with finite gradients this branch is never taken
        if (!(gsq[0] < 3.0e38f)) return;
        const float nrm = sqrtf(gsq[0]) * gs;
        const float coef = max_norm / (nrm + 1e-6f);
        if (coef < 1.0f) scale *= coef;
    }
    const float wde = (decay && !decay[i >> 6]) ? 0.f : wd;
    const int cnt = (int)min((long)4, n - i);
    for (int e = 0; e < cnt; ++e) {
        const float gr = g[i + e] * scale;
        float pv = p[i + e] * (1.0f - lr * wde);
        const float mv = b1 * m[i + e] + (1.0f - b1) * gr;
        const float vv = b2 * v[i + e] + (1.0f - b2) * gr * gr;
        m[i + e] = mv;
        v[i + e] = vv;
        pv -= (lr / bc1) * mv / (sqrtf(vv) / bc2s + eps);
        p[i + e] = pv;
    }
}
extern "C" int hs_adamw_masked(float* p, const float* g, float* m, float* v, const unsigned char* decay_mask, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, float grad_scale, const float* grad_sumsq,
                               float max_grad_norm, int step, void* stream) {
    if (!p || !g || !m || !v) return HS_ERR_ARG;
    if (n < 1 || step < 1 || (decay_mask && (n & 63))) return HS_ERR_SHAPE;
    if (grad_sumsq && !(max_grad_norm > 0.f)) return HS_ERR_ARG;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    ProfScope prof(HS_PROF_ADAMW, stream, 0, (double)n * 28);
    hipLaunchKernelGGL(adamw_masked_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, decay_mask,
                       (long)n, lr, beta1, beta2, eps, weight_decay, grad_scale, grad_sumsq, max_grad_norm, bc1, bc2s);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// hs_adamw_fused: the same update as hs_adamw_masked as ONE sweep that also (a) zeroes the gradient behind the read, (b) writes
// the 16-bit copy of every updated weight chunk to the address its 64-element chunk is mapped to (the operand copies of the next
// forward: no cast pass over the fp32 masters), (c) keeps the step / skipped-step counters on the device.
// state (int32 x 8): [0] steps applied, [1] steps skipped, [2] this call skips; as float: [4] gradient scale incl. clip, [5] bc1, [6] bc2s
__global__ void adamw_prep_kernel(int32_t* __restrict__ st, const float* __restrict__ gsq, float gs, float max_norm, float b1, float b2) {
    if (threadIdx.x) return;
    float scale = gs;
    if (gsq) {
        const float q = gsq[0];
        if (!(q < 3.0e38f)) { st[1] += 1; st[2] = 1; return; }    // inf / NaN gradient norm: the step is skipped and does not count
        if (max_norm > 0.f) {     // clip_grad_norm_: coef = max_norm / (||g|| + 1e-6), applied when < 1
            const float coef = max_norm / (sqrtf(q) * gs + 1e-6f);
            if (coef < 1.0f) scale *= coef;
        }
    }
    const int step = st[0] + 1;
    st[0] = step; st[2] = 0;
    float* f = (float*)st;
    f[4] = scale;
    f[5] = 1.0f - powf(b1, (float)step);
    f[6] = sqrtf(1.0f - powf(b2, (float)step));
}
template <bool F16>
__global__ __launch_bounds__(256) void adamw_fused_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, const unsigned char* __restrict__ decay, long n, float lr,
                                                          float b1, float b2, float eps, float wd, int zero_grad,
                                                          const uint64_t* __restrict__ table, const int32_t* __restrict__ st) {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const bool skip = st[2] != 0;  // skipped step: parameters, moments and 16-bit copies stay; the overflowed gradient is discarded
    const float* sf = (const float*)st;
    const float scale = sf[4], lr1 = lr / sf[5], bc2s = sf[6];
    // grid-stride: the synchronous pass launches one block per 1024 elements (one trip); a sweep that runs BESIDE the next forward is
    // launched with a capped grid so that it leaves wave slots and HBM bandwidth to the forward's kernels (hs_adamw_fused)
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (skip) {
            if (zero_grad) *(f32x4*)(g + i) = z4;
            continue;
        }
        const float wde = (decay && !decay[i >> 6]) ? 0.f : wd;
        f32x4 P = *(const f32x4*)(p + i), M = *(const f32x4*)(m + i), V = *(const f32x4*)(v + i);
        const f32x4 G = *(const f32x4*)(g + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = G[e] * scale;
            float pv = P[e] * (1.0f - lr * wde);
            const float mv = b1 * M[e] + (1.0f - b1) * gr;
            const float vv = b2 * V[e] + (1.0f - b2) * gr * gr;
            M[e] = mv;
            V[e] = vv;
            pv -= lr1 * mv / (sqrtf(vv) / bc2s + eps);
            P[e] = pv;
        }
        *(f32x4*)(p + i) = P;
        *(f32x4*)(m + i) = M;
        *(f32x4*)(v + i) = V;
        if (zero_grad) *(f32x4*)(g + i) = z4;
        if (table) {
            const uint64_t dst = table[i >> 6];
            if (dst) {
                u32x2 pk;
                pk[0] = pack2h<F16>(P[0], P[1]); pk[1] = pack2h<F16>(P[2], P[3]);
                *(u32x2*)(uintptr_t)(dst + (uint64_t)((i & 63) * 2)) = pk;
            }
        }
    }
}
extern "C" int hs_adamw_fused(float* p, float* g, float* m, float* v, const unsigned char* decay_mask, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, float grad_scale, const float* grad_sumsq, float max_grad_norm,
                              int zero_grad, const uint64_t* w16_table, int w16_is_f16, int32_t* state, int phase, void* stream) {
    if (!state || phase < HS_ADAMW_STEP || phase > HS_ADAMW_SWEEP) return HS_ERR_ARG;
    if ((uintptr_t)state & 3) return HS_ERR_SHAPE;
    if (phase != HS_ADAMW_SWEEP) {           // the one-thread prologue: skip decision, step counter, clip coefficient, bias corrections
        if (max_grad_norm > 0.f && !grad_sumsq) return HS_ERR_ARG;
        if (!(grad_scale > 0.f)) return HS_ERR_ARG;
    }
    if (phase != HS_ADAMW_PREPARE) {
        if (!p || !g || !m || !v) return HS_ERR_ARG;
        if (n < 64 || (n & 63)) return HS_ERR_SHAPE;
        if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) || ((uintptr_t)w16_table & 7)) return HS_ERR_SHAPE;
    }
    if (phase != HS_ADAMW_SWEEP)
        hipLaunchKernelGGL(adamw_prep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, grad_sumsq, grad_scale,
                           max_grad_norm > 0.f ? max_grad_norm : 0.f, beta1, beta2);
    if (phase != HS_ADAMW_PREPARE) {
        ProfScope prof(HS_PROF_ADAMW, stream, 0, (double)n * (28 + (zero_grad ? 4 : 0)));
        // HS_ADAMW_SWEEP runs beside the next forward: a capped grid (default 2048 blocks = 8 per CU's worth of 256-thread blocks on
        // 256 CUs; HS_ADAMW_SWEEP_BLOCKS, read once) bounds the wave slots and the HBM bandwidth it takes from the forward's kernels
        static const long sweep_cap = getenv("HS_ADAMW_SWEEP_BLOCKS") ? atol(getenv("HS_ADAMW_SWEEP_BLOCKS")) : 2048;
        long nb = (n + 1023) / 1024;
        if (phase == HS_ADAMW_SWEEP && sweep_cap > 0 && nb > sweep_cap) nb = sweep_cap;
        const dim3 grid((unsigned)nb), blk(256);
        if (w16_is_f16) hipLaunchKernelGGL(adamw_fused_kernel<true>, grid, blk, 0, (hipStream_t)stream, p, g, m, v, decay_mask, (long)n, lr, beta1,
                                           beta2, eps, weight_decay, zero_grad, w16_table, (const int32_t*)state);
        else hipLaunchKernelGGL(adamw_fused_kernel<false>, grid, blk, 0, (hipStream_t)stream, p, g, m, v, decay_mask, (long)n, lr, beta1, beta2,
                                eps, weight_decay, zero_grad, w16_table, (const int32_t*)state);
    }
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// N1: CLIP loss-head glue with the logit scale kept ON THE DEVICE (it is a trained parameter: no host read per step)
// ---------------------------------------------------------------------------------------------------------------------
// y = x * alpha_dev[0] * alpha   (logits = logit_scale * img @ txt^T, CLIP_stage2.py:173-180; alpha_dev may be NULL)
__global__ __launch_bounds__(256) void scale_dev_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ ad,
                                                        float alpha, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[i] * (ad ? ad[0] * alpha : alpha);
}
extern "C" int hs_scale_dev(const float* x, float* y, const float* alpha_dev, float alpha, int64_t n, void* stream) {
    if (!x || !y) return HS_ERR_ARG;
    if (n < 1) return HS_ERR_SHAPE;
    hipLaunchKernelGGL(scale_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, alpha_dev, alpha, (long)n);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
// out[0] += alpha * sum(a * b)      (d loss / d logit_scale = sum(dlogits * raw))
__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, float alpha,
                                                  long n) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += a[i] * b[i];
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, alpha * (red[0] + red[1] + red[2] + red[3]));
}
extern "C" int hs_dot(const float* a, const float* b, float* out, float alpha, int64_t n, void* stream) {
    if (!a || !b || !out) return HS_ERR_ARG;
    if (n < 1) return HS_ERR_SHAPE;
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(dot_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, alpha, (long)n);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
// out[i, :] = sum_j w(i, j) * X[j, :]  with w(i, j) = W[i*ldw + j] (trans = 0) or W[j*ldw + i] (trans = 1): the two backward products
// of the logits GEMM (d img = dlogits @ txt, d txt = dlogits^T @ img) for ANY global batch n (the MFMA GEMM wants multiples of 4)
__global__ __launch_bounds__(256) void rowmix_kernel(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ out,
                                                     int n_in, int d, long ldw, long ldx, long ldo, int trans) {
    const int col = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (col >= d) return;
    float acc = 0.f;
    for (int j = 0; j < n_in; ++j) acc += (trans ? W[(long)j * ldw + i] : W[(long)i * ldw + j]) * X[(long)j * ldx + col];
    out[(long)i * ldo + col] = acc;
}
extern "C" int hs_rowmix(const float* W, const float* X, float* out, int n_out, int n_in, int d, int64_t ldw, int64_t ldx, int64_t ldo,
                         int trans, void* stream) {
    if (!W || !X || !out) return HS_ERR_ARG;
    if (n_out < 1 || n_in < 1 || d < 1 || n_out > 65535) return HS_ERR_SHAPE;
    hipLaunchKernelGGL(rowmix_kernel, dim3((d + 255) / 256, n_out), dim3(256), 0, (hipStream_t)stream, W, X, out, n_in, d, (long)ldw,
                       (long)ldx, (long)ldo, trans);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
// image_text_relation_regulation (CLIP_stage2.py:187-190) on one pair of logit matrices: loss[0] += mean((l - l_ref)^2);
// dl (+)= weight * 2/n * (l - l_ref)
__global__ __launch_bounds__(256) void relation_mse_kernel(const float* __restrict__ lref, const float* __restrict__ l, float* __restrict__ dl,
                                                           float* __restrict__ loss, long n, float weight, int accumulate) {
    float acc = 0.f;
    const float k = weight * 2.0f / (float)n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float df = l[i] - lref[i];
        acc += df * df;
        if (dl) dl[i] = (accumulate ? dl[i] : 0.f) + k * df;
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) / (float)n);
}
extern "C" int hs_relation_mse(const float* l_ref, const float* l, float* dl, float* loss, int64_t n, float weight, int accumulate,
                               void* stream) {
    if (!l_ref || !l) return HS_ERR_ARG;
    if (n < 1) return HS_ERR_SHAPE;
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(relation_mse_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, l_ref, l, dl, loss, (long)n, weight,
                       accumulate);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
// F.cross_entropy(logits [rows][cols], labels = label0 + row) (mean over rows), one wave per row -- the local_loss branch of the
// contrastive loss (CLIP_stage2.py:173-177: this rank's rows against the gathered features):
// loss[0] += weight * mean_r (lse_r - logit[r, label_r]);  dlogits (+)= weight * (softmax - onehot) / rows
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, float* __restrict__ dlogits, float* __restrict__ loss,
                                                      int rows, int cols, int label0, float weight, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* x = logits + (long)r * cols;
    float m = -__builtin_inff();
    for (int j = lane; j < cols; j += 64) m = fmaxf(m, x[j]);
    m = wave_max(m);
    float z = 0.f;
    for (int j = lane; j < cols; j += 64) z += __expf(x[j] - m);
    z = wave_sum(z);
    const int lab = label0 + r;
    const float k = weight / (float)rows;
    if (dlogits) {
        float* o = dlogits + (long)r * cols;
        for (int j = lane; j < cols; j += 64) {
            const float gq = k * (__expf(x[j] - m) / z - (j == lab ? 1.f : 0.f));
            o[j] = (accumulate ? o[j] : 0.f) + gq;
        }
    }
    if (lane == 0 && loss) atomicAdd(loss, k * (m + __logf(z) - x[lab]));
}
extern "C" int hs_ce_rows(const float* logits, float* dlogits, float* loss, int rows, int cols, int label0, float weight, int accumulate,
                          void* stream) {
    if (!logits) return HS_ERR_ARG;
    if (rows < 1 || cols < 1 || label0 < 0 || label0 + rows > cols) return HS_ERR_SHAPE;
    hipLaunchKernelGGL(ce_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, dlogits, loss, rows, cols, label0,
                       weight, accumulate);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
