// Fused encoder driver: the reference's hot loops `for blk in self.blocks: x = blk(x)` + `self.norm(x)`
// (multimodal_encoder/vit.py:464-467 and 352-355) enqueued as one kernel sequence on one stream, forward and
// backward.  Host-side only: it sequences the kernels of gemm.hip / attn.hip / norm.hip / small.hip.
//
// Per block (MONAI TransformerBlock, pre-LN):            kernels
//   h1 = LN1(x)                                          layernorm_fwd (fp32 in, bf16 out, stats saved)
//   qkv = h1 Wqkv^T                                      gemm NT -> bf16 [M][3d]
//   o = softmax(q k^T / 8) v                             attn_fwd (flash, lse saved)
//   xm = x + o Wo^T + bo                                 gemm NT, epilogue bias+residual -> fp32
//   h2 = LN2(xm)                                         layernorm_fwd
//   a = gelu(h2 W1^T + b1)                               gemm NT, epilogue bias+GELU -> bf16 (gelu'(z) saved)
//   x' = xm + a W2^T + b2                                gemm NT, epilogue bias+residual -> fp32 (= hidden[l])
// The residual stream (x, xm) is fp32 throughout, as under the reference's autocast policy.
#include "driver.h"
#include <cstdlib>
#include <vector>

using namespace hsdrv;

namespace {

struct Layout {
    size_t wbf, wbft, xmid, h1, h2, qkv, o, lse, z, a, stats, fstats, dx, dh, dxb, dob, dz, dqkv, delta, qb, total;
    size_t per_w;  // bf16 weight elements per layer
};

// The dgrads dX = dY W read W [out][in] with the reduction index as the memory row (NN layout: transposed LDS reads, the slower
// kernel).  Against a transposed bf16 copy W^T [in][out] -- re-made once per backward call, 14 MB per layer -- the same product is
// an NT GEMM on the forward's kernel (256x256 tile, 16x16x32 MFMA, accumulators straight to memory).  Needs 64-multiples.
// (HS_DGRAD_NN=1, read once: A/B runs against the NN kernels)
inline bool dgrad_nt(const hs_vit_desc& c) {
    static const bool off = getenv("HS_DGRAD_NN") != nullptr;
    return !off && !(c.d & 63) && !(c.mlp & 63);
}

Layout make_layout(const hs_vit_desc& c) {
    Layout L;
    const size_t M = (size_t)c.B * c.N, d = c.d, mlp = c.mlp;
    const size_t Ls = c.save_for_bwd ? c.L : 1;
    // save_for_bwd == 2 (bf16 only): the two [rows][mlp] activations of the MLP (gelu(z) and gelu'(z): 46 % of what a layer saves) are NOT
    // kept per layer; the backward re-runs fc1 with its GELU epilogue from the saved LN2 output (same kernel, same operands: identical bits)
    const size_t Lm = c.save_for_bwd == 1 ? c.L : 1;
    const size_t es = c.prec == HS_PREC_F32 ? 4 : 2;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al256(bytes); return o; };
    L.per_w = 3 * d * d + d * d + mlp * d + d * mlp;
    // 16-bit (bf16 / fp16) weight copies: in the workspace, re-made per call -- unless the caller owns them (hs_vit_desc.w16 / .w16t)
    L.wbf = take(c.prec != HS_PREC_F32 && !c.w16 ? (size_t)c.L * L.per_w * 2 : 0);
    // TRANSPOSED bf16 weight copies for the backward pass' dgrads (see dgrad_nt below): same per-layer order [Wqkv^T | Wo^T | W1^T | W2^T]
    L.wbft = take(c.prec != HS_PREC_F32 && c.save_for_bwd && dgrad_nt(c) && !c.w16t ? (size_t)c.L * L.per_w * 2 : 0);
    L.xmid = take(Ls * M * d * 4);
    L.h1 = take(Ls * M * d * es);
    L.h2 = take(Ls * M * d * es);
    L.qkv = take(Ls * M * 3 * d * es);
    L.o = take(Ls * M * d * es);
    L.lse = take(Ls * (size_t)c.B * c.H * c.N * 4);
    L.z = take(Lm * M * mlp * es);
    L.a = take(Lm * M * mlp * es);
    L.stats = take(Ls * 4 * M * 4);
    L.fstats = take(2 * M * 4);
    const bool bwd = c.save_for_bwd != 0;
    L.dx = take(bwd ? M * d * 4 : 0);
    L.dh = take(bwd ? M * d * es : 0);     // dgrad outputs feeding LN backward: bf16 (autocast's Linear backward dtype) / f32 in parity mode
    // bf16 copies of the residual-stream gradient (GEMM operand): THREE rotating buffers -- a layer's backward produces two new ones
    // (after LN2 and after LN1 backward) and its four weight-gradient GEMMs, launched as ONE group at the end of the layer, still
    // read the previous two.  Copy j (j = 0 entering the top layer, two per layer) lives in buffer j % 3, so chunked calls agree.
    L.dxb = take(bwd && c.prec != HS_PREC_F32 ? 3 * al256(M * d * 2) : 0);
    L.dob = take(bwd ? M * d * es : 0);
    L.dz = take(bwd ? M * mlp * es : 0);
    L.dqkv = take(bwd ? M * 3 * d * es : 0);
    L.delta = take(bwd ? (size_t)c.B * c.H * c.N * 4 * 2 : 0);   // hs_attn_bwd scratch: -delta | -lse*log2e
    L.qb = take(c.qkv_bias ? (size_t)3 * d * 4 : 0);               // one layer's qkv bias with its q third pre-scaled
    L.total = off;
    return L;
}

int check_desc(const hs_vit_desc* c) {
    if (!c) return HS_ERR_ARG;
    if (c->struct_size != sizeof(hs_vit_desc)) return HS_ERR_ARG;     // a caller compiled against another header (ABI 2)
    if ((c->w16 || c->w16t) && c->prec == HS_PREC_F32) return HS_ERR_ARG;
    if (((uintptr_t)c->w16 | (uintptr_t)c->w16t) & 15) return HS_ERR_SHAPE;
    if (c->w16t && ((c->d & 63) || (c->mlp & 63))) return HS_ERR_SHAPE;
    if (c->B < 1 || c->N < 1 || c->L < 1 || c->H < 1) return HS_ERR_SHAPE;
    if (c->d != c->H * 64) return HS_ERR_SHAPE;  // attention kernels are specialised for head_dim 64
    if ((c->d & 7) || (c->mlp & 7) || c->d > 1024) return HS_ERR_SHAPE;
    if (c->prec != HS_PREC_BF16 && c->prec != HS_PREC_F32 && c->prec != HS_PREC_F16) return HS_ERR_ARG;
    if (c->save_for_bwd < 0 || c->save_for_bwd > 2) return HS_ERR_ARG;
    if (c->save_for_bwd == 2 && c->prec == HS_PREC_F32) return HS_ERR_ARG;   // MLP recompute: the 16-bit paths' option
    if (c->ln_eps < 0.f) return HS_ERR_ARG;
    if (c->qkv_bias_grad && !c->qkv_bias) return HS_ERR_ARG;
    return HS_OK;
}

// indices into the per-layer parameter list (MONAI state-dict order)
enum { P_W1 = 0, P_B1, P_W2, P_B2, P_N1W, P_N1B, P_WO, P_BO, P_WQKV, P_N2W, P_N2B, P_COUNT };

const float kQScale = 0.125f * 1.44269504088896341f;   // dh^-0.5 * log2(e), dh = 64

}  // namespace

extern "C" size_t hs_vit_workspace_bytes(const hs_vit_desc* desc) {
    if (check_desc(desc) != HS_OK) return 0;
    return make_layout(*desc).total;
}

extern "C" size_t hs_vit_weights_elems(const hs_vit_desc* desc) {
    if (check_desc(desc) != HS_OK || desc->prec == HS_PREC_F32) return 0;
    return (size_t)desc->L * make_layout(*desc).per_w;
}

namespace {
// the straight / transposed 16-bit copies of layers [l_lo, l_hi) into copy sets laid out like L.wbf (layer l at l * per_w)
int cast_weights(const hs_vit_desc& c, const float* const* lp, void* w16, int l_lo, int l_hi, void* stream) {
    const size_t d = c.d, mlp = c.mlp, per_w = 4 * d * d + 2 * d * mlp;
    std::vector<const float*> src;
    std::vector<void*> dst;
    std::vector<int64_t> cnt;
    for (int l = l_lo; l < l_hi; ++l) {
        bf16_t* wl = (bf16_t*)w16 + (size_t)l * per_w;
        const float* const* p = lp + (size_t)l * P_COUNT;
        src.push_back(p[P_WQKV]); dst.push_back(wl); cnt.push_back((int64_t)3 * d * d);
        src.push_back(p[P_WO]); dst.push_back(wl + 3 * d * d); cnt.push_back((int64_t)d * d);
        src.push_back(p[P_W1]); dst.push_back(wl + 4 * d * d); cnt.push_back((int64_t)mlp * d);
        src.push_back(p[P_W2]); dst.push_back(wl + 4 * d * d + mlp * d); cnt.push_back((int64_t)d * mlp);
    }
    return hs_cast_bf16_multi(src.data(), dst.data(), cnt.data(), (int)src.size(), stream, c.prec == HS_PREC_F16);
}
// src16 == nullptr: from the fp32 parameters; else from the straight 16-bit copy set src16
int transpose_weights(const hs_vit_desc& c, const float* const* lp, const void* src16, void* w16t, int l_lo, int l_hi, void* stream) {
    const size_t d = c.d, mlp = c.mlp, per_w = 4 * d * d + 2 * d * mlp;
    std::vector<const void*> src;
    std::vector<void*> dst;
    std::vector<int> R, Cc;
    for (int l = l_lo; l < l_hi; ++l) {
        bf16_t* wl = (bf16_t*)w16t + (size_t)l * per_w;
        const size_t off[4] = {0, 3 * d * d, 4 * d * d, 4 * d * d + mlp * d};
        const int idx[4] = {P_WQKV, P_WO, P_W1, P_W2};
        const int rc[4][2] = {{3 * c.d, c.d}, {c.d, c.d}, {c.mlp, c.d}, {c.d, c.mlp}};
        for (int i = 0; i < 4; ++i) {
            src.push_back(src16 ? (const void*)((const bf16_t*)src16 + (size_t)l * per_w + off[i]) : (const void*)lp[(size_t)l * P_COUNT + idx[i]]);
            dst.push_back(wl + off[i]); R.push_back(rc[i][0]); Cc.push_back(rc[i][1]);
        }
    }
    return hs_transpose16_multi(src.data(), src16 != nullptr, dst.data(), R.data(), Cc.data(), (int)src.size(), stream, c.prec == HS_PREC_F16);
}
}  // namespace

extern "C" int hs_vit_cast_weights(const hs_vit_desc* desc, const float* const* lp, void* w16, void* w16t, void* stream) {
    RC(check_desc(desc));
    const hs_vit_desc& c = *desc;
    if (c.prec == HS_PREC_F32 || (!w16 && !w16t)) return HS_ERR_ARG;
    if (!lp && !(w16 && w16t)) return HS_ERR_ARG;                  // without the fp32 parameters: w16 -> w16t only
    if (((uintptr_t)w16 | (uintptr_t)w16t) & 15) return HS_ERR_SHAPE;
    if (w16t && ((c.d & 63) || (c.mlp & 63))) return HS_ERR_SHAPE;
    if (lp && w16) RC(cast_weights(c, lp, w16, 0, c.L, stream));
    if (w16t) RC(transpose_weights(c, lp, lp ? nullptr : w16, w16t, 0, c.L, stream));
    return HS_OK;
}

extern "C" int hs_vit_blocks_fwd(const hs_vit_desc* desc, const float* x0, const float* const* lp, const float* const* np,
                                 float* y, float* hidden, void* ws, size_t ws_bytes, void* stream) {
    RC(check_desc(desc));
    return hs_vit_blocks_fwd_range(desc, x0, lp, np, y, hidden, ws, ws_bytes, 0, desc->L, stream);
}

extern "C" int hs_vit_blocks_fwd_range(const hs_vit_desc* desc, const float* x0, const float* const* lp, const float* const* np,
                                       float* y, float* hidden, void* ws, size_t ws_bytes, int l_lo, int l_hi, void* stream) {
    RC(check_desc(desc));
    if (!lp || !np || !hidden || !ws) return HS_ERR_ARG;
    const hs_vit_desc& c = *desc;
    if (l_lo < 0 || l_hi > c.L || l_lo >= l_hi) return HS_ERR_SHAPE;
    if ((l_lo == 0 && !x0) || (l_hi == c.L && !y)) return HS_ERR_ARG;
    if ((l_lo > 0 || l_hi < c.L) && c.prec != HS_PREC_F32 && !c.w16) return HS_ERR_ARG;      // a sub-range cannot cast every layer's weights
    const Layout L = make_layout(c);
    if (ws_bytes < L.total || ((uintptr_t)ws & 255)) return HS_ERR_SHAPE;
    char* w = (char*)ws;
    const int M = c.B * c.N, d = c.d, mlp = c.mlp;
    const size_t Md = (size_t)M * d;
    const bool bf = c.prec != HS_PREC_F32;          // 16-bit operands: bf16, or fp16 (HS_PREC_F16, forward only)
    const int f16 = c.prec == HS_PREC_F16;
    const size_t es = bf ? 2 : 4;
    const float eps = c.ln_eps > 0.f ? c.ln_eps : 1e-5f;

    // 16-bit copies of the four weight matrices of every layer: the caller's (desc.w16), or re-made here in one multi-segment launch
    const bf16_t* wcopy = bf ? (c.w16 ? (const bf16_t*)c.w16 : (const bf16_t*)(w + L.wbf)) : nullptr;
    if (bf && !c.w16) RC(cast_weights(c, lp, w + L.wbf, 0, c.L, stream));
    for (int l = l_lo; l < l_hi; ++l) {
        const size_t ls = c.save_for_bwd ? l : 0;
        const float* const* p = lp + (size_t)l * P_COUNT;
        const float* xin = l == 0 ? x0 : hidden + (size_t)(l - 1) * Md;
        float* xout = hidden + (size_t)l * Md;
        float* xmid = (float*)(w + L.xmid) + ls * Md;
        char* h1 = w + L.h1 + ls * Md * es;
        char* h2 = w + L.h2 + ls * Md * es;
        char* qkv = w + L.qkv + ls * Md * 3 * es;
        char* o = w + L.o + ls * Md * es;
        float* lse = (float*)(w + L.lse) + ls * (size_t)c.B * c.H * c.N;
        const size_t lm = c.save_for_bwd == 1 ? l : 0;
        char* z = w + L.z + lm * (size_t)M * mlp * es;
        char* a = w + L.a + lm * (size_t)M * mlp * es;
        float* st = (float*)(w + L.stats) + ls * 4 * (size_t)M;
        const void *wqkv, *wo, *w1, *w2;
        if (bf) {
            const bf16_t* wl = wcopy + (size_t)l * L.per_w;
            wqkv = wl; wo = wl + 3 * d * d; w1 = wl + 4 * d * d; w2 = wl + 4 * d * d + (size_t)mlp * d;
        } else {
            wqkv = p[P_WQKV]; wo = p[P_WO]; w1 = p[P_W1]; w2 = p[P_W2];
        }
        RC(hs_layernorm_fwd16(xin, nullptr, p[P_N1W], p[P_N1B], bf ? nullptr : (float*)h1, bf ? h1 : nullptr, st, st + M, M, d, eps, f16, stream));
        // bf16: the q third leaves the projection already multiplied by dh^-0.5 * log2(e) (one rounding), which is what the
        // attention kernels' exp2-based softmax consumes (HS_ATTN_Q_PRESCALED); the f32 parity path keeps raw q
        const float* qbias = nullptr;
        if (c.qkv_bias) {           // timm-style biased projection: the pre-scale of the q third applies to its bias too
            qbias = c.qkv_bias[l];
            if (bf) {
                float* qb = (float*)(w + L.qb);
                RC(hs_scale(qbias, qb, kQScale, d, stream));
                RC(hs_scale(qbias + d, qb + d, 1.0f, (int64_t)2 * d, stream));
                qbias = qb;
            }
        }
        RC(gemm(c.prec, HS_GEMM_NT, h1, wqkv, M, 3 * d, d, d, d, bf ? nullptr : (float*)qkv, bf ? qkv : nullptr, 3 * d, qbias, nullptr,
                nullptr, nullptr, qbias ? HS_EPI_BIAS : 0, 1, stream, nullptr, bf ? d : 0, kQScale));
        RC(hs_attn_fwd(qkv, o, lse, c.B, c.N, c.H, 64, c.prec, bf ? HS_ATTN_Q_PRESCALED : 0, stream));
        RC(gemm(c.prec, HS_GEMM_NT, o, wo, M, d, d, d, d, xmid, nullptr, d, p[P_BO], xin, nullptr, nullptr, HS_EPI_BIAS | HS_EPI_RESID, 1, stream));
        RC(hs_layernorm_fwd16(xmid, nullptr, p[P_N2W], p[P_N2B], bf ? nullptr : (float*)h2, bf ? h2 : nullptr, st + 2 * M, st + 3 * M, M, d, eps, f16, stream));
        RC(gemm(c.prec, HS_GEMM_NT, h2, w1, M, mlp, d, d, d, bf ? nullptr : (float*)a, bf ? a : nullptr, mlp, p[P_B1], nullptr, nullptr,
                c.save_for_bwd == 1 ? z : nullptr, HS_EPI_BIAS | HS_EPI_GELU | HS_EPI_SAVE_DGELU, 1, stream));   // (2: gelu' is made again in the backward)
        RC(gemm(c.prec, HS_GEMM_NT, a, w2, M, d, mlp, mlp, mlp, xout, nullptr, d, p[P_B2], xmid, nullptr, nullptr, HS_EPI_BIAS | HS_EPI_RESID, 1, stream));
    }
    if (l_hi < c.L) return HS_OK;
    float* fs = (float*)(w + L.fstats);
    RC(hs_layernorm_fwd(hidden + (size_t)(c.L - 1) * Md, nullptr, np[0], np[1], y, nullptr, fs, fs + M, M, d, eps, stream));
    return HS_OK;
}

extern "C" int hs_vit_blocks_bwd(const hs_vit_desc* desc, const float* x0, const float* const* lp, const float* const* np,
                                 const float* dy, float* dx0, float* const* lg, float* const* ng, const float* hidden, void* ws,
                                 size_t ws_bytes, int l_hi, int l_lo, void* stream) {
    RC(check_desc(desc));
    const hs_vit_desc& c = *desc;
    if (!c.save_for_bwd) return HS_ERR_ARG;
    if (!x0 || !lp || !np || !lg || !hidden || !ws) return HS_ERR_ARG;
    if (l_lo < 0 || l_hi > c.L || l_lo >= l_hi) return HS_ERR_SHAPE;
    if (l_hi == c.L && (!dy || !ng)) return HS_ERR_ARG;
    if (l_lo == 0 && !dx0) return HS_ERR_ARG;
    const Layout L = make_layout(c);
    if (ws_bytes < L.total || ((uintptr_t)ws & 255)) return HS_ERR_SHAPE;
    char* w = (char*)ws;
    const int M = c.B * c.N, d = c.d, mlp = c.mlp;
    const size_t Md = (size_t)M * d;
    // bf16 (the reference's autocast policy): GEMM operands / dgrad outputs bf16, residual-stream gradient fp32 + bf16 copies.
    // f16 (round 5): the same sequence and kernels with IEEE fp16 wherever bf16 stands (the caller's dy carries a power-of-two loss scale).
    // f32 (parity mode): everything fp32, exact-f32 MFMA GEMMs and attention.
    const bool bf = c.prec != HS_PREC_F32;
    const int f16 = c.prec == HS_PREC_F16;
    const int ln16 = f16 ? HS_LN_DY_F16 : HS_LN_DY_BF16;
    const size_t es = bf ? 2 : 4;
    const int P = c.prec;
    float* dx = (float*)(w + L.dx);
    char* dh = w + L.dh;
    bf16_t* dxb = bf ? (bf16_t*)(w + L.dxb) : nullptr;            // copy 0 (entering the top layer)
    char* dob = w + L.dob;
    char* dz = w + L.dz;
    char* dqkv = w + L.dqkv;
    float* delta = (float*)(w + L.delta);

    const bool nt = bf && dgrad_nt(c);
    const bf16_t* wcopy = bf ? (c.w16 ? (const bf16_t*)c.w16 : (const bf16_t*)(w + L.wbf)) : nullptr;     // (the forward left it there)
    const bf16_t* wcopyt = nt ? (c.w16t ? (const bf16_t*)c.w16t : (const bf16_t*)(w + L.wbft)) : nullptr;
    if (nt && !c.w16t) RC(transpose_weights(c, lp, nullptr, w + L.wbft, l_lo, l_hi, stream));   // W^T copies of this call's layers: one launch
    if (l_hi == c.L) {
        const float* fs = (const float*)(w + L.fstats);
        // (each LN backward also emits the column sums of the dx it produces = the bias gradient of the Linear below it)
        RC(hs_layernorm_bwd(dy, f16 ? HS_LN_OUT_F16 : HS_LN_DY_F32, hidden + (size_t)(c.L - 1) * Md, nullptr, np[0], fs, fs + M, nullptr, dx, dxb, ng[0], ng[1],
                            lg[(size_t)(c.L - 1) * P_COUNT + P_B2], M, d, stream));
    }
    for (int l = l_hi - 1; l >= l_lo; --l) {
        const float* const* p = lp + (size_t)l * P_COUNT;
        float* const* g = lg + (size_t)l * P_COUNT;
        const float* xin = l == 0 ? x0 : hidden + (size_t)(l - 1) * Md;
        const float* xmid = (const float*)(w + L.xmid) + (size_t)l * Md;
        const char* h1 = w + L.h1 + (size_t)l * Md * es;
        const char* h2 = w + L.h2 + (size_t)l * Md * es;
        const char* qkv = w + L.qkv + (size_t)l * Md * 3 * es;
        const char* o = w + L.o + (size_t)l * Md * es;
        const float* lse = (const float*)(w + L.lse) + (size_t)l * c.B * c.H * c.N;
        const size_t lm = c.save_for_bwd == 1 ? l : 0;
        char* z = w + L.z + lm * (size_t)M * mlp * es;
        char* a = w + L.a + lm * (size_t)M * mlp * es;
        const float* st = (const float*)(w + L.stats) + (size_t)l * 4 * M;
        const void *wqkv, *wo, *w1, *w2;
        if (bf) {
            const bf16_t* wl = wcopy + (size_t)l * L.per_w;
            wqkv = wl; wo = wl + 3 * d * d; w1 = wl + 4 * d * d; w2 = wl + 4 * d * d + (size_t)mlp * d;
        } else {
            wqkv = p[P_WQKV]; wo = p[P_WO]; w1 = p[P_W1]; w2 = p[P_W2];
        }

        if (bf) {
            // bf16: dgrad chain first, the four weight gradients as one grouped launch at the end of the layer
            const bf16_t* wlt = wcopyt + (size_t)l * L.per_w;        // (valid when nt)
            const bf16_t *wqkvt = wlt, *wot = wlt + 3 * d * d, *w1t = wlt + 4 * d * d, *w2t = wlt + 4 * d * d + (size_t)mlp * d;
            const int j = 2 * (c.L - 1 - l);
            const size_t dxs = al256(Md * 2);
            const bf16_t* dx_in = (const bf16_t*)(w + L.dxb + (size_t)(j % 3) * dxs);
            bf16_t* dx_mid = (bf16_t*)(w + L.dxb + (size_t)((j + 1) % 3) * dxs);
            bf16_t* dx_out = (bf16_t*)(w + L.dxb + (size_t)((j + 2) % 3) * dxs);
            if (c.save_for_bwd == 2)    // MLP recompute: gelu(z) and gelu'(z) of THIS layer again, exactly as the forward made them
                RC(gemm(P, HS_GEMM_NT, h2, w1, M, mlp, d, d, d, nullptr, a, mlp, p[P_B1], nullptr, nullptr, z,
                        HS_EPI_BIAS | HS_EPI_GELU | HS_EPI_SAVE_DGELU, 1, stream));
            // ---- MLP branch: x' = xm + W2 gelu(W1 h2 + b1) + b2        (db2 came with the dx that entered this layer)
            if (nt) RC(gemm(P, HS_GEMM_NT, dx_in, w2t, M, mlp, d, d, d, nullptr, dz, mlp, nullptr, nullptr, z, nullptr,
                            HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU, 1, stream, g[P_B1]));       // db1 = colsum(dz), fused in the epilogue
            else RC(gemm(P, HS_GEMM_NN, dx_in, w2, M, mlp, d, d, mlp, nullptr, dz, mlp, nullptr, nullptr, z, nullptr,
                         HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU, 1, stream, g[P_B1]));
            if (nt) RC(gemm(P, HS_GEMM_NT, dz, w1t, M, d, mlp, mlp, mlp, nullptr, dh, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
            else RC(gemm(P, HS_GEMM_NN, dz, w1, M, d, mlp, mlp, d, nullptr, dh, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
            RC(hs_layernorm_bwd(dh, ln16, xmid, nullptr, p[P_N2W], st + 2 * M, st + 3 * M, dx, dx, dx_mid, g[P_N2W], g[P_N2B], g[P_BO], M, d, stream));
            // ---- attention branch: xm = x + Wo attn(LN1 x) + bo         (dbo came from the LN2 backward just above)
            if (nt) RC(gemm(P, HS_GEMM_NT, dx_mid, wot, M, d, d, d, d, nullptr, dob, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
            else RC(gemm(P, HS_GEMM_NN, dx_mid, wo, M, d, d, d, d, nullptr, dob, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
            RC(hs_attn_bwd(qkv, o, dob, lse, dqkv, delta, c.B, c.N, c.H, 64, HS_ATTN_Q_PRESCALED | (f16 ? HS_ATTN_F16 : 0), stream));
            // qkv_bias=True: d(bias) = column sums of dqkv (hs_attn_bwd returns the gradient of the UNSCALED q, k, v)
            if (c.qkv_bias_grad && c.qkv_bias_grad[l]) RC(hs_colsum(dqkv, 1 + f16, c.qkv_bias_grad[l], M, 3 * d, 3 * d, stream));
            if (nt) RC(gemm(P, HS_GEMM_NT, dqkv, wqkvt, M, d, 3 * d, 3 * d, 3 * d, nullptr, dh, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
            else RC(gemm(P, HS_GEMM_NN, dqkv, wqkv, M, d, 3 * d, 3 * d, d, nullptr, dh, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
            float* dxo = (l == 0) ? dx0 : dx;
            float* b2_below = (l == 0) ? nullptr : lg[(size_t)(l - 1) * P_COUNT + P_B2];
            RC(hs_layernorm_bwd(dh, ln16, xin, nullptr, p[P_N1W], st, st + M, dx, dxo, l == 0 ? nullptr : dx_out, g[P_N1W], g[P_N1B], b2_below, M, d, stream));
            // ---- dW2 = dx'^T a, dW1 = dz^T h2, dWqkv = dqkv^T h1, dWo = dxm^T o (largest first)
            const int mn[4][2] = {{d, mlp}, {mlp, d}, {3 * d, d}, {d, d}};
            static const int split_env = getenv("HS_WGRAD_GROUP_SPLIT") ? atoi(getenv("HS_WGRAD_GROUP_SPLIT")) : 0;   // (A/B runs)
            const int split = split_env > 0 ? split_env : wgrad_group_split(mn, 4, M);
            hs_gemm_args wg[4] = {};
            const void* dys[4] = {dx_in, dz, dqkv, dx_mid};
            const void* xs[4] = {a, h2, h1, o};
            float* dws[4] = {g[P_W2], g[P_W1], g[P_WQKV], g[P_WO]};
            for (int i = 0; i < 4; ++i) {
                wg[i].A = dys[i]; wg[i].B = xs[i]; wg[i].M = mn[i][0]; wg[i].N = mn[i][1]; wg[i].K = M;
                wg[i].lda = mn[i][0]; wg[i].ldb = mn[i][1]; wg[i].ldc = mn[i][1]; wg[i].batch = 1; wg[i].out_f32 = dws[i];
                wg[i].alpha = 1.0f; wg[i].flags = HS_EPI_ATOMIC; wg[i].layout = HS_GEMM_TN; wg[i].split_k = split;
            }
            RC(hs_gemm_group_prec(wg, 4, P, stream));
            continue;
        }
        // ---- f32 parity mode: the same sequence with exact-f32 kernels, every buffer fp32, weight gradients where their operands
        // appear, bias gradient of fc1 by a separate column-sum pass (the fused one lives in the bf16 GEMM epilogue)
        float* dzf = (float*)dz; float* dhf = (float*)dh; float* dobf = (float*)dob;
        RC(gemm(P, HS_GEMM_TN, dx, a, d, mlp, M, d, mlp, g[P_W2], nullptr, mlp, nullptr, nullptr, nullptr, nullptr, HS_EPI_ATOMIC,
                wgrad_split(d, mlp, M, P), stream));
        RC(gemm(P, HS_GEMM_NN, dx, w2, M, mlp, d, d, mlp, dzf, nullptr, mlp, nullptr, nullptr, z, nullptr,
                HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU, 1, stream));
        RC(hs_colsum(dz, 0, g[P_B1], M, mlp, mlp, stream));
        RC(gemm(P, HS_GEMM_TN, dz, h2, mlp, d, M, mlp, d, g[P_W1], nullptr, d, nullptr, nullptr, nullptr, nullptr, HS_EPI_ATOMIC,
                wgrad_split(mlp, d, M, P), stream));
        RC(gemm(P, HS_GEMM_NN, dz, w1, M, d, mlp, mlp, d, dhf, nullptr, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
        RC(hs_layernorm_bwd(dh, 0, xmid, nullptr, p[P_N2W], st + 2 * M, st + 3 * M, dx, dx, nullptr, g[P_N2W], g[P_N2B], g[P_BO], M, d, stream));
        RC(gemm(P, HS_GEMM_TN, dx, o, d, d, M, d, d, g[P_WO], nullptr, d, nullptr, nullptr, nullptr, nullptr, HS_EPI_ATOMIC,
                wgrad_split(d, d, M, P), stream));
        RC(gemm(P, HS_GEMM_NN, dx, wo, M, d, d, d, d, dobf, nullptr, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
        RC(hs_attn_bwd(qkv, o, dob, lse, dqkv, delta, c.B, c.N, c.H, 64, HS_ATTN_F32, stream));
        if (c.qkv_bias_grad && c.qkv_bias_grad[l]) RC(hs_colsum(dqkv, 0, c.qkv_bias_grad[l], M, 3 * d, 3 * d, stream));
        RC(gemm(P, HS_GEMM_TN, dqkv, h1, 3 * d, d, M, 3 * d, d, g[P_WQKV], nullptr, d, nullptr, nullptr, nullptr, nullptr, HS_EPI_ATOMIC,
                wgrad_split(3 * d, d, M, P), stream));
        RC(gemm(P, HS_GEMM_NN, dqkv, wqkv, M, d, 3 * d, 3 * d, d, dhf, nullptr, d, nullptr, nullptr, nullptr, nullptr, 0, 1, stream));
        float* dxo = (l == 0) ? dx0 : dx;
        float* b2_below = (l == 0) ? nullptr : lg[(size_t)(l - 1) * P_COUNT + P_B2];
        RC(hs_layernorm_bwd(dh, 0, xin, nullptr, p[P_N1W], st, st + M, dx, dxo, nullptr, g[P_N1W], g[P_N1B], b2_below, M, d, stream));
    }
    return HS_OK;
}
