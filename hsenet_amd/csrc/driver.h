// Host-side helpers shared by the fused drivers (vit.hip: block stack; path.hip: embedding / scorer / packer).
#pragma once
#include "common.h"

int hs_gemm_group_prec(const hs_gemm_args* args, int count, int prec, void* stream);      // gemm.hip (declared in hsenet_hip.h too)

int hs_cast_bf16_multi(const float* const* src, void* const* dst, const int64_t* n, int count, void* stream, int f16 = 0);   // small.hip (f16: fp16 instead of bf16)
int hs_layernorm_fwd16(const float* x, const float* resid, const float* gamma, const float* beta, float* y_f32, void* y_16, float* mean,
                       float* rstd, int rows, int d, float eps, int f16, void* stream);                                        // norm.hip
// [R][C] -> 16-bit [C][R] for up to `count` matrices in one launch per 64 (R, C multiples of 64); the sources are fp32 (cast to bf16, or
// fp16 with f16 != 0) or, with src_is_16, already 16-bit (bits pass through)                                      // small.hip
int hs_transpose16_multi(const void* const* src, int src_is_16, void* const* dst, const int* R, const int* C, int count, void* stream, int f16 = 0);

// y16 = cast16(a (+ b)) [M][N] and column sums of the rounded values: columns [0, nsplit) += into cs_lo, [nsplit, N) into cs_hi   // small.hip
int hs_addcast_colsum(const float* a, const float* b, void* y16, float* cs_lo, float* cs_hi, int M, int N, int nsplit, int f16, void* stream);

namespace hsdrv {

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// bump allocator over a caller-provided workspace: offsets only (the same walk sizes and lays out the workspace)
struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off += al256(bytes); return o; }
};

inline int gemm(int prec, int layout, const void* A, const void* Bm, int M, int N, int K, long lda, long ldb, float* of32, void* obf,
                long ldc, const float* bias, const float* resid, const void* aux, void* aux_out, int flags, int split_k, void* s,
                float* colsum = nullptr, int scale_cols = 0, float scale_cols_alpha = 1.0f, int batch = 1, long sA = 0, long sB = 0,
                long sC = 0) {
    hs_gemm_args g = {};
    g.colsum_out = colsum;
    g.scale_cols = scale_cols; g.scale_cols_alpha = scale_cols_alpha;
    g.A = A; g.B = Bm; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.batch = batch;
    g.strideA = sA; g.strideB = sB; g.strideC = sC;
    g.out_f32 = of32; g.out_bf16 = obf; g.ldc = ldc; g.bias = bias; g.resid = resid; g.aux = aux; g.aux_out = aux_out;
    g.alpha = 1.0f; g.flags = flags; g.layout = layout; g.split_k = split_k;
    return hs_gemm(&g, prec, s);
}

// split-K factor for a wgrad whose output has ceil(Mo/256)*ceil(No/128) tiles and K reduction rows: pick the split
// that minimises (rounds over 256 CUs) x (K-tiles per block + a fixed per-block cost ~ the atomic epilogue).
inline int wgrad_split(int Mo, int No, int K, int prec = HS_PREC_BF16) {
    const bool bf = prec == HS_PREC_BF16;                       // exact-f32 GEMM: 64x64 tiles, 32-row K-tiles
    const int tiles = bf ? ((Mo + 255) / 256) * ((No + 127) / 128) : ((Mo + 63) / 64) * ((No + 63) / 64);
    const int kt = bf ? (K + 63) / 64 : (K + 31) / 32;
    int best = 1;
    long best_cost = -1;
    for (int s = 1; s <= 32 && s <= kt; ++s) {
        const long rounds = ((long)tiles * s + 255) / 256;
        const long cost = rounds * ((kt + s - 1) / s + 16);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = s; }
    }
    return best;
}

// common split-K factor of a GROUP of bf16 wgrads (hs_gemm_group) with output shapes mn[i] = {Mo, No} and K reduction rows: the
// same cost model over the group's total tile count -- at one volume per GPU (K = 4097 rows, 216 tiles for a ViT-B layer) that is
// ONE round of unsplit tiles instead of four launches of 5-22 K-tiles each behind an atomic epilogue.
inline int wgrad_group_split(const int (*mn)[2], int count, int K) {
    long tiles = 0;
    for (int i = 0; i < count; ++i) tiles += (long)((mn[i][0] + 255) / 256) * ((mn[i][1] + 127) / 128);
    const int kt = (K + 63) / 64;
    int best = 1;
    long best_cost = -1;
    for (int s = 1; s <= 32 && s <= kt; ++s) {
        const long rounds = (tiles * s + 255) / 256;
        const long cost = rounds * ((kt + s - 1) / s + 16);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = s; }
    }
    return best;
}

// dW[No_rows = n_out][n_in] += dY^T X   (dY bf16 [rows][ldy] columns [0, n_out), X bf16 [rows][ldx]); dW zeroed by the caller
inline int wgrad(const void* dy, long ldy, const void* x, long ldx, float* dw, int n_out, int n_in, int rows, void* s, int prec = HS_PREC_BF16) {
    return gemm(prec, HS_GEMM_TN, dy, x, n_out, n_in, rows, ldy, ldx, dw, nullptr, n_in, nullptr, nullptr, nullptr, nullptr,
                HS_EPI_ATOMIC, wgrad_split(n_out, n_in, rows), s);
}

}  // namespace hsdrv

#define RC(x) do { int rc__ = (x); if (rc__ != HS_OK) return rc__; } while (0)
