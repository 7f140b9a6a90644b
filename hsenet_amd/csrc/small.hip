// K1 / K5 / K6 / K7 / K8: the HBM-bound kernels around the GEMMs (see include/hsenet_hip.h for the reference
// operator sequence each replaces).  All use 16-B accesses per lane and wave-shuffle reductions.
#include "common.h"

// ------------------------------------------------------------------------------------------------------
// K1 patchify: x.clone() + rearrange 'b c (h p1)(w p2)(d p3) -> b (h w d)(p1 p2 p3 c)' (vit.py:437,455).
// Fast path (C == 1, p3 % 8 == 0): a lane reads 8 consecutive voxels along W (32 B), so a wave reads 2 KiB
// contiguous, and writes them as 8 consecutive features of one patch row.
// ------------------------------------------------------------------------------------------------------
template <int FMT>   // HS_PREC_*: 0 bf16, 1 f32, 2 fp16
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ vol, void* __restrict__ out, int B, int D,
                                                       int H, int W, int p1, int p2, int p3) {
    const int W8 = W >> 3;
    const long total = (long)B * D * H * W8;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int w0 = (int)(g % W8) * 8;
    long t = g / W8;
    const int hh = (int)(t % H); t /= H;
    const int dd = (int)(t % D);
    const int b = (int)(t / D);
    const int gh = H / p2, gw = W / p3;
    const long Np = (long)(D / p1) * gh * gw;
    const int Pd = p1 * p2 * p3;
    const long patch = ((long)(dd / p1) * gh + hh / p2) * gw + w0 / p3;
    const int feat = ((dd % p1) * p2 + hh % p2) * p3 + (w0 % p3);
    const f32x4 a = *(const f32x4*)(vol + g * 8);
    const f32x4 c = *(const f32x4*)(vol + g * 8 + 4);
    const long o = (b * Np + patch) * Pd + feat;
    if (FMT != HS_PREC_F32) {
        constexpr bool F16 = FMT == HS_PREC_F16;
        u32x4 pk;
        pk[0] = pack2h<F16>(a[0], a[1]); pk[1] = pack2h<F16>(a[2], a[3]);
        pk[2] = pack2h<F16>(c[0], c[1]); pk[3] = pack2h<F16>(c[2], c[3]);
        *(u32x4*)((bf16_t*)out + o) = pk;
    } else {
        *(f32x4*)((float*)out + o) = a;
        *(f32x4*)((float*)out + o + 4) = c;
    }
}
// generic path: one element per thread (any C, any patch size)
template <int FMT>
__global__ __launch_bounds__(256) void patchify_generic_kernel(const float* __restrict__ vol, void* __restrict__ out, int B,
                                                               int C, int D, int H, int W, int p1, int p2, int p3) {
    const long total = (long)B * C * D * H * W;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int ww = (int)(g % W);
    long t = g / W;
    const int hh = (int)(t % H); t /= H;
    const int dd = (int)(t % D); t /= D;
    const int cc = (int)(t % C);
    const int b = (int)(t / C);
    const int gh = H / p2, gw = W / p3;
    const long Np = (long)(D / p1) * gh * gw;
    const long Pd = (long)p1 * p2 * p3 * C;
    const long patch = ((long)(dd / p1) * gh + hh / p2) * gw + ww / p3;
    const long feat = ((long)((dd % p1) * p2 + hh % p2) * p3 + (ww % p3)) * C + cc;
    const long o = (b * Np + patch) * Pd + feat;
    if (FMT != HS_PREC_F32) ((bf16_t*)out)[o] = f2h<FMT == HS_PREC_F16>(vol[g]); else ((float*)out)[o] = vol[g];
}

extern "C" int hs_patchify(const float* vol, void* out, int B, int C, int D, int H, int W, int p1, int p2, int p3, int prec,
                           void* stream) {
    if (!vol || !out) return HS_ERR_ARG;
    if (B < 1 || C < 1 || p1 < 1 || p2 < 1 || p3 < 1 || D % p1 || H % p2 || W % p3) return HS_ERR_SHAPE;
    if (prec != HS_PREC_BF16 && prec != HS_PREC_F32 && prec != HS_PREC_F16) return HS_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(HS_PROF_PATCHIFY, stream, 0, (double)B * C * D * H * W * (prec == HS_PREC_F32 ? 8 : 6));
    if (C == 1 && (p3 % 8) == 0 && !(((uintptr_t)vol | (uintptr_t)out) & 15)) {
        const long total = (long)B * D * H * (W / 8);
        dim3 grid((unsigned)((total + 255) / 256));
        if (prec == HS_PREC_BF16) hipLaunchKernelGGL(patchify_kernel<HS_PREC_BF16>, grid, dim3(256), 0, s, vol, out, B, D, H, W, p1, p2, p3);
        else if (prec == HS_PREC_F16) hipLaunchKernelGGL(patchify_kernel<HS_PREC_F16>, grid, dim3(256), 0, s, vol, out, B, D, H, W, p1, p2, p3);
        else hipLaunchKernelGGL(patchify_kernel<HS_PREC_F32>, grid, dim3(256), 0, s, vol, out, B, D, H, W, p1, p2, p3);
    } else {
        const long total = (long)B * C * D * H * W;
        dim3 grid((unsigned)((total + 255) / 256));
        if (prec == HS_PREC_BF16) hipLaunchKernelGGL(patchify_generic_kernel<HS_PREC_BF16>, grid, dim3(256), 0, s, vol, out, B, C, D, H, W, p1, p2, p3);
        else if (prec == HS_PREC_F16) hipLaunchKernelGGL(patchify_generic_kernel<HS_PREC_F16>, grid, dim3(256), 0, s, vol, out, B, C, D, H, W, p1, p2, p3);
        else hipLaunchKernelGGL(patchify_generic_kernel<HS_PREC_F32>, grid, dim3(256), 0, s, vol, out, B, C, D, H, W, p1, p2, p3);
    }
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// tokens = cat(cls, (pe + pos) * score)   (vit.py:345-349, 459-461; pos add = PatchEmbeddingBlock.forward)
// cls == NULL (classification=False): no cls row, tokens is [B][Np][d].
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const float* __restrict__ pe, const float* __restrict__ pos,
                                                              const float* __restrict__ cls, const float* __restrict__ score,
                                                              float* __restrict__ tokens, float* __restrict__ pe_out, int B,
                                                              int Np, int d) {
    const int d4 = d >> 2;
    const int c0 = cls ? 1 : 0, Nt = Np + c0;
    const long total = (long)B * Nt * d4;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int ch = (int)(g % d4);
    const long row = g / d4;
    const int n = (int)(row % Nt);
    const int b = (int)(row / Nt);
    f32x4 v;
    if (n < c0) {
        v = *(const f32x4*)(cls + ch * 4);
    } else {
        const long pr = (long)b * Np + (n - c0);
        v = *(const f32x4*)(pe + pr * d + ch * 4);
        if (pos) v += *(const f32x4*)(pos + (long)(n - c0) * d + ch * 4);
        if (pe_out) *(f32x4*)(pe_out + pr * d + ch * 4) = v;
        if (score) v *= score[pr];
    }
    *(f32x4*)(tokens + row * d + ch * 4) = v;
}
extern "C" int hs_assemble_tokens(const float* pe, const float* pos, const float* cls, const float* score, float* tokens,
                                  float* pe_out, int B, int Np, int d, void* stream) {
    if (!pe || !tokens) return HS_ERR_ARG;
    if (B < 1 || Np < 1 || d < 4 || (d & 3)) return HS_ERR_SHAPE;
    const long total = (long)B * (Np + (cls ? 1 : 0)) * (d / 4);
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pe, pos,
                       cls, score, tokens, pe_out, B, Np, d);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// one wave per patch row p (all batches): dpe = dtok*score, dscore = <dtok, pe_pos>, dpos[p] += sum_b dpe.
// has_cls: dtok is [B][Np+1][d] and wave p == Np reduces the cls row; else dtok is [B][Np][d].
__global__ __launch_bounds__(256) void assemble_tokens_bwd_kernel(const float* __restrict__ dtok, const float* __restrict__ pe_pos,
                                                                  const float* __restrict__ score, float* __restrict__ dpe,
                                                                  float* __restrict__ dpos, float* __restrict__ dcls,
                                                                  float* __restrict__ dscore, int B, int Np, int d, int has_cls) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);   // p == Np handles the cls row
    const int Nt = Np + has_cls;
    if (p >= Nt) return;
    const int nch = d >> 2;
    for (int ch = lane; ch < nch; ch += 64) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            if (p == Np) {
                acc += *(const f32x4*)(dtok + ((long)b * Nt) * d + ch * 4);
            } else {
                f32x4 g = *(const f32x4*)(dtok + ((long)b * Nt + has_cls + p) * d + ch * 4);
                if (score) g *= score[(long)b * Np + p];
                if (dpe) *(f32x4*)(dpe + ((long)b * Np + p) * d + ch * 4) = g;
                acc += g;
            }
        }
        if (p == Np) { if (dcls) { float* o = dcls + ch * 4; for (int e = 0; e < 4; ++e) atomicAdd(o + e, acc[e]); } }
        else if (dpos) { float* o = dpos + (long)p * d + ch * 4; for (int e = 0; e < 4; ++e) atomicAdd(o + e, acc[e]); }
    }
    if (score && dscore && p < Np) {
        for (int b = 0; b < B; ++b) {
            float s = 0.f;
            for (int ch = lane; ch < nch; ch += 64) {
                const f32x4 g = *(const f32x4*)(dtok + ((long)b * Nt + has_cls + p) * d + ch * 4);
                const f32x4 x = *(const f32x4*)(pe_pos + ((long)b * Np + p) * d + ch * 4);
                s += g[0] * x[0] + g[1] * x[1] + g[2] * x[2] + g[3] * x[3];
            }
            s = wave_sum(s);
            if (lane == 0) dscore[(long)b * Np + p] = s;
        }
    }
}
// dcls == NULL  <=>  the forward had no cls row (dtok is [B][Np][d])
extern "C" int hs_assemble_tokens_bwd(const float* dtok, const float* pe_pos, const float* score, float* dpe, float* dpos,
                                      float* dcls, float* dscore, int B, int Np, int d, void* stream) {
    if (!dtok) return HS_ERR_ARG;
    if (score && dscore && !pe_pos) return HS_ERR_ARG;
    if (B < 1 || Np < 1 || d < 4 || (d & 3)) return HS_ERR_SHAPE;
    const int has_cls = dcls ? 1 : 0;
    hipLaunchKernelGGL(assemble_tokens_bwd_kernel, dim3((Np + has_cls + 3) / 4), dim3(256), 0, (hipStream_t)stream, dtok, pe_pos, score,
                       dpe, dpos, dcls, dscore, B, Np, d, has_cls);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// ------------------------------------------------------------------------------------------------------
// row softmax (fwd, in place) and its backward, rows of n <= 64: one wave per row.  Used by the single-head
// cross-attention of regular_attention (vit.py:25-33): scores [B*nq][nk=32].
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, long rows, int n) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float v = lane < n ? s[row * n + lane] : -__builtin_inff();
    const float m = wave_max(v);
    const float e = lane < n ? __expf(v - m) : 0.f;
    const float z = wave_sum(e);
    if (lane < n) s[row * n + lane] = e / z;
}
// ds = p * (dp - sum(p*dp)), in place on dp
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, long rows, int n) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float pv = lane < n ? p[row * n + lane] : 0.f;
    const float dv = lane < n ? dp[row * n + lane] : 0.f;
    const float dot = wave_sum(pv * dv);
    if (lane < n) dp[row * n + lane] = pv * (dv - dot);
}

static int launch_f32_gemm(int layout, const float* A, const float* Bm, float* C, int M, int N, int K, long lda, long ldb,
                           long ldc, int batch, long sA, long sB, long sC, float alpha, int flags, int split_k, void* stream) {
    hs_gemm_args g = {};
    g.A = A; g.B = Bm; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.batch = batch;
    g.strideA = sA; g.strideB = sB; g.strideC = sC; g.out_f32 = C; g.ldc = ldc; g.alpha = alpha; g.flags = flags;
    g.layout = layout; g.split_k = split_k;
    return hs_gemm(&g, HS_PREC_F32, stream);
}

// y[i] = x[i] * keep(i) / (1 - p): elementwise dropout whose mask is a pure function of (seed, offset, i) -- the backward pass
// applies the same call to the gradient (in place allowed)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p,
                                                      unsigned long long seed, unsigned long long offset) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;       // one Philox call = 4 consecutive elements
    if (g * 4 >= n) return;
    const Philox4 r = philox4x32_10(seed, offset + (unsigned long long)g);
    const unsigned u[4] = {r.x, r.y, r.z, r.w};
    const float inv = 1.0f / (1.0f - p);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (g * 4 + e < n) y[g * 4 + e] = (float)(u[e] >> 8) * (1.0f / 16777216.0f) >= p ? x[g * 4 + e] * inv : 0.f;
}
extern "C" int hs_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
    if (!x || !y) return HS_ERR_ARG;
    if (n < 1) return HS_ERR_SHAPE;
    if (!(p >= 0.f && p < 1.f)) return HS_ERR_ARG;
    ProfScope prof(HS_PROF_OTHER, stream, 0, (double)n * 8);
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, p,
                       (unsigned long long)seed, (unsigned long long)offset);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// K5 core as exact-f32 MFMA GEMMs + a 32-wide row softmax (tiny FLOPs: 4*nq*nk*d per volume).
// pd (optional, with drop_p > 0): receives dropout(P) -- the map the reference returns in train mode -- and out = pd V.
extern "C" int hs_cross_attn_fwd(const float* q, const float* k, const float* v, float* out, float* p, float* pd, int B, int nq,
                                 int nk, int d, float drop_p, uint64_t seed, uint64_t offset, void* stream) {
    if (!q || !k || !v || !out || !p) return HS_ERR_ARG;
    if (B < 1 || nq < 1 || nk < 4 || nk > 64 || (nk & 3) || (d & 3)) return HS_ERR_SHAPE;
    if (!(drop_p >= 0.f && drop_p < 1.f) || (drop_p > 0.f && !pd)) return HS_ERR_ARG;
    const float alpha = 1.0f / sqrtf((float)d);
    int rc = launch_f32_gemm(HS_GEMM_NT, q, k, p, nq, nk, d, d, d, nk, B, (long)nq * d, (long)nk * d, (long)nq * nk, alpha, 0, 1, stream);
    if (rc) return rc;
    const long rows = (long)B * nq;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, rows, nk);
    HS_CHECK_LAUNCH();
    const float* pu = p;
    if (drop_p > 0.f) {
        if ((rc = hs_dropout(p, pd, rows * nk, drop_p, seed, offset, stream))) return rc;
        pu = pd;
    }
    // out = P V : A = P [nq][nk], B = V stored [K=nk][N=d]
    return launch_f32_gemm(HS_GEMM_NN, pu, v, out, nq, d, nk, nk, d, d, B, (long)nq * nk, (long)nk * d, (long)nq * d, 1.0f, 0, 1, stream);
}
// ws: fp32 scratch [B][nq][nk].  dq written; dk, dv written (not accumulated).
// pd / drop_p / seed / offset: as given to the forward call (pd = NULL, drop_p = 0 in eval mode)
extern "C" int hs_cross_attn_bwd(const float* q, const float* k, const float* v, const float* p, const float* pd, const float* dout,
                                 float* dq, float* dk, float* dv, float* ws, int B, int nq, int nk, int d, float drop_p, uint64_t seed,
                                 uint64_t offset, void* stream) {
    if (!q || !k || !v || !p || !dout || !dq || !dk || !dv || !ws) return HS_ERR_ARG;
    if (!(drop_p >= 0.f && drop_p < 1.f) || (drop_p > 0.f && !pd)) return HS_ERR_ARG;
    if (B < 1 || nq < 4 || (nq & 3) || nk < 4 || nk > 64 || (nk & 3) || (d & 3)) return HS_ERR_SHAPE;
    const float alpha = 1.0f / sqrtf((float)d);
    const long sq = (long)nq * d, sk = (long)nk * d, sp = (long)nq * nk;
    // dP = dOut V^T
    int rc = launch_f32_gemm(HS_GEMM_NT, dout, v, ws, nq, nk, d, d, d, nk, B, sq, sk, sp, 1.0f, 0, 1, stream);
    if (rc) return rc;
    const long rows = (long)B * nq;
    if (drop_p > 0.f && (rc = hs_dropout(ws, ws, rows * nk, drop_p, seed, offset, stream))) return rc;     // dP = mask(dP_dropped)
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, ws, rows, nk);
    HS_CHECK_LAUNCH();
    // dQ = dS K * alpha ;  dK = dS^T Q * alpha ; dV = P^T dOut
    if ((rc = launch_f32_gemm(HS_GEMM_NN, ws, k, dq, nq, d, nk, nk, d, d, B, sp, sk, sq, alpha, 0, 1, stream))) return rc;
    // dK, dV: [nk][d] outputs (32 x 768: 12 tiles per volume) reduced over the nq = 4096 queries.  With few volumes the tiles cannot fill
    // the chip and each walks 128 K-steps (~100 us of latency at one volume): split the reduction, accumulate atomically into zeroed outputs
    const long tiles = (long)B * ((nk + 63) / 64) * ((d + 63) / 64);
    int split = (int)(256 / tiles);
    split = split < 2 ? 1 : split > 16 ? 16 : split;
    if (split > 1 && (nq >> 5) / split < 4) split = 1;
    int fl = 0;
    if (split > 1) {
        fl = HS_EPI_ATOMIC;
        if (hipMemsetAsync(dk, 0, (size_t)B * sk * 4, (hipStream_t)stream) != hipSuccess) return HS_ERR_LAUNCH;
        if (hipMemsetAsync(dv, 0, (size_t)B * sk * 4, (hipStream_t)stream) != hipSuccess) return HS_ERR_LAUNCH;
    }
    if ((rc = launch_f32_gemm(HS_GEMM_TN, ws, q, dk, nk, d, nq, nk, d, d, B, sp, sq, sk, alpha, fl, split, stream))) return rc;
    return launch_f32_gemm(HS_GEMM_TN, drop_p > 0.f ? pd : p, dout, dv, nk, d, nq, nk, d, d, B, sp, sq, sk, 1.0f, fl, split, stream);
}

// patch_score_proj + Sigmoid (vit.py:338-339): one wave per row
__global__ __launch_bounds__(256) void rowdot_sigmoid_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ b, float* __restrict__ score, int rows,
                                                                 int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int ch = lane; ch < (d >> 2); ch += 64) {
        const f32x4 a = *(const f32x4*)(x + (long)row * d + ch * 4);
        const f32x4 ww = *(const f32x4*)(w + ch * 4);
        s += a[0] * ww[0] + a[1] * ww[1] + a[2] * ww[2] + a[3] * ww[3];
    }
    s = wave_sum(s) + b[0];
    if (lane == 0) score[row] = 1.0f / (1.0f + __expf(-s));
}
#define RD_MAXCH 4
__global__ __launch_bounds__(256) void rowdot_sigmoid_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ score, const float* __restrict__ dscore,
                                                                 float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                                                 int rows, int d) {
    __shared__ float red[4][RD_MAXCH * 256];
    __shared__ float redb[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = d >> 2;
    f32x4 wv[RD_MAXCH], acc[RD_MAXCH];
#pragma unroll
    for (int i = 0; i < RD_MAXCH; ++i) {
        const int ch = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[i][e] = 0.f; wv[i][e] = 0.f; }
        if (ch < nch) wv[i] = *(const f32x4*)(w + ch * 4);
    }
    float bacc = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float s = score[row];
        const float ds = dscore[row] * s * (1.0f - s);
        bacc += ds;
#pragma unroll
        for (int i = 0; i < RD_MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                const f32x4 xv = *(const f32x4*)(x + (long)row * d + ch * 4);
                acc[i] += xv * ds;
                if (dx) *(f32x4*)(dx + (long)row * d + ch * 4) = wv[i] * ds;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RD_MAXCH; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][(lane + 64 * i) * 4 + e] = acc[i][e];
    if (lane == 0) redb[wave] = bacc;
    __syncthreads();
    for (int col = threadIdx.x; col < d; col += 256)
        if (dw) atomicAdd(&dw[col], red[0][col] + red[1][col] + red[2][col] + red[3][col]);
    if (threadIdx.x == 0 && db) atomicAdd(db, redb[0] + redb[1] + redb[2] + redb[3]);
}
extern "C" int hs_rowdot_sigmoid_fwd(const float* x, const float* w, const float* b, float* score, int rows, int d, void* stream) {
    if (!x || !w || !b || !score) return HS_ERR_ARG;
    if (rows < 1 || d < 4 || (d & 3)) return HS_ERR_SHAPE;
    hipLaunchKernelGGL(rowdot_sigmoid_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, b, score, rows, d);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
extern "C" int hs_rowdot_sigmoid_bwd(const float* x, const float* w, const float* score, const float* dscore, float* dx,
                                     float* dw, float* db, int rows, int d, void* stream) {
    if (!x || !w || !score || !dscore) return HS_ERR_ARG;
    if (rows < 1 || d < 4 || (d & 3) || d > RD_MAXCH * 256) return HS_ERR_SHAPE;
    int blocks = (rows + 3) / 4;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(rowdot_sigmoid_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, w, score, dscore, dx, dw, db,
                       rows, d);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// ------------------------------------------------------------------------------------------------------
// K6 Spatial Packer window step (spatial_pooling_projector.py:62-76,138-141)
// ------------------------------------------------------------------------------------------------------
struct Win { int gd, gh, gw, kd, kh, kw, cd, ch, cw, nwin, cells, tokens; };
__host__ __device__ inline Win make_win(int gd, int gh, int gw, int kd, int kh, int kw) {
    Win w;
    w.gd = gd; w.gh = gh; w.gw = gw; w.kd = kd; w.kh = kh; w.kw = kw;
    w.cd = gd / kd; w.ch = gh / kh; w.cw = gw / kw;
    w.nwin = kd * kh * kw; w.cells = w.cd * w.ch * w.cw; w.tokens = gd * gh * gw;
    return w;
}
// token index of child j of cell c (children beyond the pooled region do not exist: avg_pool3d floors)
__device__ __forceinline__ int child_token(const Win& w, int cell, int j) {
    const int cw_ = cell % w.cw, ch_ = (cell / w.cw) % w.ch, cd_ = cell / (w.cw * w.ch);
    const int jw = j % w.kw, jh = (j / w.kw) % w.kh, jd = j / (w.kw * w.kh);
    return ((cd_ * w.kd + jd) * w.gh + (ch_ * w.kh + jh)) * w.gw + (cw_ * w.kw + jw);
}

__global__ __launch_bounds__(256) void pool3d_fwd_kernel(const float* __restrict__ hr, float* __restrict__ lr, int B, Win w, int d) {
    const int d4 = d >> 2;
    const long total = (long)B * w.cells * d4;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int chn = (int)(g % d4);
    const long bc = g / d4;
    const int cell = (int)(bc % w.cells), b = (int)(bc / w.cells);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < w.nwin; ++j) acc += *(const f32x4*)(hr + ((long)b * w.tokens + child_token(w, cell, j)) * d + chn * 4);
    *(f32x4*)(lr + bc * d + chn * 4) = acc * (1.0f / w.nwin);
}
__global__ __launch_bounds__(256) void pool3d_bwd_kernel(const float* __restrict__ dlr, float* __restrict__ dhr, int B, Win w, int d,
                                                         int accumulate) {
    const int d4 = d >> 2;
    const long total = (long)B * w.tokens * d4;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int chn = (int)(g % d4);
    const long bt = g / d4;
    const int tok = (int)(bt % w.tokens), b = (int)(bt / w.tokens);
    const int tw = tok % w.gw, th = (tok / w.gw) % w.gh, td = tok / (w.gw * w.gh);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int cd_ = td / w.kd, ch_ = th / w.kh, cw_ = tw / w.kw;
    if (cd_ < w.cd && ch_ < w.ch && cw_ < w.cw) {
        const int cell = (cd_ * w.ch + ch_) * w.cw + cw_;
        v = *(const f32x4*)(dlr + ((long)b * w.cells + cell) * d + chn * 4) * (1.0f / w.nwin);
    }
    float* o = dhr + bt * d + chn * 4;
    if (accumulate) v += *(const f32x4*)o;
    *(f32x4*)o = v;
}

#define WA_MAXCH 4
// one wave per (b, cell): 1 query vs nwin (<=64) children, single head of width d
template <bool BWD>
__global__ __launch_bounds__(256) void window_attn_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                          const float* __restrict__ dout, float* __restrict__ out,
                                                          float* __restrict__ dq, float* __restrict__ dkv, int B, Win w, int d,
                                                          float drop_p, unsigned long long seed, unsigned long long offset) {
    const int lane = threadIdx.x & 63;
    const long bc = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bc >= (long)B * w.cells) return;
    const int cell = (int)(bc % w.cells), b = (int)(bc / w.cells);
    const int nch = d >> 2;
    const float alpha = rsqrtf((float)d);
    f32x4 qv[WA_MAXCH], gv[WA_MAXCH];
#pragma unroll
    for (int i = 0; i < WA_MAXCH; ++i) {
        const int ch = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) { qv[i][e] = 0.f; gv[i][e] = 0.f; }
        if (ch < nch) {
            qv[i] = *(const f32x4*)(q + bc * d + ch * 4);
            if (BWD) gv[i] = *(const f32x4*)(dout + bc * d + ch * 4);
        }
    }
    const float* kvb = kv + (long)b * w.tokens * 2 * d;
    float sc = -__builtin_inff(), dpl = 0.f;
    // children four at a time: the 4 x (K row [+ V row]) loads are issued together and the four wave reductions interleave -- one child
    // per iteration was a chain of nwin dependent (load -> reduce) steps, ~95 us for the 32 children of a (2,4,4) window on the 32 blocks
    // a single volume launches (pure latency)
    for (int j0 = 0; j0 < w.nwin; j0 += 4) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f}, t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + u < w.nwin) {
                const float* kr = kvb + (long)child_token(w, cell, j0 + u) * 2 * d;
#pragma unroll
                for (int i = 0; i < WA_MAXCH; ++i) {
                    const int ch = lane + 64 * i;
                    if (ch < nch) {
                        const f32x4 kk = *(const f32x4*)(kr + ch * 4);
                        s4[u] += qv[i][0] * kk[0] + qv[i][1] * kk[1] + qv[i][2] * kk[2] + qv[i][3] * kk[3];
                        if (BWD) {
                            const f32x4 vv = *(const f32x4*)(kr + d + ch * 4);
                            t4[u] += gv[i][0] * vv[0] + gv[i][1] * vv[1] + gv[i][2] * vv[2] + gv[i][3] * vv[3];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float s = wave_sum(s4[u]) * alpha;
            const float t = BWD ? wave_sum(t4[u]) : 0.f;
            if (lane == j0 + u) { sc = s; dpl = t; }
        }
    }
    const float m = wave_max(sc);
    const float e = lane < w.nwin ? __expf(sc - m) : 0.f;
    const float p = e / wave_sum(e);
    // train mode: dropout on the attention probabilities (spatial_pooling_projector.py:13-14); lane j = child j
    const float mk = drop_p > 0.f ? dropout_scale(seed, offset, (unsigned long long)bc * 64 + lane, drop_p) : 1.0f;
    if (BWD) dpl *= mk;                 // dP = dP_dropped * mask / (1 - p)
    if (!BWD) {
        f32x4 acc[WA_MAXCH];
#pragma unroll
        for (int i = 0; i < WA_MAXCH; ++i)
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) acc[i][e2] = 0.f;
#pragma unroll 4
        for (int j = 0; j < w.nwin; ++j) {
            const float pj = __shfl(p * mk, j, 64);
            const float* vr = kvb + (long)child_token(w, cell, j) * 2 * d + d;
#pragma unroll
            for (int i = 0; i < WA_MAXCH; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nch) acc[i] += *(const f32x4*)(vr + ch * 4) * pj;
            }
        }
#pragma unroll
        for (int i = 0; i < WA_MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) *(f32x4*)(out + bc * d + ch * 4) = acc[i];
        }
    } else {
        const float dot = wave_sum(p * dpl);
        const float ds = p * (dpl - dot);
        f32x4 acc[WA_MAXCH];
#pragma unroll
        for (int i = 0; i < WA_MAXCH; ++i)
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) acc[i][e2] = 0.f;
        float* dkvb = dkv + (long)b * w.tokens * 2 * d;
#pragma unroll 4
        for (int j = 0; j < w.nwin; ++j) {
            const float dsj = __shfl(ds, j, 64) * alpha, pj = __shfl(p * mk, j, 64);
            const long tok = child_token(w, cell, j);
            const float* kr = kvb + tok * 2 * d;
            float* dkr = dkvb + tok * 2 * d;
#pragma unroll
            for (int i = 0; i < WA_MAXCH; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nch) {
                    acc[i] += *(const f32x4*)(kr + ch * 4) * dsj;
                    *(f32x4*)(dkr + ch * 4) = qv[i] * dsj;
                    *(f32x4*)(dkr + d + ch * 4) = gv[i] * pj;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WA_MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) *(f32x4*)(dq + bc * d + ch * 4) = acc[i];
        }
    }
}

static int win_check(int B, int gd, int gh, int gw, int kd, int kh, int kw, int d) {
    if (B < 1 || gd < 1 || gh < 1 || gw < 1 || kd < 1 || kh < 1 || kw < 1 || d < 4 || (d & 3)) return HS_ERR_SHAPE;
    if (kd > gd || kh > gh || kw > gw || kd * kh * kw > 64) return HS_ERR_SHAPE;
    return HS_OK;
}
extern "C" int hs_pool3d_fwd(const float* hr, float* lr, int B, int gd, int gh, int gw, int kd, int kh, int kw, int d, void* stream) {
    if (!hr || !lr) return HS_ERR_ARG;
    int rc = win_check(B, gd, gh, gw, kd, kh, kw, d);
    if (rc) return rc;
    const Win w = make_win(gd, gh, gw, kd, kh, kw);
    const long total = (long)B * w.cells * (d / 4);
    hipLaunchKernelGGL(pool3d_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, hr, lr, B, w, d);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
extern "C" int hs_pool3d_bwd(const float* dlr, float* dhr, int B, int gd, int gh, int gw, int kd, int kh, int kw, int d,
                             int accumulate, void* stream) {
    if (!dlr || !dhr) return HS_ERR_ARG;
    int rc = win_check(B, gd, gh, gw, kd, kh, kw, d);
    if (rc) return rc;
    const Win w = make_win(gd, gh, gw, kd, kh, kw);
    const long total = (long)B * w.tokens * (d / 4);
    hipLaunchKernelGGL(pool3d_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dlr, dhr, B, w, d,
                       accumulate);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
extern "C" int hs_window_attn_fwd(const float* q, const float* kv, float* out, int B, int gd, int gh, int gw, int kd, int kh,
                                  int kw, int d, float drop_p, uint64_t seed, uint64_t offset, void* stream) {
    if (!q || !kv || !out) return HS_ERR_ARG;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return HS_ERR_ARG;
    int rc = win_check(B, gd, gh, gw, kd, kh, kw, d);
    if (rc) return rc;
    if (d > WA_MAXCH * 256) return HS_ERR_SHAPE;
    const Win w = make_win(gd, gh, gw, kd, kh, kw);
    const long waves = (long)B * w.cells;
    hipLaunchKernelGGL(window_attn_kernel<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, q, kv,
                       (const float*)nullptr, out, (float*)nullptr, (float*)nullptr, B, w, d, drop_p, (unsigned long long)seed,
                       (unsigned long long)offset);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
// dkv rows of tokens outside every window (grid not divisible by the kernel) are NOT written: caller zeroes dkv then.
extern "C" int hs_window_attn_bwd(const float* q, const float* kv, const float* dout, float* dq, float* dkv, int B, int gd,
                                  int gh, int gw, int kd, int kh, int kw, int d, float drop_p, uint64_t seed, uint64_t offset,
                                  void* stream) {
    if (!q || !kv || !dout || !dq || !dkv) return HS_ERR_ARG;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return HS_ERR_ARG;
    int rc = win_check(B, gd, gh, gw, kd, kh, kw, d);
    if (rc) return rc;
    if (d > WA_MAXCH * 256) return HS_ERR_SHAPE;
    const Win w = make_win(gd, gh, gw, kd, kh, kw);
    const long waves = (long)B * w.cells;
    hipLaunchKernelGGL(window_attn_kernel<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, q, kv, dout,
                       (float*)nullptr, dq, dkv, B, w, d, drop_p, (unsigned long long)seed, (unsigned long long)offset);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// ------------------------------------------------------------------------------------------------------
// K7/K8
// ------------------------------------------------------------------------------------------------------
// out[n] += sum_m x[m][n].  Block: 32 column groups (16 B each) x 8 row lanes; grid (col strips, row chunks).
template <int FMT>       // 0 fp32, 1 bf16, 2 IEEE fp16
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ x, float* __restrict__ out, int M, int N, long ldx,
                                                     int rows_per_block) {
    constexpr bool BF16 = FMT != 0;
    constexpr int V = BF16 ? 8 : 4;
    __shared__ float red[8][32 * V];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col0 = (blockIdx.x * 32 + cg) * V;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    if (col0 < N) {
        for (int r = r0 + rl; r < r1; r += 8) {
            if (BF16) {
                const u32x4 v = *(const u32x4*)((const bf16_t*)x + (long)r * ldx + col0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[2 * i] += unpack_lo<FMT == 2>(v[i]);
                    acc[2 * i + 1] += unpack_hi<FMT == 2>(v[i]);
                }
            } else {
                const f32x4 v = *(const f32x4*)((const float*)x + (long)r * ldx + col0);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += v[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) red[rl][cg * V + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 32 * V) {
        const int col = blockIdx.x * 32 * V + threadIdx.x;
        if (col < N) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += red[r][threadIdx.x];
            atomicAdd(&out[col], s);
        }
    }
}
extern "C" int hs_colsum(const void* x, int is_bf16, float* out, int M, int N, int64_t ldx, void* stream) {
    if (!x || !out) return HS_ERR_ARG;
    if (is_bf16 < 0 || is_bf16 > 2) return HS_ERR_ARG;        // 0 fp32, 1 bf16, 2 fp16
    const int V = is_bf16 ? 8 : 4;
    if (M < 1 || N < V || (N % V) || (ldx % V) || ((uintptr_t)x & 15)) return HS_ERR_SHAPE;
    // rows per block: 512 for the long matrices (81 940+ rows: hundreds of blocks either way); a one-volume matrix (4096 rows) had 8 row
    // blocks x 3 column blocks = 24 blocks on 256 CUs and took 15-40 us of pure latency -- at least ~64 row blocks now
    int rpb = M / 64;
    rpb = rpb < 32 ? 32 : rpb > 512 ? 512 : (rpb + 7) & ~7;
    dim3 grid((N + 32 * V - 1) / (32 * V), (M + rpb - 1) / rpb);
    ProfScope prof(HS_PROF_COLSUM, stream, 0, (double)M * N * (is_bf16 ? 2 : 4));
    if (is_bf16 == 2) hipLaunchKernelGGL(colsum_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, out, M, N, (long)ldx, rpb);
    else if (is_bf16) hipLaunchKernelGGL(colsum_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, out, M, N, (long)ldx, rpb);
    else hipLaunchKernelGGL(colsum_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, x, out, M, N, (long)ldx, rpb);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// out16 = cast(a (+ b)) and column sums of the ROUNDED values in one pass: the backward of an nn.Linear wants its output gradient as a
// 16-bit GEMM operand AND its column sums (the bias gradient) -- as two kernels (cast, then hs_colsum re-reading the copy) that was two
// dependent launches per Linear, 16 per step in the embedding / packer backward (1.6 % of a one-volume step).  Columns [0, nsplit) add
// into cs_lo, columns [nsplit, N) into cs_hi - nsplit (the packer's fused [Wk; Wv] projection has two bias vectors).  Internal (path.hip).
template <bool F16>
__global__ __launch_bounds__(256) void addcast_colsum_kernel(const float* __restrict__ a, const float* __restrict__ b, bf16_t* __restrict__ y,
                                                             float* __restrict__ cs_lo, float* __restrict__ cs_hi, int M, int N, int nsplit,
                                                             int rows_per_block) {
    __shared__ float red[8][256];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col0 = (blockIdx.x * 32 + cg) * 8;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (col0 < N) {
        for (int r = r0 + rl; r < r1; r += 8) {
            const long ix = (long)r * N + col0;
            f32x4 v0 = *(const f32x4*)(a + ix), v1 = *(const f32x4*)(a + ix + 4);
            if (b) { v0 += *(const f32x4*)(b + ix); v1 += *(const f32x4*)(b + ix + 4); }
            u32x4 pk;
            pk[0] = pack2h<F16>(v0[0], v0[1]); pk[1] = pack2h<F16>(v0[2], v0[3]); pk[2] = pack2h<F16>(v1[0], v1[1]); pk[3] = pack2h<F16>(v1[2], v1[3]);
            *(u32x4*)(y + ix) = pk;
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[2 * i] += unpack_lo<F16>(pk[i]); acc[2 * i + 1] += unpack_hi<F16>(pk[i]); }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
    __syncthreads();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col < N) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += red[r][threadIdx.x];
        atomicAdd(col < nsplit ? &cs_lo[col] : &cs_hi[col - nsplit], s);
    }
}
int hs_addcast_colsum(const float* a, const float* b, void* y16, float* cs_lo, float* cs_hi, int M, int N, int nsplit, int f16, void* stream) {
    if (!a || !y16 || !cs_lo) return HS_ERR_ARG;
    if (M < 1 || N < 8 || (N & 7) || nsplit < 0 || nsplit > N || (nsplit < N && !cs_hi)) return HS_ERR_SHAPE;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y16) & 15) return HS_ERR_SHAPE;
    int rpb = M / 64;
    rpb = rpb < 32 ? 32 : rpb > 512 ? 512 : (rpb + 7) & ~7;
    dim3 grid((N + 255) / 256, (M + rpb - 1) / rpb);
    ProfScope prof(HS_PROF_CAST, stream, 0, (double)M * N * (b ? 10 : 6));
    if (f16) hipLaunchKernelGGL(addcast_colsum_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, b, (bf16_t*)y16, cs_lo, cs_hi, M, N, nsplit, rpb);
    else hipLaunchKernelGGL(addcast_colsum_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, b, (bf16_t*)y16, cs_lo, cs_hi, M, N, nsplit, rpb);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

template <bool F16>
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const f32x4 a = *(const f32x4*)(x + i), c = *(const f32x4*)(x + i + 4);
        u32x4 pk;
        pk[0] = pack2h<F16>(a[0], a[1]); pk[1] = pack2h<F16>(a[2], a[3]); pk[2] = pack2h<F16>(c[0], c[1]); pk[3] = pack2h<F16>(c[2], c[3]);
        *(u32x4*)(y + i) = pk;
    } else {
        for (long j = i; j < n; ++j) y[j] = f2h<F16>(x[j]);
    }
}
__global__ __launch_bounds__(256) void cast_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const u32x4 v = *(const u32x4*)(x + i);
        f32x4 a, c;
        a[0] = __uint_as_float(v[0] << 16); a[1] = __uint_as_float(v[0] & 0xffff0000u);
        a[2] = __uint_as_float(v[1] << 16); a[3] = __uint_as_float(v[1] & 0xffff0000u);
        c[0] = __uint_as_float(v[2] << 16); c[1] = __uint_as_float(v[2] & 0xffff0000u);
        c[2] = __uint_as_float(v[3] << 16); c[3] = __uint_as_float(v[3] & 0xffff0000u);
        *(f32x4*)(y + i) = a;
        *(f32x4*)(y + i + 4) = c;
    } else {
        for (long j = i; j < n; ++j) y[j] = bf2f(x[j]);
    }
}
extern "C" int hs_cast_bf16(const float* x, void* y, int64_t n, void* stream) {
    if (!x || !y) return HS_ERR_ARG;
    if (n < 1 || (((uintptr_t)x | (uintptr_t)y) & 15)) return HS_ERR_SHAPE;
    ProfScope prof(HS_PROF_CAST, stream, 0, (double)n * 6);
    hipLaunchKernelGGL(cast_bf16_kernel<false>, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, (long)n);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
// fp32 -> IEEE fp16, round to nearest even (the operand cast of HS_PREC_F16; = torch's .half())
extern "C" int hs_cast_f16(const float* x, void* y, int64_t n, void* stream) {
    if (!x || !y) return HS_ERR_ARG;
    if (n < 1 || (((uintptr_t)x | (uintptr_t)y) & 15)) return HS_ERR_SHAPE;
    ProfScope prof(HS_PROF_CAST, stream, 0, (double)n * 6);
    hipLaunchKernelGGL(cast_bf16_kernel<true>, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, (long)n);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
// up to 64 fp32 -> bf16 segments in ONE launch (the fused encoder driver re-casts 4 weight matrices per layer at the start of every
// forward pass: 48 launches of ~7 us were 0.36 ms per step -- 3.5 % of a one-volume step).  Internal to the library (vit.hip).
struct CastSegs { const float* src[64]; bf16_t* dst[64]; long n[64]; };
template <bool F16>
__global__ __launch_bounds__(256) void cast_bf16_multi_kernel(CastSegs g) {
    const int sgm = blockIdx.y;
    const float* __restrict__ x = g.src[sgm];
    bf16_t* __restrict__ y = g.dst[sgm];
    const long n = g.n[sgm];
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long)gridDim.x * 2048) {
        if (i + 8 <= n) {
            const f32x4 a = *(const f32x4*)(x + i), c = *(const f32x4*)(x + i + 4);
            u32x4 pk;
            pk[0] = pack2h<F16>(a[0], a[1]); pk[1] = pack2h<F16>(a[2], a[3]); pk[2] = pack2h<F16>(c[0], c[1]); pk[3] = pack2h<F16>(c[2], c[3]);
            *(u32x4*)(y + i) = pk;
        } else {
            for (long j = i; j < n; ++j) y[j] = f2h<F16>(x[j]);
        }
    }
}
int hs_cast_bf16_multi(const float* const* src, void* const* dst, const int64_t* n, int count, void* stream, int f16) {
    if (!src || !dst || !n || count < 1) return HS_ERR_ARG;
    for (int c0 = 0; c0 < count; c0 += 64) {
        CastSegs g;
        const int cnt = count - c0 < 64 ? count - c0 : 64;
        long nmax = 0;
        double bytes = 0;
        for (int i = 0; i < cnt; ++i) {
            if (!src[c0 + i] || !dst[c0 + i] || n[c0 + i] < 1 || (((uintptr_t)src[c0 + i] | (uintptr_t)dst[c0 + i]) & 15)) return HS_ERR_SHAPE;
            g.src[i] = src[c0 + i]; g.dst[i] = (bf16_t*)dst[c0 + i]; g.n[i] = (long)n[c0 + i];
            nmax = g.n[i] > nmax ? g.n[i] : nmax;
            bytes += (double)n[c0 + i] * 6;
        }
        for (int i = cnt; i < 64; ++i) { g.src[i] = nullptr; g.dst[i] = nullptr; g.n[i] = 0; }
        long bx = (nmax + 2047) / 2048;
        if (bx > 512) bx = 512;                       // grid-stride: 512 x cnt blocks already saturate the memory system
        ProfScope prof(HS_PROF_CAST, stream, 0, bytes);
        if (f16) hipLaunchKernelGGL(cast_bf16_multi_kernel<true>, dim3((unsigned)bx, (unsigned)cnt), dim3(256), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL(cast_bf16_multi_kernel<false>, dim3((unsigned)bx, (unsigned)cnt), dim3(256), 0, (hipStream_t)stream, g);
        HS_CHECK_LAUNCH();
    }
    return HS_OK;
}
// fp32 [R][C] -> bf16 [C][R]: the TRANSPOSED bf16 weight copies the backward pass multiplies the output gradients with.  A dgrad
// dX = dY W reads W [out][in] with the reduction index (out) as the memory row -- the NN layout, transposed LDS reads, the slower
// GEMM kernel; against W^T [in][out] the same product is an NT GEMM and runs on the forward's kernel.  Up to 64 matrices per
// launch, 64 x 64 tiles through LDS, R and C multiples of 64.  Internal to the library (vit.hip); rounding = hs_cast_bf16's.
struct CastTSegs { const void* src[64]; bf16_t* dst[64]; int R[64], C[64]; int first[65]; };
// SRC16: the source matrices are already 16-bit (the straight copies an optimizer pass wrote): bits pass through, 4 B per element moved
template <bool F16, bool SRC16>
__global__ __launch_bounds__(256) void cast_bf16_transpose_multi_kernel(CastTSegs g) {
    __shared__ __attribute__((aligned(16))) bf16_t t[64][64 + 8];          // 144-B rows: 16-B aligned reads
    int b = blockIdx.x, sgm = 0;
    while (b >= g.first[sgm + 1]) ++sgm;                                    // (block-uniform; <= 64 segments)
    b -= g.first[sgm];
    const int R = g.R[sgm], C = g.C[sgm], tc = C >> 6;
    const int r0 = (b / tc) << 6, c0 = (b % tc) << 6;
    bf16_t* __restrict__ y = g.dst[sgm];
    const int tr = threadIdx.x >> 2, tb = (threadIdx.x & 3) << 4;
    if (SRC16) {
        const bf16_t* __restrict__ x = (const bf16_t*)g.src[sgm];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const u32x4 v = *(const u32x4*)(x + (long)(r0 + tr) * C + c0 + tb + 8 * k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                t[tb + 8 * k + 2 * e][tr] = (bf16_t)(v[e] & 0xffffu);
                t[tb + 8 * k + 2 * e + 1][tr] = (bf16_t)(v[e] >> 16);
            }
        }
    } else {
        const float* __restrict__ x = (const float*)g.src[sgm];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 v = *(const f32x4*)(x + (long)(r0 + tr) * C + c0 + tb + 4 * k);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[tb + 4 * k + e][tr] = f2h<F16>(v[e]);
        }
    }
    __syncthreads();
    const u32x4 lo = *(const u32x4*)&t[tr][tb], hi = *(const u32x4*)&t[tr][tb + 8];
    *(u32x4*)(y + (long)(c0 + tr) * R + r0 + tb) = lo;
    *(u32x4*)(y + (long)(c0 + tr) * R + r0 + tb + 8) = hi;
}
int hs_transpose16_multi(const void* const* src, int src_is_16, void* const* dst, const int* R, const int* C, int count, void* stream, int f16) {
    if (!src || !dst || !R || !C || count < 1) return HS_ERR_ARG;
    for (int c0 = 0; c0 < count; c0 += 64) {
        CastTSegs g;
        const int cnt = count - c0 < 64 ? count - c0 : 64;
        double bytes = 0;
        g.first[0] = 0;
        for (int i = 0; i < cnt; ++i) {
            const int r = R[c0 + i], c = C[c0 + i];
            if (!src[c0 + i] || !dst[c0 + i] || r < 64 || c < 64 || (r & 63) || (c & 63)) return HS_ERR_SHAPE;
            if (((uintptr_t)src[c0 + i] | (uintptr_t)dst[c0 + i]) & 15) return HS_ERR_SHAPE;
            g.src[i] = src[c0 + i]; g.dst[i] = (bf16_t*)dst[c0 + i]; g.R[i] = r; g.C[i] = c;
            g.first[i + 1] = g.first[i] + (r >> 6) * (c >> 6);
            bytes += (double)r * c * (src_is_16 ? 4 : 6);
        }
        for (int i = cnt; i < 64; ++i) { g.src[i] = nullptr; g.dst[i] = nullptr; g.R[i] = 0; g.C[i] = 0; g.first[i + 1] = g.first[cnt]; }
        ProfScope prof(HS_PROF_CAST, stream, 0, bytes);
        const dim3 grid((unsigned)g.first[cnt]), blk(256);
        if (src_is_16) hipLaunchKernelGGL((cast_bf16_transpose_multi_kernel<false, true>), grid, blk, 0, (hipStream_t)stream, g);
        else if (f16) hipLaunchKernelGGL((cast_bf16_transpose_multi_kernel<true, false>), grid, blk, 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((cast_bf16_transpose_multi_kernel<false, false>), grid, blk, 0, (hipStream_t)stream, g);
        HS_CHECK_LAUNCH();
    }
    return HS_OK;
}
extern "C" int hs_cast_f32(const void* x, float* y, int64_t n, void* stream) {
    if (!x || !y) return HS_ERR_ARG;
    if (n < 1 || (((uintptr_t)x | (uintptr_t)y) & 15)) return HS_ERR_SHAPE;
    hipLaunchKernelGGL(cast_f32_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, (long)n);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    y[i] = dy ? dy[i] * gelu_grad_f(x[i]) : gelu_f(x[i]);
}
extern "C" int hs_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
    if (!x || !y) return HS_ERR_ARG;
    if (n < 1) return HS_ERR_SHAPE;
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, y, (long)n);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
extern "C" int hs_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
    if (!x || !dy || !dx) return HS_ERR_ARG;
    if (n < 1) return HS_ERR_SHAPE;
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, (long)n);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// F.normalize(dim=-1): one wave per row
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ y,
                                                     int rows, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float ss = 0.f, dot = 0.f;
    for (int i = lane; i < d; i += 64) {
        const float v = x[(long)row * d + i];
        ss += v * v;
        if (dy) dot += v * dy[(long)row * d + i];
    }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    const float inv = 1.0f / fmaxf(nrm, eps);
    if (!dy) {
        for (int i = lane; i < d; i += 64) y[(long)row * d + i] = x[(long)row * d + i] * inv;
    } else {
        dot = wave_sum(dot);
        // y = x/n: dx = dy/n - x * <x,dy>/n^3   (clamped branch nrm<eps: dx = dy/eps)
        const float k = nrm > eps ? dot * inv * inv * inv : 0.f;
        for (int i = lane; i < d; i += 64) y[(long)row * d + i] = dy[(long)row * d + i] * inv - x[(long)row * d + i] * k;
    }
}
extern "C" int hs_l2norm_fwd(const float* x, float* y, int rows, int d, float eps, void* stream) {
    if (!x || !y) return HS_ERR_ARG;
    if (rows < 1 || d < 1) return HS_ERR_SHAPE;
    hipLaunchKernelGGL(l2norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, y, rows, d, eps);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
extern "C" int hs_l2norm_bwd(const float* x, const float* dy, float* dx, int rows, int d, float eps, void* stream) {
    if (!x || !dy || !dx) return HS_ERR_ARG;
    if (rows < 1 || d < 1) return HS_ERR_SHAPE;
    hipLaunchKernelGGL(l2norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dy, dx, rows, d, eps);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// torch.optim.AdamW semantics (decoupled weight decay, bias correction), one pass over a flat buffer
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                                                    float gs, float bc1, float bc2s) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const int cnt = (int)min((long)4, n - i);
    for (int e = 0; e < cnt; ++e) {
        const float gr = g[i + e] * gs;
        float pv = p[i + e] * (1.0f - lr * wd);
        const float mv = b1 * m[i + e] + (1.0f - b1) * gr;
        const float vv = b2 * v[i + e] + (1.0f - b2) * gr * gr;
        m[i + e] = mv;
        v[i + e] = vv;
        pv -= (lr / bc1) * mv / (sqrtf(vv) / bc2s + eps);
        p[i + e] = pv;
    }
}
extern "C" int hs_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, float grad_scale, int step, void* stream) {
    if (!p || !g || !m || !v) return HS_ERR_ARG;
    if (n < 1 || step < 1) return HS_ERR_SHAPE;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr,
                       beta1, beta2, eps, weight_decay, grad_scale, bc1, bc2s);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// ------------------------------------------------------------------------------------------------------
// N1: CLIP InfoNCE head (CLIP_stage1.py:141-155): loss = (CE(logits, arange) + CE(logits^T, arange)) / 2 on an
// n x n logits matrix, and dL/dlogits.  Tiny (n = global batch): one wave per row / per column, two launches.
// pass 0 (rows):  dlogits = softmax_row / (2n);              loss += sum_i (lse_row_i - logit_ii) / (2n)
// pass 1 (cols):  dlogits += softmax_col / (2n) - I / n;     loss += sum_j (lse_col_j - logit_jj) / (2n);
//                 loss[1] += sum(dlogits * logits)   (= logit_scale * dL/dlogit_scale)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void clip_ce_kernel(const float* __restrict__ logits, float* __restrict__ dlogits,
                                                      float* __restrict__ loss, int n, int pass) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const long stride = pass == 0 ? 1 : n, base = pass == 0 ? (long)i * n : i;
    float m = -__builtin_inff();
    for (int j = lane; j < n; j += 64) m = fmaxf(m, logits[base + j * stride]);
    m = wave_max(m);
    float z = 0.f;
    for (int j = lane; j < n; j += 64) z += __expf(logits[base + j * stride] - m);
    z = wave_sum(z);
    const float inv = 1.0f / z, w = 0.5f / n;
    float dot = 0.f;
    for (int j = lane; j < n; j += 64) {
        const float lg = logits[base + j * stride];
        const float pr = __expf(lg - m) * inv * w;
        float* o = dlogits + base + j * stride;
        if (pass == 0) {
            *o = pr;
        } else {
            const float dfin = *o + pr - (j == i ? 1.0f / n : 0.f);
            *o = dfin;
            dot += dfin * lg;        // sum(dlogits * logits) = logit_scale * dL/d(logit_scale)
        }
    }
    dot = wave_sum(dot);
    if (lane == 0) {
        atomicAdd(loss, (m + __logf(z) - logits[(long)i * n + i]) * w);
        if (pass == 1) atomicAdd(loss + 1, dot);
    }
}
extern "C" int hs_clip_ce(const float* logits, float* dlogits, float* loss, int n, void* stream) {
    if (!logits || !dlogits || !loss) return HS_ERR_ARG;
    if (n < 1) return HS_ERR_SHAPE;
    ProfScope prof(HS_PROF_OTHER, stream, 0, (double)n * n * 16);
    hipLaunchKernelGGL(clip_ce_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, dlogits, loss, n, 0);
    hipLaunchKernelGGL(clip_ce_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, dlogits, loss, n, 1);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, float* __restrict__ y, float alpha, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n) {
        *(f32x4*)(y + i) = *(const f32x4*)(x + i) * alpha;
    } else {
        for (long j = i; j < n; ++j) y[j] = x[j] * alpha;
    }
}
extern "C" int hs_scale(const float* x, float* y, float alpha, int64_t n, void* stream) {
    if (!x || !y) return HS_ERR_ARG;
    if (n < 1 || (((uintptr_t)x | (uintptr_t)y) & 15)) return HS_ERR_SHAPE;
    ProfScope prof(HS_PROF_OTHER, stream, 0, (double)n * 8);
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, x, y, alpha, (long)n);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
extern "C" int hs_prof_count(void) { return HS_PROF_COUNT; }

extern "C" int hs_version(void) { return HS_ABI_VERSION; }
extern "C" int hs_device_check(int dev) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return HS_ERR_ARCH;
    const char* a = prop.gcnArchName;
    return (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? HS_OK : HS_ERR_ARCH;
}
