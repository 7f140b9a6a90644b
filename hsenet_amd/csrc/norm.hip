// K2: LayerNorm forward / backward (nn.LayerNorm(d), eps=1e-5): MONAI TransformerBlock.norm1/norm2, the final
// `norm` (multimodal_encoder/vit.py:355,467) and the post-attention norms of regular_attention (vit.py:62) and
// resolution_attention_v3 (spatial_pooling_projector.py:82).  HBM-bound: one wavefront per row, the row lives
// in registers (d <= 1024), mean/variance by wave shuffles in fp32 (two-pass, like PyTorch's kernel), 16-B
// accesses.  The forward can fuse the residual add in front (LN(x + resid)) and emit bf16 for the next GEMM.
#include "common.h"
#include <stdlib.h>

#define LN_MAXCH 4  // float4 chunks per lane -> d <= 1024

template <bool F16>   // format of the 16-bit output: bf16, or fp16 (HS_PREC_F16 forward)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ resid,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, bf16_t* __restrict__ yb,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                            int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = d >> 2;
    const float* xr = x + (long)row * d;
    f32x4 v[LN_MAXCH];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            v[i] = *(const f32x4*)(xr + ch * 4);
            if (resid) {
                const f32x4 r = *(const f32x4*)(resid + (long)row * d + ch * 4);
                v[i] += r;
            }
            sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mu = wave_sum(sum) / d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[i][e] - mu;
                sq += t * t;
            }
        }
    }
    const float rs = rsqrtf(wave_sum(sq) / d + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            const f32x4 g = *(const f32x4*)(gamma + ch * 4);
            const f32x4 b = *(const f32x4*)(beta + ch * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + b[e];
            if (y) *(f32x4*)(y + (long)row * d + ch * 4) = o;
            if (yb) {
                u32x2 pk;
                pk[0] = pack2h<F16>(o[0], o[1]);
                pk[1] = pack2h<F16>(o[2], o[3]);
                *(u32x2*)(yb + (long)row * d + ch * 4) = pk;
            }
        }
    }
}

// The per-element arithmetic of the LayerNorm backward, ONE definition for both row loops of the kernel below (the hand-written
// one-row path and the row-pair lambdas): contraction is spelled out (explicit fma, pragma off for the rest) so that the two paths
// cannot drift apart by the compiler fusing differently in each -- tests/test_gpu_kernels.py compares their dx bit for bit.
__device__ __forceinline__ void ln_bwd_accum(float xv, float dv, float gam, float mu, float rs, float& xh, float& gy, float& s1,
                                             float& s2, float& dg, float& db) {
#pragma clang fp contract(off)
    xh = (xv - mu) * rs;
    gy = dv * gam;
    s1 += gy;
    s2 = __builtin_fmaf(gy, xh, s2);
    dg = __builtin_fmaf(dv, xh, dg);
    db += dv;
}
__device__ __forceinline__ float ln_bwd_out(float gy, float xh, float c1, float c2, float rs, float di) {
#pragma clang fp contract(off)
    const float t = __builtin_fmaf(-xh, c2, gy - c1);
    return __builtin_fmaf(rs, t, di);
}

// Each wave walks rows (grid-stride), keeps its dgamma/dbeta partials in registers, then the block reduces
// through LDS and issues one float atomic per column.
// (launch bounds: 4 blocks of 4 waves per CU, 3 for the d = 1024 instantiation whose row arrays need ~150 registers)
// NCH = 256-column chunks per row actually used (d <= NCH*256): sizes the register arrays and the LDS reduction
// buffer, which set the occupancy of this HBM-bound kernel (NCH=4 for every d cost 200 VGPRs / 48 KiB: 2 waves/SIMD).
// WPB = waves per block: 4, or 8 for small row counts (one volume per GPU: 4097 rows) where a wave's rows are a serial chain of
// load latencies -- 16 rows per block then take 2 steps per wave instead of 4 while the block's 3 x d reduction atomics stay the same.
// DYF: format of dy (0 fp32, 1 bf16, 2 IEEE fp16); OF16: the 16-bit copy of dx leaves as fp16 instead of bf16 (HS_PREC_F16 training)
template <int DYF, int NCH, int WPB = 4, bool OF16 = false>
__global__ __launch_bounds__(WPB * 64, (NCH >= 4 && WPB == 4) ? 3 : 16 / WPB) void layernorm_bwd_kernel(const void* __restrict__ dy_, const float* __restrict__ x,
                                                            const float* __restrict__ resid, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ dx_in, float* __restrict__ dx,
                                                            bf16_t* __restrict__ dxb, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dxsum, int rows, int d) {
    __shared__ float red[3][WPB][NCH * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = d >> 2;
    f32x4 g[NCH], dg[NCH], db[NCH], ds[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; g[i][e] = 0.f; ds[i][e] = 0.f; }
        if (ch < nch) g[i] = *(const f32x4*)(gamma + ch * 4);
    }
    // (WPB = 8) two rows per trip: both rows' operands are requested before either is reduced, so a wave's serial chain is half as
    // many memory round trips (it matters at 4097 rows, where a wave owns 4 rows; at 81 940 the kernel is bandwidth-bound either way)
    struct Row { f32x4 xh[NCH], gy[NCH], di[NCH]; float s1, s2, rs; };
    auto phase1 = [&](int row, Row& R) {
        const float mu = mean[row];
        R.rs = rstd[row];
        R.s1 = 0.f; R.s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                R.di[i] = dx_in ? *(const f32x4*)(dx_in + (long)row * d + ch * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 xv = *(const f32x4*)(x + (long)row * d + ch * 4);
                if (resid) xv += *(const f32x4*)(resid + (long)row * d + ch * 4);
                f32x4 dv;
                if (DYF) {
                    const u32x2 pk = *(const u32x2*)((const bf16_t*)dy_ + (long)row * d + ch * 4);
                    dv[0] = unpack_lo<DYF == 2>(pk[0]); dv[1] = unpack_hi<DYF == 2>(pk[0]);
                    dv[2] = unpack_lo<DYF == 2>(pk[1]); dv[3] = unpack_hi<DYF == 2>(pk[1]);
                } else {
                    dv = *(const f32x4*)((const float*)dy_ + (long)row * d + ch * 4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {         // (vector elements cannot bind to references: through locals)
                    float xh_, gy_, dg_ = dg[i][e], db_ = db[i][e];
                    ln_bwd_accum(xv[e], dv[e], g[i][e], mu, R.rs, xh_, gy_, R.s1, R.s2, dg_, db_);
                    R.xh[i][e] = xh_; R.gy[i][e] = gy_; dg[i][e] = dg_; db[i][e] = db_;
                }
            }
        }
    };
    auto phase2 = [&](int row, const Row& R) {
        const float c1 = wave_sum(R.s1) / d, c2 = wave_sum(R.s2) / d;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ln_bwd_out(R.gy[i][e], R.xh[i][e], c1, c2, R.rs, R.di[i][e]);
                ds[i] += o;
                if (dx) *(f32x4*)(dx + (long)row * d + ch * 4) = o;
                if (dxb) {
                    u32x2 pk;
                    pk[0] = pack2h<OF16>(o[0], o[1]);
                    pk[1] = pack2h<OF16>(o[2], o[3]);
                    *(u32x2*)(dxb + (long)row * d + ch * 4) = pk;
                }
            }
        }
    };
    const int stride = gridDim.x * WPB;
    if constexpr (WPB == 8) {            // the few-rows variant: registers for two rows in flight (one block of 8 waves per CU)
        for (int row = blockIdx.x * WPB + wave; row < rows; row += 2 * stride) {
            Row A, Bq;
            const int row_b = row + stride;
            phase1(row, A);
            if (row_b < rows) phase1(row_b, Bq);
            phase2(row, A);
            if (row_b < rows) phase2(row_b, Bq);
        }
    } else {                             // one row per trip, written out (through the lambdas hipcc spills at the 128-register budget)
        for (int row = blockIdx.x * WPB + wave; row < rows; row += stride) {
            const float mu = mean[row], rs = rstd[row];
            f32x4 xh[NCH], gy[NCH], di[NCH];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nch) {
                    di[i] = dx_in ? *(const f32x4*)(dx_in + (long)row * d + ch * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                    f32x4 xv = *(const f32x4*)(x + (long)row * d + ch * 4);
                    if (resid) xv += *(const f32x4*)(resid + (long)row * d + ch * 4);
                    f32x4 dv;
                    if (DYF) {
                        const u32x2 pk = *(const u32x2*)((const bf16_t*)dy_ + (long)row * d + ch * 4);
                        dv[0] = unpack_lo<DYF == 2>(pk[0]); dv[1] = unpack_hi<DYF == 2>(pk[0]);
                        dv[2] = unpack_lo<DYF == 2>(pk[1]); dv[3] = unpack_hi<DYF == 2>(pk[1]);
                    } else {
                        dv = *(const f32x4*)((const float*)dy_ + (long)row * d + ch * 4);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float xh_, gy_, dg_ = dg[i][e], db_ = db[i][e];
                        ln_bwd_accum(xv[e], dv[e], g[i][e], mu, rs, xh_, gy_, s1, s2, dg_, db_);
                        xh[i][e] = xh_; gy[i][e] = gy_; dg[i][e] = dg_; db[i][e] = db_;
                    }
                }
            }
            const float c1 = wave_sum(s1) / d, c2 = wave_sum(s2) / d;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nch) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = ln_bwd_out(gy[i][e], xh[i][e], c1, c2, rs, di[i][e]);
                    ds[i] += o;
                    if (dx) *(f32x4*)(dx + (long)row * d + ch * 4) = o;
                    if (dxb) {
                        u32x2 pk;
                        pk[0] = pack2h<OF16>(o[0], o[1]);
                        pk[1] = pack2h<OF16>(o[2], o[3]);
                        *(u32x2*)(dxb + (long)row * d + ch * 4) = pk;
                    }
                }
            }
        }
    }
    if (!dgamma && !dbeta && !dxsum) return;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][wave][(lane + 64 * i) * 4 + e] = dg[i][e];
            red[1][wave][(lane + 64 * i) * 4 + e] = db[i][e];
            red[2][wave][(lane + 64 * i) * 4 + e] = ds[i][e];
        }
    __syncthreads();
    for (int col = threadIdx.x; col < d; col += WPB * 64) {
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) { a += red[0][w][col]; b += red[1][w][col]; c += red[2][w][col]; }
        if (dgamma) atomicAdd(&dgamma[col], a);
        if (dbeta) atomicAdd(&dbeta[col], b);
        if (dxsum) atomicAdd(&dxsum[col], c);     // column sums of the produced dx: the bias gradient of the Linear
                                                  // whose output gradient this dx is (no separate colsum pass)
    }
}

// internal form (driver.h): f16 != 0 makes the 16-bit output IEEE fp16 instead of bf16
int hs_layernorm_fwd16(const float* x, const float* resid, const float* gamma, const float* beta, float* y_f32, void* y_16, float* mean,
                       float* rstd, int rows, int d, float eps, int f16, void* stream) {
    if (!x || !gamma || !beta || (!y_f32 && !y_16)) return HS_ERR_ARG;
    if (rows < 1 || d < 4 || (d & 3) || d > LN_MAXCH * 256) return HS_ERR_SHAPE;
    ProfScope prof(HS_PROF_LN_FWD, stream, 0, (double)rows * d * (4 + (resid ? 4 : 0) + (y_f32 ? 4 : 0) + (y_16 ? 2 : 0)));
    if (f16) hipLaunchKernelGGL(layernorm_fwd_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, resid, gamma, beta,
                                y_f32, (bf16_t*)y_16, mean, rstd, rows, d, eps);
    else hipLaunchKernelGGL(layernorm_fwd_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, resid, gamma, beta,
                            y_f32, (bf16_t*)y_16, mean, rstd, rows, d, eps);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
extern "C" int hs_layernorm_fwd(const float* x, const float* resid, const float* gamma, const float* beta, float* y_f32,
                                void* y_bf16, float* mean, float* rstd, int rows, int d, float eps, void* stream) {
    return hs_layernorm_fwd16(x, resid, gamma, beta, y_f32, y_bf16, mean, rstd, rows, d, eps, 0, stream);
}

extern "C" int hs_layernorm_bwd(const void* dy, int dy_is_bf16, const float* x, const float* resid, const float* gamma,
                                const float* mean, const float* rstd, const float* dx_in, float* dx, void* dx_bf16,
                                float* dgamma, float* dbeta, float* dxsum, int rows, int d, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || (!dx && !dx_bf16)) return HS_ERR_ARG;
    if (rows < 1 || d < 4 || (d & 3) || d > LN_MAXCH * 256) return HS_ERR_SHAPE;
    // dy_is_bf16: HS_LN_DY_F32 (0) / HS_LN_DY_BF16 (1) / HS_LN_DY_F16 (2: dy fp16 and the 16-bit dx copy fp16), or
    // HS_LN_DY_F32 | HS_LN_OUT_F16 (4: fp32 dy, the 16-bit dx copy as fp16 -- the top of an fp16 backward pass)
    const int dyf = dy_is_bf16 & 3;
    const bool of16 = dyf == 2 || (dy_is_bf16 & 4);
    if (dyf == 3 || (dy_is_bf16 & ~7) || ((dy_is_bf16 & 4) && dyf == 1)) return HS_ERR_ARG;
    // every block ends with 3 x d float atomics (dgamma, dbeta, dxsum): at one row per wave they outweigh the rows' own traffic
    // (4097 rows x 768: 45.5 us with 4 rows per block, 28.3 us with 16), so a block takes 16 rows while that still leaves >= 256 blocks
    int blocks = (rows + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (rows / 16 >= 256 && (rows + 15) / 16 < blocks) blocks = (rows + 15) / 16;
    // few rows (up to ~10 volumes per GPU): the block's 3 x d float atomics onto the SAME 3 x d addresses from every block are the
    // serial part (257 blocks at 4097 rows) -- 8 waves x 4 rows per block halve the block count at the same per-wave chain
    // Measured (tools/ln_bwd_ab.py on a hook build, d = 768, bf16 dy; wide / narrow time): 4097 rows 0.77, 8194 0.97, 12291 0.80, 16388 0.81,
    // 20485 0.77, 24582 0.84, 28679 0.91, 32776 0.86, 49164 0.96, 81940 1.05 -> the 8-wave variant up to 10 volumes' worth of rows.
    bool wide = rows / 32 >= 128 && rows <= 10 * 4100 && d <= 768;
    if (HS_DBG(g_hs_dbg[5]) == 1) wide = false;                            // (hook builds: A/B of the two variants, tools/ln_bwd_ab.py)
    if (HS_DBG(g_hs_dbg[5]) == 2) wide = rows / 32 >= 1 && d <= 768;
    if (wide) blocks = (rows + 31) / 32;
    ProfScope prof(HS_PROF_LN_BWD, stream, 0, (double)rows * d * (4 + (dyf ? 2 : 4) + (resid ? 4 : 0) + (dx_in ? 4 : 0) + (dx ? 4 : 0) + (dx_bf16 ? 2 : 0)));
    const int nchunks = (d + 255) / 256;
#define HS_LN_BWD(DYF, OF16, NCH)                                                                                         \
    do {                                                                                                                  \
        if (wide && NCH <= 3)                                                                                             \
            hipLaunchKernelGGL((layernorm_bwd_kernel<DYF, (NCH <= 3 ? NCH : 1), 8, OF16>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, dy, x, \
                               resid, gamma, mean, rstd, dx_in, dx, (bf16_t*)dx_bf16, dgamma, dbeta, dxsum, rows, d);      \
        else                                                                                                              \
            hipLaunchKernelGGL((layernorm_bwd_kernel<DYF, NCH, 4, OF16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, resid, gamma, \
                               mean, rstd, dx_in, dx, (bf16_t*)dx_bf16, dgamma, dbeta, dxsum, rows, d);                    \
    } while (0)
#define HS_LN_BWD_N(DYF, OF16)                                                                                            \
    do {                                                                                                                  \
        if (nchunks == 1) HS_LN_BWD(DYF, OF16, 1); else if (nchunks == 2) HS_LN_BWD(DYF, OF16, 2);                         \
        else if (nchunks == 3) HS_LN_BWD(DYF, OF16, 3); else HS_LN_BWD(DYF, OF16, 4);                                      \
    } while (0)
    if (dyf == 1) HS_LN_BWD_N(1, false);
    else if (dyf == 2) HS_LN_BWD_N(2, true);
    else if (of16) HS_LN_BWD_N(0, true);
    else HS_LN_BWD_N(0, false);
#undef HS_LN_BWD_N
#undef HS_LN_BWD
    HS_CHECK_LAUNCH();
    return HS_OK;
}
