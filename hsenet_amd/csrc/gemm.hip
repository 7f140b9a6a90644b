// K3: GEMM family  C[M,N] = sum_k A(m,k) B(n,k)  with fused epilogues (see include/hsenet_hip.h).
//
// bf16 kernel: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 blocks.
// Operands are staged global -> registers (range-checked buffer loads: every M/N/K tail reads as 0)
// -> LDS (write AFTER the MFMA phase of the current tile, one barrier per K-tile; cdna guide T14).
// An operand whose k index is contiguous in memory lives in LDS image R (ds_read_b128 row reads); an
// operand stored [K][dim] lives in image T and is read with ds_read_b64_tr_b16 (hardware transpose),
// so NT / NN / TN all feed the same MFMA loop and no transposed copies of weights or activations exist.
//
// f32 kernel (parity mode): 64x64x32 tile, v_mfma_f32_32x32x2_f32 (exact f32 fma chain).
#include "common.h"

struct GemmP {
    const void* A; const void* B;
    int M, N, K;
    long lda, ldb, sA, sB, sC, ldc;
    float* out_f32; void* out_bf16;
    const float* bias; const float* resid; const void* aux; void* aux_out;
    float alpha; int flags; int split_k; int tiles_m;
};

template <bool F32AUX>
__device__ __forceinline__ void epilogue_store(const GemmP& p, long cz, int m, int n, float v) {
    if (m >= p.M || n >= p.N) return;
    v *= p.alpha;
    if (p.flags & HS_EPI_BIAS) v += p.bias[(p.flags & HS_EPI_BIAS_M) ? m : n];
    const long idx = cz + (long)m * p.ldc + n;
    if (p.flags & HS_EPI_GELU) {
        if (!F32AUX) v = bf2f(f2bf(v));  // autocast: linear1 emits bf16, GELU runs on that
        if (p.aux_out) {
            if (F32AUX) ((float*)p.aux_out)[idx] = v; else ((bf16_t*)p.aux_out)[idx] = f2bf(v);
        }
        v = gelu_f(v);
    }
    if (p.flags & HS_EPI_GELU_BWD) {
        const float z = F32AUX ? ((const float*)p.aux)[idx] : bf2f(((const bf16_t*)p.aux)[idx]);
        v *= gelu_grad_f(z);
    }
    if (p.flags & HS_EPI_RESID) v += p.resid[idx];
    if (p.out_f32) {
        if (p.flags & HS_EPI_ATOMIC) atomicAdd(&p.out_f32[idx], v); else p.out_f32[idx] = v;
    }
    if (p.out_bf16) ((bf16_t*)p.out_bf16)[idx] = f2bf(v);
}

// ------------------------------------------------------------------------------------------------------
// bf16
// ------------------------------------------------------------------------------------------------------
template <bool TR>
__device__ __forceinline__ void bf_gload(u32x4 (&reg)[4], __amdgpu_buffer_rsrc_t rs, int dim0, int DIM, int k0,
                                         int K, long ld, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 256 + tid;
        unsigned off;
        if (!TR) {
            const int gi = dim0 + (idx >> 3), gk = k0 + (idx & 7) * 8;
            off = (gi < DIM && gk < K) ? (unsigned)(((long)gi * ld + gk) * 2) : HS_OOB_OFF;
        } else {
            const int gk = k0 + (idx >> 4), gi = dim0 + (idx & 15) * 8;
            off = (gk < K && gi < DIM) ? (unsigned)(((long)gk * ld + gi) * 2) : HS_OOB_OFF;
        }
        reg[it] = buf_load16(rs, off);
    }
}
template <bool TR>
__device__ __forceinline__ void bf_swrite(const u32x4 (&reg)[4], char* lds, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 256 + tid;
        const int o = TR ? offT(idx >> 4, idx & 15) : offR(idx >> 3, idx & 7);
        *(u32x4*)(lds + o) = reg[it];
    }
}
// MFMA 32x32x16 operand fragment for the 32-wide block starting at `sub` of the 128-wide tile, k-step ks.
template <bool TR>
__device__ __forceinline__ bf16x8 bf_frag(const char* lds, int sub, int ks, int lane) {
    if (!TR) {
        return *(const bf16x8*)(lds + offR(sub + (lane & 31), 2 * ks + (lane >> 5)));
    } else {
        const int g = lane >> 4, h = g >> 1, p = lane & 3, q = (lane >> 2) & 3;
        const int col = sub + 16 * (g & 1) + 4 * p;
        const int r1 = ks * 16 + 8 * h + q;
        const bf16x4 a = lds_tr16(lds + offT(r1, col >> 3) + 8 * (p & 1));
        const bf16x4 b = lds_tr16(lds + offT(r1 + 4, col >> 3) + 8 * (p & 1));
        return cat4(a, b);
    }
}

template <bool ATR, bool BTR>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x % p.tiles_m, tn = blockIdx.x / p.tiles_m;
    const int m0 = tm * 128, n0 = tn * 128;
    const int z = blockIdx.z;
    const int bz = p.split_k > 1 ? 0 : z;
    const bf16_t* A = (const bf16_t*)p.A + bz * p.sA;
    const bf16_t* B = (const bf16_t*)p.B + bz * p.sB;
    const long cz = bz * p.sC;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (size_t)(ATR ? p.K : p.M) * p.lda * 2);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (size_t)(BTR ? p.K : p.N) * p.ldb * 2);

    const int kt_total = (p.K + 63) >> 6;
    int kt0 = 0, kt1 = kt_total;
    if (p.split_k > 1) {
        const int per = (kt_total + p.split_k - 1) / p.split_k;
        kt0 = z * per;
        kt1 = min(kt_total, kt0 + per);
    }
    // double buffer: stage `cur` = [A 16 KiB | B 16 KiB] at smem + cur*32 KiB
#define AS(c) (smem + (c) * 32768)
#define BS(c) (smem + (c) * 32768 + 16384)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt0 < kt1) {
        u32x4 ra[4], rb[4];
        bf_gload<ATR>(ra, rsA, m0, p.M, kt0 * 64, p.K, p.lda, tid);
        bf_gload<BTR>(rb, rsB, n0, p.N, kt0 * 64, p.K, p.ldb, tid);
        bf_swrite<ATR>(ra, AS(0), tid);
        bf_swrite<BTR>(rb, BS(0), tid);
        __syncthreads();
        int cur = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            const bool more = kt + 1 < kt1;
            if (more) {
                bf_gload<ATR>(ra, rsA, m0, p.M, (kt + 1) * 64, p.K, p.lda, tid);
                bf_gload<BTR>(rb, rsB, n0, p.N, (kt + 1) * 64, p.K, p.ldb, tid);
            }
            const char* a_l = AS(cur);
            const char* b_l = BS(cur);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 a0 = bf_frag<ATR>(a_l, wm * 64, ks, lane);
                const bf16x8 a1 = bf_frag<ATR>(a_l, wm * 64 + 32, ks, lane);
                const bf16x8 b0 = bf_frag<BTR>(b_l, wn * 64, ks, lane);
                const bf16x8 b1 = bf_frag<BTR>(b_l, wn * 64 + 32, ks, lane);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (more) {
                bf_swrite<ATR>(ra, AS(cur ^ 1), tid);
                bf_swrite<BTR>(rb, BS(cur ^ 1), tid);
            }
            __syncthreads();
            cur ^= 1;
        }
    } else if (p.split_k > 1) {
        return;  // empty K slice: nothing to add
    }
#undef AS
#undef BS
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                epilogue_store<false>(p, cz, m0 + wm * 64 + mb * 32 + acc_row(r, h), n0 + wn * 64 + nb * 32 + c,
                                      acc[mb][nb][r]);
}

// ------------------------------------------------------------------------------------------------------
// f32 (parity mode): 64x64x32 tile, 4 waves each one 32x32 block; k order inside a group of 8 is permuted
// identically for A and B (lane half h takes k = 8s+4h+e for MFMA step e), which leaves the sum unchanged.
// ------------------------------------------------------------------------------------------------------
template <bool TR>
__device__ __forceinline__ void f32_gload(u32x4 (&reg)[2], __amdgpu_buffer_rsrc_t rs, int dim0, int DIM, int k0, int K,
                                          long ld, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = it * 256 + tid;
        unsigned off;
        if (!TR) {
            const int gi = dim0 + (idx >> 3), gk = k0 + (idx & 7) * 4;
            off = (gi < DIM && gk < K) ? (unsigned)(((long)gi * ld + gk) * 4) : HS_OOB_OFF;
        } else {
            const int gk = k0 + (idx >> 4), gi = dim0 + (idx & 15) * 4;
            off = (gk < K && gi < DIM) ? (unsigned)(((long)gk * ld + gi) * 4) : HS_OOB_OFF;
        }
        reg[it] = buf_load16(rs, off);
    }
}
template <bool TR>
__device__ __forceinline__ void f32_swrite(const u32x4 (&reg)[2], char* lds, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = it * 256 + tid;
        const int o = TR ? ((idx >> 4) * 256 + (idx & 15) * 16) : offR(idx >> 3, idx & 7);
        *(u32x4*)(lds + o) = reg[it];
    }
}
// 4 k-values (k = 8s+4h+e, e=0..3) of row/col (sub + lane&31)
template <bool TR>
__device__ __forceinline__ f32x4 f32_frag(const char* lds, int sub, int s, int lane) {
    const int i = sub + (lane & 31), h = lane >> 5;
    if (!TR) {
        return *(const f32x4*)(lds + offR(i, 2 * s + h));
    } else {
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = *(const float*)(lds + (8 * s + 4 * h + e) * 256 + i * 4);
        return r;
    }
}

template <bool ATR, bool BTR>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x % p.tiles_m, tn = blockIdx.x / p.tiles_m;
    const int m0 = tm * 64, n0 = tn * 64;
    const int z = blockIdx.z;
    const int bz = p.split_k > 1 ? 0 : z;
    const float* A = (const float*)p.A + bz * p.sA;
    const float* B = (const float*)p.B + bz * p.sB;
    const long cz = bz * p.sC;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (size_t)(ATR ? p.K : p.M) * p.lda * 4);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (size_t)(BTR ? p.K : p.N) * p.ldb * 4);
    const int kt_total = (p.K + 31) >> 5;
    int kt0 = 0, kt1 = kt_total;
    if (p.split_k > 1) {
        const int per = (kt_total + p.split_k - 1) / p.split_k;
        kt0 = z * per;
        kt1 = min(kt_total, kt0 + per);
    }
#define AS(c) (smem + (c) * 16384)
#define BS(c) (smem + (c) * 16384 + 8192)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (kt0 < kt1) {
        u32x4 ra[2], rb[2];
        f32_gload<ATR>(ra, rsA, m0, p.M, kt0 * 32, p.K, p.lda, tid);
        f32_gload<BTR>(rb, rsB, n0, p.N, kt0 * 32, p.K, p.ldb, tid);
        f32_swrite<ATR>(ra, AS(0), tid);
        f32_swrite<BTR>(rb, BS(0), tid);
        __syncthreads();
        int cur = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            const bool more = kt + 1 < kt1;
            if (more) {
                f32_gload<ATR>(ra, rsA, m0, p.M, (kt + 1) * 32, p.K, p.lda, tid);
                f32_gload<BTR>(rb, rsB, n0, p.N, (kt + 1) * 32, p.K, p.ldb, tid);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 a = f32_frag<ATR>(AS(cur), wm * 32, s, lane);
                const f32x4 b = f32_frag<BTR>(BS(cur), wn * 32, s, lane);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
            }
            if (more) {
                f32_swrite<ATR>(ra, AS(cur ^ 1), tid);
                f32_swrite<BTR>(rb, BS(cur ^ 1), tid);
            }
            __syncthreads();
            cur ^= 1;
        }
    } else if (p.split_k > 1) {
        return;
    }
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) epilogue_store<true>(p, cz, m0 + wm * 32 + acc_row(r, h), n0 + wn * 32 + c, acc[r]);
}

extern "C" int hs_gemm(const hs_gemm_args* a, int prec, void* stream) {
    if (!a || !a->A || !a->B || (!a->out_f32 && !a->out_bf16)) return HS_ERR_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch < 1 || a->split_k < 1) return HS_ERR_SHAPE;
    if (a->layout < HS_GEMM_NT || a->layout > HS_GEMM_TN) return HS_ERR_ARG;
    if ((a->flags & HS_EPI_BIAS) && !a->bias) return HS_ERR_ARG;
    if ((a->flags & HS_EPI_RESID) && !a->resid) return HS_ERR_ARG;
    if ((a->flags & HS_EPI_GELU_BWD) && !a->aux) return HS_ERR_ARG;
    if (a->split_k > 1 && (!(a->flags & HS_EPI_ATOMIC) || a->batch != 1 || !a->out_f32 || a->out_bf16)) return HS_ERR_SHAPE;
    if (prec != HS_PREC_BF16 && prec != HS_PREC_F32) return HS_ERR_ARG;
    const int esz = prec == HS_PREC_BF16 ? 2 : 4;
    const int vec = 16 / esz;  // elements per 16-B chunk
    const bool atr = a->layout == HS_GEMM_TN, btr = a->layout != HS_GEMM_NT;
    // 16-B chunk granularity along the contiguous axis of each operand
    if ((atr ? a->M : a->K) % vec || (btr ? a->N : a->K) % vec || a->lda % vec || a->ldb % vec) return HS_ERR_SHAPE;
    if (a->strideA % vec || a->strideB % vec) return HS_ERR_SHAPE;
    if (((uintptr_t)a->A | (uintptr_t)a->B) & 15) return HS_ERR_SHAPE;
    if ((size_t)(atr ? a->K : a->M) * a->lda * esz >= HS_MAX_RECORDS) return HS_ERR_SHAPE;
    if ((size_t)(btr ? a->K : a->N) * a->ldb * esz >= HS_MAX_RECORDS) return HS_ERR_SHAPE;
    GemmP p;
    p.A = a->A; p.B = a->B; p.M = a->M; p.N = a->N; p.K = a->K;
    p.lda = a->lda; p.ldb = a->ldb; p.sA = a->strideA; p.sB = a->strideB; p.sC = a->strideC; p.ldc = a->ldc;
    p.out_f32 = a->out_f32; p.out_bf16 = a->out_bf16; p.bias = a->bias; p.resid = a->resid; p.aux = a->aux;
    p.aux_out = a->aux_out; p.alpha = a->alpha; p.flags = a->flags; p.split_k = a->split_k;
    hipStream_t s = (hipStream_t)stream;
    const int zdim = a->split_k > 1 ? a->split_k : a->batch;
    const double mnk = (double)a->M * a->N * a->K * a->batch;
    ProfScope prof(HS_PROF_GEMM_NT + a->layout, stream, 2.0 * mnk,
                   ((double)a->M * a->K + (double)a->N * a->K) * a->batch * esz + (double)a->M * a->N * a->batch * (a->out_f32 ? 4 : 2));
    if (prec == HS_PREC_BF16) {
        p.tiles_m = (a->M + 127) / 128;
        dim3 grid(p.tiles_m * ((a->N + 127) / 128), 1, zdim);
        if (a->layout == HS_GEMM_NT) hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, dim3(256), 0, s, p);
        else if (a->layout == HS_GEMM_NN) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, dim3(256), 0, s, p);
    } else {
        p.tiles_m = (a->M + 63) / 64;
        dim3 grid(p.tiles_m * ((a->N + 63) / 64), 1, zdim);
        if (a->layout == HS_GEMM_NT) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, s, p);
        else if (a->layout == HS_GEMM_NN) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, s, p);
    }
    HS_CHECK_LAUNCH();
    return HS_OK;
}
