// Helpers shared by the attention kernels (attn.hip; tools/experiments/attn_bwd_one_sweep/attn_bwd1.hip): accumulator <-> operand packing, image-R fragment reads,
// asm-issued transposed LDS reads.  gfx950 only.
#pragma once
#include "common.h"

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
template <bool F16 = false>
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {  // registers 8s..8s+7 -> bf16x8 / f16x8 (k-step s)
    u32x4 u;
    u[0] = pack2h<F16>(a[8 * s + 0], a[8 * s + 1]);
    u[1] = pack2h<F16>(a[8 * s + 2], a[8 * s + 3]);
    u[2] = pack2h<F16>(a[8 * s + 4], a[8 * s + 5]);
    u[3] = pack2h<F16>(a[8 * s + 6], a[8 * s + 7]);
    return __builtin_bit_cast(bf16x8, u);
}
// column (transposed) fragment of image R: element j <-> row rowbase + 8(j>>2) + 4h + (j&3), column db*32 + (lane&31)
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int rowbase, int db, int lane) {
    const int g = lane >> 4, h = g >> 1, p = lane & 3, q = (lane >> 2) & 3;
    const int col = db * 32 + 16 * (g & 1) + 4 * p;
    const int r1 = rowbase + 4 * h + q;
    const bf16x4 a = lds_tr16(img + offR(r1, col >> 3) + 8 * (p & 1));
    const bf16x4 b = lds_tr16(img + offR(r1 + 8, col >> 3) + 8 * (p & 1));
    return cat4(a, b);
}
__device__ __forceinline__ bf16x8 row_frag(const char* img, int row, int ks, int h) {
    return *(const bf16x8*)(img + offR(row, 2 * ks + h));
}
// store a [d][query] accumulator pair as rows of a [..][ld] bf16 matrix: lane owns one query row
template <bool F16 = false>
__device__ __forceinline__ void store_dT(bf16_t* rowptr, const f32x16 (&acc)[2], float mul, int h) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            u32x2 v;
            v[0] = pack2h<F16>(acc[db][4 * g4 + 0] * mul, acc[db][4 * g4 + 1] * mul);
            v[1] = pack2h<F16>(acc[db][4 * g4 + 2] * mul, acc[db][4 * g4 + 3] * mul);
            *(u32x2*)(rowptr + db * 32 + 8 * g4 + 4 * h) = v;
        }
}

__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(__UINTPTR_TYPE__)(LDS_PTR(const char))p; }
// two transposed 4-row reads (rows r, r+8 of a 16-row group) = one 32x16 MFMA operand fragment; issued from asm so that
// hipcc does not drain the LDS-DMA ring in front of them (it cannot prove the builtin does not alias the DMA writes)
template <int IMM>
__device__ __forceinline__ void trR_issue(u32x4& f, unsigned b0, unsigned b1) {
    u32x2 lo, hi;
#if HS_ABL & 2
    asm volatile("" : "=&v"(lo), "=&v"(hi) : "v"(b0), "v"(b1));
#else
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"
                 : "=&v"(lo), "=&v"(hi) : "v"(b0), "v"(b1), "i"(IMM) : "memory");
#endif
    f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}
__device__ __forceinline__ void lds_settle2(u32x4& a, u32x4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory"); }
template <bool F16 = false>
__device__ __forceinline__ bf16x8 scale_frag(u32x4 raw, float mul) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        r[i] = pack2h<F16>(unpack_lo<F16>(raw[i]) * mul, unpack_hi<F16>(raw[i]) * mul);
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ void lds_settle4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
__device__ __forceinline__ f32x16 splat16(float v) {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = v;
    return z;
}
