// Shared device/host helpers for the hsenet_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hsenet_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;   // one ds_read_b64_tr_b16 result
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define HS_WAVE 64
#define LDS_PTR(T) __attribute__((address_space(3))) T*

// offset that is out of range for every buffer descriptor we build (num_records is clamped below it)
#define HS_OOB_OFF 0x7FFFFFF0u
#define HS_MAX_RECORDS 0x7FFFFFF0u

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even f32 -> bf16 through the compiler's cast (v_cvt_pk_bf16_f32; NaN stays NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 v = {lo, hi};
    bf2 r = __builtin_convertvector(v, bf2);
    return __builtin_bit_cast(unsigned, r);
}

// ---- 16-bit operand format as a template flag: bf16 (F16 = false: training, the reference's torch.autocast(bf16)) or IEEE fp16 (F16 = true:
// HS_PREC_F16, forward only -- the reference's generation / eval scripts run under torch.cuda.amp.autocast()'s default dtype, fp16).
// Same kernels, same LDS images, same data flow: only the packs / unpacks and the MFMA opcode differ (fp16 MFMA runs at the bf16 rate).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <bool F16> __device__ __forceinline__ unsigned pack2h(float lo, float hi) {
    if constexpr (F16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        typedef __attribute__((ext_vector_type(2))) float f2;
        f2 v = {lo, hi};
        h2 r = __builtin_convertvector(v, h2);           // round to nearest even (v_cvt_pk_f16_f32); overflow -> inf, as torch's .half()
        return __builtin_bit_cast(unsigned, r);
    } else {
        return pack2bf(lo, hi);
    }
}
template <bool F16> __device__ __forceinline__ bf16_t f2h(float f) {
    if constexpr (F16) { _Float16 h = (_Float16)f; return __builtin_bit_cast(unsigned short, h); } else return f2bf(f);
}
template <bool F16> __device__ __forceinline__ float h2f(bf16_t v) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v); else return bf2f(v);
}
template <bool F16> __device__ __forceinline__ float unpack_lo(unsigned u) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); else return __uint_as_float(u << 16);
}
template <bool F16> __device__ __forceinline__ float unpack_hi(unsigned u) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); else return __uint_as_float(u & 0xffff0000u);
}
template <bool F16> __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// row in a 32x32 MFMA accumulator held in register r by a lane of half h (cdna guide section 3)
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// exact (erf) GELU and its derivative, as nn.GELU() / MONAI act="GELU"
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Buffer descriptor with hardware range checking: loads past `bytes` return 0.
// The descriptor inputs go through readfirstlane so the compiler can PROVE it wave-uniform; otherwise it wraps every
// buffer op in a serialising "waterfall" loop (cdna guide T20).  All callers pass block-uniform bases.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
    unsigned n = bytes > (size_t)HS_MAX_RECORDS ? HS_MAX_RECORDS : (unsigned)bytes;
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    n = __builtin_amdgcn_readfirstlane(n);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}

// ---- LDS images -----------------------------------------------------------------------------
// Image R: [rows][64 bf16] (128-B rows), read by rows with ds_read_b128 (MFMA operand whose k is
// contiguous in memory) AND by columns with ds_read_b64_tr_b16.  16-B chunk c of row r sits at
// chunk c ^ swzR(r).  Derivation in DESIGN.md ("LDS images"): conflict-free for both read kinds.
__device__ __forceinline__ int swzR(int r) { return (((r >> 1) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 2) & 1); }
__device__ __forceinline__ int offR(int r, int c) { return r * 128 + ((c ^ swzR(r)) << 4); }
// Image T: [rows][128 bf16] (256-B rows), read only by columns (ds_read_b64_tr_b16): the operand
// whose k index is the memory row.  Swizzle (b) of the cdna guide T10.
__device__ __forceinline__ int offT(int r, int c) { return r * 256 + ((c ^ (((r & 3) << 2) | ((r >> 2) & 3))) << 4); }

__device__ __forceinline__ bf16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(bf16x4))(p));
}
__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) {
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

// ---- LDS reads issued from asm ----------------------------------------------------------------
// Row-fragment reads issued from asm: left to hipcc, each pair of ds_read_b128 is issued just in time and
// waited for with lgkmcnt(0) right in front of its MFMAs (one LDS round trip per two MFMAs).  Issued in groups ahead of
// use and retired with COUNTED waits instead; the waits name the registers they release ("+v") so no consumer moves up.
// HS_ABL (ablation builds of tools/build_variant.sh only, timing studies with WRONG results; never defined in the product):
// bit 0 drops the exp2 of the attention backward kernels, bit 1 every asm-issued LDS read, bit 2 the dQ / dK,dV accumulation MFMAs
#ifndef HS_ABL
#define HS_ABL 0
#endif
template <int IMM>
__device__ __forceinline__ void ldsr128(u32x4& f, unsigned base) {
#if HS_ABL & 2
    asm volatile("" : "=v"(f) : "v"(base));
#else
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(base), "i"(IMM) : "memory");
#endif
}
template <int CNT>
__device__ __forceinline__ void lds_wait4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
#if HS_ABL & 8      // (timing study: reads issued but never waited for)
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(CNT) : "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(CNT) : "memory");
#endif
}
template <int CNT>
__device__ __forceinline__ void lds_wait6(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e, u32x4& f) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "i"(CNT) : "memory");
}

// ---- Philox4x32-10 (counter-based RNG for train-mode dropout: the mask is a pure function of (seed, element index)) ----
struct Philox4 { unsigned x, y, z, w; };
__device__ __forceinline__ Philox4 philox4x32_10(unsigned long long seed, unsigned long long ctr) {
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0u, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}
// keep-scale of element i under dropout probability p: 1/(1-p) with probability 1-p, else 0
__device__ __forceinline__ float dropout_scale(unsigned long long seed, unsigned long long offset, unsigned long long i, float p) {
    const Philox4 r = philox4x32_10(seed, offset + (i >> 2));
    const unsigned u = (i & 3) == 0 ? r.x : (i & 3) == 1 ? r.y : (i & 3) == 2 ? r.z : r.w;
    return (float)(u >> 8) * (1.0f / 16777216.0f) >= p ? 1.0f / (1.0f - p) : 0.0f;
}

// per-kernel-class profiling (prof.hip); ids mirror hs_prof_name()
// (a class = ONE kernel function wherever that kernel is hot: the five 16x16x32 NT instantiations and the grouped weight-gradient
// kernel have classes of their own, so that "the dominant kernel" of a run is a kernel, not a sum over a GEMM layout; the three
// layout classes collect what is left: small tiles, generic epilogues, the packers' and embedding GEMMs)
enum { HS_PROF_GEMM_NT = 0, HS_PROF_GEMM_NN, HS_PROF_GEMM_TN, HS_PROF_ATTN_FWD, HS_PROF_ATTN_DELTA, HS_PROF_ATTN_DKDV,
       HS_PROF_ATTN_DQ, HS_PROF_LN_FWD, HS_PROF_LN_BWD, HS_PROF_COLSUM, HS_PROF_CAST, HS_PROF_PATCHIFY, HS_PROF_OTHER,
       HS_PROF_GEMM_NT16_PLAIN, HS_PROF_GEMM_NT16_RESID, HS_PROF_GEMM_NT16_GELU, HS_PROF_GEMM_NT16_DGELU, HS_PROF_GEMM_TN_GROUP,
       HS_PROF_ADAMW, HS_PROF_GEMM_NT16_BIAS, HS_PROF_COUNT };
int hs_prof_begin(int id, void* stream, double flops, double bytes);
void hs_prof_end(int token, void* stream);
struct ProfScope {
    int tok; void* s;
    ProfScope(int id, void* stream, double flops = 0, double bytes = 0) : tok(hs_prof_begin(id, stream, flops, bytes)), s(stream) {}
    ~ProfScope() { hs_prof_end(tok, s); }
};

extern int g_hs_dbg[8];   // diagnostic variant selectors (prof.hip: hs_debug_set)
// Result-changing / timing-only kernel variants and the s_memtime stamps exist only in builds made with -DHS_DEBUG_HOOKS
// (tools/build_variant.sh for tools/kbench); in the product library the selector is the constant 0 and the branches fold away.
#ifdef HS_DEBUG_HOOKS
#define HS_DBG(x) (x)
#else
#define HS_DBG(x) 0
#endif

#define HS_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return HS_ERR_LAUNCH;        \
    } while (0)
