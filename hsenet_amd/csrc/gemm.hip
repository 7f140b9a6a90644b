// K3: GEMM family  C[M,N] = sum_k A(m,k) B(n,k)  with fused epilogues (see include/hsenet_hip.h).
//
// bf16 kernel: 256x128x64 block tile, 8 waves, each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 blocks; operand
// tiles arrive by LDS-DMA into a 3-stage ring (details at the kernel).
// An operand whose k index is contiguous in memory lives in LDS image R (ds_read_b128 row reads); an
// operand stored [K][dim] lives in image T and is read with ds_read_b64_tr_b16 (hardware transpose),
// so NT / NN / TN all feed the same MFMA loop and no transposed copies of weights or activations exist.
//
// f32 kernel (parity mode): 64x64x32 tile, v_mfma_f32_32x32x2_f32 (exact f32 fma chain).
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

struct GemmP {
    const void* A; const void* B;
    int M, N, K;
    long lda, ldb, sA, sB, sC, ldc;
    float* out_f32; void* out_bf16;
    const float* bias; const float* resid; const void* aux; void* aux_out; float* colsum;
    float alpha; int flags; int split_k; int tiles_m, tiles_n;
    int scale_cols; float scale_alpha;     // columns n < scale_cols carry the extra factor scale_alpha
    int nt;                                // streaming (nontemporal) output stores, see hs_gemm
    int nblocks;                           // tiles_m * tiles_n * zdim
    int gm;                                // row tiles per group of the tile order (chosen by hs_gemm, see tile_order)
    int np;                                // column tiles per PANEL of the tile order (0: one panel = every column tile)
    // diagnostic (tools/kbench, HS_GEMM_DBG): 1 = skip the epilogue, 2 = one K-tile only, 3 = epilogue without its stores, 4 = s_memtime
    // phase sums (hs_debug_gemm_stamps), 5 = every K-tile re-reads the first (cache hits), 7 = 3 + 4, 22 = per-block timeline
    // (hs_debug_gemm_trace).  Compiled in only under -DHS_DEBUG_HOOKS (tools/build_variant.sh): in the product build HS_DBG(x) is the
    // constant 0, every variant branch folds away and the field is never read.
    int dbg_;
};

template <bool F32AUX, bool F16 = false>
__device__ __forceinline__ void epilogue_store(const GemmP& p, long cz, int m, int n, float v) {
    if (m >= p.M || n >= p.N) return;
    v *= n < p.scale_cols ? p.alpha * p.scale_alpha : p.alpha;
    if (p.flags & HS_EPI_BIAS) v += p.bias[(p.flags & HS_EPI_BIAS_M) ? m : n];
    const long idx = cz + (long)m * p.ldc + n;
    if (p.flags & HS_EPI_GELU) {
        if (!F32AUX) v = h2f<F16>(f2h<F16>(v));  // autocast: linear1 emits bf16 (fp16), GELU runs on that
        if (p.aux_out) {
            const float keep = (p.flags & HS_EPI_SAVE_DGELU) ? gelu_grad_f(v) : v;
            if (F32AUX) ((float*)p.aux_out)[idx] = keep; else ((bf16_t*)p.aux_out)[idx] = f2h<F16>(keep);
        }
        v = gelu_f(v);
    }
    if (p.flags & HS_EPI_GELU_BWD) {
        const float z = F32AUX ? ((const float*)p.aux)[idx] : h2f<F16>(((const bf16_t*)p.aux)[idx]);
        v *= (p.flags & HS_EPI_SAVE_DGELU) ? z : gelu_grad_f(z);
    }
    if (p.flags & HS_EPI_RESID) v += p.resid[idx];
    if (p.out_f32) {
        if (p.flags & HS_EPI_ATOMIC) atomicAdd(&p.out_f32[idx], v); else p.out_f32[idx] = v;
    }
    if (p.out_bf16) ((bf16_t*)p.out_bf16)[idx] = f2h<F16>(v);
}

// ------------------------------------------------------------------------------------------------------
// bf16: 256x256x64 (wave tile 128x64) or 256x128x64 (wave tile 64x64) block tiles, 8 waves (GCfg below).
// Operand tiles go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR staging, no ds_write), with
// the buffer descriptor's range check zero-filling every M/N/K tail.  LDS-DMA writes lane-linearly, so the
// bank swizzle of images R/T is applied to the per-lane SOURCE address (cdna guide rule 21) and again on the
// fragment reads.  Stage ring (2 or 3 deep): the prefetched tile is issued right after the barrier of step t, behind
// the MFMA groups; waits are counted (s_waitcnt vmcnt(6) = "all but the newest tile's 6 loads"), barriers raw.
// The epilogue is staged through LDS (see the kernel's tail) so that stores/atomics are whole row segments.
// ------------------------------------------------------------------------------------------------------

// fast erf (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7) sharing exp(-x^2/2) between GELU and its derivative
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf_x) {
    const float u = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * u);
    const float e = __expf(-u * u);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float erf_abs = 1.0f - poly * e;
    cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
    pdf_x = x * 0.3989422804014327f * e;
}

// LDS-DMA of one operand tile.  One wave-instruction = 1 KiB of LDS (R image: 8 rows x 128 B; T image: 4 rows x
// 256 B).  Piece j of this wave (j < NP) is piece index j*8 + wave of the operand's stage area.
// dma_voff(): the per-lane byte offset of piece j at k0 = 0, HS_OOB_OFF for rows/cols outside the matrix.  Full
// K-tiles then only add a wave-uniform scalar offset (kt * kstep bytes); a partial last K-tile recomputes the
// offsets with the k bound (dma_voff with K) so the descriptor's range check zero-fills the k tail.
// image R16 (rows of 128 B read as 16x16x32 operands: lane = row & 15, chunk = 4*step + (lane >> 4)): chunk c of row r sits
// at c ^ ((r >> 1) & 7).  A ds_read_b128 lane group holds every row of the 16 once, 8 of each parity; the two rows of a
// 256-B bank row differ in parity, rows of equal parity and equal chunk differ in (r >> 1), and the group's two chunk
// values differ in bit 0 while their row sets {0-3,12-15} / {4-11} never differ by 1 in (r >> 1): conflict-free.
__device__ __forceinline__ int swzR16(int r) { return (r >> 1) & 7; }
// PERMN (B image of the 16x16x32 form, see big_mainloop16): image row [j1 j0 q1 q0 r1 r0] of a wave's 64 holds matrix row
// [j1 q1 q0 j0 r1 r0], so that the lane of the transposed accumulator owns 8 CONSECUTIVE output columns.
__device__ __forceinline__ int permN(int row) { return (row & ~0x1c) | ((row & 0x0c) << 1) | ((row & 0x10) >> 2); }
template <bool TR, bool M16 = false, bool PERMN = false>
__device__ __forceinline__ unsigned dma_voff(int j, int dim0, int DIM, int k0, int K, long ld, int wave, int lane) {
    const int piece = j * 8 + wave;
    if (!TR) {
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (M16 ? swzR16(row) : swzR(row));
        const int gi = dim0 + (PERMN ? permN(row) : row), gk = k0 + c * 8;
        return (gi < DIM && gk < K) ? (unsigned)(((long)gi * ld + gk) * 2) : HS_OOB_OFF;
    } else {
        const int sub = piece >> 4;                 // 128-wide sub-image (16 pieces each)
        const int row = (piece & 15) * 4 + (lane >> 4);
        const int c = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        const int gk = k0 + row, gi = dim0 + sub * 128 + c * 8;
        return (gk < K && gi < DIM) ? (unsigned)(((long)gk * ld + gi) * 2) : HS_OOB_OFF;
    }
}
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, char* lds_area, int j, int wave, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))(lds_area + (j * 8 + wave) * 1024), 16, voff, soff, 0, 0);
}
// fragment of the 32-wide block starting at `sub` (0..255) of an operand image, k-step ks
#ifdef HS_ABLATE_TR   // ablation build only: read T images with plain row reads (wrong data, same bytes)
#define HS_TR_AS_ROW 1
#else
#define HS_TR_AS_ROW 0
#endif
template <bool TR>
__device__ __forceinline__ bf16x8 g_frag(const char* lds, int sub, int ks, int lane) {
    if (!TR || HS_TR_AS_ROW) {
        return *(const bf16x8*)(lds + offR(sub + (lane & 31), 2 * ks + (lane >> 5)));
    } else {
        const char* img = lds + (sub >> 7) * 16384;
        const int g = lane >> 4, h = g >> 1, p = lane & 3, q = (lane >> 2) & 3;
        const int col = (sub & 127) + 16 * (g & 1) + 4 * p;
        const int r1 = ks * 16 + 8 * h + q;
        const bf16x4 a = lds_tr16(img + offT(r1, col >> 3) + 8 * (p & 1));
        const bf16x4 b = lds_tr16(img + offT(r1 + 4, col >> 3) + 8 * (p & 1));
        return cat4(a, b);
    }
}

// 4 consecutive n of one output row m
// Vector epilogue for 4 consecutive n of one output row, split in two so that a group of steps can do ALL its math
// first and ALL its stores afterwards: vmcnt is an in-order counter and these steps sit in divergent control flow, so
// hipcc waits vmcnt(0) before every use of a loaded value — with a store between two uses that is one store round trip
// per step (~32 per block), which cost more than the K=768 main loop.
// epi_math: v -> final value (bias, GELU(+value to save), GELU', residual); *keep = bf16x4 to store to aux_out.
template <bool F16 = false>
__device__ __forceinline__ f32x4 epi_math(const GemmP& p, const int flags, int m, f32x4 v, f32x4 res, u32x2 zb_in, f32x4 bias_n, float alpha_n, u32x2* keep) {
    v *= alpha_n;
    if (flags & HS_EPI_BIAS) {
        if (flags & HS_EPI_BIAS_M) v += p.bias[m]; else v += bias_n;      // column bias: hoisted by the caller
    }
    if (flags & HS_EPI_GELU) {
        u32x2 zb;
        zb[0] = pack2h<F16>(v[0], v[1]); zb[1] = pack2h<F16>(v[2], v[3]);       // autocast: linear1 emits bf16 (fp16)
        v[0] = unpack_lo<F16>(zb[0]); v[1] = unpack_hi<F16>(zb[0]);
        v[2] = unpack_lo<F16>(zb[1]); v[3] = unpack_hi<F16>(zb[1]);
        f32x4 dg;
#pragma unroll
        for (int e = 0; e < 4; ++e) { float c, px; gelu_parts(v[e], c, px); v[e] *= c; dg[e] = c + px; }
        // HS_EPI_SAVE_DGELU: keep gelu'(z) (it shares exp(-z^2/2) with gelu) instead of z, so the backward epilogue is
        // one multiply instead of an erf + exp per element
        if (flags & HS_EPI_SAVE_DGELU) { zb[0] = pack2h<F16>(dg[0], dg[1]); zb[1] = pack2h<F16>(dg[2], dg[3]); }
        *keep = zb;
    }
    if (flags & HS_EPI_GELU_BWD) {
        const float z[4] = {unpack_lo<F16>(zb_in[0]), unpack_hi<F16>(zb_in[0]), unpack_lo<F16>(zb_in[1]), unpack_hi<F16>(zb_in[1])};
        if (flags & HS_EPI_SAVE_DGELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= z[e];                  // aux already holds gelu'(z)
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float c, px; gelu_parts(z[e], c, px); v[e] *= (c + px); }
        }
    }
    if (flags & HS_EPI_RESID) v += res;
    return v;
}
template <bool F16 = false>
__device__ __forceinline__ void epi_store(const GemmP& p, long idx, f32x4 v, u32x2 keep) {
    if ((p.flags & HS_EPI_GELU) && p.aux_out) *(u32x2*)((bf16_t*)p.aux_out + idx) = keep;
    if (p.out_f32) *(f32x4*)(p.out_f32 + idx) = v;
    if (p.out_bf16) {
        u32x2 o;
        o[0] = pack2h<F16>(v[0], v[1]); o[1] = pack2h<F16>(v[2], v[3]);
        *(u32x2*)((bf16_t*)p.out_bf16 + idx) = o;
    }
}

// Tile configurations.  BIG: 256x256 block, 8 waves as 2(M) x 4(N), wave tile 128x64 (4x2 MFMA blocks, 32 MFMAs per
// K-tile per wave), 2 stages x 64 KiB.  !BIG: 256x128 block, waves 4 x 2, wave tile 64x64, 3 stages x 48 KiB.
// HALF (a flag of the !BIG NT kernel, round 5): 128x128 block, the same 4 x 2 waves on 32x64 wave tiles, its own main loop
// (half_mainloop) and the !BIG epilogue on 32-row pieces.
template <bool BIG> struct GCfg {
    static constexpr int MB = BIG ? 4 : 2;            // 32-row MFMA blocks per wave along M
    static constexpr int WAVES_N = BIG ? 4 : 2;
    static constexpr int BN = WAVES_N * 64;
    static constexpr int NSTAGE = BIG ? 2 : 3;
    static constexpr int PB = BIG ? 4 : 2;            // B-operand DMA pieces per wave per K-tile (A: always 4)
    static constexpr int PIECES = 4 + PB;
    static constexpr int STAGE = 32768 + BN * 128;
    static constexpr int EPI = 8 * 64 * 68 * 4;       // per-wave 64x64 fp32 staging slices (68-float rows)
    static constexpr int SMEM = STAGE * NSTAGE > EPI ? STAGE * NSTAGE : EPI;
};
#define G_EPI_LD 68   // floats per staged row (64 + 4 pad)

// ---- transposed fragment reads from inline asm ---------------------------------------------------------------
// hipcc cannot prove that the ds_read_tr builtin does not alias the LDS-DMA writes in flight, so it puts
// `s_waitcnt vmcnt(0)` in front of every group of them — draining the prefetch ring at every k-step (NN/TN ran at
// 0.6x of NT for that reason alone).  Issued from asm they are invisible to that pass; their completion is waited
// for explicitly (tr_settle), with the fragment registers as "+v" operands so no consumer can move above the wait.
__device__ __forceinline__ unsigned lds_off(const char* p) { return (unsigned)(__UINTPTR_TYPE__)(LDS_PTR(const char))p; }
// byte addresses (k-step 0) of the two 4-row groups of the 32-wide block at `sub` of a T image
__device__ __forceinline__ void tr_bases(const char* lds, int sub, int lane, unsigned (&b)[2]) {
    const int g = lane >> 4, h = g >> 1, p = lane & 3, q = (lane >> 2) & 3;
    const int col = (sub & 127) + 16 * (g & 1) + 4 * p;
    const unsigned img = lds_off(lds) + (sub >> 7) * 16384 + 8 * (p & 1);
    b[0] = img + offT(8 * h + q, col >> 3);
    b[1] = img + offT(8 * h + q + 4, col >> 3);
}
// k-step KS adds 16 rows = 4096 B (the swizzle of a row depends on row & 15 only): an immediate offset
template <int KS>
__device__ __forceinline__ void tr_issue(u32x4& f, const unsigned (&b)[2]) {
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"
                 : "=&v"(lo), "=&v"(hi) : "v"(b[0]), "v"(b[1]), "i"(KS * 4096) : "memory");
    f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}
__device__ __forceinline__ bf16x8 as_frag(u32x4 f) { return __builtin_bit_cast(bf16x8, f); }
// wait for every LDS read issued so far; the registers are "+v" so that nothing that consumes them is scheduled above
__device__ __forceinline__ void tr_settle(u32x4& a, u32x4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory"); }
__device__ __forceinline__ void tr_settle(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

// one K-tile of MFMA work for a wave, with the DMA pieces of the prefetched K-tile (if any: `issue`, wave-uniform)
// issued behind the MFMA groups.  A single code path keeps the 64-128 accumulator registers out of phi copies.
template <bool ATR, bool BTR, bool BIG, bool F16 = false>
__device__ __forceinline__ void ktile_step(f32x16 (&acc)[GCfg<BIG>::MB][2], const char* a_l, const char* b_l, char* nxt, int row0, int col0,
                                           int wave, int lane, __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB,
                                           const unsigned (&va)[4], const unsigned (&vb)[GCfg<BIG>::PB], unsigned sa, unsigned sb, bool issue) {
    typedef GCfg<BIG> C;
    typedef std::integral_constant<int, 0> K0;
    typedef std::integral_constant<int, 1> K1;
    typedef std::integral_constant<int, 2> K2;
    typedef std::integral_constant<int, 3> K3;
    auto piece = [&](int q) {
        if (issue) {
            if (q < 4) dma_piece(rsA, nxt, q, wave, va[q], sa); else dma_piece(rsB, nxt + 32768, q - 4, wave, vb[q - 4], sb);
        }
    };
    unsigned tb[2][2];
    if (BTR) {
        tr_bases(b_l, col0, lane, tb[0]);
        tr_bases(b_l, col0 + 32, lane, tb[1]);
    }
    if (BIG) {
        // 128x64 wave tile: the accumulator already takes 128 of the 256 registers available at 2 waves/SIMD, so the
        // A fragments stream one 32-row block at a time (2 MFMAs each) instead of being double-buffered.
        auto step = [&](auto ks_tag) {
            constexpr int KS = decltype(ks_tag)::value;
            u32x4 b0, b1;
            if (BTR) {
                tr_issue<KS>(b0, tb[0]);
                tr_issue<KS>(b1, tb[1]);
                tr_settle(b0, b1);
            } else {
                b0 = __builtin_bit_cast(u32x4, g_frag<false>(b_l, col0, KS, lane));
                b1 = __builtin_bit_cast(u32x4, g_frag<false>(b_l, col0 + 32, KS, lane));
            }
#pragma unroll
            for (int i = 0; i < C::MB; ++i) {
                const bf16x8 a = g_frag<ATR>(a_l, row0 + 32 * i, KS, lane);   // (TN never uses the big tile: hs_gemm)
                acc[i][0] = mfma32<F16>(a, as_frag(b0), acc[i][0]);
                acc[i][1] = mfma32<F16>(a, as_frag(b1), acc[i][1]);
            }
            piece(2 * KS);
            piece(2 * KS + 1);                                       // 8 pieces: 2 per k-step
        };
        step(K0{}); step(K1{}); step(K2{}); step(K3{});
    } else {
        // Software pipeline inside the K-tile: the fragments of k-step ks+1 are read BEFORE the MFMAs of k-step ks are
        // issued and the DMA pieces go after those MFMAs, so LDS read latency (~200 cycles) and DMA issue cost (~65
        // cycles per piece) elapse under the matrix pipe.
        unsigned ta[2][2];
        if (ATR) {
            tr_bases(a_l, row0, lane, ta[0]);
            tr_bases(a_l, row0 + 32, lane, ta[1]);
        }
        u32x4 fa[2][2], fb[2][2];
        auto fetch = [&](auto ks_tag, auto set_tag) {
            constexpr int KS = decltype(ks_tag)::value, S = decltype(set_tag)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (ATR) tr_issue<KS>(fa[S][i], ta[i]);
                else fa[S][i] = __builtin_bit_cast(u32x4, g_frag<false>(a_l, row0 + 32 * i, KS, lane));
                if (BTR) tr_issue<KS>(fb[S][i], tb[i]);
                else fb[S][i] = __builtin_bit_cast(u32x4, g_frag<false>(b_l, col0 + 32 * i, KS, lane));
            }
        };
        auto settle = [&](auto set_tag) {
            constexpr int S = decltype(set_tag)::value;
            if (ATR || BTR) tr_settle(fa[S][0], fa[S][1], fb[S][0], fb[S][1]);
        };
        auto step = [&](auto ks_tag) {
            constexpr int KS = decltype(ks_tag)::value, CUR = KS & 1, NX = CUR ^ 1;
            if (KS < 3) fetch(std::integral_constant<int, (KS < 3 ? KS + 1 : 3)>{}, std::integral_constant<int, NX>{});
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = mfma32<F16>(as_frag(fa[CUR][i]), as_frag(fb[CUR][j]), acc[i][j]);
            if (KS < 2) { piece(2 * KS); piece(2 * KS + 1); } else piece(KS + 2);   // 6 pieces: (0,1)(2,3)(4)(5)
            if (KS < 3) settle(std::integral_constant<int, NX>{});
        };
        fetch(K0{}, K0{});
        settle(K0{});
        step(K0{}); step(K1{}); step(K2{}); step(K3{});
    }
}

// diagnostic cycle sums (dbg == 4), see hs_debug_gemm_stamps
__device__ unsigned long long g_gemm_stamps[8];

// ---- main loop of the 256x256 tile, software-pipelined by hand -----------------------------------------------------
// Left to hipcc, the fragment reads of this tile are issued just in time (4 ds_read_b128, lgkmcnt(0), 4 MFMAs, 2 reads,
// wait, 2 MFMAs, ...): three exposed LDS round trips per 8 MFMAs, ~55 % of the matrix rate.  Here every LDS read is an
// asm statement: the fragments of k-step ks+1 are issued before the MFMAs of k-step ks and retired with a COUNTED
// lgkmcnt, across the K-tile boundary too: the tile barrier sits between k-steps 2 and 3, so that k-step 3 prefetches
// the next tile's first fragments and issues half of the DMA of the tile after it.
//   per K-tile t (stage s = t & 1):   ks0: read F(t,1) | MFMA F(t,0) | DMA pieces 4-7 of tile t+1 -> stage s^1
//                                     ks1: read F(t,2) | MFMA F(t,1)
//                                     ks2: read F(t,3) | MFMA F(t,2)
//                                     lgkmcnt(0) (every read of stage s retired), vmcnt(0) (tile t+1 landed), s_barrier
//                                     ks3: read F(t+1,0) from stage s^1 | MFMA F(t,3) | DMA pieces 0-3 of tile t+2 -> stage s
// Requires K % 64 == 0 (hs_gemm picks the 256x128 tile otherwise).
template <bool BTR, bool F16 = false>
__device__ __forceinline__ void big_mainloop(f32x16 (&acc)[4][2], char* smem, int row0, int col0, int wave, int lane,
                                             __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB, const unsigned (&va)[4],
                                             const unsigned (&vb)[4], unsigned ka, unsigned kb, int kt0, int kt1, int dbg) {
    constexpr int STAGE = GCfg<true>::STAGE;       // 65536: beyond a ds_read immediate, so stage bases live in registers
    const int h = lane >> 5, c = lane & 31;
    unsigned ab0[4], bb0[4], tbb0[2][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        ab0[ks] = lds_off(smem) + offR(row0 + c, 2 * ks + h);
        bb0[ks] = lds_off(smem) + 32768 + offR(col0 + c, 2 * ks + h);
    }
    if (BTR) {
        tr_bases(smem + 32768, col0, lane, tbb0[0]);
        tr_bases(smem + 32768, col0 + 32, lane, tbb0[1]);
    }
    u32x4 fa[2][4], fb[2][2];
    // fragments of k-step KS of the tile in the stage at byte offset so -> set S
    auto rd = [&](auto set_tag, auto ks_tag, unsigned so) {
        constexpr int S = decltype(set_tag)::value, KS = decltype(ks_tag)::value;
        const unsigned a = ab0[KS] + so;
        ldsr128<0>(fa[S][0], a); ldsr128<4096>(fa[S][1], a); ldsr128<8192>(fa[S][2], a); ldsr128<12288>(fa[S][3], a);
        if (!BTR) {
            const unsigned b = bb0[KS] + so;
            ldsr128<0>(fb[S][0], b); ldsr128<4096>(fb[S][1], b);
        } else {
            unsigned t0[2] = {tbb0[0][0] + so, tbb0[0][1] + so}, t1[2] = {tbb0[1][0] + so, tbb0[1][1] + so};
            tr_issue<KS>(fb[S][0], t0);
            tr_issue<KS>(fb[S][1], t1);
        }
    };
    auto settle = [&](auto set_tag, auto cnt_tag) {
        constexpr int S = decltype(set_tag)::value, CNT = decltype(cnt_tag)::value;
        lds_wait6<CNT>(fa[S][0], fa[S][1], fa[S][2], fa[S][3], fb[S][0], fb[S][1]);
    };
    auto mma = [&](auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = mfma32<F16>(as_frag(fa[S][i]), as_frag(fb[S][j]), acc[i][j]);
    };
    auto dma = [&](int kt, unsigned so, int q0) {          // pieces q0 .. q0+3 of tile kt (0-3: A, 4-7: B)
        if (HS_DBG(dbg) == 5) kt = kt0;                              // diagnostic: every tile re-reads the first one (cache hits)
        char* sa = smem + so;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q0 == 0) dma_piece(rsA, sa, q, wave, va[q], (unsigned)kt * ka);
            else dma_piece(rsB, sa + 32768, q, wave, vb[q], (unsigned)kt * kb);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    constexpr int NR = BTR ? 8 : 6;                          // LDS read instructions per k-step
    typedef std::integral_constant<int, NR> CN;
    dma(kt0, 0, 0);
    dma(kt0, 0, 4);
    if (kt0 + 1 < kt1) {
        dma(kt0 + 1, STAGE, 0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    rd(I0{}, I0{}, 0u);
    unsigned so = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool has1 = kt + 1 < kt1, has2 = kt + 2 < kt1;
        const unsigned so1 = so ^ (unsigned)STAGE;
        rd(I1{}, I1{}, so); settle(I0{}, CN{}); mma(I0{});
        if (has1) dma(kt + 1, so1, 4);
        rd(I0{}, I2{}, so); settle(I1{}, CN{}); mma(I1{});
        rd(I1{}, I3{}, so); settle(I0{}, CN{}); mma(I0{});
        settle(I1{}, I0{});                                  // lgkmcnt(0): every read of this stage has landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (has1) rd(I0{}, I0{}, so1);
        mma(I1{});
        if (has2) dma(kt + 2, so, 0);
        so = so1;
    }
}

// ---- the same tile on v_mfma_f32_16x16x32_bf16 (NT) ----------------------------------------------------------------
// Bare MFMA loops on random data run 12.5 % faster on 16x16x32 than on 32x32x16 (the chip holds a higher clock on the small
// shape; tools/micro/mfma_shapes.cpp).  Wave tile 128x64 = 8 x 4 tiles of 16x16, two 32-deep steps per K-tile, each step in
// two halves of 16 MFMAs: four phases per K-tile with the same pipeline as big_mainloop (reads one phase ahead, barrier
// between phases 2 and 3).  Operand images are R16 (see swzR16); A and B fragments are one ds_read_b128 each.
template <int CNT>
__device__ __forceinline__ void lds_wait8g(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e, u32x4& f, u32x4& g, u32x4& i) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(i) : "i"(CNT) : "memory");
}
// Stages (round 3): THREE for A, two for B.  With two 64-KiB stages (rounds 1-2) this loop ran at ~73 % of the MFMA rate, held by the
// LATENCY of L2 misses, not by bandwidth: nearly every K-tile has a piece that comes from beyond the L2 (the A
// panel is streamed from HBM once: compulsory), a K-tile is complete when its slowest piece has landed, and two stages gave A one
// K-tile of prefetch distance (B 0.75).  This kernel's epilogue does not stage through LDS, so the whole 160 KiB are free for
// operands: 3 x 32 KiB of A + 2 x 32 KiB of B (qkv 899 -> 830 us, fc2 1013 -> 931, 8192^3 1127 -> 1252 TFLOP/s; in the engine the four
// instantiations gained 8-18 %, the step 3.1 %).  A(t+2) is issued in phase 0
// of tile t (1.75 K-tiles before the barrier that needs it), B(t+2) in phase 3 of tile t (1.0 K-tile); the issue ORDER B(t+1) < A(t+2)
// makes "everything but the newest 4 pieces" exactly the pieces barrier(t) needs (A(t+1), B(t+1)), so the wait is a counted vmcnt(4).
//   LDS: A stages at 0 / 32 K / 64 K, B stages at 96 K / 128 K.
template <bool F16 = false, bool PRO = true>      // PRO = false: the caller has issued the prologue pieces (gemm_nt16_persist_kernel)
__device__ __forceinline__ void big_mainloop16(f32x4 (&acc)[8][4], char* smem, int row0, int col0, int wave, int lane,
                                                  __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB, const unsigned (&va)[4],
                                                  const unsigned (&vb)[4], unsigned ka, unsigned kb, int kt0, int kt1) {
    constexpr unsigned AST = 32768u, BBASE = 98304u;
    const int r = lane & 15, q = lane >> 4, f = swzR16(r);
    unsigned ab0[2], bb0[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        ab0[st] = lds_off(smem) + (row0 + r) * 128 + (((4 * st + q) ^ f) << 4);
        bb0[st] = lds_off(smem) + BBASE + (col0 + r) * 128 + (((4 * st + q) ^ f) << 4);
    }
    u32x4 fa[2][4], fb[2][4];
    auto rdA = [&](auto set_tag, auto st_tag, auto hf_tag, unsigned sa) {
        constexpr int S = decltype(set_tag)::value, ST = decltype(st_tag)::value, HF = decltype(hf_tag)::value;
        const unsigned a = ab0[ST] + sa;
        ldsr128<(HF * 4 + 0) * 2048>(fa[S][0], a); ldsr128<(HF * 4 + 1) * 2048>(fa[S][1], a);
        ldsr128<(HF * 4 + 2) * 2048>(fa[S][2], a); ldsr128<(HF * 4 + 3) * 2048>(fa[S][3], a);
    };
    auto rdB = [&](auto set_tag, auto st_tag, unsigned sb) {
        constexpr int S = decltype(set_tag)::value, ST = decltype(st_tag)::value;
        const unsigned b = bb0[ST] + sb;
        ldsr128<0>(fb[S][0], b); ldsr128<2048>(fb[S][1], b); ldsr128<4096>(fb[S][2], b); ldsr128<6144>(fb[S][3], b);
    };
    auto mma = [&](auto aset_tag, auto bset_tag, auto hf_tag) {
        constexpr int SA = decltype(aset_tag)::value, SB = decltype(bset_tag)::value, HF = decltype(hf_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[HF * 4 + i][j] = mfma16<F16>(as_frag(fb[SB][j]), as_frag(fa[SA][i]), acc[HF * 4 + i][j]);
    };
    auto dmaA = [&](int kt, unsigned sa) {
#pragma unroll
        for (int p = 0; p < 4; ++p) dma_piece(rsA, smem + sa, p, wave, va[p], (unsigned)kt * ka);
    };
    auto dmaB = [&](int kt, unsigned sb) {
#pragma unroll
        for (int p = 0; p < 4; ++p) dma_piece(rsB, smem + BBASE + sb, p, wave, vb[p], (unsigned)kt * kb);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    // prologue, in the order the counted waits rely on: A(0), B(0), A(1), B(1)
    if (PRO) {
        dmaA(kt0, 0u);
        dmaB(kt0, 0u);
    }
    if (kt0 + 1 < kt1) {
        if (PRO) {
            dmaA(kt0 + 1, AST);
            dmaB(kt0 + 1, AST);
        }
        // (!PRO: the pieces were issued in front of the previous tile's epilogue, whose stores are YOUNGER: "at most 8 operations
        // outstanding" then still implies A(0), B(0) landed -- loads retire in issue order -- it only also waits for most stores)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    rdA(I0{}, I0{}, I0{}, 0u);
    rdB(I0{}, I0{}, 0u);
    unsigned sa = 0, sb = 0;                                   // stages of tile kt: A (kt - kt0) % 3, B (kt - kt0) % 2
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool has1 = kt + 1 < kt1, has2 = kt + 2 < kt1;
        const unsigned sa1 = sa == 2 * AST ? 0u : sa + AST, sa2 = sa1 == 2 * AST ? 0u : sa1 + AST, sb1 = sb ^ AST;
        // phase 0: (step 0, rows 0-63)
        rdA(I1{}, I0{}, I1{}, sa);
        lds_wait8g<4>(fa[0][0], fa[0][1], fa[0][2], fa[0][3], fb[0][0], fb[0][1], fb[0][2], fb[0][3]);
        mma(I0{}, I0{}, I0{});
        if (has2) dmaA(kt + 2, sa2);                            // the stage tile kt-1 was read from (every wave passed barrier(kt-1) since)
        // phase 1: (step 0, rows 64-127)
        rdA(I0{}, I1{}, I0{}, sa);
        rdB(I1{}, I1{}, sb);
        lds_wait4<8>(fa[1][0], fa[1][1], fa[1][2], fa[1][3]);
        mma(I1{}, I0{}, I1{});
        // phase 2: (step 1, rows 0-63)
        rdA(I1{}, I1{}, I1{}, sa);
        lds_wait8g<4>(fa[0][0], fa[0][1], fa[0][2], fa[0][3], fb[1][0], fb[1][1], fb[1][2], fb[1][3]);
        mma(I0{}, I1{}, I0{});
        lds_wait4<0>(fa[1][0], fa[1][1], fa[1][2], fa[1][3]);       // every read of this tile's stages has landed
        if (has2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // A(kt+1), B(kt+1) landed; A(kt+2) stays in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // phase 3: (step 1, rows 64-127) || first fragments of the next tile
        if (has1) { rdA(I0{}, I0{}, I0{}, sa1); rdB(I0{}, I0{}, sb1); }
        mma(I1{}, I1{}, I1{});
        if (has2) dmaB(kt + 2, sb);                             // this tile's B stage: free since the barrier
        sa = sa1; sb = sb1;
    }
}

// ---- main loop of the 128x128 tile (HALF, NT): 8 waves as 4(M) x 2(N) on 32x64 wave tiles, FIVE 32-KiB stages ------------
// For launches whose 256x128 tiles would leave more than half of the CUs idle (hs_gemm: the N = 768 outputs at one volume per GPU,
// 102 tiles on 256 CUs).  Half the MFMA work per K-tile and wave (8 MFMAs) means half the time to hide a K-tile's load latency
// in: the ring is five stages deep (A 16 KiB + B 16 KiB each = all 160 KiB; the LDS-staged epilogue reuses it), i.e. four K-tiles
// of prefetch distance, and the pipeline is big_mainloop's -- asm-issued fragment reads one k-step ahead with counted lgkmcnt, the
// tile barrier between k-steps 2 and 3, k-step 3 reading the next tile's first fragments; the tile NS ahead is issued into the stage
// that barrier has just freed.  Waits: loads retire in issue order, each tile is 4 pieces per wave, so "tile t+1 has landed" is
// vmcnt(4 x tiles issued after it).  Same k order per output element as every other tile shape: bit-identical results.
template <int CNT>
__device__ __forceinline__ void lds_wait3(u32x4& a, u32x4& b, u32x4& c) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "i"(CNT) : "memory"); }
__device__ __forceinline__ void vm_wait_tiles4(int newer) {            // at most 4 * newer vector-memory operations still outstanding
    if (newer >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (newer == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (newer == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <bool F16 = false>
__device__ __forceinline__ void half_mainloop(f32x16 (&acc)[2], char* smem, int row0, int col0, int wave, int lane,
                                              __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB, const unsigned (&va)[4],
                                              const unsigned (&vb)[2], int kt0, int kt1) {
    constexpr unsigned ST = 32768u, BOFF = 16384u;
    constexpr int NS = 5;
    const int h = lane >> 5, c = lane & 31;
    unsigned ab0[4], bb0[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        ab0[ks] = lds_off(smem) + offR(row0 + c, 2 * ks + h);
        bb0[ks] = lds_off(smem) + BOFF + offR(col0 + c, 2 * ks + h);
    }
    u32x4 fa[2], fb[2][2];
    auto rd = [&](auto set_tag, auto ks_tag, unsigned so) {
        constexpr int S = decltype(set_tag)::value, KS = decltype(ks_tag)::value;
        ldsr128<0>(fa[S], ab0[KS] + so);
        ldsr128<0>(fb[S][0], bb0[KS] + so);
        ldsr128<4096>(fb[S][1], bb0[KS] + so);
    };
    auto settle = [&](auto set_tag, auto cnt_tag) {
        constexpr int S = decltype(set_tag)::value, CNT = decltype(cnt_tag)::value;
        lds_wait3<CNT>(fa[S], fb[S][0], fb[S][1]);
    };
    auto mma = [&](auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        acc[0] = mfma32<F16>(as_frag(fa[S]), as_frag(fb[S][0]), acc[0]);
        acc[1] = mfma32<F16>(as_frag(fa[S]), as_frag(fb[S][1]), acc[1]);
    };
    auto dma = [&](int kt, unsigned so) {
#pragma unroll
        for (int q = 0; q < 2; ++q) dma_piece(rsA, smem + so, q, wave, va[q], (unsigned)kt * 128u);
#pragma unroll
        for (int q = 0; q < 2; ++q) dma_piece(rsB, smem + so + BOFF, q, wave, vb[q], (unsigned)kt * 128u);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
#pragma unroll
    for (int i = 0; i < NS; ++i)
        if (kt0 + i < kt1) dma(kt0 + i, (unsigned)i * ST);
    vm_wait_tiles4(min(NS - 1, kt1 - 1 - kt0));
    __builtin_amdgcn_s_barrier();
    rd(I0{}, I0{}, 0u);
    unsigned so = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool has1 = kt + 1 < kt1;
        const unsigned so1 = so == (NS - 1) * ST ? 0u : so + ST;
        rd(I1{}, I1{}, so); settle(I0{}, I3{}); mma(I0{});
        rd(I0{}, I2{}, so); settle(I1{}, I3{}); mma(I1{});
        rd(I1{}, I3{}, so); settle(I0{}, I3{}); mma(I0{});
        settle(I1{}, I0{});                                  // lgkmcnt(0): every read of this stage has landed
        vm_wait_tiles4(has1 ? min(NS - 2, kt1 - 2 - kt) : 0);  // tile kt+1 landed; up to three younger tiles stay in flight
        __builtin_amdgcn_s_barrier();
        if (has1) rd(I0{}, I0{}, so1);
        mma(I1{});
        if (kt + NS < kt1) dma(kt + NS, so);                 // this tile's stage: free since the barrier
        so = so1;
    }
}

// diagnostic (dbg == 4): per-phase cycle sums of the epilogue, wave 0 lane 0 of every block (tools/kbench)
extern "C" void hs_debug_gemm_stamps(unsigned long long* out, int reset) {
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps), sizeof(g_gemm_stamps));
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), z, sizeof(z)); }
}
// diagnostic (dbg == 22): per-block timeline {start, main loop end, epilogue end, HW_ID | XCC_ID << 32} for tools/kbench
__device__ unsigned long long g_gemm_trace[4 * 4096];
extern "C" void hs_debug_gemm_trace(unsigned long long* out, int nblocks) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_trace), sizeof(unsigned long long) * 4 * (nblocks < 4096 ? nblocks : 4096));
}
#define TRACE(i) do { if (HS_DBG(p.dbg_) == 22 && threadIdx.x == 0 && blockIdx.x < 4096) g_gemm_trace[blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP(i) do { if (HS_DBG(p.dbg_) == 4 || HS_DBG(p.dbg_) == 7) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) atomicAdd(&g_gemm_stamps[i], t_ - t_prev); t_prev = t_; } } while (0)

// (asm with register constraints lives in __device__ helpers, never directly in a templated __global__ body: hipcc's
// host pass then silently drops the kernel's launch stub)
__device__ __forceinline__ void keep_alive(const f32x16& v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void keep4(const f32x4& v) { asm volatile("" ::"v"(v)); }

// virtual block b of nb -> (row tile, column tile, slice): see the comment on the XCD-aware order in gemm_bf16_kernel
__device__ __forceinline__ void tile_order(int b, int nb, int tiles_m, int tiles_n, int& tm, int& tn, int& z, int GM, int NP = 0) {
    const int q = nb >> 3, r = nb & 7, x = b & 7;
    int v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    const int tiles = tiles_m * tiles_n;
    z = v / tiles;
    v -= z * tiles;
    // column panels (NP column tiles each, the last one possibly narrower), slowest inside a slice: the B rows of ONE panel stay in the
    // XCD's L2 while every row tile passes by (a 3072-column weight is 4.7 MB: as one panel it streams through the 4-MiB L2 once per
    // group of row tiles)
    int c0 = 0, wdt = tiles_n;
    if (NP > 0 && NP < tiles_n) {
        const int per_panel = tiles_m * NP, last = (tiles_n - 1) / NP;
        const int pi = min(v / per_panel, last);
        c0 = pi * NP;
        wdt = min(NP, tiles_n - c0);
        v -= pi * per_panel;
    }
    const int per_group = GM * wdt;
    const int g = v / per_group, w = v - g * per_group;
    const int gm = min(GM, tiles_m - g * GM);
    tm = g * GM + w % gm;
    tn = c0 + w / gm;
}

// lanes l and l ^ 8 (same 16-lane row) trade one 16-B piece: afterwards a (lanes & 15) < 8 lane holds {own a, partner's a} and the
// others {partner's b, own b} -- see epilogue16
__device__ __forceinline__ void half_line_swap(u32x4& a, u32x4& b, bool lo8) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned send = lo8 ? b[e] : a[e];
        const unsigned got = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0x128, 0xf, 0xf, false);   // row_ror:8
        if (lo8) b[e] = got; else a[e] = got;
    }
}
template <int EF, int NJ = 4, int J0 = 0, bool F16 = false>
__device__ __forceinline__ void epilogue16(const GemmP& p, const f32x4 (&acc16)[8][NJ], int mw, int nw, long cz, int lane) {
    // ---- 16x16x32 form: the MFMAs ran with the operands SWAPPED (D' = B-tile x A-tile), so a lane holds, for output row
    // m = 16 i + (lane & 15), the columns 16 j + 4 (lane >> 4) + r of the B IMAGE -- which permN laid out so that tiles
    // 2jp, 2jp+1 together are the 8 consecutive matrix columns 32 jp + 8 (lane >> 4) .. +7.  The epilogue therefore runs
    // straight out of the accumulators: no LDS round trip (writing and re-reading the 256-KiB fp32 tile was ~4k of the
    // ~9k epilogue cycles per tile), each store instruction = 16 rows x 64 B (bf16) or 2 x (16 rows x 128 B) (fp32).
static_assert(EF >= 0, "the 16x16x32 kernels are launched with compile-time epilogue flags");
    constexpr int flags = EF;
    const int l15 = lane & 15, q = lane >> 4;
    const bool want_res = (flags & HS_EPI_RESID) != 0;
    int ncol[2];
    bool n_in[2];
    f32x4 bias_lo[2], bias_hi[2];
    float alpha_n[2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        ncol[jp] = nw + 32 * jp + 8 * q;
        n_in[jp] = ncol[jp] < p.N;
        alpha_n[jp] = ncol[jp] < p.scale_cols ? p.alpha * p.scale_alpha : p.alpha;
        bias_lo[jp] = f32x4{0.f, 0.f, 0.f, 0.f};
        bias_hi[jp] = bias_lo[jp];
        if ((flags & HS_EPI_BIAS) && n_in[jp]) { bias_lo[jp] = *(const f32x4*)(p.bias + ncol[jp]); bias_hi[jp] = *(const f32x4*)(p.bias + ncol[jp] + 4); }
    }
    // groups of 4 steps (two 16-row tiles x two column halves): all loads in flight -> all math -> all stores (see epi_math).
    // The residual loads are software-pipelined one group ahead and UNCONDITIONAL (a row outside the matrix reads element 0
    // and is never stored): straight-line code, so hipcc retires them with counted vmcnt instead of draining the stores too.
    f32x4 rlo[2][4], rhi[2][4];
    auto load_res = [&](auto g_tag) {
        constexpr int G = decltype(g_tag)::value;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int i = 2 * G + (st >> 1), jp = st & 1;
            const int m = mw + 16 * i + l15;
            const long ix = (m < p.M && n_in[jp]) ? cz + (long)m * p.ldc + ncol[jp] : 0l;
            rlo[G & 1][st] = *(const f32x4*)(p.resid + ix);
            rhi[G & 1][st] = *(const f32x4*)(p.resid + ix + 4);
        }
    };
    // the saved gelu'(z) tile of a GELU' epilogue (dgrad of fc2 against the transposed weight copy): loaded like the residual, 16 B
    // (8 bf16 columns) per step, one group ahead, unconditionally; its column sums (the bias gradient of fc1) accumulate per lane
    // over the wave's 128 rows and are reduced across the 16 lanes of a row group at the end
    constexpr bool want_aux_in = (flags & HS_EPI_GELU_BWD) != 0;
    u32x4 zin[2][4];
    auto load_aux = [&](auto g_tag) {
        constexpr int G = decltype(g_tag)::value;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int i = 2 * G + (st >> 1), jp = st & 1;
            const int m = mw + 16 * i + l15;
            const long ix = (m < p.M && n_in[jp]) ? cz + (long)m * p.ldc + ncol[jp] : 0l;
            zin[G & 1][st] = *(const u32x4*)((const bf16_t*)p.aux + ix);
        }
    };
    f32x4 cs_lo[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, cs_hi[2] = {cs_lo[0], cs_lo[0]};
    if (want_res) load_res(std::integral_constant<int, 0>{});
    if (want_aux_in) load_aux(std::integral_constant<int, 0>{});
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 vlo[4], vhi[4];
        u32x4 zb[4];
        bool in[4];
        long idx[4];
        if (want_res) {
            if (g == 0) load_res(std::integral_constant<int, 1>{});
            if (g == 1) load_res(std::integral_constant<int, 2>{});
            if (g == 2) load_res(std::integral_constant<int, 3>{});
        }
        if (want_aux_in) {
            if (g == 0) load_aux(std::integral_constant<int, 1>{});
            if (g == 1) load_aux(std::integral_constant<int, 2>{});
            if (g == 2) load_aux(std::integral_constant<int, 3>{});
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int i = 2 * g + (st >> 1), jp = st & 1;
            const int m = mw + 16 * i + l15;
            in[st] = m < p.M && n_in[jp];
            idx[st] = cz + (long)m * p.ldc + ncol[jp];
            vlo[st] = f32x4{0.f, 0.f, 0.f, 0.f};
            vhi[st] = vlo[st];
            if (want_res) { vlo[st] = rlo[g & 1][st]; vhi[st] = rhi[g & 1][st]; }
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int i = 2 * g + (st >> 1), jp = st & 1;
            u32x2 k0 = {0u, 0u}, k1 = {0u, 0u}, zl = {0u, 0u}, zh = {0u, 0u};
            if (want_aux_in) { zl = u32x2{zin[g & 1][st][0], zin[g & 1][st][1]}; zh = u32x2{zin[g & 1][st][2], zin[g & 1][st][3]}; }
            vlo[st] = epi_math<F16>(p, flags, 0, acc16[i][J0 + 2 * jp], vlo[st], zl, bias_lo[jp], alpha_n[jp], &k0);
            vhi[st] = epi_math<F16>(p, flags, 0, acc16[i][J0 + 2 * jp + 1], vhi[st], zh, bias_hi[jp], alpha_n[jp], &k1);
            zb[st] = u32x4{k0[0], k0[1], k1[0], k1[1]};
            if (want_aux_in && in[st]) {                       // rows / columns outside the matrix contribute nothing: a SELECT, not a multiply by
                cs_lo[jp] += vlo[st]; cs_hi[jp] += vhi[st];    // 0 (their aux words are row 0's: a NaN there times 0 would poison the whole column sum)
            }
        }
        // bf16 outputs: a lane's two 16-B pieces of row l15 (column halves jp = 0, 1) are HALF lines, 16 rows x 64 B per store
        // instruction -- streamed to memory that way they ran at half the rate of full lines (tools/micro/store_ctx: 118 vs 81 us
        // for this GEMM's 377 MB).  Lanes l15 and l15 ^ 8 swap one piece each (DPP row_ror:8), after which instruction A holds rows
        // 0-7 and instruction B rows 8-15 of the 16-row tile as full 128-B lines.
        const bool lo8 = l15 < 8;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * g + ii, s0 = 2 * ii, s1 = 2 * ii + 1;
            if (HS_DBG(p.dbg_) == 3 || HS_DBG(p.dbg_) == 7) { keep4(vlo[s0]); keep4(vhi[s0]); keep4(vlo[s1]); keep4(vhi[s1]); continue; }
            if (p.out_f32) {   // fp32 rows are full lines already (8 columns x 4 B x 4 lanes)
#pragma unroll
                for (int st = s0; st <= s1; ++st)
                    if (in[st]) {
                        if (p.nt) { __builtin_nontemporal_store(vlo[st], (f32x4*)(p.out_f32 + idx[st])); __builtin_nontemporal_store(vhi[st], (f32x4*)(p.out_f32 + idx[st] + 4)); }
                        else { *(f32x4*)(p.out_f32 + idx[st]) = vlo[st]; *(f32x4*)(p.out_f32 + idx[st] + 4) = vhi[st]; }
                    }
            }
            const bool want_aux = (flags & HS_EPI_GELU) && p.aux_out;
            if (p.out_bf16 || want_aux) {
                const int mA = mw + 16 * i + (l15 & 7), mB = mA + 8, nAB = nw + 32 * (l15 >> 3) + 8 * q;
                const bool inA = mA < p.M && nAB < p.N, inB = mB < p.M && nAB < p.N;
                const long idxA = cz + (long)mA * p.ldc + nAB, idxB = idxA + 8 * p.ldc;
                if (p.out_bf16) {
                    u32x4 o0, o1;
                    o0[0] = pack2h<F16>(vlo[s0][0], vlo[s0][1]); o0[1] = pack2h<F16>(vlo[s0][2], vlo[s0][3]);
                    o0[2] = pack2h<F16>(vhi[s0][0], vhi[s0][1]); o0[3] = pack2h<F16>(vhi[s0][2], vhi[s0][3]);
                    o1[0] = pack2h<F16>(vlo[s1][0], vlo[s1][1]); o1[1] = pack2h<F16>(vlo[s1][2], vlo[s1][3]);
                    o1[2] = pack2h<F16>(vhi[s1][0], vhi[s1][1]); o1[3] = pack2h<F16>(vhi[s1][2], vhi[s1][3]);
                    half_line_swap(o0, o1, lo8);
                    if (p.nt) {
                        if (inA) __builtin_nontemporal_store(o0, (u32x4*)((bf16_t*)p.out_bf16 + idxA));
                        if (inB) __builtin_nontemporal_store(o1, (u32x4*)((bf16_t*)p.out_bf16 + idxB));
                    } else {
                        if (inA) *(u32x4*)((bf16_t*)p.out_bf16 + idxA) = o0;
                        if (inB) *(u32x4*)((bf16_t*)p.out_bf16 + idxB) = o1;
                    }
                }
                if (want_aux) {
                    u32x4 z0 = zb[s0], z1 = zb[s1];
                    half_line_swap(z0, z1, lo8);
                    if (p.nt) {
                        if (inA) __builtin_nontemporal_store(z0, (u32x4*)((bf16_t*)p.aux_out + idxA));
                        if (inB) __builtin_nontemporal_store(z1, (u32x4*)((bf16_t*)p.aux_out + idxB));
                    } else {
                        if (inA) *(u32x4*)((bf16_t*)p.aux_out + idxA) = z0;
                        if (inB) *(u32x4*)((bf16_t*)p.aux_out + idxB) = z1;
                    }
                }
            }
        }
    }
    if (want_aux_in && p.colsum) {   // fused bias gradient: column sums of this wave's 128 x 64 piece, one atomic per column
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int sh = 1; sh < 16; sh <<= 1) {       // the 16 lanes of a row group hold the same columns, different rows
                    cs_lo[jp][e] += __shfl_xor(cs_lo[jp][e], sh, 64);
                    cs_hi[jp][e] += __shfl_xor(cs_hi[jp][e], sh, 64);
                }
            }
            if (l15 == 0 && n_in[jp]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { atomicAdd(p.colsum + ncol[jp] + e, cs_lo[jp][e]); atomicAdd(p.colsum + ncol[jp] + 4 + e, cs_hi[jp][e]); }
            }
        }
    }
}

// EF >= 0: the epilogue flags as a compile-time constant (the launcher checked p.flags == EF and the 8-column
// conditions): the flag tests fold away.  Run-time flags (EF = -1) cost ~1500 cycles of branches per epilogue step --
// 24k cycles per 256x256 tile, more than half of a K = 768 main loop.
// (the body is shared by the one-problem kernel and the grouped one: b = this block's index among the nb blocks of problem p)
template <bool ATR, bool BTR, bool BIG, int EF, bool M16 = false, bool F16 = false, bool HALF = false>
__device__ __forceinline__ void gemm_bf16_body(const GemmP& p, const int b_, const int nb_) {
    static_assert(!M16 || (BIG && !ATR && !BTR), "the 16x16x32 main loop exists for the 256x256 NT tile");
    static_assert(!HALF || (!BIG && !ATR && !BTR && !M16), "the 128x128 tile exists for the NT layout");
    typedef GCfg<BIG> C;
    __shared__ __attribute__((aligned(16))) char smem[(M16 || HALF) ? 163840 : C::SMEM];     // (16x16x32 kernels: 3 A + 2 B stages, no epilogue staging; HALF: 5 stages)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = BIG ? wave >> 2 : wave >> 1, wn = BIG ? wave & 3 : wave & 1;
    const int row0 = wm * (HALF ? 32 : C::MB * 32), col0 = wn * 64;        // this wave's tile inside the block tile
    constexpr int TM = HALF ? 128 : 256;                       // block tile height
    constexpr int WROWS = HALF ? 32 : 64;                      // rows of one staged epilogue piece
    // XCD-aware tile order.  A CU can pull only ~13 B/clk from beyond its XCD's L2 (a lone block needs ~3000 cycles
    // per K-tile), so speed hinges on the ~32 blocks an XCD runs at once SHARING operand tiles out of that L2.
    // Blocks are dealt round-robin over the 8 XCDs (b % 8 labels the blocks sharing an L2): give each XCD a
    // CONTIGUOUS range of a virtual order v (bijective form, cdna guide section 5); v walks split-K slices slowest
    // (blocks of one slice share A and B, different slices share nothing), and inside a slice groups of GM row tiles x
    // all column tiles, column-major inside a group (GM small: see hs_gemm).  Placement only affects speed, never results.
    int tm, tn, z;
    tile_order(b_, nb_, p.tiles_m, p.tiles_n, tm, tn, z, p.gm, p.np);
    const int m0 = tm * TM, n0 = tn * C::BN;
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
        if (kt0 >= kt1) return;
    }
    if (HS_DBG(p.dbg_) == 2) kt1 = kt0 + 1;
    unsigned long long t_prev = (HS_DBG(p.dbg_) == 4 || HS_DBG(p.dbg_) == 7) ? __builtin_amdgcn_s_memtime() : 0ull;
    f32x16 acc[C::MB][2];
#pragma unroll
    for (int i = 0; i < C::MB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 acc16[M16 ? 8 : 1][4];                                // 16x16x32 form of the same 128x64 wave tile
#pragma unroll
    for (int i = 0; i < (M16 ? 8 : 1); ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane DMA offsets at k0 = 0 (rows/cols outside the matrix -> out-of-range offset -> zeros)
    unsigned va[4], vb[C::PB];
#pragma unroll
    for (int j = 0; j < 4; ++j) va[j] = dma_voff<ATR, M16>(j, m0, p.M, 0, 64, p.lda, wave, lane);
#pragma unroll
    for (int j = 0; j < C::PB; ++j) vb[j] = dma_voff<BTR, M16, M16>(j, n0, p.N, 0, 64, p.ldb, wave, lane);
    const unsigned ka = ATR ? (unsigned)(p.lda * 128) : 128u;   // bytes per K-tile step
    const unsigned kb = BTR ? (unsigned)(p.ldb * 128) : 128u;
    const int kt_full = p.K >> 6;                                 // K-tiles [0, kt_full) are complete

    auto issue_tile = [&](int kt, int st) {
        char* sa = smem + st * C::STAGE;
#pragma unroll
        for (int q = 0; q < C::PIECES; ++q) {
            if (kt < kt_full) {
                if (q < 4) dma_piece(rsA, sa, q, wave, va[q], (unsigned)kt * ka);
                else dma_piece(rsB, sa + 32768, q - 4, wave, vb[q - 4], (unsigned)kt * kb);
            } else {
                if (q < 4) dma_piece(rsA, sa, q, wave, dma_voff<ATR>(q, m0, p.M, kt * 64, p.K, p.lda, wave, lane), 0);
                else dma_piece(rsB, sa + 32768, q - 4, wave, dma_voff<BTR>(q - 4, n0, p.N, kt * 64, p.K, p.ldb, wave, lane), 0);
            }
        }
    };
    STAMP(0);     // setup
    TRACE(0);
    if (HS_DBG(p.dbg_) == 22 && threadIdx.x == 0 && blockIdx.x < 4096)
        g_gemm_trace[blockIdx.x * 4 + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    if constexpr (M16) {
        big_mainloop16<F16>(acc16, smem, row0, col0, wave, lane, rsA, rsB, va, vb, ka, kb, kt0, kt1);
    } else if constexpr (BIG) {
        big_mainloop<BTR, F16>(acc, smem, row0, col0, wave, lane, rsA, rsB, va, vb, ka, kb, kt0, kt1, HS_DBG(p.dbg_));
    } else if constexpr (HALF) {
        half_mainloop<F16>(acc[0], smem, row0, col0, wave, lane, rsA, rsB, va, vb, kt0, kt1);      // (K % 64 == 0: hs_gemm)
    } else {
    // ring: prefetch distance D = NSTAGE-1 tiles.  At step kt: wait until tile kt has landed (D-1 newer tiles may stay
    // in flight: counted vmcnt), barrier, then issue tile kt+D into the stage every wave finished reading before
    // that barrier (the one consumed at step kt-1).
    constexpr int D = C::NSTAGE - 1;
    issue_tile(kt0, 0);
    if (D == 2 && kt0 + 1 < kt1) issue_tile(kt0 + 1, 1);
    int st = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        if (D == 2 && kt + 1 < kt1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int stn = (D == 2) ? (st >= 1 ? st - 1 : 2) : (st ^ 1);
        const char* a_l = smem + st * C::STAGE;
        const char* b_l = a_l + 32768;
        char* nxt = smem + stn * C::STAGE;
        const int ktn = kt + D;
        const bool issue = ktn < kt1;
        unsigned sa = (unsigned)ktn * ka, sb = (unsigned)ktn * kb;
        if (issue && ktn >= kt_full) {
            // partial last K-tile: bound k per lane.  It is the last tile ever issued, so the hoisted offsets are dead
            // and can be overwritten in place.
#pragma unroll
            for (int j = 0; j < 4; ++j) va[j] = dma_voff<ATR>(j, m0, p.M, ktn * 64, p.K, p.lda, wave, lane);
#pragma unroll
            for (int j = 0; j < C::PB; ++j) vb[j] = dma_voff<BTR>(j, n0, p.N, ktn * 64, p.K, p.ldb, wave, lane);
            sa = sb = 0;
        }
        ktile_step<ATR, BTR, BIG, F16>(acc, a_l, b_l, nxt, row0, col0, wave, lane, rsA, rsB, va, vb, sa, sb, issue);
        st = (D == 2) ? (st == 2 ? 0 : st + 1) : (st ^ 1);
    }
    }
    // ---- epilogue through LDS: each wave parks a 64x64 fp32 piece of its tile in its own 17 KiB slice of the (now
    // idle) ring and reads it back row-wise, so global stores/atomics and the bias/resid/aux loads are 128-256-B row
    // segments instead of one 4-8-B piece per row (which cost more than the K=768 main loop).
    STAMP(1);     // main loop
    TRACE(1);
    __builtin_amdgcn_s_barrier();     // every wave is done reading the operand stages
    STAMP(2);     // post-loop barrier
    if (HS_DBG(p.dbg_) == 1) {
#pragma unroll
        for (int i = 0; i < C::MB; ++i) { keep_alive(acc[i][0]); keep_alive(acc[i][1]); }
#pragma unroll
        for (int i = 0; i < (M16 ? 8 : 1); ++i) { keep4(acc16[i][0]); keep4(acc16[i][1]); keep4(acc16[i][2]); keep4(acc16[i][3]); }
        return;
    }
    if constexpr (M16) {
        epilogue16<EF, 4, 0, F16>(p, acc16, m0 + row0, n0 + col0, cz, lane);
        STAMP(4);
        TRACE(2);
        return;
    }
    float* my = (float*)(smem + wave * (64 * G_EPI_LD * 4));
    const int h = lane >> 5, c = lane & 31;
    const int flags = EF >= 0 ? EF : p.flags;
    const bool vec_ok = !(p.N & 3) && !(p.ldc & 3);
#pragma unroll
    for (int half = 0; half < C::MB / 2; ++half) {
#pragma unroll
        for (int mb = 0; mb < (HALF ? 1 : 2); ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) my[(mb * 32 + acc_row(r, h)) * G_EPI_LD + nb * 32 + c] = acc[half * 2 + mb][nb][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        STAMP(3);     // staging writes
        const int mw = m0 + row0 + half * 64, nw = n0 + col0;
        if (flags & HS_EPI_ATOMIC) {
            // one float per lane: every atomic wave-instruction is 256 contiguous bytes of one row
            const int n = nw + lane;
            for (int row = 0; row < WROWS; ++row) {
                const int m = mw + row;
                if (m < p.M && n < p.N) atomicAdd(p.out_f32 + cz + (long)m * p.ldc + n, my[row * G_EPI_LD + lane] * p.alpha);
            }
        } else {
            const bool want_res = (flags & HS_EPI_RESID) != 0, want_aux = (flags & HS_EPI_GELU_BWD) != 0;
            if (EF >= 0 || (vec_ok && !(p.N & 7) && !(p.ldc & 7) && !(p.scale_cols & 7))) {
                // 8 consecutive columns per lane, 8 rows per wave-instruction: bf16 outputs leave as 16-B stores (the
                // store tail is issue-bound: half as many store instructions at equal bytes), fp32 ones as 2 x 16 B.
                const int r8 = lane >> 3, c8 = (lane & 7) * 8, n = nw + c8;
                f32x4 bias_lo = {0.f, 0.f, 0.f, 0.f}, bias_hi = {0.f, 0.f, 0.f, 0.f}, cs_lo = bias_lo, cs_hi = bias_lo;
                const float alpha_n = n < p.scale_cols ? p.alpha * p.scale_alpha : p.alpha;
                const bool n_in = n < p.N;
                if ((flags & HS_EPI_BIAS) && !(flags & HS_EPI_BIAS_M) && n_in) {
                    bias_lo = *(const f32x4*)(p.bias + n);
                    bias_hi = *(const f32x4*)(p.bias + n + 4);
                }
                // (an epilogue that LOADS -- residual, saved gelu' -- keeps 4 steps of loads in flight even on the big tile: with 2, a
                // tile pays the HBM round trip 16 times)
                constexpr int GS = (BIG && !(EF >= 0 && (EF & (HS_EPI_GELU_BWD | HS_EPI_RESID)))) ? 2 : 4;
                // groups of GS steps: all loads in flight -> all math -> all stores (see epi_math).  The 256x256 tile still holds
                // 64-128 accumulator registers here: smaller groups keep the epilogue out of scratch (a reload is a memory round trip)
#pragma unroll 1
                for (int g = 0; g < WROWS / 8 / GS; ++g) {
                    f32x4 vlo[GS], vhi[GS];
                    u32x4 zb[GS];
                    bool in[GS];
                    long idx[GS];
#pragma clang loop unroll(full)
                    for (int i = 0; i < GS; ++i) {
                        const int m = mw + (g * GS + i) * 8 + r8;
                        in[i] = m < p.M && n_in;
                        idx[i] = cz + (long)m * p.ldc + n;
                        vlo[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        vhi[i] = vlo[i];
                        zb[i] = u32x4{0u, 0u, 0u, 0u};
                        if (want_res && in[i]) { vlo[i] = *(const f32x4*)(p.resid + idx[i]); vhi[i] = *(const f32x4*)(p.resid + idx[i] + 4); }
                        if (want_aux && in[i]) zb[i] = *(const u32x4*)((const bf16_t*)p.aux + idx[i]);
                    }
#pragma clang loop unroll(full)
                    for (int i = 0; i < GS; ++i) {
                        const int row = (g * GS + i) * 8 + r8;
                        const f32x4 a = *(const f32x4*)(my + row * G_EPI_LD + c8), b = *(const f32x4*)(my + row * G_EPI_LD + c8 + 4);
                        u32x2 k0 = {0u, 0u}, k1 = {0u, 0u};
                        vlo[i] = epi_math<F16>(p, flags, mw + row, a, vlo[i], u32x2{zb[i][0], zb[i][1]}, bias_lo, alpha_n, &k0);
                        vhi[i] = epi_math<F16>(p, flags, mw + row, b, vhi[i], u32x2{zb[i][2], zb[i][3]}, bias_hi, alpha_n, &k1);
                        zb[i] = u32x4{k0[0], k0[1], k1[0], k1[1]};
                        if (in[i]) { cs_lo += vlo[i]; cs_hi += vhi[i]; }
                    }
#pragma clang loop unroll(full)
                    for (int i = 0; i < GS; ++i)
                        if (in[i] && (HS_DBG(p.dbg_) == 3 || HS_DBG(p.dbg_) == 7)) { keep4(vlo[i]); keep4(vhi[i]); }
                        else if (in[i]) {
                            u32x4 o;
                            o[0] = pack2h<F16>(vlo[i][0], vlo[i][1]); o[1] = pack2h<F16>(vlo[i][2], vlo[i][3]);
                            o[2] = pack2h<F16>(vhi[i][0], vhi[i][1]); o[3] = pack2h<F16>(vhi[i][2], vhi[i][3]);
                            if (p.nt) {
                                if ((flags & HS_EPI_GELU) && p.aux_out) __builtin_nontemporal_store(zb[i], (u32x4*)((bf16_t*)p.aux_out + idx[i]));
                                if (p.out_f32) { __builtin_nontemporal_store(vlo[i], (f32x4*)(p.out_f32 + idx[i])); __builtin_nontemporal_store(vhi[i], (f32x4*)(p.out_f32 + idx[i] + 4)); }
                                if (p.out_bf16) __builtin_nontemporal_store(o, (u32x4*)((bf16_t*)p.out_bf16 + idx[i]));
                            } else {
                                if ((flags & HS_EPI_GELU) && p.aux_out) *(u32x4*)((bf16_t*)p.aux_out + idx[i]) = zb[i];
                                if (p.out_f32) { *(f32x4*)(p.out_f32 + idx[i]) = vlo[i]; *(f32x4*)(p.out_f32 + idx[i] + 4) = vhi[i]; }
                                if (p.out_bf16) *(u32x4*)((bf16_t*)p.out_bf16 + idx[i]) = o;
                            }
                        }
                }
                if (p.colsum) {   // fused bias gradient: column sums of this wave's 64x64 piece, one atomic per column
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cs_lo[e] += __shfl_xor(cs_lo[e], 8, 64);  cs_hi[e] += __shfl_xor(cs_hi[e], 8, 64);
                        cs_lo[e] += __shfl_xor(cs_lo[e], 16, 64); cs_hi[e] += __shfl_xor(cs_hi[e], 16, 64);
                        cs_lo[e] += __shfl_xor(cs_lo[e], 32, 64); cs_hi[e] += __shfl_xor(cs_hi[e], 32, 64);
                    }
                    if (lane < 8 && n_in) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { atomicAdd(p.colsum + n + e, cs_lo[e]); atomicAdd(p.colsum + n + 4 + e, cs_hi[e]); }
                    }
                }
                if (half + 1 < C::MB / 2) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                STAMP(4);     // read back + math + stores issued
                continue;
            }
            if constexpr (EF < 0) {
            const int rr = lane >> 4, cc = (lane & 15) * 4;
            f32x4 csum = {0.f, 0.f, 0.f, 0.f};
            if (vec_ok) {
                f32x4 bias_n = {0.f, 0.f, 0.f, 0.f};
                const float alpha_n = nw + cc < p.scale_cols ? p.alpha * p.scale_alpha : p.alpha;   // scale_cols % 4 == 0
                if ((flags & HS_EPI_BIAS) && !(flags & HS_EPI_BIAS_M) && nw + cc + 3 < p.N) bias_n = *(const f32x4*)(p.bias + nw + cc);
                // groups of 8 steps: 8+8 loads in flight -> all math -> all stores (see epi_math)
#pragma unroll 1
                for (int g = 0; g < WROWS / 32; ++g) {
                    f32x4 val[8];
                    u32x2 zbv[8];
                    bool in[8];
                    long idx[8];
#pragma clang loop unroll(full)
                    for (int i = 0; i < 8; ++i) {
                        const int m = mw + (g * 8 + i) * 4 + rr, n = nw + cc;
                        in[i] = m < p.M && n + 3 < p.N;
                        idx[i] = cz + (long)m * p.ldc + n;
                        val[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        zbv[i] = u32x2{0u, 0u};
                        if (want_res && in[i]) val[i] = *(const f32x4*)(p.resid + idx[i]);
                        if (want_aux && in[i]) zbv[i] = *(const u32x2*)((const bf16_t*)p.aux + idx[i]);
                    }
#pragma clang loop unroll(full)
                    for (int i = 0; i < 8; ++i) {
                        const int row = (g * 8 + i) * 4 + rr;
                        const f32x4 v = *(const f32x4*)(my + row * G_EPI_LD + cc);
                        u32x2 keep = {0u, 0u};
                        val[i] = epi_math<F16>(p, flags, mw + row, v, val[i], zbv[i], bias_n, alpha_n, &keep);
                        zbv[i] = keep;
                        if (in[i]) csum += val[i];
                    }
#pragma clang loop unroll(full)
                    for (int i = 0; i < 8; ++i)
                        if (in[i]) epi_store<F16>(p, idx[i], val[i], zbv[i]);
                }
            } else {   // rare path (N or ldc not a multiple of 4): element-wise, rolled
#pragma unroll 1
                for (int step = 0; step < WROWS / 4; ++step) {
                    const int row = step * 4 + rr;
                    const f32x4 v = *(const f32x4*)(my + row * G_EPI_LD + cc);
#pragma unroll 1
                    for (int e = 0; e < 4; ++e) epilogue_store<false, F16>(p, cz, mw + row, nw + cc + e, v[e]);
                }
            }
            if (p.colsum) {   // fused bias gradient: column sums of this wave's 64x64 piece, one atomic per column
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    csum[e] += __shfl_xor(csum[e], 16, 64);
                    csum[e] += __shfl_xor(csum[e], 32, 64);
                }
                if (lane < 16 && nw + cc < p.N) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(p.colsum + nw + cc + e, csum[e]);
                }
            }
            }   // EF < 0
        }
        if (half + 1 < C::MB / 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    TRACE(2);
}

template <bool ATR, bool BTR, bool BIG, int EF, bool M16 = false, bool F16 = false, bool HALF = false>
__global__ __launch_bounds__(512) void gemm_bf16_kernel(GemmP p) {
    gemm_bf16_body<ATR, BTR, BIG, EF, M16, F16, HALF>(p, blockIdx.x, gridDim.x);
}

// Persistent form of the 16x16x32 NT kernel (round 4; the default whenever a launch has more tiles than CUs): ONE block per CU walks the
// tiles b, b + gridDim, ... (gridDim a multiple of 8, so b % 8 -- the XCD -- and the tile order are what the one-tile-per-block launch
// gives).  A block that owns all 160 KiB of LDS cannot overlap its successor's launch (3.6k cycles from a block's last instruction to the
// next block's first, of a 46k-cycle K = 768 tile), and the first K-tiles of every output tile are compulsory L2 misses: here the next
// tile's first two K-tiles are issued IN FRONT OF the current tile's epilogue (all five LDS stages are free after the main loop's last
// barrier, the 16x16x32 epilogue goes from the accumulators straight to memory).  No extra synchronisation between tiles: a wave issues
// that LDS-DMA only after the last barrier of the main loop, which every wave reaches after its last LDS read; the counted vmcnt waits
// stay safe with the epilogue's stores in flight (loads retire in issue order, younger stores can only make a wait longer).  Measured
// same box at 53 volumes (tools/r4_gemm2.sh): qkv 863 -> 826 us, fc1 1113 -> 1067, proj 311 -> 300, fc2 947 -> 925, fc1 + GELU 1455 ->
// 1418-1434, x gelu' 1280 -> 1266-1274; in the engine plain 600 -> 586 us, x gelu' 1388 -> 1352, + residual 671 -> 659, + GELU 1218 -> 1205:
// +0.34 % on the step.  What it cannot hide is the drain of the epilogue's stores: vmcnt is ONE counter for loads and stores, so the next
// tile's first barrier is released only when (nearly) all of them have retired.
// the first two K-tiles of an output tile, in big_mainloop16's order A(0), B(0), A(1), B(1): issued by the persistent kernel in front of the
// PREVIOUS tile's epilogue, so that the tile's first (compulsory) L2 misses elapse under that epilogue
__device__ __forceinline__ void prologue16(char* smem, int wave, __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB,
                                           const unsigned (&va)[4], const unsigned (&vb)[4], int ktiles) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t < ktiles) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dma_piece(rsA, smem + t * 32768, q, wave, va[q], (unsigned)t * 128u);
#pragma unroll
            for (int q = 0; q < 4; ++q) dma_piece(rsB, smem + 98304 + t * 32768, q, wave, vb[q], (unsigned)t * 128u);
        }
    }
}
template <int EF, bool F16>
__global__ __launch_bounds__(512) void gemm_nt16_persist_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) char smem[163840];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = (wave >> 2) * 128, col0 = (wave & 3) * 64;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, (size_t)p.M * p.lda * 2);        // (batch == 1, split_k == 1: checked by hs_gemm)
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B, (size_t)p.N * p.ldb * 2);
    const int kt1 = p.K >> 6;                                                             // (K % 64 == 0: the 256x256 tile's condition)
    int b = blockIdx.x, tm, tn, z;
    tile_order(b, p.nblocks, p.tiles_m, p.tiles_n, tm, tn, z, p.gm, p.np);
    int m0 = tm * 256, n0 = tn * 256;
    unsigned va[4], vb[4];
    auto offsets = [&](int m0_, int n0_) {
#pragma unroll
        for (int j = 0; j < 4; ++j) va[j] = dma_voff<false, true>(j, m0_, p.M, 0, 64, p.lda, wave, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) vb[j] = dma_voff<false, true, true>(j, n0_, p.N, 0, 64, p.ldb, wave, lane);
    };
    offsets(m0, n0);
    prologue16(smem, wave, rsA, rsB, va, vb, kt1);
    for (;;) {
        f32x4 acc16[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        big_mainloop16<F16, false>(acc16, smem, row0, col0, wave, lane, rsA, rsB, va, vb, 128u, 128u, 0, kt1);
        // every LDS read of this tile retired before the main loop's last barrier: all five stages are free
        const int bn = b + (int)gridDim.x;
        const bool more = bn < p.nblocks;
        int m0n = 0, n0n = 0;
        if (more) {
            tile_order(bn, p.nblocks, p.tiles_m, p.tiles_n, tm, tn, z, p.gm, p.np);
            m0n = tm * 256; n0n = tn * 256;
            offsets(m0n, n0n);
            prologue16(smem, wave, rsA, rsB, va, vb, kt1);
        }
        epilogue16<EF, 4, 0, F16>(p, acc16, m0 + row0, n0 + col0, 0l, lane);
        if (!more) break;
        b = bn; m0 = m0n; n0 = n0n;
        offsets(m0, n0);            // recomputed (a few VALU operations) rather than kept live across the epilogue: that spilled
    }
}

// Grouped launch: up to HS_GEMM_GROUP_MAX problems of one kernel variant in ONE grid (hs_gemm_group: the four weight-gradient
// GEMMs of a transformer layer).  Problem i owns blocks [first[i], first[i+1]); every range starts at a multiple of 8 so that
// (local block index) % 8 is still the XCD the block runs on, which is what tile_order builds on.
struct GemmGroup {
    GemmP p[HS_GEMM_GROUP_MAX];
    int first[HS_GEMM_GROUP_MAX + 1];
    int count;
};
template <bool ATR, bool BTR, bool BIG, int EF, bool F16 = false>
__global__ __launch_bounds__(512) void gemm_bf16_group_kernel(GemmGroup g) {
    int i = 0;
    while (i + 1 < g.count && (int)blockIdx.x >= g.first[i + 1]) ++i;
    const int b = (int)blockIdx.x - g.first[i];
    if (b >= g.p[i].nblocks) return;                                    // padding up to the next multiple of 8
    gemm_bf16_body<ATR, BTR, BIG, EF, false, F16>(g.p[i], b, g.p[i].nblocks);
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
    // grid.z = batch x split_k (exact-f32 kernel only: the scorer's dK / dV = P^T dOut are 32 x 768 outputs over 4096 rows PER VOLUME --
    // 12 tiles of 128 K-steps each at one volume per GPU, ~100 us of latency; split over K they fill the chip)
    const int z = blockIdx.z;
    const int sk = p.split_k > 1 ? p.split_k : 1;
    const int bz = z / sk, ks = z - bz * sk;
    const float* A = (const float*)p.A + bz * p.sA;
    const float* B = (const float*)p.B + bz * p.sB;
    const long cz = bz * p.sC;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (size_t)(ATR ? p.K : p.M) * p.lda * 4);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (size_t)(BTR ? p.K : p.N) * p.ldb * 4);
    const int kt_total = (p.K + 31) >> 5;
    int kt0 = 0, kt1 = kt_total;
    if (p.split_k > 1) {
        const int per = (kt_total + p.split_k - 1) / p.split_k;
        kt0 = ks * per;
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

// A/B switches of tools/kbench and tools/bench_gemm.py, read ONCE (none is set in production; hs_gemm runs ~900 times per step)
namespace {
struct GemmEnv {
    bool no_nt, no_m16;
    char tile;      // 0, 'b' (big), 's' (small: never the 128x128 form) or 'h' (128x128 wherever it exists)
    int gm;         // 0, or the tile order's group height
    int np;         // 0, or the tile order's panel width (column tiles)
    int persist;    // 16x16x32 NT kernels as persistent blocks (0 / 1), HS_GEMM_PERSIST
    GemmEnv() : no_nt(getenv("HS_GEMM_NO_NT") != nullptr), no_m16(getenv("HS_GEMM_NO_M16") != nullptr), tile(0) {
        const char* t = getenv("HS_GEMM_TILE");
        if (t && (t[0] == 'b' || t[0] == 's' || t[0] == 'h')) tile = t[0];
        gm = getenv("HS_GEMM_GM") ? atoi(getenv("HS_GEMM_GM")) : 0;
        np = getenv("HS_GEMM_NP") ? atoi(getenv("HS_GEMM_NP")) : 0;
        persist = getenv("HS_GEMM_PERSIST") ? atoi(getenv("HS_GEMM_PERSIST")) : 1;      // default on (round 4); 0 = one tile per block, for A/B runs
    }
};
const GemmEnv kEnv;
}  // namespace

// kernel selection of the 16-bit GEMMs, once per operand format (H = false: bf16, the training policy; H = true: IEEE fp16 -- since round 5
// every layout and epilogue, i.e. forward AND backward: same kernels, fp16 packs and the fp16 MFMA opcode)
struct Launch16 { int layout, f; bool big, col8, m16_ok, persist; int pgrid; bool half; };
template <bool H>
static void launch16(const GemmP& p, const Launch16& c, dim3 grid, hipStream_t s) {
    const int f = c.f;
#define HS_LAUNCH_E(ATR, BTR, EF)                                                                                    \
    do {                                                                                                             \
        if (c.big) hipLaunchKernelGGL((gemm_bf16_kernel<ATR, BTR, true, EF, false, H>), grid, dim3(512), 0, s, p);    \
        else hipLaunchKernelGGL((gemm_bf16_kernel<ATR, BTR, false, EF, false, H>), grid, dim3(512), 0, s, p);         \
    } while (0)
#define HS_LAUNCH_M16(EF)                                                                                            \
    do {                                                                                                             \
        if (c.persist) hipLaunchKernelGGL((gemm_nt16_persist_kernel<EF, H>), dim3(c.pgrid), dim3(512), 0, s, p);      \
        else hipLaunchKernelGGL((gemm_bf16_kernel<false, false, true, EF, true, H>), grid, dim3(512), 0, s, p);       \
    } while (0)
#define HS_LAUNCH_H(EF) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, false, EF, false, H, true>), grid, dim3(512), 0, s, p)
    if (c.half) {                                                  // (NT, 128x128 tiles: hs_gemm)
        if (c.col8 && f == 0) HS_LAUNCH_H(0);
        else if (c.col8 && f == (HS_EPI_BIAS | HS_EPI_RESID)) HS_LAUNCH_H(HS_EPI_BIAS | HS_EPI_RESID);
        else HS_LAUNCH_H(-1);
    } else if (c.m16_ok) {
        if (f == 0) HS_LAUNCH_M16(0);
        else if (f == HS_EPI_BIAS) HS_LAUNCH_M16(HS_EPI_BIAS);        // patch embedding, scorer / packer projections (fp32 out)
        else if (f == (HS_EPI_BIAS | HS_EPI_RESID)) HS_LAUNCH_M16(HS_EPI_BIAS | HS_EPI_RESID);
        else if (f == (HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU)) HS_LAUNCH_M16(HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU);
        else HS_LAUNCH_M16(HS_EPI_BIAS | HS_EPI_GELU | HS_EPI_SAVE_DGELU);
    } else if (c.layout == HS_GEMM_NT) {
        if (c.col8 && f == 0) HS_LAUNCH_E(false, false, 0);
        else if (c.col8 && f == (HS_EPI_BIAS | HS_EPI_RESID)) HS_LAUNCH_E(false, false, HS_EPI_BIAS | HS_EPI_RESID);
        else if (c.col8 && f == (HS_EPI_BIAS | HS_EPI_GELU | HS_EPI_SAVE_DGELU)) HS_LAUNCH_E(false, false, HS_EPI_BIAS | HS_EPI_GELU | HS_EPI_SAVE_DGELU);
        else if (c.col8 && f == (HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU)) HS_LAUNCH_E(false, false, HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU);   // fc2's dgrad against W2^T at shapes too small for the 256x256 tile
        else HS_LAUNCH_E(false, false, -1);
    } else if (c.layout == HS_GEMM_NN) {
        if (c.col8 && f == 0) HS_LAUNCH_E(false, true, 0);
        else if (c.col8 && f == (HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU)) HS_LAUNCH_E(false, true, HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU);
        else HS_LAUNCH_E(false, true, -1);
    } else {
        if (f == HS_EPI_ATOMIC) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, false, HS_EPI_ATOMIC, false, H>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<true, true, false, -1, false, H>), grid, dim3(512), 0, s, p);
    }
#undef HS_LAUNCH_E
#undef HS_LAUNCH_M16
#undef HS_LAUNCH_H
}

extern "C" int hs_gemm_group(const hs_gemm_args* args, int count, void* stream) { return hs_gemm_group_prec(args, count, HS_PREC_BF16, stream); }
extern "C" int hs_gemm_group_prec(const hs_gemm_args* args, int count, int prec, void* stream) {
    if (!args || count < 1 || count > HS_GEMM_GROUP_MAX) return HS_ERR_ARG;
    if (prec != HS_PREC_BF16 && prec != HS_PREC_F16) return HS_ERR_ARG;
    GemmGroup g = {};
    double flops = 0, bytes = 0;
    int nblocks = 0;
    for (int i = 0; i < count; ++i) {
        const hs_gemm_args* a = args + i;
        if (!a->A || !a->B || !a->out_f32 || a->out_bf16) return HS_ERR_ARG;
        if (a->layout != HS_GEMM_TN || a->flags != HS_EPI_ATOMIC || a->batch != 1 || a->split_k < 1) return HS_ERR_ARG;
        if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->scale_cols || a->colsum_out) return HS_ERR_SHAPE;
        if ((a->M & 7) || (a->N & 7) || (a->lda & 7) || (a->ldb & 7)) return HS_ERR_SHAPE;
        if (((uintptr_t)a->A | (uintptr_t)a->B) & 15) return HS_ERR_SHAPE;
        if ((size_t)a->K * a->lda * 2 >= HS_MAX_RECORDS || (size_t)a->K * a->ldb * 2 >= HS_MAX_RECORDS) return HS_ERR_SHAPE;
        GemmP& p = g.p[i];
        p.A = a->A; p.B = a->B; p.M = a->M; p.N = a->N; p.K = a->K;
        p.lda = a->lda; p.ldb = a->ldb; p.sA = 0; p.sB = 0; p.sC = 0; p.ldc = a->ldc;
        p.out_f32 = a->out_f32; p.out_bf16 = nullptr; p.bias = nullptr; p.resid = nullptr; p.aux = nullptr; p.aux_out = nullptr;
        p.colsum = nullptr; p.alpha = a->alpha; p.flags = a->flags; p.split_k = a->split_k;
        p.scale_cols = 0; p.scale_alpha = 1.0f; p.nt = 0; p.dbg_ = 0;
        p.tiles_m = (a->M + 255) / 256;
        p.tiles_n = (a->N + 127) / 128;
        p.gm = kEnv.gm > 0 ? kEnv.gm : (p.tiles_n >= 8 ? 4 : 2);
        p.np = 0;
        p.nblocks = p.tiles_m * p.tiles_n * a->split_k;
        g.first[i] = nblocks;
        nblocks += (p.nblocks + 7) & ~7;
        flops += 2.0 * a->M * a->N * a->K;
        bytes += ((double)a->M + a->N) * a->K * 2 + (double)a->M * a->N * 4;
    }
    g.first[count] = nblocks;
    g.count = count;
    ProfScope prof(HS_PROF_GEMM_TN_GROUP, stream, flops, bytes);
    if (prec == HS_PREC_F16) hipLaunchKernelGGL((gemm_bf16_group_kernel<true, true, false, HS_EPI_ATOMIC, true>), dim3(nblocks), dim3(512), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((gemm_bf16_group_kernel<true, true, false, HS_EPI_ATOMIC>), dim3(nblocks), dim3(512), 0, (hipStream_t)stream, g);
    HS_CHECK_LAUNCH();
    return HS_OK;
}

extern "C" int hs_gemm(const hs_gemm_args* a, int prec, void* stream) {
    if (!a || !a->A || !a->B || (!a->out_f32 && !a->out_bf16)) return HS_ERR_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch < 1 || a->split_k < 1) return HS_ERR_SHAPE;
    if (a->layout < HS_GEMM_NT || a->layout > HS_GEMM_TN) return HS_ERR_ARG;
    if ((a->flags & HS_EPI_BIAS) && !a->bias) return HS_ERR_ARG;
    if ((a->flags & HS_EPI_RESID) && !a->resid) return HS_ERR_ARG;
    if ((a->flags & HS_EPI_GELU_BWD) && !a->aux) return HS_ERR_ARG;
    // (split-K: atomic fp32 output; one batch entry -- except in the exact-f32 kernel, whose grid.z carries batch x split)
    if (a->split_k > 1 && (!(a->flags & HS_EPI_ATOMIC) || (a->batch != 1 && prec != HS_PREC_F32) || !a->out_f32 || a->out_bf16)) return HS_ERR_SHAPE;
    if (prec != HS_PREC_BF16 && prec != HS_PREC_F32 && prec != HS_PREC_F16) return HS_ERR_ARG;
    const bool h16 = prec == HS_PREC_F16;          // fp16 operands: every layout and epilogue the bf16 path has (round 5)
    if (a->scale_cols < 0 || (a->scale_cols & 3) || (a->scale_cols && (a->flags & HS_EPI_ATOMIC))) return HS_ERR_SHAPE;
    if (a->colsum_out && (prec == HS_PREC_F32 || (a->N & 3) || (a->ldc & 3) || a->batch != 1 || (a->flags & HS_EPI_ATOMIC))) return HS_ERR_SHAPE;
    const int esz = prec == HS_PREC_F32 ? 4 : 2;
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
    p.aux_out = a->aux_out; p.colsum = a->colsum_out; p.alpha = a->alpha; p.flags = a->flags; p.split_k = a->split_k;
    p.scale_cols = a->scale_cols; p.scale_alpha = a->scale_cols_alpha;
    p.dbg_ = HS_DBG(g_hs_dbg[3]);
    p.gm = 0;
    p.np = 0;
    // Output stores of a GEMM whose result cannot stay in the caches anyway are issued as streaming (nontemporal) stores:
    // ordinary stores park 128 KiB of dirty lines per tile in the XCD's 4-MiB L2 -- 32 tiles per round = the whole L2 --
    // evicting the operand panels the other blocks are about to hit (measured on the 81940-row ViT-B shapes: main-loop vmcnt
    // wait 7.4k -> 3.8k cycles per tile, qkv/fc1 -6 % end to end; streaming LOADS of the A panel made them 3-8 % slower).
    {
        const double out_bytes = (double)a->M * a->N * a->batch * ((a->out_f32 ? 4 : 0) + (a->out_bf16 ? 2 : 0) + (a->aux_out ? 2 : 0));
        p.nt = out_bytes >= 64.0 * 1048576.0 && !(a->flags & HS_EPI_ATOMIC) && !kEnv.no_nt;
    }
    hipStream_t s = (hipStream_t)stream;
    const int zdim = a->split_k > 1 ? a->split_k : a->batch;
    const double mnk = (double)a->M * a->N * a->K * a->batch;
    const double prof_bytes = ((double)a->M * a->K + (double)a->N * a->K) * a->batch * esz + (double)a->M * a->N * a->batch * (a->out_f32 ? 4 : 2);
    if (prec != HS_PREC_F32) {
        // tile choice: the 256x256 tile does twice the MFMA work per barrier / DMA piece / LDS read.  Measured on the layer's shapes
        // at 1-4 volumes (4097-16388 rows): it wins by 4-28 % as soon as the 256x128 tile would need a second round of the chip's CUs
        // (more than ncu/2 big tiles), and loses by 30-37 % below that (one round of half-size tiles)
        p.tiles_m = (a->M + 255) / 256;
        const long big_blocks = (long)p.tiles_m * ((a->N + 255) / 256) * zdim;
        // CU count of the CURRENT device, cached per device ordinal (a process may drive several)
        static int ncu_of[64] = {0};
        int dev_ = 0;
        (void)hipGetDevice(&dev_);
        dev_ &= 63;
        if (!ncu_of[dev_]) { int n = 0; (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev_); ncu_of[dev_] = n > 0 ? n : 256; }
        const int ncu = ncu_of[dev_];
        // (kEnv.tile: 'b' / 's' force the NT / NN tile for A/B runs; the TN kernel only exists with the 256x128 tile)
        const bool big = !(a->K & 63) && a->layout != HS_GEMM_TN && (kEnv.tile ? kEnv.tile == 'b' : 2 * big_blocks > ncu);
        p.tiles_n = (a->N + (big ? 255 : 127)) / (big ? 256 : 128);
        // 128x128 tiles when the 256x128 ones would fill at most half of the CUs (the N = 768 outputs at 4097 rows: 102 -> 198 blocks):
        // one round of the chip either way, half the main loop per block
        const bool half = !big && a->layout == HS_GEMM_NT && !(a->K & 63) && (kEnv.tile ? kEnv.tile == 'h' : 2l * p.tiles_m * p.tiles_n * zdim <= ncu);
        if (half) p.tiles_m = (a->M + 127) / 128;
        // group height of the tile order: the column tiles that share an A row panel must run AT THE SAME TIME (they then pull
        // each A K-tile into the XCD's L2 once); with few column tiles a tall group puts the sharers 16-32 dispatch slots apart and
        // the N = 768 / K = 3072 shapes re-read their 1.5-MiB A panels (measured, tools/gm_sweep.sh at M = 81940: fc2 417 -> 380 us,
        // dgrad fc1 446 -> 420, dgrad qkv 327 -> 315, fc2+resid 491 -> 460 with 2 instead of 16; 9-12 column tiles: 2..8 within noise)
        p.gm = kEnv.gm > 0 ? kEnv.gm : (p.tiles_n >= 8 ? 4 : 2);
        p.np = kEnv.np;
        dim3 grid(p.tiles_m * p.tiles_n * zdim, 1, 1);
        // epilogue specialisations (compile-time flags); anything else runs the generic kernel (EF = -1)
        const bool col8 = !(a->N & 7) && !(a->ldc & 7) && !(a->scale_cols & 7);
        const int f = a->flags;
        const bool m16 = big && a->layout == HS_GEMM_NT && (!a->colsum_out || a->flags == (HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU)) && !kEnv.no_m16;
        p.nblocks = (int)grid.x;
        // (the persistent kernel carries no debug stamps: a hook build's HS_DBG modes select the one-tile kernel they instrument)
        const bool persist = kEnv.persist && zdim == 1 && (int)grid.x > ncu && !HS_DBG(p.dbg_);
        const int pgrid = ncu & ~7;
        const bool m16_ok = m16 && col8 && (f == 0 || f == HS_EPI_BIAS || f == (HS_EPI_BIAS | HS_EPI_RESID) || f == (HS_EPI_BIAS | HS_EPI_GELU | HS_EPI_SAVE_DGELU) ||
                                            (f == (HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU) && !(((uintptr_t)a->aux) & 15)));
        // profiling class = the kernel function: the four 16x16x32 instantiations are the encoder layer's hot GEMMs
        const int pclass = !m16_ok ? HS_PROF_GEMM_NT + a->layout
                         : f == 0 ? HS_PROF_GEMM_NT16_PLAIN
                         : f == HS_EPI_BIAS ? HS_PROF_GEMM_NT16_BIAS
                         : f == (HS_EPI_BIAS | HS_EPI_RESID) ? HS_PROF_GEMM_NT16_RESID
                         : f == (HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU) ? HS_PROF_GEMM_NT16_DGELU : HS_PROF_GEMM_NT16_GELU;
        ProfScope prof(pclass, stream, 2.0 * mnk, prof_bytes);
        const Launch16 ln = {a->layout, f, big, col8, m16_ok, persist, pgrid, half};
        if (h16) launch16<true>(p, ln, grid, s);
        else launch16<false>(p, ln, grid, s);
    } else {
        ProfScope prof(HS_PROF_GEMM_NT + a->layout, stream, 2.0 * mnk, prof_bytes);
        p.tiles_m = (a->M + 63) / 64;
        dim3 grid(p.tiles_m * ((a->N + 63) / 64), 1, a->split_k > 1 ? a->split_k * a->batch : a->batch);
        if (a->layout == HS_GEMM_NT) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, s, p);
        else if (a->layout == HS_GEMM_NN) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, s, p);
    }
    HS_CHECK_LAUNCH();
    return HS_OK;
}
