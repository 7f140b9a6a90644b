// K4: global multi-head self-attention core of MONAI SABlock (reference hot loops
// multimodal_encoder/vit.py:352-354,464-466): softmax(q k^T dh^-0.5) v, unmasked, dh = 64.
//
// Flash-style (the N x N matrix never exists).  All kernels put the index that the SECOND product sums over
// on the accumulator's ROW (register) axis, so the first product's accumulator is directly the next MFMA's
// B operand after a bf16 pack (cdna guide section 3, "An accumulator tile as the next MFMA's operand"):
//   fwd  : S^T[key][q] = K Q^T        -> P^T (softmax is lane-local + one half-swap) -> O^T[d][q] += V^T P^T
//   dq   : S^T, dP^T = V dO^T          -> dS^T                                       -> dQ^T[d][q] += K^T dS^T
//   dkdv : S[q][key] = Q K^T, dP = dO V^T -> P, dS                                    -> dV^T += dO^T P ; dK^T += Q^T dS
// K/V (fwd, dq) or Q/dO (dkdv) tiles live in LDS image R (common.h): read by rows (ds_read_b128) for the first
// product and by columns (ds_read_b64_tr_b16) for the second — one image, no transposed copy.
// Tiles are staged global -> registers (range-checked buffer loads: rows >= N read 0) -> LDS, written after
// the MFMA phase, one barrier per tile, double-buffered.
#include "common.h"
#include "attn_util.h"
#include <type_traits>

#define NEG_INF (-__builtin_inff())
#define HS_F16_PSHIFT 6        // attention backward at fp16 operands: P, dS carried as P * 2^6, dS * 2^6 (see attn_bwd_dq2_body)
#if HS_ABL & 1
#define HS_BWD_EXP2(x) (x)
#else
#define HS_BWD_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif

// XCD-aware block order for the attention kernels (1-D grid of nblk * H * B blocks).  Blocks are dealt round-robin
// over the 8 XCDs; with the natural order the ~33 blocks of one (batch, head) — which all stream the same K/V (or
// Q/dO) — land on all 8 XCDs and that head's 1 MiB is fetched 8 times (measured: 1.2-1.4 GB per launch against
// 0.25-0.4 GB algorithmic).  Give each XCD a CONTIGUOUS range of the (b, head, block) order instead (bijective form,
// cdna guide section 5): the blocks of a head then share one L2.  Placement only affects speed.
__device__ __forceinline__ void attn_block_coords(int nblk, int H, int& blk, int& head, int& b) {
    const int nb = gridDim.x, id = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, x = id & 7;
    const int v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    blk = v % nblk;
    const int bh = v / nblk;
    head = bh % H;
    b = bh / H;
}


// ------------------------------------------------------------------------------------------------------
// forward, bf16.  grid (ceil(N/128), H, B), 4 waves x 32 queries, 64-key tiles.
// ------------------------------------------------------------------------------------------------------
// GUARD: run only the 128-query blocks in which the fast kernel (below) left a NaN lse, i.e. rows whose optimistic
// softmax left the f32 window; every other block exits after one load.
template <bool GUARD, bool F16 = false>
__global__ __launch_bounds__(256, 3) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                            float* __restrict__ lse, int N, int H, float c2) {
    // c2 = log2(e) x the factor still to be applied to q.k (dh^-0.5, or 1/log2(e) when q arrives pre-scaled)
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
    int blk, head, b;
    attn_block_coords((N + 127) >> 7, H, blk, head, b);
    if (GUARD) {
        const int qg = blk * 128 + tid;
        const float v = (tid < 128 && qg < N) ? lse[((long)b * H + head) * N + qg] : 0.f;
        if (!__syncthreads_or(v != v)) return;
    }
    const int q0 = blk * 128 + wave * 32;
    const int Dm = H * 64, D3 = 3 * Dm;
    const bf16_t* base = qkv + (long)b * N * D3;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, (size_t)N * D3 * 2);

    const int qrow = min(q0 + c, N - 1);
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = __builtin_bit_cast(bf16x8, buf_load16(rs, (unsigned)(((long)qrow * D3 + head * 64 + ks * 16 + 8 * h) * 2)));

    u32x4 rk[2], rv[2];
    auto gload = [&](int t) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = it * 256 + tid, key = t * 64 + (idx >> 3);
            const unsigned off = key < N ? (unsigned)(((long)key * D3 + Dm + head * 64 + (idx & 7) * 8) * 2) : HS_OOB_OFF;
            rk[it] = buf_load16(rs, off);
            rv[it] = buf_load16(rs, key < N ? off + Dm * 2 : HS_OOB_OFF);
        }
    };
    auto swrite = [&](int st) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = it * 256 + tid, o_ = offR(idx >> 3, idx & 7);
            *(u32x4*)(smem + st * 16384 + o_) = rk[it];
            *(u32x4*)(smem + st * 16384 + 8192 + o_) = rv[it];
        }
    };

    float m = NEG_INF, l = 0.f;
    f32x16 oacc[2] = {zero16(), zero16()};
    const int nt = (N + 63) >> 6;
    // one 64-key tile.  LAST (the only tile that can hold keys >= N) is a separate instantiation, so the steady-state
    // loop carries no masking code at all (if-converted it was ~34 v_cndmask + exec juggling on every tile).
    auto tile = [&](int t, auto stage_tag, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int ST = decltype(stage_tag)::value;     // compile-time stage: LDS addresses = hoisted base + immediate
        const char* Kl = smem + ST * 16384;
        const char* Vl = Kl + 8192;
        f32x16 s[2] = {zero16(), zero16()};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s[kb] = mfma32<F16>(row_frag(Kl, kb * 32 + c, ks, h), qf[ks], s[kb]);
        if (LAST) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * 64 + kb * 32 + acc_row(r, h) >= N) s[kb][r] = NEG_INF;
        }
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float mc = mn * c2;
        float rs_ = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kb][r] * c2 - mc);
                s[kb][r] = p;
                rs_ += p;
            }
        rs_ += __shfl_xor(rs_, 32, 64);
        // rescale only when some query's running max moved (wave-uniform test): after the first few tiles it rarely
        // does, and alpha == 1 exactly then, so skipping is exact (no threshold)
        if (__builtin_amdgcn_ballot_w64(mn != m) != 0) {
            const float alpha = __builtin_amdgcn_exp2f((m - mn) * c2);
            l *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
            m = mn;
        }
        l += rs_;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 pf = pack8<F16>(s[kb], sp);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    oacc[db] = mfma32<F16>(tr_frag(Vl, kb * 32 + 16 * sp, db, lane), pf, oacc[db]);
            }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    gload(0);
    swrite(0);
    __syncthreads();
    int t = 0;
    for (; t + 2 < nt; t += 2) {       // ring unrolled by two: even tiles live in stage 0, odd tiles in stage 1
        gload(t + 1);
        tile(t, S0{}, std::false_type{});
        swrite(1);
        __syncthreads();
        gload(t + 2);
        tile(t + 1, S1{}, std::false_type{});
        swrite(0);
        __syncthreads();
    }
    if (t + 1 < nt) {                   // two tiles left: t (stage 0), then the last one (stage 1)
        gload(t + 1);
        tile(t, S0{}, std::false_type{});
        swrite(1);
        __syncthreads();
        tile(t + 1, S1{}, std::true_type{});
    } else {
        tile(t, S0{}, std::true_type{});
    }
    const int query = q0 + c;
    if (query < N) {
        store_dT<F16>(o + ((long)b * N + query) * Dm + head * 64, oacc, 1.0f / l, h);
        if (h == 0) lse[((long)b * H + head) * N + query] = m * (c2 * 0.69314718055994531f) + __logf(l);
    }
}

// ------------------------------------------------------------------------------------------------------
// forward, bf16, fast path ("optimistic softmax").  4 waves x 64 queries (two 32-query blocks A, B per wave that share
// every K / V fragment read), 64-key tiles by LDS-DMA into a 3-stage ring (prefetch distance 2, counted vmcnt, raw
// barrier: no staging registers, no ds_write).
//
// Softmax without a running maximum.  softmax is invariant to the reference point subtracted before exp; the online
// algorithm re-references to the running row maximum only to stay inside the f32 range.  Here the reference is fixed
// at 0: p = exp2(q' . k) with q' = q * scale * log2(e) folded into the Q fragments, l = sum p, O = sum p v, out = O / l,
// lse = ln l.  No max, no subtraction, no rescale: per score one v_exp, one add, half a cvt_pk — at head dim 64 the
// loop is bound by vector-issue slots, not by the matrix pipe, so this is the lever.  Nothing is approximated: f32 / bf16
// keep relative precision over 2^+-126, so the result equals the max-referenced one to rounding WHENEVER every row's
// l stays inside [2^-60, 2^100] (scaled logits within about -41 .. +63 nats).  A row outside that window gets
// lse = NaN, and hs_attn_fwd always runs the guarded exact kernel (attn_fwd_bf16_kernel with GUARD) behind this one: its
// 128-query blocks recompute, with the running maximum, exactly the blocks that hold a NaN lse and exit at once
// otherwise.  tests/test_gpu_kernels.py forces both the overflow and the underflow side.
// ------------------------------------------------------------------------------------------------------

// F16 (round 5): the same kernel on IEEE fp16 operands.  fp16 has 5 exponent bits, so the window in which the un-referenced P = exp2(s')
// is exact to rounding is narrower: a row is accepted when its l = sum P lies in [1, 2^15) -- then no P can have overflowed (every P < l <
// 65504) and the row's largest P is >= l / N, a normal fp16 number; any other row gets lse = NaN and is redone by the running-maximum kernel.
// Random-init attention (l ~ N) and moderately peaked rows (largest scaled logit below ~10 nats) stay on this kernel.
template <bool F16>
__device__ __forceinline__ void attn_fwd_fast_body(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                   float* __restrict__ lse, int N, int H, float qmul) {
    // qmul: factor applied to the Q fragments (dh^-0.5 * log2(e), or exactly 1 when q arrives pre-scaled)
    __shared__ __attribute__((aligned(16))) char smem[3 * 16384];     // stage: K image 8 KiB | V image 8 KiB
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk, head, b;
    attn_block_coords((N + 255) >> 8, H, blk, head, b);
    const int q0 = blk * 256 + wave * 64;
    const bool active = q0 < N;                                        // wave-uniform; false only in the last block
    const int Dm = H * 64, D3 = 3 * Dm;
    const bf16_t* base = qkv + (long)b * N * D3;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, (size_t)N * D3 * 2);
    const int nt = (N + 63) >> 6;

    // DMA: piece p (1 KiB = rows 8p..8p+7 of an image) of K and of V for p = 2*wave, 2*wave+1; the swizzle of image R
    // is applied to the SOURCE chunk.  Tile t adds the wave-uniform soffset t*64 rows; the last tile uses offsets
    // whose rows >= N are out of range (zero fill).
    unsigned vk[2], vkl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (2 * wave + j) * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ swzR(row);
        vk[j] = (unsigned)(((long)row * D3 + Dm + head * 64 + ch * 8) * 2);
        vkl[j] = (nt - 1) * 64 + row < N ? vk[j] : HS_OOB_OFF;
    }
    const unsigned tile_bytes = (unsigned)(64 * D3 * 2), v_off = (unsigned)(Dm * 2);
    auto issue = [&](int t, int st) {
        const bool last = t == nt - 1;
        const unsigned so = (unsigned)t * tile_bytes;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vo = last ? vkl[j] : vk[j];
            char* dst = smem + st * 16384 + (2 * wave + j) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))dst, 16, vo, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))(dst + 8192), 16, vo == HS_OOB_OFF ? vo : vo + v_off, so, 0, 0);
        }
    };
    issue(0, 0);
    if (nt > 1) issue(1, 1);

    bf16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = min(q0 + 32 * qb + c, N - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[qb][ks] = scale_frag<F16>(buf_load16(rs, (unsigned)(((long)qrow * D3 + head * 64 + ks * 16 + 8 * h) * 2)), qmul);
    }
    // transposed-read bases (image R): rows 4h+q and 4h+q+8 of a 16-row group, 32-column block db
    unsigned tb0[2][2];
    {
        const int g = lane >> 4, p = lane & 3, q = (lane >> 2) & 3;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int col = db * 32 + 16 * (g & 1) + 4 * p;
            tb0[db][0] = lds_addr(smem) + offR(4 * h + q, col >> 3) + 8 * (p & 1);
            tb0[db][1] = lds_addr(smem) + offR(4 * h + q + 8, col >> 3) + 8 * (p & 1);
        }
    }

    unsigned rbk[4];                                         // K row-fragment bases: row c, chunk 2ks+h
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rbk[ks] = lds_addr(smem) + offR(c, 2 * ks + h);
    float l[2] = {0.f, 0.f};
    f32x16 oacc[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};
    // MID tiles: compile-time stage (every LDS address = hoisted base + immediate).  The LAST tile (the only one that can
    // hold keys >= N) runs once: one instantiation with the stage folded into the base registers at run time.
    auto tile = [&](int t, auto stage_tag, auto last_tag, int rst) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int ST = LAST ? 0 : decltype(stage_tag)::value;
        constexpr int VB = ST * 16384 + 8192;
        unsigned tb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) tb[i][j] = tb0[i][j] + (LAST ? rst * 16384 : 0);
        unsigned rbase[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) rbase[ks] = rbk[ks] + (LAST ? rst * 16384 : 0);
        f32x16 s[2][2];                                     // [key block][query block]
        u32x4 va[2], vb[2];                                 // V fragments [db] of one 16-key step, double-buffered
        bf16x8 pf[2][2];                                    // packed P^T [query block][sp]
        auto softmax = [&](int kb) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = __builtin_amdgcn_exp2f(s[kb][qb][r]);
                    if (LAST && t * 64 + kb * 32 + acc_row(r, h) >= N) p = 0.f;
                    s[kb][qb][r] = p;
                    // (this file is built with -fno-slp-vectorize: -O3 would pack the two query blocks' sums into
                    // v_pk_add_f32, which costs more issue time beside MFMAs than the two adds it replaces.  Not done with
                    // an asm v_add: hipcc pads no hazards inside asm, and a VALU reading a v_exp result needs a wait state)
                    l[qb] += p;
                }
                pf[qb][0] = pack8<F16>(s[kb][qb], 0);
                pf[qb][1] = pack8<F16>(s[kb][qb], 1);
            }
        };
        auto pv = [&](u32x4 (&f)[2], int sp) {               // 4 MFMAs: 16 keys x (2 d blocks) x (2 query blocks)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    oacc[qb][db] = mfma32<F16>(__builtin_bit_cast(bf16x8, f[db]), pf[qb][sp], oacc[qb][db]);
        };
        // K row fragments of both key blocks are issued up front (asm reads, counted waits); each feeds two MFMAs
        u32x4 ka[4], kc[4];
        constexpr int KO = ST * 16384;
        ldsr128<KO>(ka[0], rbase[0]); ldsr128<KO>(ka[1], rbase[1]); ldsr128<KO>(ka[2], rbase[2]); ldsr128<KO>(ka[3], rbase[3]);
        ldsr128<KO + 4096>(kc[0], rbase[0]); ldsr128<KO + 4096>(kc[1], rbase[1]); ldsr128<KO + 4096>(kc[2], rbase[2]); ldsr128<KO + 4096>(kc[3], rbase[3]);
        s[0][0] = zero16(); s[0][1] = zero16(); s[1][0] = zero16(); s[1][1] = zero16();
        lds_wait4<4>(ka[0], ka[1], ka[2], ka[3]);
        // step 1: S^T of key block 0 for both query blocks
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 kf = __builtin_bit_cast(bf16x8, ka[ks]);
            s[0][0] = mfma32<F16>(kf, qf[0][ks], s[0][0]);
            s[0][1] = mfma32<F16>(kf, qf[1][ks], s[0][1]);
        }
        // V fragments of keys 0..15 fly while step 2 runs
        trR_issue<VB + 0>(va[0], tb[0][0], tb[0][1]);
        trR_issue<VB + 0>(va[1], tb[1][0], tb[1][1]);
        lds_wait4<4>(kc[0], kc[1], kc[2], kc[3]);
        // step 2: S^T of key block 1 || softmax of key block 0
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 kf = __builtin_bit_cast(bf16x8, kc[ks]);
            s[1][0] = mfma32<F16>(kf, qf[0][ks], s[1][0]);
            s[1][1] = mfma32<F16>(kf, qf[1][ks], s[1][1]);
        }
        softmax(0);
        // steps 3, 4: O^T += V^T P^T, 16 keys at a time, the next fragments in flight behind each group of MFMAs;
        // the softmax of key block 1 sits between the two halves of key block 0's product
        lds_settle2(va[0], va[1]);
        trR_issue<VB + 2048>(vb[0], tb[0][0], tb[0][1]);
        trR_issue<VB + 2048>(vb[1], tb[1][0], tb[1][1]);
        pv(va, 0);
        lds_settle2(vb[0], vb[1]);
        trR_issue<VB + 4096>(va[0], tb[0][0], tb[0][1]);
        trR_issue<VB + 4096>(va[1], tb[1][0], tb[1][1]);
        pv(vb, 1);
        softmax(1);
        lds_settle2(va[0], va[1]);
        trR_issue<VB + 6144>(vb[0], tb[0][0], tb[0][1]);
        trR_issue<VB + 6144>(vb[1], tb[1][0], tb[1][1]);
        pv(va, 0);
        lds_settle2(vb[0], vb[1]);
        pv(vb, 1);
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 2> S2;
    // ring of 3 stages, prefetch distance 2.  At tile t: all but the newest tile's 4 DMA pieces have landed (vmcnt(4)),
    // barrier (every wave's pieces of tile t have landed AND every wave is done reading tile t-1), then tile t+2 goes
    // into the stage tile t-1 occupied.
    auto step = [&](int t, auto stage_tag) {               // a tile that is not the last one
        constexpr int ST = decltype(stage_tag)::value;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) issue(t + 2, (ST + 2) % 3);
        if (active) tile(t, stage_tag, std::false_type{}, 0);
    };
    int t = 0;
    for (; t + 3 < nt; t += 3) {
        step(t, S0{});
        step(t + 1, S1{});
        step(t + 2, S2{});
    }
    int rst = 0;
    if (t + 1 < nt) { step(t, S0{}); ++t; rst = 1; }
    if (t + 1 < nt) { step(t, S1{}); ++t; rst = 2; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (active) tile(t, S0{}, std::true_type{}, rst);
    if (!active) return;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int query = q0 + 32 * qb + c;
        const float lt = l[qb] + __shfl_xor(l[qb], 32, 64);
        const bool ok = F16 ? (lt >= 1.0f && lt < 32768.0f) : (lt > 8.6736174e-19f && lt < 1.2676506e30f);   // bf16: 2^-60 .. 2^100; fp16: 1 .. 2^15
        if (query < N) {
            store_dT<F16>(o + ((long)b * N + query) * Dm + head * 64, oacc[qb], 1.0f / lt, h);
            if (h == 0) lse[((long)b * H + head) * N + query] = ok ? __logf(lt) : __builtin_nanf("");
        }
    }
}
__global__ __launch_bounds__(256, 2) void attn_fwd_fast_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                            float* __restrict__ lse, int N, int H, float qmul) {
    attn_fwd_fast_body<false>(qkv, o, lse, N, H, qmul);
}
__global__ __launch_bounds__(256, 2) void attn_fwd_fast_f16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                                float* __restrict__ lse, int N, int H, float qmul) {
    attn_fwd_fast_body<true>(qkv, o, lse, N, H, qmul);
}

// ------------------------------------------------------------------------------------------------------
// forward, exact f32 (parity mode): same data flow with v_mfma_f32_32x32x2_f32.  K/V tiles are fp32 images
// with 256-B rows and 16-B chunk c of row r at c ^ (r & 15).  The k (= d) order inside a group of 8 is permuted
// identically for K and Q (half h takes d = 8s+4h+e at MFMA step e); the PV product takes the accumulator
// registers as they stand (half h of register r holds key acc_row(r,h)).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int offF(int r, int c) { return r * 256 + ((c ^ (r & 15)) << 4); }

__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                           float* __restrict__ lse, int N, int H, float c2) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
    const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 128 + wave * 32;
    const int Dm = H * 64, D3 = 3 * Dm;
    const float* base = qkv + (long)b * N * D3;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, (size_t)N * D3 * 4);
    const int qrow = min(q0 + c, N - 1);
    f32x4 qf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
        qf[s] = __builtin_bit_cast(f32x4, buf_load16(rs, (unsigned)(((long)qrow * D3 + head * 64 + 8 * s + 4 * h) * 4)));
    u32x4 rk[4], rv[4];
    auto gload = [&](int t) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + tid, key = t * 64 + (idx >> 4);
            const unsigned off = key < N ? (unsigned)(((long)key * D3 + Dm + head * 64 + (idx & 15) * 4) * 4) : HS_OOB_OFF;
            rk[it] = buf_load16(rs, off);
            rv[it] = buf_load16(rs, key < N ? off + Dm * 4 : HS_OOB_OFF);
        }
    };
    auto swrite = [&](int st) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + tid, o_ = offF(idx >> 4, idx & 15);
            *(u32x4*)(smem + st * 32768 + o_) = rk[it];
            *(u32x4*)(smem + st * 32768 + 16384 + o_) = rv[it];
        }
    };
    float m = NEG_INF, l = 0.f;
    f32x16 oacc[2] = {zero16(), zero16()};
    const int nt = (N + 63) >> 6;
    gload(0);
    swrite(0);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        const bool more = t + 1 < nt;
        if (more) gload(t + 1);
        const char* Kl = smem + cur * 32768;
        const char* Vl = Kl + 16384;
        f32x16 s[2] = {zero16(), zero16()};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const f32x4 a = *(const f32x4*)(Kl + offF(kb * 32 + c, 2 * s8 + h));
#pragma unroll
                for (int e = 0; e < 4; ++e) s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qf[s8][e], s[kb], 0, 0, 0);
            }
        if (t == nt - 1 && (N & 63)) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * 64 + kb * 32 + acc_row(r, h) >= N) s[kb][r] = NEG_INF;
        }
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = exp2f((m - mn) * c2);
        float rs_ = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = exp2f((s[kb][r] - mn) * c2);
                s[kb][r] = p;
                rs_ += p;
            }
        rs_ += __shfl_xor(rs_, 32, 64);
        l = l * alpha + rs_;
        m = mn;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + acc_row(r, h);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const int d = db * 32 + c;
                    const float a = *(const float*)(Vl + offF(key, d >> 2) + (d & 3) * 4);
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[kb][r], oacc[db], 0, 0, 0);
                }
            }
        if (more) swrite(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    const int query = q0 + c;
    if (query < N) {
        const float inv = 1.0f / l;
        float* orow = o + ((long)b * N + query) * Dm + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = oacc[db][4 * g4 + e] * inv;
                *(f32x4*)(orow + db * 32 + 8 * g4 + 4 * h) = v;
            }
        if (h == 0 && lse) lse[((long)b * H + head) * N + query] = m * (c2 * 0.69314718055994531f) + logf(l);
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, exact f32 (parity mode).  Same two sweeps as the bf16 kernels below, built like attn_fwd_f32_kernel on
// v_mfma_f32_32x32x2_f32 with scalar operands: the STATIONARY rows (queries for dQ, keys for dK/dV) keep their two 64-float
// vectors in registers as B operands, the STREAMED rows (keys / queries) arrive in 64-row tiles in LDS and are read as rows (A of
// the S and dP products) and as scalars (A of the accumulation products), and an accumulator register of S / dS is directly the B
// scalar of the accumulation product (its key/query index = acc_row(r, h) = the k index of that MFMA step).
//   MODE 0 (dQ):     X = q, Y = dO of the block's 128 queries; tiles T1 = K, T2 = V;   S^T = T1 X^T, dP^T = T2 Y^T,
//                    dS^T = exp2(S^T c2 - lse') (dP^T - delta)  [row constants per LANE];      dQ^T += T1^T dS^T
//   MODE 1 (dK, dV): X = k, Y = v of the block's 128 keys;     tiles T1 = Q, T2 = dO;  S = T1 X^T, dP = T2 Y^T,
//                    P = exp2(S c2 - lse'_q), dS = P (dP - delta_q)  [row constants per REGISTER, from the tile's LDS copy];
//                    dV^T += T2^T P, dK^T += T1^T dS
// ws: [0] = -delta, [1] = -lse * log2(e), each [B][H][N] (attn_stats_f32_kernel).  No rounding anywhere: fp32 end to end.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_stats_f32_kernel(const float* __restrict__ o, const float* __restrict__ d_o,
                                                             const float* __restrict__ lse, float* __restrict__ ws, int B, int N, int H) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;          // 16 lanes per (b, n, head) row of 64 floats
    const long total = (long)B * N * H * 16;
    float acc = 0.f;
    const bool ok = gid < total;
    if (ok) {
        const f32x4 a = *(const f32x4*)(o + gid * 4), g = *(const f32x4*)(d_o + gid * 4);
        acc = a[0] * g[0] + a[1] * g[1] + a[2] * g[2] + a[3] * g[3];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 8, 64);
    if (ok && (threadIdx.x & 15) == 0) {
        const long row = gid >> 4;                                  // (b*N + n)*H + head
        const int head = (int)(row % H);
        const long bn = row / H;
        const int n = (int)(bn % N), b = (int)(bn / N);
        const long i = ((long)b * H + head) * N + n;
        ws[i] = -acc;
        ws[(long)B * H * N + i] = -lse[i] * 1.44269504088896341f;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                           const float* __restrict__ ws, float* __restrict__ dqkv, int B, int N, int H,
                                                           float c2, float out_mul) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 32768 + 2 * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
    const int b = blockIdx.z, head = blockIdx.y, r0 = blockIdx.x * 128 + wave * 32;
    const int Dm = H * 64, D3 = 3 * Dm;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv + (long)b * N * D3, (size_t)N * D3 * 4);
    const __amdgpu_buffer_rsrc_t rsd = make_rsrc(d_o + (long)b * N * Dm, (size_t)N * Dm * 4);
    const float* nd = ws + ((long)b * H + head) * N;                 // -delta
    const float* nl = nd + (long)B * H * N;                          // -lse * log2(e)
    const int srow = min(r0 + c, N - 1);
    f32x4 xf[8], yf[8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
        const unsigned col = (unsigned)(head * 64 + 8 * s8 + 4 * h);
        if (MODE == 0) {
            xf[s8] = __builtin_bit_cast(f32x4, buf_load16(rs, (unsigned)(((long)srow * D3 + col) * 4)));
            yf[s8] = __builtin_bit_cast(f32x4, buf_load16(rsd, (unsigned)(((long)srow * Dm + col) * 4)));
        } else {
            xf[s8] = __builtin_bit_cast(f32x4, buf_load16(rs, (unsigned)(((long)srow * D3 + Dm + col) * 4)));
            yf[s8] = __builtin_bit_cast(f32x4, buf_load16(rs, (unsigned)(((long)srow * D3 + 2 * Dm + col) * 4)));
        }
    }
    const float cnl = nl[srow], cnd = nd[srow];                      // MODE 0: the stationary query's row constants
    u32x4 r1[4], r2[4];
    float rc = 0.f;
    auto gload = [&](int t) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + tid, row = t * 64 + (idx >> 4), ch = idx & 15;
            if (MODE == 0) {
                const unsigned off = row < N ? (unsigned)(((long)row * D3 + Dm + head * 64 + ch * 4) * 4) : HS_OOB_OFF;
                r1[it] = buf_load16(rs, off);
                r2[it] = buf_load16(rs, row < N ? off + Dm * 4 : HS_OOB_OFF);
            } else {
                r1[it] = buf_load16(rs, row < N ? (unsigned)(((long)row * D3 + head * 64 + ch * 4) * 4) : HS_OOB_OFF);
                r2[it] = buf_load16(rsd, row < N ? (unsigned)(((long)row * Dm + head * 64 + ch * 4) * 4) : HS_OOB_OFF);
            }
        }
        if (MODE == 1 && tid < 128) {
            const int j = t * 64 + (tid & 63);
            rc = j < N ? (tid < 64 ? nl[j] : nd[j]) : 0.f;
        }
    };
    auto swrite = [&](int st) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + tid, o_ = offF(idx >> 4, idx & 15);
            *(u32x4*)(smem + st * 32768 + o_) = r1[it];
            *(u32x4*)(smem + st * 32768 + 16384 + o_) = r2[it];
        }
        if (MODE == 1 && tid < 128) ((float*)(smem + 65536 + st * 512))[tid] = rc;
    };
    f32x16 acc0[2] = {zero16(), zero16()}, acc1[2] = {zero16(), zero16()};
    const int nt = (N + 63) >> 6;
    gload(0);
    swrite(0);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        const bool more = t + 1 < nt;
        if (more) gload(t + 1);
        const char* T1 = smem + cur * 32768;
        const char* T2 = T1 + 16384;
        const float* cst = (const float*)(smem + 65536 + cur * 512);
        f32x16 s[2] = {zero16(), zero16()}, dp[2] = {zero16(), zero16()};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const f32x4 a1 = *(const f32x4*)(T1 + offF(kb * 32 + c, 2 * s8 + h));
                const f32x4 a2 = *(const f32x4*)(T2 + offF(kb * 32 + c, 2 * s8 + h));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], xf[s8][e], s[kb], 0, 0, 0);
                    dp[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[e], yf[s8][e], dp[kb], 0, 0, 0);
                }
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jl = kb * 32 + acc_row(r, h);                // streamed row inside the tile
                const bool valid = t * 64 + jl < N;
                const float l_ = MODE == 0 ? cnl : cst[jl], d_ = MODE == 0 ? cnd : cst[64 + jl];
                const float p = valid ? exp2f(s[kb][r] * c2 + l_) : 0.f;
                s[kb][r] = p;
                dp[kb][r] = p * (dp[kb][r] + d_);
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = kb * 32 + acc_row(r, h);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const int d = db * 32 + c;
                    const float a1 = *(const float*)(T1 + offF(row, d >> 2) + (d & 3) * 4);
                    if (MODE == 0) {
                        acc0[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, dp[kb][r], acc0[db], 0, 0, 0);      // dQ^T += K^T dS^T
                    } else {
                        const float a2 = *(const float*)(T2 + offF(row, d >> 2) + (d & 3) * 4);
                        acc0[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, s[kb][r], acc0[db], 0, 0, 0);       // dV^T += dO^T P
                        acc1[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, dp[kb][r], acc1[db], 0, 0, 0);      // dK^T += Q^T dS
                    }
                }
            }
        if (more) swrite(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    const int srow_out = r0 + c;
    if (srow_out < N) {
        float* out = dqkv + ((long)b * N + srow_out) * D3 + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 v0, v1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = acc0[db][4 * g4 + e]; v1[e] = acc1[db][4 * g4 + e] * out_mul; }
                const int col = db * 32 + 8 * g4 + 4 * h;
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v0[e] *= out_mul;
                    *(f32x4*)(out + col) = v0;                           // dQ
                } else {
                    *(f32x4*)(out + 2 * Dm + col) = v0;                  // dV
                    *(f32x4*)(out + Dm + col) = v1;                      // dK
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, bf16
// ------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------
// backward v2.  Per (b, head, query): ws[0] = -delta = -sum_d dO*O and ws[1] = -lse*log2(e) (written by the dQ sweep, which runs first), the
// two row constants both sweeps preload into their MFMA accumulators (S' = q'.k - lse*log2e, dP' = dO.v - delta), so
// that P = exp2(S') and dS = P * dP' need no subtraction (cdna guide, attention backward, "row constants as the initial
// accumulator").  P = exp2(S' ) <= 1 always: unlike the forward fast path nothing here is optimistic.
// ------------------------------------------------------------------------------------------------------

// Image R16: the same [rows][64 bf16] tile with the swizzle the 16x16x32 operand maps need: 16-B chunk c of row r at chunk
// c ^ (r & 6).  ds_read_b128 of the 16-row x 4-chunk operand: a hardware lane group {0-3,12-15,20-27} (and its three siblings) holds
// rows {0-3,12-15} with chunk k and rows {4-11} with chunk k^1; slot = 8(r&1) + (chunk ^ (r&6)) is then a bijection onto the 16
// 16-B slots of the bank row.  ds_read_b64_tr_b16 (banking per 32-lane half = 8 rows x two chunks 2di, 2di+1): rows of one parity
// land on the four chunk pairs di ^ ((r>>1)&3).  Both conflict-free (SQ_LDS_BANK_CONFLICT = 0); with swzR the row read was 2-way.
__device__ __forceinline__ int swz16(int r) { return r & 6; }
__device__ __forceinline__ int offR16(int r, int c) { return r * 128 + ((c ^ swz16(r)) << 4); }

// dQ v2: NW waves x 32 queries per block; K/V tiles by LDS-DMA (3-stage ring, prefetch distance 2).  Per 32-key block:
// S'^T = K q'^T + (-lse*log2e), dP'^T = V dO^T + (-delta)  (row constants = splat accumulators, loop-invariant here),
// dS^T = exp2(S'^T) * dP'^T -> bf16 -> dQ^T[d][q] += K^T dS^T (K read transposed from the same image).
// NW = 4 with two blocks per CU: the two waves of a SIMD then belong to different blocks and drift out of phase, so one
// wave's exp/convert stretch runs under the other's MFMAs (two waves of ONE block are phase-locked by the tile barrier).
// The dQ sweep runs FIRST and also produces the two row constants (round 3; there used to be a separate statistics kernel that read
// O and dO once more): a wave holds the dO rows of its 32 queries as MFMA operands anyway, so it loads the matching O rows (16 B per
// lane and k-step), forms delta = sum_d dO*O in registers (8 products per k-step and lane, the two lanes of a row combined by one
// shuffle), takes lse from the forward, and writes ws[0] = -delta, ws[1] = -lse*log2(e) for the dK/dV sweep launched after it.
// F16: the same sweep on IEEE fp16 operands (HS_ATTN_F16, round 5): P = exp2(S') <= 1 and dS = P * dP' are packed as fp16, the MFMAs are the
// fp16 opcode; everything else (images, ring, waits) is format-blind.
template <int NW, bool F16>
__device__ __forceinline__ void attn_bwd_dq2_body(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                  const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                  float* __restrict__ ws, bf16_t* __restrict__ dqkv, int B,
                                                  int N, int H, float qmul, float out_mul) {
    __shared__ __attribute__((aligned(16))) char smem[3 * 16384];
    constexpr int QB = NW * 32, NP = 8 / NW;                 // queries per block; K (and V) DMA pieces per wave per tile
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk, head, b;
    attn_block_coords((N + QB - 1) / QB, H, blk, head, b);
    const int q0 = blk * QB + wave * 32;
    const bool active = q0 < N;
    const int Dm = H * 64, D3 = 3 * Dm;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv + (long)b * N * D3, (size_t)N * D3 * 2);
    const __amdgpu_buffer_rsrc_t rsd = make_rsrc(d_o + (long)b * N * Dm, (size_t)N * Dm * 2);
    const int nt = (N + 63) >> 6;
    unsigned vk[NP], vkl[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int row = (wave * NP + j) * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ swzR(row);
        vk[j] = (unsigned)(((long)row * D3 + Dm + head * 64 + ch * 8) * 2);
        vkl[j] = (nt - 1) * 64 + row < N ? vk[j] : HS_OOB_OFF;
    }
    const unsigned tile_bytes = (unsigned)(64 * D3 * 2), v_off = (unsigned)(Dm * 2);
    auto issue = [&](int t, int st) {
        const unsigned so = (unsigned)t * tile_bytes;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const unsigned vo = t == nt - 1 ? vkl[j] : vk[j];
            char* dst = smem + st * 16384 + (wave * NP + j) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))dst, 16, vo, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))(dst + 8192), 16, vo == HS_OOB_OFF ? vo : vo + v_off, so, 0, 0);
        }
    };
    issue(0, 0);
    if (nt > 1) issue(1, 1);

    const int qrow = min(q0 + c, N - 1);
    bf16x8 qf[4], dof[4];
    const __amdgpu_buffer_rsrc_t rso = make_rsrc(o + (long)b * N * Dm, (size_t)N * Dm * 2);
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = scale_frag<F16>(buf_load16(rs, (unsigned)(((long)qrow * D3 + head * 64 + ks * 16 + 8 * h) * 2)), qmul);
        const unsigned off = (unsigned)(((long)qrow * Dm + head * 64 + ks * 16 + 8 * h) * 2);
        const u32x4 g = buf_load16(rsd, off), a = buf_load16(rso, off);
        dof[ks] = __builtin_bit_cast(bf16x8, g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dsum += unpack_lo<F16>(a[i]) * unpack_lo<F16>(g[i]);
            dsum += unpack_hi<F16>(a[i]) * unpack_hi<F16>(g[i]);
        }
    }
    dsum += __shfl_xor(dsum, 32, 64);                        // lanes c and c + 32 hold the two halves of every 16-wide k-step of row qrow
    const long srow = ((long)b * H + head) * N + qrow;
    // fp16: P enters as P * 2^HS_F16_PSHIFT (the shift rides in the row constant, for free): dS = P dP' of a near-uniform row is ~|dP'| / N,
    // which for N = 4097 sits at the bottom of fp16's normal range (6.1e-5); the factor is taken out again where dQ / dK / dV are stored
    const float nd = -dsum, nl = -lse[srow] * 1.44269504088896341f + (F16 ? (float)HS_F16_PSHIFT : 0.0f);
    if (h == 0 && q0 + c < N) {                              // one writer per query row (rows past N are clamped duplicates)
        ws[srow] = nd;
        ws[(long)B * H * N + srow] = nl;
    }
    f32x16 cd = splat16(nd), cl = splat16(nl);
    asm volatile("" : "+v"(cd), "+v"(cl));          // opaque: keep the 2 x 16 registers instead of re-splatting per use
    unsigned rb0[4];                                         // row-fragment bases: row c, chunk 2ks+h (image R swizzle)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rb0[ks] = lds_addr(smem) + offR(c, 2 * ks + h);
    unsigned tb0[2][2];
    {
        const int g = lane >> 4, p = lane & 3, q = (lane >> 2) & 3;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int col = db * 32 + 16 * (g & 1) + 4 * p;
            tb0[db][0] = lds_addr(smem) + offR(4 * h + q, col >> 3) + 8 * (p & 1);
            tb0[db][1] = lds_addr(smem) + offR(4 * h + q + 8, col >> 3) + 8 * (p & 1);
        }
    }
    f32x16 dq[2] = {zero16(), zero16()};
    auto tile = [&](int t, auto stage_tag, auto last_tag, int rst) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int ST = LAST ? 0 : decltype(stage_tag)::value;
        constexpr int KB = ST * 16384;
        unsigned tb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) tb[i][j] = tb0[i][j] + (LAST ? rst * 16384 : 0);
        f32x16 st[2], dpt[2];
        u32x4 fa[4], fb[4];                                  // K^T fragments [sp*2 + db] of key block 0 / 1
        u32x4 ra[4], rb_[4];                                 // row fragments {K ks, V ks, K ks+1, V ks+1}, double-buffered
        unsigned rbase[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) rbase[ks] = rb0[ks] + (LAST ? rst * 16384 : 0);
        // group G = (kb, half): K and V row fragments of k-steps 2*half, 2*half+1 of key block kb
        auto rows = [&](u32x4 (&r)[4], auto kb_tag, auto half_tag) {
            constexpr int OFF = KB + decltype(kb_tag)::value * 4096, HF = decltype(half_tag)::value;
            ldsr128<OFF>(r[0], rbase[2 * HF]);
            ldsr128<OFF + 8192>(r[1], rbase[2 * HF]);
            ldsr128<OFF>(r[2], rbase[2 * HF + 1]);
            ldsr128<OFF + 8192>(r[3], rbase[2 * HF + 1]);
        };
        auto sdp = [&](u32x4 (&r)[4], int kb, int half) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                st[kb] = mfma32<F16>(__builtin_bit_cast(bf16x8, r[2 * i]), qf[2 * half + i], st[kb]);
                dpt[kb] = mfma32<F16>(__builtin_bit_cast(bf16x8, r[2 * i + 1]), dof[2 * half + i], dpt[kb]);
            }
        };
        auto valu = [&](int kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ds = HS_BWD_EXP2(st[kb][r]) * dpt[kb][r];
                if (LAST && t * 64 + kb * 32 + acc_row(r, h) >= N) ds = 0.f;
                dpt[kb][r] = ds;
            }
        };
        auto acc = [&](u32x4 (&f)[4], int kb) {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 dsf = pack8<F16>(dpt[kb], sp);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    dq[db] = mfma32<F16>(__builtin_bit_cast(bf16x8, f[sp * 2 + db]), dsf, dq[db]);
            }
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;
        st[0] = cl; dpt[0] = cd; st[1] = cl; dpt[1] = cd;
        rows(ra, I0{}, I0{});
        rows(rb_, I0{}, I1{});
        lds_wait4<4>(ra[0], ra[1], ra[2], ra[3]);
        sdp(ra, 0, 0);
        rows(ra, I1{}, I0{});
        lds_wait4<4>(rb_[0], rb_[1], rb_[2], rb_[3]);
        sdp(rb_, 0, 1);
        rows(rb_, I1{}, I1{});
        lds_wait4<4>(ra[0], ra[1], ra[2], ra[3]);
        sdp(ra, 1, 0);
        // (never more than 15 LDS reads in flight: lgkmcnt is a 4-bit counter)
        trR_issue<KB + 0>(fa[0], tb[0][0], tb[0][1]);            // 8 transposed reads: K^T fragments of key block 0
        trR_issue<KB + 0>(fa[1], tb[1][0], tb[1][1]);
        trR_issue<KB + 2048>(fa[2], tb[0][0], tb[0][1]);
        trR_issue<KB + 2048>(fa[3], tb[1][0], tb[1][1]);
        valu(0);
        lds_wait4<8>(rb_[0], rb_[1], rb_[2], rb_[3]);
        sdp(rb_, 1, 1);
        lds_wait4<0>(fa[0], fa[1], fa[2], fa[3]);
        trR_issue<KB + 4096>(fb[0], tb[0][0], tb[0][1]);
        trR_issue<KB + 4096>(fb[1], tb[1][0], tb[1][1]);
        trR_issue<KB + 6144>(fb[2], tb[0][0], tb[0][1]);
        trR_issue<KB + 6144>(fb[3], tb[1][0], tb[1][1]);
        acc(fa, 0);
        valu(1);
        lds_wait4<0>(fb[0], fb[1], fb[2], fb[3]);
        acc(fb, 1);
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 2> S2;
    auto step = [&](int t, auto stage_tag) {
        constexpr int ST = decltype(stage_tag)::value;
        if (NP == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) issue(t + 2, (ST + 2) % 3);
        if (active) tile(t, stage_tag, std::false_type{}, 0);
    };
    int t = 0;
    for (; t + 3 < nt; t += 3) {
        step(t, S0{});
        step(t + 1, S1{});
        step(t + 2, S2{});
    }
    int rst = 0;
    if (t + 1 < nt) { step(t, S0{}); ++t; rst = 1; }
    if (t + 1 < nt) { step(t, S1{}); ++t; rst = 2; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!active) return;
    tile(t, S0{}, std::true_type{}, rst);
    const int query = q0 + c;
    if (query < N) store_dT<F16>(dqkv + ((long)b * N + query) * D3 + head * 64, dq, out_mul, h);
}
// (thin non-template kernels: hipcc's host pass can drop the launch stub of a templated __global__ whose body holds constrained asm)
__global__ __launch_bounds__(256, 2) void attn_bwd_dq2_w4_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                                const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                                float* __restrict__ ws, bf16_t* __restrict__ dqkv, int B, int N, int H,
                                                                float qmul, float out_mul) {
    attn_bwd_dq2_body<4, false>(qkv, d_o, o, lse, ws, dqkv, B, N, H, qmul, out_mul);
}
__global__ __launch_bounds__(256, 2) void attn_bwd_dq2_w4_f16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                                    const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                                    float* __restrict__ ws, bf16_t* __restrict__ dqkv, int B, int N, int H,
                                                                    float qmul, float out_mul) {
    attn_bwd_dq2_body<4, true>(qkv, d_o, o, lse, ws, dqkv, B, N, H, qmul, out_mul);
}

// dK, dV: NW waves x 32 keys per block (K, V fragments and the dK^T, dV^T accumulators stay in registers), sweeping the queries in
// 64-row tiles that arrive by LDS-DMA: Q' image | dO image | -lse*log2e [64] | -delta [64] per stage, 3-stage ring.  Per 32-query
// block: S' = Q' K^T + (-lse*log2e)[q], dP' = dO V^T + (-delta)[q] (row constants preloaded as the accumulators, read from the stage),
// P = exp2(S'), dS = P * dP'; dV^T += dO^T P, dK^T += Q'^T dS (Q', dO read transposed from the same images).  Query rows >= N arrive as
// zeros (range-checked DMA): P = 1 there but dO = 0 and dP' = 0, so they add nothing.  Every LDS read is issued from asm in chunks
// ahead of use, with counted waits.
// Built on v_mfma_f32_16x16x32_bf16 -- the shape the chip runs 7 % faster at the power cap for this instruction mix (DESIGN.md
// section 4, tools/micro/mfma_mix.cpp); the 32x32x16 version of round 1 (git history) had the same block shape, stage layout, DMA ring
// and LDS byte counts.  The 32-query x 32-key block of a wave is 2 x 2 tiles of 16 x 16:
//   S'(qi,ki)  = sum over 2 k-steps of  A = Q' rows 16qi + (lane & 15), d = 32ks + 8g..8g+7   (row fragment: chunk 4ks + g of the row)
//                                       B = K  key  16ki + (lane & 15), same d                 (registers, loaded once)
//                accumulator: lane (key column lane & 15, g = lane >> 4) holds query rows 16qi + 4g + r, r = 0..3, so the row
//                constants of a tile are ONE broadcast 16-B read (they enter as the C operand of the first k-step: no moves)
//   P / dS as the B operand of dV^T / dK^T (k = query): the two stacked tiles (qi = 0, 1) of key tile ki give a lane its 8 k-slots:
//                slot j < 4 <-> query 4g + j, slot j >= 4 <-> query 16 + 4g + (j - 4)
//   A = dO^T / Q'^T tile di (16 d columns): two ds_read_b64_tr_b16 per fragment, 16-lane group g reading rows 4g..4g+3 and
//                16+4g..16+4g+3 of the 32-query block at columns 16di..16di+15 -- the same k-slot order
//   dV^T(di,ki), dK^T(di,ki): lane (key, g) holds d = 16di + 4g + r -> four consecutive d per 8-byte store.
template <int NW, bool F16>
__device__ __forceinline__ void attn_bwd_dkdv16_body(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                     const float* __restrict__ ws, bf16_t* __restrict__ dqkv, int B,
                                                     int N, int H, float kmul, float dk_mul, float dv_mul) {
    constexpr int STG = 16896, KBK = NW * 32, NP = 8 / NW;
    __shared__ __attribute__((aligned(16))) char smem[3 * STG];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk, head, b;
    attn_block_coords((N + KBK - 1) / KBK, H, blk, head, b);
    const int k0 = blk * KBK + wave * 32;
    const bool active = k0 < N;
    const int Dm = H * 64, D3 = 3 * Dm;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv + (long)b * N * D3, (size_t)N * D3 * 2);
    const __amdgpu_buffer_rsrc_t rsd = make_rsrc(d_o + (long)b * N * Dm, (size_t)N * Dm * 2);
    const __amdgpu_buffer_rsrc_t rst_ = make_rsrc(ws + (wave == 0 ? (long)B * H * N : 0) + ((long)b * H + head) * N, (size_t)N * 4);
    const int nt = (N + 63) >> 6;
    unsigned vq[NP], vd[NP], vql[NP], vdl[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int row = (wave * NP + j) * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ swz16(row);
        vq[j] = (unsigned)(((long)row * D3 + head * 64 + ch * 8) * 2);
        vd[j] = (unsigned)(((long)row * Dm + head * 64 + ch * 8) * 2);
        const bool in = (nt - 1) * 64 + row < N;
        vql[j] = in ? vq[j] : HS_OOB_OFF;
        vdl[j] = in ? vd[j] : HS_OOB_OFF;
    }
    const unsigned vs = (unsigned)lane * 4u, vsl = (nt - 1) * 64 + lane < N ? vs : HS_OOB_OFF;
    const unsigned qtile = (unsigned)(64 * D3 * 2), dtile = (unsigned)(64 * Dm * 2);
    auto issue = [&](int t, int st) {
        const bool last = t == nt - 1;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            char* dst = smem + st * STG + (wave * NP + j) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))dst, 16, last ? vql[j] : vq[j], (unsigned)t * qtile, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (LDS_PTR(void))(dst + 8192), 16, last ? vdl[j] : vd[j], (unsigned)t * dtile, 0, 0);
        }
        if (wave < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rst_, (LDS_PTR(void))(smem + st * STG + 16384 + wave * 256), 4, last ? vsl : vs,
                                                     (unsigned)t * 256u, 0, 0);
    };
    issue(0, 0);
    if (nt > 1) issue(1, 1);

    bf16x8 kf[2][2], vf[2][2];                                   // [key tile ki][k-step ks]
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) {
        const int key = k0 + 16 * ki + c;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned off = key < N ? (unsigned)(((long)key * D3 + Dm + head * 64 + ks * 32 + 8 * g) * 2) : HS_OOB_OFF;
            kf[ki][ks] = scale_frag<F16>(buf_load16(rs, off), kmul);
            vf[ki][ks] = __builtin_bit_cast(bf16x8, buf_load16(rs, key < N ? off + Dm * 2 : HS_OOB_OFF));
        }
    }
    unsigned rb0[2], tb0[4][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) rb0[ks] = lds_addr(smem) + offR16(c, 4 * ks + g);     // rows +16 / +32: same swizzle (bits 1-2)
    {
        const int p = lane & 3, q = (lane >> 2) & 3;
#pragma unroll
        for (int di = 0; di < 4; ++di) {
            tb0[di][0] = lds_addr(smem) + offR16(4 * g + q, 2 * di + (p >> 1)) + 8 * (p & 1);
            tb0[di][1] = lds_addr(smem) + offR16(4 * g + q + 16, 2 * di + (p >> 1)) + 8 * (p & 1);
        }
    }
    const unsigned sb0 = lds_addr(smem) + 16384 + 16 * g;        // row constants of rows 4g..4g+3 of a 16-row tile

    f32x4 dk[4][2], dv[4][2];                                     // [d tile di][key tile ki]
#pragma unroll
    for (int di = 0; di < 4; ++di)
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) { dk[di][ki] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[di][ki] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    auto tile = [&](auto stage_tag, int rst) {
        constexpr int ST = decltype(stage_tag)::value;            // -1: run-time stage (last tile)
        constexpr int SB = ST < 0 ? 0 : ST * STG;
        const unsigned dyn = ST < 0 ? (unsigned)rst * STG : 0u;
        unsigned rbase[2], tb[4][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) rbase[ks] = rb0[ks] + dyn;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) tb[i][j] = tb0[i][j] + dyn;
        const unsigned sbase = sb0 + dyn;
        u32x4 c1[2][4];                                      // row constants [qb][{lse qi0, delta qi0, lse qi1, delta qi1}]
        u32x4 ra[4], rb_[4];                                 // row fragments of one 16-query tile {Q ks0, dO ks0, Q ks1, dO ks1}
        u32x4 ta[4], tc[4];                                  // transposed fragments {dO^T di, dO^T di+1, Q^T di, Q^T di+1}
        f32x4 s[2][2][2], dp[2][2][2];                       // [qb][qi][ki]
        auto stats = [&](auto qb_tag) {
            constexpr int QB = decltype(qb_tag)::value, O = SB + QB * 128;
            ldsr128<O>(c1[QB][0], sbase); ldsr128<O + 256>(c1[QB][1], sbase);
            ldsr128<O + 64>(c1[QB][2], sbase); ldsr128<O + 320>(c1[QB][3], sbase);
        };
        auto rows = [&](u32x4 (&r)[4], auto qb_tag, auto qi_tag) {
            constexpr int OFF = SB + decltype(qb_tag)::value * 4096 + decltype(qi_tag)::value * 2048;
            ldsr128<OFF>(r[0], rbase[0]);
            ldsr128<OFF + 8192>(r[1], rbase[0]);
            ldsr128<OFF>(r[2], rbase[1]);
            ldsr128<OFF + 8192>(r[3], rbase[1]);
        };
        auto trs = [&](u32x4 (&f)[4], auto qb_tag, auto half_tag) {
            constexpr int OFF = SB + decltype(qb_tag)::value * 4096, HF = decltype(half_tag)::value;
            trR_issue<OFF + 8192>(f[0], tb[2 * HF][0], tb[2 * HF][1]);
            trR_issue<OFF + 8192>(f[1], tb[2 * HF + 1][0], tb[2 * HF + 1][1]);
            trR_issue<OFF>(f[2], tb[2 * HF][0], tb[2 * HF][1]);
            trR_issue<OFF>(f[3], tb[2 * HF + 1][0], tb[2 * HF + 1][1]);
        };
        auto sdp = [&](u32x4 (&r)[4], int qb, int qi) {           // 8 MFMAs: both key tiles, both k-steps; constants enter as C of ks = 0
            const f32x4 cl = __builtin_bit_cast(f32x4, c1[qb][2 * qi]), cd = __builtin_bit_cast(f32x4, c1[qb][2 * qi + 1]);
#pragma unroll
            for (int ki = 0; ki < 2; ++ki) {
                s[qb][qi][ki] = mfma16<F16>(__builtin_bit_cast(bf16x8, r[0]), kf[ki][0], cl);
                dp[qb][qi][ki] = mfma16<F16>(__builtin_bit_cast(bf16x8, r[1]), vf[ki][0], cd);
            }
#pragma unroll
            for (int ki = 0; ki < 2; ++ki) {
                s[qb][qi][ki] = mfma16<F16>(__builtin_bit_cast(bf16x8, r[2]), kf[ki][1], s[qb][qi][ki]);
                dp[qb][qi][ki] = mfma16<F16>(__builtin_bit_cast(bf16x8, r[3]), vf[ki][1], dp[qb][qi][ki]);
            }
        };
        auto valu = [&](int qb) {
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int ki = 0; ki < 2; ++ki)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = HS_BWD_EXP2(s[qb][qi][ki][r]);
                        s[qb][qi][ki][r] = p;
                        dp[qb][qi][ki][r] = p * dp[qb][qi][ki][r];
                    }
        };
        auto pack = [&](const f32x4& lo, const f32x4& hi) {       // k-slots 0..3 <- tile qi = 0, 4..7 <- tile qi = 1
            u32x4 u;
            u[0] = pack2h<F16>(lo[0], lo[1]); u[1] = pack2h<F16>(lo[2], lo[3]); u[2] = pack2h<F16>(hi[0], hi[1]); u[3] = pack2h<F16>(hi[2], hi[3]);
            return __builtin_bit_cast(bf16x8, u);
        };
        auto dvdk = [&](u32x4 (&f)[4], int qb, int half) {         // 8 MFMAs: d tiles 2half, 2half+1 x both key tiles
#pragma unroll
            for (int ki = 0; ki < 2; ++ki) {
                const bf16x8 pf = pack(s[qb][0][ki], s[qb][1][ki]), dsf = pack(dp[qb][0][ki], dp[qb][1][ki]);
#pragma unroll
                for (int d2 = 0; d2 < 2; ++d2) {
                    dv[2 * half + d2][ki] = mfma16<F16>(__builtin_bit_cast(bf16x8, f[d2]), pf, dv[2 * half + d2][ki]);
                    dk[2 * half + d2][ki] = mfma16<F16>(__builtin_bit_cast(bf16x8, f[2 + d2]), dsf, dk[2 * half + d2][ki]);
                }
            }
        };
        // never more than 15 LDS reads in flight (lgkmcnt is a 4-bit counter); counts in brackets = in flight after the line
        stats(I0{});                                                                                 // [4]
        rows(ra, I0{}, I0{});                                                                        // [8]
        lds_wait4<4>(c1[0][0], c1[0][1], c1[0][2], c1[0][3]);
        rows(rb_, I0{}, I1{});                                                                       // [8]
        lds_wait4<4>(ra[0], ra[1], ra[2], ra[3]);
        sdp(ra, 0, 0);
        stats(I1{});                                                                                 // [8]
        lds_wait4<4>(rb_[0], rb_[1], rb_[2], rb_[3]);
        sdp(rb_, 0, 1);
        rows(ra, I1{}, I0{});                                                                        // [8]
        lds_wait4<4>(c1[1][0], c1[1][1], c1[1][2], c1[1][3]);
        rows(rb_, I1{}, I1{});                                                                       // [8]
        valu(0);
        lds_wait4<4>(ra[0], ra[1], ra[2], ra[3]);
        sdp(ra, 1, 0);
        trs(ta, I0{}, I0{});                                                                         // [12]
        lds_wait4<8>(rb_[0], rb_[1], rb_[2], rb_[3]);
        sdp(rb_, 1, 1);
        lds_wait4<0>(ta[0], ta[1], ta[2], ta[3]);
        trs(tc, I0{}, I1{});                                                                         // [8]
        dvdk(ta, 0, 0);
        valu(1);
        lds_wait4<0>(tc[0], tc[1], tc[2], tc[3]);
        trs(ta, I1{}, I0{});                                                                         // [8]
        dvdk(tc, 0, 1);
        lds_wait4<0>(ta[0], ta[1], ta[2], ta[3]);
        trs(tc, I1{}, I1{});                                                                         // [8]
        dvdk(ta, 1, 0);
        lds_wait4<0>(tc[0], tc[1], tc[2], tc[3]);
        dvdk(tc, 1, 1);
    };
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, -1> IR;
    auto step = [&](int t, auto stage_tag) {
        constexpr int ST = decltype(stage_tag)::value;
        if (wave < 2) { if (NP == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
        else { if (NP == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) issue(t + 2, (ST + 2) % 3);
        if (active) tile(stage_tag, 0);
    };
    int t = 0;
    for (; t + 3 < nt; t += 3) {
        step(t, I0{});
        step(t + 1, I1{});
        step(t + 2, I2{});
    }
    int rst = 0;
    if (t + 1 < nt) { step(t, I0{}); ++t; rst = 1; }
    if (t + 1 < nt) { step(t, I1{}); ++t; rst = 2; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!active) return;
    tile(IR{}, rst);
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) {
        const int key = k0 + 16 * ki + c;
        if (key < N) {
            bf16_t* row = dqkv + ((long)b * N + key) * D3 + head * 64;
#pragma unroll
            for (int di = 0; di < 4; ++di) {
                u32x2 v;
                v[0] = pack2h<F16>(dk[di][ki][0] * dk_mul, dk[di][ki][1] * dk_mul);
                v[1] = pack2h<F16>(dk[di][ki][2] * dk_mul, dk[di][ki][3] * dk_mul);
                *(u32x2*)(row + Dm + 16 * di + 4 * g) = v;
                v[0] = pack2h<F16>(dv[di][ki][0] * dv_mul, dv[di][ki][1] * dv_mul);
                v[1] = pack2h<F16>(dv[di][ki][2] * dv_mul, dv[di][ki][3] * dv_mul);
                *(u32x2*)(row + 2 * Dm + 16 * di + 4 * g) = v;
            }
        }
    }
}

// (thin non-template kernel: hipcc's host pass silently dropped the launch stub of the templated __global__ form)
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv16_w4_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                                   const float* __restrict__ ws, bf16_t* __restrict__ dqkv, int B,
                                                                   int N, int H, float kmul, float dk_mul, float dv_mul) {
    attn_bwd_dkdv16_body<4, false>(qkv, d_o, ws, dqkv, B, N, H, kmul, dk_mul, dv_mul);
}
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv16_w4_f16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                                       const float* __restrict__ ws, bf16_t* __restrict__ dqkv, int B,
                                                                       int N, int H, float kmul, float dk_mul, float dv_mul) {
    attn_bwd_dkdv16_body<4, true>(qkv, d_o, ws, dqkv, B, N, H, kmul, dk_mul, dv_mul);
}

extern "C" int hs_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int dh, int prec, int flags, void* stream) {
    if (!qkv || !o) return HS_ERR_ARG;
    if (dh != 64 || B < 1 || N < 1 || H < 1) return HS_ERR_SHAPE;
    if (flags & ~HS_ATTN_Q_PRESCALED) return HS_ERR_ARG;
    const size_t esz = prec == HS_PREC_F32 ? 4 : 2;
    if ((size_t)N * 3 * H * 64 * esz >= HS_MAX_RECORDS) return HS_ERR_SHAPE;
    if (((uintptr_t)qkv | (uintptr_t)o) & 15) return HS_ERR_SHAPE;
    const float LOG2E = 1.44269504088896341f;
    const bool pre = (flags & HS_ATTN_Q_PRESCALED) != 0;
    const float c2 = pre ? 1.0f : 0.125f * LOG2E;          // what still multiplies q.k (then exp2); dh^-0.5 = 1/8
    dim3 grid((N + 127) / 128, H, B);
    dim3 grid1((unsigned)(((N + 127) / 128) * H * B));      // bf16 kernels decode (block, head, batch) themselves (XCD order)
    hipStream_t s = (hipStream_t)stream;
    const double nn = (double)B * H * N * N * 64;
    ProfScope prof(HS_PROF_ATTN_FWD, stream, 4.0 * nn, (double)B * N * H * 64 * 4 * esz);
    if (prec == HS_PREC_BF16) {
        if (!lse) return HS_ERR_ARG;
        if (g_hs_dbg[0] == 1) {       // diagnostic: the exact (running-max) kernel alone
            hipLaunchKernelGGL(attn_fwd_bf16_kernel<false>, grid1, dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)o, lse, N, H, c2);
        } else {
            // optimistic kernel, then the guarded exact kernel over the blocks it flagged (normally none: one load + exit)
            dim3 grid2((unsigned)(((N + 255) / 256) * H * B));
            hipLaunchKernelGGL(attn_fwd_fast_kernel, grid2, dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)o, lse, N, H, c2);
            if (HS_DBG(g_hs_dbg[0]) != 2)         // (2 = timing-only: no fix-up pass; hook builds only)
                hipLaunchKernelGGL(attn_fwd_bf16_kernel<true>, grid1, dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)o, lse, N, H, c2);
        }
    } else if (prec == HS_PREC_F32) {
        hipLaunchKernelGGL(attn_fwd_f32_kernel, grid, dim3(256), 0, s, (const float*)qkv, (float*)o, lse, N, H, c2);
    } else if (prec == HS_PREC_F16) {
        // fp16 operands: the optimistic kernel with fp16's narrower acceptance window (row sums in [1, 2^15): no P overflowed, the largest P
        // is a normal number), then the running-maximum kernel -- P = exp2(s - max) <= 1, what the reference's softmax -> .half() produces --
        // over exactly the 128-query blocks that hold a rejected row (normally none: one load and exit)
        if (!lse) return HS_ERR_ARG;
        if (g_hs_dbg[0] == 1) {       // diagnostic: the exact (running-max) kernel alone
            hipLaunchKernelGGL((attn_fwd_bf16_kernel<false, true>), grid1, dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)o, lse, N, H, c2);
        } else {
            dim3 grid2((unsigned)(((N + 255) / 256) * H * B));
            hipLaunchKernelGGL(attn_fwd_fast_f16_kernel, grid2, dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)o, lse, N, H, c2);
            hipLaunchKernelGGL((attn_fwd_bf16_kernel<true, true>), grid1, dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)o, lse, N, H, c2);
        }
    } else {
        return HS_ERR_ARG;
    }
    HS_CHECK_LAUNCH();
    return HS_OK;
}

extern "C" int hs_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, float* delta_ws,
                           int B, int N, int H, int dh, int flags, void* stream) {
    if (!qkv || !o || !d_o || !lse || !dqkv || !delta_ws) return HS_ERR_ARG;
    if (dh != 64 || B < 1 || N < 1 || H < 1) return HS_ERR_SHAPE;
    if (flags & ~(HS_ATTN_Q_PRESCALED | HS_ATTN_F32 | HS_ATTN_F16)) return HS_ERR_ARG;
    if ((flags & HS_ATTN_F32) && (flags & HS_ATTN_F16)) return HS_ERR_ARG;
    if (((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dqkv) & 15) return HS_ERR_SHAPE;
    const float scale = 0.125f, LOG2E = 1.44269504088896341f;
    if (flags & HS_ATTN_F32) {           // parity mode: every tensor fp32, raw (unscaled) q
        if (flags & HS_ATTN_Q_PRESCALED) return HS_ERR_ARG;
        if ((size_t)N * 3 * H * 64 * 4 >= HS_MAX_RECORDS) return HS_ERR_SHAPE;
        hipStream_t s = (hipStream_t)stream;
        const long tot = (long)B * N * H * 16;
        hipLaunchKernelGGL(attn_stats_f32_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, (const float*)o, (const float*)d_o,
                           lse, delta_ws, B, N, H);
        dim3 g((N + 127) / 128, H, B);
        hipLaunchKernelGGL(attn_bwd_f32_kernel<1>, g, dim3(256), 0, s, (const float*)qkv, (const float*)d_o, delta_ws, (float*)dqkv, B, N, H,
                           scale * LOG2E, scale);
        hipLaunchKernelGGL(attn_bwd_f32_kernel<0>, g, dim3(256), 0, s, (const float*)qkv, (const float*)d_o, delta_ws, (float*)dqkv, B, N, H,
                           scale * LOG2E, scale);
        HS_CHECK_LAUNCH();
        return HS_OK;
    }
    if ((size_t)N * 3 * H * 64 * 2 >= HS_MAX_RECORDS) return HS_ERR_SHAPE;
    const bool pre = (flags & HS_ATTN_Q_PRESCALED) != 0, h16 = (flags & HS_ATTN_F16) != 0;
    const float c2 = pre ? 1.0f : scale * LOG2E;
    const float dk_mul = pre ? 1.0f / LOG2E : scale;        // dK = dh^-0.5 dS^T q = dS^T q' / log2(e)
    const float unshift = h16 ? 1.0f / (float)(1 << HS_F16_PSHIFT) : 1.0f;     // (fp16: the kernels carry P and dS times 2^HS_F16_PSHIFT)
    hipStream_t s = (hipStream_t)stream;
    // FLOP credit follows SURVEY.md 8(d): the backward's ALGORITHMIC work is 2.0 x the forward's 4*nn (dV, dP, dQ, dK: four
    // products); the recomputation of S (both kernels) and of dP (dQ kernel) is executed but NOT counted.  dK/dV kernel: dV, dP, dK
    // = 6*nn (executes 8*nn); dQ kernel: dQ = 2*nn (executes 6*nn).
    const double nn = (double)B * H * N * N * 64;
    const double io = (double)B * N * H * 64 * 2;
    float* nstat = delta_ws;                                  // [0] = -delta, [1] = -lse*log2e   (2 x [B][H][N])
    dim3 grid((unsigned)(((N + 127) / 128) * H * B));
    {   // the dQ sweep first: it also writes the row constants the dK/dV sweep preloads (no separate statistics pass)
        ProfScope prof(HS_PROF_ATTN_DQ, stream, 2.0 * nn, 6 * io);
        hipLaunchKernelGGL(h16 ? attn_bwd_dq2_w4_f16_kernel : attn_bwd_dq2_w4_kernel, grid, dim3(256), 0, s, (const bf16_t*)qkv,
                           (const bf16_t*)d_o, (const bf16_t*)o, lse, nstat, (bf16_t*)dqkv, B, N, H, c2, scale * unshift);
    }
    {
        ProfScope prof(HS_PROF_ATTN_DKDV, stream, 6.0 * nn, 6 * io);
        hipLaunchKernelGGL(h16 ? attn_bwd_dkdv16_w4_f16_kernel : attn_bwd_dkdv16_w4_kernel, grid, dim3(256), 0, s, (const bf16_t*)qkv,
                           (const bf16_t*)d_o, nstat, (bf16_t*)dqkv, B, N, H, c2, dk_mul * unshift, unshift);
    }
    HS_CHECK_LAUNCH();
    return HS_OK;
}
