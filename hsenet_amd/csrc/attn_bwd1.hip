// K4 backward, ONE sweep (round 4): dQ, dK, dV of the MONAI SABlock core (reference hot loops multimodal_encoder/vit.py:352-354,
// 464-466) with S and dP computed ONCE per (query, key) tile: five MFMA products per tile (S = Q K^T, dP = dO V^T, dV^T += dO^T P,
// dK^T += Q^T dS, dQ^T += K^T dS^T) = 10 N^2 d executed, against the 14 N^2 d of the two-sweep kernels of attn.hip (which stay the
// generic path: any N, any batch).  Structure (cdna guide, Appendix B "Attention backward"):
//   * a workgroup = 4 waves = 256 keys of one (batch, head); a wave keeps K, V (MFMA operands) and the dK^T, dV^T accumulators of
//     its 64 keys in registers (one wave per SIMD, the whole 512-register file) and sweeps the queries in 32-row slices;
//   * S and dP are computed with the key on the lane (row constants -lse*log2e / -delta preloaded as the accumulators), so P and dS
//     are directly the B operands of the dV^T / dK^T products; dS crosses LDS once ([key][query] image, read back transposed) for the
//     dQ product, where each wave owns 16 of the 64 d columns and sums over all 256 keys of the workgroup (16x16x32 MFMAs):
//     no partial sums inside the workgroup;
//   * the dQ slice of a (batch, head) is summed over its key blocks IN A FIXED ORDER by an ordered hand-off (bitwise reproducible,
//     no float atomics: the chip adds 1.3 TB/s of them): the kernel is persistent, the 256 workgroups form
//     gangs of M = N / 256 members that walk the (batch, head) pairs together; member m adds its tile to member m-1's running sum
//     (fp32, `sc1` write-through stores and loads, a per-member progress counter: MI355X_MICROARCH.md "Workgroup dispatch ..."),
//     the last member scales, rounds and writes dq.  A member only ever waits for a member with a LOWER start ticket, so the
//     protocol cannot deadlock whatever the dispatch order; every spin is bounded and raises an error word instead of hanging.
//   * N = 256 M + rem with rem <= 1: the remainder key (the 4097th token of a 256^3 volume) is handled in fp32 by the small
//     pre-pass that also produces the row constants (attn_bwd_pre_kernel); its rank-1 dQ term seeds the chain.
// Software pipeline (one barrier per 32-query slice): iteration i runs the S / dP MFMAs of slice i+1 beside the exp2 / multiply /
// pack of slice i, then the dV / dK MFMAs of slice i and the dQ MFMAs of slice i-1.
#include "common.h"
#include "attn_util.h"
#include <type_traits>

namespace {
constexpr int FSTG = 16384 + 1024;       // ring stage: Q' image 8 KiB | dO image 8 KiB | 4 x 64 fp32 row constants (64-query tile)
constexpr int FRING = 3 * FSTG;
constexpr int FDS = 16384;               // dS^T image of one 32-query slice: [256 keys][32 q] bf16 (64-B rows)
constexpr int FKIMG = FRING + 2 * FDS;   // K image [256 keys][64 d] of the workgroup's key block (the dQ product's A operand, read transposed)
constexpr int FSMEM = FKIMG + 32768;
constexpr int FCTR = 16;                 // ctr[0] ticket, ctr[1] error word, ctr[FCTR + bh * M + m] progress (wave-slices published)
constexpr unsigned FSPIN = 1u << 18;     // bounded spin (polls) before a member gives up and raises the error word

struct FusedArgs {
    const bf16_t* qkv; const bf16_t* d_o; const float* stats; bf16_t* dqkv; float* chain; unsigned* ctr;
    int B, N, H, M, nslice, ngang, rem, mode;
    float kmul, dk_mul, dq_mul;
    int dbg;                             // HS_DEBUG_HOOKS builds only: bit 0 drain vmcnt at every wait, bit 1 agent acquire before the predecessor loads
    unsigned long long* stamps;          // HS_DEBUG_HOOKS builds only: s_memtime stamps of wave 0 of the first workgroup (else NULL)
};
enum { F_PARTIAL = 0, F_CHAIN = 1 };

template <int IMM>
__device__ __forceinline__ void ldsw64(unsigned base, unsigned lo, unsigned hi) {
    u32x2 v = {lo, hi};
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(base), "v"(v), "i"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ void tr2_issue(u32x4& f, unsigned base) {       // rows r..r+3 and r+4..r+7 of a 64-B-row image
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(lo), "=&v"(hi) : "v"(base), "i"(IMM), "i"(IMM + 256) : "memory");
    f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}
__device__ __forceinline__ void gload_sc1(u32x4& d, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void gstore_sc1(void* p, u32x4 v) {      // (the s_nop: the next instruction must not overwrite the data registers early)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void gstore_plain(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// MFMAs issued from asm so that the register FILE of every operand is ours: the dK^T / dV^T accumulators and the K / V / K^T operands
// live in the accumulator half (256 AGPRs), S' / dP' / dQ^T tiles in arch VGPRs where the VALU reads them.  Left to hipcc, every MFMA
// of a 512-register kernel gets an AGPR destination and each S' / dP' tile is copied out (1400 v_accvgpr moves in this kernel).
// hipcc pads no hazards around asm (cdna guide 5.7): every consumer of an asm MFMA result below sits >= 8 MFMAs behind it.
__device__ __forceinline__ void mfma32_first(f32x16& d, const u32x4& a, const bf16x8& b, const f32x16& c) {     // D(v) = A(v) B(a) + C(v)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
}
__device__ __forceinline__ void mfma32_next(f32x16& d, const u32x4& a, const bf16x8& b) {                        // D(v) += A(v) B(a)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma32_acc(f32x16& acc, const u32x4& a, const u32x4& b) {                        // ACC(a) += A(v) B(v)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_acc2(f32x4& acc0, f32x4& acc1, const u32x4& a, const u32x4& b0, const u32x4& b1) {   // ACCn(v) += A(v) Bn(v)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %4, %1" : "+v"(acc0), "+v"(acc1) : "v"(a), "v"(b0), "v"(b1));
}
template <int IMM>
__device__ __forceinline__ void trK_issue(u32x4& f, unsigned base) {       // keys k..k+3 and k+4..k+7 of the 128-B-row K image
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(lo), "=&v"(hi) : "v"(base), "i"(IMM), "i"(IMM + 512) : "memory");
    f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}
__device__ __forceinline__ void gload1_sc1(unsigned& d, const void* p) {
    asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(d) : "v"(p) : "memory");
}
template <int CNT>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(CNT) : "memory"); }
template <int CNT>
__device__ __forceinline__ void vm_wait2(u32x4& a, u32x4& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "i"(CNT) : "memory"); }
template <int CNT>
__device__ __forceinline__ void vm_wait1(unsigned& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "i"(CNT) : "memory"); }
__device__ __forceinline__ void lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ f32x16 cat16(const u32x4& a, const u32x4& b, const u32x4& c, const u32x4& d) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r[i] = __uint_as_float(a[i]); r[4 + i] = __uint_as_float(b[i]); r[8 + i] = __uint_as_float(c[i]); r[12 + i] = __uint_as_float(d[i]);
    }
    return r;
}
}  // namespace

// ------------------------------------------------------------------------------------------------------
// pre-pass: row constants (-delta = -sum_d dO*O, -lse*log2e) for every query row, and — when N = 256 M + 1 — the whole
// contribution of the remainder key r = N - 1 in fp32: dsr[q] = P[q][r] (dP[q][r] - delta[q]) (what the chain's first member
// multiplies with K[r] to seed dQ), dV[r] = sum_q P dO[q], dK[r] = dk_mul sum_q dS q'[q].  One block per (batch, head), 8 lanes per
// row (16 B each), fixed-order reductions.  Also zeroes the hand-off counters of its (batch, head) and the ticket / error words.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_pre_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                           const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                           float* __restrict__ stats, bf16_t* __restrict__ dqkv, unsigned* __restrict__ ctr,
                                                           int B, int N, int H, int M, int rem, float c2, float dk_mul) {
    __shared__ float red[4][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 7, rr = lane >> 3;
    const int bh = blockIdx.x, b = bh / H, head = bh % H;
    const int Dm = H * 64, D3 = 3 * Dm;
    const long BH = (long)B * H;
    if (tid < M) ctr[FCTR + bh * M + tid] = 0u;
    if (bh == 0 && tid < FCTR) ctr[tid] = 0u;
    float kr[8], vr[8], dkr[8], dvr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { kr[j] = 0.f; vr[j] = 0.f; dkr[j] = 0.f; dvr[j] = 0.f; }
    if (rem) {
        const bf16_t* kp = qkv + ((long)b * N + (N - 1)) * D3 + Dm + head * 64 + sub * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { kr[j] = bf2f(kp[j]); vr[j] = bf2f(kp[Dm + j]); }
    }
    const float* lrow = lse + bh * (long)N;
    float* s_nl = stats + bh * (long)N;
    float* s_nd = stats + (BH + bh) * (long)N;
    float* s_ds = stats + (2 * BH + bh) * (long)N;
#pragma unroll 2
    for (int row0 = wave * 8; row0 < N; row0 += 32) {
        const int row = row0 + rr;
        const bool valid = row < N;
        const int rc = valid ? row : N - 1;
        const u32x4 qv = *(const u32x4*)(qkv + ((long)b * N + rc) * D3 + head * 64 + sub * 8);
        const u32x4 gv = *(const u32x4*)(d_o + ((long)b * N + rc) * Dm + head * 64 + sub * 8);
        const u32x4 ov = *(const u32x4*)(o + ((long)b * N + rc) * Dm + head * 64 + sub * 8);
        float qf[8], gf[8];
        float dsum = 0.f, sq = 0.f, sv = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qf[2 * i] = __uint_as_float(qv[i] << 16); qf[2 * i + 1] = __uint_as_float(qv[i] & 0xffff0000u);
            gf[2 * i] = __uint_as_float(gv[i] << 16); gf[2 * i + 1] = __uint_as_float(gv[i] & 0xffff0000u);
            dsum += gf[2 * i] * __uint_as_float(ov[i] << 16);
            dsum += gf[2 * i + 1] * __uint_as_float(ov[i] & 0xffff0000u);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { sq += qf[j] * kr[j]; sv += gf[j] * vr[j]; }
#pragma unroll
        for (int o2 = 1; o2 < 8; o2 <<= 1) {
            dsum += __shfl_xor(dsum, o2, 64); sq += __shfl_xor(sq, o2, 64); sv += __shfl_xor(sv, o2, 64);
        }
        const float nl = -lrow[rc] * 1.44269504088896341f, nd = -dsum;
        float p = 0.f, ds = 0.f;
        if (rem && valid) {
            p = __builtin_amdgcn_exp2f(sq * c2 + nl);
            ds = p * (sv + nd);
#pragma unroll
            for (int j = 0; j < 8; ++j) { dvr[j] += p * gf[j]; dkr[j] += ds * qf[j]; }
        }
        if (valid && sub == 0) { s_nl[row] = nl; s_nd[row] = nd; s_ds[row] = ds; }
    }
    if (!rem) return;
    // lanes with equal `sub` hold partial sums over different rows: reduce over rr (lane bits 3..5), then over the 4 waves
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int o2 = 8; o2 < 64; o2 <<= 1) { dkr[j] += __shfl_xor(dkr[j], o2, 64); dvr[j] += __shfl_xor(dvr[j], o2, 64); }
    }
    if (rr == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[wave][0][sub * 8 + j] = dkr[j]; red[wave][1][sub * 8 + j] = dvr[j]; }
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, d = tid & 63;
        const float v = ((red[0][which][d] + red[1][which][d]) + red[2][which][d]) + red[3][which][d];
        bf16_t* row = dqkv + ((long)b * N + (N - 1)) * D3 + head * 64;
        row[(which ? 2 : 1) * Dm + d] = f2bf(which ? v : v * dk_mul);
    }
}


// ------------------------------------------------------------------------------------------------------
// the sweep
// ------------------------------------------------------------------------------------------------------
// Vector-memory accounting (vmcnt counts loads, stores, atomics and LDS-DMA together, in issue order).  In iteration j a wave issues, in
// this order:  F_j  1 flag add (ordered hand-off only, j >= 2)          L_j  1 poll load (wave 0 of a member with a predecessor)
//              D_j  5 LDS-DMA pieces (j even: the 64-query tile j/2 + 2) S_j  2 stores (the finalised slice j-1; j >= 1)
//              P_j  2 loads of the predecessor's running sum of slice j (members with a predecessor; j < nslice)
// At the top of iteration i everything older than {D, S, P} of iteration i-1 is waited for: that covers the poll L_{i-1}, the stores
// S_{i-2} (so slice i-3 may be flagged) and every DMA piece issued before iteration i-1 (a tile is read 3 iterations after its issue).
__global__ __launch_bounds__(256, 1) void attn_bwd_fused_kernel(const FusedArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[FSMEM + 16];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31, g = lane >> 4, c16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, H = a.H, M = a.M, Dm = H * 64, D3 = 3 * Dm;
    const long BH = (long)a.B * H;
    // logical id = start order (a member only ever waits for members that started before it)
    if (tid == 0) *(volatile unsigned*)(smem + FSMEM) = atomicAdd(a.ctr, 1u);
    __syncthreads();
    const unsigned ticket = __builtin_amdgcn_readfirstlane(*(volatile unsigned*)(smem + FSMEM));
    const int gx = ticket & 7, gy = ticket >> 3;
    const int gang = gx + 8 * (gy / M), m = gy % M;     // a gang's members share ticket % 8: one XCD under round-robin dealing (speed only)
    if (gang >= a.ngang) return;
    const bool chain = a.mode == F_CHAIN, has_pred = chain && m > 0, is_last = chain && m == M - 1;
    const int nslice = a.nslice, nt = (nslice + 1) >> 1;
    bool dead = false;                                   // a bounded spin ran out: stop waiting (results void, error word set)

    const unsigned sm = lds_addr(smem);
    unsigned rb0[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rb0[ks] = sm + offR(c, 2 * ks + h);
    unsigned tb0[2][2];
    {
        const int p = lane & 3, q4 = (lane >> 2) & 3;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int col = db * 32 + 16 * (g & 1) + 4 * p;
            tb0[db][0] = sm + offR(4 * h + q4, col >> 3) + 8 * (p & 1);
            tb0[db][1] = sm + offR(4 * h + q4 + 8, col >> 3) + 8 * (p & 1);
        }
    }
    const unsigned cb0 = sm + 16384 + 16 * h;
    const int kl = 64 * wave + c;
    unsigned wb0[2], db0[2];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) wb0[cc] = sm + FRING + kl * 64 + ((cc ^ ((kl >> 3) & 1)) << 5) + 8 * h;
    {
        const int p = lane & 3, qq = (lane >> 2) & 3;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) db0[qt] = sm + FRING + (8 * g + qq) * 64 + ((qt ^ (g & 1)) << 5) + 8 * p;
    }
    const unsigned kb0 = sm + FKIMG + (8 * g + ((lane >> 2) & 3)) * 128 + 32 * wave + 8 * (lane & 3);   // K^T fragment: d = 16 wave + (lane & 15), keys 32 ks8 + 8 g + j
    const unsigned srb0 = sm + 16384 + 512 + 4 * c16;     // dsr (row-constant array 2) of query 16 qt + (lane & 15)

    for (long bh = gang; bh < BH; bh += a.ngang) {
        const int b = (int)(bh / H), head = (int)(bh - (long)b * H);
        const int key0 = 256 * m;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.qkv + (long)b * N * D3, (size_t)N * D3 * 2);
        const __amdgpu_buffer_rsrc_t rsd = make_rsrc(a.d_o + (long)b * N * Dm, (size_t)N * Dm * 2);
        const int cw = wave < 3 ? wave : 2;
        const __amdgpu_buffer_rsrc_t rsc = make_rsrc(a.stats + ((long)cw * BH + bh) * N, (size_t)N * 4);
        const __amdgpu_buffer_rsrc_t rso = make_rsrc(a.dqkv + (long)b * N * D3, (size_t)N * D3 * 2);
        unsigned vq[2], vd[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (wave * 2 + j) * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ swzR(row);
            vq[j] = (unsigned)(((long)row * D3 + head * 64 + ch * 8) * 2);
            vd[j] = (unsigned)(((long)row * Dm + head * 64 + ch * 8) * 2);
        }
        auto issue = [&](int t, int st) {                 // 5 LDS-DMA pieces per wave: 2 of Q', 2 of dO, 1 of row constants
            const bool last = t >= nt - 1;                // (tiles past the end are issued too, as zero fills: the op count per iteration is fixed)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = (wave * 2 + j) * 8 + (lane >> 3);
                const bool in = !last || t * 64 + row < N;
                char* dst = smem + st * FSTG + (wave * 2 + j) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))dst, 16, in ? vq[j] : HS_OOB_OFF, (unsigned)t * (unsigned)(64 * D3 * 2), 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (LDS_PTR(void))(dst + 8192), 16, in ? vd[j] : HS_OOB_OFF, (unsigned)t * (unsigned)(64 * Dm * 2), 0, 0);
            }
            const bool inc = !last || t * 64 + lane < N;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsc, (LDS_PTR(void))(smem + st * FSTG + 16384 + wave * 256), 4, inc ? (unsigned)lane * 4u : HS_OOB_OFF,
                                                     (unsigned)t * 256u, 0, 0);
        };
        issue(0, 0);
        issue(1, 1);
        // K, V of this wave's 64 keys as B operands of S / dP (key on the lane); K staged once for the transposed dQ operand
        bf16x8 kf[2][4], vf[2][4];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const unsigned off = (unsigned)(((long)(key0 + 64 * wave + 32 * kh + c) * D3 + Dm + head * 64 + ks * 16 + 8 * h) * 2);
                kf[kh][ks] = scale_frag(buf_load16(rs, off), a.kmul);
                vf[kh][ks] = __builtin_bit_cast(bf16x8, buf_load16(rs, off + Dm * 2));
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = (wave * 8 + j) * 8 + (lane >> 3);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))(smem + FKIMG + (wave * 8 + j) * 1024), 16,
                                                     (unsigned)(((long)(key0 + row) * D3 + Dm + head * 64 + (lane & 7) * 8) * 2), 0, 0, 0);
        }
        float krem[4] = {0.f, 0.f, 0.f, 0.f};
        const bool seed = a.rem && chain && m == 0;
        if (seed) {
            const bf16_t* kp = a.qkv + ((long)b * N + (N - 1)) * D3 + Dm + head * 64 + 16 * wave + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) krem[r] = bf2f(kp[r]);
        }
        vm_wait<0>();
        __builtin_amdgcn_s_barrier();
        f32x16 dk[2][2], dv[2][2];                             // [key half][d half]: dK^T, dV^T
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                dk[kh][db] = zero16(); dv[kh][db] = zero16();
                asm volatile("" : "+a"(dk[kh][db]), "+a"(dv[kh][db]));       // accumulator-file residents from here on
            }
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+a"(kf[kh][ks]), "+a"(vf[kh][ks]));

        const unsigned* pred_ctr = a.ctr + FCTR + bh * M + (m > 0 ? m - 1 : 0);
        unsigned* my_ctr = a.ctr + FCTR + bh * M + m;
        float* cbase = a.chain + (chain ? bh * nslice : (bh * M + m) * nslice) * 2048 + (wave * 2) * 256 + lane * 4;

        f32x16 sS[2], sD[2];                                   // S', dP' [key half] of the slice whose softmax runs next
        u32x4 pred[2];                                         // the predecessor's running sum of the slice finalised next
        pred[0] = u32x4{0u, 0u, 0u, 0u}; pred[1] = pred[0];
        unsigned pollv = 0u;
        u32x4 cl4[4], cd4[4], qf[4], df[4];                    // phase A operands of the slice being prefetched
        u32x4 pw[2][2], dw[2][2];                              // P, dS of the current slice as bf16 B operands [key half][query k-step]

        // ---- phase A of slice sl (parity P): S' = Q' K^T - lse*log2e, dP' = dO V^T - delta ----
        auto a_reads1 = [&](int sl, auto par_tag) {
            constexpr int P = decltype(par_tag)::value;
            const unsigned so = (unsigned)(((sl >> 1) % 3) * FSTG + P * 4096);
            const unsigned cb = cb0 + (unsigned)(((sl >> 1) % 3) * FSTG + P * 128);
#pragma unroll
            for (int j = 0; j < 4; ++j) ldsr128<0>(cl4[j], cb + 32 * j);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ldsr128<0>(qf[ks], rb0[ks] + so);
        };
        auto a_reads2 = [&](int sl, auto par_tag) {
            constexpr int P = decltype(par_tag)::value;
            const unsigned so = (unsigned)(((sl >> 1) % 3) * FSTG + P * 4096);
            const unsigned cb = cb0 + (unsigned)(((sl >> 1) % 3) * FSTG + P * 128);
#pragma unroll
            for (int j = 0; j < 4; ++j) ldsr128<256>(cd4[j], cb + 32 * j);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ldsr128<8192>(df[ks], rb0[ks] + so);
        };
#define HS_VSTEP(KH, R)                                                                 \
        { const float p_ = __builtin_amdgcn_exp2f(sS[KH][R]); sS[KH][R] = p_; sD[KH][R] = p_ * sD[KH][R]; }
#define HS_VPACK(KH, J)                                                                 \
        { pw[KH][(J) >> 2][(J) & 3] = pack2bf(sS[KH][2 * (J)], sS[KH][2 * (J) + 1]); dw[KH][(J) >> 2][(J) & 3] = pack2bf(sD[KH][2 * (J)], sD[KH][2 * (J) + 1]); }
#define HS_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef HS_DEBUG_HOOKS
        unsigned long long stv[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bool stamp_on = a.stamps != nullptr && ticket == (unsigned)a.M * 8u - 8u + 0u * 0u && wave == 0 && bh == gang;     // the last member of gang 0
#define HS_STAMP(K, I) if (stamp_on && (I) >= 40 && (I) < 48) asm volatile("s_memtime %0" : "=s"(stv[K]));
#define HS_STAMP_OUT(I)                                                                                  \
        if (stamp_on && (I) >= 40 && (I) < 48) {                                                         \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
            if (lane == 0) { for (int k_ = 0; k_ < 12; ++k_) a.stamps[((I) - 40) * 12 + k_] = stv[k_]; }   \
        }
#else
#define HS_STAMP(K, I)
#define HS_STAMP_OUT(I)
#endif

        // ---- one pipeline iteration (PAR = i & 1): C(i-1) beside V(i) half 0 | A(i+1) beside V(i) half 1 | W(i), B(i) | hand-off of slice i-1 ----
        auto iter = [&](int i, auto par_tag, auto ha_tag, auto hv_tag, auto hc_tag) {
            constexpr int PAR = decltype(par_tag)::value;
            constexpr bool HA = decltype(ha_tag)::value, HV = decltype(hv_tag)::value, HC = decltype(hc_tag)::value;
            typedef std::integral_constant<int, PAR ^ 1> NPAR;
            // top: wait (see the accounting above), publish, poll, barrier
            HS_STAMP(0, i)
#ifdef HS_DEBUG_HOOKS
            if (a.dbg & 1) vm_wait1<0>(pollv);
#endif
            if (has_pred) { if (i == 1) vm_wait1<7>(pollv); else if (PAR == 1) vm_wait1<9>(pollv); else vm_wait1<4>(pollv); }
            else { if (PAR == 1) vm_wait<7>(); else vm_wait<2>(); }
            if (chain && i >= 3 && lane == 0) __hip_atomic_fetch_add(my_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // F_i: slice i-3
            if (has_pred && HV && wave == 0) {                 // slice i of the predecessor must be published before P_i is issued
                const unsigned need = 4u * (unsigned)(i + 1);
                unsigned v = __builtin_amdgcn_readfirstlane(pollv);
                unsigned spins = 0;
                while (v < need && !dead) {
                    unsigned t;
                    gload1_sc1(t, pred_ctr);
                    vm_wait1<0>(t);
                    v = __builtin_amdgcn_readfirstlane(t);
                    if (++spins > FSPIN) { dead = true; if (lane == 0) a.ctr[1] = 1u; }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            HS_STAMP(1, i)
            __builtin_amdgcn_s_barrier();
            HS_STAMP(2, i)
            if (has_pred && HV && wave == 0) gload1_sc1(pollv, pred_ctr);     // L_i: looked at on the next iteration
            unsigned sr0 = 0u, sr1 = 0u;
            if (HC && seed) {                                                 // dsr of slice i-1, read BEFORE D_i may refill that tile's stage
                const unsigned sa = srb0 + (unsigned)((((i - 1) >> 1) % 3) * FSTG + (PAR ^ 1) * 128);
                asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=&v"(sr0), "=&v"(sr1) : "v"(sa) : "memory");
            }
            if (PAR == 0) issue((i >> 1) + 2, ((i >> 1) + 2) % 3);            // D_i
            HS_SB();
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
            // ---- step 1: C(i-1) = dQ^T[16 d of this wave][32 q] over the workgroup's 256 keys (16 MFMAs of 16x16x32), V(i) key half 0 ----
            if (HC) {
                constexpr unsigned RO = (PAR ^ 1) * FDS;
                u32x4 gA[3], gB[3];                            // {dS^T q-tile 0, dS^T q-tile 1, K^T} fragments of one 32-key step, two steps in flight
#define HS_CREAD(G, KS8)                                                                                                       \
                tr2_issue<RO + (KS8) * 2048>(G[0], db0[0]); tr2_issue<RO + (KS8) * 2048>(G[1], db0[1]);                        \
                trK_issue<(KS8) * 4096>(G[2], kb0);
#define HS_CMMA(G, V0, V1, P0)                                                                                                 \
                mfma16_acc2(acc0, acc1, G[2], G[0], G[1]); if (HV) { HS_VSTEP(0, V0) HS_VSTEP(0, V1) HS_VPACK(0, P0) } HS_SB();
#define HS_CWAIT(G, CNT) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(G[0]), "+v"(G[1]), "+v"(G[2]) : "i"(CNT) : "memory");
                HS_CREAD(gA, 0)
                HS_CREAD(gB, 1)
                HS_CWAIT(gA, 6) HS_CMMA(gA, 0, 1, 0) HS_CREAD(gA, 2)
                HS_CWAIT(gB, 6) HS_CMMA(gB, 2, 3, 1) HS_CREAD(gB, 3)
                HS_CWAIT(gA, 6) HS_CMMA(gA, 4, 5, 2) HS_CREAD(gA, 4)
                HS_CWAIT(gB, 6) HS_CMMA(gB, 6, 7, 3) HS_CREAD(gB, 5)
                HS_CWAIT(gA, 6) HS_CMMA(gA, 8, 9, 4) HS_CREAD(gA, 6)
                HS_CWAIT(gB, 6) HS_CMMA(gB, 10, 11, 5) HS_CREAD(gB, 7)
                HS_CWAIT(gA, 6) HS_CMMA(gA, 12, 13, 6)
                HS_CWAIT(gB, 0)
                if (HA) a_reads1(i + 1, NPAR{});
                HS_SB();
                HS_CMMA(gB, 14, 15, 7)
#undef HS_CREAD
#undef HS_CMMA
#undef HS_CWAIT
            } else {
                if (HA) a_reads1(i + 1, NPAR{});
                if (HV) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) HS_VSTEP(0, r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) HS_VPACK(0, j)
                }
                HS_SB();
            }
            // ---- step 2: A(i+1) key half 0 (8 MFMAs) beside V(i) key half 1 ----
            HS_STAMP(3, i)
            if (HA) {
                lds_wait4<0>(cl4[0], cl4[1], cl4[2], cl4[3]);
                lds_wait4<0>(qf[0], qf[1], qf[2], qf[3]);
                a_reads2(i + 1, NPAR{});
                HS_SB();
                const f32x16 cl = cat16(cl4[0], cl4[1], cl4[2], cl4[3]);
                mfma32_first(sS[0], qf[0], kf[0][0], cl); if (HV) { HS_VSTEP(1, 0) HS_VSTEP(1, 1) HS_VPACK(1, 0) } HS_SB();
                mfma32_next(sS[0], qf[1], kf[0][1]);      if (HV) { HS_VSTEP(1, 2) HS_VSTEP(1, 3) HS_VPACK(1, 1) } HS_SB();
                mfma32_next(sS[0], qf[2], kf[0][2]);      if (HV) { HS_VSTEP(1, 4) HS_VSTEP(1, 5) HS_VPACK(1, 2) } HS_SB();
                mfma32_next(sS[0], qf[3], kf[0][3]);      if (HV) { HS_VSTEP(1, 6) HS_VSTEP(1, 7) HS_VPACK(1, 3) } HS_SB();
                lds_wait4<0>(cd4[0], cd4[1], cd4[2], cd4[3]);
                lds_wait4<0>(df[0], df[1], df[2], df[3]);
                const f32x16 cd = cat16(cd4[0], cd4[1], cd4[2], cd4[3]);
                mfma32_first(sD[0], df[0], vf[0][0], cd); if (HV) { HS_VSTEP(1, 8) HS_VSTEP(1, 9) HS_VPACK(1, 4) } HS_SB();
                mfma32_next(sD[0], df[1], vf[0][1]);      if (HV) { HS_VSTEP(1, 10) HS_VSTEP(1, 11) HS_VPACK(1, 5) } HS_SB();
                mfma32_next(sD[0], df[2], vf[0][2]);      if (HV) { HS_VSTEP(1, 12) HS_VSTEP(1, 13) HS_VPACK(1, 6) } HS_SB();
                mfma32_next(sD[0], df[3], vf[0][3]);      if (HV) { HS_VSTEP(1, 14) HS_VSTEP(1, 15) HS_VPACK(1, 7) } HS_SB();
                // ---- step 3: A(i+1) key half 1 (8 MFMAs) beside W(i) and the first transposed reads of B(i) ----
                HS_STAMP(4, i)
                mfma32_first(sS[1], qf[0], kf[1][0], cl); HS_SB();
                mfma32_next(sS[1], qf[1], kf[1][1]); HS_SB();
                mfma32_next(sS[1], qf[2], kf[1][2]); HS_SB();
                mfma32_next(sS[1], qf[3], kf[1][3]); HS_SB();
                mfma32_first(sD[1], df[0], vf[1][0], cd); HS_SB();
                mfma32_next(sD[1], df[1], vf[1][1]); HS_SB();
                mfma32_next(sD[1], df[2], vf[1][2]); HS_SB();
                mfma32_next(sD[1], df[3], vf[1][3]); HS_SB();
            } else if (HV) {
#pragma unroll
                for (int r = 0; r < 16; ++r) HS_VSTEP(1, r)
#pragma unroll
                for (int j = 0; j < 8; ++j) HS_VPACK(1, j)
                HS_SB();
            }
            HS_STAMP(5, i)
            if (HV) {
                // ---- W(i): dS^T image ----
                constexpr unsigned WO = PAR * FDS;
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    ldsw64<WO>(wb0[sp], dw[0][sp][0], dw[0][sp][1]); ldsw64<WO + 16>(wb0[sp], dw[0][sp][2], dw[0][sp][3]);
                    ldsw64<WO + 2048>(wb0[sp], dw[1][sp][0], dw[1][sp][1]); ldsw64<WO + 2048 + 16>(wb0[sp], dw[1][sp][2], dw[1][sp][3]);
                }
                // ---- step 4: B(i): dV^T += dO^T P, dK^T += Q'^T dS (16 MFMAs) ----
                const unsigned so = (unsigned)(((i >> 1) % 3) * FSTG + PAR * 4096);
                const unsigned t00 = tb0[0][0] + so, t01 = tb0[0][1] + so, t10 = tb0[1][0] + so, t11 = tb0[1][1] + so;
                u32x4 ta[4], tc[4];                            // {dO^T db0, dO^T db1, Q^T db0, Q^T db1} of query k-steps 0 / 1
                trR_issue<8192>(ta[0], t00, t01);
                trR_issue<8192>(ta[1], t10, t11);
                trR_issue<0>(ta[2], t00, t01);
                HS_STAMP(6, i)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[0]), "+v"(ta[1]), "+v"(ta[2])::"memory");
                HS_STAMP(7, i)
                trR_issue<0>(ta[3], t10, t11);
                trR_issue<8192 + 2048>(tc[0], t00, t01);
                trR_issue<8192 + 2048>(tc[1], t10, t11);
                trR_issue<2048>(tc[2], t00, t01);
                trR_issue<2048>(tc[3], t10, t11);
                HS_SB();
                mfma32_acc(dv[0][0], ta[0], pw[0][0]); HS_SB();
                mfma32_acc(dv[1][0], ta[0], pw[1][0]); HS_SB();
                mfma32_acc(dv[0][1], ta[1], pw[0][0]); HS_SB();
                mfma32_acc(dv[1][1], ta[1], pw[1][0]); HS_SB();
                mfma32_acc(dk[0][0], ta[2], dw[0][0]); HS_SB();
                mfma32_acc(dk[1][0], ta[2], dw[1][0]); HS_SB();
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ta[3])::"memory");
                mfma32_acc(dk[0][1], ta[3], dw[0][0]); HS_SB();
                mfma32_acc(dk[1][1], ta[3], dw[1][0]); HS_SB();
                lds_wait4<0>(tc[0], tc[1], tc[2], tc[3]);
                HS_STAMP(8, i)
                mfma32_acc(dv[0][0], tc[0], pw[0][1]); HS_SB();
                mfma32_acc(dv[1][0], tc[0], pw[1][1]); HS_SB();
                mfma32_acc(dv[0][1], tc[1], pw[0][1]); HS_SB();
                mfma32_acc(dv[1][1], tc[1], pw[1][1]); HS_SB();
                mfma32_acc(dk[0][0], tc[2], dw[0][1]); HS_SB();
                mfma32_acc(dk[1][0], tc[2], dw[1][1]); HS_SB();
                mfma32_acc(dk[0][1], tc[3], dw[0][1]); HS_SB();
                mfma32_acc(dk[1][1], tc[3], dw[1][1]); HS_SB();
            }
            // hipcc pads no hazard behind an asm MFMA (cdna guide 5.7): whatever it places next — the hand-off's adds on the dQ tile, or
            // v_accvgpr copies of the accumulators where two instantiations of this body meet — must not read a result that is still
            // in the matrix pipe (an 8-pass MFMA needs 12 wait states; without them register 0 of every accumulator was stale).
            asm volatile("s_nop 15" ::: "memory");
            HS_STAMP(9, i)
            if (HC) {
                // ---- hand-off of slice i-1 ----
                float* out = cbase + (long)(i - 1) * 2048;
#ifdef HS_DEBUG_HOOKS
                if (a.dbg & 1) vm_wait2<0>(pred[0], pred[1]);
#endif
                if (has_pred) {
                    // ops behind P_{i-1}: F_i (i >= 3), L_i (wave 0), D_i (i even)
                    if (i < 3) vm_wait2<0>(pred[0], pred[1]); else if (PAR == 0) vm_wait2<6>(pred[0], pred[1]); else vm_wait2<1>(pred[0], pred[1]);
                    acc0 += __builtin_bit_cast(f32x4, pred[0]);
                    acc1 += __builtin_bit_cast(f32x4, pred[1]);
                } else if (seed) {
                    const float d0 = __uint_as_float(sr0), d1 = __uint_as_float(sr1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { acc0[r] += d0 * krem[r]; acc1[r] += d1 * krem[r]; }
                }
                if (is_last) {                                                   // S_i: dq rows (8 B per lane and query tile), range-checked
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) {
                        const f32x4 v = qt ? acc1 : acc0;
                        const int q = 32 * (i - 1) + 16 * qt + c16;
                        u32x2 o2;
                        o2[0] = pack2bf(v[0] * a.dq_mul, v[1] * a.dq_mul);
                        o2[1] = pack2bf(v[2] * a.dq_mul, v[3] * a.dq_mul);
                        __builtin_amdgcn_raw_buffer_store_b64(o2, rso, q < N ? (unsigned)(((long)q * D3 + head * 64 + 16 * wave + 4 * g) * 2) : HS_OOB_OFF, 0, 0);
                    }
                } else if (chain) {
                    gstore_sc1(out, __builtin_bit_cast(u32x4, acc0));
                    gstore_sc1(out + 256, __builtin_bit_cast(u32x4, acc1));
                } else {
                    gstore_plain(out, __builtin_bit_cast(u32x4, acc0));
                    gstore_plain(out + 256, __builtin_bit_cast(u32x4, acc1));
                }
            }
            HS_STAMP(10, i)
            if (has_pred && HV) {                                                // P_i
#ifdef HS_DEBUG_HOOKS
                if (a.dbg & 2) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
                const float* pp = cbase + (long)i * 2048;
                gload_sc1(pred[0], pp);
                gload_sc1(pred[1], pp + 256);
            }
            if (HV) lgkm0();                                                     // this wave's dS^T writes have landed before it reaches the next barrier
            HS_STAMP(11, i)
            HS_STAMP_OUT(i)
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;
        typedef std::true_type T;
        typedef std::false_type F;
        {   // prologue: S', dP' of slice 0
            a_reads1(0, I0{});
            lds_wait4<0>(cl4[0], cl4[1], cl4[2], cl4[3]);
            lds_wait4<0>(qf[0], qf[1], qf[2], qf[3]);
            a_reads2(0, I0{});
            lds_wait4<0>(cd4[0], cd4[1], cd4[2], cd4[3]);
            lds_wait4<0>(df[0], df[1], df[2], df[3]);
            asm volatile("s_nop 7" ::: "memory");
            const f32x16 cl = cat16(cl4[0], cl4[1], cl4[2], cl4[3]), cd = cat16(cd4[0], cd4[1], cd4[2], cd4[3]);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                mfma32_first(sS[kh], qf[0], kf[kh][0], cl);
                mfma32_next(sS[kh], qf[1], kf[kh][1]); mfma32_next(sS[kh], qf[2], kf[kh][2]); mfma32_next(sS[kh], qf[3], kf[kh][3]);
                mfma32_first(sD[kh], df[0], vf[kh][0], cd);
                mfma32_next(sD[kh], df[1], vf[kh][1]); mfma32_next(sD[kh], df[2], vf[kh][2]); mfma32_next(sD[kh], df[3], vf[kh][3]);
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        }
        if (nslice > 1) iter(0, I0{}, T{}, T{}, F{}); else iter(0, I0{}, F{}, T{}, F{});
        int i = 1;
        for (; i + 1 <= nslice - 2; i += 2) {
            iter(i, I1{}, T{}, T{}, T{});
            iter(i + 1, I0{}, T{}, T{}, T{});
        }
        if (i <= nslice - 2) { iter(i, I1{}, T{}, T{}, T{}); ++i; }
        if (i == nslice - 1) {                                                   // last slice: nothing left to prefetch
            if (i & 1) iter(i, I1{}, F{}, T{}, T{}); else iter(i, I0{}, F{}, T{}, T{});
            ++i;
        }
        if (i & 1) iter(i, I1{}, F{}, F{}, T{}); else iter(i, I0{}, F{}, F{}, T{});  // i == nslice: the last slice's dQ
        vm_wait<0>();
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                  // (asm MFMA results -> compiler reads of dk / dv below)
        if (chain && lane == 0) {                                                // slices nslice-2 (if any) and nslice-1
            const unsigned left = nslice >= 2 ? 2u : 1u;
            __hip_atomic_fetch_add(my_ctr, left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // dK, dV of this wave's 64 keys
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            bf16_t* row = a.dqkv + ((long)b * N + key0 + 64 * wave + 32 * kh + c) * D3 + head * 64;
            store_dT(row + Dm, dk[kh], a.dk_mul, h);
            store_dT(row + 2 * Dm, dv[kh], 1.0f, h);
        }
        vm_wait<0>();
        __builtin_amdgcn_s_barrier();
    }
}

// fixed-order sum of the per-key-block dQ tiles (F_PARTIAL mode: every member stored its own tile) + the remainder key's term
__global__ __launch_bounds__(256) void attn_bwd_reduce_kernel(const float* __restrict__ chain, const float* __restrict__ stats,
                                                              const bf16_t* __restrict__ qkv, bf16_t* __restrict__ dqkv, int B, int N,
                                                              int H, int M, int nslice, int rem, float dq_mul) {
    const long unit = (long)blockIdx.x * 256 + threadIdx.x;            // one f32x4 of a slice tile
    const long per_bh = (long)nslice * 512;
    const long bh = unit / per_bh;
    if (bh >= (long)B * H) return;
    const long rest = unit - bh * per_bh;
    const int slice = (int)(rest >> 9), u = (int)(rest & 511);
    const int wave = u >> 7, qt = (u >> 6) & 1, lane = u & 63, g = lane >> 4, c16 = lane & 15;
    const int b = (int)(bh / H), head = (int)(bh % H), Dm = H * 64, D3 = 3 * Dm;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q = 32 * slice + 16 * qt + c16, d0 = 16 * wave + 4 * g;
    if (rem && q < N) {
        const float dsr = stats[(2 * (long)B * H + bh) * N + q];
        const bf16_t* kp = qkv + ((long)b * N + (N - 1)) * D3 + Dm + head * 64 + d0;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = dsr * bf2f(kp[r]);
    }
    for (int m = 0; m < M; ++m) acc += *(const f32x4*)(chain + ((bh * M + m) * nslice + slice) * 2048 + (long)u * 4);
    if (q < N) {
        u32x2 o2;
        o2[0] = pack2bf(acc[0] * dq_mul, acc[1] * dq_mul);
        o2[1] = pack2bf(acc[2] * dq_mul, acc[3] * dq_mul);
        *(u32x2*)(dqkv + ((long)b * N + q) * D3 + head * 64 + d0) = o2;
    }
}

// ------------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------------
namespace {
struct FusedPlan { bool ok; int M, rem, nslice, ngang, mode; size_t off_chain, off_ctr, bytes; };
int fused_env() {          // HS_ATTN_FUSED: 0 = never, 1 = when the shape fits and the grid fills (default), 2 = whenever the shape fits; +10: F_PARTIAL
    static const int v = [] { const char* e = getenv("HS_ATTN_FUSED"); return e ? atoi(e) : 1; }();
    return v;
}
FusedPlan fused_plan(int B, int N, int H) {
    FusedPlan p{};
    const int env = fused_env();
    if (env % 10 == 0) return p;
    const int M = N / 256, rem = N % 256;
    if (M < 1 || M > 32 || (M & (M - 1)) || rem > 1) return p;
    if ((size_t)N * 3 * H * 64 * 2 >= HS_MAX_RECORDS) return p;
    p.M = M; p.rem = rem; p.nslice = (N + 31) / 32; p.ngang = 8 * (32 / M);
    if (env % 10 == 1 && (long)B * H < p.ngang) return p;          // fewer (batch, head) pairs than gangs: the two-sweep kernels fill the chip better
    p.mode = env >= 10 ? F_PARTIAL : F_CHAIN;
    const size_t BH = (size_t)B * H;
    p.off_chain = (3 * BH * N * 4 + 255) & ~(size_t)255;
    const size_t chain_bytes = BH * (p.mode == F_PARTIAL ? M : 1) * (size_t)p.nslice * 2048 * 4;
    p.off_ctr = p.off_chain + chain_bytes;
    p.bytes = p.off_ctr + ((FCTR + BH * M) * 4 + 255 & ~(size_t)255);
    p.ok = true;
    return p;
}
}  // namespace

#ifdef HS_DEBUG_HOOKS
static unsigned long long* g_fused_stamps = nullptr;
// diagnostic (hook builds only): device buffer of 64 u64 that receives the phase stamps of iterations 40..47 (wave 0, last member of gang 0)
extern "C" void hs_debug_attn_stamps(unsigned long long* dev_buf) { g_fused_stamps = dev_buf; }
#endif

extern "C" size_t hs_attn_bwd_ws_bytes(int B, int N, int H, int flags) {
    if (B < 1 || N < 1 || H < 1) return 0;
    const size_t two_sweep = (size_t)B * H * N * 4 * 2;
    if (flags & HS_ATTN_F32) return two_sweep;
    const FusedPlan p = fused_plan(B, N, H);
    return p.ok && p.bytes > two_sweep ? p.bytes : two_sweep;
}

extern "C" int hs_attn_bwd2(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, void* ws, size_t ws_bytes,
                            int B, int N, int H, int dh, int flags, void* stream) {
    if (!qkv || !o || !d_o || !lse || !dqkv || !ws) return HS_ERR_ARG;
    if (dh != 64 || B < 1 || N < 1 || H < 1) return HS_ERR_SHAPE;
    if (flags & ~(HS_ATTN_Q_PRESCALED | HS_ATTN_F32)) return HS_ERR_ARG;
    if (((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dqkv | (uintptr_t)ws) & 15) return HS_ERR_SHAPE;
    if (ws_bytes < (size_t)B * H * N * 4 * 2) return HS_ERR_SHAPE;
    const FusedPlan p = (flags & HS_ATTN_F32) ? FusedPlan{} : fused_plan(B, N, H);
    if (!p.ok || ws_bytes < p.bytes) return hs_attn_bwd(qkv, o, d_o, lse, dqkv, (float*)ws, B, N, H, dh, flags, stream);
    const float scale = 0.125f, LOG2E = 1.44269504088896341f;
    const bool pre = (flags & HS_ATTN_Q_PRESCALED) != 0;
    const float c2 = pre ? 1.0f : scale * LOG2E;
    const float dk_mul = pre ? 1.0f / LOG2E : scale;
    hipStream_t s = (hipStream_t)stream;
    float* stats = (float*)ws;
    float* chain = (float*)((char*)ws + p.off_chain);
    unsigned* ctr = (unsigned*)((char*)ws + p.off_ctr);
    const double nn = (double)B * H * N * N * 64, io = (double)B * N * H * 64 * 2;
    {
        ProfScope prof(HS_PROF_ATTN_DELTA, stream, 0.0, 3 * io);
        hipLaunchKernelGGL(attn_bwd_pre_kernel, dim3((unsigned)(B * H)), dim3(256), 0, s, (const bf16_t*)qkv, (const bf16_t*)d_o, (const bf16_t*)o, lse,
                           stats, (bf16_t*)dqkv, ctr, B, N, H, p.M, p.rem, c2, dk_mul);
    }
    FusedArgs a;
    a.qkv = (const bf16_t*)qkv; a.d_o = (const bf16_t*)d_o; a.stats = stats; a.dqkv = (bf16_t*)dqkv; a.chain = chain; a.ctr = ctr;
    a.B = B; a.N = N; a.H = H; a.M = p.M; a.nslice = p.nslice; a.ngang = p.ngang; a.rem = p.rem; a.mode = p.mode;
    a.kmul = c2; a.dk_mul = dk_mul; a.dq_mul = scale;
    a.stamps = nullptr; a.dbg = 0;
#ifdef HS_DEBUG_HOOKS
    a.stamps = g_fused_stamps; a.dbg = g_hs_dbg[4];
#endif
    {
        // FLOP credit: the whole backward = 2.0 x the forward's 4 nn (dV, dP, dQ, dK); executed: 10 nn (S once)
        ProfScope prof(HS_PROF_ATTN_BWD1, stream, 8.0 * nn, 7 * io);
        hipLaunchKernelGGL(attn_bwd_fused_kernel, dim3(256), dim3(256), 0, s, a);
    }
    if (p.mode == F_PARTIAL) {
        ProfScope prof(HS_PROF_OTHER, stream, 0.0, 0.0);
        const long units = (long)B * H * p.nslice * 512;
        hipLaunchKernelGGL(attn_bwd_reduce_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, chain, stats, (const bf16_t*)qkv,
                           (bf16_t*)dqkv, B, N, H, p.M, p.nslice, p.rem, scale);
    }
    HS_CHECK_LAUNCH();
    return HS_OK;
}

// error word of the last fused launch that used this workspace (0 = none; 1 = a bounded hand-off spin ran out: results void).
// Synchronises the stream; diagnostic / test use only.
extern "C" int hs_attn_bwd_ws_error(const void* ws, int B, int N, int H, void* stream) {
    if (!ws) return HS_ERR_ARG;
    const FusedPlan p = fused_plan(B, N, H);
    if (!p.ok) return 0;
    unsigned v[2] = {0u, 0u};
    if (hipMemcpyAsync(v, (const char*)ws + p.off_ctr, 8, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return HS_ERR_LAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return HS_ERR_LAUNCH;
    return (int)v[1];
}
