// K4 backward, ONE sweep (round 4): dQ, dK, dV of the MONAI SABlock core (reference hot loops multimodal_encoder/vit.py:352-354,
// 464-466) with S and dP computed ONCE per (query, key) tile: five MFMA products per tile (S = Q K^T, dP = dO V^T, dV^T += dO^T P,
// dK^T += Q^T dS, dQ^T += K^T dS^T) = 10 N^2 d executed, against the 14 N^2 d of the two-sweep kernels of attn.hip.
// STATUS: correct (bit-level agreement of dK / dV with the two sweeps to 7e-5, dQ / dK / dV within 3.1e-3 of an fp32 reference at
// N = 256 ... 4097), deterministic, and SLOWER than the two sweeps on MI355X (9.1 vs 8.1 ms at 53 volumes): a caller
// has to ask for it (HS_ATTN_ONE_SWEEP).  It stays in the library as the measured answer to "compute S once", with its tests.
// Structure (cdna guide, Appendix B "Attention backward"):
//   * a workgroup = 4 waves = 256 keys of one (batch, head); a wave keeps K, V (MFMA operands) and the dK^T, dV^T accumulators of
//     its 64 keys in the accumulator registers (one wave per SIMD, the whole 512-register file) and sweeps the queries in 32-row slices;
//   * S and dP are computed with the key on the lane (row constants -lse*log2e / -delta preloaded as the accumulators), so P and dS
//     are directly the B operands of the dV^T / dK^T products; dS crosses LDS once ([key][query] image, read back transposed) for the
//     dQ product, where each wave owns 16 of the 64 d columns and sums over all 256 keys of the workgroup (16x16x32 MFMAs):
//     no partial sums inside the workgroup;
//   * the dQ slice of a (batch, head) is summed over its key blocks IN A FIXED ORDER by an ordered hand-off (bitwise reproducible,
//     no float atomics: the chip adds 1.3 TB/s of them): the kernel is persistent, the 256 workgroups form
//     gangs of M = N / 256 members that walk the (batch, head) pairs together; member m adds its tile to member m-1's running sum.
//     The hand-off is self-validating: a 16-byte granule = four fp32 sums whose last word carries a tag {launch epoch, member} in its
//     low byte; the consumer loads the granules two slices ahead, checks all 128 tags of its wave and re-reads (from the memory side,
//     system scope) what is not there yet.  No flag, no counter, no store ever has to be waited for — which matters because on this
//     chip only LOADS complete in issue order: a counted vmcnt wait behind a store or an atomic released early (measured).
//     A member only ever waits for a member with a LOWER start ticket, so the protocol cannot deadlock whatever the dispatch order;
//     every spin is bounded and raises an error word instead of hanging.
//   * N = 256 M + rem with rem <= 1: the remainder key (the 4097th token of a 256^3 volume) is handled in fp32 by the small
//     pre-pass that also produces the row constants (attn_bwd_pre_kernel); its rank-1 dQ term seeds the chain.
// Software pipeline (one barrier per 32-query slice), every MFMA, LDS and vector-memory instruction hand-placed (asm + sched_barrier):
// see `iter` below.  What it taught (all measured with the s_memtime stamps of the HS_DEBUG_HOOKS build, tools/kbench attnb):
//   * hipcc pads no hazard around asm MFMAs: an accumulator copy it placed directly in front of one left register 0 of every dK / dV
//     accumulator stale -> `s_nop 1` inside the statement; but an instruction between two DEPENDENT MFMAs costs the chain its
//     forwarding (~64 cycles per link) -> chains are either issued back to back or alternated;
//   * one wave per SIMD issues ONE instruction stream: the slice's 56 MFMAs (1536 pipe cycles) share it with ~80 LDS instructions
//     (~12 cycles each), ~100 VALU instructions (528 cycles of exp2 / multiply / pack) and 7 vector-memory instructions (110-180 cycles
//     each with four waves in step) -> ~4400 cycles per slice, 35 % matrix-pipe utilisation, whatever the placement.  Two waves per
//     SIMD issue two streams; that is why the two-sweep kernels (57-62 % busy on 14 N^2 d) still win.
#include "../../../hsenet_amd/csrc/common.h"
#include "../../../hsenet_amd/csrc/attn_util.h"
#include "one_sweep.h"
#define HS_PROF_ATTN_BWD1 HS_PROF_OTHER      // (the product library has no profiling class for this kernel any more)
#include <type_traits>

namespace {
constexpr int FSTG = 4096 + 4096 + 768;  // ring stage = one 32-query slice: Q' image 4 KiB | dO image 4 KiB | 3 row-constant arrays (256 B slots)
constexpr int FNST = 6;                  // stages; slice s lives in stage s % 6, its LDS-DMA is issued 4 iterations ahead
constexpr int FRING = FNST * FSTG;
constexpr int FDS = 16384;               // dS^T image of one slice: [256 keys][32 q] bf16 (64-B rows, 8-B slots XOR-swizzled)
constexpr int FKT = FRING + 2 * FDS;     // K^T image [64 d][256 keys] bf16 (512-B rows, 16-B chunks XOR-swizzled): the dQ product's A operand
constexpr int FSMEM = FKT + 32768;
constexpr int FCTR = 16;                 // ctr[0] ticket, ctr[1] error word, ctr[2] launch epoch (the tag of the hand-off granules)
constexpr unsigned FSPIN = 1u << 18;     // bounded spin (reloads) before a member gives up and raises the error word

struct FusedArgs {
    const bf16_t* qkv; const bf16_t* d_o; const float* stats; bf16_t* dqkv; float* chain; unsigned* ctr;
    int B, N, H, M, nslice, ngang, rem, mode;
    float kmul, dk_mul, dq_mul;
    int dbg;                             // HS_DEBUG_HOOKS builds only
    unsigned long long* stamps;          // HS_DEBUG_HOOKS builds only: s_memtime stamps of wave 0 of one workgroup (else NULL)
};
enum { F_PARTIAL = 0, F_CHAIN = 1 };

template <int IMM>
__device__ __forceinline__ void ldsw64(unsigned base, unsigned lo, unsigned hi) {
    u32x2 v = {lo, hi};
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(base), "v"(v), "i"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ void tr2_issue(u32x4& f, unsigned b0, unsigned b1) {       // two 4-row transposed reads = one 16x16x32 B fragment
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"
                 : "=&v"(lo), "=&v"(hi) : "v"(b0), "v"(b1), "i"(IMM) : "memory");
    f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}
// hand-off traffic at AGENT scope (sc1: write-through stores, L1-bypassing loads).  System-scope (sc0 sc1) stores were measured 2x slower
// for the whole kernel (their acknowledgements take microseconds and every counted vmcnt wait has to sit them out), so they are used only
// for the RE-READ of a granule whose tag is not there yet: that read comes from the memory side, never from a per-XCD L2 line that another
// XCD's store has made stale.  The s_nop behind a store: the next instruction must not overwrite its data registers early (guide 5.7).
__device__ __forceinline__ void gload_sys(u32x4& d, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void gload_agent(u32x4& d, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void gstore_agent(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void gstore_plain(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int CNT>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(CNT) : "memory"); }
template <int CNT>
__device__ __forceinline__ void vm_wait2(u32x4& a, u32x4& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "i"(CNT) : "memory"); }
__device__ __forceinline__ void lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ f32x16 cat16(const u32x4& a, const u32x4& b, const u32x4& c, const u32x4& d) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r[i] = __uint_as_float(a[i]); r[4 + i] = __uint_as_float(b[i]); r[8 + i] = __uint_as_float(c[i]); r[12 + i] = __uint_as_float(d[i]);
    }
    return r;
}
// MFMAs issued from asm so that the register FILE of every operand is ours: the dK^T / dV^T accumulators and the K / V operands live
// in the accumulator half (AGPRs), the S' / dP' / dQ^T tiles in arch VGPRs where the VALU reads them.  Left to hipcc, every MFMA of a
// 512-register kernel gets an AGPR destination and each S' / dP' tile is copied out (1400 v_accvgpr moves in this kernel).
// hipcc pads NO hazard around asm (cdna guide 5.7).  In front of each MFMA: `s_nop 1` — hipcc may have placed a v_mov / v_accvgpr_mov that
// writes an operand directly in front of the statement (where two instantiations of the loop body meet it copies whole accumulators,
// element 0 last: without the nop register 0 of every dK / dV accumulator was stale).  Behind: every reader of a result sits >= 8 MFMAs
// later, or behind an explicit s_nop 15.
__device__ __forceinline__ void mfma32_first(f32x16& d, const u32x4& a, const bf16x8& b, const f32x16& c) {     // D(v) = A(v) B(a) + C(v)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
}
__device__ __forceinline__ void mfma32_next(f32x16& d, const u32x4& a, const bf16x8& b) {                        // D(v) += A(v) B(a), straight behind its
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));                          // predecessor (an instruction in between costs the chain ~60 cycles)
}
__device__ __forceinline__ void mfma32_acc(f32x16& acc, const u32x4& a, const u32x4& b) {                        // ACC(a) += A(v) B(v), first use in an iteration
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_acc_n(f32x16& acc, const u32x4& a, const u32x4& b) {                      // the same, later uses (hipcc was seen to
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));             // place accumulator copies in front of these too)
}
template <bool FIRST>
__device__ __forceinline__ void mfma16_acc2(f32x4& acc0, f32x4& acc1, const u32x4& a, const u32x4& b0, const u32x4& b1) {   // ACCn(v) += A(v) Bn(v)
    if (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %4, %1" : "+v"(acc0), "+v"(acc1) : "v"(a), "v"(b0), "v"(b1));
    else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %4, %1" : "+v"(acc0), "+v"(acc1) : "v"(a), "v"(b0), "v"(b1));
}
}  // namespace

// ------------------------------------------------------------------------------------------------------
// pre-pass: row constants (-delta = -sum_d dO*O, -lse*log2e) for every query row, and — when N = 256 M + 1 — the whole
// contribution of the remainder key r = N - 1 in fp32: dsr[q] = P[q][r] (dP[q][r] - delta[q]) (what the chain's first member
// multiplies with K[r] to seed dQ), dV[r] = sum_q P dO[q], dK[r] = dk_mul sum_q dS q'[q].  One block per (batch, head), 8 lanes per
// row (16 B each), fixed-order reductions.  Block 0 also resets the ticket / error words and advances the launch epoch.
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
    if (bh == 0 && tid == 0) { ctr[0] = 0u; ctr[1] = 0u; ctr[2] = (ctr[2] + 1u) & 0xffffffu; }   // ticket, error word, launch epoch (>= 1 in a zeroed workspace)
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
// Vector-memory accounting.  vmcnt counts loads, LDS-DMA pieces, stores and atomics together, but only the LOADS (VGPR loads and LDS-DMA)
// are known to complete in issue order: a counted wait written for "store ... load" order released early on this chip (measured: the
// flag atomics of the first version overtook older loads).  So every counted wait below counts the loads that may be younger than the
// one it needs and NOTHING else; an older store that is still in flight only makes the wait longer, never wrong.  Per iteration j a
// wave issues, in order:  D_j  3 LDS-DMA pieces (slice j + 4)       S_j  the hand-off stores of slice j - 1
//                         P_j  3 loads of the predecessor's granules of slice j + 1 (consumed in iteration j + 2)
// No store ever has to be waited for: the hand-off is self-validating (each 16-byte granule = 3 fp32 of the running dQ sum + a tag word
// {launch epoch, member}; 16-byte stores arrive whole), the consumer checks the tags and re-reads a granule that is not there yet.
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
    const int nslice = a.nslice;
    const unsigned epoch = a.ctr[2];
    // tag byte of a hand-off granule: never 0, changes with the launch and with the writing member
    const unsigned tag_mine = 0x80u | ((epoch & 7u) << 4) | (unsigned)(m & 15), tag_pred = 0x80u | ((epoch & 7u) << 4) | (unsigned)((m - 1) & 15);
    const long sstride = 2048;                           // floats per slice in the hand-off buffer (2 granules / tiles per lane)
    bool dead = false;                                   // a bounded spin ran out: stop waiting (results void, error word set)

    const unsigned sm = lds_addr(smem);
    unsigned rb0[4], tb0[2][2], wbj[4], dbr[2][2], ktb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rb0[ks] = sm + offR(c, 2 * ks + h);
    {
        const int p = lane & 3, q4 = (lane >> 2) & 3;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int col = db * 32 + 16 * (g & 1) + 4 * p;
            tb0[db][0] = sm + offR(4 * h + q4, col >> 3) + 8 * (p & 1);
            tb0[db][1] = sm + offR(4 * h + q4 + 8, col >> 3) + 8 * (p & 1);
        }
        // dS^T image: row = key (64 B = 8 slots of 4 queries), slot s of row r at slot s ^ ((r >> 1) & 7): the 16 lanes of a ds_write_b64 group
        // and the 32 lanes of a ds_read_b64_tr_b16 half each cover every bank once
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r4 = 0; r4 < 2; ++r4) {
                const int key = 8 * g + q4 + 4 * r4;
                dbr[qt][r4] = sm + FRING + key * 64 + (((4 * qt + p) ^ ((key >> 1) & 7)) << 3);
            }
    }
    const unsigned cb0 = sm + 8192 + 16 * h;
    const int kl = 64 * wave + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) wbj[j] = sm + FRING + kl * 64 + (((2 * j + h) ^ ((kl >> 1) & 7)) << 3);
    // K^T image: row = d (512 B = 32 chunks of 8 keys), chunk ch of row d at chunk ch ^ (d & 15); fragment ks8 of lane (d = 16 wave + (lane & 15), g):
    // chunk 4 ks8 + g
#pragma unroll
    for (int k = 0; k < 4; ++k) ktb[k] = sm + FKT + (16 * wave + c16) * 512 + (((4 * k + g) ^ c16) << 4);
    const unsigned srb0 = sm + 8192 + 512 + 4 * c16;      // dsr (row-constant array 2) of query 16 qt + (lane & 15)

    for (long bh = gang; bh < BH; bh += a.ngang) {
        const int b = (int)(bh / H), head = (int)(bh - (long)b * H);
        const int key0 = 256 * m;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.qkv + (long)b * N * D3, (size_t)N * D3 * 2);
        const __amdgpu_buffer_rsrc_t rsd = make_rsrc(a.d_o + (long)b * N * Dm, (size_t)N * Dm * 2);
        const int cw = wave < 3 ? wave : 2;
        const __amdgpu_buffer_rsrc_t rsc = make_rsrc(a.stats + ((long)cw * BH + bh) * N, (size_t)N * 4);
        const __amdgpu_buffer_rsrc_t rso = make_rsrc(a.dqkv + (long)b * N * D3, (size_t)N * D3 * 2);
        const int drow = wave * 8 + (lane >> 3);                                  // this lane's row of a slice's Q' / dO piece
        const unsigned vq = (unsigned)(((long)drow * D3 + head * 64 + ((lane & 7) ^ swzR(drow)) * 8) * 2);
        const unsigned vd = (unsigned)(((long)drow * Dm + head * 64 + ((lane & 7) ^ swzR(drow)) * 8) * 2);
        auto issue = [&](int sl) {                        // D: 3 LDS-DMA pieces per wave = one Q' piece, one dO piece, one row-constant array
            const int st = sl % FNST;
            const bool in = sl * 32 + drow < N, inc = lane < 32 && sl * 32 + lane < N;
            char* dst = smem + st * FSTG + wave * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))dst, 16, in ? vq : HS_OOB_OFF, (unsigned)sl * (unsigned)(32 * D3 * 2), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (LDS_PTR(void))(dst + 4096), 16, in ? vd : HS_OOB_OFF, (unsigned)sl * (unsigned)(32 * Dm * 2), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsc, (LDS_PTR(void))(smem + st * FSTG + 8192 + cw * 256), 4, inc ? (unsigned)lane * 4u : HS_OOB_OFF,
                                                     (unsigned)sl * 128u, 0, 0);
        };
        issue(0); issue(1); issue(2); issue(3);
        // K, V of this wave's 64 keys as B operands of S / dP (key on the lane)
        bf16x8 kf[2][4], vf[2][4];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const unsigned off = (unsigned)(((long)(key0 + 64 * wave + 32 * kh + c) * D3 + Dm + head * 64 + ks * 16 + 8 * h) * 2);
                kf[kh][ks] = scale_frag(buf_load16(rs, off), a.kmul);
                vf[kh][ks] = __builtin_bit_cast(bf16x8, buf_load16(rs, off + Dm * 2));
            }
        // K of the workgroup's 256 keys, staged row-major in the dS^T buffers, then rewritten transposed (K^T image)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = (wave * 8 + j) * 8 + (lane >> 3);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))(smem + FRING + (wave * 8 + j) * 1024), 16,
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
        {
            const int p = lane & 3, qq = (lane >> 2) & 3;
            const unsigned kb = sm + FRING + (8 * g + qq) * 128 + 32 * wave + 8 * p;
#pragma unroll
            for (int ks8 = 0; ks8 < 8; ++ks8) {
                u32x2 lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:512\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(lo), "=&v"(hi) : "v"(kb + ks8 * 4096) : "memory");
                u32x4 f; f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
                asm volatile("ds_write_b128 %0, %1" ::"v"(ktb[ks8 & 3] + (unsigned)((ks8 >> 2) * 256)), "v"(f) : "memory");
            }
        }
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

        float* cbase = a.chain + (chain ? bh * nslice : (bh * M + m) * nslice) * sstride + (long)wave * 512 + lane * 4;

        f32x16 sS[2], sD[2];                                   // S', dP' [key half] of the slice whose softmax runs next
        u32x4 predE[2], predO[2];                              // the predecessor's granules of the next even / odd slice to be finalised
#pragma unroll
        for (int k = 0; k < 2; ++k) { predE[k] = u32x4{0u, 0u, 0u, 0u}; predO[k] = predE[k]; }
        u32x4 cl4[4], cd4[4], qf[4], df[4];                    // phase A operands of the slice being prefetched
        u32x4 pw1[2], dw1[2];                                  // P, dS (bf16 B operands) of key half 1 of the current slice [query k-step]
        u32x4 pw0[2][2], dw0[2][2];                            // ... of key half 0 [slice parity][query k-step]: made one iteration early, beside B

        auto cl_keep = [&](u32x4 (&c4)[4]) { return cat16(c4[0], c4[1], c4[2], c4[3]); };
        auto a_reads1 = [&](int sl) {
            const unsigned so = (unsigned)((sl % FNST) * FSTG);
#pragma unroll
            for (int j = 0; j < 4; ++j) ldsr128<0>(cl4[j], cb0 + so + 32 * j);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ldsr128<0>(qf[ks], rb0[ks] + so);
        };
        auto a_reads2 = [&](int sl) {
            const unsigned so = (unsigned)((sl % FNST) * FSTG);
#pragma unroll
            for (int j = 0; j < 4; ++j) ldsr128<256>(cd4[j], cb0 + so + 32 * j);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ldsr128<4096>(df[ks], rb0[ks] + so);
        };
        auto p_issue1 = [&](u32x4& pr, int sl, int k) {        // P: granule k of the predecessor's running sum of slice sl
            gload_agent(pr, cbase + (long)sl * sstride + 256 * k);
        };
        auto p_reissue = [&](u32x4 (&pr)[2], int sl) {         // the same from the memory side (a granule that was not there yet)
            const float* pp = cbase + (long)sl * sstride;
            gload_sys(pr[0], pp); gload_sys(pr[1], pp + 256);
        };
#define HS_VSTEP(KH, R)                                                                 \
        { const float p_ = __builtin_amdgcn_exp2f(sS[KH][R]); sS[KH][R] = p_; sD[KH][R] = p_ * sD[KH][R]; }
#define HS_VPACK1(J)                                                                    \
        { pw1[(J) >> 2][(J) & 3] = pack2bf(sS[1][2 * (J)], sS[1][2 * (J) + 1]); dw1[(J) >> 2][(J) & 3] = pack2bf(sD[1][2 * (J)], sD[1][2 * (J) + 1]); }
#define HS_VPACK0(PP, J)                                                                \
        { pw0[PP][(J) >> 2][(J) & 3] = pack2bf(sS[0][2 * (J)], sS[0][2 * (J) + 1]); dw0[PP][(J) >> 2][(J) & 3] = pack2bf(sD[0][2 * (J)], sD[0][2 * (J) + 1]); }
#define HS_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef HS_DEBUG_HOOKS
        unsigned long long stv[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bool stamp_on = a.stamps != nullptr && gang == 0 && m == M / 2 && wave == 0 && bh == gang + a.ngang;     // a middle member of gang 0, its second (batch, head)
#define HS_STAMP(K, I) if (stamp_on && (I) >= 40 && (I) < 48) asm volatile("s_memtime %0" : "=s"(stv[K]));
#define HS_STAMP_OUT(I)                                                                                  \
        if (stamp_on && (I) >= 40 && (I) < 48) {                                                         \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
            if (lane == 0) { for (int k_ = 0; k_ < 12; ++k_) a.stamps[((I) - 40) * 12 + k_] = stv[k_]; }  \
        }
#else
#define HS_STAMP(K, I)
#define HS_STAMP_OUT(I)
#endif

        // issue() split into its three pieces so that each can sit behind an MFMA group of its own (a wave that issues vector-memory
        // instructions back to back stalls until the texture path has taken them: ~110-180 cycles apiece with four waves in step)
        auto issue_q = [&](int sl) {
            const bool in = sl * 32 + drow < N;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))(smem + (sl % FNST) * FSTG + wave * 1024), 16, in ? vq : HS_OOB_OFF, (unsigned)sl * (unsigned)(32 * D3 * 2), 0, 0);
        };
        auto issue_d = [&](int sl) {
            const bool in = sl * 32 + drow < N;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (LDS_PTR(void))(smem + (sl % FNST) * FSTG + 4096 + wave * 1024), 16, in ? vd : HS_OOB_OFF, (unsigned)sl * (unsigned)(32 * Dm * 2), 0, 0);
        };
        auto issue_c = [&](int sl) {
            const bool inc = lane < 32 && sl * 32 + lane < N;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsc, (LDS_PTR(void))(smem + (sl % FNST) * FSTG + 8192 + cw * 256), 4, inc ? (unsigned)lane * 4u : HS_OOB_OFF, (unsigned)sl * 128u, 0, 0);
        };
        // ---- one pipeline iteration (PAR = i & 1) ----
        //   step 1  C(i-1): dQ^T of slice i-1 (16 MFMAs 16x16x32), the three LDS-DMA pieces of slice i+4 behind its MFMA groups
        //   step 2  A(i+1) key half 0 (8 MFMAs) beside V(i) key half 1 (exp2, multiply, pack)
        //   step 3  A(i+1) key half 1 (8 MFMAs, the two chains alternating) beside the hand-off of slice i-1
        //   step 4  B(i): dV^T, dK^T (16 MFMAs) beside V(i+1) key half 0, the dS^T image writes of slice i and the predecessor loads of slice i+1
        // Vector-memory operations per iteration j, in issue order: D_j (3 loads), S_j (2 stores), P_j (2 loads: slice j+1).
        auto iter = [&](int i, auto par_tag, auto ha_tag, auto hv_tag, auto hc_tag) {
            constexpr int PAR = decltype(par_tag)::value;
            constexpr bool HA = decltype(ha_tag)::value, HV = decltype(hv_tag)::value, HC = decltype(hc_tag)::value;
            HS_STAMP(0, i)
            // top: slice i+1 (issued as D_{i-3}) must have landed; loads that may be younger: P_{i-3}, D_{i-2}, P_{i-2}, D_{i-1}, P_{i-1}
            if (has_pred) vm_wait<12>(); else vm_wait<6>();
            HS_STAMP(1, i)
            __builtin_amdgcn_s_barrier();
            HS_STAMP(2, i)
            unsigned sr0 = 0u, sr1 = 0u;
            if (HC && seed) {                                                 // dsr of slice i-1 (its stage is refilled by D_{i+1})
                const unsigned sa = srb0 + (unsigned)(((i - 1) % FNST) * FSTG);
                asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=&v"(sr0), "=&v"(sr1) : "v"(sa) : "memory");
            }
#ifdef HS_DEBUG_HOOKS
            const bool dma_on = !((a.dbg & 16) && i >= 8);
#else
            constexpr bool dma_on = true;
#endif
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
            if (HC) {
                constexpr unsigned RO = (PAR ^ 1) * FDS;
                u32x4 gA[3], gB[3], gC[3];                     // {dS^T q-tile 0, dS^T q-tile 1, K^T} fragments of one 32-key step; three steps in flight
#define HS_CREAD(G, KS8)                                                                                                       \
                tr2_issue<RO + (KS8) * 2048>(G[0], dbr[0][0], dbr[0][1]); tr2_issue<RO + (KS8) * 2048>(G[1], dbr[1][0], dbr[1][1]); \
                ldsr128<((KS8) >> 2) * 256>(G[2], ktb[(KS8) & 3]);
#define HS_CMMA(G, FIRST)                                                                                                      \
                mfma16_acc2<FIRST>(acc0, acc1, G[2], G[0], G[1]); HS_SB();
#define HS_CWAIT(G, CNT) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(G[0]), "+v"(G[1]), "+v"(G[2]) : "i"(CNT) : "memory");
                HS_CREAD(gA, 0) HS_CREAD(gB, 1) HS_CREAD(gC, 2)
                HS_CWAIT(gA, 10) HS_CMMA(gA, true) HS_CREAD(gA, 3)
                if (dma_on) issue_q(i + 4);
                HS_CWAIT(gB, 10) HS_CMMA(gB, false) HS_CREAD(gB, 4)
                HS_CWAIT(gC, 10) HS_CMMA(gC, false) HS_CREAD(gC, 5)
                if (dma_on) issue_d(i + 4);
                HS_CWAIT(gA, 10) HS_CMMA(gA, false) HS_CREAD(gA, 6)
                HS_CWAIT(gB, 10) HS_CMMA(gB, false) HS_CREAD(gB, 7)
                if (dma_on) issue_c(i + 4);
                HS_CWAIT(gC, 10) HS_CMMA(gC, false)
                HS_CWAIT(gA, 5)
                if (HA) a_reads1(i + 1);
                HS_SB();
                HS_CMMA(gA, false)
                if (HA) { HS_CWAIT(gB, 8) } else { HS_CWAIT(gB, 0) }
                HS_CMMA(gB, false)
#undef HS_CREAD
#undef HS_CMMA
#undef HS_CWAIT
            } else {
                if (dma_on) { issue_q(i + 4); issue_d(i + 4); issue_c(i + 4); }
                if (HA) a_reads1(i + 1);
                HS_SB();
            }
            HS_STAMP(3, i)
            // ---- step 2 ----
            if (HA) {
                lds_wait4<0>(cl4[0], cl4[1], cl4[2], cl4[3]);
                lds_wait4<0>(qf[0], qf[1], qf[2], qf[3]);
                a_reads2(i + 1);
                HS_SB();
                // (the VALU work between them spaces these dependent MFMAs ~65 cycles apart: no latency stall; step 3 has no such filler and
                // alternates the two chains instead)
                const f32x16 cl = cat16(cl4[0], cl4[1], cl4[2], cl4[3]);
                mfma32_first(sS[0], qf[0], kf[0][0], cl); if (HV) { HS_VSTEP(1, 0) HS_VSTEP(1, 1) HS_VPACK1(0) } HS_SB();
                mfma32_next(sS[0], qf[1], kf[0][1]);      if (HV) { HS_VSTEP(1, 2) HS_VSTEP(1, 3) HS_VPACK1(1) } HS_SB();
                mfma32_next(sS[0], qf[2], kf[0][2]);      if (HV) { HS_VSTEP(1, 4) HS_VSTEP(1, 5) HS_VPACK1(2) } HS_SB();
                mfma32_next(sS[0], qf[3], kf[0][3]);      if (HV) { HS_VSTEP(1, 6) HS_VSTEP(1, 7) HS_VPACK1(3) } HS_SB();
                lds_wait4<0>(cd4[0], cd4[1], cd4[2], cd4[3]);
                lds_wait4<0>(df[0], df[1], df[2], df[3]);
                const f32x16 cd = cat16(cd4[0], cd4[1], cd4[2], cd4[3]);
                mfma32_first(sD[0], df[0], vf[0][0], cd); if (HV) { HS_VSTEP(1, 8) HS_VSTEP(1, 9) HS_VPACK1(4) } HS_SB();
                mfma32_next(sD[0], df[1], vf[0][1]);      if (HV) { HS_VSTEP(1, 10) HS_VSTEP(1, 11) HS_VPACK1(5) } HS_SB();
                mfma32_next(sD[0], df[2], vf[0][2]);      if (HV) { HS_VSTEP(1, 12) HS_VSTEP(1, 13) HS_VPACK1(6) } HS_SB();
                mfma32_next(sD[0], df[3], vf[0][3]);      if (HV) { HS_VSTEP(1, 14) HS_VSTEP(1, 15) HS_VPACK1(7) } HS_SB();
            } else if (HV) {
#pragma unroll
                for (int r = 0; r < 16; ++r) HS_VSTEP(1, r)
#pragma unroll
                for (int j = 0; j < 8; ++j) HS_VPACK1(j)
                HS_SB();
            }
            HS_STAMP(4, i)
            // ---- step 3: A(i+1) key half 1 and the hand-off of slice i-1 (the dQ tile of C(i-1) + the predecessor's running sum) ----
            u32x4 g0 = u32x4{0u, 0u, 0u, 0u}, g1 = g0;
            float* out = cbase + (long)(i - 1) * sstride;
#ifdef HS_DEBUG_HOOKS
            const bool ho_on = HC && !(a.dbg & 8);
#else
            constexpr bool ho_on = HC;
#endif
            auto handoff_sum = [&]() {                         // after the pred wait: validate, add, tag
                if (has_pred) {
                    u32x4 (&pr)[2] = PAR ? predE : predO;      // slice i-1 has the other parity
                    // issued in step 4 of iteration i-2; loads that may be younger: D_{i-1}, P_{i-1}, D_i
                    vm_wait2<8>(pr[0], pr[1]);
                    unsigned spins = 0;
                    while (__builtin_amdgcn_ballot_w64((pr[0][3] & 0xffu) != tag_pred || (pr[1][3] & 0xffu) != tag_pred) != 0ull && !dead) {
                        p_reissue(pr, i - 1);                  // not there yet: read the granules again, from the memory side (rare: the predecessor runs slices ahead)
                        vm_wait2<0>(pr[0], pr[1]);
                        if (++spins > FSPIN) { dead = true; if (lane == 0) a.ctr[1] = 1u; }
                        __builtin_amdgcn_s_sleep(8);
                    }
#ifdef HS_DEBUG_HOOKS
                    if (stamp_on && lane == 0) { a.stamps[97] += 1ull; if (spins) { a.stamps[96] += 1ull; a.stamps[98] += spins; } }
#endif
                    acc0[0] += __uint_as_float(pr[0][0]); acc0[1] += __uint_as_float(pr[0][1]); acc0[2] += __uint_as_float(pr[0][2]);
                    acc0[3] += __uint_as_float(pr[0][3] & 0xffffff00u);
                    acc1[0] += __uint_as_float(pr[1][0]); acc1[1] += __uint_as_float(pr[1][1]); acc1[2] += __uint_as_float(pr[1][2]);
                    acc1[3] += __uint_as_float(pr[1][3] & 0xffffff00u);
                } else if (seed) {
                    const float d0 = __uint_as_float(sr0), d1 = __uint_as_float(sr1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { acc0[r] += d0 * krem[r]; acc1[r] += d1 * krem[r]; }
                }
                // a granule = the four running sums of one query tile; the low byte of the last word carries the tag (8 of its 24 mantissa bits:
                // 2^-16 relative on one value in four, deterministic; dq leaves in bf16)
                g0 = __builtin_bit_cast(u32x4, acc0); g1 = __builtin_bit_cast(u32x4, acc1);
                if (chain && !is_last) { g0[3] = (g0[3] & 0xffffff00u) | tag_mine; g1[3] = (g1[3] & 0xffffff00u) | tag_mine; }
            };
            auto handoff_store = [&](int k) {                  // S: one of the two stores of the finalised slice
                const u32x4 gk = k ? g1 : g0;
                if (is_last) {                                 // dq rows (8 B per lane and query tile), range-checked
                    const f32x4 v = __builtin_bit_cast(f32x4, gk);
                    const int q = 32 * (i - 1) + 16 * k + c16;
                    u32x2 o2;
                    o2[0] = pack2bf(v[0] * a.dq_mul, v[1] * a.dq_mul);
                    o2[1] = pack2bf(v[2] * a.dq_mul, v[3] * a.dq_mul);
                    __builtin_amdgcn_raw_buffer_store_b64(o2, rso, q < N ? (unsigned)(((long)q * D3 + head * 64 + 16 * wave + 4 * g) * 2) : HS_OOB_OFF, 0, 0);
                } else if (chain) gstore_agent(out + 256 * k, gk);
                else gstore_plain(out + 256 * k, gk);
            };
            if (HA) {
                constexpr unsigned WO = PAR * FDS;
                (void)WO;
                // (asm MFMA results -> VALU: the last C(i-1) MFMA sits 8 MFMAs back)
                mfma32_first(sS[1], qf[0], kf[1][0], cl_keep(cl4)); HS_SB();
                mfma32_first(sD[1], df[0], vf[1][0], cl_keep(cd4)); if (ho_on) handoff_sum(); HS_SB();
                mfma32_next(sS[1], qf[1], kf[1][1]); HS_SB();
                mfma32_next(sD[1], df[1], vf[1][1]); if (ho_on) handoff_store(0); HS_SB();
                mfma32_next(sS[1], qf[2], kf[1][2]); HS_SB();
                mfma32_next(sD[1], df[2], vf[1][2]); if (ho_on) handoff_store(1); HS_SB();
                mfma32_next(sS[1], qf[3], kf[1][3]); HS_SB();
                mfma32_next(sD[1], df[3], vf[1][3]); HS_SB();
            } else if (ho_on) {
                asm volatile("s_nop 15" ::: "memory");
                handoff_sum(); handoff_store(0); handoff_store(1);
            }
            HS_STAMP(5, i)
            // ---- step 4 ----
            if (HV) {
                lgkm0();
                const unsigned so = (unsigned)((i % FNST) * FSTG);
                const unsigned t00 = tb0[0][0] + so, t01 = tb0[0][1] + so, t10 = tb0[1][0] + so, t11 = tb0[1][1] + so;
                u32x4 ta[4], tc[4];                            // {dO^T db0, dO^T db1, Q^T db0, Q^T db1} of query k-steps 0 / 1
                trR_issue<4096>(ta[0], t00, t01);
                trR_issue<4096>(ta[1], t10, t11);
                trR_issue<0>(ta[2], t00, t01);
                trR_issue<0>(ta[3], t10, t11);
                trR_issue<4096 + 2048>(tc[0], t00, t01);
                trR_issue<4096 + 2048>(tc[1], t10, t11);
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ta[0]), "+v"(ta[1])::"memory");
                HS_STAMP(6, i)
                HS_SB();
                constexpr unsigned WO = PAR * FDS;
#ifdef HS_DEBUG_HOOKS
                const bool w_on = !(a.dbg & 4), p_on = has_pred && i + 1 < nslice && !(a.dbg & 8);
#else
                constexpr bool w_on = true;
                const bool p_on = has_pred && i + 1 < nslice;
#endif
#define HS_V0(R) if (HA) { HS_VSTEP(0, R) }
#define HS_P0(J) if (HA) { HS_VPACK0(PAR ^ 1, J) }
                mfma32_acc(dv[0][0], ta[0], pw0[PAR][0]); HS_V0(0) HS_SB();
                trR_issue<2048>(tc[2], t00, t01);
                mfma32_acc(dv[1][0], ta[0], pw1[0]); HS_V0(1) HS_P0(0) HS_SB();
                trR_issue<2048>(tc[3], t10, t11);
                mfma32_acc(dv[0][1], ta[1], pw0[PAR][0]); HS_V0(2) HS_SB();
                mfma32_acc(dv[1][1], ta[1], pw1[0]); HS_V0(3) HS_P0(1) HS_SB();
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ta[2]), "+v"(ta[3])::"memory");
                mfma32_acc(dk[0][0], ta[2], dw0[PAR][0]); HS_V0(4) HS_SB();
                if (p_on) p_issue1((PAR ? predE : predO)[0], i + 1, 0);                                   // P_i: slice i+1 (parity of i-1), consumed in iteration i+2
                mfma32_acc(dk[1][0], ta[2], dw1[0]); HS_V0(5) HS_P0(2) HS_SB();
                mfma32_acc(dk[0][1], ta[3], dw0[PAR][0]); HS_V0(6) HS_SB();
                mfma32_acc(dk[1][1], ta[3], dw1[0]); HS_V0(7) HS_P0(3) HS_SB();
                lds_wait4<0>(tc[0], tc[1], tc[2], tc[3]);
                HS_STAMP(7, i)
                if (p_on) p_issue1((PAR ? predE : predO)[1], i + 1, 1);
                // W(i): the dS^T image of this slice, one write behind each MFMA
                mfma32_acc_n(dv[0][0], tc[0], pw0[PAR][1]); HS_V0(8) if (w_on) ldsw64<WO>(wbj[0], dw0[PAR][0][0], dw0[PAR][0][1]); HS_SB();
                mfma32_acc_n(dv[1][0], tc[0], pw1[1]); HS_V0(9) HS_P0(4) if (w_on) ldsw64<WO>(wbj[1], dw0[PAR][0][2], dw0[PAR][0][3]); HS_SB();
                mfma32_acc_n(dv[0][1], tc[1], pw0[PAR][1]); HS_V0(10) if (w_on) ldsw64<WO>(wbj[2], dw0[PAR][1][0], dw0[PAR][1][1]); HS_SB();
                mfma32_acc_n(dv[1][1], tc[1], pw1[1]); HS_V0(11) HS_P0(5) if (w_on) ldsw64<WO>(wbj[3], dw0[PAR][1][2], dw0[PAR][1][3]); HS_SB();
                mfma32_acc_n(dk[0][0], tc[2], dw0[PAR][1]); HS_V0(12) if (w_on) ldsw64<WO + 2048>(wbj[0], dw1[0][0], dw1[0][1]); HS_SB();
                mfma32_acc_n(dk[1][0], tc[2], dw1[1]); HS_V0(13) HS_P0(6) if (w_on) ldsw64<WO + 2048>(wbj[1], dw1[0][2], dw1[0][3]); HS_SB();
                mfma32_acc_n(dk[0][1], tc[3], dw0[PAR][1]); HS_V0(14) if (w_on) ldsw64<WO + 2048>(wbj[2], dw1[1][0], dw1[1][1]); HS_SB();
                mfma32_acc_n(dk[1][1], tc[3], dw1[1]); HS_V0(15) HS_P0(7) if (w_on) ldsw64<WO + 2048>(wbj[3], dw1[1][2], dw1[1][3]); HS_SB();
#undef HS_V0
#undef HS_P0
            }
            // (results of the asm MFMAs above may be read by whatever hipcc places next, e.g. accumulator copies where two instantiations meet)
            asm volatile("s_nop 15" ::: "memory");
            HS_STAMP(8, i)
            if (HV) lgkm0();                                                     // this wave's dS^T writes have landed before it reaches the next barrier
            HS_STAMP(9, i) HS_STAMP(10, i) HS_STAMP(11, i)
            HS_STAMP_OUT(i)
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;
        typedef std::true_type T;
        typedef std::false_type F;
        if (has_pred) { p_issue1(predE[0], 0, 0); p_issue1(predE[1], 0, 1); }
        {   // prologue: S', dP' of slice 0
            a_reads1(0);
            lds_wait4<0>(cl4[0], cl4[1], cl4[2], cl4[3]);
            lds_wait4<0>(qf[0], qf[1], qf[2], qf[3]);
            a_reads2(0);
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
#pragma unroll
            for (int r = 0; r < 16; ++r) HS_VSTEP(0, r)
#pragma unroll
            for (int j = 0; j < 8; ++j) HS_VPACK0(0, j)
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
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                      // (asm MFMA results -> compiler reads of dk / dv below)
        // dK, dV of this wave's 64 keys
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            bf16_t* row = a.dqkv + ((long)b * N + key0 + 64 * wave + 32 * kh + c) * D3 + head * 64;
            store_dT(row + Dm, dk[kh], a.dk_mul, h);
            store_dT(row + 2 * Dm, dv[kh], 1.0f, h);
        }
        vm_wait<0>();
        __builtin_amdgcn_s_barrier();
#undef HS_VSTEP
#undef HS_VPACK0
#undef HS_VPACK1
#undef HS_SB
#undef HS_STAMP
#undef HS_STAMP_OUT
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
// the one-sweep kernel runs only when the caller asks for it (HS_ATTN_ONE_SWEEP) and the shape fits: N = 256 * 2^k (+ 1), k <= 5
FusedPlan fused_plan(int B, int N, int H, int flags) {
    FusedPlan p{};
    if (!(flags & HS_ATTN_ONE_SWEEP) || (flags & HS_ATTN_F32)) return p;
    const int M = N / 256, rem = N % 256;
    if (M < 1 || M > 32 || (M & (M - 1)) || rem > 1) return p;
    if ((size_t)N * 3 * H * 64 * 2 >= HS_MAX_RECORDS) return p;
    p.M = M; p.rem = rem; p.nslice = (N + 31) / 32; p.ngang = 8 * (32 / M);
    p.mode = (flags & HS_ATTN_ONE_SWEEP_TILES) ? F_PARTIAL : F_CHAIN;
    const size_t BH = (size_t)B * H;
    p.off_chain = (3 * BH * N * 4 + 255) & ~(size_t)255;
    const size_t chain_bytes = BH * (p.mode == F_PARTIAL ? (size_t)M : (size_t)1) * 2048 * (size_t)p.nslice * 4;
    p.off_ctr = p.off_chain + chain_bytes;
    p.bytes = p.off_ctr + 256;
    p.ok = true;
    return p;
}
#ifdef HS_DEBUG_HOOKS
unsigned long long* g_fused_stamps = nullptr;
#endif
}  // namespace

#ifdef HS_DEBUG_HOOKS
// diagnostic (hook builds only): device buffer of 100 u64 that receives the phase stamps of iterations 40..47 of one wave
extern "C" void hs_debug_attn_stamps(unsigned long long* dev_buf) { g_fused_stamps = dev_buf; }
#endif

extern "C" size_t hs_attn_bwd_ws_bytes(int B, int N, int H, int flags) {
    if (B < 1 || N < 1 || H < 1) return 0;
    const size_t two_sweep = (size_t)B * H * N * 4 * 2;
    if (flags & HS_ATTN_F32) return two_sweep;
    const FusedPlan p = fused_plan(B, N, H, flags);
    return p.ok && p.bytes > two_sweep ? p.bytes : two_sweep;
}

extern "C" int hs_attn_bwd2(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, void* ws, size_t ws_bytes,
                            int B, int N, int H, int dh, int flags, void* stream) {
    if (!qkv || !o || !d_o || !lse || !dqkv || !ws) return HS_ERR_ARG;
    if (dh != 64 || B < 1 || N < 1 || H < 1) return HS_ERR_SHAPE;
    if (flags & ~(HS_ATTN_Q_PRESCALED | HS_ATTN_F32 | HS_ATTN_ONE_SWEEP | HS_ATTN_ONE_SWEEP_TILES)) return HS_ERR_ARG;
    if (((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dqkv | (uintptr_t)ws) & 15) return HS_ERR_SHAPE;
    if (ws_bytes < (size_t)B * H * N * 4 * 2) return HS_ERR_SHAPE;
    const FusedPlan p = fused_plan(B, N, H, flags);
    if (!p.ok || ws_bytes < p.bytes)
        return hs_attn_bwd(qkv, o, d_o, lse, dqkv, (float*)ws, B, N, H, dh, flags & (HS_ATTN_Q_PRESCALED | HS_ATTN_F32), stream);
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

// error word of the last one-sweep launch that used this workspace (0 = none; 1 = a bounded hand-off wait ran out: results void).
// Synchronises the stream; diagnostic / test use only.
extern "C" int hs_attn_bwd_ws_error(const void* ws, int B, int N, int H, int flags, void* stream) {
    if (!ws) return HS_ERR_ARG;
    const FusedPlan p = fused_plan(B, N, H, flags);
    if (!p.ok) return 0;
    unsigned v[2] = {0u, 0u};
    if (hipMemcpyAsync(v, (const char*)ws + p.off_ctr, 8, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return HS_ERR_LAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return HS_ERR_LAUNCH;
    return (int)v[1];
}
