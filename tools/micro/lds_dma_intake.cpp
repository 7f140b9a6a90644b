// What can ONE compute unit take in through LDS-DMA (buffer_load_dwordx4 ... lds) when the data sits in its XCD's L2 -- the operand
// path of the GEMM kernels?  8 waves per block (one block per CU, 128 KiB of LDS as 2 x 64 KiB stages like gemm.hip's big tile), each
// wave issues 8 x 1-KiB pieces per K-tile into the stage the block is not "reading", waits with a counted vmcnt + barrier, repeats.
// No MFMAs, no LDS reads: this is the intake ceiling alone.  Source = a per-block window of `span` KiB swept cyclically (small span: L2
// hits after the first pass; large span: every piece comes from beyond the L2).  Reports B/clk per CU (s_memtime ticks) and TB/s.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_dma_intake.cpp -o tools/micro/lds_dma_intake && tools/micro/lds_dma_intake
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define LDS_PTR(T) __attribute__((address_space(3))) T*

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// shared != 0: every block of an XCD (blockIdx % 8) sweeps the SAME window (operand panels shared by the tiles of a row/column group)
__global__ __launch_bounds__(512) void intake(const char* src, long span_bytes, int ktiles, int shared, unsigned long long* ticks) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 65536];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long win = shared ? (long)(blockIdx.x & 7) : (long)blockIdx.x;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src + win * span_bytes, (unsigned)span_bytes);
    const unsigned tiles_in_span = (unsigned)(span_bytes / 65536);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < ktiles; ++kt) {
        char* st = smem + (kt & 1) * 65536;
        const unsigned so = (unsigned)(kt % tiles_in_span) * 65536u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned piece = j * 8 + wave;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))(st + piece * 1024), 16, piece * 1024 + lane * 16, so, 0, 0);
        }
        if (kt > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the previous tile has landed; this one stays in flight
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    const long total = 4l << 30;
    char* src; CK(hipMalloc(&src, total)); CK(hipMemset(src, 1, total));
    unsigned long long* ticks; CK(hipMalloc(&ticks, 4096 * 8));
    std::vector<unsigned long long> h(4096);
    const int nb = 256, ktiles = 512;
    struct Cfg { long span_kib; int shared; const char* what; } cfgs[] = {
        {128, 0, "own 128-KiB window per block (L2 hits after the first pass)"},
        {1024, 1, "1-MiB window shared by the 32 blocks of an XCD (operand panels of a tile group: L2 hits)"},
        {4096, 1, "4-MiB window shared per XCD (the L2's size)"},
        {8192, 1, "8-MiB window shared per XCD (2x the L2; 64 MiB in all: Infinity-Cache resident)"},
        {16384, 1, "16-MiB window shared per XCD (128 MiB in all: Infinity-Cache resident)"},
        {512, 0, "own 512-KiB window per block (16 MiB per XCD, 128 MiB in all: Infinity-Cache resident, nothing shared)"},
        {16384, 0, "own 16-MiB window per block (4 GiB in all: beyond L2 and Infinity Cache)"}};
    for (const Cfg& c : cfgs) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(intake, dim3(nb), dim3(512), 0, 0, src, c.span_kib * 1024, ktiles, c.shared, ticks);
            CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep < 2) continue;
            CK(hipMemcpy(h.data(), ticks, nb * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.begin() + nb);
            const double bytes = (double)ktiles * 65536;
            printf("%-92s: median %6.1f B/tick per CU (p10 %.1f, p90 %.1f), kernel %.1f us = %.2f TB/s chip-wide\n", c.what, bytes / h[nb / 2],
                   bytes / h[nb * 9 / 10], bytes / h[nb / 10], ms * 1e3, nb * bytes / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
