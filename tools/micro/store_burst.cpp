// How fast can ONE compute unit get a GEMM tile's epilogue out, and what happens when all 256 do it at once?
// Each block = 8 waves storing a 256x256 bf16 tile (128 KiB) as 16-B pieces, 8 rows of 128 contiguous bytes per wave
// instruction (the layout of gemm.hip's epilogue).  Reports s_memtime cycles until the stores are ISSUED and until they
// are acknowledged (vmcnt 0), per block, for grids of 8 .. 2048 blocks.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_burst.cpp -o tools/micro/store_burst && tools/micro/store_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// tiles_n > 0: the block is tile (blockIdx / tiles_n, blockIdx % tiles_n) of a row-major [M][ld] matrix (GEMM placement);
// wide: every wave instruction covers 2 rows x 512 B (wave w owns rows 32w..32w+31 of the tile) instead of 8 rows x 128 B
__global__ __launch_bounds__(512) void burst(unsigned short* out, long ld, unsigned long long* t_issue, unsigned long long* t_done, int spin, int tiles_n, int wide, int nt, const u32x4* src, int rd_kb, long src_n) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 2, wn = wave & 3;
    const int tm = tiles_n ? blockIdx.x / tiles_n : blockIdx.x, tn = tiles_n ? blockIdx.x % tiles_n : 0;
    out += (long)tn * 256;
    long m0 = (long)tm * 256 + wm * 128;
    int n0 = wn * 64 + (lane & 7) * 8;
    int rstep = 8, rlane = lane >> 3;
    if (wide) { m0 = (long)tm * 256 + wave * 32; n0 = (lane & 31) * 8; rstep = 2; rlane = lane >> 5; }
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    // optional busy phase before the burst (de-phases nothing: every block spins the same), keeps clocks up
    float f = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) f = f * 1.0001f + 0.5f;
    if (rd_kb) {   // GEMM-like operand traffic: blocks of one 9-tile row read the same A panel (L2 hits), everyone reads the B panel
        unsigned acc = 0;
        const long a0 = ((long)tm * 24576) % src_n, b0 = ((long)tn * 24576 + 7777777) % src_n;
        for (int i = threadIdx.x; i < rd_kb * 32; i += 512) { acc += src[(a0 + i) % src_n][0]; acc += src[(b0 + i) % src_n][1]; }
        if (acc == 0x12345u) v[1] = acc;
    }
    if (f == 12345.678f) v[0] = 7u;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const long m = m0 + s * rstep + rlane;
        if (nt) __builtin_nontemporal_store(v, (u32x4*)(out + m * ld + n0)); else *(u32x4*)(out + m * ld + n0) = v;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { t_issue[blockIdx.x] = t1 - t0; t_done[blockIdx.x] = t2 - t0; }
}

__global__ void dirty_fill(u32x4* p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = u32x4{1u, 2u, 3u, (unsigned)i};
}
int main() {
    u32x4* scratch; const long nscr = (1l << 30) / 16; CK(hipMalloc(&scratch, nscr * 16));
    const int maxb = 4096;
    unsigned short* out; CK(hipMalloc(&out, (size_t)4096 * 256 * 256 * 2));
    unsigned long long *ti, *td; CK(hipMalloc(&ti, maxb * 8)); CK(hipMalloc(&td, maxb * 8));
    std::vector<unsigned long long> hi(maxb), hd(maxb);
    struct Cfg { long ld; int tiles_n, wide, nt, dirty, spin, rd_kb; } cfgs[] = {{2304, 9, 0, 0, 0, 1500, 0}, {2304, 9, 1, 0, 0, 1500, 0}, {2304, 9, 0, 1, 0, 1500, 0},
                                                                                 {2304, 9, 1, 1, 0, 1500, 0}, {2304, 9, 0, 1, 0, 1500, 384}, {2304, 9, 1, 1, 0, 1500, 384}};
    for (const Cfg& c : cfgs) {
        const long ld = c.ld; const int spin = c.spin;
        printf("ld %ld tiles_n %d wide %d nontemporal %d caches pre-filled with dirty lines %d\n", c.ld, c.tiles_n, c.wide, c.nt, c.dirty); printf("  spin %d, reads %d KiB per operand per block\n", c.spin, c.rd_kb);
        for (int nb : {32, 256, 2880}) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
                if (c.dirty) hipLaunchKernelGGL(dirty_fill, dim3(2048), dim3(256), 0, 0, scratch, nscr);
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(burst, dim3(nb), dim3(512), 0, 0, out, ld, ti, td, spin, c.tiles_n, c.wide, c.nt, scratch, c.rd_kb, nscr);
                CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (rep < 2) continue;
                CK(hipMemcpy(hi.data(), ti, nb * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hd.data(), td, nb * 8, hipMemcpyDeviceToHost));
                double si = 0, sd = 0; unsigned long long mi = 0, md = 0;
                for (int i = 0; i < nb; ++i) { si += hi[i]; sd += hd[i]; mi = std::max(mi, hi[i]); md = std::max(md, hd[i]); }
                printf("spin %5d blocks %4d: issue avg %7.0f max %7llu | acked avg %7.0f max %7llu (memtime ticks) | kernel %.1f us = %.2f TB/s\n", spin, nb,
                       si / nb, mi, sd / nb, md, ms * 1e3, nb * 131072.0 / (ms * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
