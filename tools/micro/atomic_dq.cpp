// Is an FA2-style fused attention backward viable on MI355X?  With keys stationary, every (128-key block, 32-query step) would add a
// 32 x 64 fp32 tile (8 KiB per wave) into dQ[b, head] with float atomics: N^2*d/32 bytes per head = 8 GB per launch at 20 volumes.
// This measures what the chip sustains for exactly that pattern: `blocks` four-wave blocks, each wave adding `steps` tiles of
// 32 rows x 256 B into its head's 1-MiB dQ array (rows walk the array like the query sweep does; `heads` arrays, block -> head by
// blockIdx % heads, so an array's writers sit on several XCDs as they would under plain dispatch; same_xcd=1: block -> head chosen so
// that all writers of an array share an XCD).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/atomic_dq.cpp -o tools/micro/atomic_dq && tools/micro/atomic_dq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void atomic_tiles(float* dq, int heads, int rows, int steps, int same_xcd, int mode) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int head;
    if (same_xcd) { const int x = blockIdx.x & 7, j = (blockIdx.x >> 3) % (heads / 8); head = x + 8 * j; }   // heads % 8 == 0
    else head = blockIdx.x % heads;
    float* base = dq + (long)head * rows * 64;
    const float v = 1.0f + lane * 0.001f;
    for (int s = 0; s < steps; ++s) {
        const int r0 = (s * 32) % (rows - 32);                // every block of a head sweeps the same query rows, like the real kernel
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
            float* p = base + (long)(r0 + ((r + wave * 8) & 31)) * 64 + lane;
            if (mode == 0) unsafeAtomicAdd(p, v);                                       // global_atomic_add_f32, no return
            else if (mode == 1) __builtin_nontemporal_store(v, p);                      // plain stores of the same bytes (yardstick)
            else *p = v;
        }
    }
}

int main(int argc, char** argv) {
    const int heads = 240, rows = 4097;
    float* dq;
    CK(hipMalloc(&dq, (size_t)heads * rows * 64 * 4));
    CK(hipMemset(dq, 0, (size_t)heads * rows * 64 * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 3; ++mode)
        for (int same = 0; same < 2; ++same)
            for (int blocks : {512, 2048, 7920}) {
                const int steps = 129;                        // a 4097-query sweep = 129 steps of 32
                hipLaunchKernelGGL(atomic_tiles, dim3(blocks), dim3(256), 0, 0, dq, heads, rows, 8, same, mode);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(atomic_tiles, dim3(blocks), dim3(256), 0, 0, dq, heads, rows, steps, same, mode);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                const double bytes = (double)blocks * 4 * steps * 32 * 256;
                printf("%-8s same_xcd=%d blocks=%5d: %8.3f ms  %8.1f GB/s of %s payload (%.2f GB)\n", mode == 0 ? "atomic" : mode == 1 ? "nt-store" : "store",
                       same, blocks, ms, bytes / ms / 1e6, mode == 0 ? "atomic" : "store", bytes / 1e9);
            }
    return 0;
}
