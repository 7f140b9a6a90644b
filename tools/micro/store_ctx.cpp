// Does the machine state of a GEMM block (139 KiB of LDS, ~200 VGPRs per wave, MFMAs just retired, data coming out of the
// accumulators) slow down the SAME 16 stores per wave that tools/micro/store_burst.cpp issues in ~1.8k cycles?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_ctx.cpp -o tools/micro/store_ctx && tools/micro/store_ctx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

template <int LDSB, bool MFMA, bool CVT, int PAT = 0, bool NT = true>
__global__ __launch_bounds__(512) void k(unsigned short* out, long ld, int tiles_n, unsigned long long* t_issue, unsigned long long* t_done, const bf16x8* in, int iters) {
    __shared__ char smem[LDSB > 0 ? LDSB : 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (LDSB > 0) smem[threadIdx.x] = (char)lane;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    out += (long)tn * 256;
    const long m0 = (long)tm * 256 + (wave >> 2) * 128;
    const int n0 = (wave & 3) * 64;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{(float)lane, 1.f, 2.f, (float)i};
    if (MFMA) {
        bf16x8 a[4], b[4];
        for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) % 4096]; b[i] = in[(threadIdx.x * 8 + 4 + i) % 4096]; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i & 3], acc[i][j], 0, 0, 0);
            asm volatile("" : "+v"(a[0]), "+v"(b[0]));
        }
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int l15 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            u32x4 o;
            if (CVT) {
                const f32x4 x = acc[i][2 * jp], y = acc[i][2 * jp + 1];
                o[0] = (__float_as_uint(x[0]) >> 16) | (__float_as_uint(x[1]) & 0xffff0000u); o[1] = (__float_as_uint(x[2]) >> 16) | (__float_as_uint(x[3]) & 0xffff0000u);
                o[2] = (__float_as_uint(y[0]) >> 16) | (__float_as_uint(y[1]) & 0xffff0000u); o[3] = (__float_as_uint(y[2]) >> 16) | (__float_as_uint(y[3]) & 0xffff0000u);
            } else o = u32x4{(unsigned)lane, 2u, 3u, (unsigned)i};
            // PAT 0: 16 rows x 64 B per instruction (the transposed-accumulator layout); PAT 1: 8 rows x 128 B (lanes l15 >= 8 take the
            // other half of the line of row l15 - 8)
            long m = m0 + 16 * i + l15; int n = n0 + 32 * jp + 8 * q;
            if (PAT == 1) { m = m0 + 16 * i + 8 * jp + (l15 & 7); n = n0 + 32 * (l15 >> 3) + 8 * q; }
            if (NT) __builtin_nontemporal_store(o, (u32x4*)(out + m * ld + n)); else *(u32x4*)(out + m * ld + n) = o;
        }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { t_issue[blockIdx.x] = t1 - t0; t_done[blockIdx.x] = t2 - t0; }
    if (LDSB > 0 && smem[(threadIdx.x * 7) % 512] == 99) out[0] = 1;
}

template <int LDSB, bool MFMA, bool CVT, int PAT = 0, bool NT = true>
void run(const char* name, unsigned short* out, unsigned long long* ti, unsigned long long* td, const bf16x8* in, int iters) {
    std::vector<unsigned long long> hi(4096), hd(4096);
    for (int nb : {256, 2880}) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL((k<LDSB, MFMA, CVT, PAT, NT>), dim3(nb), dim3(512), 0, 0, out, 2304l, 9, ti, td, in, iters);
            CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, a, b));
        }
        CK(hipMemcpy(hi.data(), ti, nb * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hd.data(), td, nb * 8, hipMemcpyDeviceToHost));
        double si = 0, sd = 0;
        for (int i = 0; i < nb; ++i) { si += hi[i]; sd += hd[i]; }
        printf("%-44s blocks %4d: issue avg %7.0f | acked avg %7.0f ticks | kernel %7.1f us\n", name, nb, si / nb, sd / nb, ms * 1e3);
    }
}
int main() {
    unsigned short* out; CK(hipMalloc(&out, (size_t)4096 * 65536 * 2));
    unsigned long long *ti, *td; CK(hipMalloc(&ti, 4096 * 8)); CK(hipMalloc(&td, 4096 * 8));
    bf16x8* in; CK(hipMalloc(&in, 4096 * 16)); CK(hipMemset(in, 0x3c, 4096 * 16));
    run<139264, false, true, 0, true>("16 rows x 64 B, nontemporal", out, ti, td, in, 0);
    run<139264, false, true, 0, false>("16 rows x 64 B, plain stores", out, ti, td, in, 0);
    run<139264, false, true, 1, true>("8 rows x 128 B, nontemporal", out, ti, td, in, 0);
    run<139264, false, true, 1, false>("8 rows x 128 B, plain stores", out, ti, td, in, 0);
    run<139264, true, true, 0, true>("16 x 64 B nt, 24 MFMA k-steps before", out, ti, td, in, 24);
    run<139264, true, true, 1, true>("8 x 128 B nt, 24 MFMA k-steps before", out, ti, td, in, 24);
    run<139264, true, true, 1, false>("8 x 128 B plain, 24 MFMA k-steps before", out, ti, td, in, 24);
    return 0;
}
