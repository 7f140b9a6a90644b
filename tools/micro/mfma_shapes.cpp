// Bare bf16 MFMA loops, operands in registers, random data: which shape does the chip run faster (clock included)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shapes.cpp -o tools/micro/mfma_shapes && tools/micro/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// wave tile 128x64 per K=16 step: 8 x 32x32x16 MFMAs (4 A fragments x 2 B fragments), waves per SIMD = W
template <int ITER>
__global__ __launch_bounds__(512, 2) void k32(const bf16x8* __restrict__ in, float* __restrict__ out) {
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = in[threadIdx.x * 6 + i];
    for (int i = 0; i < 2; ++i) b[i] = in[threadIdx.x * 6 + 4 + i];
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
// same 128x64 wave tile with 16x16x32: 8 x 4 tiles, K = 32 per MFMA -> 32 MFMAs per K=32 step = same FLOPs as 2 steps above
template <int ITER>
__global__ __launch_bounds__(512, 2) void k16(const bf16x8* __restrict__ in, float* __restrict__ out) {
    bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = in[(threadIdx.x * 12 + i) % 3072];
    for (int i = 0; i < 4; ++i) b[i] = in[(threadIdx.x * 12 + 8 + i) % 3072];
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < ITER / 2; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    const int ITER = 4096, NB = 256 * 8;
    std::vector<unsigned short> h(3072 * 8);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    bf16x8* in; float* out;
    CK(hipMalloc(&in, h.size() * 2)); CK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, NB * 512 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = (double)NB * 8 /*waves*/ * ITER * 8 /*mfma*/ * 2.0 * 32 * 32 * 16;
    for (int round = 0; round < 4; ++round) {
        for (int which = 0; which < 2; ++which) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) {
                if (which == 0) hipLaunchKernelGGL(k32<ITER>, dim3(NB), dim3(512), 0, 0, in, out);
                else hipLaunchKernelGGL(k16<ITER>, dim3(NB), dim3(512), 0, 0, in, out);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("round %d %s: %8.3f ms  %7.1f TFLOP/s\n", round, which == 0 ? "32x32x16" : "16x16x32", ms / 5, flops / (ms / 5 * 1e-3) / 1e12);
        }
    }
    return 0;
}
