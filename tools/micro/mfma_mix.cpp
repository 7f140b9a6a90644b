// Does the 16x16x32 MFMA shape still beat 32x32x16 at the power cap when the loop looks like an attention sweep rather than a bare
// MFMA loop?  Per 32-cycle slice of matrix-pipe work (one 32x32x16 or two 16x16x32 MFMAs) each wave also issues one 1-KiB LDS
// fragment read (which becomes the next A operand: real data, toggling like real activations), two v_exp_f32, one v_mul_f32 and
// half a v_cvt_pk_bf16_f32 -- the dK/dV kernel's ratios.  Two waves per SIMD, all 256 CUs.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_mix.cpp -o tools/micro/mfma_mix && tools/micro/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SHAPE, int MIX>
__global__ __launch_bounds__(512, 1) void mix(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
    __shared__ u32x4 lds[2048];                                   // 32 KiB of random bf16
    for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = __builtin_bit_cast(bf16x8, in[(threadIdx.x * 8 + i) & 2047]); b[i] = __builtin_bit_cast(bf16x8, in[(threadIdx.x * 8 + 4 + i) & 2047]); }
    f32x16 acc32[4];
    f32x4 acc16[4][4];
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f; for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f; }
    float x[8];
    for (int r = 0; r < 8; ++r) x[r] = -0.01f * (lane + r);
    unsigned ofs = (wave * 64 + lane) & 2047;
    for (int it = 0; it < iters; ++it) {
        u32x4 nxt[4];
        if (MIX & 1) {                                              // next iteration's four A fragments: 4 x 1 KiB per wave, in flight
#pragma unroll                                                      // behind this iteration's MFMAs (waited for at the bottom)
            for (int i = 0; i < 4; ++i) { ofs = (ofs + 67) & 2047; nxt[i] = lds[ofs]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                               // 4 slices of 32 matrix-pipe cycles
            if (SHAPE == 32) {
                acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], acc32[i], 0, 0, 0);
            } else {
                acc16[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[i], acc16[i][0], 0, 0, 0);
                acc16[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[(i + 1) & 3], acc16[i][1], 0, 0, 0);
            }
            if (MIX & 2) {
                x[2 * i] = __builtin_amdgcn_exp2f(x[2 * i]) - 1.5f;
                x[2 * i + 1] = __builtin_amdgcn_exp2f(x[2 * i + 1]) * x[2 * i];
            }
        }
        if (MIX & 2) {                                              // 2 packs per 4 slices
            unsigned p0, p1;
            asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5" : "=v"(p0), "=v"(p1) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
            u32x4 t = __builtin_bit_cast(u32x4, b[3]);
            t[0] ^= (p0 ^ p1) & 0x00010001u;
            b[3] = __builtin_bit_cast(bf16x8, t);
        }
        if (MIX & 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, nxt[i]);
        }
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) s += acc32[i][r]; for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc16[i][j][r]; }
    for (int r = 0; r < 8; ++r) s += x[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int SHAPE, int MIX> double run(const u32x4* in, float* out, int iters, int nb) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mix<SHAPE, MIX>), dim3(nb), dim3(512), 0, 0, in, out, iters / 8);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 6; ++r) hipLaunchKernelGGL((mix<SHAPE, MIX>), dim3(nb), dim3(512), 0, 0, in, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 6;
}
int main() {
    const int iters = 16384, nb = 256 * 4;
    std::vector<unsigned short> h(2048 * 8);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    u32x4* in; float* out;
    CK(hipMalloc(&in, h.size() * 2)); CK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, nb * 512 * 4));
    const double flops = (double)nb * 8 * iters * 4 * 2.0 * 32 * 32 * 16;
    for (int round = 0; round < 2; ++round) {
        const double t[8] = {run<32, 0>(in, out, iters, nb), run<16, 0>(in, out, iters, nb), run<32, 1>(in, out, iters, nb), run<16, 1>(in, out, iters, nb),
                             run<32, 2>(in, out, iters, nb), run<16, 2>(in, out, iters, nb), run<32, 3>(in, out, iters, nb), run<16, 3>(in, out, iters, nb)};
        const char* names[8] = {"32x32x16 bare", "16x16x32 bare", "32x32x16 + LDS reads", "16x16x32 + LDS reads", "32x32x16 + exp/mul/pack",
                                "16x16x32 + exp/mul/pack", "32x32x16 + LDS reads + exp/mul/pack", "16x16x32 + LDS reads + exp/mul/pack"};
        for (int i = 0; i < 8; ++i) printf("round %d %-38s %8.3f ms  %7.1f TFLOP/s\n", round, names[i], t[i], flops / (t[i] * 1e-3) / 1e12);
    }
    return 0;
}
