// N2: CT volume preprocessing in front of the encoder path, as HBM-bound gather/stream kernels.
// Reference: Data/data_processing/CT-RATE/CT-RATE_nii_to_3D_volume_npy_file.py:25-124 (HU rescale + clamp, transpose
// to (D,H,W), trilinear respacing in float64, cast to float32, min-max, MONAI CropForeground, MONAI Resize to
// (32,256,256)) and the train-time augmentation LaMed/src/dataset/multi_dataset.py:148-158.
// Every kernel writes W-contiguous 4-B (or 8-B) elements per lane; gathers read at most 8 taps per output voxel.
#include "common.h"

// out[d][h][w] = clamp(slope * raw[h][w][d] + intercept, lo, hi) in float64 (the reference holds the volume in f64 up to
// the first interpolation).  32(w) x 32(d) tiles through LDS: reads are d-contiguous, writes w-contiguous.
template <typename T>
__global__ __launch_bounds__(256) void hu_window_kernel(const T* __restrict__ raw, double* __restrict__ out, int H, int W, int D,
                                                        double slope, double intercept, double lo, double hi) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int w0 = blockIdx.x * 32, d0 = blockIdx.y * 32, hh = blockIdx.z;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int w = w0 + ty + 8 * i, d = d0 + tx;
        if (w < W && d < D) {
            const double v = slope * (double)raw[((long)hh * W + w) * D + d] + intercept;
            tile[ty + 8 * i][tx] = fmin(fmax(v, lo), hi);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = d0 + ty + 8 * i, w = w0 + tx;
        if (w < W && d < D) out[((long)d * H + hh) * W + w] = tile[tx][ty + 8 * i];
    }
}

// source index / weight of torch's upsample_linear (align_corners = False): src = scale * (dst + 0.5) - 0.5, clamped at 0
template <typename A>
__device__ __forceinline__ void lin_coord(int dst, A scale, int in_size, int& i0, int& i1, A& l1) {
    A src = scale * ((A)dst + (A)0.5) - (A)0.5;
    if (src < (A)0) src = (A)0;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (A)i0;
}
// Trilinear resize of the window [c0, c0 + csize) of a (iD,iH,iW) volume to (oD,oH,oW); each tap is first mapped
// v -> (v - sub) / den (the min-max normalisation of the reference, applied to the taps so that the window test and the
// interpolation see exactly the normalised values).  A = arithmetic type (double for the respacing, float for the Resize).
template <typename TI, typename A>
__global__ __launch_bounds__(256) void resample_kernel(const TI* __restrict__ in, float* __restrict__ out, int iH, int iW, int cd0,
                                                       int ch0, int cw0, int cD, int cH, int cW, int oD, int oH, int oW, A sub, A den) {
    const long total = (long)oD * oH * oW;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int ow = (int)(g % oW);
    const long t = g / oW;
    const int oh = (int)(t % oH), od = (int)(t / oH);
    int d0, d1, h0, h1, w0, w1;
    A ld, lh, lw;
    lin_coord<A>(od, (A)cD / (A)oD, cD, d0, d1, ld);
    lin_coord<A>(oh, (A)cH / (A)oH, cH, h0, h1, lh);
    lin_coord<A>(ow, (A)cW / (A)oW, cW, w0, w1, lw);
    auto tap = [&](int d, int hh, int w) -> A {
        const A v = (A)in[((long)(cd0 + d) * iH + (ch0 + hh)) * iW + (cw0 + w)];
        return (v - sub) / den;
    };
    const A one = (A)1;
    const A r0 = (one - lh) * ((one - lw) * tap(d0, h0, w0) + lw * tap(d0, h0, w1)) + lh * ((one - lw) * tap(d0, h1, w0) + lw * tap(d0, h1, w1));
    const A r1 = (one - lh) * ((one - lw) * tap(d1, h0, w0) + lw * tap(d1, h0, w1)) + lh * ((one - lw) * tap(d1, h1, w0) + lw * tap(d1, h1, w1));
    out[g] = (float)((one - ld) * r0 + ld * r1);
}

// global min / max (out2 initialised to {+inf, -inf} by the caller): wave reduce, one atomic pair per wave.
__device__ __forceinline__ void atomic_min_f(float* a, float v) {
    // order-preserving integer view of IEEE floats
    if (v >= 0) atomicMin((int*)a, __float_as_int(v)); else atomicMax((unsigned*)a, __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f(float* a, float v) {
    if (v >= 0) atomicMax((int*)a, __float_as_int(v)); else atomicMin((unsigned*)a, __float_as_uint(v));
}
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, long n, float* __restrict__ out2) {
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { atomic_min_f(out2, mn); atomic_max_f(out2 + 1, mx); }
}

// bounding box of {x > thresh} (MONAI CropForeground, select_fn = x > 0 on the normalised volume <=> x > min):
// box6 = {min d, min h, min w, max d, max h, max w}, initialised to {D, H, W, -1, -1, -1}
__global__ __launch_bounds__(256) void bbox_kernel(const float* __restrict__ x, int D, int H, int W, float thresh, int* __restrict__ box6) {
    const long total = (long)D * H * W;
    int lo[3] = {D, H, W}, hi[3] = {-1, -1, -1};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        if (x[i] > thresh) {
            const int w = (int)(i % W);
            const long t = i / W;
            const int hh = (int)(t % H), d = (int)(t / H);
            lo[0] = min(lo[0], d); lo[1] = min(lo[1], hh); lo[2] = min(lo[2], w);
            hi[0] = max(hi[0], d); hi[1] = max(hi[1], hh); hi[2] = max(hi[2], w);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = max(hi[a], __shfl_xor(hi[a], o, 64));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(box6 + a, lo[a]); atomicMax(box6 + 3 + a, hi[a]); }
    }
}

// train-time augmentation, one gather: np.rot90(k, axes=(H,W)) -> flips (bit0 D, bit1 H, bit2 W) -> * scale -> + shift
// (MONAI RandRotate90 / RandFlip x3 / RandScaleIntensity / RandShiftIntensity with their random draws made on the host).
// in: (C, D, H, W); out: (C, D, Ho, Wo) with (Ho, Wo) = (W, H) for odd k.
__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int D, int H, int W,
                                                      int k, int flips, float scale, float shift) {
    const int Ho = (k & 1) ? W : H, Wo = (k & 1) ? H : W;
    const long total = (long)C * D * Ho * Wo;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    int j = (int)(g % Wo);
    long t = g / Wo;
    int i = (int)(t % Ho); t /= Ho;
    int d = (int)(t % D);
    const int c = (int)(t / D);
    if (flips & 1) d = D - 1 - d;
    if (flips & 2) i = Ho - 1 - i;
    if (flips & 4) j = Wo - 1 - j;
    // inverse of rot90: out_rot[i][j] = in[..] with  k=1: in[j][W-1-i];  k=2: in[H-1-i][W-1-j];  k=3: in[H-1-j][i]
    int hh, w;
    switch (k & 3) {
        case 1: hh = j; w = W - 1 - i; break;
        case 2: hh = H - 1 - i; w = W - 1 - j; break;
        case 3: hh = H - 1 - j; w = i; break;
        default: hh = i; w = j; break;
    }
    // two roundings like the reference's separate transforms: this file is built with -ffp-contract=off (bit-exact)
    out[g] = in[(((long)c * D + d) * H + hh) * W + w] * scale + shift;
}

extern "C" int hs_prep_hu_window(const void* raw, int raw_is_f64, double* out, int H, int W, int D, double slope, double intercept,
                                 double lo, double hi, void* stream) {
    if (!raw || !out) return HS_ERR_ARG;
    if (H < 1 || W < 1 || D < 1 || H > 65535) return HS_ERR_SHAPE;
    ProfScope prof(HS_PROF_OTHER, stream, 0, (double)H * W * D * ((raw_is_f64 ? 8 : 4) + 8));
    dim3 grid((W + 31) / 32, (D + 31) / 32, H);
    if (raw_is_f64) hipLaunchKernelGGL(hu_window_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double*)raw, out, H, W, D, slope, intercept, lo, hi);
    else hipLaunchKernelGGL(hu_window_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)raw, out, H, W, D, slope, intercept, lo, hi);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
extern "C" int hs_prep_resample(const void* in, int in_is_f64, float* out, int iD, int iH, int iW, int cd0, int ch0, int cw0, int cD,
                                int cH, int cW, int oD, int oH, int oW, float sub, float den, void* stream) {
    if (!in || !out) return HS_ERR_ARG;
    if (cD < 1 || cH < 1 || cW < 1 || oD < 1 || oH < 1 || oW < 1 || cd0 < 0 || ch0 < 0 || cw0 < 0 || cd0 + cD > iD || ch0 + cH > iH ||
        cw0 + cW > iW || den == 0.f) return HS_ERR_SHAPE;
    const long total = (long)oD * oH * oW;
    ProfScope prof(HS_PROF_OTHER, stream, 0, (double)total * 4 + (double)cD * cH * cW * (in_is_f64 ? 8 : 4));
    const unsigned nb = (unsigned)((total + 255) / 256);
    if (in_is_f64) hipLaunchKernelGGL((resample_kernel<double, double>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const double*)in, out, iH, iW, cd0, ch0, cw0, cD, cH, cW, oD, oH, oW, (double)sub, (double)den);
    else hipLaunchKernelGGL((resample_kernel<float, float>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)in, out, iH, iW, cd0, ch0, cw0, cD, cH, cW, oD, oH, oW, sub, den);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
extern "C" int hs_prep_minmax(const float* x, int64_t n, float* out2, void* stream) {
    if (!x || !out2) return HS_ERR_ARG;
    if (n < 1) return HS_ERR_SHAPE;
    ProfScope prof(HS_PROF_OTHER, stream, 0, (double)n * 4);
    const unsigned nb = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(minmax_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, (long)n, out2);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
extern "C" int hs_prep_foreground_bbox(const float* x, int D, int H, int W, float thresh, int* box6, void* stream) {
    if (!x || !box6) return HS_ERR_ARG;
    if (D < 1 || H < 1 || W < 1) return HS_ERR_SHAPE;
    const long n = (long)D * H * W;
    ProfScope prof(HS_PROF_OTHER, stream, 0, (double)n * 4);
    const unsigned nb = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(bbox_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, D, H, W, thresh, box6);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
extern "C" int hs_prep_augment(const float* in, float* out, int C, int D, int H, int W, int rot_k, int flip_mask, float scale, float shift,
                               void* stream) {
    if (!in || !out) return HS_ERR_ARG;
    if (C < 1 || D < 1 || H < 1 || W < 1 || rot_k < 0 || rot_k > 3 || (flip_mask & ~7)) return HS_ERR_SHAPE;
    const long n = (long)C * D * H * W;
    ProfScope prof(HS_PROF_OTHER, stream, 0, (double)n * 8);
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, C, D, H, W, rot_k, flip_mask, scale, shift);
    HS_CHECK_LAUNCH();
    return HS_OK;
}
