// Standalone micro-bench + A/B checker for the hot kernels, through the C ABI only (no torch: starts in < 1 s).
//   build : hipcc --offload-arch=gfx950 -O2 tools/kbench.cpp -o tools/kbench -ldl        (tools/build_kbench.sh)
//   run   : tools/kbench attnf <B> <reps> <variantA> <variantB>     forward attention, A/B interleaved
//           tools/kbench attnb <B> <reps> <variantA> <variantB>     backward attention
//           tools/kbench gemm  <B> <reps> [filter]                  the ViT-B layer's GEMM shapes
// Variants are selected with hs_debug_set(key, value) (diagnostic export, not part of include/hsenet_hip.h);
// variant A is the reference the outputs of B are compared against (both are GPU kernels: this is a regression
// screen for kernel work, the parity tests proper are tests/test_gpu_*.py).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include "../include/hsenet_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static void* g_lib;
static int g_variant = 0;
template <typename T> static T sym(const char* n) {
    void* p = dlsym(g_lib, n);
    if (!p) { fprintf(stderr, "missing symbol %s\n", n); exit(2); }
    return (T)p;
}

static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }
static inline float gauss() {   // sum of 4 uniforms, unit variance
    float s = 0;
    uint64_t r = rnd();
    for (int i = 0; i < 4; ++i) s += (float)((r >> (16 * i)) & 0xffff) * (1.0f / 65536.0f);
    return (s - 2.0f) * 1.7320508f;
}
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

static void* dev_bf16_gauss(size_t n, float std_) {
    std::vector<uint16_t> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2bf(gauss() * std_);
    void* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}
static void* dev_zero(size_t bytes) { void* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes)); return d; }

struct Cmp { double rel, maxabs; };
static Cmp cmp_bf16(const void* a, const void* b, size_t n) {
    std::vector<uint16_t> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, n * 2, hipMemcpyDeviceToHost));
    double num = 0, den = 0, mx = 0; size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const double x = bf2f(ha[i]), y = bf2f(hb[i]);
        if (!(std::isfinite(x) && std::isfinite(y))) { ++bad; continue; }
        num += (x - y) * (x - y); den += x * x; mx = std::max(mx, std::fabs(x - y));
    }
    if (bad) printf("   !! %zu non-finite elements\n", bad);
    return {std::sqrt(num / std::max(den, 1e-30)), mx};
}
static Cmp cmp_f32(const void* a, const void* b, size_t n) {
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0, mx = 0; size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const double x = ha[i], y = hb[i];
        if (!(std::isfinite(x) && std::isfinite(y))) { ++bad; continue; }
        num += (x - y) * (x - y); den += x * x; mx = std::max(mx, std::fabs(x - y));
    }
    if (bad) printf("   !! %zu non-finite elements\n", bad);
    return {std::sqrt(num / std::max(den, 1e-30)), mx};
}

template <typename F> static double time_us(F&& f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms * 1e3 / reps;
}

typedef int (*attn_fwd_t)(const void*, void*, float*, int, int, int, int, int, int, void*);
typedef int (*attn_bwd_t)(const void*, const void*, const void*, const float*, void*, float*, int, int, int, int, int, void*);
typedef int (*attn_bwd2_t)(const void*, const void*, const void*, const float*, void*, void*, size_t, int, int, int, int, int, void*);
typedef size_t (*attn_ws_t)(int, int, int, int);
typedef int (*attn_err_t)(const void*, int, int, int, int, void*);
typedef int (*gemm_t)(const hs_gemm_args*, int, void*);
typedef void (*dbg_t)(int, int);

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: kbench attnf|attnb|gemm ...\n"); return 1; }
    std::string exe = argv[0];
    std::string dir = exe.substr(0, exe.find_last_of('/'));
    std::string libp = dir + "/../hsenet_amd/lib/libhsenet_hip.so";
    if (getenv("HS_LIB")) libp = getenv("HS_LIB");
    g_lib = dlopen(libp.c_str(), RTLD_NOW);
    if (!g_lib) { fprintf(stderr, "dlopen %s: %s\n", libp.c_str(), dlerror()); return 2; }
    auto dbg = sym<dbg_t>("hs_debug_set");
    if (getenv("HS_DBG4")) dbg(4, atoi(getenv("HS_DBG4")));
    const std::string mode = argv[1];
    const int B = argc > 2 ? atoi(argv[2]) : 4, reps = argc > 3 ? atoi(argv[3]) : 10;
    const int N = getenv("HS_N") ? atoi(getenv("HS_N")) : 4097, H = 12, Dm = 768;
    CK(hipSetDevice(0));

    if (mode == "attnf" || mode == "attnb") {
        const int vA = argc > 4 ? atoi(argv[4]) : 0, vB = argc > 5 ? atoi(argv[5]) : 0;
        const float std_ = getenv("HS_STD") ? atof(getenv("HS_STD")) : 1.0f;
        const int aflags = getenv("HS_PRESCALED") ? HS_ATTN_Q_PRESCALED : 0;   // timing only: the fp64 check assumes raw q
        const size_t nq = (size_t)B * N * 3 * Dm, no = (size_t)B * N * Dm, nl = (size_t)B * H * N;
        void* qkv = dev_bf16_gauss(nq, std_);
        auto fwd = sym<attn_fwd_t>("hs_attn_fwd");
        void* o[2] = {dev_zero(no * 2), dev_zero(no * 2)};
        float* lse[2] = {(float*)dev_zero(nl * 4), (float*)dev_zero(nl * 4)};
        const double fl_f = 4.0 * B * H * (double)N * N * 64;
        const int key = mode == "attnf" ? 0 : 1;
        int var[2] = {vA, vB};
        if (mode == "attnf") {
            for (int i = 0; i < 2; ++i) { dbg(key, var[i]); int rc = fwd(qkv, o[i], lse[i], B, N, H, 64, HS_PREC_BF16, aflags, 0); if (rc) { printf("variant %d rc=%d\n", var[i], rc); return 3; } }
            CK(hipDeviceSynchronize());
            Cmp c = cmp_bf16(o[0], o[1], no), cl = cmp_f32(lse[0], lse[1], nl);
            printf("attn fwd B=%d N=%d: variant %d vs %d: o rel %.3e maxabs %.3e | lse rel %.3e maxabs %.3e\n", B, N, vA, vB, c.rel, c.maxabs, cl.rel, cl.maxabs);
            {   // fp64 host reference on sampled (b, head, query) rows, from the same bf16 inputs
                std::vector<uint16_t> hq(nq), ho[2] = {std::vector<uint16_t>(no), std::vector<uint16_t>(no)};
                std::vector<float> hl[2] = {std::vector<float>(nl), std::vector<float>(nl)};
                CK(hipMemcpy(hq.data(), qkv, nq * 2, hipMemcpyDeviceToHost));
                for (int i = 0; i < 2; ++i) { CK(hipMemcpy(ho[i].data(), o[i], no * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl[i].data(), lse[i], nl * 4, hipMemcpyDeviceToHost)); }
                double num[2] = {0, 0}, den = 0, lerr[2] = {0, 0};
                std::vector<double> sc(N);
                for (int smp = 0; smp < 96; ++smp) {
                    const int bb = rnd() % B, hh = rnd() % H, qi = smp < 8 ? (smp < 4 ? smp : N - 1 - (smp - 4)) : rnd() % N;
                    const uint16_t* qp = &hq[((size_t)bb * N + qi) * 3 * Dm + hh * 64];
                    double mx = -1e300;
                    for (int k = 0; k < N; ++k) {
                        const uint16_t* kp = &hq[((size_t)bb * N + k) * 3 * Dm + Dm + hh * 64];
                        double d = 0; for (int e = 0; e < 64; ++e) d += (double)bf2f(qp[e]) * bf2f(kp[e]);
                        sc[k] = d * 0.125; mx = std::max(mx, sc[k]);
                    }
                    double lsum = 0, od[64] = {0};
                    for (int k = 0; k < N; ++k) {
                        const double pk = std::exp(sc[k] - mx); lsum += pk;
                        const uint16_t* vp = &hq[((size_t)bb * N + k) * 3 * Dm + 2 * Dm + hh * 64];
                        for (int e = 0; e < 64; ++e) od[e] += pk * bf2f(vp[e]);
                    }
                    for (int e = 0; e < 64; ++e) {
                        const double r = od[e] / lsum; den += r * r;
                        for (int i = 0; i < 2; ++i) { const double g = bf2f(ho[i][((size_t)bb * N + qi) * Dm + hh * 64 + e]); num[i] += (g - r) * (g - r); }
                    }
                    for (int i = 0; i < 2; ++i) lerr[i] = std::max(lerr[i], std::fabs(hl[i][((size_t)bb * H + hh) * N + qi] - (mx + std::log(lsum))));
                }
                printf("  vs fp64 reference (96 rows): variant %d rel %.3e lse maxabs %.2e | variant %d rel %.3e lse maxabs %.2e\n", vA, std::sqrt(num[0] / den), lerr[0], vB, std::sqrt(num[1] / den), lerr[1]);
            }
            for (int round = 0; round < 3; ++round)
                for (int i = 0; i < 2; ++i) {
                    dbg(key, var[i]);
                    const double us = time_us([&] { fwd(qkv, o[i], lse[i], B, N, H, 64, HS_PREC_BF16, aflags, 0); }, reps);
                    printf("  round %d variant %d: %9.1f us  %7.1f TFLOP/s\n", round, var[i], us, fl_f / us / 1e6);
                }
        } else {
            auto bwd1 = sym<attn_bwd_t>("hs_attn_bwd");
            // the retired one-sweep kernel: present only in tools/experiments/attn_bwd_one_sweep/libhsenet_one_sweep.so (HS_LIB=...)
            auto bwd2 = (attn_bwd2_t)dlsym(g_lib, "hs_attn_bwd2");
            auto wsb = (attn_ws_t)dlsym(g_lib, "hs_attn_bwd_ws_bytes");
            auto wserr = (attn_err_t)dlsym(g_lib, "hs_attn_bwd_ws_error");
            const int kOneSweep = 4, kOneSweepTiles = 8;
            if ((vA >= 100 || vB >= 100) && (!bwd2 || !wsb || !wserr)) { fprintf(stderr, "variant >= 100 needs HS_LIB=tools/experiments/attn_bwd_one_sweep/libhsenet_one_sweep.so\n"); return 2; }
            const int oflags = aflags | kOneSweep | (getenv("HS_TILES") ? kOneSweepTiles : 0);
            const size_t ws_bytes = wsb ? wsb(B, N, H, oflags) : (size_t)B * H * N * 8;
            void* ws2 = dev_zero(ws_bytes);
            printf("attn bwd workspace: %.1f MB (two-sweep scratch %.1f MB)\n", ws_bytes / 1e6, (double)B * H * N * 8 / 1e6);
            // variant >= 100: hs_attn_bwd2 with HS_ATTN_ONE_SWEEP (HS_TILES=1 in the environment: the per-key-block tile form); else hs_attn_bwd
            auto bwd = [&](const void* q_, const void* o_, const void* g_, const float* l_, void* out_, float* dl_, int B_, int N_, int H_, int dh_, int fl_, void* st_) -> int {
                if (g_variant >= 100) return bwd2(q_, o_, g_, l_, out_, ws2, ws_bytes, B_, N_, H_, dh_, oflags, st_);
                return bwd1(q_, o_, g_, l_, out_, dl_, B_, N_, H_, dh_, fl_, st_);
            };
            dbg(0, 0);
            fwd(qkv, o[0], lse[0], B, N, H, 64, HS_PREC_BF16, aflags, 0);
            void* d_o = dev_bf16_gauss(no, 1.0f);
            void* dqkv[2] = {dev_zero(nq * 2), dev_zero(nq * 2)};
            float* delta = (float*)dev_zero(nl * 4 * 2 + (size_t)B * N * Dm * 4);
            for (int i = 0; i < 2; ++i) { g_variant = var[i]; dbg(1, var[i] % 10); dbg(2, (var[i] / 10) % 10); int rc = bwd(qkv, o[0], d_o, lse[0], dqkv[i], delta, B, N, H, 64, aflags, 0); if (rc) { printf("variant %d rc=%d\n", var[i], rc); return 3; } }
            CK(hipDeviceSynchronize());
            if (vA >= 100 || vB >= 100) printf("one-sweep error word: %d\n", wserr(ws2, B, N, H, oflags, 0));
            if (void* stp = dlsym(g_lib, "hs_debug_attn_stamps")) {        // hook builds: phase stamps of the one-sweep kernel
                unsigned long long* dbuf = (unsigned long long*)dev_zero(100 * 8);
                ((void (*)(unsigned long long*))stp)(dbuf);
                g_variant = 100; bwd(qkv, o[0], d_o, lse[0], dqkv[1], delta, B, N, H, 64, aflags, 0); CK(hipDeviceSynchronize());
                unsigned long long hs[100]; CK(hipMemcpy(hs, dbuf, sizeof hs, hipMemcpyDeviceToHost));
                printf("hand-offs %llu, of which %llu re-read their granules (%llu re-reads)\n", hs[97], hs[96], hs[98]);
                ((void (*)(unsigned long long*))stp)(nullptr);
                printf("one-sweep stamps, cycles: wait | barrier | C + 3 DMA | A0+V1 | A1 + hand-off | B reads | 8 MFMA | 8 MFMA + W | lgkm0 | - | - | total\n");
                for (int it = 0; it < 8; ++it) {
                    const unsigned long long* t = hs + it * 12;
                    printf("   iter %d:", 40 + it);
                    for (int k = 1; k < 12; ++k) printf(" %5lld", (long long)(t[k] - t[k - 1]));
                    printf(" | %6lld\n", (long long)(t[11] - t[0]));
                }
            }
            Cmp c = cmp_bf16(dqkv[0], dqkv[1], nq);
            printf("attn bwd B=%d N=%d: variant %d vs %d: dqkv rel %.3e maxabs %.3e\n", B, N, vA, vB, c.rel, c.maxabs);
            {   // per section (dq | dk | dv)
                std::vector<uint16_t> ha(nq), hb(nq);
                CK(hipMemcpy(ha.data(), dqkv[0], nq * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), dqkv[1], nq * 2, hipMemcpyDeviceToHost));
                double num[3] = {0, 0, 0}, den[3] = {0, 0, 0};
                for (size_t i = 0; i < nq; ++i) { const int sec = (int)((i % (3 * Dm)) / Dm); const double x = bf2f(ha[i]), y = bf2f(hb[i]); num[sec] += (x - y) * (x - y); den[sec] += x * x; }
                printf("   dq rel %.3e | dk rel %.3e | dv rel %.3e\n", std::sqrt(num[0] / den[0]), std::sqrt(num[1] / den[1]), std::sqrt(num[2] / den[2]));
            }
            for (int round = 0; round < 3; ++round)
                for (int i = 0; i < 2; ++i) {
                    g_variant = var[i]; dbg(1, var[i] % 10); dbg(2, (var[i] / 10) % 10);      // variant = 10 * (dK/dV kernel) + (dQ kernel); >= 100: hs_attn_bwd2
                    const double us = time_us([&] { bwd(qkv, o[0], d_o, lse[0], dqkv[i], delta, B, N, H, 64, aflags, 0); }, reps);
                    printf("  round %d variant %d: %9.1f us  %7.1f TFLOP/s (algorithmic 10 N^2 d)\n", round, var[i], us, 2.5 * fl_f / us / 1e6);
                }
        }
        return 0;
    }
    if (mode == "gemm") {
        auto gemm = sym<gemm_t>("hs_gemm");
        const char* filt = argc > 4 ? argv[4] : "";
        if (getenv("HS_GEMM_DBG")) dbg(3, atoi(getenv("HS_GEMM_DBG")));
        const int M = B * N;
        struct Sh { const char* name; int lay, m, n, k; int flags; };
        const Sh shapes[] = {
            {"qkv NT", HS_GEMM_NT, M, 2304, 768, 0}, {"proj NT", HS_GEMM_NT, M, 768, 768, 0}, {"fc1 NT", HS_GEMM_NT, M, 3072, 768, 0},
            {"fc2 NT", HS_GEMM_NT, M, 768, 3072, 0}, {"dgrad fc2 NN", HS_GEMM_NN, M, 3072, 768, 0}, {"dgrad fc1 NN", HS_GEMM_NN, M, 768, 3072, 0},
            {"dgrad qkv NN", HS_GEMM_NN, M, 768, 2304, 0}, {"dgrad proj NN", HS_GEMM_NN, M, 768, 768, 0},
            {"dgrad qkv as NT", HS_GEMM_NT, M, 768, 2304, 0},   // (the dgrads against a transposed weight copy: fc1 = "fc2 NT", proj = "proj NT", fc2 = "fc1 NT")
            {"wgrad fc1 TN", HS_GEMM_TN, 3072, 768, M, 0}, {"wgrad fc2 TN", HS_GEMM_TN, 768, 3072, M, 0}, {"wgrad qkv TN", HS_GEMM_TN, 2304, 768, M, 0},
            {"wgrad proj TN", HS_GEMM_TN, 768, 768, M, 0}, {"sq NT 4096", HS_GEMM_NT, 4096, 4096, 4096, 0}, {"sq NT 8192", HS_GEMM_NT, 8192, 8192, 8192, 0},
            // the fused epilogues of the encoder layer, as vit.hip launches them
            {"fc1+gelu NT epi", HS_GEMM_NT, M, 3072, 768, HS_EPI_BIAS | HS_EPI_GELU | HS_EPI_SAVE_DGELU}, {"fc2+resid NT epi", HS_GEMM_NT, M, 768, 3072, HS_EPI_BIAS | HS_EPI_RESID},
            {"proj+resid NT epi", HS_GEMM_NT, M, 768, 768, HS_EPI_BIAS | HS_EPI_RESID}, {"dgrad fc2 gelu' NN epi", HS_GEMM_NN, M, 3072, 768, HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU},
            {"dgrad fc2 gelu' as NT epi", HS_GEMM_NT, M, 3072, 768, HS_EPI_GELU_BWD | HS_EPI_SAVE_DGELU}};
        for (const Sh& s : shapes) {
            if (*filt && !strstr(s.name, filt)) continue;
            const bool tn = s.lay == HS_GEMM_TN;
            const size_t na = (size_t)s.m * s.k, nb = (size_t)s.n * s.k;
            void* A = dev_bf16_gauss(na, 1.0f); void* Bm = dev_bf16_gauss(nb, 1.0f);
            hs_gemm_args a; memset(&a, 0, sizeof a);
            a.A = A; a.B = Bm; a.M = s.m; a.N = s.n; a.K = s.k; a.layout = s.lay;
            a.lda = tn ? s.m : s.k; a.ldb = s.lay == HS_GEMM_NT ? s.k : s.n; a.ldc = s.n; a.alpha = 1.0f; a.batch = 1;
            void* out = dev_zero((size_t)s.m * s.n * 4);
            a.split_k = 1;
            if (tn) {
                a.out_f32 = (float*)out; a.flags = HS_EPI_ATOMIC;
                // same heuristic as hsenet_amd/functional.py:_wgrad_split
                const int tiles = ((s.m + 255) / 256) * ((s.n + 127) / 128), nkt = (s.k + 63) / 64;
                long best_cost = -1;
                for (int sp = 1; sp <= std::min(32, nkt); ++sp) {
                    const long cost = (long)((tiles * sp + 255) / 256) * ((nkt + sp - 1) / sp + 16);
                    if (best_cost < 0 || cost < best_cost) { best_cost = cost; a.split_k = sp; }
                }
                if (getenv("HS_SPLITK")) a.split_k = atoi(getenv("HS_SPLITK"));
            } else a.out_bf16 = out;
            void *e_bias = nullptr, *e_res = nullptr, *e_aux = nullptr;
            if (s.flags) {
                a.flags = s.flags;
                if (s.flags & HS_EPI_BIAS) { e_bias = dev_zero((size_t)s.n * 4); a.bias = (const float*)e_bias; }
                if (s.flags & HS_EPI_RESID) { e_res = dev_zero((size_t)s.m * s.n * 4); a.resid = (const float*)e_res; a.out_bf16 = nullptr; a.out_f32 = (float*)out; }
                if (s.flags & HS_EPI_GELU) { e_aux = dev_zero((size_t)s.m * s.n * 2); a.aux_out = e_aux; }
                if (s.flags & HS_EPI_GELU_BWD) { e_aux = dev_bf16_gauss((size_t)s.m * s.n, 0.5f); a.aux = e_aux; }
            }
            int rc = gemm(&a, HS_PREC_BF16, 0);
            if (rc) { printf("%s rc=%d\n", s.name, rc); continue; }
            CK(hipDeviceSynchronize());
            if (getenv("HS_GEMM_AB") && !tn && !s.flags) {       // A/B: the same GEMM with the 16x16x32 main loop toggled (HS_GEMM_NO_M16) must agree (both accumulate in f32)
                void* out2 = dev_zero((size_t)s.m * s.n * 2);
                hs_gemm_args a2 = a; a2.out_bf16 = out2;
                const bool had = getenv("HS_GEMM_NO_M16") != nullptr;
                if (had) unsetenv("HS_GEMM_NO_M16"); else setenv("HS_GEMM_NO_M16", "1", 1);
                gemm(&a2, HS_PREC_BF16, 0); CK(hipDeviceSynchronize());
                if (had) setenv("HS_GEMM_NO_M16", "1", 1); else unsetenv("HS_GEMM_NO_M16");
                Cmp c = cmp_bf16(out, out2, (size_t)s.m * s.n);
                printf("    M16 on/off: rel %.3e maxabs %.3e\n", c.rel, c.maxabs);
                CK(hipFree(out2));
            }
            const double us = time_us([&] { gemm(&a, HS_PREC_BF16, 0); }, reps);
            printf("%-16s M=%6d N=%5d K=%6d: %8.1f us  %7.1f TFLOP/s\n", s.name, s.m, s.n, s.k, us, 2.0 * s.m * s.n * s.k / us / 1e6);
            if (getenv("HS_GEMM_DBG") && (atoi(getenv("HS_GEMM_DBG")) == 4 || atoi(getenv("HS_GEMM_DBG")) >= 7)) {
                typedef void (*st_t)(unsigned long long*, int);
                auto st = sym<st_t>("hs_debug_gemm_stamps");
                unsigned long long v[8];
                st(nullptr, 1);
                gemm(&a, HS_PREC_BF16, 0); CK(hipDeviceSynchronize());
                st(v, 0);
                const double nb = (double)((s.m + 255) / 256) * ((s.n + 255) / 256);
                printf("    cycles/block (wave 0): setup %.0f | main loop %.0f (barrier wait %.0f, vmcnt wait %.0f, last-read wait %.0f) | barrier %.0f | staging %.0f | readback+math+stores %.0f\n", v[0] / nb, v[1] / nb, v[5] / nb, v[7] / nb, v[6] / nb, v[2] / nb, v[3] / nb, v[4] / nb);
            }
            if (getenv("HS_GEMM_DBG") && atoi(getenv("HS_GEMM_DBG")) == 22) {
                typedef void (*tr_t)(unsigned long long*, int);
                auto tr = sym<tr_t>("hs_debug_gemm_trace");
                const int nb = std::min(4096, ((s.m + 255) / 256) * ((s.n + 255) / 256));
                std::vector<unsigned long long> v(4 * (size_t)nb);
                gemm(&a, HS_PREC_BF16, 0); CK(hipDeviceSynchronize());
                tr(v.data(), nb);
                const char* path = getenv("HS_TRACE_OUT") ? getenv("HS_TRACE_OUT") : "gpurun_out/gemm_trace.csv";
                FILE* f = fopen(path, "w");
                if (f) {
                    fprintf(f, "block,start,main_end,epi_end,hw_id,xcc_id\n");
                    for (int b = 0; b < nb; ++b) fprintf(f, "%d,%llu,%llu,%llu,%llu,%llu\n", b, v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3] & 0xffffffffull, v[4 * b + 3] >> 32);
                    fclose(f);
                    printf("    timeline of %d blocks -> %s\n", nb, path);
                }
            }
            fflush(stdout);
            CK(hipFree(A)); CK(hipFree(Bm)); CK(hipFree(out));
            if (e_bias) CK(hipFree(e_bias));
            if (e_res) CK(hipFree(e_res));
            if (e_aux) CK(hipFree(e_aux));
        }
        return 0;
    }
    fprintf(stderr, "unknown mode %s\n", mode.c_str());
    return 1;
}
