// What a streaming read+write pass can reach at the U-Net's activation sizes: plain 16-byte copy kernels (one shot / grid-stride with
// 1, 2, 4 loads in flight / non-temporal) next to fmc_layernorm_fwd and fmc_groupnorm_silu_fwd through the C ABI.
//   hipcc -O3 --offload-arch=gfx950 bw_bench.cpp -o bw_bench -ldl ; ./bw_bench libfmc_hip.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_oneshot(const u32x4* __restrict__ x, u32x4* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_stride(const u32x4* __restrict__ x, u32x4* __restrict__ y, int64_t n) {
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * step < n) v[u] = NT ? __builtin_nontemporal_load(x + i + u * step) : x[i + u * step];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * step < n) { if (NT) __builtin_nontemporal_store(v[u], y + i + u * step); else y[i + u * step] = v[u]; }
    }
}
typedef int (*ln_fn)(const void*, void*, const float*, const float*, const float*, int64_t, int, float, int, int, int, void*);
typedef int (*gn_fn)(const void*, void*, const float*, const float*, float*, void*, int, int, int, int, float, int, int, const void*, int, void*);
typedef int64_t (*gnws_fn)(int, int, int);
template <typename F> static float timeit(F f, int iters = 30) {
    for (int i = 0; i < 3; ++i) f();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters * 1e3f;
}
int main(int argc, char** argv) {
    void* lib = argc > 1 ? dlopen(argv[1], RTLD_NOW) : nullptr;
    ln_fn ln = lib ? (ln_fn)dlsym(lib, "fmc_layernorm_fwd") : nullptr;
    gn_fn gn = lib ? (gn_fn)dlsym(lib, "fmc_groupnorm_silu_fwd") : nullptr;
    gnws_fn gnws = lib ? (gnws_fn)dlsym(lib, "fmc_groupnorm_workspace_bytes") : nullptr;
    const size_t maxb = (size_t)81920 * 1280 * 2;
    uint16_t *x, *y; float *g, *b, *stats, *pe; void* ws;
    hipMalloc(&pe, 16 * 1280 * 4); hipMemset(pe, 0, 16 * 1280 * 4);
    hipMalloc(&x, maxb); hipMalloc(&y, maxb); hipMalloc(&g, 8192 * 4); hipMalloc(&b, 8192 * 4); hipMalloc(&stats, 1 << 20); hipMalloc(&ws, 64 << 20);
    hipMemset(x, 0x3c, maxb); hipMemset(g, 0, 8192 * 4); hipMemset(b, 0, 8192 * 4);
    struct { int64_t M; int C; int n_img; } shapes[] = {{81920, 320, 32}, {20480, 640, 32}, {5120, 1280, 32}, {81920, 640, 32}, {81920, 1280, 32}};
    for (auto& s : shapes) {
        const size_t bytes = (size_t)s.M * s.C * 2;
        const int64_t n = bytes / 16;
        const double mb = 2.0 * bytes / 1e6;
        printf("[M=%lld C=%d]  %.1f MB in + out\n", (long long)s.M, s.C, mb);
        auto rep = [&](const char* what, float us) { printf("   %-44s %7.1f us  %5.2f TB/s\n", what, us, mb / us); };
        rep("copy, one 16 B element per thread", timeit([&] { copy_oneshot<<<(unsigned)((n + 255) / 256), 256>>>((const u32x4*)x, (u32x4*)y, n); }));
        for (int wg : {1024, 2048, 4096}) {
            char nm[96];
            snprintf(nm, 96, "copy, grid-stride %d WGs, 1 in flight", wg); rep(nm, timeit([&] { copy_stride<1, false><<<wg, 256>>>((const u32x4*)x, (u32x4*)y, n); }));
            snprintf(nm, 96, "copy, grid-stride %d WGs, 4 in flight", wg); rep(nm, timeit([&] { copy_stride<4, false><<<wg, 256>>>((const u32x4*)x, (u32x4*)y, n); }));
            snprintf(nm, 96, "copy, grid-stride %d WGs, 4 in flight, nt", wg); rep(nm, timeit([&] { copy_stride<4, true><<<wg, 256>>>((const u32x4*)x, (u32x4*)y, n); }));
        }
        if (ln && s.C <= 1280) rep("fmc_layernorm_fwd", timeit([&] { ln(x, y, g, b, nullptr, s.M, s.C, 1e-5f, 1, 1, 0, nullptr); }));
        if (ln && s.C <= 1280) rep("fmc_layernorm_fwd + positional encoding (16 frames)", timeit([&] { ln(x, y, g, b, pe, s.M, s.C, 1e-5f, (int)(s.M / 32), 16, 0, nullptr); }));
        if (gn) rep("fmc_groupnorm_silu_fwd (32 groups, SiLU)", timeit([&] { gn(x, y, g, b, stats, ws, s.n_img, (int)(s.M / s.n_img), s.C, 32, 1e-5f, 1, 0, nullptr, 0, nullptr); }));
    }
    return 0;
}
