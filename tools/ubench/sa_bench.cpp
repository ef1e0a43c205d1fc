// Standalone timing + spot check of fmc_spatial_attn_fwd through the C ABI (no torch): level-0 self-attention shape by default.
//   hipcc -O2 sa_bench.cpp -o sa_bench -ldl ;  ./sa_bench <libfmc_hip.so> [B S H D iters]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
typedef int (*sa_fn)(const void*, const void*, const void*, void*, float*, int, int, int, int, int, int64_t, int64_t, int64_t, int64_t,
                     int64_t, int64_t, int, float, int, void*);
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: sa_bench lib.so [B S H D iters]\n"); return 2; }
    const int B = argc > 2 ? atoi(argv[2]) : 32, S = argc > 3 ? atoi(argv[3]) : 2560, H = argc > 4 ? atoi(argv[4]) : 8,
              D = argc > 5 ? atoi(argv[5]) : 40, iters = argc > 6 ? atoi(argv[6]) : 20;
    const int Skv = argc > 7 ? atoi(argv[7]) : S, kvdiv = argc > 8 ? atoi(argv[8]) : 1;   // cross attention: K/V [B / kvdiv, Skv, 2C]
    const bool cross = argc > 7;
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
    sa_fn fn = (sa_fn)dlsym(lib, "fmc_spatial_attn_fwd");
    const int C = H * D;
    const size_t n = (size_t)B * S * 3 * C;
    std::vector<uint16_t> h(n);
    uint32_t rng = 12345u;
    for (size_t i = 0; i < n; ++i) {   // sum of 4 uniforms: roughly normal, sigma ~ 1
        float a = 0.f;
        for (int j = 0; j < 4; ++j) { rng = rng * 1664525u + 1013904223u; a += (float)(rng >> 8) * (1.f / 16777216.f) - 0.5f; }
        h[i] = f2bf(a * 1.73f);
    }
    uint16_t *qkv, *o; float* lse = nullptr;
    const bool dbg = getenv("SA_BENCH_DBG") != nullptr;
    if (dbg) { hipMalloc(&lse, (size_t)B * H * S * 4 + (1 << 20)); hipMemset(lse, 0, (size_t)B * H * S * 4 + (1 << 20)); }
    hipMalloc(&qkv, n * 2); hipMalloc(&o, (size_t)B * S * C * 2);
    hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemset(o, 0, (size_t)B * S * C * 2);
    std::vector<uint16_t> hkv;
    uint16_t* kv = nullptr;
    if (cross) {
        hkv.resize((size_t)(B / kvdiv) * Skv * 2 * C);
        for (size_t i = 0; i < hkv.size(); ++i) { float a = 0.f; for (int j = 0; j < 4; ++j) { rng = rng * 1664525u + 1013904223u; a += (float)(rng >> 8) * (1.f / 16777216.f) - 0.5f; } hkv[i] = f2bf(a * 1.73f); }
        hipMalloc(&kv, hkv.size() * 2); hipMemcpy(kv, hkv.data(), hkv.size() * 2, hipMemcpyHostToDevice);
    }
    const float scale = 1.f / std::sqrt((float)D);
    auto call = [&]() {
        if (cross) return fn(qkv, kv, kv + C, o, lse, B, H, S, Skv, D, (int64_t)S * 3 * C, 3 * C, (int64_t)Skv * 2 * C, 2 * C, (int64_t)S * C, C, kvdiv, scale, 0, nullptr);
        return fn(qkv, qkv + C, qkv + 2 * C, o, lse, B, H, S, S, D, (int64_t)S * 3 * C, 3 * C, (int64_t)S * 3 * C, 3 * C, (int64_t)S * C, C, 1,
                  scale, /*FMC_BF16*/ 0, nullptr);
    };
    int rc = call();
    if (rc) { printf("rc=%d\n", rc); return 1; }
    hipDeviceSynchronize();
    for (int i = 0; i < 3; ++i) call();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) call();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    const double fl = 4.0 * B * H * (double)S * Skv * D;
    // spot check: a few (b, h, q) rows against a double-precision softmax over the bf16 inputs
    std::vector<uint16_t> ho((size_t)B * S * C);
    hipMemcpy(ho.data(), o, ho.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0, ref_max = 0;
    const int picks[][3] = {{0, 0, 0}, {B - 1, H - 1, S - 1}, {B / 2, 3 % H, S / 2 + 17}, {1 % B, 5 % H, 255}, {7 % B, 2 % H, 256}, {B - 1, 0, 511},
                            {3 % B, H - 1, 512}, {11 % B, 1 % H, S - 65}, {B / 3, 4 % H, 63}, {B / 2 + 1, 6 % H, 64}};
    for (auto& p : picks) {
        const int b = p[0], hh = p[1], qi = p[2] < S ? p[2] : S - 1;
        const uint16_t* base = h.data() + (size_t)b * S * 3 * C;
        const uint16_t* kb = cross ? hkv.data() + (size_t)(b / kvdiv) * Skv * 2 * C : base + C;
        const uint16_t* vb = cross ? kb + C : base + 2 * C;
        const int kvs = cross ? 2 * C : 3 * C;
        std::vector<double> sc(Skv);
        double mx = -1e300;
        for (int k = 0; k < Skv; ++k) {
            double a = 0;
            for (int d = 0; d < D; ++d) a += (double)bf2f(base[(size_t)qi * 3 * C + hh * D + d]) * bf2f(kb[(size_t)k * kvs + hh * D + d]);
            sc[k] = a * scale; mx = std::max(mx, sc[k]);
        }
        double l = 0;
        for (int k = 0; k < Skv; ++k) { sc[k] = std::exp(sc[k] - mx); l += sc[k]; }
        for (int d = 0; d < D; ++d) {
            double a = 0;
            for (int k = 0; k < Skv; ++k) a += sc[k] * bf2f(vb[(size_t)k * kvs + hh * D + d]);
            a /= l;
            const double got = bf2f(ho[((size_t)b * S + qi) * C + hh * D + d]);
            worst = std::max(worst, std::fabs(got - a)); ref_max = std::max(ref_max, std::fabs(a));
        }
    }
    if (dbg) {
        const int nwg = B * H * (S / 256) * (getenv("FMC_SA_PIPE") && atoi(getenv("FMC_SA_PIPE")) == 0 ? 1 : 1);
        std::vector<float> hl((size_t)nwg * 16);
        hipMemcpy(hl.data(), lse, hl.size() * 4, hipMemcpyDeviceToHost);
        double c = 0, cl = 0, r = 0; int n = 0;
        double xr[8] = {0}, xc[8] = {0}; int xn[8] = {0}; float rmin = 1e30f, rmax = 0;
        for (int i = 0; i < nwg * 4; ++i) {
            if (hl[i * 4 + 2] <= 0) continue;
            c += hl[i * 4]; cl += hl[i * 4 + 1]; r += hl[i * 4 + 2]; ++n;
            const int x = ((int)hl[i * 4 + 3]) & 7;
            xr[x] += hl[i * 4 + 2]; xc[x] += hl[i * 4]; ++xn[x];
            rmin = std::min(rmin, hl[i * 4 + 2]); rmax = std::max(rmax, hl[i * 4 + 2]);
        }
        c /= n; cl /= n; r /= n;
        printf("  %d waves reporting; per wave: %.0f cycles total, %.0f in the pipelined sweeps, %.2f us (min %.2f max %.2f) -> shader clock %.2f GHz\n", n, c, cl,
               r / 100.0, rmin / 100.0, rmax / 100.0, c / (r * 10.0));
        printf("  per XCD (blockIdx & 7): us ");
        for (int x = 0; x < 8; ++x) printf("%.1f ", xr[x] / std::max(1, xn[x]) / 100.0);
        printf(" GHz ");
        for (int x = 0; x < 8; ++x) printf("%.2f ", xc[x] / std::max(1.0, xr[x]) / 10.0);
        printf("\n");
    }
    printf("%-44s B=%d S=%d H=%d d=%d: %7.4f ms  %7.1f TF/s  frac %.4f  spot err %.2e (ref max %.3f)\n", argv[1], B, S, H, D, ms, fl / ms / 1e9,
           fl / ms / 1e9 / 2500.0, worst, ref_max);
    return 0;
}
