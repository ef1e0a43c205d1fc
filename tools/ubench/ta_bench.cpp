// Standalone timing + spot check of fmc_temporal_attn_fwd (bf16) through the C ABI on the fused [B, F, P, 3C] projection layout.
//   hipcc -O2 ta_bench.cpp -o ta_bench -ldl ; ./ta_bench lib.so [clips F P H D]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
typedef int (*ta_fn)(const void*, const void*, const void*, void*, int, int, int, int, int, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, float,
                     int, void*);
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
int main(int argc, char** argv) {
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
    ta_fn fn = (ta_fn)dlsym(lib, "fmc_temporal_attn_fwd");
    const int B = argc > 2 ? atoi(argv[2]) : 2, F = argc > 3 ? atoi(argv[3]) : 16, P = argc > 4 ? atoi(argv[4]) : 2560, H = argc > 5 ? atoi(argv[5]) : 8,
              D = argc > 6 ? atoi(argv[6]) : 40;
    const int C = H * D;
    const size_t n = (size_t)B * F * P * 3 * C;
    std::vector<uint16_t> h(n);
    uint32_t rng = 777u;
    for (size_t i = 0; i < n; ++i) {
        float a = 0.f;
        for (int j = 0; j < 4; ++j) { rng = rng * 1664525u + 1013904223u; a += (float)(rng >> 8) * (1.f / 16777216.f) - 0.5f; }
        h[i] = f2bf(a * 1.73f);
    }
    uint16_t *qkv, *o;
    hipMalloc(&qkv, n * 2); hipMalloc(&o, (size_t)B * F * P * C * 2);
    hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
    const float scale = 1.f / std::sqrt((float)D);
    auto call = [&]() {
        return fn(qkv, qkv + C, qkv + 2 * C, o, B, P, F, H, D, (int64_t)F * P * 3 * C, (int64_t)P * 3 * C, 3 * C, (int64_t)F * P * C, (int64_t)P * C, C, scale, 0, nullptr);
    };
    int rc = call();
    if (rc) { printf("rc=%d\n", rc); return 1; }
    hipDeviceSynchronize();
    for (int i = 0; i < 3; ++i) call();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) call();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    std::vector<uint16_t> ho((size_t)B * F * P * C);
    hipMemcpy(ho.data(), o, ho.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0, ref_max = 0;
    const int picks[][3] = {{0, 0, 0}, {B - 1, P - 1, H - 1}, {0, P / 2 + 3, 3 % H}, {B - 1, 17 % P, 5 % H}, {0, P - 2, 1 % H}};
    for (auto& pk : picks) {
        const int b = pk[0], p = pk[1], hh = pk[2];
        for (int fq = 0; fq < F; ++fq) {
            std::vector<double> sc(F); double mx = -1e300;
            for (int fk = 0; fk < F; ++fk) {
                double a = 0;
                for (int d = 0; d < D; ++d)
                    a += (double)bf2f(h[(((size_t)b * F + fq) * P + p) * 3 * C + hh * D + d]) * bf2f(h[(((size_t)b * F + fk) * P + p) * 3 * C + C + hh * D + d]);
                sc[fk] = a * scale; mx = std::max(mx, sc[fk]);
            }
            double l = 0; for (int fk = 0; fk < F; ++fk) { sc[fk] = std::exp(sc[fk] - mx); l += sc[fk]; }
            for (int d = 0; d < D; ++d) {
                double a = 0;
                for (int fk = 0; fk < F; ++fk) a += sc[fk] * bf2f(h[(((size_t)b * F + fk) * P + p) * 3 * C + 2 * C + hh * D + d]);
                a /= l;
                const double got = bf2f(ho[(((size_t)b * F + fq) * P + p) * C + hh * D + d]);
                worst = std::max(worst, std::fabs(got - a)); ref_max = std::max(ref_max, std::fabs(a));
            }
        }
    }
    const double bytes = 4.0 * B * F * P * C * 2;
    printf("%-40s clips=%d F=%d P=%d H=%d d=%d: %7.2f us  %5.2f TB/s (%.1f MB)  frac of 8 TB/s %.3f  spot err %.2e (ref max %.2f)\n", argv[1], B, F, P, H, D,
           ms * 1e3, bytes / ms / 1e9, bytes / 1e6, bytes / ms / 1e9 / 8.0, worst, ref_max);
    return 0;
}
