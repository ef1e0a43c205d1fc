// Standalone timing of fmc_linear_bf16 through the C ABI (no torch): fixed cost vs per-k-tile cost of every arm on small shapes.
//   hipcc -O2 gemm_bench.cpp -o gemm_bench -ldl ; ./gemm_bench libfmc_hip.so M N K arm[,arm..] [bias res]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
typedef int (*lin_fn)(const void*, const void*, const void*, const void*, void*, int64_t, int, int, int64_t, int64_t, int64_t, float, int, int, int,
                      void*, int64_t, const void*, int64_t, int, const void*, void*);
int main(int argc, char** argv) {
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
    lin_fn fn = (lin_fn)dlsym(lib, "fmc_linear_bf16");
    const int64_t M = atoll(argv[2]); const int N = atoi(argv[3]), K = atoi(argv[4]);
    std::vector<int> arms; for (char* t = strtok(argv[5], ","); t; t = strtok(nullptr, ",")) arms.push_back(atoi(t));
    const bool use_bias = argc > 6 && atoi(argv[6]), use_res = argc > 7 && atoi(argv[7]);
    uint16_t *x, *w, *b, *r, *o; void* ws;
    hipMalloc(&x, M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&b, N * 2); hipMalloc(&r, M * N * 2); hipMalloc(&o, M * N * 2);
    const int64_t wsb = 16 * M * N * 4 + (500 << 20); hipMalloc(&ws, wsb); hipMemset(ws, 0, wsb);
    {   // random operands (sum of 4 uniforms), so that the arms can be compared with each other
        auto fill = [](uint16_t* d, size_t n, float scale, uint32_t seed) {
            std::vector<uint16_t> h(n);
            uint32_t rng = seed;
            for (size_t i = 0; i < n; ++i) {
                float a = 0.f;
                for (int j = 0; j < 4; ++j) { rng = rng * 1664525u + 1013904223u; a += (float)(rng >> 8) * (1.f / 16777216.f) - 0.5f; }
                a *= scale; uint32_t u; memcpy(&u, &a, 4); u += 0x7fffu + ((u >> 16) & 1u); h[i] = (uint16_t)(u >> 16);
            }
            hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
        };
        fill(x, (size_t)M * K, 1.7f, 1u); fill(w, (size_t)N * K, 0.1f, 2u); fill(b, N, 1.f, 3u); fill(r, (size_t)M * N, 1.7f, 4u);
    }
    std::vector<uint16_t> ref, got((size_t)M * N);
    printf("M=%lld N=%d K=%d bias=%d residual=%d: %.2f GFLOP\n", (long long)M, N, K, use_bias, use_res, 2.0 * M * N * K / 1e9);
    for (int arm : arms) {
        int tile = arm & 15, split = 1;
        if (arm >= 256) { tile = arm - 256; split = -2; }               // hybrid: whole rounds plain, the last partial round stream-K
        else if (arm >= 128) { tile = arm - 128; split = -1; }          // stream-K arms
        else if (arm >= 16) { tile = arm & 15; split = 1 << (arm >> 4); }
        auto call = [&]() { return fn(x, w, use_bias ? b : nullptr, use_res ? r : nullptr, o, M, N, K, K, N, N, 1.f, 0, tile, split, ws, wsb, nullptr, 0, 0, nullptr, nullptr); };
        int rc = call();
        if (rc) { printf("  arm %3d: rc=%d\n", arm, rc); continue; }
        hipDeviceSynchronize();
        hipMemcpy(got.data(), o, got.size() * 2, hipMemcpyDeviceToHost);
        double worst = 0;
        if (ref.empty()) ref = got;
        else for (size_t i = 0; i < got.size(); ++i) {
            uint32_t ua = (uint32_t)got[i] << 16, ub = (uint32_t)ref[i] << 16; float fa, fb; memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
            const double d = std::fabs((double)fa - fb); if (!(d <= worst)) worst = d;
        }
        hipMemset(o, 0, got.size() * 2);
        hipGraph_t g; hipGraphExec_t ge; hipStream_t st; hipStreamCreate(&st);
        lin_fn f2 = fn;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 16; ++i) f2(x, w, use_bias ? b : nullptr, use_res ? r : nullptr, o, M, N, K, K, N, N, 1.f, 0, tile, split, ws, wsb, nullptr, 0, 0, nullptr, st);
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9;
        for (int k = 0; k < 3; ++k) {
            hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 16 < best) best = ms / 16;
        }
        printf("  arm %3d: %7.1f us  %6.0f TF/s   max |diff| vs first arm %.3g\n", arm, best * 1e3, 2.0 * M * N * K / best / 1e9, worst);
    }
    return 0;
}
