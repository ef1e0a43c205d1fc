// Micro-benchmark: can the softmax VALU work of one wave run under the MFMAs of the other wave of the same SIMD?
// A 512-thread workgroup puts waves w and w+4 on SIMD w%4.  Per iteration a "matrix" role issues 28 v_mfma_f32_32x32x16_bf16
// (one 64-key tile of the d = 40 attention: 2 query blocks x 2 key blocks x 7) and a "softmax" role issues 64 v_exp_f32 +
// 32 v_cvt_pk_bf16_f32 + 32 v_max3_f32 (the same tile's softmax).  hipcc --offload-arch=gfx950 -O3 pingpong.hip -o pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void matrix_role(f32x16 (&acc)[4], const bf16x8& a, const bf16x8& b) {
#pragma unroll
    for (int i = 0; i < 28; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    union { bf16x2_t v; uint32_t u; } r;
    r.v = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return r.u;
}
// 4 scores: the compiler may not hoist or drop them (zero-instruction value barriers), schedules them itself
template <bool MAX3>
__device__ __forceinline__ void softmax4(float (&x)[64], uint32_t (&pk)[32], float& w, int i) {
    asm volatile("" : "+v"(x[i]), "+v"(x[i + 1]), "+v"(x[i + 2]), "+v"(x[i + 3]));
    const float e0 = __builtin_amdgcn_exp2f(x[i]), e1 = __builtin_amdgcn_exp2f(x[i + 1]);
    const float e2 = __builtin_amdgcn_exp2f(x[i + 2]), e3 = __builtin_amdgcn_exp2f(x[i + 3]);
    if (MAX3) w = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(x[i], x[i + 1]), __builtin_fmaxf(x[i + 2], w)), x[i + 3]);
    pk[i / 2] = pack_bf2(e0, e1);
    pk[i / 2 + 1] = pack_bf2(e2, e3);
    asm volatile("" : "+v"(pk[i / 2]), "+v"(pk[i / 2 + 1]));
}
template <bool MAX3>
__device__ __forceinline__ void softmax_role(float (&x)[64], uint32_t (&pk)[32], float& w) {
#pragma unroll
    for (int i = 0; i < 64; i += 4) softmax4<MAX3>(x, pk, w, i);
}
// MODE 0 matrix role only (waves 0-3), 1 softmax role only (waves 4-7), 2 both with fixed roles, 3 roles swap every
// iteration (ping-pong, one barrier per phase), 4 every wave does matrix then softmax (serial; barrier per iteration),
// 5 as 4 without barriers, 6 as 3 with the second half at s_setprio 1, 7: one wave per SIMD (waves 4-7 exit) doing matrix
// then softmax, 8: one wave per SIMD, MFMAs and softmax interleaved in the source (1 MFMA : ~5 VALU)
template <int MODE, bool MAX3>
__global__ __launch_bounds__(512) void pp_kernel(float* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (threadIdx.x + i)); b[i] = (__bf16)(seed * i); }
    float x[64]; uint32_t pk[32]; float w = -1e30f;
    for (int i = 0; i < 64; ++i) x[i] = -seed * (i + threadIdx.x);
    for (int i = 0; i < 32; ++i) pk[i] = 0;
    const bool second = wave >= 4;
    if (MODE == 6 && second) __builtin_amdgcn_s_setprio(1);
    if ((MODE == 7 || MODE == 8) && second) return;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { if (!second) matrix_role(acc, a, b); __builtin_amdgcn_s_barrier(); }
        else if (MODE == 1) { if (second) softmax_role<MAX3>(x, pk, w); __builtin_amdgcn_s_barrier(); }
        else if (MODE == 2) { if (!second) matrix_role(acc, a, b); else softmax_role<MAX3>(x, pk, w); __builtin_amdgcn_s_barrier(); }
        else if (MODE == 3 || MODE == 6) {
            if (!second) matrix_role(acc, a, b); else softmax_role<MAX3>(x, pk, w);
            __builtin_amdgcn_s_barrier();
            if (second) matrix_role(acc, a, b); else softmax_role<MAX3>(x, pk, w);
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 4) { matrix_role(acc, a, b); softmax_role<MAX3>(x, pk, w); __builtin_amdgcn_s_barrier(); }
        else if (MODE == 5 || MODE == 7) { matrix_role(acc, a, b); __builtin_amdgcn_sched_barrier(0); softmax_role<MAX3>(x, pk, w); __builtin_amdgcn_sched_barrier(0); }
        else if (MODE == 8) {
#pragma unroll
            for (int i = 0; i < 28; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i < 16) {
                    softmax4<MAX3>(x, pk, w, 4 * i);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = w;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 32; ++i) s += (float)pk[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x < 256) {   // shader cycles and 100 MHz ticks of this workgroup
        out[1024 + blockIdx.x * 2] = (float)(__builtin_readcyclecounter() - c0);
        out[1024 + blockIdx.x * 2 + 1] = (float)(__builtin_amdgcn_s_memrealtime() - r0);
    }
}

template <int MODE, bool MAX3> void run(const char* what, float* out) {
    const int iters = 2000, grid = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    pp_kernel<MODE, MAX3><<<grid, 512>>>(out, 10, 0.001f);
    hipEventRecord(e0);
    pp_kernel<MODE, MAX3><<<grid, 512>>>(out, iters, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // grid / 256 workgroups per CU one after the other; per iteration per workgroup:
    const double us_it = ms * 1e3 / iters / (grid / 256);
    std::vector<float> h(512);
    hipMemcpy(h.data(), out + 1024, 512 * 4, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < 256; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
    const double ghz = c / (r * 10.0);
    printf("mode %d max3=%d  %-58s %8.3f us/iter  %5.0f cycles/iter at the measured shader clock %.2f GHz\n", MODE, (int)MAX3, what, us_it, c / 256 / iters, ghz);
}
int main() {
    float* out; hipMalloc(&out, 8192);
    run<0, true>("matrix role only (28 MFMA, waves 0-3)", out);
    run<1, true>("softmax role only (64 exp+32 cvt+32 max3, waves 4-7)", out);
    run<1, false>("softmax role only, no max3", out);
    run<2, true>("both, fixed roles", out);
    run<2, false>("both, fixed roles, no max3", out);
    run<3, true>("ping-pong (2 phases = one tile for all 8 waves)", out);
    run<3, false>("ping-pong, no max3", out);
    run<6, false>("ping-pong, no max3, second half prio 1", out);
    run<4, true>("8 waves serial matrix->softmax, barrier per tile", out);
    run<5, true>("8 waves serial, no barrier", out);
    run<5, false>("8 waves serial, no barrier, no max3", out);
    run<7, true>("4 waves (1/SIMD) serial", out);
    run<8, true>("4 waves (1/SIMD) interleaved 1 MFMA : 4-5 VALU", out);
    run<8, false>("4 waves (1/SIMD) interleaved, no max3", out);
    return 0;
}
