// Do LDS-DMA operand requests and MFMAs of one CU overlap, and what does a request cost by shape?  (torch-free; hipcc --offload-arch=gfx950)
//   dma_mfma <shape> <dma per iteration and wave> <mfma per iteration and wave> <iterations> [waves per workgroup = 8] [workgroups = 256] [lds = 1]
// Every wave runs `iterations` times { D buffer_load ... lds (1 KiB each) ; s_waitcnt vmcnt(2 D) ; M v_mfma_f32_16x16x32_bf16 on constant
// registers }.  shape 0: 16 rows x 64 B per wave instruction (gemm160's sub-tile piece), 1: 8 rows x 128 B (whole lines), 2: 1 KiB linear
// (a pre-packed operand), 3: 4 rows x 256 B.  Rows are 1280 B apart (K = 640).  Each workgroup streams its own 64-KiB region (L2 resident).
// lds = 0: the same requests into registers (buffer_load_dwordx4).
// Reading: t(D, M) ~ max(t(D, 0), t(0, M)) -> the request stream hides under the matrix work; ~ sum -> it does not.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int LDS, int M>
__global__ __launch_bounds__(1024) void k(const unsigned char* base, int shape, int D, int iters, float* sink, int depth, int inter, int row_bytes_arg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long region = 65536 * 8;
    const unsigned char* mine = base + (long)blockIdx.x * region;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, (int)region, 0x00020000);
    const int row_bytes = row_bytes_arg;
    unsigned voff;
    int step;                                            // byte advance between consecutive pieces of a wave
    if (shape == 0) { voff = (lane >> 2) * row_bytes + (lane & 3) * 16; step = 64; }
    else if (shape == 1) { voff = (lane >> 3) * row_bytes + (lane & 7) * 16; step = 128; }
    else if (shape == 3) { voff = (lane >> 4) * row_bytes + (lane & 15) * 16; step = 256; }
    else { voff = lane * 16; step = 1024; }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
    u32x4 v = {0, 0, 0, 0};
    int soff = wave * 4096;
    for (int it = 0; it < iters; ++it) {
        auto req = [&](int d) {
            if (LDS) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wave * 8192 + (d & 7) * 1024), 16, (int)voff, soff, 0, 0);
            else { u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0); v[0] ^= t[0]; }
            soff += step;
            if (shape != 2 && (soff % row_bytes) + step > row_bytes) soff += (shape == 0 ? 16 : shape == 1 ? 8 : 4) * row_bytes - (soff % row_bytes);
            if (soff > region - 24 * row_bytes) soff = wave * 4096 % 1024;
        };
        if (inter) {                                     // 4 requests, each followed by a quarter of the MFMAs
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                req(q);
                if (inter == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int m = q * M / 4; m < (q + 1) * M / 4; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
                if (inter == 2) __builtin_amdgcn_s_setprio(0);
            }
            if (depth == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            continue;
        }
        for (int d = 0; d < D; ++d) req(d);
        if (LDS) {
            if (depth == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (depth == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (depth == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (depth == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (sink && s + (float)v[0] == 1234.5f) sink[0] = s;
}

int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 0;
    const int D = argc > 2 ? atoi(argv[2]) : 4;
    const int M = argc > 3 ? atoi(argv[3]) : 25;
    const int iters = argc > 4 ? atoi(argv[4]) : 2000;
    const int waves = argc > 5 ? atoi(argv[5]) : 8;
    const int wgs = argc > 6 ? atoi(argv[6]) : 256;
    const int use_lds = argc > 7 ? atoi(argv[7]) : 1;
    const int row_bytes = argc > 10 ? atoi(argv[10]) : 1280;   // pitch of the operand rows (K * 2 bytes)
    const int inter = argc > 9 ? atoi(argv[9]) : 0;   // 1: 4 x {request, M/4 MFMAs} per iteration; 2: the same with s_setprio around the MFMAs
    const int depth = argc > 8 ? atoi(argv[8]) : 2;   // s_waitcnt vmcnt(0 / 4 / 8 / 16 / 32) after each group of D requests
    unsigned char* buf;
    hipMalloc(&buf, (size_t)wgs * 65536 * 8 + 4096);
    hipMemset(buf, 1, (size_t)wgs * 65536 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
#define L(LD, MM) hipLaunchKernelGGL((k<LD, MM>), dim3(wgs), dim3(64 * waves), waves * 8192, 0, buf, shape, D, iters, nullptr, depth, inter, row_bytes)
        if (use_lds) { if (M == 0) L(1, 0); else if (M == 25) L(1, 25); else L(1, 50); }
        else { if (M == 0) L(0, 0); else if (M == 25) L(0, 25); else L(0, 50); }
    };
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double ns_iter = ms * 1e6 / iters;
    printf("pitch %5d inter %d depth %d shape %d lds %d  D %2d M %3d waves %d wgs %3d: %7.1f ns / iteration   %6.1f GB/s per CU   %6.1f ns per request and CU   MFMA %5.1f %% of 16-cycle issue at 2.4 GHz\n", row_bytes, inter, depth, shape, use_lds, D, M,
           waves, wgs, ns_iter, D * waves * 1024.0 / ns_iter, D ? ns_iter / (D * waves) : 0.0, M ? 100.0 * (M * (waves / 4.0) * 16 / 2.4) / ns_iter : 0.0);
    return 0;
}
