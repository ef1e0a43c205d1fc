// Per-CU read bandwidth on gfx950 as a function of where the data lives and how it is requested (torch-free; hipcc --offload-arch=gfx950).
//   cu_bw <workgroups> <bytes per workgroup region> <mode> <waves per workgroup> <passes>
// Every workgroup streams its OWN region (no sharing between workgroups) `passes` times: a region of 64 KiB stays in the L2 after the first
// pass, 512 KiB x 256 workgroups = 128 MiB lives in the Infinity Cache, 4 MiB x 256 = 1 GiB comes from HBM.
// mode 0: buffer_load_dwordx4 to registers, 16 rows x 64 B per wave instruction (the gemm160 sub-tile piece shape)
// mode 1: the same, 8 rows x 128 B (full lines)        mode 2: fully linear (64 lanes x 16 B = 1 KiB contiguous)
// mode 3 / 4 / 5: as 0 / 1 / 2 through buffer_load ... lds (LDS-DMA)
// Up to 12 wave instructions in flight per wave (s_waitcnt vmcnt(8) after every group of 4), like the GEMM loops.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>
__global__ __launch_bounds__(512) void cu_bw_kernel(const unsigned char* base, long region, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned char* mine = base + (long)blockIdx.x * region;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, (int)region, 0x00020000);
    constexpr int M = MODE % 3;
    // a "row" is 1280 bytes (K = 640 bf16): piece p covers 16 rows x 64 B (M = 0), 8 rows x 128 B (M = 1) or 1 KiB linear (M = 2)
    const int row_bytes = 1280;
    unsigned voff;
    if (M == 0) voff = (lane >> 2) * row_bytes + (lane & 3) * 16;
    else if (M == 1) voff = (lane >> 3) * row_bytes + (lane & 7) * 16;
    else voff = lane * 16;
    const long piece_bytes = M == 2 ? 1024 : (M == 0 ? 16 : 8) * (long)row_bytes;      // address span one piece group advances by
    // pieces of a group: column blocks across the 1280-byte row (20 x 64 B or 10 x 128 B), then the next row group
    const int cols = M == 2 ? 1 : (M == 0 ? 20 : 10);
    const long groups = region / piece_bytes;
    const long total = groups * cols;                      // pieces in the region
    u32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int pass = 0; pass < passes; ++pass) {
        for (long p = wave * 4; p + 3 < total; p += nw * 4) {
            u32x4 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long q = p + e, g = q / cols, c = q - g * cols;
                const int soff = (int)(g * piece_bytes + c * (M == 0 ? 64 : 128));
                if (MODE < 3) v[e] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wave * 4096 + e * 1024), 16, (int)voff, soff, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (MODE < 3) { asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && acc0[0] + acc1[0] + acc2[0] + acc3[0] == 0x1234567u) sink[0] = 1;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    const long region = argc > 2 ? atol(argv[2]) : 65536;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const int waves = argc > 4 ? atoi(argv[4]) : 8;
    const int passes = argc > 5 ? atoi(argv[5]) : 16;
    unsigned char* buf;
    hipMalloc(&buf, (size_t)wgs * region + 4096);
    hipMemset(buf, 1, (size_t)wgs * region);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        const size_t l = 8 * 4096;
        switch (mode) {
            case 0: hipLaunchKernelGGL(cu_bw_kernel<0>, dim3(wgs), dim3(64 * waves), l, 0, buf, region, passes, nullptr); break;
            case 1: hipLaunchKernelGGL(cu_bw_kernel<1>, dim3(wgs), dim3(64 * waves), l, 0, buf, region, passes, nullptr); break;
            case 2: hipLaunchKernelGGL(cu_bw_kernel<2>, dim3(wgs), dim3(64 * waves), l, 0, buf, region, passes, nullptr); break;
            case 3: hipLaunchKernelGGL(cu_bw_kernel<3>, dim3(wgs), dim3(64 * waves), l, 0, buf, region, passes, nullptr); break;
            case 4: hipLaunchKernelGGL(cu_bw_kernel<4>, dim3(wgs), dim3(64 * waves), l, 0, buf, region, passes, nullptr); break;
            default: hipLaunchKernelGGL(cu_bw_kernel<5>, dim3(wgs), dim3(64 * waves), l, 0, buf, region, passes, nullptr); break;
        }
    };
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double bytes = (double)wgs * region * passes;
    printf("wgs %4d region %8ld B mode %d waves %d passes %3d: %8.1f us  %7.1f GB/s per workgroup  %6.2f TB/s total\n", wgs, region, mode, waves,
           passes, ms * 1e3, bytes / wgs / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
    return 0;
}
