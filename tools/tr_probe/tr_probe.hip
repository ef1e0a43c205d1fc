// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds element index e at bf16 slot e (as an integer pattern); every lane
// supplies an address and the four 16-bit values it receives are printed.  Two address patterns:
//   A: all lanes the same address (0)            -> what does a uniform address return?
//   B: lane i of each 16-lane group g: row (i / 4), 8-byte piece (i % 4) of a [4][pitch] block that starts at
//      row 4 * g: addr = ((4 * g + i / 4) * PITCH + 4 * (i % 4)) * 2 bytes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int PITCH = 96;   // elements
__global__ void probe(uint16_t* out, int mode) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    unsigned addr = 0;
    if (mode == 1) addr = ((4 * g + i / 4) * PITCH + 4 * (i % 4)) * 2;
    if (mode == 2) addr = (i * PITCH + 4 * g) * 2;          // lane i -> row i, group g -> 4 columns
    addr += (unsigned)(uintptr_t)lds & 0xffff;
    typedef unsigned __attribute__((ext_vector_type(2))) u2;
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16;
    out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (PITCH %d): lane: 4 values as (row,col) = e / PITCH, e %% PITCH\n", mode, PITCH);
        for (int l = 0; l < 64; ++l) {
            printf("  l%02d:", l);
            for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l * 4 + j] / PITCH, h[l * 4 + j] % PITCH);
            if (l % 4 == 3) printf("\n");
        }
    }
    return 0;
}
