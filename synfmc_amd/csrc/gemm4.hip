// bf16 token GEMM for the SMALL-M projections of the inner U-Net levels (M = 5120 / 1280 rows, N, K = 1280 .. 5120): nn.Linear of diffusers'
// Attention / Transformer2D / the motion module at the 10x16 and 5x8 levels (call sites fmc/models/attention_processor.py:50-69,255-283,
// fmc/models/motion_module.py:219,228,284) with the `+ residual` / Camera-Adapter axpy behind them (`alpha (x W^T + b) + r [+ r2]`).
//
// Why another GEMM.  Those launches are 200 - 256 output tiles of 20 - 40 k-steps: on the vendor arm (hipBLASLt 128x256 tiles, 200 workgroups,
// 36 us for 16.8 GFLOP = 0.19 of the MFMA peak; 121 of the step's 218 `linear` calls) and on this library's ring kernels alike the launch is
// prologue + a short main loop + epilogue, each exposed.  Here the software-pipelined loop of conv_halo4_kernel carries a plain GEMM:
//   * 160 x 160 output tiles (M = 5120, N = 1280: exactly 256), 4 waves = 2 x 2, one per SIMD, each 80 x 80 outputs as 5 x 5
//     v_mfma_f32_16x16x32_bf16;
//   * a 32-deep sub-tile is 20 one-KiB LDS-DMA pieces (A 160 rows + W 160 rows of 64 bytes, the 16-byte chunks XOR-swizzled through the
//     SOURCE address: conflict-free fragment reads) = FIVE per wave, in a ring of 6 sub-tiles (120 KiB) requested five ahead;
//   * fragments of sub-tile s + 1 are read into a second register set and the requests of sub-tile s + 5 issued BETWEEN the 25 MFMAs of
//     sub-tile s (`sched_group_barrier`); one barrier per sub-tile; counted `vmcnt(15)`, never a drain.
// Roofline: MFMA bound for K >= 1280.  Algorithmic flops per launch = 2 M N K.
#include <type_traits>

#include "common.h"

namespace {

constexpr int BM = 160, BN = 160, NBUF = 6, DLEAD = NBUF - 1;
constexpr int SUB = (BM + BN) * 64;                        // one 32-deep sub-tile: 320 rows x 64 B = 20 KiB
constexpr int LDS_BYTES = NBUF * SUB;                      // 122,880
constexpr unsigned OOB = 0x80000000u;

struct G4Params {
    const bf16_t* a; const bf16_t* w; const bf16_t* bias; const bf16_t* res; const bf16_t* res2; bf16_t* out;
    int64_t M; int N, K;
    int64_t lda, ldres, ldo;
    float alpha;
    int tiles_m, tiles_n;
    int64_t a_bytes, w_bytes;
};

__global__ __launch_bounds__(256, 1)
void gemm4_kernel(const G4Params P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, kq = lane >> 4;

    // ---- my tile: XCD x owns a contiguous range of the launch order; inside it n runs fastest (the A rows of a tile row come from that L2 once) ----
    int tile_m, tile_n;
    {
        const int total = P.tiles_m * P.tiles_n;
        const int id = blockIdx.x, q = total >> 3, r = total & 7, xcd = id & 7;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        tile_m = lin / P.tiles_n;
        tile_n = lin - tile_m * P.tiles_n;
    }
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;
    const int nks = P.K >> 5;

    // ---- my five DMA pieces per sub-tile: piece i = 5 wave + e; i < 10: A rows 16 i .., else W rows 16 (i - 10) ..; lane -> row lane / 4, physical
    //      chunk lane % 4, which holds logical chunk (lane % 4) ^ (3 * ((row >> 3) & 1)) ----------------------------------------------------------------
    // (waves 0, 1 hold the ten A pieces, waves 2, 3 the ten W pieces: one buffer descriptor per wave, no branch in the request stream)
    const bool isw = wave >= 2;
    const __amdgpu_buffer_rsrc_t rs = isw ? __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)P.w_bytes, 0x00020000)
                                          : __builtin_amdgcn_make_buffer_rsrc((void*)P.a, 0, (int)P.a_bytes, 0x00020000);
    const int prow = lane >> 2, pch = lane & 3, psrc = pch ^ (3 * ((prow >> 3) & 1));
    unsigned p_vo[5];
    const int p_lds0 = (isw ? BM * 64 : 0) + 5 * (wave & 1) * 1024;          // my first piece inside a sub-tile buffer; piece e at + e KiB
#pragma unroll
    for (int e = 0; e < 5; ++e) {
        const int row = 16 * (5 * (wave & 1) + e) + prow;
        if (isw) p_vo[e] = n0 + row < P.N ? (unsigned)(((int64_t)(n0 + row) * P.K + psrc * 8) * 2) : OOB;
        else p_vo[e] = m0 + row < P.M ? (unsigned)(((m0 + row) * P.lda + psrc * 8) * 2) : OOB;
    }
    int iss_k = 0, iss_slot = 0;                             // byte offset of the next sub-tile inside a row / its ring slot
    auto issue = [&]() {
        unsigned char* stage = smem_raw + iss_slot * SUB + p_lds0;
#pragma unroll
        for (int e = 0; e < 5; ++e)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(stage + e * 1024), 16, (int)p_vo[e], iss_k, 0, 0);
        iss_k = iss_k + 64 >= P.K * 2 ? 0 : iss_k + 64;    // (past the end of K the stream wraps to valid addresses: the counts stay exact)
        iss_slot = iss_slot + 1 == NBUF ? 0 : iss_slot + 1;
    };

    f32x4 acc[5][5];
    bf16x8 wf[2][5], af[2][5];
    const int frag = (l15 * 32 + (kq ^ (3 * ((l15 >> 3) & 1))) * 8) * 2;
    const int afrag = wr * 80 * 64 + frag, wfrag = BM * 64 + wc * 80 * 64 + frag;      // + slot * SUB + block * 1024
    int rd_slot = 0;
    auto read_frags = [&](int set) {
        const unsigned char* base = smem_raw + rd_slot * SUB;
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) wf[set][nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + wfrag + nb * 1024));
#pragma unroll
        for (int mb = 0; mb < 5; ++mb) af[set][mb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + afrag + mb * 1024));
        rd_slot = rd_slot + 1 == NBUF ? 0 : rd_slot + 1;
    };

    // ---- prologue: sub-tiles 0 .. 4 in flight, sub-tile 0 retired and published, its fragments read ---------------------------------------------------
#pragma unroll
    for (int d = 0; d < DLEAD; ++d) issue();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (DLEAD - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0);
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto step = [&](auto setc) {                             // multiplies sub-tile s (fragments in set `setc`), reads sub-tile s + 1, requests s + 5
        constexpr int set = decltype(setc)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (DLEAD - 2)) : "memory");       // my pieces of sub-tile s + 1 have landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                        // ... everybody's have; everybody has finished reading sub-tile s - 1's slot
        __builtin_amdgcn_sched_barrier(0);
        read_frags(set ^ 1);
        issue();
#pragma unroll
        for (int mb = 0; mb < 5; ++mb)
#pragma unroll
            for (int nb = 0; nb < 5; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[set][nb], af[set][mb], acc[mb][nb], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 10; ++k) {                       // the fragment reads and the five requests go out between the matrix instructions
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < nks; s += 2) {                       // (K % 64 == 0: an even number of sub-tiles)
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the wrap-around requests of the tail have landed: LDS is free for the epilogue
    __syncthreads();

    // ---- epilogue: alpha (acc + bias) in registers, residual(s) through the staging tile, whole-row 16-byte stores ---------------------------------------
    constexpr int OP = BN + 8, CPR = BN / 8;
    bf16_t* Os = reinterpret_cast<bf16_t*>(smem_raw);        // [160][168]
#pragma unroll
    for (int nb = 0; nb < 5; ++nb) {
        const int n = n0 + wc * 80 + nb * 16 + 4 * kq;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (P.bias && n < P.N) {
            const u32x2 t = *reinterpret_cast<const u32x2*>(P.bias + n);
            b4[0] = __uint_as_float(t[0] << 16); b4[1] = __uint_as_float(t[0] & 0xffff0000u);
            b4[2] = __uint_as_float(t[1] << 16); b4[3] = __uint_as_float(t[1] & 0xffff0000u);
        }
#pragma unroll
        for (int mb = 0; mb < 5; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mb][nb][j] = (acc[mb][nb][j] + b4[j]) * P.alpha;
    }
#pragma unroll 1
    for (int rz = 0; rz < 2; ++rz) {
        const bf16_t* rp = rz == 0 ? P.res : P.res2;
        if (rp == nullptr) continue;                         // (uniform)
        for (int c = tid; c < BM * CPR; c += 256) {
            const int r = c / CPR, ch = c - r * CPR;
            const bool ok = m0 + r < P.M && n0 + ch * 8 < P.N;
            *reinterpret_cast<u32x4*>(Os + r * OP + ch * 8) = ok ? *reinterpret_cast<const u32x4*>(rp + (m0 + r) * P.ldres + n0 + ch * 8) : u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < 5; ++mb)
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                const u32x2 t = *reinterpret_cast<const u32x2*>(Os + (wr * 80 + mb * 16 + l15) * OP + wc * 80 + nb * 16 + 4 * kq);
                acc[mb][nb][0] += __uint_as_float(t[0] << 16); acc[mb][nb][1] += __uint_as_float(t[0] & 0xffff0000u);
                acc[mb][nb][2] += __uint_as_float(t[1] << 16); acc[mb][nb][3] += __uint_as_float(t[1] & 0xffff0000u);
            }
        __syncthreads();
    }
#pragma unroll
    for (int mb = 0; mb < 5; ++mb)
#pragma unroll
        for (int nb = 0; nb < 5; ++nb)
            *reinterpret_cast<u32x2*>(Os + (wr * 80 + mb * 16 + l15) * OP + wc * 80 + nb * 16 + 4 * kq) =
                u32x2{pack_bf2(acc[mb][nb][0], acc[mb][nb][1]), pack_bf2(acc[mb][nb][2], acc[mb][nb][3])};
    __syncthreads();
    for (int c = tid; c < BM * CPR; c += 256) {
        const int r = c / CPR, ch = c - r * CPR;
        if (m0 + r < P.M && n0 + ch * 8 < P.N)
            *reinterpret_cast<u32x4*>(P.out + (m0 + r) * P.ldo + n0 + ch * 8) = *reinterpret_cast<const u32x4*>(Os + r * OP + ch * 8);
    }
}

}  // namespace

extern "C" int fmc_linear4_supported(int64_t M, int N, int K, int64_t ldx) {
    if (M < 1 || N < 8 || K < 64 || K % 64 || N % 8 || ldx % 8) return 0;
    if (((M - 1) * ldx + K) * 2 >= ((int64_t)1 << 31) || (int64_t)N * K * 2 >= ((int64_t)1 << 31)) return 0;
    return 1;
}

/* out[m, n] = alpha * (sum_k x[m, k] w[n, k] + bias[n]) + residual[m, n] + residual2[m, n]   (bf16; x rows ldx apart, residual(s) ldres, out ldo;
 * any of bias / residual / residual2 may be NULL, residual2 needs residual).  160 x 160 tiles, 4 waves, software-pipelined (csrc/gemm4.hip): the arm for the
 * M <= 5120 projections of the 10x16 / 5x8 levels.  K % 64 == 0, N % 8 == 0; edge tiles are masked. */
extern "C" int fmc_linear4_bf16(const void* x, const void* w, const void* bias, const void* residual, const void* residual2, void* out, int64_t M, int N,
                                int K, int64_t ldx, int64_t ldres, int64_t ldo, float alpha, void* stream) {
    if (!x || !w || !out) FMC_FAIL(FMC_E_NULL, "linear4_bf16: NULL tensor");
    if (!fmc_linear4_supported(M, N, K, ldx) || ldo % 8 || (residual && ldres % 8))
        FMC_FAIL(FMC_E_SHAPE, "linear4_bf16: need K %% 64 == 0, N %% 8 == 0, strides %% 8 == 0, operands < 2 GiB (M=%lld N=%d K=%d)", (long long)M, N, K);
    if (residual2 && !residual) FMC_FAIL(FMC_E_NULL, "linear4_bf16: residual2 needs residual");
    if (!fmc_aligned16(x) || !fmc_aligned16(w) || !fmc_aligned16(out) || (residual && !fmc_aligned16(residual)) || (residual2 && !fmc_aligned16(residual2)) ||
        (bias && (reinterpret_cast<uintptr_t>(bias) & 7)))
        FMC_FAIL(FMC_E_ALIGN, "linear4_bf16: tensors must be 16-byte aligned (bias 8)");
    G4Params P;
    P.a = (const bf16_t*)x; P.w = (const bf16_t*)w; P.bias = (const bf16_t*)bias; P.res = (const bf16_t*)residual; P.res2 = (const bf16_t*)residual2;
    P.out = (bf16_t*)out; P.M = M; P.N = N; P.K = K; P.lda = ldx; P.ldres = ldres; P.ldo = ldo; P.alpha = alpha;
    P.tiles_m = (int)((M + BM - 1) / BM); P.tiles_n = (N + BN - 1) / BN;
    P.a_bytes = ((M - 1) * ldx + K) * 2; P.w_bytes = (int64_t)N * K * 2;
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        raised = true;
    }
    hipLaunchKernelGGL(gemm4_kernel, dim3((unsigned)(P.tiles_m * P.tiles_n)), dim3(256), LDS_BYTES, (hipStream_t)stream, P);
    FMC_CHECK_LAUNCH("fmc_linear4_bf16");
    return 0;
}
