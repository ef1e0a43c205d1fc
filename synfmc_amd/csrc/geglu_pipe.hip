// LayerNorm + GEGLU projection of diffusers' FeedForward with the gate SOFTWARE-PIPELINED under the matrix work (round 6), gfx950.
//
//     out[M][Cff] = (n W_v^T + b_v) * gelu(n W_g^T + b_g),  n = LayerNorm(h)         fmc/models/motion_module.py:295-299, diffusers 0.24 attention.py (GEGLU)
//
// Same arithmetic, operand layouts and packed weight (hip_ops.pack_geglu_frag80) as geglu_direct_kernel (temporal_block640.hip).  What that kernel left on
// the table, measured there (NOTEBOOK.md round 6: s_memtime stamps + SQ counters): at K = 320 the gate is NOT a small epilogue -- per 160-column chunk a
// wave issues 250 MFMAs (4000 matrix-pipe cycles) and then ~1000 VALU / transcendental instructions, two staging barriers and its stores (~11000 cycles)
// with the matrix pipe idle; the second workgroup of the CU runs the same phases at the same time, so the two do not cover each other (matrix pipe 35 % busy,
// VALU 15 %, the rest parked).  Setting priorities, staggering the workgroups or prefetching the bias moved 5 % between the phases.
// Here ONE wave per SIMD owns both jobs and interleaves them in its own instruction stream:
//   * two accumulator sets (2 x 100 registers, one wave per SIMD: 512 registers): while the MFMAs of chunk c fill one set, the gate of chunk c - 1 drains the
//     other -- 3 to 4 VALU instructions in the shadow of every 16-cycle MFMA, placed with sched_group_barrier;
//   * no staging tile and no barrier inside the chunk loop: a lane's 4 gated values (8 bytes) go out straight from registers with buffer stores -- in the
//     tile-major layout of the feed-forward's second GEMM (`out_blocked`) the 4 lanes of a row quarter fill 32 contiguous bytes; the waves of a workgroup
//     never wait for each other after the LayerNorm;
//   * the weight stream runs continuously across chunks through a 4-stage fragment ring (prefetch distance 2 k-steps).
// Workgroup = 4 waves = 80 rows resident in LDS (50 / 100 KiB at C = 320 / 640), a wave = 80 rows x 32 gated columns (64 weight rows: [v 0-15 | v 16-31 |
// g 0-15 | g 16-31], `hip_ops.pack_geglu_frag64`) per chunk of 128: both accumulator sets (2 x 80) stay in architectural VGPRs next to the gate's temporaries,
// the weight ring lives in AGPRs.  (The first build kept geglu_direct_kernel's 80-weight-row waves: 2 x 100 accumulators went to AGPRs and every chunk paid
// 300 v_accvgpr_read / _write to hand them to the VALU.)
#include <type_traits>

#include "common.h"

namespace {

struct GPParams {
    const bf16_t* h; bf16_t* out;                          // h [M][C]; out [M][Cff] row-major or tile-major [M / 160][Cff / 32][160][32]
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const bf16_t* w;                                       // [Cff / G column groups][C / 32 k-steps][G / 8 blocks][lane][8], G = 32 | 16 gated columns per wave (hip_ops.pack_geglu_frag)
    const bf16_t* bias;                                    // [2 Cff] (value | gate) or NULL
    int64_t M; int cff;
    int out_blocked;
};

__device__ __forceinline__ void gp_dma(const __amdgpu_buffer_rsrc_t& rs, unsigned voff, void* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, 0, 0, 0);
}
__device__ __forceinline__ f32x2_t gp_unpack2(unsigned w) { return f32x2_t{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
// v * gelu(g) on pairs (common.h: fmc_gelu_fast; two-element vectors so that the multiplies / fmas can pair into v_pk_* instructions)
__device__ __forceinline__ f32x2_t gp_gate2(f32x2_t v, f32x2_t g) {
    f32x2_t g2 = g * g;
    g2[0] = fminf(g2[0], 81.f); g2[1] = fminf(g2[1], 81.f);
    const f32x2_t c2 = f32x2_t{1.0142630198970437e-3f, 1.0142630198970437e-3f}, c1 = f32x2_t{-0.10677571594715118f, -0.10677571594715118f},
                  c0 = f32x2_t{-2.301121234893799f, -2.301121234893799f};
    const f32x2_t s = (c2 * g2 + c1) * g2 + c0;
    const f32x2_t t = g * s;
    f32x2_t e;
    e[0] = __builtin_amdgcn_exp2f(t[0]); e[1] = __builtin_amdgcn_exp2f(t[1]);
    e = e + f32x2_t{1.f, 1.f};
    f32x2_t r;
    r[0] = __builtin_amdgcn_rcpf(e[0]); r[1] = __builtin_amdgcn_rcpf(e[1]);
    return v * g * r;
}

// VALU / transcendental instructions placed behind EACH MFMA of a k-step that carries a gate unit (a unit = 4 values: ~55 instructions over 20 MFMAs)
#ifndef GP_VPM
#define GP_VPM 3
#endif
#ifndef GP_WD
#define GP_WD 3              // weight prefetch distance in k-steps (4-stage ring: at most 3)
#endif
// the 8-wave form lives on 256 registers per wave (2 x 80 accumulators): a weight prefetch distance of 1 is what fits without scratch (distance 2: 38 spilled
// registers, 3: 62; neither faster in a build without the gate); its partner wave on the SIMD covers the shorter prefetch.  Three gate instructions per MFMA
// slot (since the accumulators start from the bias the gate is ~45 instructions per unit): 145 us against 150 with two
#ifndef GP_WD8
#define GP_WD8 1
#endif
#ifndef GP_VPM8
#define GP_VPM8 3
#endif
// A-fragment lookahead of the two-waves-per-SIMD form, in row blocks (3 and 4 measured the same as 2 once the order is pinned, see `step`)
#ifndef GP_AFD8
#define GP_AFD8 2
#endif

// GC_ = channels; NW_ waves; MB_ = 16-row blocks of the resident tile (every wave runs all of them); NBK_ = 16-row weight blocks per wave and chunk (half
// values, half gates): <C, 4, 5, 4> = 80 rows, one wave per SIMD, 32 gated columns per wave; <320, 8, 10, 2> = 160 rows, two waves per SIMD, 16 gated columns
// per wave -- every weight fragment a wave loads then serves 160 rows: the fragment stream is 25 B / clk / CU against the 64 B / clk of the CU's vector-memory
// path (the 80-row forms of this and of geglu_direct_kernel need 50: what throttles their MFMA stream to half rate even with no gate at all).
template <int GC_, int NW_, int MB_, int NBK_>
__global__ __launch_bounds__(64 * NW_, NW_ / 4)
void geglu_pipe_kernel(const GPParams P) {
    constexpr int C = GC_, NW = NW_, MB = MB_, NBK = NBK_, NT = 64 * NW, ROWS = 16 * MB, CPR = C / 8, KS = C / 32, GW = 8 * NBK, GCOLS = GW * NW, LPR = CPR / 10,
                  WAVE_W = KS * NBK * 512, NP = NBK / 2, NU = MB * NP;
    constexpr bool AF2 = NW == 4;                                 // one wave per SIMD: next k-step's A fragments in a second register set
    constexpr int WD = NW == 8 ? GP_WD8 : GP_WD, VPMG = NW == 8 ? GP_VPM8 : GP_VPM;
    constexpr int NRG = NT / CPR;                                // row groups of the in-place normalisation pass
    static_assert((2 * KS) % 4 == 0, "the 4-stage weight ring must close over two chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* X = reinterpret_cast<bf16_t*>(smem_raw);             // [80][C], chunk c of row r at chunk c ^ ((r >> 1) & 7)
    float* stats = reinterpret_cast<float*>(X + ROWS * C);       // (mean, rstd) x 80 rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int xsw = (l15 >> 1) & 7;
    const int64_t m0 = (int64_t)blockIdx.x * ROWS;
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)P.h, 0, (int)(P.M * C * 2), 0x00020000);
    // ---- phase A: rows -> X, LayerNorm in place (geglu_direct_kernel's) ----
#pragma unroll
    for (int j = 0; j < (ROWS * CPR / 64 + NW - 1) / NW; ++j) {
        const int q = wave + NW * j;
        if (q < ROWS * CPR / 64) {
            const int idx = 64 * q + lane, r = idx / CPR, pc = idx - r * CPR, c = pc ^ ((r >> 1) & 7);
            gp_dma(rsH, (unsigned)(((m0 + r) * C + c * 8) * 2), X + 64 * q * 8);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int r = tid / LPR; r < ROWS; r += NT / LPR) {
        const int q = tid % LPR;
        const bf16_t* xr = X + r * C;
        u32x4 x4[10];
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            x4[i] = *reinterpret_cast<const u32x4*>(xr + (q + LPR * i) * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) s1 += __uint_as_float(x4[i][j] << 16) + __uint_as_float(x4[i][j] & 0xffff0000u);
        }
        s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
        if (LPR == 8) s1 += __shfl_xor(s1, 4);
        const float mean = s1 * (1.f / C);
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 10; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = __uint_as_float(x4[i][j] << 16) - mean, b = __uint_as_float(x4[i][j] & 0xffff0000u) - mean;
                s2 += a * a + b * b;
            }
        s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
        if (LPR == 8) s2 += __shfl_xor(s2, 4);
        if (q == 0) *reinterpret_cast<f32x2_t*>(stats + 2 * r) = f32x2_t{mean, rsqrtf(s2 * (1.f / C) + P.ln_eps)};
    }
    __syncthreads();
    if (tid < CPR * NRG) {
        const int nc = tid % CPR, nrg = tid / CPR;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(P.ln_gamma + nc * 8), g1 = *reinterpret_cast<const f32x4*>(P.ln_gamma + nc * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(P.ln_beta + nc * 8), b1 = *reinterpret_cast<const f32x4*>(P.ln_beta + nc * 8 + 4);
#pragma unroll 2
        for (int r = nrg; r < ROWS; r += NRG) {
            u32x4* px = reinterpret_cast<u32x4*>(X + r * C + (nc ^ ((r >> 1) & 7)) * 8);
            const u32x4 x4 = *px;
            const f32x2_t st = *reinterpret_cast<const f32x2_t*>(stats + 2 * r);
            const float m = st[0], rs = st[1];
            u32x4 o4;
            o4[0] = pack_bf2((__uint_as_float(x4[0] << 16) - m) * rs * g0[0] + b0[0], (__uint_as_float(x4[0] & 0xffff0000u) - m) * rs * g0[1] + b0[1]);
            o4[1] = pack_bf2((__uint_as_float(x4[1] << 16) - m) * rs * g0[2] + b0[2], (__uint_as_float(x4[1] & 0xffff0000u) - m) * rs * g0[3] + b0[3]);
            o4[2] = pack_bf2((__uint_as_float(x4[2] << 16) - m) * rs * g1[0] + b1[0], (__uint_as_float(x4[2] & 0xffff0000u) - m) * rs * g1[1] + b1[1]);
            o4[3] = pack_bf2((__uint_as_float(x4[3] << 16) - m) * rs * g1[2] + b1[2], (__uint_as_float(x4[3] & 0xffff0000u) - m) * rs * g1[3] + b1[3]);
            *px = o4;
        }
    }
    __syncthreads();                                              // X = LayerNorm(h): read-only from here on, the waves run free

    // ---- the pipelined chunk loop ----
    const int nchunks = P.cff / GCOLS;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((int64_t)2 * P.cff * C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)P.out, 0, (int)(P.M * P.cff * 2), 0x00020000);
    // (no bias: an empty range -- every load returns zero; chunks past the end read zeros the same way: the steady-state steps carry no branch, so that
    //  the whole k-step stays ONE scheduling region for the MFMA / gate interleave)
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(P.bias ? P.bias : P.w), 0, P.bias ? 2 * P.cff * 2 : 0, 0x00020000);
    f32x4 acc[2][MB][NBK];
    u32x4 wfr[4][NBK];
    u32x4 afb[2][AF2 ? MB : 1];                                   // (AF2) A fragments: this k-step's and the next one's
    u32x2 braw[2][NP];                                            // bias words [value | gate][p] of the NEXT chunk (requested one chunk ahead)
    f32x4 bq[NBK];                                                // ... of the chunk whose MFMAs are starting: its accumulators start from them
    int ooff[NP];                                                 // store offsets of the chunk being gated: (row l15, column 16 p + 4 kq) of my group                                          // [chunk parity][value | gate][block p]: bias words of the chunk whose MFMAs run in that parity
    auto wbase = [&](int ch) {
        int v = lane * 16 + (ch * NW + wave) * (WAVE_W * 2);
        asm volatile("" : "+v"(v));
        return v;
    };
    // store offset (bytes) of (tile row r, gated column col): tile-major [M / 160][Cff / 32][160][32] or row-major
    const int mblk = (int)(m0 / 160), mrow0 = (int)(m0 % 160);
    const int blk_base = mblk * (P.cff >> 5) * (160 * 32);
    const int m0i = (int)m0, blocked = P.out_blocked;
    auto out_off = [&](int r, int col) {
        const int ob = (blk_base + (col >> 5) * (160 * 32) + (mrow0 + r) * 32 + (col & 31)) * 2, orm = ((m0i + r) * P.cff + col) * 2;
        return blocked ? ob : orm;
    };
    // the accumulators START from the bias (the first k-step's MFMAs take it as their C operand: lane (row l15, kq) holds columns 4 kq .. + 3 of a block, the
    // same four words for every row block), so the gate adds nothing: ~40 packed adds per chunk less, and no bias registers alive during the gate
    auto load_bias = [&](int ch) {
        const int gc0 = ch * GCOLS + wave * GW;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c0 = gc0 + 16 * p + 4 * kq;
            braw[0][p] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsB, c0 * 2, 0, 0));
            braw[1][p] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsB, (P.cff + c0) * 2, 0, 0));
        }
    };
    auto take_bias = [&]() {
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
            const u32x2 w2 = braw[nb / NP][nb % NP];
            const f32x2_t lo = gp_unpack2(w2[0]), hi = gp_unpack2(w2[1]);
            bq[nb] = f32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    };
    // per gated chunk, once: the store offsets of row block 0 (row blocks are `rstep` bytes apart)
    const int rstep = blocked ? 16 * 32 * 2 : 16 * P.cff * 2;
    auto gate_setup = [&](int ch) {
        const int gc0 = ch * GCOLS + wave * GW;
#pragma unroll
        for (int p = 0; p < NP; ++p) ooff[p] = out_off(l15, gc0 + 16 * p + 4 * kq);
    };
    // gate unit u = (row block mb = u / NP, column block p = u % NP) out of accumulator set `par`: 4 values of my lane -> one 8-byte store
    auto gate_unit = [&](auto par_c, auto u_c, int ch) {
        constexpr int par = decltype(par_c)::value, u = decltype(u_c)::value, mb = u / NP, p = u % NP;
        const f32x2_t v01 = f32x2_t{acc[par][mb][p][0], acc[par][mb][p][1]}, v23 = f32x2_t{acc[par][mb][p][2], acc[par][mb][p][3]};
        const f32x2_t g01 = f32x2_t{acc[par][mb][NP + p][0], acc[par][mb][NP + p][1]}, g23 = f32x2_t{acc[par][mb][NP + p][2], acc[par][mb][NP + p][3]};
#ifdef GP_NOGATE   // diagnostic build: the MFMA stream with the cheapest possible consumer (wrong results)
        const f32x2_t o01 = v01 + g01, o23 = v23 + g23;
#else
        const f32x2_t o01 = gp_gate2(v01, g01);
        const f32x2_t o23 = gp_gate2(v23, g23);
#endif
        const u32x2 o = u32x2{pack_bf2(o01[0], o01[1]), pack_bf2(o23[0], o23[1])};
        __builtin_amdgcn_raw_buffer_store_b64(o, rsO, ooff[p], mb * rstep, 0);
    };
    // step S of a two-chunk round (chunks c0 = even, c0 + 1): MFMAs of (chunk c0 + S / KS, k-step S % KS) into accumulator set S / KS, the weight request of
    // step S + 2, and the gate units of the chunk BEFORE (set 1 - S / KS) that fall on this k-step
    auto step = [&](auto s_c, auto mode_c, int c0, int wl0, int wl1, int wl2) {
        constexpr int S = decltype(s_c)::value, par = S / KS, g = S % KS, MODE = decltype(mode_c)::value;
        constexpr bool mma = MODE & 1, gate = MODE & 2;
        const int ch = c0 + par;
        {   // weight fragments of step S + GP_WD (this round's or the first ones of the next round's): the stage step S - 1 has just left (GP_WD = 3)
            constexpr int S2 = S + WD, r2 = S2 / KS, g2 = S2 % KS;   // r2: 0, 1 = this round's chunks, 2 = the next round's first
            const int wl = r2 == 0 ? wl0 : (r2 == 1 ? wl1 : wl2);   // (past the last chunk: out of range, zeros)
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) wfr[S2 % 4][nb] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wl, (g2 * NBK + nb) * 1024, 0);
        }
        if constexpr (g == 0 && mma) {
            take_bias();                                          // (requested a whole chunk ago)
            load_bias(ch + 1);                                    // past the last chunk: out of range, zeros
        }
        if constexpr (g == 0 && gate) gate_setup(ch - 1);
        int kqx = kq ^ xsw, xrow_o = l15 * C;
        asm volatile("" : "+v"(kqx), "+v"(xrow_o));
        // AF2: the A fragments of THIS k-step were read during the previous one (afb[S & 1]); the next k-step's (k-step 0 again behind the last: X is the same
        // for every chunk) are read now, one in front of each row block's MFMAs -- one wave per SIMD has nobody to cover an LDS round trip.  Otherwise
        // (two waves per SIMD): three registers sets, two row blocks ahead.
        constexpr int gn = (g + 1) % KS;
        const int xn = xrow_o + ((gn >> 1) * 8 + (((gn & 1) * 4) ^ kqx)) * 8;
        const int xo = xrow_o + ((g >> 1) * 8 + (((g & 1) * 4) ^ kqx)) * 8;
        __builtin_amdgcn_sched_barrier(0);
        // gate units of this k-step: unit u runs in k-step (u * KS) / NU
        constexpr int u_lo = (g * NU + KS - 1) / KS, u_hi = ((g + 1) * NU + KS - 1) / KS;      // units u with u_lo <= u < u_hi (at most one)
        constexpr int AFD = GP_AFD8;
        u32x4 af3[AFD + 1];
        if constexpr (mma && !AF2) {
#pragma unroll
            for (int i = 0; i < AFD; ++i) af3[i] = *reinterpret_cast<const u32x4*>(X + i * 16 * C + xo);
        }
        if constexpr (mma) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                union { bf16x8 v; u32x4 u; } a;
                if constexpr (AF2) {
                    afb[(S + 1) & 1][mb] = *reinterpret_cast<const u32x4*>(X + mb * 16 * C + xn);
                    a.u = afb[S & 1][mb];
                } else {
                    if (mb + AFD < MB) af3[(mb + AFD) % (AFD + 1)] = *reinterpret_cast<const u32x4*>(X + (mb + AFD) * 16 * C + xo);
                    // pin the fragment pipeline: LDS reads and MFMAs may not cross this point (VALU / SALU / transcendentals -- the gate -- may).  Left to the
                    // scheduler's register-pressure heuristic the ring collapsed into ONE register quad: read, lgkmcnt(0), two MFMAs, read, ... -- a full
                    // LDS round trip per row block (found in the ISA: 75 `s_waitcnt lgkmcnt(0)` per chunk; 161 -> 153 us once pinned).  Carrying the ring
                    // across k-step boundaries as well costs 47 spilled registers: not done.
                    __builtin_amdgcn_sched_barrier(0x406);
                    a.u = af3[mb % (AFD + 1)];
                }
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb) {
                    union { bf16x8 v; u32x4 u; } w;
                    w.u = wfr[S % 4][nb];
                    if constexpr (g == 0) acc[par][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, a.v, bq[nb], 0, 0, 0);
                    else acc[par][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, a.v, acc[par][mb][nb], 0, 0, 0);
                }
                if constexpr (!AF2) __builtin_amdgcn_sched_barrier(0x406);
            }
        }
        if constexpr (gate) {
            static_assert(u_hi - u_lo <= 1, "one gate unit per k-step");
            if constexpr (u_lo < u_hi) gate_unit(std::integral_constant<int, 1 - par>{}, std::integral_constant<int, u_lo>{}, ch - 1);
        }
        // placement.  An in-order wave overlaps VALU with its own MFMAs only at the granularity of ONE matrix instruction (behind several back-to-back MFMAs
        // the issue port waits for the pipe, 16 cycles each, and a VALU block behind them runs in the open): one fragment read per row block, then per
        // MFMA GP_VPM gate instructions in its shadow.
        constexpr int NUS = gate ? u_hi - u_lo : 0, VPM = NUS == 0 ? 0 : VPMG;
        if constexpr (mma) {
            if constexpr (!AF2) __builtin_amdgcn_sched_group_barrier(0x100, AFD, 0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                if (AF2 || mb + AFD < MB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (VPM) __builtin_amdgcn_sched_group_barrier(0x402, VPM, 0);
                }
            }
        }
    };

    load_bias(0);
    int wl0 = wbase(0), wl1 = wbase(1), wl2 = wbase(2);
#pragma unroll
    for (int st = 0; st < WD; ++st)
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) wfr[st][nb] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wl0, (st * NBK + nb) * 1024, 0);
    if constexpr (AF2) {
        int kqx = kq ^ xsw, xrow_o = l15 * C;
        asm volatile("" : "+v"(kqx), "+v"(xrow_o));
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) afb[0][mb] = *reinterpret_cast<const u32x4*>(X + mb * 16 * C + xrow_o + kqx * 8);
    }
    // one chunk = KS steps in accumulator set PAR with MODE = 1 (MFMAs only: the first chunk), 3 (MFMAs + the previous chunk's gate), 2 (gate only: behind the last)
    auto chunk = [&](auto par_c, auto mode_c, int c0) {
        constexpr int B = decltype(par_c)::value * KS;
#define GP_S(S) step(std::integral_constant<int, B + S>{}, mode_c, c0, wl0, wl1, wl2)
        GP_S(0); GP_S(1); GP_S(2); GP_S(3); GP_S(4); GP_S(5); GP_S(6); GP_S(7); GP_S(8); GP_S(9);
        if constexpr (KS == 20) { GP_S(10); GP_S(11); GP_S(12); GP_S(13); GP_S(14); GP_S(15); GP_S(16); GP_S(17); GP_S(18); GP_S(19); }
#undef GP_S
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // chunk 0 (no gate yet), then pairs, then the gate of the last chunk
    chunk(I0{}, I1{}, 0);
    if (nchunks > 1) chunk(I1{}, I3{}, 0); else chunk(I1{}, I2{}, 0);
#pragma unroll 1
    for (int c0 = 2; c0 < nchunks; c0 += 2) {
        wl0 = wbase(c0); wl1 = wbase(c0 + 1); wl2 = wbase(c0 + 2);
        chunk(I0{}, I3{}, c0);
        if (c0 + 1 < nchunks) chunk(I1{}, I3{}, c0); else chunk(I1{}, I2{}, c0);
    }
    if (nchunks > 1 && nchunks % 2 == 0) {                         // the last chunk ran in set 1: its gate in a set-0 pass without MFMAs
        wl0 = wl1 = wl2 = wbase(nchunks);
        chunk(I0{}, I2{}, nchunks);
    }
}

}  // namespace

// variant: 0 = 80-row tiles, 4 waves, 32 gated columns per wave (C = 320 | 640; weights packed with group 32); 1 = 160-row tiles, 8 waves, 16 gated columns
// per wave (C = 320 only, M % 160 == 0; weights packed with group 16)
extern "C" int fmc_geglu_pipe_supported(int64_t M, int cff, int C, int variant) {
    const bool common = M > 0 && cff > 0 && cff % 128 == 0 && M * (int64_t)cff * 2 < ((int64_t)1 << 31) && (int64_t)2 * cff * C * 2 < ((int64_t)1 << 31) &&
                        M * (int64_t)C * 2 < ((int64_t)1 << 31);
    if (variant == 1) return common && C == 320 && M % 160 == 0;
    return common && (C == 320 || C == 640) && M % 80 == 0;
}

extern "C" int fmc_geglu_pipe_ln_bf16(const void* h, void* out, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w_packed, const void* bias,
                                      int64_t M, int cff, int C, int out_blocked, int variant, void* stream) {
    if (!h || !out || !ln_gamma || !ln_beta || !w_packed) FMC_FAIL(FMC_E_NULL, "geglu_pipe_ln_bf16: NULL tensor");
    if ((variant != 0 && variant != 1) || !fmc_geglu_pipe_supported(M, cff, C, variant) || (out_blocked && (M % 160 || cff % 32)))
        FMC_FAIL(FMC_E_SHAPE, "geglu_pipe_ln_bf16: C in {320, 640} (variant 1: 320), M %% 80 == 0 (%% 160 tile-major / variant 1), cff %% 128 == 0, tensors below "
                              "2 GiB (got M=%lld cff=%d C=%d variant=%d)", (long long)M, cff, C, variant);
    if (!fmc_aligned16(h) || !fmc_aligned16(out) || !fmc_aligned16(w_packed) || !fmc_aligned16(ln_gamma) || !fmc_aligned16(ln_beta) || (bias && ((uintptr_t)bias & 7)))
        FMC_FAIL(FMC_E_ALIGN, "geglu_pipe_ln_bf16: tensors must be 16-byte aligned");
    GPParams P;
    P.h = (const bf16_t*)h; P.out = (bf16_t*)out; P.ln_gamma = ln_gamma; P.ln_beta = ln_beta; P.ln_eps = ln_eps;
    P.w = (const bf16_t*)w_packed; P.bias = (const bf16_t*)bias; P.M = M; P.cff = cff; P.out_blocked = out_blocked != 0;
    constexpr int lds320 = 80 * 320 * 2 + 640, lds640 = 80 * 640 * 2 + 640, lds320w = 160 * 320 * 2 + 1280;
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&geglu_pipe_kernel<320, 4, 5, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds320);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&geglu_pipe_kernel<640, 4, 5, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds640);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&geglu_pipe_kernel<320, 8, 10, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds320w);
        raised = true;
    }
    hipStream_t st = (hipStream_t)stream;
    if (variant == 1) hipLaunchKernelGGL((geglu_pipe_kernel<320, 8, 10, 2>), dim3((unsigned)(M / 160)), dim3(512), lds320w, st, P);
    else if (C == 320) hipLaunchKernelGGL((geglu_pipe_kernel<320, 4, 5, 4>), dim3((unsigned)(M / 80)), dim3(256), lds320, st, P);
    else hipLaunchKernelGGL((geglu_pipe_kernel<640, 4, 5, 4>), dim3((unsigned)(M / 80)), dim3(256), lds640, st, P);
    FMC_CHECK_LAUNCH("fmc_geglu_pipe_ln_bf16");
    return 0;
}
