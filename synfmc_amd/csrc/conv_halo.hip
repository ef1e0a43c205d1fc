// Implicit-GEMM 3x3 convolution for gfx950 with the INPUT HALO TILE RESIDENT IN LDS and GroupNorm + SiLU applied while it is staged
// ("GN+SiLU prologue fused into implicit-GEMM 3x3 conv, temb add + GN stats epilogue").
// Replaces, on the FMC path, conv1 / conv2 of diffusers' ResnetBlock2D together with the `nonlinearity(norm(x))` in front of them
// (ctor args fmc/models/unet_blocks.py:306-317; `InflatedConv3d` fmc/models/resnet.py:16-24, `InflatedGroupNorm` :27-37) and the conv of
// Upsample2D (unet_blocks.py:625).
//
// Why a second conv kernel.  gemm160_kernel / gemm8_kernel (gemm_conv.hip) fetch the A operand tap by tap: every input pixel travels
// L2 -> LDS NINE times, and the loop is bound by its `buffer_load ... lds` instruction stream (~15 ns per 1-KiB request and CU, serial to the
// matrix work: gemm_conv.hip, "What bounds the loop").  Here a workgroup owns 10 x 32 output pixels of ONE image x 160 output channels:
//   * the 12 x 34 input halo of a 64-channel chunk is staged ONCE (global -> registers -> LDS) and serves all 9 taps x 2 k-halves: A traffic
//     falls 9 x 320 / 408 = 7x, and because it passes through registers the consumer's GroupNorm + SiLU is applied on the way
//     (`silu(x * a[c] + b[c])`, a / b per (image, channel) from fmc_groupnorm_coef) -- the normalised tensor is never written or read;
//   * only W is streamed by LDS-DMA: 10 one-KiB requests per 32-deep sub-tile and CU instead of 30 (the tile is 320 pixels x 160 channels:
//     the cheap operand gets the long side), from a copy of the filter pre-packed in the kernel's own sub-tile order;
//   * the MFMA loop itself is gemm160_kernel's: 8 waves = 4 (pixels) x 2 (channels), a wave = 80 x 80 outputs as 5 x 5
//     v_mfma_f32_16x16x32_bf16, one 32-deep sub-tile per phase {LOAD | barrier | 25 MFMAs | barrier}, the two wave halves one barrier apart.
// LDS: halo double buffer 2 x 53,248 B ([8 sixteen-byte channel groups][416 pixels] x 16 B: a fragment read of ANY tap is conflict free,
// see halo_px_slot) + W ring 5 x 10,240 B + GroupNorm coefficients 2 x 512 B = 158,720 B.
//
// Roofline: MFMA bound.  Algorithmic flops per launch = 2 * n_img*H*W * Cout * 9*Cin; algorithmic bytes = x + w + out (+ residual).
#include <type_traits>

#include "common.h"

namespace {

constexpr int TH = 10, TW = 32, BM = TH * TW, BN = 160;
constexpr int HWID = TW + 2, HPIX = (TH + 2) * HWID;       // 34 x 12 = 408 halo pixels
constexpr int NP = 416;                                    // pixels per channel-group plane (a multiple of 16: every plane starts at the same bank)
constexpr int PLANE = NP * 16;                             // bytes
constexpr int HALO = 8 * PLANE;                            // one 64-channel chunk of the halo: 53,248 B
constexpr int NBW = 5;                                     // W ring depth: three sub-tiles in flight (requested at LOAD(s - 3))
constexpr int WSUB = BN * 64;                              // one 32-deep W sub-tile: 160 rows x 64 B
constexpr int OFF_W = 2 * HALO, OFF_COEF = OFF_W + NBW * WSUB;
constexpr int LDS_BYTES = OFF_COEF + 2 * 512;              // 158,720
constexpr int NPIECE = 7;                                  // halo pieces (16 B) per thread and chunk: 51 blocks of 8 pixels x 8 channel groups over 8 waves
constexpr unsigned OOB = 0x80000000u;

struct CHParams {
    const bf16_t* x; const bf16_t* x2;      // input [n_img, Hs, Ws, c1] (+ second channel block [n_img, Hs, Ws, cin - c1]: the up blocks' skip connection)
    int c1;                                 // channels [0, c1) come from x; c1 == cin without x2; c1 % 64 == 0
    const bf16_t* w;                        // packed filter, see fmc_conv3x3_halo_pack_weight
    const bf16_t* bias; const bf16_t* temb; const bf16_t* res; bf16_t* out;
    int n_img, H, W, cin, cout, ups;        // H, W = OUTPUT size; ups: x is [n_img, H/2, W/2, .] read through a nearest 2x upsample
    int64_t temb_ld; int temb_div;
    const float* gn_coef; int gn_act;       // [n_img, cin, 2] (scale, shift) or NULL; gn_act: SiLU behind the affine map
    float* gn_part;                         // [n_img, tiles_y * tiles_x, 32, 2] partial (sum, sum of squares) of the ROUNDED outputs, or NULL
    int tiles_y, tiles_x, tiles_n;
    int64_t x_bytes, x2_bytes, w_bytes;
};

template <int I> using IC = std::integral_constant<int, I>;

__device__ __forceinline__ float silu_fast(float z) { return z * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z)); }

// halo staging pieces issued / written in sub-tile i of a chunk (18 sub-tiles): piece j is requested at LOAD(2 j) and written at LOAD(2 j + 3)
constexpr int nh(int i) { return (i >= 0 && i <= 12 && (i & 1) == 0) ? 1 : 0; }

// GN: 0 = plain convolution, 1 = operand silu(x * scale + shift), 2 = operand x * scale + shift (compile-time: the normalisation has to sit in
// the SAME basic block as the MFMAs it is interleaved with)
template <int GN>
__global__ __launch_bounds__(512, 2)
void conv_halo_kernel(const CHParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, wq = wave & 3;
    const int wc = wq & 1, wr = half * 2 + (wq >> 1);        // my 80 channels / my 80 pixels of the tile
    const int l15 = lane & 15, kq = lane >> 4;
    const bool clsA = wave < 2;                              // waves 0, 1 issue two W pieces per sub-tile, the others one

    // ---- my tile ---------------------------------------------------------------------------------------------------------------------
    int tile_p, tile_n;
    {
        const int total = P.n_img * P.tiles_y * P.tiles_x * P.tiles_n;
        const int id = blockIdx.x, q = total >> 3, r = total & 7, xcd = id & 7;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);      // XCD x owns a contiguous range of tiles:
        tile_p = lin / P.tiles_n;                                                                 // the channel tiles of a pixel tile are neighbours
        tile_n = lin - tile_p * P.tiles_n;                                                        // (its halo comes from that XCD's L2 once)
    }
    const int tpi = P.tiles_y * P.tiles_x;
    const int img = tile_p / tpi, tin = tile_p - img * tpi;
    const int y0 = (tin / P.tiles_x) * TH, x0 = (tin % P.tiles_x) * TW;
    const int n0 = tile_n * BN;
    const int nchunk = P.cin >> 6, nsub = nchunk * 18;

    // ---- halo staging: my seven (pixel, channel group) pieces --------------------------------------------------------------------------
    // block b = 8 j + wave holds halo pixels 8 b .. 8 b + 7 x 8 channel groups; lane = 8 g + p takes pixel p, channel group (p + g) & 7: the eight
    // lanes of a ds_write_b128 group write eight different pixels (bank groups) of eight planes, and the wave still covers 8 whole 128-byte lines
    const int Hs = P.ups ? P.H >> 1 : P.H, Ws = P.ups ? P.W >> 1 : P.W;
    const int pp = lane & 7, pg = ((lane & 7) + (lane >> 3)) & 7;
    int h_pix[NPIECE];                                       // source pixel index (image-major), or -1: outside the image / past the halo
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
        const int b = 8 * j + wave, px = 8 * b + pp;
        const int hy = px / HWID, hx = px - hy * HWID;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool in = b < HPIX / 8 && (unsigned)y < (unsigned)P.H && (unsigned)x < (unsigned)P.W;
        const int ys = P.ups ? y >> 1 : y, xs = P.ups ? x >> 1 : x;
        h_pix[j] = in ? (img * Hs + ys) * Ws + xs : -1;
    }
    const int h_lds = pg * PLANE + (8 * wave + pp) * 16;     // + j * 1024 (+ buffer): plane of my channel group, pixel 8 (8 j + wave) + pp
    const bool last_piece = wave < HPIX / 8 - 48;            // piece 6 exists for waves 0 .. 2 only (51 blocks of 8 pixels)
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, (int)P.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX2 = __builtin_amdgcn_make_buffer_rsrc((void*)(P.x2 ? P.x2 : P.x), 0, (int)(P.x2 ? P.x2_bytes : P.x_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)P.w_bytes, 0x00020000);
    const int c2 = P.cin - P.c1;
    u32x4 hreg[2];
    // chunk `ch64` of the input: which source, its row pitch and the channel offset inside it (all wave-uniform); branch-free per lane
    auto halo_load = [&](int j, int ch64, u32x4& dst) {
        const int cbeg = ch64 * 64;
        const bool second = cbeg >= P.c1, past = ch64 >= nchunk;
        const int pitch = past ? 0 : (second ? c2 : P.c1) * 2;
        const unsigned coff = past ? OOB : (unsigned)(((second ? cbeg - P.c1 : cbeg) + pg * 8) * 2);
        unsigned vo = (unsigned)(h_pix[j] * pitch) + coff;
        vo = h_pix[j] < 0 ? OOB : vo;
        const __amdgpu_buffer_rsrc_t rs = second ? rsX2 : rsX;
        dst = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, 0, 0));
    };
    // GroupNorm + SiLU of one staged piece, registers only (runs INSIDE an MFMA phase: its ~64 VALU instructions fill the issue slots between
    // the 25 matrix instructions instead of lengthening a LOAD phase); the 16 coefficients come from LDS ([64 channels][scale, shift] per chunk)
    auto halo_transform = [&](int j, int coefbuf, const u32x4& raw) -> u32x4 {
        if constexpr (GN == 0) return raw;
        const f32x4* src = reinterpret_cast<const f32x4*>(smem_raw + OFF_COEF + coefbuf * 512 + pg * 64);
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { f[2 * k] = __uint_as_float(raw[k] << 16); f[2 * k + 1] = __uint_as_float(raw[k] & 0xffff0000u); }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 t = src[k];
            const float z0 = fmaf(f[2 * k], t[0], t[1]), z1 = fmaf(f[2 * k + 1], t[2], t[3]);
            f[2 * k] = GN == 1 ? silu_fast(z0) : z0;
            f[2 * k + 1] = GN == 1 ? silu_fast(z1) : z1;
        }
        const bool in = h_pix[j] >= 0;                       // the convolution pads the NORMALISED tensor with zeros
        u32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = in ? pack_bf2(f[2 * k], f[2 * k + 1]) : 0u;
        return v;
    };
    auto halo_store = [&](int j, int buf, const u32x4& v) {
        if (j == NPIECE - 1 && !last_piece) return;          // (wave-uniform)
        *reinterpret_cast<u32x4*>(smem_raw + buf * HALO + h_lds + j * 1024) = v;
    };
    // GroupNorm coefficients of chunk `ch64`: 128 floats, every thread fetches one (4 x redundant: identical per-wave VMEM counts)
    auto coef_fetch = [&](int ch64) -> float {
        const int cc = ch64 < nchunk ? ch64 : nchunk - 1;
        const float* src = GN ? P.gn_coef + ((size_t)img * P.cin + cc * 64) * 2 + (tid & 127) : reinterpret_cast<const float*>(P.w);
        return *src;
    };

    // ---- W stream ------------------------------------------------------------------------------------------------------------------------
    // packed filter: [tiles_n][sub-tile s = (chunk, tap, half)][160 rows][32] with the LDS chunk swizzle already applied: piece p = KiB p of the block
    const int my_piece = clsA ? 2 * wave : wave + 2;
    const unsigned w_vo0 = (unsigned)(lane * 16 + my_piece * 1024);
    const int w_dst0 = OFF_W + my_piece * 1024;
    int iss_soff = tile_n * nsub * WSUB, iss_left = nsub, iss_slot = 0;
    auto w_issue = [&](auto cls) {
        unsigned char* dst = smem_raw + w_dst0 + iss_slot * WSUB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, (int)w_vo0, iss_soff, 0, 0);
        if constexpr (decltype(cls)::value)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + 1024), 16, (int)w_vo0, iss_soff + 1024, 0, 0);
        iss_soff += WSUB;
        if (--iss_left == 0) { iss_left = nsub; iss_soff = tile_n * nsub * WSUB; }      // (past the end the stream wraps to valid addresses: the counts stay exact)
        iss_slot = iss_slot + 1 == NBW ? 0 : iss_slot + 1;
    };

    // ---- fragments -----------------------------------------------------------------------------------------------------------------------
    f32x4 acc[5][5];
    bf16x8 wf[5], af[5];
    const int wfrag = OFF_W + ((wc * 80 + l15) * 32 + (kq ^ (3 * ((l15 >> 3) & 1))) * 8) * 2;     // + slot * WSUB + nb * 1024
    int afrag[5];                                            // + buf * HALO + half * 4 * PLANE + (ky * 34 + kx) * 16
#pragma unroll
    for (int mb = 0; mb < 5; ++mb) {
        const int idx = wr * 80 + mb * 16 + l15, ty = idx >> 5, tx = idx & 31;
        afrag[mb] = kq * PLANE + (ty * HWID + tx) * 16;
    }

    // ---- prologue: halo chunk 0, GroupNorm coefficients of chunks 0 and 1 ---------------------------------------------------------------------
    {
        u32x4 t[NPIECE];
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) halo_load(j, 0, t[j]);
        const float c0v = coef_fetch(0), c1v = coef_fetch(1);
        if (tid < 128) {
            *reinterpret_cast<float*>(smem_raw + OFF_COEF + tid * 4) = c0v;
            *reinterpret_cast<float*>(smem_raw + OFF_COEF + 512 + tid * 4) = c1v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) halo_store(j, 0, halo_transform(j, 0, t[j]));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- main loop, instantiated per wave class (two / one W request per sub-tile: the counted waits differ) -----------------------------------
    auto main_loop = [&](auto cls) {
        constexpr int NW = decltype(cls)::value ? 2 : 1;
        int rd_slot = 0;
        w_issue(cls); w_issue(cls); w_issue(cls);             // W sub-tiles 0 .. 2 in flight, sub-tile 0 retired and published
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NW) : "memory");
        __builtin_amdgcn_s_barrier();
        if (half == 1) __builtin_amdgcn_s_barrier();       // wave half 1 runs one barrier behind half 0
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);

        int cbuf = 0;                                        // halo (and coefficient) buffer of the chunk being multiplied
        float coef_next = 0.f;
        u32x4 hpk = {0u, 0u, 0u, 0u};
        for (int c = 0; c < nchunk; ++c) {
            const int abase = cbuf * HALO, nbuf = cbuf ^ 1;
            auto sub = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int tap = i >> 1, hk = i & 1, ky = tap / 3, kx = tap % 3;
                constexpr int aimm = hk * 4 * PLANE + (ky * HWID + kx) * 16;
                // 1. a staged piece of the NEXT chunk's halo: requested three sub-tiles ago, normalised during the last MFMA phase
                if constexpr (i >= 3 && i <= 15 && (i & 1) == 1) halo_store((i - 3) / 2, nbuf, hpk);
                if constexpr (i == 16) {                     // coefficients of chunk c + 2 into the buffer chunk c + 1's staging just stopped using
                    if (tid < 128) *reinterpret_cast<float*>(smem_raw + OFF_COEF + cbuf * 512 + tid * 4) = coef_next;
                }
                // 2. this sub-tile's fragments
                {
                    const unsigned char* Wp = smem_raw + wfrag + rd_slot * WSUB;
#pragma unroll
                    for (int nb = 0; nb < 5; ++nb) wf[nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wp + nb * 1024));
#pragma unroll
                    for (int mb = 0; mb < 5; ++mb)
                        af[mb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem_raw + abase + afrag[mb] + aimm));
                    rd_slot = rd_slot + 1 == NBW ? 0 : rd_slot + 1;
                }
                // 3. requests: one halo piece of the next chunk (even sub-tiles 0 .. 12), the coefficient of chunk c + 2, W sub-tile s + 3
                if constexpr (nh(i) == 1) halo_load(i / 2, c + 1, hreg[(i / 2) & 1]);
                if constexpr (i == 13) coef_next = coef_fetch(c + 2);
                w_issue(cls);
                // 4. counted wait: W sub-tile s + 1 (requested two LOADs ago) and everything older has landed
                {
                    constexpr int extra = nh(i - 1) + nh(i) + (i == 13 || i == 14 ? 1 : 0);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NW + extra) : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                // (piece j = (i - 2) / 2 was requested at LOAD(i - 2) in front of that phase's W request, which the wait above retired: it is here)
                if constexpr (i >= 2 && i <= 14 && (i & 1) == 0) hpk = halo_transform((i - 2) / 2, nbuf, hreg[((i - 2) / 2) & 1]);
#pragma unroll
                for (int mb = 0; mb < 5; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 5; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb], af[mb], acc[mb][nb], 0, 0, 0);
                if constexpr (GN != 0 && i >= 2 && i <= 14 && (i & 1) == 0) {      // one matrix instruction, then three of the piece's VALU / LDS-read instructions
#pragma unroll
                    for (int k = 0; k < 25; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x102, 3, 0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            };
            sub(IC<0>{}); sub(IC<1>{}); sub(IC<2>{}); sub(IC<3>{}); sub(IC<4>{}); sub(IC<5>{});
            sub(IC<6>{}); sub(IC<7>{}); sub(IC<8>{}); sub(IC<9>{}); sub(IC<10>{}); sub(IC<11>{});
            sub(IC<12>{}); sub(IC<13>{}); sub(IC<14>{}); sub(IC<15>{}); sub(IC<16>{}); sub(IC<17>{});
            cbuf = nbuf;
        }
    };
    if (clsA) main_loop(std::true_type{}); else main_loop(std::false_type{});
    if (half == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the wrap-around W requests of the tail have landed: LDS is free for the epilogue
    __syncthreads();

    // ---- epilogue (gemm160_kernel's: bias / temb in registers, residual through the staging tile, whole-row 16-byte stores) ---------------
    // my outputs: acc[mb][nb][j] = (tile pixel 80 wr + 16 mb + l15, tile channel 80 wc + 16 nb + 4 kq + j)
    constexpr int OP = BN + 8;                               // bf16 pitch of the staging rows (336 B)
    bf16_t* Os = reinterpret_cast<bf16_t*>(smem_raw);        // [320][OP] = 107,520 B
    constexpr int CPR = BN / 8;                              // 20 sixteen-byte chunks per row
    const int vrows = min(TH, P.H - y0) * TW;                // (rows of a tile that hangs over the image's last row are not stored)
    auto row_pixel = [&](int r) -> int64_t { return ((int64_t)img * P.H + y0 + (r >> 5)) * P.W + x0 + (r & 31); };
    {   // bias and time-embedding words of my five channel blocks: all requested before the first is used (one branch per block put each load behind its
        // own s_waitcnt vmcnt(0): ten dependent round trips)
        u32x2 bt[5], tt[5];
        if (P.bias) {
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) bt[nb] = *reinterpret_cast<const u32x2*>(P.bias + n0 + wc * 80 + nb * 16 + 4 * kq);
        }
        if (P.temb) {
            const bf16_t* trow = P.temb + (int64_t)(img / P.temb_div) * P.temb_ld + n0 + wc * 80 + 4 * kq;
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) tt[nb] = *reinterpret_cast<const u32x2*>(trow + nb * 16);
        }
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) {
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (P.bias) {
                b4[0] = __uint_as_float(bt[nb][0] << 16); b4[1] = __uint_as_float(bt[nb][0] & 0xffff0000u);
                b4[2] = __uint_as_float(bt[nb][1] << 16); b4[3] = __uint_as_float(bt[nb][1] & 0xffff0000u);
            }
            if (P.temb) {
                b4[0] += __uint_as_float(tt[nb][0] << 16); b4[1] += __uint_as_float(tt[nb][0] & 0xffff0000u);
                b4[2] += __uint_as_float(tt[nb][1] << 16); b4[3] += __uint_as_float(tt[nb][1] & 0xffff0000u);
            }
#pragma unroll
            for (int mb = 0; mb < 5; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mb][nb][j] += b4[j];
        }
    }
    if (P.res) {
        // the residual rows in two bursts of 7 loads per thread (320 x 20 chunks = 12.5 per thread) -- rolled, the loop was load -> s_waitcnt vmcnt(0) ->
        // ds_write per iteration: thirteen dependent round trips per tile
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 rv[7];
#pragma unroll
            for (int it = 0; it < 7; ++it) {
                const int cidx = min(tid + (7 * h + it) * 512, BM * CPR - 1), r = cidx / CPR, ch = cidx - r * CPR;
                rv[it] = *reinterpret_cast<const u32x4*>(P.res + row_pixel(r < vrows ? r : 0) * P.cout + n0 + ch * 8);
            }
#pragma unroll
            for (int it = 0; it < 7; ++it) {
                const int cidx = tid + (7 * h + it) * 512, r = cidx / CPR, ch = cidx - r * CPR;
                if (cidx < BM * CPR) *reinterpret_cast<u32x4*>(Os + r * OP + ch * 8) = rv[it];
            }
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
    // GroupNorm statistics of what I just produced (the ROUNDED values, as the consumer would read them): thread t sums group t % GT over rows
    // t / GT, t / GT + RP, ...; fixed summation order, one (sum, sum of squares) pair per (image, pixel tile, group) -- no atomics
    float gs = 0.f, gss = 0.f;
    const int cpg = P.cout >> 5, GT = BN / cpg, RP = 512 / GT;
    if (P.gn_part) {
        const int gl = tid % GT;
        for (int r = tid / GT; r < vrows; r += RP) {
            const unsigned* wsrc = reinterpret_cast<const unsigned*>(Os + r * OP + gl * cpg);
            for (int k = 0; k < cpg / 2; ++k) {
                const unsigned u = wsrc[k];
                const float a = __uint_as_float(u << 16), b = __uint_as_float(u & 0xffff0000u);
                gs += a + b;
                gss += a * a + b * b;
            }
        }
    }
    for (int cidx = tid; cidx < BM * CPR; cidx += 512) {
        const int r = cidx / CPR, ch = cidx - r * CPR;
        if (r < vrows)
            *reinterpret_cast<u32x4*>(P.out + row_pixel(r) * P.cout + n0 + ch * 8) = *reinterpret_cast<const u32x4*>(Os + r * OP + ch * 8);
    }
    if (P.gn_part) {
        float* red = reinterpret_cast<float*>(smem_raw + (size_t)BM * OP * 2);
        red[2 * tid] = gs;
        red[2 * tid + 1] = gss;
        __syncthreads();
        if (tid < GT) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < RP; ++k) { a += red[2 * (tid + k * GT)]; b += red[2 * (tid + k * GT) + 1]; }
            float* dst = P.gn_part + (((int64_t)img * tpi + tin) * 32 + (n0 / cpg + tid)) * 2;
            dst[0] = a;
            dst[1] = b;
        }
    }
}

// ---- filter packing: [Cout][3][3][Cin] (channels-last filter) -> [Cout / 160][Cin / 64][9 taps][2 halves][160 rows][32], the 16-byte chunks of a row
// already in their LDS places (physical chunk p of row r holds logical chunk p ^ (3 * ((r >> 3) & 1))) --------------------------------------------
__global__ __launch_bounds__(256) void conv_halo_pack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ dst, int cout, int cin) {
    const int64_t total = (int64_t)cout * 9 * cin / 8;      // 16-byte chunks
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = id;
        const int p = (int)(t & 3); t >>= 2;
        const int row = (int)(t % BN); t /= BN;
        const int hk = (int)(t & 1); t >>= 1;
        const int tap = (int)(t % 9); t /= 9;
        const int nchunk = cin >> 6;
        const int ch64 = (int)(t % nchunk); t /= nchunk;
        const int nt = (int)t;
        const int lc = p ^ (3 * ((row >> 3) & 1));
        const int64_t src = (((int64_t)(nt * BN + row) * 9 + tap) * cin + ch64 * 64 + hk * 32 + lc * 8);
        *reinterpret_cast<u32x4*>(dst + id * 8) = *reinterpret_cast<const u32x4*>(w + src);
    }
}

// ---- GroupNorm -> per-(image, channel) affine map: partial (sum, sum of squares) per (image, split, group) -> coef[n][c] = (rstd gamma_c,
// beta_c - mean rstd gamma_c); combined in fp64 like gn_apply_fwd_kernel (norm_kernels.hip) -------------------------------------------------------
__global__ __launch_bounds__(256) void gn_coef_kernel(const float* __restrict__ part, int splits, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ coef, float* __restrict__ stats,
                                                      int C, int G, double count, float eps) {
    __shared__ float s_mean[64], s_rstd[64];
    const int n = blockIdx.x;
    if ((int)threadIdx.x < G) {
        double a = 0.0, b = 0.0;
        for (int s = 0; s < splits; ++s) {
            const float* p = part + (((int64_t)n * splits + s) * G + threadIdx.x) * 2;
            a += (double)p[0];
            b += (double)p[1];
        }
        const double mean = a / count;
        double var = b / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        s_mean[threadIdx.x] = (float)mean;
        s_rstd[threadIdx.x] = rstd;
        if (stats) { stats[((int64_t)n * G + threadIdx.x) * 2] = (float)mean; stats[((int64_t)n * G + threadIdx.x) * 2 + 1] = rstd; }
    }
    __syncthreads();
    const int cpg = C / G;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const float a = s_rstd[g] * gamma[c];
        coef[((int64_t)n * C + c) * 2] = a;
        coef[((int64_t)n * C + c) * 2 + 1] = beta[c] - s_mean[g] * a;
    }
}

}  // namespace

extern "C" int64_t fmc_conv3x3_halo_packed_bytes(int Cin, int Cout) { return (int64_t)Cout * 9 * Cin * 2; }

extern "C" int fmc_conv3x3_halo_pack_weight(const void* w, void* dst, int Cin, int Cout, void* stream) {
    if (!w || !dst) FMC_FAIL(FMC_E_NULL, "conv3x3_halo_pack_weight: NULL pointer");
    if (Cin % 64 || Cout % BN) FMC_FAIL(FMC_E_SHAPE, "conv3x3_halo_pack_weight: Cin %% 64 / Cout %% 160 (Cin=%d Cout=%d)", Cin, Cout);
    if (!fmc_aligned16(w) || !fmc_aligned16(dst)) FMC_FAIL(FMC_E_ALIGN, "conv3x3_halo_pack_weight: pointers must be 16-byte aligned");
    const int64_t chunks = (int64_t)Cout * 9 * Cin / 8;
    const int grid = (int)((chunks + 255) / 256 < 4096 ? (chunks + 255) / 256 : 4096);
    hipLaunchKernelGGL(conv_halo_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, (bf16_t*)dst, Cout, Cin);
    FMC_CHECK_LAUNCH("fmc_conv3x3_halo_pack_weight");
    return 0;
}

extern "C" int fmc_conv3x3_halo_supported(int n_img, int H, int W, int Cin, int Cin1, int Cout, int upsample2x) {
    if (n_img < 1 || H < 1 || W < 1 || W % TW || Cin % 64 || Cout % BN) return 0;
    if (Cin1 <= 0 || Cin1 > Cin || Cin1 % 64) return 0;
    if (upsample2x && ((H | W) & 1)) return 0;
    const int64_t hs = upsample2x ? H / 2 : H, ws = upsample2x ? W / 2 : W;
    if ((int64_t)n_img * hs * ws * Cin1 * 2 >= (1ll << 31) || (int64_t)n_img * hs * ws * (Cin - Cin1) * 2 >= (1ll << 31)) return 0;
    if ((int64_t)Cout * 9 * Cin * 2 >= (1ll << 31)) return 0;
    return 1;
}

extern "C" int fmc_conv3x3_halo_bf16(const void* x, const void* x2, int Cin1, const void* w_packed, const void* bias, const void* temb,
                                     const void* residual, void* out, int n_img, int H, int W, int Cin, int Cout, int64_t temb_row_stride,
                                     int temb_img_div, int upsample2x, const float* gn_coef, int gn_act, float* gn_partials, void* stream) {
    if (!x || !w_packed || !out) FMC_FAIL(FMC_E_NULL, "conv3x3_halo: NULL x / w / out");
    if (!x2) Cin1 = Cin;
    if (!fmc_conv3x3_halo_supported(n_img, H, W, Cin, Cin1, Cout, upsample2x))
        FMC_FAIL(FMC_E_SHAPE, "conv3x3_halo: needs W %% 32 == 0, Cin %% 64 == 0 (both sources), Cout %% 160 == 0, operands < 2 GiB "
                 "(n=%d H=%d W=%d Cin=%d+%d Cout=%d ups=%d)", n_img, H, W, Cin1, Cin - Cin1, Cout, upsample2x);
    if (!fmc_aligned16(x) || !fmc_aligned16(w_packed) || !fmc_aligned16(out) || (x2 && !fmc_aligned16(x2)) || (residual && !fmc_aligned16(residual)) ||
        (bias && (reinterpret_cast<uintptr_t>(bias) & 7)) || (temb && ((reinterpret_cast<uintptr_t>(temb) & 7) || temb_row_stride % 4)))
        FMC_FAIL(FMC_E_ALIGN, "conv3x3_halo: x / w / out / residual must be 16-byte aligned, bias / temb rows 8-byte aligned");
    if (temb && temb_img_div < 1) FMC_FAIL(FMC_E_SHAPE, "conv3x3_halo: temb_img_div %d", temb_img_div);
    if (gn_partials && (Cout % 64 || BN % (Cout / 32)))     // a channel tile must hold whole GroupNorm groups of an even number of channels
        FMC_FAIL(FMC_E_SHAPE, "conv3x3_halo: the statistics epilogue needs Cout %% 64 == 0 and 160 %% (Cout / 32) == 0 (Cout=%d)", Cout);
    CHParams P;
    P.x = (const bf16_t*)x; P.x2 = (const bf16_t*)x2; P.c1 = Cin1;
    P.w = (const bf16_t*)w_packed; P.bias = (const bf16_t*)bias; P.temb = (const bf16_t*)temb; P.res = (const bf16_t*)residual; P.out = (bf16_t*)out;
    P.n_img = n_img; P.H = H; P.W = W; P.cin = Cin; P.cout = Cout; P.ups = upsample2x ? 1 : 0;
    P.temb_ld = temb_row_stride; P.temb_div = temb ? temb_img_div : 1;
    P.gn_coef = gn_coef; P.gn_act = gn_act; P.gn_part = gn_partials;
    P.tiles_y = (H + TH - 1) / TH; P.tiles_x = W / TW; P.tiles_n = Cout / BN;
    const int64_t hs = upsample2x ? H / 2 : H, ws = upsample2x ? W / 2 : W;
    P.x_bytes = (int64_t)n_img * hs * ws * Cin1 * 2; P.x2_bytes = (int64_t)n_img * hs * ws * (Cin - Cin1) * 2;
    P.w_bytes = (int64_t)Cout * 9 * Cin * 2;
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        raised = true;
    }
    const unsigned grid = (unsigned)(n_img * P.tiles_y * P.tiles_x * P.tiles_n);
    if (!gn_coef) hipLaunchKernelGGL(conv_halo_kernel<0>, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, P);
    else if (gn_act) hipLaunchKernelGGL(conv_halo_kernel<1>, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(conv_halo_kernel<2>, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, P);
    FMC_CHECK_LAUNCH("fmc_conv3x3_halo_bf16");
    return 0;
}

extern "C" int fmc_conv3x3_halo_tiles_per_image(int H, int W) { return ((H + TH - 1) / TH) * (W / TW); }

extern "C" int fmc_groupnorm_coef(const float* partials, int part_splits, const float* gamma, const float* beta, float* coef, float* stats,
                                  int N, int HW, int C, int G, float eps, void* stream) {
    if (!partials || !gamma || !beta || !coef) FMC_FAIL(FMC_E_NULL, "groupnorm_coef: NULL pointer");
    if (N < 1 || C < 1 || G < 1 || G > 64 || C % G || part_splits < 1) FMC_FAIL(FMC_E_SHAPE, "groupnorm_coef: N=%d C=%d G=%d splits=%d", N, C, G, part_splits);
    hipLaunchKernelGGL(gn_coef_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, partials, part_splits, gamma, beta, coef, stats, C, G,
                       (double)HW * (double)(C / G), eps);
    FMC_CHECK_LAUNCH("fmc_groupnorm_coef");
    return 0;
}
