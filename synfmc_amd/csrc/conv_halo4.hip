// The halo-resident 3x3 convolution (conv_halo.hip) for the SMALL feature maps of the U-Net -- 10x16 and 5x8 pixels per image at the two inner levels
// (diffusers ResnetBlock2D / Upsample2D convs of down_blocks.2-3, mid_block, up_blocks.0-1; ctor args fmc/models/unet_blocks.py:306-317, :625).
//
// Why a second geometry.  At 1280 channels those convolutions are 32 images x 160 (or 40) pixels: M = 5120 (1280) rows against K = 11520 .. 23040.
// conv_halo_kernel's 10 x 32-pixel x 160-channel tiles do not exist there, and the ring kernel's k-lockstep stream-K pass + finishing launch ran them
// at 0.17 - 0.36 of the MFMA peak (profiles/r05d_kernel_by_grid.md: 27 launches, 3.7 ms of a 28.9 ms step).  Here
//   * a tile is 320 output pixels = NB ROW BLOCKS of TH x TW pixels (TW = 16 where the image width is a multiple of 16: two 10 x 16 blocks = two
//     whole images at the 10x16 level; TW = 8 otherwise: eight 5 x 8 blocks = eight images at the 5x8 level; wider / taller images are cut into
//     such blocks) x 80 output channels: 256 tiles at the 10x16 level = one per CU, no split, no finishing pass;
//   * the input halos of the tile's row blocks ((TH + 2) x (TW + 2) pixels each, 64 channels at a time) are staged ONCE per chunk through registers
//     into LDS planes [8 channel groups][pixels] (any tap's fragment read is conflict free for TW >= 16, 2-way for TW = 8) and serve 9 taps x 2
//     k-halves; only W is streamed by LDS-DMA: 5 one-KiB requests per 32-deep sub-tile and CU;
//   * 4 waves, one per SIMD, each 80 pixels x 80 channels (5 x 5 v_mfma_f32_16x16x32_bf16): a wave keeps its matrix pipe busy alone (a 16x16x32
//     MFMA issues every 16 cycles from one wave) by software pipelining -- the fragments of sub-tile s + 1 are read into a second register set
//     while the 25 MFMAs of sub-tile s run -- with ONE barrier per sub-tile (it publishes the W sub-tile the next reads need).
// LDS: halo double buffer 2 x 8 x NP x 16 B + W ring 5 x 5 KiB (TW = 16: 135,168 B; TW = 8: 163,840 - the ring shrinks to 4).
// Roofline: MFMA bound.  Algorithmic flops per launch = 2 * n_img*H*W * Cout * 9*Cin.
#include <type_traits>

#include "common.h"

namespace {

constexpr int BM = 320;
#ifndef FMC_C4_INTERLEAVE
#define FMC_C4_INTERLEAVE 1
#endif
constexpr bool INTERLEAVE = FMC_C4_INTERLEAVE != 0;          // A/B switch (compile time): C - E issued between the MFMAs instead of in front of them
constexpr unsigned OOB = 0x80000000u;

// NWV waves: 4 (one per SIMD, 320 pixels x 80 channels) or 8 (two per SIMD, 320 x 160: wave = pixel block wave % 4, channel block wave / 4)
template <int TW, int NWV> struct Geo {
    static constexpr int BN = 20 * NWV, NT = 64 * NWV;
    static constexpr int WSUB = BN * 64;                   // one 32-deep W sub-tile: BN rows x 64 B (5 / 10 KiB)
    static constexpr int WPIECES = BN / 16;                // its 1-KiB pieces: 5 / 10; wave w issues piece w, the first WPIECES - NWV waves a second one
    static constexpr int TH = TW == 8 ? 5 : 10;
    static constexpr int RB = TH * TW;                     // pixels per row block
    static constexpr int NB = BM / RB;                     // row blocks per tile: 1 / 2 / 8
    static constexpr int HW_ = TW + 2, HB = (TH + 2) * HW_;    // halo pitch / halo pixels per row block
    static constexpr int HPIX = NB * HB;                   // 408 / 432 / 560
    static constexpr int NP = (HPIX + 15) / 16 * 16;       // pixels per channel-group plane (a multiple of 16: every plane starts at the same bank)
    static constexpr int PLANE = NP * 16, HALO = 8 * PLANE;
    static constexpr int NBW = (2 * HALO + 5 * WSUB <= 163840) ? 5 : 4;      // W ring depth
    static constexpr int DW = NBW - 1;                     // W sub-tile s + DW is requested at LOAD(s), into the slot sub-tile s - 1 was read from
    static constexpr int OFF_W = 2 * HALO;
    static constexpr int LDS_BYTES = OFF_W + NBW * WSUB;
    static constexpr int NPIECE = (HPIX / 8 + NWV - 1) / NWV;      // halo pieces (16 B) per thread and chunk: blocks of 8 pixels x 8 channel groups over the waves
    static_assert(HPIX % 8 == 0 && LDS_BYTES <= 163840 && NPIECE <= 18, "geometry");
};

struct C4Params {
    const bf16_t* x; const bf16_t* x2; int c1;              // as conv_halo.hip: two-source input, channels [0, c1) from x
    const bf16_t* w;                                        // fmc_conv3x3_halo4_pack_weight: [Cout / 80][Cin / 64][9][2][80 rows][32], chunk-swizzled
    const bf16_t* bias; const bf16_t* temb; const bf16_t* res; bf16_t* out;
    int n_img, H, W, cin, cout, ups;
    int64_t temb_ld; int temb_div;
    float* gn_part;                                         // [n_img, tiles_y, 32, 2] partial sums of the rounded outputs (one split per row block), or NULL
    int tiles_y, tiles_x, tiles_p, tiles_n;                 // row blocks per image column / per image row, pixel tiles, channel tiles
    int splits; float* ws;                                  // split-K: the 64-channel chunks are dealt to `splits` workgroups per tile, which leave fp32 partial
                                                            //   sums in ws[split][pixel][Cout]; conv_halo4_finish_kernel adds them (fixed order) and runs the epilogue
    int64_t x_bytes, x2_bytes, w_bytes;
    int xcd_pc;                                             // tile order: the 8 XCDs as xcd_pc filter-slice groups x 8 / xcd_pc pixel groups (launch_c4), 0 = contiguous ranges
};

template <int I> using IC = std::integral_constant<int, I>;

// halo pieces requested in sub-tile i of a chunk (requested at LOAD(i), written at LOAD(i + 3)): PPS = 2 per sub-tile where a thread has more than
// 12 pieces per chunk (4 waves), else one (8 waves: fewer staging registers in flight)
template <int NPIECE> constexpr int pps() { return NPIECE > 12 ? 2 : 1; }
template <int NPIECE> constexpr int nh(int i) {
    return (i >= 0 && pps<NPIECE>() * i < NPIECE) ? (pps<NPIECE>() == 2 && 2 * i + 1 < NPIECE ? 2 : 1) : 0;
}

template <int TW, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV / 4)
void conv_halo4_kernel(const C4Params P) {
    using G = Geo<TW, NWV>;
    constexpr int BN = G::BN, NT = G::NT, WSUB = G::WSUB;
    constexpr int TH = G::TH, RB = G::RB, NB = G::NB, HW_ = G::HW_, HB = G::HB, PLANE = G::PLANE, HALO = G::HALO, NBW = G::NBW, OFF_W = G::OFF_W;
    constexpr int NPIECE = G::NPIECE, DW = G::DW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int wp = wave & 3, wn = wave >> 2;                 // my 80 pixels / my 80 channels of the tile
    const bool clsA = wave < G::WPIECES - NWV;               // these waves issue two W pieces per sub-tile (w and NWV + w), the others one

    // ---- my tile.  Workgroup id -> XCD id & 7 (round robin).  The filter of these levels (29.5 MB at 1280 -> 1280) is larger than the input (13 MB at
    // 10x16, 3.3 MB at 5x8) and every pixel tile streams its channel tile's slice of it, so WHICH tiles share an XCD (and its L2) decides how often the
    // filter crosses the fabric: contiguous tile ranges (two pixel tiles x all 16 channel tiles per XCD at 10x16) bring the WHOLE filter into every L2 --
    // 302 MB fetched per launch for 56 MB of operands (profiles/r06fin_step_hbm_traffic_by_kernel.md).  xcd_pc > 0: the XCDs form a grid of xcd_pc
    // filter-slice groups x 8 / xcd_pc pixel groups; launch_c4 picks the split that minimises filter * pixel groups + input * slice groups. ----------
    int tile_p, tile_n, split;
    {
        const int total = P.tiles_p * P.tiles_n * P.splits;
        const int id = blockIdx.x, xcd = id & 7;
        if (P.xcd_pc > 0) {
            const int j = id >> 3, pc = P.xcd_pc;
            const int tnl = P.tiles_n * P.splits / pc, tpl = P.tiles_p / (8 / pc);      // (filter slice = (channel tile, split)) per slice group, pixel tiles per pixel group
            const int tnv = (xcd % pc) * tnl + j % tnl;
            tile_p = (xcd / pc) * tpl + j / tnl;
            split = tnv % P.splits;
            tile_n = tnv / P.splits;
        } else {
            const int q = total >> 3, r = total & 7;
            int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
            split = lin % P.splits;                          // (the splits of a tile are neighbours)
            lin /= P.splits;
            tile_p = lin / P.tiles_n;
            tile_n = lin - tile_p * P.tiles_n;
        }
    }
    const int n0 = tile_n * BN;
    const int nchunk_all = P.cin >> 6;
    const int per = (nchunk_all + P.splits - 1) / P.splits;
    const int ck0 = split * per, nchunk = max(0, min(nchunk_all, ck0 + per) - ck0);      // my chunks ck0 .. ck0 + nchunk - 1
    const int nsub = max(nchunk, 1) * 18;
    const int tpi = P.tiles_y * P.tiles_x;                   // row blocks per image: block rb = (image rb / tpi, rows ((rb % tpi) / tiles_x) TH .., columns ((rb % tpi) % tiles_x) TW ..)
    const int rb0 = tile_p * NB, rb_total = P.n_img * tpi;   // my row blocks rb0 .. rb0 + NB - 1

    // ---- halo staging: block b = 4 j + wave holds halo pixels 8 b .. 8 b + 7 x 8 channel groups; lane = 8 g + p takes pixel p, group (p + g) & 7 -----
    const int Hs = P.ups ? P.H >> 1 : P.H, Ws = P.ups ? P.W >> 1 : P.W;
    const int pp = lane & 7, pg = ((lane & 7) + (lane >> 3)) & 7;
    int h_pix[NPIECE];                                       // source pixel index (image-major), or -1
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
        const int b = NWV * j + wave, px = 8 * b + pp;
        const int blk = px / HB, rem = px - blk * HB;
        const int hy = rem / HW_, hx = rem - hy * HW_;
        const int rb = rb0 + blk;
        const int img = rb / tpi, rin = rb - img * tpi, yb = rin / P.tiles_x, xb = rin - yb * P.tiles_x;
        const int y = yb * TH - 1 + hy, x = xb * TW - 1 + hx;
        const bool in = px < G::HPIX && rb < rb_total && (unsigned)y < (unsigned)P.H && (unsigned)x < (unsigned)P.W;
        const int ys = P.ups ? y >> 1 : y, xs = P.ups ? x >> 1 : x;
        h_pix[j] = in ? (img * Hs + ys) * Ws + xs : -1;
    }
    const int h_lds = pg * PLANE + (8 * wave + pp) * 16;     // + j * 128 NWV (+ buffer)
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, (int)P.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX2 = __builtin_amdgcn_make_buffer_rsrc((void*)(P.x2 ? P.x2 : P.x), 0, (int)(P.x2 ? P.x2_bytes : P.x_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)P.w_bytes, 0x00020000);
    const int c2 = P.cin - P.c1;
    auto halo_load = [&](int j, int crel) -> u32x4 {             // crel: chunk relative to my range
        const int cbeg = (ck0 + crel) * 64;
        const bool second = cbeg >= P.c1, past = crel >= nchunk;
        const int pitch = past ? 0 : (second ? c2 : P.c1) * 2;
        const unsigned coff = past ? OOB : (unsigned)(((second ? cbeg - P.c1 : cbeg) + pg * 8) * 2);
        unsigned vo = (unsigned)(h_pix[j] * pitch) + coff;
        vo = h_pix[j] < 0 ? OOB : vo;
        const __amdgpu_buffer_rsrc_t rs = second ? rsX2 : rsX;
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, 0, 0));
    };
    auto halo_store = [&](int j, int buf, const u32x4& v) {
        if (8 * (NWV * j + wave) < G::HPIX)                   // (wave-uniform: the last piece index exists for the first waves only)
            *reinterpret_cast<u32x4*>(smem_raw + buf * HALO + h_lds + j * (128 * NWV)) = v;
    };

    // ---- W stream: piece p = KiB p of the 5-KiB sub-tile block; wave w issues piece w, wave 0 also piece 4 --------------------------------------------
    const unsigned w_vo0 = (unsigned)(lane * 16 + wave * 1024);
    const int w_base = (tile_n * nchunk_all + ck0) * 18 * WSUB;       // my first sub-tile inside the channel tile's block
    int iss_soff = w_base, iss_left = nsub, iss_slot = 0;
    auto w_issue = [&](auto cls) {
        unsigned char* dst = smem_raw + OFF_W + iss_slot * WSUB + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, (int)w_vo0, iss_soff, 0, 0);
        if constexpr (decltype(cls)::value)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + NWV * 1024), 16, (int)w_vo0, iss_soff + NWV * 1024, 0, 0);
        iss_soff += WSUB;
        if (--iss_left == 0) { iss_left = nsub; iss_soff = w_base; }      // (past the end the stream wraps to valid addresses)
        iss_slot = iss_slot + 1 == NBW ? 0 : iss_slot + 1;
    };

    // ---- fragments: my 80 pixels x all 80 channels ------------------------------------------------------------------------------------------------------
    f32x4 acc[5][5];
    bf16x8 wf[2][5], af[2][5];
    const int wfrag = OFF_W + ((wn * 80 + l15) * 32 + (kq ^ (3 * ((l15 >> 3) & 1))) * 8) * 2;      // + slot * WSUB + nb * 1024
    int afrag[5];                                            // + buf * HALO + half * 4 * PLANE + (ky * HW_ + kx) * 16
#pragma unroll
    for (int mb = 0; mb < 5; ++mb) {
        const int idx = wp * 80 + mb * 16 + l15, blk = idx / RB, r = idx - blk * RB, ty = r / TW, tx = r - ty * TW;
        afrag[mb] = kq * PLANE + (blk * HB + ty * HW_ + tx) * 16;
    }

    // ---- prologue: halo chunk 0 ------------------------------------------------------------------------------------------------------------------------------
    {
        u32x4 t[NPIECE];
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) t[j] = halo_load(j, 0);
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) halo_store(j, 0, t[j]);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    auto main_loop = [&](auto cls) {
        constexpr int NW = decltype(cls)::value ? 2 : 1;
        constexpr int PPS = pps<NPIECE>();
        u32x4 hreg[3][PPS];                                  // staged pieces in flight: requested at LOAD(i), written at LOAD(i + 3)
        // W sub-tiles 0 .. DW - 1 in flight; sub-tile 0 retired, published, its fragments (and chunk 0's halo) read into set 0
        w_issue(cls); w_issue(cls); w_issue(cls);
        if constexpr (DW == 4) w_issue(cls);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DW - 1) * NW) : "memory");
        __builtin_amdgcn_s_barrier();
        int rd_slot = 0;
        auto read_frags = [&](int set, int abase, int aimm) {
            const unsigned char* Wp = smem_raw + wfrag + rd_slot * WSUB;
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) wf[set][nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wp + nb * 1024));
#pragma unroll
            for (int mb = 0; mb < 5; ++mb)
                af[set][mb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem_raw + abase + afrag[mb] + aimm));
            rd_slot = rd_slot + 1 == NBW ? 0 : rd_slot + 1;
        };
        read_frags(0, 0, 0);
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

        int cbuf = 0;
        for (int c = 0; c < nchunk; ++c) {
            const int abase = cbuf * HALO, nbuf = cbuf ^ 1;
            auto sub = [&](auto ic) {
                constexpr int i = decltype(ic)::value;       // sub-tile s = 18 c + i: multiplied here; sub-tile s + 1's fragments are read here
                constexpr int n1 = (i + 1) % 18, tap1 = n1 >> 1, hk1 = n1 & 1;
                constexpr int aimm1 = hk1 * 4 * PLANE + ((tap1 / 3) * HW_ + (tap1 % 3)) * 16;
                // A. counted wait: my pieces of W sub-tile s + 1 (requested at LOAD(s + 1 - DW), behind that phase's halo requests) and everything
                //    older have landed; what I issued since may stay in flight: DW - 2 W requests and the halo pieces of those phases
                {
                    constexpr int extra = nh<NPIECE>(i - 1) + (DW == 4 ? nh<NPIECE>(i - 2) : 0);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DW - 2) * NW + extra) : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                // B. publish: every wave's pieces of W sub-tile s + 1 are in LDS; every wave has finished reading sub-tile s - 1's slot
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // C. the staged halo pieces requested three sub-tiles ago (retired by wait A) go to the NEXT chunk's buffer
                if constexpr (nh<NPIECE>(i - 3) >= 1) halo_store(PPS * (i - 3), nbuf, hreg[(i - 3) % 3][0]);
                if constexpr (nh<NPIECE>(i - 3) == 2) halo_store(PPS * (i - 3) + 1, nbuf, hreg[(i - 3) % 3][PPS - 1]);
                // D. sub-tile s + 1's fragments into the other register set (the first sub-tile of the next chunk reads the buffer just filled)
                read_frags((i + 1) & 1, i == 17 ? nbuf * HALO : abase, aimm1);
                // E. requests: two halo pieces of the next chunk, then W sub-tile s + DW (into the slot sub-tile s - 1 was read from)
                if constexpr (nh<NPIECE>(i) >= 1) hreg[i % 3][0] = halo_load(PPS * i, c + 1);
                if constexpr (nh<NPIECE>(i) == 2) hreg[i % 3][PPS - 1] = halo_load(PPS * i + 1, c + 1);
                w_issue(cls);
                // F. 25 MFMAs of sub-tile s, with C - E in their issue shadow: a 16x16x32 MFMA occupies the matrix pipe for 16 cycles and the issue
                //    port for 4 -- the fragment reads, the staging store / loads and the W request (all independent of this sub-tile's operands) go out
                //    between the matrix instructions instead of in front of them (one wave per SIMD: nobody else would cover that time)
#pragma unroll
                for (int mb = 0; mb < 5; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 5; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i & 1][nb], af[i & 1][mb], acc[mb][nb], 0, 0, 0);
                if constexpr (INTERLEAVE) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {                // staging stores first (their data is oldest), one per matrix instruction
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
#pragma unroll
                    for (int k = 0; k < 10; ++k) {               // the 10 fragment reads of sub-tile s + 1
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {                // halo requests and the W request(s)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            sub(IC<0>{}); sub(IC<1>{}); sub(IC<2>{}); sub(IC<3>{}); sub(IC<4>{}); sub(IC<5>{});
            sub(IC<6>{}); sub(IC<7>{}); sub(IC<8>{}); sub(IC<9>{}); sub(IC<10>{}); sub(IC<11>{});
            sub(IC<12>{}); sub(IC<13>{}); sub(IC<14>{}); sub(IC<15>{}); sub(IC<16>{}); sub(IC<17>{});
            cbuf = nbuf;
        }
    };
    if (clsA) main_loop(std::true_type{}); else main_loop(std::false_type{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the wrap-around W requests of the tail have landed: LDS is free for the epilogue
    __syncthreads();

    // ---- epilogue: bias / temb in registers, residual through the staging tile, whole-row 16-byte stores ------------------------------------------------
    // my outputs: acc[mb][nb][j] = (tile pixel 80 wave + 16 mb + l15, tile channel 16 nb + 4 kq + j)
    if (P.splits > 1) {                                      // raw fp32 partial sums of my chunks: the finishing pass owns the epilogue
        const int64_t mtot = (int64_t)P.n_img * P.H * P.W;
#pragma unroll
        for (int mb = 0; mb < 5; ++mb) {
            const int r = wp * 80 + mb * 16 + l15;
            const int blk = r / RB, q = r - blk * RB, ty = q / TW, tx = q - ty * TW;
            const int rb = rb0 + blk, img = rb / tpi, rin = rb - img * tpi, yb = rin / P.tiles_x, xb = rin - yb * P.tiles_x, y = yb * TH + ty;
            if (rb < rb_total && y < P.H) {
                float* dst = P.ws + ((int64_t)split * mtot + ((int64_t)img * P.H + y) * P.W + xb * TW + tx) * P.cout + n0 + wn * 80 + 4 * kq;
#pragma unroll
                for (int nb = 0; nb < 5; ++nb) *reinterpret_cast<f32x4*>(dst + nb * 16) = acc[mb][nb];
            }
        }
        return;
    }
    constexpr int OP = BN + 8;                               // bf16 pitch of the staging rows (176 / 336 B)
    bf16_t* Os = reinterpret_cast<bf16_t*>(smem_raw);        // [320][OP] = 56,320 B
    constexpr int CPR = BN / 8;                              // sixteen-byte chunks per row
    auto row_pixel = [&](int r, bool& ok) -> int64_t {       // tile row -> global pixel (rows past the image's last row / past the last image: not stored)
        const int blk = r / RB, q = r - blk * RB, ty = q / TW, tx = q - ty * TW;
        const int rb = rb0 + blk, img = rb / tpi, rin = rb - img * tpi, yb = rin / P.tiles_x, xb = rin - yb * P.tiles_x, y = yb * TH + ty;
        ok = rb < rb_total && y < P.H;
        return ((int64_t)img * P.H + y) * P.W + xb * TW + tx;
    };
    {   // bias and time-embedding words of my outputs: every load of a kind issued before the first is used (one `if (P.temb)` per accumulator block put
        // a load and its own s_waitcnt vmcnt(0) in each of 25 branches: 25 dependent round trips in the epilogue of every first conv of a ResNet block)
        u32x2 bt[5], tt[5][5];
        if (P.bias) {
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) bt[nb] = *reinterpret_cast<const u32x2*>(P.bias + n0 + wn * 80 + nb * 16 + 4 * kq);
        }
        if (P.temb) {                                        // (a tile spans several images: the row's own image)
#pragma unroll
            for (int mb = 0; mb < 5; ++mb) {
                const int idx = wp * 80 + mb * 16 + l15, rb = min(rb0 + idx / RB, rb_total - 1), img = rb / tpi;
                const bf16_t* trow = P.temb + (int64_t)(img / P.temb_div) * P.temb_ld + n0 + wn * 80 + 4 * kq;
#pragma unroll
                for (int nb = 0; nb < 5; ++nb) tt[mb][nb] = *reinterpret_cast<const u32x2*>(trow + nb * 16);
            }
        }
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) {
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (P.bias) {
                b4[0] = __uint_as_float(bt[nb][0] << 16); b4[1] = __uint_as_float(bt[nb][0] & 0xffff0000u);
                b4[2] = __uint_as_float(bt[nb][1] << 16); b4[3] = __uint_as_float(bt[nb][1] & 0xffff0000u);
            }
#pragma unroll
            for (int mb = 0; mb < 5; ++mb) {
                float t4[4] = {0.f, 0.f, 0.f, 0.f};
                if (P.temb) {
                    t4[0] = __uint_as_float(tt[mb][nb][0] << 16); t4[1] = __uint_as_float(tt[mb][nb][0] & 0xffff0000u);
                    t4[2] = __uint_as_float(tt[mb][nb][1] << 16); t4[3] = __uint_as_float(tt[mb][nb][1] & 0xffff0000u);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mb][nb][j] += b4[j] + t4[j];
            }
        }
    }
    if (P.res) {
        // the residual rows in bursts of RBU loads per thread (rows that are not stored read pixel 0: never used) -- rolled, the loop was
        // load -> s_waitcnt vmcnt(0) -> ds_write per iteration: BM * CPR / NT (12.5 at 4 waves) dependent round trips
        constexpr int NIT = (BM * CPR + NT - 1) / NT, RBU = 7;
#pragma unroll
        for (int h = 0; h < (NIT + RBU - 1) / RBU; ++h) {
            u32x4 rv[RBU];
#pragma unroll
            for (int it = 0; it < RBU; ++it) {
                const int cidx = min(tid + (h * RBU + it) * NT, BM * CPR - 1), r = cidx / CPR, ch = cidx - r * CPR;
                bool ok;
                const int64_t m = row_pixel(r, ok);
                rv[it] = *reinterpret_cast<const u32x4*>(P.res + (ok ? m : 0) * P.cout + n0 + ch * 8);
            }
#pragma unroll
            for (int it = 0; it < RBU; ++it) {
                const int cidx = tid + (h * RBU + it) * NT, r = cidx / CPR, ch = cidx - r * CPR;
                if (cidx < BM * CPR) *reinterpret_cast<u32x4*>(Os + r * OP + ch * 8) = rv[it];
            }
        }
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < 5; ++mb)
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                const u32x2 t = *reinterpret_cast<const u32x2*>(Os + (wp * 80 + mb * 16 + l15) * OP + wn * 80 + nb * 16 + 4 * kq);
                acc[mb][nb][0] += __uint_as_float(t[0] << 16); acc[mb][nb][1] += __uint_as_float(t[0] & 0xffff0000u);
                acc[mb][nb][2] += __uint_as_float(t[1] << 16); acc[mb][nb][3] += __uint_as_float(t[1] & 0xffff0000u);
            }
        __syncthreads();
    }
#pragma unroll
    for (int mb = 0; mb < 5; ++mb)
#pragma unroll
        for (int nb = 0; nb < 5; ++nb)
            *reinterpret_cast<u32x2*>(Os + (wp * 80 + mb * 16 + l15) * OP + wn * 80 + nb * 16 + 4 * kq) =
                u32x2{pack_bf2(acc[mb][nb][0], acc[mb][nb][1]), pack_bf2(acc[mb][nb][2], acc[mb][nb][3])};
    __syncthreads();
    for (int cidx = tid; cidx < BM * CPR; cidx += NT) {
        const int r = cidx / CPR, ch = cidx - r * CPR;
        bool ok;
        const int64_t m = row_pixel(r, ok);
        if (ok) *reinterpret_cast<u32x4*>(P.out + m * P.cout + n0 + ch * 8) = *reinterpret_cast<const u32x4*>(Os + r * OP + ch * 8);
    }
    // GroupNorm statistics of the ROUNDED outputs, one (sum, sum of squares) pair per (image, row block, group): thread t sums group t % GT over
    // rows t / GT, t / GT + RP, ... of one row block at a time; fixed summation order, no atomics
    if (P.gn_part) {
        const int cpg = P.cout >> 5, GT = BN / cpg, RP = NT / GT, gl = tid % GT;
        float* red = reinterpret_cast<float*>(smem_raw + (size_t)BM * OP * 2);
        for (int blk = 0; blk < NB; ++blk) {
            const int rb = rb0 + blk;
            if (rb >= rb_total) break;                       // (uniform)
            const int img = rb / tpi, sp = rb - img * tpi;
            const int vr = min(TH, P.H - (sp / P.tiles_x) * TH) * TW;       // rows of this block inside the image
            float gs = 0.f, gss = 0.f;
            for (int r = tid / GT; r < vr; r += RP) {
                const unsigned* wsrc = reinterpret_cast<const unsigned*>(Os + (blk * RB + r) * OP + gl * cpg);
                for (int k = 0; k < cpg / 2; ++k) {
                    const unsigned u = wsrc[k];
                    const float a = __uint_as_float(u << 16), b = __uint_as_float(u & 0xffff0000u);
                    gs += a + b;
                    gss += a * a + b * b;
                }
            }
            red[2 * tid] = gs;
            red[2 * tid + 1] = gss;
            __syncthreads();
            if (tid < GT) {
                float a = 0.f, b = 0.f;
                for (int k = 0; k < RP; ++k) { a += red[2 * (tid + k * GT)]; b += red[2 * (tid + k * GT) + 1]; }
                float* dst = P.gn_part + (((int64_t)img * tpi + sp) * 32 + (n0 / cpg + tid)) * 2;
                dst[0] = a;
                dst[1] = b;
            }
            __syncthreads();
        }
    }
}

// filter [Cout][3][3][Cin] -> [Cout / 80][Cin / 64][9 taps][2 halves][80 rows][32], 16-byte chunks in their LDS places (chunk p of row r holds
// logical chunk p ^ (3 * ((r >> 3) & 1)))
__global__ __launch_bounds__(256) void conv_halo4_pack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ dst, int cout, int cin, int BN) {
    const int64_t total = (int64_t)cout * 9 * cin / 8;
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

// second pass of a split launch: out = sum over the splits (fixed order) + bias + temb + residual, 8 channels of one pixel per thread
__global__ __launch_bounds__(256) void conv_halo4_finish_kernel(const C4Params P) {
    const int64_t mtot = (int64_t)P.n_img * P.H * P.W, cpr = P.cout / 8, total = mtot * cpr;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = id / cpr;
        const int n = (int)(id - m * cpr) * 8;
        // every operand of the chunk requested before the first is used: bias / temb / residual words, then the partial sums four splits at a time
        // (one load -> wait -> add per split and per side operand was seven dependent round trips for a launch that moves 10 MB)
        u32x4 tb = u32x4{0u, 0u, 0u, 0u}, tt = tb, tr = tb;
        if (P.bias) tb = *reinterpret_cast<const u32x4*>(P.bias + n);
        if (P.temb) tt = *reinterpret_cast<const u32x4*>(P.temb + ((m / ((int64_t)P.H * P.W)) / P.temb_div) * P.temb_ld + n);
        if (P.res) tr = *reinterpret_cast<const u32x4*>(P.res + m * P.cout + n);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int sp0 = 0; sp0 < P.splits; sp0 += 4) {
            f32x4 pa[4], pb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* src = P.ws + ((int64_t)min(sp0 + k, P.splits - 1) * mtot + m) * P.cout + n;
                pa[k] = *reinterpret_cast<const f32x4*>(src);
                pb[k] = *reinterpret_cast<const f32x4*>(src + 4);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)                       // (fixed order: split 0, 1, 2, ...)
                if (sp0 + k < P.splits) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] += pa[k][j]; v[4 + j] += pb[k][j]; }
                }
        }
        const u32x4* side[3] = {&tb, &tt, &tr};
#pragma unroll
        for (int q = 0; q < 3; ++q)                           // + bias, + temb, + residual, in the order of the one-pass epilogue (absent ones add zero words)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] += __uint_as_float((*side[q])[j] << 16);
                v[2 * j + 1] += __uint_as_float((*side[q])[j] & 0xffff0000u);
            }
        Vec8<bf16_t>::store(P.out + m * P.cout + n, v);
    }
}

template <int TW, int NWV> int launch_c4(C4Params& P, hipStream_t st) {
    using G = Geo<TW, NWV>;
    P.tiles_y = (P.H + G::TH - 1) / G::TH;
    P.tiles_x = P.W / TW;
    P.tiles_p = (P.n_img * P.tiles_y * P.tiles_x + G::NB - 1) / G::NB;
    {   // XCD grid (see the kernel): FMC_C4_XCD = 0 contiguous ranges (rounds 5's order), 1 / 2 / 4 / 8 forced where it divides, unset = least fabric traffic
        static const int forced = getenv("FMC_C4_XCD") ? atoi(getenv("FMC_C4_XCD")) : -1;
        const int tnv = P.tiles_n * P.splits;
        double best = 0.;
        P.xcd_pc = 0;
        if (forced != 0 && (P.tiles_p * tnv) % 8 == 0) {
            for (int pc = 1; pc <= 8; pc *= 2) {
                if (tnv % pc || P.tiles_p % (8 / pc) || (forced > 0 && pc != forced)) continue;
                const double cost = (double)P.w_bytes * (8 / pc) + (double)(P.x_bytes + P.x2_bytes) * pc;
                if (!P.xcd_pc || cost < best) { best = cost; P.xcd_pc = pc; }
            }
        }
    }
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo4_kernel<TW, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        raised = true;
    }
    hipLaunchKernelGGL((conv_halo4_kernel<TW, NWV>), dim3((unsigned)(P.tiles_p * P.tiles_n * P.splits)), dim3(G::NT), G::LDS_BYTES, st, P);
    if (P.splits > 1) {
        const int64_t chunks = (int64_t)P.n_img * P.H * P.W * (P.cout / 8);
        const unsigned grid = (unsigned)((chunks + 255) / 256 < 2048 ? (chunks + 255) / 256 : 2048);
        hipLaunchKernelGGL(conv_halo4_finish_kernel, dim3(grid), dim3(256), 0, st, P);
    }
    return 0;
}

}  // namespace

// `wide` = 0: 4 waves, 80 output channels per tile; 1: 8 waves, 160 (16-pixel-wide row blocks only).  The packed filter differs between the two.
extern "C" int fmc_conv3x3_halo4_pack_weight(const void* w, void* dst, int Cin, int Cout, int wide, void* stream) {
    const int BN = wide ? 160 : 80;
    if (!w || !dst) FMC_FAIL(FMC_E_NULL, "conv3x3_halo4_pack_weight: NULL pointer");
    if (Cin % 64 || Cout % BN) FMC_FAIL(FMC_E_SHAPE, "conv3x3_halo4_pack_weight: Cin %% 64 / Cout %% %d (Cin=%d Cout=%d)", BN, Cin, Cout);
    if (!fmc_aligned16(w) || !fmc_aligned16(dst)) FMC_FAIL(FMC_E_ALIGN, "conv3x3_halo4_pack_weight: pointers must be 16-byte aligned");
    const int64_t chunks = (int64_t)Cout * 9 * Cin / 8;
    const int grid = (int)((chunks + 255) / 256 < 4096 ? (chunks + 255) / 256 : 4096);
    hipLaunchKernelGGL(conv_halo4_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, (bf16_t*)dst, Cout, Cin, BN);
    FMC_CHECK_LAUNCH("fmc_conv3x3_halo4_pack_weight");
    return 0;
}

// row blocks are 10 x 16 pixels where W % 16 == 0, else 5 x 8 (W % 8 == 0); any H
static int c4_tw(int W) { return W % 16 == 0 ? 16 : 8; }
extern "C" int fmc_conv3x3_halo4_supported(int n_img, int H, int W, int Cin, int Cin1, int Cout, int upsample2x, int wide) {
    if (n_img < 1 || H < 1 || W < 8 || W % 8 || Cin % 64 || Cout % (wide ? 160 : 80) || (wide && W % 16)) return 0;
    if (Cin1 <= 0 || Cin1 > Cin || Cin1 % 64) return 0;
    if (upsample2x && ((H | W) & 1)) return 0;
    const int64_t hs = upsample2x ? H / 2 : H, ws = upsample2x ? W / 2 : W;
    if ((int64_t)n_img * hs * ws * Cin1 * 2 >= (1ll << 31) || (int64_t)n_img * hs * ws * (Cin - Cin1) * 2 >= (1ll << 31)) return 0;
    if ((int64_t)Cout * 9 * Cin * 2 >= (1ll << 31)) return 0;
    return 1;
}

extern "C" int fmc_conv3x3_halo4_row_blocks_per_image(int H, int W) {
    const int tw = c4_tw(W);
    return (tw == 8 ? (H + 4) / 5 : (H + 9) / 10) * (W / tw);
}

extern "C" int fmc_conv3x3_halo4_tiles(int n_img, int H, int W, int Cout, int wide) {
    const int nb = c4_tw(W) == 8 ? 8 : 2;
    return (n_img * fmc_conv3x3_halo4_row_blocks_per_image(H, W) + nb - 1) / nb * (Cout / (wide ? 160 : 80));
}

/* As fmc_conv3x3_halo_bf16 (fmc_hip.h) without the GroupNorm operand path, for images 8 / 16 / 32 pixels wide; gn_partials
 * [n_img, fmc_conv3x3_halo4_row_blocks_per_image(H, W), 32, 2]; w_packed = fmc_conv3x3_halo4_pack_weight.  split_k > 1: the 64-channel chunks of the
 * reduction are dealt to split_k workgroups per tile (5x8-pixel images: 64 tiles on 256 CUs otherwise), fp32 partials in `workspace`
 * (>= split_k * n_img * H * W * Cout * 4 bytes), summed in a fixed order by a second launch that runs the epilogue; no statistics epilogue then. */
extern "C" int fmc_conv3x3_halo4_bf16(const void* x, const void* x2, int Cin1, const void* w_packed, const void* bias, const void* temb,
                                      const void* residual, void* out, int n_img, int H, int W, int Cin, int Cout, int64_t temb_row_stride,
                                      int temb_img_div, int upsample2x, float* gn_partials, int split_k, void* workspace, int64_t workspace_bytes,
                                      int wide, void* stream) {
    const int BN = wide ? 160 : 80;
    if (!x || !w_packed || !out) FMC_FAIL(FMC_E_NULL, "conv3x3_halo4: NULL x / w / out");
    if (!x2) Cin1 = Cin;
    if (!fmc_conv3x3_halo4_supported(n_img, H, W, Cin, Cin1, Cout, upsample2x, wide))
        FMC_FAIL(FMC_E_SHAPE, "conv3x3_halo4: needs W %% 8 == 0 (wide: %% 16), Cin %% 64 == 0 (both sources), Cout %% %d == 0, operands < 2 GiB "
                 "(n=%d H=%d W=%d Cin=%d+%d Cout=%d ups=%d)", BN, n_img, H, W, Cin1, Cin - Cin1, Cout, upsample2x);
    if (!fmc_aligned16(x) || !fmc_aligned16(w_packed) || !fmc_aligned16(out) || (x2 && !fmc_aligned16(x2)) || (residual && !fmc_aligned16(residual)) ||
        (bias && (reinterpret_cast<uintptr_t>(bias) & 7)) || (temb && ((reinterpret_cast<uintptr_t>(temb) & 7) || temb_row_stride % 4)))
        FMC_FAIL(FMC_E_ALIGN, "conv3x3_halo4: x / w / out / residual must be 16-byte aligned, bias / temb rows 8-byte aligned");
    if (temb && temb_img_div < 1) FMC_FAIL(FMC_E_SHAPE, "conv3x3_halo4: temb_img_div %d", temb_img_div);
    if (gn_partials && (Cout % 64 || BN % (Cout / 32)))
        FMC_FAIL(FMC_E_SHAPE, "conv3x3_halo4: the statistics epilogue needs Cout %% 64 == 0 and %d %% (Cout / 32) == 0 (Cout=%d)", BN, Cout);
    if (split_k < 1) split_k = 1;
    if (split_k > Cin / 64) split_k = Cin / 64;
    if (split_k > 1) {
        if (gn_partials) FMC_FAIL(FMC_E_SHAPE, "conv3x3_halo4: the statistics epilogue is not available with split_k > 1");
        if (!workspace || workspace_bytes < (int64_t)split_k * n_img * H * W * Cout * 4 || !fmc_aligned16(workspace))
            FMC_FAIL(FMC_E_SHAPE, "conv3x3_halo4: split_k %d needs a 16-byte aligned workspace of %lld bytes", split_k, (long long)split_k * n_img * H * W * Cout * 4);
    }
    C4Params P;
    P.x = (const bf16_t*)x; P.x2 = (const bf16_t*)x2; P.c1 = Cin1;
    P.w = (const bf16_t*)w_packed; P.bias = (const bf16_t*)bias; P.temb = (const bf16_t*)temb; P.res = (const bf16_t*)residual; P.out = (bf16_t*)out;
    P.n_img = n_img; P.H = H; P.W = W; P.cin = Cin; P.cout = Cout; P.ups = upsample2x ? 1 : 0;
    P.temb_ld = temb_row_stride; P.temb_div = temb ? temb_img_div : 1;
    P.gn_part = gn_partials; P.tiles_n = Cout / BN;
    P.splits = split_k; P.ws = (float*)workspace;
    const int64_t hs = upsample2x ? H / 2 : H, ws = upsample2x ? W / 2 : W;
    P.x_bytes = (int64_t)n_img * hs * ws * Cin1 * 2; P.x2_bytes = (int64_t)n_img * hs * ws * (Cin - Cin1) * 2;
    P.w_bytes = (int64_t)Cout * 9 * Cin * 2;
    hipStream_t st = (hipStream_t)stream;
    if (c4_tw(W) == 8) launch_c4<8, 4>(P, st);
    else if (wide) launch_c4<16, 8>(P, st);
    else launch_c4<16, 4>(P, st);
    FMC_CHECK_LAUNCH("fmc_conv3x3_halo4_bf16");
    return 0;
}
