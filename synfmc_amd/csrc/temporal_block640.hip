// The fused temporal attention block of the motion modules at the 20x32 level (C = 640, 8 heads x 80, F = 16 frames), gfx950.
//
// Same chain as temporal_block.hip (fmc/models/motion_module.py:287-300, fmc/models/attention_processor.py:255-291):
//     n = LayerNorm_i(h) + pe;  m = s qkv_merge(n + pose) + n (block 0);  q | k | v = to_q / to_k / to_v (m);  o = softmax(q k^T d^-1/2) v over the frames;
//     h' = to_out(o) + b + h
// which at this level ran as a LayerNorm launch + three GEMM launches + the temporal attention launch (~160 us per block, 10 blocks per step).
//
// C = 640 changes the geometry of temporal_block.hip, not the idea.  A tile is 5 pixels x 16 frames = 80 rows (M = 20480 = 256 tiles: one per CU); X, the
// resident 80 x 640 bf16 tile, takes 100 KiB, so a 32-deep sub-tile of a 640-row weight (40 KiB) cannot be double-buffered in what is left.  The weights
// therefore never touch LDS: the 8 waves sit side by side along N (a wave = 80 rows x 80 columns = 5 x 5 v_mfma_f32_16x16x32_bf16 accumulators, as in
// every 160x320 kernel of this library), so a wave's W rows are its own, and it reads them straight into registers from a copy PRE-PACKED IN FRAGMENT
// ORDER (one contiguous KiB per 16-row block and k-step, `hip_ops.pack_w_frag80`), three k-steps ahead -- the scheme of phase D of temporal_block.hip,
// here for all four projections.  No ring, no counted vmcnt and no barrier inside a main loop: X is read-only while a projection runs.
//   A  h rows -> X by LDS-DMA (row = 16 pixel + frame; 16-byte chunks XOR-swizzled through the source address: conflict-free ds_read_b128),
//      LayerNorm + (beta + pe[frame]) in place.
//   B  (block 0) m = s x W_m^T + pose_term + x, in place; the pose-term rows arrive in three 16-row pass buffers (60 KiB) under the main loop.
//   D  wave = head: q | k | v of the head for the 80 rows in registers (swapped product for q, k; un-swapped for v), scores = two 16x16x32 + one
//      16x16x16 MFMA per pixel (d = 80 = 32 + 32 + 16: no padding), softmax across 4 lanes, PV = five 16x16x16; o parked in registers, then written over m.
//   E  h' = o W_out^T + b + h: the residual rows come back through the pass buffers, h' is formed in place there and leaves with whole-row stores.
// Roofline: MFMA.  Algorithmic flops = 2 M C (C [merge] + 3 C + C) + 4 M F C = 132 GF (block 0) / 106 GF (block 1) at M = 20480; HBM-side bytes 79 / 52 MB.
#include <type_traits>

#include "common.h"

namespace {

constexpr int T6_C = 640, T6_ROWS = 80, T6_PIX = 5, T6_F = 16, T6_CPR = 80;   // CPR: 16-byte chunks per row
constexpr int T6_X_ELEMS = T6_ROWS * T6_C;                 // 51200 bf16 = 100 KiB
constexpr int T6_PASS = 16 * T6_C;                         // one pass buffer: 16 rows (20 KiB)
constexpr int T6_LDS = (T6_X_ELEMS + 3 * T6_PASS) * 2;     // 163840 B
constexpr int T6_WAVE_W = 20 * 5 * 512;                    // bf16 elements of one wave's packed 80 x 640 weight slice (100 KiB)

struct T6Params {
    const bf16_t* h; bf16_t* out;                          // [clips, 16, hw, 640] channels-last video tokens
    const float* ln_gamma;                                 // [640]
    const float* ln_bpe;                                   // [16][640]: LayerNorm beta + positional-encoding row of frame f
    float ln_eps;
    const bf16_t* w_merge;                                 // fragment order [8 waves][20 k-steps][5 blocks][lane][8], or NULL
    const bf16_t* pose_term;                               // s (W_m pose + b_m), layout of h (read when w_merge)
    float merge_scale;
    const bf16_t* w_qkv;                                   // [8 heads][q | k | v][20 k-steps][5 blocks][lane][8]
    const bf16_t* w_out;                                   // fragment order as w_merge
    const bf16_t* b_out;                                   // [640] or NULL
    int n_clips, hw, tpc;
    float scale_log2;
    // XATT (the text cross-attention block of the spatial transformer at this level, same skeleton): tokens [images][hw][640], a tile = 80 consecutive rows
    const bf16_t* kvfrag;                                  // [batch][8 heads][K: 5 key blocks x (512 + 512 + 256) | V^T: 5 x 5 x 256] (fmc_xattn_pack_kv)
    int n_keys, frames_per_batch;                          // text tokens (<= 80); images per text row (kv_batch_div)
    int64_t total_rows;
};

__device__ __forceinline__ void t6_dma(const __amdgpu_buffer_rsrc_t& rs, unsigned voff, void* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, 0, 0, 0);
}
__device__ __forceinline__ void t6_unpack4(const u32x2& w, float (&o)[4]) {
    o[0] = __uint_as_float(w[0] << 16); o[1] = __uint_as_float(w[0] & 0xffff0000u);
    o[2] = __uint_as_float(w[1] << 16); o[3] = __uint_as_float(w[1] & 0xffff0000u);
}
// (see TB_SETTLE in temporal_block.hip: MFMA results are handed to VALU code only behind a scheduling fence + idle issue cycles)
#define T6_SETTLE()                                                                                                      \
    do {                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    } while (0)
#define T6_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

template <bool HAS_MERGE, bool XATT>
__global__ __launch_bounds__(512, 2)
void temporal_block640_kernel(const T6Params P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* X = reinterpret_cast<bf16_t*>(smem_raw);             // [80][640], chunk c of row r at chunk c ^ ((r >> 1) & 7)
    bf16_t* PB = X + T6_X_ELEMS;                                 // 3 pass buffers [16][640], X's swizzle
    float* stats = reinterpret_cast<float*>(PB + 2 * T6_PASS);   // phase A only: (mean, rstd) x 80 rows, in pass buffer 2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int xsw = (l15 >> 1) & 7;                              // chunk swizzle of my fragment rows (row = 16 mb + l15)
    const int ecol = wave * 80 + 4 * kq;                         // my accumulator columns in phases B, E: ecol + 16 nb + j

    const int tile = blockIdx.x;
    const int clip = tile / P.tpc, p0 = (tile - clip * P.tpc) * T6_PIX;
    // tile row r = 16 hi + lo lives at element row0 + lo * fstride + hi * pstride: temporal = (frame lo, pixel hi) of the clip; XATT = row 80 tile + r
    const unsigned row0 = XATT ? (unsigned)((int64_t)tile * T6_ROWS * T6_C) : (unsigned)(((int64_t)clip * T6_F * P.hw + p0) * T6_C);
    const unsigned fstride = XATT ? (unsigned)T6_C : (unsigned)(P.hw * T6_C);
    const unsigned pstride = XATT ? (unsigned)(16 * T6_C) : (unsigned)T6_C;
    const int64_t total_elems = P.total_rows * T6_C;
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)P.h, 0, (int)(total_elems * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsPT = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_MERGE ? P.pose_term : P.h), 0, (int)(total_elems * 2), 0x00020000);

    // rows [r0, r0 + nr) of the tile of `rs` -> LDS at `dst`, row-major with X's chunk swizzle; nr * 80 chunks = pieces of 64, piece q = wave + 8 j
    auto issue_rows = [&](const __amdgpu_buffer_rsrc_t& rs, int r0, int nr, bf16_t* dst) {
        const int pieces = nr * T6_CPR / 64;
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            const int q = wave + 8 * j;
            if (q < pieces) {
                const int idx = 64 * q + lane, rl = idx / T6_CPR, pc = idx - rl * T6_CPR, r = r0 + rl, c = pc ^ ((r >> 1) & 7);
                const unsigned src = (row0 + (unsigned)(r & 15) * fstride + (unsigned)(r >> 4) * pstride + (unsigned)c * 8) * 2;
                t6_dma(rs, src, dst + 64 * q * 8);
            }
        }
    };
    auto issue_pass = [&](const __amdgpu_buffer_rsrc_t& rs, int pss) { issue_rows(rs, 16 * pss, 16, PB + (pss % 3) * T6_PASS); };

    // ================= phase A: h -> X, LayerNorm (+ pe) in place =================
    issue_rows(rsH, 0, T6_ROWS, X);
    T6_VMCNT0();
    __syncthreads();
    {
        // statistics: 8 lanes per row, 10 chunks each, centred variance
#pragma unroll 1
        for (int r = tid >> 3; r < T6_ROWS; r += 64) {
            const int q = tid & 7;
            const bf16_t* xr = X + r * T6_C;
            u32x4 x4[10];
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                x4[i] = *reinterpret_cast<const u32x4*>(xr + (q + 8 * i) * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) s1 += __uint_as_float(x4[i][j] << 16) + __uint_as_float(x4[i][j] & 0xffff0000u);
            }
            s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4);
            const float mean = s1 * (1.f / 640.f);
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 10; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = __uint_as_float(x4[i][j] << 16) - mean, b = __uint_as_float(x4[i][j] & 0xffff0000u) - mean;
                    s2 += a * a + b * b;
                }
            s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4);
            if (q == 0) *reinterpret_cast<f32x2_t*>(stats + 2 * r) = f32x2_t{mean, rsqrtf(s2 * (1.f / 640.f) + P.ln_eps)};
        }
    }
    __syncthreads();
    if (tid < 480) {
        // normalising thread = (logical chunk nc of 80, row group nrg of 6): rows nrg, nrg + 6, ...; (beta + pe) of the row's frame from the L1-resident table
        const int nc = tid % 80, nrg = tid / 80;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(P.ln_gamma + nc * 8), g1 = *reinterpret_cast<const f32x4*>(P.ln_gamma + nc * 8 + 4);
#pragma unroll 2
        for (int r = nrg; r < T6_ROWS; r += 6) {
            const float* bp = P.ln_bpe + (r & 15) * T6_C + nc * 8;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
            u32x4* px = reinterpret_cast<u32x4*>(X + r * T6_C + (nc ^ ((r >> 1) & 7)) * 8);
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
    __syncthreads();                                              // X = x = LayerNorm(h) + pe; the pass buffers are free

    // ---- one 80 x 640 x 640 projection with the A operand resident in X and MY 80 weight rows streamed from `wbase` (fragment order) -----------------
    f32x4 acc[5][5];
    auto project = [&](const bf16_t* wbase) {
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(wbase + (size_t)wave * T6_WAVE_W), 0, T6_WAVE_W * 2, 0x00020000);
        int wl = lane * 16;
        asm volatile("" : "+v"(wl));
        u32x4 wfr[3][5];
        auto load_w = [&](int g) {
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) wfr[g % 3][nb] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wl, (g * 5 + nb) * 1024, 0);
        };
        load_w(0);
        load_w(1);
        auto step = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + 2 < 20) load_w(g + 2);
            int kqx = kq ^ xsw, xrow_o = l15 * T6_C;
            asm volatile("" : "+v"(kqx), "+v"(xrow_o));
            const int xo = xrow_o + ((g >> 1) * 8 + (((g & 1) * 4) ^ kqx)) * 8;
            __builtin_amdgcn_sched_barrier(0);
            u32x4 af3[3];
            af3[0] = *reinterpret_cast<const u32x4*>(X + 0 * 16 * T6_C + xo);
            af3[1] = *reinterpret_cast<const u32x4*>(X + 1 * 16 * T6_C + xo);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int mb = 0; mb < 5; ++mb) {
                if (mb + 2 < 5) af3[(mb + 2) % 3] = *reinterpret_cast<const u32x4*>(X + (mb + 2) * 16 * T6_C + xo);
                union { bf16x8 v; u32x4 u; } a;
                a.u = af3[mb % 3];
#pragma unroll
                for (int nb = 0; nb < 5; ++nb) {
                    union { bf16x8 v; u32x4 u; } w;
                    w.u = wfr[g % 3][nb];
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, a.v, acc[mb][nb], 0, 0, 0);
                }
                if (mb + 2 < 5) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
            }
        };
#define T6_G(G) step(std::integral_constant<int, G>{})
        T6_G(0); T6_G(1); T6_G(2); T6_G(3); T6_G(4); T6_G(5); T6_G(6); T6_G(7); T6_G(8); T6_G(9);
        T6_G(10); T6_G(11); T6_G(12); T6_G(13); T6_G(14); T6_G(15); T6_G(16); T6_G(17); T6_G(18); T6_G(19);
#undef T6_G
        T6_SETTLE();
    };

    // ================= phase B: m = s x W_m^T + pose_term + x (in place) =================
    if (HAS_MERGE) {
        issue_pass(rsPT, 0);
        issue_pass(rsPT, 1);
        issue_pass(rsPT, 2);
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        project(P.w_merge);
        T6_VMCNT0();
        __syncthreads();                                          // every wave is done reading x; passes 0-2 have landed
#pragma unroll
        for (int pss = 0; pss < 5; ++pss) {
            if (pss >= 3) {
                T6_VMCNT0();
                __syncthreads();
            }
            {
                const bf16_t* Pt = PB + (pss % 3) * T6_PASS + l15 * T6_C;
                bf16_t* Xr = X + (pss * 16 + l15) * T6_C;
#pragma unroll
                for (int nb = 0; nb < 5; ++nb) {
                    const int col = ecol + nb * 16, off = (((col >> 3) ^ xsw) << 3) + (col & 7);
                    u32x2* px = reinterpret_cast<u32x2*>(Xr + off);
                    float xv[4], pv[4];
                    t6_unpack4(*px, xv);
                    t6_unpack4(*reinterpret_cast<const u32x2*>(Pt + off), pv);
                    *px = u32x2{pack_bf2(acc[pss][nb][0] * P.merge_scale + pv[0] + xv[0], acc[pss][nb][1] * P.merge_scale + pv[1] + xv[1]),
                                pack_bf2(acc[pss][nb][2] * P.merge_scale + pv[2] + xv[2], acc[pss][nb][3] * P.merge_scale + pv[3] + xv[3])};
                }
            }
            if (pss + 3 < 5) {
                __syncthreads();                                  // everybody has read pass pss: its buffer takes pass pss + 3
                issue_pass(rsPT, pss + 3);
            }
        }
        __syncthreads();                                          // X = m
    }

    // ================= phase D: wave = head.  q | k | v projections + attention, all in registers =================
    u32x2 o_pk[5][5];                                             // o of my head: (row 16 m + l15, channels 16 b + 4 kq ..) as 4 bf16
    {
        constexpr int NPARTS = XATT ? 1 : 3;                      // XATT: only to_q (k | v come from the text, packed per step by fmc_xattn_pack_kv)
        const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(P.w_qkv + (size_t)wave * (NPARTS * T6_WAVE_W)), 0, NPARTS * T6_WAVE_W * 2, 0x00020000);
        int qkv_lane = lane * 16;
        asm volatile("" : "+v"(qkv_lane));
        u32x4 wq[3][5];                                           // [stage][block]
        auto load_step = [&](int s) {
#pragma unroll
            for (int b = 0; b < 5; ++b) wq[s % 3][b] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, qkv_lane, (s * 5 + b) * 1024, 0);
        };
        load_step(0);
        load_step(1);
        f32x4 pacc[5][5];
        bf16x8 q8a[5], q8b[5];
        s16x4 q4[5];
        u32x2 p_pk[5];
        auto zero_pacc = [&]() {
#pragma unroll
            for (int m = 0; m < 5; ++m)
#pragma unroll
                for (int b = 0; b < 5; ++b) pacc[m][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        auto step = [&](auto part_c, auto ks_c) {
            constexpr int part = decltype(part_c)::value, ks = decltype(ks_c)::value, s = part * 20 + ks;
            if constexpr (s + 2 < 20 * NPARTS) load_step(s + 2);
            int kqx = kq ^ xsw, xrow_o = l15 * T6_C;
            asm volatile("" : "+v"(kqx), "+v"(xrow_o));
            const int xo = xrow_o + ((ks >> 1) * 8 + (((ks & 1) * 4) ^ kqx)) * 8;
            __builtin_amdgcn_sched_barrier(0);
            u32x4 af3[3];
            af3[0] = *reinterpret_cast<const u32x4*>(X + 0 * 16 * T6_C + xo);
            af3[1] = *reinterpret_cast<const u32x4*>(X + 1 * 16 * T6_C + xo);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                if (m + 2 < 5) af3[(m + 2) % 3] = *reinterpret_cast<const u32x4*>(X + (m + 2) * 16 * T6_C + xo);
                union { bf16x8 v; u32x4 u; } a;
                a.u = af3[m % 3];
#pragma unroll
                for (int b = 0; b < 5; ++b) {
                    union { bf16x8 v; u32x4 u; } w;
                    w.u = wq[s % 3][b];
                    if constexpr (part < 2) pacc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, a.v, pacc[m][b], 0, 0, 0);   // (frame l15, 4 channels)
                    else pacc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, w.v, pacc[m][b], 0, 0, 0);                     // (channel l15, 4 frames)
                }
                if (m + 2 < 5) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
            }
        };
#define T6_STEP(PART, KS) step(std::integral_constant<int, PART>{}, std::integral_constant<int, KS>{})
#define T6_PART(PART)                                                                                                    \
    T6_STEP(PART, 0); T6_STEP(PART, 1); T6_STEP(PART, 2); T6_STEP(PART, 3); T6_STEP(PART, 4);                            \
    T6_STEP(PART, 5); T6_STEP(PART, 6); T6_STEP(PART, 7); T6_STEP(PART, 8); T6_STEP(PART, 9);                            \
    T6_STEP(PART, 10); T6_STEP(PART, 11); T6_STEP(PART, 12); T6_STEP(PART, 13); T6_STEP(PART, 14);                       \
    T6_STEP(PART, 15); T6_STEP(PART, 16); T6_STEP(PART, 17); T6_STEP(PART, 18); T6_STEP(PART, 19)
        auto pack8 = [](const f32x4& x, const f32x4& y) {
            union { bf16x8 v; u32x4 u; } t;
            t.u = u32x4{pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
            return t.v;
        };
        auto pack4 = [](const f32x4& x) {
            union { u32x2 u; s16x4 s; } t;
            t.u = u32x2{pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3])};
            return t.s;
        };
        // ---- q ----
        zero_pacc();
        T6_PART(0);
        T6_SETTLE();
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            q8a[m] = pack8(pacc[m][0], pacc[m][1]);
            q8b[m] = pack8(pacc[m][2], pacc[m][3]);
            q4[m] = pack4(pacc[m][4]);
        }
        if constexpr (XATT) {
            // ---- scores against the text keys (5 blocks of 16, keys >= n_keys masked), softmax, o = P V: K and V^T fragments straight from the packed copy ----
            const int batch = (int)(((int64_t)tile * T6_ROWS / P.hw) / P.frames_per_batch);
            const bf16_t* KF = P.kvfrag + ((size_t)batch * 8 + wave) * 12800;
            u32x2 pp[5][5];                                       // probabilities P^T: [query block m][key block]: keys 4 kq .. of query l15
            {
                bf16x8 ka[5], kbb[5];
                s16x4 k4[5];
#pragma unroll
                for (int kb = 0; kb < 5; ++kb) {
                    ka[kb] = *reinterpret_cast<const bf16x8*>(KF + kb * 1280 + lane * 8);
                    kbb[kb] = *reinterpret_cast<const bf16x8*>(KF + kb * 1280 + 512 + lane * 8);
                    k4[kb] = *reinterpret_cast<const s16x4*>(KF + kb * 1280 + 1024 + lane * 4);
                }
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    f32x4 sc[5];
                    // (an accumulator is touched again only four MFMAs later, the order is pinned: back-to-back dependent MFMAs gave wrong scores for the
                    //  middle pixel blocks -- the hazard class of TB_SETTLE in temporal_block.hip)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kb = 0; kb < 5; ++kb) sc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[kb], q8a[m], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kb = 0; kb < 5; ++kb) sc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kbb[kb], q8b[m], sc[kb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kb = 0; kb < 5; ++kb) sc[kb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(k4[kb], q4[m], sc[kb], 0, 0, 0);
                    T6_SETTLE();
                    float mx = -3.0e38f;
#pragma unroll
                    for (int kb = 0; kb < 5; ++kb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            sc[kb][j] = (16 * kb + 4 * kq + j < P.n_keys) ? sc[kb][j] * P.scale_log2 : -3.0e38f;
                            mx = fmaxf(mx, sc[kb][j]);
                        }
                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    float sum = 0.f;
#pragma unroll
                    for (int kb = 0; kb < 5; ++kb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { sc[kb][j] = __builtin_amdgcn_exp2f(sc[kb][j] - mx); sum += sc[kb][j]; }
                    sum += __shfl_xor(sum, 16, 64);
                    sum += __shfl_xor(sum, 32, 64);
                    const float inv = 1.f / sum;
#pragma unroll
                    for (int kb = 0; kb < 5; ++kb) pp[m][kb] = u32x2{pack_bf2(sc[kb][0] * inv, sc[kb][1] * inv), pack_bf2(sc[kb][2] * inv, sc[kb][3] * inv)};
                }
            }
            {
                s16x4 vf[5][5];                                   // V^T: [channel block][key block]: (channel 16 b + l15, keys 16 kb + 4 kq ..)
#pragma unroll
                for (int b = 0; b < 5; ++b)
#pragma unroll
                    for (int kb = 0; kb < 5; ++kb) vf[b][kb] = *reinterpret_cast<const s16x4*>(KF + 6400 + ((b * 5 + kb) * 64 + lane) * 4);
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    f32x4 o[5];
#pragma unroll
                    for (int b = 0; b < 5; ++b) o[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 5; ++kb) {              // (key block outer: the five accumulators take turns)
                        union { u32x2 u; s16x4 s; } pb;
                        pb.u = pp[m][kb];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int b = 0; b < 5; ++b) o[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[b][kb], pb.s, o[b], 0, 0, 0);
                    }
                    T6_SETTLE();
#pragma unroll
                    for (int b = 0; b < 5; ++b) o_pk[m][b] = u32x2{pack_bf2(o[b][0], o[b][1]), pack_bf2(o[b][2], o[b][3])};
                }
            }
        } else {
        // ---- k, scores, softmax ----
        zero_pacc();
        T6_PART(1);
        T6_SETTLE();
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            // S^T[key][query] = K Q^T: A = k rows, B = q rows (both index the reduction by the same channel permutation)
            const bf16x8 k8a = pack8(pacc[m][0], pacc[m][1]), k8b = pack8(pacc[m][2], pacc[m][3]);
            const s16x4 k4 = pack4(pacc[m][4]);
            const f32x4 sc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k8a, q8a[m], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const f32x4 sc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k8b, q8b[m], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const f32x4 sc3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(k4, q4[m], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            T6_SETTLE();
            const f32x4 sc = sc1 + sc2 + sc3;
            float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])) * P.scale_log2;
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float e[4], sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { e[j] = __builtin_amdgcn_exp2f(sc[j] * P.scale_log2 - mx); sum += e[j]; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
            p_pk[m] = u32x2{pack_bf2(e[0] * inv, e[1] * inv), pack_bf2(e[2] * inv, e[3] * inv)};
        }
        // ---- v, o = P V ----
        zero_pacc();
        T6_PART(2);
        T6_SETTLE();
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            union { u32x2 u; s16x4 s; } pb;
            pb.u = p_pk[m];
            f32x4 o[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) o[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pack4(pacc[m][b]), pb.s, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);   // (query l15, channels 16 b + 4 kq ..)
            T6_SETTLE();
#pragma unroll
            for (int b = 0; b < 5; ++b) o_pk[m][b] = u32x2{pack_bf2(o[b][0], o[b][1]), pack_bf2(o[b][2], o[b][3])};
        }
        }  // (!XATT)
#undef T6_PART
#undef T6_STEP
    }
    __syncthreads();                                              // every head is done with m: o may overwrite it
    {
        int orow = l15 * T6_C, okq = kq, oxs = xsw;
        asm volatile("" : "+v"(orow), "+v"(okq), "+v"(oxs));
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const int col = 80 * wave + 16 * b + 4 * okq;
            const int off = orow + (((col >> 3) ^ oxs) << 3) + (col & 7);
#pragma unroll
            for (int m = 0; m < 5; ++m) *reinterpret_cast<u32x2*>(X + m * 16 * T6_C + off) = o_pk[m][b];
        }
    }
    // the residual rows (h of this tile) come back through the pass buffers under the out-projection
    issue_pass(rsH, 0);
    issue_pass(rsH, 1);
    issue_pass(rsH, 2);
    __syncthreads();                                              // X = o

    // ================= phase E: h' = o W_out^T + b + h =================
#pragma unroll
    for (int nb = 0; nb < 5; ++nb) {
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (P.b_out) t6_unpack4(*reinterpret_cast<const u32x2*>(P.b_out + ecol + nb * 16), b4);
#pragma unroll
        for (int mb = 0; mb < 5; ++mb) acc[mb][nb] = f32x4{b4[0], b4[1], b4[2], b4[3]};
    }
    project(P.w_out);
#pragma unroll
    for (int pss = 0; pss < 5; ++pss) {
        T6_VMCNT0();
        __syncthreads();                                          // pass pss has landed (all waves' pieces)
        bf16_t* Hs = PB + (pss % 3) * T6_PASS;
        {
            bf16_t* Hr = Hs + l15 * T6_C;
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                const int col = ecol + nb * 16, off = (((col >> 3) ^ xsw) << 3) + (col & 7);
                u32x2* ph = reinterpret_cast<u32x2*>(Hr + off);
                float hv[4];
                t6_unpack4(*ph, hv);
                *ph = u32x2{pack_bf2(acc[pss][nb][0] + hv[0], acc[pss][nb][1] + hv[1]), pack_bf2(acc[pss][nb][2] + hv[2], acc[pss][nb][3] + hv[3])};
            }
        }
        __syncthreads();                                          // the 16 finished rows (pixel pss, frames 0-15) are in the pass buffer
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = tid + 512 * j;
            if (j < 2 || tid < 256) {                             // 16 rows x 80 chunks = 2.5 x 512
                const int rl = c / T6_CPR, pc = c - rl * T6_CPR, r = 16 * pss + rl, lc = pc ^ ((r >> 1) & 7);
                const unsigned dst = row0 + (unsigned)(r & 15) * fstride + (unsigned)(r >> 4) * pstride + (unsigned)lc * 8;
                *reinterpret_cast<u32x4*>(P.out + dst) = *reinterpret_cast<const u32x4*>(Hs + c * 8);
            }
        }
        if (pss + 3 < 5) {
            __syncthreads();                                      // everybody has read the buffer: it takes pass pss + 3
            issue_pass(rsH, pss + 3);
        }
    }
}

// ---- LayerNorm + GEGLU projection of a feed-forward at the 20x32 level with the A operand RESIDENT (round 4) ------------------------------------------
// out[M][Cff] = (n W_v^T + b_v) * gelu(n W_g^T + b_g), n = LayerNorm(h): diffusers FeedForward(GEGLU) as BasicTransformerBlock / TemporalTransformerBlock
// call it (fmc/models/motion_module.py:295-299; diffusers 0.24 attention.py).  The 160x320 kernel re-requests the A tile for every one of its 16 n-tiles and
// spends 7 of 19.75 us per tile in its epilogue; here a workgroup keeps its 80 normalised rows in LDS (100 KiB), walks the Cff / 320 column chunks, and
// streams each chunk's 640 weight rows in fragment order straight into registers (the projections of temporal_block640_kernel).  Weight rows are
// permuted per wave to [v 0-15 | v 16-31 | v 32-39, g 32-39 | g 0-15 | g 16-31] (hip_ops.pack_geglu_frag80): value and gate of a column meet in one lane
// for four of the five 16-row blocks, one v_permlane32_swap serves the fifth.
__device__ __forceinline__ float t6_gelu_erf(float g) {       // Abramowitz & Stegun 7.1.26 (|error| < 1.5e-7), as gemm_conv.hip; default: common.h's fast form
    if (!FMC_GELU_EXACT) return fmc_gelu_fast(g);
    const float x = fabsf(g) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.f - p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);
    return 0.5f * g + 0.5f * fabsf(g) * e;
}

// G6_KNOCK (diagnostic builds, tools/scratch/r06/knock.sh; results are wrong by construction): bit 0 = no weight-fragment loads inside the loop, bit 1 = no
// A-fragment LDS reads, bit 2 = no gate / staging / stores (one dummy store keeps the accumulators alive)
#ifndef G6_KNOCK
#define G6_KNOCK 0
#endif
// G6_STAMP = 1 (diagnostic): wave 0 of workgroups 0 and 300 writes s_memtime stamps of its first 4 chunks over the head of `out` (results destroyed;
// tools/scratch/r06/stamp_geglu.py reads them back)
#ifndef G6_STAMP
#define G6_STAMP 0
#endif
#define G6_T(i)                                                                                                          \
    do {                                                                                                                 \
        if (G6_STAMP && tid == 0 && ch < 4 && (blockIdx.x == 0 || blockIdx.x == 300))                                    \
            reinterpret_cast<long long*>(P.out)[(blockIdx.x ? 64 : 0) + ch * 8 + (i)] = (long long)__builtin_readcyclecounter(); \
    } while (0)
#ifndef FMC_GEGLU320_ROWS160_DEFAULT
#define FMC_GEGLU320_ROWS160_DEFAULT 0
#endif
struct G6Params {
    const bf16_t* h; bf16_t* out;                          // h [M][640]; out [M][Cff] row-major
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const bf16_t* w;                                       // [Cff / 320 chunks][8 waves][20 k-steps][5 blocks][lane][8]
    const bf16_t* bias;                                    // [2 Cff] (value | gate) or NULL
    int64_t M; int cff;
    int out_blocked;                                       // out tile-major [M / 160][cff / 32][160][32] (GemmParams::a_blocked of the feed-forward's second GEMM)
};

// GC_ = 640 (8 waves, one tile per CU) | 320 (4 waves, 80 KiB: two workgroups per CU).  RH = 2 (GC_ = 320 only, round 5): ONE workgroup of 8 waves owns 160
// rows -- wave w and wave w + 4 run the same columns on the two 80-row halves in step (the per-chunk barriers keep them together), so the two requests for every
// weight fragment reach the CU's vector cache together instead of from two unrelated workgroups.
// SEQ (RH = 2, round 6): 4 waves own the 160 rows -- a wave runs BOTH 80-row halves against every weight fragment it loads (two accumulator sets, 200
// registers, one wave per SIMD).  What bounds the 80-row forms is the CU's vector-memory path, not L2: a k-step is 5 one-KiB fragment loads per wave = 16
// cycles each at 64 B / clk, 8 waves x 5 x 16 = 640 cycles per 800 cycles of MFMA (2 waves per SIMD x 25 MFMAs x 16), whether the fragments hit L1 (RH = 2
// without SEQ: same instruction count, measured no gain) or not.  Reusing a loaded fragment for twice the rows halves that: 4 x 5 x 16 = 320 per 800.
template <int GC_, int RH = 1, bool SEQ = false>
__global__ __launch_bounds__(GC_ / 80 * 64 * (SEQ ? 1 : RH), SEQ ? 1 : 2)
void geglu_direct_kernel(const G6Params P) {
    constexpr int RS = SEQ ? RH : 1;                             // 80-row halves a wave runs per loaded weight fragment
    constexpr int C = GC_, NW = C / 80, NT = 64 * NW * (SEQ ? 1 : RH), ROWS = T6_ROWS * RH, CPR = C / 8, KS = C / 32, GCOLS = 40 * NW, LPR = CPR / 10, WAVE_W = KS * 5 * 512;
    constexpr int NRG = NT / CPR;                                // row groups of the in-place normalisation pass
    constexpr int NWV = NT / 64;                                 // waves of the workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* X = reinterpret_cast<bf16_t*>(smem_raw);             // [80][640], chunk c of row r at chunk c ^ ((r >> 1) & 7)
    bf16_t* S = X + ROWS * C;                                    // staging [ROWS][GCOLS + 8]; phase A: (mean, rstd) x ROWS rows
    float* stats = reinterpret_cast<float*>(S);
    constexpr int SP = GCOLS + 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all % NW, rh = SEQ ? 0 : wave_all / NW;   // my column group / my (first) 80-row half
    const int l15 = lane & 15, kq = lane >> 4;
    const int xsw = (l15 >> 1) & 7;
    const int64_t m0 = (int64_t)blockIdx.x * ROWS;
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)P.h, 0, (int)(P.M * C * 2), 0x00020000);
    // ---- phase A: rows -> X, LayerNorm in place ----
#pragma unroll
    for (int j = 0; j < (ROWS * CPR / 64 + NWV - 1) / NWV; ++j) {
        const int q = wave_all + NWV * j;
        if (q < ROWS * CPR / 64) {
            const int idx = 64 * q + lane, r = idx / CPR, pc = idx - r * CPR, c = pc ^ ((r >> 1) & 7);
            t6_dma(rsH, (unsigned)(((m0 + r) * C + c * 8) * 2), X + 64 * q * 8);
        }
    }
    T6_VMCNT0();
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
    __syncthreads();                                              // X = LayerNorm(h); the staging region is free

    f32x4 acc[RS][5][5];
    const int nchunks = P.cff / GCOLS;
    // one descriptor over the whole packed weight; the chunk's first two k-steps are requested BEFORE the previous chunk's gating / staging / stores
    // (the fragment registers are idle there: requested at the head of a chunk their L2 round trip was exposed once per chunk)
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((int64_t)2 * P.cff * C * 2), 0x00020000);
    u32x4 wfr[3][5];
    int wl = 0;
    auto load_w = [&](int g) {
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) wfr[g % 3][nb] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wl, (g * 5 + nb) * 1024, 0);
    };
    auto chunk_base = [&](int ch) {
        int v = lane * 16 + (ch * NW + wave) * (WAVE_W * 2);
        asm volatile("" : "+v"(v));
        return v;
    };
    wl = chunk_base(0);
    load_w(0);
    load_w(1);
    if constexpr (G6_KNOCK & 1) load_w(2);                        // (diagnostic: every fragment register defined)
#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
        G6_T(0);
#pragma unroll
        for (int s = 0; s < RS; ++s)
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int b = 0; b < 5; ++b) acc[s][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto step = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + 2 < KS && !(G6_KNOCK & 1)) load_w(g + 2);
            int kqx = kq ^ xsw, xrow_o = (rh * T6_ROWS + l15) * C;
            asm volatile("" : "+v"(kqx), "+v"(xrow_o));
            const int xo = (G6_KNOCK & 2) ? 0 : xrow_o + ((g >> 1) * 8 + (((g & 1) * 4) ^ kqx)) * 8;
            __builtin_amdgcn_sched_barrier(0);
            // the RS * 5 row blocks of my halves as one sequence (block i = half i / 5, block i % 5: X rows 16 i ..), fragments two blocks ahead
            constexpr int NB_ = 5 * RS;
            u32x4 af3[3];
            if constexpr (G6_KNOCK & 2) {
                af3[0] = af3[1] = af3[2] = u32x4{(unsigned)xo + 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
            } else {
                af3[0] = *reinterpret_cast<const u32x4*>(X + 0 * 16 * C + xo);
                af3[1] = *reinterpret_cast<const u32x4*>(X + 1 * 16 * C + xo);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
#pragma unroll
            for (int mb = 0; mb < NB_; ++mb) {
                if constexpr (!(G6_KNOCK & 2))
                if (mb + 2 < NB_) af3[(mb + 2) % 3] = *reinterpret_cast<const u32x4*>(X + (mb + 2) * 16 * C + xo);
                union { bf16x8 v; u32x4 u; } a;
                a.u = af3[mb % 3];
#pragma unroll
                for (int nb = 0; nb < 5; ++nb) {
                    union { bf16x8 v; u32x4 u; } w;
                    w.u = wfr[g % 3][nb];
                    acc[mb / 5][mb % 5][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, a.v, acc[mb / 5][mb % 5][nb], 0, 0, 0);
                }
                if (!(G6_KNOCK & 2) && mb + 2 < NB_) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
            }
        };
#define G6_G(G) step(std::integral_constant<int, G>{})
        G6_G(0); G6_G(1); G6_G(2); G6_G(3); G6_G(4); G6_G(5); G6_G(6); G6_G(7); G6_G(8); G6_G(9);
        if constexpr (KS == 20) { G6_G(10); G6_G(11); G6_G(12); G6_G(13); G6_G(14); G6_G(15); G6_G(16); G6_G(17); G6_G(18); G6_G(19); }
#undef G6_G
        T6_SETTLE();
        G6_T(1);
        if constexpr (G6_KNOCK & 4) {
            float keep = 0.f;
#pragma unroll
            for (int s_ = 0; s_ < RS; ++s_)
#pragma unroll
                for (int a = 0; a < 5; ++a)
#pragma unroll
                    for (int b = 0; b < 5; ++b) keep += acc[s_][a][b][0] + acc[s_][a][b][1] + acc[s_][a][b][2] + acc[s_][a][b][3];
            if (keep == 12345.678f) P.out[tid] = (bf16_t)1;
            if (ch + 1 < nchunks) {
                wl = chunk_base(ch + 1);
                load_w(0);
                load_w(1);
            }
            continue;
        }
        // ---- gate: my 40 gated columns gc0 .. gc0 + 39 of this chunk (the bias first: its loads must be OLDER than the weight prefetch below, or waiting
        //      for them waits for the prefetch too) ----
        const int gc0 = ch * GCOLS + wave * 40;
        float bv[3][4], bg[3][4];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int c0 = gc0 + 16 * p + 4 * (p == 2 ? (kq & 1) : kq);
            if (P.bias) {
                t6_unpack4(*reinterpret_cast<const u32x2*>(P.bias + c0), bv[p]);
                t6_unpack4(*reinterpret_cast<const u32x2*>(P.bias + P.cff + c0), bg[p]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[p][j] = bg[p][j] = 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        G6_T(2);
        if (ch + 1 < nchunks) {                                   // the next chunk's first two k-steps (stages 0, 1: the loop's last steps have left them)
            wl = chunk_base(ch + 1);
            load_w(0);
            load_w(1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < RS; ++s)
#pragma unroll
        for (int mb = 0; mb < 5; ++mb) {
            bf16_t* Sr = S + ((rh + s) * T6_ROWS + mb * 16 + l15) * SP + wave * 40;
#pragma unroll
            for (int p = 0; p < 2; ++p) {                          // blocks (0, 3) and (1, 4): value and gate in the same lane
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (acc[s][mb][p][j] + bv[p][j]) * t6_gelu_erf(acc[s][mb][3 + p][j] + bg[p][j]);
                *reinterpret_cast<u32x2*>(Sr + 16 * p + 4 * kq) = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
            }
            {                                                     // block 2: lanes kq < 2 hold value columns 32 + 4 kq .., lanes kq >= 2 their gates
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned own = __float_as_uint(acc[s][mb][2][j]);
                    const auto sw = __builtin_amdgcn_permlane32_swap(own, own, false, false);
                    const float gate = __uint_as_float((unsigned)sw[1]);          // (lanes 0-31 see lane + 32's value)
                    o[j] = (acc[s][mb][2][j] + bv[2][j]) * t6_gelu_erf(gate + bg[2][j]);
                }
                if (kq < 2) *reinterpret_cast<u32x2*>(Sr + 32 + 4 * kq) = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
            }
        }
        G6_T(3);
        __syncthreads();
        G6_T(4);
        // ---- whole-row stores: ROWS rows x GCOLS / 8 sixteen-byte chunks = 6.25 per thread (12.5 with SEQ) ----
        // (the per-thread row / column of the seven stores recomputed from an opaque copy of tid in every chunk: visible as loop invariants, hipcc
        //  hoists the seven 64-bit row offsets out of the chunk loop, spills them -- 20 VGPRs -- and reloads each behind its own s_waitcnt vmcnt(0))
        int tq = tid;
        asm volatile("" : "+v"(tq));
        constexpr int NCH = ROWS * (GCOLS / 8), NIT = (NCH + NT - 1) / NT;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = tq + NT * it;
            if (it < NIT - 1 || c < NCH) {
                const int r = c / (GCOLS / 8), cc = c - r * (GCOLS / 8);
                const int64_t m = m0 + r;
                const int col = ch * GCOLS + cc * 8;
                const int64_t dst = P.out_blocked ? (((m / 160) * (P.cff >> 5) + (col >> 5)) * 160 + m % 160) * 32 + (col & 31) : m * P.cff + col;
                *reinterpret_cast<u32x4*>(P.out + dst) = *reinterpret_cast<const u32x4*>(S + r * SP + cc * 8);
            }
        }
        G6_T(5);
        __syncthreads();                                          // the staging region is free again
        G6_T(6);
    }
}

// text k | v `[batch][S][2 C]` (C = 640, 8 heads x 80) -> the MFMA fragments phase D of the XATT block reads: per (batch, head)
//   K: 5 key blocks x [a: lane x 8 | b: lane x 8 | tail: lane x 4]: lane (l15 = key in block, kq) holds channels {4 kq .. + 3, 16 + 4 kq .. + 3} (+ 32 for b),
//      64 + 4 kq .. + 3 for the tail -- the channel permutation q comes out of the swapped product in;
//   V^T: [channel block 5][key block 5][lane][4]: lane (l15 = channel in block, kq) holds keys 16 kb + 4 kq .. + 3.   Keys >= S are zero.
__global__ __launch_bounds__(256) void xattn_pack_kv_kernel(const bf16_t* __restrict__ kv, bf16_t* __restrict__ out, int S, int64_t ldb) {
    const int bh = blockIdx.x, b = bh >> 3, h = bh & 7;
    const bf16_t* kb_ = kv + (int64_t)b * ldb + h * 80;            // k of (b, h): row stride 1280
    const bf16_t* vb_ = kb_ + 640;
    bf16_t* o = out + (size_t)bh * 12800;
    for (int g = threadIdx.x; g < 3200; g += 256) {               // groups of 4 output elements
        u32x2 val = u32x2{0u, 0u};
        if (g < 1600) {                                           // K: key block kb = g / 320; inside: a (128 groups), b (128), tail (64)
            const int kb = g / 320, r = g - kb * 320;
            int lane, ch;
            if (r < 256) {
                const int part = r >> 7, q = r & 127;
                lane = q >> 1;
                ch = 32 * part + 16 * (q & 1) + 4 * (lane >> 4);
            } else {
                lane = r - 256;
                ch = 64 + 4 * (lane >> 4);
            }
            const int key = 16 * kb + (lane & 15);
            if (key < S) val = *reinterpret_cast<const u32x2*>(kb_ + (int64_t)key * 1280 + ch);
        } else {                                                  // V^T: (cb, kb, lane)
            const int q = g - 1600, lane = q & 63, kb = (q >> 6) % 5, cb = (q >> 6) / 5;
            const int ch = 16 * cb + (lane & 15), key0 = 16 * kb + 4 * (lane >> 4);
            unsigned short e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = key0 + j < S ? vb_[(int64_t)(key0 + j) * 1280 + ch] : (unsigned short)0;
            val = u32x2{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16)};
        }
        *reinterpret_cast<u32x2*>(o + g * 4) = val;
    }
}

}  // namespace

// called by fmc_temporal_block_bf16 (temporal_block.hip) for channels == 640; the arguments are already checked for NULL / alignment there
int fmc_temporal_block640_launch(const void* h, void* out, const float* ln_gamma, const float* ln_bpe, float ln_eps, const void* w_merge_frag,
                                 const void* pose_term, float merge_scale, const void* w_qkv_packed, const void* w_out_frag, const void* b_out,
                                 int n_clips, int hw, float scale, hipStream_t st) {
    if (hw % 5) FMC_FAIL(FMC_E_SHAPE, "temporal_block_bf16: C = 640 wants pixels %% 5 == 0 (got %d)", hw);
    T6Params P{};
    P.h = (const bf16_t*)h; P.out = (bf16_t*)out; P.ln_gamma = ln_gamma; P.ln_bpe = ln_bpe; P.ln_eps = ln_eps;
    P.w_merge = (const bf16_t*)w_merge_frag; P.pose_term = (const bf16_t*)pose_term; P.merge_scale = merge_scale;
    P.w_qkv = (const bf16_t*)w_qkv_packed; P.w_out = (const bf16_t*)w_out_frag; P.b_out = (const bf16_t*)b_out;
    P.n_clips = n_clips; P.hw = hw; P.tpc = hw / 5;
    P.total_rows = (int64_t)n_clips * T6_F * hw;
    P.scale_log2 = scale * 1.4426950408889634f;
    const unsigned grid = (unsigned)(n_clips * P.tpc);
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block640_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, T6_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block640_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, T6_LDS);
        raised = true;
    }
    if (w_merge_frag) hipLaunchKernelGGL((temporal_block640_kernel<true, false>), dim3(grid), dim3(512), T6_LDS, st, P);
    else hipLaunchKernelGGL((temporal_block640_kernel<false, false>), dim3(grid), dim3(512), T6_LDS, st, P);
    FMC_CHECK_LAUNCH("fmc_temporal_block_bf16 (C = 640)");
    return 0;
}

// The text cross-attention block of the spatial transformer at the 20x32 level in one launch (BasicTransformerBlock, diffusers 0.24 as driven by
// fmc/models/unet_blocks.py:323-333 / fmc/models/attention_processor.py:50-69,148-154):   out = to_out(softmax(to_q(LayerNorm(h)) k^T d^-1/2) v) + b + h
// for tokens h [images][hw][640] (hw % 80 == 0), 8 heads x 80, text k | v packed by fmc_xattn_pack_kv.  Replaces LayerNorm + to_q GEMM + cross-attention
// kernel + to_out GEMM.  ln_bpe: fp32 [16][640] rows all equal to the LayerNorm beta (the kernel shares phase A with the temporal block).
extern "C" int fmc_xattn_block640_bf16(const void* h, void* out, const float* ln_gamma, const float* ln_bpe, float ln_eps, const void* w_q_packed,
                                       const void* kvfrag, const void* w_out_frag, const void* b_out, int n_images, int hw, int n_keys, int images_per_text,
                                       float scale, void* stream) {
    if (!h || !out || !ln_gamma || !ln_bpe || !w_q_packed || !kvfrag || !w_out_frag) FMC_FAIL(FMC_E_NULL, "xattn_block640_bf16: NULL tensor");
    if (n_images <= 0 || hw <= 0 || hw % 80 || n_keys <= 0 || n_keys > 80 || images_per_text <= 0 || n_images % images_per_text)
        FMC_FAIL(FMC_E_SHAPE, "xattn_block640_bf16: hw %% 80 == 0, 1 <= keys <= 80, images %% images_per_text == 0 (got hw=%d keys=%d images=%d / %d)", hw, n_keys,
                 n_images, images_per_text);
    if ((int64_t)n_images * hw * 640 * 2 >= ((int64_t)1 << 31)) FMC_FAIL(FMC_E_SHAPE, "xattn_block640_bf16: tensor of 2 GiB or more");
    if (!fmc_aligned16(h) || !fmc_aligned16(out) || !fmc_aligned16(w_q_packed) || !fmc_aligned16(kvfrag) || !fmc_aligned16(w_out_frag) || !fmc_aligned16(ln_gamma) ||
        !fmc_aligned16(ln_bpe) || (b_out && !fmc_aligned16(b_out)))
        FMC_FAIL(FMC_E_ALIGN, "xattn_block640_bf16: tensors must be 16-byte aligned");
    T6Params P{};
    P.h = (const bf16_t*)h; P.out = (bf16_t*)out; P.ln_gamma = ln_gamma; P.ln_bpe = ln_bpe; P.ln_eps = ln_eps;
    P.w_qkv = (const bf16_t*)w_q_packed; P.w_out = (const bf16_t*)w_out_frag; P.b_out = (const bf16_t*)b_out;
    P.kvfrag = (const bf16_t*)kvfrag; P.n_keys = n_keys; P.frames_per_batch = images_per_text;
    P.n_clips = 1; P.hw = hw; P.tpc = 1;
    P.total_rows = (int64_t)n_images * hw;
    P.scale_log2 = scale * 1.4426950408889634f;
    const unsigned grid = (unsigned)(P.total_rows / T6_ROWS);
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block640_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, T6_LDS);
        raised = true;
    }
    hipLaunchKernelGGL((temporal_block640_kernel<false, true>), dim3(grid), dim3(512), T6_LDS, (hipStream_t)stream, P);
    FMC_CHECK_LAUNCH("fmc_xattn_block640_bf16");
    return 0;
}

// kv: bf16 [batch][S][1280] (k | v of the fused to_k / to_v projection, rows `ld_batch` elements apart per batch entry) -> out: bf16 [batch][8][12800]
extern "C" int fmc_xattn_pack_kv(const void* kv, void* out, int batch, int S, int64_t ld_batch, void* stream) {
    if (!kv || !out) FMC_FAIL(FMC_E_NULL, "xattn_pack_kv: NULL tensor");
    if (batch <= 0 || S <= 0 || S > 80 || ld_batch < (int64_t)S * 1280) FMC_FAIL(FMC_E_SHAPE, "xattn_pack_kv: 1 <= S <= 80 text tokens of 2 x 640 channels");
    if (((uintptr_t)kv & 7) || !fmc_aligned16(out)) FMC_FAIL(FMC_E_ALIGN, "xattn_pack_kv: alignment");
    hipLaunchKernelGGL(xattn_pack_kv_kernel, dim3((unsigned)(batch * 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)kv, (bf16_t*)out, S, ld_batch);
    FMC_CHECK_LAUNCH("fmc_xattn_pack_kv");
    return 0;
}

// LayerNorm + GEGLU projection with the A operand resident (see geglu640_kernel): h bf16 [M][640] (M % 80 == 0), out bf16 [M][cff] row-major,
// w_packed = hip_ops.pack_geglu_frag80 of the [2 cff, 640] projection (cff % 320 == 0), bias bf16 [2 cff] or NULL, ln_gamma / ln_beta fp32 [640].
extern "C" int fmc_geglu640_ln_bf16(const void* h, void* out, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w_packed, const void* bias,
                                    int64_t M, int cff, int out_blocked, void* stream) {
    if (!h || !out || !ln_gamma || !ln_beta || !w_packed) FMC_FAIL(FMC_E_NULL, "geglu640_ln_bf16: NULL tensor");
    if (M <= 0 || M % 80 || cff <= 0 || cff % 320 || M * 640 * 2 >= ((int64_t)1 << 31) || M * cff * 2 >= ((int64_t)1 << 40))
        FMC_FAIL(FMC_E_SHAPE, "geglu640_ln_bf16: M %% 80 == 0, cff %% 320 == 0 (got M=%lld cff=%d)", (long long)M, cff);
    if (out_blocked && M % 160) FMC_FAIL(FMC_E_SHAPE, "geglu640_ln_bf16: the tile-major output needs M %% 160 == 0 (got M=%lld)", (long long)M);
    if (!fmc_aligned16(h) || !fmc_aligned16(out) || !fmc_aligned16(w_packed) || !fmc_aligned16(ln_gamma) || !fmc_aligned16(ln_beta) || (bias && ((uintptr_t)bias & 7)))
        FMC_FAIL(FMC_E_ALIGN, "geglu640_ln_bf16: tensors must be 16-byte aligned");
    G6Params P{};
    P.h = (const bf16_t*)h; P.out = (bf16_t*)out; P.ln_gamma = ln_gamma; P.ln_beta = ln_beta; P.ln_eps = ln_eps;
    P.w = (const bf16_t*)w_packed; P.bias = (const bf16_t*)bias; P.M = M; P.cff = cff; P.out_blocked = out_blocked != 0;
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&geglu_direct_kernel<640>), hipFuncAttributeMaxDynamicSharedMemorySize, T6_LDS);
        raised = true;
    }
    hipLaunchKernelGGL(geglu_direct_kernel<640>, dim3((unsigned)(M / 80)), dim3(512), T6_LDS, (hipStream_t)stream, P);
    FMC_CHECK_LAUNCH("fmc_geglu640_ln_bf16");
    return 0;
}

// the same at the 40x64 level: h bf16 [M][320], w_packed = hip_ops.pack_geglu_frag80 of the [2 cff, 320] projection, cff % 160 == 0; 4 waves and 77 KiB of
// LDS per workgroup, two workgroups per CU
extern "C" int fmc_geglu320_ln_bf16(const void* h, void* out, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* w_packed, const void* bias,
                                    int64_t M, int cff, int out_blocked, void* stream) {
    if (!h || !out || !ln_gamma || !ln_beta || !w_packed) FMC_FAIL(FMC_E_NULL, "geglu320_ln_bf16: NULL tensor");
    if (M <= 0 || M % 80 || cff <= 0 || cff % 160 || M * 320 * 2 >= ((int64_t)1 << 31))
        FMC_FAIL(FMC_E_SHAPE, "geglu320_ln_bf16: M %% 80 == 0, cff %% 160 == 0 (got M=%lld cff=%d)", (long long)M, cff);
    if (out_blocked && M % 160) FMC_FAIL(FMC_E_SHAPE, "geglu320_ln_bf16: the tile-major output needs M %% 160 == 0 (got M=%lld)", (long long)M);
    if (!fmc_aligned16(h) || !fmc_aligned16(out) || !fmc_aligned16(w_packed) || !fmc_aligned16(ln_gamma) || !fmc_aligned16(ln_beta) || (bias && ((uintptr_t)bias & 7)))
        FMC_FAIL(FMC_E_ALIGN, "geglu320_ln_bf16: tensors must be 16-byte aligned");
    G6Params P{};
    P.h = (const bf16_t*)h; P.out = (bf16_t*)out; P.ln_gamma = ln_gamma; P.ln_beta = ln_beta; P.ln_eps = ln_eps;
    P.w = (const bf16_t*)w_packed; P.bias = (const bf16_t*)bias; P.M = M; P.cff = cff; P.out_blocked = out_blocked != 0;
    constexpr int lds = (80 * 320 + 80 * 168) * 2;
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&geglu_direct_kernel<320>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&geglu_direct_kernel<320, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&geglu_direct_kernel<320, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * lds);
        raised = true;
    }
    // FMC_GEGLU320_ROWS160: 0 = 80-row workgroups of 4 waves, two per CU; 1 = 160 rows on 8 waves (round 5 A/B: 215 - 225 us either way on 81920 x 2560 x 320,
    // gpurun_out/r05n/geglu_ab.txt -- pairing the halves in one workgroup leaves every wave's fragment loads in place); 2 = 160 rows on 4 waves, each weight
    // fragment used for both halves (SEQ, round 6)
    static const int rows160 = [] { const char* e = getenv("FMC_GEGLU320_ROWS160"); return e ? atoi(e) : FMC_GEGLU320_ROWS160_DEFAULT; }();
    if (rows160 == 2 && M % 160 == 0) hipLaunchKernelGGL((geglu_direct_kernel<320, 2, true>), dim3((unsigned)(M / 160)), dim3(256), 2 * lds, (hipStream_t)stream, P);
    else if (rows160 == 1 && M % 160 == 0) hipLaunchKernelGGL((geglu_direct_kernel<320, 2>), dim3((unsigned)(M / 160)), dim3(512), 2 * lds, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(geglu_direct_kernel<320>, dim3((unsigned)(M / 80)), dim3(256), lds, (hipStream_t)stream, P);
    FMC_CHECK_LAUNCH("fmc_geglu320_ln_bf16");
    return 0;
}
