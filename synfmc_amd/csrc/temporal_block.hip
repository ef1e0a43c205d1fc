// The fused temporal attention block of the motion modules at the 40x64 level (C = 320, 8 heads x 40, F = 16 frames), gfx950.
//
// Replaces, per attention block of `TemporalTransformerBlock.forward` (fmc/models/motion_module.py:287-300), the chain
//     n = LayerNorm_i(h) (+ pos_encoder)                                   motion_module.py:288-289, 349-356
//     m = qkv_merge(n + pose) * scale + n          (Camera Adapter, block 0) attention_processor.py:255-258
//     q, k, v = to_q(m), to_k(m), to_v(m)                                  attention_processor.py:260-272
//     o = softmax(q k^T d^-1/2) v   per (pixel, head) over the F frames    attention_processor.py:271-281 / :61-67
//     h' = to_out(o) + bias + h                                            attention_processor.py:283-291, motion_module.py:289-297
// which ran as LayerNorm-in-epilogue + three GEMM launches + the temporal attention launch with `m`, `q | k | v` and `o` making a round trip
// through HBM each (~520 MB per block at 16x320x512, CFG batch 2).  Everything after the LayerNorm is local to a PIXEL (temporal attention
// mixes the 16 frames of one pixel, the projections are row-wise), so a tile of 10 pixels x 16 frames = 160 rows can be taken from `h` to
// `h'` without leaving the CU:  h is read once, the pose term once, h' written once (157 MB per block).
//
// One persistent workgroup of 8 waves per CU, 160 KiB of LDS = [X: the 160 x 320 bf16 tile, 100 KiB | RING: 3 x 20-KiB weight sub-tiles].
// Per tile (rows ordered pixel-major: row = 16 * pixel + frame, so a 16-row MFMA block IS one pixel):
//   A  h rows -> X by LDS-DMA (row-major, 16-byte chunks XOR-swizzled through the source address so that every ds_read_b128 fragment read
//      below is bank-conflict free), LayerNorm (+ positional-encoding row of the frame) in place: x.
//   B  (block 0) m = s * x W_m^T + pose_term + x: the 160 x 320 x 320 product exactly as gemm160_kernel runs it (8 waves = 2 x 4, a wave 80 x 80
//      outputs as 5 x 5 v_mfma_f32_16x16x32_bf16, two wave rows one barrier apart, W sub-tiles of 32 k by `buffer_load ... lds` into the ring,
//      counted vmcnt) -- but the A operand is RESIDENT: fragments come straight out of X, no A requests.  The pose term arrives in accumulator
//      layout under the main loop; m overwrites x in place (each lane rewrites exactly the words it read).
//   D  wave w = head w.  q, k, v of the head for all 160 rows: three passes of 10 k-steps, A fragments from X, the head's weight fragments
//      (pre-packed in fragment order: one contiguous KiB per 16-row block and k-step) by plain buffer loads two k-steps ahead -- no LDS
//      staging of weights (nothing is shared between heads), no barrier inside the phase.  q and k come out of the "swapped" product
//      (lane = frame, 4 consecutive head channels), which IS the operand layout of S^T = K Q^T once both use the same permutation of the
//      reduction index; v comes out of the un-swapped product (lane = channel, 4 consecutive frames) = the A operand of O^T = V^T P^T; P^T
//      leaves the softmax in the B-operand layout.  So q, k, v, the scores, the probabilities and o never touch LDS: 40 = 32 + 8 head
//      channels = one 16x16x32 plus one 16x16x16 MFMA for the scores (the 8-channel tails of q and k share one 16-row weight block), three
//      16x16x16 MFMAs for PV.  o (bf16) is parked in registers until every wave is done with m, then written over it.
//   E  h' = o W_out^T + b + h like B (h in accumulator layout under the main loop), staged per wave row in the ring region and stored
//      with whole-row 16-byte stores; optionally the (mean, rstd) of the h' rows for a consumer GEMM that applies the next LayerNorm itself.
// Roofline: MFMA.  Algorithmic flops per block = 2 M C (C [merge] + 3 C + C) + 4 M F C = 84.3 GF (block 0) / 67.5 GF (block 1) at
// M = 81920; HBM-side bytes 157 / 105 MB.  bf16 only (fp32 parity mode keeps the un-fused chain on the split-bf16 kernels).
#include <type_traits>

#include "common.h"

namespace {

constexpr int TB_C = 320, TB_ROWS = 160, TB_PIX = 10, TB_F = 16;
constexpr int TB_X_ELEMS = TB_ROWS * TB_C;             // 51200 bf16 = 100 KiB
constexpr int TB_SUB = 320 * 32;                       // one weight sub-tile: 320 rows x 32 k (20 KiB)
constexpr int TB_LDS = (TB_X_ELEMS + 3 * TB_SUB) * 2;  // 163840 B
constexpr int TB_QKV_HEAD = 8 * 10 * 512;              // bf16 elements of one head's packed q|k|v weights: 8 blocks x 10 k-steps x 1 KiB

struct TBParams {
    const bf16_t* h; bf16_t* out;                      // [clips, 16, hw, 320] channels-last video tokens
    const float* ln_gamma;                             // [320]
    const float* ln_bpe;                               // [16][320]: LayerNorm beta + positional-encoding row of frame f
    float ln_eps;
    const bf16_t* w_merge;                             // tile-major [10][320][32] or NULL (no Camera-Adapter merge)
    const bf16_t* pose_term;                           // s * (W_m pose + b_m), layout of h (read when w_merge)
    float merge_scale;
    const bf16_t* w_qkv;                               // [8 heads][q: 3 blocks | k: 2 | v: 3][10 k-steps][block][lane][8]  (hip_ops.pack_temporal_qkv)
    const bf16_t* w_out;                               // tile-major [10][320][32]
    const bf16_t* b_out;                               // [320] or NULL
    float* ln_stats; float ln_stats_eps;               // optional: (mean, rstd) of every h' row -> [rows][2]
    int n_clips, hw, tiles;
    float scale_log2;                                  // d^-1/2 * log2(e)
    long long* dbg_times;                              // diagnostic (fmc_temporal_block_set_debug): [workgroup][tile slot 0..3][8] s_memrealtime stamps of wave 0
    // XATT (the text cross-attention block of the spatial transformer at this level on the same skeleton): tokens [images][hw][320], a tile = 160 consecutive rows
    const bf16_t* kvfrag;                              // [batch][8 heads][K: 5 key blocks x (512 + 256) | V^T: 3 x 5 x 256] (fmc_xattn_pack_kv, C = 320)
    int n_keys, images_per_text;
    int64_t total_rows;
};

__device__ __forceinline__ void tb_dma(const __amdgpu_buffer_rsrc_t& rs, unsigned voff, int soff, void* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, soff, 0, 0);
}
__device__ __forceinline__ void unpack4(const u32x2& w, float (&o)[4]) {
    o[0] = __uint_as_float(w[0] << 16); o[1] = __uint_as_float(w[0] & 0xffff0000u);
    o[2] = __uint_as_float(w[1] << 16); o[3] = __uint_as_float(w[1] & 0xffff0000u);
}
// MFMA results are consumed by VALU code only behind TB_SETTLE(): a scheduling fence + 32 idle issue cycles.  hipcc (ROCm 7.2) interleaves the
// consumers of a k-loop's accumulators (v_cvt_pk, the score MFMAs) into the loop's last step with the wait states ITS tables give the gfx950
// v_mfma_f32_16x16x32_bf16 -- observed: deterministic garbage in the pixel whose accumulators were read first, moving from pixel to pixel with
// the register allocation of the build (q right, probabilities wrong); with the fence every build since has been right.  ~0.1 % of a tile.
#define TB_SETTLE()                                                                                                      \
    do {                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    } while (0)
#define TB_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define TB_STAMP(i)                                                                                                      \
    do {                                                                                                                 \
        if (P.dbg_times && tid == 0 && tslot < 4) P.dbg_times[((int64_t)blockIdx.x * 4 + tslot) * 8 + (i)] = (long long)wall_clock64(); \
    } while (0)

// HAS_MERGE: attention block 0 (Camera Adapter); STATS: also emit the (mean, rstd) of the output rows
template <bool HAS_MERGE, bool STATS, bool XATT = false>
__global__ __launch_bounds__(512, 2)
void temporal_block_kernel(const TBParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* X = reinterpret_cast<bf16_t*>(smem_raw);             // [160][320], chunk c of row r at chunk c ^ ((r >> 1) & 7)
    bf16_t* RING = X + TB_X_ELEMS;                               // 3 sub-tile buffers [320][32] (64-byte rows, gemm160's image)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, kq = lane >> 4;
    const bool w_wave = wave < 5;                                // waves 0-4 request the weight sub-tiles (4 one-KiB pieces each)

    const int64_t total_elems = P.total_rows * TB_C;
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)P.h, 0, (int)(total_elems * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsPT = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_MERGE ? P.pose_term : P.h), 0, (int)(total_elems * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsWM = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_MERGE ? P.w_merge : P.w_out), 0, 320 * 320 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsWO = __builtin_amdgcn_make_buffer_rsrc((void*)P.w_out, 0, 320 * 320 * 2, 0x00020000);

    // ---- weight sub-tile requests (gemm160's image): piece i = 4 wave + e covers rows 16 i .. 16 i + 15 of the sub-tile ---------------------
    const int prow = lane >> 2, pch = lane & 3, psrc = pch ^ (3 * ((prow >> 3) & 1));
    unsigned w_vo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w_vo[e] = (unsigned)(((16 * (4 * wave + e) + prow) * 32 + psrc * 8) * 2);
    auto issue_w = [&](const __amdgpu_buffer_rsrc_t& rs, int sub, int buf) {
        if (w_wave) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tb_dma(rs, w_vo[e], sub * (TB_SUB * 2), RING + buf * TB_SUB + 16 * (4 * wave + e) * 32);
        }
    };
    // fragment addresses: W fragment (row l15 of a 16-row block, logical k-chunk kq) inside a ring buffer; A fragment inside X
    const int wfrag = (wc * 80 + l15) * 32 + (kq ^ (3 * ((l15 >> 3) & 1))) * 8;
    const int xsw = (l15 >> 1) & 7;                              // chunk swizzle of my fragment rows (row = 16 mb + l15 + 80 wr: the same for all)
    // my accumulator rows / columns in phases B, E: acc[mb][nb][j] = (row 80 wr + 16 mb + l15, column 80 wc + 16 nb + 4 kq + j)
    const int ecol = wc * 80 + 4 * kq;

    f32x4 acc[5][5];
    bf16x8 wf[5], af[5];
    // one 160 x 320 x 320 product with the A operand resident in X: W sub-tiles `rs` through the ring, the two wave rows one barrier apart.
    // `mid(g)` runs in LOAD(g) right after the requests (prefetch hooks).  On entry: sub-tiles 0, 1 requested into buffers 0, 1 and
    // `vm_first` leaves everything younger than sub-tile 0 outstanding.  VM_MID = loads `mid` issues at g == 2 (per lane).
    auto read_frags = [&](int g, int buf) {
        // (the per-lane fragment offsets are recomputed here from an opaque copy: left visible as loop invariants of the persistent tile loop,
        //  hipcc materialises the addresses of all 20 + 30 unrolled steps up front, spills them and reloads them inside the main loops -- and
        //  every scratch reload carries an `s_waitcnt vmcnt(0)` that drains the weight stream)
        int wfrag_o = wfrag, xrow_o = (wr * 80 + l15) * TB_C, kqx = kq ^ xsw;
        asm volatile("" : "+v"(wfrag_o), "+v"(xrow_o), "+v"(kqx));
        const bf16_t* Ws = RING + buf * TB_SUB + wfrag_o;
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) {
            union { bf16x8 v; u32x4 u; } t;
            t.u = *reinterpret_cast<const u32x4*>(Ws + nb * 16 * 32);
            wf[nb] = t.v;
        }
        // chunk (4 g + kq) ^ xsw = 8 (g >> 1) + ((4 (g & 1) + kq) ^ xsw) = 8 (g >> 1) + ((4 (g & 1)) ^ (kq ^ xsw))   (kq < 4, xsw < 8)
        const bf16_t* As = X + xrow_o + ((g >> 1) * 8 + (((g & 1) * 4) ^ kqx)) * 8;
#pragma unroll
        for (int mb = 0; mb < 5; ++mb) {
            union { bf16x8 v; u32x4 u; } t;
            t.u = *reinterpret_cast<const u32x4*>(As + mb * 16 * TB_C);
            af[mb] = t.v;
        }
    };
#define TB_MMA()                                                                                                         \
    do {                                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        __builtin_amdgcn_s_barrier();                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                   \
        _Pragma("unroll") for (int mb = 0; mb < 5; ++mb)                                                                 \
            _Pragma("unroll") for (int nb = 0; nb < 5; ++nb)                                                             \
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb], af[mb], acc[mb][nb], 0, 0, 0);             \
        __builtin_amdgcn_s_setprio(0);                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        __builtin_amdgcn_s_barrier();                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    } while (0)

    const int n_tiles_per_clip = P.hw / TB_PIX;
    // tile row r = 16 hi + lo lives at element row0 + lo * fstride + hi * pstride: temporal = (frame lo, pixel hi) of the clip; XATT = row 160 tile + r
    const unsigned fstride = XATT ? (unsigned)TB_C : (unsigned)(P.hw * TB_C), pstride = XATT ? (unsigned)(16 * TB_C) : (unsigned)TB_C;
    auto tile_row0 = [&](int t) {
        if (XATT) return (unsigned)((int64_t)t * TB_ROWS * TB_C);
        const int clip = t / n_tiles_per_clip, p0 = (t - clip * n_tiles_per_clip) * TB_PIX;
        return (unsigned)(((int64_t)clip * TB_F * P.hw + p0) * TB_C);
    };

    // rows [r0, r0 + nr) of a tile of `rs` (h or the pose term) -> LDS at `dst` by LDS-DMA, row-major with X's chunk swizzle (16-byte chunk c of tile
    // row r at chunk c ^ ((r >> 1) & 7): the source address carries it).  nr * 40 chunks = pieces of 64; piece q = wave + 8 j.
    auto issue_rows = [&](const __amdgpu_buffer_rsrc_t& rs, int t, int r0, int nr, bf16_t* dst) {
        int ln = lane;
        asm volatile("" : "+v"(ln));                  // (per-tile address arithmetic must not be hoisted out of the persistent loop: 40 live registers)
        const unsigned row0 = tile_row0(t);
        const int pieces = nr * 40 / 64;
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            const int q = wave + 8 * j;
            if (q < pieces) {
                const int idx = 64 * q + ln, rl = idx / 40, pc = idx - rl * 40, r = r0 + rl, c = pc ^ ((r >> 1) & 7);
                const unsigned src = (row0 + (unsigned)(r & 15) * fstride + (unsigned)(r >> 4) * pstride + (unsigned)c * 8) * 2;
                tb_dma(rs, src, 0, dst + 64 * q * 8);
            }
        }
    };
    auto issue_h = [&](int t) { issue_rows(rsH, t, 0, TB_ROWS, X); };
    // epilogue side operand (pose term) of pass p = the rows of accumulator block mb = p of BOTH wave rows (tile rows 16 p .. and 80 + 16 p ..): 32 rows
    // = one 20-KiB ring buffer, rows 0-15 / 16-31, X's swizzle.  20 one-KiB pieces: 3 from waves 0-3, 2 from waves 4-7.
    auto issue_pass = [&](const __amdgpu_buffer_rsrc_t& rs, int t, int pss, int buf) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const unsigned row0 = tile_row0(t);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = wave + 8 * j;
            if (q < 20) {
                const int idx = 64 * q + ln, rl = idx / 40, pc = idx - rl * 40, r = (rl < 16 ? 16 * pss + rl : 64 + 16 * pss + rl), c = pc ^ ((r >> 1) & 7);
                const unsigned src = (row0 + (unsigned)(r & 15) * fstride + (unsigned)(r >> 4) * pstride + (unsigned)c * 8) * 2;
                tb_dma(rs, src, 0, RING + buf * TB_SUB + 64 * q * 8);
            }
        }
    };
#define TB_VMCNT2(a, b)                                                                                                  \
    do {                                                                                                                 \
        if (wave < 4) TB_VMCNT(a); else TB_VMCNT(b);                                                                     \
    } while (0)
    if ((int)blockIdx.x < P.tiles) issue_h(blockIdx.x);
    float* stats = reinterpret_cast<float*>(smem_raw + TB_X_ELEMS * 2 + 53248);      // (mean, rstd) x 160 rows: behind the epilogue's staging rows in the ring region

    int tslot = -1;
    for (int tile = blockIdx.x; tile < P.tiles; tile += gridDim.x) {
        ++tslot;
        TB_STAMP(0);
        const int clip = tile / n_tiles_per_clip, p0 = (tile - clip * n_tiles_per_clip) * TB_PIX;
        // global element offset of tile row (pixel p, frame f): ((clip * 16 + f) * hw + p0 + p) * 320
        const unsigned row0 = tile_row0(tile);                                          // (pixel 0, frame 0)
        // ================= phase A: LayerNorm (+ pe) of the h rows in X, in place =================
        // (the rows were requested under the previous tile's epilogue; the merge's first two weight sub-tiles go out behind the norm's constants)
        __builtin_amdgcn_sched_barrier(0);
        if (HAS_MERGE) {
            issue_w(rsWM, 0, 0);
            issue_w(rsWM, 1, 1);
        }
        if (HAS_MERGE && w_wave) TB_VMCNT(8); else TB_VMCNT(0);   // the h rows have landed; the two weight sub-tiles may still be in flight
        __syncthreads();
        TB_STAMP(1);
        {
            // statistics: 4 lanes per row, the row's 40 chunks in registers, centred variance
#pragma unroll 1
            for (int r = tid >> 2; r < TB_ROWS; r += 128) {
                const int q = tid & 3;
                const bf16_t* xr = X + r * TB_C;
                u32x4 x4[10];
                float s1 = 0.f;
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    x4[i] = *reinterpret_cast<const u32x4*>(xr + (q + 4 * i) * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s1 += __uint_as_float(x4[i][j] << 16) + __uint_as_float(x4[i][j] & 0xffff0000u);
                }
                s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
                const float mean = s1 * (1.f / 320.f);
                float s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 10; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = __uint_as_float(x4[i][j] << 16) - mean, b = __uint_as_float(x4[i][j] & 0xffff0000u) - mean;
                        s2 += a * a + b * b;
                    }
                s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
                if (q == 0) *reinterpret_cast<f32x2_t*>(stats + 2 * r) = f32x2_t{mean, rsqrtf(s2 * (1.f / 320.f) + P.ln_eps)};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // normalising thread = (logical chunk c, row group rg): rows rg, rg + 12, ...; frame of row r = r & 15 -> only the four frames (rg + 12 j) & 15
        // occur.  gamma of its 8 columns stays in registers; the (beta + pe) row of frame class j is fetched (L1 / L2 resident) while class
        // j - 1 is normalised -- all four at once were spilled by the register allocator the moment they arrived (two scratch round trips per row)
        int tid_t = tid;
        asm volatile("" : "+v"(tid_t));
        const int nc = tid_t % 40, nrg = tid_t / 40;
        f32x4 ng0 = f32x4{0.f, 0.f, 0.f, 0.f}, ng1 = ng0, cb0 = ng0, cb1 = ng0;
        if (tid < 480) {
            ng0 = *reinterpret_cast<const f32x4*>(P.ln_gamma + nc * 8);
            ng1 = *reinterpret_cast<const f32x4*>(P.ln_gamma + nc * 8 + 4);
            const float* bp = P.ln_bpe + (nrg & 15) * 320 + nc * 8;
            cb0 = *reinterpret_cast<const f32x4*>(bp);
            cb1 = *reinterpret_cast<const f32x4*>(bp + 4);
        }
        __syncthreads();
        if (tid < 480) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 nx0 = cb0, nx1 = cb1;
                if (j < 3) {
                    const float* bp = P.ln_bpe + ((nrg + 12 * (j + 1)) & 15) * 320 + nc * 8;
                    nx0 = *reinterpret_cast<const f32x4*>(bp);
                    nx1 = *reinterpret_cast<const f32x4*>(bp + 4);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int r = nrg + 12 * (4 * kk + j);
                    if (r < TB_ROWS) {
                        u32x4* px = reinterpret_cast<u32x4*>(X + r * TB_C + (nc ^ ((r >> 1) & 7)) * 8);
                        const u32x4 x4 = *px;
                        const f32x2_t st = *reinterpret_cast<const f32x2_t*>(stats + 2 * r);
                        const float m = st[0], rs = st[1];
                        u32x4 o4;
                        o4[0] = pack_bf2((__uint_as_float(x4[0] << 16) - m) * rs * ng0[0] + cb0[0], (__uint_as_float(x4[0] & 0xffff0000u) - m) * rs * ng0[1] + cb0[1]);
                        o4[1] = pack_bf2((__uint_as_float(x4[1] << 16) - m) * rs * ng0[2] + cb0[2], (__uint_as_float(x4[1] & 0xffff0000u) - m) * rs * ng0[3] + cb0[3]);
                        o4[2] = pack_bf2((__uint_as_float(x4[2] << 16) - m) * rs * ng1[0] + cb1[0], (__uint_as_float(x4[2] & 0xffff0000u) - m) * rs * ng1[1] + cb1[1]);
                        o4[3] = pack_bf2((__uint_as_float(x4[3] << 16) - m) * rs * ng1[2] + cb1[2], (__uint_as_float(x4[3] & 0xffff0000u) - m) * rs * ng1[3] + cb1[3]);
                        *px = o4;
                    }
                }
                cb0 = nx0; cb1 = nx1;
            }
        }
        __syncthreads();                                          // X = x = LayerNorm(h) + pe
        TB_STAMP(2);

        // ================= phase B: m = s x W_m^T + pose_term + x (in place) =================
        if (HAS_MERGE) {
            TB_VMCNT(4);                                          // sub-tile 0 (requested at the head of the tile)
            __builtin_amdgcn_s_barrier();
            if (wr == 1) __builtin_amdgcn_s_barrier();            // wave row 1 runs one barrier behind wave row 0
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 10; ++g) {
                read_frags(g, g % 3);
                if (g + 2 < 10) issue_w(rsWM, g + 2, (g + 2) % 3);
                // the ring buffers the weight stream no longer needs take the first pose-term passes (same rule as the stream: the buffer read at LOAD(g - 1))
                if (g == 8) issue_pass(rsPT, tile, 0, 1);
                if (g == 9) issue_pass(rsPT, tile, 1, 2);
                if (g < 8) TB_VMCNT(4);                           // retires sub-tile g + 1, the one just requested stays in flight
                else if (g == 8) TB_VMCNT2(3, 2);                 // sub-tile 9; my pieces of pass 0 stay in flight
                TB_MMA();
            }
            if (wr == 0) __builtin_amdgcn_s_barrier();            // the wave rows meet again: every fragment read of x is done, the ring is free
            __builtin_amdgcn_sched_barrier(0);
            issue_pass(rsPT, tile, 2, 0);
            // m = s acc + pose term + x, in place, accumulator block row by block row (pass p = block mb = p of both wave rows).  The pose-term rows
            // arrive in the ring by LDS-DMA (whole 640-byte rows) three passes ahead and are picked up in accumulator layout: as 25 eight-byte
            // global loads per lane they cost the texture path 16 quarter-lines per instruction (3 us of the epilogue), requested under the main
            // loop 50 live registers that spill; as two 80-row passes behind the loop their latency was exposed twice (epilogue 9 us).
#pragma unroll
            for (int pss = 0; pss < 5; ++pss) {
                // queue (oldest first): pass pss, pss + 1, pss + 2 (where they exist)
                if (pss == 0) TB_VMCNT2(6, 4);
                else if (pss < 4) TB_VMCNT2(3, 2);
                else TB_VMCNT(0);
                __syncthreads();                                  // pass pss published; everybody is done with pass pss - 1
                if (pss >= 1 && pss + 2 < 5) issue_pass(rsPT, tile, pss + 2, (pss + 3) % 3);      // into the buffer pass pss - 1 has just left
                {
                    const bf16_t* Pt = RING + ((pss + 1) % 3) * TB_SUB + (wr * 16 + l15) * TB_C;
                    bf16_t* Xr = X + (wr * 80 + pss * 16 + l15) * TB_C;
#pragma unroll
                    for (int nb = 0; nb < 5; ++nb) {
                        const int col = ecol + nb * 16, off = (((col >> 3) ^ xsw) << 3) + (col & 7);
                        u32x2* px = reinterpret_cast<u32x2*>(Xr + off);
                        float xv[4], pv[4];
                        unpack4(*px, xv);
                        unpack4(*reinterpret_cast<const u32x2*>(Pt + off), pv);
                        *px = u32x2{pack_bf2(acc[pss][nb][0] * P.merge_scale + pv[0] + xv[0], acc[pss][nb][1] * P.merge_scale + pv[1] + xv[1]),
                                    pack_bf2(acc[pss][nb][2] * P.merge_scale + pv[2] + xv[2], acc[pss][nb][3] * P.merge_scale + pv[3] + xv[3])};
                    }
                }
            }
            __syncthreads();                                      // X = m
        }

        TB_STAMP(3);
        // ================= phase D: wave = head.  q | k | v projections + attention, all in registers =================
        // the out-projection's first two weight sub-tiles stream into the ring meanwhile
        issue_w(rsWO, 0, 0);
        issue_w(rsWO, 1, 1);
        u32x2 o_pk[10][3];                                        // o of my head: (row 16 m + l15, channels 16 nb + 4 kq ..) as 4 bf16
        {
            // my head's 80-KiB stream through its own descriptor: the step offsets below are then compile-time constants (soffset literals),
            // not 80 precomputed scalar registers
            constexpr int QKV_STREAM = XATT ? 10 * 3 * 512 : TB_QKV_HEAD;      // XATT: the head's to_q only ([q0 | q1 | (q tail, zeros)] per k-step)
            const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(P.w_qkv + (size_t)wave * QKV_STREAM), 0, QKV_STREAM * 2, 0x00020000);
            int qkv_lane = lane * 16;
            asm volatile("" : "+v"(qkv_lane));
            // weight fragment stream of the head: step s = 0 .. 29 -> (part = s / 10, k-step = s % 10); bytes before step s:
            //   part 0 (q0 q1 tail): 3 KiB per step; part 1 (k0 k1): 2 KiB; part 2 (v0 v1 v2): 3 KiB
            auto step_off = [](int s) { return s < 10 ? s * 3072 : (s < 20 ? 30720 + (s - 10) * 2048 : 51200 + (s - 20) * 3072); };
            u32x4 wq[3][3];                                       // [stage][block]
            auto load_step = [&](int s) {
                const int nbk = (s >= 10 && s < 20) ? 2 : 3;
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    if (b < nbk) wq[s % 3][b] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, qkv_lane, step_off(s) + b * 1024, 0);
            };
            load_step(0);
            load_step(1);
            f32x4 pacc[10][3];
            bf16x8 q8[10];
            u32x2 t4[10];                                         // tail block: lanes kq < 2: q channels 32 + 4 kq ..; kq >= 2: k channels 32 + 4 (kq - 2) ..
            u32x2 p_pk[10];                                       // softmax probabilities P^T (keys 4 kq .. of query l15), bf16
            // (every step is instantiated with compile-time (part, k-step): the 30-step loop nest is beyond the unroller's size limit, and left
            // rolled it indexes the fragment ring dynamically -> scratch)
            auto zero_pacc = [&]() {
#pragma unroll
                for (int m = 0; m < 10; ++m)
#pragma unroll
                    for (int b = 0; b < 3; ++b) pacc[m][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            };
            auto step = [&](auto part_c, auto ks_c) {
                constexpr int part = decltype(part_c)::value, ks = decltype(ks_c)::value, s = part * 10 + ks;
                if constexpr (s + 2 < (XATT ? 10 : 30)) load_step(s + 2);
                int kqx = kq ^ xsw, xrow_o = l15 * TB_C;
                asm volatile("" : "+v"(kqx), "+v"(xrow_o));        // (see read_frags: no address of a later step may be computed ahead and spilled)
                const int xo = xrow_o + ((ks >> 1) * 8 + (((ks & 1) * 4) ^ kqx)) * 8;
                __builtin_amdgcn_sched_barrier(0);                // (keeps later k-steps' fragment reads from being hoisted over this one: registers)
                // the pixel blocks' A fragments through a 3-register-set pipeline: the read of block m + 2 is issued in front of the MFMAs of block m
                // (left to itself hipcc reads one fragment, waits lgkmcnt(0), issues its 3 MFMAs, reads the next: the LDS latency of every block exposed)
                constexpr int NBK = part == 1 ? 2 : 3;
                u32x4 af3[3];
                af3[0] = *reinterpret_cast<const u32x4*>(X + 0 * 16 * TB_C + xo);
                af3[1] = *reinterpret_cast<const u32x4*>(X + 1 * 16 * TB_C + xo);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // (the two leading reads first)
#pragma unroll
                for (int m = 0; m < 10; ++m) {
                    if (m + 2 < 10) af3[(m + 2) % 3] = *reinterpret_cast<const u32x4*>(X + (m + 2) * 16 * TB_C + xo);
                    union { bf16x8 v; u32x4 u; } a;
                    a.u = af3[m % 3];
#pragma unroll
                    for (int b = 0; b < NBK; ++b) {
                        union { bf16x8 v; u32x4 u; } w;
                        w.u = wq[s % 3][b];
                        if constexpr (part < 2) pacc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, a.v, pacc[m][b], 0, 0, 0);   // (frame l15, 4 channels)
                        else pacc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, w.v, pacc[m][b], 0, 0, 0);                     // (channel l15, 4 frames)
                    }
                    // one LDS read, then this block's MFMAs: pins the interleave (mask 0x100 = DS read, 0x008 = MFMA)
                    if (m + 2 < 10) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NBK, 0);
                }
            };
#define TB_STEP(PART, KS) step(std::integral_constant<int, PART>{}, std::integral_constant<int, KS>{})
#define TB_PART(PART)                                                                                                    \
    TB_STEP(PART, 0); TB_STEP(PART, 1); TB_STEP(PART, 2); TB_STEP(PART, 3); TB_STEP(PART, 4);                            \
    TB_STEP(PART, 5); TB_STEP(PART, 6); TB_STEP(PART, 7); TB_STEP(PART, 8); TB_STEP(PART, 9)
            // ---- q (+ the shared tail block) ----
            zero_pacc();
            TB_PART(0);
            TB_SETTLE();
#pragma unroll
            for (int m = 0; m < 10; ++m) {
                union { bf16x8 v; u32x4 u; } t;
                t.u = u32x4{pack_bf2(pacc[m][0][0], pacc[m][0][1]), pack_bf2(pacc[m][0][2], pacc[m][0][3]),
                            pack_bf2(pacc[m][1][0], pacc[m][1][1]), pack_bf2(pacc[m][1][2], pacc[m][1][3])};
                q8[m] = t.v;
                t4[m] = u32x2{pack_bf2(pacc[m][2][0], pacc[m][2][1]), pack_bf2(pacc[m][2][2], pacc[m][2][3])};
            }
            if constexpr (XATT) {
                // ---- scores against the text keys (5 blocks of 16, keys >= n_keys masked), softmax and o = P V block by block; K and V^T fragments from the packed copy ----
                const int batch = (int)(((int64_t)tile * TB_ROWS / P.hw) / P.images_per_text);
                const bf16_t* KF = P.kvfrag + ((size_t)batch * 8 + wave) * 7680;
                bf16x8 ka[5];
                s16x4 kt[5], vf[3][5];
#pragma unroll
                for (int kb = 0; kb < 5; ++kb) {
                    ka[kb] = *reinterpret_cast<const bf16x8*>(KF + kb * 768 + lane * 8);
                    kt[kb] = *reinterpret_cast<const s16x4*>(KF + kb * 768 + 512 + lane * 4);
                }
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int kb = 0; kb < 5; ++kb) vf[b][kb] = *reinterpret_cast<const s16x4*>(KF + 3840 + ((b * 5 + kb) * 64 + lane) * 4);
#pragma unroll
                for (int m = 0; m < 10; ++m) {
                    union { u32x2 u; s16x4 s; } qt;
                    const unsigned keep = kq < 2 ? 0xffffffffu : 0u;
                    qt.u = u32x2{t4[m][0] & keep, t4[m][1] & keep};
                    f32x4 sc[5];
                    // (an accumulator is touched again only four MFMAs later, the order pinned: see temporal_block640.hip)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kb = 0; kb < 5; ++kb) sc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[kb], q8[m], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kb = 0; kb < 5; ++kb) sc[kb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt[kb], qt.s, sc[kb], 0, 0, 0);
                    TB_SETTLE();
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
                    f32x4 o[3];
#pragma unroll
                    for (int b = 0; b < 3; ++b) o[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 5; ++kb) {
                        union { u32x2 u; s16x4 s; } pb;
                        pb.u = u32x2{pack_bf2(sc[kb][0] * inv, sc[kb][1] * inv), pack_bf2(sc[kb][2] * inv, sc[kb][3] * inv)};
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int b = 0; b < 3; ++b) o[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[b][kb], pb.s, o[b], 0, 0, 0);
                        if (kb & 1) asm volatile("s_nop 7" ::: "memory");   // (three accumulators take turns: spacing for the dependent ones)
                    }
                    TB_SETTLE();
#pragma unroll
                    for (int b = 0; b < 3; ++b) o_pk[m][b] = u32x2{pack_bf2(o[b][0], o[b][1]), pack_bf2(o[b][2], o[b][3])};
                }
            } else {
            // ---- k, scores, softmax ----
            zero_pacc();
            TB_PART(1);
            TB_SETTLE();
#pragma unroll
            for (int m = 0; m < 10; ++m) {
                union { bf16x8 v; u32x4 u; } k8;
                k8.u = u32x4{pack_bf2(pacc[m][0][0], pacc[m][0][1]), pack_bf2(pacc[m][0][2], pacc[m][0][3]),
                             pack_bf2(pacc[m][1][0], pacc[m][1][1]), pack_bf2(pacc[m][1][2], pacc[m][1][3])};
                // S^T[key][query] = K Q^T: A = k rows, B = q rows (both index the reduction by the same channel permutation)
                const f32x4 sc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k8.v, q8[m], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                // tails: q channels 32..39 sit in lanes kq < 2 of t4, k channels 32..39 in lanes kq >= 2 -> bring those down by a half-wave swap
                union { u32x2 u; s16x4 s; } qt, kt;
                const unsigned keep = kq < 2 ? 0xffffffffu : 0u;
                qt.u = u32x2{t4[m][0] & keep, t4[m][1] & keep};
                const auto s0 = __builtin_amdgcn_permlane32_swap(t4[m][0], t4[m][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(t4[m][1], t4[m][1], false, false);
                kt.u = u32x2{(unsigned)s0[1] & keep, (unsigned)s1[1] & keep};       // [1]: lanes 0-31 see the upper half's value of lane + 32
                const f32x4 sc2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt.s, qt.s, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                TB_SETTLE();
                const f32x4 sc = sc1 + sc2;
                // softmax over the 16 keys of query l15: 4 in-lane values x 4 lanes (kq)
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
            TB_PART(2);
            TB_SETTLE();
#pragma unroll
            for (int m = 0; m < 10; ++m) {
                union { u32x2 u; s16x4 s; } pb;
                pb.u = p_pk[m];
                f32x4 o[3];
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    union { u32x2 u; s16x4 s; } vt;                  // V^T: (channel 16 b + l15, keys 4 kq ..)
                    vt.u = u32x2{pack_bf2(pacc[m][b][0], pacc[m][b][1]), pack_bf2(pacc[m][b][2], pacc[m][b][3])};
                    o[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt.s, pb.s, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);   // (query l15, channels 16 b + 4 kq ..)
                }
                TB_SETTLE();
#pragma unroll
                for (int b = 0; b < 3; ++b) o_pk[m][b] = u32x2{pack_bf2(o[b][0], o[b][1]), pack_bf2(o[b][2], o[b][3])};
            }
            }  // (!XATT)
#undef TB_PART
#undef TB_STEP
        }
        TB_STAMP(4);
        __syncthreads();                                          // every head is done with m: o may overwrite it
        TB_STAMP(5);
        {
            int orow = l15 * TB_C, okq = kq, oxs = xsw;
            asm volatile("" : "+v"(orow), "+v"(okq), "+v"(oxs));   // (addresses computed here, not ahead of phase D: see read_frags)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int ch = 16 * b + 4 * okq;                 // channel inside the head
                const int col = 40 * wave + ch;
                const int off = orow + (((col >> 3) ^ oxs) << 3) + (col & 7);
                if (ch < 40) {
#pragma unroll
                    for (int m = 0; m < 10; ++m) *reinterpret_cast<u32x2*>(X + m * 16 * TB_C + off) = o_pk[m][b];
                }
            }
        }
        // ================= phase E: h' = o W_out^T + b + h =================
        TB_VMCNT(0);                                              // (sub-tiles 0, 1 of W_out have long landed)
        __syncthreads();                                          // X = o; ring buffers 0, 1 published
        if (wr == 1) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 10; ++g) {
            read_frags(g, g % 3);
            if (g + 2 < 10) issue_w(rsWO, g + 2, (g + 2) % 3);
            if (g < 8) TB_VMCNT(4); else TB_VMCNT(0);
            TB_MMA();
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        TB_STAMP(6);
        issue_h(tile);                                            // X (= o) is dead: this tile's h rows come back for the residual add (see phase B)
        {
            // epilogue: wave row `pass` stages its 80 rows (bf16, pitch 328) in the ring region; whole-row stores
            constexpr int OP = 328, CPR = 40;
            bf16_t* Os = RING;
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                float b4[4] = {0.f, 0.f, 0.f, 0.f};
                if (P.b_out) unpack4(*reinterpret_cast<const u32x2*>(P.b_out + ecol + nb * 16), b4);
#pragma unroll
                for (int mb = 0; mb < 5; ++mb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mb][nb][j] += b4[j];
            }
            TB_VMCNT(0);
            __syncthreads();
#pragma unroll
            for (int mb = 0; mb < 5; ++mb) {
                const int r = wr * 80 + mb * 16 + l15;
#pragma unroll
                for (int nb = 0; nb < 5; ++nb) {
                    const int col = ecol + nb * 16;
                    float hv[4];
                    unpack4(*reinterpret_cast<const u32x2*>(X + r * TB_C + (((col >> 3) ^ xsw) << 3) + (col & 7)), hv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mb][nb][j] += hv[j];
                }
            }
            __syncthreads();
            // the NEXT tile's h rows stream into X under the staging passes and stores below
            __builtin_amdgcn_sched_barrier(0);
            if (tile + (int)gridDim.x < P.tiles) issue_h(tile + gridDim.x);
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                if (wr == pass) {
#pragma unroll
                    for (int mb = 0; mb < 5; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 5; ++nb)
                            *reinterpret_cast<u32x2*>(Os + (mb * 16 + l15) * OP + ecol + nb * 16) =
                                u32x2{pack_bf2(acc[mb][nb][0], acc[mb][nb][1]), pack_bf2(acc[mb][nb][2], acc[mb][nb][3])};
                }
                __syncthreads();
                // staged row rr (0..79) = tile row 80 pass + rr = (pixel 5 pass + rr / 16, frame rr % 16)
                for (int c = tid; c < 80 * CPR; c += 512) {
                    const int rr = c / CPR, ch = c - rr * CPR;
                    const unsigned dst = row0 + (unsigned)(rr & 15) * fstride + (unsigned)(5 * pass + (rr >> 4)) * pstride + (unsigned)ch * 8;
                    *reinterpret_cast<u32x4*>(P.out + dst) = *reinterpret_cast<const u32x4*>(Os + rr * OP + ch * 8);
                }
                if (STATS && tid < 320) {
                    const int rr = tid >> 2, q = tid & 3;
                    float s1 = 0.f;
                    u32x4 x4[10];
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        x4[i] = *reinterpret_cast<const u32x4*>(Os + rr * OP + (q + 4 * i) * 8);
#pragma unroll
                        for (int j = 0; j < 4; ++j) s1 += __uint_as_float(x4[i][j] << 16) + __uint_as_float(x4[i][j] & 0xffff0000u);
                    }
                    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
                    const float mean = s1 * (1.f / 320.f);
                    float s2 = 0.f;
#pragma unroll
                    for (int i = 0; i < 10; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a = __uint_as_float(x4[i][j] << 16) - mean, b = __uint_as_float(x4[i][j] & 0xffff0000u) - mean;
                            s2 += a * a + b * b;
                        }
                    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
                    if (q == 0) {
                        const int64_t grow = XATT ? (int64_t)tile * TB_ROWS + 80 * pass + rr : ((int64_t)clip * TB_F + (rr & 15)) * P.hw + p0 + 5 * pass + (rr >> 4);
                        *reinterpret_cast<f32x2_t*>(P.ln_stats + grow * 2) = f32x2_t{mean, rsqrtf(s2 * (1.f / 320.f) + P.ln_stats_eps)};
                    }
                }
                __syncthreads();
            }
        }
        TB_STAMP(7);
    }
#undef TB_MMA
}

}  // namespace

// text k | v `[batch][S][2 C]` (C = 320, 8 heads x 40) -> the MFMA fragments phase D of the XATT block reads: per (batch, head) 7680 bf16:
//   K: 5 key blocks x [a: lane x 8 = channels {4 kq .. + 3, 16 + 4 kq .. + 3} | tail: lane x 4 = channels 32 + 4 kq .. + 3 for kq < 2, zeros beyond];
//   V^T: [channel block 3][key block 5][lane][4]: lane (l15 = channel 16 cb + l15 < 40, kq) holds keys 16 kb + 4 kq .. + 3.   Keys >= S are zero.
__global__ __launch_bounds__(256) void xattn_pack_kv40_kernel(const bf16_t* __restrict__ kv, bf16_t* __restrict__ out, int S, int64_t ldb) {
    const int bh = blockIdx.x, b = bh >> 3, h = bh & 7;
    const bf16_t* kb_ = kv + (int64_t)b * ldb + h * 40;            // k of (b, h): row stride 640
    const bf16_t* vb_ = kb_ + 320;
    bf16_t* o = out + (size_t)bh * 7680;
    for (int g = threadIdx.x; g < 1920; g += 256) {               // groups of 4 output elements
        u32x2 val = u32x2{0u, 0u};
        if (g < 960) {                                            // K: key block kb = g / 192; inside: a (128 groups), tail (64)
            const int kb = g / 192, r = g - kb * 192;
            int lane, ch;
            if (r < 128) { lane = r >> 1; ch = 16 * (r & 1) + 4 * (lane >> 4); }
            else { lane = r - 128; ch = (lane >> 4) < 2 ? 32 + 4 * (lane >> 4) : -1; }
            const int key = 16 * kb + (lane & 15);
            if (key < S && ch >= 0) val = *reinterpret_cast<const u32x2*>(kb_ + (int64_t)key * 640 + ch);
        } else {                                                  // V^T: (cb, kb, lane)
            const int q = g - 960, lane = q & 63, kb = (q >> 6) % 5, cb = (q >> 6) / 5;
            const int ch = 16 * cb + (lane & 15), key0 = 16 * kb + 4 * (lane >> 4);
            unsigned short e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = (key0 + j < S && ch < 40) ? vb_[(int64_t)(key0 + j) * 640 + ch] : (unsigned short)0;
            val = u32x2{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16)};
        }
        *reinterpret_cast<u32x2*>(o + g * 4) = val;
    }
}

extern "C" int fmc_xattn_pack_kv40(const void* kv, void* out, int batch, int S, int64_t ld_batch, void* stream) {
    if (!kv || !out) FMC_FAIL(FMC_E_NULL, "xattn_pack_kv40: NULL tensor");
    if (batch <= 0 || S <= 0 || S > 80 || ld_batch < (int64_t)S * 640) FMC_FAIL(FMC_E_SHAPE, "xattn_pack_kv40: 1 <= S <= 80 text tokens of 2 x 320 channels");
    if (((uintptr_t)kv & 7) || !fmc_aligned16(out)) FMC_FAIL(FMC_E_ALIGN, "xattn_pack_kv40: alignment");
    hipLaunchKernelGGL(xattn_pack_kv40_kernel, dim3((unsigned)(batch * 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)kv, (bf16_t*)out, S, ld_batch);
    FMC_CHECK_LAUNCH("fmc_xattn_pack_kv40");
    return 0;
}

// The text cross-attention block of the spatial transformer at the 40x64 level in one launch (see fmc_xattn_block640_bf16): tokens h [images][hw][320],
// hw % 160 == 0, 8 heads x 40; w_q_packed = hip_ops.pack_xattn_q40, w_out_tm = hip_ops._w_tilemajor, kvfrag = fmc_xattn_pack_kv40; optional row statistics.
extern "C" int fmc_xattn_block320_bf16(const void* h, void* out, const float* ln_gamma, const float* ln_bpe, float ln_eps, const void* w_q_packed,
                                       const void* kvfrag, const void* w_out_tm, const void* b_out, float* ln_stats, float ln_stats_eps, int n_images, int hw,
                                       int n_keys, int images_per_text, float scale, void* stream) {
    if (!h || !out || !ln_gamma || !ln_bpe || !w_q_packed || !kvfrag || !w_out_tm) FMC_FAIL(FMC_E_NULL, "xattn_block320_bf16: NULL tensor");
    if (n_images <= 0 || hw <= 0 || hw % 160 || n_keys <= 0 || n_keys > 80 || images_per_text <= 0 || n_images % images_per_text)
        FMC_FAIL(FMC_E_SHAPE, "xattn_block320_bf16: hw %% 160 == 0, 1 <= keys <= 80, images %% images_per_text == 0 (got hw=%d keys=%d images=%d / %d)", hw, n_keys,
                 n_images, images_per_text);
    if ((int64_t)n_images * hw * 320 * 2 >= ((int64_t)1 << 31)) FMC_FAIL(FMC_E_SHAPE, "xattn_block320_bf16: tensor of 2 GiB or more");
    if (!fmc_aligned16(h) || !fmc_aligned16(out) || !fmc_aligned16(w_q_packed) || !fmc_aligned16(kvfrag) || !fmc_aligned16(w_out_tm) || !fmc_aligned16(ln_gamma) ||
        !fmc_aligned16(ln_bpe) || (b_out && !fmc_aligned16(b_out)) || (ln_stats && ((uintptr_t)ln_stats & 7)))
        FMC_FAIL(FMC_E_ALIGN, "xattn_block320_bf16: tensors must be 16-byte aligned");
    TBParams P{};
    P.h = (const bf16_t*)h; P.out = (bf16_t*)out; P.ln_gamma = ln_gamma; P.ln_bpe = ln_bpe; P.ln_eps = ln_eps;
    P.w_qkv = (const bf16_t*)w_q_packed; P.w_out = (const bf16_t*)w_out_tm; P.b_out = (const bf16_t*)b_out;
    P.ln_stats = ln_stats; P.ln_stats_eps = ln_stats_eps;
    P.kvfrag = (const bf16_t*)kvfrag; P.n_keys = n_keys; P.images_per_text = images_per_text;
    P.n_clips = 1; P.hw = hw; P.total_rows = (int64_t)n_images * hw;
    P.tiles = (int)(P.total_rows / TB_ROWS);
    P.scale_log2 = scale * 1.4426950408889634f;
    P.dbg_times = nullptr;
    const int cus = fmc_cu_count();
    const unsigned grid = (unsigned)(P.tiles < cus ? P.tiles : cus);
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block_kernel<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS);
        raised = true;
    }
    if (ln_stats) hipLaunchKernelGGL((temporal_block_kernel<false, true, true>), dim3(grid), dim3(512), TB_LDS, (hipStream_t)stream, P);
    else hipLaunchKernelGGL((temporal_block_kernel<false, false, true>), dim3(grid), dim3(512), TB_LDS, (hipStream_t)stream, P);
    FMC_CHECK_LAUNCH("fmc_xattn_block320_bf16");
    return 0;
}

static long long* g_tb_dbg = nullptr;
// diagnostic: device buffer of [workgroups][4][8] int64 that receives s_memrealtime stamps (100 MHz) of wave 0 at the phase boundaries; NULL = off
extern "C" int fmc_temporal_block_set_debug(void* buf) {
    g_tb_dbg = (long long*)buf;
    return 0;
}

int fmc_temporal_block640_launch(const void* h, void* out, const float* ln_gamma, const float* ln_bpe, float ln_eps, const void* w_merge_frag,
                                 const void* pose_term, float merge_scale, const void* w_qkv_packed, const void* w_out_frag, const void* b_out,
                                 int n_clips, int hw, float scale, hipStream_t st);      // temporal_block640.hip

extern "C" int fmc_temporal_block_bf16(const void* h, void* out, const float* ln_gamma, const float* ln_bpe, float ln_eps, const void* w_merge_tm,
                                       const void* pose_term, float merge_scale, const void* w_qkv_packed, const void* w_out_tm, const void* b_out,
                                       float* ln_stats, float ln_stats_eps, int n_clips, int frames, int hw, int channels, int heads, float scale,
                                       void* stream) {
    if (!h || !out || !ln_gamma || !ln_bpe || !w_qkv_packed || !w_out_tm) FMC_FAIL(FMC_E_NULL, "temporal_block_bf16: NULL tensor");
    if (channels == 640 && frames == 16 && heads == 8 && hw > 0 && n_clips > 0) {     // the 20x32 level: its own kernel and weight formats (see fmc_hip.h)
        if ((w_merge_tm != nullptr) != (pose_term != nullptr)) FMC_FAIL(FMC_E_NULL, "temporal_block_bf16: w_merge and pose_term come together");
        if (ln_stats) FMC_FAIL(FMC_E_SHAPE, "temporal_block_bf16: no row statistics at C = 640");
        if ((int64_t)n_clips * frames * hw * channels * 2 >= ((int64_t)1 << 31)) FMC_FAIL(FMC_E_SHAPE, "temporal_block_bf16: tensor of 2 GiB or more");
        if (!fmc_aligned16(h) || !fmc_aligned16(out) || !fmc_aligned16(w_qkv_packed) || !fmc_aligned16(w_out_tm) || !fmc_aligned16(ln_gamma) || !fmc_aligned16(ln_bpe) ||
            (w_merge_tm && (!fmc_aligned16(w_merge_tm) || !fmc_aligned16(pose_term))) || (b_out && !fmc_aligned16(b_out)))
            FMC_FAIL(FMC_E_ALIGN, "temporal_block_bf16: tensors must be 16-byte aligned");
        return fmc_temporal_block640_launch(h, out, ln_gamma, ln_bpe, ln_eps, w_merge_tm, pose_term, merge_scale, w_qkv_packed, w_out_tm, b_out, n_clips, hw, scale,
                                            (hipStream_t)stream);
    }
    if (frames != 16 || channels != 320 || heads != 8 || hw <= 0 || hw % 10 || n_clips <= 0)
        FMC_FAIL(FMC_E_SHAPE, "temporal_block_bf16: the fused block exists for F = 16, C = 320, 8 heads, pixels %% 10 == 0 (got F=%d C=%d H=%d hw=%d)", frames,
                 channels, heads, hw);
    if ((w_merge_tm != nullptr) != (pose_term != nullptr)) FMC_FAIL(FMC_E_NULL, "temporal_block_bf16: w_merge and pose_term come together");
    if ((int64_t)n_clips * frames * hw * channels * 2 >= ((int64_t)1 << 31)) FMC_FAIL(FMC_E_SHAPE, "temporal_block_bf16: tensor of 2 GiB or more");
    if (!fmc_aligned16(h) || !fmc_aligned16(out) || !fmc_aligned16(w_qkv_packed) || !fmc_aligned16(w_out_tm) || !fmc_aligned16(ln_gamma) || !fmc_aligned16(ln_bpe) ||
        (w_merge_tm && (!fmc_aligned16(w_merge_tm) || !fmc_aligned16(pose_term))) || (b_out && !fmc_aligned16(b_out)) || (ln_stats && ((uintptr_t)ln_stats & 7)))
        FMC_FAIL(FMC_E_ALIGN, "temporal_block_bf16: tensors must be 16-byte aligned");
    TBParams P{};
    P.h = (const bf16_t*)h; P.out = (bf16_t*)out; P.ln_gamma = ln_gamma; P.ln_bpe = ln_bpe; P.ln_eps = ln_eps;
    P.w_merge = (const bf16_t*)w_merge_tm; P.pose_term = (const bf16_t*)pose_term; P.merge_scale = merge_scale;
    P.w_qkv = (const bf16_t*)w_qkv_packed; P.w_out = (const bf16_t*)w_out_tm; P.b_out = (const bf16_t*)b_out;
    P.ln_stats = ln_stats; P.ln_stats_eps = ln_stats_eps;
    P.n_clips = n_clips; P.hw = hw; P.tiles = n_clips * (hw / 10);
    P.total_rows = (int64_t)n_clips * frames * hw;
    P.scale_log2 = scale * 1.4426950408889634f;
    P.dbg_times = g_tb_dbg;
    const int cus = fmc_cu_count();
    const unsigned grid = (unsigned)(P.tiles < cus ? P.tiles : cus);
    hipStream_t st = (hipStream_t)stream;
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS);
        raised = true;
    }
    if (w_merge_tm) {
        if (ln_stats) hipLaunchKernelGGL((temporal_block_kernel<true, true>), dim3(grid), dim3(512), TB_LDS, st, P);
        else hipLaunchKernelGGL((temporal_block_kernel<true, false>), dim3(grid), dim3(512), TB_LDS, st, P);
    } else {
        if (ln_stats) hipLaunchKernelGGL((temporal_block_kernel<false, true>), dim3(grid), dim3(512), TB_LDS, st, P);
        else hipLaunchKernelGGL((temporal_block_kernel<false, false>), dim3(grid), dim3(512), TB_LDS, st, P);
    }
    FMC_CHECK_LAUNCH("fmc_temporal_block_bf16");
    return 0;
}
