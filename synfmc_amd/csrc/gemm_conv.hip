// bf16 MFMA GEMM for gfx950 with two A-operand loaders (token matrix / implicit 3x3 convolution on NHWC) and
// fused epilogues.  Replaces, on the FMC path,
//   * nn.Linear of diffusers' Attention / FeedForward / Transformer2D proj_in|out (call sites
//     fmc/models/attention_processor.py:50-69,255-283; fmc/models/motion_module.py:219,228,284) including the
//     `+ residual` that follows them (motion_module.py:289-297, diffusers BasicTransformerBlock), the Camera-Adapter
//     axpy `qkv_merge(h + pose) * scale + h` (attention_processor.py:257) and the GEGLU gate;
//   * the 3x3 convolutions of diffusers' ResnetBlock2D / Upsample2D (ctor args fmc/models/unet_blocks.py:306-317,
//     :625) with `+ time_emb_proj(silu(temb))[:, :, None, None]` and the `input + h` residual fused into the epilogue.
//
// out[m, n] = epi( sum_k A[m, k] * W[n, k] )      A: [M, K] bf16 (K contiguous), W: [N, K] bf16 (K contiguous)
//   conv mode: m = (img, y, x) pixel of an NHWC image, k = (tap, ci): A[m, k] = X[img, y+dy-1, x+dx-1, ci] (zero
//   outside), W = the filter in channels-last memory format [Cout][3][3][Cin] -- exactly K-contiguous.
//
// Structure (CDNA4, wave = 64):
//   * a workgroup of WM x WN waves computes a (64*WM) x (64*WN) output tile, every wave a 64x64 sub-tile as 2x2
//     v_mfma_f32_32x32x16_bf16 accumulators; three geometries are built: 128x128 (4 waves, 2 workgroups per CU),
//     256x128 (8 waves) and 256x256 (16 waves, one workgroup per CU).  The first version (128x128 only) ran at
//     ~64 flop per byte moved L2->LDS and measured 10.6 TB/s of that traffic = 680 TFLOP/s on every shape: it was
//     L2-bandwidth bound, so the larger tiles (85 / 128 flop per byte) are the lever, not the inner loop;
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction): no staging
//     registers, no ds_write pass; the 16-byte chunks of a 128-byte tile row are XOR-swizzled through the SOURCE
//     address so that the ds_read_b128 fragment reads are bank-conflict free; out-of-range rows / padding taps read
//     a 16-byte zero page; two LDS stages, the DMA of k-tile t+1 runs under the MFMAs of k-tile t, one raw
//     s_barrier per k-tile (a __syncthreads() would drain the DMA);
//   * products are computed "swapped" (MFMA A operand = W rows) so a lane ends up with 4 consecutive output
//     columns; the C tile leaves through LDS in fp32 (64-row slabs) with full-row 16-byte stores, bias / alpha /
//     temb / residual / GEGLU applied on the way out;
//   * XCD-aware tile order: the N-tiles of one M-tile run back to back on one XCD (A rows come from that L2).
//
// Roofline: MFMA bound for K >= ~640, HBM bound (output write) for the K = 320 level-0 projections.
// Algorithmic flops per launch = 2*M*N*K.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace {

constexpr int BK_MAX = 64;   // K must be a multiple of this (every FMC projection / conv is)

struct GemmParams {
    const bf16_t* a; const bf16_t* w; const bf16_t* bias; const bf16_t* temb; const bf16_t* res; bf16_t* out;
    const bf16_t* res2;               // optional second residual (rows ldres apart, like res)
    int64_t M; int N, K;
    int64_t lda, ldres, ldo;
    int img_h, img_w, cin, hw;        // conv mode
    int64_t temb_ld; int temb_div;    // temb row of image i = temb + (i / temb_div) * temb_ld
    const bf16_t* a2; int64_t lda2; int ksplit;   // token mode: columns [ksplit, K) of A come from a2 (rows lda2 apart)
    int ups;                          // conv mode: 1 = x is [n, H/2, W/2, Cin], read through a nearest 2x upsample;
                                      //            2 = stride-2 convolution, x is [n, 2H, 2W, Cin]  (H, W = output size)
    float alpha;
    int tiles_m, tiles_n;
    int regepi;                       // A/B switch FMC_GEMM_REGEPI: 0 = everything through the fp32 slab, 2 = residual calls only, 1 = none
    int tap_outer;                    // conv mode A/B switch (FMC_CONV_TAP_OUTER=1): the old (tap, channel) k order
    int group_m;                      // tile order: m runs fastest inside groups of group_m m-tiles (see lin_to_tile)
    int split_k;                      // > 1: k-tiles are dealt to split_k workgroups per output tile, which write fp32
    float* ws;                        //      partial sums to ws[split][M][N]; splitk_reduce_kernel applies the epilogue
    int sk;                           // stream-K: grid = sk persistent workgroups (a multiple of 8), each owning a
    int* sk_flags;                    //   contiguous range of (tile, k-tile) iterations; ws[block][BM][BN] partials
    int64_t sk_ws_bytes;
    int sk_maxseg;                    // 8-phase stream-K: workspace slots per workgroup (segments it may own)
    int sk_whole;                     // 8-phase stream-K: 1 = the persistent pass (ROLE 3) finished the whole tiles itself
    int sk_t0;                        // 8-phase hybrid: tiles [0, sk_t0) (launch order) run on the plain grid, only the rest is stream-K'd
    int sk_hybrid;                    // split_k == -2 was asked for
    int sk_lock, sk_lock_len;         // split_k == -3, 8-phase kernel: K-LOCKSTEP split -- every tile's reduction is cut into sk_lock chunks of sk_lock_len
                                      //   k-tiles; unit u = chunk * T + tile (tiles n-outer / m-inner), workgroup b runs units (b & 7) * (sk / 8) + (b >> 3)
                                      //   + round * sk: the 32 CUs of an XCD work on the SAME k-chunk of (at most two) filter column blocks at the same
                                      //   time, so a weight byte leaves the Infinity Cache once per XCD and round instead of once per tile
    int64_t a_bytes;                  // 8-phase kernel: size of the A operand in bytes (buffer descriptor range)
    const float* q8_inv;              // EPI 2 (fp8 e4m3 output, three equal column blocks q | k | v): 1 / scale per block (device)
    unsigned* q8_amax;                // EPI 2: running max |value| per block as float bits (device, atomicMax), may be NULL
    float* gn_part;                   // gemm160 kernels, plain bf16 epilogue: per-(image, 160-row tile, group of N / 32 channels) partial
    int gn_hw;                        //   (sum, sum of squares) of the ROUNDED outputs -> gn_part[img][hw / 160][32][2]; gn_hw = pixels per image
    bf16_t* ln_out;                   // gemm160p_kernel, plain epilogue, N == 320: ALSO write LayerNorm(out rows) * gamma + beta (+ pe row) here
    const float* ln_gamma; const float* ln_beta; const float* ln_pe;   //   (the consumer's norm: the tile holds whole rows, x is not read again);
    float ln_eps; int ln_pe_inner, ln_pe_frames;                        //   pe row of output row m = ((m / ln_pe_inner) % ln_pe_frames), ln_pe_inner % 160 == 0
    int w_blocked;                    // gemm160 kernels: the W operand is PRE-PACKED tile-major in the kernel's own sub-tile order,
                                      //   [N / 320][K / 32][320 rows][32] (conv: sub-tile s = (64-channel chunk, tap, 32-channel half)): a W piece is a
                                      //   contiguous KiB (fmc_pack_weight_tilemajor) -- linear requests move 30-50 % more bytes per CU than 16 rows x 64 B
    int a_blocked;                    // gemm160p_kernel: the A operand is stored tile-major, [M / 160][K / 32][160 rows][32] -- the sub-tile a workgroup requests is
                                      //   ONE contiguous 10-KiB block (what the GEGLU epilogue writes with out_blocked for the feed-forward's second GEMM)
    int out_blocked;                  // gemm160p_kernel, GEGLU epilogue: write the gated output in that layout (K of the consumer = N / 2)
    float* ln_stats;                  // ... or only the rows' (mean, rstd) -> ln_stats[M][2], for a consumer GEMM that applies the LayerNorm itself:
    const float* lnc_stats; const float* lnc_c; const float* lnc_bias;  // gemm160p_kernel as that consumer (W pre-scaled by gamma): out = rstd[m] (acc -
                                      //   mean[m] c[n]) + lnc_bias[n], c[n] = sum_k W'[n, k], lnc_bias = W beta + bias (fp32 [N]); alpha 1, no bf16 bias
    const float* bias_img; int64_t w_img_stride; int img_rows;   // gemm160p_kernel<IMG = 1>: every img_rows rows of A (one image, a multiple of 160) have their
                                      //   OWN weight (w + image * w_img_stride elements, same layout) and fp32 bias row (bias_img + image * N): the
                                      //   GroupNorm in front of the layer folded into it per image (fmc_groupnorm_fold_linear, fmc_linear_bf16_imgw)
    int f32io;                        // fp32-storage ("parity") mode: A / W are split-bf16 x3 operands (fmc_split_bf16x3), bias / temb /
                                      // residual(s) / out are FP32 tensors (the bf16_t pointers above are reinterpreted), see epi_f32_*
};

// ---- fp32-storage mode: the epilogue straight from the accumulator registers, fp32 in, fp32 out ---------------------------
// The split-bf16 x3 operands ([hi | hi | lo] x [hi | lo | hi] along K, fp32 accumulate) give the fp32 product to ~2^-17; this
// epilogue keeps bias / alpha / temb / residual(s) / GEGLU in fp32 as well, so the SAME tile maps, loaders, fragment layouts
// and reduction orders the bf16 product path runs are checked against the fp32 CPU oracle at 1e-3 instead of 1e-2.
// A lane holds 4 consecutive output columns of one row per accumulator register quad: one f32x4 store per quad.
template <int MODE>
__device__ __forceinline__ void epi_f32_quad(const GemmParams& P, const float (&a)[4], int64_t m, int n) {
    if (m >= P.M || n >= P.N) return;
    const float* bias = reinterpret_cast<const float*>(P.bias);
    const float* temb = reinterpret_cast<const float*>(P.temb);
    const float* res = reinterpret_cast<const float*>(P.res);
    const float* res2 = reinterpret_cast<const float*>(P.res2);
    float* out = reinterpret_cast<float*>(P.out);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (a[j] + (bias ? bias[n + j] : 0.f)) * P.alpha;
    if (MODE == 1 && temb) v += *reinterpret_cast<const f32x4*>(temb + ((m / P.hw) / P.temb_div) * P.temb_ld + n);
    if (res) v += *reinterpret_cast<const f32x4*>(res + m * P.ldres + n);
    if (res2) v += *reinterpret_cast<const f32x4*>(res2 + m * P.ldres + n);
    *reinterpret_cast<f32x4*>(out + m * P.ldo + n) = v;
}
// GEGLU: `av` / `ag` = value and gate accumulators of the same 4 output columns; nb = weight row of the first value column
__device__ __forceinline__ void epi_f32_geglu_quad(const GemmParams& P, const float (&av)[4], const float (&ag)[4], int64_t m, int nb, int no) {
    if (m >= P.M || no >= P.N / 2) return;
    const float* bias = reinterpret_cast<const float*>(P.bias);
    float* out = reinterpret_cast<float*>(P.out);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float g = ag[j] + (bias ? bias[nb + 32 + j] : 0.f);
        // exact erf here (libdevice): the fast polynomial of gelu_erf is a bf16-grade approximation (1.5e-7 absolute is fine,
        // but parity mode should not depend on it)
        v[j] = (av[j] + (bias ? bias[nb + j] : 0.f)) * (0.5f * g * (1.f + erff(g * 0.70710678118654752f)));
    }
    *reinterpret_cast<f32x4*>(out + m * P.ldo + no) = v;
}

// Launch-order index -> output tile.  Inside a group of `group_m` m-tiles the order is n-outer / m-inner, so the C
// workgroups an XCD runs at any moment cover ~group_m m-tiles x C/group_m n-tiles: each k-step they pull group_m
// A sub-tiles and C/group_m W sub-tiles through that XCD's L2 instead of 1 and C (row-major order re-streamed the whole
// weight matrix from the Infinity Cache once per m-tile row as soon as it outgrew the 4 MB L2: PMC showed 9 TB/s of
// L2-miss traffic on the GEGLU projections, 12x their algorithmic bytes).
__device__ __forceinline__ void lin_to_tile(int lin, const GemmParams& P, int& tile_m, int& tile_n) {
    const int gsz = P.group_m * P.tiles_n;
    const int g = lin / gsz, r = lin - g * gsz, first = g * P.group_m;
    const int gm = min(P.group_m, P.tiles_m - first);
    tile_n = r / gm;
    tile_m = first + (r - tile_n * gm);
}

// exact-erf GELU for a bf16 result: erf by Abramowitz & Stegun 7.1.26 (|error| < 1.5e-7, far below bf16's 2^-9) with the
// hardware reciprocal and exp2 -- libdevice's erff costs ~3x as many VALU slots, and the level-0 GEGLU projection
// evaluates 105 M of them per launch
__device__ __forceinline__ float gelu_erf(float g) {
    if (!FMC_GELU_EXACT) return fmc_gelu_fast(g);         // common.h: the sigmoid-form fit (2.6e-5 absolute), half the VALU work
    const float x = fabsf(g) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.f - p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);   // erf(|g|/sqrt 2)
    return 0.5f * g + 0.5f * fabsf(g) * e;                                                 // g/2 (1 + sign(g) erf)
}

// physical 16-byte chunk of logical chunk c in tile row r: 8 chunks per 128-byte row (BK = 64) or 4 per 64-byte row
// (BK = 32); both make the ds_read_b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} bank-conflict free
template <int BK> __device__ __forceinline__ int swz(int r, int c) {
    return BK == 64 ? (c ^ ((r >> 1) & 7)) : (c ^ ((r >> 2) & 3));
}

// 16 zero bytes in global memory: the LDS-DMA source of out-of-range rows / padding taps
__device__ u32x4 g_zero16 = {0u, 0u, 0u, 0u};

// one 1-KiB LDS-DMA piece: lane i's 16 bytes land at lds_base + 16*i (global_load_lds_dwordx4)
__device__ __forceinline__ void dma16(const bf16_t* gsrc, bf16_t* lds_base_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_base_wave_uniform, 16, 0, 0);
}

// stream-K hand-off payload: 16-byte buffer accesses with sc0 sc1 (write-through / cache-bypassing, coherent across the
// 8 XCD L2s) -- no release / acquire fence, which on gfx950 means a whole-L2 write-back / invalidate per workgroup
// (measured: the fenced version ran 2-3x slower than the plain grid)
constexpr int SK_SC = 17;                                // cache-policy bits: sc0 | sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_rsrc(float* ws) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)ws, 0, 0x7fffffff, 0x00020000);
}

// MODE 0: token GEMM, 1: implicit 3x3 conv.  EPI 0: alpha*(acc+bias) (+temb)(+residual); 1: GEGLU (weights
// pre-interleaved per 64 rows: 32 value rows then their 32 gate rows; out is [M, N/2]).
//
// BK = 64: two 128-byte-row stages; BK = 32: half the LDS per workgroup, i.e. twice the workgroups per CU -- the arm
// for the K = 320 / 640 projections, whose 5-10 k-tiles cannot hide the DMA latency behind their own MFMAs and need
// other workgroups on the CU to do it.
//
// STAGES: depth of the LDS ring.  STAGES-1 k-tiles are in flight under the MFMAs of the current one (counted
// s_waitcnt vmcnt(N), never a drain): what bounds these kernels is bytes in flight per CU (L2/HBM latency x
// bandwidth), so the ring is as deep as the 160 KiB of LDS allow for the geometry.
template <int WM, int WN, int MI, int BK, int STAGES> constexpr int gemm_min_waves() {
    if (MI == 4) return (WM * WN + 3) / 4;              // 128 accumulator registers per wave: one workgroup per CU
    const size_t lds = (size_t)STAGES * 64 * (WM + WN) * BK * 2;
    const int wgs = (int)(160 * 1024 / lds) > 0 ? (int)(160 * 1024 / lds) : 1;     // workgroups per CU the LDS allows
    const int w = wgs * WM * WN / 4;                                                 // waves per SIMD
    return w > 4 ? 4 : (w < 1 ? 1 : w);
}

//
// MI: 32-row A blocks per wave (2 = every wave a 64x64 sub-tile; 4 = 128 (m) x 64 (n), kept for experiments).
// WN = 5 gives tiles that span N = 320 / 640 / 960 exactly: no padded columns and A read once per 320 columns -- the
// K = 320 projections are bound by L2 -> LDS operand traffic (64 flop/B with 128x128 tiles, 91 with 128x320).
template <int MODE, int EPI, int WM, int WN, int MI, int BK, int STAGES, bool SK>
__global__ __launch_bounds__(64 * WM * WN, (gemm_min_waves<WM, WN, MI, BK, STAGES>()))
void gemm_kernel(const GemmParams P) {
    constexpr int BM = 32 * MI * WM, BN = 64 * WN, NW = WM * WN, NT = 64 * NW;
    constexpr int STAGE_ELEMS = (BM + BN) * BK;
    constexpr int RPP = 512 / BK;                    // tile rows per 1-KiB DMA piece (8 x 128 B or 16 x 64 B)
    constexpr int CPRW = BK / 8;                     // 16-byte chunks per tile row
    constexpr int PIECES = (BM + BN) / RPP;          // pieces per k-tile
    constexpr int PPW = (PIECES + NW - 1) / NW;      // pieces per wave (the last round may be ragged)
    constexpr int CP = BN + 8;                       // fp32 C slab pitch
    constexpr bool FRAG_ALL = NW <= 8 && BK == 64 && MI == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    // ---- work assignment ---------------------------------------------------------------------------------------
    // plain / split-K: one output tile per workgroup, XCD aware (blockIdx % 8 = XCD): the n-tiles of one m-tile stay on
    // one XCD.  stream-K (P.sk persistent workgroups, one per CU slot): the tiles are dealt to the 8 XCD groups at tile
    // granularity; inside a group the (tile, k-tile) iteration space is cut into equal contiguous ranges, one per
    // workgroup, so every CU does the same number of k-tiles whatever tiles/256 is.  A tile cut by a range boundary is
    // finished by the workgroup that holds its FIRST k-tiles (it reaches them at the END of its range, when the others
    // -- who meet their share of that tile at the START of theirs -- are long done): they leave fp32 partials in
    // ws[block] and raise sk_flags[block]; ranges are handed out in reverse block order, so a workgroup only ever
    // waits for lower block ids.
    const int nkt = P.K / BK;
    int64_t sk_it = 0, sk_end = 0, sk_gbase = 0, sk_I = 0;
    int sk_r = 0, sk_per = 1;
    if (SK) {
        const int T = P.tiles_m * P.tiles_n, x = blockIdx.x & 7, q = blockIdx.x >> 3;
        sk_per = P.sk >> 3;
        const int tg0 = (int)((int64_t)T * x / 8), tg1 = (int)((int64_t)T * (x + 1) / 8);
        sk_gbase = (int64_t)tg0 * nkt;
        sk_I = (int64_t)(tg1 - tg0) * nkt;
        sk_r = sk_per - 1 - q;
        sk_it = sk_gbase + sk_I * sk_r / sk_per;
        sk_end = sk_gbase + sk_I * (sk_r + 1) / sk_per;
    }
    for (;;) {                                           // one pass per segment (exactly one without stream-K)
    int tile_m, tile_n, kt0 = 0, nk = nkt;
    int sk_np = 0;                                       // stream-K owner: partials to add (from blocks id - 8, - 16, ..)
    bool sk_partial = false;                             // stream-K: this segment leaves a partial instead of output
    if (SK) {
        if (sk_it >= sk_end) break;
        const int lin = (int)(sk_it / nkt);
        kt0 = (int)(sk_it - (int64_t)lin * nkt);
        nk = (int)min((int64_t)nkt, kt0 + (sk_end - sk_it));
        sk_it += nk - kt0;
        lin_to_tile(lin, P, tile_m, tile_n);
        sk_partial = kt0 > 0;
        if (kt0 == 0 && nk < nkt) {                      // I hold the head of a cut tile: who holds the rest?
            const int64_t tile_end = (int64_t)(lin + 1) * nkt;
            int64_t e = sk_end;
            while (e < tile_end) {
                ++sk_np;
                e = sk_gbase + sk_I * (sk_r + 1 + sk_np) / sk_per;
            }
        }
    } else {
        const int total = P.tiles_m * P.tiles_n;
        int id = blockIdx.x;
        if (P.split_k > 1) id /= P.split_k;              // the splits of one tile are neighbours in launch order
        int lin = id;
        if (P.split_k == 1) {                            // XCD x = id % 8 owns a contiguous range of the tile order
            const int q = total >> 3, r = total & 7, x = id & 7;
            lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
        }
        lin_to_tile(lin, P, tile_m, tile_n);
    }
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- my LDS-DMA pieces: piece p = wave + NW*j; p < BM/8 -> rows 8p.. of A, else rows of W --------------------
    const bf16_t* src[PPW];
    const bf16_t* src2[PPW];                         // second A source (token mode, two-source operand)
    bool val[PPW];
    int py[PPW], px[PPW];
    const int prow = lane / CPRW, pphys = lane % CPRW;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = wave + NW * j;
        if (p >= PIECES) {
            src[j] = src2[j] = nullptr; val[j] = false; py[j] = px[j] = 0;
        } else if (p < BM / RPP) {
            const int rloc = RPP * p + prow;
            const int sc = swz<BK>(rloc, pphys);
            const int64_t m = m0 + rloc;
            val[j] = m < P.M;
            const int64_t mm = val[j] ? m : 0;
            if (MODE == 0) {
                src[j] = P.a + mm * P.lda + sc * 8;
                src2[j] = P.a2 ? P.a2 + mm * P.lda2 + sc * 8 : nullptr;
                py[j] = px[j] = 0;
            } else {
                const int pix = (int)(mm % P.hw);
                src2[j] = nullptr;
                py[j] = pix / P.img_w;
                px[j] = pix - py[j] * P.img_w;
                if (P.ups == 1) src[j] = P.a + (mm / P.hw) * (int64_t)(P.hw >> 2) * P.cin + sc * 8;   // image base (half-res)
                else if (P.ups == 2) src[j] = P.a + (mm / P.hw) * (int64_t)(P.hw << 2) * P.cin + sc * 8;   // image base (2x res)
                else src[j] = P.a + mm * P.cin + sc * 8;
            }
        } else {
            const int rloc = RPP * (p - BM / RPP) + prow;
            const int sc = swz<BK>(rloc, pphys);
            const int n = n0 + rloc;
            val[j] = n < P.N;
            src[j] = P.w + (int64_t)(val[j] ? n : 0) * P.K + sc * 8;
            src2[j] = nullptr;
            py[j] = px[j] = 0;
        }
    }
    auto dma_issue = [&](int kt, int buf) {
        int k0 = kt * BK;
        bf16_t* stage = smem + buf * STAGE_ELEMS;
        const bf16_t* zero = reinterpret_cast<const bf16_t*>(&g_zero16);
        int dy = 0, dx = 0, ci0 = 0;
        int tap = 0;
        if (MODE == 1) {
            // k-tile order = (channel chunk outer, tap inner): the 9 taps of a chunk re-read the same input lines (shifted by
            // a pixel) in 9 consecutive k-steps, so they hit the L2.  With (tap outer, channel inner) the re-use distance was
            // the whole channel depth of every co-resident tile -- the input came back from the Infinity Cache once per tap
            // (PMC: 565 MB of L2 misses per 1280->640 launch whose input is 65 MB).
            if (P.tap_outer) {
                tap = k0 / P.cin;
                ci0 = k0 - tap * P.cin;
            } else {
                const int chunk = kt / 9;
                tap = kt - chunk * 9;
                ci0 = chunk * BK;
                k0 = tap * P.cin + ci0;
            }
            dy = tap / 3 - 1;
            dx = tap - (tap / 3) * 3 - 1;
        }
        const int64_t shift = MODE == 1 ? ((int64_t)dy * P.img_w + dx) * P.cin + ci0 : (int64_t)k0;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int p = wave + NW * j;             // wave-uniform
            if (p >= PIECES) {                       // ragged last round: with a counted vmcnt every wave must issue the
                if (STAGES > 2) dma16(zero, smem + STAGES * STAGE_ELEMS);   // same number of DMAs -> dummy into a scratch KiB
                continue;
            }
            if (p < BM / RPP) {
                bool ok = val[j];
                if (MODE == 1)
                    ok = ok && (unsigned)(py[j] + dy) < (unsigned)P.img_h && (unsigned)(px[j] + dx) < (unsigned)P.img_w;
                if (MODE == 1 && P.ups == 1) {       // tap (y+dy, x+dx) of the upsampled image = source pixel (.. >> 1)
                    const int64_t so = ((int64_t)((py[j] + dy) >> 1) * (P.img_w >> 1) + ((px[j] + dx) >> 1)) * P.cin + ci0;
                    dma16(ok ? src[j] + so : zero, stage + p * 512);
                } else if (MODE == 1 && P.ups == 2) {   // stride 2, pad 1: input pixel (2y+dy, 2x+dx) of the 2H x 2W image
                    const int iy = 2 * py[j] + dy, ix = 2 * px[j] + dx;
                    const bool ok2 = val[j] && iy >= 0 && ix >= 0;                // (the high side is always inside)
                    dma16(ok2 ? src[j] + ((int64_t)iy * (2 * P.img_w) + ix) * P.cin + ci0 : zero, stage + p * 512);
                } else if (MODE == 0 && P.a2 && k0 >= P.ksplit) {
                    dma16(ok ? src2[j] + (k0 - P.ksplit) : zero, stage + p * 512);
                } else {
                    dma16(ok ? src[j] + shift : zero, stage + p * 512);
                }
            } else {
                dma16(val[j] ? src[j] + k0 : zero, stage + p * 512);       // W rows follow the A rows in the stage
            }
        }
    };

    f32x16 acc[2][MI];                               // [ni][mi]: rows = n (registers), cols = m (lanes)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int split = P.split_k > 1 ? (int)(blockIdx.x % P.split_k) : 0;
    if (P.split_k > 1) {
        const int per = (nk + P.split_k - 1) / P.split_k;
        kt0 = split * per;
        nk = min(nk, kt0 + per);
    }
    // ring: tile kt lives in stage (kt - kt0) % STAGES; tiles kt+1 .. kt+STAGES-1 stream in under compute(kt)
#pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0)
        if (kt0 + s0 < nk) dma_issue(kt0 + s0, s0);
    int st_cur = 0, st_free = STAGES - 1;            // stage of tile kt, stage that tile kt+STAGES-1 will use
    for (int kt = kt0; kt < nk; ++kt) {
        // my pieces of tile kt have landed (the younger STAGES-2 tiles may still be in flight); after the barrier
        // everybody's have, and everybody has finished reading the stage of tile kt-1, which is the one tile
        // kt+STAGES-1 streams into
        {
            const int younger = min(STAGES - 2, nk - 1 - kt);
            if (STAGES >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            else if (STAGES >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kt + STAGES - 1 < nk) dma_issue(kt + STAGES - 1, st_free);
        const bf16_t* As = smem + st_cur * STAGE_ELEMS;
        st_free = st_cur;
        st_cur = st_cur + 1 == STAGES ? 0 : st_cur + 1;
        const bf16_t* Ws = As + BM * BK;
        // fragments are double-buffered in registers: the ds_read_b128s of k-step ks+1 are issued before the MFMAs
        // of k-step ks (sched_barrier pins that order; left alone the compiler re-uses one register set and every
        // k-step eats its own LDS latency)
        auto load_frags = [&](int ks, bf16x8 (&wf)[2], bf16x8 (&af)[MI]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rw = wn * 64 + i * 32 + l31;
                union { bf16x8 v; u32x4 u; } tw;
                tw.u = *reinterpret_cast<const u32x4*>(Ws + rw * BK + swz<BK>(rw, 2 * ks + half) * 8);
                wf[i] = tw.v;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int rm = wm * (32 * MI) + i * 32 + l31;
                union { bf16x8 v; u32x4 u; } ta;
                ta.u = *reinterpret_cast<const u32x4*>(As + rm * BK + swz<BK>(rm, 2 * ks + half) * 8);
                af[i] = ta.v;
            }
        };
        if constexpr (FRAG_ALL) {
            // every fragment of the k-tile requested up front: with the double-buffered form below a ds_read_b128 has the 4 MFMAs
            // of one k-step (128 cycles) to land, which it does not under load -- +8..13 % on the deep-K shapes.  Needs 32 more
            // VGPRs: the geometries below 16 waves per workgroup have them
            bf16x8 wfa[BK / 16][2], afa[BK / 16][MI];
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) load_frags(ks, wfa[ks], afa[ks]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfa[ks][ni], afa[ks][mi], acc[ni][mi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            bf16x8 wf[2][2], af[2][MI];
            load_frags(0, wf[0], af[0]);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                if (ks + 1 < BK / 16) load_frags(ks + 1, wf[(ks + 1) & 1], af[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][ni], af[ks & 1][mi], acc[ni][mi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    if (SK && sk_np > 0 && tid == 0) {
        // stream-K owner: the other shares of this tile were produced at the START of their workgroups' ranges.  Flag and
        // payload are sc1 (agent-scope) accesses, so no cache-wide acquire is needed; the __syncthreads() at the top of
        // the slab loop orders the payload reads of the workgroup after this wait.  The spin is bounded: a bug must not
        // hang the GPU.
        for (int j = 1; j <= sk_np; ++j) {
            const int* f = P.sk_flags + (blockIdx.x - 8 * j);
            int spins = 0;
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++spins < (1 << 22))
                __builtin_amdgcn_s_sleep(8);
        }
    }

    // ---- fp32-storage mode (never with stream-K; split-K writes its raw partials below and splitk_reduce_kernel finishes) ------
    if (!SK && P.f32io && P.split_k == 1) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int64_t m = m0 + wm * (32 * MI) + mi * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (EPI == 1) {
                    const float av[4] = {acc[0][mi][4 * g], acc[0][mi][4 * g + 1], acc[0][mi][4 * g + 2], acc[0][mi][4 * g + 3]};
                    const float ag[4] = {acc[1][mi][4 * g], acc[1][mi][4 * g + 1], acc[1][mi][4 * g + 2], acc[1][mi][4 * g + 3]};
                    epi_f32_geglu_quad(P, av, ag, m, n0 + wn * 64 + 8 * g + 4 * half, n0 / 2 + wn * 32 + 8 * g + 4 * half);
                } else {
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const float a4[4] = {acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]};
                        epi_f32_quad<MODE>(P, a4, m, n0 + wn * 64 + ni * 32 + 8 * g + 4 * half);
                    }
                }
            }
        }
        break;
    }

    // ---- GEGLU epilogue of the plain grid: in registers.  The weight rows are interleaved per 64 so that acc[0] holds the
    // value and acc[1] the gate of the SAME 32 output columns, register for register: every wave gates its own 64 x 32
    // outputs at once, stages them as bf16 (one LDS round for the whole tile instead of one fp32 slab round per wave
    // row, a quarter of the LDS bytes) and the tile leaves with whole-row 16-byte stores.
    if (EPI == 1 && !SK && P.regepi) {
        constexpr int OP = BN / 2 + 8;                   // bf16 pitch of the staged output tile (16-byte aligned rows)
        bf16_t* Os = smem;                               // [BM][OP]
        float ba[16], bg[16];
        const bool cols_ok = n0 + wn * 64 < P.N;        // N % 64 == 0: a 64-column group is inside or outside as a whole
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nb = n0 + wn * 64 + 8 * g + 4 * half + j;
                ba[4 * g + j] = (P.bias && cols_ok) ? bf2f(P.bias[nb]) : 0.f;
                bg[4 * g + j] = (P.bias && cols_ok) ? bf2f(P.bias[nb + 32]) : 0.f;
            }
        __syncthreads();                                 // every wave is done with the operand ring
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = (acc[0][mi][4 * g + j] + ba[4 * g + j]) * gelu_erf(acc[1][mi][4 * g + j] + bg[4 * g + j]);
                const int row = wm * (32 * MI) + mi * 32 + l31, col = wn * 32 + 8 * g + 4 * half;
                *reinterpret_cast<u32x2*>(Os + row * OP + col) = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
            }
        __syncthreads();
        constexpr int OCPR = BN / 16;                    // 16-byte chunks per output row
#pragma unroll 4
        for (int c = tid; c < BM * OCPR; c += NT) {
            const int r = c / OCPR, ch = c - r * OCPR;
            const int64_t m = m0 + r;
            const int no = n0 / 2 + ch * 8;
            if (m < P.M && no < P.N / 2)
                *reinterpret_cast<u32x4*>(P.out + m * P.ldo + no) = *reinterpret_cast<const u32x4*>(Os + r * OP + ch * 8);
        }
        break;
    }

    // ---- the same for the plain epilogue when there is no residual to add (QKV / proj_in projections, the first conv of a
    // ResNet block with its time-embedding row): bias, alpha and temb are applied in the accumulator registers, the
    // tile is staged once as bf16.  A residual goes through the same tile first (see below), so it is still added in fp32
    // before the one rounding.  Measured alternatives: the residual read from global memory in accumulator layout (8
    // bytes per lane and row) was 20 % slower than whole-row 16-byte loads; adding it at the read-out of the bf16
    // tile (two roundings, like the un-fused ops) was no faster.  Two residuals keep the fp32 slab path below.
    if (EPI == 0 && !SK && P.split_k == 1 && !P.res2 && P.regepi && (P.regepi == 1 || !P.res)) {
        constexpr int OP = BN + 8;
        bf16_t* Os = smem;                               // [BM][OP]
        // residual: whole-row 16-byte loads into registers now, into the staging tile behind the barrier; every lane then
        // picks its own (row, 4 columns) words out of LDS, adds them in fp32 and overwrites them with the result
        constexpr int R_CPR = BN / 8, R_IT = BM * R_CPR / NT;
        static_assert(BM * R_CPR % NT == 0, "tile chunks must divide over the threads");
        const bool has_res = P.res != nullptr;
        u32x4 rchunk[R_IT];
        if (has_res) {
#pragma unroll
            for (int it = 0; it < R_IT; ++it) {
                const int c = tid + it * NT, r = c / R_CPR, ch = c - r * R_CPR;
                rchunk[it] = *reinterpret_cast<const u32x4*>(P.res + min(m0 + r, P.M - 1) * P.ldres + min(n0 + ch * 8, P.N - 8));
            }
        }
        __syncthreads();                                 // every wave is done with the operand ring
        if (has_res) {
#pragma unroll
            for (int it = 0; it < R_IT; ++it) {
                const int c = tid + it * NT, r = c / R_CPR, ch = c - r * R_CPR;
                *reinterpret_cast<u32x4*>(Os + r * OP + ch * 8) = rchunk[it];
            }
            __syncthreads();
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            float bv[16];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nb = n0 + wn * 64 + ni * 32 + 8 * g + 4 * half + j;
                    bv[4 * g + j] = (P.bias && nb < P.N) ? bf2f(P.bias[nb]) : 0.f;
                }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wm * (32 * MI) + mi * 32 + l31;
                const bf16_t* trow = nullptr;
                if (MODE == 1 && P.temb) {
                    const int64_t m = min(m0 + row, P.M - 1);
                    trow = P.temb + ((m / P.hw) / P.temb_div) * P.temb_ld;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = wn * 64 + ni * 32 + 8 * g + 4 * half;
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (acc[ni][mi][4 * g + j] + bv[4 * g + j]) * P.alpha;
                    if (MODE == 1 && P.temb) {
                        const u32x2 t = *reinterpret_cast<const u32x2*>(trow + min(n0 + col, P.N - 4));
                        o[0] += __uint_as_float(t[0] << 16); o[1] += __uint_as_float(t[0] & 0xffff0000u);
                        o[2] += __uint_as_float(t[1] << 16); o[3] += __uint_as_float(t[1] & 0xffff0000u);
                    }
                    if (has_res) {
                        const u32x2 t = *reinterpret_cast<const u32x2*>(Os + row * OP + col);
                        o[0] += __uint_as_float(t[0] << 16); o[1] += __uint_as_float(t[0] & 0xffff0000u);
                        o[2] += __uint_as_float(t[1] << 16); o[3] += __uint_as_float(t[1] & 0xffff0000u);
                    }
                    *reinterpret_cast<u32x2*>(Os + row * OP + col) = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                }
            }
        }
        __syncthreads();
        constexpr int OCPR = BN / 8;                     // 16-byte chunks per output row
#pragma unroll 4
        for (int c = tid; c < BM * OCPR; c += NT) {
            const int r = c / OCPR, ch = c - r * OCPR;
            const int64_t m = m0 + r;
            const int n = n0 + ch * 8;
            if (m < P.M && n < P.N)
                *reinterpret_cast<u32x4*>(P.out + m * P.ldo + n) = *reinterpret_cast<const u32x4*>(Os + r * OP + ch * 8);
        }
        break;
    }

    // ---- epilogue: fp32 C slabs of 64 rows through LDS, bias / alpha / temb / residual / GEGLU on the way out ----------
    float* Cs = reinterpret_cast<float*>(smem_raw);             // [64][CP] fp32
#pragma unroll 1
    for (int hw = 0; hw < WM; ++hw)
#pragma unroll
    for (int part = 0; part < MI / 2; ++part) {      // 64-row slab hm: rows [64*part, 64*part+64) of wave row hw
        const int hm = hw * (MI / 2) + part;
        // the residual rows of this slab are requested NOW, all at once and branch-free, so that their latency runs under
        // the slab staging below (one load per row inside the output loop is one exposed memory round trip per row:
        // 8 of them were ~12 us of a 26 us workgroup on the K = 320 projections)
        constexpr int E_CPR = BN / 8, E_RSTEP = NT / E_CPR, E_IT = 64 / E_RSTEP;
        static_assert(64 % E_RSTEP == 0, "slab rows must divide over the threads");
        u32x4 resv[E_IT], resv2[E_IT];
        if (EPI == 0 && P.res && !(SK && sk_partial) && P.split_k == 1) {
            const int ch = tid % E_CPR;
            const int nn = min(n0 + ch * 8, P.N - 8);
#pragma unroll
            for (int it = 0; it < E_IT; ++it) {
                const int64_t m = min(m0 + hm * 64 + tid / E_CPR + it * E_RSTEP, P.M - 1);
                resv[it] = *reinterpret_cast<const u32x4*>(P.res + m * P.ldres + nn);
                if (P.res2) resv2[it] = *reinterpret_cast<const u32x4*>(P.res2 + m * P.ldres + nn);
            }
        }
        __syncthreads();
        if (wm == hw) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int m = mi * 32 + l31;
                        const int n = wn * 64 + ni * 32 + 8 * g + 4 * half;
                        const f32x16& a = acc[ni][part * 2 + mi];
                        *reinterpret_cast<f32x4*>(Cs + m * CP + n) = f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                    }
        }
        __syncthreads();
        if (SK && sk_np > 0) {                           // stream-K owner: fold the other shares of this tile into the slab
            constexpr int C4 = BN / 4;
            const __amdgpu_buffer_rsrc_t rs = sk_rsrc(P.ws);
#pragma unroll
            for (int c = tid; c < 64 * C4; c += NT) {
                const int r = c / C4, c4 = c - r * C4;
                f32x4 v = *reinterpret_cast<const f32x4*>(Cs + r * CP + c4 * 4);
                for (int j = 1; j <= sk_np; ++j) {
                    union { u32x4 u; f32x4 f; } t;
                    t.u = __builtin_amdgcn_raw_buffer_load_b128(
                        rs, (int)((((int64_t)(blockIdx.x - 8 * j) * BM + hm * 64 + r) * BN + c4 * 4) * 4), 0, SK_SC);
                    v += t.f;
                }
                *reinterpret_cast<f32x4*>(Cs + r * CP + c4 * 4) = v;
            }
            __syncthreads();
        }
        if (SK && sk_partial) {                          // stream-K: tile-local fp32 partial in my slot
            constexpr int C4 = BN / 4;                   // 16-byte chunks per row; consecutive lanes = consecutive chunks
            const __amdgpu_buffer_rsrc_t rs = sk_rsrc(P.ws);
#pragma unroll
            for (int c = tid; c < 64 * C4; c += NT) {
                const int r = c / C4, c4 = c - r * C4;
                const u32x4 v = *reinterpret_cast<const u32x4*>(Cs + r * CP + c4 * 4);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)((((int64_t)blockIdx.x * BM + hm * 64 + r) * BN + c4 * 4) * 4), 0, SK_SC);
            }
        } else if (EPI == 0 && P.split_k > 1) {          // raw fp32 partial sums; epilogue in splitk_reduce_kernel
            constexpr int CPR = BN / 8;
            constexpr int RSTEP = NT / CPR;
            const int ch = tid % CPR, n = n0 + ch * 8;
            if (n < P.N) {
                for (int r = tid / CPR; r < 64; r += RSTEP) {
                    const int64_t m = m0 + hm * 64 + r;
                    if (m >= P.M) continue;
                    float* dst = P.ws + ((int64_t)split * P.M + m) * P.N + n;
                    *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(Cs + r * CP + ch * 8);
                    *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(Cs + r * CP + ch * 8 + 4);
                }
            }
        } else if (EPI == 0) {
            constexpr int CPR = BN / 8;                  // 8-column chunks per slab row
            constexpr int RSTEP = NT / CPR;              // rows covered per pass
            const int ch = tid % CPR, n = n0 + ch * 8;
            if (n < P.N) {
                float bv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) bv[i] = P.bias ? bf2f(P.bias[n + i]) : 0.f;
#pragma unroll
                for (int it = 0; it < E_IT; ++it) {
                    const int r = tid / CPR + it * RSTEP;
                    const int64_t m = m0 + hm * 64 + r;
                    if (m >= P.M) continue;
                    float v[8];
                    Vec8<float>::load(Cs + r * CP + ch * 8, v);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (v[k] + bv[k]) * P.alpha;
                    if (MODE == 1 && P.temb) {
                        float t[8];
                        Vec8<bf16_t>::load(P.temb + ((m / P.hw) / P.temb_div) * P.temb_ld + n, t);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += t[k];
                    }
                    if (P.res) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            v[2 * k] += __uint_as_float(resv[it][k] << 16);
                            v[2 * k + 1] += __uint_as_float(resv[it][k] & 0xffff0000u);
                        }
                        if (P.res2) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                v[2 * k] += __uint_as_float(resv2[it][k] << 16);
                                v[2 * k + 1] += __uint_as_float(resv2[it][k] & 0xffff0000u);
                            }
                        }
                    }
                    Vec8<bf16_t>::store(P.out + m * P.ldo + n, v);
                }
            }
        } else {
            // GEGLU: within every 64 tile columns, [0,32) = value, [32,64) = gate -> 32 output columns
            constexpr int CPR = BN / 16;                 // 8-column output chunks per slab row
            constexpr int RSTEP = NT / CPR;
            const int ch = tid % CPR;
            const int grp = ch / 4, cc = ch % 4;         // 64-column group inside the tile, chunk inside the group
            const int ncol = grp * 64 + cc * 8;          // tile column of the value chunk
            const int no = (n0 / 2) + grp * 32 + cc * 8; // output column
            if (no < P.N / 2) {
                float ba[8], bg[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    ba[i] = P.bias ? bf2f(P.bias[n0 + ncol + i]) : 0.f;
                    bg[i] = P.bias ? bf2f(P.bias[n0 + ncol + 32 + i]) : 0.f;
                }
                for (int r = tid / CPR; r < 64; r += RSTEP) {
                    const int64_t m = m0 + hm * 64 + r;
                    if (m >= P.M) continue;
                    float a[8], g[8];
                    Vec8<float>::load(Cs + r * CP + ncol, a);
                    Vec8<float>::load(Cs + r * CP + ncol + 32, g);
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[k] = (a[k] + ba[k]) * gelu_erf(g[k] + bg[k]);
                    Vec8<bf16_t>::store(P.out + m * P.ldo + no, a);
                }
            }
        }
    }
    if (!SK) break;
    if (sk_partial) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my sc1 payload stores have completed ...
        __syncthreads();
        if (tid == 0)                                    // ... everybody's have: raise the flag
            __hip_atomic_store(P.sk_flags + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (sk_np > 0) {
        __syncthreads();                                 // everybody has read the partials: hand the flags back as zeros
        if (tid == 0)
            for (int j = 1; j <= sk_np; ++j)
                __hip_atomic_store(P.sk_flags + (blockIdx.x - 8 * j), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                     // the C slab aliases the operand stages of the next segment
    }  // segment loop
}

// =====================================================================================================================
// The 8-phase 256x256 kernel (tile arms 13 / 14): 8 waves (2 along M x 4 along N), every wave a 128 x 64 output block as
// 2 x 4 v_mfma_f32_32x32x16_bf16 accumulators (128 accumulator registers), BK = 64, TWO k-tile buffers in LDS.
//
// What differs from gemm_kernel above (whose 16-wave 256x256 tile re-reads every fragment by 4 waves -- the LDS port was its
// bound, DESIGN.md section 5 -- and whose waves all load, wait, compute in lockstep):
//   * a k-tile is FOUR phases; a phase = {LOAD: the ds_read_b128s of the fragments this phase needs + the LDS-DMA issue of one
//     half-tile of a LATER k-tile + a counted vmcnt} | barrier | {8 MFMAs = one quadrant (64 rows x 32 columns) of the wave's
//     block over the whole 64-deep k-tile, under s_setprio 1} | barrier;
//   * the two wave rows run ONE BARRIER APART (wave row 1 passes an extra barrier before the loop, wave row 0 one after it): in
//     every inter-barrier segment one wave of each SIMD issues MFMAs while the other wave of that SIMD does its LDS reads /
//     DMA issue / waiting -- the matrix pipe never waits for a fragment;
//   * operands are DMA'd in half-tiles ordered by first use -- A early rows (the first 64 of each wave row's 128), W early
//     rows (the first 32 of each wave column's 64), W late, A late -- PF half-tiles ahead of their first read, never drained:
//     `s_waitcnt vmcnt(2 (PF - 2))` at the end of LOAD(g) retires exactly what LOAD(g + 1) reads, the barrier after it
//     publishes it (a wave reads DMA'd data one phase AFTER the wait that retired it, CDNA4 guide section 5); past the end of
//     K the issue slot sends a dummy KiB to a scratch area so the count stays exact;
//   * fragments per wave and k-tile: 16 A + 8 W ds_read_b128 (24 KiB) instead of 2 x 16 KiB x 4 re-reads.
// WAR: slot(h) is re-filled at LOAD(h - PF); its previous content (h - 8) was last read at LOAD(h - 8) of the LATER wave row,
// retired by that row's lgkmcnt(0) one segment on: safe for PF <= 6.
template <int MODE, int EPI, int PF, int ROLE>
__global__ __launch_bounds__(512, 2)
void gemm8_kernel(const GemmParams P) {
    constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
    constexpr int STAGE_ELEMS = (BM + BN) * BK;      // 64 KiB per k-tile buffer
    constexpr int VMW = 2 * (PF - 2);                // DMA instructions that may stay in flight at the end of a LOAD part
    static_assert(PF >= 3 && PF <= 6, "prefetch distance (half-tiles)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nkt = P.K / BK;
    // ROLE 0: plain grid, one output tile per workgroup (main loop + epilogue).
    // ROLE 1 + ROLE 2: stream-K as TWO launches.  ROLE 1 = P.sk persistent workgroups (one per CU); the (tile, k-tile) iteration
    // space of every XCD group is cut into equal contiguous ranges, one per workgroup, so every CU does the same number of
    // k-tiles whatever tiles / CUs is; every segment (= the part of one tile inside my range) leaves its fp32 accumulators in a
    // workspace slot in ACCUMULATOR layout (slot[wave][register quad][lane] x 16 B: every store moves one contiguous KiB) and
    // nothing else -- no epilogue code, no flags, no spinning.  ROLE 2 = one workgroup per output tile: it sums the slots of
    // the segments that cover its tile (same lane <-> element map, so the sum lands in accumulator registers) and runs the normal
    // epilogue.  (ROLE 3 = a persistent pass that also finishes the tiles it holds whole, ROLE 2 then only the cut ones:
    // measured slower -- the plain-epilogue instantiation spills again, and even the spill-free GEGLU one only ties the plain
    // grid at 1600 tiles; not instantiated.)  The kernel boundary is the synchronisation; the summation order is fixed.  (A single persistent kernel that
    // also finishes the tiles -- built first -- left the register allocator with two live versions of the 128 accumulators
    // around the segment loop: 350-800 spilled registers, +22 us per launch.)  Costs 2 x M x N x 4 bytes of extra traffic:
    // the arm for small outputs with a long reduction (the 10x16 / 5x8-level convolutions, M*N <= ~8 M elements).
    constexpr bool SK = ROLE == 1 || ROLE == 3;
    int64_t sk_it = 0, sk_end = 0, sk_gbase = 0, sk_I = 0;
    int sk_r = 0, sk_per = 1, sk_first = 0;
    int lk_u = 0;                                     // k-lockstep form: my next unit
    if (SK && P.sk_lock) {
        lk_u = (int)(blockIdx.x & 7) * (P.sk >> 3) + (int)(blockIdx.x >> 3);
    } else if (SK) {
        const int T = P.tiles_m * P.tiles_n - P.sk_t0, x = blockIdx.x & 7, q = blockIdx.x >> 3;
        sk_per = P.sk >> 3;
        const int tg0 = (int)((int64_t)T * x / 8), tg1 = (int)((int64_t)T * (x + 1) / 8);
        sk_gbase = (int64_t)tg0 * nkt;
        sk_I = (int64_t)(tg1 - tg0) * nkt;
        sk_r = sk_per - 1 - q;
        sk_it = sk_gbase + sk_I * sk_r / sk_per;
        sk_end = sk_gbase + sk_I * (sk_r + 1) / sk_per;
        sk_first = (int)(sk_it / nkt);
    }
    const __amdgpu_buffer_rsrc_t rsP = sk_rsrc(P.ws);
    f32x16 acc[2][4];                                 // [ni][mi]: rows = n (registers), cols = m (lanes); zeroed right before the k loop
    bf16x8 wf[2][4], af[2][4];                        // W fragments of both 32-column halves; A fragments of the current 64-row half
    for (;;) {                                       // one pass per segment (exactly one for ROLE 0 / 2)
    int tid = threadIdx.x;
    if (SK) asm volatile("" : "+v"(tid));            // (per-lane indices of a segment must not be hoisted out of the persistent loop)
    const int lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int prow = lane >> 3, pch = lane & 7;      // my row / 16-byte chunk inside a 1-KiB DMA piece (8 rows x 128 B)
    int tile_m, tile_n, kt0 = 0, nk = nkt, sk_slot = 0;
    if (SK && P.sk_lock) {
        const int T = P.tiles_m * P.tiles_n;
        if (lk_u >= T * P.sk_lock) break;
        const int c = lk_u / T, t = lk_u - c * T;
        tile_n = t / P.tiles_m;
        tile_m = t - tile_n * P.tiles_m;
        kt0 = c * P.sk_lock_len;
        nk = min(nkt, kt0 + P.sk_lock_len);
        sk_slot = lk_u;
        lk_u += P.sk;
    } else if (SK) {
        if (sk_it >= sk_end) break;
        const int lin = (int)(sk_it / nkt);
        kt0 = (int)(sk_it - (int64_t)lin * nkt);
        nk = (int)min((int64_t)nkt, kt0 + (sk_end - sk_it));
        sk_it += nk - kt0;
        sk_slot = blockIdx.x * P.sk_maxseg + (lin - sk_first);
        lin_to_tile(lin + P.sk_t0, P, tile_m, tile_n);
    } else {
        // (hybrid: the plain launch covers the first sk_t0 tiles of the launch order, the finishing launch the rest)
        const int total = ROLE == 2 ? P.tiles_m * P.tiles_n - P.sk_t0 : (P.sk_t0 > 0 ? P.sk_t0 : P.tiles_m * P.tiles_n);
        const int id = blockIdx.x, q = total >> 3, r = total & 7, x = id & 7;
        const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);   // XCD x owns a contiguous range
        lin_to_tile(ROLE == 2 ? lin + P.sk_t0 : lin, P, tile_m, tile_n);
        if (ROLE == 2 && P.sk_lock) {                 // k-lockstep pass: chunk c of tile t left its accumulators in slot c * T + t
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a2][b2][e] = 0.f;
            const int T = P.tiles_m * P.tiles_n, t = tile_n * P.tiles_m + tile_m;
#pragma unroll 1
            for (int c = 0; c < P.sk_lock; ++c) {
                const int slot = c * T + t;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        u32x4 t4[4];
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const int quad = (wave * 8 + ni * 4 + mi) * 4 + gq;
                            t4[gq] = __builtin_amdgcn_raw_buffer_load_b128(rsP, (int)(((int64_t)slot * (BM * BN / 4) + quad * 64 + lane) * 16), 0, 0);
                        }
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            union { u32x4 u; f32x4 f; } cv;
                            cv.u = t4[gq];
                            acc[ni][mi][4 * gq] += cv.f[0]; acc[ni][mi][4 * gq + 1] += cv.f[1];
                            acc[ni][mi][4 * gq + 2] += cv.f[2]; acc[ni][mi][4 * gq + 3] += cv.f[3];
                        }
                    }
            }
        } else if (ROLE == 2) {
            // the segments of ROLE 1 that cover tile `lin`: XCD group g holds tiles [T g / 8, T (g + 1) / 8); inside it range r of
            // `per` covers iterations [I r / per, I (r + 1) / per) and was run by block (per - 1 - r) * 8 + g
            const int T = total, per = P.sk >> 3;
            int g = 0;
            while ((int)((int64_t)T * (g + 1) / 8) <= lin) ++g;
            const int tg0 = (int)((int64_t)T * g / 8), tg1 = (int)((int64_t)T * (g + 1) / 8);
            const int64_t gbase = (int64_t)tg0 * nkt, I = (int64_t)(tg1 - tg0) * nkt;
            const int64_t lo = (int64_t)lin * nkt - gbase, hi = lo + nkt;
            int r0 = (int)(lo * per / I);
            while (r0 + 1 < per && I * (r0 + 1) / per <= lo) ++r0;
            while (r0 > 0 && I * r0 / per > lo) --r0;
            if (P.sk_whole && I * r0 / per <= lo && I * (r0 + 1) / per >= hi) return;   // a whole tile of one range: finished by the ROLE 3 pass
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a2][b2][e] = 0.f;
#pragma unroll 1
            for (int r2 = r0; r2 < per && I * r2 / per < hi; ++r2) {
                const int64_t rs = I * r2 / per, re = I * (r2 + 1) / per;
                if (min(re, hi) <= max(rs, lo)) continue;                    // (an empty range)
                const int blk = (per - 1 - r2) * 8 + g;
                const int slot = blk * P.sk_maxseg + (lin - (int)((gbase + rs) / nkt));
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        u32x4 t[4];
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const int quad = (wave * 8 + ni * 4 + mi) * 4 + gq;
                            t[gq] = __builtin_amdgcn_raw_buffer_load_b128(rsP, (int)(((int64_t)slot * (BM * BN / 4) + quad * 64 + lane) * 16), 0, 0);
                        }
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            union { u32x4 u; f32x4 f; } c;
                            c.u = t[gq];
                            acc[ni][mi][4 * gq] += c.f[0]; acc[ni][mi][4 * gq + 1] += c.f[1];
                            acc[ni][mi][4 * gq + 2] += c.f[2]; acc[ni][mi][4 * gq + 3] += c.f[3];
                        }
                    }
            }
        }
    }
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;
    // half-tiles stream in the order h = 4 t + {0: A early, 1: W early, 2: W late, 3: A late}

    // ---- my DMA pieces: every half-tile is 16 pieces of 8 rows; wave w moves pieces u = 2w, 2w + 1 of each -----------------
    // A piece u of (early | late): rows (u >> 3) * 128 + late * 64 + (u & 7) * 8 ..;  W piece u: rows (u >> 2) * 64 + late * 32 + (u & 3) * 8 ..
    // The DMAs are BUFFER loads (`buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds`): the per-lane byte offset of my
    // (row, swizzled chunk) is loop invariant, the k position is a scalar offset, rows beyond M / N carry an offset
    // beyond the descriptor's size and read as zeros -- no 64-bit address arithmetic, no zero page, no branch in the loop
    // (an LDS-DMA piece costs its wave 60-185 cycles of issue time; what surrounds it is what can be saved).
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.a, 0, (int)P.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((int64_t)P.N * P.K * 2), 0x00020000);
    unsigned a_vo[4];                                // [late * 2 + i]: byte offset of my row start (token) / pixel (conv) + chunk; OOB = row beyond M
    unsigned a_ok[4];                                // conv: bit t = tap t of my pixel lies inside the image; (ups == 1: (y << 16) | x instead)
    unsigned w_vo[4];
    int a_lds[4], w_lds[4];                          // wave-uniform LDS element offsets of the pieces inside a buffer
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int late = e >> 1, u = 2 * wave + (e & 1);
        const int rowa = (u >> 3) * 128 + late * 64 + (u & 7) * 8;
        a_lds[e] = rowa * BK;
        const int lr = rowa + prow;
        const int sc = swz<BK>(lr, pch);
        const int64_t m = m0 + lr;
        const bool ok = m < P.M;
        a_ok[e] = 0;
        if (MODE == 0) {
            a_vo[e] = ok ? (unsigned)((m * P.lda + sc * 8) * 2) : OOB;
        } else {
            const int64_t mm = ok ? m : 0;
            const int pix = (int)(mm % P.hw);
            const int py = pix / P.img_w, px = pix - py * P.img_w;
            const int64_t img = mm / P.hw;
            if (P.ups == 1) {                         // half-resolution source: the tap's source pixel is computed per tap
                a_vo[e] = ok ? (unsigned)((img * (int64_t)(P.hw >> 2) * P.cin + sc * 8) * 2) : OOB;
                a_ok[e] = ((unsigned)py << 16) | (unsigned)px;
            } else {
                unsigned mask = 0;
                for (int t = 0; t < 9; ++t) {
                    const int dy = t / 3 - 1, dx = t % 3 - 1;
                    const bool in = P.ups == 2 ? (2 * py + dy >= 0 && 2 * px + dx >= 0)      // (the high side is always inside)
                                               : ((unsigned)(py + dy) < (unsigned)P.img_h && (unsigned)(px + dx) < (unsigned)P.img_w);
                    mask |= (unsigned)in << t;
                }
                a_ok[e] = ok ? mask : 0u;
                if (P.ups == 2) a_vo[e] = (unsigned)(((img * (int64_t)(P.hw << 2) + (int64_t)(2 * py) * (2 * P.img_w) + 2 * px) * P.cin + sc * 8) * 2);
                else a_vo[e] = (unsigned)((mm * P.cin + sc * 8) * 2);
            }
        }
        const int roww = (u >> 2) * 64 + late * 32 + (u & 3) * 8;
        w_lds[e] = (BM + roww) * BK;
        const int lw = roww + prow;
        const int n = n0 + lw;
        w_vo[e] = n < P.N ? (unsigned)(((int64_t)n * P.K + swz<BK>(lw, pch) * 8) * 2) : OOB;
    }
    // the k-tile the NEXT issue belongs to (scalars, advanced after its fourth half-tile; past the end of the k range the stream
    // wraps to its first k-tile: the re-fetched data lands in a slot nobody reads any more, the addresses stay valid and the DMA
    // count exact)
    int it_kt = kt0, it_buf = 0, it_tap = 0, it_ci0 = 0;
    auto seek = [&](int kt) {                         // (conv: k-tile -> (tap, channel chunk); one division per segment)
        it_kt = kt;
        if (MODE == 1) {
            if (P.tap_outer) {
                it_tap = kt * BK / P.cin;
                it_ci0 = kt * BK - it_tap * P.cin;
            } else {
                const int chunk = kt / 9;
                it_tap = kt - chunk * 9;
                it_ci0 = chunk * BK;
            }
        }
    };
    seek(kt0);
    auto advance = [&]() {
        it_buf ^= 1;
        if (++it_kt == nk) {                          // past the end of my k range: wrap to its first k-tile
            seek(kt0);
        } else if (MODE == 1) {
            if (P.tap_outer) {
                it_ci0 += BK;
                if (it_ci0 == P.cin) { it_ci0 = 0; ++it_tap; }
            } else if (++it_tap == 9) {               // (channel chunk outer, tap inner), see gemm_kernel
                it_tap = 0;
                it_ci0 += BK;
            }
        }
    };
    auto dma = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, int soff, bf16_t* lds) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, soff, 0, 0);
    };
    auto issue = [&](int q) {                         // q: 0 = A early, 1 = W early, 2 = W late, 3 = A late (a constant at every call site)
        bf16_t* stage = smem + it_buf * STAGE_ELEMS;
        if (q == 1 || q == 2) {
            const int e0 = (q - 1) * 2;
            const int soff = (MODE == 1 ? it_tap * P.cin + it_ci0 : it_kt * BK) * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) dma(rsW, w_vo[e0 + i], soff, stage + w_lds[e0 + i]);
        } else {
            const int e0 = q == 0 ? 0 : 2;
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) dma(rsA, a_vo[e0 + i], it_kt * BK * 2, stage + a_lds[e0 + i]);
            } else {
                const int dy = (it_tap >= 3) + (it_tap >= 6) - 1, dx = it_tap - 3 * (dy + 1) - 1;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int e = e0 + i;
                    unsigned vo;
                    if (P.ups == 1) {                 // tap (y+dy, x+dx) of the upsampled image = source pixel (.. >> 1)
                        const int py = (int)(a_ok[e] >> 16) + dy, px = (int)(a_ok[e] & 0xffffu) + dx;
                        const bool in = (unsigned)py < (unsigned)P.img_h && (unsigned)px < (unsigned)P.img_w;
                        vo = a_vo[e] + (unsigned)(((py >> 1) * (P.img_w >> 1) + (px >> 1)) * P.cin * 2);
                        if (!in || a_vo[e] == OOB) vo = OOB;
                    } else {
                        const int shift = (P.ups == 2 ? (dy * 2 * P.img_w + dx) : (dy * P.img_w + dx)) * P.cin * 2;   // scalar, may be negative
                        vo = ((a_ok[e] >> it_tap) & 1u) ? a_vo[e] + (unsigned)shift : OOB;
                    }
                    dma(rsA, vo, it_ci0 * 2, stage + a_lds[e]);
                }
            }
        }
        if (q == 3) advance();
    };

    auto read_w = [&](const bf16_t* Ws, int ni) {
        const int rw = wc * 64 + ni * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            union { bf16x8 v; u32x4 u; } t;
            t.u = *reinterpret_cast<const u32x4*>(Ws + rw * BK + swz<BK>(rw, 2 * ks + half) * 8);
            wf[ni][ks] = t.v;
        }
    };
    auto read_a = [&](const bf16_t* As, int mh) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rm = wr * 128 + (2 * mh + j) * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                union { bf16x8 v; u32x4 u; } t;
                t.u = *reinterpret_cast<const u32x4*>(As + rm * BK + swz<BK>(rm, 2 * ks + half) * 8);
                af[j][ks] = t.v;
            }
        }
    };
#define G8_MMA(NI, MH)                                                                                               \
    do {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
                acc[NI][2 * (MH) + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[NI][ks], af[j][ks], acc[NI][2 * (MH) + j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#define G8_FEED(C)                                                                                                   \
    do {                                                                                                             \
        issue(((C) + PF) & 3);                                                                                       \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");                                                   \
    } while (0)

    if constexpr (ROLE != 2) {
    // ---- prologue: PF half-tiles in flight, the two that phase 0 reads retired and published -----------------------------------
#pragma unroll
    for (int h = 0; h < PF; ++h) issue(h & 3);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs one barrier behind wave row 0
    __builtin_amdgcn_sched_barrier(0);
    // (zeroed HERE, behind the scheduling fence: 128 live zeros across the segment set-up above is what the register allocator
    // answers with accumulator-tuple spills in the persistent instantiations)
    float zero_v = 0.f;
    if (SK) asm volatile("" : "+v"(zero_v));          // (a zero the optimiser cannot hoist out of the persistent loop as 8 constant tuples)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = zero_v;
    __builtin_amdgcn_sched_barrier(0);

    for (int kt = 0; kt < nk - kt0; ++kt) {
        const bf16_t* As = smem + (kt & 1) * STAGE_ELEMS;
        const bf16_t* Ws = As + BM * BK;
        read_w(Ws, 0);                                // phase 0: quadrant (rows 0-63, columns 0-31)
        read_a(As, 0);
        G8_FEED(0);
        G8_MMA(0, 0);
        read_w(Ws, 1);                                // phase 1: (rows 0-63, columns 32-63)
        G8_FEED(1);
        G8_MMA(1, 0);
        read_a(As, 1);                                // phase 2: (rows 64-127, columns 32-63)
        G8_FEED(2);
        G8_MMA(1, 1);
        G8_FEED(3);                                   // phase 3: (rows 64-127, columns 0-31): everything is in registers
        G8_MMA(0, 1);
    }
#undef G8_MMA
#undef G8_FEED
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the wrap-around DMAs of the tail have landed: LDS is free for the epilogue
    __syncthreads();

    }  // (ROLE 2 has no main loop: its accumulators are the sums of the stored partials)

    if (ROLE == 1 || (ROLE == 3 && (kt0 > 0 || nk < nkt))) {   // my segment's accumulators -> its workspace slot, accumulator layout
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    union { u32x4 u; f32x4 f; } t;
                    t.f = f32x4{acc[ni][mi][4 * gq], acc[ni][mi][4 * gq + 1], acc[ni][mi][4 * gq + 2], acc[ni][mi][4 * gq + 3]};
                    const int quad = (wave * 8 + ni * 4 + mi) * 4 + gq;
                    __builtin_amdgcn_raw_buffer_store_b128(t.u, rsP, (int)(((int64_t)sk_slot * (BM * BN / 4) + quad * 64 + lane) * 16), 0, 0);
                }
        continue;                                         // (the next segment's prologue only writes LDS, already released above)
    } else {
    if constexpr (ROLE == 0 && EPI != 2) {
        if (P.f32io) {                                    // fp32-storage mode: see epi_f32_quad
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int64_t m = m0 + wr * 128 + mi * 32 + l31;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    if (EPI == 1) {
                        const float av[4] = {acc[0][mi][4 * gq], acc[0][mi][4 * gq + 1], acc[0][mi][4 * gq + 2], acc[0][mi][4 * gq + 3]};
                        const float ag[4] = {acc[1][mi][4 * gq], acc[1][mi][4 * gq + 1], acc[1][mi][4 * gq + 2], acc[1][mi][4 * gq + 3]};
                        epi_f32_geglu_quad(P, av, ag, m, n0 + wc * 64 + 8 * gq + 4 * half, n0 / 2 + wc * 32 + 8 * gq + 4 * half);
                    } else {
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const float a4[4] = {acc[ni][mi][4 * gq], acc[ni][mi][4 * gq + 1], acc[ni][mi][4 * gq + 2], acc[ni][mi][4 * gq + 3]};
                            epi_f32_quad<MODE>(P, a4, m, n0 + wc * 64 + ni * 32 + 8 * gq + 4 * half);
                        }
                    }
                }
            }
            break;
        }
    }
    // ---- epilogue: the output tile is staged as bf16 in LDS, 128 rows (the same 64-row half of both wave rows) per pass,
    // and leaves with whole-row 16-byte stores.  Residuals travel through the same staging rows first (whole-row 16-byte
    // loads), every lane adds its own words to its accumulators in fp32: one rounding, as in gemm_kernel.
    constexpr int ON = EPI != 0 ? BN / 2 : BN;        // output columns of the tile in 2-byte units (EPI 2: two fp8 per unit)
    constexpr int OP = ON + 8;                        // bf16 pitch of the staging rows
    bf16_t* Os = smem;                                // [256][OP]
    constexpr int CPR = ON / 8;                       // 16-byte chunks per staged row
    constexpr int IT = 128 * CPR / NT;                // chunks per thread and pass
    const int no0 = EPI != 0 ? n0 / 2 : n0;           // first output column of the tile
    const int n_out = EPI != 0 ? P.N / 2 : P.N;
    auto pass = [&](auto hp_c) {
        constexpr int hp = decltype(hp_c)::value;     // compile-time: the accumulator blocks of a pass must be static indices
        auto stage_rows = [&](const bf16_t* src, int64_t ld) {     // global [rows of this pass][ON] -> Os
            // GR chunks per thread in flight at a time (512 threads x GR x 16 B per round): all IT at once cost 32 registers
            // next to the 128 accumulators, which is what pushed the persistent instantiation into scratch
            constexpr int GR = IT >= 4 ? 4 : IT;
#pragma unroll
            for (int g0 = 0; g0 < IT; g0 += GR) {
                u32x4 v[GR];
#pragma unroll
                for (int it = 0; it < GR; ++it) {
                    const int c = tid + (g0 + it) * NT, ri = c / CPR, ch = c - ri * CPR;
                    const int row = (ri >> 6) * 128 + hp * 64 + (ri & 63);
                    v[it] = *reinterpret_cast<const u32x4*>(src + min(m0 + row, P.M - 1) * ld + min(no0 + ch * 8, n_out - 8));
                }
#pragma unroll
                for (int it = 0; it < GR; ++it) {
                    const int c = tid + (g0 + it) * NT, ri = c / CPR, ch = c - ri * CPR;
                    const int row = (ri >> 6) * 128 + hp * 64 + (ri & 63);
                    *reinterpret_cast<u32x4*>(Os + row * OP + ch * 8) = v[it];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };
        if (EPI == 0) {
            // alpha * (acc + bias) (+ temb) in the accumulator registers; bias / temb words are fetched where they are used
            // (4 bf16 = one 8-byte load, L1 / L2 resident): holding all 32 bias values of a lane next to the 128 accumulators
            // is what tipped the stream-K instantiations into scratch
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int nb = n0 + wc * 64 + ni * 32 + 8 * gq + 4 * half;
                    float b4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (P.bias) {
                        const u32x2 t = *reinterpret_cast<const u32x2*>(P.bias + min(nb, P.N - 4));
                        b4[0] = __uint_as_float(t[0] << 16); b4[1] = __uint_as_float(t[0] & 0xffff0000u);
                        b4[2] = __uint_as_float(t[1] << 16); b4[3] = __uint_as_float(t[1] & 0xffff0000u);
                    }
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const int mi = 2 * hp + j2, row = wr * 128 + mi * 32 + l31;
                        float t4[4] = {0.f, 0.f, 0.f, 0.f};
                        if (MODE == 1 && P.temb) {
                            const bf16_t* trow = P.temb + ((min(m0 + row, P.M - 1) / P.hw) / P.temb_div) * P.temb_ld;
                            const u32x2 t = *reinterpret_cast<const u32x2*>(trow + min(nb, P.N - 4));
                            t4[0] = __uint_as_float(t[0] << 16); t4[1] = __uint_as_float(t[0] & 0xffff0000u);
                            t4[2] = __uint_as_float(t[1] << 16); t4[3] = __uint_as_float(t[1] & 0xffff0000u);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[ni][mi][4 * gq + j] = (acc[ni][mi][4 * gq + j] + b4[j]) * P.alpha + t4[j];
                    }
                }
            // residual(s): stage, add my words.  The additions are UNconditional (a missing residual contributes zero words):
            // accumulators modified inside an `if` leave two versions of them alive at the join, which the register
            // allocator resolves with copies and, in the persistent instantiation, spills
#pragma unroll
            for (int rz = 0; rz < 2; ++rz) {
                const bf16_t* rp = rz == 0 ? P.res : P.res2;
                const bool has = rp != nullptr;
                if (has) stage_rows(rp, P.ldres);
                const unsigned keep = has ? 0xffffffffu : 0u;
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    const int mi = 2 * hp + j2, row = wr * 128 + mi * 32 + l31;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            u32x2 t = *reinterpret_cast<const u32x2*>(Os + row * OP + wc * 64 + ni * 32 + 8 * gq + 4 * half);
                            t[0] &= keep; t[1] &= keep;
                            acc[ni][mi][4 * gq] += __uint_as_float(t[0] << 16); acc[ni][mi][4 * gq + 1] += __uint_as_float(t[0] & 0xffff0000u);
                            acc[ni][mi][4 * gq + 2] += __uint_as_float(t[1] << 16); acc[ni][mi][4 * gq + 3] += __uint_as_float(t[1] & 0xffff0000u);
                        }
                }
                if (has) __syncthreads();             // everybody has picked up its words: the rows may be overwritten
            }
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int mi = 2 * hp + j2, row = wr * 128 + mi * 32 + l31;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        *reinterpret_cast<u32x2*>(Os + row * OP + wc * 64 + ni * 32 + 8 * gq + 4 * half) =
                            u32x2{pack_bf2(acc[ni][mi][4 * gq], acc[ni][mi][4 * gq + 1]), pack_bf2(acc[ni][mi][4 * gq + 2], acc[ni][mi][4 * gq + 3])};
                    }
            }
        } else if (EPI == 2) {
            // fp8 (OCP e4m3) output for the fused q | k | v projection of the fp8 temporal attention: value / scale[block],
            // clamped to +-448, four consecutive columns of a lane = one 32-bit word.  A wave's 64 columns lie inside one of the
            // three blocks (C % 64 == 0), so the running |max| of the block costs one atomicMax per wave and pass.
            const int cq = P.N / 3;
            const int blk = min((n0 + wc * 64) / cq, 2);
            const float inv = P.q8_inv[blk];
            float amax = 0.f;
            unsigned char* Ob = reinterpret_cast<unsigned char*>(Os);
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int mi = 2 * hp + j2, row = wr * 128 + mi * 32 + l31;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = acc[ni][mi][4 * gq + j] * P.alpha;
                            amax = fmaxf(amax, fabsf(v));
                            o[j] = fminf(fmaxf(v * inv, -448.f), 448.f);
                        }
                        int w = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], 0, false);
                        w = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], w, true);
                        *reinterpret_cast<int*>(Ob + row * (OP * 2) + wc * 64 + ni * 32 + 8 * gq + 4 * half) = w;
                    }
            }
            if (P.q8_amax && n0 + wc * 64 < P.N) {
                // running max: only a wave that RAISES it touches the word (a relaxed read first): 20 000 unconditional
                // atomicMax on three addresses serialised in the L2 and cost the projection +50 us
                amax = wave_max(amax);
                if (lane == 0) {
                    const unsigned bits = __float_as_uint(amax);                      // (non-negative floats order like their bit patterns)
                    if (bits > __hip_atomic_load(P.q8_amax + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(P.q8_amax + blk, bits);
                }
            }
        } else {
            // GEGLU: weight rows interleaved per 64 -> acc[0] = value, acc[1] = gate of the SAME 32 output columns
            float ba[16], bg[16];
            const bool cols_ok = n0 + wc * 64 < P.N;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nb = n0 + wc * 64 + 8 * gq + 4 * half + j;
                    ba[4 * gq + j] = (P.bias && cols_ok) ? bf2f(P.bias[nb]) : 0.f;
                    bg[4 * gq + j] = (P.bias && cols_ok) ? bf2f(P.bias[nb + 32]) : 0.f;
                }
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int mi = 2 * hp + j2, row = wr * 128 + mi * 32 + l31;
                const f32x16& av = acc[0][mi];
                const f32x16& ag = acc[1][mi];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (av[4 * gq + j] + ba[4 * gq + j]) * gelu_erf(ag[4 * gq + j] + bg[4 * gq + j]);
                    *reinterpret_cast<u32x2*>(Os + row * OP + wc * 32 + 8 * gq + 4 * half) = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                }
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < IT; ++it) {
            const int c = tid + it * NT, ri = c / CPR, ch = c - ri * CPR;
            const int row = (ri >> 6) * 128 + hp * 64 + (ri & 63);
            const int64_t m = m0 + row;
            const int no = no0 + ch * 8;
            if (m < P.M && no < n_out)
                *reinterpret_cast<u32x4*>(P.out + m * P.ldo + no) = *reinterpret_cast<const u32x4*>(Os + row * OP + ch * 8);
        }
        // (the next pass touches the other 128 staging rows: no barrier needed between the passes)
    };
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
    if (ROLE != 3) break;
    __syncthreads();                                  // the staging rows alias the operand buffers of the next segment
    }  // (epilogue)
    }  // segment loop
}

// ---- GroupNorm statistics out of the producing GEMM / conv epilogue ----------------------------------------
// The consumer of most level-0 outputs is a GroupNorm(32 groups) over the same tensor; its first pass (`gn_partial_kernel`) re-reads the
// whole tensor only to sum x and x^2 per (image, group).  The 160 x 320 kernels have the finished bf16 tile in LDS anyway: thread t sums
// group t % GT (GT = 320 / cpg groups per tile) over rows t / GT, t / GT + RP, ... from the staged tile -- the ROUNDED values, exactly what
// gn_partial would read -- the 512 / GT partial pairs of a group are added in a fixed order, and the tile's pair lands in
// gn_part[image][tile of the image][group][2], the layout `gn_apply_fwd_kernel` combines in fp64 (deterministic: no atomics).
__device__ __forceinline__ void gn_tile_accumulate(const bf16_t* Os, int OP, int rows, int cpg, int tid, float& s, float& ss) {
    const int GT = 320 / cpg, gl = tid % GT, rp = tid / GT, RP = 512 / GT;
    for (int r = rp; r < rows; r += RP) {
        const unsigned* w = reinterpret_cast<const unsigned*>(Os + r * OP + gl * cpg);
        for (int k = 0; k < cpg / 2; ++k) {
            const unsigned u = w[k];
            const float a = __uint_as_float(u << 16), b = __uint_as_float(u & 0xffff0000u);
            s += a + b;
            ss += a * a + b * b;
        }
    }
}
__device__ __forceinline__ void gn_tile_finish(const GemmParams& P, float* red, int64_t m0, int n0, int cpg, int tid, float s, float ss) {
    const int GT = 320 / cpg, RP = 512 / GT;
    red[2 * tid] = s;
    red[2 * tid + 1] = ss;
    __syncthreads();
    if (tid < GT) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < RP; ++k) { a += red[2 * (tid + k * GT)]; b += red[2 * (tid + k * GT) + 1]; }
        const int64_t img = m0 / P.gn_hw;
        const int nsplit = P.gn_hw / 160, sp = (int)((m0 - img * P.gn_hw) / 160);
        float* dst = P.gn_part + ((img * nsplit + sp) * 32 + (n0 / cpg + tid)) * 2;
        dst[0] = a;
        dst[1] = b;
    }
}

// =====================================================================================================================
// The 160 x 320 kernel (C-ABI tile 16): the staggered-phase schedule of gemm8_kernel on a tile that DIVIDES the FMC problem sizes.
//
// Why: every token / pixel count of the U-Net is a multiple of 160 rows x 256 CUs at the two levels that hold 2/3 of the GEMM time
// (M = 81920 = 512 x 160 at 40x64, M = 20480 = 128 x 160 at 20x32), and every width there is a multiple of 320 (N = 320, 640, 960, 1920,
// 2560, 5120).  256 x 256 tiles pay for that twice -- padded columns (N = 320 runs as 512: 37 % of the MFMAs multiply zeros; 640 as 768)
// and tile quantisation (640 or 240 or 400 workgroups on 256 CUs: 78-94 % of a round is filled).  With 160 x 320 tiles the level-0
// launches are exactly 2 / 6 / 16 rounds of 256 workgroups and the level-1 launches exactly 1 / 3 / 8: no padded column, no partial round.
//
// Geometry: 8 waves = 2 (m) x 4 (n), a wave computes 80 x 80 outputs as 5 x 5 v_mfma_f32_16x16x32_bf16 accumulators (100 registers; 80
// is not a multiple of 32, hence the 16-wide MFMA; products are "swapped" as everywhere in this file: MFMA A operand = W rows, so a
// lane holds 4 consecutive output columns of one row).
//
// Schedule.  The reduction is cut into 32-deep SUB-TILES (one MFMA k-step): a sub-tile is 30 KiB in LDS (A 160 rows, W 320 rows of 64
// bytes), FIVE of them form a ring (150 KiB).  One phase per sub-tile:
//     {LOAD: the 10 ds_read_b128 of the sub-tile's fragments + the DMA of the sub-tile THREE ahead + `s_waitcnt vmcnt(8)`} | barrier |
//     {25 MFMAs under s_setprio 1} | barrier,
// the two wave rows one barrier apart, so on every SIMD one wave runs its 25 MFMAs (425 cycles) while the other loads.  (The first
// version used gemm8_kernel's 64-deep k-tiles with four 10 / 15-MFMA phases each: 2.1 us per 64-deep step = 32 % of the matrix pipe --
// the fixed cost of a phase (two barriers, an LDS round trip, a counted vmcnt) was paid twice as often as here.)  A sub-tile's DMA is 30
// one-KiB pieces (16 rows x 64 B; 4 instructions per wave, 2 dummies keep every wave's vmcnt arithmetic identical); sub-tile s is
// issued at LOAD(s - 3) into buffer s % 5, whose previous content (s - 5) was last read at LOAD(s - 5): two whole phases earlier for
// both wave rows (WAR safe with a phase to spare); `vmcnt(8)` at the end of LOAD(g) retires sub-tile g + 1, the barrier publishes it,
// LOAD(g + 1) reads it (a wave reads DMA'd data one phase after the wait that retired it).  Past the end of K the stream wraps to
// valid addresses so the count stays exact.
// What bounds the loop (measured, tools/scratch/probe_g160_dbg*.py + tools/ubench/cu_bw): the DMA instruction stream.  With the 32 DMA
// instructions per sub-tile and CU taken out, the conv 32x20x32 640->640 launch drops from 222 to 113 us; with them in but pointed out of
// range (no memory traffic at all) it stays at 194; the stream alone (no MFMA, no ds_read, no barrier, no wait) takes 221 us on 8
// workgroups and on 256.  So it is neither HBM / Infinity Cache / L2 bandwidth nor latency, but ~15 ns per `buffer_load ... lds` wave
// instruction and CU, serial to the matrix work.  Tried on top and dropped: a register-staged stream (global -> VGPR ring of 3
// sub-tiles -> ds_write_b128, 2 LDS buffers; latency-bound at the same ~1.2 us per sub-tile: 217 vs 198 us), half of the DMAs behind the
// wave's own MFMAs (no change), reduction start offsets skewed across the co-resident workgroups (no change).
// LDS image: 64-byte rows, 16-byte chunk c of row r at physical chunk c ^ (3 * ((r >> 3) & 1)): the four 16-lane groups of a
// ds_read_b128 of a 16x16x32 fragment (lane l: row l % 16, chunk l / 16) then hit 16 distinct 16-byte slots of the 256-byte bank row
// (checked against the lane-group table of MI355X_MICROARCH.md); the DMA applies the same involution to its SOURCE chunk.
// Conv mode walks (64-channel chunk, tap, 32-channel half): the two halves of a 128-byte line are fetched in consecutive phases.
// Epilogue: bias / alpha / temb in registers, residual(s) through the bf16 staging tile, whole-row 16-byte stores.  GEGLU: weight rows
// ordered [8 value | 8 gate] per 16 (layers.interleave_geglu(block=8)), so lane l holds values where lane l ^ 32 holds the gates of
// the same outputs; one v_permlane32_swap per register pair of two blocks gives every lane a (value, gate) pair -- all 64 lanes gate.
// =====================================================================================================================
__device__ __forceinline__ void lane32_swap(float& lo_keeps, float& hi_keeps) {
    // lanes 0-31 of `hi_keeps` <-> lanes 32-63 of `lo_keeps`:  afterwards lo_keeps = [lo_keeps.low | hi_keeps.low], hi_keeps = [lo_keeps.high | hi_keeps.high]
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_keeps), __float_as_uint(hi_keeps), false, false);
    lo_keeps = __uint_as_float(r[0]);
    hi_keeps = __uint_as_float(r[1]);
}

template <int MODE, int EPI>
__global__ __launch_bounds__(512, 2)
void gemm160_kernel(const GemmParams P) {
    constexpr int BM = 160, BN = 320, BK = 32, NT = 512, NBUF = 5;
    constexpr int SUB_ELEMS = (BM + BN) * BK;        // 30 KiB per sub-tile buffer
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int l15 = lane & 15, kq = lane >> 4;       // fragment row / 16-byte k chunk (output: column quad)
    const int prow = lane >> 2, pch = lane & 3;      // my row / physical chunk inside a 1-KiB DMA piece (16 rows x 64 B)
    const int psrc = pch ^ (3 * ((prow >> 3) & 1));  // the LOGICAL chunk that lives there

    int nks = P.K / BK, ks0 = 0;                      // my range of sub-tiles: [ks0, ks0 + nks)
    int tile_m, tile_n;
    const int split = P.split_k > 1 ? (int)(blockIdx.x % P.split_k) : 0;
    {
        const int total = P.tiles_m * P.tiles_n;
        int lin = blockIdx.x;
        if (P.split_k > 1) {                          // split-K: the splits of a tile are neighbours in launch order; fp32 partials -> P.ws,
            lin = blockIdx.x / P.split_k;             // splitk_reduce_kernel sums them in a fixed order and applies the epilogue
            const int per = ((nks + P.split_k - 1) / P.split_k + 1) & ~1;
            ks0 = split * per;
            nks = max(0, min(nks, ks0 + per) - ks0);
        } else {
            const int id = blockIdx.x, q = total >> 3, r = total & 7, x = id & 7;
            lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);   // XCD x owns a contiguous range
        }
        lin_to_tile(lin, P, tile_m, tile_n);
    }
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- my DMA pieces: entry i = 4 wave + e; i < 20: W piece i (rows 16 i ..), 20 <= i < 30: A piece i - 20, else a dummy -------------
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.a, 0, (int)P.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)((int64_t)P.N * P.K * 2), 0x00020000);
    unsigned e_vo[4], e_ok[4];
    int e_lds[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = 4 * wave + e;
        e_vo[e] = OOB; e_ok[e] = 0; e_lds[e] = NBUF * SUB_ELEMS;         // dummy: zeros into the scratch KiB behind the ring
        if (i < 20) {
            const int lw = 16 * i + prow, n = n0 + lw;
            e_lds[e] = (BM + 16 * i) * BK;
            e_vo[e] = n >= P.N ? OOB : (P.w_blocked ? (unsigned)((((int64_t)tile_n * (P.K / BK)) * (BN * BK) + lw * BK + psrc * 8) * 2)
                                                     : (unsigned)(((int64_t)n * P.K + psrc * 8) * 2));
        } else if (i < 30) {
            const int lr = 16 * (i - 20) + prow;
            e_lds[e] = 16 * (i - 20) * BK;
            const int64_t m = m0 + lr;
            const bool ok = m < P.M;
            if (MODE == 0) {
                e_vo[e] = ok ? (P.a_blocked ? (unsigned)((((int64_t)tile_m * (P.K / BK)) * (BM * BK) + lr * BK + psrc * 8) * 2)     // tile-major A: GemmParams
                                            : (unsigned)((m * P.lda + psrc * 8) * 2)) : OOB;
            } else {
                const int64_t mm = ok ? m : 0;
                const int pix = (int)(mm % P.hw);
                const int py = pix / P.img_w, px = pix - py * P.img_w;
                const int64_t img = mm / P.hw;
                if (P.ups == 1) {
                    e_vo[e] = ok ? (unsigned)((img * (int64_t)(P.hw >> 2) * P.cin + psrc * 8) * 2) : OOB;
                    e_ok[e] = ((unsigned)py << 16) | (unsigned)px;
                } else {
                    unsigned mask = 0;
                    for (int t = 0; t < 9; ++t) {
                        const int dy = t / 3 - 1, dx = t % 3 - 1;
                        const bool in = P.ups == 2 ? (2 * py + dy >= 0 && 2 * px + dx >= 0)
                                                   : ((unsigned)(py + dy) < (unsigned)P.img_h && (unsigned)(px + dx) < (unsigned)P.img_w);
                        mask |= (unsigned)in << t;
                    }
                    e_ok[e] = ok ? mask : 0u;
                    if (P.ups == 2) e_vo[e] = (unsigned)(((img * (int64_t)(P.hw << 2) + (int64_t)(2 * py) * (2 * P.img_w) + 2 * px) * P.cin + psrc * 8) * 2);
                    else e_vo[e] = (unsigned)((mm * P.cin + psrc * 8) * 2);
                }
            }
        }
    }
    const bool w_wave = wave < 5, a_first = wave >= 5;   // waves 0-4: four W pieces each; waves 5-7: A pieces (wave 7: two A + two dummies)
    // the sub-tile the NEXT issue belongs to (scalars).  Conv: (64-channel chunk, tap, 32-channel half)
    int it_s = 0, it_buf = 0, it_tap = 0, it_ci0 = 0, it_half = 0;
    auto seek0 = [&]() {                               // to the first sub-tile of my range
        it_s = ks0;
        if (MODE == 1) {
            const int chunk = ks0 / 18, rem = ks0 - chunk * 18;
            it_ci0 = chunk * 64; it_tap = rem >> 1; it_half = rem & 1;
        }
    };
    seek0();
    auto advance = [&]() {
        it_buf = it_buf + 1 == NBUF ? 0 : it_buf + 1;
        if (++it_s == ks0 + nks || nks == 0) {
            seek0();
        } else if (MODE == 1) {
            if (++it_half == 2) {
                it_half = 0;
                if (++it_tap == 9) { it_tap = 0; it_ci0 += 64; }
            }
        }
    };
    auto dma = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, int soff, bf16_t* lds) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, soff, 0, 0);
    };
    auto issue = [&]() {
        bf16_t* stage = smem + it_buf * SUB_ELEMS;
        const int kw = P.w_blocked ? it_s * (BN * BK * 2)                                            // the next 20-KiB block of my column tile
                                   : (MODE == 1 ? it_tap * P.cin + it_ci0 + it_half * 32 : it_s * BK) * 2;     // byte offset inside a W row
        if (w_wave) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dma(rsW, e_vo[e], kw, stage + e_lds[e]);
        } else {
            const int dy = (it_tap >= 3) + (it_tap >= 6) - 1, dx = it_tap - 3 * (dy + 1) - 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bf16_t* dst = e_lds[e] == NBUF * SUB_ELEMS ? smem + NBUF * SUB_ELEMS : stage + e_lds[e];
                if (MODE == 0) {
                    dma(rsA, e_vo[e], P.a_blocked ? it_s * (BM * BK * 2) : it_s * BK * 2, dst);
                } else {
                    unsigned vo;
                    if (P.ups == 1) {
                        const int py = (int)(e_ok[e] >> 16) + dy, px = (int)(e_ok[e] & 0xffffu) + dx;
                        const bool in = (unsigned)py < (unsigned)P.img_h && (unsigned)px < (unsigned)P.img_w;
                        vo = e_vo[e] + (unsigned)(((py >> 1) * (P.img_w >> 1) + (px >> 1)) * P.cin * 2);
                        if (!in || e_vo[e] == OOB) vo = OOB;
                    } else {
                        const int shift = (P.ups == 2 ? (dy * 2 * P.img_w + dx) : (dy * P.img_w + dx)) * P.cin * 2;
                        vo = ((e_ok[e] >> it_tap) & 1u) ? e_vo[e] + (unsigned)shift : OOB;
                    }
                    dma(rsA, vo, (it_ci0 + it_half * 32) * 2, dst);
                }
            }
        }
        advance();
    };
    (void)a_first;

    f32x4 acc[5][5];                                   // [mb][nb]: 4 consecutive n (registers) of row m = l15
    bf16x8 wf[5], af[5];
    // fragment address of (block row 0, my lane): row l15, logical chunk kq
    const int frag_off = l15 * BK + (kq ^ (3 * ((l15 >> 3) & 1))) * 8;
    auto read_frags = [&](const bf16_t* sub) {
        const bf16_t* Ws = sub + BM * BK + (wc * 80) * BK + frag_off;
        const bf16_t* As = sub + (wr * 80) * BK + frag_off;
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) {
            union { bf16x8 v; u32x4 u; } t;
            t.u = *reinterpret_cast<const u32x4*>(Ws + nb * 16 * BK);
            wf[nb] = t.v;
        }
#pragma unroll
        for (int mb = 0; mb < 5; ++mb) {
            union { bf16x8 v; u32x4 u; } t;
            t.u = *reinterpret_cast<const u32x4*>(As + mb * 16 * BK);
            af[mb] = t.v;
        }
    };

    // ---- prologue: sub-tiles 0, 1, 2 in flight, sub-tile 0 retired and published -----------------------------------------------------
    issue(); issue(); issue();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs one barrier behind wave row 0
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);

    int rbuf = 0;
    for (int g = 0; g < nks; ++g) {
        read_frags(smem + rbuf * SUB_ELEMS);
        rbuf = rbuf + 1 == NBUF ? 0 : rbuf + 1;
        issue();                                      // sub-tile g + 3
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mb = 0; mb < 5; ++mb)
#pragma unroll
            for (int nb = 0; nb < 5; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb], af[mb], acc[mb][nb], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the wrap-around DMAs of the tail have landed: LDS is free for the epilogue
    __syncthreads();

    // ---- epilogue --------------------------------------------------------------------------------------------------------------------
    // my outputs: acc[mb][nb][j] = (row m0 + 80 wr + 16 mb + l15, tile column 80 wc + 16 nb + 4 kq + j)
    if (EPI == 0 && P.split_k > 1) {                  // raw fp32 partial sums of my k range
#pragma unroll
        for (int mb = 0; mb < 5; ++mb) {
            const int64_t m = m0 + wr * 80 + mb * 16 + l15;
            if (m >= P.M) continue;
#pragma unroll
            for (int nb = 0; nb < 5; ++nb)
                *reinterpret_cast<f32x4*>(P.ws + ((int64_t)split * P.M + m) * P.N + n0 + wc * 80 + nb * 16 + 4 * kq) = acc[mb][nb];
        }
        return;
    }
    if (EPI == 1) {
        // GEGLU.  Weight rows per 16: [8 value | 8 gate] -> lanes with kq < 2 hold values of outputs 4 kq .. 4 kq + 3 of the block, lanes with
        // kq >= 2 (= lane + 32) the gates of the same outputs.  Blocks are paired (nb 0-1, 2-3 of a row block; the nb = 4 blocks of row
        // blocks 0-1, 2-3; (4, 4) alone): after the half-wave swap the low lanes hold (value, gate) of the first block of a pair, the high
        // lanes of the second.  Output column of a block: 40 wc + 8 nb + 4 (kq & 1) inside the tile's 160.
        const bool hi = lane >= 32;
        const int oq = 4 * (kq & 1);
        auto gate_pair = [&](f32x4& x, f32x4& y, int mbx, int nbx, int mby, int nby, bool single) {
            // x, y: two blocks; returns in x the gated outputs of (hi ? block y : block x), row / column through the out-params below
#pragma unroll
            for (int j = 0; j < 4; ++j) {                                 // x = [x.value | y.value], y = [x.gate | y.gate]
                float xa = x[j], ya = y[j];
                lane32_swap(xa, ya);
                x[j] = xa; y[j] = ya;
            }
            const int nbm = hi ? nby : nbx;
            const int tn = wc * 80 + nbm * 16;                            // first weight row of my block inside the tile
            float o[4];
            if (P.f32io) {
                const float* bias = reinterpret_cast<const float*>(P.bias);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gg = y[j] + (bias ? bias[n0 + tn + 8 + oq + j] : 0.f);
                    o[j] = (x[j] + (bias ? bias[n0 + tn + oq + j] : 0.f)) * (0.5f * gg * (1.f + erff(gg * 0.70710678118654752f)));
                }
            } else {
                float bv[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
                if (P.bias) {
                    const u32x2 t = *reinterpret_cast<const u32x2*>(P.bias + n0 + tn + oq);
                    const u32x2 u = *reinterpret_cast<const u32x2*>(P.bias + n0 + tn + 8 + oq);
                    bv[0] = __uint_as_float(t[0] << 16); bv[1] = __uint_as_float(t[0] & 0xffff0000u);
                    bv[2] = __uint_as_float(t[1] << 16); bv[3] = __uint_as_float(t[1] & 0xffff0000u);
                    bg[0] = __uint_as_float(u[0] << 16); bg[1] = __uint_as_float(u[0] & 0xffff0000u);
                    bg[2] = __uint_as_float(u[1] << 16); bg[3] = __uint_as_float(u[1] & 0xffff0000u);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (x[j] + bv[j]) * gelu_erf(y[j] + bg[j]);
            }
            const int mbm = hi ? mby : mbx;
            const int row = wr * 80 + mbm * 16 + l15, col = wc * 40 + nbm * 8 + oq;      // inside the tile's [160][160] outputs
            if (single && hi) return;                                     // (the unpaired block: the high lanes hold nothing)
            if (P.f32io) {
                const int64_t m = m0 + row;
                if (m < P.M) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(P.out) + m * P.ldo + n0 / 2 + col) = f32x4{o[0], o[1], o[2], o[3]};
            } else {
                *reinterpret_cast<u32x2*>(smem + row * 168 + col) = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
            }
        };
#pragma unroll
        for (int mb = 0; mb < 5; ++mb) {
            gate_pair(acc[mb][0], acc[mb][1], mb, 0, mb, 1, false);
            gate_pair(acc[mb][2], acc[mb][3], mb, 2, mb, 3, false);
        }
        gate_pair(acc[0][4], acc[1][4], 0, 4, 1, 4, false);
        gate_pair(acc[2][4], acc[3][4], 2, 4, 3, 4, false);
        {
            f32x4 none = f32x4{0.f, 0.f, 0.f, 0.f};
            gate_pair(acc[4][4], none, 4, 4, 4, 4, true);
        }
        if (P.f32io) return;
        __syncthreads();
        for (int c = tid; c < BM * 20; c += NT) {                          // 160 rows x 20 chunks of 8 outputs
            const int r = c / 20, ch = c - r * 20;
            const int64_t m = m0 + r;
            if (m < P.M)
                *reinterpret_cast<u32x4*>(P.out + m * P.ldo + n0 / 2 + ch * 8) = *reinterpret_cast<const u32x4*>(smem + r * 168 + ch * 8);
        }
        return;
    }
    if (P.f32io) {                                    // fp32-storage mode: see epi_f32_quad
#pragma unroll
        for (int mb = 0; mb < 5; ++mb) {
            const int64_t m = m0 + wr * 80 + mb * 16 + l15;
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                const float a4[4] = {acc[mb][nb][0], acc[mb][nb][1], acc[mb][nb][2], acc[mb][nb][3]};
                epi_f32_quad<MODE>(P, a4, m, n0 + wc * 80 + nb * 16 + 4 * kq);
            }
        }
        return;
    }
    {
        constexpr int OP = BN + 8;                         // bf16 pitch of the staging rows (656 B: 16-byte aligned)
        bf16_t* Os = smem;                                 // [160][OP] = 104 960 B
        constexpr int CPR = BN / 8;                        // 40 chunks per row
        // alpha * (acc + bias) (+ temb) in the accumulator registers; the five bias words in one burst (a load inside each block's `if (P.bias)` was
        // followed by its own s_waitcnt vmcnt(0): five dependent round trips per tile)
        u32x2 bt[5];
        if (P.bias) {
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) bt[nb] = *reinterpret_cast<const u32x2*>(P.bias + n0 + wc * 80 + nb * 16 + 4 * kq);
        }
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) {
            const int n = n0 + wc * 80 + nb * 16 + 4 * kq;
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (P.bias) {
                const u32x2 t = bt[nb];
                b4[0] = __uint_as_float(t[0] << 16); b4[1] = __uint_as_float(t[0] & 0xffff0000u);
                b4[2] = __uint_as_float(t[1] << 16); b4[3] = __uint_as_float(t[1] & 0xffff0000u);
            }
#pragma unroll
            for (int mb = 0; mb < 5; ++mb) {
                float t4[4] = {0.f, 0.f, 0.f, 0.f};
                if (MODE == 1 && P.temb) {
                    const int64_t m = min(m0 + wr * 80 + mb * 16 + l15, P.M - 1);
                    const u32x2 t = *reinterpret_cast<const u32x2*>(P.temb + ((m / P.hw) / P.temb_div) * P.temb_ld + n);
                    t4[0] = __uint_as_float(t[0] << 16); t4[1] = __uint_as_float(t[0] & 0xffff0000u);
                    t4[2] = __uint_as_float(t[1] << 16); t4[3] = __uint_as_float(t[1] & 0xffff0000u);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mb][nb][j] = (acc[mb][nb][j] + b4[j]) * P.alpha + t4[j];
            }
        }
        // residual(s): whole-row 16-byte loads into the staging tile, every lane adds its own words in fp32
#pragma unroll 1
        for (int rz = 0; rz < 2; ++rz) {
            const bf16_t* rp = rz == 0 ? P.res : P.res2;
            if (rp == nullptr) continue;                   // (uniform)
            // four bursts of 4 loads per thread (160 x 40 chunks = 12.5 per thread) instead of a rolled load -> wait -> ds_write loop: thirteen dependent
            // round trips per residual became four
#pragma unroll 1
            for (int h = 0; h < 4; ++h) {                  // (4 x 4: sixteen staging registers fit beside the accumulators; 2 x 7 spilled 43)
                u32x4 rv[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int c = min(tid + (4 * h + it) * NT, BM * CPR - 1), r = c / CPR, ch = c - r * CPR;
                    rv[it] = *reinterpret_cast<const u32x4*>(rp + min(m0 + r, P.M - 1) * P.ldres + n0 + ch * 8);
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int c = tid + (4 * h + it) * NT, r = c / CPR, ch = c - r * CPR;
                    if (c < BM * CPR) *reinterpret_cast<u32x4*>(Os + r * OP + ch * 8) = rv[it];
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
        float gs = 0.f, gss = 0.f;
        if (P.gn_part) gn_tile_accumulate(Os, OP, BM, P.N / 32, tid, gs, gss);
        for (int c = tid; c < BM * CPR; c += NT) {
            const int r = c / CPR, ch = c - r * CPR;
            const int64_t m = m0 + r;
            if (m < P.M)
                *reinterpret_cast<u32x4*>(P.out + m * P.ldo + n0 + ch * 8) = *reinterpret_cast<const u32x4*>(Os + r * OP + ch * 8);
        }
        if (P.gn_part) gn_tile_finish(P, reinterpret_cast<float*>(smem_raw + (size_t)BM * OP * 2), m0, n0, P.N / 32, tid, gs, gss);
    }
}

// =====================================================================================================================
// gemm160p_kernel: the PERSISTENT form of gemm160_kernel for the token projections (MODE 0, M % 160 == 0, more tiles than CUs).
//
// At K = 320 / 640 a 160 x 320 tile is 10 / 20 sub-tiles of main loop (5-11 us) between a prologue that waits for the first operands
// (~1.5-2 us: nothing to compute yet) and an epilogue (gate / residual / stage / store: 3-5 us) during which nothing is loaded: a third
// of the launch.  Here one workgroup per CU walks tiles id, id + CUs, ... and the operand stream never stops at a tile boundary: the
// DMA of the NEXT tile's first two sub-tiles is issued in the last two phases of the current tile and lands under its epilogue.  For
// that the epilogue stages through its own LDS region (a three-buffer ring of 30-KiB sub-tiles + 54 KiB of staging = 145 KiB) -- the
// ring is never drained.
// vmcnt bookkeeping across the boundary (stores count in vmcnt on gfx9-class hardware): every thread issues EXACTLY S_EPI global stores
// per tile (a thread without a last chunk repeats its previous one: same bytes, same address), so in the first phase after an epilogue
// `vmcnt(S_EPI + 4)` retires the sub-tile that was requested BEFORE the stores without waiting for their acknowledgements; one phase
// later the plain `vmcnt(4)` does wait for them, a whole phase after they were issued.
// Ring: lead 2 (sub-tile g + 2 is requested at LOAD(g) into the buffer LOAD(g - 1) read; every LOAD ends with lgkmcnt(0) in front of
// its barrier, so those reads -- of both wave rows -- are complete).  Everything else (geometry, staggered wave rows, LDS image, GEGLU by
// half-wave swap) is gemm160_kernel's.
// =====================================================================================================================
// MB = 16-row blocks per wave row: 5 = the 160 x 320 tile above; 8 = a 256 x 320 tile (GEGLU only, arm 17) -- a wave then issues 5 operand
// requests for 40 MFMAs per sub-tile instead of 4 for 25: a `buffer_load ... lds` blocks its wave for ~140 ns whatever else the CU does
// (tools/ubench/dma_mfma), so requests per MFMA is what this loop's speed follows.
// LN = 1 (plain epilogue, N == 320: the tile holds WHOLE output rows): the epilogue also writes the consumer's LayerNorm of the rows it
// just produced, from the bf16-ROUNDED rows in the staging tile: statistics by 4 lanes per row (16-byte reads, one pass, two shuffles),
// `(v - mean) * rstd * gamma + (beta + pe row)` in place by (8-column chunk, row group) threads that keep their gamma / beta in registers,
// whole-row stores.  (Statistics and normalisation straight from the accumulator registers -- the first form -- cost 110-200 spilled
// registers, reloaded inside the main loop.)  The separate LayerNorm launch and its read of the tensor disappear.
// A2 = 1 (plain epilogue, tile-major A): the reduction has a SECOND segment -- columns [ksplit, K) of A are rows of the row-major tensor a2 (lda2 apart).
// The feed-forward's output projection and the transformer's proj_out as ONE product: out = [gated | h] [Wp W2 | Wp]^T + b' + x (hip_ops.ff_tail); the
// stream's A requests of a sub-tile past ksplit / 32 go to a2 with ONE per-lane offset (row in piece, 16-byte chunk) and the piece / k position in the
// scalar offset, so the variant costs one register over the plain one.
// IMG = 1 (plain epilogue): per-image weights and fp32 bias rows, see GemmParams::bias_img.
template <int EPI, int MB, int LN = 0, int LNC = 0, int A2 = 0, int IMG = 0>
__global__ __launch_bounds__(512, 2)
void gemm160p_kernel(const GemmParams P) {
    constexpr int BM = 32 * MB, BN = 320, BK = 32, NT = 512, NBUF = 3;
    constexpr int NE = (20 + BM / 16 + 7) / 8;       // operand requests per wave and sub-tile (20 W pieces + BM / 16 A pieces + dummies)
    static_assert(20 % NE == 0, "a wave's requests are all W or all A");
    constexpr int SUB_ELEMS = (BM + BN) * BK;        // 30 (36) KiB per sub-tile buffer
    constexpr int S_EPI = MB == 5 ? (EPI == 1 ? 7 : (LN == 1 ? 28 : (LN == 2 ? 16 : 14))) : 10;   // global stores per thread and tile (see the header)
    static_assert(!LN || (EPI == 0 && MB == 5), "LayerNorm rides in the plain epilogue of the 160-row tile");
    static_assert(MB == 5 || (MB == 8 && EPI == 1), "the 256-row tile has the GEGLU epilogue only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    auto lds0 = (__attribute__((address_space(3))) unsigned char*)smem_raw;
    bf16_t* Os = smem + NBUF * SUB_ELEMS + 512;      // staging: [80][328] (plain epilogue, per pass) or [160][168] (GEGLU)
    // LN: [gamma 320 | beta + pe row 320 | (mean, rstd) x 80 rows]; LNC: [c 320 | bias' 320 | (mean, rstd) x BM rows]  (behind the staging tile;
    // shares the GroupNorm scratch)
    float* lnS = reinterpret_cast<float*>(Os + (EPI == 1 ? (MB == 5 ? 160 : 128) * 168 : 80 * (BN + 8)));
    if (LN == 1 && threadIdx.x < 320) lnS[threadIdx.x] = P.ln_gamma[threadIdx.x];   // (published by the main loop's barriers long before the first epilogue)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int l15 = lane & 15, kq = lane >> 4;
    const int prow = lane >> 2, pch = lane & 3;
    const int psrc = pch ^ (3 * ((prow >> 3) & 1));
    const int nks = P.K / BK;
    const int nksA = A2 ? P.ksplit / BK : nks;       // sub-tiles (= 10-KiB blocks per row tile) of the first A segment
    const int total = P.tiles_m * P.tiles_n;
    auto tile_of = [&](int id, int& tm, int& tn) {
        const int q = total >> 3, r = total & 7, x = id & 7;
        const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);   // XCD x owns a contiguous range
        lin_to_tile(lin, P, tm, tn);
    };

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.a, 0, (int)P.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)(IMG ? (P.M / P.img_rows) * P.w_img_stride * 2 : (int64_t)P.N * P.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(A2 ? P.a2 : P.a), 0, A2 ? (int)(((P.M - 1) * P.lda2 + (P.K - P.ksplit)) * 2) : 0, 0x00020000);
    const unsigned lane_a2 = A2 ? (unsigned)((prow * P.lda2 + psrc * 8) * 2) : 0u;
    int s_m0 = 0;                                     // first row of the stream's current tile (A2), < 0 past the last tile
    // ---- the operand stream: entry i = 4 wave + e; i < 20: W piece i, 20 <= i < 30: A piece i - 20, else a dummy ---------------------
    unsigned e_vo[NE];
    int e_lds[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int i = NE * wave + e;
        e_lds[e] = i < 20 ? (BM + 16 * i) * BK : (i < 20 + BM / 16 ? 16 * (i - 20) * BK : NBUF * SUB_ELEMS);
    }
    const bool w_wave = wave < 20 / NE;
    int s_tile = blockIdx.x, it_s = 0, it_buf = 0;
    auto stream_setup = [&]() {                       // per-lane source offsets of the stream's current tile (or out of range past the last)
        int tm = 0, tn = 0;
        const bool live = s_tile < total;
        if (live) tile_of(s_tile, tm, tn);
        if (A2) s_m0 = live ? tm * BM : -1;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = NE * wave + e;
            unsigned vo = OOB;
            if (live && i < 20) {
                const int n = tn * BN + 16 * i + prow;
                vo = P.w_blocked ? (unsigned)((((int64_t)tn * nks) * (BN * BK) + (16 * i + prow) * BK + psrc * 8) * 2)
                                 : (unsigned)(((int64_t)n * P.K + psrc * 8) * 2);
                if (IMG) vo += (unsigned)(((int64_t)tm * BM / P.img_rows) * P.w_img_stride * 2);       // this tile's image's weight
            } else if (live && i < 20 + BM / 16) {
                const int64_t m = (int64_t)tm * BM + 16 * (i - 20) + prow;
                vo = P.a_blocked ? (unsigned)((((int64_t)tm * nksA) * (BM * BK) + (16 * (i - 20) + prow) * BK + psrc * 8) * 2)
                                 : (unsigned)((m * P.lda + psrc * 8) * 2);
            }
            e_vo[e] = vo;
        }
    };
    auto issue = [&]() {
        const int soff = w_wave ? (P.w_blocked ? it_s * (BN * BK * 2) : it_s * BK * 2)          // (tile-major operands: the next sub-tile is the next block)
                                : (P.a_blocked ? it_s * (BM * BK * 2) : it_s * BK * 2);
        if (A2 && !w_wave && it_s >= nksA) {          // second A segment: piece i - 20 of rows s_m0 ... of a2, k position (it_s - nksA) * 32   (wave-uniform branch)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int i = NE * wave + e;
                const bool real = s_m0 >= 0 && i < 20 + BM / 16;
                const int dst = e_lds[e] == NBUF * SUB_ELEMS ? NBUF * SUB_ELEMS : it_buf * SUB_ELEMS + e_lds[e];
                const int so2 = real ? (int)((((int64_t)s_m0 + 16 * (i - 20)) * P.lda2 + (it_s - nksA) * BK) * 2) : 0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (__attribute__((address_space(3))) void*)(lds0 + 2 * dst), 16, real ? (int)lane_a2 : (int)OOB, so2, 0, 0);
            }
        } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) {                // (LDS destinations as wave-uniform 32-bit offsets: they travel through M0)
            const int dst = e_lds[e] == NBUF * SUB_ELEMS ? NBUF * SUB_ELEMS : it_buf * SUB_ELEMS + e_lds[e];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_wave ? rsW : rsA, (__attribute__((address_space(3))) void*)(lds0 + 2 * dst), 16, (int)e_vo[e], soff, 0, 0);
        }
        }
        it_buf = it_buf + 1 == NBUF ? 0 : it_buf + 1;
        if (++it_s == nks) {
            it_s = 0;
            s_tile += gridDim.x;
            stream_setup();
        }
    };

    f32x4 acc[MB][5];
    bf16x8 wf[5], af[MB];
    const int frag_off = l15 * BK + (kq ^ (3 * ((l15 >> 3) & 1))) * 8;
    auto read_frags = [&](const bf16_t* sub) {
        const bf16_t* Ws = sub + BM * BK + (wc * 80) * BK + frag_off;
        const bf16_t* As = sub + (wr * 16 * MB) * BK + frag_off;
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) {
            union { bf16x8 v; u32x4 u; } t;
            t.u = *reinterpret_cast<const u32x4*>(Ws + nb * 16 * BK);
            wf[nb] = t.v;
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            union { bf16x8 v; u32x4 u; } t;
            t.u = *reinterpret_cast<const u32x4*>(As + mb * 16 * BK);
            af[mb] = t.v;
        }
    };

    // ---- prologue: sub-tiles 0, 1 of my first tile requested, sub-tile 0 retired and published ------------------------------------------
    stream_setup();
    issue(); issue();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NE) : "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs one barrier behind wave row 0
    int rbuf = 0;
    bool after_epi = false;                           // S_EPI stores sit between the two youngest sub-tile requests and the next one
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        int tile_m, tile_n;
        tile_of(tile, tile_m, tile_n);
        const int64_t m0 = (int64_t)tile_m * BM;
        const int n0 = tile_n * BN;
        // LNC: this tile's row statistics and column constants are requested NOW (three loads per thread, every lane active so that all
        // waves count alike) and parked in LDS at the head of the epilogue -- read there from global memory they were two exposed round trips per tile
        f32x2_t pre_st = f32x2_t{0.f, 0.f};
        float pre_c = 0.f, pre_b = 0.f;
        if (LNC) {
            pre_st = *reinterpret_cast<const f32x2_t*>(P.lnc_stats + (m0 + tid % BM) * 2);
            pre_c = P.lnc_c[n0 + tid % BN];
            pre_b = P.lnc_bias[n0 + tid % BN];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
        for (int g = 0; g < nks; ++g) {
            read_frags(smem + rbuf * SUB_ELEMS);
            rbuf = rbuf + 1 == NBUF ? 0 : rbuf + 1;
            issue();                                  // sub-tile g + 2 (of this tile or the next)
            if (g == 0 && after_epi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S_EPI + NE + 3 * LNC) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NE) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < 5; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb], af[mb], acc[mb][nb], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();    // the wave rows meet for the epilogue
        __builtin_amdgcn_sched_barrier(0);
        if (LNC) {                                    // lnS: [c 320 | bias' 320 | (mean, rstd) x BM]
            if (tid < BM) *reinterpret_cast<f32x2_t*>(lnS + 640 + 2 * tid) = pre_st;
            if (tid < BN) { lnS[tid] = pre_c; lnS[320 + tid] = pre_b; }
            __syncthreads();
        }

        // ---- epilogue through Os (the ring keeps streaming) ------------------------------------------------------------------------------
        if (EPI == 1) {
            const bool hi = lane >= 32;
            const int oq = 4 * (kq & 1);
            float lmu[MB], lrs[MB];                                      // (mean, rstd) of my rows when this GEMM applies its input's LayerNorm
            if (LNC) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const f32x2_t t = *reinterpret_cast<const f32x2_t*>(lnS + 640 + (wr * 16 * MB + mb * 16 + l15) * 2);
                    lmu[mb] = t[0]; lrs[mb] = t[1];
                }
            }
            auto gate_pair = [&](f32x4& x, f32x4& y, int mbx, int nbx, int mby, int nby, bool single) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {             // x = [x.value | y.value], y = [x.gate | y.gate]
                    float xa = x[j], ya = y[j];
                    lane32_swap(xa, ya);
                    x[j] = xa; y[j] = ya;
                }
                const int nbm = hi ? nby : nbx, mbm = hi ? mby : mbx;
                const int tn = wc * 80 + nbm * 16;
                float bv[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
                if (LNC) {                        // the LayerNorm of my A rows, applied here (W carries gamma): see GemmParams
                    const float mu = hi ? lmu[mby] : lmu[mbx], rs = hi ? lrs[mby] : lrs[mbx];
                    const f32x4 cv = *reinterpret_cast<const f32x4*>(lnS + tn + oq), cg = *reinterpret_cast<const f32x4*>(lnS + tn + 8 + oq);
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(lnS + 320 + tn + oq), b1 = *reinterpret_cast<const f32x4*>(lnS + 320 + tn + 8 + oq);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { x[j] = rs * (x[j] - mu * cv[j]); y[j] = rs * (y[j] - mu * cg[j]); bv[j] = b0[j]; bg[j] = b1[j]; }
                } else
                if (P.bias) {
                    const u32x2 t = *reinterpret_cast<const u32x2*>(P.bias + n0 + tn + oq);
                    const u32x2 u = *reinterpret_cast<const u32x2*>(P.bias + n0 + tn + 8 + oq);
                    bv[0] = __uint_as_float(t[0] << 16); bv[1] = __uint_as_float(t[0] & 0xffff0000u);
                    bv[2] = __uint_as_float(t[1] << 16); bv[3] = __uint_as_float(t[1] & 0xffff0000u);
                    bg[0] = __uint_as_float(u[0] << 16); bg[1] = __uint_as_float(u[0] & 0xffff0000u);
                    bg[2] = __uint_as_float(u[1] << 16); bg[3] = __uint_as_float(u[1] & 0xffff0000u);
                }
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (x[j] + bv[j]) * gelu_erf(y[j] + bg[j]);
                if (single && hi) return;
                *reinterpret_cast<u32x2*>(Os + ((MB == 5 ? wr * 80 : 0) + mbm * 16 + l15) * 168 + wc * 40 + nbm * 8 + oq) = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
            };
            auto gate_all = [&]() {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    gate_pair(acc[mb][0], acc[mb][1], mb, 0, mb, 1, false);
                    gate_pair(acc[mb][2], acc[mb][3], mb, 2, mb, 3, false);
                }
#pragma unroll
                for (int mb = 0; mb + 1 < MB; mb += 2) gate_pair(acc[mb][4], acc[mb + 1][4], mb, 4, mb + 1, 4, false);
                if (MB & 1) {
                    f32x4 none = f32x4{0.f, 0.f, 0.f, 0.f};
                    gate_pair(acc[MB - 1][4], none, MB - 1, 4, MB - 1, 4, true);
                }
            };
            if constexpr (MB == 5) {
                gate_all();
                __syncthreads();
#pragma unroll
                for (int it = 0; it < S_EPI; ++it) {                     // 160 rows x 20 chunks = 3200 = 6.25 per thread: the 7th repeats the 6th
                    int c = tid + it * NT;
                    if (c >= BM * 20) c -= NT;
                    // row-major: chunk c = (row, 8-column chunk).  out_blocked (tile-major output for the feed-forward's second GEMM): my
                    // 160 x 160 tile = 5 blocks [160 rows][32] of 10 KiB, block kb of the whole tensor at (m0 / 160) (N / 64) + n0 / 64 + kb;
                    // chunk c = (kb, row, 8-column quarter) in storage order.  One store either way (offsets selected, no divergent code).
                    const int kb = c / 640, rem = c - kb * 640;
                    const int rB = rem >> 2, r = P.out_blocked ? rB : c / 20;
                    const int col = P.out_blocked ? kb * 32 + (rem & 3) * 8 : (c - (c / 20) * 20) * 8;
                    const int64_t off = P.out_blocked ? (((m0 / BM) * (int64_t)(P.N / 64) + n0 / 64 + kb) * BM + rB) * 32 + (rem & 3) * 8
                                                      : (m0 + r) * P.ldo + n0 / 2 + col;
                    *reinterpret_cast<u32x4*>(P.out + off) = *reinterpret_cast<const u32x4*>(Os + r * 168 + col);
                }
                __syncthreads();                                         // Os is free again
            } else {
#pragma unroll 1
                for (int pass = 0; pass < 2; ++pass) {                   // one wave row (128 rows) per pass through the [128][168] staging tile
                    if (wr == pass) gate_all();
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < S_EPI / 2; ++it) {             // 128 rows x 20 chunks = 2560 = 5 per thread
                        const int c = tid + it * NT;
                        const int r = c / 20, ch = c - r * 20;
                        *reinterpret_cast<u32x4*>(P.out + (m0 + pass * 128 + r) * P.ldo + n0 / 2 + ch * 8) = *reinterpret_cast<const u32x4*>(Os + r * 168 + ch * 8);
                    }
                    __syncthreads();                                     // Os is free again
                }
            }
        } else if constexpr (MB == 5) {
            constexpr int OP = BN + 8, CPR = BN / 8;                     // staging rows of 328 bf16, 40 chunks per row
            if (LN == 1 && tid < 320)                                    // beta + the positional-encoding row of this tile's frame (tile-uniform)
                lnS[320 + tid] = P.ln_beta[tid] + (P.ln_pe ? P.ln_pe[(size_t)((m0 / P.ln_pe_inner) % P.ln_pe_frames) * 320 + tid] : 0.f);
            if (LNC) {                                           // the LayerNorm of my A rows, applied here (W carries gamma): see GemmParams
                float mu[5], rs[5];
#pragma unroll
                for (int mb = 0; mb < 5; ++mb) {
                    const f32x2_t t = *reinterpret_cast<const f32x2_t*>(lnS + 640 + (wr * 80 + mb * 16 + l15) * 2);
                    mu[mb] = t[0]; rs[mb] = t[1];
                }
#pragma unroll
                for (int nb = 0; nb < 5; ++nb) {
                    const int n = n0 + wc * 80 + nb * 16 + 4 * kq;
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(lnS + n - n0), b4 = *reinterpret_cast<const f32x4*>(lnS + 320 + n - n0);
#pragma unroll
                    for (int mb = 0; mb < 5; ++mb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[mb][nb][j] = rs[mb] * (acc[mb][nb][j] - mu[mb] * c4[j]) + b4[j];
                }
            } else {
            // alpha * (acc + bias) in the accumulator registers; the five bias words in one burst where the registers allow it (LN == 0: the
            // LayerNorm-writing variants sit at the 256-register limit and keep the load beside its use)
            u32x2 bt[5];
            if (LN == 0 && !IMG && P.bias) {
#pragma unroll
                for (int nb = 0; nb < 5; ++nb) bt[nb] = *reinterpret_cast<const u32x2*>(P.bias + n0 + wc * 80 + nb * 16 + 4 * kq);
            }
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                const int n = n0 + wc * 80 + nb * 16 + 4 * kq;
                float b4[4] = {0.f, 0.f, 0.f, 0.f};
                if (IMG) {                                               // fp32 bias row of this tile's image
                    const f32x4 t = *reinterpret_cast<const f32x4*>(P.bias_img + (m0 / P.img_rows) * P.N + n);
                    b4[0] = t[0]; b4[1] = t[1]; b4[2] = t[2]; b4[3] = t[3];
                } else
                if (P.bias) {
                    u32x2 t;
                    if (LN == 0) t = bt[nb];
                    else t = *reinterpret_cast<const u32x2*>(P.bias + n);
                    b4[0] = __uint_as_float(t[0] << 16); b4[1] = __uint_as_float(t[0] & 0xffff0000u);
                    b4[2] = __uint_as_float(t[1] << 16); b4[3] = __uint_as_float(t[1] & 0xffff0000u);
                }
#pragma unroll
                for (int mb = 0; mb < 5; ++mb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mb][nb][j] = (acc[mb][nb][j] + b4[j]) * P.alpha;
            }
            }
            float gs = 0.f, gss = 0.f;
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {                       // rows [80 pass, 80 pass + 80): wave row `pass`
#pragma unroll 1
                for (int rz = 0; rz < 2; ++rz) {
                    const bf16_t* rp = rz == 0 ? P.res : P.res2;
                    if (rp == nullptr) continue;                       // (uniform)
                    {   // the 80 residual rows in ONE burst of 7 loads per thread (80 x 40 chunks = 6.25 per thread: the 7th repeats the 6th): rolled, the loop
                        // was load -> s_waitcnt vmcnt(0) -> ds_write per iteration, seven dependent HBM round trips per wave row and residual
                        constexpr int RB = LN == 1 ? 2 : 7;              // (the LayerNorm-writing variant has 2 free registers: 7 in flight spilled 18, 4 spilled 5)
#pragma unroll
                        for (int h = 0; h < (7 + RB - 1) / RB; ++h) {
                            u32x4 rv[RB];
#pragma unroll
                            for (int it = 0; it < RB; ++it) {
                                int c = tid + min(h * RB + it, 6) * NT;
                                if (c >= 80 * CPR) c -= NT;
                                const int r = c / CPR, ch = c - r * CPR;
                                rv[it] = *reinterpret_cast<const u32x4*>(rp + (m0 + pass * 80 + r) * P.ldres + n0 + ch * 8);
                            }
#pragma unroll
                            for (int it = 0; it < RB; ++it) {
                                int c = tid + min(h * RB + it, 6) * NT;
                                if (c >= 80 * CPR) c -= NT;
                                const int r = c / CPR, ch = c - r * CPR;
                                *reinterpret_cast<u32x4*>(Os + r * OP + ch * 8) = rv[it];
                            }
                        }
                    }
                    __syncthreads();
                    if (wr == pass) {
#pragma unroll
                        for (int mb = 0; mb < 5; ++mb)
#pragma unroll
                            for (int nb = 0; nb < 5; ++nb) {
                                const u32x2 t = *reinterpret_cast<const u32x2*>(Os + (mb * 16 + l15) * OP + wc * 80 + nb * 16 + 4 * kq);
                                acc[mb][nb][0] += __uint_as_float(t[0] << 16); acc[mb][nb][1] += __uint_as_float(t[0] & 0xffff0000u);
                                acc[mb][nb][2] += __uint_as_float(t[1] << 16); acc[mb][nb][3] += __uint_as_float(t[1] & 0xffff0000u);
                            }
                    }
                    __syncthreads();
                }
                if (wr == pass) {
#pragma unroll
                    for (int mb = 0; mb < 5; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 5; ++nb)
                            *reinterpret_cast<u32x2*>(Os + (mb * 16 + l15) * OP + wc * 80 + nb * 16 + 4 * kq) =
                                u32x2{pack_bf2(acc[mb][nb][0], acc[mb][nb][1]), pack_bf2(acc[mb][nb][2], acc[mb][nb][3])};
                }
                __syncthreads();
                if (P.gn_part) gn_tile_accumulate(Os, OP, 80, P.N / 32, tid, gs, gss);
#pragma unroll
                for (int it = 0; it < 7; ++it) {                         // 80 rows x 40 chunks = 3200 = 6.25 per thread: the 7th repeats the 6th
                    int c = tid + it * NT;
                    if (c >= 80 * CPR) c -= NT;
                    const int r = c / CPR, ch = c - r * CPR;
                    *reinterpret_cast<u32x4*>(P.out + (m0 + pass * 80 + r) * P.ldo + n0 + ch * 8) = *reinterpret_cast<const u32x4*>(Os + r * OP + ch * 8);
                }
                if (LN) {
                    // (1) statistics of the 80 staged (rounded) rows: 4 lanes per row, 16-byte reads, one pass (sum, sum of squares), two shuffles
                    if (tid < 320) {
                        const int r = tid >> 2, q = tid & 3;
                        // (centred variance, as the stand-alone LayerNorm kernel and torch compute it: E[x^2] - mean^2 loses the variance of rows with
                        //  |mean| >> std, and which path a row takes depends on the autotuned arm.  The row is read from the staging tile twice: held in
                        //  registers between the passes it costs 40 VGPRs next to the other wave row's live accumulators -- 375 spilled registers.)
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int i = 0; i < 10; ++i) {
                            const u32x4 x4 = *reinterpret_cast<const u32x4*>(Os + r * OP + (q + 4 * i) * 8);
#pragma unroll
                            for (int j = 0; j < 4; ++j) s1 += __uint_as_float(x4[j] << 16) + __uint_as_float(x4[j] & 0xffff0000u);
                        }
                        s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
                        const float mean = s1 * (1.f / 320.f);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 10; ++i) {
                            const u32x4 x4 = *reinterpret_cast<const u32x4*>(Os + r * OP + (q + 4 * i) * 8);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float lo = __uint_as_float(x4[j] << 16) - mean, hi = __uint_as_float(x4[j] & 0xffff0000u) - mean;
                                s2 += lo * lo + hi * hi;
                            }
                        }
                        s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
                        if (q == 0) *reinterpret_cast<f32x2_t*>(lnS + 640 + 2 * r) = f32x2_t{mean, rsqrtf(s2 * (1.f / 320.f) + P.ln_eps)};
                    }
                    __syncthreads();                                     // (also: every thread's copy of the rows to `out` has left the staging tile)
                    if (LN == 2) {                                       // the consumer GEMM normalises: it only needs (mean, rstd) per row; one store per thread
                        const int r = tid % 80;
                        *reinterpret_cast<f32x2_t*>(P.ln_stats + (m0 + pass * 80 + r) * 2) = *reinterpret_cast<const f32x2_t*>(lnS + 640 + 2 * r);
                    }
                    // (2) normalise in place: thread = (chunk column, row group) -- gamma / beta' of its 8 columns in registers, rows rg, rg + 12, ...
                    if (LN == 1 && tid < 480) {
                        const int ch = tid % 40, rg = tid / 40;
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(lnS + ch * 8), g1 = *reinterpret_cast<const f32x4*>(lnS + ch * 8 + 4);
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(lnS + 320 + ch * 8), b1 = *reinterpret_cast<const f32x4*>(lnS + 324 + ch * 8);
#pragma unroll 1
                        for (int r = rg; r < 80; r += 12) {
                            u32x4* px = reinterpret_cast<u32x4*>(Os + r * OP + ch * 8);
                            const u32x4 x4 = *px;
                            const f32x2_t st = *reinterpret_cast<const f32x2_t*>(lnS + 640 + 2 * r);
                            const float m = st[0], rs = st[1];
                            u32x4 o4;
                            o4[0] = pack_bf2((__uint_as_float(x4[0] << 16) - m) * rs * g0[0] + b0[0], (__uint_as_float(x4[0] & 0xffff0000u) - m) * rs * g0[1] + b0[1]);
                            o4[1] = pack_bf2((__uint_as_float(x4[1] << 16) - m) * rs * g0[2] + b0[2], (__uint_as_float(x4[1] & 0xffff0000u) - m) * rs * g0[3] + b0[3]);
                            o4[2] = pack_bf2((__uint_as_float(x4[2] << 16) - m) * rs * g1[0] + b1[0], (__uint_as_float(x4[2] & 0xffff0000u) - m) * rs * g1[1] + b1[1]);
                            o4[3] = pack_bf2((__uint_as_float(x4[3] << 16) - m) * rs * g1[2] + b1[2], (__uint_as_float(x4[3] & 0xffff0000u) - m) * rs * g1[3] + b1[3]);
                            *px = o4;
                        }
                    }
                    if (LN == 1) __syncthreads();
                    // (3) whole-row stores of the LayerNorm output
#pragma unroll
                    for (int it = 0; it < (LN == 1 ? 7 : 0); ++it) {
                        int c = tid + it * NT;
                        if (c >= 80 * CPR) c -= NT;
                        const int r = c / CPR, ch = c - r * CPR;
                        *reinterpret_cast<u32x4*>(P.ln_out + (m0 + pass * 80 + r) * (int64_t)BN + ch * 8) = *reinterpret_cast<const u32x4*>(Os + r * OP + ch * 8);
                    }
                }
                __syncthreads();                                         // Os is free again
            }
            if (P.gn_part) {
                gn_tile_finish(P, reinterpret_cast<float*>(Os + 80 * OP), m0, n0, P.N / 32, tid, gs, gss);
                __syncthreads();                                         // (the reduction scratch is re-used by the next tile)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (wr == 1) __builtin_amdgcn_s_barrier();    // and part again for the next tile's phases
        after_epi = true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the requests past my last tile: out of range, zeros into the scratch KiB / dead buffers)
}

int gemm_geometry_override() {   // FMC_GEMM_TILE = 0 (caller's choice) | 1..10: see fmc_hip.h
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FMC_GEMM_TILE");
        v = e ? atoi(e) : 0;
    }
    return v;
}

int gemm_group_m_override() {   // FMC_GEMM_GM: m-tiles per group of the tile order (1 = row-major); unset = per launch
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("FMC_GEMM_GM");
        v = e ? atoi(e) : -1;
    }
    return v;
}

// second pass of a split-K launch: out = alpha * (sum_s ws[s] + bias) (+ temb) (+ residual), fixed summation order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams P) {
    const int cpr = P.N / 8;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= P.M * cpr) return;
    const int64_t m = idx / cpr;
    const int n = (int)(idx - m * cpr) * 8;
    float v[8];
    Vec8<float>::load(P.ws + m * P.N + n, v);
    for (int sp = 1; sp < P.split_k; ++sp) {
        float t[8];
        Vec8<float>::load(P.ws + ((int64_t)sp * P.M + m) * P.N + n, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += t[k];
    }
    if (P.f32io) {                                        // fp32-storage mode: fp32 bias / temb / residual(s) / out
        const float* bias = reinterpret_cast<const float*>(P.bias);
        const float* temb = reinterpret_cast<const float*>(P.temb);
        const float* res = reinterpret_cast<const float*>(P.res);
        const float* res2 = reinterpret_cast<const float*>(P.res2);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = (v[k] + (bias ? bias[n + k] : 0.f)) * P.alpha;
            if (temb) v[k] += temb[((m / P.hw) / P.temb_div) * P.temb_ld + n + k];
            if (res) v[k] += res[m * P.ldres + n + k];
            if (res2) v[k] += res2[m * P.ldres + n + k];
        }
        Vec8<float>::store(reinterpret_cast<float*>(P.out) + m * P.ldo + n, v);
        return;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (v[k] + (P.bias ? bf2f(P.bias[n + k]) : 0.f)) * P.alpha;
    if (P.temb) {
        float t[8];
        Vec8<bf16_t>::load(P.temb + ((m / P.hw) / P.temb_div) * P.temb_ld + n, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += t[k];
    }
    if (P.res) {
        float t[8];
        Vec8<bf16_t>::load(P.res + m * P.ldres + n, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += t[k];
        if (P.res2) {
            Vec8<bf16_t>::load(P.res2 + m * P.ldres + n, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += t[k];
        }
    }
    Vec8<bf16_t>::store(P.out + m * P.ldo + n, v);
}

// finishing pass of the k-lockstep split (split_k == -3 / <= -16) on the 8-phase kernel: out = epilogue(sum over the sk_lock chunk slots of a tile).
// The slots are in ACCUMULATOR layout (gemm8_kernel: slot[wave][ni * 4 + mi][gq][lane] x 16 B); a 256-thread block owns one 32 x 32 block (wave, ni, mi) of
// one tile, thread = (l31, gq, half) with (gq, half) fastest: a wave's 16-byte loads cover eight whole 128-byte lines of every slot, its 8-byte stores
// eight 64-byte row segments.  One launch of tiles x 64 small blocks instead of one 512-thread workgroup per tile (100 of 256 CUs on the 10x16-level
// convolutions, each walking 1.3 MB of partials on its own): the pass is a plain bandwidth kernel.  Summation order: chunk 0, 1, ... (fixed).
template <int MODE>
__global__ __launch_bounds__(256) void sk_finish_kernel(const GemmParams P) {
    const int blk = blockIdx.x & 63, t = blockIdx.x >> 6;
    const int T = P.tiles_m * P.tiles_n;
    const int tile_n = t / P.tiles_m, tile_m = t - tile_n * P.tiles_m;
    const int wave = blk >> 3, ni = (blk >> 2) & 1, mi = blk & 3;
    const int tid = threadIdx.x, half = tid & 1, gq = (tid >> 1) & 3, l31 = tid >> 3;
    const int quad = (wave * 8 + ni * 4 + mi) * 4 + gq;
    const float* src = P.ws + ((int64_t)t * (256 * 256 / 4) + quad * 64 + half * 32 + l31) * 4;
    const int64_t cstride = (int64_t)T * 256 * 256;
    f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll 4
    for (int c = 1; c < P.sk_lock; ++c) v += *reinterpret_cast<const f32x4*>(src + c * cstride);
    const int64_t m = (int64_t)tile_m * 256 + (wave >> 2) * 128 + mi * 32 + l31;
    const int n = tile_n * 256 + (wave & 3) * 64 + ni * 32 + 8 * gq + 4 * half;
    if (m >= P.M || n >= P.N) return;
    auto up4 = [](const bf16_t* p, float (&o)[4]) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(p);
        o[0] = __uint_as_float(w[0] << 16); o[1] = __uint_as_float(w[0] & 0xffff0000u);
        o[2] = __uint_as_float(w[1] << 16); o[3] = __uint_as_float(w[1] & 0xffff0000u);
    };
    float o[4] = {v[0], v[1], v[2], v[3]}, a[4];
    if (P.bias) {
        up4(P.bias + n, a);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += a[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] *= P.alpha;
    if (MODE == 1 && P.temb) {
        up4(P.temb + ((m / P.hw) / P.temb_div) * P.temb_ld + n, a);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += a[j];
    }
    if (P.res) {
        up4(P.res + m * P.ldres + n, a);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += a[j];
        if (P.res2) {
            up4(P.res2 + m * P.ldres + n, a);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += a[j];
        }
    }
    *reinterpret_cast<u32x2*>(P.out + m * P.ldo + n) = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
}

template <int MODE, int EPI, int WM, int WN, int MI, int BK, int STAGES, bool SK>
void launch_gemm_k(GemmParams& P, unsigned grid, size_t lds, hipStream_t st) {
    static size_t raised = 0;                            // (the LDS size of one instantiation depends on the epilogue variant)
    if (lds > raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<MODE, EPI, WM, WN, MI, BK, STAGES, SK>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        raised = lds;
    }
    hipLaunchKernelGGL((gemm_kernel<MODE, EPI, WM, WN, MI, BK, STAGES, SK>), dim3(grid), dim3(64 * WM * WN), lds, st, P);
}

template <int MODE, int EPI, int WM, int WN, int BK, int STAGES, int MI = 2>
void launch_gemm_g(GemmParams& P, hipStream_t st) {
    constexpr int BM = 32 * MI * WM, BN = 64 * WN;
    P.tiles_m = (int)((P.M + BM - 1) / BM);
    P.tiles_n = (P.N + BN - 1) / BN;
    P.group_m = 1;
    {
        static const int tap_outer = getenv("FMC_CONV_TAP_OUTER") ? atoi(getenv("FMC_CONV_TAP_OUTER")) : 0;
        P.tap_outer = tap_outer;
        static const int regepi = getenv("FMC_GEMM_REGEPI") ? atoi(getenv("FMC_GEMM_REGEPI")) : 1;
        P.regepi = regepi;
    }
    size_t lds = (size_t)STAGES * (BM + BN) * BK * sizeof(bf16_t) + 1024;
    const size_t slab = (size_t)64 * (BN + 8) * sizeof(float);
    if (slab > lds) lds = slab;
    // the plain grid stages the whole output tile as bf16 (in-register epilogues); stream-K and split-K keep the fp32 slab
    const size_t staged = (size_t)BM * ((EPI == 1 ? BN / 2 : BN) + 8) * sizeof(bf16_t);
    const size_t lds_plain = (P.split_k == 1 && (EPI == 1 || !P.res2) && staged > lds) ? staged : lds;
    unsigned grid = (unsigned)(P.tiles_m * P.tiles_n * P.split_k);
    // co-resident workgroups per CU (LDS / wave-slot limited)
    constexpr int by_waves = 16 / (WM * WN) > 0 ? 16 / (WM * WN) : 1;
    const int by_lds = (int)(160 * 1024 / (P.sk ? lds : lds_plain));
    const int per_cu = by_lds < by_waves ? (by_lds < 1 ? 1 : by_lds) : by_waves;
    // Tile order (lin_to_tile).  Row-major is optimal while the weight matrix fits an XCD's L2 or the c workgroups an
    // XCD runs at once already span all n-tiles; otherwise make their footprint square in bytes: gm x c/gm tiles with
    // gm * BM = (c / gm) * BN.  Measured (tools/pmc_tile_order.sh): L2-miss bytes of the level-1 / level-2 GEGLU
    // projections 810 -> 172 MB and 1015 -> 221 MB per launch, launch time -4 .. -8 %; convs (3-5 n-tiles, k-lockstep
    // already shares W) only lost L2 hits to grouping, so they keep row-major.
    if (gemm_group_m_override() > 0) {
        P.group_m = gemm_group_m_override();
    } else if (MODE == 0 && (int64_t)P.N * P.K * 2 > (int64_t)5 << 19) {
        const double c = fmin(32.0 * per_cu, (double)P.tiles_m * P.tiles_n / 8.0);
        const int gm = (int)lround(sqrt(c * BN / BM));
        if (gm > 1 && P.tiles_n * 2 > 3 * (c / gm)) P.group_m = gm;
    }
    if (P.sk) {
        // persistent grid: exactly the co-resident workgroups, a multiple of 8
        const int64_t iters = (int64_t)P.tiles_m * P.tiles_n * (P.K / BK);
        const int g = (fmc_cu_count() * per_cu) & ~7;
        const int64_t need = (int64_t)g * BM * BN * (int64_t)sizeof(float) + 4096;
        if (STAGES != 2 || MI != 2 || g < 8 || iters < 4 * (int64_t)g || P.sk_ws_bytes < need || g * (int)sizeof(int) > 4096) {
            P.sk = 0;                                    // too little work (or workspace): the plain grid
        } else {
            P.sk = g;
            grid = (unsigned)g;
        }
    }
    if constexpr (STAGES == 2 && MI == 2) {
        if (P.sk) {
            launch_gemm_k<MODE, EPI, WM, WN, MI, BK, STAGES, true>(P, grid, lds, st);
            return;
        }
    }
    launch_gemm_k<MODE, EPI, WM, WN, MI, BK, STAGES, false>(P, grid, lds_plain, st);
    if (P.split_k > 1) {
        const int64_t chunks = P.M * (P.N / 8);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, P);
    }
}

// the 8-phase 256x256 kernel: plain grid or stream-K (one persistent workgroup per CU); offsets of A and W must fit 32 bits
template <int MODE, int EPI, int PF>
void launch_gemm8(GemmParams& P, hipStream_t st) {
    P.tiles_m = (int)((P.M + 255) / 256);
    P.tiles_n = (P.N + 255) / 256;
    P.group_m = 1;
    static const int tap_outer = getenv("FMC_CONV_TAP_OUTER") ? atoi(getenv("FMC_CONV_TAP_OUTER")) : 0;
    P.tap_outer = tap_outer;
    if (gemm_group_m_override() > 0) {
        P.group_m = gemm_group_m_override();
    } else if (MODE == 0 && (int64_t)P.N * P.K * 2 > (int64_t)5 << 19) {     // as launch_gemm_g: square per-XCD footprint
        const double c = fmin(32.0, (double)P.tiles_m * P.tiles_n / 8.0);
        const int gm = (int)lround(sqrt(c));
        if (gm > 1 && P.tiles_n * 2 > 3 * (c / gm)) P.group_m = gm;
    }
    constexpr size_t ring = (size_t)2 * 512 * 64 * sizeof(bf16_t) + 1024;
    constexpr size_t staged = (size_t)256 * ((EPI != 0 ? 128 : 256) + 8) * sizeof(bf16_t);
    constexpr size_t lds = staged > ring ? staged : ring;
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8_kernel<MODE, EPI, PF, 0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8_kernel<MODE, EPI, PF, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8_kernel<MODE, EPI, PF, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        raised = true;
    }
    const int tiles = P.tiles_m * P.tiles_n;
    P.sk_t0 = 0;
    if (P.sk && P.sk_hybrid && !P.sk_lock) {
        // hybrid: the whole rounds on the plain grid, only the last partial round through stream-K -- the partial traffic (2 x 256 KiB per
        // segment) and the finishing launch then cover tiles % CUs tiles instead of all of them
        const int g = fmc_cu_count() & ~7;
        const int t0 = g >= 8 ? (tiles / g) * g : 0, rem = tiles - t0;
        const int64_t iters = (int64_t)rem * (P.K / 64);
        const int maxseg = (rem + g - 1) / (g > 0 ? g : 1) + 2;
        const int64_t need = (int64_t)g * maxseg * 256 * 256 * (int64_t)sizeof(float) + 4096;
        if (t0 > 0 && rem > 0 && iters >= 2 * (int64_t)g && P.sk_ws_bytes >= need && need < ((int64_t)1 << 31)) {
            P.sk = g;
            P.sk_maxseg = maxseg;
            P.sk_whole = 0;
            P.sk_t0 = t0;
            hipLaunchKernelGGL((gemm8_kernel<MODE, EPI, PF, 1>), dim3((unsigned)g), dim3(512), lds, st, P);
            hipLaunchKernelGGL((gemm8_kernel<MODE, EPI, PF, 0>), dim3((unsigned)t0), dim3(512), lds, st, P);
            hipLaunchKernelGGL((gemm8_kernel<MODE, EPI, PF, 2>), dim3((unsigned)rem), dim3(512), lds, st, P);
            return;
        }
        if (rem == 0) P.sk = 0;                           // whole rounds only: the plain grid (else: fewer tiles than CUs -> full stream-K below)
    }
    if (P.sk && P.sk_lock) {
        // k-lockstep split: S chunks per tile, unit = (chunk, tile); S = the caller's (split_k <= -16) or the cheapest by a small model:
        // rounds x (chunk length x 1.9 us per 64-deep k-tile + 4 us of prologue / partial store) + 0.06 us of finishing traffic per unit
        // (constants fitted to tools/scratch/r04/probe_lock.py: profiles/r04_probe_lock.txt)
        const int g = fmc_cu_count() & ~7, nkt = P.K / 64;
        int S = P.sk_lock > 1 ? P.sk_lock : 0;
        if (!S) {
            double best = 1e30;
            for (int s2 = 2; s2 <= 16 && s2 <= nkt / 4; ++s2) {
                const int L = (nkt + s2 - 1) / s2;
                if ((s2 - 1) * L >= nkt) continue;
                const int rounds = (tiles * s2 + g - 1) / (g > 0 ? g : 1);
                const double cost = rounds * (L * 1.9 + 4.0) + tiles * s2 * 0.06;
                if (cost < best) { best = cost; S = s2; }
            }
        }
        const int L = S ? (nkt + S - 1) / S : 0;
        const int64_t need = (int64_t)tiles * S * 256 * 256 * (int64_t)sizeof(float) + 4096;
        if (g >= 8 && S >= 2 && (S - 1) * L < nkt && P.sk_ws_bytes >= need && need < ((int64_t)1 << 31)) {
            P.sk = g;
            P.sk_lock = S;
            P.sk_lock_len = L;
            hipLaunchKernelGGL((gemm8_kernel<MODE, EPI, PF, 1>), dim3((unsigned)g), dim3(512), lds, st, P);
            if constexpr (EPI == 0) hipLaunchKernelGGL((sk_finish_kernel<MODE>), dim3((unsigned)tiles * 64), dim3(256), 0, st, P);
            else hipLaunchKernelGGL((gemm8_kernel<MODE, EPI, PF, 2>), dim3((unsigned)tiles), dim3(512), lds, st, P);
            return;
        }
        P.sk_lock = 0;                                    // too little work / workspace: plain stream-K rules below
    }
    if (P.sk) {
        // stream-K = persistent main kernel (one 128-KiB workgroup per CU, partials only) + one finishing workgroup per tile
        const int64_t iters = (int64_t)tiles * (P.K / 64);
        const int g = fmc_cu_count() & ~7;
        const int maxseg = (tiles + g - 1) / g + 2;       // a range of ceil(T / g) tiles' worth of k-tiles touches at most this many tiles
        const int64_t need = (int64_t)g * maxseg * 256 * 256 * (int64_t)sizeof(float) + 4096;
        if (g >= 8 && iters >= 2 * (int64_t)g && P.sk_ws_bytes >= need && need < ((int64_t)1 << 31)) {
            P.sk = g;
            P.sk_maxseg = maxseg;
            P.sk_whole = 0;
            hipLaunchKernelGGL((gemm8_kernel<MODE, EPI, PF, 1>), dim3((unsigned)g), dim3(512), lds, st, P);
            hipLaunchKernelGGL((gemm8_kernel<MODE, EPI, PF, 2>), dim3((unsigned)tiles), dim3(512), lds, st, P);
            return;
        }
        P.sk = 0;                                         // too little work (or workspace): the plain grid
    }
    hipLaunchKernelGGL((gemm8_kernel<MODE, EPI, PF, 0>), dim3((unsigned)tiles), dim3(512), lds, st, P);
}

// =====================================================================================================================
// W-stationary persistent kernel for the K = 320 token projections (arm 15; MODE 0, plain epilogue, N % 320 == 0).
//
// Why: at the 40x64 level (M = 81920) a projection with K = 320 is five k-tiles per output tile.  Knock-outs of the ring kernels
// on M = 81920, N = K = 320 (tools/ubench/gemm_bench): 36 us as shipped, 12.5 us with the main loop removed (= the output stores:
// 52 MB at write bandwidth), i.e. 23.5 us of main loop for 11 us of MFMA work and a 52 MB read -- every tile pays its prologue,
// five DMA round trips and its epilogue one after the other, and the co-resident workgroups of a CU run in phase.
// Here a workgroup of 5 waves owns ALL 320 columns of a column block and keeps the block's weights in REGISTERS for its whole
// life (wave w: columns 64 w .. 64 w + 63, 20 k-steps x 2 fragments = 160 VGPRs); it walks its share of the 64-row A tiles:
//   * the A tile (64 x 320, 40 KB) arrives by buffer_load ... lds in ONE burst, two tiles ahead, double-buffered;
//   * 80 MFMAs per wave and tile straight from LDS fragments against the resident weights -- no W traffic at all after the prologue;
//   * the epilogue goes through one bf16 staging tile that the residual rows were DMA'd into during the main loop (every lane
//     adds its own words in fp32 and overwrites them), and leaves with whole-row 16-byte stores;
//   * ONE `s_waitcnt vmcnt(0)` per tile, placed right after the MFMAs: by then the next A tile, this tile's residual rows and the
//     previous tile's stores have had a whole main loop to finish, so it does not stall -- and it needs no assumption about the
//     order in which loads and stores retire.
// Loads of tile t+2 / t+1 and the stores of tile t are all in flight under the MFMAs of tile t+1: the kernel runs at the speed of
// its bytes.
// =====================================================================================================================
constexpr int K320_K = 320;
// MI x NI 32x32 blocks per wave: the tile is 32 MI rows x 160 NI columns (5 waves side by side).  (2, 2) = 64 x 320: A is read once per 320
// columns, one workgroup per CU; (1, 1) = 32 x 160: 80 weight registers per wave, three workgroups per CU (15 waves on 4 SIMDs)
template <int MI, int NI> constexpr int k320_stage_chunks() { return ((32 * MI * ((160 * NI + 8) / 8) + 63) / 64) * 64; }   // staging tile in 16-B chunks, whole DMA pieces
template <int MI, int NI> constexpr size_t k320_lds() {
    return (size_t)2 * 32 * MI * K320_K * 2 + (size_t)k320_stage_chunks<MI, NI>() * 16 + 160 * NI * sizeof(float);
}

template <int MI, int NI, bool HAS_RES>
__global__ __launch_bounds__(320, (MI * NI == 1 ? 3 : (MI * NI == 2 ? 2 : 1))) void gemm_k320_kernel(const GemmParams P) {
    constexpr int K = K320_K, KS = K / 16, BM = 32 * MI, BN = 160 * NI, OP = BN + 8, CPR = K / 8, OCPR = OP / 8, WC = 32 * NI;
    constexpr int A_ELEMS = BM * K, A_PIECES = BM * CPR / 64, ST_PIECES = k320_stage_chunks<MI, NI>() / 64;
    constexpr int A_PPW = (A_PIECES + 4) / 5, ST_PPW = (ST_PIECES + 4) / 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* Abuf = reinterpret_cast<bf16_t*>(smem_raw);                         // [2][BM][K], chunk-swizzled rows
    bf16_t* Os = Abuf + 2 * A_ELEMS;                                             // [BM][OP] staging: residual in, output out
    float* bias_s = reinterpret_cast<float*>(Os + k320_stage_chunks<MI, NI>() * 8);  // [BN]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const int ntiles = (int)(P.M / BM);

    // ---- resident weights: W^T fragments of my 64 columns, all 20 k-steps (A operand of the swapped product: rows = n) ----------
    bf16x8 wf[KS][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const bf16_t* wrow = P.w + (int64_t)(n0 + wave * WC + ni * 32 + l31) * K + half * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            union { bf16x8 v; u32x4 u; } t;
            t.u = *reinterpret_cast<const u32x4*>(wrow + ks * 16);
            wf[ks][ni] = t.v;
        }
    }
    for (int i = tid; i < BN; i += 320) bias_s[i] = P.bias ? bf2f(P.bias[n0 + i]) : 0.f;

    // ---- DMA pieces (1 KiB = 64 lanes x 16 B, LDS-linear).  A: chunk L = 64 piece + lane -> row L / 40, physical chunk L % 40 holds
    // logical chunk (L % 40) ^ ((row >> 1) & 7) (conflict-free b128 fragment reads at the 640-byte pitch).  Residual: L -> row L / 41,
    // chunk L % 41 of the padded staging row (chunk 40 is the pad: out of range, zeros) -------------------------------------------
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.a, 0, (int)(((P.M - 1) * P.lda + K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_RES ? P.res + n0 : P.a), 0,
                                                                         HAS_RES ? (int)(((P.M - 1) * P.ldres + BN) * 2) : 0, 0x00020000);
    constexpr unsigned OOB = 0x7ffffff0u;
    // (the per-lane offsets are recomputed at every request -- a dozen VALU per piece -- instead of living in 17 registers next to the
    // 160 of the weights and the 64 accumulators)
    int lane_v = lane, tid_v = tid;                      // (made opaque once per tile: see the loop)
    auto a_vo = [&](int piece) {
        const int L = piece * 64 + lane_v, row = L / CPR, phys = L - row * CPR;
        return (unsigned)(row * (int)P.lda * 2 + (phys ^ ((row >> 1) & 7)) * 16);
    };
    auto r_vo = [&](int piece) {
        const int L = piece * 64 + lane_v, row = L / OCPR, cc = L - row * OCPR;
        return (row < BM && cc < BN / 8) ? (unsigned)(row * (int)P.ldres * 2 + cc * 16) : OOB;
    };
    auto dma = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, int soff, bf16_t* lds) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, soff, 0, 0);
    };
    auto load_a = [&](int tile, int buf) {
        const int soff = tile * BM * (int)P.lda * 2;
#pragma unroll
        for (int i = 0; i < A_PPW; ++i)
            if (wave + 5 * i < A_PIECES) dma(rsA, a_vo(wave + 5 * i), soff, Abuf + buf * A_ELEMS + (wave + 5 * i) * 512);
    };
    auto load_res = [&](int tile) {
        const int soff = tile * BM * (int)P.ldres * 2;
#pragma unroll
        for (int i = 0; i < ST_PPW; ++i)
            if (wave + 5 * i < ST_PIECES) dma(rsR, r_vo(wave + 5 * i), soff, Os + (wave + 5 * i) * 512);
    };

    // fragment read offsets: row (mi*32 + l31), logical chunk 2 ks + half at physical chunk (2 ks + half) ^ sw, sw = (l31 >> 1) & 7.
    // (2 ks) ^ x = 2 ks + x - 2 (x & (2 ks & 6)) for x = sw & 6: four per-lane constants, the rest is an immediate
    const int sw = (l31 >> 1) & 7, x6 = sw & 6, b0 = half ^ (sw & 1);
    int dsel[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) dsel[m] = l31 * K + (x6 - 2 * (x6 & (2 * m)) + b0) * 8;

    int t = blockIdx.x;
    if (t < ntiles) load_a(t, 0);
    if (t + (int)gridDim.x < ntiles) load_a(t + gridDim.x, 1);
    if (HAS_RES && t < ntiles) load_res(t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
        const bf16_t* As = Abuf + (it & 1) * A_ELEMS;
        // per-lane offsets of the DMA requests and of the store loop are recomputed every tile from these: hoisted out of the persistent
        // loop they were ~40 spilled registers, and every scratch reload is a VMEM operation whose wait also drains the DMA queue
        asm volatile("" : "+v"(lane_v), "+v"(tid_v));
        // the accumulators start from the bias (out = alpha * (acc + bias) + residual): eight b128 reads here, all in flight together,
        // instead of one dependent LDS round trip in front of each of the epilogue's 16 word writes
        f32x16 acc[NI][MI];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + wave * WC + a * 32 + 8 * g + 4 * half);
#pragma unroll
                for (int b = 0; b < MI; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[a][b][4 * g + j] = bv[j];
            }
        // ---- 20 k-steps: A^T fragments from LDS (next step's requested before this step's MFMAs), W from registers ------------
        bf16x8 af[2][MI];
        auto read_a = [&](int ks, bf16x8 (&f)[MI]) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                union { bf16x8 v; u32x4 u; } ta;
                ta.u = *reinterpret_cast<const u32x4*>(As + mi * 32 * K + dsel[ks & 3] + ks * 16);
                f[mi] = ta.v;
            }
        };
        read_a(0, af[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) read_a(ks + 1, af[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][ni], af[ks & 1][mi], acc[ni][mi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // next A tile, this tile's residual rows (and the previous tile's stores) are done; everybody is done with As
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        asm volatile("" : "+v"(lane_v), "+v"(tid_v));       // (what the epilogue derives from the lane id is not kept alive across the MFMAs)
        const int l31_e = lane_v & 31, half_e = (lane_v >> 5) & 1;
        if (t + 2 * (int)gridDim.x < ntiles) load_a(t + 2 * gridDim.x, it & 1);

        // ---- epilogue: alpha * acc (+ residual words from the staging tile, four words requested at a time), one rounding, into the
        // staging tile ------------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                bf16_t* wbase = Os + (mi * 32 + l31_e) * OP + wave * WC + ni * 32 + 4 * half_e;
                u32x2 rw[4];
                if (HAS_RES) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) rw[g] = *reinterpret_cast<const u32x2*>(wbase + 8 * g);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = P.alpha * acc[ni][mi][4 * g + j];
                    if (HAS_RES) {
                        o[0] += __uint_as_float(rw[g][0] << 16); o[1] += __uint_as_float(rw[g][0] & 0xffff0000u);
                        o[2] += __uint_as_float(rw[g][1] << 16); o[3] += __uint_as_float(rw[g][1] & 0xffff0000u);
                    }
                    *reinterpret_cast<u32x2*>(wbase + 8 * g) = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            bf16_t* og = P.out + (int64_t)t * BM * P.ldo + n0;
            for (int c = tid_v; c < BM * (BN / 8); c += 320) {     // (a run-time loop: unrolled, its 8 staged chunks spill next to the resident weights)
                const int r = c / (BN / 8), ch = c - r * (BN / 8);
                *reinterpret_cast<u32x4*>(og + (int64_t)r * P.ldo + ch * 8) = *reinterpret_cast<const u32x4*>(Os + r * OP + ch * 8);
            }
        }
        if (HAS_RES) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // the staging tile has been read out
            if (t + (int)gridDim.x < ntiles) load_res(t + gridDim.x);
        }
    }
}

inline int gemm_k320_cfg() {          // FMC_K320_CFG = 22 (default) | 11 | 21 | 12: (MI, NI) blocks per wave (A/B)
    static const int v = [] {
        const char* e = getenv("FMC_K320_CFG");
        return e ? atoi(e) : 22;
    }();
    return v;
}
bool gemm_k320_ok(const GemmParams& P) {
    const int bn = (gemm_k320_cfg() % 10 == 1) ? 160 : 320;
    return !P.f32io && P.hw <= 1 && P.K == K320_K && P.N % bn == 0 && P.M % 64 == 0 && !P.a2 && !P.res2 && P.split_k == 1 && !P.sk && !P.temb &&
           ((P.M - 1) * P.lda + P.K) * 2 < ((int64_t)1 << 31) && (!P.res || ((P.M - 1) * P.ldres + P.N) * 2 < ((int64_t)1 << 31));
}

template <int MI, int NI>
void launch_gemm_k320_c(GemmParams& P, hipStream_t st) {
    const size_t lds = k320_lds<MI, NI>();
    const int per_cu = MI * NI == 1 ? 3 : (MI * NI == 2 ? 2 : 1);
    const int tiles = (int)(P.M / (32 * MI)), slots = fmc_cu_count() * per_cu;
    dim3 grid((unsigned)(tiles < slots ? tiles : slots), (unsigned)(P.N / (160 * NI)));
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_k320_kernel<MI, NI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_k320_kernel<MI, NI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        raised = true;
    }
    if (P.res) hipLaunchKernelGGL((gemm_k320_kernel<MI, NI, true>), grid, dim3(320), lds, st, P);
    else hipLaunchKernelGGL((gemm_k320_kernel<MI, NI, false>), grid, dim3(320), lds, st, P);
}
void launch_gemm_k320(GemmParams& P, hipStream_t st) {
    switch (gemm_k320_cfg()) {
        case 11: launch_gemm_k320_c<1, 1>(P, st); break;
        case 21: launch_gemm_k320_c<2, 1>(P, st); break;
        case 12: launch_gemm_k320_c<1, 2>(P, st); break;
        default: launch_gemm_k320_c<2, 2>(P, st); break;
    }
}

// tile 16: plain grid only (whole rounds are the point); N must be a multiple of 320 (GEGLU: weight rows per 16 = [8 value | 8 gate])
bool gemm160_ok(const GemmParams& P) {
    return P.N % 320 == 0 && !P.sk && (!P.a2 || P.a_blocked);   // (split_k > 1: plain epilogue only -- set_split_k has checked; a2: the persistent form's A2 variant, linear_impl has checked)
}
template <int MODE, int EPI>
void launch_gemm160(GemmParams& P, hipStream_t st) {
    P.tiles_m = (int)((P.M + 159) / 160);
    P.tiles_n = P.N / 320;
    P.group_m = 1;
    P.tap_outer = 0;
    if (gemm_group_m_override() > 0) {
        P.group_m = gemm_group_m_override();
    } else if (MODE == 0 && (int64_t)P.N * P.K * 2 > (int64_t)5 << 19) {     // as launch_gemm8: square per-XCD footprint in bytes
        const double c = fmin(32.0, (double)P.tiles_m * P.tiles_n / 8.0);
        const int gm = (int)lround(sqrt(c * 2.0));
        if (gm > 1 && P.tiles_n * 2 > 3 * (c / gm)) P.group_m = gm;
    }
    constexpr size_t lds = (size_t)5 * (160 + 320) * 32 * sizeof(bf16_t) + 1024;          // five sub-tile buffers + the dummies' KiB
    static_assert(lds >= (size_t)160 * 328 * 2 && lds >= (size_t)160 * 164 * 4, "epilogue staging fits the operand buffers");
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm160_kernel<MODE, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        raised = true;
    }
    if (P.split_k > 1) {
        if constexpr (EPI == 0) {
            hipLaunchKernelGGL((gemm160_kernel<MODE, 0>), dim3((unsigned)(P.tiles_m * P.tiles_n * P.split_k)), dim3(512), lds, st, P);
            const int64_t chunks = P.M * (P.N / 8);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, P);
        }
        return;
    }
    if constexpr (MODE == 0) {
        // more tiles than CUs on a token projection: the persistent form (the next tile's operands stream in under this tile's epilogue)
        static const int persist = getenv("FMC_G160_PERSIST") ? atoi(getenv("FMC_G160_PERSIST")) : 1;
        const int cus = fmc_cu_count() & ~7;
        // (FMC_G160_PERSIST=2: also launches of exactly one round -- tiles == CUs, the level-1 N = 640 projections -- A/B switch)
        // (a tile-major A operand exists in the persistent form only: it also takes launches of exactly one round -- the two halves of a split feed-forward)
        if (persist && !P.f32io && P.M % 160 == 0 && ((persist == 2 || P.a_blocked || P.bias_img) ? P.tiles_m * P.tiles_n >= cus : P.tiles_m * P.tiles_n > cus) && cus >= 8) {
            constexpr size_t ldsp = (size_t)3 * (160 + 320) * 32 * sizeof(bf16_t) + 1024 + (EPI == 1 ? (size_t)160 * 168 * 2 + 4096 : (size_t)80 * 328 * 2 + 5120);
            static FmcPerDeviceFlag raisedp;
            if (!raisedp) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm160p_kernel<EPI, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm160p_kernel<EPI, 5, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
                if constexpr (EPI == 0) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm160p_kernel<0, 5, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm160p_kernel<0, 5, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
                }
                raisedp = true;
            }
            if constexpr (EPI == 0) {
                if (P.bias_img) {                             // per-image weights / fp32 bias rows (linear_impl has checked; with or without the LayerNorm statistics)
                    static FmcPerDeviceFlag raised3;
                    if (!raised3) {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm160p_kernel<0, 5, 0, 0, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm160p_kernel<0, 5, 2, 0, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
                        raised3 = true;
                    }
                    if (P.ln_stats) hipLaunchKernelGGL((gemm160p_kernel<0, 5, 2, 0, 0, 1>), dim3((unsigned)cus), dim3(512), ldsp, st, P);
                    else hipLaunchKernelGGL((gemm160p_kernel<0, 5, 0, 0, 0, 1>), dim3((unsigned)cus), dim3(512), ldsp, st, P);
                    return;
                }
                if (P.a2) {                                   // two-segment reduction (linear_impl has checked: tile-major first segment, plain epilogue)
                    static FmcPerDeviceFlag raised2;
                    if (!raised2) {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm160p_kernel<0, 5, 0, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
                        raised2 = true;
                    }
                    hipLaunchKernelGGL((gemm160p_kernel<0, 5, 0, 0, 1>), dim3((unsigned)cus), dim3(512), ldsp, st, P);
                    return;
                }
            }
            if (P.lnc_stats) {                                // this GEMM applies the LayerNorm of its input rows (linear_impl has checked the rest)
                hipLaunchKernelGGL((gemm160p_kernel<EPI, 5, 0, 1>), dim3((unsigned)cus), dim3(512), ldsp, st, P);
                return;
            }
            if constexpr (EPI == 0) {
                if (P.ln_out) {                               // (linear_impl has checked N == 320 and no GroupNorm partials)
                    hipLaunchKernelGGL((gemm160p_kernel<0, 5, 1>), dim3((unsigned)cus), dim3(512), ldsp, st, P);
                    return;
                }
                if (P.ln_stats) {
                    hipLaunchKernelGGL((gemm160p_kernel<0, 5, 2>), dim3((unsigned)cus), dim3(512), ldsp, st, P);
                    return;
                }
            }
            hipLaunchKernelGGL((gemm160p_kernel<EPI, 5>), dim3((unsigned)cus), dim3(512), ldsp, st, P);
            return;
        }
    }
    // (not reached with ln_out / ln_stats / lnc_stats set: linear_impl checks the persistent form's conditions)
    hipLaunchKernelGGL((gemm160_kernel<MODE, EPI>), dim3((unsigned)(P.tiles_m * P.tiles_n)), dim3(512), lds, st, P);
}

// tile 17: 256 x 320 tiles, persistent, GEGLU epilogue only (the feed-forward input projections: M % 256 == 0, more tiles than CUs)
bool gemm256p_ok(const GemmParams& P) {
    const int cus = fmc_cu_count() & ~7;
    return P.N % 320 == 0 && P.M % 256 == 0 && !P.sk && !P.a2 && P.split_k == 1 && !P.f32io && cus >= 8 && (P.M / 256) * (P.N / 320) > cus;
}
void launch_gemm256p(GemmParams& P, hipStream_t st) {
    P.tiles_m = (int)(P.M / 256);
    P.tiles_n = P.N / 320;
    P.group_m = 1;
    P.tap_outer = 0;
    if (gemm_group_m_override() > 0) {
        P.group_m = gemm_group_m_override();
    } else if ((int64_t)P.N * P.K * 2 > (int64_t)5 << 19) {                   // as launch_gemm160: square per-XCD footprint in bytes
        const double c = fmin(32.0, (double)P.tiles_m * P.tiles_n / 8.0);
        const int gm = (int)lround(sqrt(c * 1.25));
        if (gm > 1 && P.tiles_n * 2 > 3 * (c / gm)) P.group_m = gm;
    }
    constexpr size_t ldsp = (size_t)3 * (256 + 320) * 32 * sizeof(bf16_t) + 1024 + (size_t)128 * 168 * 2;
    static_assert(ldsp <= 160 * 1024, "ring + staging fit the LDS");
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm160p_kernel<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
        raised = true;
    }
    hipLaunchKernelGGL((gemm160p_kernel<1, 8>), dim3((unsigned)(fmc_cu_count() & ~7)), dim3(512), ldsp, st, P);
}

bool gemm8_ok(GemmParams& P) {
    // operand sizes as the buffer descriptors see them (conv: the whole input tensor; token: up to the end of the last row)
    const int64_t a_elems = P.hw > 1 ? (P.ups == 1 ? (P.M / 4) * (int64_t)P.cin : P.M * (int64_t)P.cin * (P.ups == 2 ? 4 : 1))
                                     : (P.a2 && P.a_blocked ? P.M * (int64_t)P.ksplit : (P.M - 1) * P.lda + P.K);
    P.a_bytes = a_elems * 2;
    return P.split_k == 1 && (!P.a2 || P.a_blocked) && P.a_bytes < ((int64_t)1 << 31) && (int64_t)P.N * P.K * 2 < ((int64_t)1 << 31);
}

// tile arms (fmc_hip.h): geometry x k-tile depth x ring depth
constexpr int GEMM_TILE_MAX = 22;

// geometry: the largest tile that still gives every CU work and does not waste more than ~20 % of N
template <int MODE, int EPI>
void launch_gemm(GemmParams& P, int tile, hipStream_t st) {
    int g = gemm_geometry_override();
    if (g == 0) g = tile;
    if (g == 0) {
        auto tiles = [&](int bm, int bn) { return ((P.M + bm - 1) / bm) * ((P.N + bn - 1) / bn); };
        auto waste = [&](int bn) { return (double)(((P.N + bn - 1) / bn) * bn) / P.N; };
        if (tiles(256, 256) >= 256 && waste(256) <= 1.2) g = 3;
        else if (tiles(256, 128) >= 256 && waste(128) <= 1.25) g = 2;
        else g = 1;
    }
    if (g == 17) {                                        // 256 x 320 persistent tiles, GEGLU only; anything else takes the 160 x 320 kernel
        if constexpr (MODE == 0 && EPI == 1) {
            if (gemm8_ok(P) && gemm256p_ok(P)) { launch_gemm256p(P, st); return; }
        }
        g = 16;
    }
    if (g == 16) {                                        // 160 x 320 tiles (8-phase schedule, 16x16x32 MFMA): N % 320 == 0, plain grid
        const int sk_keep = P.split_k;
        P.split_k = 1;
        const bool fits = gemm8_ok(P);                    // (32-bit operand offsets; sets P.a_bytes)
        P.split_k = sk_keep;
        if (fits && gemm160_ok(P)) { launch_gemm160<MODE, EPI>(P, st); return; }
        g = 13;
    }
    if ((g == 13 || g == 14) && !gemm8_ok(P)) g = 3;      // 8-phase kernel: plain grid, 32-bit operand offsets
    if (g == 15) {                                        // W-stationary persistent kernel of the K = 320 token projections
        if constexpr (MODE == 0 && EPI == 0) {
            if (gemm_k320_ok(P)) { launch_gemm_k320(P, st); return; }
        }
        g = 5;
    }
    if constexpr (MODE == 0 && EPI == 0) {                // small-M experiment (round 4): 64 x 128 tiles, a wave = 32 x 64; in-register epilogue only
        if (g >= 19 && g <= 22 && (P.split_k != 1 || P.res2 || P.sk || P.f32io)) g = 1;
        if (g == 19) { launch_gemm_g<0, 0, 2, 2, 64, 3, 1>(P, st); return; }      // 3-stage ring of 24-KiB k-tiles: 2 workgroups per CU
        if (g == 20) { launch_gemm_g<0, 0, 2, 2, 32, 4, 1>(P, st); return; }      // 32-deep k-tiles, 4-stage ring (48 KiB): 3 per CU
        if (g == 21) { launch_gemm_g<0, 0, 4, 2, 64, 2, 1>(P, st); return; }      // 128 x 128 as 8 waves of 32 x 64: 64 KiB, two workgroups = 16 waves per CU
        if (g == 22) { launch_gemm_g<0, 0, 2, 4, 64, 2, 1>(P, st); return; }      // 64 x 256, 8 waves: 80 KiB, two per CU
    }
    switch (g) {
        case 14: launch_gemm8<MODE, EPI, 5>(P, st); break;              // 8-phase 256x256, 5 half-tiles ahead
        case 13: launch_gemm8<MODE, EPI, 4>(P, st); break;              // 8-phase 256x256, 4 half-tiles ahead
        case 12: launch_gemm_g<MODE, EPI, 2, 5, 32, 4>(P, st); break;   // 128x320, 32-deep k-tiles, 4-stage ring
        case 11: launch_gemm_g<MODE, EPI, 2, 5, 64, 2>(P, st); break;   // 128x320: 2 x 5 waves
        case 10: launch_gemm_g<MODE, EPI, 4, 2, 32, 4>(P, st); break;
        case 9: launch_gemm_g<MODE, EPI, 2, 2, 32, 4>(P, st); break;
        case 8: launch_gemm_g<MODE, EPI, 4, 4, 32, 4>(P, st); break;
        case 7: launch_gemm_g<MODE, EPI, 4, 2, 64, 3>(P, st); break;
        case 6: launch_gemm_g<MODE, EPI, 4, 4, 32, 2>(P, st); break;
        case 5: launch_gemm_g<MODE, EPI, 4, 2, 32, 2>(P, st); break;
        case 4: launch_gemm_g<MODE, EPI, 2, 2, 32, 2>(P, st); break;
        case 3: launch_gemm_g<MODE, EPI, 4, 4, 64, 2>(P, st); break;
        case 2: launch_gemm_g<MODE, EPI, 4, 2, 64, 2>(P, st); break;
        default: launch_gemm_g<MODE, EPI, 2, 2, 64, 2>(P, st); break;
    }
}

int set_split_k(GemmParams& P, int split_k, void* workspace, int64_t workspace_bytes, bool plain_epilogue, const char* who) {
    P.split_k = 1;
    P.ws = nullptr;
    P.sk = 0;
    P.sk_t0 = 0;
    P.sk_flags = nullptr;
    P.sk_ws_bytes = 0;
    P.sk_hybrid = split_k == -2;
    P.sk_lock = split_k == -3 ? 1 : (split_k <= -16 ? -split_k - 16 : 0);      // k-lockstep split on the 8-phase arms (other arms: plain stream-K)
    P.sk_lock_len = 0;
    if (split_k <= -16 && (P.sk_lock < 2 || P.sk_lock > 64)) FMC_FAIL(FMC_E_SHAPE, "%s: split_k %d (k-lockstep chunks = -split_k - 16 in 2..64)", who, split_k);
    if (split_k == -1 || split_k == -2 || split_k == -3 || split_k <= -16) {                // stream-K: [4096 B of flags (zero on entry and on return) | partials]
        if (!workspace || !fmc_aligned16(workspace) || workspace_bytes < 8192)
            FMC_FAIL(FMC_E_NULL, "%s: stream-K needs a 16-byte aligned, zero-initialised workspace", who);
        P.sk = 1;
        P.sk_flags = (int*)workspace;
        P.ws = (float*)((char*)workspace + 4096);
        P.sk_ws_bytes = workspace_bytes;
        return 0;
    }
    if (split_k <= 1) return 0;
    if (split_k > 16 || !plain_epilogue) FMC_FAIL(FMC_E_SHAPE, "%s: split_k %d (1..16, not with the GEGLU epilogue)", who, split_k);
    if (!workspace || !fmc_aligned16(workspace)) FMC_FAIL(FMC_E_NULL, "%s: split_k needs a 16-byte aligned workspace", who);
    if (workspace_bytes < (int64_t)split_k * P.M * P.N * (int64_t)sizeof(float))
        FMC_FAIL(FMC_E_SHAPE, "%s: workspace of %lld bytes < split_k*M*N*4", who, (long long)workspace_bytes);
    P.split_k = split_k;
    P.ws = (float*)workspace;
    return 0;
}

}  // namespace

static int linear_impl(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M,
                       int N, int K, int64_t ldx, int64_t ldres, int64_t ldo, float alpha, int epilogue, int tile,
                       int split_k, void* workspace, int64_t workspace_bytes, const void* x2, int64_t ldx2,
                       int k_split, const void* residual2, void* stream, int f32io, void* gn_partials = nullptr, int gn_hw = 0,
                       void* ln_out = nullptr, const float* ln_gamma = nullptr, const float* ln_beta = nullptr, float ln_eps = 0.f,
                       const float* ln_pe = nullptr, int ln_pe_inner = 1, int ln_pe_frames = 1, float* ln_stats = nullptr,
                       const float* lnc_stats = nullptr, const float* lnc_c = nullptr, const float* lnc_bias = nullptr, int a_blocked = 0,
                       int out_blocked = 0, int w_tilemajor = 0, const float* bias_img = nullptr, int img_rows = 0) {
    if (!x || !w || !out) FMC_FAIL(FMC_E_NULL, "linear_bf16: NULL tensor");
    if (tile == 18) { tile = 16; w_tilemajor = 1; }           // tile 18 = tile 16 on a weight pre-packed tile-major (fmc_hip.h)
    if (M <= 0 || N <= 0 || K <= 0 || K % BK_MAX || N % 8 || ldx % 8 || ldo % (f32io ? 4 : 8) || (residual && ldres % (f32io ? 4 : 8)))
        FMC_FAIL(FMC_E_SHAPE, "linear_bf16: need K%%64==0, N%%8==0 and strides %%8==0 (M=%lld N=%d K=%d)", (long long)M, N, K);
    if (f32io && (x2 || split_k < 0)) FMC_FAIL(FMC_E_SHAPE, "linear_x3_f32: no two-source operand (split it into one buffer) and no stream-K");
    if (epilogue != 0 && epilogue != 1) FMC_FAIL(FMC_E_SHAPE, "linear_bf16: epilogue %d", epilogue);
    if (epilogue == 1 && (N % 64 || residual)) FMC_FAIL(FMC_E_SHAPE, "linear_bf16: GEGLU needs N%%64==0 and no residual");
    if (!fmc_aligned16(x) || !fmc_aligned16(w) || !fmc_aligned16(out) || (residual && !fmc_aligned16(residual)) ||
        (bias && !fmc_aligned16(bias)))
        FMC_FAIL(FMC_E_ALIGN, "linear_bf16: tensors must be 16-byte aligned");
    GemmParams P{};
    P.a = (const bf16_t*)x; P.w = (const bf16_t*)w; P.bias = (const bf16_t*)bias; P.temb = nullptr;
    if (residual2 && (!residual || !fmc_aligned16(residual2)))
        FMC_FAIL(FMC_E_NULL, "linear_bf16: residual2 needs residual (same row stride) and 16-byte alignment");
    P.res = (const bf16_t*)residual; P.res2 = (const bf16_t*)residual2; P.out = (bf16_t*)out;
    P.M = M; P.N = N; P.K = K; P.lda = ldx; P.ldres = ldres; P.ldo = ldo;
    P.img_h = P.img_w = P.cin = 0; P.hw = 1; P.alpha = alpha; P.temb_ld = 0; P.temb_div = 1; P.ups = 0;
    P.f32io = f32io;
    if (gn_partials && (f32io || epilogue != 0 || tile != 16 || split_k != 1 || gn_hw <= 0 || gn_hw % 160 || M % gn_hw || N % 320 || (x2 && !a_blocked) ||
                        ((M - 1) * ldx + K) * 2 >= ((int64_t)1 << 31) || (int64_t)N * K * 2 >= ((int64_t)1 << 31)))
        FMC_FAIL(FMC_E_SHAPE, "linear_bf16: GroupNorm partials come out of tile 16's plain bf16 epilogue only (N %% 320 == 0, pixels per image %% 160 == 0)");
    P.gn_part = (float*)gn_partials; P.gn_hw = gn_hw;
    if (lnc_stats) {                                          // consumer: LayerNorm of the A rows applied in the epilogue (W pre-scaled by gamma)
        const int cus = fmc_cu_count() & ~7;
        if (f32io || tile != 16 || split_k != 1 || gn_partials || ln_out || ln_stats || x2 || bias || residual || residual2 || alpha != 1.f || N % 320 ||
            M % 160 || (M / 160) * (N / 320) <= cus || cus < 8 || !lnc_c || !lnc_bias || !fmc_aligned16(lnc_c) || !fmc_aligned16(lnc_bias) ||
            ((M - 1) * ldx + K) * 2 >= ((int64_t)1 << 31) || (int64_t)N * K * 2 >= ((int64_t)1 << 31) ||
            (getenv("FMC_G160_PERSIST") && atoi(getenv("FMC_G160_PERSIST")) == 0))
            FMC_FAIL(FMC_E_SHAPE, "linear_bf16_lnc: tile 16's persistent form only (bf16, N %% 320 == 0, M %% 160 == 0, more tiles than CUs, no bf16 bias / "
                                  "residual, alpha 1)");
    }
    P.lnc_stats = lnc_stats; P.lnc_c = lnc_c; P.lnc_bias = lnc_bias;
    if (a_blocked || out_blocked) {                           // tile-major intermediate of the feed-forward: persistent form of tile 16 only
        const int cus = fmc_cu_count() & ~7;
        // (x2 with a tile-major x: the reduction's second segment [k_split, K) comes from row-major x2 -- gemm160p_kernel's A2 variant, plain epilogue,
        //  GroupNorm partials allowed: fmc_linear_bf16_fftail)
        if (f32io || tile != 16 || split_k != 1 || (gn_partials && !(a_blocked && x2 && !out_blocked)) || (x2 && (!a_blocked || out_blocked)) || N % 320 || M % 160 || cus < 8 ||
            (out_blocked && (M / 160) * (N / 320) <= cus) ||                      // (the GEGLU epilogue that writes tile-major is the persistent form's)
            (a_blocked && (M / 160) * (N / 320) < cus) ||
            (a_blocked && (epilogue != 0 || ldx != (x2 ? k_split : K) || K % 32)) ||
            (x2 && (ln_out || ln_stats || lnc_stats || ((M - 1) * ldx2 + (K - k_split)) * 2 >= ((int64_t)1 << 31))) || (out_blocked && (epilogue != 1 || ldo != N / 2 || (N / 2) % 32)) ||
            (int64_t)M * (a_blocked ? K : N / 2) * 2 >= ((int64_t)1 << 31) || (getenv("FMC_G160_PERSIST") && atoi(getenv("FMC_G160_PERSIST")) == 0))
            FMC_FAIL(FMC_E_SHAPE, "linear_bf16: the tile-major feed-forward intermediate needs tile 16's persistent form (M %% 160 == 0, N %% 320 == 0, "
                                  "more tiles than CUs, dense rows)");
    }
    P.a_blocked = a_blocked; P.out_blocked = out_blocked;
    P.bias_img = nullptr; P.w_img_stride = 0; P.img_rows = 0;
    if (bias_img) {                                           // per-image weights (a GroupNorm folded in): persistent form of tile 16, plain epilogue
        const int cus = fmc_cu_count() & ~7;
        if (f32io || epilogue != 0 || tile != 16 || split_k != 1 || gn_partials || x2 || bias || residual || residual2 || alpha != 1.f || ln_out || lnc_stats ||
            a_blocked || out_blocked || N % 320 || M % 160 || (M / 160) * (N / 320) < cus || cus < 8 || img_rows < 160 || img_rows % 160 || M % img_rows ||
            ((uintptr_t)bias_img & 15) || ((M - 1) * ldx + K) * 2 >= ((int64_t)1 << 31) || (M / img_rows) * (int64_t)N * K * 2 >= ((int64_t)1 << 31) ||
            (getenv("FMC_G160_PERSIST") && atoi(getenv("FMC_G160_PERSIST")) == 0))
            FMC_FAIL(FMC_E_SHAPE, "linear_bf16_imgw: tile 16's persistent form only (bf16, N %% 320 == 0, M %% 160 == 0, at least as many tiles as CUs, images of a multiple "
                                  "of 160 rows, no bf16 bias / residual, all weights < 2 GiB)");
        P.bias_img = bias_img; P.w_img_stride = (int64_t)N * K; P.img_rows = img_rows;
    }
    if (ln_out || ln_stats) {
        const int cus = fmc_cu_count() & ~7;
        if ((ln_out != nullptr) == (ln_stats != nullptr)) FMC_FAIL(FMC_E_NULL, "linear_bf16_ln: exactly one of ln_out / ln_stats");
        if (f32io || epilogue != 0 || tile != 16 || split_k != 1 || gn_partials || x2 || N != 320 || M % 160 || (bias_img ? M / 160 < cus : M / 160 <= cus) || cus < 8 ||
            (ln_out && (!ln_gamma || !ln_beta || !fmc_aligned16(ln_out))) || (ln_stats && ((uintptr_t)ln_stats & 7)) || (ln_pe && (ln_pe_inner <= 0 || ln_pe_inner % 160 || ln_pe_frames <= 0)) ||
            ((M - 1) * ldx + K) * 2 >= ((int64_t)1 << 31) || (getenv("FMC_G160_PERSIST") && atoi(getenv("FMC_G160_PERSIST")) == 0))
            FMC_FAIL(FMC_E_SHAPE, "linear_bf16_ln: the LayerNorm output comes out of tile 16's persistent form only (bf16, N == 320, M %% 160 == 0, "
                                  "M / 160 > CUs, plain epilogue, positional-encoding frames of a multiple of 160 rows)");
    }
    P.ln_stats = ln_stats;
    P.ln_out = (bf16_t*)ln_out; P.ln_gamma = ln_gamma; P.ln_beta = ln_beta; P.ln_eps = ln_eps; P.ln_pe = ln_pe;
    P.ln_pe_inner = ln_pe_inner; P.ln_pe_frames = ln_pe_frames;
    if (x2 && (k_split <= 0 || k_split >= K || k_split % BK_MAX || ldx2 % 8 || !fmc_aligned16(x2)))
        FMC_FAIL(FMC_E_SHAPE, "linear_bf16: two-source input needs 0 < k_split < K, k_split %% 64 == 0 (k_split=%d K=%d)", k_split, K);
    P.a2 = (const bf16_t*)x2; P.lda2 = ldx2; P.ksplit = x2 ? k_split : 0;
    hipStream_t st = (hipStream_t)stream;
    if (tile < 0 || tile > GEMM_TILE_MAX) FMC_FAIL(FMC_E_SHAPE, "linear_bf16: tile %d", tile);
    P.w_blocked = 0;
    if (w_tilemajor) {                                        // no other kernel can read that weight: everything that would leave tile 16 is an error
        if (tile != 16 || f32io || N % 320 || (x2 && !a_blocked) || split_k < 1 || ((M - 1) * ldx + K) * 2 >= ((int64_t)1 << 31) || (int64_t)N * K * 2 >= ((int64_t)1 << 31))
            FMC_FAIL(FMC_E_SHAPE, "linear_bf16: a tile-major weight (tile 18) needs bf16, N %% 320 == 0, no two-source operand, no stream-K, operands < 2 GiB");
        P.w_blocked = 1;
    }
    if (int rc = set_split_k(P, split_k, workspace, workspace_bytes, epilogue == 0, "linear_bf16")) return rc;
    if (epilogue == 0) launch_gemm<0, 0>(P, tile, st); else launch_gemm<0, 1>(P, tile, st);
    FMC_CHECK_LAUNCH("fmc_linear_bf16");
    return 0;
}

extern "C" int fmc_linear_bf16(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M,
                               int N, int K, int64_t ldx, int64_t ldres, int64_t ldo, float alpha, int epilogue, int tile,
                               int split_k, void* workspace, int64_t workspace_bytes, const void* x2, int64_t ldx2,
                               int k_split, const void* residual2, void* stream) {
    return linear_impl(x, w, bias, residual, out, M, N, K, ldx, ldres, ldo, alpha, epilogue, tile, split_k, workspace,
                       workspace_bytes, x2, ldx2, k_split, residual2, stream, 0);
}

extern "C" int fmc_linear_bf16_gn(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M,
                                  int N, int K, int64_t ldx, int64_t ldres, int64_t ldo, float alpha, const void* residual2,
                                  float* gn_partials, int gn_hw, int w_tilemajor, void* stream) {
    if (!gn_partials) FMC_FAIL(FMC_E_NULL, "linear_bf16_gn: NULL gn_partials");
    return linear_impl(x, w, bias, residual, out, M, N, K, ldx, ldres, ldo, alpha, 0, 16, 1, nullptr, 0, nullptr, 0, 0, residual2, stream, 0,
                       gn_partials, gn_hw, nullptr, nullptr, nullptr, 0.f, nullptr, 1, 1, nullptr, nullptr, nullptr, nullptr, 0, 0, w_tilemajor);
}

extern "C" int fmc_linear_bf16_ln(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M,
                                  int N, int K, int64_t ldx, int64_t ldres, int64_t ldo, float alpha, const void* residual2,
                                  void* ln_out, const float* ln_gamma, const float* ln_beta, float ln_eps, const float* ln_pe,
                                  int ln_pe_inner, int ln_pe_frames, float* ln_stats, int w_tilemajor, void* stream) {
    if (!ln_out && !ln_stats) FMC_FAIL(FMC_E_NULL, "linear_bf16_ln: NULL ln_out and ln_stats");
    if (ln_out && (!ln_gamma || !ln_beta)) FMC_FAIL(FMC_E_NULL, "linear_bf16_ln: NULL gamma / beta");
    return linear_impl(x, w, bias, residual, out, M, N, K, ldx, ldres, ldo, alpha, 0, 16, 1, nullptr, 0, nullptr, 0, 0, residual2, stream, 0,
                       nullptr, 0, ln_out, ln_gamma, ln_beta, ln_eps, ln_pe, ln_pe_inner, ln_pe_frames, ln_stats, nullptr, nullptr, nullptr, 0, 0, w_tilemajor);
}

extern "C" int fmc_linear_bf16_lnc(const void* x, const void* w_gamma, void* out, int64_t M, int N, int K, int64_t ldx, int64_t ldo, int epilogue,
                                   const float* ln_stats, const float* ln_c, const float* ln_bias, int w_tilemajor, void* stream) {
    if (!ln_stats || !ln_c || !ln_bias) FMC_FAIL(FMC_E_NULL, "linear_bf16_lnc: NULL ln_stats / ln_c / ln_bias");
    return linear_impl(x, w_gamma, nullptr, nullptr, out, M, N, K, ldx, 0, ldo, 1.f, epilogue, 16, 1, nullptr, 0, nullptr, 0, 0, nullptr, stream, 0,
                       nullptr, 0, nullptr, nullptr, nullptr, 0.f, nullptr, 1, 1, nullptr, ln_stats, ln_c, ln_bias, 0, 0, w_tilemajor);
}

extern "C" int fmc_linear_bf16_ffblk(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M, int N, int K,
                                     int64_t ldres, float alpha, int epilogue, int x_blocked, int out_blocked, const float* ln_stats,
                                     const float* ln_c, const float* ln_bias, int w_tilemajor, void* stream) {
    if (!x_blocked && !out_blocked) FMC_FAIL(FMC_E_SHAPE, "linear_bf16_ffblk: neither operand is tile-major (use fmc_linear_bf16)");
    const int n_out = epilogue == 1 ? N / 2 : N;
    return linear_impl(x, w, bias, residual, out, M, N, K, K, ldres, n_out, alpha, epilogue, 16, 1, nullptr, 0, nullptr, 0, 0, nullptr, stream, 0,
                       nullptr, 0, nullptr, nullptr, nullptr, 0.f, nullptr, 1, 1, nullptr, ln_stats, ln_c, ln_bias, x_blocked, out_blocked, w_tilemajor);
}

extern "C" int fmc_linear_bf16_imgw(const void* x, const void* w_img, const float* bias_img, void* out, int64_t M, int N, int K, int64_t ldx, int64_t ldo,
                                    int img_rows, float* ln_stats, float ln_eps, int w_tilemajor, void* stream) {
    if (!bias_img) FMC_FAIL(FMC_E_NULL, "linear_bf16_imgw: NULL bias_img");
    return linear_impl(x, w_img, nullptr, nullptr, out, M, N, K, ldx, 0, ldo, 1.f, 0, 16, 1, nullptr, 0, nullptr, 0, 0, nullptr, stream, 0,
                       nullptr, 0, nullptr, nullptr, nullptr, ln_eps, nullptr, 1, 1, ln_stats, nullptr, nullptr, nullptr, 0, 0, w_tilemajor, bias_img, img_rows);
}

extern "C" int fmc_linear_bf16_fftail(const void* x_blocked, const void* x2, const void* w, const void* bias, const void* residual, void* out, int64_t M,
                                      int N, int K, int k_split, int64_t ldx2, int64_t ldres, float* gn_partials, int gn_hw, int w_tilemajor, void* stream) {
    if (!x2) FMC_FAIL(FMC_E_NULL, "linear_bf16_fftail: NULL x2 (use fmc_linear_bf16_ffblk)");
    return linear_impl(x_blocked, w, bias, residual, out, M, N, K, k_split, ldres, N, 1.f, 0, 16, 1, nullptr, 0, x2, ldx2, k_split, nullptr, stream, 0,
                       gn_partials, gn_hw, nullptr, nullptr, nullptr, 0.f, nullptr, 1, 1, nullptr, nullptr, nullptr, nullptr, 1, 0, w_tilemajor);
}

extern "C" int fmc_linear_x3_f32(const void* x3, const void* w3, const float* bias, const float* residual, float* out, int64_t M,
                                 int N, int K3, int64_t ldx, int64_t ldres, int64_t ldo, float alpha, int epilogue, int tile,
                                 int split_k, void* workspace, int64_t workspace_bytes, const float* residual2, void* stream) {
    if (K3 % 3 || (K3 / 3) % 8) FMC_FAIL(FMC_E_SHAPE, "linear_x3_f32: K3 = 3 K of the split operands (K3=%d)", K3);
    return linear_impl(x3, w3, bias, residual, out, M, N, K3, ldx, ldres, ldo, alpha, epilogue, tile, split_k, workspace,
                       workspace_bytes, nullptr, 0, 0, residual2, stream, 1);
}

static int conv3x3_impl(const void* x, const void* w, const void* bias, const void* temb, const void* residual,
                        void* out, int n_img, int H, int W, int Cin, int Cout, int64_t temb_row_stride,
                        int temb_img_div, int upsample2x, int tile, int split_k,
                        void* workspace, int64_t workspace_bytes, void* stream, int f32io, void* gn_partials = nullptr, int w_tilemajor = 0) {
    if (!x || !w || !out) FMC_FAIL(FMC_E_NULL, "conv3x3_bf16: NULL tensor");
    if (tile == 18) { tile = 16; w_tilemajor = 1; }           // tile 18 = tile 16 on a filter pre-packed tile-major (fmc_hip.h)
    if (n_img <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % BK_MAX || Cout % 8)
        FMC_FAIL(FMC_E_SHAPE, "conv3x3_bf16: need Cin%%64==0 and Cout%%8==0 (Cin=%d Cout=%d)", Cin, Cout);
    if (!fmc_aligned16(x) || !fmc_aligned16(w) || !fmc_aligned16(out) || (residual && !fmc_aligned16(residual)) ||
        (temb && !fmc_aligned16(temb)) || (bias && !fmc_aligned16(bias)))
        FMC_FAIL(FMC_E_ALIGN, "conv3x3_bf16: tensors must be 16-byte aligned");
    GemmParams P{};
    P.a = (const bf16_t*)x; P.w = (const bf16_t*)w; P.bias = (const bf16_t*)bias; P.temb = (const bf16_t*)temb;
    P.res = (const bf16_t*)residual; P.res2 = nullptr; P.out = (bf16_t*)out;
    P.M = (int64_t)n_img * H * W; P.N = Cout; P.K = 9 * Cin; P.lda = Cin; P.ldres = Cout; P.ldo = Cout;
    P.a2 = nullptr; P.lda2 = 0; P.ksplit = 0;
    P.img_h = H; P.img_w = W; P.cin = Cin; P.hw = H * W; P.alpha = 1.f;
    if (temb && (temb_img_div < 1 || temb_row_stride % (f32io ? 4 : 8))) FMC_FAIL(FMC_E_SHAPE, "conv3x3_bf16: temb_img_div >= 1, temb_row_stride %% 8 == 0");
    P.f32io = f32io;
    if (gn_partials && (f32io || tile != 16 || split_k != 1 || (H * W) % 160 || Cout % 320 ||
                        (int64_t)n_img * H * W * Cin * 2 * (upsample2x == 2 ? 4 : 1) >= ((int64_t)1 << 31) || (int64_t)Cout * 9 * Cin * 2 >= ((int64_t)1 << 31)))
        FMC_FAIL(FMC_E_SHAPE, "conv3x3_bf16: GroupNorm partials come out of tile 16's plain bf16 epilogue only (Cout %% 320 == 0, H W %% 160 == 0)");
    P.gn_part = (float*)gn_partials; P.gn_hw = H * W;
    if (f32io && split_k < 0) FMC_FAIL(FMC_E_SHAPE, "conv3x3_x3_f32: no stream-K in fp32-storage mode");
    P.temb_ld = temb_row_stride; P.temb_div = temb_img_div < 1 ? 1 : temb_img_div;
    if (upsample2x < 0 || upsample2x > 2) FMC_FAIL(FMC_E_SHAPE, "conv3x3_bf16: resample mode %d", upsample2x);
    if (upsample2x == 1 && ((H | W) & 1)) FMC_FAIL(FMC_E_SHAPE, "conv3x3_bf16: upsample2x needs even H, W (the OUTPUT size)");
    P.ups = upsample2x;
    if (tile < 0 || tile > GEMM_TILE_MAX) FMC_FAIL(FMC_E_SHAPE, "conv3x3_bf16: tile %d", tile);
    P.w_blocked = 0;
    if (w_tilemajor) {                                        // the filter is packed in the kernel's (chunk, tap, half) sub-tile order
        if (tile != 16 || f32io || Cout % 320 || Cin % 64 || split_k < 1 || (int64_t)Cout * 9 * Cin * 2 >= ((int64_t)1 << 31) ||
            (int64_t)n_img * H * W * Cin * 2 * (upsample2x == 2 ? 4 : 1) >= ((int64_t)1 << 31))
            FMC_FAIL(FMC_E_SHAPE, "conv3x3_bf16: tile 18 (tile-major filter) needs bf16, Cout %% 320 == 0, Cin %% 64 == 0, no stream-K, operands < 2 GiB");
        P.w_blocked = 1;
    }
    if (int rc = set_split_k(P, split_k, workspace, workspace_bytes, true, "conv3x3_bf16")) return rc;
    launch_gemm<1, 0>(P, tile, (hipStream_t)stream);
    FMC_CHECK_LAUNCH("fmc_conv3x3_bf16");
    return 0;
}

extern "C" int fmc_conv3x3_bf16(const void* x, const void* w, const void* bias, const void* temb, const void* residual,
                                void* out, int n_img, int H, int W, int Cin, int Cout, int64_t temb_row_stride,
                                int temb_img_div, int upsample2x, int tile, int split_k,
                                void* workspace, int64_t workspace_bytes, void* stream) {
    return conv3x3_impl(x, w, bias, temb, residual, out, n_img, H, W, Cin, Cout, temb_row_stride, temb_img_div, upsample2x, tile,
                        split_k, workspace, workspace_bytes, stream, 0);
}

extern "C" int fmc_conv3x3_bf16_gn(const void* x, const void* w, const void* bias, const void* temb, const void* residual,
                                   void* out, int n_img, int H, int W, int Cin, int Cout, int64_t temb_row_stride,
                                   int temb_img_div, int upsample2x, float* gn_partials, int w_tilemajor, void* stream) {
    if (!gn_partials) FMC_FAIL(FMC_E_NULL, "conv3x3_bf16_gn: NULL gn_partials");
    return conv3x3_impl(x, w, bias, temb, residual, out, n_img, H, W, Cin, Cout, temb_row_stride, temb_img_div, upsample2x, 16, 1,
                        nullptr, 0, stream, 0, gn_partials, w_tilemajor);
}

extern "C" int fmc_conv3x3_x3_f32(const void* x3, const void* w3, const float* bias, const float* temb, const float* residual,
                                  float* out, int n_img, int H, int W, int Cin3, int Cout, int64_t temb_row_stride,
                                  int temb_img_div, int upsample2x, int tile, int split_k,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
    if (Cin3 % 3) FMC_FAIL(FMC_E_SHAPE, "conv3x3_x3_f32: Cin3 = 3 Cin of the split operands (Cin3=%d)", Cin3);
    return conv3x3_impl(x3, w3, bias, temb, residual, out, n_img, H, W, Cin3, Cout, temb_row_stride, temb_img_div, upsample2x, tile,
                        split_k, workspace, workspace_bytes, stream, 1);
}

// ---- fp32 -> split-bf16 x3 operand (see fmc_hip.h) ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows, int C,
                                                           int64_t ld_src, int64_t ld_dst, int col0, int K, int role) {
    const int cpr = C / 8;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * cpr) return;
    const int64_t r = idx / cpr;
    const int c = (int)(idx - r * cpr) * 8;
    float v[8];
    Vec8<float>::load(src + r * ld_src + c, v);
    bf16x8 hi, lo;
    split_bf16x8(v, hi, lo);
    union { bf16x8 v; u32x4 u; } h, l;
    h.v = hi; l.v = lo;
    bf16_t* d = dst + r * ld_dst + col0 + c;
    // activation: [hi | hi | lo];  weight: [hi | lo | hi]  ->  sum over 3K = hi hi' + hi lo' + lo hi'
    *reinterpret_cast<u32x4*>(d) = h.u;
    *reinterpret_cast<u32x4*>(d + K) = role == 0 ? h.u : l.u;
    *reinterpret_cast<u32x4*>(d + 2 * (int64_t)K) = role == 0 ? l.u : h.u;
}
}  // namespace

extern "C" int fmc_split_bf16x3(const float* src, void* dst, int64_t rows, int C, int64_t ld_src, int64_t ld_dst, int col0, int K,
                                int role, void* stream) {
    if (!src || !dst) FMC_FAIL(FMC_E_NULL, "split_bf16x3: NULL tensor");
    if (rows <= 0 || C <= 0 || C % 8 || K % 8 || col0 % 8 || col0 < 0 || col0 + C > K || ld_src % 4 || ld_dst % 8 || ld_dst < 3 * (int64_t)K ||
        (role != 0 && role != 1))
        FMC_FAIL(FMC_E_SHAPE, "split_bf16x3: rows=%lld C=%d K=%d col0=%d ld_src=%lld ld_dst=%lld role=%d", (long long)rows, C, K, col0,
                 (long long)ld_src, (long long)ld_dst, role);
    if (!fmc_aligned16(src) || !fmc_aligned16(dst)) FMC_FAIL(FMC_E_ALIGN, "split_bf16x3: tensors must be 16-byte aligned");
    const int64_t chunks = rows * (C / 8);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst, rows, C, ld_src, ld_dst, col0, K, role);
    FMC_CHECK_LAUNCH("fmc_split_bf16x3");
    return 0;
}

extern "C" int fmc_linear_fp8_qkv(const void* x, const void* w, void* out_fp8, int64_t M, int N, int K, int64_t ldx,
                                  const void* inv_scales, void* amax_bits, void* stream) {
    if (!x || !w || !out_fp8 || !inv_scales) FMC_FAIL(FMC_E_NULL, "linear_fp8_qkv: NULL tensor");
    if (M <= 0 || N <= 0 || K <= 0 || K % BK_MAX || N % 192 || ldx % 8)
        FMC_FAIL(FMC_E_SHAPE, "linear_fp8_qkv: need K%%64==0, N = 3*C with C%%64==0, ldx%%8==0 (M=%lld N=%d K=%d)", (long long)M, N, K);
    if (!fmc_aligned16(x) || !fmc_aligned16(w) || !fmc_aligned16(out_fp8)) FMC_FAIL(FMC_E_ALIGN, "linear_fp8_qkv: tensors must be 16-byte aligned");
    GemmParams P{};
    P.a = (const bf16_t*)x; P.w = (const bf16_t*)w; P.out = (bf16_t*)out_fp8;
    P.M = M; P.N = N; P.K = K; P.lda = ldx; P.ldo = N / 2;            // output rows in 2-byte units (two fp8 each)
    P.hw = 1; P.alpha = 1.f; P.temb_div = 1; P.split_k = 1;
    P.q8_inv = (const float*)inv_scales; P.q8_amax = (unsigned*)amax_bits;
    if (!gemm8_ok(P)) FMC_FAIL(FMC_E_SHAPE, "linear_fp8_qkv: operands beyond 2 GiB");
    launch_gemm8<0, 2, 4>(P, (hipStream_t)stream);
    FMC_CHECK_LAUNCH("fmc_linear_fp8_qkv");
    return 0;
}
