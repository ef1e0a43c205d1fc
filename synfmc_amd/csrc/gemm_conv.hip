// bf16 MFMA GEMM for gfx950 with two A-operand loaders (token matrix / implicit 3x3 convolution on NHWC) and
// fused epilogues.  Replaces, on the FMC path,
//   * nn.Linear of diffusers' Attention / FeedForward / Transformer2D proj_in|out (call sites
//     fmc/models/attention_processor.py:50-69,255-283; fmc/models/motion_module.py:219,228,284) including the
//     `+ residual` that follows them (motion_module.py:289-297, diffusers BasicTransformerBlock) and the GEGLU gate;
//   * the 3x3 convolutions of diffusers' ResnetBlock2D / Downsample2D(stride 1 only here) / Upsample2D
//     (ctor args fmc/models/unet_blocks.py:306-317) with `+ time_emb_proj(silu(temb))[:, :, None, None]` and the
//     `input + h` residual fused into the epilogue.
//
// out[m, n] = epi( sum_k A[m, k] * W[n, k] )      A: [M, K] bf16 (K contiguous), W: [N, K] bf16 (K contiguous)
//   conv mode: m = (img, y, x) pixel of an NHWC image, k = (tap, ci): A[m, k] = X[img, y+dy-1, x+dx-1, ci] (zero
//   outside), W = the filter in channels-last memory format [Cout][3][3][Cin] -- exactly K-contiguous.
//
// Structure (CDNA4, wave = 64): 128x128 output tile per 256-thread workgroup (2x2 waves, 64x64 per wave as 2x2
// v_mfma_f32_32x32x16_bf16 tiles), BK = 64, operands staged global -> registers -> LDS with the loads of tile k+1 in
// flight under the MFMAs of tile k, two LDS stages (one barrier per k-tile), XOR-swizzled 16-byte chunks so that
// both the ds_write_b128 of the staging pass and the ds_read_b128 fragment reads are bank-conflict free, products
// computed "swapped" (MFMA A operand = W rows) so a lane ends up with 4 consecutive output columns, C tile staged
// through LDS and written with full-row 16-byte stores with bias / temb / residual / GEGLU applied on the way out.
// XCD-aware tile order: the N-tiles of one M-tile run on one XCD (A rows come from that L2).
//
// Roofline: MFMA bound for K >= ~640, HBM bound (output write) for the K = 320 level-0 projections.
// Algorithmic flops per launch = 2*M*N*K.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_ELEMS = (BM + BN) * BK;          // bf16 elements per LDS stage (32 KiB)
constexpr int CP = BN + 8;                           // C tile pitch (elements)

struct GemmParams {
    const bf16_t* a; const bf16_t* w; const bf16_t* bias; const bf16_t* temb; const bf16_t* res; bf16_t* out;
    int64_t M; int N, K;
    int64_t lda, ldres, ldo;
    int img_h, img_w, cin, hw;        // conv mode
    float alpha;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ float gelu_erf(float g) { return 0.5f * g * (1.f + erff(g * 0.70710678118654752f)); }

// physical 16-byte chunk of logical chunk c in tile row r (8 chunks per 128-byte row)
__device__ __forceinline__ int swz(int r, int c) { return c ^ ((r >> 1) & 7); }

// MODE 0: token GEMM, 1: implicit 3x3 conv.  EPI 0: (+bias)(+temb)(+residual); 1: GEGLU (weights pre-interleaved so
// that tile columns [0,64) are the value half and [64,128) the matching gate half; out is [M, N/2]).
template <int MODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    // ---- tile order: XCD aware (blockIdx % 8 = XCD): n-tiles of one m-tile stay on one XCD --------------------
    int tile_m, tile_n;
    {
        const int id = blockIdx.x, total = P.tiles_m * P.tiles_n;
        const int per_xcd = (total + 7) / 8;
        const int xcd = id & 7, j = id >> 3;
        int lin = xcd * per_xcd + j;                 // contiguous range of the linear (m-major) order per XCD
        if ((total & 7) != 0) lin = id;              // keep it bijective when 8 does not divide the grid
        tile_m = lin / P.tiles_n;
        tile_n = lin - tile_m * P.tiles_n;
        if (tile_m >= P.tiles_m) return;
    }
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- per-thread staging slots: 4 A chunks + 4 W chunks of 16 bytes per k-tile ----------------------------------
    const int srow = tid >> 3, sc = tid & 7;        // rows srow + 32*j, chunk sc
    const bf16_t* aptr[4];
    bool aval[4];
    int ay[4], ax[4];
    const bf16_t* wptr[4];
    bool wval[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t m = m0 + srow + 32 * j;
        aval[j] = m < P.M;
        if (MODE == 0) {
            aptr[j] = P.a + (aval[j] ? m : 0) * P.lda + sc * 8;
            ay[j] = ax[j] = 0;
        } else {
            const int64_t mm = aval[j] ? m : 0;
            const int pix = (int)(mm % P.hw);
            ay[j] = pix / P.img_w;
            ax[j] = pix - ay[j] * P.img_w;
            aptr[j] = P.a + mm * P.cin + sc * 8;
        }
        const int n = n0 + srow + 32 * j;
        wval[j] = n < P.N;
        wptr[j] = P.w + (int64_t)(wval[j] ? n : 0) * P.K + sc * 8;
    }

    u32x4 ra[4], rb[4];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ra[j] = aval[j] ? *reinterpret_cast<const u32x4*>(aptr[j] + k0) : u32x4{0u, 0u, 0u, 0u};
        } else {
            const int tap = k0 / P.cin, ci0 = k0 - tap * P.cin;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int64_t shift = ((int64_t)dy * P.img_w + dx) * P.cin + ci0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = aval[j] && (unsigned)(ay[j] + dy) < (unsigned)P.img_h &&
                                (unsigned)(ax[j] + dx) < (unsigned)P.img_w;
                ra[j] = ok ? *reinterpret_cast<const u32x4*>(aptr[j] + shift) : u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            rb[j] = wval[j] ? *reinterpret_cast<const u32x4*>(wptr[j] + k0) : u32x4{0u, 0u, 0u, 0u};
    };
    auto lstore = [&](int buf) {
        bf16_t* As = smem + buf * STAGE_ELEMS;
        bf16_t* Ws = As + BM * BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = srow + 32 * j;
            *reinterpret_cast<u32x4*>(As + r * BK + swz(r, sc) * 8) = ra[j];
            *reinterpret_cast<u32x4*>(Ws + r * BK + swz(r, sc) * 8) = rb[j];
        }
    };

    f32x16 acc[2][2];                                // [ni][mi]: rows = n (registers), cols = m (lanes)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = P.K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        const bf16_t* As = smem + (kt & 1) * STAGE_ELEMS;
        const bf16_t* Ws = As + BM * BK;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 wf[2], af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rw = wn * 64 + i * 32 + l31;
                const int rm = wm * 64 + i * 32 + l31;
                union { bf16x8 v; u32x4 u; } tw, ta;
                tw.u = *reinterpret_cast<const u32x4*>(Ws + rw * BK + swz(rw, 2 * ks + half) * 8);
                ta.u = *reinterpret_cast<const u32x4*>(As + rm * BK + swz(rm, 2 * ks + half) * 8);
                wf[i] = tw.v;
                af[i] = ta.v;
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: C tile -> LDS (bf16 would lose the fp32 sum before bias/residual: keep fp32 pairs packed later)
    // The tile is staged as fp32-accurate bf16 AFTER adding nothing; bias / temb / residual are added in fp32 on the
    // way out from a second fp32 staging would cost 64 KiB, so the sum is rounded once here and once at the store:
    // instead we stage fp32 in two halves of 64 rows to keep full precision until the final rounding.
    float* Cs = reinterpret_cast<float*>(smem_raw);             // [64][CP] fp32 = 34 KiB per half
#pragma unroll
    for (int hm = 0; hm < 2; ++hm) {
        __syncthreads();
        if (wm == hm) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int m = mi * 32 + l31;
                        const int n = wn * 64 + ni * 32 + 8 * g + 4 * half;
                        *reinterpret_cast<f32x4*>(Cs + m * CP + n) =
                            f32x4{acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]};
                    }
        }
        __syncthreads();
        if (EPI == 0) {
            // 64 rows x 16 chunks of 8 columns; thread: chunk = tid & 15, rows tid>>4 + 16*i
            const int ch = tid & 15, n = n0 + ch * 8;
            if (n < P.N) {
                float bv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) bv[i] = P.bias ? bf2f(P.bias[n + i]) : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = (tid >> 4) + 16 * i;
                    const int64_t m = m0 + hm * 64 + r;
                    if (m >= P.M) continue;
                    float v[8];
                    Vec8<float>::load(Cs + r * CP + ch * 8, v);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (v[k] + bv[k]) * P.alpha;
                    if (MODE == 1 && P.temb) {
                        float t[8];
                        Vec8<bf16_t>::load(P.temb + (m / P.hw) * P.N + n, t);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += t[k];
                    }
                    if (P.res) {
                        float t[8];
                        Vec8<bf16_t>::load(P.res + m * P.ldres + n, t);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += t[k];
                    }
                    Vec8<bf16_t>::store(P.out + m * P.ldo + n, v);
                }
            }
        } else {
            // GEGLU: value columns [0,64), gate columns [64,128) of the tile -> 64 output columns
            const int ch = tid & 7, no = tile_n * 64 + ch * 8;      // output column
            if (no < P.N / 2) {
                float ba[8], bg[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    ba[i] = P.bias ? bf2f(P.bias[n0 + ch * 8 + i]) : 0.f;
                    bg[i] = P.bias ? bf2f(P.bias[n0 + 64 + ch * 8 + i]) : 0.f;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = (tid >> 3) + 32 * i;
                    const int64_t m = m0 + hm * 64 + r;
                    if (m >= P.M) continue;
                    float a[8], g[8];
                    Vec8<float>::load(Cs + r * CP + ch * 8, a);
                    Vec8<float>::load(Cs + r * CP + 64 + ch * 8, g);
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[k] = (a[k] + ba[k]) * gelu_erf(g[k] + bg[k]);
                    Vec8<bf16_t>::store(P.out + m * P.ldo + no, a);
                }
            }
        }
    }
}

template <int MODE, int EPI>
int launch_gemm(GemmParams& P, hipStream_t st) {
    P.tiles_m = (int)((P.M + BM - 1) / BM);
    P.tiles_n = (P.N + BN - 1) / BN;
    const int total = P.tiles_m * P.tiles_n;
    const int grid = (total % 8 == 0) ? total : total;      // non-multiples of 8 use the identity map in-kernel
    const size_t lds = (size_t)2 * STAGE_ELEMS * sizeof(bf16_t);   // 64 KiB; also covers the 34 KiB fp32 C staging
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<MODE, EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        raised = true;
    }
    hipLaunchKernelGGL((gemm_kernel<MODE, EPI>), dim3(grid), dim3(256), lds, st, P);
    return 0;
}

}  // namespace

extern "C" int fmc_linear_bf16(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M,
                               int N, int K, int64_t ldx, int64_t ldres, int64_t ldo, float alpha, int epilogue, void* stream) {
    if (!x || !w || !out) FMC_FAIL(FMC_E_NULL, "linear_bf16: NULL tensor");
    if (M <= 0 || N <= 0 || K <= 0 || K % BK || N % 8 || ldx % 8 || ldo % 8 || (residual && ldres % 8))
        FMC_FAIL(FMC_E_SHAPE, "linear_bf16: need K%%64==0, N%%8==0 and strides %%8==0 (M=%lld N=%d K=%d)", (long long)M, N, K);
    if (epilogue != 0 && epilogue != 1) FMC_FAIL(FMC_E_SHAPE, "linear_bf16: epilogue %d", epilogue);
    if (epilogue == 1 && (N % BN || residual)) FMC_FAIL(FMC_E_SHAPE, "linear_bf16: GEGLU needs N%%128==0 and no residual");
    if (!fmc_aligned16(x) || !fmc_aligned16(w) || !fmc_aligned16(out) || (residual && !fmc_aligned16(residual)) ||
        (bias && !fmc_aligned16(bias)))
        FMC_FAIL(FMC_E_ALIGN, "linear_bf16: tensors must be 16-byte aligned");
    GemmParams P{};
    P.a = (const bf16_t*)x; P.w = (const bf16_t*)w; P.bias = (const bf16_t*)bias; P.temb = nullptr;
    P.res = (const bf16_t*)residual; P.out = (bf16_t*)out;
    P.M = M; P.N = N; P.K = K; P.lda = ldx; P.ldres = ldres; P.ldo = ldo;
    P.img_h = P.img_w = P.cin = 0; P.hw = 1; P.alpha = alpha;
    hipStream_t st = (hipStream_t)stream;
    if (epilogue == 0) launch_gemm<0, 0>(P, st); else launch_gemm<0, 1>(P, st);
    FMC_CHECK_LAUNCH("fmc_linear_bf16");
    return 0;
}

extern "C" int fmc_conv3x3_bf16(const void* x, const void* w, const void* bias, const void* temb, const void* residual,
                                void* out, int n_img, int H, int W, int Cin, int Cout, void* stream) {
    if (!x || !w || !out) FMC_FAIL(FMC_E_NULL, "conv3x3_bf16: NULL tensor");
    if (n_img <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % BK || Cout % 8)
        FMC_FAIL(FMC_E_SHAPE, "conv3x3_bf16: need Cin%%64==0 and Cout%%8==0 (Cin=%d Cout=%d)", Cin, Cout);
    if (!fmc_aligned16(x) || !fmc_aligned16(w) || !fmc_aligned16(out) || (residual && !fmc_aligned16(residual)) ||
        (temb && !fmc_aligned16(temb)) || (bias && !fmc_aligned16(bias)))
        FMC_FAIL(FMC_E_ALIGN, "conv3x3_bf16: tensors must be 16-byte aligned");
    GemmParams P{};
    P.a = (const bf16_t*)x; P.w = (const bf16_t*)w; P.bias = (const bf16_t*)bias; P.temb = (const bf16_t*)temb;
    P.res = (const bf16_t*)residual; P.out = (bf16_t*)out;
    P.M = (int64_t)n_img * H * W; P.N = Cout; P.K = 9 * Cin; P.lda = Cin; P.ldres = Cout; P.ldo = Cout;
    P.img_h = H; P.img_w = W; P.cin = Cin; P.hw = H * W; P.alpha = 1.f;
    launch_gemm<1, 0>(P, (hipStream_t)stream);
    FMC_CHECK_LAUNCH("fmc_conv3x3_bf16");
    return 0;
}
