// GroupNorm(+SiLU) fwd/bwd, LayerNorm(+PE), GEGLU -- HBM-bound channels-last passes for gfx950.
//
// Layout: activations are [N, HW, C] with C contiguous.  A thread always owns the same 8
// consecutive channels (one 16-byte bf16 load) and walks rows, so every wave-level load is a run
// of full rows (coalesced) and the per-channel affine terms live in registers.
//
// Roofline: all kernels here are HBM bound.  Algorithmic bytes:
//   groupnorm_silu_fwd : 2 * N*HW*C * e   (read x + write y; the second read of x for the apply
//                                          pass is expected to hit L2 / Infinity Cache)
//   layernorm_fwd      : 2 * M*C * e      geglu_fwd : 3 * M*Cff * e
#include "common.h"

namespace {

constexpr int GN_MAX_SPLIT = 64;
constexpr int GN_MAX_G = 64;

struct GnGeom {
    int tpr;    // threads per row  = C / 8
    int rpi;    // rows per block iteration
    int block;  // tpr * rpi
    int split;  // blocks per sample
    int rows_per_split;
};

constexpr int GN_U = 4;          // rows per trip of the two-pass kernels = rows per thread: every load of a thread in flight at once

GnGeom gn_geom(int HW, int C) {
    GnGeom g;
    g.tpr = C / 8;
    g.rpi = 512 / g.tpr;
    if (g.rpi < 1) g.rpi = 1;
    if (g.rpi > HW) g.rpi = HW;
    g.block = g.tpr * g.rpi;
    int rows_target = 8 * g.rpi;  // every thread sees ~8 rows
    g.split = (HW + rows_target - 1) / rows_target;
    if (g.split > GN_MAX_SPLIT) g.split = GN_MAX_SPLIT;
    if (g.split < 1) g.split = 1;
    g.rows_per_split = (HW + g.split - 1) / g.split;
    g.split = (HW + g.rows_per_split - 1) / g.rows_per_split;
    return g;
}

// z * sigmoid(z); the hardware reciprocal (1 ulp) instead of an IEEE division (~10 VALU ops per element)
__device__ __forceinline__ float silu_f(float z) { return z * __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
__device__ __forceinline__ float dsilu_f(float z) {
    float s = 1.f / (1.f + __expf(-z));
    return s * (1.f + z * (1.f - s));
}

// --------------------------------------------------------------------------------------------
// pass 1: per (sample, split, group) partial sums.  MODE 0: (sum x, sum x^2).
// MODE 1 (backward): with z = a*x+b, dxh = dy*act'(z)*gamma, xh = (x-mean)*rstd:
//                    (sum dxh, sum dxh*xh).
// --------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ void gn_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, const float* __restrict__ stats,
                                  float* __restrict__ part, int HW, int C, int G, int tpr, int rpi,
                                  int rows_per_split, int act, const T* __restrict__ x2 = nullptr, int C1 = 0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][rpi][C]
    const int n = blockIdx.y, s = blockIdx.x, nsplit = gridDim.x;
    const int tid = threadIdx.x;
    const int cc = tid % tpr, rsub = tid / tpr;
    const int c0 = cc * 8;
    const int cpg = C / G;
    const int row0 = s * rows_per_split;
    int row1 = row0 + rows_per_split;
    if (row1 > HW) row1 = HW;

    float aco[8], bco[8], mu[8], rs[8], gm[8];
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int c = c0 + i, g = c / cpg;
            mu[i] = stats[((size_t)n * G + g) * 2];
            rs[i] = stats[((size_t)n * G + g) * 2 + 1];
            gm[i] = gamma[c];
            aco[i] = rs[i] * gm[i];
            bco[i] = beta[c] - mu[i] * aco[i];
        }
    }
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    // two-source input (channels [0, C1) from x, [C1, C) from x2): the channel concat of the up blocks is never written
    int xs = C;
    const T* xb = x + (size_t)n * HW * C + c0;
    if (x2) {
        xs = c0 < C1 ? C1 : C - C1;
        xb = c0 < C1 ? x + (size_t)n * HW * C1 + c0 : x2 + (size_t)n * HW * (C - C1) + (c0 - C1);
    }
    const T* dyb = (MODE == 1) ? dy + (size_t)n * HW * C + c0 : nullptr;
    if (MODE == 0) {
        for (int r = row0 + rsub; r < row1; r += GN_U * rpi) {    // GN_U rows per trip, all loads first
            float v4[GN_U][8];
#pragma unroll
            for (int u = 0; u < GN_U; ++u) {
                const int rr = r + u * rpi;
                Vec8<T>::load(xb + (size_t)(rr < row1 ? rr : row1 - 1) * xs, v4[u]);
            }
#pragma unroll
            for (int u = 0; u < GN_U; ++u) {
                const float m = (r + u * rpi) < row1 ? 1.f : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float t = v4[u][i] * m; s1[i] += t; s2[i] += t * v4[u][i]; }
            }
        }
    }
    for (int r = row0 + rsub; MODE == 1 && r < row1; r += rpi) {
        float v[8];
        Vec8<T>::load(xb + (size_t)r * C, v);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { s1[i] += v[i]; s2[i] += v[i] * v[i]; }
        } else {
            float d[8];
            Vec8<T>::load(dyb + (size_t)r * C, d);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float z = v[i] * aco[i] + bco[i];
                float dz = act ? d[i] * dsilu_f(z) : d[i];
                float dxh = dz * gm[i];
                float xh = (v[i] - mu[i]) * rs[i];
                s1[i] += dxh;
                s2[i] += dxh * xh;
            }
        }
    }
    float* l1 = smem;
    float* l2 = smem + (size_t)rpi * C;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        l1[rsub * C + c0 + i] = s1[i];
        l2[rsub * C + c0 + i] = s2[i];
    }
    __syncthreads();
    for (int c = tid; c < C; c += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < rpi; ++r) { a += l1[r * C + c]; b += l2[r * C + c]; }
        l1[c] = a;  // row 0 of each plane now holds the per-channel totals
        l2[c] = b;
    }
    __syncthreads();
    for (int g = tid; g < G; g += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += l1[c]; b += l2[c]; }
        float* p = part + (((size_t)n * nsplit + s) * G + g) * 2;
        p[0] = a;
        p[1] = b;
    }
}

// --------------------------------------------------------------------------------------------
// pass 2 (forward): combine partials -> mean / rstd, then y = act(x * a_c + b_c)
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ void gn_apply_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ part,
                                    float* __restrict__ stats, int HW, int C, int G, int tpr, int rpi,
                                    int rows_per_split, float eps, int act, const T* __restrict__ x2 = nullptr,
                                    int C1 = 0, int part_splits = 0) {
    __shared__ float sh_mean[GN_MAX_G], sh_rstd[GN_MAX_G];
    __shared__ float sh_part[GN_MAX_SPLIT * GN_MAX_G * 2];
    // part_splits > 0: the partial pairs come from the PRODUCER of x (a GEMM / conv epilogue, one slab per 160-row tile of the image)
    const int n = blockIdx.y, s = blockIdx.x, nsplit = part_splits > 0 ? part_splits : (int)gridDim.x;
    const int tid = threadIdx.x;
    const int cpg = C / G;
    // the nsplit x G partial pairs of this sample: ONE global round trip for the whole workgroup (every thread fetches its share),
    // then a fixed-order sum per group out of LDS.  (First version: thread g walked its nsplit partials through a loop of dependent
    // global loads -- ~27 L2 round trips, ~10 us, in front of every workgroup's streaming part: a third of the launch.)
    for (int i = tid; i < nsplit * G * 2; i += blockDim.x) sh_part[i] = part[(size_t)n * nsplit * G * 2 + i];
    __syncthreads();
    for (int g = tid; g < G; g += blockDim.x) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < nsplit; ++k) {
            const float* p = sh_part + ((size_t)k * G + g) * 2;
            a += (double)p[0];
            b += (double)p[1];
        }
        double cnt = (double)HW * cpg;
        double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        float rstd = (float)(1.0 / sqrt(var + (double)eps));
        sh_mean[g] = (float)mean;
        sh_rstd[g] = rstd;
        if (s == 0) {
            stats[((size_t)n * G + g) * 2] = (float)mean;
            stats[((size_t)n * G + g) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    const int cc = tid % tpr, rsub = tid / tpr;
    const int c0 = cc * 8;
    float aco[8], bco[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = c0 + i, g = c / cpg;
        aco[i] = sh_rstd[g] * gamma[c];
        bco[i] = beta[c] - sh_mean[g] * aco[i];
    }
    const int row0 = s * rows_per_split;
    int row1 = row0 + rows_per_split;
    if (row1 > HW) row1 = HW;
    int xs = C;
    const T* xb = x + (size_t)n * HW * C + c0;
    if (x2) {
        xs = c0 < C1 ? C1 : C - C1;
        xb = c0 < C1 ? x + (size_t)n * HW * C1 + c0 : x2 + (size_t)n * HW * (C - C1) + (c0 - C1);
    }
    T* yb = y + (size_t)n * HW * C + c0;
    // GN_U rows per trip, loads first (branch-free, clamped): a row per trip leaves one 16-byte load in flight per thread
    for (int r = row0 + rsub; r < row1; r += GN_U * rpi) {
        float v[GN_U][8];
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int rr = r + u * rpi;
            Vec8<T>::load(xb + (size_t)(rr < row1 ? rr : row1 - 1) * xs, v[u]);
        }
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int rr = r + u * rpi;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float z = v[u][i] * aco[i] + bco[i];
                v[u][i] = act ? silu_f(z) : z;
            }
            if (rr < row1) Vec8<T>::store(yb + (size_t)rr * C, v[u]);
        }
    }
}

// --------------------------------------------------------------------------------------------
// single-pass forward for the small levels: one workgroup owns ALL pixels of (image n, a slab of GPB whole groups =
// CS channels), keeps them in registers (<= GN1_MAXCH 8-element chunks per thread), reduces, normalises, writes.  x is
// read once and there is one launch instead of two -- at [32, 160, 1280] the two-pass pair costs 20 us against a 6 us
// copy.  Thread (pl, ch) always handles the same 8 channels (chunk ch of the slab) of pixels pl, pl+PL, ...
// --------------------------------------------------------------------------------------------
constexpr int GN1_MAXCH = 16;

template <typename T, int MAXCH>
__global__ __launch_bounds__(256) void gn_fused_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ stats, int HW, int C, int G, int CS, int PL,
                                                           float eps, int act, const T* __restrict__ x2, int C1, int xcd_order) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][PL][CS] partial sums, then [2][CS] totals
    __shared__ float sh_mean[GN_MAX_G], sh_rstd[GN_MAX_G];
    // blockIdx -> (image, slab): workgroup ids go round-robin over the 8 XCDs, and a slab is an 80..160-byte run of every pixel row,
    // i.e. it shares its 128-byte lines with the neighbouring slabs of the SAME image -- so all slabs of an image run on one XCD
    // (ids = xcd mod 8) and its L2 fetches every line once; slab-fastest order had each line fetched by up to 8 L2s
    // (52 MB level-1 launch: 20.6 -> see DESIGN us)
    const int nslab = C / CS, tid = threadIdx.x;
    int n, slab;
    if (xcd_order) {                              // (an explicit flag: a 2-D grid of ONE image also has gridDim.y == 1)
        const int id = blockIdx.x, xcd = id & 7, within = id >> 3;
        n = (within / nslab) * 8 + xcd;
        slab = within % nslab;
    } else {
        n = blockIdx.y;
        slab = blockIdx.x;
    }
    const int CPS = CS / 8, cpg = C / G, GPB = CS / cpg;
    const int ch = tid % CPS, pl = tid / CPS;
    const int c0 = slab * CS + ch * 8;                             // my 8 channels (global index)
    int xs = C;
    const T* xb = x + (size_t)n * HW * C + c0;
    if (x2) {
        xs = c0 < C1 ? C1 : C - C1;
        xb = c0 < C1 ? x + (size_t)n * HW * C1 + c0 : x2 + (size_t)n * HW * (C - C1) + (c0 - C1);
    }
    // the slab stays in registers in its STORAGE type (bf16: 4 registers per 8-element chunk, unpacked where used; as fp32 the 16-chunk
    // instantiation needed 182 VGPRs = 2 waves per SIMD, and a workgroup's load burst, reduction and store burst do not overlap with
    // anything but other workgroups)
    typename Vec8<T>::raw_t v[MAXCH];
    float gm[8], bt[8];                                            // requested with the slab, not behind the reduction
    Vec8<float>::load(gamma + c0, gm);
    Vec8<float>::load(beta + c0, bt);
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    // branch-free loads (clamped row) so that all of them are in flight together: a load inside `if (p < HW)` is
    // waited for inside its own branch region, i.e. one memory round trip per chunk
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int p = pl + PL * j;
        v[j] = Vec8<T>::load_raw(xb + (size_t)(p < HW ? p : HW - 1) * xs);
    }
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const float m = (pl + PL * j) < HW ? 1.f : 0.f;
        float f[8];
        Vec8<T>::unpack(v[j], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float t = f[i] * m; s1[i] += t; s2[i] += t * f[i]; }
    }
    float* l1 = smem;
    float* l2 = smem + (size_t)PL * CS;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        l1[pl * CS + ch * 8 + i] = s1[i];
        l2[pl * CS + ch * 8 + i] = s2[i];
    }
    __syncthreads();
    for (int c = tid; c < CS; c += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < PL; ++r) { a += l1[r * CS + c]; b += l2[r * CS + c]; }
        l1[c] = a;
        l2[c] = b;
    }
    __syncthreads();
    if (tid < GPB) {
        double a = 0.0, b = 0.0;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += (double)l1[c]; b += (double)l2[c]; }
        const double cnt = (double)HW * cpg, mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        sh_mean[tid] = (float)mean;
        sh_rstd[tid] = rstd;
        const int g = slab * GPB + tid;
        stats[((size_t)n * G + g) * 2] = (float)mean;
        stats[((size_t)n * G + g) * 2 + 1] = rstd;
    }
    __syncthreads();
    float aco[8], bco[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gl = (ch * 8 + i) / cpg;
        aco[i] = sh_rstd[gl] * gm[i];
        bco[i] = bt[i] - sh_mean[gl] * aco[i];
    }
    T* yb = y + (size_t)n * HW * C + c0;
    if constexpr (sizeof(T) == 2) {
        // (the raw words are "modified" here as far as the optimiser knows: it must unpack them again below instead of keeping the
        // fp32 values of the statistics pass alive across the barriers)
#pragma unroll
        for (int j = 0; j < MAXCH; ++j) asm volatile("" : "+v"(v[j]));
    }
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int p = pl + PL * j;
        if (p < HW) {
            float o[8], f[8];
            Vec8<T>::unpack(v[j], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float z = f[i] * aco[i] + bco[i];
                o[i] = act ? silu_f(z) : z;
            }
            Vec8<T>::store(yb + (size_t)p * C, o);
        }
    }
}

// slab choice for the single-pass kernel: GPB whole groups per workgroup such that the slab is a whole number of
// 16-byte chunks and all HW pixels of it fit the register budget; longest contiguous runs first, then most workgroups
struct Gn1Geom { int CS, PL, block, chunks; };
static bool gn1_geom(int HW, int C, int G, Gn1Geom& out) {
    const int cpg = C / G;
    int best = 0, best_score = -1;
    for (int gpb = 1; gpb <= G; ++gpb) {
        if (G % gpb || gpb > GN_MAX_G) continue;
        const int cs = gpb * cpg;
        if (cs % 8) continue;
        const int cps = cs / 8;
        if (cps > 256) break;
        const int pl = 256 / cps < HW ? 256 / cps : HW;
        const int chunks = (HW + pl - 1) / pl;
        if (chunks > GN1_MAXCH) continue;
        if ((size_t)2 * pl * cs * sizeof(float) > 60 * 1024) continue;
        // contiguous bytes per pixel (as bf16) up to 160 count; beyond that prefer small slabs: more workgroups than CUs
        // (so one's load burst overlaps another's arithmetic) and few chunks per thread (registers -> occupancy)
        const int run = cs * 2 < 160 ? cs * 2 : 160;
        const int score = run * 1024 - chunks * 8 - gpb;
        if (score > best_score) { best_score = score; best = gpb; }
    }
    if (!best) return false;
    out.CS = best * cpg;
    const int cps = out.CS / 8;
    out.PL = 256 / cps < HW ? 256 / cps : HW;
    out.block = cps * out.PL;
    out.chunks = (HW + out.PL - 1) / out.PL;
    return true;
}

template <typename T>
static void launch_gn1(const Gn1Geom& g1, const void* x, void* y, const float* gamma, const float* beta, float* stats, int N,
                       int HW, int C, int G, float eps, int act, hipStream_t st, const void* x2, int C1) {
    dim3 grid1(C / g1.CS, N), block1(g1.block);
    const int xcd_order = N % 8 == 0;
    if (xcd_order) grid1 = dim3((unsigned)(C / g1.CS) * N, 1);        // XCD-aware 1-D order (see the kernel)
    const size_t lds1 = (size_t)2 * g1.PL * g1.CS * sizeof(float);
    if (g1.chunks <= 8)
        hipLaunchKernelGGL((gn_fused_fwd_kernel<T, 8>), grid1, block1, lds1, st, (const T*)x, (T*)y, gamma, beta, stats, HW, C,
                           G, g1.CS, g1.PL, eps, act, (const T*)x2, C1, xcd_order);
    else
        hipLaunchKernelGGL((gn_fused_fwd_kernel<T, GN1_MAXCH>), grid1, block1, lds1, st, (const T*)x, (T*)y, gamma, beta,
                           stats, HW, C, G, g1.CS, g1.PL, eps, act, (const T*)x2, C1, xcd_order);
}


// pass 2 (backward): dx = rstd * (dxh - S1/cnt - xh * S2/cnt)
template <typename T>
__global__ void gn_apply_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ stats, const float* __restrict__ part, int HW, int C,
                                    int G, int tpr, int rpi, int rows_per_split, int act, const T* __restrict__ addend = nullptr) {
    // addend (same shape as dx): the gradient that reaches x along its OTHER use, the skip connection around the normalised branch
    // (`h + f(norm(h))`): dx = addend + dX(norm) in this pass instead of a separate elementwise add per residual connection.
    __shared__ float sh_m1[GN_MAX_G], sh_m2[GN_MAX_G];
    __shared__ float sh_part[GN_MAX_SPLIT * GN_MAX_G * 2];
    const int n = blockIdx.y, s = blockIdx.x, nsplit = gridDim.x;
    const int tid = threadIdx.x;
    const int cpg = C / G;
    // the sample's partial pairs in ONE global round trip (as in gn_apply_fwd_kernel; a chain of nsplit dependent loads per group before)
    for (int i = tid; i < nsplit * G * 2; i += blockDim.x) sh_part[i] = part[(size_t)n * nsplit * G * 2 + i];
    __syncthreads();
    for (int g = tid; g < G; g += blockDim.x) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < nsplit; ++k) {
            const float* p = sh_part + ((size_t)k * G + g) * 2;
            a += (double)p[0];
            b += (double)p[1];
        }
        double cnt = (double)HW * cpg;
        sh_m1[g] = (float)(a / cnt);
        sh_m2[g] = (float)(b / cnt);
    }
    __syncthreads();
    const int cc = tid % tpr, rsub = tid / tpr;
    const int c0 = cc * 8;
    float aco[8], bco[8], mu[8], rs[8], gm[8], m1[8], m2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = c0 + i, g = c / cpg;
        mu[i] = stats[((size_t)n * G + g) * 2];
        rs[i] = stats[((size_t)n * G + g) * 2 + 1];
        gm[i] = gamma[c];
        aco[i] = rs[i] * gm[i];
        bco[i] = beta[c] - mu[i] * aco[i];
        m1[i] = sh_m1[g];
        m2[i] = sh_m2[g];
    }
    const int row0 = s * rows_per_split;
    int row1 = row0 + rows_per_split;
    if (row1 > HW) row1 = HW;
    const size_t base = (size_t)n * HW * C + c0;
    // two rows per trip, loads first (branch-free, clamped): one row per trip left two 16-byte loads in flight per thread
    for (int r = row0 + rsub; r < row1; r += 2 * rpi) {
        float v[2][8], d[2][8], ad[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rr = r + u * rpi < row1 ? r + u * rpi : row1 - 1;
            Vec8<T>::load(x + base + (size_t)rr * C, v[u]);
            Vec8<T>::load(dy + base + (size_t)rr * C, d[u]);
            if (addend) Vec8<T>::load(addend + base + (size_t)rr * C, ad[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rr = r + u * rpi;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float z = v[u][i] * aco[i] + bco[i];
                float dz = act ? d[u][i] * dsilu_f(z) : d[u][i];
                float dxh = dz * gm[i];
                float xh = (v[u][i] - mu[i]) * rs[i];
                v[u][i] = rs[i] * (dxh - m1[i] - xh * m2[i]);
                if (addend) v[u][i] += ad[u][i];
            }
            if (rr < row1) Vec8<T>::store(dx + base + (size_t)rr * C, v[u]);
        }
    }
}

// --------------------------------------------------------------------------------------------
// LayerNorm (+ positional-encoding add): one wave normalises R token rows at a time; all R rows are
// loaded (16 B per lane per chunk) before the first reduction so that R*NCH loads are in flight per lane.
// --------------------------------------------------------------------------------------------
template <typename T, int NCH, int R>  // NCH = ceil(C/8/64) chunks of 8 per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ pe, int64_t M, int C,
                                 float eps, int pe_inner, int pe_frames) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int nchunks = C / 8;
    const float inv_c = 1.f / (float)C;
    for (int64_t row0 = ((int64_t)blockIdx.x * wpb + wave) * R; row0 < M; row0 += (int64_t)gridDim.x * wpb * R) {
        float v[R][NCH][8];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int ch = lane + 64 * k;
                if (row < M && ch < nchunks) Vec8<T>::load(x + row * C + ch * 8, v[r][k]);
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[r][k][i] = 0.f;
                }
            }
        }
        float mean[R], rstd[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) s += v[r][k][i];
            mean[r] = s;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < R; ++r) mean[r] += __shfl_xor(mean[r], o, 64);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            mean[r] *= inv_c;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if (lane + 64 * k < nchunks) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { float d = v[r][k][i] - mean[r]; q += d * d; }
                }
            }
            rstd[r] = q;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < R; ++r) rstd[r] += __shfl_xor(rstd[r], o, 64);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + 64 * k;
            if (ch >= nchunks) continue;
            float gm[8], bt[8];
            Vec8<float>::load(gamma + ch * 8, gm);
            Vec8<float>::load(beta + ch * 8, bt);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = row0 + r;
                if (row >= M) continue;
                const float rs = rsqrtf(rstd[r] * inv_c + eps);
                const float* per = pe ? pe + (size_t)((row / pe_inner) % pe_frames) * C + ch * 8 : nullptr;
                float o[8], pv[8];
                if (per) Vec8<float>::load(per, pv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    o[i] = (v[r][k][i] - mean[r]) * rs * gm[i] + bt[i];
                    if (per) o[i] += pv[i];
                }
                Vec8<T>::store(y + row * C + ch * 8, o);
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// LayerNorm, LPR = C/40 lanes per row (C = 320 / 640 / 1280: every FMC width; 5 chunks of 8 elements per lane).
// The kernel above leaves 24 of 64 lanes idle at C = 320 and reduces over the whole wave through the LDS crossbar; here
// a wave works on (64/LPR) x U rows at once with every lane active, a row's lanes read LPR*16 contiguous bytes per
// load, and the two row reductions are 3-4 DPP steps (+ one cross-row shuffle at LPR = 32).
// --------------------------------------------------------------------------------------------
template <int LPR> __device__ __forceinline__ float group_sum(float v) {   // sum over LPR consecutive lanes, in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    if (LPR >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    if (LPR >= 32) v += __shfl_xor(v, 16, 64);
    return v;
}

template <typename T> __device__ __forceinline__ void round_as_stored(float (&v)[8]);
template <> __device__ __forceinline__ void round_as_stored<float>(float (&)[8]) {}
template <> __device__ __forceinline__ void round_as_stored<bf16_t>(float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned r = pack_bf2(v[2 * i], v[2 * i + 1]);
        v[2 * i] = __uint_as_float(r << 16);
        v[2 * i + 1] = __uint_as_float(r & 0xffff0000u);
    }
}

template <typename T, int LPR, int U>   // 5 eight-element chunks per lane, U rows per lane group
__global__ __launch_bounds__(256) void layernorm_g_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ pe, int64_t M, float eps, int pe_inner,
                                                          int pe_frames, const T* __restrict__ addend = nullptr, T* __restrict__ sum_out = nullptr) {
    // addend / sum_out (both or neither): the rows normalised are h = round_T(x + addend) and h is written to sum_out -- the residual add
    // that a vendor-library projection leaves to a separate elementwise launch, done where its result is consumed (fmc_layernorm_add_fwd)
    constexpr int NCH = 5, C = LPR * NCH * 8, RPW = 64 / LPR;
    // gamma / beta through LDS: requested together with the rows (read from global memory behind the statistics they were a second
    // memory round trip in the life of every wave, which is one load burst, a reduction and one store burst long)
    __shared__ __attribute__((aligned(16))) float sh_g[C], sh_b[C];
    // positional-encoding rows the same way: the 4 * RPW * U consecutive rows of a workgroup lie in at most two frames (pe_inner rows
    // per frame, >= the workgroup's rows whenever two frames suffice; otherwise the rows read pe from global memory as before)
    __shared__ __attribute__((aligned(16))) float sh_pe[2][C];
    const int64_t wg_row0 = (int64_t)blockIdx.x * (4 * RPW * U);
    const int64_t wg_f0 = pe ? wg_row0 / pe_inner : 0;
    const int64_t pe_next = (wg_f0 + 1) * pe_inner;           // first row of the workgroup's second frame
    const bool pe_lds = pe && pe_inner >= 4 * RPW * U && 4 * RPW * U >= 32;   // (8 rows per workgroup at C = 1280: staging two PE rows costs as much as the rows)
    const int lane = threadIdx.x & 63, t = lane % LPR, g = lane / LPR;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t row0 = wave * (RPW * U) + g;               // this group's rows: row0 + RPW*u
    float v[U][NCH][8];
    // every chunk of the rows requested in ONE burst from clamped addresses (rows past M repeat the last row and are never stored): a load inside
    // `if (row < M)` is followed by its own s_waitcnt vmcnt(0) -- five dependent round trips, 9 of the 10.7 us this launch took at every size
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t row = row0 + RPW * u, rowc = row < M ? row : M - 1;
#pragma unroll
        for (int j = 0; j < NCH; ++j) Vec8<T>::load(x + rowc * C + (j * LPR + t) * 8, v[u][j]);
    }
    {   // gamma / beta / the two positional-encoding rows -> LDS, requested BEHIND the rows in the same burst (every value loaded before the first is
        // stored: a rolled `sh[i] = g[i]` loop was one round trip per 256 elements, five of them at C = 1280, ahead of everything else)
        constexpr int NI = (C + 255) / 256;
        float gv[NI], bv[NI], pv[2 * NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = min((int)threadIdx.x + 256 * i, C - 1);
            gv[i] = gamma[idx];
            bv[i] = beta[idx];
        }
        if (pe_lds) {
#pragma unroll
            for (int i = 0; i < 2 * NI; ++i) {
                const int idx = min((int)threadIdx.x + 256 * i, 2 * C - 1), which = idx / C, c = idx - which * C;
                pv[i] = pe[(size_t)((wg_f0 + which) % pe_frames) * C + c];
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = (int)threadIdx.x + 256 * i;
            if (idx < C) { sh_g[idx] = gv[i]; sh_b[idx] = bv[i]; }
        }
        if (pe_lds) {
#pragma unroll
            for (int i = 0; i < 2 * NI; ++i) {
                const int idx = (int)threadIdx.x + 256 * i;
                if (idx < 2 * C) sh_pe[idx / C][idx % C] = pv[i];
            }
        }
    }
    float pg[U][NCH][8];                                     // positional-encoding chunks of my rows where they do not come through LDS: same burst
    if (pe && !pe_lds) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = row0 + RPW * u, rowc = row < M ? row : M - 1;
            const float* prow = pe + (size_t)((rowc / pe_inner) % pe_frames) * C;
#pragma unroll
            for (int j = 0; j < NCH; ++j) Vec8<float>::load(prow + (j * LPR + t) * 8, pg[u][j]);
        }
    }
    if (addend) {                                            // (kernel-uniform)
        float a8[U][NCH][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = row0 + RPW * u, rowc = row < M ? row : M - 1;
#pragma unroll
            for (int j = 0; j < NCH; ++j) Vec8<T>::load(addend + rowc * C + (j * LPR + t) * 8, a8[u][j]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = row0 + RPW * u;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[u][j][i] += a8[u][j][i];
                if (row < M) Vec8<T>::store(sum_out + row * C + (j * LPR + t) * 8, v[u][j]);
                round_as_stored<T>(v[u][j]);                 // the norm sees the stored (rounded) sum, as a separate pass would
            }
        }
    }
    float mean[U], rs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[u][j][i];
        mean[u] = group_sum<LPR>(s) * (1.f / C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[u][j][i] - mean[u]; q += d * d; }
        rs[u] = rsqrtf(group_sum<LPR>(q) * (1.f / C) + eps);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        float gm[8], bt[8];
        Vec8<float>::load(sh_g + (j * LPR + t) * 8, gm);
        Vec8<float>::load(sh_b + (j * LPR + t) * 8, bt);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = row0 + RPW * u;
            if (row >= M) continue;
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (v[u][j][i] - mean[u]) * rs[u] * gm[i] + bt[i];
            if (pe) {
                float pv[8];
                if (pe_lds) {
                    Vec8<float>::load(sh_pe[row >= pe_next ? 1 : 0] + (j * LPR + t) * 8, pv);   // (no 64-bit division per row)
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) pv[i] = pg[u][j][i];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] += pv[i];
            }
            Vec8<T>::store(y + row * C + (j * LPR + t) * 8, o);
        }
    }
}

template <typename T, int LPR, int U>
static void launch_ln_g(const void* x, void* y, const float* gamma, const float* beta, const float* pe, int64_t M, float eps,
                        int pe_inner, int pe_frames, hipStream_t st, const void* addend = nullptr, void* sum_out = nullptr) {
    const int64_t rpw = (64 / LPR) * U, waves = (M + rpw - 1) / rpw;
    hipLaunchKernelGGL((layernorm_g_kernel<T, LPR, U>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, (const T*)x,
                       (T*)y, gamma, beta, pe, M, eps, pe_inner, pe_frames, (const T*)addend, (T*)sum_out);
}

template <typename T>
static bool try_launch_ln16(const void* x, void* y, const float* gamma, const float* beta, const float* pe, int64_t M, int C,
                            float eps, int pe_inner, int pe_frames, hipStream_t st, const void* addend = nullptr, void* sum_out = nullptr) {
    static const int force_u = getenv("FMC_LN_U") ? atoi(getenv("FMC_LN_U")) : 0;      // A/B switch: 1 / 2 rows per lane group whatever M
    // (one row per lane group everywhere: two measured 7-35 % slower at every FMC shape, also at M = 81920 -- tools/scratch/r04/probe_ln.py)
    const bool many = force_u == 2;
    switch (C) {
        case 320: many ? launch_ln_g<T, 8, 2>(x, y, gamma, beta, pe, M, eps, pe_inner, pe_frames, st, addend, sum_out)
                       : launch_ln_g<T, 8, 1>(x, y, gamma, beta, pe, M, eps, pe_inner, pe_frames, st, addend, sum_out); return true;
        case 640: many ? launch_ln_g<T, 16, 2>(x, y, gamma, beta, pe, M, eps, pe_inner, pe_frames, st, addend, sum_out)
                       : launch_ln_g<T, 16, 1>(x, y, gamma, beta, pe, M, eps, pe_inner, pe_frames, st, addend, sum_out); return true;
        case 1280: force_u == 2 ? launch_ln_g<T, 32, 2>(x, y, gamma, beta, pe, M, eps, pe_inner, pe_frames, st, addend, sum_out)
                                : launch_ln_g<T, 32, 1>(x, y, gamma, beta, pe, M, eps, pe_inner, pe_frames, st, addend, sum_out); return true;
        default: return false;
    }
}

// --------------------------------------------------------------------------------------------
// GEGLU: y = a * gelu_erf(g)
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ void geglu_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t M, int Cff) {
    const int cpr = Cff / 8;
    const int64_t total = M * cpr;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t m = idx / cpr;
        int c = (int)(idx - m * cpr) * 8;
        float a[8], g[8];
        Vec8<T>::load(x + m * 2 * Cff + c, a);
        Vec8<T>::load(x + m * 2 * Cff + Cff + c, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = a[i] * (0.5f * g[i] * (1.f + erff(g[i] * 0.70710678118654752f)));
        Vec8<T>::store(y + m * Cff + c, a);
    }
}

// --------------------------------------------------------------------------------------------
// LayerNorm backward: dx = rstd * (dyh - mean(dyh) - xh * mean(dyh * xh)), dyh = dy * gamma, per row; mean / rstd are
// recomputed from x (nothing saved by the forward).  Optional dgamma / dbeta: per-wave register partials over its
// rows, then one fp32 atomicAdd per channel per wave (buffers zeroed by the caller).
// --------------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
                                     T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                     int64_t M, int C, float eps, const T* __restrict__ addend = nullptr) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int nchunks = C / 8;
    const float inv_c = 1.f / (float)C;
    float gm[NCH][8], ag[NCH][8], ab[NCH][8];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + 64 * k;
#pragma unroll
        for (int i = 0; i < 8; ++i) { gm[k][i] = ch < nchunks ? gamma[ch * 8 + i] : 0.f; ag[k][i] = ab[k][i] = 0.f; }
    }
    for (int64_t row = (int64_t)blockIdx.x * wpb + wave; row < M; row += (int64_t)gridDim.x * wpb) {
        float v[NCH][8], d[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + 64 * k;
            if (ch < nchunks) {
                Vec8<T>::load(x + row * C + ch * 8, v[k]);
                Vec8<T>::load(dy + row * C + ch * 8, d[k]);
#pragma unroll
                for (int i = 0; i < 8; ++i) s += v[k][i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[k][i] = d[k][i] = 0.f;
            }
        }
        const float mean = wave_sum(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (lane + 64 * k < nchunks) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float t = v[k][i] - mean; q += t * t; }
            }
        const float rstd = rsqrtf(wave_sum(q) * inv_c + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (lane + 64 * k < nchunks) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xh = (v[k][i] - mean) * rstd;
                    const float dyh = d[k][i] * gm[k][i];
                    s1 += dyh;
                    s2 += dyh * xh;
                    ag[k][i] += d[k][i] * xh;
                    ab[k][i] += d[k][i];
                    v[k][i] = xh;
                    d[k][i] = dyh;
                }
            }
        const float m1 = wave_sum(s1) * inv_c, m2 = wave_sum(s2) * inv_c;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + 64 * k;
            if (ch < nchunks) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = rstd * (d[k][i] - m1 - v[k][i] * m2);
                if (addend) {                            // + the gradient along the skip connection around the normalised branch
                    float a8[8];
                    Vec8<T>::load(addend + row * C + ch * 8, a8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] += a8[i];
                }
                Vec8<T>::store(dx + row * C + ch * 8, o);
            }
        }
    }
    if (dgamma) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + 64 * k;
            if (ch < nchunks) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    atomicAdd(dgamma + ch * 8 + i, ag[k][i]);
                    atomicAdd(dbeta + ch * 8 + i, ab[k][i]);
                }
            }
        }
    }
}

// GEGLU backward: y = a * gelu(g):  da = dy * gelu(g),  dg = dy * a * (Phi(g) + g * phi(g))
template <typename T>
__global__ void geglu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx, int64_t M, int Cff) {
    const int cpr = Cff / 8;
    const int64_t total = M * cpr;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = idx / cpr;
        const int c = (int)(idx - m * cpr) * 8;
        float a[8], g[8], d[8], da[8], dg[8];
        Vec8<T>::load(x + m * 2 * Cff + c, a);
        Vec8<T>::load(x + m * 2 * Cff + Cff + c, g);
        Vec8<T>::load(dy + m * Cff + c, d);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float cdf = 0.5f * (1.f + erff(g[i] * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * g[i] * g[i]);
            da[i] = d[i] * g[i] * cdf;
            dg[i] = d[i] * a[i] * (cdf + g[i] * pdf);
        }
        Vec8<T>::store(dx + m * 2 * Cff + c, da);
        Vec8<T>::store(dx + m * 2 * Cff + Cff + c, dg);
    }
}

int gn_check(const void* x, const void* y, int N, int HW, int C, int G, int dtype) {
    if (!x || !y) FMC_FAIL(FMC_E_NULL, "groupnorm: NULL tensor");
    if (dtype != FMC_BF16 && dtype != FMC_F32) FMC_FAIL(FMC_E_DTYPE, "groupnorm: dtype %d", dtype);
    if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % 8 || C % G || G > GN_MAX_G || C > 8 * 512)
        FMC_FAIL(FMC_E_SHAPE, "groupnorm: need C%%8==0, C%%G==0, G<=%d, C<=4096 (N=%d HW=%d C=%d G=%d)", GN_MAX_G, N,
                 HW, C, G);
    if (!fmc_aligned16(x) || !fmc_aligned16(y)) FMC_FAIL(FMC_E_ALIGN, "groupnorm: tensors must be 16-byte aligned");
    return 0;
}

}  // namespace

extern "C" int64_t fmc_groupnorm_workspace_bytes(int N, int C, int G) {
    (void)C;
    return (int64_t)N * GN_MAX_SPLIT * G * 2 * (int64_t)sizeof(float);
}

extern "C" int fmc_groupnorm_silu_fwd(const void* x, void* y, const float* gamma, const float* beta, float* stats,
                                      void* workspace, int N, int HW, int C, int G, float eps, int act, int dtype,
                                      const void* x2, int C1, void* stream) {
    if (int rc = gn_check(x, y, N, HW, C, G, dtype)) return rc;
    if (x2 && (C1 <= 0 || C1 >= C || C1 % 8 || !fmc_aligned16(x2)))
        FMC_FAIL(FMC_E_SHAPE, "groupnorm_fwd: two-source input needs 0 < C1 < C, C1 %% 8 == 0 (C1=%d C=%d)", C1, C);
    if (!x2) C1 = 0;
    if (!gamma || !beta || !stats || !workspace) FMC_FAIL(FMC_E_NULL, "groupnorm_fwd: NULL gamma/beta/stats/workspace");
    hipStream_t st = (hipStream_t)stream;
    Gn1Geom g1;
    if (gn1_geom(HW, C, G, g1)) {                        // small levels: one launch, x read once
        if (dtype == FMC_BF16) launch_gn1<bf16_t>(g1, x, y, gamma, beta, stats, N, HW, C, G, eps, act, st, x2, C1);
        else launch_gn1<float>(g1, x, y, gamma, beta, stats, N, HW, C, G, eps, act, st, x2, C1);
        FMC_CHECK_LAUNCH("fmc_groupnorm_silu_fwd");
        return 0;
    }
    GnGeom g = gn_geom(HW, C);
    dim3 grid(g.split, N), block(g.block);
    size_t lds = (size_t)2 * g.rpi * C * sizeof(float);
    float* part = (float*)workspace;
    if (dtype == FMC_BF16) {
        hipLaunchKernelGGL((gn_partial_kernel<bf16_t, 0>), grid, block, lds, st, (const bf16_t*)x, nullptr, gamma, beta,
                           nullptr, part, HW, C, G, g.tpr, g.rpi, g.rows_per_split, act, (const bf16_t*)x2, C1);
        hipLaunchKernelGGL((gn_apply_fwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, gamma, beta,
                           part, stats, HW, C, G, g.tpr, g.rpi, g.rows_per_split, eps, act, (const bf16_t*)x2, C1);
    } else {
        hipLaunchKernelGGL((gn_partial_kernel<float, 0>), grid, block, lds, st, (const float*)x, nullptr, gamma, beta,
                           nullptr, part, HW, C, G, g.tpr, g.rpi, g.rows_per_split, act, (const float*)x2, C1);
        hipLaunchKernelGGL((gn_apply_fwd_kernel<float>), grid, block, 0, st, (const float*)x, (float*)y, gamma, beta,
                           part, stats, HW, C, G, g.tpr, g.rpi, g.rows_per_split, eps, act, (const float*)x2, C1);
    }
    FMC_CHECK_LAUNCH("fmc_groupnorm_silu_fwd");
    return 0;
}

// the statistics pass of fmc_groupnorm_silu_fwd alone: per-(sample, split, group) (sum, sum of squares) -> partials [N][splits][G][2], splits =
// fmc_groupnorm_partial_splits(HW, C); consumed by fmc_groupnorm_coef (conv_halo.hip) or fmc_groupnorm_apply_fwd
extern "C" int fmc_groupnorm_partial_splits(int HW, int C) { return (HW > 0 && C >= 8 && C <= 8 * 512 && C % 8 == 0) ? gn_geom(HW, C).split : 0; }

extern "C" int fmc_groupnorm_partials(const void* x, const void* x2, int C1, float* partials, int N, int HW, int C, int G, int dtype, void* stream) {
    if (int rc = gn_check(x, x, N, HW, C, G, dtype)) return rc;
    if (x2 && (C1 <= 0 || C1 >= C || C1 % 8 || !fmc_aligned16(x2)))
        FMC_FAIL(FMC_E_SHAPE, "groupnorm_partials: two-source input needs 0 < C1 < C, C1 %% 8 == 0 (C1=%d C=%d)", C1, C);
    if (!x2) C1 = 0;
    if (!partials) FMC_FAIL(FMC_E_NULL, "groupnorm_partials: NULL partials");
    GnGeom g = gn_geom(HW, C);
    dim3 grid(g.split, N), block(g.block);
    const size_t lds = (size_t)2 * g.rpi * C * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FMC_BF16)
        hipLaunchKernelGGL((gn_partial_kernel<bf16_t, 0>), grid, block, lds, st, (const bf16_t*)x, nullptr, nullptr, nullptr, nullptr, partials, HW, C, G,
                           g.tpr, g.rpi, g.rows_per_split, 0, (const bf16_t*)x2, C1);
    else
        hipLaunchKernelGGL((gn_partial_kernel<float, 0>), grid, block, lds, st, (const float*)x, nullptr, nullptr, nullptr, nullptr, partials, HW, C, G,
                           g.tpr, g.rpi, g.rows_per_split, 0, (const float*)x2, C1);
    FMC_CHECK_LAUNCH("fmc_groupnorm_partials");
    return 0;
}

extern "C" int fmc_groupnorm_apply_fwd(const void* x, void* y, const float* gamma, const float* beta, float* stats,
                                       const float* partials, int part_splits, int N, int HW, int C, int G, float eps, int act,
                                       int dtype, void* stream) {
    if (int rc = gn_check(x, y, N, HW, C, G, dtype)) return rc;
    if (!gamma || !beta || !stats || !partials) FMC_FAIL(FMC_E_NULL, "groupnorm_apply_fwd: NULL gamma/beta/stats/partials");
    if (part_splits < 1 || part_splits > GN_MAX_SPLIT) FMC_FAIL(FMC_E_SHAPE, "groupnorm_apply_fwd: part_splits %d (1..%d)", part_splits, GN_MAX_SPLIT);
    hipStream_t st = (hipStream_t)stream;
    GnGeom g = gn_geom(HW, C);
    dim3 grid(g.split, N), block(g.block);
    if (dtype == FMC_BF16)
        hipLaunchKernelGGL((gn_apply_fwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, gamma, beta, partials, stats, HW,
                           C, G, g.tpr, g.rpi, g.rows_per_split, eps, act, (const bf16_t*)nullptr, 0, part_splits);
    else
        hipLaunchKernelGGL((gn_apply_fwd_kernel<float>), grid, block, 0, st, (const float*)x, (float*)y, gamma, beta, partials, stats, HW, C,
                           G, g.tpr, g.rpi, g.rows_per_split, eps, act, (const float*)nullptr, 0, part_splits);
    FMC_CHECK_LAUNCH("fmc_groupnorm_apply_fwd");
    return 0;
}

// --------------------------------------------------------------------------------------------
// GroupNorm folded into the linear layer behind it (no activation in between: the norm in front of a transformer's proj_in).
//   proj(GN(x))[m, n] = sum_c W[n, c] (x[m, c] a_c + b_c) + bias[n],   a_c = rstd[img, g(c)] gamma[c],  b_c = beta[c] - mean[img, g(c)] a_c
//                     = sum_c W'_img[n, c] x[m, c] + bias'_img[n]
// One workgroup = 8 weight rows of one image x 8-channel chunks: W'_img = bf16(W a) (row-major, or tile-major [N / 320][C / 32][320][32] as tile 18 reads
// it) and, in fp32 and FROM THE ROUNDED W', bias'_img[n] = bias[n] + sum_c (W[n, c] beta[c] - W'_img[n, c] mean[img, g(c)]): the mean term then cancels
// against sum_c W' x exactly as (x - mean) would, whatever |mean| / std is -- the only new rounding is W a to bf16, in place of GN(x) to bf16.
// --------------------------------------------------------------------------------------------
constexpr int GNF_ROWS = 8, GNF_PASSES = 5;                  // a workgroup = 40 weight rows of one image: 8 rows x 64 chunk lanes per pass
__global__ __launch_bounds__(512) void gn_fold_linear_kernel(const float* __restrict__ part, int nsplit, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias,
                                                              bf16_t* __restrict__ w_out, float* __restrict__ b_out, int HW, int C, int G, int N, float eps,
                                                              int tilemajor) {
    __shared__ float sh_mean[GN_MAX_G], sh_rstd[GN_MAX_G];
    __shared__ float sh_part[GN_MAX_SPLIT * GN_MAX_G * 2];
    const int img = blockIdx.y, n0 = blockIdx.x * (GNF_ROWS * GNF_PASSES), tid = threadIdx.x;
    const int cpg = C / G, nch = C / 8;
    const int r = tid >> 6, cc = tid & 63;                     // row n0 + 8 pass + r, chunks cc, cc + 64, ...
    // everything that does not depend on the statistics is requested first: my weight chunks of the five passes, gamma / beta of my first chunk column
    Vec8<bf16_t>::raw_t wraw[GNF_PASSES];
    float gm[8], bt[8];
    const bool has = cc < nch;
#pragma unroll
    for (int p = 0; p < GNF_PASSES; ++p) {
        const int n = n0 + p * GNF_ROWS + r;
        wraw[p] = Vec8<bf16_t>::load_raw(w + (size_t)(has && n < N ? n : 0) * C + (has ? cc * 8 : 0));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { gm[i] = 0.f; bt[i] = 0.f; }
    if (has) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + cc * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + cc * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + cc * 8), b1 = *reinterpret_cast<const f32x4*>(beta + cc * 8 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { gm[i] = g0[i]; gm[4 + i] = g1[i]; bt[i] = b0[i]; bt[4 + i] = b1[i]; }
    }
    for (int i = tid; i < nsplit * G * 2; i += blockDim.x) sh_part[i] = part[(size_t)img * nsplit * G * 2 + i];
    __syncthreads();
    for (int g = tid; g < G; g += blockDim.x) {               // (as gn_apply_fwd_kernel: fixed order, double)
        double a = 0.0, b = 0.0;
        for (int k = 0; k < nsplit; ++k) {
            const float* p = sh_part + ((size_t)k * G + g) * 2;
            a += (double)p[0];
            b += (double)p[1];
        }
        const double cnt = (double)HW * cpg, mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        sh_mean[g] = (float)mean;
        sh_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    // per-lane constants of my first chunk column (the only one at C <= 512): a_c = rstd gamma and the group mean -- one integer division per channel, not per row
    float ac[8], mu[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int g = has ? (cc * 8 + i) / cpg : 0;
        ac[i] = sh_rstd[g] * gm[i];
        mu[i] = sh_mean[g];
    }
#pragma unroll
    for (int p = 0; p < GNF_PASSES; ++p) {
        const int n = n0 + p * GNF_ROWS + r;                       // (wave-uniform: a wave = one weight row)
        float acc = 0.f;
        if (n < N) {
            for (int ch = cc; ch < nch; ch += 64) {
                const int c0 = ch * 8;
                float wv[8], o[8];
                if (ch == cc) Vec8<bf16_t>::unpack(wraw[p], wv);
                else Vec8<bf16_t>::load(w + (size_t)n * C + c0, wv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float a_c = ac[i], m_c = mu[i], be = bt[i];
                    if (ch != cc) {
                        const int c = c0 + i, g = c / cpg;
                        a_c = sh_rstd[g] * gamma[c]; m_c = sh_mean[g]; be = beta[c];
                    }
                    const float wr = bf2f(f2bf(wv[i] * a_c));
                    o[i] = wr;
                    acc += wv[i] * be - wr * m_c;
                }
                const size_t dst = tilemajor ? ((size_t)(n / 320) * (C / 32) + c0 / 32) * (320 * 32) + (size_t)(n % 320) * 32 + (c0 % 32)
                                             : (size_t)n * C + c0;
                Vec8<bf16_t>::store(w_out + (size_t)img * N * C + dst, o);
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);          // fixed tree: deterministic
        if (cc == 0 && n < N) b_out[(size_t)img * N + n] = acc + (bias ? bf2f(bias[n]) : 0.f);
    }
}

extern "C" int fmc_groupnorm_fold_linear(const float* partials, int part_splits, const float* gamma, const float* beta, const void* w, const void* bias,
                                         void* w_out, float* bias_out, int n_img, int HW, int C, int G, int N, float eps, int w_tilemajor, void* stream) {
    if (!partials || !gamma || !beta || !w || !w_out || !bias_out) FMC_FAIL(FMC_E_NULL, "groupnorm_fold_linear: NULL argument");
    if (n_img < 1 || HW < 1 || C < 8 || C % 8 || G < 1 || G > GN_MAX_G || C % G || N < 1 || (w_tilemajor && (N % 320 || C % 32)))
        FMC_FAIL(FMC_E_SHAPE, "groupnorm_fold_linear: C %% 8, C %% G, G <= %d (tile-major: N %% 320, C %% 32) (C=%d G=%d N=%d)", GN_MAX_G, C, G, N);
    if (part_splits < 1 || part_splits > GN_MAX_SPLIT) FMC_FAIL(FMC_E_SHAPE, "groupnorm_fold_linear: part_splits %d (1..%d)", part_splits, GN_MAX_SPLIT);
    if (!fmc_aligned16(w) || !fmc_aligned16(w_out) || !fmc_aligned16(gamma) || !fmc_aligned16(beta))
        FMC_FAIL(FMC_E_ALIGN, "groupnorm_fold_linear: w / w_out / gamma / beta must be 16-byte aligned");
    dim3 grid((unsigned)((N + GNF_ROWS * GNF_PASSES - 1) / (GNF_ROWS * GNF_PASSES)), (unsigned)n_img);
    hipLaunchKernelGGL(gn_fold_linear_kernel, grid, dim3(512), 0, (hipStream_t)stream, partials, part_splits, gamma, beta, (const bf16_t*)w,
                       (const bf16_t*)bias, (bf16_t*)w_out, bias_out, HW, C, G, N, eps, w_tilemajor);
    FMC_CHECK_LAUNCH("fmc_groupnorm_fold_linear");
    return 0;
}

extern "C" int fmc_groupnorm_silu_bwd_add(const void* dy, const void* x, void* dx, const float* gamma, const float* beta,
                                          const float* stats, void* workspace, int N, int HW, int C, int G, int act,
                                          const void* addend, int dtype, void* stream) {
    if (int rc = gn_check(x, dx, N, HW, C, G, dtype)) return rc;
    if (!dy || !gamma || !beta || !stats || !workspace) FMC_FAIL(FMC_E_NULL, "groupnorm_bwd: NULL argument");
    if (!fmc_aligned16(dy) || (addend && !fmc_aligned16(addend))) FMC_FAIL(FMC_E_ALIGN, "groupnorm_bwd: dy / addend must be 16-byte aligned");
    GnGeom g = gn_geom(HW, C);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(g.split, N), block(g.block);
    size_t lds = (size_t)2 * g.rpi * C * sizeof(float);
    float* part = (float*)workspace;
    if (dtype == FMC_BF16) {
        hipLaunchKernelGGL((gn_partial_kernel<bf16_t, 1>), grid, block, lds, st, (const bf16_t*)x, (const bf16_t*)dy,
                           gamma, beta, stats, part, HW, C, G, g.tpr, g.rpi, g.rows_per_split, act);
        hipLaunchKernelGGL((gn_apply_bwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)x,
                           (bf16_t*)dx, gamma, beta, stats, part, HW, C, G, g.tpr, g.rpi, g.rows_per_split, act, (const bf16_t*)addend);
    } else {
        hipLaunchKernelGGL((gn_partial_kernel<float, 1>), grid, block, lds, st, (const float*)x, (const float*)dy, gamma,
                           beta, stats, part, HW, C, G, g.tpr, g.rpi, g.rows_per_split, act);
        hipLaunchKernelGGL((gn_apply_bwd_kernel<float>), grid, block, 0, st, (const float*)dy, (const float*)x,
                           (float*)dx, gamma, beta, stats, part, HW, C, G, g.tpr, g.rpi, g.rows_per_split, act, (const float*)addend);
    }
    FMC_CHECK_LAUNCH("fmc_groupnorm_silu_bwd");
    return 0;
}

extern "C" int fmc_groupnorm_silu_bwd(const void* dy, const void* x, void* dx, const float* gamma, const float* beta,
                                      const float* stats, void* workspace, int N, int HW, int C, int G, int act,
                                      int dtype, void* stream) {
    return fmc_groupnorm_silu_bwd_add(dy, x, dx, gamma, beta, stats, workspace, N, HW, C, G, act, nullptr, dtype, stream);
}

template <typename T, int NCH, int R>
static void launch_ln_r(const void* x, void* y, const float* gamma, const float* beta, const float* pe, int64_t M, int C,
                        float eps, int pe_inner, int pe_frames, hipStream_t st) {
    const int wpb = 4;
    int64_t blocks = (M + wpb * R - 1) / (wpb * R);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL((layernorm_kernel<T, NCH, R>), dim3((unsigned)blocks), dim3(64 * wpb), 0, st, (const T*)x, (T*)y,
                       gamma, beta, pe, M, C, eps, pe_inner, pe_frames);
}

// rows per wave: 4 when there are plenty of rows (loads in flight per lane), fewer when the row count alone cannot
// fill 256 CUs x 8 waves (level 2/3 token matrices have only 1280-5120 rows)
template <typename T, int NCH>
static void launch_ln_n(const void* x, void* y, const float* gamma, const float* beta, const float* pe, int64_t M, int C,
                        float eps, int pe_inner, int pe_frames, hipStream_t st) {
    if (NCH <= 3 && M >= 32768) launch_ln_r<T, NCH, 4>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st);
    else if (M >= 16384) launch_ln_r<T, NCH, 2>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st);
    else launch_ln_r<T, NCH, 1>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st);
}

template <typename T>
static void launch_ln(const void* x, void* y, const float* gamma, const float* beta, const float* pe, int64_t M, int C,
                      float eps, int pe_inner, int pe_frames, hipStream_t st) {
    if (try_launch_ln16<T>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st)) return;
    switch ((C / 8 + 63) / 64) {
        case 1: launch_ln_n<T, 1>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st); break;
        case 2: launch_ln_n<T, 2>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st); break;
        case 3: launch_ln_n<T, 3>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st); break;
        case 4: launch_ln_n<T, 4>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st); break;
        case 5: launch_ln_n<T, 5>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st); break;
        default: break;
    }
}

extern "C" int fmc_layernorm_add_fwd(const void* x, const void* addend, void* sum_out, void* y, const float* gamma, const float* beta,
                                     const float* pe, int64_t M, int C, float eps, int pe_inner, int pe_frames, int dtype, void* stream) {
    if (!x || !addend || !sum_out || !y || !gamma || !beta) FMC_FAIL(FMC_E_NULL, "layernorm_add: NULL argument");
    if (M <= 0 || (C != 320 && C != 640 && C != 1280)) FMC_FAIL(FMC_E_SHAPE, "layernorm_add: C must be 320, 640 or 1280 (C=%d)", C);
    if (pe && (pe_inner <= 0 || pe_frames <= 0)) FMC_FAIL(FMC_E_SHAPE, "layernorm_add: pe_inner/pe_frames must be > 0");
    if (!fmc_aligned16(x) || !fmc_aligned16(y) || !fmc_aligned16(addend) || !fmc_aligned16(sum_out))
        FMC_FAIL(FMC_E_ALIGN, "layernorm_add: tensors must be 16-byte aligned");
    if (!pe) { pe_inner = 1; pe_frames = 1; }
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
    if (dtype == FMC_BF16) ok = try_launch_ln16<bf16_t>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st, addend, sum_out);
    else if (dtype == FMC_F32) ok = try_launch_ln16<float>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st, addend, sum_out);
    else FMC_FAIL(FMC_E_DTYPE, "layernorm_add: dtype %d", dtype);
    if (!ok) FMC_FAIL(FMC_E_SHAPE, "layernorm_add: unsupported width %d", C);
    FMC_CHECK_LAUNCH("fmc_layernorm_add_fwd");
    return 0;
}

extern "C" int fmc_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, const float* pe,
                                 int64_t M, int C, float eps, int pe_inner, int pe_frames, int dtype, void* stream) {
    if (!x || !y || !gamma || !beta) FMC_FAIL(FMC_E_NULL, "layernorm: NULL argument");
    if (M <= 0 || C <= 0 || C % 8 || C > 8 * 64 * 5) FMC_FAIL(FMC_E_SHAPE, "layernorm: need C%%8==0 and C<=2560 (C=%d)", C);
    if (pe && (pe_inner <= 0 || pe_frames <= 0)) FMC_FAIL(FMC_E_SHAPE, "layernorm: pe_inner/pe_frames must be > 0");
    if (!fmc_aligned16(x) || !fmc_aligned16(y)) FMC_FAIL(FMC_E_ALIGN, "layernorm: tensors must be 16-byte aligned");
    if (!pe) { pe_inner = 1; pe_frames = 1; }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FMC_BF16) launch_ln<bf16_t>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st);
    else if (dtype == FMC_F32) launch_ln<float>(x, y, gamma, beta, pe, M, C, eps, pe_inner, pe_frames, st);
    else FMC_FAIL(FMC_E_DTYPE, "layernorm: dtype %d", dtype);
    FMC_CHECK_LAUNCH("fmc_layernorm_fwd");
    return 0;
}

extern "C" int fmc_geglu_fwd(const void* x, void* y, int64_t M, int Cff, int dtype, void* stream) {
    if (!x || !y) FMC_FAIL(FMC_E_NULL, "geglu: NULL argument");
    if (M <= 0 || Cff <= 0 || Cff % 8) FMC_FAIL(FMC_E_SHAPE, "geglu: need Cff%%8==0 (Cff=%d)", Cff);
    if (!fmc_aligned16(x) || !fmc_aligned16(y)) FMC_FAIL(FMC_E_ALIGN, "geglu: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int64_t total = M * (Cff / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == FMC_BF16) hipLaunchKernelGGL((geglu_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, M, Cff);
    else if (dtype == FMC_F32) hipLaunchKernelGGL((geglu_kernel<float>), grid, block, 0, st, (const float*)x, (float*)y, M, Cff);
    else FMC_FAIL(FMC_E_DTYPE, "geglu: dtype %d", dtype);
    FMC_CHECK_LAUNCH("fmc_geglu_fwd");
    return 0;
}

template <typename T>
static void launch_ln_bwd(const void* dy, const void* x, const float* gamma, void* dx, float* dgamma, float* dbeta,
                          int64_t M, int C, float eps, hipStream_t st, const void* addend = nullptr) {
    const int wpb = 4;
    int64_t blocks = (M + wpb - 1) / wpb;
    if (blocks > 2048) blocks = 2048;            // grid-stride: every wave owns many rows -> few dgamma/dbeta atomics
    dim3 grid((unsigned)blocks), block(64 * wpb);
#define LNB_CASE(K)                                                                                                   \
    case K:                                                                                                           \
        hipLaunchKernelGGL((layernorm_bwd_kernel<T, K>), grid, block, 0, st, (const T*)dy, (const T*)x, gamma, (T*)dx, \
                           dgamma, dbeta, M, C, eps, (const T*)addend);                                               \
        break;
    switch ((C / 8 + 63) / 64) {
        LNB_CASE(1) LNB_CASE(2) LNB_CASE(3) LNB_CASE(4) LNB_CASE(5)
        default: break;
    }
#undef LNB_CASE
}

extern "C" int fmc_layernorm_bwd_add(const void* dy, const void* x, const float* gamma, void* dx, float* dgamma,
                                     float* dbeta, const void* addend, int64_t M, int C, float eps, int dtype, void* stream) {
    if (!dy || !x || !gamma || !dx) FMC_FAIL(FMC_E_NULL, "layernorm_bwd: NULL argument");
    if (addend && !fmc_aligned16(addend)) FMC_FAIL(FMC_E_ALIGN, "layernorm_bwd: addend must be 16-byte aligned");
    if ((dgamma == nullptr) != (dbeta == nullptr)) FMC_FAIL(FMC_E_NULL, "layernorm_bwd: pass both dgamma and dbeta or neither");
    if (M <= 0 || C <= 0 || C % 8 || C > 8 * 64 * 5) FMC_FAIL(FMC_E_SHAPE, "layernorm_bwd: need C%%8==0 and C<=2560 (C=%d)", C);
    if (!fmc_aligned16(dy) || !fmc_aligned16(x) || !fmc_aligned16(dx)) FMC_FAIL(FMC_E_ALIGN, "layernorm_bwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FMC_BF16) launch_ln_bwd<bf16_t>(dy, x, gamma, dx, dgamma, dbeta, M, C, eps, st, addend);
    else if (dtype == FMC_F32) launch_ln_bwd<float>(dy, x, gamma, dx, dgamma, dbeta, M, C, eps, st, addend);
    else FMC_FAIL(FMC_E_DTYPE, "layernorm_bwd: dtype %d", dtype);
    FMC_CHECK_LAUNCH("fmc_layernorm_bwd");
    return 0;
}

extern "C" int fmc_layernorm_bwd(const void* dy, const void* x, const float* gamma, void* dx, float* dgamma,
                                 float* dbeta, int64_t M, int C, float eps, int dtype, void* stream) {
    return fmc_layernorm_bwd_add(dy, x, gamma, dx, dgamma, dbeta, nullptr, M, C, eps, dtype, stream);
}

extern "C" int fmc_geglu_bwd(const void* dy, const void* x, void* dx, int64_t M, int Cff, int dtype, void* stream) {
    if (!dy || !x || !dx) FMC_FAIL(FMC_E_NULL, "geglu_bwd: NULL argument");
    if (M <= 0 || Cff <= 0 || Cff % 8) FMC_FAIL(FMC_E_SHAPE, "geglu_bwd: need Cff%%8==0 (Cff=%d)", Cff);
    if (!fmc_aligned16(dy) || !fmc_aligned16(x) || !fmc_aligned16(dx)) FMC_FAIL(FMC_E_ALIGN, "geglu_bwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int64_t total = M * (Cff / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == FMC_BF16) hipLaunchKernelGGL((geglu_bwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, M, Cff);
    else if (dtype == FMC_F32) hipLaunchKernelGGL((geglu_bwd_kernel<float>), grid, block, 0, st, (const float*)dy, (const float*)x, (float*)dx, M, Cff);
    else FMC_FAIL(FMC_E_DTYPE, "geglu_bwd: dtype %d", dtype);
    FMC_CHECK_LAUNCH("fmc_geglu_bwd");
    return 0;
}
