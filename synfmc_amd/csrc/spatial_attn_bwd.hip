// Backward of the spatial attention (gfx950).  Needed by the FMC training stages: the U-Net is frozen but the
// activation gradient must flow through every attention layer back to the OMC / CMC injection points
// (reference backward = autograd through attention_processor.py:61-67).
//
// With S = scale * Q K^T, P = softmax(S) = exp(S - LSE), O = P V:
//   D_q = rowsum(dO .* O);  dV = P^T dO;  dP = dO V^T;  dS = P .* (dP - D);  dQ = scale * dS K;  dK = scale * dS^T Q.
// Three kernels, all on v_mfma_f32_32x32x16_bf16 with the forward's "swapped" layouts (no S x S tensor, no atomics):
//   1. rowdot:  D[b,h,q]                                                    (HBM bound, 2 reads)
//   2. dq:      forward-shaped (a workgroup owns 128 queries, streams key tiles): S^T = K Q^T, dP^T = V dO^T,
//               dS^T from registers is the B operand of dQ^T += K^T dS^T   (K staged row-major AND transposed)
//   3. dkdv:    key-shaped (a wave owns 32 keys = its lanes, streams query tiles): S = Q K^T, dP = dO V^T,
//               P / dS from registers are the B operands of dV^T += dO^T P and dK^T += Q^T dS
//               (Q, dO staged row-major AND transposed); frames sharing one text K/V (kv_batch_div) are looped inside
//               the workgroup, so dK / dV need no cross-workgroup reduction.
// S is recomputed in both 2 and 3 (7 matrix products instead of 5) in exchange for atomic-free, deterministic grads.
// Algorithmic flops per backward = 14 * B*H*Sq*Skv*D (2.5x + 1 recompute of the forward's 4).
#include "attn_common.h"
#include <cstdlib>

namespace {

constexpr float LOG2E_B = 1.4426950408889634f;
constexpr int BK2 = 64;            // keys per LDS tile (dq kernel) / queries per LDS tile (dkdv kernel)

struct SABwdParams {
    const void* q; const void* k; const void* v; const void* o; const void* d_o; const float* lse; float* dvec;
    void* dq; void* dk; void* dv;
    int B, H, Sq, Skv, D;
    int64_t qbs, qrs, kbs, krs, obs, ors;            // q / (k,v) / (o, dO) strides
    int64_t dqbs, dqrs, dkbs, dkrs;                  // dq / (dk,dv) strides
    int kv_batch_div;
    float scale, scale_log2;
    int nblk;
    int xcd;                                         // block -> (batch*head, block) map: all heads and blocks of a batch entry on one XCD
};

// ---- 1. D[b,h,q] = sum_d dO * O ------------------------------------------------------------------------------
template <typename T>
__global__ void rowdot_kernel(const SABwdParams P) {
    const int64_t total = (int64_t)P.B * P.H * P.Sq;
    const int CH = P.D / 8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % P.Sq);
        const int64_t bh = idx / P.Sq;
        const int h = (int)(bh % P.H), b = (int)(bh / P.H);
        const T* op = (const T*)P.o + (int64_t)b * P.obs + (int64_t)q * P.ors + h * P.D;
        const T* gp = (const T*)P.d_o + (int64_t)b * P.obs + (int64_t)q * P.ors + h * P.D;
        float acc = 0.f;
        for (int c = 0; c < CH; ++c) {
            float a[8], g[8];
            Vec8<T>::load(op + c * 8, a);
            Vec8<T>::load(gp + c * 8, g);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += a[i] * g[i];
        }
        P.dvec[((int64_t)b * P.H + h) * P.Sq + q] = acc;
    }
}

// stage `rows` rows of a [*, D] operand: row-major into Rs[rows][KP] (pad columns zeroed by the caller) and, if Tt,
// transposed into Tt[D][VP]; rows >= limit are zero
template <typename T>
__device__ __forceinline__ void stage_tile(const T* g, int64_t row_stride, int row0, int limit, int rows, int CH, T* Rs,
                                           int KP, T* Tt, int VP, int tid, int nthreads) {
    for (int c = tid; c < rows * CH; c += nthreads) {
        const int row = c / CH, ch = c - row * CH;
        float v[8];
        if (row0 + row < limit) Vec8<T>::load(g + (int64_t)(row0 + row) * row_stride + ch * 8, v);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 0.f;
        }
        if (Rs) Vec8<T>::store(Rs + row * KP + ch * 8, v);
        if (Tt) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) Tt[(ch * 8 + i) * VP + row] = f2bf(v[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) Tt[(ch * 8 + i) * VP + row] = v[i];
            }
        }
    }
}

// blockIdx -> (batch * H + head, block).  Workgroup ids go round-robin over the 8 XCDs; with the plain order the blocks of one (batch,
// head) -- which all sweep the same K / V (or Q / dO) rows -- land on 8 different L2s, and a head's 80-byte row slice shares its
// 128-byte lines with the neighbouring heads.  As in the forward kernel: everything of a batch entry on one XCD when the batch count allows.
__device__ __forceinline__ void sab_decode(const SABwdParams& P, int id, int& bh, int& blk) {
    if (P.xcd) {
        const int xcd = id & 7, within = id >> 3, per_b = P.H * P.nblk;
        const int rem = within % per_b;
        bh = ((within / per_b) * 8 + xcd) * P.H + rem / P.nblk;
        blk = rem % P.nblk;
    } else {
        bh = id / P.nblk;
        blk = id % P.nblk;
    }
}

// ---- 2. dQ -----------------------------------------------------------------------------------------------------
template <typename T, int NKS>
__global__ __launch_bounds__(256) void attn_dq_kernel(const SABwdParams P) {
    constexpr int NDT = (NKS + 1) / 2, DP16 = NKS * 16, KP = DP16 + 8, VP = BK2 + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Ks = reinterpret_cast<T*>(smem_raw);          // [BK2][KP]
    T* Vs = Ks + BK2 * KP;                           // [BK2][KP]
    T* Kt = Vs + BK2 * KP;                           // [NDT*32][VP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int D = P.D, CH = D / 8;
    int bh, qblk;
    sab_decode(P, blockIdx.x, bh, qblk);
    const int b = bh / P.H, h = bh - b * P.H;
    const T* qg = (const T*)P.q + (int64_t)b * P.qbs + (int64_t)h * D;
    const T* gg = (const T*)P.d_o + (int64_t)b * P.obs + (int64_t)h * D;
    const T* kg = (const T*)P.k + (int64_t)(b / P.kv_batch_div) * P.kbs + (int64_t)h * D;
    const T* vg = (const T*)P.v + (int64_t)(b / P.kv_batch_div) * P.kbs + (int64_t)h * D;

    const int qrow = qblk * 128 + wave * 32 + l31;
    const bool qok = qrow < P.Sq;
    Frag<T> qf[NKS], gf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int d0 = ks * 16 + half * 8;
        if (qok && d0 < D) {
            make_frag<T>(qg + (int64_t)qrow * P.qrs + d0, qf[ks]);
            make_frag<T>(gg + (int64_t)qrow * P.ors + d0, gf[ks]);
        } else {
            zero(qf[ks]);
            zero(gf[ks]);
        }
    }
    const float lse2 = qok ? P.lse[((int64_t)b * P.H + h) * P.Sq + qrow] * LOG2E_B : INFINITY;
    const float dq_ = qok ? P.dvec[((int64_t)b * P.H + h) * P.Sq + qrow] : 0.f;
    if (DP16 > D)
        for (int r = tid; r < BK2; r += blockDim.x)
#pragma unroll
            for (int i = 0; i < 8; ++i) { Ks[r * KP + D + i] = T(0); Vs[r * KP + D + i] = T(0); }

    f32x16 acc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;

    const int ntiles = (P.Skv + BK2 - 1) / BK2;
    for (int tile = 0; tile < ntiles; ++tile) {
        const int kv0 = tile * BK2;
        __syncthreads();
        stage_tile<T>(kg, P.krs, kv0, P.Skv, BK2, CH, Ks, KP, Kt, VP, tid, blockDim.x);
        stage_tile<T>(vg, P.krs, kv0, P.Skv, BK2, CH, Vs, KP, (T*)nullptr, VP, tid, blockDim.x);
        __syncthreads();
#pragma unroll
        for (int sb = 0; sb < BK2 / 32; ++sb) {
            const int kvb = kv0 + sb * 32;
            if (kvb >= P.Skv) break;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                Frag<T> kf, vf;
                make_frag<T>(Ks + (sb * 32 + l31) * KP + ks * 16 + half * 8, kf);
                make_frag<T>(Vs + (sb * 32 + l31) * KP + ks * 16 + half * 8, vf);
                mma32(kf, qf[ks], s);                 // S^T : rows keys, col query (lane)
                mma32(vf, gf[ks], dp);                // dP^T
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kvb + (r & 3) + 8 * (r >> 2) + 4 * half;
                float p = __builtin_amdgcn_exp2f(fmaf(s[r], P.scale_log2, -lse2));
                if (kv >= P.Skv) p = 0.f;
                ds[r] = p * (dp[r] - dq_) * P.scale;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float d8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) d8[i] = ds[s2 * 8 + i];
                Frag<T> df;
                p_frag(d8, df);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const T* krow = Kt + (dt * 32 + l31) * VP + sb * 32 + s2 * 16 + half * 4;
                    Frag<T> kt;
                    make_frag_2x4<T>(krow, krow + 8, kt);
                    mma32(kt, df, acc[dt]);           // dQ^T[d, q] += K^T[d, kv] dS^T[kv, q]
                }
            }
        }
    }
    if (qok) {
        T* orow = (T*)P.dq + (int64_t)b * P.dqbs + (int64_t)qrow * P.dqrs + (int64_t)h * D;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * half;
                if (d < D) store4<T>(orow + d, acc[dt][4 * g], acc[dt][4 * g + 1], acc[dt][4 * g + 2], acc[dt][4 * g + 3]);
            }
    }
}

// ---- 3. dK, dV ---------------------------------------------------------------------------------------------------
template <typename T, int NKS, int WAVES, int BQ>
__global__ __launch_bounds__(64 * WAVES) void attn_dkdv_kernel(const SABwdParams P) {
    constexpr int NDT = (NKS + 1) / 2, DP16 = NKS * 16, KP = DP16 + 8, VP = BQ + 4, BKV = 32 * WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Kb = reinterpret_cast<T*>(smem_raw);          // [BKV][KP]   this workgroup's keys
    T* Vb = Kb + BKV * KP;                           // [BKV][KP]
    T* Qs = Vb + BKV * KP;                           // [BQ][KP]   query tile, row-major
    T* Gs = Qs + BQ * KP;                           // [BQ][KP]   dO tile, row-major
    T* Qt = Gs + BQ * KP;                           // [NDT*32][VP] query tile, transposed
    T* Gt = Qt + NDT * 32 * VP;                      // [NDT*32][VP]
    float* Ls = reinterpret_cast<float*>(Gt + NDT * 32 * VP);   // [BQ] lse*log2e, [BQ] D
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int D = P.D, CH = D / 8;
    int bkvh, kblk;
    sab_decode(P, blockIdx.x, bkvh, kblk);
    const int bkv = bkvh / P.H, h = bkvh - bkv * P.H;
    const T* kg = (const T*)P.k + (int64_t)bkv * P.kbs + (int64_t)h * D;
    const T* vg = (const T*)P.v + (int64_t)bkv * P.kbs + (int64_t)h * D;
    const int kv_base = kblk * BKV;

    if (DP16 > D) {                                   // pad columns [D, DP16) of every row-major tile: zero, once
        for (int r = tid; r < BKV; r += blockDim.x)
#pragma unroll
            for (int i = 0; i < 8; ++i) { Kb[r * KP + D + i] = T(0); Vb[r * KP + D + i] = T(0); }
        for (int r = tid; r < BQ; r += blockDim.x)
#pragma unroll
            for (int i = 0; i < 8; ++i) { Qs[r * KP + D + i] = T(0); Gs[r * KP + D + i] = T(0); }
    }
    stage_tile<T>(kg, P.krs, kv_base, P.Skv, BKV, CH, Kb, KP, (T*)nullptr, VP, tid, blockDim.x);
    stage_tile<T>(vg, P.krs, kv_base, P.Skv, BKV, CH, Vb, KP, (T*)nullptr, VP, tid, blockDim.x);

    f32x16 acck[NDT], accv[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acck[dt][r] = accv[dt][r] = 0.f;

    const int nqt = (P.Sq + BQ - 1) / BQ;
    for (int rep = 0; rep < P.kv_batch_div; ++rep) {
        const int b = bkv * P.kv_batch_div + rep;
        const T* qg = (const T*)P.q + (int64_t)b * P.qbs + (int64_t)h * D;
        const T* gg = (const T*)P.d_o + (int64_t)b * P.obs + (int64_t)h * D;
        const float* lse = P.lse + ((int64_t)b * P.H + h) * P.Sq;
        const float* dvec = P.dvec + ((int64_t)b * P.H + h) * P.Sq;
        for (int qt = 0; qt < nqt; ++qt) {
            const int q0 = qt * BQ;
            __syncthreads();
            stage_tile<T>(qg, P.qrs, q0, P.Sq, BQ, CH, Qs, KP, Qt, VP, tid, blockDim.x);
            stage_tile<T>(gg, P.ors, q0, P.Sq, BQ, CH, Gs, KP, Gt, VP, tid, blockDim.x);
            for (int i = tid; i < BQ; i += blockDim.x) {
                const bool ok = q0 + i < P.Sq;
                Ls[i] = ok ? lse[q0 + i] * LOG2E_B : INFINITY;      // p = 0 for padding queries
                Ls[BQ + i] = ok ? dvec[q0 + i] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int sb = 0; sb < BQ / 32; ++sb) {
                if (q0 + sb * 32 >= P.Sq) break;
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    Frag<T> qf, gf, kf, vf;
                    make_frag<T>(Qs + (sb * 32 + l31) * KP + ks * 16 + half * 8, qf);
                    make_frag<T>(Gs + (sb * 32 + l31) * KP + ks * 16 + half * 8, gf);
                    make_frag<T>(Kb + (wave * 32 + l31) * KP + ks * 16 + half * 8, kf);
                    make_frag<T>(Vb + (wave * 32 + l31) * KP + ks * 16 + half * 8, vf);
                    mma32(qf, kf, s);                 // S : rows queries, col key (lane)
                    mma32(gf, vf, dp);                // dP
                }
                float p[16], ds[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    p[r] = __builtin_amdgcn_exp2f(fmaf(s[r], P.scale_log2, -Ls[ql]));
                    ds[r] = p[r] * (dp[r] - Ls[BQ + ql]) * P.scale;
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float p8[8], d8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { p8[i] = p[s2 * 8 + i]; d8[i] = ds[s2 * 8 + i]; }
                    Frag<T> pf, df;
                    p_frag(p8, pf);
                    p_frag(d8, df);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) {
                        const int off = (dt * 32 + l31) * VP + sb * 32 + s2 * 16 + half * 4;
                        Frag<T> gt, qtf;
                        make_frag_2x4<T>(Gt + off, Gt + off + 8, gt);
                        make_frag_2x4<T>(Qt + off, Qt + off + 8, qtf);
                        mma32(gt, pf, accv[dt]);      // dV^T[d, kv] += dO^T[d, q] P[q, kv]
                        mma32(qtf, df, acck[dt]);     // dK^T[d, kv] += Q^T[d, q] dS[q, kv]
                    }
                }
            }
        }
    }
    const int kv = kv_base + wave * 32 + l31;
    if (kv < P.Skv) {
        T* dkrow = (T*)P.dk + (int64_t)bkv * P.dkbs + (int64_t)kv * P.dkrs + (int64_t)h * D;
        T* dvrow = (T*)P.dv + (int64_t)bkv * P.dkbs + (int64_t)kv * P.dkrs + (int64_t)h * D;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * half;
                if (d < D) {
                    store4<T>(dkrow + d, acck[dt][4 * g], acck[dt][4 * g + 1], acck[dt][4 * g + 2], acck[dt][4 * g + 3]);
                    store4<T>(dvrow + d, accv[dt][4 * g], accv[dt][4 * g + 1], accv[dt][4 * g + 2], accv[dt][4 * g + 3]);
                }
            }
    }
}

template <typename K>
void raise_lds(K kernel, size_t lds, FmcPerDeviceFlag& raised) {
    if (lds > 64 * 1024 && !raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        raised = true;
    }
}

template <typename T, int NKS>
void launch_bwd(SABwdParams P, hipStream_t st) {
    constexpr int NDT = (NKS + 1) / 2, KP = NKS * 16 + 8, VP = BK2 + 4;
    // dkdv geometry: LDS holds the workgroup's K,V block + a query tile in two layouts; sized to fit 160 KiB
    constexpr int WAVES = sizeof(T) == 2 ? (NKS >= 8 ? 2 : 4) : 1;
    constexpr int BQ = sizeof(T) == 2 ? 64 : 32;
    {
        const int64_t total = (int64_t)P.B * P.H * P.Sq;
        int64_t blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL((rowdot_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    }
    {
        const size_t lds = sizeof(T) * ((size_t)2 * BK2 * KP + (size_t)NDT * 32 * VP);
        static FmcPerDeviceFlag raised;
        raise_lds(&attn_dq_kernel<T, NKS>, lds, raised);
        P.nblk = (P.Sq + 127) / 128;
        P.xcd = (P.B % 8 == 0 && !getenv("FMC_SAB_XCD0")) ? 1 : 0;
        hipLaunchKernelGGL((attn_dq_kernel<T, NKS>), dim3((unsigned)(P.B * P.H * P.nblk)), dim3(256), lds, st, P);
    }
    if (P.dk) {                                       // dk == dv == NULL: the key/value side needs no gradient
        constexpr int BKV = 32 * WAVES;
        const size_t lds = sizeof(T) * ((size_t)2 * BKV * KP + (size_t)2 * BQ * KP + (size_t)2 * NDT * 32 * (BQ + 4)) +
                           2 * BQ * sizeof(float);
        static FmcPerDeviceFlag raised;
        raise_lds(&attn_dkdv_kernel<T, NKS, WAVES, BQ>, lds, raised);
        P.nblk = (P.Skv + BKV - 1) / BKV;
        const int bkv = P.B / P.kv_batch_div;
        P.xcd = (bkv % 8 == 0 && !getenv("FMC_SAB_XCD0")) ? 1 : 0;
        hipLaunchKernelGGL((attn_dkdv_kernel<T, NKS, WAVES, BQ>), dim3((unsigned)(bkv * P.H * P.nblk)), dim3(64 * WAVES), lds, st, P);
    }
}

template <typename T>
int dispatch_bwd(const SABwdParams& P, hipStream_t st) {
    switch ((P.D + 15) / 16) {
        case 1: launch_bwd<T, 1>(P, st); break;
        case 2: launch_bwd<T, 2>(P, st); break;
        case 3: launch_bwd<T, 3>(P, st); break;
        case 4: launch_bwd<T, 4>(P, st); break;
        case 5: launch_bwd<T, 5>(P, st); break;
        case 6: launch_bwd<T, 6>(P, st); break;
        case 8: launch_bwd<T, 8>(P, st); break;
        case 10: launch_bwd<T, 10>(P, st); break;
        default: FMC_FAIL(FMC_E_SHAPE, "spatial_attn_bwd: head dim %d not built", P.D);
    }
    return 0;
}

}  // namespace

extern "C" int fmc_spatial_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                    const float* lse, float* dvec, void* dq, void* dk, void* dv, int B, int H, int Sq,
                                    int Skv, int D, int64_t q_batch_stride, int64_t q_row_stride,
                                    int64_t kv_batch_stride, int64_t kv_row_stride, int64_t o_batch_stride,
                                    int64_t o_row_stride, int64_t dq_batch_stride, int64_t dq_row_stride,
                                    int64_t dkv_batch_stride, int64_t dkv_row_stride, int kv_batch_div, float scale,
                                    int dtype, void* stream) {
    if (!q || !k || !v || !o || !d_o || !lse || !dvec || !dq || (!dk != !dv))
        FMC_FAIL(FMC_E_NULL, "spatial_attn_bwd: NULL tensor (dk and dv may be NULL together)");
    if (dtype != FMC_BF16 && dtype != FMC_F32) FMC_FAIL(FMC_E_DTYPE, "spatial_attn_bwd: dtype %d", dtype);
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0 || D <= 0 || D % 8 || D > 160 || kv_batch_div <= 0 || B % kv_batch_div)
        FMC_FAIL(FMC_E_SHAPE, "spatial_attn_bwd: bad shape (B=%d H=%d Sq=%d Skv=%d D=%d div=%d)", B, H, Sq, Skv, D, kv_batch_div);
    const int64_t strides[] = {q_batch_stride, q_row_stride, kv_batch_stride, kv_row_stride, o_batch_stride, o_row_stride,
                               dq_batch_stride, dq_row_stride, dkv_batch_stride, dkv_row_stride};
    for (int64_t s : strides)
        if (s % 8) FMC_FAIL(FMC_E_ALIGN, "spatial_attn_bwd: strides must be multiples of 8 elements");
    const void* ptrs[] = {q, k, v, o, d_o, dq, dk, dv};
    for (const void* p : ptrs)
        if (p && !fmc_aligned16(p)) FMC_FAIL(FMC_E_ALIGN, "spatial_attn_bwd: tensors must be 16-byte aligned");
    SABwdParams P;
    P.q = q; P.k = k; P.v = v; P.o = o; P.d_o = d_o; P.lse = lse; P.dvec = dvec; P.dq = dq; P.dk = dk; P.dv = dv;
    P.B = B; P.H = H; P.Sq = Sq; P.Skv = Skv; P.D = D;
    P.qbs = q_batch_stride; P.qrs = q_row_stride; P.kbs = kv_batch_stride; P.krs = kv_row_stride;
    P.obs = o_batch_stride; P.ors = o_row_stride; P.dqbs = dq_batch_stride; P.dqrs = dq_row_stride;
    P.dkbs = dkv_batch_stride; P.dkrs = dkv_row_stride;
    P.kv_batch_div = kv_batch_div; P.scale = scale; P.scale_log2 = scale * LOG2E_B; P.nblk = 0;
    hipStream_t st = (hipStream_t)stream;
    int rc = dtype == FMC_BF16 ? dispatch_bwd<bf16_t>(P, st) : dispatch_bwd<float>(P, st);
    if (rc) return rc;
    FMC_CHECK_LAUNCH("fmc_spatial_attn_bwd");
    return 0;
}
