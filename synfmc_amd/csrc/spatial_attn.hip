// Spatial (per-frame) attention for gfx950: flash-style softmax(Q K^T * scale) V on the bf16 MFMA pipe.
//
// Replaces head_to_batch_dim + baddbmm + softmax + bmm + batch_to_head_dim of
// fmc/models/attention_processor.py:61-67 / :148-154 (attn1: self, attn2: text cross, S_kv = 77).
//
// Design (CDNA4, wave = 64):
//   * one workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 rows;
//   * K/V tiles of 64 keys are staged in LDS: K row-major [64][D+pad] (read as MFMA A operand with
//     ds_read_b128, pitch chosen bank-conflict free), V transposed [D][64+pad] (read as A operand of
//     the PV product with two ds_read_b64);
//   * "swapped" products:  S^T = K Q^T  and  O^T = V^T P^T  with v_mfma_f32_32x32x16_bf16, so a lane
//     holds 16 scores of ONE query column -> the online softmax is lane-local plus one
//     cross-half shuffle, and the rescale of O^T is a per-lane scalar;
//   * the P^T B-operand is taken straight from the S^T accumulator registers: the key order inside
//     a 16-wide K step is a fixed permutation, and the V^T fragment is read with the same permutation;
//   * head dims 40 / 80 / 160 (any multiple of 8 up to 160): the QK^T reduction dim is padded to a
//     multiple of 16 (48/80/160), the PV output dim to a multiple of 32 (64/96/160);
//   * blockIdx -> (batch*head, q-block) is XCD aware: all q-blocks of one (batch, head) run on the
//     same XCD so its K/V (<= 400 KB) is served from that XCD's L2 -- and, when the batch count allows, all HEADS
//     of a batch entry too: a head's 80-byte row slice of the fused [.., 3C] projection shares its 128-byte lines
//     with the neighbouring heads, so spreading the heads over the XCDs made every L2 fetch those lines again
//     (measured 2.7x the algorithmic read bytes);
//   * FMC_F32 storage runs every product as split-bf16 x3 (hi*hi + hi*lo + lo*hi), fp32 accumulate:
//     the parity mode (agrees with an fp32 reference to ~1e-5), 3x the MFMA work.
//
// Roofline: MFMA bound.  Algorithmic flops per launch = 4 * B*H * Sq*Skv * D.
#include "attn_common.h"
#include <cstdlib>

namespace {

constexpr int SA_WAVES = 4;
constexpr int SA_BQ = 32 * SA_WAVES;  // query rows per workgroup
constexpr bool SA_PREFETCH_DEFAULT = true;
constexpr int SA_BK = 64;             // keys per LDS tile
constexpr float LOG2E = 1.4426950408889634f;

struct SAParams {
    const void* q; const void* k; const void* v; void* o; float* lse;
    int B, H, Sq, Skv, D;
    int64_t qbs, qrs, kbs, krs, obs, ors;
    int kv_batch_div;
    float scale_log2;
    int nqblk;
    int xcd_remap;
};

// One 32-key x 32-query block for one wave: S^T = K Q^T, softmax numerators, O^T += V^T P^T.
//
// The softmax is arranged so that a block costs ~2 VALU per score (exp2 + its share of cvt_pk / max3) and has no
// branch and no conditionally updated state (a conditional rescale of the O^T tuples makes the register allocator
// copy them on every block):
//  * Q is pre-multiplied by scale*log2(e) when its fragments are built: scores leave the MFMA in log2 units;
//  * FIXED reference: m_ref is the row maximum over the first K tile (a prologue).  -m_ref lives in a 16-register
//    tuple that is the C operand of the first QK^T MFMA, so the accumulators already hold s - m_ref and
//    p = exp2(acc).  p, l and O are all relative to m_ref and O / l does not depend on it; bf16 and fp32 share an
//    8-bit exponent, so nothing overflows while max(s) - m_ref <= REF_LIMIT (log2 units).  Every block folds its
//    maximum into `worst`; if some row exceeded the limit the workgroup redoes the K/V sweep once with the now
//    exactly known row maxima (never seen on real activations; tested with adversarial inputs);
//  * when the padded V^T tile has a spare row (odd NKS: D = 40, 80, ...) that row holds ones, so the PV MFMAs
//    accumulate l = sum(p) in O^T row NDT*32-1 for free (and from the same rounded p as the numerator).
constexpr float REF_LIMIT = 64.f;
#ifndef SA_DBG
#define SA_DBG 0   // timing knock-outs (tools/ubench): 1 no staging, 2 barriers only, 3 staging without barriers, 4 no exp, 5 no max, 6 neither;
                   // pipelined kernel: 11 no LDS / DMA traffic, 12 no DMA, 13 no exp, 14 no barrier, 15 = 11 + 13; 20: cycle counters into `lse`
#endif

// max of three; this file is built with -fno-honor-nans so that the fmaxf chain selects v_max3_f32 without the NaN
// canonicalisation (v_max x,x) of every MFMA output.  (Not inline asm: the compiler must see these reads to insert the
// MFMA -> VALU wait states itself.)
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// scores of one block relative to the reference in `cinit` (= -m_ref, or 0 in the prologue)
template <typename T, int NKS, bool TAIL>
__device__ __forceinline__ f32x16 qk_block(const T* __restrict__ Ks, const Frag<T> (&qf)[NKS], const f32x16& cinit, int sb,
                                           int kvb, int Skv, int l31, int half) {
    constexpr int KP = NKS * 16 + 8;
    f32x16 s = cinit;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        Frag<T> kf;
        make_frag<T>(Ks + (sb * 32 + l31) * KP + ks * 16 + half * 8, kf);
        mma32(kf, qf[ks], s);
    }
    if (TAIL) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (kvb + (r & 3) + 8 * (r >> 2) + 4 * half >= Skv) s[r] = -INFINITY;
    }
    return s;
}

// max(w, s[0..15]) in 8 v_max3
__device__ __forceinline__ float fold_max(const f32x16& s, float w) {
    float ma = max3(s[0], s[1], s[2]), mb = max3(s[8], s[9], s[10]);
    ma = max3(ma, s[3], s[4]);
    mb = max3(mb, s[11], s[12]);
    ma = max3(ma, s[5], s[6]);
    mb = max3(mb, s[13], s[14]);
    return max3(max3(ma, mb, s[7]), s[15], w);
}

// VR (bf16): V stays ROW-major in LDS ([key][d], one ds_write_b128 per staged chunk instead of eight ds_write_b16 plus
// the unpacking) and the V^T fragments of the PV products come out of ds_read_b64_tr_b16: inside a 16-lane group lane i
// supplies the address of row i/4, columns 4(i%4)..+3 of a [4 keys][16 d] block and receives column i of it -- 4
// consecutive keys of one d, exactly one half of a fragment.  The pitch makes the 8 x 32-byte pieces a 32-lane half
// touches tile the 64 LDS banks exactly (pitch * 2 = 64 or 192 mod 256 bytes).
template <int NDT> constexpr int sa_vr_pitch() { return NDT == 1 ? 32 : (NDT == 2 ? 96 : 160); }
typedef short __attribute__((ext_vector_type(4))) sa_s4;
__device__ __forceinline__ sa_s4 lds_tr16(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sa_s4*)(p));
}

// NQ 32-query blocks per wave share every K / V^T fragment read: per 32 keys 3 + 4 LDS fragment loads feed 7*NQ MFMAs
// (at NQ = 1 the LDS port is ~80 % busy at d = 40: 12 waves per CU x 44 LDS cycles per 224 MFMA cycles)
// One 32-key block in two phases, so that the caller can issue the QK^T MFMAs of the NEXT block before the softmax of this
// one (the matrix pipe then works under the exp / max / convert VALU work instead of waiting for it).
template <typename T, int NKS, int NQ, bool MK>
__device__ __forceinline__ void qk_scores(const T* __restrict__ Ks, const Frag<T> (&qf)[NQ][NKS], const f32x16 (&negm)[NQ],
                                          f32x16 (&s)[NQ], int sb, int l31, int half) {
    constexpr int KP = NKS * 16 + 8;
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        if (MK) {                                    // the reference rides in the reduction (see the kernel): C = 0, no
#pragma unroll                                       // 16-register splat of -m_ref per block
            for (int r = 0; r < 16; ++r) s[nq][r] = 0.f;
        } else {
            s[nq] = negm[nq];
        }
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        Frag<T> kf;
        make_frag<T>(Ks + (sb * 32 + l31) * KP + ks * 16 + half * 8, kf);
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) mma32(kf, qf[nq][ks], s[nq]);
    }
}

template <typename T, int NKS, int NQ, bool TAIL, bool VR>
__device__ __forceinline__ void softmax_pv(const T* __restrict__ Vt, f32x16 (&s)[NQ], f32x16 (&oacc)[NQ][(NKS + 1) / 2],
                                           float (&worst)[NQ], float (&l_run)[NQ], int sb, int kvb, int Skv, int l31, int half) {
    constexpr int NDT = (NKS + 1) / 2;
    constexpr int VP = SA_BK + 4;
    constexpr bool LROW = (NKS & 1) != 0;
    Frag<T> pf[NQ][2];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        if (TAIL) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kvb + (r & 3) + 8 * (r >> 2) + 4 * half >= Skv) s[nq][r] = -INFINITY;
        }
        worst[nq] = fold_max(s[nq], worst[nq]);
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(s[nq][r]);
        if (!LROW) {
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) psum += p[r];
            l_run[nq] += psum;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float p8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) p8[i] = p[s2 * 8 + i];
            p_frag(p8, pf[nq][s2]);
        }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            Frag<T> vf;
            if constexpr (VR && sizeof(T) == 2) {
                constexpr int VPR = sa_vr_pitch<NDT>();
                const bf16_t* vp = reinterpret_cast<const bf16_t*>(Vt) +
                                   (sb * 32 + s2 * 16 + half * 4 + ((l31 & 15) >> 2)) * VPR + dt * 32 + (l31 >> 4) * 16 + (l31 & 3) * 4;
                union { bf16x8 v; sa_s4 h[2]; } r;
                r.h[0] = lds_tr16(vp);                   // keys +0..3 of this lane's d
                r.h[1] = lds_tr16(vp + 8 * VPR);         // keys +8..11 (the P^T fragment's key permutation)
                vf.hi = r.v;
            } else {
                const T* vrow = Vt + (dt * 32 + l31) * VP + sb * 32 + s2 * 16 + half * 4;
                make_frag_2x4<T>(vrow, vrow + 8, vf);
            }
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq) mma32(vf, pf[nq][s2], oacc[nq][dt]);
        }
    }
}

// The same, query block by query block with the V^T fragments of the 32-key block loaded once up front: the PV MFMAs of
// query block 0 are in flight while the VALU does the softmax of query block 1.
template <typename T, int NKS, int NQ, bool VR>
__device__ __forceinline__ void softmax_pv_perq(const T* __restrict__ Vt, f32x16 (&s)[NQ], f32x16 (&oacc)[NQ][(NKS + 1) / 2],
                                                float (&worst)[NQ], float (&l_run)[NQ], int sb, int l31, int half) {
    constexpr int NDT = (NKS + 1) / 2;
    constexpr int VP = SA_BK + 4;
    constexpr bool LROW = (NKS & 1) != 0;
    Frag<T> vf[2][NDT];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            if constexpr (VR && sizeof(T) == 2) {
                constexpr int VPR = sa_vr_pitch<NDT>();
                const bf16_t* vp = reinterpret_cast<const bf16_t*>(Vt) +
                                   (sb * 32 + s2 * 16 + half * 4 + ((l31 & 15) >> 2)) * VPR + dt * 32 + (l31 >> 4) * 16 + (l31 & 3) * 4;
                union { bf16x8 v; sa_s4 h[2]; } r;
                r.h[0] = lds_tr16(vp);
                r.h[1] = lds_tr16(vp + 8 * VPR);
                vf[s2][dt].hi = r.v;
            } else {
                const T* vrow = Vt + (dt * 32 + l31) * VP + sb * 32 + s2 * 16 + half * 4;
                make_frag_2x4<T>(vrow, vrow + 8, vf[s2][dt]);
            }
        }
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        if (SA_DBG != 5 && SA_DBG != 6) worst[nq] = fold_max(s[nq], worst[nq]);
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = (SA_DBG == 4 || SA_DBG == 6) ? s[nq][r] * 0.001f : __builtin_amdgcn_exp2f(s[nq][r]);
        if (!LROW) {
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) psum += p[r];
            l_run[nq] += psum;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float p8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) p8[i] = p[s2 * 8 + i];
            Frag<T> pf;
            p_frag(p8, pf);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) mma32(vf[s2][dt], pf, oacc[nq][dt]);
        }
    }
}

template <typename T, int NKS, int NQ, bool TAIL, bool MK = false, bool VR = false>
__device__ __forceinline__ void attn_block(const T* __restrict__ Ks, const T* __restrict__ Vt, const Frag<T> (&qf)[NQ][NKS],
                                           f32x16 (&oacc)[NQ][(NKS + 1) / 2], const f32x16 (&negm)[NQ], float (&worst)[NQ],
                                           float (&l_run)[NQ], int sb, int kvb, int Skv, int l31, int half) {
    f32x16 s[NQ];
    qk_scores<T, NKS, NQ, MK>(Ks, qf, negm, s, sb, l31, half);
    softmax_pv<T, NKS, NQ, TAIL, VR>(Vt, s, oacc, worst, l_run, sb, kvb, Skv, l31, half);
}

// SHORT_KV only separates the text cross-attention launches (S_kv = 77) from the self-attention ones in profiles
// (same code path): the two differ by >10x in work per launch and would blur a per-kernel-name average.
template <typename T> __device__ __forceinline__ T to_elem(float v);
template <> __device__ __forceinline__ bf16_t to_elem<bf16_t>(float v) { return f2bf(v); }
template <> __device__ __forceinline__ float to_elem<float>(float v) { return v; }

// PREFETCH keeps the next K/V tile in registers under the current tile's MFMAs (costs ~40 VGPRs).
template <typename T, int NKS, bool SHORT_KV, bool PREFETCH, int NQ, bool MK, bool VR, bool PIPE = (PREFETCH && sizeof(T) == 2)>
__global__ __launch_bounds__(64 * SA_WAVES, (sizeof(T) == 2) ? (NKS <= 3 && NQ == 1 ? 3 : 2) : 1) void spatial_attn_kernel(const SAParams P) {
    constexpr int NDT = (NKS + 1) / 2;
    constexpr int DP16 = NKS * 16;
    constexpr int KP = DP16 + 8;        // K tile pitch (elements)
    constexpr int VP = SA_BK + 4;       // V^T tile pitch (elements)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Ks = reinterpret_cast<T*>(smem_raw);              // [SA_BK][KP]
    T* Vt = Ks + SA_BK * KP;                             // [NDT*32][VP]  (VR: V row-major, [SA_BK][VPR])
    constexpr int VPR = sa_vr_pitch<NDT>();
    static_assert(!VR || (PREFETCH && sizeof(T) == 2), "VR: bf16 prefetch variants only");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int D = P.D, CH = D / 8;
    const uint64_t dbg_c0 = SA_DBG >= 20 ? __builtin_readcyclecounter() : 0, dbg_r0 = SA_DBG >= 20 ? __builtin_amdgcn_s_memrealtime() : 0;

    // ---- block -> (bh, q-block), XCD aware -----------------------------------------------------
    int bh, qblk;
    {
        const int id = blockIdx.x;
        if (P.xcd_remap == 2) {                         // all heads of a batch entry on one XCD (see the header)
            const int xcd = id & 7, within = id >> 3, per_b = P.H * P.nqblk;
            const int rem = within % per_b;
            bh = ((within / per_b) * 8 + xcd) * P.H + rem / P.nqblk;
            qblk = rem % P.nqblk;
        } else if (P.xcd_remap) {
            const int xcd = id & 7, within = id >> 3;
            bh = (within / P.nqblk) * 8 + xcd;
            qblk = within % P.nqblk;
        } else {
            bh = id / P.nqblk;
            qblk = id % P.nqblk;
        }
    }
    const int b = bh / P.H, h = bh - b * P.H;
    const T* qg = (const T*)P.q + (int64_t)b * P.qbs + (int64_t)h * D;
    const T* kg = (const T*)P.k + (int64_t)(b / P.kv_batch_div) * P.kbs + (int64_t)h * D;
    const T* vg = (const T*)P.v + (int64_t)(b / P.kv_batch_div) * P.kbs + (int64_t)h * D;
    T* og = (T*)P.o + (int64_t)b * P.obs + (int64_t)h * D;

    // K/V staging is split (T14-style): the 16-byte global loads of tile t+1 are issued right after tile t has been
    // written to LDS and stay in flight under tile t's MFMAs; they are written to LDS after the next barrier.
    constexpr int NSLOT = (SA_BK * NKS * 2 + 64 * SA_WAVES - 1) / (64 * SA_WAVES);   // 8-element chunks per thread
    constexpr int RW = sizeof(T) * 2;                                                  // dwords per chunk
    typedef uint32_t __attribute__((ext_vector_type(RW))) raw_t;
    raw_t kreg[NSLOT], vreg[NSLOT];
    auto gload = [&](int kv0) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const int c = tid + j * 64 * SA_WAVES;
            const int row = c / CH, ch = c - row * CH;
            const int kv = kv0 + row;
            if (c < SA_BK * CH && kv < P.Skv) {
                kreg[j] = *reinterpret_cast<const raw_t*>(kg + (int64_t)kv * P.krs + ch * 8);
                vreg[j] = *reinterpret_cast<const raw_t*>(vg + (int64_t)kv * P.krs + ch * 8);
            } else {
                kreg[j] = raw_t(0);
                vreg[j] = raw_t(0);
            }
        }
    };
    // tile 0 is requested BEFORE the Q rows: the two round trips overlap (behind the Q fragments it was a second dependent one --
    // the short cross-attention launches are a chain of such round trips: Q, tile 0, tile 1, stores)
    if (PREFETCH) gload(0);

    // ---- Q^T fragments (B operand of S^T = K Q^T), kept in registers -----------------------------
    int qrow[NQ];
    Frag<T> qf[NQ][NKS];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        qrow[nq] = (qblk * SA_WAVES + wave) * (32 * NQ) + nq * 32 + l31;
        if constexpr (sizeof(T) == 2) {
            // all chunks of the row requested at once from clamped addresses, masked afterwards: a load inside a branch is
            // followed by its own vmcnt(0) -- NKS dependent round trips at the head of every workgroup
            const bf16_t* qp = reinterpret_cast<const bf16_t*>(qg) + (int64_t)(qrow[nq] < P.Sq ? qrow[nq] : P.Sq - 1) * P.qrs;
            u32x4 qraw[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int d0 = ks * 16 + half * 8;
                qraw[ks] = *reinterpret_cast<const u32x4*>(qp + (d0 < D ? d0 : 0));
            }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bool live = qrow[nq] < P.Sq && ks * 16 + half * 8 < D;
                float qv[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    qv[2 * i] = live ? __uint_as_float(qraw[ks][i] << 16) * P.scale_log2 : 0.f;
                    qv[2 * i + 1] = live ? __uint_as_float(qraw[ks][i] & 0xffff0000u) * P.scale_log2 : 0.f;
                }
                p_frag(qv, qf[nq][ks]);
            }
            continue;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d0 = ks * 16 + half * 8;
            if (qrow[nq] < P.Sq && d0 < D) {
                float qv[8];
                Vec8<T>::load(qg + (int64_t)qrow[nq] * P.qrs + d0, qv);
#pragma unroll
                for (int i = 0; i < 8; ++i) qv[i] *= P.scale_log2;
                p_frag(qv, qf[nq][ks]);   // fp32 values -> fragment(s): bf16 rounding / hi+lo split
            } else {
                zero(qf[nq][ks]);
            }
        }
    }

    // zero the K pad columns once (d in [D, DP16)): 0 * garbage must stay 0
    if (DP16 > D) {
        for (int r = tid; r < SA_BK; r += blockDim.x) {
#pragma unroll
            for (int i = 0; i < 8; ++i) Ks[r * KP + D + i] = (MK && i == 0) ? to_elem<T>(1.f) : T(0);
        }
    }

    constexpr bool LROW = (NKS & 1) != 0;
    if (LROW) {   // ones in the last (spare) V^T row: O^T row NDT*32-1 accumulates l = sum(p)
        for (int c = tid; c < SA_BK; c += 64 * SA_WAVES) Vt[VR ? c * VPR + NDT * 32 - 1 : (NDT * 32 - 1) * VP + c] = to_elem<T>(1.f);
    }

    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const int c = tid + j * 64 * SA_WAVES;
            if (c < SA_BK * CH) {
                const int row = c / CH, ch = c - row * CH;
                *reinterpret_cast<raw_t*>(Ks + row * KP + ch * 8) = kreg[j];
                if constexpr (VR) {
                    *reinterpret_cast<raw_t*>(Vt + row * VPR + ch * 8) = vreg[j];
                } else if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        Vt[(ch * 8 + 2 * i) * VP + row] = (T)(vreg[j][i] & 0xffffu);
                        Vt[(ch * 8 + 2 * i + 1) * VP + row] = (T)(vreg[j][i] >> 16);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) Vt[(ch * 8 + i) * VP + row] = __uint_as_float(vreg[j][i]);
                }
            }
        }
    };
    auto stage_rolled = [&](int kv0) {   // one chunk live at a time (register-starved variants)
        for (int c = tid; c < SA_BK * CH; c += 64 * SA_WAVES) {
            const int row = c / CH, ch = c - row * CH;
            const int kv = kv0 + row;
            float kvals[8], vvals[8];
            if (kv < P.Skv) {
                Vec8<T>::load(kg + (int64_t)kv * P.krs + ch * 8, kvals);
                Vec8<T>::load(vg + (int64_t)kv * P.krs + ch * 8, vvals);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) kvals[i] = vvals[i] = 0.f;
            }
            Vec8<T>::store(Ks + row * KP + ch * 8, kvals);
#pragma unroll
            for (int i = 0; i < 8; ++i) Vt[(ch * 8 + i) * VP + row] = to_elem<T>(vvals[i]);
        }
    };

    const int ntiles = (P.Skv + SA_BK - 1) / SA_BK, nfull = P.Skv / SA_BK;

    // ---- prologue: tile 0 -> LDS, reference m_ref = row maximum over its keys -----------------------
    if (PREFETCH) { lstore(); } else { stage_rolled(0); }
    __syncthreads();
    if (PREFETCH && ntiles > 1) gload(SA_BK);
    float m_ref[NQ];
    f32x16 negm[NQ];
    // MK (bf16, D % 16 == 8): the reference is subtracted by the QK^T MFMAs themselves -- K's first pad column holds 1, the
    // matching (otherwise zero) element of the Q^T fragment holds -m_ref -- so the score accumulator starts from the
    // inline constant 0 instead of a 16-register splat of -m_ref that the compiler re-materialised for every block
    // (60 v_mov per 64-key tile).  The reference is rounded to bf16 for that; it only has to be the SAME for all keys
    // of a row (softmax is shift invariant), and the rounded value is what the LSE and the redo test use.
    auto set_ref = [&](int nq) {
        if (MK) {
            const bf16_t mb = f2bf(-m_ref[nq]);
            m_ref[nq] = -bf2f(mb);
            union { bf16x8 v; bf16_t e[8]; } u;
            u.v = qf[nq][NKS - 1].hi;
            if (half == 1) u.e[0] = mb;                  // d = D of k-step NKS-1 (D % 16 == 8: upper half, element 0)
            qf[nq][NKS - 1].hi = u.v;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[nq][r] = -m_ref[nq];
    };
    {
        f32x16 zero16;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) {
            float mx = fold_max(qk_block<T, NKS, true>(Ks, qf[nq], zero16, 0, 0, P.Skv, l31, half), -INFINITY);
            if (P.Skv > 32) mx = fold_max(qk_block<T, NKS, true>(Ks, qf[nq], zero16, 1, 32, P.Skv, l31, half), mx);
            m_ref[nq] = fmaxf(mx, __shfl_xor(mx, 32, 64));   // finite: key 0 always exists
            set_ref(nq);
        }
    }

    f32x16 oacc[NQ][NDT];
    float l_run[NQ];
    for (int pass = 0; pass < 2; ++pass) {
        float worst[NQ];           // max over all keys of s - m_ref (this lane's half of the keys)
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[nq][dt][r] = 0.f;
            l_run[nq] = 0.f;
            worst[nq] = -INFINITY;
        }

        auto stage = [&](int tile) {
            if (tile > 0 || pass > 0) {             // tile 0 is already resident after the prologue
                if (SA_DBG == 1) return;
                if (pass > 0 && tile == 0 && PREFETCH) gload(0);
                if (SA_DBG != 3) __syncthreads();  // previous tile fully consumed
                if (SA_DBG != 2) { if (PREFETCH) lstore(); else stage_rolled(tile * SA_BK); }
                if (SA_DBG != 3) __syncthreads();
                if (SA_DBG != 2) if (PREFETCH && tile + 1 < ntiles) gload((tile + 1) * SA_BK);
            }
        };
        // full tiles: no key masking anywhere, no branch in the body (a second path merging into the loop makes the
        // register allocator copy the O^T tuples every iteration); the ragged last tile is peeled.
        for (int tile = 0; tile < nfull; ++tile) {
            stage(tile);
            if constexpr (SA_BK == 64 && sizeof(T) == 2 && PIPE) {
                // QK^T of both 32-key blocks first: the second one's MFMAs run under the softmax VALU work of the first, the
                // first one's PV MFMAs under the softmax of the second (PMC: matrix pipe 34 % + other VALU ~40 % of the SIMD
                // cycles, one after the other, in the block-by-block order)
                f32x16 s0[NQ], s1[NQ];
                qk_scores<T, NKS, NQ, MK>(Ks, qf, negm, s0, 0, l31, half);
                qk_scores<T, NKS, NQ, MK>(Ks, qf, negm, s1, 1, l31, half);
                softmax_pv_perq<T, NKS, NQ, VR>(Vt, s0, oacc, worst, l_run, 0, l31, half);
                softmax_pv_perq<T, NKS, NQ, VR>(Vt, s1, oacc, worst, l_run, 1, l31, half);
            } else {
#pragma unroll
                for (int sb = 0; sb < SA_BK / 32; ++sb)
                    attn_block<T, NKS, NQ, false, MK, VR>(Ks, Vt, qf, oacc, negm, worst, l_run, sb, tile * SA_BK + sb * 32, P.Skv, l31, half);
            }
        }
        if (nfull < ntiles) {
            stage(nfull);
#pragma unroll
            for (int sb = 0; sb < SA_BK / 32; ++sb)
                if (nfull * SA_BK + sb * 32 < P.Skv)           // block-uniform
                    attn_block<T, NKS, NQ, true, MK, VR>(Ks, Vt, qf, oacc, negm, worst, l_run, sb, nfull * SA_BK + sb * 32, P.Skv, l31, half);
        }
        // did any row of this workgroup leave the safe range of the fixed reference?  (block-uniform decision)
        bool bad = false;
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) bad = bad || worst[nq] > REF_LIMIT;
        if (!__syncthreads_or(bad)) break;
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) {
            m_ref[nq] += fmaxf(worst[nq], __shfl_xor(worst[nq], 32, 64));     // now the exact row maximum
            set_ref(nq);
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------------
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        float l_tot;
        if (LROW) l_tot = __shfl(oacc[nq][NDT - 1][15], l31 + 32, 64);   // row NDT*32-1 lives in the upper half's register 15
        else l_tot = l_run[nq] + __shfl_xor(l_run[nq], 32, 64);
        const float inv = 1.f / l_tot;
        if (qrow[nq] < P.Sq) {
            T* orow = og + (int64_t)qrow[nq] * P.ors;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * half;
                    if (d < D)
                        store4<T>(orow + d, oacc[nq][dt][4 * g] * inv, oacc[nq][dt][4 * g + 1] * inv,
                                  oacc[nq][dt][4 * g + 2] * inv, oacc[nq][dt][4 * g + 3] * inv);
                }
            }
            if (SA_DBG < 20 && P.lse && half == 0)
                P.lse[((int64_t)b * P.H + h) * P.Sq + qrow[nq]] = m_ref[nq] * 0.6931471805599453f + logf(l_tot);
        }
    }
    if (SA_DBG >= 20 && P.lse && lane == 0) {
        float* dst = P.lse + ((int64_t)blockIdx.x * 4 + wave) * 4;
        dst[0] = (float)(__builtin_readcyclecounter() - dbg_c0); dst[1] = 0.f;
        dst[2] = (float)(__builtin_amdgcn_s_memrealtime() - dbg_r0); dst[3] = (float)blockIdx.x;
    }
}


// =====================================================================================================================
// Software-pipelined kernel for the level-0 self-attention (bf16, D = 40, S_kv % 64 == 0, S_q % 256 == 0).
//
// Same fragment mapping, fixed-reference softmax (reference through the QK^T reduction) and ones-row denominator as
// the NQ = 2 / MK / VR variant above; what changes is the instruction order and how the tiles get into LDS.
// Knock-out timings of that variant (tools/ubench/sa_bench + SA_DBG) showed the matrix pipe 34 % busy although neither
// the exp work (-8 % when removed) nor the barriers (0 %) nor the LDS bandwidth were the limit: the wave stalls on
// its OWN dependencies (ds_read -> MFMA, MFMA -> v_exp, v_cvt -> MFMA), and a micro-benchmark
// (tools/ubench/pingpong.hip) reaches 85-95 % matrix pipe occupancy with the same instruction mix once every MFMA is
// followed by ~4 independent VALU.  So the sweep is a pipeline over units u = (tile, 32-key block, 32-query block),
// one step per unit:
//     matrix pipe:  QK^T of unit u+1 (3 MFMA)  and  P V of unit u-1 (4 MFMA), alternating
//     VALU:         softmax numerators of unit u (16 v_exp + 8 v_cvt_pk), 3-4 after each MFMA
//     LDS:          the fragments of the NEXT step's MFMAs, issued at the head of the step
// Every operand of a step was produced a full step earlier.  `sched_barrier(0)` pins the order; the compiler still
// inserts the wait counts and hazard nops.  Measured inside the kernel (SA_DBG=20: s_memtime / s_memrealtime): 2130
// cycles per 64-key tile and wave with two waves per SIMD, i.e. the matrix pipe is 84 % busy during the sweep -- at a
// shader clock of 1.5-1.6 GHz (the block-by-block kernel: 1.75 GHz): the launch is power limited, which is why the
// gain in wall time (0.44 -> 0.36 ms) is smaller than the gain in cycles.
// No running maximum: an out-of-range row shows as l > 2^64 at the end (the denominator rides in the PV MFMAs), the
// workgroup then finds the exact row maxima in a plain sweep and repeats the pipelined sweep once (never on real
// activations; adversarial test).
//
// The tiles are brought in by buffer_load ... lds (no staging registers, no ds_write).
// A wave's DMA instruction writes 64 consecutive 16-byte chunks, so the tiles are unpadded (5 chunks per row):
//   * K rows in key order; a b128 fragment read of 32 consecutive rows at 80-byte pitch is conflict free;
//   * V rows at slot 16 (k/16) + (k/4)%4 + 4 (k%4) -- a 4x4 transpose inside every 16 keys -- so that the 4 consecutive keys
//     a 16-lane group of ds_read_b64_tr_b16 touches start 16 banks apart;
//   * what the padded layout kept in pad columns now comes from constant regions through per-lane base addresses (the
//     immediates of the reads are shared by all lanes, so a region spans the largest immediate): the K chunk [1, 0 x7]
//     for the upper-half lanes of k-step 2 (d = 40..47: the reference rides in the reduction), zeros for the lanes
//     supplying V columns 40..59, the pattern [0, 0, 0, 1] for columns 60..63 (the denominator row).  Zeros rather
//     than a neighbouring row's bytes: the kernel runs power limited (shader clock 1.55-1.75 GHz, tools/ubench) and a
//     multiplier fed zeros costs less.
// Four buffers; tile t+3 is requested in step 2 of tile t, behind the barrier that retires tile t-1's buffer.
// Persistent workgroups (2 per CU) walk the (batch, head, 256-query block) items: the next item's DMAs and Q loads go
// out together, and no workgroup launch sits between two items.
struct PFrag { unsigned u[4]; };
__device__ __forceinline__ bf16x8 as_frag(const PFrag& p) {
    union { bf16x8 v; unsigned u[4]; } r;
    r.u[0] = p.u[0]; r.u[1] = p.u[1]; r.u[2] = p.u[2]; r.u[3] = p.u[3];
    return r.v;
}
__device__ __forceinline__ bf16x8 lds_frag(const bf16_t* p) {
    union { bf16x8 v; u32x4 u; } r;
    r.u = *reinterpret_cast<const u32x4*>(p);
    return r.v;
}
#define P40_SB() __builtin_amdgcn_sched_barrier(0)
#define P40_EXP(x) ((SA_DBG == 13 || SA_DBG == 15) ? (x) * 0.001f : __builtin_amdgcn_exp2f(x))
constexpr int D40_TILE = SA_BK * 40;                                        // elements of one unpadded tile (5120 B)
constexpr int D40_BUF = 2 * D40_TILE;                                       // K tile, then V tile
constexpr int D40_KCONST = 4 * D40_BUF;                                     // [1, 0 x7] at +0 and at +32 rows (the two key blocks)
constexpr int D40_DUMP = D40_KCONST + 16;                                   // 2 KB scratch for the two dummy DMAs (inside the gap)
constexpr int D40_VZERO = D40_KCONST + 32 * 40 + 16;                        // 2048 elements of zeros
constexpr int D40_VCONST = D40_VZERO + 2048 + 8;                            // 2048 elements of [0, 0, 0, 1]; +8: not the zero region's banks
constexpr size_t D40_LDS = (size_t)(D40_VCONST + 2048) * 2;
static_assert(D40_DUMP + 1024 <= D40_KCONST + 32 * 40, "scratch must fit between the two K constants");

__global__ __launch_bounds__(256, 2) void sa40d_kernel(const SAParams P, const int nitems) {
    constexpr int NKS = 3, D = 40, CH = 5, RP = 40;                        // RP: row pitch (elements) of both tiles
    constexpr int KB_STEP = 32 * RP, VS_STEP = 16 * RP, V8 = 2 * RP;      // key block / 16-key group / keys +8 (2 slots)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const uint64_t dbg_c0 = SA_DBG >= 20 ? __builtin_readcyclecounter() : 0, dbg_r0 = SA_DBG >= 20 ? __builtin_amdgcn_s_memrealtime() : 0;
    uint64_t dbg_sweep = 0;

    // constant regions (once per workgroup)
    if (tid < 16) smem[D40_KCONST + (tid >> 3) * KB_STEP + (tid & 7)] = ((tid & 7) == 0) ? f2bf(1.f) : (bf16_t)0;
    for (int i = tid; i < 2048; i += 256) {
        smem[D40_VZERO + i] = (bf16_t)0;
        smem[D40_VCONST + i] = ((i & 3) == 3) ? f2bf(1.f) : (bf16_t)0;
    }

    // per-lane pieces of the DMA (see above): chunk c = 64 * block + lane of a tile -> (row or slot, chunk in row)
    const int krsB = (int)P.krs * 2, tile_bytes = SA_BK * krsB;
    auto vkey = [](int slot) { return (slot & ~15) + (slot & 3) * 4 + ((slot >> 2) & 3); };
    const int c0 = wave * 64 + lane, c2 = 256 + lane;
    const unsigned vo0 = (unsigned)((c0 / CH) * krsB + (c0 % CH) * 16);
    const unsigned vo1 = (unsigned)(vkey(c0 / CH) * krsB + (c0 % CH) * 16);
    const unsigned vo2 = wave == 0 ? (unsigned)((c2 / CH) * krsB + (c2 % CH) * 16)
                                   : (wave == 1 ? (unsigned)(vkey(c2 / CH) * krsB + (c2 % CH) * 16) : 0x7fff0000u);
    const int ld0 = wave * 512, ld1 = D40_TILE + wave * 512, ld2 = wave == 0 ? 4 * 512 : D40_TILE + 4 * 512;
    const int kv_bytes = (int)(((int64_t)(P.Skv - 1) * P.krs + D) * 2);
    const int ntiles = P.Skv / SA_BK;

    // per-lane fragment addresses inside a buffer (elements).  K: row kb*32 + l31, chunk 2 ks + half.  V^T: slot
    // 16 (2 kb + s2) + half + 4 r (+2 for keys +8), r = (l31 & 15) >> 2, columns dt*32 + (l31 >> 4)*16 + (l31 & 3)*4.
    const int ka = l31 * RP + half * 8;
    const int va = D40_TILE + (half + 4 * ((l31 & 15) >> 2)) * RP + (l31 >> 4) * 16 + (l31 & 3) * 4;
    const bool kconst = half == 1;
    const int vsel = (l31 >> 4) == 0 ? ((l31 & 3) >= 2 ? 1 : 0) : ((l31 & 3) == 3 ? 2 : 1);   // dt = 1: 0 data, 1 zeros, 2 [0,0,0,1]

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        int bh, qblk;
        if (P.xcd_remap == 2) {
            const int xcd = item & 7, within = item >> 3, per_b = P.H * P.nqblk;
            const int rem = within % per_b;
            bh = ((within / per_b) * 8 + xcd) * P.H + rem / P.nqblk;
            qblk = rem % P.nqblk;
        } else if (P.xcd_remap) {
            const int xcd = item & 7, within = item >> 3;
            bh = (within / P.nqblk) * 8 + xcd;
            qblk = within % P.nqblk;
        } else {
            bh = item / P.nqblk;
            qblk = item % P.nqblk;
        }
        const int b = bh / P.H, h = bh - b * P.H;
        const bf16_t* qg = (const bf16_t*)P.q + (int64_t)b * P.qbs + (int64_t)h * D;
        const bf16_t* kg = (const bf16_t*)P.k + (int64_t)(b / P.kv_batch_div) * P.kbs + (int64_t)h * D;
        const bf16_t* vg = (const bf16_t*)P.v + (int64_t)(b / P.kv_batch_div) * P.kbs + (int64_t)h * D;
        bf16_t* og = (bf16_t*)P.o + (int64_t)b * P.obs + (int64_t)h * D;
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kg, 0, kv_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vg, 0, kv_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs2 = (wave & 1) ? rsV : rsK;
        auto dma = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, int soff, bf16_t* lds) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, soff, 0, 0);
        };
        auto stage_tile = [&](int tile, int buf) {
            const int so = tile * tile_bytes;
            bf16_t* base = smem + buf * D40_BUF;
            dma(rsK, vo0, so, base + ld0);
            dma(rsV, vo1, so, base + ld1);
            dma(rs2, vo2, so, wave < 2 ? base + ld2 : smem + D40_DUMP + (wave - 2) * 512);
        };
        // fragment loads of key block kb out of buffer `buf`
        auto kfrags = [&](const bf16_t* buf, int kb, bf16x8 (&kf)[NKS]) {
            const bf16_t* k01 = buf + ka + kb * KB_STEP;
            const bf16_t* k2 = kconst ? smem + D40_KCONST + kb * KB_STEP : k01 + 32;
            kf[0] = lds_frag(k01); kf[1] = lds_frag(k01 + 16); kf[2] = lds_frag(k2);
        };
        auto vfrags = [&](const bf16_t* buf, int kb, bf16x8 (&vf)[2][2]) {
            const bf16_t* v0 = buf + va + kb * 2 * VS_STEP;
            const bf16_t* v1 = vsel == 0 ? v0 + 32 : smem + (vsel == 1 ? D40_VZERO : D40_VCONST) + kb * 2 * VS_STEP;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                union { bf16x8 v; sa_s4 h[2]; } r0, r1;
                r0.h[0] = lds_tr16(v0 + s2 * VS_STEP); r0.h[1] = lds_tr16(v0 + s2 * VS_STEP + V8);
                r1.h[0] = lds_tr16(v1 + s2 * VS_STEP); r1.h[1] = lds_tr16(v1 + s2 * VS_STEP + V8);
                vf[s2][0] = r0.v; vf[s2][1] = r1.v;
            }
        };

        // ---- item prologue: tiles 0..2 requested, then the Q^T fragments while they fly ---------------------------------
        __syncthreads();                                   // the previous item's fragment reads (and the constants) are done
        stage_tile(0, 0); stage_tile(1, 1); stage_tile(2, 2);
        int qrow[2];
        bf16x8 qf[2][NKS];
#pragma unroll
        for (int nq = 0; nq < 2; ++nq) {
            qrow[nq] = (qblk * SA_WAVES + wave) * 64 + nq * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int d0 = ks * 16 + half * 8;
                Frag<bf16_t> f;
                if (d0 < D) {
                    float qv[8];
                    Vec8<bf16_t>::load(qg + (int64_t)qrow[nq] * P.qrs + d0, qv);
#pragma unroll
                    for (int i = 0; i < 8; ++i) qv[i] *= P.scale_log2;
                    p_frag(qv, f);
                } else {
                    zero(f);
                }
                qf[nq][ks] = f.hi;
            }
        }
        float m_ref[2];
        auto set_ref = [&](int nq) {
            const bf16_t mb = f2bf(-m_ref[nq]);
            m_ref[nq] = -bf2f(mb);
            union { bf16x8 v; bf16_t e[8]; } u;
            u.v = qf[nq][NKS - 1];
            if (half == 1) u.e[0] = mb;
            qf[nq][NKS - 1] = u.v;
        };
        auto scores = [&](const bf16x8 (&kf)[NKS], int nq) {            // plain order, relative to whatever reference sits in qf
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[nq][ks], sacc, 0, 0, 0);
            return sacc;
        };

        f32x16 oacc[2][2];
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) { __syncthreads(); stage_tile(0, 0); stage_tile(1, 1); stage_tile(2, 2); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            bf16x8 kf0[NKS], kf1[NKS], vf0[2][2], vf1[2][2];
            kfrags(smem, 0, kf0);
            if (pass == 0) {                   // reference: row maximum over tile 0 (finite: its keys exist)
                kfrags(smem, 1, kf1);
#pragma unroll
                for (int nq = 0; nq < 2; ++nq) {
                    float mx = fold_max(scores(kf0, nq), -INFINITY);
                    mx = fold_max(scores(kf1, nq), mx);
                    m_ref[nq] = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    set_ref(nq);
                }
            }
#pragma unroll
            for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[nq][dt][r] = 0.f;

            // ---- pipeline fill: S of unit (0, kb0, q0); an all-zero P for the "previous" unit (times finite V of tile 0) ----
            f32x16 sA, sB;
            PFrag pA[2], pB[2];
            vfrags(smem, 0, vf0);
            vfrags(smem, 1, vf1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < 4; ++i) pB[s2].u[i] = 0u;
            sA = scores(kf0, 0);

            // one pipeline step.  sS -> pS: softmax numerators of this unit; sQ = K(kfQ) Q^T(qfQ): the next unit's scores;
            // oP += V^T(vfP) P^T(pP): the previous unit's output.  `head` issues this step's LDS / DMA traffic.
            auto step = [&](const f32x16& sS, PFrag (&pS)[2], f32x16& sQ, const bf16x8 (&kfQ)[NKS], const bf16x8 (&qfQ)[NKS],
                            const PFrag (&pP)[2], const bf16x8 (&vfP)[2][2], f32x16 (&oP)[2], auto&& head) {
                float e[16];
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                P40_SB();
                if (SA_DBG != 11 && SA_DBG != 15) head();
                P40_SB();
                sQ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfQ[0], qfQ[0], z, 0, 0, 0);
                P40_SB();
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = P40_EXP(sS[i]);
                P40_SB();
                oP[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfP[0][0], as_frag(pP[0]), oP[0], 0, 0, 0);
                P40_SB();
                pS[0].u[0] = pack_bf2(e[0], e[1]); pS[0].u[1] = pack_bf2(e[2], e[3]);
                e[4] = P40_EXP(sS[4]); e[5] = P40_EXP(sS[5]);
                P40_SB();
                sQ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfQ[1], qfQ[1], sQ, 0, 0, 0);
                P40_SB();
                e[6] = P40_EXP(sS[6]); e[7] = P40_EXP(sS[7]);
                pS[0].u[2] = pack_bf2(e[4], e[5]);
                e[8] = P40_EXP(sS[8]);
                pS[0].u[3] = pack_bf2(e[6], e[7]);
                P40_SB();
                oP[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfP[0][1], as_frag(pP[0]), oP[1], 0, 0, 0);
                P40_SB();
                e[9] = P40_EXP(sS[9]); e[10] = P40_EXP(sS[10]); e[11] = P40_EXP(sS[11]);
                P40_SB();
                sQ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfQ[2], qfQ[2], sQ, 0, 0, 0);
                P40_SB();
                pS[1].u[0] = pack_bf2(e[8], e[9]); pS[1].u[1] = pack_bf2(e[10], e[11]);
                e[12] = P40_EXP(sS[12]); e[13] = P40_EXP(sS[13]);
                P40_SB();
                oP[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfP[1][0], as_frag(pP[1]), oP[0], 0, 0, 0);
                P40_SB();
                e[14] = P40_EXP(sS[14]); e[15] = P40_EXP(sS[15]);
                pS[1].u[2] = pack_bf2(e[12], e[13]);
                P40_SB();
                oP[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfP[1][1], as_frag(pP[1]), oP[1], 0, 0, 0);
                P40_SB();
                pS[1].u[3] = pack_bf2(e[14], e[15]);
                P40_SB();
            };

            const uint64_t dbg_s0 = SA_DBG >= 20 ? __builtin_readcyclecounter() : 0;
            for (int t = 0; t < ntiles; ++t) {
                const bf16_t* cur = smem + (t & 3) * D40_BUF;
                const bf16_t* nxt = smem + ((t + 1) & 3) * D40_BUF;
                // step 0: unit (kb0, q0); QK^T of (kb0, q1); PV of the previous tile's (kb1, q1); K fragments of kb1
                step(sA, pA, sB, kf0, qf[1], pB, vf1, oacc[1], [&] { kfrags(cur, 1, kf1); });
                // step 1: unit (kb0, q1); QK^T of (kb1, q0); PV of (kb0, q0); V^T fragments of kb1
                step(sB, pB, sA, kf1, qf[0], pA, vf0, oacc[0], [&] { vfrags(cur, 1, vf1); });
                // behind this barrier every wave is done reading tile t-1's buffer (tile t+3 goes there) and tile t+1 has landed:
                // the three youngest DMAs of a wave belong to tile t+2
                if (SA_DBG != 11 && SA_DBG != 14 && SA_DBG != 15) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                // step 2: unit (kb1, q0); QK^T of (kb1, q1); PV of (kb0, q1); tile t+3 requested; K fragments of the next tile
                step(sA, pA, sB, kf1, qf[1], pB, vf0, oacc[1], [&] {
                    if (SA_DBG != 12) stage_tile(t + 3, (t + 3) & 3);
                    kfrags(nxt, 0, kf0);
                });
                // step 3: unit (kb1, q1); QK^T of the next tile's (kb0, q0); PV of (kb1, q0); V^T fragments of the next tile's kb0
                step(sB, pB, sA, kf0, qf[0], pA, vf1, oacc[0], [&] { vfrags(nxt, 0, vf0); });
            }
            if (SA_DBG >= 20) dbg_sweep += __builtin_readcyclecounter() - dbg_s0;
            // drain: PV of the last unit
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    oacc[1][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1[s2][dt], as_frag(pB[s2]), oacc[1][dt], 0, 0, 0);

            // ---- did a row leave the range of the fixed reference?  (block-uniform) -----------------------------------------
            bool bad = false;
#pragma unroll
            for (int nq = 0; nq < 2; ++nq) bad = bad || (half == 1 && oacc[nq][1][15] > 1.8446744e19f);
            if (pass == 1 || !__syncthreads_or(bad)) break;
            // exact row maxima in a plain sweep (tile by tile through buffer 0), then once more
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the out-of-range tiles the pipeline still had in flight
            float worst[2] = {-INFINITY, -INFINITY};
            for (int t = 0; t < ntiles; ++t) {
                __syncthreads();
                stage_tile(t, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                kfrags(smem, 0, kf0);
                kfrags(smem, 1, kf1);
#pragma unroll
                for (int nq = 0; nq < 2; ++nq) {
                    worst[nq] = fold_max(scores(kf0, nq), worst[nq]);
                    worst[nq] = fold_max(scores(kf1, nq), worst[nq]);
                }
            }
#pragma unroll
            for (int nq = 0; nq < 2; ++nq) {
                m_ref[nq] += fmaxf(worst[nq], __shfl_xor(worst[nq], 32, 64));
                set_ref(nq);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // out-of-range tiles still in flight must not land in the next item's buffers

        // ---- item epilogue ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int nq = 0; nq < 2; ++nq) {
            const float l_tot = __shfl(oacc[nq][1][15], l31 + 32, 64);   // O^T row 63 (the [0,0,0,1] column of V): upper half, register 15
            const float inv = 1.f / l_tot;
            bf16_t* orow = og + (int64_t)qrow[nq] * P.ors;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * half;
                    if (d < D)
                        store4<bf16_t>(orow + d, oacc[nq][dt][4 * g] * inv, oacc[nq][dt][4 * g + 1] * inv, oacc[nq][dt][4 * g + 2] * inv,
                                       oacc[nq][dt][4 * g + 3] * inv);
                }
            }
            if (SA_DBG < 20 && P.lse && half == 0) P.lse[((int64_t)b * P.H + h) * P.Sq + qrow[nq]] = m_ref[nq] * 0.6931471805599453f + logf(l_tot);
        }
    }
    if (SA_DBG >= 20 && P.lse && lane == 0) {     // shader cycles: whole workgroup, pipelined sweeps; 100 MHz ticks: whole workgroup
        float* dst = P.lse + ((int64_t)blockIdx.x * 4 + wave) * 4;
        dst[0] = (float)(__builtin_readcyclecounter() - dbg_c0); dst[1] = (float)dbg_sweep;
        dst[2] = (float)(__builtin_amdgcn_s_memrealtime() - dbg_r0); dst[3] = (float)blockIdx.x;
    }
}

// =====================================================================================================================
// Text cross-attention (bf16, D = 40, H = 8, S_kv <= 96: the 77 text tokens), K / V stationary.
//
// The block-by-block kernel spends a 77-key launch on per-workgroup latencies (Q rows, K/V tile 0, a reference pass, tile 1,
// two barriers each; 53 us at level 0 against 17 us for the launch's bytes).  Here a wave owns one head and keeps that head's K
// fragments (3 key blocks x 3 k-steps) and V^T fragments (3 x 2 x 2, transposed once through a wave-private LDS tile with
// ds_read_b64_tr_b16, ones in column 63 so that the PV MFMAs also produce the denominator) in REGISTERS for its whole life, and
// walks 32-query units: Q fragments straight from global memory (the next unit's requested before this unit's MFMAs), 9 + 12
// MFMAs, an exact softmax (all 96 scores of a query are in two lanes' registers: no online rescaling, no reference pass), 8-byte
// stores.  No LDS traffic and no barrier after the prologue.  The 8 waves of a workgroup are the 8 heads of the same query rows,
// so the 80-byte head slices of a row are touched together.
// =====================================================================================================================
constexpr int XA_VP = 64;                                                  // pitch (elements) of the wave-private V tile [96][64]
__global__ __launch_bounds__(512, 1) void xattn40_kernel(const SAParams P, const int nq32, const int units_per_kvb, const int nkvb) {
    constexpr int NKS = 3, NDT = 2, D = 40, NKB = 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    bf16_t* Vs = reinterpret_cast<bf16_t*>(smem_raw) + h * (96 * XA_VP);
    const int kvb = blockIdx.x % nkvb, wg = blockIdx.x / nkvb, wgs = gridDim.x / nkvb;
    const bf16_t* kg = (const bf16_t*)P.k + (int64_t)kvb * P.kbs + (int64_t)h * D;
    const bf16_t* vg = (const bf16_t*)P.v + (int64_t)kvb * P.kbs + (int64_t)h * D;

    // ---- K fragments (A operand of S^T = K Q^T: rows = keys), straight from global memory; keys >= S_kv and d >= 40 are zero ----------
    bf16x8 kf[NKB][NKS];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int key = kb * 32 + l31, d0 = ks * 16 + half * 8;
            union { bf16x8 v; u32x4 u; } t;
            t.u = (key < P.Skv && d0 < D) ? *reinterpret_cast<const u32x4*>(kg + (int64_t)key * P.krs + d0) : u32x4{0u, 0u, 0u, 0u};
            kf[kb][ks] = t.v;
        }
    // ---- V rows of my head -> wave-private LDS [96][64] (columns 40..62 zero, column 63 one), then the V^T fragments out of it ----------
    {   // 96 rows x 5 data chunks = 480 chunks over 64 lanes: all 8 loads of a lane in flight before the first LDS write (a loop of
        // load -> store pairs was 12 dependent round trips, most of this kernel's prologue); then the constant chunks 5..7
        u32x4 w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = lane + 64 * j, row = c / 5, ch = c - row * 5;
            w[j] = (c < 480 && row < P.Skv) ? *reinterpret_cast<const u32x4*>(vg + (int64_t)row * P.krs + ch * 8) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = lane + 64 * j, row = c / 5, ch = c - row * 5;
            if (c < 480) *reinterpret_cast<u32x4*>(Vs + row * XA_VP + ch * 8) = w[j];
        }
        for (int c = lane; c < 96 * 3; c += 64) {
            const int row = c / 3, ch = 5 + (c - row * 3);
            u32x4 z = u32x4{0u, 0u, 0u, 0u};
            if (ch == 7) z[3] = (unsigned)f2bf(1.f) << 16;
            *reinterpret_cast<u32x4*>(Vs + row * XA_VP + ch * 8) = z;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (wave-private: no barrier)
    bf16x8 vf[NKB][2][NDT];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const bf16_t* vp = Vs + (kb * 32 + s2 * 16 + half * 4 + ((l31 & 15) >> 2)) * XA_VP + dt * 32 + (l31 >> 4) * 16 + (l31 & 3) * 4;
                union { bf16x8 v; sa_s4 hh[2]; } r;
                r.hh[0] = lds_tr16(vp);
                r.hh[1] = lds_tr16(vp + 8 * XA_VP);
                vf[kb][s2][dt] = r.v;
            }

    auto load_q = [&](int u, bf16x8 (&qf)[NKS]) {
        const int b = kvb * P.kv_batch_div + u / nq32, q = (u - (u / nq32) * nq32) * 32 + l31;
        const bf16_t* qg = (const bf16_t*)P.q + (int64_t)b * P.qbs + (int64_t)q * P.qrs + h * D;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d0 = ks * 16 + half * 8;
            Frag<bf16_t> f;
            if (d0 < D) {
                float qv[8];
                Vec8<bf16_t>::load(qg + d0, qv);
#pragma unroll
                for (int i = 0; i < 8; ++i) qv[i] *= P.scale_log2;
                p_frag(qv, f);
            } else {
                zero(f);
            }
            qf[ks] = f.hi;
        }
    };

    int u = wg;
    bf16x8 qf[NKS], qn[NKS];
    if (u < units_per_kvb) load_q(u, qf);
    for (; u < units_per_kvb; u += wgs) {
        if (u + wgs < units_per_kvb) load_q(u + wgs, qn);                      // the next unit's rows are on their way under this unit's work
        f32x16 s[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][ks], qf[ks], s[kb], 0, 0, 0);
        }
        // keys >= S_kv out; exact row maximum (a query's 96 scores: 48 here, 48 in lane ^ 32)
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (key >= P.Skv) s[kb][r] = -INFINITY;
                mx = fmaxf(mx, s[kb][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        f32x16 oacc[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_exp2f(s[kb][r] - mx);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float p8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) p8[i] = pv[s2 * 8 + i];
                Frag<bf16_t> pf;
                p_frag(p8, pf);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kb][s2][dt], pf.hi, oacc[dt], 0, 0, 0);
            }
        }
        const float l_tot = __shfl(oacc[NDT - 1][15], l31 + 32, 64);             // O^T row 63 (the ones column of V): upper half, register 15
        const float inv = 1.f / l_tot;
        const int b = kvb * P.kv_batch_div + u / nq32, q = (u - (u / nq32) * nq32) * 32 + l31;
        bf16_t* orow = (bf16_t*)P.o + (int64_t)b * P.obs + (int64_t)q * P.ors + h * D;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * half;
                if (d < D)
                    store4<bf16_t>(orow + d, oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
            }
        if (P.lse && half == 0) P.lse[((int64_t)b * P.H + h) * P.Sq + q] = mx * 0.6931471805599453f + logf(l_tot);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = qn[ks];
    }
}

inline int sa_xattn_env() {           // FMC_SA_XATTN=0: the block-by-block kernel for the text cross-attention too (A/B)
    static const int v = [] {
        const char* e = getenv("FMC_SA_XATTN");
        return e ? atoi(e) : 1;
    }();
    return v;
}
inline bool xattn40_ok(const SAParams& P) {
    return P.D == 40 && P.H == 8 && P.Skv <= 96 && P.Skv >= 1 && P.Sq % 32 == 0 && P.B % P.kv_batch_div == 0 && sa_xattn_env() != 0;
}
inline void launch_xattn40(const SAParams& P, hipStream_t st) {
    const int nkvb = P.B / P.kv_batch_div, nq32 = P.Sq / 32, units = P.kv_batch_div * nq32;
    static const int cus = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    int per = cus / nkvb;
    if (per < 1) per = 1;
    if (per > units) per = units;
    const size_t lds = (size_t)8 * 96 * XA_VP * 2;
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn40_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        raised = true;
    }
    hipLaunchKernelGGL(xattn40_kernel, dim3((unsigned)(per * nkvb)), dim3(512), lds, st, P, nq32, units, nkvb);
}

// =====================================================================================================================
// Whole-K/V kernel for the inner levels (bf16, D = 160, S_kv <= 160): the self-attention of the 10x16 and 5x8 feature
// maps (S = 160 / 40) and the text cross-attention there (S_kv = 77).
//
// At these sizes a launch of the tiled kernel above is one wave of workgroups whose time is a chain of round trips
// (Q, tile 0 -> LDS -> barrier, tile 1 -> ..., three tiles at S = 160: 32 us for 4.2 GF) with every second workgroup
// of S = 160 holding one busy wave (128 + 32 query rows).  Here a workgroup = 5 waves = 160 query rows takes ALL the keys
// of its (batch, head) into LDS in one round trip (K row-major, V row-major read through ds_read_b64_tr_b16 as above;
// 105 KiB at 160 keys), keeps the whole score row in registers (5 x 16 per lane) and does the softmax exactly: the
// row maximum is known before the first exp, so there is no reference, no rescale and no redo pass.  One barrier.
// Same fragment mapping and the same rounding points as the tiled kernel (p rounded to bf16 for the PV product, the
// denominator summed in fp32 from the unrounded p: the even-NKS rule above).
template <int NB>   // resident 32-key blocks: 3 (S_kv <= 96) or 5 (S_kv <= 160)
__global__ __launch_bounds__(320, 1) void sa_small160_kernel(const SAParams P) {
    typedef bf16_t T;
    constexpr int NKS = 10, NDT = 5, KP = NKS * 16 + 8, VPR = sa_vr_pitch<NDT>(), CH = 20, NKEY = NB * 32, NT = 320;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Ks = reinterpret_cast<T*>(smem_raw);              // [NKEY][KP]
    T* Vr = Ks + NKEY * KP;                              // [NKEY][VPR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.x / P.nqblk, qblk = blockIdx.x - bh * P.nqblk;
    const int b = bh / P.H, h = bh - b * P.H;
    const T* qg = (const T*)P.q + (int64_t)b * P.qbs + (int64_t)h * 160;
    const T* kg = (const T*)P.k + (int64_t)(b / P.kv_batch_div) * P.kbs + (int64_t)h * 160;
    const T* vg = (const T*)P.v + (int64_t)(b / P.kv_batch_div) * P.kbs + (int64_t)h * 160;
    T* og = (T*)P.o + (int64_t)b * P.obs + (int64_t)h * 160;

    constexpr int NSLOT = NKEY * CH / NT;                // 16-byte chunks per thread and operand: 10 / 6
    static_assert(NKEY * CH % NT == 0, "whole chunks per thread");
    // request order K, Q, V: the loads return in order, so K is in LDS and the scores are under way while V still travels
    // (a launch moves its ~40 MB of q | k | v in one burst: the V half of it now hides under the QK^T products)
    u32x4 kreg[NSLOT], vreg[NSLOT];
    auto gload = [&](const T* base, u32x4 (&reg)[NSLOT]) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const int c = tid + j * NT, row = c / CH, ch = c - row * CH;
            // rows past S_kv repeat the last key (no branch, no select behind the load): their scores are masked to -inf
            // below and their p = 0 multiplies a finite V row
            reg[j] = *reinterpret_cast<const u32x4*>(base + (int64_t)(row < P.Skv ? row : P.Skv - 1) * P.krs + ch * 8);
        }
    };
    auto lstore = [&](T* dst, int pitch, const u32x4 (&reg)[NSLOT]) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const int c = tid + j * NT, row = c / CH, ch = c - row * CH;
            *reinterpret_cast<u32x4*>(dst + row * pitch + ch * 8) = reg[j];
        }
    };
    gload(kg, kreg);
    const int qrow = (qblk * 5 + wave) * 32 + l31;
    const bool active = (qblk * 5 + wave) * 32 < P.Sq;   // wave-uniform; idle waves still stage and meet the barriers
    // the Q row is loaded unconditionally (clamped): a load inside a branch is followed by its own vmcnt(0), ten dependent
    // round trips instead of one
    u32x4 qraw[NKS];
    {
        const T* qp = qg + (int64_t)(qrow < P.Sq ? qrow : P.Sq - 1) * P.qrs + half * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qraw[ks] = *reinterpret_cast<const u32x4*>(qp + ks * 16);
    }
    gload(vg, vreg);
    lstore(Ks, KP, kreg);
    Frag<T> qf[1][NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        float qv[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qv[2 * i] = __uint_as_float(qraw[ks][i] << 16) * P.scale_log2;
            qv[2 * i + 1] = __uint_as_float(qraw[ks][i] & 0xffff0000u) * P.scale_log2;
        }
        p_frag(qv, qf[0][ks]);
    }
    __syncthreads();

    // ---- scores of the whole row, exact maximum, p = exp2(s - m) ---------------------------------------------------
    Frag<T> pf[NB][2];
    float m = 0.f, l_run = 0.f;
    if (active) {
        f32x16 s[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)                 // k-step outermost: NB independent accumulators in flight
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                Frag<T> kf;
                make_frag<T>(Ks + (blk * 32 + l31) * KP + ks * 16 + half * 8, kf);
                mma32(kf, qf[0][ks], s[blk]);
            }
        float mx = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            if ((blk + 1) * 32 > P.Skv) {                // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= P.Skv) s[blk][r] = -INFINITY;
            }
            mx = fold_max(s[blk], mx);
        }
        m = fmaxf(mx, __shfl_xor(mx, 32, 64));           // finite: key 0 always exists
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(s[blk][r] - m);
                l_run += p[r];
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float p8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) p8[i] = p[s2 * 8 + i];
                p_frag(p8, pf[blk][s2]);
            }
        }
    }
    lstore(Vr, VPR, vreg);
    __syncthreads();
    if (!active) return;

    // ---- O^T += V^T P^T -------------------------------------------------------------------------------------------
    f32x16 oacc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const T* vp = Vr + (blk * 32 + s2 * 16 + half * 4 + ((l31 & 15) >> 2)) * VPR + dt * 32 + (l31 >> 4) * 16 + (l31 & 3) * 4;
                union { bf16x8 v; sa_s4 hh[2]; } rr;
                rr.hh[0] = lds_tr16(vp);
                rr.hh[1] = lds_tr16(vp + 8 * VPR);
                Frag<T> vf;
                vf.hi = rr.v;
                mma32(vf, pf[blk][s2], oacc[dt]);
            }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qrow < P.Sq) {
        T* orow = og + (int64_t)qrow * P.ors;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                store4<T>(orow + dt * 32 + 8 * g + 4 * half, oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv,
                          oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
        if (P.lse && half == 0) P.lse[((int64_t)b * P.H + h) * P.Sq + qrow] = m * 0.6931471805599453f + logf(l_tot);
    }
}

// =====================================================================================================================
// Big-tile kernel for the 20x32 level (bf16, D = 80, S_kv % 32 == 0): the self-attention of the 640-pixel feature maps.
//
// The tiled kernel above moves K / V in 64-key tiles with ONE tile of register prefetch; at d = 80 a tile is ~1400 cycles
// of work per wave and the prefetch round trip is longer, so every one of the 10 tiles of S = 640 waits for its operands
// (68 us for 33.5 GF = 0.20 of peak, 1280 workgroups of 4 waves).  Here a workgroup = 10 waves = 320 query rows and a
// tile = 320 keys = 155 KiB of LDS (K row-major 88-element pitch, V row-major 160-element pitch read through
// ds_read_b64_tr_b16: the layouts of the tiled kernel, so its block functions are used unchanged): S = 640 is TWO round
// trips per workgroup instead of ten, each a single burst of clamped, unconditional loads.  Inside a tile the ten 32-key
// blocks run back to back with no barrier -- QK^T of block b+1 is issued before the softmax of block b -- on the
// fixed-reference softmax of the tiled kernel (reference = row maximum over the first 32 keys, redo pass with the exact
// maxima if a row leaves the safe range).
template <int NW>      // waves per workgroup = 32-query blocks: 10 (168 VGPRs: block loop rolled) | 8 (256 VGPRs: unrolled, QK^T of block b+1 before the softmax of b)
                       // | 5: 160 rows on 160-key tiles (77.5 KiB): TWO workgroups per CU, one staging while the other computes
__device__ __forceinline__ void sa_big80_body(const SAParams& P) {
    typedef bf16_t T;
    constexpr int NKS = 5, NDT = 3, KP = NKS * 16 + 8, VPR = sa_vr_pitch<NDT>(), CH = 10, BKT = NW == 5 ? 160 : 320, NBLK = BKT / 32, NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Ks = reinterpret_cast<T*>(smem_raw);              // [BKT][KP]
    T* Vr = Ks + BKT * KP;                               // [BKT][VPR] (columns 80..95 are read by the padded PV product: zeroed once)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int bh, qblk;
    {   // all query blocks AND heads of a batch entry on one XCD (the tiled kernel's map 2: a head's 160-byte row slices share lines with its neighbours)
        const int id = blockIdx.x;
        if (P.xcd_remap == 2) {
            const int xcd = id & 7, within = id >> 3, per_b = P.H * P.nqblk, rem = within % per_b;
            bh = ((within / per_b) * 8 + xcd) * P.H + rem / P.nqblk;
            qblk = rem % P.nqblk;
        } else {
            bh = id / P.nqblk;
            qblk = id - bh * P.nqblk;
        }
    }
    const int b = bh / P.H, h = bh - b * P.H;
    const T* qg = (const T*)P.q + (int64_t)b * P.qbs + (int64_t)h * 80;
    const T* kg = (const T*)P.k + (int64_t)(b / P.kv_batch_div) * P.kbs + (int64_t)h * 80;
    const T* vg = (const T*)P.v + (int64_t)(b / P.kv_batch_div) * P.kbs + (int64_t)h * 80;
    T* og = (T*)P.o + (int64_t)b * P.obs + (int64_t)h * 80;

    constexpr int NSLOT = (BKT * CH + NT - 1) / NT;      // 5 (6.25 -> 7 at 8 waves) sixteen-byte chunks per thread and operand
    u32x4 kreg[NSLOT], vreg[NSLOT];
    auto gload = [&](int kv0) {                          // rows past S_kv repeat the last key: their blocks are never visited (S_kv % 32 == 0)
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const int c = min(tid + j * NT, BKT * CH - 1), row = c / CH, ch = c - row * CH;      // (the last slot of the 8-wave form repeats the last chunk)
            const int kv = kv0 + row < P.Skv ? kv0 + row : P.Skv - 1;
            kreg[j] = *reinterpret_cast<const u32x4*>(kg + (int64_t)kv * P.krs + ch * 8);
            vreg[j] = *reinterpret_cast<const u32x4*>(vg + (int64_t)kv * P.krs + ch * 8);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const int c = min(tid + j * NT, BKT * CH - 1), row = c / CH, ch = c - row * CH;
            *reinterpret_cast<u32x4*>(Ks + row * KP + ch * 8) = kreg[j];
            *reinterpret_cast<u32x4*>(Vr + row * VPR + ch * 8) = vreg[j];
        }
    };
    gload(0);
    const int qrow = (qblk * NW + wave) * 32 + l31;
    const bool active = (qblk * NW + wave) * 32 < P.Sq;  // wave-uniform; idle waves still stage and meet the barriers
    Frag<T> qf[1][NKS];
    {
        const T* qp = qg + (int64_t)(qrow < P.Sq ? qrow : P.Sq - 1) * P.qrs + half * 8;
        u32x4 qraw[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qraw[ks] = *reinterpret_cast<const u32x4*>(qp + ks * 16);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            float qv[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                qv[2 * i] = __uint_as_float(qraw[ks][i] << 16) * P.scale_log2;
                qv[2 * i + 1] = __uint_as_float(qraw[ks][i] & 0xffff0000u) * P.scale_log2;
            }
            p_frag(qv, qf[0][ks]);
        }
    }
    for (int c = tid; c < BKT * 2; c += NT)              // V columns 80..94: zeros; column 95: ones -- O^T row 95 accumulates l = sum(p) in the PV
        *reinterpret_cast<u32x4*>(Vr + (c >> 1) * VPR + 80 + (c & 1) * 8) = u32x4{0u, 0u, 0u, (c & 1) ? 0x3f800000u : 0u};      // products (the tiled kernel's LROW)
    lstore();
    __syncthreads();

    const int ntiles = (P.Skv + BKT - 1) / BKT;
    // reference: row maximum over the first 32 keys (in LDS now)
    float m_ref[1];
    f32x16 negm[1];
    {
        f32x16 zero16;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
        const float mx = fold_max(qk_block<T, NKS, false>(Ks, qf[0], zero16, 0, 0, P.Skv, l31, half), -INFINITY);
        m_ref[0] = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[0][r] = -m_ref[0];
    }
    f32x16 oacc[1][NDT];
    float l_run[1];
    for (int pass = 0; pass < 2; ++pass) {
        float worst[1] = {-INFINITY};
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[0][dt][r] = 0.f;
        l_run[0] = 0.f;
        for (int tile = 0; tile < ntiles; ++tile) {
#ifndef SA_BIG_DBG
#define SA_BIG_DBG 0
#endif
            if ((tile > 0 || pass > 0) && SA_BIG_DBG != 2) {                  // tile 0 of the first pass is resident
                gload(tile * BKT);
                __syncthreads();                         // the previous tile is consumed
                lstore();
                __syncthreads();
            }
            if (active && SA_BIG_DBG != 1) {
                const int nblk = min(NBLK, (P.Skv - tile * BKT) >> 5);       // block-uniform
                if (NW == 8 && nblk == NBLK) {
                    f32x16 s[2][1];
                    qk_scores<T, NKS, 1, false>(Ks, qf, negm, s[0], 0, l31, half);
#pragma unroll
                    for (int sb = 0; sb < NBLK; ++sb) {
                        if (sb + 1 < NBLK) qk_scores<T, NKS, 1, false>(Ks, qf, negm, s[(sb + 1) & 1], sb + 1, l31, half);
                        softmax_pv_perq<T, NKS, 1, true>(Vr, s[sb & 1], oacc, worst, l_run, sb, l31, half);
                    }
                } else {
#pragma unroll 1
                    for (int sb = 0; sb < nblk; ++sb)
                        attn_block<T, NKS, 1, false, false, true>(Ks, Vr, qf, oacc, negm, worst, l_run, sb, tile * BKT + sb * 32, P.Skv, l31, half);
                }
            }
        }
        const bool bad = active && worst[0] > REF_LIMIT;
        if (!__syncthreads_or(bad)) break;
        m_ref[0] += fmaxf(worst[0], __shfl_xor(worst[0], 32, 64));          // now the exact row maximum
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[0][r] = -m_ref[0];
    }
    if (!active) return;
    const float l_tot = __shfl(oacc[0][NDT - 1][15], l31 + 32, 64);          // row 95 lives in the upper half's register 15
    const float inv = 1.f / l_tot;
    if (qrow < P.Sq) {
        T* orow = og + (int64_t)qrow * P.ors;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * half;
                if (d < 80)
                    store4<T>(orow + d, oacc[0][dt][4 * g] * inv, oacc[0][dt][4 * g + 1] * inv, oacc[0][dt][4 * g + 2] * inv, oacc[0][dt][4 * g + 3] * inv);
            }
        if (P.lse && half == 0) P.lse[((int64_t)b * P.H + h) * P.Sq + qrow] = m_ref[0] * 0.6931471805599453f + logf(l_tot);
    }
}

// one thin kernel per form (HIP: the second launch bound is the minimum number of WAVES PER SIMD; 10 waves per CU = three on one SIMD = 168 VGPRs)
__global__ __launch_bounds__(640, 1) void sa_big80_kernel_w10(const SAParams P) { sa_big80_body<10>(P); }
__global__ __launch_bounds__(512, 1) void sa_big80_kernel_w8(const SAParams P) { sa_big80_body<8>(P); }
__global__ __launch_bounds__(320, 3) void sa_big80_kernel_w5(const SAParams P) { sa_big80_body<5>(P); }

inline bool sa_big80_ok(const SAParams& P) {     // FMC_SA_BIG80=0: the tiled kernel at the 20x32 level too (A/B)
    static const bool on = [] {
        const char* e = getenv("FMC_SA_BIG80");
        return !e || atoi(e) != 0;
    }();
    return on && P.D == 80 && P.Skv % 32 == 0 && P.Skv >= 256 && P.Sq >= 160;
}
inline void launch_sa_big80(const SAParams& Pin, hipStream_t st) {
    SAParams P = Pin;
    static const int nw = [] { const char* e = getenv("FMC_SA_BIG80_WAVES"); const int v = e ? atoi(e) : 10; return v == 8 || v == 5 ? v : 10; }();
    P.nqblk = (P.Sq + 32 * nw - 1) / (32 * nw);
    if (P.xcd_remap != 2) P.xcd_remap = 0;
    const size_t lds = (size_t)(nw == 5 ? 160 : 320) * (5 * 16 + 8 + sa_vr_pitch<3>()) * 2;
    static FmcPerDeviceFlag raised;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sa_big80_kernel_w10), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sa_big80_kernel_w8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sa_big80_kernel_w5), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        raised = true;
    }
    if (nw == 5) hipLaunchKernelGGL(sa_big80_kernel_w5, dim3((unsigned)(P.B * P.H * P.nqblk)), dim3(320), lds, st, P);
    else if (nw == 8) hipLaunchKernelGGL(sa_big80_kernel_w8, dim3((unsigned)(P.B * P.H * P.nqblk)), dim3(512), lds, st, P);
    else hipLaunchKernelGGL(sa_big80_kernel_w10, dim3((unsigned)(P.B * P.H * P.nqblk)), dim3(640), lds, st, P);
}

inline bool sa_small_ok(const SAParams& P) {     // FMC_SA_SMALL=0: the tiled kernel at the inner levels too (A/B)
    static const bool on = [] {
        const char* e = getenv("FMC_SA_SMALL");
        return !e || atoi(e) != 0;
    }();
    return on && P.D == 160 && P.Skv <= 160;
}
template <int NB>
void launch_sa_small_nb(const SAParams& P, hipStream_t st) {
    const size_t lds = (size_t)NB * 32 * (10 * 16 + 8 + sa_vr_pitch<5>()) * 2;
    if (lds > 64 * 1024) {
        static FmcPerDeviceFlag raised;
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sa_small160_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            raised = true;
        }
    }
    hipLaunchKernelGGL((sa_small160_kernel<NB>), dim3((unsigned)(P.B * P.H * P.nqblk)), dim3(320), lds, st, P);
}
inline void launch_sa_small(const SAParams& Pin, hipStream_t st) {
    SAParams P = Pin;
    P.nqblk = (P.Sq + 159) / 160;
    if (P.Skv <= 96) launch_sa_small_nb<3>(P, st);
    else launch_sa_small_nb<5>(P, st);
}

inline int sa_pipe_env() {            // FMC_SA_PIPE=0: the block-by-block kernel at d = 40 too (A/B); 4: persistent workgroups
    static const int v = [] {
        const char* e = getenv("FMC_SA_PIPE");
        return e ? atoi(e) : 1;
    }();
    return v;
}
inline bool sa40_ok(const SAParams& P) {
    return P.D == 40 && P.Skv % SA_BK == 0 && P.Skv >= 2 * SA_BK && P.Sq % 256 == 0 && P.krs * 2 * (int64_t)P.Skv < (1ll << 31) &&
           sa_pipe_env() != 0;
}
inline void launch_sa40(const SAParams& Pin, hipStream_t st) {
    SAParams P = Pin;
    P.nqblk = P.Sq / 256;
    const int nitems = P.B * P.H * P.nqblk;
    // One workgroup per item by default: the hardware's dispatch balances the CUs inside an XCD (workgroup times spread
    // 53..78 us, the XCDs' clocks 1.48..1.61 GHz); FMC_SA_PIPE=4 runs 2 persistent workgroups per CU over a static item
    // list instead -- measured 4 % slower for that reason.
    int grid = nitems;
    if (sa_pipe_env() == 4) {
        static const int resident = [] {
            int dev = 0, cus = 256;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            return 2 * cus;
        }();
        if (nitems > resident) grid = resident;
    }
    hipLaunchKernelGGL(sa40d_kernel, dim3((unsigned)grid), dim3(256), D40_LDS, st, P, nitems);
}

template <typename T, int NKS, bool SHORT_KV, bool PF, int NQ = 1, bool MK = false, bool VR = false>
void launch_sa_v(const SAParams& Pin, hipStream_t st) {
    SAParams P = Pin;
    P.nqblk = (P.Sq + SA_BQ * NQ - 1) / (SA_BQ * NQ);
    constexpr int NDT = (NKS + 1) / 2;
    const size_t lds = sizeof(T) * ((size_t)SA_BK * (NKS * 16 + 8) + (VR ? (size_t)SA_BK * sa_vr_pitch<NDT>() : (size_t)NDT * 32 * (SA_BK + 4)));
    dim3 grid((unsigned)(P.B * P.H * P.nqblk)), block(64 * SA_WAVES);
    if (lds > 64 * 1024) {  // gfx950 has 160 KiB of LDS per CU; opting in is needed above 64 KiB
        static FmcPerDeviceFlag raised;
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_attn_kernel<T, NKS, SHORT_KV, PF, NQ, MK, VR>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            raised = true;
        }
    }
    hipLaunchKernelGGL((spatial_attn_kernel<T, NKS, SHORT_KV, PF, NQ, MK, VR>), grid, block, lds, st, P);
}

// FMC_SA_PREFETCH=0/1 overrides the default policy (experiments only)
inline int sa_prefetch_env() {
    static const int v = [] {
        const char* e = getenv("FMC_SA_PREFETCH");
        return e ? atoi(e) : -1;
    }();
    return v;
}

inline int sa_nq_env() {              // FMC_SA_NQ = 1 | 2 query blocks per wave for d <= 48 (default 2)
    static const int v = [] {
        const char* e = getenv("FMC_SA_NQ");
        return e ? atoi(e) : 2;
    }();
    return v;
}

inline int sa_vr_env() {              // FMC_SA_VR=0: V transposed while staging instead of by the transpose read (A/B)
    static const int v = [] {
        const char* e = getenv("FMC_SA_VR");
        return e ? atoi(e) : 1;
    }();
    return v;
}

inline int sa_mk_env() {              // FMC_SA_MK=0: softmax reference through the accumulator instead of the QK^T reduction
    static const int v = [] {               // (the reduction form frees 33 VGPRs: -2 % on top of the transpose-read V path)
        const char* e = getenv("FMC_SA_MK");
        return e ? atoi(e) : 1;
    }();
    return v;
}

template <typename T, int NKS>
void launch_sa(const SAParams& P, hipStream_t st) {
    if (P.Skv <= 2 * SA_BK) {
        if constexpr (sizeof(T) == 2 && NKS <= 6) {
            if (sa_prefetch_env() != 0 && P.Skv > SA_BK) {
                if constexpr (NKS <= 3) {
                    if (sa_nq_env() == 2 && P.Sq >= 2 * SA_BQ) {
                        if (sa_vr_env()) {
                            if (P.D % 16 == 8 && sa_mk_env()) return launch_sa_v<T, NKS, true, true, 2, true, true>(P, st);
                            return launch_sa_v<T, NKS, true, true, 2, false, true>(P, st);
                        }
                        return launch_sa_v<T, NKS, true, true, 2>(P, st);
                    }
                }
                return launch_sa_v<T, NKS, true, true>(P, st);
            }
        }
        launch_sa_v<T, NKS, true, false>(P, st);
    } else {
        if constexpr (sizeof(T) == 2 && NKS <= 6) {
            const int e = sa_prefetch_env();
            if (e < 0 ? SA_PREFETCH_DEFAULT : e != 0) {
                if constexpr (NKS <= 3) {
                    if (sa_nq_env() == 2 && P.Sq >= 2 * SA_BQ) {
                        if (sa_vr_env()) {
                            if (P.D % 16 == 8 && sa_mk_env()) return launch_sa_v<T, NKS, false, true, 2, true, true>(P, st);
                            return launch_sa_v<T, NKS, false, true, 2, false, true>(P, st);
                        }
                        return launch_sa_v<T, NKS, false, true, 2>(P, st);
                    }
                }
                return launch_sa_v<T, NKS, false, true>(P, st);
            }
        }
        launch_sa_v<T, NKS, false, false>(P, st);
    }
}

template <typename T>
int dispatch_sa(const SAParams& P, hipStream_t st) {
    switch ((P.D + 15) / 16) {
        case 1: launch_sa<T, 1>(P, st); break;
        case 2: launch_sa<T, 2>(P, st); break;
        case 3: launch_sa<T, 3>(P, st); break;
        case 4: launch_sa<T, 4>(P, st); break;
        case 5: launch_sa<T, 5>(P, st); break;
        case 6: launch_sa<T, 6>(P, st); break;
        case 8: launch_sa<T, 8>(P, st); break;
        case 10: launch_sa<T, 10>(P, st); break;
        default: FMC_FAIL(FMC_E_SHAPE, "spatial_attn: head dim %d not built (supported: <=96, 113..128, 145..160)", P.D);
    }
    return 0;
}

}  // namespace

extern "C" int fmc_spatial_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H,
                                    int Sq, int Skv, int D, int64_t q_batch_stride, int64_t q_row_stride,
                                    int64_t kv_batch_stride, int64_t kv_row_stride, int64_t o_batch_stride,
                                    int64_t o_row_stride, int kv_batch_div, float scale, int dtype, void* stream) {
    if (!q || !k || !v || !o) FMC_FAIL(FMC_E_NULL, "spatial_attn: NULL tensor");
    if (dtype != FMC_BF16 && dtype != FMC_F32) FMC_FAIL(FMC_E_DTYPE, "spatial_attn: dtype %d", dtype);
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0 || D <= 0 || D % 8 || D > 160 || kv_batch_div <= 0)
        FMC_FAIL(FMC_E_SHAPE, "spatial_attn: need D%%8==0, D<=160, positive sizes (B=%d H=%d Sq=%d Skv=%d D=%d)", B, H, Sq,
                 Skv, D);
    const int64_t strides[] = {q_batch_stride, q_row_stride, kv_batch_stride, kv_row_stride, o_batch_stride, o_row_stride};
    for (int64_t s : strides)
        if (s % 8) FMC_FAIL(FMC_E_ALIGN, "spatial_attn: strides must be multiples of 8 elements");
    if (!fmc_aligned16(q) || !fmc_aligned16(k) || !fmc_aligned16(v) || !fmc_aligned16(o))
        FMC_FAIL(FMC_E_ALIGN, "spatial_attn: tensors must be 16-byte aligned");
    SAParams P;
    P.q = q; P.k = k; P.v = v; P.o = o; P.lse = lse;
    P.B = B; P.H = H; P.Sq = Sq; P.Skv = Skv; P.D = D;
    P.qbs = q_batch_stride; P.qrs = q_row_stride; P.kbs = kv_batch_stride; P.krs = kv_row_stride;
    P.obs = o_batch_stride; P.ors = o_row_stride;
    P.kv_batch_div = kv_batch_div;
    P.scale_log2 = scale * LOG2E;
    P.nqblk = (Sq + SA_BQ - 1) / SA_BQ;
    P.xcd_remap = (B % 8 == 0) ? 2 : ((B * H) % 8 == 0) ? 1 : 0;
    if (const char* e = getenv("FMC_SA_XCD")) {          // A/B switch: 0 = plain order, 1 = (batch, head) per XCD
        const int want = atoi(e);
        if (want == 0 || (want == 1 && (B * H) % 8 == 0)) P.xcd_remap = want;
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FMC_BF16 && xattn40_ok(P)) {
        launch_xattn40(P, st);
        FMC_CHECK_LAUNCH("fmc_spatial_attn_fwd");
        return 0;
    }
    if (dtype == FMC_BF16 && sa_big80_ok(P)) {
        launch_sa_big80(P, st);
        FMC_CHECK_LAUNCH("fmc_spatial_attn_fwd");
        return 0;
    }
    if (dtype == FMC_BF16 && sa_small_ok(P)) {
        launch_sa_small(P, st);
        FMC_CHECK_LAUNCH("fmc_spatial_attn_fwd");
        return 0;
    }
    if (dtype == FMC_BF16 && sa40_ok(P)) {
        launch_sa40(P, st);
        FMC_CHECK_LAUNCH("fmc_spatial_attn_fwd");
        return 0;
    }
    int rc = (dtype == FMC_BF16) ? dispatch_sa<bf16_t>(P, st) : dispatch_sa<float>(P, st);
    if (rc) return rc;
    FMC_CHECK_LAUNCH("fmc_spatial_attn_fwd");
    return 0;
}
