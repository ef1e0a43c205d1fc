// Multi-head attention for the steps either side of the denoising loop, gfx950: the shapes the hot-path kernels of
// spatial_attn.hip do not take -- head widths up to 512 and masked scores.
//
//   * the VAE mid block's single-head self-attention, d = 512 (diffusers `Attention(512, heads=1, dim_head=512)` inside AutoencoderKL's UNetMidBlock2D;
//     reached from `decode_latents` / `vae.encode`, fmc/pipelines/pipeline_animation_cm_om.py:465-478, train_cam_obj_ctrl.py:786);
//   * CLIP's text self-attention, 12 heads x 64 over 77 tokens with the causal mask (and, for `use_attention_mask` configurations, a key-padding mask):
//     `_encode_prompt`, pipeline_animation_cm_om.py:480-568 -> transformers CLIPAttention.
//
// Both run once per clip, outside the metric: the kernel is written for generality (any S, any d = 32 k up to 512, causal / key mask, bf16 and the
// fp32-storage parity mode), not for the roofline.  Flash-style, one pass over the keys:
//   workgroup = 4 waves = 64 queries of one (batch, head); a wave owns 16 queries, its Q fragments and O^T accumulators live in registers;
//   a 32-key stage of K ([key][d], row pitch d + 8) and V^T ([d][key], pitch 40) is staged in LDS by all four waves;
//   S^T[16 keys x 16 queries] = K Q^T on v_mfma_f32_16x16x32_bf16 -- the accumulator layout (lane = query, 4 consecutive keys) IS the B-operand layout of
//   v_mfma_f32_16x16x16_bf16, so P^T feeds O^T[d x 16 queries] += V^T P^T from registers (the scheme of temporal_attn.hip); online softmax in the exp2
//   domain, row statistics replicated over the four lanes of a query.
// fp32 storage: every product as split-bf16 x3 (hi hi + hi lo + lo hi, fp32 accumulate), like the other attention kernels.
#include "common.h"

namespace {

struct AGParams {
    const void* q; const void* k; const void* v; void* o;
    const unsigned char* key_keep;          // [B][Skv] 1 = attend, 0 = masked; or NULL
    int B, H, Sq, Skv;
    int64_t qbs, qrs, kbs, krs, obs, ors;   // element strides: batch, row (heads are D apart inside a row)
    float scale_log2;
    int causal;
};

constexpr int AG_KB = 32;                   // keys per stage
constexpr int AG_VP = AG_KB + 8;            // V^T row pitch (elements)

template <typename T> struct AGFrag8;       // 8 k-values of a 16x16x32 operand
template <> struct AGFrag8<bf16_t> { bf16x8 hi; };
template <> struct AGFrag8<float> { bf16x8 hi, lo; };
template <typename T> struct AGFrag4;       // 4 k-values of a 16x16x16 operand
template <> struct AGFrag4<bf16_t> { s16x4 hi; };
template <> struct AGFrag4<float> { s16x4 hi, lo; };

__device__ __forceinline__ void ag_mma32(const AGFrag8<bf16_t>& a, const AGFrag8<bf16_t>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, acc, 0, 0, 0);
}
__device__ __forceinline__ void ag_mma32(const AGFrag8<float>& a, const AGFrag8<float>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, acc, 0, 0, 0);
}
__device__ __forceinline__ void ag_mma16(const AGFrag4<bf16_t>& a, const AGFrag4<bf16_t>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.hi, b.hi, acc, 0, 0, 0);
}
__device__ __forceinline__ void ag_mma16(const AGFrag4<float>& a, const AGFrag4<float>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.hi, b.hi, acc, 0, 0, 0);
}

// 8 consecutive elements of a global row -> 8 floats (zeros when `ok` is false)
template <typename T> __device__ __forceinline__ void ag_load8(const T* p, bool ok, float (&v)[8]) {
    if (ok) {
        Vec8<T>::load(p, v);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
}

// LDS planes: T = bf16_t keeps one bf16 plane, T = float a hi and a lo plane (the split happens once, while staging)
template <typename T> struct AGPlanes { static constexpr int N = sizeof(T) == 2 ? 1 : 2; };

template <typename T, int NKS>
__global__ __launch_bounds__(256) void attn_generic_kernel(const AGParams P) {
    constexpr int D = 32 * NKS, KP = D + 8, NPL = AGPlanes<T>::N, NDT = D / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);              // [NPL][AG_KB][KP]
    bf16_t* Vt = Ks + NPL * AG_KB * KP;                            // [NPL][D][AG_VP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int bh = blockIdx.y, b = bh / P.H, h = bh - b * P.H;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const T* qp = static_cast<const T*>(P.q) + b * P.qbs + h * D;
    const T* kp = static_cast<const T*>(P.k) + b * P.kbs + h * D;
    const T* vp = static_cast<const T*>(P.v) + b * P.kbs + h * D;
    const unsigned char* keep = P.key_keep ? P.key_keep + (int64_t)b * P.Skv : nullptr;

    // ---- Q fragments: lane (l15 = query, kq) holds d = 32 ks + 8 kq .. + 7 ----
    AGFrag8<T> qf[NKS];
    const int qrow = q0 + l15;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        float v8[8];
        ag_load8<T>(qp + (int64_t)qrow * P.qrs + 32 * ks + 8 * kq, qrow < P.Sq, v8);
        if constexpr (sizeof(T) == 2) {
            union { bf16x8 v; unsigned u[4]; } r;
#pragma unroll
            for (int i = 0; i < 4; ++i) r.u[i] = pack_bf2(v8[2 * i], v8[2 * i + 1]);
            qf[ks].hi = r.v;
        } else {
            split_bf16x8(v8, qf[ks].hi, qf[ks].lo);
        }
    }

    f32x4 oacc[NDT];
#pragma unroll
    for (int t = 0; t < NDT; ++t) oacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_part = 0.f;       // running maximum of my query (replicated over kq); MY lanes' share of the denominator

    // causal: keys beyond the last query of the workgroup never contribute
    const int kv_end = P.causal ? min(P.Skv, blockIdx.x * 64 + 64) : P.Skv;
    for (int kb = 0; kb < kv_end; kb += AG_KB) {
        __syncthreads();                          // the previous stage's fragments are consumed
        // ---- stage K [key][d] and V^T [d][key]: 8-element pieces, zero beyond Skv ----
        for (int pc = tid; pc < AG_KB * (D / 8); pc += 256) {
            const int key = pc / (D / 8), c = pc - key * (D / 8);
            const bool ok = kb + key < P.Skv;
            float k8[8], v8[8];
            ag_load8<T>(kp + (int64_t)(kb + key) * P.krs + 8 * c, ok, k8);
            ag_load8<T>(vp + (int64_t)(kb + key) * P.krs + 8 * c, ok, v8);
            union { bf16x8 v; u32x4 u; bf16_t s[8]; } kh, kl, vh, vl;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { kh.s[i] = f2bf(k8[i]); vh.s[i] = f2bf(v8[i]); }
            } else {
                split_bf16x8(k8, kh.v, kl.v);
                split_bf16x8(v8, vh.v, vl.v);
            }
            *reinterpret_cast<u32x4*>(Ks + key * KP + 8 * c) = kh.u;
#pragma unroll
            for (int i = 0; i < 8; ++i) Vt[(8 * c + i) * AG_VP + key] = vh.s[i];
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<u32x4*>(Ks + AG_KB * KP + key * KP + 8 * c) = kl.u;
#pragma unroll
                for (int i = 0; i < 8; ++i) Vt[D * AG_VP + (8 * c + i) * AG_VP + key] = vl.s[i];
            }
        }
        __syncthreads();

        // ---- scores of my 16 queries against the stage's 2 x 16 keys: s[kt][j] = key kb + 16 kt + 4 kq + j ----
        f32x4 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                AGFrag8<T> kf;
                union { bf16x8 v; u32x4 u; } r;
                r.u = *reinterpret_cast<const u32x4*>(Ks + (16 * kt + l15) * KP + 32 * ks + 8 * kq);
                kf.hi = r.v;
                if constexpr (sizeof(T) == 4) {
                    r.u = *reinterpret_cast<const u32x4*>(Ks + AG_KB * KP + (16 * kt + l15) * KP + 32 * ks + 8 * kq);
                    kf.lo = r.v;
                }
                ag_mma32(kf, qf[ks], s[kt]);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = kb + 16 * kt + 4 * kq + j;
                bool ok = key < P.Skv && (!P.causal || key <= qrow);
                if (ok && keep) ok = keep[key] != 0;
                s[kt][j] = ok ? s[kt][j] * P.scale_log2 : -INFINITY;
                mx = fmaxf(mx, s[kt][j]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);      // (m_run = -inf: exp2(-inf) = 0)
        float p[2][4], psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p[kt][j] = (m_new == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(s[kt][j] - m_new);
                psum += p[kt][j];
            }
        l_part = l_part * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < NDT; ++t) oacc[t] *= alpha;

        // ---- O^T[d x my queries] += V^T[d x keys] P^T: A = V^T fragment (lane: d = 16 t + l15, keys 4 kq .. + 3), B = P^T from the score registers ----
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            AGFrag4<T> pf;
            if constexpr (sizeof(T) == 2) {
                union { s16x4 s; unsigned u[2]; } r;
                r.u[0] = pack_bf2(p[kt][0], p[kt][1]);
                r.u[1] = pack_bf2(p[kt][2], p[kt][3]);
                pf.hi = r.s;
            } else {
                bf16_t hh[4], ll[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { hh[i] = f2bf(p[kt][i]); ll[i] = f2bf(p[kt][i] - bf2f(hh[i])); }
                pf.hi = s16x4{(short)hh[0], (short)hh[1], (short)hh[2], (short)hh[3]};
                pf.lo = s16x4{(short)ll[0], (short)ll[1], (short)ll[2], (short)ll[3]};
            }
#pragma unroll
            for (int t = 0; t < NDT; ++t) {
                AGFrag4<T> vf;
                union { s16x4 s; u32x2 u; } r;
                r.u = *reinterpret_cast<const u32x2*>(Vt + (16 * t + l15) * AG_VP + 16 * kt + 4 * kq);
                vf.hi = r.s;
                if constexpr (sizeof(T) == 4) {
                    r.u = *reinterpret_cast<const u32x2*>(Vt + D * AG_VP + (16 * t + l15) * AG_VP + 16 * kt + 4 * kq);
                    vf.lo = r.s;
                }
                ag_mma16(vf, pf, oacc[t]);
            }
        }
    }

    // ---- normalise and store: lane (l15 = query, kq) holds d = 16 t + 4 kq .. + 3 ----
    float l = l_part;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.f / l : 0.f;                    // (a fully masked row yields zeros)
    if (qrow < P.Sq) {
        T* op = static_cast<T*>(P.o) + b * P.obs + (int64_t)qrow * P.ors + h * D;
#pragma unroll
        for (int t = 0; t < NDT; ++t) {
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<u32x2*>(op + 16 * t + 4 * kq) = u32x2{pack_bf2(oacc[t][0] * inv, oacc[t][1] * inv), pack_bf2(oacc[t][2] * inv, oacc[t][3] * inv)};
            } else {
                *reinterpret_cast<f32x4*>(op + 16 * t + 4 * kq) = f32x4{oacc[t][0] * inv, oacc[t][1] * inv, oacc[t][2] * inv, oacc[t][3] * inv};
            }
        }
    }
}

template <typename T, int NKS>
int ag_launch(const AGParams& P, hipStream_t st) {
    constexpr int D = 32 * NKS, NPL = AGPlanes<T>::N;
    constexpr int lds = NPL * (AG_KB * (D + 8) + D * AG_VP) * 2;
    static FmcPerDeviceFlag raised;
    if (!raised) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_generic_kernel<T, NKS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        raised = true;
    }
    hipLaunchKernelGGL((attn_generic_kernel<T, NKS>), dim3((unsigned)((P.Sq + 63) / 64), (unsigned)(P.B * P.H)), dim3(256), lds, st, P);
    return 0;
}

template <typename T>
int ag_dispatch(const AGParams& P, int D, hipStream_t st) {
    switch (D / 32) {
        case 1: return ag_launch<T, 1>(P, st);
        case 2: return ag_launch<T, 2>(P, st);
        case 4: return ag_launch<T, 4>(P, st);
        case 8: return ag_launch<T, 8>(P, st);
        case 16: return ag_launch<T, 16>(P, st);
        default: FMC_FAIL(FMC_E_SHAPE, "attention_fwd: head width %d (supported: 32, 64, 128, 256, 512)", D);
    }
}

}  // namespace

extern "C" int fmc_attention_supported(int D) { return D == 32 || D == 64 || D == 128 || D == 256 || D == 512; }

extern "C" int fmc_attention_fwd(const void* q, const void* k, const void* v, void* o, const unsigned char* key_keep, int B, int H, int Sq, int Skv, int D,
                                 int64_t q_batch_stride, int64_t q_row_stride, int64_t kv_batch_stride, int64_t kv_row_stride, int64_t o_batch_stride,
                                 int64_t o_row_stride, float scale, int causal, int dtype, void* stream) {
    if (!q || !k || !v || !o) FMC_FAIL(FMC_E_NULL, "attention_fwd: NULL tensor");
    if (dtype != FMC_BF16 && dtype != FMC_F32) FMC_FAIL(FMC_E_DTYPE, "attention_fwd: dtype %d", dtype);
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0 || !fmc_attention_supported(D) || (int64_t)B * H > 65535)
        FMC_FAIL(FMC_E_SHAPE, "attention_fwd: B=%d H=%d Sq=%d Skv=%d D=%d (D in {32, 64, 128, 256, 512}, B*H <= 65535)", B, H, Sq, Skv, D);
    const int64_t strides[] = {q_batch_stride, q_row_stride, kv_batch_stride, kv_row_stride, o_batch_stride, o_row_stride};
    for (int64_t s : strides)
        if (s % 8) FMC_FAIL(FMC_E_ALIGN, "attention_fwd: strides must be multiples of 8 elements");
    if (!fmc_aligned16(q) || !fmc_aligned16(k) || !fmc_aligned16(v) || !fmc_aligned16(o)) FMC_FAIL(FMC_E_ALIGN, "attention_fwd: tensors must be 16-byte aligned");
    AGParams P;
    P.q = q; P.k = k; P.v = v; P.o = o; P.key_keep = key_keep;
    P.B = B; P.H = H; P.Sq = Sq; P.Skv = Skv;
    P.qbs = q_batch_stride; P.qrs = q_row_stride; P.kbs = kv_batch_stride; P.krs = kv_row_stride; P.obs = o_batch_stride; P.ors = o_row_stride;
    P.scale_log2 = scale * 1.4426950408889634f;
    P.causal = causal != 0;
    const int rc = dtype == FMC_BF16 ? ag_dispatch<bf16_t>(P, D, (hipStream_t)stream) : ag_dispatch<float>(P, D, (hipStream_t)stream);
    if (rc) return rc;
    FMC_CHECK_LAUNCH("fmc_attention_fwd");
    return 0;
}
