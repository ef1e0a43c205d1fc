// Temporal attention for gfx950: per (pixel, head) softmax(Q K^T * scale) V over F in {16, 32} frames.
//
// Replaces head_to_batch_dim + baddbmm + softmax + bmm + batch_to_head_dim of
// fmc/models/attention_processor.py:271-281 (PoseAdaptorAttnProcessor) and :61-67 (AttnProcessor)
// when reached from the motion modules / camera encoder (fmc/models/motion_module.py:365-389).
//
// This kernel is HBM bound (arithmetic intensity F/2 flop per byte): the design goal is to move
// Q, K, V, O exactly once with full-line transactions and keep the MFMA work off the critical path.
//   * work unit = (clip, pixel, head group); a head group is GH heads = CW <= 320 contiguous
//     channels, so every frame row of a unit is one contiguous 640-byte run (bf16);
//   * one wave (= one 64-thread workgroup) per unit: all 3*F rows are fetched with 16-byte loads,
//     every one in flight before the first use, and staged in LDS [F][CW+8] (pitch is conflict free
//     for the ds_read_b128 fragment reads);
//   * strided addressing (clip / frame / pixel strides) lets the kernel read the channels-last
//     activation `[(b f), hw, c]` in place: the reference's `b c f h w <-> (b h w) f c`
//     transposes (fmc/models/motion_module.py:218,232) do not exist here;
//   * per head: S^T = K Q^T with v_mfma_f32_16x16x32_bf16 (head dim padded to 32s), softmax over
//     keys = 4 in-lane values + two cross-lane steps, P^T goes from the accumulator registers
//     straight into v_mfma_f32_16x16x16_bf16 as the B operand of O^T = V^T P^T;
//   * O is written back into the LDS slot of the Q rows it came from and leaves with the same
//     coalesced 16-byte stores;
//   * FMC_F32 storage = split-bf16 x3 products (parity mode), like the spatial kernel.
//
// Algorithmic bytes per launch = 4 * n_clips*n_pix*F*H*D * e; flops = 4 * n_clips*n_pix*H*F*F*D.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

struct TAParams {
    const void* q; const void* k; const void* v; void* o;
    int n_clips, n_pix, F, H, D, GH;
    int64_t cs, fs, ps, ocs, ofs, ops;
    float scale_log2;
};

template <typename T> struct F8;   // 8-wide fragment (QK^T operands)
template <> struct F8<bf16_t> { bf16x8 hi; };
template <> struct F8<float> { bf16x8 hi, lo; };
template <typename T> struct F4;   // 4-wide fragment (PV operands)
template <> struct F4<bf16_t> { s16x4 hi; };
template <> struct F4<float> { s16x4 hi, lo; };

__device__ __forceinline__ void load_f8(const bf16_t* p, bool valid, F8<bf16_t>& f) {
    union { bf16x8 v; u32x4 u; } r;
    r.u = valid ? *reinterpret_cast<const u32x4*>(p) : u32x4{0u, 0u, 0u, 0u};
    f.hi = r.v;
}
__device__ __forceinline__ void load_f8(const float* p, bool valid, F8<float>& f) {
    float v[8];
    if (valid) Vec8<float>::load(p, v);
    else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
    split_bf16x8(v, f.hi, f.lo);
}
__device__ __forceinline__ void mma_qk(const F8<bf16_t>& a, const F8<bf16_t>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_qk(const F8<float>& a, const F8<float>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, acc, 0, 0, 0);
}
__device__ __forceinline__ void make_f4(const float (&v)[4], F4<bf16_t>& f) {
    union { s16x4 s; unsigned u[2]; } r;
    r.u[0] = pack_bf2(v[0], v[1]);
    r.u[1] = pack_bf2(v[2], v[3]);
    f.hi = r.s;
}
__device__ __forceinline__ void make_f4(const float (&v)[4], F4<float>& f) {
    bf16_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { h[i] = f2bf(v[i]); l[i] = f2bf(v[i] - bf2f(h[i])); }
    f.hi = s16x4{(short)h[0], (short)h[1], (short)h[2], (short)h[3]};
    f.lo = s16x4{(short)l[0], (short)l[1], (short)l[2], (short)l[3]};
}
__device__ __forceinline__ void mma_pv(const F4<bf16_t>& a, const F4<bf16_t>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.hi, b.hi, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_pv(const F4<float>& a, const F4<float>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.hi, b.hi, acc, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ F4<T> tr_f4(s16x4 v);
template <> __device__ __forceinline__ F4<bf16_t> tr_f4<bf16_t>(s16x4 v) { F4<bf16_t> f; f.hi = v; return f; }
template <> __device__ __forceinline__ F4<float> tr_f4<float>(s16x4 v) { F4<float> f; f.hi = v; f.lo = v; return f; }   // (never taken: fp32 rows are not 2-byte)
__device__ __forceinline__ float ldsf(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ float ldsf(const float* p) { return *p; }

template <typename T> __device__ __forceinline__ void st4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
}
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&v)[4]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
}

// FT = F/16 frame tiles; NK32 = ceil(D/32) k-steps of the QK^T reduction
template <typename T, int FT, int NK32, int NWV>     // NWV waves per unit (they split its heads)
__global__ __launch_bounds__(64 * NWV) void temporal_attn_kernel(const TAParams P) {
    constexpr int F = FT * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int D = P.D, GH = P.GH, CW = GH * D, CPR = CW / 8, PITCH = CW + 8;
    T* Qs = reinterpret_cast<T*>(smem_raw);  // [F][PITCH]; O overwrites it head by head
    T* Ks = Qs + F * PITCH;
    T* Vs = Ks + F * PITCH;
    // up to 4 waves per unit: they stage the rows together and split the unit's heads, so that the short dependent
    // chains of one head (LDS read -> MFMA -> softmax -> MFMA -> LDS write) overlap with other heads' (one wave per
    // unit ran at ~1 wave per SIMD -- the LDS footprint caps a CU at 5 units -- and every latency was exposed)
    constexpr int NTH = 64 * NWV;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int lane = tid & 63;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- unit decode ---------------------------------------------------------------------------
    const int groups = P.H / GH;
    int u = blockIdx.x;
    const int hg = u % groups; u /= groups;
    const int pix = u % P.n_pix;
    const int clip = u / P.n_pix;
    const int64_t in_off = (int64_t)clip * P.cs + (int64_t)pix * P.ps + (int64_t)hg * CW;
    const int64_t out_off = (int64_t)clip * P.ocs + (int64_t)pix * P.ops + (int64_t)hg * CW;

    // ---- stage Q, K, V: global -> LDS, 16 bytes per lane per load -------------------------------------
    const int chunks = F * CPR;
    if constexpr (sizeof(T) == 2) {
        // every 16-byte load of the unit (Q, K and V rows: 30 KB at F = 16) is in flight before the first LDS write: the unit pays ONE
        // memory round trip (tensor by tensor, as first written, it paid three), and the rows pass through as raw words
        constexpr int MAXC = (F * 40 + NTH - 1) / NTH;     // chunks per thread and tensor at the widest unit (CW = 320)
        if (CPR <= 40 && MAXC <= 3) {               // (5 chunks per tensor in registers -- the 2-wave units at d = 160 -- measured slower than the loop)
            u32x4 r[3][MAXC];
#pragma unroll
            for (int which = 0; which < 3; ++which) {
                const T* src = (const T*)(which == 0 ? P.q : (which == 1 ? P.k : P.v)) + in_off;
#pragma unroll
                for (int j = 0; j < MAXC; ++j) {
                    const int c = tid + j * NTH, f = c / CPR, ch = c - f * CPR;
                    if (c < chunks) r[which][j] = *reinterpret_cast<const u32x4*>(src + (int64_t)f * P.fs + ch * 8);
                }
            }
#pragma unroll
            for (int which = 0; which < 3; ++which) {
                T* dst = which == 0 ? Qs : (which == 1 ? Ks : Vs);
#pragma unroll
                for (int j = 0; j < MAXC; ++j) {
                    const int c = tid + j * NTH, f = c / CPR, ch = c - f * CPR;
                    if (c < chunks) *reinterpret_cast<u32x4*>(dst + f * PITCH + ch * 8) = r[which][j];
                }
            }
        } else {
#pragma unroll 1
            for (int which = 0; which < 3; ++which) {
                const T* src = (const T*)(which == 0 ? P.q : (which == 1 ? P.k : P.v)) + in_off;
                T* dst = which == 0 ? Qs : (which == 1 ? Ks : Vs);
#pragma unroll 5
                for (int c = tid; c < chunks; c += NTH) {
                    const int f = c / CPR, ch = c - f * CPR;
                    *reinterpret_cast<u32x4*>(dst + f * PITCH + ch * 8) = *reinterpret_cast<const u32x4*>(src + (int64_t)f * P.fs + ch * 8);
                }
            }
        }
    } else {
#pragma unroll 1
        for (int which = 0; which < 3; ++which) {
            const T* src = (const T*)(which == 0 ? P.q : (which == 1 ? P.k : P.v)) + in_off;
            T* dst = which == 0 ? Qs : (which == 1 ? Ks : Vs);
#pragma unroll 5
            for (int c = tid; c < chunks; c += NTH) {
                const int f = c / CPR, ch = c - f * CPR;
                float v[8];
                Vec8<T>::load(src + (int64_t)f * P.fs + ch * 8, v);
                Vec8<T>::store(dst + f * PITCH + ch * 8, v);
            }
        }
    }
    __syncthreads();

#define TA_EXP2(x) __builtin_amdgcn_exp2f(x)
#ifndef TA_DBG
#define TA_DBG 0      // timing knock-outs (tools/ubench/ta_bench): 1 no attention arithmetic (rows in, Q rows out)
#endif
    for (int hh = wave; hh < (TA_DBG == 1 ? 0 : GH); hh += NWV) {
        const int hc = hh * D;
#pragma unroll
        for (int qt = 0; qt < FT; ++qt) {
            // ---- S^T = K Q^T for this query tile --------------------------------------------------------
            F8<T> qf[NK32];
#pragma unroll
            for (int ks = 0; ks < NK32; ++ks) {
                const int d0 = ks * 32 + lg * 8;
                load_f8(Qs + (qt * 16 + l15) * PITCH + hc + d0, d0 < D, qf[ks]);
            }
            f32x4 s[FT];
#pragma unroll
            for (int kt = 0; kt < FT; ++kt) {
                s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NK32; ++ks) {
                    const int d0 = ks * 32 + lg * 8;
                    F8<T> kf;
                    load_f8(Ks + (kt * 16 + l15) * PITCH + hc + d0, d0 < D, kf);
                    mma_qk(kf, qf[ks], s[kt]);
                }
            }
            // ---- softmax over keys: lane holds keys kt*16 + lg*4 + r of query qt*16 + l15 --------------------
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < FT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[kt][r] *= P.scale_log2; mx = fmaxf(mx, s[kt][r]); }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < FT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[kt][r] = TA_EXP2(s[kt][r] - mx); sum += s[kt][r]; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
            F4<T> pf[FT];
#pragma unroll
            for (int kt = 0; kt < FT; ++kt) {
                float p4[4] = {s[kt][0] * inv, s[kt][1] * inv, s[kt][2] * inv, s[kt][3] * inv};
                make_f4(p4, pf[kt]);
            }
            // ---- O^T = V^T P^T, 16 output channels at a time; result replaces Q(qt, head) in LDS ---------
            const int ndt = (D + 15) / 16;
            for (int dt = 0; dt < ndt; ++dt) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
                const int dA = dt * 16 + l15;       // channel of this lane's V^T row
#pragma unroll
                for (int kt = 0; kt < FT; ++kt) {
                    F4<T> vf;
                    if constexpr (sizeof(T) == 2) {
                        // transpose read: inside a 16-lane group lane i supplies row (key) i/4, columns 4(i%4).. of the [4 keys][16 d] block and
                        // receives column i -- keys lg*4..+3 of channel dt*16 + l15, the A fragment as it is (was: four 2-byte reads and
                        // four conversions).  Channels >= D of the last block read the neighbouring head / the pad: those output rows
                        // are never stored
                        const bf16_t* vp = reinterpret_cast<const bf16_t*>(Vs) + (kt * 16 + lg * 4 + (l15 >> 2)) * PITCH + hc + dt * 16 + (l15 & 3) * 4;
                        typedef short __attribute__((ext_vector_type(4))) ta_s4;
                        const ta_s4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ta_s4*)(vp));
                        vf = tr_f4<T>(s16x4{t4[0], t4[1], t4[2], t4[3]});
                    } else {
                        float v4[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            v4[i] = dA < D ? ldsf(Vs + (kt * 16 + lg * 4 + i) * PITCH + hc + dA) : 0.f;
                        make_f4(v4, vf);
                    }
                    mma_pv(vf, pf[kt], o);
                }
                const int dO = dt * 16 + lg * 4;    // 4 consecutive output channels of query l15
                if (dO < D) {
                    float o4[4] = {o[0], o[1], o[2], o[3]};
                    st4<T>(Qs + (qt * 16 + l15) * PITCH + hc + dO, o4);
                }
            }
        }
    }
    __syncthreads();

    // ---- O: LDS -> global ------------------------------------------------------------------------------
    T* og = (T*)P.o + out_off;
#pragma unroll 5
    for (int c = tid; c < chunks; c += NTH) {
        const int f = c / CPR, ch = c - f * CPR;
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<u32x4*>(og + (int64_t)f * P.ofs + ch * 8) = *reinterpret_cast<const u32x4*>(Qs + f * PITCH + ch * 8);
        } else {
            float v[8];
            Vec8<T>::load(Qs + f * PITCH + ch * 8, v);
            Vec8<T>::store(og + (int64_t)f * P.ofs + ch * 8, v);
        }
    }
}

// --------------------------------------------------------------------------------------------
// fp8 variant (BASELINE configs[4]: "fp8 MFMA temporal attention").  Q, K, V arrive as OCP e4m3 bytes with one scale per
// tensor (value = byte * scale; written by the QKV projection's epilogue, fmc_linear_fp8_qkv) -- the kernel is HBM bound, so
// halving the bytes it reads is the point; O leaves as bf16.  S^T = K Q^T runs on v_mfma_f32_16x16x32_fp8_fp8 straight on
// the staged bytes (the product of the two scales joins the softmax scale); P V keeps the bf16 MFMA with V converted
// e4m3 -> bf16 in registers (exact), so against the oracle evaluated on the SAME fp8-rounded q, k, v the result carries only
// the bf16 rounding of P and O, like the bf16 kernel.
// --------------------------------------------------------------------------------------------
struct TA8Params {
    const unsigned char* q; const unsigned char* k; const unsigned char* v; bf16_t* o;
    const float* scales;              // device: {scale_q, scale_k, scale_v}
    int n_clips, n_pix, F, H, D, GH;
    int64_t cs, fs, ps, ocs, ofs, ops;   // q/k/v strides in BYTES (= elements), o strides in bf16 elements
    float scale_log2;
};

__device__ __forceinline__ float fp8_to_f32(unsigned char b) { return __builtin_amdgcn_cvt_f32_fp8((int)b, 0); }

template <int FT, int NK32>
__global__ __launch_bounds__(256) void temporal_attn_fp8_kernel(const TA8Params P) {
    constexpr int F = FT * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int D = P.D, GH = P.GH, CW = GH * D, CPR16 = CW / 16, PB = CW + 16;      // byte pitch of the fp8 rows
    const int OPITCH = CW + 8;                                                      // bf16 pitch of the O rows
    unsigned char* Qs = smem_raw;                    // [F][PB]
    unsigned char* Ks = Qs + F * PB;
    unsigned char* Vs = Ks + F * PB;
    bf16_t* Os = reinterpret_cast<bf16_t*>(Vs + F * PB);   // [F][OPITCH]
    const int tid = threadIdx.x, NTH = blockDim.x, wave = tid >> 6, NWV = NTH >> 6;
    const int lane = tid & 63, l15 = lane & 15, lg = lane >> 4;

    const int groups = P.H / GH;
    int u = blockIdx.x;
    const int hg = u % groups; u /= groups;
    const int pix = u % P.n_pix;
    const int clip = u / P.n_pix;
    const int64_t in_off = (int64_t)clip * P.cs + (int64_t)pix * P.ps + (int64_t)hg * CW;
    const int64_t out_off = (int64_t)clip * P.ocs + (int64_t)pix * P.ops + (int64_t)hg * CW;
    const float sq = P.scales[0], sk = P.scales[1], sv = P.scales[2];
    const float sl2 = P.scale_log2 * sq * sk;

    const int chunks = F * CPR16;                    // 16 bytes = 16 channels per lane and load
    constexpr int MAXC8 = (F * 20 + 255) / 256;      // chunks per thread and tensor at the widest unit on 256 threads
    if (NTH == 256 && CPR16 <= 20) {                 // every load of the unit in flight before the first LDS write (one round trip, not three)
        u32x4 r[3][MAXC8];
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            const unsigned char* src = (which == 0 ? P.q : (which == 1 ? P.k : P.v)) + in_off;
#pragma unroll
            for (int j = 0; j < MAXC8; ++j) {
                const int c = tid + j * 256, f = c / CPR16, ch = c - f * CPR16;
                if (c < chunks) r[which][j] = *reinterpret_cast<const u32x4*>(src + (int64_t)f * P.fs + ch * 16);
            }
        }
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            unsigned char* dst = which == 0 ? Qs : (which == 1 ? Ks : Vs);
#pragma unroll
            for (int j = 0; j < MAXC8; ++j) {
                const int c = tid + j * 256, f = c / CPR16, ch = c - f * CPR16;
                if (c < chunks) *reinterpret_cast<u32x4*>(dst + f * PB + ch * 16) = r[which][j];
            }
        }
    } else {
#pragma unroll 1
        for (int which = 0; which < 3; ++which) {
            const unsigned char* src = (which == 0 ? P.q : (which == 1 ? P.k : P.v)) + in_off;
            unsigned char* dst = which == 0 ? Qs : (which == 1 ? Ks : Vs);
#pragma unroll 5
            for (int c = tid; c < chunks; c += NTH) {
                const int f = c / CPR16, ch = c - f * CPR16;
                *reinterpret_cast<u32x4*>(dst + f * PB + ch * 16) = *reinterpret_cast<const u32x4*>(src + (int64_t)f * P.fs + ch * 16);
            }
        }
    }
    __syncthreads();

    for (int hh = wave; hh < GH; hh += NWV) {
        const int hc = hh * D;
#pragma unroll
        for (int qt = 0; qt < FT; ++qt) {
            long qf[NK32];
#pragma unroll
            for (int ks = 0; ks < NK32; ++ks) {
                const int d0 = ks * 32 + lg * 8;
                qf[ks] = d0 < D ? *reinterpret_cast<const long*>(Qs + (qt * 16 + l15) * PB + hc + d0) : 0L;
            }
            f32x4 s[FT];
#pragma unroll
            for (int kt = 0; kt < FT; ++kt) {
                s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NK32; ++ks) {
                    const int d0 = ks * 32 + lg * 8;
                    const long kf = d0 < D ? *reinterpret_cast<const long*>(Ks + (kt * 16 + l15) * PB + hc + d0) : 0L;
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(kf, qf[ks], s[kt], 0, 0, 0);
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < FT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[kt][r] *= sl2; mx = fmaxf(mx, s[kt][r]); }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < FT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - mx); sum += s[kt][r]; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
            F4<bf16_t> pf[FT];
#pragma unroll
            for (int kt = 0; kt < FT; ++kt) {
                float p4[4] = {s[kt][0] * inv, s[kt][1] * inv, s[kt][2] * inv, s[kt][3] * inv};
                make_f4(p4, pf[kt]);
            }
            const int ndt = (D + 15) / 16;
            for (int dt = 0; dt < ndt; ++dt) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
                const int dA = dt * 16 + l15;
#pragma unroll
                for (int kt = 0; kt < FT; ++kt) {
                    float v4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        v4[i] = dA < D ? fp8_to_f32(Vs[(kt * 16 + lg * 4 + i) * PB + hc + dA]) : 0.f;
                    F4<bf16_t> vf;
                    make_f4(v4, vf);                 // e4m3 -> bf16 is exact
                    mma_pv(vf, pf[kt], o);
                }
                const int dO = dt * 16 + lg * 4;
                if (dO < D) {
                    float o4[4] = {o[0] * sv, o[1] * sv, o[2] * sv, o[3] * sv};
                    st4<bf16_t>(Os + (qt * 16 + l15) * OPITCH + hc + dO, o4);
                }
            }
        }
    }
    __syncthreads();
    bf16_t* og = P.o + out_off;
    const int CPR = CW / 8, ochunks = F * CPR;
#pragma unroll 5
    for (int c = tid; c < ochunks; c += NTH) {
        const int f = c / CPR, ch = c - f * CPR;
        *reinterpret_cast<u32x4*>(og + (int64_t)f * P.ofs + ch * 8) = *reinterpret_cast<const u32x4*>(Os + f * OPITCH + ch * 8);
    }
}

// --------------------------------------------------------------------------------------------
// Backward.  Same unit decomposition and staging as the forward; Q, K, V, dO in, dQ, dK, dV out (7 LDS tiles).
// With P = softmax(S), S = scale * Q K^T:   dV = P^T dO,  dP = dO V^T,  dS = P .* (dP - rowsum(P .* dP)),
// dQ = scale * dS K,  dK = scale * dS^T Q.  The 16x16 score tiles are computed twice, once per register layout:
//   L1 (lane = query column, registers = keys): softmax statistics are lane-local; dS^T is the B operand of
//       dQ^T = K^T dS^T (contraction over keys);
//   L2 (lane = key column, registers = queries): P and dS are the B operands of dV^T = dO^T P and dK^T = Q^T dS
//       (contraction over queries); the per-query statistics come from L1 by lane shuffles.
// All of it is noise next to the 7 HBM passes (the kernel is HBM bound like the forward).
// --------------------------------------------------------------------------------------------
struct TABwdParams {
    const void* q; const void* k; const void* v; const void* d_o; void* dq; void* dk; void* dv;
    int n_clips, n_pix, F, H, D, GH;
    int64_t cs, fs, ps, ocs, ofs, ops, dcs, dfs, dps;
    float scale, scale_log2;
    const float* q8_scales;           // non-NULL: q, k, v are e4m3 bytes (strides in bytes) with these three scales; they are
};                                    // dequantised while being staged, the arithmetic below is unchanged

template <typename T> __device__ __forceinline__ void col_f4(const T* base, int pitch, bool valid, F4<T>& f) {
    float v4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v4[i] = valid ? ldsf(base + i * pitch) : 0.f;
    make_f4(v4, f);
}

template <typename T, int FT, int NK32>
__global__ __launch_bounds__(256) void temporal_attn_bwd_kernel(const TABwdParams P) {
    constexpr int F = FT * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int D = P.D, GH = P.GH, CW = GH * D, CPR = CW / 8, PITCH = CW + 8;
    T* Qs = reinterpret_cast<T*>(smem_raw);
    T* Ks = Qs + F * PITCH;
    T* Vs = Ks + F * PITCH;
    T* Gs = Vs + F * PITCH;      // dO
    T* dQs = Gs + F * PITCH;
    T* dKs = dQs + F * PITCH;
    T* dVs = dKs + F * PITCH;
    // up to 4 waves per unit split its heads (see the forward kernel); every head works on its own channel columns of
    // the seven tiles, so the waves never touch the same LDS words
    const int tid = threadIdx.x, NTH = blockDim.x, wave = tid >> 6, NWV = NTH >> 6;
    const int lane = tid & 63;
    const int l15 = lane & 15, lg = lane >> 4;

    const int groups = P.H / GH;
    int u = blockIdx.x;
    const int hg = u % groups; u /= groups;
    const int pix = u % P.n_pix;
    const int clip = u / P.n_pix;
    const int64_t in_off = (int64_t)clip * P.cs + (int64_t)pix * P.ps + (int64_t)hg * CW;
    const int64_t go_off = (int64_t)clip * P.ocs + (int64_t)pix * P.ops + (int64_t)hg * CW;
    const int64_t dq_off = (int64_t)clip * P.dcs + (int64_t)pix * P.dps + (int64_t)hg * CW;

    const int chunks = F * CPR;
    bool staged = false;
    if constexpr (sizeof(T) == 2 && FT == 1) {
        if (NTH == 256 && CPR <= 40 && !P.q8_scales) {   // Q, K, V, dO rows: every load in flight before the first LDS write (one round trip, not four)
            constexpr int MAXCB = (F * 40 + 255) / 256;
            u32x4 r[4][MAXCB];
#pragma unroll
            for (int which = 0; which < 4; ++which) {
                const T* src = which == 0 ? (const T*)P.q + in_off : which == 1 ? (const T*)P.k + in_off
                             : which == 2 ? (const T*)P.v + in_off : (const T*)P.d_o + go_off;
                const int64_t fstride = which == 3 ? P.ofs : P.fs;
#pragma unroll
                for (int j = 0; j < MAXCB; ++j) {
                    const int c = tid + j * 256, f = c / CPR, ch = c - f * CPR;
                    if (c < chunks) r[which][j] = *reinterpret_cast<const u32x4*>(src + (int64_t)f * fstride + ch * 8);
                }
            }
#pragma unroll
            for (int which = 0; which < 4; ++which) {
                T* dst = which == 0 ? Qs : which == 1 ? Ks : which == 2 ? Vs : Gs;
#pragma unroll
                for (int j = 0; j < MAXCB; ++j) {
                    const int c = tid + j * 256, f = c / CPR, ch = c - f * CPR;
                    if (c < chunks) *reinterpret_cast<u32x4*>(dst + f * PITCH + ch * 8) = r[which][j];
                }
            }
            staged = true;
        }
    }
#pragma unroll 1
    for (int which = 0; which < (staged ? 0 : 4); ++which) {
        T* dst = which == 0 ? Qs : which == 1 ? Ks : which == 2 ? Vs : Gs;
        if (P.q8_scales && which < 3) {
            const unsigned char* src8 = (const unsigned char*)(which == 0 ? P.q : which == 1 ? P.k : P.v) + in_off;
            const float sc = P.q8_scales[which];
            for (int c = tid; c < chunks; c += NTH) {
                const int f = c / CPR, ch = c - f * CPR;
                const u32x2 w = *reinterpret_cast<const u32x2*>(src8 + (int64_t)f * P.fs + ch * 8);
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fp8_to_f32((unsigned char)((w[i >> 2] >> (8 * (i & 3))) & 0xffu)) * sc;
                Vec8<T>::store(dst + f * PITCH + ch * 8, v);
            }
            continue;
        }
        const T* src = which == 0 ? (const T*)P.q + in_off : which == 1 ? (const T*)P.k + in_off
                     : which == 2 ? (const T*)P.v + in_off : (const T*)P.d_o + go_off;
        const int64_t fstride = which == 3 ? P.ofs : P.fs;
#pragma unroll 5
        for (int c = tid; c < chunks; c += NTH) {
            const int f = c / CPR, ch = c - f * CPR;
            float v[8];
            Vec8<T>::load(src + (int64_t)f * fstride + ch * 8, v);
            Vec8<T>::store(dst + f * PITCH + ch * 8, v);
        }
    }
    __syncthreads();

    const int ndt = (D + 15) / 16;
    for (int hh = wave; hh < GH; hh += NWV) {
        const int hc = hh * D;
        float st_m[FT], st_inv[FT], st_d[FT];          // per query (lane l15 of tile qt), from layout L1
        // ================= L1: lane = query, registers = keys =================
#pragma unroll
        for (int qt = 0; qt < FT; ++qt) {
            F8<T> qf[NK32], gf[NK32];
#pragma unroll
            for (int ks = 0; ks < NK32; ++ks) {
                const int d0 = ks * 32 + lg * 8;
                load_f8(Qs + (qt * 16 + l15) * PITCH + hc + d0, d0 < D, qf[ks]);
                load_f8(Gs + (qt * 16 + l15) * PITCH + hc + d0, d0 < D, gf[ks]);
            }
            f32x4 s[FT], dp[FT];
#pragma unroll
            for (int kt = 0; kt < FT; ++kt) {
                s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NK32; ++ks) {
                    const int d0 = ks * 32 + lg * 8;
                    F8<T> kf, vf;
                    load_f8(Ks + (kt * 16 + l15) * PITCH + hc + d0, d0 < D, kf);
                    load_f8(Vs + (kt * 16 + l15) * PITCH + hc + d0, d0 < D, vf);
                    mma_qk(kf, qf[ks], s[kt]);
                    mma_qk(vf, gf[ks], dp[kt]);
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < FT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[kt][r] *= P.scale_log2; mx = fmaxf(mx, s[kt][r]); }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < FT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - mx); sum += s[kt][r]; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
            float dsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < FT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[kt][r] *= inv; dsum += s[kt][r] * dp[kt][r]; }
            dsum += __shfl_xor(dsum, 16, 64);
            dsum += __shfl_xor(dsum, 32, 64);
            st_m[qt] = mx; st_inv[qt] = inv; st_d[qt] = dsum;
            F4<T> dsf[FT];
#pragma unroll
            for (int kt = 0; kt < FT; ++kt) {
                float d4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) d4[r] = s[kt][r] * (dp[kt][r] - dsum) * P.scale;
                make_f4(d4, dsf[kt]);
            }
            // dQ^T[d, q] = sum_kv K^T[d, kv] dS^T[kv, q]
            for (int dt = 0; dt < ndt; ++dt) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
                const int dA = dt * 16 + l15;
#pragma unroll
                for (int kt = 0; kt < FT; ++kt) {
                    F4<T> kf4;
                    col_f4(Ks + (kt * 16 + lg * 4) * PITCH + hc + dA, PITCH, dA < D, kf4);
                    mma_pv(kf4, dsf[kt], o);
                }
                const int dO_ = dt * 16 + lg * 4;
                if (dO_ < D) {
                    float o4[4] = {o[0], o[1], o[2], o[3]};
                    st4<T>(dQs + (qt * 16 + l15) * PITCH + hc + dO_, o4);
                }
            }
        }
        // ================= L2: lane = key, registers = queries =================
#pragma unroll
        for (int kt = 0; kt < FT; ++kt) {
            F8<T> kf[NK32], vf[NK32];
#pragma unroll
            for (int ks = 0; ks < NK32; ++ks) {
                const int d0 = ks * 32 + lg * 8;
                load_f8(Ks + (kt * 16 + l15) * PITCH + hc + d0, d0 < D, kf[ks]);
                load_f8(Vs + (kt * 16 + l15) * PITCH + hc + d0, d0 < D, vf[ks]);
            }
            F4<T> pf[FT], dsf[FT];
#pragma unroll
            for (int qt = 0; qt < FT; ++qt) {
                f32x4 s2 = f32x4{0.f, 0.f, 0.f, 0.f}, dp2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NK32; ++ks) {
                    const int d0 = ks * 32 + lg * 8;
                    F8<T> qf, gf;
                    load_f8(Qs + (qt * 16 + l15) * PITCH + hc + d0, d0 < D, qf);
                    load_f8(Gs + (qt * 16 + l15) * PITCH + hc + d0, d0 < D, gf);
                    mma_qk(qf, kf[ks], s2);          // rows q = qt*16 + lg*4 + r, col kv = kt*16 + l15
                    mma_qk(gf, vf[ks], dp2);
                }
                float p4[4], d4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = lg * 4 + r;       // lane that owns this query's statistics in L1
                    const float m = __shfl(st_m[qt], ql, 64), inv = __shfl(st_inv[qt], ql, 64);
                    const float dsum = __shfl(st_d[qt], ql, 64);
                    p4[r] = exp2f(s2[r] * P.scale_log2 - m) * inv;
                    d4[r] = p4[r] * (dp2[r] - dsum) * P.scale;
                }
                make_f4(p4, pf[qt]);
                make_f4(d4, dsf[qt]);
            }
            // dV^T[d, kv] = sum_q dO^T[d, q] P[q, kv];  dK^T[d, kv] = sum_q Q^T[d, q] dS[q, kv]
            for (int dt = 0; dt < ndt; ++dt) {
                f32x4 ov = f32x4{0.f, 0.f, 0.f, 0.f}, ok = f32x4{0.f, 0.f, 0.f, 0.f};
                const int dA = dt * 16 + l15;
#pragma unroll
                for (int qt = 0; qt < FT; ++qt) {
                    F4<T> gf4, qf4;
                    col_f4(Gs + (qt * 16 + lg * 4) * PITCH + hc + dA, PITCH, dA < D, gf4);
                    col_f4(Qs + (qt * 16 + lg * 4) * PITCH + hc + dA, PITCH, dA < D, qf4);
                    mma_pv(gf4, pf[qt], ov);
                    mma_pv(qf4, dsf[qt], ok);
                }
                const int dO_ = dt * 16 + lg * 4;
                if (dO_ < D) {
                    float v4[4] = {ov[0], ov[1], ov[2], ov[3]}, k4[4] = {ok[0], ok[1], ok[2], ok[3]};
                    st4<T>(dVs + (kt * 16 + l15) * PITCH + hc + dO_, v4);
                    st4<T>(dKs + (kt * 16 + l15) * PITCH + hc + dO_, k4);
                }
            }
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int which = 0; which < 3; ++which) {
        T* dst = (T*)(which == 0 ? P.dq : which == 1 ? P.dk : P.dv) + dq_off;
        const T* srcl = which == 0 ? dQs : which == 1 ? dKs : dVs;
#pragma unroll 5
        for (int c = tid; c < chunks; c += NTH) {
            const int f = c / CPR, ch = c - f * CPR;
            float v[8];
            Vec8<T>::load(srcl + f * PITCH + ch * 8, v);
            Vec8<T>::store(dst + (int64_t)f * P.dfs + ch * 8, v);
        }
    }
}

template <typename T, int FT, int NK32>
void launch_ta_bwd(const TABwdParams& P, hipStream_t st) {
    const int CW = P.GH * P.D;
    const size_t lds = sizeof(T) * 7 * (size_t)(FT * 16) * (CW + 8);
    if (lds > 64 * 1024) {
        static FmcPerDeviceFlag raised;
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_bwd_kernel<T, FT, NK32>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised = true;
        }
    }
    const int waves = P.GH >= 4 ? 4 : (P.GH >= 2 ? 2 : 1);
    dim3 grid((unsigned)((int64_t)P.n_clips * P.n_pix * (P.H / P.GH))), block(64 * waves);
    hipLaunchKernelGGL((temporal_attn_bwd_kernel<T, FT, NK32>), grid, block, lds, st, P);
}

template <typename T, int FT>
int dispatch_ta_bwd_k(const TABwdParams& P, hipStream_t st) {
    switch ((P.D + 31) / 32) {
        case 1: launch_ta_bwd<T, FT, 1>(P, st); break;
        case 2: launch_ta_bwd<T, FT, 2>(P, st); break;
        case 3: launch_ta_bwd<T, FT, 3>(P, st); break;
        case 4: launch_ta_bwd<T, FT, 4>(P, st); break;
        case 5: launch_ta_bwd<T, FT, 5>(P, st); break;
        default: FMC_FAIL(FMC_E_SHAPE, "temporal_attn_bwd: head dim %d > 160", P.D);
    }
    return 0;
}

template <typename T, int FT, int NK32>
void launch_ta(const TAParams& P, hipStream_t st) {
    const int CW = P.GH * P.D;
    const size_t lds = sizeof(T) * 3 * (size_t)(FT * 16) * (CW + 8);
    dim3 grid((unsigned)((int64_t)P.n_clips * P.n_pix * (P.H / P.GH)));
    auto go = [&](auto nw) {                                          // the waves of a workgroup split the unit's heads
        constexpr int NW = decltype(nw)::value;
        if (lds > 64 * 1024) {
            static FmcPerDeviceFlag raised;
            if (!raised) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_kernel<T, FT, NK32, NW>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                raised = true;
            }
        }
        hipLaunchKernelGGL((temporal_attn_kernel<T, FT, NK32, NW>), grid, dim3(64 * NW), lds, st, P);
    };
    if (P.GH >= 4) go(std::integral_constant<int, 4>{});
    else if (P.GH >= 2) go(std::integral_constant<int, 2>{});
    else go(std::integral_constant<int, 1>{});
}

template <typename T, int FT>
int dispatch_ta_k(const TAParams& P, hipStream_t st) {
    switch ((P.D + 31) / 32) {
        case 1: launch_ta<T, FT, 1>(P, st); break;
        case 2: launch_ta<T, FT, 2>(P, st); break;
        case 3: launch_ta<T, FT, 3>(P, st); break;
        case 4: launch_ta<T, FT, 4>(P, st); break;
        case 5: launch_ta<T, FT, 5>(P, st); break;
        default: FMC_FAIL(FMC_E_SHAPE, "temporal_attn: head dim %d > 160", P.D);
    }
    return 0;
}

template <typename T>
int dispatch_ta(const TAParams& P, hipStream_t st) {
    if (P.F == 16) return dispatch_ta_k<T, 1>(P, st);
    if (P.F == 32) return dispatch_ta_k<T, 2>(P, st);
    FMC_FAIL(FMC_E_SHAPE, "temporal_attn: F must be 16 or 32 (got %d)", P.F);
}

}  // namespace

extern "C" int fmc_temporal_attn_fwd(const void* q, const void* k, const void* v, void* o, int n_clips, int n_pix,
                                     int F, int H, int D, int64_t clip_stride, int64_t frame_stride,
                                     int64_t pix_stride, int64_t o_clip_stride, int64_t o_frame_stride,
                                     int64_t o_pix_stride, float scale, int dtype, void* stream) {
    if (!q || !k || !v || !o) FMC_FAIL(FMC_E_NULL, "temporal_attn: NULL tensor");
    if (dtype != FMC_BF16 && dtype != FMC_F32) FMC_FAIL(FMC_E_DTYPE, "temporal_attn: dtype %d", dtype);
    if (n_clips <= 0 || n_pix <= 0 || H <= 0 || D <= 0 || D % 8 || D > 160)
        FMC_FAIL(FMC_E_SHAPE, "temporal_attn: need D%%8==0, D<=160, positive sizes (clips=%d pix=%d H=%d D=%d)", n_clips,
                 n_pix, H, D);
    const int64_t strides[] = {clip_stride, frame_stride, pix_stride, o_clip_stride, o_frame_stride, o_pix_stride};
    for (int64_t s : strides)
        if (s % 8) FMC_FAIL(FMC_E_ALIGN, "temporal_attn: strides must be multiples of 8 elements");
    if (!fmc_aligned16(q) || !fmc_aligned16(k) || !fmc_aligned16(v) || !fmc_aligned16(o))
        FMC_FAIL(FMC_E_ALIGN, "temporal_attn: tensors must be 16-byte aligned");
    TAParams P;
    P.q = q; P.k = k; P.v = v; P.o = o;
    P.n_clips = n_clips; P.n_pix = n_pix; P.F = F; P.H = H; P.D = D;
    // head group: the largest divisor GH of H with GH*D <= 320 channels (one 640-byte bf16 run per frame row)
    int gh = 1;
    for (int g = 1; g <= H; ++g)
        if (H % g == 0 && g * D <= 320) gh = g;
    P.GH = gh;
    P.cs = clip_stride; P.fs = frame_stride; P.ps = pix_stride;
    P.ocs = o_clip_stride; P.ofs = o_frame_stride; P.ops = o_pix_stride;
    P.scale_log2 = scale * LOG2E;
    hipStream_t st = (hipStream_t)stream;
    int rc = (dtype == FMC_BF16) ? dispatch_ta<bf16_t>(P, st) : dispatch_ta<float>(P, st);
    if (rc) return rc;
    FMC_CHECK_LAUNCH("fmc_temporal_attn_fwd");
    return 0;
}

extern "C" int fmc_temporal_attn_bwd(const void* q, const void* k, const void* v, const void* d_o, void* dq, void* dk,
                                     void* dv, int n_clips, int n_pix, int F, int H, int D, int64_t clip_stride,
                                     int64_t frame_stride, int64_t pix_stride, int64_t do_clip_stride,
                                     int64_t do_frame_stride, int64_t do_pix_stride, int64_t dq_clip_stride,
                                     int64_t dq_frame_stride, int64_t dq_pix_stride, float scale, int dtype,
                                     void* stream) {
    if (!q || !k || !v || !d_o || !dq || !dk || !dv) FMC_FAIL(FMC_E_NULL, "temporal_attn_bwd: NULL tensor");
    if (dtype != FMC_BF16 && dtype != FMC_F32) FMC_FAIL(FMC_E_DTYPE, "temporal_attn_bwd: dtype %d", dtype);
    if (n_clips <= 0 || n_pix <= 0 || H <= 0 || D <= 0 || D % 8 || D > 160 || (F != 16 && F != 32))
        FMC_FAIL(FMC_E_SHAPE, "temporal_attn_bwd: need F in {16,32}, D%%8==0, D<=160 (F=%d H=%d D=%d)", F, H, D);
    const int64_t strides[] = {clip_stride, frame_stride, pix_stride, do_clip_stride, do_frame_stride, do_pix_stride,
                               dq_clip_stride, dq_frame_stride, dq_pix_stride};
    for (int64_t s : strides)
        if (s % 8) FMC_FAIL(FMC_E_ALIGN, "temporal_attn_bwd: strides must be multiples of 8 elements");
    const void* ptrs[] = {q, k, v, d_o, dq, dk, dv};
    for (const void* p : ptrs)
        if (!fmc_aligned16(p)) FMC_FAIL(FMC_E_ALIGN, "temporal_attn_bwd: tensors must be 16-byte aligned");
    TABwdParams P;
    P.q = q; P.k = k; P.v = v; P.d_o = d_o; P.dq = dq; P.dk = dk; P.dv = dv;
    P.n_clips = n_clips; P.n_pix = n_pix; P.F = F; P.H = H; P.D = D;
    // head group: largest divisor of H whose 7 LDS tiles fit in ~150 KiB and whose rows are <= 320 channels
    const size_t esz = dtype == FMC_BF16 ? 2 : 4;
    int gh = 1;
    for (int g = 1; g <= H; ++g)
        if (H % g == 0 && g * D <= 320 && 7 * (size_t)F * (g * D + 8) * esz <= 150 * 1024) gh = g;
    if (7 * (size_t)F * (gh * D + 8) * esz > 160 * 1024) FMC_FAIL(FMC_E_SHAPE, "temporal_attn_bwd: F=%d D=%d does not fit LDS in this dtype", F, D);
    P.GH = gh;
    P.cs = clip_stride; P.fs = frame_stride; P.ps = pix_stride;
    P.ocs = do_clip_stride; P.ofs = do_frame_stride; P.ops = do_pix_stride;
    P.dcs = dq_clip_stride; P.dfs = dq_frame_stride; P.dps = dq_pix_stride;
    P.scale = scale; P.scale_log2 = scale * LOG2E;
    P.q8_scales = nullptr;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dtype == FMC_BF16) rc = F == 16 ? dispatch_ta_bwd_k<bf16_t, 1>(P, st) : dispatch_ta_bwd_k<bf16_t, 2>(P, st);
    else rc = F == 16 ? dispatch_ta_bwd_k<float, 1>(P, st) : dispatch_ta_bwd_k<float, 2>(P, st);
    if (rc) return rc;
    FMC_CHECK_LAUNCH("fmc_temporal_attn_bwd");
    return 0;
}

namespace {
template <int FT, int NK32>
void launch_ta8(const TA8Params& P, hipStream_t st) {
    const int CW = P.GH * P.D;
    const size_t lds = 3 * (size_t)(FT * 16) * (CW + 16) + sizeof(bf16_t) * (size_t)(FT * 16) * (CW + 8);
    if (lds > 64 * 1024) {
        static FmcPerDeviceFlag raised;
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_fp8_kernel<FT, NK32>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised = true;
        }
    }
    const int waves = P.GH >= 4 ? 4 : (P.GH >= 2 ? 2 : 1);
    dim3 grid((unsigned)((int64_t)P.n_clips * P.n_pix * (P.H / P.GH))), block(64 * waves);
    hipLaunchKernelGGL((temporal_attn_fp8_kernel<FT, NK32>), grid, block, lds, st, P);
}
template <int FT>
int dispatch_ta8(const TA8Params& P, hipStream_t st) {
    switch ((P.D + 31) / 32) {
        case 1: launch_ta8<FT, 1>(P, st); break;
        case 2: launch_ta8<FT, 2>(P, st); break;
        case 3: launch_ta8<FT, 3>(P, st); break;
        case 4: launch_ta8<FT, 4>(P, st); break;
        case 5: launch_ta8<FT, 5>(P, st); break;
        default: FMC_FAIL(FMC_E_SHAPE, "temporal_attn_fp8: head dim %d > 160", P.D);
    }
    return 0;
}
}  // namespace

extern "C" int fmc_temporal_attn_fp8_fwd(const void* q, const void* k, const void* v, void* o, const void* scales,
                                         int n_clips, int n_pix, int F, int H, int D, int64_t clip_stride,
                                         int64_t frame_stride, int64_t pix_stride, int64_t o_clip_stride,
                                         int64_t o_frame_stride, int64_t o_pix_stride, float scale, void* stream) {
    if (!q || !k || !v || !o || !scales) FMC_FAIL(FMC_E_NULL, "temporal_attn_fp8: NULL tensor");
    if (n_clips <= 0 || n_pix <= 0 || H <= 0 || D <= 0 || D % 8 || D > 160 || (F != 16 && F != 32))
        FMC_FAIL(FMC_E_SHAPE, "temporal_attn_fp8: need F in {16,32}, D%%8==0, D<=160 (F=%d H=%d D=%d)", F, H, D);
    const int64_t strides[] = {clip_stride, frame_stride, pix_stride};
    for (int64_t s : strides)
        if (s % 16) FMC_FAIL(FMC_E_ALIGN, "temporal_attn_fp8: q/k/v strides must be multiples of 16 bytes");
    const int64_t ostrides[] = {o_clip_stride, o_frame_stride, o_pix_stride};
    for (int64_t s : ostrides)
        if (s % 8) FMC_FAIL(FMC_E_ALIGN, "temporal_attn_fp8: o strides must be multiples of 8 elements");
    if (!fmc_aligned16(q) || !fmc_aligned16(k) || !fmc_aligned16(v) || !fmc_aligned16(o))
        FMC_FAIL(FMC_E_ALIGN, "temporal_attn_fp8: tensors must be 16-byte aligned");
    TA8Params P;
    P.q = (const unsigned char*)q; P.k = (const unsigned char*)k; P.v = (const unsigned char*)v; P.o = (bf16_t*)o;
    P.scales = (const float*)scales;
    P.n_clips = n_clips; P.n_pix = n_pix; P.F = F; P.H = H; P.D = D;
    int gh = 1;                        // head group: GH*D channels <= 320 and a multiple of 16 bytes per staged row
    for (int g = 1; g <= H; ++g)
        if (H % g == 0 && g * D <= 320 && (g * D) % 16 == 0) gh = g;
    if ((gh * D) % 16) FMC_FAIL(FMC_E_SHAPE, "temporal_attn_fp8: no head group with (GH*D) %% 16 == 0 (H=%d D=%d)", H, D);
    P.GH = gh;
    P.cs = clip_stride; P.fs = frame_stride; P.ps = pix_stride;
    P.ocs = o_clip_stride; P.ofs = o_frame_stride; P.ops = o_pix_stride;
    P.scale_log2 = scale * LOG2E;
    hipStream_t st = (hipStream_t)stream;
    const int rc = F == 16 ? dispatch_ta8<1>(P, st) : dispatch_ta8<2>(P, st);
    if (rc) return rc;
    FMC_CHECK_LAUNCH("fmc_temporal_attn_fp8_fwd");
    return 0;
}

extern "C" int fmc_temporal_attn_fp8_bwd(const void* q, const void* k, const void* v, const void* scales, const void* d_o,
                                         void* dq, void* dk, void* dv, int n_clips, int n_pix, int F, int H, int D,
                                         int64_t clip_stride, int64_t frame_stride, int64_t pix_stride,
                                         int64_t do_clip_stride, int64_t do_frame_stride, int64_t do_pix_stride,
                                         int64_t dq_clip_stride, int64_t dq_frame_stride, int64_t dq_pix_stride, float scale,
                                         void* stream) {
    if (!q || !k || !v || !scales || !d_o || !dq || !dk || !dv) FMC_FAIL(FMC_E_NULL, "temporal_attn_fp8_bwd: NULL tensor");
    if (n_clips <= 0 || n_pix <= 0 || H <= 0 || D <= 0 || D % 8 || D > 160 || (F != 16 && F != 32))
        FMC_FAIL(FMC_E_SHAPE, "temporal_attn_fp8_bwd: need F in {16,32}, D%%8==0, D<=160 (F=%d H=%d D=%d)", F, H, D);
    const int64_t strides[] = {clip_stride, frame_stride, pix_stride, do_clip_stride, do_frame_stride, do_pix_stride,
                               dq_clip_stride, dq_frame_stride, dq_pix_stride};
    for (int64_t s : strides)
        if (s % 8) FMC_FAIL(FMC_E_ALIGN, "temporal_attn_fp8_bwd: strides must be multiples of 8");
    TABwdParams P;
    P.q = q; P.k = k; P.v = v; P.d_o = d_o; P.dq = dq; P.dk = dk; P.dv = dv;
    P.n_clips = n_clips; P.n_pix = n_pix; P.F = F; P.H = H; P.D = D;
    int gh = 1;
    for (int g = 1; g <= H; ++g)
        if (H % g == 0 && g * D <= 320 && 7 * (size_t)F * (g * D + 8) * 2 <= 150 * 1024) gh = g;
    if (7 * (size_t)F * (gh * D + 8) * 2 > 160 * 1024) FMC_FAIL(FMC_E_SHAPE, "temporal_attn_fp8_bwd: F=%d D=%d does not fit LDS", F, D);
    P.GH = gh;
    P.cs = clip_stride; P.fs = frame_stride; P.ps = pix_stride;          // bytes = elements for the e4m3 inputs
    P.ocs = do_clip_stride; P.ofs = do_frame_stride; P.ops = do_pix_stride;
    P.dcs = dq_clip_stride; P.dfs = dq_frame_stride; P.dps = dq_pix_stride;
    P.scale = scale; P.scale_log2 = scale * LOG2E;
    P.q8_scales = (const float*)scales;
    hipStream_t st = (hipStream_t)stream;
    const int rc = F == 16 ? dispatch_ta_bwd_k<bf16_t, 1>(P, st) : dispatch_ta_bwd_k<bf16_t, 2>(P, st);
    if (rc) return rc;
    FMC_CHECK_LAUNCH("fmc_temporal_attn_fp8_bwd");
    return 0;
}

namespace {
__global__ void fp8_scales_roll_kernel(float* amax, float* scale, float* inv_scale, float k) {
    const int i = threadIdx.x;
    if (i < 3) {
        const float s = fmaxf(amax[i] * k, 1e-12f);
        scale[i] = s;
        inv_scale[i] = 1.f / s;
        amax[i] = 0.f;
    }
}
}  // namespace

extern "C" int fmc_fp8_scales_roll(void* amax, void* scale, void* inv_scale, float margin, void* stream) {
    if (!amax || !scale || !inv_scale) FMC_FAIL(FMC_E_NULL, "fp8_scales_roll: NULL tensor");
    hipLaunchKernelGGL(fp8_scales_roll_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (float*)amax, (float*)scale, (float*)inv_scale,
                       margin / 448.f);
    FMC_CHECK_LAUNCH("fmc_fp8_scales_roll");
    return 0;
}
