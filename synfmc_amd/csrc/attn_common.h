// Fragment helpers shared by the spatial attention forward / backward kernels (gfx950, v_mfma_f32_32x32x16_bf16).
// Storage type T = bf16_t: one MFMA per product; T = float: split-bf16 x3 (hi*hi + hi*lo + lo*hi), the parity mode.
#pragma once
#include "common.h"

namespace {

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 hi; };
template <> struct Frag<float> { bf16x8 hi, lo; };

__device__ __forceinline__ bf16x8 zero_frag() {
    union { bf16x8 v; u32x4 u; } z;
    z.u = u32x4{0u, 0u, 0u, 0u};
    return z.v;
}

// 8 consecutive elements (global or LDS) -> MFMA fragment(s)
template <typename T> __device__ __forceinline__ void make_frag(const T* p, Frag<T>& f);
template <> __device__ __forceinline__ void make_frag<bf16_t>(const bf16_t* p, Frag<bf16_t>& f) {
    union { bf16x8 v; u32x4 u; } r;
    r.u = *reinterpret_cast<const u32x4*>(p);
    f.hi = r.v;
}
template <> __device__ __forceinline__ void make_frag<float>(const float* p, Frag<float>& f) {
    float v[8];
    Vec8<float>::load(p, v);
    split_bf16x8(v, f.hi, f.lo);
}
// two runs of 4 consecutive elements -> fragment
template <typename T> __device__ __forceinline__ void make_frag_2x4(const T* p0, const T* p1, Frag<T>& f);
template <> __device__ __forceinline__ void make_frag_2x4<bf16_t>(const bf16_t* p0, const bf16_t* p1, Frag<bf16_t>& f) {
    union { bf16x8 v; u32x2 u[2]; } r;
    r.u[0] = *reinterpret_cast<const u32x2*>(p0);
    r.u[1] = *reinterpret_cast<const u32x2*>(p1);
    f.hi = r.v;
}
template <> __device__ __forceinline__ void make_frag_2x4<float>(const float* p0, const float* p1, Frag<float>& f) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p0);
    f32x4 b = *reinterpret_cast<const f32x4*>(p1);
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    split_bf16x8(v, f.hi, f.lo);
}
template <typename T> __device__ __forceinline__ void zero(Frag<T>& f);
template <> __device__ __forceinline__ void zero<bf16_t>(Frag<bf16_t>& f) { f.hi = zero_frag(); }
template <> __device__ __forceinline__ void zero<float>(Frag<float>& f) { f.hi = zero_frag(); f.lo = zero_frag(); }

// acc += A * B  (bf16: one MFMA; fp32 storage: hi*hi + hi*lo + lo*hi)
__device__ __forceinline__ void mma32(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(const Frag<float>& a, const Frag<float>& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
}

// 8 probabilities (fp32) -> P^T fragment(s)
__device__ __forceinline__ void p_frag(const float (&p)[8], Frag<bf16_t>& f) {
    union { bf16x8 v; unsigned u[4]; } r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.u[i] = pack_bf2(p[2 * i], p[2 * i + 1]);
    f.hi = r.v;
}
__device__ __forceinline__ void p_frag(const float (&p)[8], Frag<float>& f) { split_bf16x8(p, f.hi, f.lo); }

template <typename T> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf2(a, b), pack_bf2(c, d)};
}
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}

// NKS: number of 16-wide k-steps of the QK^T reduction (D padded to 16*NKS); NDT = ceil(NKS/2)
}  // namespace
