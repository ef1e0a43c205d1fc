// Shared helpers for the gfx950 kernels of libfmc_hip.so (wave = 64 lanes, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/fmc_hip.h"

// ---- error plumbing (thread-local message, negative status codes) -----------------------------
void fmc_set_error(const char* fmt, ...);

#define FMC_FAIL(code, ...)          \
    do {                             \
        fmc_set_error(__VA_ARGS__);  \
        return (code);               \
    } while (0)

#define FMC_CHECK_LAUNCH(name)                                                       \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) FMC_FAIL(FMC_E_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// ---- per-device host state -------------------------------------------------------------------------
// A process may drive several GPUs: the CU count and the "dynamic-LDS attribute raised" flags are kept per device id.
int fmc_device();                 // the calling thread's current device (0 when the runtime cannot say)
int fmc_cu_count();               // compute units of that device, looked up once per device
struct FmcPerDeviceFlag {         // `static FmcPerDeviceFlag raised; if (!raised) { ...; raised = true; }` -- once per device, not per process
    unsigned long long mask = 0;
    bool operator!() const { return !((mask >> (fmc_device() & 63)) & 1ull); }
    FmcPerDeviceFlag& operator=(bool v) { if (v) mask |= 1ull << (fmc_device() & 63); return *this; }
};

static inline bool fmc_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- vector / fragment types ---------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

typedef unsigned short bf16_t;  // raw storage

// float -> bf16, round-to-nearest-even, through the hardware converter (v_cvt_pk_bf16_f32 on gfx950: one VALU op
// per PAIR of values; the integer add/shift sequence costs ~5 ops per value)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    union { bf16x2_t v; unsigned u; } r;
    r.v = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return r.u;
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

// GELU of a gate whose product is rounded to bf16 (the GEGLU epilogues of the bf16 kernels): g * sigmoid(2 g (c0 + c1 g^2 + c2 g^4)), the minimax fit of
// that form to the exact g/2 (1 + erf(g / sqrt 2)) of torch.nn.functional.gelu (diffusers GEGLU.gelu) on |g| <= 9 -- |error| <= 2.6e-5 absolute everywhere
// (tests/test_gelu_fast.py; bf16's half-ulp is 2e-3 relative), 7 VALU + exp2 + rcp per value against 14 + 2 for the erf of Abramowitz & Stegun 7.1.26: the
// level-0 / level-1 GEGLU launches spend a quarter of their time in this function (226 -> 203 us and 151 -> 138 us with it, 176 / 132 us with no GELU at
// all).  The constants carry the -2 log2(e) of the exponent; g^2 is clamped where the quartic would turn around (the tails are exact: 0 and g).
// `make GELU_EXACT=1` builds the erf form into every bf16 kernel instead (A/B and parity experiments); the fp32 parity mode always calls erff.
#ifndef FMC_GELU_EXACT
#define FMC_GELU_EXACT 0
#endif
__device__ __forceinline__ float fmc_gelu_fast(float g) {
    const float g2 = fminf(g * g, 81.f);
    const float s = fmaf(fmaf(1.0142630198970437e-3f, g2, -0.10677571594715118f), g2, -2.301121234893799f);
    return g * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(g * s));
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }

// 8 consecutive activations <-> 8 floats, for both storage types
template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> {
    typedef u32x4 raw_t;                                  // 8 elements as stored
    __device__ __forceinline__ static raw_t load_raw(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
    __device__ __forceinline__ static void unpack(const raw_t& r, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(r[i] << 16);
            v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ static void load(const bf16_t* p, float (&v)[8]) { unpack(load_raw(p), v); }
    __device__ __forceinline__ static void store(bf16_t* p, const float (&v)[8]) {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<u32x4*>(p) = r;
    }
};
template <> struct Vec8<float> {
    struct raw_t { f32x4 a, b; };
    __device__ __forceinline__ static raw_t load_raw(const float* p) {
        return raw_t{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)};
    }
    __device__ __forceinline__ static void unpack(const raw_t& r, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = r.a[i]; v[4 + i] = r.b[i]; }
    }
    __device__ __forceinline__ static void load(const float* p, float (&v)[8]) {
        f32x4 a = *reinterpret_cast<const f32x4*>(p);
        f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
    __device__ __forceinline__ static void store(float* p, const float (&v)[8]) {
        f32x4 a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
        *reinterpret_cast<f32x4*>(p) = a;
        *reinterpret_cast<f32x4*>(p + 4) = b;
    }
};

// split 8 fp32 values into a bf16 "hi" fragment and the bf16 of the remainder ("lo"):
// a ~= hi + lo to ~2^-17 relative, so a*b ~= hi*hi' + hi*lo' + lo*hi' on the bf16 MFMA pipe.
__device__ __forceinline__ void split_bf16x8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    union { bf16x8 v; bf16_t s[8]; } h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bf16_t b = f2bf(v[i]);
        h.s[i] = b;
        l.s[i] = f2bf(v[i] - bf2f(b));
    }
    hi = h.v;
    lo = l.v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
