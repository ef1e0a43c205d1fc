// Conditioning-path kernels for gfx950: Pluecker rays, OMC rasteriser, mask modulation, OMC feature
// add, fused CFG + DDIM update.  All are write- or read-once HBM passes; each thread produces
// 16-byte (or wider) contiguous runs so that a wave writes whole cache lines.
//
// Algorithmic bytes: plucker = B*F*H*W*6*e written; rasterize = BF*H*W*(n_obj*4 read
// + 13*e written + 4 mask); mask_modulate = 2*N*h*w*C*e; feature_add = 3*n*e (2*n*e in place on
// the conditioned half only); cfg_ddim = n*(2*e + 8).
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ T cvt_out(float v);
template <> __device__ __forceinline__ float cvt_out<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t cvt_out<bf16_t>(float v) { return f2bf(v); }

// ---- Pluecker ---------------------------------------------------------------------------------
struct Cam {
    float fx, fy, cx, cy;
    float R[3][3];
    float t[3];
};

__device__ __forceinline__ Cam load_cam(const float* K, const float* c2w, int bf, int c2w_rows) {
    Cam c;
    c.fx = K[bf * 4 + 0]; c.fy = K[bf * 4 + 1]; c.cx = K[bf * 4 + 2]; c.cy = K[bf * 4 + 3];
    const float* m = c2w + (size_t)bf * c2w_rows * 4;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) c.R[a][b] = m[a * 4 + b];
        c.t[a] = m[a * 4 + 3];
    }
    return c;
}

// (o x d, d) for pixel centre (x+0.5, y+0.5); same operation order as ray_condition
// (fmc/data/dataset.py:955-969): direction, normalise, rotate by R, cross with the translation.
__device__ __forceinline__ void plucker_pixel(const Cam& c, int x, int y, float (&out)[6]) {
    float xs = ((float)x + 0.5f - c.cx) / c.fx;
    float ys = ((float)y + 0.5f - c.cy) / c.fy;
    float nrm = sqrtf(xs * xs + ys * ys + 1.0f);
    float d0 = xs / nrm, d1 = ys / nrm, d2 = 1.0f / nrm;
    float w[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) w[a] = d0 * c.R[a][0] + d1 * c.R[a][1] + d2 * c.R[a][2];
    out[0] = c.t[1] * w[2] - c.t[2] * w[1];
    out[1] = c.t[2] * w[0] - c.t[0] * w[2];
    out[2] = c.t[0] * w[1] - c.t[1] * w[0];
    out[3] = w[0]; out[4] = w[1]; out[5] = w[2];
}

template <typename T, int LAYOUT>
__global__ void plucker_planar_kernel(const float* __restrict__ K, const float* __restrict__ c2w, T* __restrict__ out,
                                      int B, int F, int H, int W, int c2w_rows) {
    const int bf = blockIdx.y;
    const Cam cam = load_cam(K, c2w, bf, c2w_rows);
    const int HW = H * W;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
        int y = p / W, x = p - y * W;
        float v[6];
        plucker_pixel(cam, x, y, v);
        if (LAYOUT == 0) {
            T* o = out + ((size_t)bf * HW + p) * 6;
#pragma unroll
            for (int c = 0; c < 6; ++c) o[c] = cvt_out<T>(v[c]);
        } else {  // [B, 6, F, H, W]
            int b = bf / F, f = bf - b * F;
#pragma unroll
            for (int c = 0; c < 6; ++c) out[(((size_t)b * 6 + c) * F + f) * HW + p] = cvt_out<T>(v[c]);
        }
    }
}

// layout 2: [BF, H/8, W/8, 384], channel = c*64 + dy*8 + dx.  One thread = one (block, dy) row of 8 pixels.
template <typename T>
__global__ void plucker_unshuffle_kernel(const float* __restrict__ K, const float* __restrict__ c2w,
                                         T* __restrict__ out, int H, int W, int c2w_rows) {
    const int bf = blockIdx.y;
    const Cam cam = load_cam(K, c2w, bf, c2w_rows);
    const int hb = H / 8, wb = W / 8;
    const int total = hb * wb * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int dy = idx & 7, blk = idx >> 3;
        int by = blk / wb, bx = blk - by * wb;
        float v[8][6];
#pragma unroll
        for (int dx = 0; dx < 8; ++dx) plucker_pixel(cam, bx * 8 + dx, by * 8 + dy, v[dx]);
        T* o = out + ((size_t)bf * hb * wb + blk) * 384 + dy * 8;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            float r[8];
#pragma unroll
            for (int dx = 0; dx < 8; ++dx) r[dx] = v[dx][c];
            Vec8<T>::store(o + c * 64, r);
        }
    }
}

// ---- OMC rasteriser ---------------------------------------------------------------------------
// last object with mask > 0 wins (fmc/util.py:173-183); then everything * mask (util.py:201).
__device__ __forceinline__ void raster_pixel(const float* __restrict__ poses, const float* __restrict__ masks,
                                             int bf, int n_obj, size_t HW, size_t p, float (&feat)[13], float& m_out) {
    int win = -1;
    float m = 0.f;
    for (int o = 0; o < n_obj; ++o) {
        float mo = masks[((size_t)bf * n_obj + o) * HW + p];
        if (mo > 0.f) { win = o; m = mo; }
    }
    m_out = m;
    if (win < 0) {
#pragma unroll
        for (int c = 0; c < 13; ++c) feat[c] = 0.f;
        return;
    }
    const float* ps = poses + ((size_t)bf * n_obj + win) * 12;
#pragma unroll
    for (int c = 0; c < 12; ++c) feat[c] = (ps[c] * m) * m;
    feat[12] = m * m;
}

template <typename T>
__global__ void raster_planar_kernel(const float* __restrict__ poses, const float* __restrict__ masks,
                                     T* __restrict__ feat, float* __restrict__ mask_out, int n_obj, int H, int W) {
    const int bf = blockIdx.y;
    const size_t HW = (size_t)H * W;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (size_t)gridDim.x * blockDim.x) {
        float v[13], m;
        raster_pixel(poses, masks, bf, n_obj, HW, p, v, m);
#pragma unroll
        for (int c = 0; c < 13; ++c) feat[((size_t)bf * 13 + c) * HW + p] = cvt_out<T>(v[c]);
        mask_out[(size_t)bf * HW + p] = m;
    }
}

template <typename T>
__global__ void raster_unshuffle_kernel(const float* __restrict__ poses, const float* __restrict__ masks,
                                        T* __restrict__ feat, float* __restrict__ mask_out, int n_obj, int H, int W) {
    const int bf = blockIdx.y;
    const size_t HW = (size_t)H * W;
    const int hb = H / 8, wb = W / 8;
    const int total = hb * wb * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int dy = idx & 7, blk = idx >> 3;
        int by = blk / wb, bx = blk - by * wb;
        float v[8][13], m[8];
#pragma unroll
        for (int dx = 0; dx < 8; ++dx) {
            size_t p = (size_t)(by * 8 + dy) * W + bx * 8 + dx;
            raster_pixel(poses, masks, bf, n_obj, HW, p, v[dx], m[dx]);
        }
        Vec8<float>::store(mask_out + (size_t)bf * HW + (size_t)(by * 8 + dy) * W + bx * 8, m);
        T* o = feat + ((size_t)bf * hb * wb + blk) * 832 + dy * 8;
#pragma unroll
        for (int c = 0; c < 13; ++c) {
            float r[8];
#pragma unroll
            for (int dx = 0; dx < 8; ++dx) r[dx] = v[dx][c];
            Vec8<T>::store(o + c * 64, r);
        }
    }
}

// ---- mask modulation --------------------------------------------------------------------------
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

template <typename T>
__global__ void mask_modulate_kernel(const T* __restrict__ x, const float* __restrict__ mask_in, T* __restrict__ y,
                                     float* __restrict__ mask_out, int N, int h, int w, int C, int Hin, int Win) {
    const int cpr = C / 8;
    const int64_t total = (int64_t)N * h * w * cpr;
    const float sh = (float)Hin / (float)h, sw = (float)Win / (float)w;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t pix = idx / cpr;
        int cc = (int)(idx - pix * cpr);
        int j = (int)(pix % w);
        int64_t t = pix / w;
        int i = (int)(t % h);
        int n = (int)(t / h);
        float m = mask_in[((size_t)n * Hin + nearest_src(i, sh, Hin)) * Win + nearest_src(j, sw, Win)];
        float v[8];
        Vec8<T>::load(x + pix * C + cc * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= m;
        Vec8<T>::store(y + pix * C + cc * 8, v);
        if (mask_out && cc == 0) mask_out[pix] = m;
    }
}

// ---- OMC feature add --------------------------------------------------------------------------
template <typename T>
__global__ void feature_add_kernel(const T* __restrict__ h, const T* __restrict__ t, T* __restrict__ out,
                                   int64_t n_chunks, int64_t skip_chunks, bool copy_skipped) {
    const int64_t start = copy_skipped ? 0 : skip_chunks;
    for (int64_t idx = start + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_chunks;
         idx += (int64_t)gridDim.x * blockDim.x) {
        float a[8];
        Vec8<T>::load(h + idx * 8, a);
        if (idx >= skip_chunks) {
            float b[8];
            Vec8<T>::load(t + (idx - skip_chunks) * 8, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += b[k];
        }
        Vec8<T>::store(out + idx * 8, a);
    }
}

// ---- CFG + DDIM -------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf2f(*p); }

template <typename T>
__global__ void cfg_ddim_kernel(const T* __restrict__ eps_uc, const float* __restrict__ x, float* __restrict__ x_out,
                                int64_t n, int has_uncond, float g, float sa_t, float s1a_t, float sa_p, float s1a_p) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float e;
        if (has_uncond) {
            float eu = ld1<T>(eps_uc + i), ec = ld1<T>(eps_uc + n + i);
            e = eu + g * (ec - eu);
        } else {
            e = ld1<T>(eps_uc + i);
        }
        float x0 = (x[i] - s1a_t * e) / sa_t;
        x_out[i] = sa_p * x0 + s1a_p * e;
    }
}

inline unsigned grid_for(int64_t work, int block, int cap) {
    int64_t b = (work + block - 1) / block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int fmc_plucker_fwd(const float* K, const float* c2w, void* out, int B, int F, int H, int W, int c2w_rows,
                               int layout, int dtype, void* stream) {
    if (!K || !c2w || !out) FMC_FAIL(FMC_E_NULL, "plucker: NULL argument");
    if (B <= 0 || F <= 0 || H <= 0 || W <= 0 || (c2w_rows != 3 && c2w_rows != 4))
        FMC_FAIL(FMC_E_SHAPE, "plucker: bad shape B=%d F=%d H=%d W=%d c2w_rows=%d", B, F, H, W, c2w_rows);
    if (layout < 0 || layout > 2) FMC_FAIL(FMC_E_SHAPE, "plucker: layout %d", layout);
    if (layout == 2 && (H % 8 || W % 8)) FMC_FAIL(FMC_E_SHAPE, "plucker: layout 2 needs H,W %% 8 == 0");
    if (dtype != FMC_BF16 && dtype != FMC_F32) FMC_FAIL(FMC_E_DTYPE, "plucker: dtype %d", dtype);
    if (layout == 2 && !fmc_aligned16(out)) FMC_FAIL(FMC_E_ALIGN, "plucker: out must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int BF = B * F;
    if (layout == 2) {
        dim3 grid(grid_for((int64_t)H * W / 8, 256, 1024), BF), block(256);
        if (dtype == FMC_BF16)
            hipLaunchKernelGGL((plucker_unshuffle_kernel<bf16_t>), grid, block, 0, st, K, c2w, (bf16_t*)out, H, W, c2w_rows);
        else
            hipLaunchKernelGGL((plucker_unshuffle_kernel<float>), grid, block, 0, st, K, c2w, (float*)out, H, W, c2w_rows);
    } else {
        dim3 grid(grid_for((int64_t)H * W, 256, 1024), BF), block(256);
#define PL(T, L) hipLaunchKernelGGL((plucker_planar_kernel<T, L>), grid, block, 0, st, K, c2w, (T*)out, B, F, H, W, c2w_rows)
        if (dtype == FMC_BF16) { if (layout == 0) PL(bf16_t, 0); else PL(bf16_t, 1); }
        else { if (layout == 0) PL(float, 0); else PL(float, 1); }
#undef PL
    }
    FMC_CHECK_LAUNCH("fmc_plucker_fwd");
    return 0;
}


namespace {
// ---- Gaussian circle masks (fmc/data/dataset.py:5365-5380, the analytic part: cv2.minEnclosingCircle stays on the host)
// circles [N,3] = (cx, cy, radius) in pixels.  mask = [ (x-int(cx))^2 + (y-int(cy))^2 <= int(r)^2 ] * g / max(g) with
// g = exp(-d^2 / (2 (r/2)^2)), d the distance to the float centre; max(g) is attained at the pixel nearest the centre,
// so g / max(g) = exp(-(d^2 - dmin^2) * 2 / r^2) without a reduction.  HBM bound: one 4-byte store per pixel.
__global__ void gaussian_circle_mask_kernel(const float* __restrict__ circles, float* __restrict__ out, int H, int W) {
    const int n = blockIdx.y;
    const float cx = circles[n * 3], cy = circles[n * 3 + 1], r = circles[n * 3 + 2];
    const float nx = fminf(fmaxf(rintf(cx), 0.f), (float)(W - 1)), ny = fminf(fmaxf(rintf(cy), 0.f), (float)(H - 1));
    const float dmin2 = (nx - cx) * (nx - cx) + (ny - cy) * (ny - cy);
    const float k = -2.f / (r * r) * 1.4426950408889634f;
    const int icx = (int)cx, icy = (int)cy, ir = (int)r;
    float* o = out + (int64_t)n * H * W;
    const int total = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const float dx = (float)x - cx, dy = (float)y - cy;
        const bool inside = (x - icx) * (x - icx) + (y - icy) * (y - icy) <= ir * ir;
        o[i] = inside ? exp2f((dx * dx + dy * dy - dmin2) * k) : 0.f;
    }
}
}  // namespace

extern "C" int fmc_omc_rasterize_fwd(const float* poses, const float* masks, void* feat, float* mask_out, int BF,
                                     int n_obj, int H, int W, int layout, int dtype, void* stream) {
    if (!poses || !masks || !feat || !mask_out) FMC_FAIL(FMC_E_NULL, "omc_rasterize: NULL argument");
    if (BF <= 0 || n_obj < 0 || H <= 0 || W <= 0) FMC_FAIL(FMC_E_SHAPE, "omc_rasterize: bad shape");
    if (layout != 0 && layout != 2) FMC_FAIL(FMC_E_SHAPE, "omc_rasterize: layout %d", layout);
    if (layout == 2 && (H % 8 || W % 8)) FMC_FAIL(FMC_E_SHAPE, "omc_rasterize: layout 2 needs H,W %% 8 == 0");
    if (dtype != FMC_BF16 && dtype != FMC_F32) FMC_FAIL(FMC_E_DTYPE, "omc_rasterize: dtype %d", dtype);
    if (layout == 2 && (!fmc_aligned16(feat) || !fmc_aligned16(mask_out)))
        FMC_FAIL(FMC_E_ALIGN, "omc_rasterize: outputs must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (layout == 2) {
        dim3 grid(grid_for((int64_t)H * W / 8, 256, 1024), BF), block(256);
        if (dtype == FMC_BF16)
            hipLaunchKernelGGL((raster_unshuffle_kernel<bf16_t>), grid, block, 0, st, poses, masks, (bf16_t*)feat, mask_out, n_obj, H, W);
        else
            hipLaunchKernelGGL((raster_unshuffle_kernel<float>), grid, block, 0, st, poses, masks, (float*)feat, mask_out, n_obj, H, W);
    } else {
        dim3 grid(grid_for((int64_t)H * W, 256, 1024), BF), block(256);
        if (dtype == FMC_BF16)
            hipLaunchKernelGGL((raster_planar_kernel<bf16_t>), grid, block, 0, st, poses, masks, (bf16_t*)feat, mask_out, n_obj, H, W);
        else
            hipLaunchKernelGGL((raster_planar_kernel<float>), grid, block, 0, st, poses, masks, (float*)feat, mask_out, n_obj, H, W);
    }
    FMC_CHECK_LAUNCH("fmc_omc_rasterize_fwd");
    return 0;
}

extern "C" int fmc_gaussian_circle_mask_fwd(const float* circles, float* out, int N, int H, int W, void* stream) {
    if (!circles || !out) FMC_FAIL(FMC_E_NULL, "gaussian_circle_mask: NULL argument");
    if (N <= 0 || H <= 0 || W <= 0 || N > 65535) FMC_FAIL(FMC_E_SHAPE, "gaussian_circle_mask: bad shape N=%d H=%d W=%d", N, H, W);
    dim3 grid(grid_for((int64_t)H * W, 256, 256), N), block(256);
    hipLaunchKernelGGL(gaussian_circle_mask_kernel, grid, block, 0, (hipStream_t)stream, circles, out, H, W);
    FMC_CHECK_LAUNCH("fmc_gaussian_circle_mask_fwd");
    return 0;
}

extern "C" int fmc_mask_modulate_fwd(const void* x, const float* mask_in, void* y, float* mask_out, int N, int h, int w,
                                     int C, int Hin, int Win, int dtype, void* stream) {
    if (!x || !mask_in || !y) FMC_FAIL(FMC_E_NULL, "mask_modulate: NULL argument");
    if (N <= 0 || h <= 0 || w <= 0 || C <= 0 || C % 8 || Hin <= 0 || Win <= 0)
        FMC_FAIL(FMC_E_SHAPE, "mask_modulate: need C%%8==0 and positive sizes (C=%d)", C);
    if (!fmc_aligned16(x) || !fmc_aligned16(y)) FMC_FAIL(FMC_E_ALIGN, "mask_modulate: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(grid_for((int64_t)N * h * w * (C / 8), 256, 4096)), block(256);
    if (dtype == FMC_BF16)
        hipLaunchKernelGGL((mask_modulate_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x, mask_in, (bf16_t*)y, mask_out, N, h, w, C, Hin, Win);
    else if (dtype == FMC_F32)
        hipLaunchKernelGGL((mask_modulate_kernel<float>), grid, block, 0, st, (const float*)x, mask_in, (float*)y, mask_out, N, h, w, C, Hin, Win);
    else
        FMC_FAIL(FMC_E_DTYPE, "mask_modulate: dtype %d", dtype);
    FMC_CHECK_LAUNCH("fmc_mask_modulate_fwd");
    return 0;
}

extern "C" int fmc_feature_add_fwd(const void* h, const void* t, void* out, int64_t n_elems, int64_t skip_elems,
                                   int dtype, void* stream) {
    if (!h || !t || !out) FMC_FAIL(FMC_E_NULL, "feature_add: NULL argument");
    if (n_elems <= 0 || skip_elems < 0 || skip_elems > n_elems || n_elems % 8 || skip_elems % 8)
        FMC_FAIL(FMC_E_SHAPE, "feature_add: element counts must be multiples of 8");
    if (!fmc_aligned16(h) || !fmc_aligned16(t) || !fmc_aligned16(out))
        FMC_FAIL(FMC_E_ALIGN, "feature_add: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const bool copy_skipped = (h != out);
    const int64_t work = (copy_skipped ? n_elems : n_elems - skip_elems) / 8;
    if (work == 0) return 0;
    dim3 grid(grid_for(work, 256, 4096)), block(256);
    if (dtype == FMC_BF16)
        hipLaunchKernelGGL((feature_add_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)h, (const bf16_t*)t, (bf16_t*)out, n_elems / 8, skip_elems / 8, copy_skipped);
    else if (dtype == FMC_F32)
        hipLaunchKernelGGL((feature_add_kernel<float>), grid, block, 0, st, (const float*)h, (const float*)t, (float*)out, n_elems / 8, skip_elems / 8, copy_skipped);
    else
        FMC_FAIL(FMC_E_DTYPE, "feature_add: dtype %d", dtype);
    FMC_CHECK_LAUNCH("fmc_feature_add_fwd");
    return 0;
}

extern "C" int fmc_cfg_ddim_step(const void* eps_uc, const float* x, float* x_out, int64_t n, int has_uncond,
                                 float guidance, float alpha_t, float alpha_prev, int dtype, void* stream) {
    if (!eps_uc || !x || !x_out) FMC_FAIL(FMC_E_NULL, "cfg_ddim_step: NULL argument");
    if (n <= 0 || alpha_t <= 0.f || alpha_t > 1.f || alpha_prev <= 0.f || alpha_prev > 1.f)
        FMC_FAIL(FMC_E_SHAPE, "cfg_ddim_step: bad n / alphas");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(grid_for(n, 256, 2048)), block(256);
    const float sa_t = sqrtf(alpha_t), s1a_t = sqrtf(1.f - alpha_t);
    const float sa_p = sqrtf(alpha_prev), s1a_p = sqrtf(1.f - alpha_prev);
    if (dtype == FMC_BF16)
        hipLaunchKernelGGL((cfg_ddim_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)eps_uc, x, x_out, n, has_uncond, guidance, sa_t, s1a_t, sa_p, s1a_p);
    else if (dtype == FMC_F32)
        hipLaunchKernelGGL((cfg_ddim_kernel<float>), grid, block, 0, st, (const float*)eps_uc, x, x_out, n, has_uncond, guidance, sa_t, s1a_t, sa_p, s1a_p);
    else
        FMC_FAIL(FMC_E_DTYPE, "cfg_ddim_step: dtype %d", dtype);
    FMC_CHECK_LAUNCH("fmc_cfg_ddim_step");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 convolutions (training stages 2-3: OMC Adapter, camera encoder; train_cam_obj_ctrl.py:861-943).
//   dW[co][dy][dx][ci] = sum over (img, y, x) of dY[img, y, x, co] * X[img, y + dy - 1, x + dx - 1, ci]
// is a GEMM whose REDUCTION index is the pixel -- the slow index of both NHWC operands.  fmc_linear_bf16 wants both operands
// reduction-contiguous, so this kernel re-lays them out pixel-minor over a zero-PADDED pixel index
//   p = (img * Hp + yp) * Wp + xp,  Hp = H + 2, Wp = round_up(W + 2, 8),  (yp, xp) = (y + 1, x + 1)
//   dst[s][c][G + p] = src[img][yp - 1][xp - 1 + (s - S/2)]   (0 outside the image / beyond the last image)
// with S = 3 copies shifted by dx = -1, 0, +1 for X (S = 1 for dY) and G guard elements in front of every row.  A vertical
// tap offset dy is then the POINTER offset dy * Wp (a multiple of 8 elements = 16 bytes: DMA-aligned), the rows (dx, ci) of
// the three copies have one uniform stride, and the padding zeros of dY^T switch off every product that would wrap across
// an image border.  dW for one dy = ONE fmc_linear_bf16 call [3 Cin x Lk] x [Cout x Lk]^T (split-K over the pixels).
template <int TP>
__global__ __launch_bounds__(256) void nhwc_to_cmajor_padded_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int n_img,
                                                                    int H, int W, int C, int Hp, int Wp, int64_t row_len, int guard,
                                                                    int shifts) {
    __shared__ bf16_t tile[64][64 + 2];              // [position][channel]
    const int s = blockIdx.z, c0 = blockIdx.y * 64;
    const int64_t p0 = (int64_t)blockIdx.x * 64;     // first row position (guard included) of this tile
    const int dx = s - shifts / 2;
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pos = (tid >> 3) + 32 * it, ch = (tid & 7) * 8;
        const int64_t p = p0 + pos - guard;          // padded pixel index
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (p >= 0 && c0 + ch < C) {
            const int64_t img = p / ((int64_t)Hp * Wp);
            const int r = (int)(p - img * Hp * Wp), yp = r / Wp, xp = r - yp * Wp;
            const int y = yp - 1, x = xp - 1 + dx;
            if (img < n_img && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && xp - 1 >= -1 && xp - 1 <= W)
                v = *reinterpret_cast<const u32x4*>(src + (((int64_t)img * H + y) * W + x) * C + c0 + ch);
        }
        union { u32x4 u; bf16_t e[8]; } t;
        t.u = v;
#pragma unroll
        for (int i = 0; i < 8; ++i) tile[pos][ch + i] = t.e[i];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (tid >> 3) + 32 * it, pc = (tid & 7) * 8;
        if (c0 + c >= C || p0 + pc >= row_len) continue;
        union { u32x4 u; bf16_t e[8]; } t;
#pragma unroll
        for (int i = 0; i < 8; ++i) t.e[i] = tile[pc + i][c];
        *reinterpret_cast<u32x4*>(dst + ((int64_t)s * C + c0 + c) * row_len + p0 + pc) = t.u;
    }
}

extern "C" int fmc_nhwc_to_cmajor_padded(const void* src, void* dst, int n_img, int H, int W, int C, int64_t row_len, int guard,
                                         int shifts, void* stream) {
    if (!src || !dst) FMC_FAIL(FMC_E_NULL, "nhwc_to_cmajor_padded: NULL tensor");
    const int Hp = H + 2, Wp = (W + 2 + 7) / 8 * 8;
    if (n_img <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || row_len % 8 || guard % 8 || guard < 0 || (shifts != 1 && shifts != 3) ||
        row_len < guard + (int64_t)n_img * Hp * Wp)
        FMC_FAIL(FMC_E_SHAPE, "nhwc_to_cmajor_padded: need C%%8==0, row_len%%8==0, guard%%8==0, shifts in {1,3}, row_len >= guard + n*Hp*Wp");
    if (!fmc_aligned16(src) || !fmc_aligned16(dst)) FMC_FAIL(FMC_E_ALIGN, "nhwc_to_cmajor_padded: tensors must be 16-byte aligned");
    dim3 grid((unsigned)((row_len + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)shifts);
    hipLaunchKernelGGL((nhwc_to_cmajor_padded_kernel<64>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst,
                       n_img, H, W, C, Hp, Wp, row_len, guard, shifts);
    FMC_CHECK_LAUNCH("fmc_nhwc_to_cmajor_padded");
    return 0;
}
