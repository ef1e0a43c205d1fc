// The vendor arm of the token projections, called directly: out = x W^T + bias + residual as ONE hipBLASLt launch.
//
// The autotuner's arm 0 used to be torch's F.linear (hipBLASLt with the bias epilogue) followed, for the projections that end a residual branch
// (attention to_out, feed-forward output, proj_out: fmc/models/attention_processor.py:69, diffusers attention.py FeedForward, transformer_2d proj_out),
// by a separate torch add -- 42 elementwise launches per denoising step at the 10x16 / 5x8 levels (0.25 ms), where no own kernel beats the library.
// hipBLASLt computes D = A B + beta C + bias natively; torch's addmm does not expose C and bias together.  Same library, same Tensile kernels
// (the process binds to the libhipblaslt.so.1 torch has already loaded), one launch less per projection; `algo` indexes the heuristic's candidate
// list, so the per-shape autotuner can also try the runners-up.
//
// Row-major out[M][N] = x[M][K] w[N][K]^T is the column-major product D(N x M) = w^T(N x K) x(K x M): transA = T (lda = K), transB = N (ldb = ldx);
// the bias vector runs along D's rows (= output features).
//
// Ownership (round 6; the header's convention "the caller owns every buffer" holds here too): the library handle lives between fmc_vendor_init() and
// fmc_vendor_destroy() of a device, the split-K / stream-K scratch is the CALLER's (`workspace`, `workspace_bytes` per call -- one buffer per stream on the
// host side, so two streams never share scratch); nothing in this file allocates device memory.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.h"

namespace {

constexpr size_t VG_WORKSPACE = 64u << 20;          // the workspace size candidates are planned for (fmc_vendor_workspace_bytes)
constexpr int VG_MAX_ALGOS = 8;

struct VgDevice {
    hipblasLtHandle_t handle = nullptr;
};

struct VgPlan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
    std::vector<hipblasLtMatmulHeuristicResult_t> algos;
};

typedef std::tuple<int, int64_t, int, int, int64_t, int64_t, int64_t, int, int> VgKey;   // device, M, N, K, ldx, ldres, ldo, bias?, residual?

void vg_free_plan(VgPlan& p) {
    if (p.a) (void)hipblasLtMatrixLayoutDestroy(p.a);
    if (p.b) (void)hipblasLtMatrixLayoutDestroy(p.b);
    if (p.c) (void)hipblasLtMatrixLayoutDestroy(p.c);
    if (p.d) (void)hipblasLtMatrixLayoutDestroy(p.d);
    if (p.desc) (void)hipblasLtMatmulDescDestroy(p.desc);
    p = VgPlan();
}

std::mutex g_mu;
VgDevice g_dev[64];
std::map<VgKey, VgPlan> g_plans;

#define VG_CHECK(expr, what)                                                                                   \
    do {                                                                                                       \
        const hipblasStatus_t st__ = (expr);                                                                   \
        if (st__ != HIPBLAS_STATUS_SUCCESS) FMC_FAIL(FMC_E_LAUNCH, "vendor_linear_bf16: %s failed (hipblasStatus %d)", what, (int)st__); \
    } while (0)

int vg_device(VgDevice*& out) {
    VgDevice& d = g_dev[fmc_device() & 63];
    if (!d.handle) FMC_FAIL(FMC_E_NULL, "vendor_linear_bf16: fmc_vendor_init() has not been called on device %d", fmc_device());
    out = &d;
    return 0;
}

int vg_plan(VgDevice& dev, const VgKey& key, int64_t M, int N, int K, int64_t ldx, int64_t ldres, int64_t ldo, bool has_bias, bool has_res, VgPlan*& out) {
    auto it = g_plans.find(key);
    if (it != g_plans.end()) {
        out = &it->second;
        return 0;
    }
    VgPlan p;
    VG_CHECK(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F), "MatmulDescCreate");
    const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    VG_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)), "TRANSA");
    VG_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)), "TRANSB");
    if (has_bias) {
        const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
        const int32_t bt = (int32_t)HIP_R_16BF;
        VG_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)), "EPILOGUE");
        VG_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)), "BIAS_DATA_TYPE");
    }
    // A = w: stored [N][K] row-major = column-major K x N with lda = K (op T -> N x K); B = x: column-major K x M, ldb = ldx
    VG_CHECK(hipblasLtMatrixLayoutCreate(&p.a, HIP_R_16BF, (uint64_t)K, (uint64_t)N, K), "layout A");
    VG_CHECK(hipblasLtMatrixLayoutCreate(&p.b, HIP_R_16BF, (uint64_t)K, (uint64_t)M, ldx), "layout B");
    VG_CHECK(hipblasLtMatrixLayoutCreate(&p.c, HIP_R_16BF, (uint64_t)N, (uint64_t)M, has_res ? ldres : ldo), "layout C");
    VG_CHECK(hipblasLtMatrixLayoutCreate(&p.d, HIP_R_16BF, (uint64_t)N, (uint64_t)M, ldo), "layout D");
    hipblasLtMatmulPreference_t pref = nullptr;
    VG_CHECK(hipblasLtMatmulPreferenceCreate(&pref), "PreferenceCreate");
    const uint64_t ws = VG_WORKSPACE;
    VG_CHECK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws)), "MAX_WORKSPACE_BYTES");
    hipblasLtMatmulHeuristicResult_t res[VG_MAX_ALGOS];
    int found = 0;
    const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(dev.handle, p.desc, p.a, p.b, p.c, p.d, pref, VG_MAX_ALGOS, res, &found);
    (void)hipblasLtMatmulPreferenceDestroy(pref);
    if (st != HIPBLAS_STATUS_SUCCESS || found <= 0)
        FMC_FAIL(FMC_E_SHAPE, "vendor_linear_bf16: hipBLASLt offers no kernel for M=%lld N=%d K=%d (status %d)", (long long)M, N, K, (int)st);
    for (int i = 0; i < found; ++i)
        if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= VG_WORKSPACE) p.algos.push_back(res[i]);
    if (p.algos.empty()) FMC_FAIL(FMC_E_SHAPE, "vendor_linear_bf16: no usable hipBLASLt candidate for M=%lld N=%d K=%d", (long long)M, N, K);
    out = &(g_plans[key] = p);
    return 0;
}

}  // namespace

extern "C" int64_t fmc_vendor_workspace_bytes(void) { return (int64_t)VG_WORKSPACE; }

// hipBLASLt's version number (candidate indices of one library version mean other kernels in another: the arm table records it)
extern "C" int fmc_vendor_version(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    VgDevice& d = g_dev[fmc_device() & 63];
    if (!d.handle) FMC_FAIL(FMC_E_NULL, "vendor_version: fmc_vendor_init() has not been called on device %d", fmc_device());
    int v = 0;
    VG_CHECK(hipblasLtGetVersion(d.handle, &v), "hipblasLtGetVersion");
    return v;
}

// library handle of the current device (idempotent); no device memory is allocated
extern "C" int fmc_vendor_init(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    VgDevice& d = g_dev[fmc_device() & 63];
    if (!d.handle) VG_CHECK(hipblasLtCreate(&d.handle), "hipblasLtCreate");
    return 0;
}

// drops the current device's plans and its handle (idempotent)
extern "C" int fmc_vendor_destroy(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    const int dev = fmc_device();
    for (auto it = g_plans.begin(); it != g_plans.end();) {
        if (std::get<0>(it->first) == dev) {
            vg_free_plan(it->second);
            it = g_plans.erase(it);
        } else {
            ++it;
        }
    }
    VgDevice& d = g_dev[dev & 63];
    if (d.handle) {
        (void)hipblasLtDestroy(d.handle);
        d.handle = nullptr;
    }
    return 0;
}

// number of heuristic candidates for this problem (>= 1), or a negative error code
extern "C" int fmc_vendor_linear_candidates(int64_t M, int N, int K, int64_t ldx, int64_t ldres, int64_t ldo, int has_bias, int has_residual) {
    if (M <= 0 || N <= 0 || K <= 0) FMC_FAIL(FMC_E_SHAPE, "vendor_linear_candidates: M=%lld N=%d K=%d", (long long)M, N, K);
    std::lock_guard<std::mutex> lock(g_mu);
    VgDevice* dev = nullptr;
    if (int rc = vg_device(dev)) return rc;
    VgPlan* plan = nullptr;
    const VgKey key(fmc_device(), M, N, K, ldx, has_residual ? ldres : 0, ldo, has_bias != 0, has_residual != 0);
    if (int rc = vg_plan(*dev, key, M, N, K, ldx, ldres, ldo, has_bias != 0, has_residual != 0, plan)) return rc;
    return (int)plan->algos.size();
}

extern "C" int fmc_vendor_linear_bf16(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M, int N, int K,
                                      int64_t ldx, int64_t ldres, int64_t ldo, int algo, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!x || !w || !out) FMC_FAIL(FMC_E_NULL, "vendor_linear_bf16: NULL tensor");
    if (M <= 0 || N <= 0 || K <= 0 || ldx < K || ldo < N || (residual && ldres < N))
        FMC_FAIL(FMC_E_SHAPE, "vendor_linear_bf16: M=%lld N=%d K=%d ldx=%lld ldres=%lld ldo=%lld", (long long)M, N, K, (long long)ldx, (long long)ldres, (long long)ldo);
    if (!fmc_aligned16(x) || !fmc_aligned16(w) || !fmc_aligned16(out) || (residual && !fmc_aligned16(residual)) || (bias && ((uintptr_t)bias & 1)))
        FMC_FAIL(FMC_E_ALIGN, "vendor_linear_bf16: tensors must be 16-byte aligned");
    std::lock_guard<std::mutex> lock(g_mu);
    VgDevice* dev = nullptr;
    if (int rc = vg_device(dev)) return rc;
    VgPlan* plan = nullptr;
    const VgKey key(fmc_device(), M, N, K, ldx, residual ? ldres : 0, ldo, bias != nullptr, residual != nullptr);
    if (int rc = vg_plan(*dev, key, M, N, K, ldx, ldres, ldo, bias != nullptr, residual != nullptr, plan)) return rc;
    if (algo < 0 || algo >= (int)plan->algos.size())
        FMC_FAIL(FMC_E_SHAPE, "vendor_linear_bf16: candidate %d of %d", algo, (int)plan->algos.size());
    const size_t need = plan->algos[algo].workspaceSize;
    if (need && (!workspace || workspace_bytes < (int64_t)need || !fmc_aligned16(workspace)))
        FMC_FAIL(FMC_E_NULL, "vendor_linear_bf16: candidate %d needs %zu bytes of 16-byte aligned workspace, got %lld", algo, need, (long long)workspace_bytes);
    if (bias) VG_CHECK(hipblasLtMatmulDescSetAttribute(plan->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)), "BIAS_POINTER");
    const float alpha = 1.f, beta = residual ? 1.f : 0.f;
    VG_CHECK(hipblasLtMatmul(dev->handle, plan->desc, &alpha, w, plan->a, x, plan->b, &beta, residual ? residual : out, plan->c, out, plan->d,
                             &plan->algos[algo].algo, need ? workspace : nullptr, need ? (size_t)workspace_bytes : 0, (hipStream_t)stream),
             "hipblasLtMatmul");
    return 0;
}
