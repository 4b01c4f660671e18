// axw.cpp -- libflex_axw.so: Out = A * X * W around the engine's SpMM (include/flex_axw.h).
// ≙ run1 / run2 (cusp.cu:3-104, 106-208) with cusparseSpMM replaced by flex_spmm, cuBLAS by rocBLAS
// and row-major dense operands throughout.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <new>

#include "../../include/flex_axw.h"

extern "C" hipError_t flex_axw_gemm_launch(const float *L, const float *Wp, float *Out, int n, int dim, int cp, int n_cus,
                                           hipStream_t s);  // axw_kernels.hip

struct flex_axw {
    int32_t n = 0;
    int n_cus = 0;
    bool use_blas = false;  // FLEX_AXW_USE_BLAS in `flags`, or a shape the MFMA kernel does not take (dim % 4 != 0, dim > 256, n < 32)
    int dim = 0, c = 0, cp = 0, device = 0;
    flex_plan *plan_c = nullptr, *plan_dim = nullptr;
    float *d_xw = nullptr;  // n x cp
    float *d_ax = nullptr;  // n x dim
    float *d_wp = nullptr;  // dim x cp (W with zero columns appended)
    rocblas_handle blas = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};

static thread_local int g_blas_status = 0;

namespace {

int hip_fail(hipError_t e) { return e == hipSuccess ? FLEX_OK : (e == hipErrorOutOfMemory ? FLEX_ERR_NOMEM : FLEX_ERR_HIP); }

// C_rm[n x cp] = L_rm[n x dim] * Wp_rm[dim x cp]   <=>   column-major  C^T = Wp^T * L^T
int gemm_rm(flex_axw *h, const float *L, float *Cout, hipStream_t s) {
    if (!h->use_blas) return hip_fail(flex_axw_gemm_launch(L, h->d_wp, Cout, h->n, h->dim, h->cp, h->n_cus, s));
    const float one = 1.0f, zero = 0.0f;
    const rocblas_status st = rocblas_sgemm(h->blas, rocblas_operation_none, rocblas_operation_none, h->cp, h->n, h->dim, &one,
                                            h->d_wp, h->cp, L, h->dim, &zero, Cout, h->cp);
    if (st != rocblas_status_success) {
        g_blas_status = static_cast<int>(st);
        return FLEX_ERR_UNSUPPORTED;
    }
    return FLEX_OK;
}

}  // namespace

extern "C" {

// Whole 128-byte lines per row: a B row that starts mid-line costs every gather one extra line
// (measured, reddit shape: k=100 999 us, k=104 973 us, k=96 646 us, k=128 669 us -- DESIGN.md 3.3).
int flex_axw_ld(int c) { return c <= 0 ? 0 : (c + 31) / 32 * 32; }
int flex_axw_last_blas_status(void) { return g_blas_status; }

int flex_axw_destroy(flex_axw *h) {
    if (!h) return FLEX_OK;
    int cur = -1;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(h->device);
    flex_plan_destroy(h->plan_c);
    flex_plan_destroy(h->plan_dim);
    (void)hipFree(h->d_xw);
    (void)hipFree(h->d_ax);
    (void)hipFree(h->d_wp);
    if (h->blas) rocblas_destroy_handle(h->blas);
    for (hipEvent_t e : h->ev)
        if (e) (void)hipEventDestroy(e);
    if (cur >= 0) (void)hipSetDevice(cur);
    delete h;
    return FLEX_OK;
}

int flex_axw_create(flex_axw **out, const flex_csr *A, int dim, int c, int device, unsigned flags) {
    if (!out) return FLEX_ERR_INVALID;
    *out = nullptr;
    if (!A || dim <= 0 || c <= 0 || device < 0 || A->m != A->n) return FLEX_ERR_INVALID;
    flex_axw *h = new (std::nothrow) flex_axw();
    if (!h) return FLEX_ERR_NOMEM;
    h->n = A->n;
    h->dim = dim;
    h->c = c;
    h->cp = flex_axw_ld(c);
    h->device = device;
    int prev = -1;
    (void)hipGetDevice(&prev);
    int rc = hip_fail(hipSetDevice(device));
    const bool want_blas = (flags & FLEX_AXW_USE_BLAS) != 0;
    flags &= ~FLEX_AXW_USE_BLAS;
    if (!rc) rc = flex_plan_create(&h->plan_c, A, h->cp, device, flags);
    if (!rc) rc = flex_plan_create(&h->plan_dim, A, dim, device, flags);
    const size_t n1 = static_cast<size_t>(h->n > 0 ? h->n : 1);
    if (!rc) rc = hip_fail(hipMalloc(reinterpret_cast<void **>(&h->d_xw), n1 * h->cp * sizeof(float)));
    if (!rc) rc = hip_fail(hipMalloc(reinterpret_cast<void **>(&h->d_ax), n1 * dim * sizeof(float)));
    if (!rc) rc = hip_fail(hipMalloc(reinterpret_cast<void **>(&h->d_wp), static_cast<size_t>(dim) * h->cp * sizeof(float)));
    if (!rc) rc = hip_fail(hipMemset(h->d_wp, 0, static_cast<size_t>(dim) * h->cp * sizeof(float)));
    for (int i = 0; i < 3 && !rc; ++i) rc = hip_fail(hipEventCreate(&h->ev[i]));
    if (!rc && rocblas_create_handle(&h->blas) != rocblas_status_success) rc = FLEX_ERR_UNSUPPORTED;
    if (!rc) {
        hipDeviceProp_t prop;
        rc = hip_fail(hipGetDeviceProperties(&prop, device));
        h->n_cus = prop.multiProcessorCount;
        h->use_blas = want_blas || dim % 4 != 0 || dim > 256 || h->n < 32;
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (rc) {
        flex_axw_destroy(h);
        return rc;
    }
    *out = h;
    return FLEX_OK;
}

int flex_axw_run(flex_axw *h, int order, const float *dX, const float *dW, float *dOut, flex_stream_t stream,
                 float *gemm_ms, float *spmm_ms) {
    if (!h || !dX || !dW || !dOut || order < FLEX_AXW_AUTO || order > FLEX_AXW_AX_W) return FLEX_ERR_INVALID;
    if (h->n == 0) return FLEX_OK;
    if (order == FLEX_AXW_AUTO) order = h->cp <= h->dim ? FLEX_AXW_A_XW : FLEX_AXW_AX_W;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int cur = -1;
    (void)hipGetDevice(&cur);
    int rc = hip_fail(hipSetDevice(h->device));
    const bool timed = gemm_ms || spmm_ms;
    auto mark = [&](int i) { return timed ? hip_fail(hipEventRecord(h->ev[i], s)) : FLEX_OK; };
    if (!rc && rocblas_set_stream(h->blas, s) != rocblas_status_success) rc = FLEX_ERR_UNSUPPORTED;
    // W -> Wp: dim rows of c floats into rows of cp floats (the zero padding was written once at create)
    if (!rc)
        rc = hip_fail(hipMemcpy2DAsync(h->d_wp, static_cast<size_t>(h->cp) * sizeof(float), dW, static_cast<size_t>(h->c) * sizeof(float),
                                       static_cast<size_t>(h->c) * sizeof(float), static_cast<size_t>(h->dim), hipMemcpyDeviceToDevice, s));
    if (!rc) rc = mark(0);
    if (order == FLEX_AXW_A_XW) {  // run1: cusp.cu:18-75
        if (!rc) rc = gemm_rm(h, dX, h->d_xw, s);
        if (!rc) rc = mark(1);
        if (!rc) rc = flex_spmm(h->plan_c, h->d_xw, dOut, stream);
        if (!rc) rc = mark(2);
    } else {  // run2: cusp.cu:121-178
        if (!rc) rc = flex_spmm(h->plan_dim, dX, h->d_ax, stream);
        if (!rc) rc = mark(1);
        if (!rc) rc = gemm_rm(h, h->d_ax, dOut, s);
        if (!rc) rc = mark(2);
    }
    if (!rc && timed) {
        rc = hip_fail(hipEventSynchronize(h->ev[2]));
        float a = 0.f, b = 0.f;
        if (!rc) rc = hip_fail(hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
        if (!rc) rc = hip_fail(hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
        const bool gemm_first = order == FLEX_AXW_A_XW;
        if (gemm_ms) *gemm_ms = gemm_first ? a : b;
        if (spmm_ms) *spmm_ms = gemm_first ? b : a;
    }
    if (cur >= 0 && cur != h->device) (void)hipSetDevice(cur);
    return rc;
}

}  // extern "C"
