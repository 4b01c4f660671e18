// vendor.cpp -- libflex_vendor.so: hipSPARSE SpMM behind a C ABI (include/flex_vendor.h).
// ≙ cuSpmm (flex.cu:5717-5804) with cuSPARSE replaced by hipSPARSE; same descriptors, same
// algorithm enum, row-major dense operands, alpha = 1, beta = 0.
#include <hip/hip_runtime.h>
#include <hipsparse/hipsparse.h>

#include <new>

#include "../../include/flex_vendor.h"

struct flex_vendor {
    hipsparseHandle_t handle = nullptr;
    hipsparseSpMatDescr_t A = nullptr;
    hipsparseDnMatDescr_t B = nullptr, C = nullptr;
    void *buffer = nullptr;
    float alpha = 1.0f, beta = 0.0f;
    hipsparseSpMMAlg_t alg = HIPSPARSE_SPMM_CSR_ALG3;
};

static thread_local int g_status = 0;

#define VENDOR_TRY(expr)                       \
    do {                                       \
        hipsparseStatus_t st_ = (expr);        \
        if (st_ != HIPSPARSE_STATUS_SUCCESS) { \
            g_status = static_cast<int>(st_);  \
            flex_vendor_spmm_destroy(h);       \
            return -8;                         \
        }                                      \
    } while (0)

extern "C" {

int flex_vendor_last_status(void) { return g_status; }

int flex_vendor_spmm_destroy(flex_vendor *h) {
    if (!h) return 0;
    if (h->A) hipsparseDestroySpMat(h->A);
    if (h->B) hipsparseDestroyDnMat(h->B);
    if (h->C) hipsparseDestroyDnMat(h->C);
    if (h->handle) hipsparseDestroy(h->handle);
    if (h->buffer) (void)hipFree(h->buffer);
    delete h;
    return 0;
}

int flex_vendor_spmm_create(flex_vendor **out, int32_t m, int32_t n, int64_t nnz, const uint32_t *d_rowPtr,
                            const uint32_t *d_col, const float *d_vals, int k, const float *dB, float *dC) {
    return flex_vendor_spmm_create_alg(out, m, n, nnz, d_rowPtr, d_col, d_vals, k, dB, dC, 3);
}

int flex_vendor_spmm_create_alg(flex_vendor **out, int32_t m, int32_t n, int64_t nnz, const uint32_t *d_rowPtr,
                                const uint32_t *d_col, const float *d_vals, int k, const float *dB, float *dC, int alg) {
    if (alg < 0 || alg > 3) return -1;
    if (!out || m < 0 || n < 0 || nnz < 0 || k <= 0 || !d_rowPtr || !dC) return -1;
    *out = nullptr;
    flex_vendor *h = new (std::nothrow) flex_vendor();
    if (!h) return -2;
    h->alg = alg == 0 ? HIPSPARSE_SPMM_ALG_DEFAULT : alg == 1 ? HIPSPARSE_SPMM_CSR_ALG1 : alg == 2 ? HIPSPARSE_SPMM_CSR_ALG2 : HIPSPARSE_SPMM_CSR_ALG3;
    VENDOR_TRY(hipsparseCreate(&h->handle));
    VENDOR_TRY(hipsparseCreateCsr(&h->A, m, n, nnz, const_cast<uint32_t *>(d_rowPtr), const_cast<uint32_t *>(d_col),
                                  const_cast<float *>(d_vals), HIPSPARSE_INDEX_32I, HIPSPARSE_INDEX_32I,
                                  HIPSPARSE_INDEX_BASE_ZERO, HIP_R_32F));
    VENDOR_TRY(hipsparseCreateDnMat(&h->B, n, k, k, const_cast<float *>(dB), HIP_R_32F, HIPSPARSE_ORDER_ROW));
    VENDOR_TRY(hipsparseCreateDnMat(&h->C, m, k, k, dC, HIP_R_32F, HIPSPARSE_ORDER_ROW));
    size_t bytes = 0;
    VENDOR_TRY(hipsparseSpMM_bufferSize(h->handle, HIPSPARSE_OPERATION_NON_TRANSPOSE, HIPSPARSE_OPERATION_NON_TRANSPOSE,
                                        &h->alpha, h->A, h->B, &h->beta, h->C, HIP_R_32F, h->alg,
                                        &bytes));
    if (hipMalloc(&h->buffer, bytes ? bytes : 4) != hipSuccess) {
        flex_vendor_spmm_destroy(h);
        return -3;
    }
    *out = h;
    return 0;
}

int flex_vendor_spmm_run(flex_vendor *h, flex_vendor_stream_t stream) {
    if (!h) return -1;
    hipsparseStatus_t st = hipsparseSetStream(h->handle, reinterpret_cast<hipStream_t>(stream));
    if (st == HIPSPARSE_STATUS_SUCCESS)
        st = hipsparseSpMM(h->handle, HIPSPARSE_OPERATION_NON_TRANSPOSE, HIPSPARSE_OPERATION_NON_TRANSPOSE, &h->alpha,
                           h->A, h->B, &h->beta, h->C, HIP_R_32F, h->alg, h->buffer);
    if (st != HIPSPARSE_STATUS_SUCCESS) {
        g_status = static_cast<int>(st);
        return -8;
    }
    return 0;
}

}  // extern "C"
