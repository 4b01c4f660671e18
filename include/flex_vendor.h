/*
 * flex_vendor.h -- C ABI of libflex_vendor.so: the vendor SpMM used as side-by-side
 * baseline and sanity check.
 *
 * ≙ cuSpmm(DataLoader&, Perfs&) (flex.cu:5717-5804): cusparseSpMM, CSR, 32-bit indices,
 * row-major B and C, alpha=1, beta=0, algorithm CSR_ALG3 -> hipsparseSpMM with
 * HIPSPARSE_SPMM_CSR_ALG3 / HIPSPARSE_ORDER_ROW.  Kept in its own library so that the
 * engine (libflex_spmm.so) never depends on hipSPARSE.
 */
#ifndef FLEX_VENDOR_H
#define FLEX_VENDOR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct flex_vendor flex_vendor;
typedef struct ihipStream_t *flex_vendor_stream_t;

/* All pointers are DEVICE pointers that must outlive the handle (the reference passes
 * DataLoader's rowPtr_dev/col_dev/vals_dev/gpuX/gpuC, flex.cu:5740-5750). Returns 0 or a
 * negative code (-3: HIP error, -8: hipSPARSE status, see flex_vendor_last_status). */
int flex_vendor_spmm_create(flex_vendor **out, int32_t m, int32_t n, int64_t nnz, const uint32_t *d_rowPtr,
                            const uint32_t *d_col, const float *d_vals, int k, const float *dB, float *dC);
/* Same with the algorithm named: 0 = HIPSPARSE_SPMM_ALG_DEFAULT, 1..3 = HIPSPARSE_SPMM_CSR_ALG1..3 (3 is what the
 * reference's protocol uses; the others are for a "best of vendor" side-by-side). */
int flex_vendor_spmm_create_alg(flex_vendor **out, int32_t m, int32_t n, int64_t nnz, const uint32_t *d_rowPtr,
                                const uint32_t *d_col, const float *d_vals, int k, const float *dB, float *dC, int alg);
/* one hipsparseSpMM on `stream` (the reference times 5 warm-up + 10 of these, flex.cu:5766-5789) */
int flex_vendor_spmm_run(flex_vendor *h, flex_vendor_stream_t stream);
int flex_vendor_spmm_destroy(flex_vendor *h);
int flex_vendor_last_status(void);

#ifdef __cplusplus
}
#endif
#endif
