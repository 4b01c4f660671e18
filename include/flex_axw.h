/*
 * flex_axw.h -- C ABI of libflex_axw.so: the GCN layer product  Out = A * X * W  either side of the
 * SpMM.
 *
 * ≙ run1 / run2 of cusp.cu (3-104, 106-208), the AXW block of main.cu:22-77 (compiled out in the
 * reference: `//#define AXW 1`):  run1 = A*(X*W): SGEMM then SpMM at k = c;  run2 = (A*X)*W: SpMM at
 * k = dim then SGEMM.  Here the SpMM is the engine's (flex_spmm, a plan per width) instead of
 * cusparseSpMM, the dense product is a hand-written fp32 MFMA kernel for dim <= 128 (axw_kernels.hip;
 * rocBLAS SGEMM for other widths), and all dense operands are ROW-major (the reference's are
 * column-major, cusp.cu:31-32, 55-60) so that they chain with flex_spmm without a transpose.
 * Kept in its own library: the engine (libflex_spmm.so) never depends on rocBLAS.
 */
#ifndef FLEX_AXW_H
#define FLEX_AXW_H
#include "flex_spmm.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct flex_axw flex_axw;

#define FLEX_AXW_A_XW 1 /* run1: B = X*W (n x c), Out = A*B : cheapest when c < dim */
#define FLEX_AXW_AX_W 2 /* run2: B = A*X (n x dim), Out = B*W */
#define FLEX_AXW_AUTO 0 /* the order with fewer SpMM columns (SpMM dominates both) */
#define FLEX_AXW_USE_BLAS 0x10000u /* flex_axw_create flag: rocBLAS SGEMM for the dense half instead of the hand-written MFMA kernel (tools/probe_axw.py) */

/* Leading dimension of Out and of the X*W intermediate: c rounded up to a multiple of 32 floats, so
 * that every row is a whole number of 128-byte cache lines (a row that starts mid-line costs each
 * gather one extra line: k=100 runs 50 % slower than k=128 on the reddit shape); the extra columns
 * come out as zeros. */
int flex_axw_ld(int c);

/* Plans A once per SpMM width (k = flex_axw_ld(c) and k = dim) and allocates the intermediates
 * (n x flex_axw_ld(c) and n x dim floats) and the padded copy of W on `device`.  `flags` as for
 * flex_plan_create (row schedule).  A must be square (a graph). */
int flex_axw_create(flex_axw **out, const flex_csr *hostA, int dim, int c, int device, unsigned flags);

/* Out[n x flex_axw_ld(c)] = A * dX[n x dim] * dW[dim x c], all device, row-major, fp32, on `stream`.
 * order: FLEX_AXW_*.  gemm_ms / spmm_ms (or NULL): device time of the two stages of THIS call; asking
 * for them makes the call synchronise.  ≙ Metrics.gemm_t / spmm_t (common.h:14-36). */
int flex_axw_run(flex_axw *h, int order, const float *dX, const float *dW, float *dOut, flex_stream_t stream,
                 float *gemm_ms, float *spmm_ms);
int flex_axw_destroy(flex_axw *h);
/* last rocBLAS status seen (rocblas_status), for FLEX_ERR_UNSUPPORTED returns caused by rocBLAS */
int flex_axw_last_blas_status(void);

#ifdef __cplusplus
}
#endif
#endif
