// shard.cpp -- contiguous, cost-balanced row ranges for row-sharded multi-GPU SpMM.
//
// New work: the reference is single-GPU (one cudaSetDevice, flex.cu:4137).  Rows of
// A (and of C) are independent units, B is replicated, so shards need no reduction.
#include "internal.h"

extern "C" int flex_shard_rows(const flex_csr *A, int k, int nparts, int64_t *row_bounds) {
    if (!row_bounds || nparts <= 0 || k <= 0) return FLEX_ERR_INVALID;
    const int rc = flex::validate_csr(A);
    if (rc) return rc;
    // cost(row) = nnz*(4k+8) + 4k bytes: gathered B bytes + (col,val) records + the C row
    const double per_nz = 4.0 * k + 8.0, per_row = 4.0 * k;
    const double total = per_nz * static_cast<double>(A->nnz) + per_row * static_cast<double>(A->m);
    row_bounds[0] = 0;
    int64_t r = 0;
    for (int p = 1; p < nparts; ++p) {
        const double want = total * p / nparts;
        // first r whose prefix cost reaches `want`; prefix(r) = per_nz*rowPtr[r] + per_row*r is monotone
        int64_t lo = r, hi = A->m;
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            const double c = per_nz * static_cast<double>(A->rowPtr[mid]) + per_row * static_cast<double>(mid);
            if (c < want) lo = mid + 1; else hi = mid;
        }
        r = lo;
        row_bounds[p] = r;
    }
    row_bounds[nparts] = A->m;
    return FLEX_OK;
}
