// mg.cpp -- libflex_mg.so: row-sharded SpMM over the GPUs of one node in a single process
// (include/flex_mg.h).  Uses the engine's C ABI for everything per-GPU and RCCL for the one
// broadcast of B.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <new>
#include <vector>

#include "../../include/flex_mg.h"

struct flex_mg {
    int ngpus = 0, k = 0;
    int32_t m = 0, n = 0;
    std::vector<int> dev;
    std::vector<int64_t> bounds, shard_nnz;
    std::vector<int32_t> vo_mp;  // empty when the order is natural
    std::vector<flex_plan *> plan;
    std::vector<float *> dB, dC;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> e0, e1;
    std::vector<ncclComm_t> comm;
};

static thread_local int g_rccl = 0;

#define MG_HIP(expr)                     \
    do {                                 \
        if ((expr) != hipSuccess) {      \
            flex_mg_destroy(h);          \
            return FLEX_ERR_HIP;         \
        }                                \
    } while (0)
#define MG_FLEX(expr)                    \
    do {                                 \
        int rc_ = (expr);                \
        if (rc_ != FLEX_OK) {            \
            flex_mg_destroy(h);          \
            return rc_;                  \
        }                                \
    } while (0)

extern "C" {

int flex_mg_last_rccl(void) { return g_rccl; }

int flex_mg_destroy(flex_mg *h) {
    if (!h) return FLEX_OK;
    for (int i = 0; i < h->ngpus; ++i) {
        (void)hipSetDevice(h->dev[i]);
        if (i < static_cast<int>(h->comm.size()) && h->comm[i]) ncclCommDestroy(h->comm[i]);
        if (i < static_cast<int>(h->plan.size())) flex_plan_destroy(h->plan[i]);
        if (i < static_cast<int>(h->dB.size())) (void)hipFree(h->dB[i]);
        if (i < static_cast<int>(h->dC.size())) (void)hipFree(h->dC[i]);
        if (i < static_cast<int>(h->e0.size()) && h->e0[i]) (void)hipEventDestroy(h->e0[i]);
        if (i < static_cast<int>(h->e1.size()) && h->e1[i]) (void)hipEventDestroy(h->e1[i]);
        if (i < static_cast<int>(h->stream.size()) && h->stream[i]) (void)hipStreamDestroy(h->stream[i]);
    }
    delete h;
    return FLEX_OK;
}

int flex_mg_create(flex_mg **out, const flex_csr *A, int k, int ngpus, const int *devices, unsigned order) {
    if (!out || !A || k <= 0 || ngpus <= 0) return FLEX_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return FLEX_ERR_HIP;
    flex_mg *h = new (std::nothrow) flex_mg();
    if (!h) return FLEX_ERR_NOMEM;
    h->ngpus = ngpus;
    h->k = k;
    h->m = A->m;
    h->n = A->n;
    for (int i = 0; i < ngpus; ++i) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= ndev) {
            delete h;
            return FLEX_ERR_INVALID;
        }
        h->dev.push_back(d);
    }
    // re-order once on the host, then shard: each shard's columns form a set of communities / a band
    flex_csr Ap = *A;
    std::vector<uint32_t> rp2, col2;
    std::vector<float> val2;
    if ((order & FLEX_ORDER_MASK) != FLEX_ORDER_NATURAL) {
        std::vector<uint32_t> rank(static_cast<size_t>(A->m));
        switch (order & FLEX_ORDER_MASK) {
            case FLEX_ORDER_RCM: MG_FLEX(flex_order_rcm(A, rank.data())); break;
            case FLEX_ORDER_CLUSTER: MG_FLEX(flex_order_cluster(A, rank.data())); break;
            case FLEX_ORDER_GORDER: MG_FLEX(flex_order_gorder(A, 3, rank.data())); break;
            default: delete h; return FLEX_ERR_INVALID;
        }
        h->vo_mp.resize(A->m);
        rp2.resize(static_cast<size_t>(A->m) + 1);
        col2.resize(static_cast<size_t>(A->nnz));
        val2.resize(static_cast<size_t>(A->nnz));
        MG_FLEX(flex_perm_csr(A, rank.data(), h->vo_mp.data(), rp2.data(), col2.data(), val2.data()));
        Ap.rowPtr = rp2.data();
        Ap.col = col2.data();
        Ap.vals = val2.data();
    }
    h->bounds.resize(ngpus + 1);
    MG_FLEX(flex_shard_rows(&Ap, k, ngpus, h->bounds.data()));
    h->plan.assign(ngpus, nullptr);
    h->dB.assign(ngpus, nullptr);
    h->dC.assign(ngpus, nullptr);
    h->stream.assign(ngpus, nullptr);
    h->e0.assign(ngpus, nullptr);
    h->e1.assign(ngpus, nullptr);
    h->shard_nnz.assign(ngpus, 0);
    for (int i = 0; i < ngpus; ++i) {
        MG_HIP(hipSetDevice(h->dev[i]));
        const int64_t r0 = h->bounds[i], r1 = h->bounds[i + 1];
        h->shard_nnz[i] = static_cast<int64_t>(Ap.rowPtr[r1]) - Ap.rowPtr[r0];
        MG_FLEX(flex_plan_create_rows(&h->plan[i], &Ap, r0, r1, h->vo_mp.empty() ? nullptr : h->vo_mp.data(), k,
                                      h->dev[i], FLEX_ORDER_NATURAL));
        MG_HIP(hipMalloc(reinterpret_cast<void **>(&h->dB[i]), sizeof(float) * std::max<size_t>(1, static_cast<size_t>(A->n) * k)));
        MG_HIP(hipMalloc(reinterpret_cast<void **>(&h->dC[i]), sizeof(float) * std::max<size_t>(1, static_cast<size_t>(r1 - r0) * k)));
        MG_HIP(hipStreamCreate(&h->stream[i]));
        MG_HIP(hipEventCreate(&h->e0[i]));
        MG_HIP(hipEventCreate(&h->e1[i]));
    }
    h->comm.assign(ngpus, nullptr);
    const ncclResult_t nr = ncclCommInitAll(h->comm.data(), ngpus, h->dev.data());
    if (nr != ncclSuccess) {
        g_rccl = static_cast<int>(nr);
        flex_mg_destroy(h);
        return FLEX_ERR_HIP;
    }
    *out = h;
    return FLEX_OK;
}

int flex_mg_set_B(flex_mg *h, const float *hostB, double *bcast_ms) {
    if (!h || !hostB) return FLEX_ERR_INVALID;
    const size_t count = static_cast<size_t>(h->n) * h->k;
    if (hipSetDevice(h->dev[0]) != hipSuccess) return FLEX_ERR_HIP;
    if (hipMemcpy(h->dB[0], hostB, count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return FLEX_ERR_HIP;
    const auto t0 = std::chrono::steady_clock::now();
    if (h->ngpus > 1) {
        ncclResult_t nr = ncclGroupStart();
        for (int i = 0; i < h->ngpus && nr == ncclSuccess; ++i)
            nr = ncclBroadcast(h->dB[i], h->dB[i], count, ncclFloat, /*root=*/0, h->comm[i], h->stream[i]);
        if (nr == ncclSuccess) nr = ncclGroupEnd();
        if (nr != ncclSuccess) {
            g_rccl = static_cast<int>(nr);
            return FLEX_ERR_HIP;
        }
    }
    const int rc = flex_mg_sync(h);
    if (bcast_ms) *bcast_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int flex_mg_spmm(flex_mg *h) {
    if (!h) return FLEX_ERR_INVALID;
    for (int i = 0; i < h->ngpus; ++i) {
        if (hipSetDevice(h->dev[i]) != hipSuccess) return FLEX_ERR_HIP;
        const int rc = flex_spmm(h->plan[i], h->dB[i], h->dC[i], reinterpret_cast<flex_stream_t>(h->stream[i]));
        if (rc) return rc;
    }
    return FLEX_OK;
}

int flex_mg_sync(flex_mg *h) {
    if (!h) return FLEX_ERR_INVALID;
    for (int i = 0; i < h->ngpus; ++i) {
        if (hipSetDevice(h->dev[i]) != hipSuccess) return FLEX_ERR_HIP;
        if (hipStreamSynchronize(h->stream[i]) != hipSuccess) return FLEX_ERR_HIP;
    }
    return FLEX_OK;
}

int flex_mg_time(flex_mg *h, int warmup, int reps, double *us_per_step) {
    if (!h || reps <= 0 || !us_per_step) return FLEX_ERR_INVALID;
    int rc;
    for (int w = 0; w < warmup; ++w)
        if ((rc = flex_mg_spmm(h))) return rc;
    if ((rc = flex_mg_sync(h))) return rc;
    for (int i = 0; i < h->ngpus; ++i) {
        if (hipSetDevice(h->dev[i]) != hipSuccess || hipEventRecord(h->e0[i], h->stream[i]) != hipSuccess) return FLEX_ERR_HIP;
    }
    for (int r = 0; r < reps; ++r)
        if ((rc = flex_mg_spmm(h))) return rc;
    for (int i = 0; i < h->ngpus; ++i) {
        if (hipSetDevice(h->dev[i]) != hipSuccess || hipEventRecord(h->e1[i], h->stream[i]) != hipSuccess) return FLEX_ERR_HIP;
    }
    if ((rc = flex_mg_sync(h))) return rc;
    double worst = 0;
    for (int i = 0; i < h->ngpus; ++i) {
        float ms = 0;
        if (hipSetDevice(h->dev[i]) != hipSuccess || hipEventElapsedTime(&ms, h->e0[i], h->e1[i]) != hipSuccess) return FLEX_ERR_HIP;
        worst = std::max<double>(worst, ms);
    }
    *us_per_step = worst * 1e3 / reps;
    return FLEX_OK;
}

int flex_mg_get_C(flex_mg *h, float *hostC) {
    if (!h || !hostC) return FLEX_ERR_INVALID;
    int rc = flex_mg_sync(h);
    if (rc) return rc;
    std::vector<float> tmp;
    for (int i = 0; i < h->ngpus; ++i) {
        const int64_t r0 = h->bounds[i], r1 = h->bounds[i + 1];
        if (hipSetDevice(h->dev[i]) != hipSuccess) return FLEX_ERR_HIP;
        if (h->vo_mp.empty()) {
            if (hipMemcpy(hostC + r0 * h->k, h->dC[i], sizeof(float) * static_cast<size_t>(r1 - r0) * h->k, hipMemcpyDeviceToHost) != hipSuccess)
                return FLEX_ERR_HIP;
        } else {  // shard rows are in the re-ordered numbering: scatter them back
            tmp.resize(static_cast<size_t>(r1 - r0) * h->k);
            if (hipMemcpy(tmp.data(), h->dC[i], sizeof(float) * tmp.size(), hipMemcpyDeviceToHost) != hipSuccess) return FLEX_ERR_HIP;
            for (int64_t r = r0; r < r1; ++r)
                std::copy(tmp.begin() + (r - r0) * h->k, tmp.begin() + (r - r0 + 1) * h->k,
                          hostC + static_cast<size_t>(h->vo_mp[r]) * h->k);
        }
    }
    return FLEX_OK;
}

int flex_mg_shard_info(const flex_mg *h, int64_t *row_bounds, int64_t *shard_nnz) {
    if (!h) return FLEX_ERR_INVALID;
    if (row_bounds) std::copy(h->bounds.begin(), h->bounds.end(), row_bounds);
    if (shard_nnz) std::copy(h->shard_nnz.begin(), h->shard_nnz.end(), shard_nnz);
    return FLEX_OK;
}

}  // extern "C"
