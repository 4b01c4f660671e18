// DataLoader.h (host mirror) -- same class surface as the reference's DataLoader
// (DataLoader.cuh:21-146) so that main.cu's body compiles against it unchanged: constructor
// (path, k), public rowPtr/col/vals/vo_mp/cpuX/h_ref_c, device pointers rowPtr_dev/col_dev/
// vals_dev/gpuX/gpuC, sizes and graph statistics, c_cuSpmm_run, gpuC_zero, and the reordered
// loaders.  Device memory is HIP; parsing, statistics and orderings go through the engine's
// C ABI (include/flex_spmm.h).  A path "synth:<name>[*scale]" builds the stand-in graph of
// that name instead of reading a file (the reference's data files other than pubmed.csv are absent).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "common.h"

struct flex_axw;  // include/flex_axw.h

class DataLoader {
   public:
    DataLoader(const std::string &st, const int di);  // DataLoader.cu:9-124
    DataLoader(const DataLoader &dl);                 // DataLoader.cu:230-241 (sizes only; arrays filled by subclasses)
    ~DataLoader() { freeAll(); }
    DataLoader &operator=(const DataLoader &) = delete;

    void cuda_alloc_cpy();            // DataLoader.cu:166-218 (name kept for source compatibility; allocates with HIP)
    void c_cuSpmm_run(Perfs &perfRes);  // DataLoader.cu:220-227
    void gpuC_zero();                 // DataLoader.cu:229-233
    void perm_apply(const DataLoader &dl);  // DataLoader.cu:244-321: fill this loader from dl through vo_mp (+ checksum self-test)
    void print_data();
    void getDegDist();
    void axw_alloc();  // W, the two result buffers and the A*X*W plans (host/cusp.cpp); only the --axw path calls it
    bool compare();    // DataLoader.cu:859-869: cpuRef1 (A(XW)) against cpuRef2 ((AX)W)

    const DataLoader *const dl_original;
    std::vector<unsigned int> rowPtr, col;
    std::vector<float> vals;
    std::vector<int> vo_mp;  // vo_mp[new] = old

    std::vector<float> cpuX;     // n * dim
    std::vector<float> h_ref_c;  // vendor (hipSPARSE) result, the gold of resCheck
    std::vector<float> cpuW;     // dim * c                                   (DataLoader.cuh:46)
    std::vector<float> cpuRef1, cpuRef2;  // n * flex_axw_ld(c): A(XW) and (AX)W  (DataLoader.cuh:47-48)
    float *gpuW = nullptr, *gpuRef1 = nullptr, *gpuRef2 = nullptr;
    flex_axw *axw = nullptr;

    std::string vertex_order_abbr;
    unsigned int *rowPtr_dev = nullptr, *col_dev = nullptr;
    float *vals_dev = nullptr;

    int64_t gpuX_bytes = 0, C_elts = 0, gpuC_bytes = 0;
    int64_t uni_nb = 0;
    float *gpuX = nullptr, *gpuC = nullptr;

    bool is_directed = false;
    int n_nodes_z_out = 0, n_nodes_z_in = 0, n_nodes_z_deg = 0;
    size_t n_edges_one_way = 0, n_edges_asymmetric = 0;
    size_t m = 0, n = 0, dim = 0, c = 0, nnz = 0;
    std::string graph_name;

    flex_csr csr_view() const {
        return flex_csr{static_cast<int32_t>(m), static_cast<int32_t>(n), static_cast<int64_t>(nnz), rowPtr.data(),
                        col.data(), vals.data()};
    }
    void freeA();
    void freeAll();

   protected:
    // shared body of the reordered loaders: rank[old] = new -> vo_mp, permuted CSR, device copy
    void adopt_rank(const DataLoader &dl, const std::vector<uint32_t> &rank, const char *abbr);
};

class DataLoaderRcm : public DataLoader {  // DataLoader.cu:723-787
   public:
    explicit DataLoaderRcm(const DataLoader &dl);
};
class DataLoaderDeg : public DataLoader {  // DataLoader.cu:657-721
   public:
    explicit DataLoaderDeg(const DataLoader &dl);
};
class DataLoaderDFS : public DataLoader {  // DataLoader.cu:324-451
   public:
    explicit DataLoaderDFS(const DataLoader &dl);
};
class DataLoaderGorder : public DataLoader {  // DataLoader.cu:789-857 (window 3)
   public:
    explicit DataLoaderGorder(const DataLoader &dl);
};
class DataLoaderRabbit : public DataLoader {  // DataLoader.cu:453-655 (flex_order_rabbit: the reference's merges and dendrogram walk)
   public:
    explicit DataLoaderRabbit(const DataLoader &dl);
};
