// DataLoader.cpp (host mirror) -- see DataLoader.h.  Behaviour follows DataLoader.cu of the
// reference (line numbers cited per function); the code is new and sits on the engine's C ABI.
#include "DataLoader.h"
#include "lazy_lib.h"

#include <cassert>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <numeric>

#include "../../../include/flex_axw.h"
#include "flex.h"

namespace {

// "synth:flickr" or "synth:flickr*8"
bool parse_synth(const std::string &path, std::string &name, int &scale) {
    if (path.rfind("synth:", 0) != 0) return false;
    name = path.substr(6);
    scale = 1;
    if (auto star = name.find('*'); star != std::string::npos) {
        scale = std::max(1, std::atoi(name.c_str() + star + 1));
        name = name.substr(0, star);
    }
    return true;
}

}  // namespace

DataLoader::DataLoader(const std::string &data_path, const int di) : dl_original(this), dim(di) {
    vertex_order_abbr = "OVO";  // Original Vertex Order
    flex_host_csr h{};
    std::string sname;
    int scale = 1;
    if (parse_synth(data_path, sname, scale)) {
        flex_synth_params sp{};
        FLEX_CHECK(flex_synth_preset(sname.c_str(), scale, &sp));
        FLEX_CHECK(flex_synth_graph(&sp, &h));
        graph_name = sname;
    } else {
        const std::string data_name = data_path.substr(data_path.find_last_of("/") + 1);
        graph_name = data_name.substr(0, data_name.find("."));
        const std::string ext = data_name.find('.') == std::string::npos ? "" : data_name.substr(data_name.find_last_of('.'));
        if (ext == ".mtx")  // MatrixMarket directly (the reference converts first: data/SuiteSparse/mtx2csr.cc)
            FLEX_CHECK(flex_mtx_load(data_path.c_str(), /*sort_columns=*/1, &h));
        else if (ext == ".bin")  // binary CSR cache written by flex_csr_save_bin
            FLEX_CHECK(flex_csr_load_bin(data_path.c_str(), &h));
        else
            FLEX_CHECK(flex_csv_load(data_path.c_str(), &h));  // amazon.csv draws its values from rand() here
        if (h.m != h.n) {
            flex_host_csr_free(&h);
            throw std::runtime_error("DataLoader needs a square (graph) matrix");
        }
    }
    rowPtr.assign(h.rowPtr, h.rowPtr + h.m + 1);
    col.assign(h.col, h.col + h.nnz);
    vals.assign(h.vals, h.vals + h.nnz);
    if (run_options().debug_values) std::fill(vals.begin(), vals.end(), 1.0f);  // opt_debug, DataLoader.cu:51
    m = n = static_cast<size_t>(h.m);
    nnz = static_cast<size_t>(h.nnz);
    c = static_cast<size_t>(h.c);
    uni_nb = h.uni_nb;
    is_directed = h.is_directed != 0;
    n_edges_one_way = static_cast<size_t>(h.n_edges_one_way);
    n_edges_asymmetric = static_cast<size_t>(h.n_edges_asymmetric);
    n_nodes_z_out = h.n_nodes_z_out;
    n_nodes_z_in = h.n_nodes_z_in;
    n_nodes_z_deg = h.n_nodes_z_deg;
    flex_host_csr_free(&h);
    vo_mp.resize(m);
    std::iota(vo_mp.begin(), vo_mp.end(), 0);
    cuda_alloc_cpy();
}

DataLoader::DataLoader(const DataLoader &dl)
    : dl_original(&dl), uni_nb(dl.uni_nb), m(dl.m), n(dl.n), dim(dl.dim), c(dl.c), nnz(dl.nnz), graph_name(dl.graph_name) {}

void DataLoader::cuda_alloc_cpy() {
    HIP_CHECK(hipMalloc(&rowPtr_dev, sizeof(unsigned int) * (m + 1)));
    HIP_CHECK(hipMemcpy(rowPtr_dev, rowPtr.data(), sizeof(unsigned int) * (m + 1), hipMemcpyHostToDevice));
    HIP_CHECK(hipMalloc(&col_dev, sizeof(unsigned int) * std::max<size_t>(nnz, 1)));
    HIP_CHECK(hipMemcpy(col_dev, col.data(), sizeof(unsigned int) * nnz, hipMemcpyHostToDevice));
    HIP_CHECK(hipMalloc(&vals_dev, sizeof(float) * std::max<size_t>(nnz, 1)));
    HIP_CHECK(hipMemcpy(vals_dev, vals.data(), sizeof(float) * nnz, hipMemcpyHostToDevice));

    C_elts = static_cast<int64_t>(m * dim);
    gpuC_bytes = C_elts * static_cast<int64_t>(sizeof(float));
    HIP_CHECK(hipMalloc(&gpuC, std::max<int64_t>(gpuC_bytes, 4)));
    HIP_CHECK(hipMemset(gpuC, 0, gpuC_bytes));
    gpuX_bytes = static_cast<int64_t>(n * dim * sizeof(float));
    if (vertex_order_abbr == "OVO") {  // B is drawn once, for the original order only (DataLoader.cu:198-217)
        cpuX.resize(n * dim);
        if (run_options().debug_values) {  // opt_debug, DataLoader.cu:202-203
            for (size_t i = 0; i < n; ++i) std::fill(cpuX.begin() + i * dim, cpuX.begin() + (i + 1) * dim, static_cast<float>(i));
        } else {
            FLEX_CHECK(flex_fill_dense_rand(cpuX.data(), static_cast<int64_t>(n), static_cast<int>(dim)));
        }
        HIP_CHECK(hipMalloc(&gpuX, std::max<int64_t>(gpuX_bytes, 4)));
        HIP_CHECK(hipMemcpy(gpuX, cpuX.data(), gpuX_bytes, hipMemcpyHostToDevice));
    }
}

void DataLoader::c_cuSpmm_run(Perfs &perfRes) {
    cuSpmm(*this, perfRes);
    h_ref_c.resize(C_elts);
    HIP_CHECK(hipMemcpy(h_ref_c.data(), gpuC, gpuC_bytes, hipMemcpyDeviceToHost));
}

void DataLoader::gpuC_zero() { HIP_CHECK(hipMemset(gpuC, 0, gpuC_bytes)); }

void DataLoader::freeA() {
    hip_freez(rowPtr_dev);
    hip_freez(col_dev);
    hip_freez(vals_dev);
}

void DataLoader::freeAll() {
    try {
        freeA();
        if (vertex_order_abbr == "OVO") hip_freez(gpuX);  // reordered loaders alias the original's B (DataLoader.cuh:104)
        hip_freez(gpuC);
        hip_freez(gpuW);
        hip_freez(gpuRef1);
        hip_freez(gpuRef2);
        if (axw) FLEX_AXW(flex_axw_destroy)(axw);
        axw = nullptr;
    } catch (...) {
    }
}

void DataLoader::adopt_rank(const DataLoader &dl, const std::vector<uint32_t> &rank, const char *abbr) {
    gpuX = dl.gpuX;
    vertex_order_abbr = abbr;
    assert(dl.rowPtr.size() == n + 1);
    vo_mp.resize(m);
    rowPtr.resize(m + 1);
    col.resize(nnz);
    vals.resize(nnz);
    const flex_csr a = dl.csr_view();
    FLEX_CHECK(flex_perm_csr(&a, rank.data(), vo_mp.data(), rowPtr.data(), col.data(), vals.data()));
    cuda_alloc_cpy();
}

void DataLoader::perm_apply(const DataLoader &dl) {
    // this->vo_mp[new] = old is already set: derive rank, permute, then the reference's checksum
    // self-test (DataLoader.cu:294-320): per-column sums of (src & 0xf) and of weights survive
    assert(rowPtr.empty());
    std::vector<uint32_t> rank(n);
    for (size_t v_new = 0; v_new < n; ++v_new) rank[vo_mp[v_new]] = static_cast<uint32_t>(v_new);
    std::vector<int> vo(n);
    rowPtr.resize(n + 1);
    col.resize(dl.col.size());
    vals.resize(dl.col.size());
    const flex_csr a = dl.csr_view();
    FLEX_CHECK(flex_perm_csr(&a, rank.data(), vo.data(), rowPtr.data(), col.data(), vals.data()));
    std::vector<int64_t> check_old(n, 0), check_new(n, 0);
    std::vector<double> w_old(n, 0.0), w_new(n, 0.0);
    for (size_t v_old = 0; v_old < n; ++v_old) {
        const int inc = static_cast<int>(v_old & 0xf);
        for (unsigned e = dl.rowPtr[v_old]; e < dl.rowPtr[v_old + 1]; ++e) {
            check_old[dl.col[e]] += inc;
            w_old[dl.col[e]] += dl.vals[e];
        }
        for (unsigned e = rowPtr[rank[v_old]]; e < rowPtr[rank[v_old] + 1]; ++e) {
            check_new[col[e]] += inc;
            w_new[col[e]] += vals[e];
        }
    }
    for (size_t v_old = 0; v_old < n; ++v_old) {
        if (check_old[v_old] != check_new[rank[v_old]]) throw std::runtime_error("perm_apply: edge checksum mismatch");
        // weights are summed in a different order after the column sort: compare with a float-sized slack
        const double d = std::abs(w_old[v_old] - w_new[rank[v_old]]);
        if (d > 1e-6 * (1.0 + std::abs(w_old[v_old]))) throw std::runtime_error("perm_apply: weight checksum mismatch");
    }
}

namespace {

// rank[old] = new from `compute`, or from <perm_cache>/<graph>.<ORD>.perm when that file matches this
// matrix (the reference recomputes every ordering on every run).
template <typename F>
std::vector<uint32_t> cached_rank(const DataLoader &dl, const char *abbr, F compute) {
    std::vector<uint32_t> rank(dl.n);
    const flex_csr a = dl.csr_view();
    const std::string &dir = run_options().perm_cache;
    std::string path;
    uint64_t fp = 0;
    if (!dir.empty()) {
        path = dir + "/" + dl.graph_name + "." + abbr + ".perm";
        fp = flex_csr_fingerprint(&a);
        if (flex_perm_load(path.c_str(), rank.data(), static_cast<int64_t>(dl.n), fp) == FLEX_OK) {
            std::printf("%s order: cached (%s)\n", abbr, path.c_str());
            return rank;
        }
    }
    FLEX_CHECK(compute(&a, rank.data()));
    if (!dir.empty() && flex_perm_save(path.c_str(), rank.data(), static_cast<int64_t>(dl.n), fp) != FLEX_OK)
        std::printf("warning: could not write %s\n", path.c_str());
    return rank;
}

}  // namespace

DataLoaderRcm::DataLoaderRcm(const DataLoader &dl) : DataLoader(dl) {
    adopt_rank(dl, cached_rank(dl, "RCM", [](const flex_csr *a, uint32_t *r) { return flex_order_rcm(a, r); }), "RCM");
}

DataLoaderDeg::DataLoaderDeg(const DataLoader &dl) : DataLoader(dl) {
    adopt_rank(dl, cached_rank(dl, "DEG", [](const flex_csr *a, uint32_t *r) { return flex_order_deg(a, /*descending=*/1, r); }), "DEG");
}

DataLoaderDFS::DataLoaderDFS(const DataLoader &dl) : DataLoader(dl) {
    adopt_rank(dl, cached_rank(dl, "DFS", [](const flex_csr *a, uint32_t *r) { return flex_order_dfs(a, r); }), "DFS");
}

DataLoaderGorder::DataLoaderGorder(const DataLoader &dl) : DataLoader(dl) {
    // window 3 as DataLoader.cu:808
    adopt_rank(dl, cached_rank(dl, "GOR", [](const flex_csr *a, uint32_t *r) { return flex_order_gorder(a, /*window_sz=*/3, r); }), "GOR");
}

DataLoaderRabbit::DataLoaderRabbit(const DataLoader &dl) : DataLoader(dl) {
    // the reference's own algorithm (flex_order_rabbit restates DataLoader.cu:455-655 merge for merge) on the sizes the
    // reference runs it on; its per-vertex weight maps make it impractical past a few 10^7 nonzeros (the reference's too),
    // where the engine's parallel clustering stands in -- said out loud, never silently
    const bool is_dir = dl.is_directed;
    if (dl.nnz <= 50'000'000) {
        adopt_rank(dl, cached_rank(dl, "RBT", [is_dir](const flex_csr *a, uint32_t *r) { return flex_order_rabbit(a, is_dir ? 1 : 0, r); }), "RBT");
    } else {
        std::printf("RBT: %zu nonzeros is beyond the reference's map-based Rabbit; using the engine's parallel clustering (flex_order_cluster)\n", static_cast<size_t>(dl.nnz));
        adopt_rank(dl, cached_rank(dl, "RBC", [](const flex_csr *a, uint32_t *r) { return flex_order_cluster(a, r); }), "RBT");
    }
}

void DataLoader::getDegDist() {  // DataLoader.cu:126-145
    int deg[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i + 1 < rowPtr.size(); ++i) {
        const unsigned nbs = rowPtr[i + 1] - rowPtr[i];
        deg[nbs <= 8 ? 0 : nbs <= 16 ? 1 : nbs <= 32 ? 2 : nbs <= 256 ? 3 : 4]++;
    }
    const char *label[5] = {"( 0, 8]", "( 8, 16]", "( 16, 32]", "( 32, 256]", "( 256, +OO)"};
    for (int i = 0; i < 5; ++i) std::printf("%s: %f\n", label[i], deg[i] * 1.0 / m);
}

void DataLoader::print_data() {  // DataLoader.cu:871-915 (first/last five of each array)
    auto show = [](const char *what, auto first, auto last) {
        std::cout << what;
        for (auto it = first; it != last; ++it) std::cout << *it << " ";
        std::cout << std::endl;
    };
    const size_t k5 = std::min<size_t>(5, nnz), r5 = std::min<size_t>(5, rowPtr.size());
    show("The first 5 elements of rowptr: ", rowPtr.begin(), rowPtr.begin() + r5);
    show("The last 5 elements of rowptr: ", rowPtr.end() - r5, rowPtr.end());
    show("The first 5 elements of indies: ", col.begin(), col.begin() + k5);
    show("The last 5 elements of indies: ", col.end() - k5, col.end());
    show("The first 5 elements of vals: ", vals.begin(), vals.begin() + k5);
    show("The last 5 elements of vals: ", vals.end() - k5, vals.end());
    if (!cpuX.empty()) show("The first 5 elements of X: ", cpuX.begin(), cpuX.begin() + std::min<size_t>(5, cpuX.size()));
}
